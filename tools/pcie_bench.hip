// PCIe copy rates on the GPU box, in the shapes the host-buffer pipeline of vpt_predict_batch uses (diagnostics, not product
// code): pinned host memory, hipMemcpyAsync, one or two streams per direction, whole-batch and chunked sizes.
//   hipcc --offload-arch=gfx950 -O3 -o tools/pcie_bench tools/pcie_bench.hip && ./tools/pcie_bench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e_), #x); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t MB = 1 << 20, big = 32 * MB;
    char *h_in, *h_out, *d_in, *d_out;
    CHECK(hipHostMalloc((void**)&h_in, big, hipHostMallocDefault)); CHECK(hipHostMalloc((void**)&h_out, big, hipHostMallocDefault));
    CHECK(hipMalloc((void**)&d_in, big)); CHECK(hipMalloc((void**)&d_out, big));
    hipStream_t s[4]; for (auto& x : s) CHECK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    auto timeit = [&](const char* what, double bytes, auto fn) {
        fn(); for (auto& x : s) CHECK(hipStreamSynchronize(x));
        const int reps = 10; const double t0 = now();
        for (int r = 0; r < reps; ++r) { fn(); for (auto& x : s) CHECK(hipStreamSynchronize(x)); }
        const double dt = (now() - t0) / reps;
        printf("%-78s %7.3f ms  %6.1f GB/s\n", what, dt * 1e3, bytes / dt / 1e9);
    };
    const size_t IN = 21 * MB, OUT = 31 * MB;   // configs[1]: text + offsets in, scores + labels out
    timeit("H2D 21 MB, one copy", IN, [&] { CHECK(hipMemcpyAsync(d_in, h_in, IN, hipMemcpyHostToDevice, s[0])); });
    timeit("D2H 31 MB, one copy", OUT, [&] { CHECK(hipMemcpyAsync(h_out, d_out, OUT, hipMemcpyDeviceToHost, s[0])); });
    timeit("D2H 31 MB as 7 copies on one stream", OUT, [&] { for (int k = 0; k < 7; ++k) CHECK(hipMemcpyAsync(h_out + k * (OUT / 7), d_out + k * (OUT / 7), OUT / 7, hipMemcpyDeviceToHost, s[0])); });
    timeit("D2H 31 MB as 2 copies on two streams", OUT, [&] { for (int k = 0; k < 2; ++k) CHECK(hipMemcpyAsync(h_out + k * (OUT / 2), d_out + k * (OUT / 2), OUT / 2, hipMemcpyDeviceToHost, s[k])); });
    timeit("D2H 31 MB as 14 copies alternating over two streams", OUT, [&] { for (int k = 0; k < 14; ++k) CHECK(hipMemcpyAsync(h_out + k * (OUT / 14), d_out + k * (OUT / 14), OUT / 14, hipMemcpyDeviceToHost, s[k & 1])); });
    timeit("H2D 21 MB + D2H 31 MB at the same time, one stream each (bytes = both)", IN + OUT, [&] {
        CHECK(hipMemcpyAsync(d_in, h_in, IN, hipMemcpyHostToDevice, s[0])); CHECK(hipMemcpyAsync(h_out, d_out, OUT, hipMemcpyDeviceToHost, s[1])); });
    timeit("the same as 7 + 7 chunk copies", IN + OUT, [&] {
        for (int k = 0; k < 7; ++k) { CHECK(hipMemcpyAsync(d_in + k * (IN / 7), h_in + k * (IN / 7), IN / 7, hipMemcpyHostToDevice, s[0]));
                                      CHECK(hipMemcpyAsync(h_out + k * (OUT / 7), d_out + k * (OUT / 7), OUT / 7, hipMemcpyDeviceToHost, s[1])); } });
    timeit("the same with two streams per direction", IN + OUT, [&] {
        for (int k = 0; k < 7; ++k) { CHECK(hipMemcpyAsync(d_in + k * (IN / 7), h_in + k * (IN / 7), IN / 7, hipMemcpyHostToDevice, s[k & 1]));
                                      CHECK(hipMemcpyAsync(h_out + k * (OUT / 7), d_out + k * (OUT / 7), OUT / 7, hipMemcpyDeviceToHost, s[2 + (k & 1)])); } });
    std::vector<char> pageable(big);
    timeit("D2H 31 MB into PAGEABLE memory, one copy", OUT, [&] { CHECK(hipMemcpyAsync(pageable.data(), d_out, OUT, hipMemcpyDeviceToHost, s[0])); });
    timeit("H2D 21 MB from PAGEABLE memory, one copy", IN, [&] { CHECK(hipMemcpyAsync(d_in, pageable.data(), IN, hipMemcpyHostToDevice, s[0])); });
    return 0;
}
