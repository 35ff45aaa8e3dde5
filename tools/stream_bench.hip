// Microbenchmark (diagnostics, not product code): what the memory system gives a kernel with the TRAFFIC of fill_tags' front end and nothing else.
//   hipcc --offload-arch=gfx950 -O3 -o tools/stream_bench tools/stream_bench.hip && tools/stream_bench [chars]
// The front end (kernels_tags.hip, tag_front_flat_kernel) reads 4 + 1 bytes per char (the decoded char, its label) and writes 4 + 4 * n_tags
// (the writer's tok_model word, the None entries): on BASELINE's configs[4] (133 M chars, 2 tags) 0.67 GB in, 1.6 GB out per launch.  Modes:
//   0  write only: a dword and a dwordx2 per char (the None entries alone)            1  hipMemsetAsync over the same 12 bytes per char
//   2  read only: a dword and a byte per char                                         3  both (mode 0 + mode 2: the front end's traffic)
//   4  as 3 with steps of 128 chars per wave and the loads of the next step issued before the stores of this one (the front end's shape)
//   5  as 3 with 16 bytes per lane (4 chars per lane: dwordx4 load, dwordx4 + 2 x dwordx4 stores)
// Each is timed with events over `reps` launches; GB/s = bytes the mode moves / time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e_), #x); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void stream(const uint32_t* __restrict__ cps, const uint8_t* __restrict__ labels, int32_t* __restrict__ tm,
                                              int2* __restrict__ tags, uint64_t n, uint32_t* out) {
    const uint64_t stride = uint64_t(gridDim.x) * blockDim.x;
    uint32_t acc = 0;
    for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (MODE & 2) acc += cps[i] + labels[i];
        if (MODE != 2) { tm[i] = 0; tags[i] = make_int2(-1, -1); }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// a wave walks a run of 2 K chars in steps of 128, the next step's loads in flight over this step's stores
__global__ __launch_bounds__(256) void stream_steps(const uint32_t* __restrict__ cps, const uint8_t* __restrict__ labels, int32_t* __restrict__ tm,
                                                    int2* __restrict__ tags, uint64_t n, uint32_t* out) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6, n_waves = (uint64_t(gridDim.x) * blockDim.x) >> 6;
    uint32_t acc = 0;
    for (uint64_t run = wave; run * 2048 < n; run += n_waves) {
        const uint64_t r0 = run * 2048, r1 = r0 + 2048 < n ? r0 + 2048 : n;
        uint32_t c0 = 0, c1 = 0, b0 = 0, b1 = 0;
        if (r0 + lane < r1) { c0 = cps[r0 + lane]; b0 = labels[r0 + lane]; }
        if (r0 + 64 + lane < r1) { c1 = cps[r0 + 64 + lane]; b1 = labels[r0 + 64 + lane]; }
        for (uint64_t base = r0; base < r1; base += 128) {
            const uint32_t x0 = c0 + b0, x1 = c1 + b1;
            const uint64_t nb = base + 128;
            if (nb + lane < r1) { c0 = cps[nb + lane]; b0 = labels[nb + lane]; }
            if (nb + 64 + lane < r1) { c1 = cps[nb + 64 + lane]; b1 = labels[nb + 64 + lane]; }
            acc += x0 ^ x1;
            if (base + lane < r1) { tm[base + lane] = 0; tags[base + lane] = make_int2(-1, -1); }
            if (base + 64 + lane < r1) { tm[base + 64 + lane] = 0; tags[base + 64 + lane] = make_int2(-1, -1); }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

__global__ __launch_bounds__(256) void stream_wide(const uint4* __restrict__ cps, const uint32_t* __restrict__ labels, int4* __restrict__ tm,
                                                   int4* __restrict__ tags, uint64_t n4, uint32_t* out) {
    const uint64_t stride = uint64_t(gridDim.x) * blockDim.x;
    uint32_t acc = 0;
    for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const uint4 c = cps[i];
        acc += c.x + c.y + c.z + c.w + labels[i];
        tm[i] = make_int4(0, 0, 0, 0);
        tags[2 * i] = make_int4(-1, -1, -1, -1);
        tags[2 * i + 1] = make_int4(-1, -1, -1, -1);
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char** argv) {
    const uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 133000000ull;
    const int reps = 10;
    uint32_t *cps, *out; uint8_t* labels; int32_t* tm; int2* tags;
    CHECK(hipMalloc(&cps, n * 4 + 64)); CHECK(hipMalloc(&labels, n + 64)); CHECK(hipMalloc(&tm, n * 4 + 64)); CHECK(hipMalloc(&tags, n * 8 + 64)); CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(cps, 1, n * 4)); CHECK(hipMemset(labels, 1, n)); CHECK(hipMemset(out, 0, 64));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("%s, %d CUs, %llu chars\n", prop.name, cus, (unsigned long long)n);
    for (int mode = 0; mode <= 5; ++mode) {
        for (int wg_per_cu : {8, 32, 0}) {   // 0: one thread per element
            if ((mode == 1) && wg_per_cu != 8) continue;
            if ((mode == 4) && wg_per_cu == 0) continue;
            const uint64_t elems = mode == 5 ? n / 4 : n;
            const uint32_t blocks = wg_per_cu ? uint32_t(cus * wg_per_cu) : uint32_t((elems + 255) / 256);
            float best = 1e30f, total = 0;
            for (int r = 0; r < reps + 2; ++r) {
                CHECK(hipEventRecord(e0, 0));
                switch (mode) {
                    case 0: hipLaunchKernelGGL(stream<0>, dim3(blocks), dim3(256), 0, 0, cps, labels, tm, tags, n, out); break;
                    case 1: CHECK(hipMemsetAsync(tm, 0, n * 4, 0)); CHECK(hipMemsetAsync(tags, 0xFF, n * 8, 0)); break;
                    case 2: hipLaunchKernelGGL(stream<2>, dim3(blocks), dim3(256), 0, 0, cps, labels, tm, tags, n, out); break;
                    case 3: hipLaunchKernelGGL(stream<3>, dim3(blocks), dim3(256), 0, 0, cps, labels, tm, tags, n, out); break;
                    case 4: hipLaunchKernelGGL(stream_steps, dim3(blocks), dim3(256), 0, 0, cps, labels, tm, tags, n, out); break;
                    case 5: hipLaunchKernelGGL(stream_wide, dim3(blocks), dim3(256), 0, 0, reinterpret_cast<const uint4*>(cps), reinterpret_cast<const uint32_t*>(labels),
                                               reinterpret_cast<int4*>(tm), reinterpret_cast<int4*>(tags), n / 4, out); break;
                }
                CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
                float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (r >= 2) { total += ms; if (ms < best) best = ms; }
            }
            const double rd = (mode & 2) || mode >= 4 ? double(n) * 5 : 0, wr = mode != 2 ? double(n) * 12 : 0;
            const double rd2 = (mode == 3 || mode >= 4) ? double(n) * 5 : rd;
            const double bytes = (mode == 2 ? rd : (mode == 0 || mode == 1) ? wr : rd2 + wr);
            printf("mode %d  grid %6u x 256  avg %.3f ms  best %.3f ms  %.0f GB/s (%.2f GB)\n", mode, mode == 1 ? 0u : blocks, total / reps, best, bytes / (total / reps) / 1e6, bytes / 1e9);
        }
    }
    return 0;
}
