// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access SHAPES of score_tiles_fast_kernel
// (diagnostics, not product code).  MI355X_MICROARCH.md, HBM section: FETCH_SIZE tallies a wide coalesced stream at
// half its bytes; "other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own
// access pattern".  Every kernel below moves a KNOWN number of bytes (printed as JSON, one line per kernel) in one of
// those shapes over a 1 GiB table (4x the Infinity Cache, so re-use does not hide fetches):
//   calib_stream_kernel    16 B per lane, coalesced                  (the text staging)
//   calib_pair128_kernel   lane pairs read the two 64-B halves of a random 128-B record   (the record gathers)
//   calib_half64_kernel    every lane reads a random 64-B half line (4 x 16 B)
//   calib_q16_kernel       every lane reads a random 16-B unit        (unigram rows, trie entries)
//   calib_store4_kernel    4 B per lane, coalesced stores            (the i32 scores)
//   calib_store1_kernel    1 B per lane, coalesced stores            (the u8 labels)
//   hipcc --offload-arch=gfx950 -O3 -o tools/calib_fetch tools/calib_fetch.hip
//   rocprofv3 --pmc FETCH_SIZE ... -- tools/calib_fetch          (tools/profile.sh does the passes and the arithmetic)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e_), #x); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}

__global__ __launch_bounds__(256) void calib_stream_kernel(const uint4* __restrict__ tab, size_t n16, uint32_t* out) {
    uint32_t acc = 0;
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n16; i += size_t(gridDim.x) * 256) { const uint4 v = tab[i]; acc ^= v.x ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void calib_pair128_kernel(const uint4* __restrict__ tab, uint32_t mask16, int iters, uint32_t* out) {
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        uint32_t h = mix((tid >> 1) * 0x9E3779B1u + uint32_t(i) * 0x85EBCA77u) & mask16;
        h = (h & ~7u) + ((tid & 1u) << 2);
        const uint4 a = tab[h], b = tab[h + 1], c = tab[h + 2], d = tab[h + 3];
        acc ^= a.x ^ b.y ^ c.z ^ d.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void calib_half64_kernel(const uint4* __restrict__ tab, uint32_t mask16, int iters, uint32_t* out) {
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        const uint32_t h = mix(tid * 0x9E3779B1u + uint32_t(i) * 0x85EBCA77u) & mask16 & ~3u;
        const uint4 a = tab[h], b = tab[h + 1], c = tab[h + 2], d = tab[h + 3];
        acc ^= a.x ^ b.y ^ c.z ^ d.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void calib_q16_kernel(const uint4* __restrict__ tab, uint32_t mask16, int iters, uint32_t* out) {
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        const uint4 a = tab[mix(tid * 0x9E3779B1u + uint32_t(i) * 0x85EBCA77u) & mask16];
        acc ^= a.x ^ a.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void calib_store4_kernel(uint32_t* __restrict__ dst, size_t n) {
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += size_t(gridDim.x) * 256) dst[i] = uint32_t(i);
}
__global__ __launch_bounds__(256) void calib_store1_kernel(uint8_t* __restrict__ dst, size_t n) {
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += size_t(gridDim.x) * 256) dst[i] = uint8_t(i);
}

int main() {
    const size_t bytes = size_t(1) << 30;
    const int blocks = 256 * 8, iters = 32, reps = 3;
    uint4* tab; uint32_t* out;
    CHECK(hipMalloc(&tab, bytes + 256)); CHECK(hipMemset(tab, 1, bytes + 256)); CHECK(hipMalloc(&out, 64));
    const uint32_t mask16 = uint32_t(bytes / 16 - 1);
    const double lanes = double(blocks) * 256 * iters;
    for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL(calib_stream_kernel, dim3(blocks), dim3(256), 0, 0, tab, bytes / 16, out);
        hipLaunchKernelGGL(calib_pair128_kernel, dim3(blocks), dim3(256), 0, 0, tab, mask16, iters, out);
        hipLaunchKernelGGL(calib_half64_kernel, dim3(blocks), dim3(256), 0, 0, tab, mask16, iters, out);
        hipLaunchKernelGGL(calib_q16_kernel, dim3(blocks), dim3(256), 0, 0, tab, mask16, iters, out);
        hipLaunchKernelGGL(calib_store4_kernel, dim3(blocks), dim3(256), 0, 0, reinterpret_cast<uint32_t*>(tab), bytes / 4);
        hipLaunchKernelGGL(calib_store1_kernel, dim3(blocks), dim3(256), 0, 0, reinterpret_cast<uint8_t*>(tab), bytes / 4);
        CHECK(hipDeviceSynchronize());
    }
    // known bytes per launch; "lines" = distinct 128-byte lines requested (an upper bound: random picks may repeat, 1 GiB
    // holds 8.4 M lines and a launch asks for 8.4 M (pair128) or 16.8 M (half64, q16) of them with replacement)
    printf("{\"kernel\": \"calib_stream_kernel\", \"bytes\": %.0f}\n", double(bytes));
    printf("{\"kernel\": \"calib_pair128_kernel\", \"bytes\": %.0f, \"requests\": %.0f, \"request_bytes\": 128}\n", lanes / 2 * 128, lanes / 2);
    printf("{\"kernel\": \"calib_half64_kernel\", \"bytes\": %.0f, \"requests\": %.0f, \"request_bytes\": 64}\n", lanes * 64, lanes);
    printf("{\"kernel\": \"calib_q16_kernel\", \"bytes\": %.0f, \"requests\": %.0f, \"request_bytes\": 16}\n", lanes * 16, lanes);
    printf("{\"kernel\": \"calib_store4_kernel\", \"bytes\": %.0f}\n", double(bytes));
    printf("{\"kernel\": \"calib_store1_kernel\", \"bytes\": %.0f}\n", double(bytes / 4));
    return 0;
}
