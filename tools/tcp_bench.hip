// Microbenchmark (diagnostics, not product code): what a CU's vector L1 (TCP) takes per clock, by access shape -- the unit the scoring kernel's
// time is made of.  Every lane issues ITER independent random loads of one shape from a table that fits the L1 (16 KB), the L2
// (2 MB) or neither (256 MB); one kernel NAME per shape, so that a `rocprofv3 --pmc TCP_TOTAL_ACCESSES_sum GRBM_GUI_ACTIVE` pass over this
// binary gives the counter's units per lane-load and per clock for each:
//   hipcc --offload-arch=gfx950 -O3 -o tools/tcp_bench tools/tcp_bench.hip && tools/tcp_bench
// Shapes: b4 (one dword), b4x2 (two dwords of one 16-byte slot), b8 (dwordx2), b16 (dwordx4), b32 (two adjacent dwordx4: a 32-byte node), b64 (four adjacent dwordx4: a 64-byte entry),
// mix (the scoring kernel's per-position mix on M1, profiles/r05_t_hit_share_sweep.jsonl: a dword (cid), a 16-byte unigram node, a 32-byte bigram
// node, a 16-byte trigram node for 46 % of the lanes, a 64-byte deep entry for 21 %; mix32: the same with 32-byte deep entries).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e_), #x); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
constexpr int kIter = 64;

#define SHAPE_KERNEL(name, ...)                                                                                           \
    __global__ __launch_bounds__(256) void name(const uint4* __restrict__ tab, uint32_t mask16, uint32_t* out) {         \
        const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;                                                       \
        uint32_t acc = 0;                                                                                                 \
        for (int i = 0; i < kIter; ++i) {                                                                                 \
            const uint32_t r = mix(tid * 0x9E3779B1u + uint32_t(i) * 0x85EBCA77u);                                        \
            uint32_t h = r & mask16;                                                                                      \
            __VA_ARGS__                                                                                                   \
        }                                                                                                                 \
        if (acc == 0x12345678u) out[0] = acc;                                                                             \
    }
SHAPE_KERNEL(gather_b4, { acc ^= reinterpret_cast<const uint32_t*>(tab)[h * 4 + (r >> 30)]; })
// (every component is used: the compiler narrows a load to the dwords that are)
#define X4(v) ((v).x ^ (v).y ^ (v).z ^ (v).w)
SHAPE_KERNEL(gather_b8, { const uint2 v = reinterpret_cast<const uint2*>(tab)[h * 2 + (r >> 31)]; acc ^= v.x ^ v.y; })
SHAPE_KERNEL(gather_b16, { const uint4 v = tab[h]; acc ^= X4(v); })
SHAPE_KERNEL(gather_b32, { h &= ~1u; const uint4 a = tab[h], b = tab[h + 1]; acc ^= X4(a) ^ X4(b); })
SHAPE_KERNEL(gather_b64, { h &= ~3u; const uint4 a = tab[h], b = tab[h + 1], c = tab[h + 2], d = tab[h + 3]; acc ^= X4(a) ^ X4(b) ^ X4(c) ^ X4(d); })
SHAPE_KERNEL(gather_b4x2, { acc ^= reinterpret_cast<const uint32_t*>(tab)[h * 4] ^ reinterpret_cast<const uint32_t*>(tab)[h * 4 + 3]; })
// the kernel's mix: independent loads of one position (no dependent trips: the rate of the memory pipe, not the latency of a walk)
#define MIX_BODY(DEEP16)                                                                                                  \
    {                                                                                                                     \
        acc ^= reinterpret_cast<const uint32_t*>(tab)[(mix(r + 1) & mask16) * 4];                                         \
        const uint4 u = tab[mix(r + 2) & mask16];                                                                         \
        const uint32_t nb = mix(r + 3) & mask16 & ~1u;                                                                    \
        const uint4 b0 = tab[nb], b1 = tab[nb + 1];                                                                       \
        acc ^= X4(u) ^ X4(b0) ^ X4(b1);                                                                                   \
        if ((mix(r + 4) & 1023u) < 471u) { const uint4 t = tab[mix(r + 5) & mask16]; acc ^= X4(t); }                     \
        if ((mix(r + 6) & 1023u) < 215u) {                                                                                \
            const uint32_t nd = mix(r + 7) & mask16 & ~3u;                                                                \
            for (int q = 0; q < DEEP16; ++q) { const uint4 d = tab[nd + q]; acc ^= X4(d); }                               \
        }                                                                                                                 \
    }
SHAPE_KERNEL(gather_mix64, MIX_BODY(4))
SHAPE_KERNEL(gather_mix32, MIX_BODY(2))

template <typename K>
double run(K kernel, const uint4* tab, uint32_t mask16, int blocks, uint32_t* out) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, tab, mask16, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, tab, mask16, out);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / 5.0;
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate / 1e6;
    const int blocks = cus * 8 * 4;
    const double lanes = double(blocks) * 256 * kIter;
    uint32_t* out; CHECK(hipMalloc(&out, 64));
    printf("{\"cus\": %d, \"clock_ghz\": %.3f, \"lanes_per_launch\": %.0f, \"note\": \"per_clk_cu = lane-items per clock and CU at the reported clock\"}\n", cus, ghz, lanes);
    const size_t sizes[3] = {size_t(16) << 10, size_t(2) << 20, size_t(256) << 20};
    const char* where[3] = {"L1 (16 KB)", "L2 (2 MB)", "HBM / MALL (256 MB)"};
    for (int s = 0; s < 3; ++s) {
        uint4* tab; CHECK(hipMalloc(&tab, sizes[s] + 256)); CHECK(hipMemset(tab, 1, sizes[s] + 256));
        const uint32_t mask16 = uint32_t(sizes[s] / 16 - 1);
        struct { const char* name; double ms; } rows[] = {
            {"b4", run(gather_b4, tab, mask16, blocks, out)}, {"b4x2", run(gather_b4x2, tab, mask16, blocks, out)}, {"b8", run(gather_b8, tab, mask16, blocks, out)}, {"b16", run(gather_b16, tab, mask16, blocks, out)},
            {"b32", run(gather_b32, tab, mask16, blocks, out)}, {"b64", run(gather_b64, tab, mask16, blocks, out)},
            {"mix64", run(gather_mix64, tab, mask16, blocks, out)}, {"mix32", run(gather_mix32, tab, mask16, blocks, out)}};
        for (auto& r : rows)
            printf("{\"table\": \"%s\", \"shape\": \"%s\", \"ms\": %.4f, \"G_items_s\": %.1f, \"items_per_clk_cu\": %.3f}\n", where[s], r.name, r.ms, lanes / r.ms / 1e6,
                   lanes / (r.ms * 1e-3) / (double(cus) * ghz * 1e9));
        CHECK(hipFree(tab));
    }
    return 0;
}
