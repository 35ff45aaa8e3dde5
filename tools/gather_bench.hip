// Microbenchmark (diagnostics, not product code): rate of random gathers on MI355X by access shape.
//   hipcc --offload-arch=gfx950 -O3 -o gather_bench tools/gather_bench.hip && ./gather_bench
// Every thread issues ITER independent random accesses (address = hash(thread, i)); modes:
//   0: 16 B from a random 16-B slot                1: 64 B (4 x 16) from a random 64-B record
//   2: 128 B (8 x 16) from a random 128-B record   3: lane PAIRS share a random 128-B record, 64 B each
//   4: 32 B (2 x 16) from a random 32-B record     5: as 1 with non-temporal loads
//   6: lanes L and L+32 share a random 128-B record, 64 B each (the pairing v_permlane32_swap_b32 can undo)
// Second table (the kernel's access MIX, 256 MB of records, 77 % of the accesses to a 2 MB hot set = its L2 hit rate),
// every lane needs its own record, rates in records/s:
//   7: lane pairs fetch both halves of the even lane's record, then of the odd lane's (what score_tiles_fast_kernel does)
//   8: every lane fetches the first 64 B of its record, 43 % of the lanes also the second 64 B ("light" records)
//   9: as 8 with the second half requested by the partner lane in the same instruction (pair issue, masked)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e_), #x); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}

template <int MODE>
__global__ __launch_bounds__(256) void gather(const uint4* __restrict__ tab, uint32_t mask16, int iters, uint32_t* out) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        uint32_t who = (MODE == 3) ? (tid >> 1) : (MODE == 6) ? ((tid >> 6) * 32 + (tid & 31)) : tid;
        uint32_t h = mix(who * 0x9E3779B1u + uint32_t(i) * 0x85EBCA77u) & mask16;   // index in 16-B units
        if (MODE == 0) { uint4 v = tab[h]; acc ^= v.x ^ v.w; }
        else if (MODE == 4) { h &= ~1u; uint4 a = tab[h], b = tab[h + 1]; acc ^= a.x ^ b.w; }
        else if (MODE == 1) { h &= ~3u; uint4 a = tab[h], b = tab[h + 1], c = tab[h + 2], d = tab[h + 3]; acc ^= a.x ^ b.y ^ c.z ^ d.w; }
        else if (MODE == 5) {
            h &= ~3u;
            const uint32_t* p = reinterpret_cast<const uint32_t*>(tab + h);
            uint32_t a = __builtin_nontemporal_load(p), b = __builtin_nontemporal_load(p + 5), c = __builtin_nontemporal_load(p + 10), d = __builtin_nontemporal_load(p + 15);
            acc ^= a ^ b ^ c ^ d;
        }
        else if (MODE == 2) {
            h &= ~7u;
            uint4 a = tab[h], b = tab[h + 1], c = tab[h + 2], d = tab[h + 3], e = tab[h + 4], f = tab[h + 5], g = tab[h + 6], k = tab[h + 7];
            acc ^= a.x ^ b.y ^ c.z ^ d.w ^ e.x ^ f.y ^ g.z ^ k.w;
        } else if (MODE == 6) {
            h = (h & ~7u) + (((tid >> 5) & 1u) << 2);
            uint4 a = tab[h], b = tab[h + 1], c = tab[h + 2], d = tab[h + 3]; acc ^= a.x ^ b.y ^ c.z ^ d.w;
        } else if (MODE == 3) {
            h = (h & ~7u) + ((tid & 1u) << 2);
            uint4 a = tab[h], b = tab[h + 1], c = tab[h + 2], d = tab[h + 3]; acc ^= a.x ^ b.y ^ c.z ^ d.w;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;  // keep the loads alive
}

// record index for the skewed mix: 77 % from the first `hot` records, the rest uniform over the table
__device__ uint32_t g_hot_per_1024 = 788u;   // share of the accesses that go to the hot set
__device__ __forceinline__ uint32_t skewed(uint32_t who, uint32_t i, uint32_t nrec_mask, uint32_t hot_mask) {
    const uint32_t r = mix(who * 0x9E3779B1u + i * 0x85EBCA77u);
    const uint32_t pick = mix(r ^ 0x5bd1e995u);
    return ((pick & 1023u) < g_hot_per_1024) ? (r & hot_mask) : (r & nrec_mask);
}

template <int MODE>
__global__ __launch_bounds__(256) void gather_mix(const uint4* __restrict__ tab, uint32_t nrec_mask, uint32_t hot_mask, int iters, uint32_t* out) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const bool odd = tid & 1u;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        const uint32_t mine = skewed(tid, uint32_t(i), nrec_mask, hot_mask);
        const uint32_t partner = skewed(tid ^ 1u, uint32_t(i), nrec_mask, hot_mask);
        const bool heavy = (mix(mine + 77u) & 127u) < 55u, partner_heavy = (mix(partner + 77u) & 127u) < 55u;   // 43 %
        if (MODE == 7) {
            const uint32_t ra = ((odd ? partner : mine) << 3) + (odd ? 4u : 0u), rb = ((odd ? mine : partner) << 3) + (odd ? 0u : 4u);
            uint4 a = tab[ra], b = tab[ra + 1], c = tab[ra + 2], d = tab[ra + 3], e = tab[rb], f = tab[rb + 1], g = tab[rb + 2], k = tab[rb + 3];
            acc ^= a.x ^ b.y ^ c.z ^ d.w ^ e.x ^ f.y ^ g.z ^ k.w;
        } else if (MODE == 8) {
            const uint32_t r = mine << 3;
            uint4 a = tab[r], b = tab[r + 1], c = tab[r + 2], d = tab[r + 3];
            acc ^= a.x ^ b.y ^ c.z ^ d.w;
            if (heavy) { uint4 e = tab[r + 4], f = tab[r + 5], g = tab[r + 6], k = tab[r + 7]; acc ^= e.x ^ f.y ^ g.z ^ k.w; }
        } else {   // 9: first the even lane's record (even lane: half 0, odd lane: half 1 if that record is heavy), then the odd lane's
            const uint32_t ra = ((odd ? partner : mine) << 3) + (odd ? 4u : 0u), rb = ((odd ? mine : partner) << 3) + (odd ? 0u : 4u);
            if (!odd || partner_heavy) { uint4 a = tab[ra], b = tab[ra + 1], c = tab[ra + 2], d = tab[ra + 3]; acc ^= a.x ^ b.y ^ c.z ^ d.w; }
            if (odd || partner_heavy) { uint4 e = tab[rb], f = tab[rb + 1], g = tab[rb + 2], k = tab[rb + 3]; acc ^= e.x ^ f.y ^ g.z ^ k.w; }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}


// ---- whole-position access mixes (round 2): what ONE start position asks of the memory system, all lanes busy, no ALU work.
//   v1: today's kernel -- a seed byte (512 KB array, random), a 16-byte unigram row (1 MB table, 4 K rows in use, 256 B apart),
//       the pair-issued 128-byte record (256 MB, 77 % of the accesses to a 2 MB hot set)
//   v2: a double-array trie walk -- a 16-byte unigram node (64 KB, dense), a 32-byte bigram node (64 MB of nodes, hot set 16 K nodes),
//       a 16-byte trigram node for `tri_per_1024` of the lanes (32 MB of nodes, hot set 32 K nodes)
__device__ __forceinline__ uint32_t zipfish(uint32_t r, uint32_t n_mask, uint32_t hot_mask) {
    const uint32_t pick = mix(r ^ 0x5bd1e995u);
    return ((pick & 1023u) < 788u) ? (r & hot_mask) : (r & n_mask);
}
__global__ __launch_bounds__(256) void mix_v1(const uint4* __restrict__ rec, const uint4* __restrict__ uni, const uint8_t* __restrict__ seed, int iters, uint32_t* out) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const bool odd = tid & 1u;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        const uint32_t r = mix(tid * 0x9E3779B1u + uint32_t(i) * 0x85EBCA77u), rp = mix((tid ^ 1u) * 0x9E3779B1u + uint32_t(i) * 0x85EBCA77u);
        const uint32_t mine = zipfish(r, (1u << 21) - 1, (1u << 14) - 1), partner = zipfish(rp, (1u << 21) - 1, (1u << 14) - 1);
        const uint32_t sd = seed[mix(r + 1) & ((1u << 19) - 1)];
        const uint4 u = uni[zipfish(mix(r + 2), 4095u, 255u) << 4];
        const uint32_t ra = ((odd ? partner : mine) << 3) + (odd ? 4u : 0u), rb = ((odd ? mine : partner) << 3) + (odd ? 0u : 4u);
        const uint4 a = rec[ra], b = rec[ra + 1], c = rec[ra + 2], d = rec[ra + 3], e = rec[rb], f = rec[rb + 1], g = rec[rb + 2], k = rec[rb + 3];
        acc ^= a.x ^ b.y ^ c.z ^ d.w ^ e.x ^ f.y ^ g.z ^ k.w ^ u.x ^ sd;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void mix_v2(const uint4* __restrict__ bi, const uint4* __restrict__ tri, const uint4* __restrict__ uni, uint32_t tri_per_1024, int iters, uint32_t* out) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        const uint32_t r = mix(tid * 0x9E3779B1u + uint32_t(i) * 0x85EBCA77u);
        const uint4 u = uni[zipfish(mix(r + 2), 4095u, 255u)];
        const uint32_t nb = zipfish(r, (1u << 21) - 1, (1u << 14) - 1) << 1;
        const uint4 b0 = bi[nb], b1 = bi[nb + 1];
        acc ^= u.x ^ b0.y ^ b1.z;
        if ((mix(r + 3) & 1023u) < tri_per_1024) { const uint4 t = tri[zipfish(mix(r + 4), (1u << 21) - 1, (1u << 15) - 1)]; acc ^= t.w; }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

//   v3: the trigram children of a bigram node in the node's own 128-byte line (a 32-byte node + six 16-byte child slots): the child is
//       asked for together with the node (its slot follows from the third char alone), one L2 request for both levels; `over_per_1024`
//       of the trigram lanes find their child in the separate table all the same (bigrams with more children than fit)
__global__ __launch_bounds__(256) void mix_v3(const uint4* __restrict__ bi, const uint4* __restrict__ tri, const uint4* __restrict__ uni, uint32_t tri_per_1024,
                                              uint32_t over_per_1024, int iters, uint32_t* out) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        const uint32_t r = mix(tid * 0x9E3779B1u + uint32_t(i) * 0x85EBCA77u);
        const uint4 u = uni[zipfish(mix(r + 2), 4095u, 255u)];
        const uint32_t nb = zipfish(r, (1u << 21) - 1, (1u << 14) - 1) << 3;   // 128-byte records
        const bool want_tri = (mix(r + 3) & 1023u) < tri_per_1024, over = (mix(r + 5) & 1023u) < over_per_1024;
        const uint4 b0 = bi[nb], b1 = bi[nb + 1];
        uint4 t = make_uint4(0, 0, 0, 0);
        if (want_tri && !over) t = bi[nb + 2 + (mix(r + 4) % 6u)];
        acc ^= u.x ^ b0.y ^ b1.z ^ t.w;
        if (want_tri && over) { const uint4 t2 = tri[zipfish(mix(r + 4), (1u << 21) - 1, (1u << 15) - 1)]; acc ^= t2.w; }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int MODE>
double run_mix(const uint4* tab, uint32_t nrec_mask, uint32_t hot_mask, int blocks, int iters, uint32_t* out) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(gather_mix<MODE>, dim3(blocks), dim3(256), 0, 0, tab, nrec_mask, hot_mask, iters, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(gather_mix<MODE>, dim3(blocks), dim3(256), 0, 0, tab, nrec_mask, hot_mask, iters, out);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / 5.0;
}

template <int MODE>
double run(const uint4* tab, uint32_t mask16, int blocks, int iters, uint32_t* out) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(gather<MODE>, dim3(blocks), dim3(256), 0, 0, tab, mask16, iters, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(gather<MODE>, dim3(blocks), dim3(256), 0, 0, tab, mask16, iters, out);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / 5.0;
}

int main() {
    const int blocks = 256 * 8, iters = 64;
    uint32_t* out; CHECK(hipMalloc(&out, 64));
    const double positions = double(blocks) * 256 * iters;
    printf("positions per launch: %.1f M (mode 3: half as many records)\n", positions / 1e6);
    for (int logmb : {1, 2, 5, 7, 9}) {   // 2 MB, 4 MB, 32 MB, 128 MB, 512 MB
        const size_t bytes = size_t(1) << (20 + logmb);
        uint4* tab; CHECK(hipMalloc(&tab, bytes + 256)); CHECK(hipMemset(tab, 1, bytes + 256));
        const uint32_t mask16 = uint32_t(bytes / 16 - 1);
        double t0 = run<0>(tab, mask16, blocks, iters, out), t4 = run<4>(tab, mask16, blocks, iters, out), t1 = run<1>(tab, mask16, blocks, iters, out);
        double t5 = run<5>(tab, mask16, blocks, iters, out), t2 = run<2>(tab, mask16, blocks, iters, out), t3 = run<3>(tab, mask16, blocks, iters, out);
        double t6 = run<6>(tab, mask16, blocks, iters, out);
        printf("table %4zu MB | G accesses/s: 16B %.1f | 32B %.1f | 64B %.1f | 64B-nt %.1f | 128B %.1f | pair128B %.1f (records/s %.1f) | half-wave pair records/s %.1f\n", bytes >> 20,
               positions / t0 / 1e6, positions / t4 / 1e6, positions / t1 / 1e6, positions / t5 / 1e6, positions / t2 / 1e6, positions / t3 / 1e6, positions / 2 / t3 / 1e6, positions / 2 / t6 / 1e6);
        CHECK(hipFree(tab));
    }
    {   // the kernel's mix: 2 M records of 128 B, hot set 16 K records
        const size_t bytes = size_t(256) << 20;
        uint4* tab; CHECK(hipMalloc(&tab, bytes + 256)); CHECK(hipMemset(tab, 1, bytes + 256));
        const uint32_t nrec_mask = uint32_t(bytes / 128 - 1), hot_mask = (1u << 14) - 1;
        const double t7 = run_mix<7>(tab, nrec_mask, hot_mask, blocks, iters, out), t8 = run_mix<8>(tab, nrec_mask, hot_mask, blocks, iters, out);
        const double t9 = run_mix<9>(tab, nrec_mask, hot_mask, blocks, iters, out);
        printf("skewed mix, one record per lane | G records/s: pair-issued 128 B %.1f | 64 B + 43%% second half, own lane %.1f | same, pair-issued %.1f\n",
               positions / t7 / 1e6, positions / t8 / 1e6, positions / t9 / 1e6);
        // the same pair-issued record gather with a hotter mix: what filtering the absent keys (14 % of the positions,
        // all of them cold) before the fetch would leave -- 788/1024 hot of 100 % -> 788/(1024 - 0.8 * 143) = 88 % hot
        for (uint32_t hot : {788u, 900u, 1024u}) {
            CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_hot_per_1024), &hot, sizeof(hot)));
            const double t = run_mix<7>(tab, nrec_mask, hot_mask, blocks, iters, out);
            printf("pair-issued 128 B, hot share %u/1024: %.1f G records/s\n", hot, positions / t / 1e6);
        }
        CHECK(hipFree(tab));
    }
    {   // whole-position mixes
        uint4 *rec, *uni, *bi, *tri; uint8_t* seed;
        CHECK(hipMalloc(&rec, (size_t(256) << 20) + 256)); CHECK(hipMemset(rec, 1, (size_t(256) << 20) + 256));
        CHECK(hipMalloc(&uni, (size_t(1) << 20) + 256)); CHECK(hipMemset(uni, 1, (size_t(1) << 20) + 256));
        CHECK(hipMalloc(&seed, (size_t(1) << 19) + 256)); CHECK(hipMemset(seed, 1, (size_t(1) << 19) + 256));
        CHECK(hipMalloc(&bi, (size_t(64) << 20) + 256)); CHECK(hipMemset(bi, 1, (size_t(64) << 20) + 256));
        CHECK(hipMalloc(&tri, (size_t(32) << 20) + 256)); CHECK(hipMemset(tri, 1, (size_t(32) << 20) + 256));
        auto time5 = [&](auto launch) {
            hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
            launch(); CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(a)); for (int r = 0; r < 5; ++r) launch(); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
            float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b)); return double(ms) / 5.0; };
        const double t1 = time5([&] { hipLaunchKernelGGL(mix_v1, dim3(blocks), dim3(256), 0, 0, rec, uni, seed, iters, out); });
        printf("position mix v1 (seed + unigram row + pair-issued 128-B record): %.1f G positions/s\n", positions / t1 / 1e6);
        for (uint32_t tp : {0u, 300u, 560u, 1024u}) {
            const double t2 = time5([&] { hipLaunchKernelGGL(mix_v2, dim3(blocks), dim3(256), 0, 0, bi, tri, uni, tp, iters, out); });
            printf("position mix v2 (16-B unigram node + 32-B bigram node + 16-B trigram node for %u/1024 of the lanes): %.1f G positions/s\n", tp, positions / t2 / 1e6);
        }
        uint4* bi3;   // the same 2 M bigram nodes as 128-byte records
        CHECK(hipMalloc(&bi3, (size_t(256) << 20) + 256)); CHECK(hipMemset(bi3, 1, (size_t(256) << 20) + 256));
        for (uint32_t ov : {0u, 300u, 600u}) {
            const double t3 = time5([&] { hipLaunchKernelGGL(mix_v3, dim3(blocks), dim3(256), 0, 0, bi3, tri, uni, 560u, ov, iters, out); });
            printf("position mix v3 (trigram child in its bigram node's 128-B line, 560/1024 of the lanes want one, %u/1024 of those from the separate table): %.1f G positions/s\n", ov, positions / t3 / 1e6);
        }
    }
    return 0;
}
