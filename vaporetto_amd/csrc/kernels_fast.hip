// Specialised tile kernel for the model geometry every distributed Vaporetto / KyTea model has:
// char window 3, char n-grams of at most 3 chars (+ dictionary words of any length), type scores from the
// 8^(2W) window table (W <= 3) or none.  Same algorithm and tables as kernels.hip (which stays the general
// path); what changes is how the work is scheduled on a CDNA4 wave:
//
//   * geometry is compile-time: level rows are 6/5/4 slots at boundary offsets -3/-2/-1, one 32-byte entry =
//     two dwordx4 loads, so the per-start-position code is branch-free: all six loads (unigram row, bigram slot,
//     trigram slot) are issued before anything is compared, the three rows are summed in registers and land in
//     the LDS score array with six ds_add_u32;
//   * everything data-dependent -- a probe that has to continue past its home slot (kDisplacedBit), a
//     non-BMP unigram, a dictionary word longer than 3 chars walking the trie -- is NOT done in place (64 lanes
//     would wait for the unluckiest one) but pushed, ballot/mbcnt-compacted, onto a wave-private LDS queue and
//     replayed 64 items at a time with every lane busy; a trie step that matches re-queues its continuation;
//   * UTF-8 decode is two-step: a chunk scan finds (byte position, sentence) of every char, then one thread
//     per CHAR decodes from the LDS-staged text, so lanes are dense there too;
//   * 20 KB of LDS per workgroup and <= 64 VGPRs: 8 workgroups = 32 waves per CU.
#include <hip/hip_runtime.h>

#include "device_common.h"
#include "kernels.hpp"

namespace vpt {
namespace {

constexpr int kQCap = 192;                   // deferred items per wave
constexpr int kQHigh = kQCap - 64;           // replay until there is room for one more push of 64
constexpr uint32_t kCpMask = 0x1FFFFFu;      // sym = scalar value | tile-local sentence index << 21
constexpr int kPerThread = kFastCap / kThreads;
constexpr int kWavesF = kThreads / 64;

struct FastLds {
    uint32_t sym[kFastCap + kMargin];        // decode step 1 keeps (byte pos | sentence << 16) per char here
    int32_t score[kFastCap + kMargin];       // staged text bytes during decode
    uint2 queue[kWavesF][kQCap];             // sentence-start bitmap during decode
    uint8_t typ[kFastCap + kMargin];
    uint32_t wtot[8];
};
static_assert(sizeof(FastLds) <= 20480, "8 workgroups per CU need <= 20 KB each");
static_assert((kFastCap + kMargin) * 4 % 16 == 0, "carve offsets stay 16-byte aligned");

__device__ __forceinline__ uint32_t lane_rank(uint64_t mask) {  // set bits of `mask` below this lane
    return __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u));
}

// deferred item: a = s | kind << 11 | depth << 13 ; b = slot index (kinds 0..2) or trie node (kind 3)
constexpr uint32_t kKindWalk = 3;
__device__ __forceinline__ uint2 make_item(uint32_t s, uint32_t kind, uint32_t depth, uint32_t b) {
    return make_uint2(s | (kind << 11) | (depth << 13), b);
}

struct WaveQueue {
    uint2* q;
    uint32_t n;  // wave-uniform
    __device__ __forceinline__ void push(bool pred, uint2 item) {
        const uint64_t m = __ballot(pred);
        if (m == 0) return;
        if (pred) q[n + lane_rank(m)] = item;
        n += uint32_t(__popcll(m));
    }
};

__device__ __forceinline__ void add_row6(int32_t* score, int s, const int32_t (&acc)[6]) {
#pragma unroll
    for (int j = 0; j < 6; ++j) atomicAdd(score + s - 3 + j, acc[j]);
}

// Replays up to 64 queued items with all lanes busy.
__device__ __forceinline__ void replay(const PatternTableView& T, FastLds& L, WaveQueue& Q, int lane) {
    const uint32_t take = Q.n < 64u ? Q.n : 64u;
    Q.n -= take;
    const bool have = uint32_t(lane) < take;
    uint2 it = have ? Q.q[Q.n + lane] : make_uint2(0u, 0u);
    const uint32_t s = it.x & 0x7FFu, kind = (it.x >> 11) & 3u, depth = it.x >> 13;
    bool again = false;
    uint2 next = make_uint2(0u, 0u);
    if (have && kind != kKindWalk) {
        // continue a level-(kind+1) lookup at slot it.y (probe chain past the home slot, or a non-BMP unigram)
        const uint32_t c1 = L.sym[s] & kCpMask;
        const uint32_t c2 = kind >= 1 ? (L.sym[s + 1] & kCpMask) : 0u;
        const uint32_t c3 = kind >= 2 ? (L.sym[s + 2] & kCpMask) : 0u;
        const uint64_t key = short_key(c1, c2, c3);
        uint32_t idx = it.y;
        for (;;) {
            const uint4* p = reinterpret_cast<const uint4*>(T.short_tab + size_t(idx) * 8);
            const uint4 e0 = p[0], e1 = p[1];
            if (e0.x == uint32_t(key) && (e0.y & ~kDisplacedBit) == uint32_t(key >> 32)) {
                if (kind == 0) {
                    const int32_t r[6] = {int32_t(e0.z), int32_t(e0.w), int32_t(e1.x), int32_t(e1.y), int32_t(e1.z), int32_t(e1.w)};
                    add_row6(L.score, int(s), r);
                } else if (kind == 1) {
                    const int32_t r[6] = {0, int32_t(e0.z), int32_t(e0.w), int32_t(e1.x), int32_t(e1.y), int32_t(e1.z)};
                    add_row6(L.score, int(s), r);
                } else {
                    const int32_t r[6] = {0, 0, int32_t(e0.z), int32_t(e0.w), int32_t(e1.x), int32_t(e1.y)};
                    add_row6(L.score, int(s), r);
                    if (e1.w != 0) { again = true; next = make_item(s, kKindWalk, 3, e1.w); }
                }
                break;
            }
            if ((e0.x | e0.y) == 0) break;
            idx = (idx + 1) & T.short_mask;
        }
    } else if (have) {
        // one trie step: the edge (node, sym[s + depth])
        const uint32_t c = s + depth < uint32_t(kFastCap + kMargin) ? (L.sym[s + depth] & kCpMask) : 0u;
        if (c != 0) {  // 0 = the sentence ended
            const uint64_t key = edge_key(it.y, c);
            uint32_t idx = hash_slot(key, T.edge_shift);
            bool home = true;
            for (;;) {
                const uint4 ed = reinterpret_cast<const uint4*>(T.edges)[idx];
                if (ed.x == uint32_t(key) && (ed.y & ~kDisplacedBit) == uint32_t(key >> 32)) {
                    const uint32_t m = depth + 1;  // length of the string matched so far
                    if (ed.w != kNoRow) {           // a pattern of m chars: m + 1 weights from boundary s - 1
                        const int32_t* w = T.wdata + ed.w;
                        for (uint32_t j = 0; j <= m; ++j) {
                            const int32_t v = w[j];
                            if (v != 0) atomicAdd(L.score + s - 1 + j, v);
                        }
                    }
                    if (ed.z & kHasKidsBit) { again = true; next = make_item(s, kKindWalk, m, ed.z & ~kHasKidsBit); }
                    break;
                }
                if ((ed.x | ed.y) == 0) break;
                if (home && !(ed.y & kDisplacedBit)) break;  // nothing was displaced from the home slot
                home = false;
                idx = (idx + 1) & T.edge_mask;
            }
        }
    }
    Q.push(again, next);
}

template <int WT>
__global__ __launch_bounds__(kThreads) void score_tiles_fast_kernel(const ScoreParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    FastLds& L = *reinterpret_cast<FastLds*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int pad = 3;

    const uint32_t t = blockIdx.x;
    const uint64_t i0 = P.tile_first[t], i1 = P.tile_first[t + 1];
    if (i0 >= i1) return;
    const uint64_t O0 = P.ooff[i0], O1 = P.ooff[i1];
    const uint64_t flat_len64 = uint64_t(pad) + (O1 + i1 * 4) - (O0 + i0 * 4);
    const uint64_t B0 = P.boff[i0], B1 = P.boff[i1];
    const uint8_t* tbase = P.text + B0;
    const uintptr_t a0 = reinterpret_cast<uintptr_t>(tbase) & ~uintptr_t(15);
    const uint32_t head = uint32_t(reinterpret_cast<uintptr_t>(tbase) - a0);
    const uint64_t nbytes_al64 = head + (B1 - B0);
    if (flat_len64 > kFastCap || nbytes_al64 > uint64_t(kFastCap) * 4 + 15) {  // does not fit in LDS: defer
        if (tid == 0) P.slow_list[atomicAdd(P.slow_count, 1u)] = t;
        return;
    }
    const uint32_t flat_len = uint32_t(flat_len64), nbytes_al = uint32_t(nbytes_al64);
    const uint32_t nsent = uint32_t(i1 - i0);
    const uint32_t expect_chars = uint32_t((O1 + i1) - (O0 + i0));
    const uint32_t nchunks = (nbytes_al + 15) >> 4;
    uint32_t err = 0;

    // ---------------------------------------------------------------- A. decode
    uint32_t* bitmap = reinterpret_cast<uint32_t*>(&L.queue[0][0]);
    uint32_t* raw = reinterpret_cast<uint32_t*>(&L.score[0]);
    for (uint32_t i = tid; i < ((nbytes_al + 31) >> 5) + 1; i += kThreads) bitmap[i] = 0;
    __syncthreads();
    for (uint32_t j = tid; j < nsent; j += kThreads) {
        const uint64_t b = P.boff[i0 + j], bn = P.boff[i0 + j + 1];
        if (bn <= b) err |= kErrEmptySentence;
        const uint32_t pos = head + uint32_t(b - B0);
        atomicOr(&bitmap[pos >> 5], 1u << (pos & 31));
    }
    __syncthreads();
    uint32_t base_leads = 0, base_starts = 0;
    for (uint32_t c0 = 0; c0 < nchunks; c0 += kThreads) {
        const uint32_t c = c0 + tid;
        uint32_t lm = 0, sm = 0;
        const uint32_t pos0 = c * 16;
        if (c < nchunks) {
            const uint4 v = reinterpret_cast<const uint4*>(a0)[c];
            reinterpret_cast<uint4*>(raw)[c] = v;  // stage the text for the per-char decode
            const uint32_t lo = pos0 < head ? head - pos0 : 0u;
            const uint32_t rem = nbytes_al - pos0;
            const uint32_t hi = rem < 16 ? rem : 16u;
            const uint32_t vm = ((1u << hi) - 1u) & ~((1u << lo) - 1u);
            lm = (lead_nibble(v.x) | (lead_nibble(v.y) << 4) | (lead_nibble(v.z) << 8) | (lead_nibble(v.w) << 12)) & vm;
            sm = (bitmap[pos0 >> 5] >> (pos0 & 31)) & 0xFFFFu;
        }
        const uint32_t mine = __popc(lm) | (__popc(sm) << 16);
        uint32_t incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t u = __shfl_up(incl, d);
            if (lane >= d) incl += u;
        }
        if (lane == 63) L.wtot[wave] = incl;
        __syncthreads();
        uint32_t woff = 0, total = 0;
#pragma unroll
        for (int k = 0; k < kWavesF; ++k) {
            const uint32_t u = L.wtot[k];
            if (k < wave) woff += u;
            total += u;
        }
        const uint32_t excl = woff + incl - mine;
        uint32_t ci = base_leads + (excl & 0xFFFFu);
        const uint32_t si0 = base_starts + (excl >> 16);
        base_leads += total & 0xFFFFu;
        base_starts += total >> 16;
        __syncthreads();
        uint32_t m = lm;
        while (m) {
            const uint32_t k = uint32_t(__ffs(int(m))) - 1u;
            m &= m - 1;
            const uint32_t si = si0 + uint32_t(__popc(sm & ((2u << k) - 1u))) - 1u;
            if (ci < uint32_t(kFastCap)) L.sym[ci] = (pos0 + k) | (si << 16);
            ++ci;
        }
    }
    const uint32_t nchars = base_leads;
    if (nchars != expect_chars) err |= kErrBadOffsets;
    if (tid == 0) raw[nchunks * 4] = 0;  // the dword after the staged text is read (as padding) by the last char
    __syncthreads();

    // one thread per char: decode from the staged text
    uint32_t cps[kPerThread], meta[kPerThread];  // meta = flat | type << 12 | last-of-sentence << 15 | sentence << 16
#pragma unroll
    for (int k = 0; k < kPerThread; ++k) {
        const uint32_t ci = uint32_t(tid) + uint32_t(k) * kThreads;
        cps[k] = 0; meta[k] = 0xFFFFFFFFu;
        if (ci < nchars && ci < uint32_t(kFastCap)) {
            const uint32_t info = L.sym[ci];
            const uint32_t pos = info & 0xFFFFu, si = info >> 16;
            const uint32_t nsi = (ci + 1 < nchars) ? (L.sym[ci + 1] >> 16) : 0xFFFFu;
            const uint32_t w0 = raw[pos >> 2], w1 = raw[(pos >> 2) + 1];
            const uint32_t b4 = __builtin_amdgcn_alignbyte(w1, w0, pos & 3u);
            const uint32_t cp = utf8_scalar(b4);
            if (cp == 0) err |= kErrNulChar;
            const uint32_t flat = uint32_t(pad) + ci + uint32_t(pad) * si;
            if (flat + pad < uint32_t(kFastCap + kMargin) && si < 1024u) {
                cps[k] = cp;
                meta[k] = flat | (char_type(cp) << 12) | ((nsi != si ? 1u : 0u) << 15) | (si << 16);
            } else err |= kErrBadOffsets;
        }
    }
    __syncthreads();  // every (pos, sentence) record has been read; sym and score can be reused
    for (uint32_t i = tid; i < (uint32_t(kFastCap + kMargin) * 4) / 16; i += kThreads)
        reinterpret_cast<uint4*>(L.score)[i] = make_uint4(0, 0, 0, 0);
    if (tid < pad) { L.sym[tid] = 0; L.typ[tid] = 0; }
    if (tid >= 64 && tid < 64 + kMargin) {  // slack past the tile for the s+1, s+2 look-ahead
        const uint32_t p = flat_len + uint32_t(tid - 64);
        if (p < uint32_t(kFastCap + kMargin)) { L.sym[p] = 0; L.typ[p] = 0; }
    }
#pragma unroll
    for (int k = 0; k < kPerThread; ++k) {
        if (meta[k] == 0xFFFFFFFFu) continue;
        const uint32_t flat = meta[k] & 0xFFFu;
        L.sym[flat] = cps[k] | ((meta[k] >> 16) << 21);
        L.typ[flat] = uint8_t((meta[k] >> 12) & 7u);
        if (meta[k] & 0x8000u) {
#pragma unroll
            for (int z = 1; z <= pad; ++z) { L.sym[flat + z] = 0; L.typ[flat + z] = 0; }
        }
    }
    __syncthreads();

    // ---------------------------------------------------------------- B. patterns
    const PatternTableView& T = P.ct;
    WaveQueue Q{&L.queue[wave][0], 0u};
    const uint4* uni4 = reinterpret_cast<const uint4*>(T.uni);
    const uint4* tab4 = reinterpret_cast<const uint4*>(T.short_tab);
    for (int k = 0; k < kPerThread; ++k) {
        const uint32_t s = uint32_t(tid) + uint32_t(k) * kThreads;
        if (s - uint32_t(lane) >= flat_len) break;  // wave-uniform: this wave's 64 positions are past the tile
        const bool in = s < flat_len;
        const uint32_t c1 = in ? (L.sym[s] & kCpMask) : 0u;
        const uint32_t c2 = L.sym[s + 1] & kCpMask, c3 = L.sym[s + 2] & kCpMask;
        const bool live = c1 != 0;
        const bool has2 = live && c2 != 0;
        const bool has3 = has2 && c3 != 0;
        const bool big1 = c1 >= kUniDirectChars;
        const uint64_t k1 = short_key(c1, 0, 0), k2 = short_key(c1, c2, 0), k3 = short_key(c1, c2, c3);
        const uint32_t h1 = hash_slot(k1, T.short_shift), h2 = hash_slot(k2, T.short_shift), h3 = hash_slot(k3, T.short_shift);
        // all loads first (row 0 of `uni` and whatever slot a dead lane hashes to are harmless to read)
        const uint32_t urow = big1 ? 0u : c1;
        const uint4 u0 = uni4[size_t(urow) * 2], u1 = uni4[size_t(urow) * 2 + 1];
        const uint4 b0 = tab4[size_t(h2) * 2], b1 = tab4[size_t(h2) * 2 + 1];
        const uint4 t0 = tab4[size_t(h3) * 2], t1 = tab4[size_t(h3) * 2 + 1];

        const bool hit2 = has2 && b0.x == uint32_t(k2) && (b0.y & ~kDisplacedBit) == uint32_t(k2 >> 32);
        const bool hit3 = has3 && t0.x == uint32_t(k3) && (t0.y & ~kDisplacedBit) == uint32_t(k3 >> 32);
        const bool more2 = has2 && !hit2 && (b0.y & kDisplacedBit);
        const bool more3 = has3 && !hit3 && (t0.y & kDisplacedBit);
        int32_t acc[6] = {int32_t(u0.x), int32_t(u0.y), int32_t(u0.z), int32_t(u0.w), int32_t(u1.x), int32_t(u1.y)};
        if (hit2) { acc[1] += int32_t(b0.z); acc[2] += int32_t(b0.w); acc[3] += int32_t(b1.x); acc[4] += int32_t(b1.y); acc[5] += int32_t(b1.z); }
        if (hit3) { acc[2] += int32_t(t0.z); acc[3] += int32_t(t0.w); acc[4] += int32_t(t1.x); acc[5] += int32_t(t1.y); }
        if (live) add_row6(L.score, int(s), acc);
        const uint32_t node = hit3 ? t1.w : 0u;

        if (__any(live && big1)) {
            while (Q.n > uint32_t(kQHigh)) replay(T, L, Q, lane);
            Q.push(live && big1, make_item(s, 0, 0, h1));
        }
        while (Q.n > uint32_t(kQHigh)) replay(T, L, Q, lane);
        Q.push(more2, make_item(s, 1, 0, (h2 + 1) & T.short_mask));
        while (Q.n > uint32_t(kQHigh)) replay(T, L, Q, lane);
        Q.push(more3, make_item(s, 2, 0, (h3 + 1) & T.short_mask));
        while (Q.n > uint32_t(kQHigh)) replay(T, L, Q, lane);
        Q.push(node != 0, make_item(s, kKindWalk, 3, node));
    }
    while (Q.n > 0) replay(T, L, Q, lane);
    __syncthreads();

    // ---------------------------------------------------------------- C. boundaries
    for (uint32_t p = uint32_t(pad) + tid; p + 1 < flat_len; p += kThreads) {
        const uint32_t x = L.sym[p];
        if ((x & kCpMask) == 0 || (L.sym[p + 1] & kCpMask) == 0) continue;
        int32_t y = P.bias + L.score[p];
        if (WT > 0) {
            uint32_t id = 0;  // window t[b-W+1 .. b+W], 3 bits each (boundary_scorer_cache.rs:59-81)
#pragma unroll
            for (int i = 1 - WT; i <= WT; ++i) id = (id << 3) | L.typ[int(p) + i];
            y += P.type_table[id];
        }
        const uint32_t si = x >> 21;
        const uint64_t o = O0 + (p - uint32_t(pad)) - uint64_t(pad + 1) * si;
        if (P.scores) P.scores[o] = y;
        if (P.labels) P.labels[o] = y > 0 ? 1 : 0;
    }
    if (err) atomicOr(P.status, err);
}

}  // namespace

bool fast_path_supported(const ScoreParams& P) {
    const PatternTableView& T = P.ct;
    if (!T.present || T.window != 3 || T.stride_dw != 8 || T.uni_dw != 8 || T.uni_n != kUniDirectChars || T.ext_slot != 5) return false;
    if (P.pad != 3) return false;
    if (P.type_kind == kTypeNone) return true;
    return P.type_kind == kTypeWindowTable && P.type_window >= 1 && P.type_window <= 3;
}

hipError_t launch_score_tiles_fast(const ScoreParams& P, uint32_t n_tiles, hipStream_t stream) {
    const size_t lds = sizeof(FastLds);
    const int wt = P.type_kind == kTypeWindowTable ? P.type_window : 0;
    switch (wt) {
        case 0: hipLaunchKernelGGL(score_tiles_fast_kernel<0>, dim3(n_tiles), dim3(kThreads), lds, stream, P); break;
        case 1: hipLaunchKernelGGL(score_tiles_fast_kernel<1>, dim3(n_tiles), dim3(kThreads), lds, stream, P); break;
        case 2: hipLaunchKernelGGL(score_tiles_fast_kernel<2>, dim3(n_tiles), dim3(kThreads), lds, stream, P); break;
        case 3: hipLaunchKernelGGL(score_tiles_fast_kernel<3>, dim3(n_tiles), dim3(kThreads), lds, stream, P); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace vpt
