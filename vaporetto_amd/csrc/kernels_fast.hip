// Specialised tile kernel for the model geometry every distributed Vaporetto / KyTea model has:
// char window 3, char n-grams of at most 3 chars (+ dictionary words of any length), type scores from the
// 8^(2W) window table (W <= 3) or none.  Same algorithm and tables as kernels.hip (which stays the general
// path); what changes is how the work is scheduled on a CDNA4 wave:
//
//   * geometry is compile-time: level rows are 6/5/4 slots at boundary offsets -3/-2/-1, one bucket = 64 bytes
//     = four dwordx4 loads, so the per-start-position code is branch-free: the unigram row, the bigram bucket
//     and the trigram bucket are all requested before anything is compared, the three rows are summed in
//     registers and land in the LDS score array with six ds_add_u32;
//   * everything data-dependent -- a lookup that must continue past its home bucket (kDisplacedBit), a non-BMP
//     unigram, a dictionary word longer than 3 chars walking the trie -- is NOT done in place (64 lanes would
//     wait for the unluckiest one): it is pushed, ballot/mbcnt-compacted, onto a wave-private LDS stack and
//     replayed 64 items at a time with every lane busy; a trie step that matches re-queues its continuation.
//     Trie steps and lookup continuations use separate stacks, so a replay has no per-lane kind divergence;
//   * UTF-8 decode is two-step: a chunk scan finds (byte position, sentence) of every char, then one thread
//     per CHAR decodes from the LDS-staged text (branch-free) and classifies it with a 64 KB table;
//   * 22 KB of LDS per workgroup and <= 64 VGPRs: 7 workgroups = 28 waves per CU.
#include <hip/hip_runtime.h>

#include "device_common.h"
#include "kernels.hpp"

namespace vpt {
namespace {

constexpr int kQCap = 256;                   // deferred items per wave (both stacks together)
constexpr uint32_t kQHigh = kQCap - 128;     // replay until one more iteration (<= 64 + 64 pushes) fits
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
static_assert(sizeof(FastLds) <= 23400, "7 workgroups per CU need <= 22.8 KB each");
static_assert((kFastCap + kMargin) * 4 % 16 == 0, "carve offsets stay 16-byte aligned");

__device__ __forceinline__ uint32_t lane_rank(uint64_t mask) {  // set bits of `mask` below this lane
    return __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u));
}

// branch-free UTF-8 -> scalar value; b4 = the lead byte and the three bytes after it, little-endian
__device__ __forceinline__ uint32_t utf8_scalar_bf(uint32_t b4) {
    const uint32_t b0 = b4 & 0xFF, b1 = (b4 >> 8) & 0x3F, b2 = (b4 >> 16) & 0x3F, b3 = (b4 >> 24) & 0x3F;
    const uint32_t c2 = ((b0 & 0x1F) << 6) | b1;
    const uint32_t c3 = ((b0 & 0x0F) << 12) | (b1 << 6) | b2;
    const uint32_t c4 = ((b0 & 0x07) << 18) | (b1 << 12) | (b2 << 6) | b3;
    uint32_t cp = b0;
    cp = b0 >= 0xC0 ? c2 : cp;
    cp = b0 >= 0xE0 ? c3 : cp;
    cp = b0 >= 0xF0 ? c4 : cp;
    return cp;
}

// Two stacks in one wave-private buffer: trie steps grow from the bottom, lookup continuations from the top.
//   trie step      x = s | depth << 11          y = node
//   continuation   x = s | levels << 11 (bit 0: non-BMP unigram, 1: bigram, 2: trigram lookups still open)
struct WaveStacks {
    uint2* q;
    uint32_t nw, nr;  // wave-uniform counts
    __device__ __forceinline__ void push_walk(bool pred, uint32_t x, uint32_t y) {
        const uint64_t m = __ballot(pred);
        if (m == 0) return;
        if (pred) q[nw + lane_rank(m)] = make_uint2(x, y);
        nw += uint32_t(__popcll(m));
    }
    __device__ __forceinline__ void push_retry(bool pred, uint32_t x, uint32_t y) {
        const uint64_t m = __ballot(pred);
        if (m == 0) return;
        if (pred) q[kQCap - 1 - (nr + lane_rank(m))] = make_uint2(x, y);
        nr += uint32_t(__popcll(m));
    }
};

__device__ __forceinline__ void add_row6(int32_t* score, uint32_t s, int32_t a0, int32_t a1, int32_t a2, int32_t a3,
                                         int32_t a4, int32_t a5) {
    int32_t* p = score + s - 3;
    atomicAdd(p, a0); atomicAdd(p + 1, a1); atomicAdd(p + 2, a2);
    atomicAdd(p + 3, a3); atomicAdd(p + 4, a4); atomicAdd(p + 5, a5);
}

// Up to 64 queued trie steps, all lanes busy: the edge (node, sym[s + depth]).
__device__ __forceinline__ void replay_walk(const PatternTableView& T, FastLds& L, WaveStacks& Q, int lane) {
    const uint32_t take = Q.nw < 64u ? Q.nw : 64u;
    Q.nw -= take;
    const bool have = uint32_t(lane) < take;
    const uint2 it = have ? Q.q[Q.nw + lane] : make_uint2(0u, 0u);
    const uint32_t s = it.x & 0x7FFu, depth = it.x >> 11;
    const uint32_t at = s + depth;
    const uint32_t c = (have && at < uint32_t(kFastCap + kMargin)) ? (L.sym[at] & kCpMask) : 0u;  // 0: sentence over
    bool again = false;
    uint32_t nx = 0, ny = 0;
    if (c != 0) {
        const uint64_t key = edge_key(it.y, c);
        const uint32_t klo = uint32_t(key), khi = uint32_t(key >> 32);
        uint32_t b = hash_slot(key, T.edge_shift);
        bool home = true;
        for (;;) {
            const uint4* p = reinterpret_cast<const uint4*>(T.edges) + size_t(b) * kEdgeBucket;
            const uint4 e0 = p[0], e1 = p[1], e2 = p[2], e3 = p[3];
            const bool m0 = e0.x == klo && (e0.y & ~kDisplacedBit) == khi;
            const bool m1 = e1.x == klo && e1.y == khi, m2 = e2.x == klo && e2.y == khi, m3 = e3.x == klo && e3.y == khi;
            if (m0 || m1 || m2 || m3) {
                const uint32_t child = m0 ? e0.z : m1 ? e1.z : m2 ? e2.z : e3.z;
                const uint32_t woff = m0 ? e0.w : m1 ? e1.w : m2 ? e2.w : e3.w;
                const uint32_t m = depth + 1;  // chars matched so far: a pattern of m chars has m + 1 weights
                if (woff != kNoRow) {           // starting at boundary s - 1
                    const int32_t* w = T.wdata + woff;
                    int32_t* dst = L.score + s - 1;
                    const int32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], w4 = w[4];  // m >= 4
                    atomicAdd(dst, w0); atomicAdd(dst + 1, w1); atomicAdd(dst + 2, w2); atomicAdd(dst + 3, w3); atomicAdd(dst + 4, w4);
                    for (uint32_t j = 5; j <= m; ++j) atomicAdd(dst + j, w[j]);
                }
                if (child & kHasKidsBit) { again = true; nx = s | (m << 11); ny = child & ~kHasKidsBit; }
                break;
            }
            const bool free_slot = (e0.x | e0.y) == 0 || (e1.x | e1.y) == 0 || (e2.x | e2.y) == 0 || (e3.x | e3.y) == 0;
            if (free_slot || (home && !(e0.y & kDisplacedBit))) break;
            home = false;
            b = (b + 1) & T.edge_mask;
        }
    }
    Q.push_walk(again, nx, ny);
}

// Up to 64 queued lookup continuations: for the start position s, the flagged levels still have to be looked
// up past the home bucket (a key displaced by a full bucket), or at all (non-BMP unigram: no direct row).
__device__ __forceinline__ void replay_retry(const PatternTableView& T, FastLds& L, WaveStacks& Q, int lane) {
    const uint32_t take = Q.nr < 64u ? Q.nr : 64u;
    Q.nr -= take;
    const bool have = uint32_t(lane) < take;
    const uint2 it = have ? Q.q[kQCap - 1 - (Q.nr + lane)] : make_uint2(0u, 0u);
    const uint32_t s = it.x & 0x7FFu;
    uint32_t levels = have ? (it.x >> 11) & 7u : 0u;
    bool walk = false;
    uint32_t node = 0;
    const uint32_t c1 = L.sym[s] & kCpMask, c2 = L.sym[s + 1] & kCpMask, c3 = L.sym[s + 2] & kCpMask;
    while (levels) {
        const uint32_t level = uint32_t(__ffs(int(levels))) - 1u;  // 0..2 = 1..3 chars
        levels &= levels - 1;
        const uint64_t key = short_key(c1, level >= 1 ? c2 : 0u, level >= 2 ? c3 : 0u);
        const uint32_t klo = uint32_t(key), khi = uint32_t(key >> 32);
        uint32_t b = hash_slot(key, T.short_shift);
        if (level != 0) b = (b + 1) & T.short_mask;  // the home bucket was already examined
        for (;;) {
            const uint4* p = reinterpret_cast<const uint4*>(T.short_tab) + size_t(b) * 4;
            const uint4 a0 = p[0], a1 = p[1], b0 = p[2], b1 = p[3];
            const bool ma = a0.x == klo && (a0.y & ~kDisplacedBit) == khi, mb = b0.x == klo && b0.y == khi;
            if (ma || mb) {
                const int32_t r0 = int32_t(ma ? a0.z : b0.z), r1 = int32_t(ma ? a0.w : b0.w), r2 = int32_t(ma ? a1.x : b1.x);
                const int32_t r3 = int32_t(ma ? a1.y : b1.y), r4 = int32_t(ma ? a1.z : b1.z), r5 = int32_t(ma ? a1.w : b1.w);
                if (level == 0) add_row6(L.score, s, r0, r1, r2, r3, r4, r5);
                else if (level == 1) add_row6(L.score, s, 0, r0, r1, r2, r3, r4);
                else {
                    add_row6(L.score, s, 0, 0, r0, r1, r2, r3);
                    if (r5 != 0) { walk = true; node = uint32_t(r5); }
                }
                break;
            }
            if ((a0.x | a0.y) == 0 || (b0.x | b0.y) == 0) break;  // a bucket with a free slot ends the chain
            b = (b + 1) & T.short_mask;
        }
    }
    Q.push_walk(walk, s | (3u << 11), node);
}

__device__ __forceinline__ void make_room(const PatternTableView& T, FastLds& L, WaveStacks& Q, int lane) {
    while (Q.nw + Q.nr > kQHigh) {
        if (Q.nw >= Q.nr) replay_walk(T, L, Q, lane);
        else replay_retry(T, L, Q, lane);
    }
}

template <int WT>
__global__ __launch_bounds__(kThreads) void score_tiles_fast_kernel(const ScoreParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    FastLds& L = *reinterpret_cast<FastLds*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr uint32_t pad = 3;

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
    uint32_t cps[kPerThread], meta[kPerThread];  // meta = flat | last-of-sentence << 15 | sentence << 16
#pragma unroll
    for (int k = 0; k < kPerThread; ++k) {
        const uint32_t ci = uint32_t(tid) + uint32_t(k) * kThreads;
        const bool ok = ci < nchars && ci < uint32_t(kFastCap);
        const uint32_t info = L.sym[ok ? ci : 0u];
        const uint32_t ninfo = L.sym[ok ? ci + 1 : 0u];
        const uint32_t pos = info & 0xFFFFu, si = info >> 16;
        const uint32_t nsi = (ci + 1 < nchars) ? (ninfo >> 16) : 0xFFFFu;
        const uint32_t pw = (pos >> 2) < uint32_t(kFastCap + kMargin - 1) ? (pos >> 2) : 0u;
        const uint32_t w0 = raw[pw], w1 = raw[pw + 1];
        const uint32_t cp = utf8_scalar_bf(__builtin_amdgcn_alignbyte(w1, w0, pos & 3u));
        const uint32_t flat = pad + ci + pad * si;
        const bool fits = flat + pad < uint32_t(kFastCap + kMargin) && si < 1024u;
        if (ok && cp == 0) err |= kErrNulChar;
        if (ok && !fits) err |= kErrBadOffsets;
        cps[k] = cp;
        meta[k] = (ok && fits) ? (flat | ((nsi != si ? 1u : 0u) << 15) | (si << 16)) : 0xFFFFFFFFu;
    }
    __syncthreads();  // every (pos, sentence) record has been read; sym and score can be reused
    for (uint32_t i = tid; i < (uint32_t(kFastCap + kMargin) * 4) / 16; i += kThreads)
        reinterpret_cast<uint4*>(L.score)[i] = make_uint4(0, 0, 0, 0);
    if (tid < int(pad)) { L.sym[tid] = 0; L.typ[tid] = 0; }
    if (tid >= 64 && tid < 64 + kMargin) {  // slack past the tile for the s+1, s+2 look-ahead
        const uint32_t p = flat_len + uint32_t(tid - 64);
        if (p < uint32_t(kFastCap + kMargin)) { L.sym[p] = 0; L.typ[p] = 0; }
    }
#pragma unroll
    for (int k = 0; k < kPerThread; ++k) {
        if (meta[k] == 0xFFFFFFFFu) continue;
        const uint32_t flat = meta[k] & 0xFFFu, cp = cps[k];
        const uint32_t ty = cp < 0x10000u ? uint32_t(P.ctype[cp]) : char_type(cp);
        L.sym[flat] = cp | ((meta[k] >> 16) << 21);
        L.typ[flat] = uint8_t(ty);
        if (meta[k] & 0x8000u) {
#pragma unroll
            for (uint32_t z = 1; z <= pad; ++z) { L.sym[flat + z] = 0; L.typ[flat + z] = 0; }
        }
    }
    __syncthreads();

    // ---------------------------------------------------------------- B. patterns
    const PatternTableView& T = P.ct;
    WaveStacks Q{&L.queue[wave][0], 0u, 0u};
    const uint4* uni4 = reinterpret_cast<const uint4*>(T.uni);
    const uint4* tab4 = reinterpret_cast<const uint4*>(T.short_tab);
    for (int k = 0; k < kPerThread; ++k) {
#ifdef VPT_ABLATE_NO_PATTERNS
        break;
#endif
        const uint32_t s = uint32_t(tid) + uint32_t(k) * kThreads;
        if (s - uint32_t(lane) >= flat_len) break;  // wave-uniform: this wave's 64 positions are past the tile
        const uint32_t c1 = s < flat_len ? (L.sym[s] & kCpMask) : 0u;
        const uint32_t c2 = L.sym[s + 1] & kCpMask, c3 = L.sym[s + 2] & kCpMask;
        const bool live = c1 != 0;
        const bool has2 = live && c2 != 0;
        const bool has3 = has2 && c3 != 0;
        const bool big1 = c1 >= kUniDirectChars;
        const uint64_t k2 = short_key(c1, c2, 0), k3 = short_key(c1, c2, c3);
        const uint32_t h2 = hash_slot(k2, T.short_shift), h3 = hash_slot(k3, T.short_shift);
        // every load first (row 0 of `uni` and whatever bucket a dead lane hashes to are harmless to read)
        const uint32_t urow = big1 ? 0u : c1;
        const uint4 u0 = uni4[size_t(urow) * 2], u1 = uni4[size_t(urow) * 2 + 1];
        const uint4* pb = tab4 + size_t(h2) * 4;
        const uint4* pt = tab4 + size_t(h3) * 4;
        const uint4 ba0 = pb[0], ba1 = pb[1], bb0 = pb[2], bb1 = pb[3];
        const uint4 ta0 = pt[0], ta1 = pt[1], tb0 = pt[2], tb1 = pt[3];

        const uint32_t k2lo = uint32_t(k2), k2hi = uint32_t(k2 >> 32), k3lo = uint32_t(k3), k3hi = uint32_t(k3 >> 32);
        const bool m2a = has2 && ba0.x == k2lo && (ba0.y & ~kDisplacedBit) == k2hi;
        const bool m2b = has2 && bb0.x == k2lo && bb0.y == k2hi;
        const bool m3a = has3 && ta0.x == k3lo && (ta0.y & ~kDisplacedBit) == k3hi;
        const bool m3b = has3 && tb0.x == k3lo && tb0.y == k3hi;
        const bool more2 = has2 && !m2a && !m2b && (ba0.y & kDisplacedBit);
        const bool more3 = has3 && !m3a && !m3b && (ta0.y & kDisplacedBit);
        int32_t a0 = int32_t(u0.x), a1 = int32_t(u0.y), a2 = int32_t(u0.z), a3 = int32_t(u0.w), a4 = int32_t(u1.x), a5 = int32_t(u1.y);
        a1 += int32_t(m2a ? ba0.z : m2b ? bb0.z : 0u);
        a2 += int32_t(m2a ? ba0.w : m2b ? bb0.w : 0u) + int32_t(m3a ? ta0.z : m3b ? tb0.z : 0u);
        a3 += int32_t(m2a ? ba1.x : m2b ? bb1.x : 0u) + int32_t(m3a ? ta0.w : m3b ? tb0.w : 0u);
        a4 += int32_t(m2a ? ba1.y : m2b ? bb1.y : 0u) + int32_t(m3a ? ta1.x : m3b ? tb1.x : 0u);
        a5 += int32_t(m2a ? ba1.z : m2b ? bb1.z : 0u) + int32_t(m3a ? ta1.y : m3b ? tb1.y : 0u);
        if (live) add_row6(L.score, s, a0, a1, a2, a3, a4, a5);
        const uint32_t node = m3a ? ta1.w : m3b ? tb1.w : 0u;

        make_room(T, L, Q, lane);
        Q.push_walk(node != 0, s | (3u << 11), node);
        const uint32_t levels = (live && big1 ? 1u : 0u) | (more2 ? 2u : 0u) | (more3 ? 4u : 0u);
        Q.push_retry(levels != 0, s | (levels << 11), 0u);
    }
    while (Q.nr > 0) replay_retry(T, L, Q, lane);
    while (Q.nw > 0) replay_walk(T, L, Q, lane);
    __syncthreads();

    // ---------------------------------------------------------------- C. boundaries
    for (uint32_t p = pad + uint32_t(tid); p + 1 < flat_len; p += kThreads) {
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
        const uint64_t o = O0 + (p - pad) - uint64_t(pad + 1) * si;
        if (P.scores) P.scores[o] = y;
        if (P.labels) P.labels[o] = y > 0 ? 1 : 0;
    }
    if (err) atomicOr(P.status, err);
}

}  // namespace

bool fast_path_supported(const ScoreParams& P) {
    const PatternTableView& T = P.ct;
    if (!T.present || T.window != 3 || T.stride_dw != 8 || T.uni_dw != 8 || T.uni_n != kUniDirectChars || T.ext_slot != 5) return false;
    if (P.pad != 3 || !P.ctype) return false;
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
