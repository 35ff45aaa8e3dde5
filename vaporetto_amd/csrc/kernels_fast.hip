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

#include <cstdlib>

#include "device_common.h"
#include "kernels.hpp"

namespace vpt {
namespace {

constexpr int kQCap = 256;                   // deferred items per wave (both stacks together)
constexpr uint32_t kQHigh = kQCap - 128;     // replay until one more iteration (<= 64 + 64 pushes) fits
constexpr uint32_t kCpMask = 0xFFFFu;        // sym = char (non-BMP -> 0xFFFF) | tile-local sentence index << 16
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
//   trie step      x = s | depth << 11          y = parent id (layout.h: tri slot, or kPackedEdgeId | edge slot)
//   continuation   x = s | levels << 11 (bit 1: bigram, bit 2: trigram lookup to be continued past its home slot)
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

__device__ __forceinline__ int32_t lo16(uint32_t x) { return int32_t(x << 16) >> 16; }
__device__ __forceinline__ int32_t hi16(uint32_t x) { return int32_t(x) >> 16; }

__device__ __forceinline__ void add_row6(int32_t* score, uint32_t s, int32_t a0, int32_t a1, int32_t a2, int32_t a3,
                                         int32_t a4, int32_t a5) {
    int32_t* p = score + s - 3;
    atomicAdd(p, a0); atomicAdd(p + 1, a1); atomicAdd(p + 2, a2);
    atomicAdd(p + 3, a3); atomicAdd(p + 4, a4); atomicAdd(p + 5, a5);
}

// Up to 64 queued trie steps, all lanes busy: the edge (parent, sym[s + depth]).
__device__ __forceinline__ void replay_walk(const PackedView& K, FastLds& L, WaveStacks& Q, int lane) {
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
        const uint4* tab = reinterpret_cast<const uint4*>(K.edge);
        uint32_t b = packed_hash2(it.y, c, K.edge_shift);
        bool home = true;
        for (;;) {
            const uint4 e = tab[b];
            if (e.x == it.y && (e.y & 0xFFFFu) == c) {
                const uint32_t m = depth + 1;  // chars matched so far: a pattern of m chars has m + 1 weights,
                if (e.y & (kPkHasRow << 16)) {  // the first on boundary s - 1
                    const uint4* wr = reinterpret_cast<const uint4*>(K.wrows) + e.z;
                    int32_t* dst = L.score + s - 1;
                    if (e.y & (kPkWide << 16)) {  // a value outside i16: the row is stored as i32 (rare)
                        const int32_t* w32 = reinterpret_cast<const int32_t*>(wr);
                        for (uint32_t j = 0; j <= m; ++j) atomicAdd(dst + j, w32[j]);
                        if (e.y & (kPkHasKids << 16)) { again = true; nx = s | (m << 11); ny = kPackedEdgeId | b; }
                        break;
                    }
                    const uint4 r = wr[0];
                    atomicAdd(dst, lo16(r.x)); atomicAdd(dst + 1, hi16(r.x)); atomicAdd(dst + 2, lo16(r.y));
                    atomicAdd(dst + 3, hi16(r.y)); atomicAdd(dst + 4, lo16(r.z));  // m >= 4
                    if (m >= 5) atomicAdd(dst + 5, hi16(r.z));
                    if (m >= 6) atomicAdd(dst + 6, lo16(r.w));
                    if (m >= 7) atomicAdd(dst + 7, hi16(r.w));
                    for (uint32_t j0 = 8; j0 <= m; j0 += 8) {
                        const uint4 q = wr[j0 >> 3];
                        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                        for (uint32_t j = 0; j < 8; ++j)
                            if (j0 + j <= m) atomicAdd(dst + j0 + j, (j & 1) ? hi16(w[j >> 1]) : lo16(w[j >> 1]));
                    }
                }
                if (e.y & (kPkHasKids << 16)) { again = true; nx = s | (m << 11); ny = kPackedEdgeId | b; }
                break;
            }
            if (e.y == 0 || (home && !(e.y & (kPkDisp << 16)))) break;
            home = false;
            b = (b + 1) & K.edge_mask;
        }
    }
    Q.push_walk(again, nx, ny);
}

// Row of a <= 3-char string in the GENERAL short table (layout.h; 32-byte entries, buckets of two): used for the
// rare packed slots marked kPkWide.
__device__ __forceinline__ bool general_row(const PatternTableView& T, uint64_t key, uint4& r0, uint4& r1) {
    const uint4* tab = reinterpret_cast<const uint4*>(T.short_tab);
    const uint32_t klo = uint32_t(key), khi = uint32_t(key >> 32);
    uint32_t b = hash_slot(key, T.short_shift);
    bool home = true;
    for (;;) {
        const uint4 a0 = tab[size_t(b) * 4], b0 = tab[size_t(b) * 4 + 2];
        if (a0.x == klo && (a0.y & ~kDisplacedBit) == khi) { r0 = a0; r1 = tab[size_t(b) * 4 + 1]; return true; }
        if (b0.x == klo && b0.y == khi) { r0 = b0; r1 = tab[size_t(b) * 4 + 3]; return true; }
        if ((a0.x | a0.y) == 0 || (b0.x | b0.y) == 0 || (home && !(a0.y & kDisplacedBit))) return false;
        home = false;
        b = (b + 1) & T.short_mask;
    }
}

// Up to 64 queued lookup continuations for start position s.  levels: 2 / 4 = the bigram / trigram lookup goes on
// past its home slot; 1 / 8 / 16 = the unigram / bigram / trigram row is kPkWide and comes from the general tables.
__device__ __forceinline__ void replay_retry(const PackedView& K, const PatternTableView& T, FastLds& L, WaveStacks& Q, int lane) {
    const uint32_t take = Q.nr < 64u ? Q.nr : 64u;
    Q.nr -= take;
    const bool have = uint32_t(lane) < take;
    const uint2 it = have ? Q.q[kQCap - 1 - (Q.nr + lane)] : make_uint2(0u, 0u);
    const uint32_t s = it.x & 0x7FFu;
    uint32_t levels = have ? (it.x >> 11) & 31u : 0u;
    bool walk = false;
    uint32_t parent = 0;
    const uint32_t c1 = L.sym[s] & kCpMask, c2 = L.sym[s + 1] & kCpMask, c3 = L.sym[s + 2] & kCpMask;
    const uint32_t kb = c1 | (c2 << 16);
    if (levels & 2u) {
        const uint4* tab = reinterpret_cast<const uint4*>(K.bi);
        uint32_t b = packed_hash1(kb, K.bi_shift);
        for (;;) {
            b = (b + 1) & K.bi_mask;
            const uint4 e = tab[b];
            if (e.x == kb) {
                if (e.w & (kPkWide << 16)) levels |= 8u;
                else add_row6(L.score, s, 0, lo16(e.y), hi16(e.y), lo16(e.z), hi16(e.z), lo16(e.w));
                break;
            }
            if (e.x == 0) break;
        }
    }
    if (levels & 4u) {
        const uint4* tab = reinterpret_cast<const uint4*>(K.tri);
        uint32_t b = packed_hash2(kb, c3, K.tri_shift);
        for (;;) {
            b = (b + 1) & K.tri_mask;
            const uint4 e = tab[b];
            if (e.x == kb && (e.y & 0xFFFFu) == c3) {
                if (e.y & (kPkWide << 16)) levels |= 16u;
                else add_row6(L.score, s, 0, 0, lo16(e.z), hi16(e.z), lo16(e.w), hi16(e.w));
                if (e.y & (kPkHasKids << 16)) { walk = true; parent = b; }
                break;
            }
            if (e.x == 0) break;
        }
    }
    if (levels & 25u) {  // wide rows (rare): i32 rows of the general tables
        uint4 r0, r1;
        if (levels & 1u) {
            const uint4* u = reinterpret_cast<const uint4*>(T.uni) + size_t(c1) * 2;
            r0 = u[0]; r1 = u[1];
            add_row6(L.score, s, int32_t(r0.x), int32_t(r0.y), int32_t(r0.z), int32_t(r0.w), int32_t(r1.x), int32_t(r1.y));
        }
        if ((levels & 8u) && general_row(T, short_key(c1, c2, 0), r0, r1))
            add_row6(L.score, s, 0, int32_t(r0.z), int32_t(r0.w), int32_t(r1.x), int32_t(r1.y), int32_t(r1.z));
        if ((levels & 16u) && general_row(T, short_key(c1, c2, c3), r0, r1))
            add_row6(L.score, s, 0, 0, int32_t(r0.z), int32_t(r0.w), int32_t(r1.x), int32_t(r1.y));
    }
    Q.push_walk(walk, s | (3u << 11), parent);
}

__device__ __forceinline__ void make_room(const PackedView& K, const PatternTableView& T, FastLds& L, WaveStacks& Q, int lane) {
    while (Q.nw + Q.nr > kQHigh) {
        if (Q.nw >= Q.nr) replay_walk(K, L, Q, lane);
        else replay_retry(K, T, L, Q, lane);
    }
}

// optional phase timing (VPT_PROFILE_PHASES): wave 0 of every workgroup adds the shader cycles it spent per phase
__device__ __forceinline__ uint64_t phase_mark(uint64_t* prof, int slot, uint64_t t_prev) {
    if (!prof) return 0;
    const uint64_t now = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) atomicAdd(reinterpret_cast<unsigned long long*>(prof + slot), (unsigned long long)(now - t_prev));
    return now;
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
    uint64_t* const prof = P.prof;
    uint64_t tmark = prof ? __builtin_amdgcn_s_memtime() : 0;

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
    tmark = phase_mark(prof, 0, tmark);
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
        L.sym[flat] = (cp < kPackedNoMatchSym ? cp : kPackedNoMatchSym) | (meta[k] & 0xFFFF0000u);
        L.typ[flat] = uint8_t(ty);
        if (meta[k] & 0x8000u) {
#pragma unroll
            for (uint32_t z = 1; z <= pad; ++z) { L.sym[flat + z] = 0; L.typ[flat + z] = 0; }
        }
    }
    __syncthreads();

    tmark = phase_mark(prof, 1, tmark);
    // ---------------------------------------------------------------- B. patterns
    const PackedView& K = P.pk;
    WaveStacks Q{&L.queue[wave][0], 0u, 0u};
    const uint4* uni4 = reinterpret_cast<const uint4*>(K.uni);
    const uint4* bi4 = reinterpret_cast<const uint4*>(K.bi);
    const uint4* tri4 = reinterpret_cast<const uint4*>(K.tri);
    for (int k = 0; k < kPerThread; ++k) {
        const uint32_t s = uint32_t(tid) + uint32_t(k) * kThreads;
        if (s - uint32_t(lane) >= flat_len) break;  // wave-uniform: this wave's 64 positions are past the tile
        const uint32_t c1 = s < flat_len ? (L.sym[s] & kCpMask) : 0u;
        const uint32_t c2 = L.sym[s + 1] & kCpMask, c3 = L.sym[s + 2] & kCpMask;
        const bool live = c1 != 0;
        const bool has2 = live && c2 != 0;
        const bool has3 = has2 && c3 != 0;
        const uint32_t kb = c1 | (c2 << 16);
        uint32_t hb = packed_hash1(kb, K.bi_shift), ht = packed_hash2(kb, c3, K.tri_shift);
        if (P.debug) {  // timing ablations (VPT_DEBUG_ABLATE; results are wrong): pin a lookup to slot 0
            if (P.debug & 1u) hb = 0;
            if (P.debug & 2u) ht = 0;
        }
        // the three loads first (row 0 of `uni` and whatever slot a dead lane hashes to are harmless to read)
        const uint4 u = uni4[(P.debug & 4u) ? 0u : c1];
        const uint4 eb = bi4[hb];
        const uint4 et = tri4[ht];
        const bool mb = has2 && eb.x == kb;
        const bool mt = has3 && et.x == kb && (et.y & 0xFFFFu) == c3;
        const bool moreb = has2 && !mb && (eb.w & (kPkDisp << 16));
        const bool moret = has3 && !mt && (et.y & (kPkDisp << 16));
        const uint32_t b1 = mb ? eb.y : 0u, b2 = mb ? eb.z : 0u, b3 = mb ? eb.w : 0u;
        const uint32_t t2 = mt ? et.z : 0u, t3 = mt ? et.w : 0u;
        const int32_t a0 = lo16(u.x);
        const int32_t a1 = hi16(u.x) + lo16(b1);
        const int32_t a2 = lo16(u.y) + hi16(b1) + lo16(t2);
        const int32_t a3 = hi16(u.y) + lo16(b2) + hi16(t2);
        const int32_t a4 = lo16(u.z) + hi16(b2) + lo16(t3);
        const int32_t a5 = hi16(u.z) + lo16(b3) + hi16(t3);
        if (live) add_row6(L.score, s, a0, a1, a2, a3, a4, a5);

        make_room(K, P.ct, L, Q, lane);
        Q.push_walk(mt && (et.y & (kPkHasKids << 16)) && !(P.debug & 8u), s | (3u << 11), ht);
        const uint32_t levels = ((live && u.w != 0) ? 1u : 0u) | (moreb ? 2u : 0u) | (moret ? 4u : 0u) |
                                ((mb && (eb.w & (kPkWide << 16))) ? 8u : 0u) | ((mt && (et.y & (kPkWide << 16))) ? 16u : 0u);
        Q.push_retry(levels != 0, s | (levels << 11), 0u);
    }
    while (Q.nr > 0) replay_retry(K, P.ct, L, Q, lane);
    while (Q.nw > 0) replay_walk(K, L, Q, lane);
    tmark = phase_mark(prof, 2, tmark);
    __syncthreads();
    tmark = phase_mark(prof, 3, tmark);

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
        const uint32_t si = x >> 16;
        const uint64_t o = O0 + (p - pad) - uint64_t(pad + 1) * si;
        if (P.scores) P.scores[o] = y;
        if (P.labels) P.labels[o] = y > 0 ? 1 : 0;
    }
    if (err) atomicOr(P.status, err);
    phase_mark(prof, 4, tmark);
}

}  // namespace

bool fast_path_supported(const ScoreParams& P) {
    if (!P.pk.present || P.pad != 3 || !P.ctype) return false;
    if (!P.ct.present || P.ct.stride_dw != 8 || P.ct.uni_dw != 8 || P.ct.uni_n != kUniDirectChars) return false;  // kPkWide rows
    if (P.type_kind == kTypeNone) return true;
    return P.type_kind == kTypeWindowTable && P.type_window >= 1 && P.type_window <= 3;
}

hipError_t launch_score_tiles_fast(const ScoreParams& P, uint32_t n_tiles, hipStream_t stream) {
    size_t lds = sizeof(FastLds);
    if (const char* padv = std::getenv("VPT_DEBUG_LDS_PAD")) lds += size_t(std::atoi(padv));  // occupancy experiments
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
