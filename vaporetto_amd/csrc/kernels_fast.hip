// Specialised tile kernel for the model geometry every distributed Vaporetto / KyTea model has: char window 3,
// BMP patterns (layout.h, "PACKED TABLES"), type scores from type rows in LDS, the 8^(2W) window table (W <= 3) or
// none.  Same all-matches algorithm as kernels.hip (which stays the general path); what changes is how the work is
// laid out for a CDNA4 CU, whose limiter on this workload is the vector L1's address pipeline: every 16-byte load of
// a lane costs a slot of it and every distinct line it touches two more (profiles/r02_*), while VALU work is two
// orders of magnitude cheaper per lane.  So a start position issues as few loads as the data structure allows:
//
//   * the patterns form a DOUBLE-ARRAY trie over their first three symbols: one 16-byte unigram node (indexed by the
//     char's id), one 32-byte bigram node at (unigram base + id of the next char), and -- only when the bigram's 64-bit
//     filter admits the third char -- one 16-byte trigram node at (bigram base + its id): 4 loads to 3 lines, no hash,
//     no seed table, no probing; a node names its parent, so a lookup that lands on a foreign node knows it;
//   * the three loads depend on each other, so they are software-pipelined: a trip of the main loop loads the
//     unigram nodes of the positions two trips ahead, the bigram nodes of the next trip's positions and the trigram
//     nodes of its own -- all independent of each other -- and waits once;
//   * rows are added to the LDS score array where they arrive (ds_add_u32: integer => order-free => bit-exact);
//   * what is data-dependent beyond depth 3 is NOT done in place (64 lanes would wait for the unluckiest one): it is
//     pushed, ballot/mbcnt-compacted, onto wave-private LDS stacks, so that a replay runs one short code path with
//     every lane busy: W trie steps of dictionary words longer than 3 chars (a step that matches re-queues its
//     continuation), M the rare rows with a value outside their fields (taken from the general tables);
//   * UTF-8 decode: the text is staged in LDS, a chunk scan numbers chars and sentences, the thread that scanned a
//     chunk decodes its chars (branch-free) straight into their flat positions, a second pass maps them to ids;
//   * 24 KB of LDS and at most 80 VGPRs per workgroup: 6 workgroups per CU.
#include <hip/hip_runtime.h>

#include <cstddef>

#include "device_common.h"
#include "kernels.hpp"

namespace vpt {
namespace {

constexpr int kQCap = 128;                   // W items per wave
constexpr uint32_t kQHigh = kQCap - 64;      // replay until one more round of pushes (<= 64) fits
constexpr int kMCap = 64;                    // M items per wave: one round of pushes (<= 64) always fits an empty stack
constexpr uint32_t kCpMask = 0xFFFFu;        // sym = id (kNoId: in no pattern) | type << 16 | tile-local sentence << 19 | linebreak << 29
constexpr uint32_t kSymLinebreak = 1u << 29;
constexpr int kWavesF = kThreads / 64;
constexpr int kTypeRows = 4;                 // TM value: type rows in LDS (1..3 = window table of that W, 0 = none)
static_assert(kMargin >= int(kPackedMaxSkip), "replay_w reads up to kPackedMaxSkip symbols past a char");
constexpr int kTrowCount = int(kTypeRowCount);   // layout.h, type_row_index

template <int CAP>
struct FastLdsT {
    uint32_t sym[CAP + kMargin];             // zero except for the tile's chars (scalar value | sentence << 21 until classified)
    int32_t score[CAP + kMargin];            // staged text bytes during decode
    uint2 queue[kWavesF][kQCap];             // sentence-start bitmap during decode
    uint32_t mqueue[kWavesF][kMCap];
    uint32_t wtot[8];
    union {                                  // never needed together; the launch allocates the one in use
        uint8_t typ[CAP + kMargin];          // window-table modes
        uint4 trow[kTrowCount];              // TM == kTypeRows
    };
};
// the replay routines only index the arrays: they see the layout through the largest geometry's type (sym and score start
// every geometry; the queues are passed as pointers)
struct FastLds {
    uint32_t* sym;
    int32_t* score;
};
template <int CAP>
constexpr bool fast_lds_ok(int wg_per_cu) {
    return offsetof(FastLdsT<CAP>, typ) % 16 == 0 && (CAP + kMargin) * 4 % 16 == 0 &&                    // carve offsets stay 16-byte aligned
           sizeof(uint2) * kWavesF * kQCap >= size_t(CAP) * 4 / 8 + 16 &&                                 // the decode phase's sentence-start bitmap lives in the W queues
           offsetof(FastLdsT<CAP>, typ) + sizeof(uint4) * kTrowCount <= size_t(128 / wg_per_cu) * 1280;   // gfx950 hands out LDS in 1280-byte granules, 128 per CU
}
static_assert(fast_lds_ok<kFastCapSmall>(kFastWgSmall) && fast_lds_ok<kFastCapLarge>(kFastWgLarge), "LDS budget of the two tile geometries");

__device__ __forceinline__ uint32_t lane_rank(uint64_t mask) {  // set bits of `mask` below this lane
    return __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u));
}

// 16 bytes at base + byte offset (32-bit): one scalar base for all packed arrays
__device__ __forceinline__ uint4 ld16(const unsigned char* base, uint32_t byte_off) {
    return *reinterpret_cast<const uint4*>(base + byte_off);
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

// Wave-private stacks of deferred work (wave-uniform counts):
//   W  x = s | depth << 11   y = mini-table ref (in `deep`) of the children to search      trie step
//   M  s | kinds << 11                                                                     a row outside its fields
constexpr uint32_t kWideUni = 4u, kWideBi = 8u, kWideTri = 16u;
struct WaveStacks {
    uint2* q;
    uint32_t* mq;
    uint32_t nw, nm;
    __device__ __forceinline__ void push_w(bool pred, uint32_t x, uint32_t y) {
        const uint64_t m = __ballot(pred);
        if (m == 0) return;
        if (pred) q[nw + lane_rank(m)] = make_uint2(x, y);
        nw = wave_uniform(nw + uint32_t(__popcll(m)));
    }
    __device__ __forceinline__ void push_m(bool pred, uint32_t x) {
        const uint64_t m = __ballot(pred);
        if (m == 0) return;
        if (pred) mq[nm + lane_rank(m)] = x;
        nm += uint32_t(__popcll(m));
    }
};

__device__ __forceinline__ int32_t sext(uint32_t x, int bits) { return int32_t(x << (32 - bits)) >> (32 - bits); }   // low `bits` bits, signed
__device__ __forceinline__ int32_t lo16(uint32_t x) { return int32_t(x << 16) >> 16; }
__device__ __forceinline__ int32_t hi16(uint32_t x) { return int32_t(x) >> 16; }

__device__ __forceinline__ void add_row6(int32_t* score, uint32_t s, int32_t a0, int32_t a1, int32_t a2, int32_t a3,
                                         int32_t a4, int32_t a5) {
    int32_t* p = score + s - 3;
    atomicAdd(p, a0); atomicAdd(p + 1, a1); atomicAdd(p + 2, a2);
    atomicAdd(p + 3, a3); atomicAdd(p + 4, a4); atomicAdd(p + 5, a5);
}
// the row of a 3-char string starting at `st`: boundaries st-1 .. st+2
__device__ __forceinline__ void add_child(int32_t* score, uint32_t st, uint32_t w01, uint32_t w23) {
    int32_t* p = score + st - 1;
    atomicAdd(p, lo16(w01)); atomicAdd(p + 1, hi16(w01)); atomicAdd(p + 2, lo16(w23)); atomicAdd(p + 3, hi16(w23));
}

// W: up to 64 queued trie steps, all lanes busy: the child sym[s + depth] in the mini-table `ref` of `deep`, whose
// 64-byte entries also name the up-to-8 symbols that must follow (compressed single-child chains; layout.h).
// The home entry and the next one are read together (a mini-table keeps a quarter of its entries free, so nearly
// every search ends within two); the rare longer search loops.
__device__ __forceinline__ void replay_w(const PackedView& K, FastLds& L, WaveStacks& Q, int lane) {
    VPT_WAVE_LOCKSTEP();   // the queue entries were written by other lanes
    const uint32_t take = wave_uniform(Q.nw < 64u ? Q.nw : 64u);   // opaque: nw - min(nw, 64) would become a VALU-only saturating subtract
    Q.nw = wave_uniform(Q.nw - take);
    const bool have = uint32_t(lane) < take;
    const uint2 it = have ? Q.q[Q.nw + lane] : make_uint2(0u, 0u);
    const uint32_t s = it.x & 0x7FFu, depth = it.x >> 11;
    const uint32_t at = s + depth;
    // at <= flat_len: the steps before this one matched chars, and the array is zero from the end of the tile on
    const uint32_t c = have ? (L.sym[at] & kCpMask) : 0u;  // 0: sentence over
    const uint32_t tab = K.off_deep + ((it.y >> 5) << 6);   // byte offset of the mini-table (64-byte entries)
    const uint32_t last = (1u << (it.y & 31u)) - 1u;
    const uint32_t i0 = packed_mini_slot(c, it.y), i1 = (i0 + 1) & last;
    const uint4 ea = ld16(K.base, tab + (i0 << 6)), eb = ld16(K.base, tab + (i1 << 6));
    const bool ma = c != 0 && (ea.x & 0xFFFFu) == c;
    const bool mb = c != 0 && !ma && ea.x != 0 && (eb.x & 0xFFFFu) == c;   // last == 0: eb is ea again, no match
    bool found = ma || mb;
    uint4 e = ma ? ea : eb;
    uint32_t idx = ma ? i0 : i1;
    bool open = c != 0 && !found && ea.x != 0 && eb.x != 0 && last > 1u;
    if (__ballot(open) != 0) {  // rare
        uint32_t i = (i1 + 1) & last, n = 2;
        while (__ballot(open) != 0) {
            if (open) {
                e = ld16(K.base, tab + (i << 6));
                found = (e.x & 0xFFFFu) == c;
                idx = i;
                open = !found && e.x != 0 && n < last;
                i = (i + 1) & last;
                ++n;
            }
        }
    }
    const uint32_t ent = tab + (idx << 6);
    // the symbols that must follow (none for most entries of a dense trie, several for a long rare word)
    const uint32_t nskip = found ? (e.x >> 24) & 15u : 0u;
    if (__ballot(nskip != 0) != 0) {
        uint4 e1 = make_uint4(0, 0, 0, 0);
        if (__ballot(nskip > 2) != 0) { if (nskip > 2) e1 = ld16(K.base, ent + 16); }
        const uint32_t sk[4] = {e.z, e.w, e1.x, e1.y};
#pragma unroll
        for (uint32_t j = 0; j < kPackedMaxSkip; ++j) {
            if (__ballot(j < nskip) == 0) break;
            const uint32_t q = at + 1 + j;   // < flat_len + kPackedMaxSkip <= kFastCap + kMargin (found: sym[at] is a char)
            const uint32_t have_c = j < nskip ? (L.sym[q] & kCpMask) : 0u;
            const uint32_t want = (sk[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
            if (j < nskip && have_c != want) found = false;
        }
    }
    const uint32_t m = depth + 1 + nskip;  // chars matched so far: a pattern of m chars has m + 1 weights, the first
    int32_t* dst = L.score + s - 1;        // on boundary s - 1; unused row slots hold zero, adding them is harmless
    const bool row = found && (e.x & (kPkHasRow << 16));
    if (__ballot(row) != 0) {
        uint4 f0 = make_uint4(0, 0, 0, 0), f1 = make_uint4(0, 0, 0, 0);
        if (row) f0 = ld16(K.base, ent + 32);   // (loading the home entry's row speculatively with the entry: no faster, profiles/r02_c5_ab*.jsonl)
        if (__ballot(row && m >= 8) != 0) { if (row && m >= 8) f1 = ld16(K.base, ent + 48); }
        if (row) {
            atomicAdd(dst, lo16(f0.x)); atomicAdd(dst + 1, hi16(f0.x)); atomicAdd(dst + 2, lo16(f0.y)); atomicAdd(dst + 3, hi16(f0.y));
            atomicAdd(dst + 4, lo16(f0.z)); atomicAdd(dst + 5, hi16(f0.z)); atomicAdd(dst + 6, lo16(f0.w)); atomicAdd(dst + 7, hi16(f0.w));
        }
        if (__ballot(row && m >= 8) != 0) {
            if (row && m >= 8) {
                atomicAdd(dst + 8, lo16(f1.x)); atomicAdd(dst + 9, hi16(f1.x)); atomicAdd(dst + 10, lo16(f1.y));
                atomicAdd(dst + 11, hi16(f1.y)); atomicAdd(dst + 12, lo16(f1.z)); atomicAdd(dst + 13, hi16(f1.z));
            }
        }
    }
    if (__ballot(found && (e.x & (kPkExtRow << 16))) != 0) {  // more than 14 weights or a value outside i16 (rare)
        if (found && (e.x & (kPkExtRow << 16))) {
            const int32_t* w32 = reinterpret_cast<const int32_t*>(K.base + K.off_xrows) + ld16(K.base, ent + 32).x;
            for (uint32_t j = 0; j <= m; ++j) atomicAdd(dst + j, w32[j]);
        }
    }
    Q.push_w(found && e.y != 0, s | (m << 11), e.y);
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

// rows with a value outside their fields (rare): the general tables hold them as i32, keyed by code points
__device__ __forceinline__ void add_wide_rows(const PackedView& K, const PatternTableView& T, FastLds& L, uint32_t kinds, uint32_t s) {
    const uint32_t* cpid = reinterpret_cast<const uint32_t*>(K.base + K.off_cpid);
    // a wide row belongs to a pattern that matched here: its chars are in the alphabet (ids below n_uni)
    const uint32_t last = K.n_uni - 1u;
    const uint32_t i1 = L.sym[s] & kCpMask, i2 = L.sym[s + 1] & kCpMask, i3 = L.sym[s + 2] & kCpMask;
    const uint32_t c1 = cpid[i1 < last ? i1 : last], c2 = cpid[i2 < last ? i2 : last], c3 = cpid[i3 < last ? i3 : last];
    uint4 r0, r1;
    if (kinds & kWideUni) {
        const uint4* u = reinterpret_cast<const uint4*>(T.uni) + size_t(c1) * 2;
        r0 = u[0]; r1 = u[1];
        add_row6(L.score, s, int32_t(r0.x), int32_t(r0.y), int32_t(r0.z), int32_t(r0.w), int32_t(r1.x), int32_t(r1.y));
    }
    if ((kinds & kWideBi) && general_row(T, short_key(c1, c2, 0), r0, r1))
        add_row6(L.score, s, 0, int32_t(r0.z), int32_t(r0.w), int32_t(r1.x), int32_t(r1.y), int32_t(r1.z));
    if ((kinds & kWideTri) && general_row(T, short_key(c1, c2, c3), r0, r1))
        add_row6(L.score, s, 0, 0, int32_t(r0.z), int32_t(r0.w), int32_t(r1.x), int32_t(r1.y));
}

// M: up to 64 queued rows with a value outside their fields -- taken from the general tables (i32).
__device__ __forceinline__ void replay_m(const PackedView& K, const PatternTableView& T, FastLds& L, WaveStacks& Q, int lane) {
    VPT_WAVE_LOCKSTEP();   // the queue entries were written by other lanes
    const uint32_t take = wave_uniform(Q.nm < 64u ? Q.nm : 64u);
    Q.nm -= take;
    if (uint32_t(lane) < take) {
        const uint32_t it = Q.mq[Q.nm + lane];
        add_wide_rows(K, T, L, it >> 11, it & 0x7FFu);
    }
}

// W replays until at most `mark` items are left (a W item becomes at most one W item, so the loop ends).
__device__ __forceinline__ void drain_w(const PackedView& K, FastLds& L, WaveStacks& Q, int lane, uint32_t mark) {
    while (Q.nw > mark) replay_w(K, L, Q, lane);
}

// optional phase timing (VPT_PROFILE_PHASES): wave 0 of every workgroup adds the shader cycles it spent per phase
__device__ __forceinline__ uint64_t phase_mark(uint64_t* prof, int slot, uint64_t t_prev) {
    if (!prof) return 0;
    const uint64_t now = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) atomicAdd(reinterpret_cast<unsigned long long*>(prof + slot), (unsigned long long)(now - t_prev));
    return now;
}

// DBG: the diagnostics build (VPT_DEBUG_ABLATE timing ablations, VPT_PROFILE_PHASES) -- compiled out of the kernel
// production launches use.
template <int TM, bool DBG, int CAP, int WG>
__global__ __launch_bounds__(kThreads, WG) void score_tiles_fast_kernel(const ScoreParams P_in) {
    ScoreParams P = P_in;
    if (!DBG) { P.debug = 0; P.prof = nullptr; }
    VPT_DYNAMIC_LDS(smem);
    FastLdsT<CAP>& M = *reinterpret_cast<FastLdsT<CAP>*>(smem);
    FastLds L{M.sym, M.score};
    constexpr int kFastCap = CAP, kPerThread = CAP / kThreads;
    const int tid = threadIdx.x, lane = tid & 63, wave = int(wave_uniform(uint32_t(tid) >> 6));
    const uint32_t wbase = uint32_t(wave) << 6;   // this wave's first thread, as a scalar
    constexpr uint32_t pad = 3;

    const uint32_t t = blockIdx.x;
    uint64_t i0, i1;
    if (P.tile_first) {
        i0 = P.tile_first[t]; i1 = P.tile_first[t + 1];
    } else {   // no assign_tiles_kernel in front: lanes 0 and 1 find the tile's two ends (a few dependent, L2-hot loads)
        uint64_t mine = 0;
        if (tid < 2) {
            const uint32_t tt = t + uint32_t(tid);
            mine = tt >= P.n_tiles ? P.n_sent : first_sentence_at(P.ooff, P.n_sent, uint64_t(1 + pad), uint64_t(tt) * P.tile_flat);
        }
        if (tid < 2) M.wtot[tid * 2] = uint32_t(mine), M.wtot[tid * 2 + 1] = uint32_t(mine >> 32);
        __syncthreads();
        i0 = uint64_t(wave_uniform(M.wtot[0])) | (uint64_t(wave_uniform(M.wtot[1])) << 32);
        i1 = uint64_t(wave_uniform(M.wtot[2])) | (uint64_t(wave_uniform(M.wtot[3])) << 32);
        __syncthreads();   // wtot is reused by the scan
    }
    if (i0 >= i1) return;
    const uint64_t O0 = P.ooff[i0], O1 = P.ooff[i1];
    const uint64_t flat_len64 = uint64_t(pad) + (O1 + i1 * 4) - (O0 + i0 * 4);
    const uint64_t B0 = P.boff[i0], B1 = P.boff[i1];
    const uint8_t* tbase = P.text + B0;
    const uintptr_t a0 = reinterpret_cast<uintptr_t>(tbase) & ~uintptr_t(15);
    const uint32_t head = uint32_t(reinterpret_cast<uintptr_t>(tbase) - a0);
    const uint64_t nbytes_al64 = head + (B1 - B0);
    if (flat_len64 > kFastCap || nbytes_al64 > uint64_t(kFastCap) * 4 + 15) {  // does not fit in LDS: defer
        if (tid == 0) {
            if (P.scratch_cap == 0) atomicOr(P.status, kErrScratchTooSmall);   // the caller's length bounds were understated
            else P.slow_list[atomicAdd(P.slow_count, 1u)] = t;
        }
        return;
    }
    const uint32_t flat_len = uint32_t(flat_len64), nbytes_al = uint32_t(nbytes_al64);
    const uint32_t nsent = uint32_t(i1 - i0);
    const uint32_t expect_chars = uint32_t((O1 + i1) - (O0 + i0));
    const uint32_t nchunks = (nbytes_al + 15) >> 4;
    uint32_t err = 0;
    if (TM == kTypeRows) {  // 512 type rows -> LDS (not aliased by the decode scratch; barriers follow before use)
        for (uint32_t i = tid; i < uint32_t(kTrowCount); i += kThreads) M.trow[i] = ld16(P.pk.base, P.pk.off_trow + (i << 4));
    }
    uint64_t* const prof = P.prof;
    uint64_t tmark = prof ? __builtin_amdgcn_s_memtime() : 0;

    // ---------------------------------------------------------------- A. decode
    // Flat layout: char g of the tile's sentence j sits at pad + g + pad * j, everything else is zero (separators,
    // the slack past the tile).  The text is staged in LDS (the score array is free until phase B), a chunk scan
    // numbers the chars and the sentences, and the thread that scanned a chunk decodes its chars straight into
    // their flat positions; a second pass classifies them (one cache-hot table read each, all in flight together).
    uint32_t* bitmap = reinterpret_cast<uint32_t*>(&M.queue[0][0]);   // one bit per text byte: a sentence starts here
    uint32_t* raw = reinterpret_cast<uint32_t*>(&L.score[0]);
    for (uint32_t i = tid; i < ((nbytes_al + 31) >> 5) + 1; i += kThreads) bitmap[i] = 0;
    for (uint32_t i = tid; i < (uint32_t(kFastCap + kMargin) * 4) / 16; i += kThreads) reinterpret_cast<uint4*>(L.sym)[i] = make_uint4(0, 0, 0, 0);
    if (TM != kTypeRows) {
        for (uint32_t i = tid; i < uint32_t(kFastCap + kMargin) / 4; i += kThreads) reinterpret_cast<uint32_t*>(M.typ)[i] = 0;
    }
    for (uint32_t c = tid; c < nchunks; c += kThreads) reinterpret_cast<uint4*>(raw)[c] = reinterpret_cast<const uint4*>(a0)[c];   // (non-temporal loads / stores here measured 1-2 % slower: profiles/r02_c1_ab.jsonl, r02_c3_ab.jsonl)
    if (tid == 0) raw[nchunks * 4] = 0;  // the dword after the staged text is read (as padding) by the last char
    __syncthreads();
    for (uint32_t j = tid; j < nsent; j += kThreads) {
        const uint64_t b = P.boff[i0 + j], bn = P.boff[i0 + j + 1];
        if (bn <= b) err |= kErrEmptySentence;
        const uint32_t pos = head + uint32_t(b - B0);
        atomicOr(&bitmap[pos >> 5], 1u << (pos & 31));
    }
    __syncthreads();
    uint32_t base_leads = 0, base_starts = 0;
    uint32_t min_cp = 0xFFFFFFFFu, max_si = 0, max_flat = 0;   // over this thread's chars: NUL / a char outside the tile
    for (uint32_t c0 = 0; c0 < nchunks; c0 += kThreads) {   // one pass for up to 4 KB of tile text, else two
        const uint32_t c = c0 + tid;
        uint32_t lm = 0, sm = 0;
        const uint32_t pos0 = c * 16;
        if (c < nchunks) {
            const uint4 v = reinterpret_cast<const uint4*>(raw)[c];
            const uint32_t lo = pos0 < head ? head - pos0 : 0u;
            const uint32_t rem = nbytes_al - pos0;
            const uint32_t hi = rem < 16 ? rem : 16u;
            const uint32_t vm = ((1u << hi) - 1u) & ~((1u << lo) - 1u);
            lm = (lead_nibble(v.x) | (lead_nibble(v.y) << 4) | (lead_nibble(v.z) << 8) | (lead_nibble(v.w) << 12)) & vm;
            sm = (bitmap[pos0 >> 5] >> (pos0 & 31)) & 0xFFFFu;
        }
        const uint32_t mine = __popc(lm) | (__popc(sm) << 16);
        const uint32_t incl = wave_inclusive_scan(mine);
        if (lane == 63) M.wtot[wave] = incl;
        __syncthreads();
        uint32_t woff = 0, total = 0;
#pragma unroll
        for (int k = 0; k < kWavesF; ++k) {
            const uint32_t u = wave_uniform(M.wtot[k]);
            if (k < wave) woff += u;
            total += u;
        }
        const uint32_t excl = woff + incl - mine;
        const uint32_t ci = base_leads + (excl & 0xFFFFu);
        const uint32_t si0 = base_starts + (excl >> 16);
        base_leads += total & 0xFFFFu;
        base_starts += total >> 16;
        __syncthreads();   // wtot is rewritten by the next pass
        uint32_t m = lm;
        const uint32_t sib = si0 - 1u;                        // sentence of a char = sib + starts up to it in this chunk
        uint32_t fb = pad + ci + sib + (sib << 1);            // flat = pad + char index + pad * si (mod 2^32: sib may be -1)
        while (m) {
            const uint32_t k = uint32_t(__ffs(int(m))) - 1u;
            m &= m - 1;
            const uint32_t r = uint32_t(__popc(sm & ((2u << k) - 1u)));
            const uint32_t si = sib + r;                      // 0xFFFFFFFF before the first start
            const uint32_t pos = pos0 + k;
            const uint32_t cp = utf8_scalar_bf(__builtin_amdgcn_alignbyte(raw[(pos >> 2) + 1], raw[pos >> 2], pos & 3u));
            const uint32_t flat = fb + __umul24(r, pad);
            ++fb;
            static_assert(pad == 3, "fb adds 3 * sib");
            min_cp = cp < min_cp ? cp : min_cp;
            max_si = si > max_si ? si : max_si;
            max_flat = flat > max_flat ? flat : max_flat;
            if (si < 1024u && flat + pad < uint32_t(kFastCap + kMargin)) L.sym[flat] = cp | (si << 21);   // 21 + 10 bits; classified below
        }
    }
    const uint32_t nchars = base_leads;
    if (nchars != expect_chars) err |= kErrBadOffsets;
    if (min_cp == 0) err |= kErrNulChar;
    if (max_si >= 1024u || max_flat + pad >= uint32_t(kFastCap + kMargin)) err |= kErrBadOffsets;   // a char that did not fit
    tmark = phase_mark(prof, 0, tmark);
    __syncthreads();  // the staged text has been read: the score array can be zeroed; every char is in place
    for (uint32_t i = tid; i < (uint32_t(kFastCap + kMargin) * 4) / 16; i += kThreads) reinterpret_cast<uint4*>(L.score)[i] = make_uint4(0, 0, 0, 0);
    {   // classify: table reads first (all in flight together), then the final symbols
        uint32_t xs[kPerThread], info[kPerThread];
#pragma unroll
        for (int k = 0; k < kPerThread; ++k) {
            xs[k] = L.sym[uint32_t(tid) + uint32_t(k) * kThreads];   // zero past the tile
            const uint32_t cp = xs[k] & 0x1FFFFFu, idx = cp < 0x10000u ? cp : 0u;
            // id of the char it is scored as | CharacterType << 16 | linebreak << 19: one word of a 256 KB table (plain, or --
            // with VPT_FLAG_KYTEA_FULLWIDTH -- the one that looks through KyteaFullwidthFilter)
            if (DBG && (P.debug & 64u)) info[k] = (cp & 7u) == 0 ? P.cid[idx] : ((cp * 2654435761u) >> 20) | (3u << 16);   // timing ablation: one gather in eight
            else info[k] = P.cid[idx];
        }
#pragma unroll
        for (int k = 0; k < kPerThread; ++k) {
            const uint32_t cp = xs[k] & 0x1FFFFFu;
            uint32_t v = (info[k] & 0x7FFFFu) | ((xs[k] >> 21) << 19) | ((info[k] & kCinfoLinebreak) ? kSymLinebreak : 0u);
            if (__ballot(cp >= 0x10000u) != 0) {   // rare: outside the BMP nothing is tabulated (and nothing can match)
                if (cp >= 0x10000u) v = kNoId | (char_type(cp) << 16) | ((xs[k] >> 21) << 19);
            }
            v = cp != 0 ? v : 0u;                  // a separator (or NUL, which has raised kErrNulChar)
            L.sym[uint32_t(tid) + uint32_t(k) * kThreads] = v;
            if (P.cps_out) {   // wave-uniform: a fill_tags call on this batch follows and wants the chars decoded (it then skips its own pass)
                const uint32_t flat = uint32_t(tid) + uint32_t(k) * kThreads, si = xs[k] >> 21;
                const uint32_t scored = (P.cinfo && cp < 0x10000u) ? (P.cinfo[cp] & 0xFFFFu) : cp;   // through KyteaFullwidthFilter when that flag is on
                if (cp != 0 && flat >= pad * (si + 1u) && flat - pad * (si + 1u) < expect_chars) {
                    const uint64_t at = (O0 + i0) + (flat - pad * (si + 1u));
                    if (at < P.total_chars) P.cps_out[at] = scored | (((v >> 16) & 7u) << 24);
                    else err |= kErrBadOffsets;   // out_offsets that run past the total the caller stated
                }
            }
            if (TM != kTypeRows) M.typ[uint32_t(tid) + uint32_t(k) * kThreads] = uint8_t((v >> 16) & 7u);
        }
    }
    __syncthreads();

    tmark = phase_mark(prof, 1, tmark);
    // ---------------------------------------------------------------- B. patterns
    // One lane per start position, three trips in flight: trip j loads the TRIGRAM nodes of its own positions
    // (tid + 256 j), the BIGRAM nodes of the next trip's and the UNIGRAM nodes of the positions two trips ahead -- loads
    // that do not depend on each other -- waits once, adds the rows that arrived to the LDS score array and computes the
    // addresses the next trip needs (child slot = base in the parent + id of the next char; layout.h).
    const PackedView& K = P.pk;
    WaveStacks Q{&M.queue[wave][0], &M.mqueue[wave][0], 0u, 0u};
    const uint32_t uni_last = K.n_uni - 1u;
    const uint32_t off_bi = K.off_bi & ~255u, off_tri = K.off_tri & ~255u;   // they ARE 256-byte aligned (capi.cpp); now the compiler knows
    uint32_t b_slot = 0, b_key = 0, b_id3 = 0;   // next trip's bigram stage: node slot, its key (0: no bigram starts there), id of the third char
    uint32_t t_slot = ~0u, t_par = 0;            // this trip's trigram stage: node slot (~0: none) and parent slot + 1
    for (int j = -2; j < kPerThread; ++j) {
        if (P.debug & 16u) break;  // timing ablation: no pattern phase at all
        // wave-uniform: which stages have positions of this wave inside the tile
        const bool do_t = j >= 0 && wbase + uint32_t(j) * kThreads < flat_len;
        const bool do_b = j + 1 >= 0 && j + 1 < kPerThread && wbase + uint32_t(j + 1) * kThreads < flat_len;
        const bool do_u = j + 2 < kPerThread && wbase + uint32_t(j + 2) * kThreads < flat_len;
        if (!do_t && !do_b && !do_u) break;
        const uint32_t s_t = uint32_t(tid) + uint32_t(j) * kThreads, s_b = s_t + kThreads, s_u = s_b + kThreads;
        // ---- every load first
        uint4 tn = make_uint4(0, 0, 0, 0), n0 = make_uint4(0, 0, 0, 0), n1 = make_uint4(0, 0, 0, 0), u = make_uint4(0, 0, 0, 0);
        uint32_t x1 = 0, x2 = 0, x3 = 0;
        if (do_t) { if (t_slot != ~0u) tn = ld16(K.base, off_tri + (((P.debug & 1u) ? 0u : t_slot) << 4)); }
        if (do_b) {
            const uint32_t a = off_bi + (((P.debug & 1u) ? 0u : b_slot) << 5);
            n0 = ld16(K.base, a); n1 = ld16(K.base, a | 16u);
        }
        if (do_u) {
            x1 = L.sym[s_u]; x2 = L.sym[s_u + 1]; x3 = L.sym[s_u + 2];   // the array is zero past the tile; s_u + 2 < kFastCap + kMargin
            const uint32_t id1 = x1 & kCpMask;
            if (DBG && (P.debug & 128u)) { if ((id1 & 3u) == 0) u = ld16(K.base, K.off_uni + ((id1 < uni_last ? id1 : uni_last) << 4)); }   // timing ablation: one unigram node in four
            else u = ld16(K.base, K.off_uni + (((P.debug & 4u) ? 0u : (id1 < uni_last ? id1 : uni_last)) << 4));
        }
        // ---- trigram stage of positions s_t: the node is ours if it names our bigram node as its parent
        if (do_t) {
            const bool hit = t_slot != ~0u && (tn.x & kTriParentMask) == t_par;
            if (hit) add_child(L.score, s_t, tn.y, tn.z);   // a kPkWide node holds zero weights
            const bool wide = hit && (tn.x & (kPkWide << 24));
            uint32_t kids = hit ? tn.w : 0u;
            VPT_PIN(kids);
            drain_w(K, L, Q, lane, kQHigh);                 // room for one more round of pushes
            Q.push_w(kids != 0 && !(P.debug & 8u), s_t | (3u << 11), kids);
            const uint64_t mm = __ballot(wide);
            if (mm != 0) {
                while (Q.nm + uint32_t(__popcll(mm)) > uint32_t(kMCap)) replay_m(K, P.ct, L, Q, lane);
                Q.push_m(wide, s_t | (kWideTri << 11));
            }
        }
        // ---- bigram stage of positions s_b: key check, the row, and the address of the trigram node
        t_slot = ~0u;
        if (do_b) {
            const bool keyok = b_key != 0 && n0.x == b_key;
            if (keyok) {
                // five 19-bit fields at bits 0, 19, 38, 57, 76 of dwords 1..3 (layout.h); bit 95 = the row is wide (M stack)
                int32_t* p = L.score + s_b - 2;
                atomicAdd(p, sext(n0.y, kBiFieldBits));
                atomicAdd(p + 1, sext(__builtin_amdgcn_alignbit(n0.z, n0.y, 19), kBiFieldBits));
                atomicAdd(p + 2, sext(n0.z >> 6, kBiFieldBits));
                atomicAdd(p + 3, sext(__builtin_amdgcn_alignbit(n0.w, n0.z, 25), kBiFieldBits));
                atomicAdd(p + 4, sext(n0.w >> 12, kBiFieldBits));
            }
            const uint32_t bit = packed_filter_bit(b_id3);
            const bool cont = keyok && b_id3 != 0 && (((bit < 32 ? n1.y >> bit : n1.z >> (bit - 32)) & 1u) != 0);
            const uint32_t ts = n1.x + b_id3;               // modulo 2^32 (layout.h); a false positive of the filter may point anywhere
            t_slot = (cont && ts < K.n_tri && !(P.debug & 2u)) ? ts : ~0u;
            t_par = b_slot + 1u;
            const uint64_t mm = __ballot(keyok && (n0.w & kBiWideBit));
            if (mm != 0) {
                while (Q.nm + uint32_t(__popcll(mm)) > uint32_t(kMCap)) replay_m(K, P.ct, L, Q, lane);
                Q.push_m(keyok && (n0.w & kBiWideBit), s_b | (kWideBi << 11));
            }
        }
        // ---- unigram stage of positions s_u: the row (+ the type row), and the address of the bigram node
        b_key = 0;
        if (do_u) {
            const uint32_t id1 = x1 & kCpMask, id2 = x2 & kCpMask;
            const bool live = id1 != 0;
            // six 18-bit fields at bits 0, 18, 36, 54, 72, 90 (layout.h); bits 108..126 = the base of the bigram nodes; bit 127 = wide
            int32_t a0 = sext(u.x, kUniFieldBits), a1 = sext(__builtin_amdgcn_alignbit(u.y, u.x, 18), kUniFieldBits);
            int32_t a2 = sext(u.y >> 4, kUniFieldBits), a3 = sext(__builtin_amdgcn_alignbit(u.z, u.y, 22), kUniFieldBits);
            int32_t a4 = sext(u.z >> 8, kUniFieldBits), a5 = sext(__builtin_amdgcn_alignbit(u.w, u.z, 26), kUniFieldBits);
            if (TM == kTypeRows) {
                // a dead lane (t1 = 0) reads the 16 bytes in front of the rows; nothing is added for it
                const uint4 tr = M.trow[int32_t(type_row_index((x1 >> 16) & 7u, (x2 >> 16) & 7u, (x3 >> 16) & 7u))];
                // six 18-bit signed fields at bits 0, 18, 36, 54, 72, 90 (layout.h, trow_field)
                a0 += int32_t(tr.x << 14) >> 14;
                a1 += int32_t(__builtin_amdgcn_alignbit(tr.y, tr.x, 18) << 14) >> 14;
                a2 += int32_t(tr.y << 10) >> 14;
                a3 += int32_t(__builtin_amdgcn_alignbit(tr.z, tr.y, 22) << 14) >> 14;
                a4 += int32_t(tr.z << 6) >> 14;
                a5 += int32_t(__builtin_amdgcn_alignbit(tr.w, tr.z, 26) << 14) >> 14;
            }
            if (live) add_row6(L.score, s_u, a0, a1, a2, a3, a4, a5);
            b_slot = (((u.w >> kUniBaseShift) & kUniBaseMask) << K.bi_shift) + id2;
            b_key = (live && id2 != 0) ? (id1 | (id2 << 16)) : 0u;
            b_id3 = x3 & kCpMask;
            const uint64_t mm = __ballot(live && (u.w & kUniWideBit) && !(P.debug & 32u));
            if (mm != 0) {
                while (Q.nm + uint32_t(__popcll(mm)) > uint32_t(kMCap)) replay_m(K, P.ct, L, Q, lane);
                Q.push_m(live && (u.w & kUniWideBit), s_u | (kWideUni << 11));
            }
        }
    }
    while (Q.nm > 0) replay_m(K, P.ct, L, Q, lane);
    while (Q.nw > 0) replay_w(K, L, Q, lane);
    tmark = phase_mark(prof, 2, tmark);
    __syncthreads();
    tmark = phase_mark(prof, 3, tmark);

    // ---------------------------------------------------------------- C. boundaries
    // boundary p lies between flat positions p and p + 1, both chars of one sentence; its output index is the
    // tile's first boundary + (p - pad) - (pad + 1) * (sentence in tile): 32-bit offsets from scalar bases
    const uint32_t nb = uint32_t(O1 - O0);
    int32_t* const sc = P.scores ? P.scores + O0 : nullptr;
    uint8_t* const lb = P.labels ? P.labels + O0 : nullptr;
#pragma unroll
    for (int k = 0; k < kPerThread; ++k) {
        if (wbase + uint32_t(k) * kThreads + pad + 1 >= flat_len) break;   // wave-uniform
        const uint32_t p = pad + uint32_t(tid) + uint32_t(k) * kThreads;   // p + 1 < kFastCap + kMargin; zero past the tile
        const uint32_t x = L.sym[p], x2 = L.sym[p + 1];
        int32_t y = P.bias + L.score[p];
        if ((x & kCpMask) == 0 || (x2 & kCpMask) == 0) continue;
        if (TM >= 1 && TM <= 3) {
            uint32_t id = 0;  // window t[b-W+1 .. b+W], 3 bits each (boundary_scorer_cache.rs:59-81)
#pragma unroll
            for (int i = 1 - TM; i <= TM; ++i) id = (id << 3) | (M.typ[int(p) + i] & 7u);
            y += P.type_table[id];
        }
        const uint32_t o = (p - pad) - (pad + 1) * ((x >> 19) & 1023u);
        if (o >= nb) { err |= kErrBadOffsets; continue; }  // only with offsets that do not match the text
        if (sc) sc[o] = y;
        if (lb) {
            uint32_t label = y > 0 ? 1u : 0u;
            if (P.post) {   // wave-uniform: KyteaWsConstFilter / SplitLinebreaksFilter on the label
                const uint32_t t1 = (x >> 16) & 7u, t2 = (x2 >> 16) & 7u;
                if (t1 == t2 && ((P.post >> t1) & 1u) && t1 != 0 && t1 != 7) label = 0;
                if ((P.post & 0x80u) && ((x | x2) & kSymLinebreak)) label = 1;
            }
            lb[o] = uint8_t(label);
        }
    }
    if (err) atomicOr(P.status, err);
    phase_mark(prof, 4, tmark);
}

}  // namespace

bool fast_path_supported(const ScoreParams& P) {
    if (!P.pk.present || P.pad != 3 || !P.cid) return false;
    if (!P.ct.present || P.ct.stride_dw != 8 || P.ct.uni_dw != 8 || P.ct.uni_n != kUniDirectChars) return false;  // kPkWide rows
    if (P.type_kind == kTypeNone) return true;
    return P.type_kind == kTypeWindowTable && P.type_window >= 1 && P.type_window <= 3;
}

static bool use_type_rows(const ScoreParams& P) {
    return P.type_kind == kTypeWindowTable && P.pk.has_trow && !P.force_window_table;
}
size_t score_tiles_fast_lds_bytes(const ScoreParams& P, int cap) {
    const size_t head = cap == kFastCapSmall ? offsetof(FastLdsT<kFastCapSmall>, typ) : offsetof(FastLdsT<kFastCapLarge>, typ);
    return head + (use_type_rows(P) ? sizeof(uint4) * kTrowCount : size_t(cap + kMargin));
}

hipError_t launch_score_tiles_fast(const ScoreParams& P, int cap, uint32_t n_tiles, hipStream_t stream) {
    const bool rows = use_type_rows(P);
    size_t lds = score_tiles_fast_lds_bytes(P, cap);
    lds += P.lds_pad;  // occupancy experiments
    const int tm = rows ? kTypeRows : P.type_kind == kTypeWindowTable ? P.type_window : 0;
    const bool dbg = P.debug != 0 || P.prof != nullptr;
    if (cap != kFastCapSmall && cap != kFastCapLarge) return hipErrorInvalidValue;
#define VPT_LAUNCH_FAST2(TM_, DBG_)                                                                                                          \
    if (cap == kFastCapSmall) hipLaunchKernelGGL((score_tiles_fast_kernel<TM_, DBG_, kFastCapSmall, kFastWgSmall>), dim3(n_tiles), dim3(kThreads), lds, stream, P); \
    else hipLaunchKernelGGL((score_tiles_fast_kernel<TM_, DBG_, kFastCapLarge, kFastWgLarge>), dim3(n_tiles), dim3(kThreads), lds, stream, P);
#define VPT_LAUNCH_FAST(TM_)                                  \
    if (dbg) { VPT_LAUNCH_FAST2(TM_, true) } else { VPT_LAUNCH_FAST2(TM_, false) } \
    break;
    switch (tm) {
        case 0: VPT_LAUNCH_FAST(0)
        case 1: VPT_LAUNCH_FAST(1)
        case 2: VPT_LAUNCH_FAST(2)
        case 3: VPT_LAUNCH_FAST(3)
        case 4: VPT_LAUNCH_FAST(4)
        default: return hipErrorInvalidValue;
    }
#undef VPT_LAUNCH_FAST
#undef VPT_LAUNCH_FAST2
    return hipGetLastError();
}

}  // namespace vpt
