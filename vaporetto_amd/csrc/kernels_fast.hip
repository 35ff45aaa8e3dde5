// Specialised tile kernel for the models a Vaporetto / KyTea trainer produces: 16-bit weights, char and type
// windows up to 8 (layout.h, "PACKED TABLES"; one instance per ROW WINDOW wl = max(3, W_c, W_t) -- every distributed model has 3),
// type scores from type rows (in LDS, or in global memory for long type n-grams), the 8^(2W) window table (wl = 3) or none.  Same
// all-matches algorithm as kernels.hip (which stays the general path); what changes is how the work is laid out for a CDNA4 CU,
// whose limiter on this workload is the vector L1's address pipeline: every 16-byte load of a lane costs a slot of it and every
// distinct line it touches two more (profiles/r02_*, r03_b_*), while VALU work is two orders of magnitude cheaper per lane.  So a
// start position issues as few loads as the data structure allows:
//
//   * the patterns form a DOUBLE-ARRAY trie over their first three symbols: one unigram node (indexed by the char's id), one
//     bigram node at (unigram base + id of the next char), and -- only when the bigram's 64-bit filter admits the third char --
//     one trigram node at (bigram base + its id): for wl = 3 that is 4 loads of 16 bytes to 3 lines; no hash, no seed table, no
//     probing; a node names its parent, so a lookup that lands on a foreign node knows it;
//   * the three loads depend on each other, so they are software-pipelined: a trip of the main loop loads the
//     unigram nodes of the positions two trips ahead, the bigram nodes of the next trip's positions and the trigram
//     nodes of its own -- all independent of each other -- and waits once;
//   * only lanes that can match load at all: separators, chars no pattern contains and positions past the tile issue nothing;
//   * rows are added to the LDS score array where they arrive (ds_add_u32: integer => order-free => bit-exact); the type row of a
//     position is added to its unigram row in registers first;
//   * what is data-dependent beyond depth 3 is NOT done in place (64 lanes would wait for the unluckiest one): it is
//     pushed, ballot/mbcnt-compacted, onto wave-private LDS stacks, so that a replay runs one short code path with
//     every lane busy: W trie steps of dictionary words longer than 3 chars (a step that matches re-queues its
//     continuation), M the rare rows with a value outside their fields (taken from the general tables);
//   * UTF-8 decode: the text is staged in LDS, a chunk scan numbers chars and sentences, the thread that scanned a
//     chunk decodes its chars straight into their flat positions (a 3-byte-only round when the wave's chars allow it),
//     a second pass maps them to ids;
//   * a tile is EITHER a run of whole sentences OR a range of flat positions with a halo of max(pattern length) chars on
//     either side (kernels.hpp, TileDesc): sentences of any length are scored here, by as many tiles as they span.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstddef>

#include "device_common.h"
#include "kernels.hpp"

namespace vpt {
namespace {

#ifndef VPT_FAST_PAIR_BI
#define VPT_FAST_PAIR_BI 1                   // the bigram node's halves fetched by lane pairs (A/B builds: -DVPT_FAST_PAIR_BI=0)
#endif
constexpr int kQCap = 128;                   // W items per wave
constexpr uint32_t kQHigh = kQCap - 64;      // replay until one more round of pushes (<= 64) fits
constexpr int kMCap = 64;                    // M items per wave: one round of pushes (<= 64) always fits an empty stack
constexpr uint32_t kCpMask = 0xFFFFu;        // sym = id (kNoId: in no pattern) | type << 16 | tile-local sentence << 19 | linebreak << 29
constexpr uint32_t kSymLinebreak = 1u << 29;
static_assert(kSymLinebreak == kCinfoLinebreak, "the char table's words are symbol words");
constexpr int kWavesF = kThreads / 64;
constexpr int kTypeRows = 4;                 // TM value: type rows (1..3 = window table of that W, 0 = none)
static_assert(kMargin >= int(kPackedMaxSkip) && kMargin >= kMaxWindow, "replay_w reads up to kPackedMaxSkip symbols past a char; rows reach wl - 1 boundaries past one");
constexpr int kTrowCount = int(kTypeRowCount);   // layout.h, type_row_index

// geometry of the instance for row window WL (layout.h, kernels.hpp)
template <int WL>
struct FastGeom {
    static constexpr int kCapG = fast_cap(WL);
    static constexpr uint32_t kPadG = uint32_t(pk_pad(WL));         // separator slots between sentences
    static constexpr int kSymSlots = kCapG + kMargin + 4;           // the last one is the DUMP slot: where a char outside the tile's window is written
    static constexpr uint32_t kDump = uint32_t(kCapG + kMargin);
    static constexpr int kPerThread = kCapG / kThreads;
    static constexpr int kTrowQ = pk_trow_dw(WL) / 4;               // 16-byte words per LDS type row
    static constexpr int kUniQ = ((kUniFieldBits * 2 * WL + kUniBaseBits + 1 + 31) / 32 + 3) / 4;   // 16-byte loads of a unigram node
    static constexpr int kBiQ = (pk_bi_used_dw(WL) + 3) / 4;
    static constexpr int kTriQ = (pk_tri_used_dw(WL) + 3) / 4;
    static_assert(kCapG % kThreads == 0 && kCapG <= 2047, "geometry (a W item keeps its position in 11 bits)");
};

template <int WL>
struct FastLdsT {
    using G = FastGeom<WL>;
    uint32_t sym[G::kSymSlots];              // zero except for the tile's chars (scalar value | sentence << 21 until classified)
    int32_t score[G::kSymSlots];             // staged text bytes during decode
    uint2 queue[kWavesF][kQCap];             // sentence-start bitmap during decode
    uint32_t mqueue[kWavesF][kMCap];
    uint32_t wtot[8];
    uint32_t geo[8];                         // the tile's geometry for phase C (written in phase A: it does not wait in registers over phase B)
    union {                                  // never needed together; the launch allocates the one in use
        uint8_t typ[G::kSymSlots];           // window-table modes
        uint4 trow[kTrowCount * G::kTrowQ];  // TM == kTypeRows, rows in LDS
    };
};
// the replay routines only index the arrays (the queues are passed as pointers)
struct FastLds {
    uint32_t* sym;
    int32_t* score;
};
template <int WL>
constexpr bool fast_lds_ok() {
    using T = FastLdsT<WL>;
    using G = FastGeom<WL>;
    return offsetof(T, typ) % 16 == 0 && offsetof(T, score) % 16 == 0 &&
           sizeof(uint2) * kWavesF * kQCap >= size_t(G::kCapG) * 4 / 8 + 80 &&                              // the decode phase's sentence-start bitmap lives in the W queues
           sizeof(int32_t) * G::kSymSlots >= size_t(G::kCapG) * 4 + 36 &&                                    // ... and the staged text in the score array
           offsetof(T, typ) + sizeof(uint4) * kTrowCount * G::kTrowQ <= size_t(128 / fast_wg(WL)) * 1280;    // gfx950 hands out LDS in 1280-byte granules, 128 per CU
}
static_assert(fast_lds_ok<3>() && fast_lds_ok<4>() && fast_lds_ok<5>() && fast_lds_ok<6>() && fast_lds_ok<7>() && fast_lds_ok<8>(), "LDS budget of the tile geometries");

__device__ __forceinline__ uint32_t lane_rank(uint64_t mask) {  // set bits of `mask` below this lane
    return __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u));
}

// Diagnostics (VPT_PROFILE_PHASES): counts the node reads the lanes ISSUE per level (slots 8 ..: unigram, bigram, trigram nodes, deep entries, deep rows, global type
// rows): the "useful bytes" of the gathers = reads x node size, to set beside the 128-byte lines the L2 fetches for them
__device__ __forceinline__ void count_reads(uint64_t* prof, int slot, bool pred, uint32_t per_lane = 1) {
    if (!prof) return;
    const uint64_t m = __ballot(pred);
    if (m != 0 && int(threadIdx.x & 63) == __builtin_ctzll(m)) atomicAdd(reinterpret_cast<unsigned long long*>(prof + 8 + slot), (unsigned long long)(__popcll(m)) * per_lane);
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

__device__ __forceinline__ int32_t lo16(uint32_t x) { return int32_t(x << 16) >> 16; }
__device__ __forceinline__ int32_t hi16(uint32_t x) { return int32_t(x) >> 16; }

// signed NB-bit field at bit BIT of a node held in registers (d[] is indexed by constants only)
template <int BIT, int NB>
__device__ __forceinline__ int32_t sfield(const uint32_t* d) {
    constexpr int q = BIT >> 5, r = BIT & 31;
    if constexpr (r + NB <= 32) return int32_t(d[q] << (32 - r - NB)) >> (32 - NB);
    else return int32_t(__builtin_amdgcn_alignbit(d[q + 1], d[q], r) << (32 - NB)) >> (32 - NB);
}
template <int BIT, int NB>
__device__ __forceinline__ uint32_t ufield(const uint32_t* d) {
    constexpr int q = BIT >> 5, r = BIT & 31;
    if constexpr (r + NB <= 32) return (d[q] >> r) & ((1u << NB) - 1u);
    else return __builtin_amdgcn_alignbit(d[q + 1], d[q], r) & ((1u << NB) - 1u);
}
// f(j, value of field j) for the N fields of NB bits from bit 0 of d[], unrolled at compile time
template <int J, int N, int NB, typename F>
__device__ __forceinline__ void each_field(const uint32_t* d, F&& f) {
    if constexpr (J < N) {
        f(J, sfield<NB * J, NB>(d));
        each_field<J + 1, N, NB>(d, f);
    }
}
template <int Q>
__device__ __forceinline__ void unpack4(const uint4 (&v)[Q], uint32_t (&d)[4 * Q + 1]) {
#pragma unroll
    for (int i = 0; i < Q; ++i) { d[4 * i] = v[i].x; d[4 * i + 1] = v[i].y; d[4 * i + 2] = v[i].z; d[4 * i + 3] = v[i].w; }
    d[4 * Q] = 0;
}

// What the trie replays read of the packed tables: scalars of the launch, kept in registers over the pattern phase.
struct DeepView {
    const unsigned char* base;
    uint32_t off_deep;
};

// W: up to 64 queued trie steps, all lanes busy: the child sym[s + depth] in the mini-table `ref` of `deep`, whose
// 64-byte entries also name the up-to-8 symbols that must follow (compressed single-child chains; layout.h).
// The home entry and the next one are read together (a mini-table keeps a quarter of its entries free, so nearly
// every search ends within two); the rare longer search loops.  `P`: the parameter block, for what only the rare rows need.
template <int WL>
__device__ __forceinline__ void replay_w(VPT_KARG(ScoreParams) P, const DeepView& K, FastLds& L, WaveStacks& Q, int lane, uint64_t* prof = nullptr) {
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
    count_reads(prof, 3, have, 2);
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
    const uint32_t m = depth + 1 + nskip;  // chars matched so far (>= 4): the row of a pattern of m chars has row_len(m, WL) weights, the
    // first on boundary s + row_lo(m, WL) (wl = 3: m + 1 weights from s - 1); unused row slots hold zero, adding them is harmless
    const int32_t lo = WL == 3 ? -1 : (int32_t(m) - 1 - WL < -1 ? int32_t(m) - 1 - WL : -1);
    const uint32_t rlen = WL == 3 ? m + 1 : uint32_t(int32_t(m - 1 > uint32_t(WL - 1) ? m - 1 : uint32_t(WL - 1)) - lo + 1);
    int32_t* dst = L.score + int32_t(s) + lo;   // s >= WL (the separators in front of the tile's first char)
    const bool row = found && (e.x & (kPkHasRow << 16));
    if (__ballot(row) != 0) {
        uint4 f0 = make_uint4(0, 0, 0, 0), f1 = make_uint4(0, 0, 0, 0);
        if (row) f0 = ld16(K.base, ent + 32);   // (loading the home entry's row speculatively with the entry: no faster, profiles/r02_c5_ab*.jsonl)
        count_reads(prof, 4, row);
        if (__ballot(row && rlen > 8) != 0) { if (row && rlen > 8) f1 = ld16(K.base, ent + 48); }
        if (row) {
            atomicAdd(dst, lo16(f0.x)); atomicAdd(dst + 1, hi16(f0.x)); atomicAdd(dst + 2, lo16(f0.y)); atomicAdd(dst + 3, hi16(f0.y));
            atomicAdd(dst + 4, lo16(f0.z)); atomicAdd(dst + 5, hi16(f0.z)); atomicAdd(dst + 6, lo16(f0.w)); atomicAdd(dst + 7, hi16(f0.w));
        }
        if (__ballot(row && rlen > 8) != 0) {
            if (row && rlen > 8) {
                atomicAdd(dst + 8, lo16(f1.x)); atomicAdd(dst + 9, hi16(f1.x)); atomicAdd(dst + 10, lo16(f1.y));
                atomicAdd(dst + 11, hi16(f1.y)); atomicAdd(dst + 12, lo16(f1.z)); atomicAdd(dst + 13, hi16(f1.z));
            }
        }
    }
    if (__ballot(found && (e.x & (kPkExtRow << 16))) != 0) {  // more than 14 weights or a value outside i16 (rare)
        if (found && (e.x & (kPkExtRow << 16))) {
            VPT_KARG(ScoreParams) R = P;
            VPT_KARG_FENCE(R);   // (asked for here, not kept over the pattern phase)
            const int32_t* w32 = reinterpret_cast<const int32_t*>(K.base + R->pk.off_xrows) + ld16(K.base, ent + 32).x;
            for (uint32_t j = 0; j < rlen; ++j) atomicAdd(dst + j, w32[j]);
        }
    }
    Q.push_w(found && e.y != 0, s | (m << 11), e.y);
}

// The GENERAL tables are laid out for the same row window (tables.cpp: build_table(pats, wl, ..)), so their geometry is a
// compile-time function of WL too -- the rare wide rows cost the main loop no scalar registers for strides and row bounds.
template <int WL>
struct GeneralGeom {
    static constexpr uint32_t kStride = uint32_t((2 + 2 * WL + 3) & ~3);   // dwords per short entry: key + max(len) slots (+ the continuation slot)
    static constexpr uint32_t kUniDw = uint32_t((2 * WL + 3) & ~3);        // dwords per direct unigram row
};
// Entry of a <= 3-char string in the GENERAL short table (layout.h; buckets of two entries): used for the rare packed nodes whose
// row is marked wide.  Returns the entry's first dword or nullptr.
template <int WL>
__device__ __forceinline__ const uint32_t* general_row(const uint32_t* short_tab, uint32_t short_shift, uint32_t short_mask, uint64_t key) {
    constexpr uint32_t kStride = GeneralGeom<WL>::kStride;
    const uint32_t klo = uint32_t(key), khi = uint32_t(key >> 32);
    uint32_t b = hash_slot(key, short_shift);
    bool home = true;
    for (;;) {
        const uint32_t* e0 = short_tab + size_t(b) * (2 * kStride);
        const uint32_t* e1 = e0 + kStride;
        const uint32_t a0 = e0[0], a1 = e0[1], b0 = e1[0], b1 = e1[1];
        if (a0 == klo && (a1 & ~kDisplacedBit) == khi) return e0;
        if (b0 == klo && b1 == khi) return e1;
        if ((a0 | a1) == 0 || (b0 | b1) == 0 || (home && !(a1 & kDisplacedBit))) return nullptr;
        home = false;
        b = (b + 1) & short_mask;
    }
}

// M: up to 64 queued rows with a value outside their fields (rare) -- taken from the general tables, which hold them as i32 keyed by
// code points; the row of a string of n chars covers the boundaries s + row_lo(n, WL) .. (layout.h).  An item is what ONE lane of ONE
// trip of the main loop found wide: su | kinds << 11 with the unigram's start su, the bigram's su - 256, the trigram's su - 512 (the
// three stages a lane has in flight).  Everything this needs of the parameter block is asked for here.
template <int WL>
__device__ __forceinline__ void replay_m(VPT_KARG(ScoreParams) P, FastLds& L, WaveStacks& Q, int lane) {
    VPT_WAVE_LOCKSTEP();   // the queue entries were written by other lanes
    VPT_KARG(ScoreParams) R = P;
    VPT_KARG_FENCE(R);
    const uint32_t take = wave_uniform(Q.nm < 64u ? Q.nm : 64u);
    Q.nm -= take;
    if (uint32_t(lane) >= take) return;
    const uint32_t it = Q.mq[Q.nm + lane];
    const uint32_t kinds = it >> 11, su = it & 0x7FFu;
    const unsigned char* base = R->pk.base;
    const uint32_t* cpid = reinterpret_cast<const uint32_t*>(base + R->pk.off_cpid);
    const uint32_t* short_tab = R->ct.short_tab;
    const uint32_t* uni = R->ct.uni;
    const uint32_t short_shift = R->ct.short_shift, short_mask = R->ct.short_mask;
    // a wide row belongs to a pattern that matched here: its chars are in the alphabet (ids below n_uni)
    const uint32_t last = R->pk.n_uni - 1u;
    auto cp_at = [&](uint32_t pos) { const uint32_t i = L.sym[pos] & kCpMask; return cpid[i < last ? i : last]; };
    if (kinds & kWideUni) {   // (the general tables index the chars of the BMP directly and keep the others' rows with the short strings)
        const uint32_t s = su, c1 = cp_at(s);
        const uint32_t* u = c1 < kUniDirectChars ? uni + size_t(c1) * GeneralGeom<WL>::kUniDw : general_row<WL>(short_tab, short_shift, short_mask, short_key(c1, 0, 0));
        if (u && c1 >= kUniDirectChars) u += 2;
        if (u)
            for (int j = 0; j < row_len(1, WL); ++j) atomicAdd(L.score + int32_t(s) + row_lo(1, WL) + j, int32_t(u[j]));
    }
    if (kinds & kWideBi) {
        const uint32_t s = su - uint32_t(kThreads);
        if (const uint32_t* e = general_row<WL>(short_tab, short_shift, short_mask, short_key(cp_at(s), cp_at(s + 1), 0)))
            for (int j = 0; j < row_len(2, WL); ++j) atomicAdd(L.score + int32_t(s) + row_lo(2, WL) + j, int32_t(e[2 + j]));
    }
    if (kinds & kWideTri) {
        const uint32_t s = su - 2u * uint32_t(kThreads);
        if (const uint32_t* e = general_row<WL>(short_tab, short_shift, short_mask, short_key(cp_at(s), cp_at(s + 1), cp_at(s + 2))))
            for (int j = 0; j < row_len(3, WL); ++j) atomicAdd(L.score + int32_t(s) + row_lo(3, WL) + j, int32_t(e[2 + j]));
    }
}

// W replays until at most `mark` items are left (a W item becomes at most one W item, so the loop ends).
template <int WL>
__device__ __forceinline__ void drain_w(VPT_KARG(ScoreParams) P, const DeepView& K, FastLds& L, WaveStacks& Q, int lane, uint32_t mark, uint64_t* prof = nullptr) {
    while (Q.nw > mark) replay_w<WL>(P, K, L, Q, lane, prof);
}

// optional phase timing (VPT_PROFILE_PHASES): wave 0 of every workgroup adds the shader cycles it spent per phase
__device__ __forceinline__ uint64_t phase_mark(uint64_t* prof, int slot, uint64_t t_prev) {
    if (!prof) return 0;
    const uint64_t now = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) atomicAdd(reinterpret_cast<unsigned long long*>(prof + slot), (unsigned long long)(now - t_prev));
    return now;
}

// The parameter block is read through a pointer, phase by phase (device_common.h, VPT_KARG); what phase C needs of the tile's geometry
// waits in LDS (FastLdsT::geo) instead of in registers over the pattern phase.
template <int WL, int TM, bool DBG>
__global__ __launch_bounds__(kThreads, fast_wg(WL)) void score_tiles_fast_kernel(const ScoreParams P_in) {
    using G = FastGeom<WL>;
    constexpr uint32_t kPad = G::kPadG, kDump = G::kDump;
    constexpr int kFastCap = G::kCapG, kSymSlots = G::kSymSlots, kPerThread = G::kPerThread;
    constexpr int kNU = pk_uni_fields(WL), kNB = pk_bi_fields(WL), kNT = pk_tri_fields(WL);
    // (row window 3: the 32-byte bigram node.  The same for the 64-byte nodes of windows 4 .. 8 -- two requests a node instead of three or four, at the
    // price of a fourth load instruction per lane -- measured equal on charw4, 0.1250 against 0.1247 ms, profiles/r06_n_*: not kept)
    constexpr bool kPairBi = VPT_FAST_PAIR_BI != 0 && G::kBiQ == 2;
    VPT_KARG(ScoreParams) P = VPT_KARG_PTR(ScoreParams, P_in);
    const uint32_t dbg = DBG ? P->debug : 0u;
    uint64_t* const prof = DBG ? P->prof : nullptr;
    VPT_DYNAMIC_LDS(smem);
    FastLdsT<WL>& M = *reinterpret_cast<FastLdsT<WL>*>(smem);
    FastLds L{M.sym, M.score};
    const int tid = threadIdx.x, lane = tid & 63, wave = int(wave_uniform(uint32_t(tid) >> 6));
    const uint32_t wbase = uint32_t(wave) << 6;   // this wave's first thread, as a scalar

    // ---------------------------------------------------------------- the tile
    // Cut tiles come with a description (kernels.hpp, TileDesc: four scalar loads).  A whole-sentence tile is described by its
    // first sentence and its neighbour's (the general kernel's assign_tiles_kernel: the cheapest thing in front of this launch); the
    // rest follows from the offsets of the two -- a second trip of scalar loads that other tiles' work hides.
    const uint32_t tile = blockIdx.x;
    uint64_t i0, byte0, g0;
    uint32_t nbytes, nsent, flat_len, own_lo, own_hi, expect_chars, strict_end;
    int32_t c_off, sib0;
    const uint64_t* const p_boff = P->boff;
    const uint64_t* const p_ooff = P->ooff;
    if (P->tiles) {
        const TileDesc* const dp = P->tiles + tile;
        nbytes = dp->nbytes;
        if (nbytes == 0) return;   // an empty tile (past the batch's end; reported as not fitting)
        i0 = dp->i0; nsent = dp->nsent; flat_len = dp->flat_len; own_lo = dp->own_lo; own_hi = dp->own_hi;
        c_off = dp->c_off; sib0 = dp->sib0;
        byte0 = uint64_t(dp->byte0_lo) | (uint64_t(dp->byte0_hi) << 32); g0 = uint64_t(dp->g0_lo) | (uint64_t(dp->g0_hi) << 32);
        expect_chars = dp->expect_chars; strict_end = dp->strict_end;
    } else {
        i0 = P->tile_first[tile];
        const uint64_t i1 = P->tile_first[tile + 1];
        if (i0 >= i1) return;      // no sentence starts in this tile's range
        const uint64_t O0 = p_ooff[i0], O1 = p_ooff[i1], B0 = p_boff[i0], B1 = p_boff[i1];
        const uint64_t fl = uint64_t(kPad) + (O1 + i1 * (kPad + 1)) - (O0 + i0 * (kPad + 1));
        if (O1 < O0 || B1 <= B0 || fl > uint64_t(kFastCap) || B1 - B0 > uint64_t(kFastCap) * 4 + 15 || i1 - i0 > 1023) {
            if (tid == 0) atomicOr(P->status, kErrScratchTooSmall);   // a sentence longer than the caller's bound (or offsets that are no offsets)
            return;
        }
        nsent = uint32_t(i1 - i0); byte0 = B0; nbytes = uint32_t(B1 - B0); c_off = int32_t(kPad); sib0 = -1;
        flat_len = uint32_t(fl); own_lo = kPad; own_hi = flat_len; g0 = O0 + i0;
        expect_chars = uint32_t((O1 + i1) - (O0 + i0)); strict_end = 1;
    }
    const uint8_t* tbase = P->text + byte0;
    const uintptr_t a0 = reinterpret_cast<uintptr_t>(tbase) & ~uintptr_t(15);
    const uint32_t head = uint32_t(reinterpret_cast<uintptr_t>(tbase) - a0);
    const uint32_t nbytes_al = head + nbytes;   // <= 4 * kFastCap + 15 + 16: the assign kernels see to it
    const uint32_t nchunks = (nbytes_al + 15) >> 4;
    // LDS positions in use, rounded up to whole waves of start positions plus the look-ahead margin: what is zeroed and walked
    const uint32_t span = ((flat_len + 63u) & ~63u) + uint32_t(kMargin) + 4u < uint32_t(kSymSlots) ? ((flat_len + 63u) & ~63u) + uint32_t(kMargin) + 4u : uint32_t(kSymSlots);
    uint32_t err = 0;
    uint64_t tmark = prof ? __builtin_amdgcn_s_memtime() : 0;
    const bool trow_lds = TM == kTypeRows && P->pk.trow_mode == kTypeRowsLds;   // wave-uniform (a kernel argument)

    // ---------------------------------------------------------------- A. decode
    // Flat layout: lead number ci of the staged text, in the tile's sentence si, sits at c_off + ci + kPad * si; everything else is
    // zero (separators, the slack past the tile).  The text is staged in LDS (the score array is free until phase B), a chunk
    // scan numbers the chars and the sentences, and the thread that scanned a chunk decodes its chars straight into their flat
    // positions; a second pass classifies them.  Chars outside [kPad, flat_len) (a cut tile stages whole 256-byte blocks) go to
    // the dump slot.
    uint32_t* bitmap = reinterpret_cast<uint32_t*>(&M.queue[0][0]);   // one bit per text byte: a sentence starts here
    uint32_t* raw = reinterpret_cast<uint32_t*>(&L.score[0]);
    // this thread's sentence (the first 256 of the tile: a tile of more, i.e. of sentences of a char or two, loops below): its
    // offsets are asked for NOW, together with the text and the tables -- one trip to memory instead of three on the tile's
    // critical path -- and used after the barriers
    uint64_t my_b = 0, my_bn = 1, my_oa = 0, my_ob = 0;
    if (uint32_t(tid) < nsent) { my_b = p_boff[i0 + tid]; my_bn = p_boff[i0 + tid + 1]; my_oa = p_ooff[i0 + tid]; my_ob = p_ooff[i0 + tid + 1]; }
    for (uint32_t i = tid; i < ((nbytes_al + 31) >> 5) + 1; i += kThreads) bitmap[i] = 0;
    for (uint32_t i = tid; i < (span + 3) / 4; i += kThreads) reinterpret_cast<uint4*>(L.sym)[i] = make_uint4(0, 0, 0, 0);
    if (TM != kTypeRows) {
        for (uint32_t i = tid; i < (span + 3) / 4; i += kThreads) reinterpret_cast<uint32_t*>(M.typ)[i] = 0;
    }
    for (uint32_t c = tid; c < nchunks; c += kThreads) reinterpret_cast<uint4*>(raw)[c] = reinterpret_cast<const uint4*>(a0)[c];   // (non-temporal loads / stores here measured 1-2 % slower: profiles/r02_c1_ab.jsonl, r02_c3_ab.jsonl)
    if (tid == 0) {
        raw[nchunks * 4] = 0;  // the dword after the staged text is read (as padding) by the last char
        // what phase C needs of the tile: read back there, so that it does not wait in scalar registers over the pattern phase
        const uint64_t obase = g0 - i0;                        // >= 0: every sentence in front of the tile has a char
        M.geo[0] = uint32_t(obase); M.geo[1] = uint32_t(obase >> 32); M.geo[2] = uint32_t(c_off); M.geo[3] = own_lo; M.geo[4] = own_hi;
    }
    // the LDS-resident type rows (coalesced 16-byte loads from the predictor's arena)
    if (trow_lds) {
        const unsigned char* const tb = P->pk.base + P->pk.off_trow;
        for (uint32_t i = tid; i < uint32_t(kTrowCount * G::kTrowQ); i += kThreads) M.trow[i] = *reinterpret_cast<const uint4*>(tb + (i << 4));
    }
    __syncthreads();
    tmark = phase_mark(prof, 0, tmark);   // zeroing, staging, table loads
    for (uint32_t j = tid; j < nsent; j += kThreads) {
        const uint64_t b = j == uint32_t(tid) ? my_b : p_boff[i0 + j], bn = j == uint32_t(tid) ? my_bn : p_boff[i0 + j + 1];
        if (bn <= b) err |= kErrEmptySentence;
        if (b >= byte0 && b - byte0 < nbytes) {   // (sentence i0 of a cut tile may have started before the staged text)
            const uint32_t pos = head + uint32_t(b - byte0);
            atomicOr(&bitmap[pos >> 5], 1u << (pos & 31));
        }
    }
    __syncthreads();
    tmark = phase_mark(prof, 1, tmark);   // sentence starts
    uint32_t base_leads = 0, base_starts = 0;
    uint32_t min_lead = 0xFFu;     // over this thread's chars that are not three-byte sequences: the smallest lead byte (a NUL char is the byte 0; a
                                   // cut tile's last staged char may miss its continuation bytes, so the decoded value is not what is looked at)
    int32_t max_p = -1;            // ... and the last flat position (a char past the tile)
    for (uint32_t c0 = 0; c0 < nchunks; c0 += kThreads) {   // one pass for up to 4 KB of tile text, else two
        const uint32_t c = c0 + tid;
        uint32_t lm = 0, sm = 0;
        const uint32_t pos0 = c < nchunks ? c * 16 : 0u;   // (a thread without a chunk still runs the rounds below, at position 0)
        if (c < nchunks) {
            const uint4 v = reinterpret_cast<const uint4*>(raw)[c];
            const uint32_t lo = pos0 < head ? head - pos0 : 0u;
            const uint32_t rem = nbytes_al - pos0;
            const uint32_t hi = rem < 16 ? rem : 16u;
            const uint32_t vm = ((1u << hi) - 1u) & ~((1u << lo) - 1u);
            lm = lead_mask16(v) & vm;
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
        const int32_t sib = sib0 + int32_t(base_starts + (excl >> 16));   // sentence of a char = sib + starts up to it in this chunk
        base_leads += total & 0xFFFFu;
        base_starts += total >> 16;
        __syncthreads();   // wtot is rewritten by the next pass
        uint32_t m = lm;
        int32_t fb = c_off + int32_t(ci) + int32_t(kPad) * sib;   // flat = c_off + char index + kPad * sentence
        int32_t p = -1;                                          // the chunk's chars go to ascending positions: its last one is its largest
        const uint32_t* const rp = raw + (pos0 >> 2);            // this chunk's first dword of the staged text
        // Every round takes one char per lane; a three-byte sequence (Japanese text mostly is) decodes with five instructions.
        while (m != 0) {   // (a loop every lane stays in until the wave's last char, with the idle lanes writing the dump slot, measured
            const uint32_t k = uint32_t(__builtin_ctz(m));   //  0.5 % slower: profiles/r03_j_ab_m1.jsonl)
            const uint32_t t = m - 1u;
            const uint32_t r = uint32_t(__popc(sm & (m ^ t)));   // m ^ (m - 1): the bits up to and including bit k
            m &= t;
            const uint32_t w = __builtin_amdgcn_alignbyte(rp[(k >> 2) + 1], rp[k >> 2], k & 3u);   // the char's four bytes (pos0 is a multiple of 16)
            uint32_t cp;
            if ((w & 0xF0u) == 0xE0u) {   // a three-byte sequence (the side of the branch nobody takes is skipped); its lead byte is no NUL
                cp = (((((w & 0xFu) << 6) | ((w >> 8) & 0x3Fu)) << 6) | ((w >> 16) & 0x3Fu));
            } else {
                cp = utf8_scalar_bf(w);
                min_lead = (w & 0xFFu) < min_lead ? (w & 0xFFu) : min_lead;
            }
            p = fb + int32_t(kPad) * int32_t(r);
            ++fb;
            const bool inside = uint32_t(p - int32_t(kPad)) < flat_len - kPad;   // (p < kPad wraps to a large number)
            L.sym[inside ? uint32_t(p) : kDump] = cp | (uint32_t(sib + int32_t(r)) << 21);   // 21 + 10 bits; classified below
        }
        max_p = p > max_p ? p : max_p;
    }
    if (expect_chars != 0xFFFFFFFFu && base_leads != expect_chars) err |= kErrBadOffsets;
    if (min_lead == 0) err |= kErrNulChar;
    if (strict_end && max_p >= int32_t(flat_len)) err |= kErrBadOffsets;   // more chars than the offsets promise
    tmark = phase_mark(prof, 2, tmark);   // chunk scan + decode rounds
    __syncthreads();  // the staged text has been read: the score array can be zeroed; every char is in place
    for (uint32_t i = tid; i < (span + 3) / 4; i += kThreads) reinterpret_cast<uint4*>(L.score)[i] = make_uint4(0, 0, 0, 0);
    VPT_KARG_FENCE(P);
    {   // classify: one word of the char table per char (all of a thread's reads in flight together), then the final symbols
        const uint32_t* const cid = P->cid;
        uint32_t* const cps_out = P->cps_out;   // (wave-uniform) a fill_tags call on this batch follows and wants the chars decoded (it then skips its own pass)
        // a char's final symbol word `v` is in place: the other things that want it
        auto placed = [&](uint32_t pos, uint32_t cp, uint32_t si, uint32_t v) {
            L.sym[pos] = v;
            if (cps_out) {
                const uint32_t* const cinfo = P->cinfo;
                const uint32_t scored = (cinfo && cp < 0x10000u) ? (cinfo[cp] & 0xFFFFu) : cp;   // through KyteaFullwidthFilter when that flag is on
                if (cp != 0 && pos >= own_lo && pos < own_hi) {
                    const uint64_t at = g0 + uint64_t(int64_t(int32_t(pos) - c_off - int32_t(kPad) * int32_t(si)));
                    if (at < P->total_chars) cps_out[at] = scored | (((v >> 16) & 7u) << 24);
                    else err |= kErrBadOffsets;   // out_offsets that run past the total the caller stated
                }
            }
            if (TM != kTypeRows) M.typ[pos] = uint8_t((v >> 16) & 7u);
        };
        uint32_t xs[kPerThread], info[kPerThread];
#pragma unroll
        for (int k = 0; k < kPerThread; ++k) {
            xs[k] = 0;
            if (wbase + uint32_t(k) * kThreads < flat_len) xs[k] = L.sym[uint32_t(tid) + uint32_t(k) * kThreads];   // (wave-uniform)
        }
#pragma unroll
        for (int k = 0; k < kPerThread; ++k) {
            info[k] = 0;
            // id of the char it is scored as | CharacterType << 16 | linebreak << 29: one word of a 256 KB table (plain, or -- with
            // VPT_FLAG_KYTEA_FULLWIDTH -- the one that looks through KyteaFullwidthFilter); a separator or a char outside the BMP asks for nothing
            const uint32_t cp = xs[k] & 0x1FFFFFu;
            if (uint32_t(cp - 1u) < 0xFFFFu) info[k] = cid[cp];
        }
        uint32_t outside = 0;   // this lane's chars outside the BMP (bit k), for the pass below
#pragma unroll
        for (int k = 0; k < kPerThread; ++k) {
            if (wbase + uint32_t(k) * kThreads >= flat_len) continue;   // wave-uniform
            const uint32_t pos = uint32_t(tid) + uint32_t(k) * kThreads;
            const uint32_t cp = xs[k] & 0x1FFFFFu, si = xs[k] >> 21;
            if (cp >= 0x10000u) { outside |= 1u << k; continue; }   // its word stays as decoded
            placed(pos, cp, si, cp != 0 ? (info[k] | (si << 19)) : 0u);   // 0: a separator (or NUL, which has raised kErrNulChar)
        }
        if (__ballot(outside != 0) != 0) {   // rare: a char outside the BMP has its type computed and its id -- the model's alphabet may hold a few
            VPT_KARG(ScoreParams) R = P;     // such chars -- looked up in `xcid` (layout.h); KyteaFullwidthFilter leaves it alone.  One copy of this code.
            VPT_KARG_FENCE(R);
            const uint32_t ox = R->pk.off_xcid;
            const uint32_t* const xt = reinterpret_cast<const uint32_t*>(R->pk.base + ox);
#pragma nounroll
            for (int k = 0; k < kPerThread; ++k) {
                if (!((outside >> k) & 1u)) continue;
                const uint32_t pos = uint32_t(tid) + uint32_t(k) * kThreads;
                const uint32_t x = L.sym[pos], cp = x & 0x1FFFFFu, si = x >> 21;
                const uint32_t id = ox ? xcid_find(xt, cp) : kNoId;
                placed(pos, cp, si, id | (char_type(cp) << 16) | (si << 19));
            }
        }
    }
    __syncthreads();
    // the sentences' places, as out_offsets state them, against where the text put them: sentence j of the tile starts at
    // e = c_off + (its first char's global number - g0) + kPad * j and is followed by a separator -- checked wherever e or its end
    // falls into the tile's window.  (A sentence longer or shorter than stated moves everything behind it.)
    for (uint32_t j = tid; j < nsent; j += kThreads) {
        const uint64_t oa = j == uint32_t(tid) ? my_oa : p_ooff[i0 + j], ob = j == uint32_t(tid) ? my_ob : p_ooff[i0 + j + 1];
        const int64_t e = int64_t(c_off) + int64_t(oa + i0 + j - g0) + int64_t(kPad) * int64_t(j);
        const int64_t end = e + int64_t(ob - oa) + 1;
        if (ob < oa) err |= kErrBadOffsets;
        if (e >= int64_t(kPad) && e < int64_t(flat_len)) {
            const uint32_t x = L.sym[e], xm = L.sym[e - 1];
            if ((x & kCpMask) == 0 || ((x >> 19) & 1023u) != (j & 1023u) || (xm & kCpMask) != 0) err |= kErrBadOffsets;
        }
        if (end >= int64_t(kPad) && end < int64_t(flat_len) && (L.sym[end] & kCpMask) != 0) err |= kErrBadOffsets;
    }

    tmark = phase_mark(prof, 3, tmark);   // classify + sentence checks
    // ---------------------------------------------------------------- B. patterns
    // One lane per start position, three trips in flight: trip j loads the TRIGRAM nodes of its own positions
    // (tid + 256 j), the BIGRAM nodes of the next trip's and the UNIGRAM nodes of the positions two trips ahead -- loads
    // that do not depend on each other -- waits once, adds the rows that arrived to the LDS score array and computes the
    // addresses the next trip needs (child slot = base in the parent + id of the next char; layout.h).
    VPT_KARG_FENCE(P);
    const unsigned char* const kbase = P->pk.base;
    const DeepView K{kbase, P->pk.off_deep};
    WaveStacks Q{&M.queue[wave][0], &M.mqueue[wave][0], 0u, 0u};
    const uint32_t off_bi = P->pk.off_bi & ~255u, off_tri = P->pk.off_tri & ~255u;   // they ARE 256-byte aligned (capi.cpp); now the compiler knows
    const uint32_t n_tri = P->pk.n_tri, bi_shift = P->pk.bi_shift;                    // (off_uni is 0: the unigram nodes lead the packed tables)
    uint32_t b_slot = 0, b_key = 0, b_id3 = 0;   // next trip's bigram stage: node slot, its key (0: no bigram starts there), id of the third char
    uint32_t t_slot = ~0u, t_par = 0;            // this trip's trigram stage: node slot (~0: none) and parent slot + 1
    for (int j = -2; j < kPerThread; ++j) {
        if (dbg & 16u) break;  // timing ablation: no pattern phase at all
        // wave-uniform: which stages have positions of this wave inside the tile
        const bool do_t = j >= 0 && wbase + uint32_t(j) * kThreads < flat_len;
        const bool do_b = j + 1 >= 0 && j + 1 < kPerThread && wbase + uint32_t(j + 1) * kThreads < flat_len;
        const bool do_u = j + 2 < kPerThread && wbase + uint32_t(j + 2) * kThreads < flat_len;
        if (!do_t && !do_b && !do_u) break;
        const uint32_t s_t = uint32_t(tid) + uint32_t(j) * kThreads, s_b = s_t + kThreads, s_u = s_b + kThreads;
        // ---- every load first; only lanes that can match issue one (a node nobody loaded is never looked at: no zeroing)
        uint4 tn[G::kTriQ], nn[G::kBiQ], un[G::kUniQ];
#pragma unroll
        for (int q = 0; q < G::kTriQ; ++q) VPT_UNDEF4(tn[q]);
#pragma unroll
        for (int q = 0; q < G::kBiQ; ++q) VPT_UNDEF4(nn[q]);
#pragma unroll
        for (int q = 0; q < G::kUniQ; ++q) un[q] = make_uint4(0, 0, 0, 0);   // (a char of no pattern has a zero row)
        uint4 pa, pb;   // (kPairBi) the halves of the lane pair's two bigram nodes this lane fetches
        VPT_UNDEF4(pa); VPT_UNDEF4(pb);
        uint32_t x1 = 0, x2 = 0, x3 = 0;
        uint32_t wide_kinds = 0;   // what this lane's three stages find outside their fields this trip
        if (DBG && do_t) count_reads(prof, 2, t_slot != ~0u);
        if (DBG && do_b) count_reads(prof, 1, b_key != 0);
        if (do_t) {
            if (t_slot != ~0u) {
                const uint32_t a = off_tri + (((dbg & 1u) ? 0u : t_slot) * uint32_t(4 * pk_tri_dw(WL)));
#pragma unroll
                for (int q = 0; q < G::kTriQ; ++q) tn[q] = ld16(kbase, a + 16u * uint32_t(q));
            }
        }
        if (do_b) {
            if constexpr (kPairBi) {
                // The two halves of a 32-byte node as ONE request each way: two loads of one lane that miss the vector L1 are two requests to the
                // L2 -- nothing merges them, and the L1's outstanding misses (some 200 a CU) over their latency are what this phase runs at
                // (tools/tcp_bench, profiles/r06_j_*: adjacent pairs of 16-byte loads 132 G nodes/s where single ones make 264) -- while two LANES
                // of one instruction that ask for one line are one request.  So a pair of lanes fetches the even lane's node (a half each), then
                // the odd lane's, and they swap what they fetched for each other (four DPP moves a half).
                const uint32_t p_slot = uint32_t(__builtin_amdgcn_mov_dpp(int(b_slot), 0xB1, 0xF, 0xF, true));   // quad_perm [1, 0, 3, 2]: the partner's
                const bool p_on = __builtin_amdgcn_mov_dpp(int(b_key != 0 ? 1u : 0u), 0xB1, 0xF, 0xF, true) != 0;
                const bool odd = (lane & 1) != 0;
                const uint32_t slot_a = odd ? p_slot : b_slot, slot_b = odd ? b_slot : p_slot;
                const bool on_a = odd ? p_on : b_key != 0, on_b = odd ? b_key != 0 : p_on;
                const uint32_t half = odd ? 16u : 0u;
                if (on_a) pa = ld16(kbase, (off_bi + (((dbg & 1u) ? 0u : slot_a) * uint32_t(4 * pk_bi_dw(WL)))) | half);
                if (on_b) pb = ld16(kbase, (off_bi + (((dbg & 1u) ? 0u : slot_b) * uint32_t(4 * pk_bi_dw(WL)))) | half);
            } else if (b_key != 0) {
                const uint32_t a = off_bi + (((dbg & 1u) ? 0u : b_slot) * uint32_t(4 * pk_bi_dw(WL)));
#pragma unroll
                for (int q = 0; q < G::kBiQ; ++q) nn[q] = ld16(kbase, a | (16u * uint32_t(q)));
            }
        }
        bool live = false;
        if (do_u) {
            x1 = L.sym[s_u]; x2 = L.sym[s_u + 1]; x3 = L.sym[s_u + 2];   // the array is zero past the tile; s_u + 2 < kSymSlots
            const uint32_t id1 = x1 & kCpMask;
            live = id1 != 0;
            const bool want = uint32_t(id1 - 1u) < kNoId - 1u;   // a separator has no node; a char no pattern contains has none either (its row is zero, its base unused)
            if (DBG) count_reads(prof, 0, want);
            if (__ballot(want) != 0) {
                if (want) {
                    const uint32_t a = ((dbg & 4u) ? 0u : id1) * uint32_t(4 * pk_uni_dw(WL));
#pragma unroll
                    for (int q = 0; q < G::kUniQ; ++q) un[q] = ld16(kbase, a + 16u * uint32_t(q));
                }
            }
        }
        // ---- trigram stage of positions s_t: the node is ours if it names our bigram node as its parent
        if (do_t) {
            uint32_t td[4 * G::kTriQ + 1];
            unpack4(tn, td);
            const bool hit = t_slot != ~0u && (td[0] & kTriParentMask) == t_par;
            if (hit) {   // 2 WL - 2 weights of 16 bits from boundary s - WL + 2 (a kPkWide node holds zero weights)
                int32_t* p = L.score + s_t - uint32_t(WL - 2);
#pragma unroll
                for (int q = 0; q < kNT; q += 2) {
                    const uint32_t w = td[pk_tri_w_dw(WL) + q / 2];
                    atomicAdd(p + q, lo16(w)); atomicAdd(p + q + 1, hi16(w));
                }
            }
            if (hit && (td[0] & (kPkWide << kTriFlagShift))) wide_kinds |= kWideTri;
            // the node's kids, unless its child filter says that the char behind the trigram is none of them (layout.h, "tri": the queued trie
            // step that this saves is two 64-byte entry reads)
            uint32_t kids = 0;
            if (hit && (td[pk_tri_kids_dw(WL)] & kTriKidsRefMask) != 0) {
                const uint32_t c4 = L.sym[s_t + 3] & kCpMask;   // (zero past the tile; s_t + 3 < kSymSlots)
                if (uint32_t(c4 - 1u) < kNoId - 1u && ((tri_kid_filter(td[0], td[pk_tri_kids_dw(WL)]) >> packed_kid_filter_bit(c4)) & 1u)) kids = td[pk_tri_kids_dw(WL)] & kTriKidsRefMask;
            }
            VPT_PIN(kids);
            drain_w<WL>(P, K, L, Q, lane, kQHigh, prof);           // room for one more round of pushes
            Q.push_w(kids != 0 && !(dbg & 8u), s_t | (3u << 11), kids);
        }
        // ---- bigram stage of positions s_b: key check, the row, and the address of the trigram node
        t_slot = ~0u;
        if (do_b) {
            if constexpr (kPairBi) {   // pa: a half of the pair's even node, pb: of its odd node -- the lane's own node is its half and the partner's
                const bool odd = (lane & 1) != 0;
                const auto sel = [&](uint32_t a, uint32_t b) { return odd ? b : a; };
                const auto swp = [&](uint32_t v) { return uint32_t(__builtin_amdgcn_mov_dpp(int(v), 0xB1, 0xF, 0xF, true)); };
                // the half I fetched for my partner (even: of pb, odd: of pa) goes over; what comes back is the other half of my node
                const uint32_t ox = swp(sel(pb.x, pa.x)), oy = swp(sel(pb.y, pa.y)), oz = swp(sel(pb.z, pa.z)), ow = swp(sel(pb.w, pa.w));
                nn[0] = make_uint4(sel(pa.x, ox), sel(pa.y, oy), sel(pa.z, oz), sel(pa.w, ow));   // first half: the even lane's own, the odd lane's from its partner
                nn[1] = make_uint4(sel(ox, pb.x), sel(oy, pb.y), sel(oz, pb.z), sel(ow, pb.w));
            }
            uint32_t nd[4 * G::kBiQ + 1];
            unpack4(nn, nd);
            const bool keyok = b_key != 0 && nd[pk_bi_key_dw(WL)] == b_key;
            const uint32_t* const rowd = nd + pk_bi_row_dw(WL);
            if (keyok) {   // 2 WL - 1 fields of 19 bits from boundary s - WL + 1; behind them the wide flag (the row then holds zeros: M stack)
                int32_t* p = L.score + s_b - uint32_t(WL - 1);
                each_field<0, kNB, kBiFieldBits>(rowd, [&](int f, int32_t v) { atomicAdd(p + f, v); });
            }
            // the child filter: bit packed_filter_bit(id3) of a 64-bit word (a 64-bit shift: the bit number selects the half)
            const uint64_t filt = uint64_t(nd[pk_bi_filter_dw(WL)]) | (uint64_t(nd[pk_bi_filter_dw(WL) + 1]) << 32);
            const bool cont = keyok && uint32_t(b_id3 - 1u) < kNoId - 1u && ((filt >> packed_filter_bit(b_id3)) & 1u) != 0;
            const uint32_t ts = nd[pk_bi_base_dw(WL)] + b_id3;   // modulo 2^32 (layout.h); a false positive of the filter may point anywhere
            t_slot = (cont && ts < n_tri && !(dbg & 2u)) ? ts : ~0u;
            t_par = b_slot + 1u;
            if (keyok && ufield<pk_bi_wide_bit(WL), 1>(rowd) != 0) wide_kinds |= kWideBi;
        }
        // ---- unigram stage of positions s_u: the row (+ the type row), and the address of the bigram node
        b_key = 0;
        if (do_u) {
            uint32_t ud[4 * G::kUniQ + 1];
            unpack4(un, ud);
            const uint32_t id1 = x1 & kCpMask, id2 = x2 & kCpMask;
            // 2 WL fields of 18 bits from boundary s - WL; behind them the base of the bigram nodes (19 bits) and the wide flag
            int32_t a[kNU];
            each_field<0, kNU, kUniFieldBits>(ud, [&](int f, int32_t v) { a[f] = v; });
            if (TM == kTypeRows) {
                if (trow_lds) {   // (wave-uniform)
                    // a dead lane (t1 = 0) reads in front of the rows; nothing is added for it
                    const int32_t ri = int32_t(type_row_index((x1 >> 16) & 7u, (x2 >> 16) & 7u, (x3 >> 16) & 7u)) * G::kTrowQ;
                    uint4 tr[G::kTrowQ];
#pragma unroll
                    for (int q = 0; q < G::kTrowQ; ++q) tr[q] = M.trow[ri + q];
                    uint32_t rd[4 * G::kTrowQ + 1];
                    unpack4(tr, rd);
                    each_field<0, kNU, kUniFieldBits>(rd, [&](int f, int32_t v) { a[f] += v; });
                } else {
                    // rows of i32 in global memory, indexed by the types of s .. s + levels - 1 (layout.h, "TYPE ROWS"; the array is zero past the
                    // tile: code 0 ends the prefix)
                    uint32_t idx = 0;
                    for (int32_t i = int32_t(P->pk.trow_levels) - 1; i >= 3; --i) idx = idx * 7u + ((L.sym[s_u + uint32_t(i)] >> 16) & 7u);
                    idx = (idx * 7u + ((x3 >> 16) & 7u)) * 7u + ((x2 >> 16) & 7u);
                    idx = idx * 6u + (((x1 >> 16) & 7u) - 1u);
                    if (DBG) count_reads(prof, 5, live);
                    if (live) {
                        const uint32_t ra = P->pk.off_trow + idx * uint32_t(4 * pk_trow_global_dw(WL));
                        uint4 tr[pk_trow_global_dw(WL) / 4];
#pragma unroll
                        for (int q = 0; q < pk_trow_global_dw(WL) / 4; ++q) tr[q] = ld16(kbase, ra + 16u * uint32_t(q));
                        uint32_t rd[pk_trow_global_dw(WL) + 1];
                        unpack4(tr, rd);
#pragma unroll
                        for (int f = 0; f < kNU; ++f) a[f] += int32_t(rd[f]);
                    }
                }
            }
            if (live) {
                int32_t* p = L.score + s_u - uint32_t(WL);
#pragma unroll
                for (int f = 0; f < kNU; ++f) atomicAdd(p + f, a[f]);
            }
            b_slot = (ufield<pk_uni_base_bit(WL), kUniBaseBits>(ud) << bi_shift) + id2;
            // a bigram starts here when both chars are chars of the alphabet (ids in [1, kNoId): 0 is a separator, kNoId a char of no pattern)
            b_key = (uint32_t(id1 - 1u) < kNoId - 1u && uint32_t(id2 - 1u) < kNoId - 1u) ? (id1 | (id2 << 16)) : 0u;
            b_id3 = x3 & kCpMask;
            if (live && ufield<pk_uni_base_bit(WL) + kUniBaseBits, 1>(ud) != 0 && !(dbg & 32u)) wide_kinds |= kWideUni;
        }
        // ---- rows outside their fields (rare): ONE item per lane and trip; they wait on the M stack until a replay finds many of them
        // (a replay per trip for an item or two cost 22 % more vector-memory instructions than the pattern phase had: profiles/r05_f_*)
        const uint64_t mm = __ballot(wide_kinds != 0);
        if (mm != 0) {
            while (Q.nm + uint32_t(__popcll(mm)) > uint32_t(kMCap)) replay_m<WL>(P, L, Q, lane);
            Q.push_m(wide_kinds != 0, s_u | (wide_kinds << 11));
        }
    }
    while (Q.nm > 0) replay_m<WL>(P, L, Q, lane);
    while (Q.nw > 0) replay_w<WL>(P, K, L, Q, lane, prof);
    tmark = phase_mark(prof, 4, tmark);   // patterns
    __syncthreads();
    tmark = phase_mark(prof, 5, tmark);   // waiting for the other waves

    // ---------------------------------------------------------------- C. boundaries
    // boundary p lies between flat positions p and p + 1, both chars of one sentence; the tile writes those in [own_lo, own_hi).
    // Its output index is (g0 - i0) + (p - c_off) - (kPad + 1) * (sentence in tile): a 32-bit offset from a scalar base.
    // (The tile's geometry comes back from LDS, the outputs' addresses from the parameter block: none of it waited in registers.)
    VPT_KARG_FENCE(P);
    const uint64_t obase = uint64_t(wave_uniform(M.geo[0])) | (uint64_t(wave_uniform(M.geo[1])) << 32);
    const int32_t c_off_c = int32_t(wave_uniform(M.geo[2]));
    const uint32_t own_lo_c = wave_uniform(M.geo[3]), own_hi_c = wave_uniform(M.geo[4]);
    const uint64_t total_b = P->total_chars - P->n_sent;         // boundaries of the call (or the caller's upper bound of them)
    const uint32_t o_lim = obase >= total_b ? 0u : (total_b - obase > 0xFFFFFFFFull ? 0xFFFFFFFFu : uint32_t(total_b - obase));
    int32_t* const sc = P->scores ? P->scores + obase : nullptr;
    uint8_t* const lb = P->labels ? P->labels + obase : nullptr;
    const int32_t bias = P->bias;
    const uint32_t post = P->post;
#pragma unroll
    for (int k = 0; k < kPerThread; ++k) {
        const uint32_t p = kPad + uint32_t(tid) + uint32_t(k) * kThreads;   // p + 1 < kSymSlots; zero past the tile
        if (wbase + uint32_t(k) * kThreads + kPad + 1 >= flat_len) break;   // wave-uniform
        const uint32_t x = L.sym[p], x2 = L.sym[p + 1];
        int32_t y = bias + L.score[p];
        if ((x & kCpMask) == 0 || (x2 & kCpMask) == 0 || p < own_lo_c || p >= own_hi_c) continue;
        if (TM >= 1 && TM <= 3) {
            uint32_t id = 0;  // window t[b-W+1 .. b+W], 3 bits each (boundary_scorer_cache.rs:59-81)
#pragma unroll
            for (int i = 1 - TM; i <= TM; ++i) id = (id << 3) | (M.typ[int(p) + i] & 7u);
            y += P->type_table[id];
        }
        const uint32_t o = uint32_t(int32_t(p) - c_off_c) - (kPad + 1) * ((x >> 19) & 1023u);
        if (o >= o_lim) { err |= kErrBadOffsets; continue; }  // only with offsets that do not match the text
        if (sc) sc[o] = y;
        uint32_t label = y > 0 ? 1u : 0u;
        if (lb) {
            if (post) {   // wave-uniform: KyteaWsConstFilter / SplitLinebreaksFilter on the label
                const uint32_t t1 = (x >> 16) & 7u, t2 = (x2 >> 16) & 7u;
                if (t1 == t2 && ((post >> t1) & 1u) && t1 != 0 && t1 != 7) label = 0;
                if ((post & 0x80u) && ((x | x2) & kSymLinebreak)) label = 1;
            }
            lb[o] = uint8_t(label);
        }
    }
    tmark = phase_mark(prof, 6, tmark);           // boundaries out

    if (err) atomicOr(P->status, err);
    phase_mark(prof, 7, tmark);
}

// ------------------------------------------------------------------------------------------------------------
// The tiles.  F(i) = ooff[i] + i * (1 + pad) is the flat position of sentence i's first separator; its chars follow at F(i) + pad
// (pad = the separator slots of the kernel instance, pk_pad(wl)).
// ------------------------------------------------------------------------------------------------------------
__host__ __device__ constexpr uint32_t stage_bytes(uint32_t cap) { return cap * 4 + 15; }   // what a tile may stage besides its alignment head

__device__ __forceinline__ void put_desc(TileDesc* out, const TileDesc& d) {
    const uint4* s = reinterpret_cast<const uint4*>(&d);
    uint4* o = reinterpret_cast<uint4*>(out);
    o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; o[3] = s[3];
}

// ---- CUT anywhere.  Index of the text: lead bytes per 256-byte block, so that a tile can start in the middle of a sentence of any
// length.  Blocks are absolute (block b = bytes [256 b, 256 b + 256) of `text`); a superblock is 256 blocks (64 KB).
// cut_local[b - first block] = leads in the superblock's earlier blocks, cut_super[s] = leads in the earlier superblocks.
// (Block positions are relative to `text` rounded DOWN to 16 bytes -- `mis` = what was rounded off -- so that every load is aligned.)
__global__ __launch_bounds__(256) void cut_count_kernel(const uint8_t* __restrict__ text, uint32_t mis, const uint64_t* __restrict__ boff, uint64_t n_sent,
                                                        uint32_t* __restrict__ cut_local, uint64_t* __restrict__ cut_super) {
    __shared__ uint32_t cnt[256];
    __shared__ uint32_t wsum[4];
    text -= mis;
    const uint64_t T0 = boff[0] + mis, T1 = boff[n_sent] + mis;
    const uint64_t sb0 = T0 >> 16, sb = sb0 + blockIdx.x;
    if ((sb << 16) >= T1) return;
    const int tid = threadIdx.x, lane = tid & 63;
    for (int it = 0; it < 16; ++it) {   // 4 KB per trip: a lane takes 16 bytes, a row of 16 lanes one block
        const uint64_t at = (sb << 16) + uint64_t(it) * 4096 + uint64_t(tid) * 16;
        uint32_t n = 0;
        if (at + 16 > T0 && at < T1) {
            const uint4 v = *reinterpret_cast<const uint4*>(text + at);   // 16-byte aligned when `text` is; bytes outside the batch are masked
            const uint32_t lo = at < T0 ? uint32_t(T0 - at) : 0u, hi = T1 - at < 16 ? uint32_t(T1 - at) : 16u;
            const uint32_t vm = ((1u << hi) - 1u) & ~((1u << lo) - 1u);
            n = uint32_t(__popc(lead_mask16(v) & vm));
        }
        n += uint32_t(__builtin_amdgcn_update_dpp(0, int(n), 0x111, 0xF, 0xF, false));  // row_shr:1 .. 8: lane 15 of a row holds the row's sum
        n += uint32_t(__builtin_amdgcn_update_dpp(0, int(n), 0x112, 0xF, 0xF, false));
        n += uint32_t(__builtin_amdgcn_update_dpp(0, int(n), 0x114, 0xF, 0xF, false));
        n += uint32_t(__builtin_amdgcn_update_dpp(0, int(n), 0x118, 0xF, 0xF, false));
        if ((lane & 15) == 15) cnt[it * 16 + (tid >> 4)] = n;
    }
    __syncthreads();
    const uint32_t mine = cnt[tid];
    const uint32_t incl = wave_inclusive_scan(mine);
    if (lane == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (int k = 0; k < 4; ++k) { if (k < (tid >> 6)) before += wsum[k]; total += wsum[k]; }
    cut_local[uint64_t(blockIdx.x) * 256 + uint32_t(tid)] = before + incl - mine;
    if (tid == 0) cut_super[blockIdx.x + 1] = total;   // scanned in place by the next kernel
}
// cut_super[s] -> leads in the superblocks before s (exclusive), one workgroup; n_super_max bounds the array
__global__ __launch_bounds__(256) void cut_scan_kernel(uint32_t mis, const uint64_t* __restrict__ boff, uint64_t n_sent, uint64_t* __restrict__ cut_super, uint64_t n_super_max) {
    __shared__ uint32_t wsum[4];
    __shared__ uint64_t carry_s;
    const uint64_t T0 = boff[0] + mis, T1 = boff[n_sent] + mis;
    uint64_t n_super = T1 > T0 ? ((T1 - 1) >> 16) - (T0 >> 16) + 1 : 0;
    if (n_super > n_super_max) n_super = n_super_max;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) { carry_s = 0; cut_super[0] = 0; }
    __syncthreads();
    for (uint64_t base = 1; base <= n_super; base += 256) {
        const uint64_t i = base + uint64_t(tid);
        const uint32_t v = i <= n_super ? uint32_t(cut_super[i]) : 0u;   // a superblock holds at most 65536 leads
        const uint32_t incl = wave_inclusive_scan(v);
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        uint64_t before = carry_s;
        for (int k = 0; k < w; ++k) before += wsum[k];
        if (i <= n_super) cut_super[i] = before + incl;   // inclusive over the counts = exclusive for superblock i
        __syncthreads();
        if (tid == 255) carry_s = before + incl;
        __syncthreads();
    }
}

struct CutIndex {
    const uint32_t* local;
    const uint64_t* super;
    uint64_t blk0;      // first block of the first superblock
    uint64_t n_blocks;  // blocks the index covers
    __device__ __forceinline__ uint64_t leads_before(uint64_t b) const {   // leads in front of block b (b - blk0 <= n_blocks)
        const uint64_t r = b - blk0;
        if (r >= n_blocks) return super[n_blocks >> 8];                      // the batch's total (n_blocks is a multiple of 256)
        return super[r >> 8] + local[r];
    }
};
// the last block b in [lo, hi] with leads_before(b) <= G, or lo - 1 when there is none
__device__ __forceinline__ uint64_t last_block_le(const CutIndex& X, uint64_t lo, uint64_t hi, uint64_t G) {
    if (lo > hi || X.leads_before(lo) > G) return lo - 1;
    uint64_t a = lo, b = hi;   // leads_before(a) <= G
    // chars are spread almost evenly over a sentence's bytes: start where the straight line says and gallop
    const uint64_t ga = X.leads_before(lo), gb = X.leads_before(hi);
    if (gb <= G) return hi;
    uint64_t g = gb > ga ? lo + uint64_t((unsigned __int128)(G - ga) * (hi - lo) / (gb - ga)) : lo;
    if (g > hi) g = hi;
    if (X.leads_before(g) <= G) {
        a = g;
        for (uint64_t w = 1;; w <<= 2) { const uint64_t q = a + w < hi ? a + w : hi; if (X.leads_before(q) > G) { b = q; break; } a = q; if (q == hi) return hi; }
    } else {
        b = g;
        for (uint64_t w = 1;; w <<= 2) { const uint64_t q = b > lo + w ? b - w : lo; if (X.leads_before(q) <= G) { a = q; break; } b = q; }
    }
    while (b - a > 1) { const uint64_t mid = a + (b - a) / 2; if (X.leads_before(mid) <= G) a = mid; else b = mid; }
    return a;
}

// One thread per tile.  Tile t writes the boundaries at flat positions [t * TF, (t + 1) * TF); its window starts halo_left + pad in front
// (LDS position 0) and is cap_eff long.
__global__ __launch_bounds__(256) void assign_tiles_cut_kernel(const uint64_t* __restrict__ boff, const uint64_t* __restrict__ ooff, uint64_t n_sent,
                                                               CutGeometry Gm, uint32_t n_tiles, const uint32_t* __restrict__ cut_local,
                                                               const uint64_t* __restrict__ cut_super, uint64_t n_super_max, TileDesc* __restrict__ tiles,
                                                               uint32_t* __restrict__ ctrl) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t == 1) ctrl[1] = 0;
    if (t >= n_tiles) return;
    TileDesc d{};
    const uint64_t kPad = Gm.pad;
    const uint64_t step = 1 + kPad;
    auto F = [&](uint64_t i) { return ooff[i] + i * step; };
    const uint64_t total_flat = F(n_sent);
    const uint64_t own0 = uint64_t(t) * Gm.tile_flat;
    if (own0 >= total_flat || n_sent == 0) { put_desc(tiles + t, d); return; }
    const int64_t base = int64_t(own0) - int64_t(Gm.halo_left) - int64_t(kPad);   // global flat position of LDS position 0
    const uint64_t mis = Gm.mis;   // block positions are relative to `text` rounded down to 16 bytes
    const uint64_t T0 = boff[0] + mis, T1 = boff[n_sent] + mis;
    CutIndex X{cut_local, cut_super, (T0 >> 16) << 8, 0};
    {
        uint64_t n_super = T1 > T0 ? ((T1 - 1) >> 16) - (T0 >> 16) + 1 : 0;
        if (n_super > n_super_max) n_super = n_super_max;
        X.n_blocks = n_super << 8;
    }
    // ---- where the staged text starts: the sentence that holds the first flat position the window keeps
    const uint64_t f_need = base + int64_t(kPad) > 0 ? uint64_t(base + int64_t(kPad)) : 0;
    const uint64_t i_lo = first_sentence_at(ooff, n_sent, step, f_need + 1) - 1;          // the last sentence with F <= f_need (F(0) = 0)
    const uint64_t F_lo = F(i_lo), G_lo = ooff[i_lo] + i_lo, B_lo = boff[i_lo], B_lo1 = boff[i_lo + 1];
    const uint64_t k_need = f_need > F_lo + kPad ? f_need - (F_lo + kPad) : 0;          // chars of it in front of the window
    uint64_t byte0 = B_lo, g0 = G_lo;
    if (k_need > 32 && B_lo1 > B_lo) {   // deep inside a long sentence: the last block boundary at or before the first char wanted
        const uint64_t b_first = (B_lo + mis + 255) >> kCutBlockShift, b_last = (B_lo1 + mis - 1) >> kCutBlockShift;
        const uint64_t b = last_block_le(X, b_first, b_last, G_lo + k_need);
        if (b + 1 != b_first && (b << kCutBlockShift) > B_lo + mis) { byte0 = (b << kCutBlockShift) - mis; g0 = X.leads_before(b); }
    }
    // ---- where it ends: the window's last position, or the batch's
    const uint64_t f_end = uint64_t(base + int64_t(Gm.cap_eff));                         // exclusive
    const bool last = f_end >= total_flat;
    const uint64_t i_hi = last ? n_sent : first_sentence_at(ooff, n_sent, step, f_end);  // sentences below i_hi start inside the window
    uint64_t byte1 = boff[i_hi];
    if (!last && i_hi > 0) {
        const uint64_t ie = i_hi - 1, Fe = F(ie), Be = boff[ie], Be1 = boff[ie + 1];
        const uint64_t k_end = f_end > Fe + kPad ? f_end - (Fe + kPad) : 0;              // chars of the last sentence the window holds
        if (k_end < ooff[ie + 1] - ooff[ie] + 1 && Be1 > Be) {
            const uint64_t b_first = (Be + mis + 255) >> kCutBlockShift, b_last = (Be1 + mis - 1) >> kCutBlockShift;
            const uint64_t b = last_block_le(X, b_first, b_last, ooff[ie] + ie + k_end);
            // block b holds (or precedes) the first char past the window: the text up to the end of block b covers the window
            const uint64_t cut = (b + 1 == b_first ? (b_first << kCutBlockShift) : ((b + 1) << kCutBlockShift)) - mis;
            if (cut < byte1) byte1 = cut;
        }
    }
    const uint64_t flat_len = last ? total_flat - uint64_t(base) : uint64_t(Gm.cap_eff);   // (base < total_flat: own0 < total_flat)
    if (byte1 <= byte0 || byte1 - byte0 > uint64_t(stage_bytes(Gm.cap)) || flat_len > uint64_t(Gm.cap) || i_hi - i_lo > 1023 || B_lo1 < B_lo) {
        atomicOr(ctrl, kErrBadOffsets);   // only with offsets that do not match the text (4 bytes per char at most)
        put_desc(tiles + t, d);
        return;
    }
    d.i0 = uint32_t(i_lo); d.nsent = uint32_t(i_hi - i_lo);
    d.byte0_lo = uint32_t(byte0); d.byte0_hi = uint32_t(byte0 >> 32); d.nbytes = uint32_t(byte1 - byte0);
    d.c_off = int32_t(int64_t(g0 + kPad * (i_lo + 1)) - base);   // char G of sentence i sits at global flat G + kPad * (i + 1)
    d.sib0 = byte0 > B_lo ? 0 : -1;
    d.own_lo = Gm.halo_left + uint32_t(kPad);
    d.own_hi = d.own_lo + Gm.tile_flat < uint32_t(flat_len) ? d.own_lo + Gm.tile_flat : uint32_t(flat_len);
    d.flat_len = uint32_t(flat_len);
    d.g0_lo = uint32_t(g0); d.g0_hi = uint32_t(g0 >> 32);
    d.expect_chars = 0xFFFFFFFFu; d.strict_end = last ? 1u : 0u;
    put_desc(tiles + t, d);
}

}  // namespace

// The instance that scores a predictor: its row window and where its type scores come from, or nothing (the general kernels).
static bool fast_instance(const ScoreParams& P, int* wl_out, int* tm_out) {
    if (!P.pk.present || !P.cid || P.pk.wl < 3 || P.pk.wl > uint32_t(kMaxWindow) || P.pad != pk_pad(int(P.pk.wl))) return false;
    if (!P.ct.present || P.ct.uni_n != kUniDirectChars || P.ct.window != int32_t(P.pk.wl)) return false;   // the rows marked wide come from the general tables,
    if (P.ct.stride_dw != uint32_t((2 + 2 * P.pk.wl + 3) & ~3u) || P.ct.uni_dw != uint32_t((2 * P.pk.wl + 3) & ~3u)) return false;   // whose geometry the kernel knows (GeneralGeom)
    const int wl = int(P.pk.wl);
    int tm;
    if (P.type_kind == kTypeNone) tm = 0;
    else if (P.pk.trow_mode != kTypeRowsNone && !(P.force_window_table && P.type_kind == kTypeWindowTable && wl == 3)) tm = kTypeRows;
    else if (P.type_kind == kTypeWindowTable && wl == 3 && P.type_window >= 1 && P.type_window <= 3) tm = P.type_window;
    else return false;   // type n-grams outside the row forms with a window above 3 (or a row window above 3): the general kernels
    if (tm == kTypeRows) {
        if (P.pk.trow_mode == kTypeRowsLds ? P.pk.trow_levels != 3 : (P.pk.trow_levels < 3 || P.pk.trow_levels > uint32_t(kMaxTypeRowLevels))) return false;
    }
    if (wl_out) *wl_out = wl;
    if (tm_out) *tm_out = tm;
    return true;
}
bool fast_path_supported(const ScoreParams& P) { return fast_instance(P, nullptr, nullptr); }
int fast_path_cap(const ScoreParams& P) { return fast_cap(int(P.pk.wl)); }
int fast_path_wg(const ScoreParams& P) { return fast_wg(int(P.pk.wl)); }

template <int WL>
static size_t lds_bytes_for(int tm, uint32_t trow_mode) {
    using T = FastLdsT<WL>;
    constexpr size_t kSlots16 = size_t((FastGeom<WL>::kSymSlots + 15) & ~15);
    if (tm == kTypeRows) return offsetof(T, typ) + (trow_mode == kTypeRowsLds ? sizeof(uint4) * kTrowCount * FastGeom<WL>::kTrowQ : size_t(16));
    return offsetof(T, typ) + (tm == 0 ? size_t(16) : kSlots16);
}
size_t score_tiles_fast_lds_bytes(const ScoreParams& P) {
    int wl = 3, tm = 0;
    if (!fast_instance(P, &wl, &tm)) return 0;
    switch (wl) {
        case 3: return lds_bytes_for<3>(tm, P.pk.trow_mode);
        case 4: return lds_bytes_for<4>(tm, P.pk.trow_mode);
        case 5: return lds_bytes_for<5>(tm, P.pk.trow_mode);
        case 6: return lds_bytes_for<6>(tm, P.pk.trow_mode);
        case 7: return lds_bytes_for<7>(tm, P.pk.trow_mode);
        default: return lds_bytes_for<8>(tm, P.pk.trow_mode);
    }
}

// The instances: row window 3 with every type source and the diagnostics build; the wider windows with type rows or none.
template <int WL>
static hipError_t launch_wide(const ScoreParams& P, int tm, uint32_t n_tiles, size_t lds, hipStream_t stream) {
    if (tm != kTypeRows && tm != 0) return hipErrorInvalidValue;
    if (tm == kTypeRows) hipLaunchKernelGGL((score_tiles_fast_kernel<WL, kTypeRows, false>), dim3(n_tiles), dim3(kThreads), lds, stream, P);
    else hipLaunchKernelGGL((score_tiles_fast_kernel<WL, 0, false>), dim3(n_tiles), dim3(kThreads), lds, stream, P);
    return hipGetLastError();
}

hipError_t launch_score_tiles_fast(const ScoreParams& P, uint32_t n_tiles, hipStream_t stream) {
    int wl = 3, tm = 0;
    if (!fast_instance(P, &wl, &tm)) return hipErrorInvalidValue;
    const size_t lds = score_tiles_fast_lds_bytes(P);
    const bool dbg = P.debug != 0 || P.prof != nullptr;
    switch (wl) {
        case 4: return launch_wide<4>(P, tm, n_tiles, lds, stream);
        case 5: return launch_wide<5>(P, tm, n_tiles, lds, stream);
        case 6: return launch_wide<6>(P, tm, n_tiles, lds, stream);
        case 7: return launch_wide<7>(P, tm, n_tiles, lds, stream);
        case 8: return launch_wide<8>(P, tm, n_tiles, lds, stream);
        default: break;
    }
#define VPT_LAUNCH_FAST(TM_)                                                                                                                  \
    if (dbg) hipLaunchKernelGGL((score_tiles_fast_kernel<3, TM_, true>), dim3(n_tiles), dim3(kThreads), lds, stream, P);                      \
    else hipLaunchKernelGGL((score_tiles_fast_kernel<3, TM_, false>), dim3(n_tiles), dim3(kThreads), lds, stream, P);                         \
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
    return hipGetLastError();
}

void cut_index_entries(uint64_t total_chars_bound, size_t* n_local, size_t* n_super) {
    // a char takes at most 4 bytes; the batch's text may start and end inside a superblock
    const uint64_t supers = ((total_chars_bound * 4) >> 16) + 2;
    *n_super = size_t(supers + 1);
    *n_local = size_t(supers << 8);
}

hipError_t launch_assign_tiles_cut(const ScoreParams& P, const CutGeometry& G, uint32_t n_tiles, uint64_t total_chars_bound, uint32_t* cut_local,
                                   uint64_t* cut_super, TileDesc* tiles, uint32_t* ctrl, hipStream_t stream) {
    size_t n_local = 0, n_super = 0;
    cut_index_entries(total_chars_bound, &n_local, &n_super);
    const uint64_t supers = uint64_t(n_super - 1);
    hipLaunchKernelGGL(cut_count_kernel, dim3(uint32_t(supers)), dim3(256), 0, stream, P.text, G.mis, P.boff, P.n_sent, cut_local, cut_super);
    hipLaunchKernelGGL(cut_scan_kernel, dim3(1), dim3(256), 0, stream, G.mis, P.boff, P.n_sent, cut_super, supers);
    hipLaunchKernelGGL(assign_tiles_cut_kernel, dim3((n_tiles + 1 + 255) / 256), dim3(256), 0, stream, P.boff, P.ooff, P.n_sent, G, n_tiles, cut_local, cut_super,
                       supers, tiles, ctrl);
    return hipGetLastError();
}

}  // namespace vpt
