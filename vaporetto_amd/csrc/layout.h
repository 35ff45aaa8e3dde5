// Table layout shared by the host-side builder (tables.cpp) and the HIP kernels (kernels.hip).
//
// What Predictor::new builds in the reference -- a double-array Aho-Corasick automaton over all char n-grams
// and dictionary words with suffix-merged weight vectors (char_scorer/boundary_scorer.rs:56-89,
// char_scorer.rs:50-78) -- is replaced by position-parallel lookup tables.  Integer addition is associative
// and commutative, so "for every start position, add the weights of EVERY pattern that starts there" gives
// bit-identical scores to the reference's "longest match + pre-merged suffix weights".
//
// Pattern table (one for characters; one for character types when the window table cannot be used):
//
//   level n = 1..3   strings of <= 3 symbols, key = s1 | s2 << 21 | s3 << 42 (absent symbols = 0; a symbol is
//                    never 0: sentence.rs:174-179 bans U+0000 and type ids are 1..6).
//                    Entry = [key_lo, key_hi, slot_0 .. slot_{SC-1}] as dwords, stride padded to 16 bytes,
//                    key 0 = empty.  Entries sit in BUCKETS of kShortBucket = 2 (64 bytes at the common 32-byte
//                    entry): hash -> home bucket, a lookup reads the whole bucket; a key is displaced to the
//                    next bucket with a free slot only when its home bucket is full.  Keys use 63 bits; bit 63
//                    (kDisplacedBit of key_hi) of a bucket's FIRST entry says "some key whose home is this
//                    bucket lives further on".  A lookup that finds neither its key nor that mark in the home
//                    bucket is a definite miss after one 64-byte read (most absent strings end this way); a
//                    continued lookup stops at the first bucket that has a free slot.
//                    slot_j of a level-n entry = total weight this string adds to boundary (start + lo[n] + j):
//                      lo[n]  = min(n-1-W, -1)      hi[n] = max(W-1, n-1)      len[n] = hi[n]-lo[n]+1
//                    (n-gram w[k] lands on boundary start+n-1-W+k, char_scorer/boundary_scorer.rs:63 with
//                     predictor.rs:178-179; dict-word w[k] on boundary start-1+k, boundary_scorer.rs:67-74;
//                     when the same string is both, the two vectors are summed like CharWeightMerger::add).
//                    Level-3 entries keep one extra slot (index ext_slot): id of the trie node that continues
//                    this 3-symbol prefix, or 0.
//   unigram rows     symbols < uni_n index `uni` directly (row = the slots of the level-1 entry, no key).
//   long trie        strings of > 3 symbols: edge table keyed (parent_node << 32 | symbol) ->
//                    {child node, woff}; the row of a node at depth n has len[n] (same formula) entries at
//                    wdata[woff ..]; woff = kNoRow when the node is only a prefix.  Buckets of kEdgeBucket = 4
//                    (64 bytes), same kDisplacedBit rule (parent ids stay below 2^31); kHasKidsBit on the child
//                    id = the child has edges of its own.
//
// PACKED TABLES (specialised kernel, kernels_fast.hip).  Every model a Vaporetto / KyTea trainer produces has
// char window 3, weights quantised to 16 bits (trainer.rs:18,383-397; kytea_model.rs:72-79) and BMP-only patterns.
// For such models -- W = 3, every pattern symbol in [1, 0xFFFE], every merged row value within i16 -- the same
// all-matches tables are also emitted with 16-BYTE entries, so that one lookup is ONE dwordx4 load per lane
// (the kernel is bound by the number of divergent vector-memory lane requests, not by bytes):
//
//   uni    65536 rows, indexed by the char:   {w[0..5] i16 (boundaries s-3 .. s+2), 0}
//   bi     open addressing, key = c1 | c2<<16 {key, w[0..4] i16 (boundaries s-2 .. s+2), flags u16}
//   tri    key = (c1 | c2<<16, c3)            {c1|c2<<16, c3 | flags<<16, w[0..3] i16 (boundaries s-1 .. s+2)}
//          holds every 3-char pattern and the 3-char prefix of every longer one (zero weights).
//   edge   trie of the patterns longer than 3 chars, key = (parent id, sym):
//                                             {parent, sym | flags<<16, row offset (16-byte units), 0}
//          parent id = slot index of the 3-char prefix in `tri`, or kPackedEdgeId | slot index in `edge`.
//   wrows  i16 weight rows of those patterns (a pattern of m chars has m+1 weights, boundaries s-1 .. s+m-1),
//          each row padded to a multiple of 16 bytes.
//
// A key lives in its home slot hash(key) or, if that was taken, in the next free slot (linear probing).  The
// flags of a SLOT carry kPkDisp when some key whose home is this slot lives further on: a lookup that finds
// neither its key nor kPkDisp in the home slot is done after one load; otherwise it is continued (replayed)
// until the key or an empty slot is met.  Empty slot: bi/tri dword 0 == 0, edge dword 1 == 0.
// kPkWide marks a row with a value outside i16 (an n-gram and a dictionary word with the same string can sum past
// 16 bits): a uni/bi/tri slot then keeps zero weights and the row is taken from the general tables above
// (`uni` row flag: dword 3 == kPkWide); an edge's row in `wrows` is then stored as i32.
// A text char >= 0x10000 is mapped to 0xFFFF before lookups: no pattern contains either, so it matches nothing.
#pragma once
#include <cstdint>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#endif

namespace vpt {

constexpr int kMaxWindow = 8;                 // windows above this are rejected (model contract)
constexpr uint32_t kUniDirectChars = 0x10000; // BMP code points index the unigram rows directly
constexpr uint32_t kUniDirectTypes = 256;     // type ids are bytes
constexpr uint32_t kNoRow = 0xFFFFFFFFu;
constexpr uint32_t kDisplacedBit = 0x80000000u;  // in key_hi of a hash slot
constexpr uint32_t kHasKidsBit = 0x80000000u;    // in the child id of an edge
constexpr uint32_t kShortBucket = 2;             // entries per bucket of the short table
constexpr uint32_t kEdgeBucket = 4;              // edges per bucket of the trie edge table
constexpr uint32_t kHashMulLo = 0x9E3779B1u, kHashMulHi = 0x85EBCA77u;

constexpr uint32_t kPkDisp = 1u, kPkHasKids = 2u, kPkHasRow = 4u, kPkWide = 8u;  // flags of a packed slot
constexpr uint32_t kPackedEdgeId = 0x80000000u;                    // parent ids that name an `edge` slot
constexpr uint32_t kPackedNoMatchSym = 0xFFFFu;

#if defined(__HIPCC__)
#define VPT_HD __host__ __device__ __forceinline__
#else
#define VPT_HD inline
#endif

VPT_HD uint64_t short_key(uint32_t s1, uint32_t s2, uint32_t s3) {
    return uint64_t(s1) | (uint64_t(s2) << 21) | (uint64_t(s3) << 42);
}
VPT_HD uint64_t edge_key(uint32_t parent, uint32_t sym) { return (uint64_t(parent) << 32) | sym; }
// two-word multiplicative hash; `shift` = 32 - log2(capacity)
VPT_HD uint32_t hash_slot(uint64_t key, uint32_t shift) {
    return (uint32_t(key) * kHashMulLo + uint32_t(key >> 32) * kHashMulHi) >> shift;
}

// packed-table hashes; `shift` = 32 - log2(capacity)
VPT_HD uint32_t packed_hash1(uint32_t k, uint32_t shift) { return (k * kHashMulLo) >> shift; }
VPT_HD uint32_t packed_hash2(uint32_t a, uint32_t b, uint32_t shift) { return (a * kHashMulLo + b * kHashMulHi) >> shift; }

// geometry of the row of a pattern of n symbols under window W (see the header comment)
VPT_HD int row_lo(int n, int W) { return (n - 1 - W) < -1 ? (n - 1 - W) : -1; }
VPT_HD int row_hi(int n, int W) { return (W - 1) > (n - 1) ? (W - 1) : (n - 1); }
VPT_HD int row_len(int n, int W) { return row_hi(n, W) - row_lo(n, W) + 1; }

// Device view of one pattern table; passed to kernels by value.
struct PatternTableView {
    const uint32_t* short_tab;
    const uint32_t* uni;
    const uint32_t* edges;   // 4 dwords per edge slot: key_lo, key_hi, child, woff
    const int32_t* wdata;
    uint32_t short_shift, short_mask;   // hash_slot shift and mask in BUCKETS
    uint32_t edge_shift, edge_mask;
    uint32_t stride_dw;      // dwords per short entry (2 + slots, rounded up to a multiple of 4)
    uint32_t uni_dw;         // dwords per unigram row (rounded up to a multiple of 4)
    uint32_t uni_n;          // symbols below this use `uni`
    uint32_t ext_slot;       // slot of a level-3 entry that holds the continuation node
    int32_t window;
    int32_t lo[3], len[3];
    uint32_t has_long;       // any pattern longer than 3 symbols
    uint32_t debug;          // profiling ablation bits (0 in production)
    uint32_t present;        // 0 = this scorer is None (char_scorer.rs:98-100 / type_scorer.rs:109-111)
};

// Device view of the packed tables; passed to the specialised kernel by value.
struct PackedView {
    const uint32_t *uni, *bi, *tri, *edge, *wrows;   // 16-byte entries (uint4)
    uint32_t bi_shift, bi_mask, tri_shift, tri_mask, edge_shift, edge_mask;
    uint32_t present;
};

}  // namespace vpt
