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
// For such models -- W = 3, every pattern symbol in [1, 0xFFFE] -- the same all-matches information is also emitted as a
// DOUBLE-ARRAY TRIE over the first three symbols of the patterns, walked from every start position.  What shapes it is
// what bounds the kernel on MI355X (profiles/r02_*): the vector L1's address pipeline -- a lane's 16-byte load costs
// about 1.4 ns / 256 CUs of it and every distinct line it touches another 2.7 -- so a position should issue few
// loads to few lines, and nothing it does not need.  One position = one 16-byte unigram node, one 32-byte bigram node and,
// when the bigram's filter says the third char may continue it, one 16-byte trigram node: 4 loads to 3 lines, each
// address computed from the node before it (Tarjan-Yao row displacement: child slot = parent's base + child symbol, so
// there is no hash, no seed table and no probing; a node names its parent so that a lookup which lands on somebody
// else's node knows it).  The three dependent loads of a position are software-pipelined over three iterations of the
// kernel's main loop, so every iteration still waits once.
//
//   ids    pattern symbols are renumbered 1 .. n_alpha (the alphabet of the model: every char of every n-gram and dictionary
//          word), the char that the most pattern symbols are first: a small id is a frequent char.  The kernel's char
//          classification table (cid, below) maps a text char to its id, 0xFFFF (kNoId) when no pattern contains it.  0 stays
//          "outside the sentence".
//   cinfo  65536 words per mode (plain / through KyteaFullwidthFilter): id | CharacterType << 16 | linebreak << 19
//   uni    n_alpha + 2 rows of 16 bytes indexed by id (row 0 and the last row are zero: separators, chars outside the
//          alphabet): six signed 18-bit fields (boundaries s-3 .. s+2) at bits 0, 18, .., 90; bits 108..126 = B1, the
//          base of this char's bigram nodes in units of (1 << bi_shift) nodes; bit 127 = kUniWideBit
//   bi     32-byte bigram nodes, node (id1, id2) at slot (B1[id1] << bi_shift) + id2 of its first char's displaced row.  (Build option
//          kBiDenseCols > 0: the nodes of the kBiDenseCols most frequent second chars sit in a dense matrix at the head of the array
//          instead, slot id1 * kBiDenseCols + id2, a row's nodes four to a cache line -- an experiment in L2 locality that measured
//          no gain, off by default.)
//            dword 0      key = id1 | id2 << 16 (0 = free)
//            dwords 1..3  the bigram row: five signed 19-bit fields (boundaries s-2 .. s+2) at bits 0, 19, .., 76 of the
//                         96; bit 95 = kBiWideBit (the row has a value outside its fields: it comes from the general tables)
//            dword 4      B2 = base of this prefix's trigram nodes: node (id1, id2, id3) at slot B2 + id3 (mod 2^32)
//            dwords 5, 6  64-bit filter over packed_filter_bit(id3) of the children (0: none)
//            dword 7      number of children (statistics)
//   tri    16-byte trigram nodes: dword 0 = (slot of the parent bigram node + 1) | cflags << 24 (0 = free; cflags: kPkWide),
//          dwords 1, 2 = w0|w1<<16, w2|w3<<16 (boundaries s-1 .. s+2), dword 3 = `kids`: mini-table ref of its depth-4
//          children in `deep` (0: none).  A node = a 3-char pattern and/or the 3-char prefix of longer ones.
//   deep   64-byte entries for the trie below depth 3.  An entry stands for a child symbol PLUS the chain of up to 8
//          further symbols that must follow it (nodes with a single child and no row of their own are compressed
//          away), and ends at a node of depth m:
//            dword 0      id | dflags<<16 | nskip<<24              dword 1   kids (mini-table of the end node's children)
//            dwords 2..5  the nskip further ids, 16 bits each
//            dwords 8..14 the row of the end node's pattern: m+1 weights i16 (boundaries s-1 .. s+m-1), inline when
//                         m+1 <= 14 and every value fits i16 (kPkHasRow); otherwise kPkExtRow and dword 8 = offset of
//                         an i32 row in `xrows`.
//   mini-table ref = base << 5 | log2(size): `size` consecutive entries of `deep` holding the children of ONE
//          node (so a hot node's children are contiguous and cache-hot together); entry index
//          (id * kHashMulLo >> 15) & (size-1), linear probing inside the mini-table, at most `size` probes,
//          an entry with dword 0 == 0 ends the search.  Entry 0 of the arena is unused so that ref 0 = none.
//   cpid   n_alpha + 2 words: the code point of an id (only the rare kPkWide replay needs it: the general tables are
//          keyed by code points).
//
// Bases come from first-fit placement at load time (rows with the most children first; a row = the set of child ids of
// one parent): every present key is found by exactly one node read, an absent key lands on a free node or on another
// parent's.  kPkWide / kBiWideBit / kUniWideBit mark a row with a value outside its fields (i16 in a trigram node or
// `deep` entry, 19 bits in a bigram row, 18 in a unigram row -- a single weight has 16, a merged row sums an n-gram and
// the dictionary word of the same string): the node keeps zero weights and the row comes from the general tables above.
//
// TYPE ROWS.  When every type n-gram has at most 3 symbols and W_t <= 3 (the trainer's defaults), the type scores
// are folded into the same start-position form: trow[type_row_index(t1, t2, t3)] = the six totals (boundaries
// s-3 .. s+2; 18-bit signed fields, three i16 weights always fit) of the type unigram t1, bigram (t1,t2) and
// trigram (t1,t2,t3) -- 6 * 7 * 7 = 294 rows of 16 bytes (t1 = 1..6, t2 and t3 = 0..6) that the
// kernel keeps in LDS, which removes the per-boundary gather from the 8^(2W) window table.  Code 0 (outside the
// sentence) only ever appears as t2/t3 and selects the shorter n-grams, exactly what the window table encodes
// (boundary_scorer_cache.rs:30-57).  Models outside this shape use the window table (W_t <= 3) as before.
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

constexpr uint32_t kPkExtRow = 2u, kPkHasRow = 4u, kPkWide = 8u;  // packed flags (deep entries: dflags; trigram nodes: cflags)
constexpr uint32_t kNoId = 0xFFFFu;            // id of a text char that no pattern contains
constexpr uint32_t kPackedInlineRow = 14;      // weights a `deep` entry holds inline
constexpr uint32_t kPackedMaxSkip = 8;         // further symbols a `deep` entry can require (path compression)
constexpr int kUniFieldBits = 18;              // unigram node: six fields
constexpr int kBiFieldBits = 19;               // bigram node (dwords 1..3): five fields
constexpr uint32_t kUniBaseShift = 12, kUniBaseMask = 0x7FFFFu;   // dword 3 of a unigram node: B1 at bits 12..30
constexpr uint32_t kUniWideBit = 0x80000000u;  // dword 3 of a unigram node: the row is in the general tables (i32)
constexpr uint32_t kBiWideBit = 0x80000000u;   // dword 3 of a bigram node: likewise
constexpr uint32_t kTriParentMask = 0x0FFFFFFFu, kTriFlagShift = 28;   // dword 0 of a trigram node: parent slot + 1; cflags above
constexpr uint32_t kCinfoLinebreak = 1u << 29; // cid word: the (scored) char is '\n' or '\r' (where the kernel's symbol word keeps it)
constexpr uint32_t kCharCacheNoEntry = 0u;     // char cache (layout below): an empty slot
#ifndef VPT_BI_DENSE
#define VPT_BI_DENSE 0    // measured on MI355X (profiles/r03_f_ab_*.jsonl, r03_j_ab_m1.jsonl): 16 / 64 columns 1 % slower than none on M1, 256 columns
#endif                    // 1-2 % faster on M1 and M2 for 32 MB of matrix -- within the noise of either; the displaced rows alone serve
constexpr uint32_t kBiDenseCols = VPT_BI_DENSE;   // second chars (ids below this) whose bigram nodes live in the dense matrix (A/B builds: -D)
// CHAR CACHE (LDS): slot cp & (entries - 1) holds  cp << 16 | CharacterType << 13 | id  of ONE char with that slot index -- the one
// that the most patterns contain -- or 0; only chars of the alphabet with ids below 0x1FFF that are no line breaks are cached.
// UNIGRAM CACHE (LDS): slot id & (nodes - 1) holds the unigram node of one id with that slot index (utag = the id, 0: empty).

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

// packed-table hashes
VPT_HD uint32_t packed_mini_slot(uint32_t sym, uint32_t ref) { return ((sym * kHashMulLo) >> 15) & ((1u << (ref & 31u)) - 1u); }
VPT_HD uint32_t packed_filter_bit(uint32_t sym) { return (sym * kHashMulHi) >> 26; }   // 0..63

// Tag token table (HostTagTables::tok_tab): a surface is hashed from its length and the low 16 bits of its first four chars
// (lo = c0 | c1 << 16, hi = c2 | c3 << 16, zero past the end) -- a lane builds the key with four reads and no loop over the
// token; surfaces that agree in both share a probe sequence and are told apart by the slot.  The top bits are the slot.
VPT_HD uint32_t tag_token_hash_key(uint32_t lo, uint32_t hi, uint32_t len) {
    uint32_t h = lo * kHashMulLo + hi * kHashMulHi + len * 0x7FEB352Du;
    h ^= h >> 15;
    return h * kHashMulLo;
}
// tok_model words (fill_tags -> the writer, a workspace array): tag model index + 1 of the token that ENDS at the char (0: none) in the
// low 24 bits, above them the bytes of its "/tag/tag.." suffix (Sentence::write_tokenized_text, sentence.rs:866-881) -- 255: longer
// than 254, the writer walks the tags itself
constexpr uint32_t kTokModelMask = 0xFFFFFFu, kTokSuffixShift = 24, kTokSuffixLong = 255;
constexpr uint32_t kTagFilterLog2 = 5;    // filter bits per slot of the token table, as a power of two (the filter follows the slots)
constexpr uint32_t kTagTokInline = 1u << 31, kTagTokFast = 1u << 30, kTagTokLenMask = (1u << 30) - 1u;
constexpr uint32_t kTagFiltStride = 32;   // dwords per model in HostTagTables::mfilt

// signed `bits`-wide field number j of a 128-bit little-endian row (unigram rows, bigram rows)
VPT_HD int32_t row_field(uint32_t x, uint32_t y, uint32_t z, uint32_t w, int j, int bits) {
    const uint32_t d[5] = {x, y, z, w, 0u};
    const int bit = bits * j, q = bit >> 5, r = bit & 31;
    const uint64_t v = (uint64_t(d[q]) | (uint64_t(d[q + 1]) << 32)) >> r;
    return int32_t(uint32_t(v) << (32 - bits)) >> (32 - bits);
}
VPT_HD bool fits_field(int32_t v, int bits) { return v >= -(1 << (bits - 1)) && v < (1 << (bits - 1)); }

constexpr uint32_t kTypeRowCount = 6 * 7 * 7;
VPT_HD uint32_t type_row_index(uint32_t t1, uint32_t t2, uint32_t t3) { return (t1 - 1u) + 6u * t2 + 42u * t3; }   // t1 in 1..6

// type row: six 18-bit signed fields packed little-endian into dwords 0..3 (bits 0..107)
VPT_HD int32_t trow_field(uint32_t x, uint32_t y, uint32_t z, uint32_t w, int j) {
    const uint64_t lo = uint64_t(x) | (uint64_t(y) << 32), hi = uint64_t(z) | (uint64_t(w) << 32);
    const int bit = 18 * j;
    uint64_t v = bit < 64 ? (lo >> bit) : (hi >> (bit - 64));
    if (bit < 64 && bit + 18 > 64) v |= hi << (64 - bit);
    return int32_t(uint32_t(v) << 14) >> 14;
}

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

// Device view of the packed tables; passed to the specialised kernel by value.  All arrays live in ONE device
// allocation (below 4 GB), addressed as base + 32-bit byte offset: one scalar base pointer instead of seven.
struct PackedView {
    const unsigned char* base;
    uint32_t off_uni, off_bi, off_tri, off_deep, off_xrows, off_trow, off_cpid;   // byte offsets, 256-byte aligned
    uint32_t off_cc, off_utag, off_urow;   // the LDS-resident caches' contents (off_cc: for the batch's char table mode)
    uint32_t n_uni;                 // unigram nodes (ids above n_uni - 1 read the last, all-zero one)
    uint32_t n_tri;                 // trigram nodes (a filter false positive may point past them)
    uint32_t bi_shift;              // bigram slot = (B1 << bi_shift) + id2
    uint32_t present;
    uint32_t has_trow;              // type rows available (else: window table / none)
};

}  // namespace vpt
