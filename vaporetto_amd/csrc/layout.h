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
// PACKED TABLES (specialised kernel, kernels_fast.hip).  Every model a Vaporetto / KyTea trainer produces has weights quantised
// to 16 bits (trainer.rs:18,383-397; kytea_model.rs:72-79); the distributed ones have char window 3, and
// train/src/main.rs:33-51 lets --charw / --typew be anything.  For such models -- any pattern chars (at most 65 533 distinct ones), windows up to
// 8 -- the same all-matches information is also emitted as a DOUBLE-ARRAY TRIE over the first three symbols of the patterns,
// walked from every start position.  What shapes it is what bounds the kernel on MI355X (profiles/r02_*): the vector L1's address
// pipeline -- a lane's 16-byte load costs about 1.4 ns / 256 CUs of it and every distinct line it touches another 2.7 -- so a
// position should issue few loads to few lines, and nothing it does not need.  One position = one unigram node, one bigram node
// and, when the bigram's filter says the third char may continue it, one trigram node, each address computed from the node before
// it (Tarjan-Yao row displacement: child slot = parent's base + child symbol, so there is no hash, no seed table and no probing; a
// node names its parent so that a lookup which lands on somebody else's node knows it).  The three dependent loads of a position
// are software-pipelined over three iterations of the kernel's main loop, so every iteration still waits once.
//
//   ROW WINDOW  wl = max(3, W_c, W_t): the window the rows of the packed tables are laid out for.  A pattern of n symbols that
//          starts at s touches the boundaries s + row_lo(n, wl) .. s + row_hi(n, wl); a model with narrower windows has its
//          weights at the boundaries they belong to and zeros around them, so windows 1..3 share the nodes (and the kernel
//          instance) of window 3 -- 16-byte unigram, 32-byte bigram, 16-byte trigram nodes: 4 loads to 3 lines per position --
//          and every wider window has node sizes of its own (pk_*_dw below), with the same protocol.
//   ids    pattern symbols are renumbered 1 .. n_alpha (the alphabet of the model: every char of every n-gram and dictionary
//          word), the char that the most pattern symbols are first: a small id is a frequent char.  The kernel's char
//          classification table (cid, below) maps a text char to its id, 0xFFFF (kNoId) when no pattern contains it.  0 stays
//          "outside the sentence".
//   cinfo  65536 words per mode (plain / through KyteaFullwidthFilter): id | CharacterType << 16 | linebreak << 29
//   uni    n_alpha + 2 nodes of pk_uni_dw(wl) dwords indexed by id (node 0 and the last one are zero: separators, chars outside
//          the alphabet): 2 wl signed 18-bit fields (boundaries s-wl .. s+wl-1) from bit 0; behind them (bit 36 wl) B1 in 19 bits,
//          the base of this char's bigram nodes in units of (1 << bi_shift) nodes, and the wide flag (kUniWideBit when wl = 3)
//   bi     bigram nodes of pk_bi_dw(wl) dwords, node (id1, id2) at slot (B1[id1] << bi_shift) + id2 of its first char's displaced row:
//            key = id1 | id2 << 16 (0 = free); the bigram row: 2 wl - 1 signed 19-bit fields (boundaries s-wl+1 .. s+wl-1) and,
//            behind them, the wide flag (the row has a value outside its fields: it comes from the general tables);
//            B2 = base of this prefix's trigram nodes: node (id1, id2, id3) at slot B2 + id3 (mod 2^32); a 64-bit filter over
//            packed_filter_bit(id3) of the children (0: none).
//            wl = 3: dword 0 key, 1..3 row, 4 B2, 5..6 filter, 7 number of children (statistics)
//            wl > 3: dword 0 key, 1 B2, 2..3 filter, 4.. row  (the first 16 bytes decide, the rest is the row)
//   tri    trigram nodes of pk_tri_dw(wl) dwords: (slot of the parent bigram node + 1) | cflags << 28 (0 = free; cflags: kPkWide),
//          2 wl - 2 weights of 16 bits (boundaries s-wl+2 .. s+wl-1), `kids`: mini-table ref of its depth-4 children in `deep`
//          (0: none) in the low 27 bits.  A node = a 3-char pattern and/or the 3-char prefix of longer ones.  An 8-bit FILTER over
//          packed_kid_filter_bit(id) of those children (round 5) sits in what the two words leave free -- bits 27..31 of `kids`, bits 28..30
//          of dword 0 --: most trigram prefixes have a child or two, the char behind most trigrams in a text is none of them, and a trie step
//          that the filter turns away costs the kernel nothing where a queued one costs two 64-byte entry reads.
//            wl = 3: dword 0 parent, 1..2 weights, 3 kids          wl > 3: dword 0 parent, 1 kids, 2.. weights
//   deep   64-byte entries for the trie below depth 3.  An entry stands for a child symbol PLUS the chain of up to 8
//          further symbols that must follow it (nodes with a single child and no row of their own are compressed
//          away), and ends at a node of depth m:
//            dword 0      id | dflags<<16 | nskip<<24              dword 1   kids (mini-table of the end node's children)
//            dwords 2..5  the nskip further ids, 16 bits each
//            dwords 8..14 the row of the end node's pattern: row_len(m, wl) weights i16 from boundary s + row_lo(m, wl) (wl = 3:
//                         m + 1 weights from s - 1), inline when there are at most 14 and every value fits i16 (kPkHasRow);
//                         otherwise kPkExtRow and dword 8 = offset of an i32 row in `xrows`.
//   mini-table ref = base << 5 | log2(size): `size` consecutive entries of `deep` holding the children of ONE
//          node (so a hot node's children are contiguous and cache-hot together); entry index
//          packed_mini_slot(id, ref), linear probing inside the mini-table, at most `size` probes,
//          an entry with dword 0 == 0 ends the search.  Entry 0 of the arena is unused so that ref 0 = none.
//   cpid   n_alpha + 2 words: the code point of an id (only the rare kPkWide replay needs it: the general tables are
//          keyed by code points).
//   xcid   the alphabet's chars OUTSIDE the BMP (UniDic-derived dictionaries have a few: U+20B9F, U+29E3D ..; any `String` is a
//          pattern to the reference, char_scorer/boundary_scorer.rs:56-89): entry 0 = {log2(entries), 0}, then 2^bits entries
//          {code point, id} (code point 0 = free; at most half of them taken), home = xcid_slot, linear probing.  The kernel's
//          classification reads `cid` for a BMP char and probes here for the others -- a branch a wave takes only when one of its
//          64 chars lies outside the BMP.  No such char in the alphabet: no table (PackedView::off_xcid = 0).
//
// Bases come from first-fit placement at load time (rows with the most children first; a row = the set of child ids of
// one parent): every present key is found by exactly one node read, an absent key lands on a free node or on another
// parent's.  The wide flags mark a row with a value outside its fields (i16 in a trigram node or `deep` entry, 19 bits in a
// bigram row, 18 in a unigram row -- a single weight has 16, a merged row sums an n-gram and the dictionary word of the same
// string): the node keeps zero weights and the row comes from the general tables above.
//
// TYPE ROWS.  The type scores are folded into the same start-position form: the row of (t1, .., tN) holds, for the 2 wl boundaries
// s-wl .. s+wl-1, the totals of the type n-grams t1, (t1,t2), .. that start at s -- code 0 (outside the sentence) ends the prefix,
// exactly what the 8^(2W) window table encodes (boundary_scorer_cache.rs:30-57) -- so a position reads ONE row, added together with
// its unigram row.  Two homes:
//   LDS rows     every type n-gram has at most 3 symbols (the trainer's default) and every total fits 18 bits: 6 * 7 * 7 = 294 rows
//                (type_row_index) of pk_trow_dw(wl) dwords, 2 wl signed 18-bit fields from bit 0; the kernel keeps them in LDS.
//   global rows  n-grams of up to 6 symbols (or totals outside 18 bits): 6 * 7^(N-1) rows of i32, (2 wl + 3) & ~3 dwords each,
//                indexed by type_row_index_n over s .. s+N-1; an L2-resident gather per position.
// Models outside both (an n-gram that contains code 0, or longer than 6) use the window table (W_t <= 3, wl = 3) or the general
// kernels.
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
constexpr int kUniFieldBits = 18;              // unigram node / type row: 2 wl fields
constexpr int kBiFieldBits = 19;               // bigram node: 2 wl - 1 fields
constexpr int kUniBaseBits = 19;
constexpr uint32_t kUniBaseMask = 0x7FFFFu;    // B1 of a unigram node: 19 bits behind its fields, the wide flag behind that
constexpr uint32_t kTriParentMask = 0x0FFFFFFFu, kTriFlagShift = 28;   // dword 0 of a trigram node: parent slot + 1; cflags above
constexpr uint32_t kTriKidsRefMask = 0x07FFFFFFu;                     // its kids word: the mini-table ref (base < 2^22); above it 5 bits of the child filter
constexpr uint32_t kTriKidsMaxBase = 1u << 22;
constexpr uint32_t kCinfoLinebreak = 1u << 29; // cid word: the (scored) char is '\n' or '\r' (where the kernel's symbol word keeps it)

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
// (ids are 16 bits wide: a 16 x 16-bit product is ONE full-rate v_mul_u32_u24 on gfx950, where a 32-bit v_mul_lo_u32 takes four issue
// slots.  The kernels ask for it by name: with a plain `*` the optimiser first drops the `& 0xFFFF` -- the bits looked at do not depend
// on the operand's high half -- and then no longer knows that the operand fits 24 bits.)
VPT_HD uint32_t mul_u16(uint32_t a16, uint32_t b24) {   // the low 32 bits of a 16-bit x 24-bit product
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(a16, b24);
#else
    return uint32_t(uint64_t(a16 & 0xFFFFu) * uint64_t(b24 & 0xFFFFFFu));
#endif
}
#if defined(VPT_AB_OLD_HASH)   /* A/B builds: the 32-bit multiplies of rounds 2-4 */
VPT_HD uint32_t packed_mini_slot(uint32_t sym, uint32_t ref) { return ((sym * kHashMulLo) >> 15) & ((1u << (ref & 31u)) - 1u); }
VPT_HD uint32_t packed_filter_bit(uint32_t sym) { return (sym * kHashMulHi) >> 26; }
#elif defined(VPT_AB_LOW_HASH) /* A/B builds: 16 x 16-bit products */
VPT_HD uint32_t packed_mini_slot(uint32_t sym, uint32_t ref) { return (mul_u16(sym & 0xFFFFu, 0x9E37u) >> 7) & ((1u << (ref & 31u)) - 1u); }
VPT_HD uint32_t packed_filter_bit(uint32_t sym) { return (mul_u16(sym & 0xFFFFu, 0x85EBu) >> 10) & 63u; }   // 0..63
#else
// Fibonacci hashing in 24 bits: the multiplier is 2^24 / phi, so that the TOP bits of the product's low 24 spread consecutive ids (ids go by
// frequency: the chars that matter most are 1, 2, 3 ..) evenly -- taking other bits of the product does not (bits 26 .. 31 of id x 0x85EBCB map
// eight consecutive small ids to ONE filter bit: 9 % more trigram nodes read, profiles/r05_d_*)
VPT_HD uint32_t packed_mini_slot(uint32_t sym, uint32_t ref) { return (mul_u16(sym & 0xFFFFu, 0x9E3779u) & 0xFFFFFFu) >> (24u - (ref & 31u)); }   // mini-tables have at most 2^17 entries
VPT_HD uint32_t packed_filter_bit(uint32_t sym) { return (mul_u16(sym & 0xFFFFu, 0x9E3779u) >> 18) & 63u; }   // 0..63
#endif
// the child filter of a trigram node (header comment, "tri"): 8 bits, bits 0..4 above the kids ref, bits 5..7 in dword 0's bits 28..30
VPT_HD uint32_t packed_kid_filter_bit(uint32_t sym) { return (mul_u16(sym & 0xFFFFu, 0x9E3779u) >> 21) & 7u; }   // 0..7
VPT_HD uint32_t tri_kid_filter(uint32_t dword0, uint32_t kids_word) { return (kids_word >> 27) | (((dword0 >> 28) & 7u) << 5); }
// the alphabet outside the BMP (header comment, "xcid"): `tab` = the section's first dword
VPT_HD uint32_t xcid_slot(uint32_t cp, uint32_t bits) { return (cp * kHashMulLo) >> (32u - bits); }   // bits in 1..31
VPT_HD uint32_t xcid_find(const uint32_t* tab, uint32_t cp) {
    const uint32_t bits = tab[0], mask = (1u << bits) - 1u;
    const uint32_t* const e = tab + 2;
    for (uint32_t i = xcid_slot(cp, bits), n = 0; n <= mask; i = (i + 1u) & mask, ++n) {
        const uint32_t k = e[2u * i];
        if (k == cp) return e[2u * i + 1u];
        if (k == 0u) break;
    }
    return 0xFFFFu;   // kNoId
}

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

VPT_HD bool fits_field(int32_t v, int bits) { return v >= -(1 << (bits - 1)) && v < (1 << (bits - 1)); }

constexpr uint32_t kTypeRowCount = 6 * 7 * 7;
VPT_HD uint32_t type_row_index(uint32_t t1, uint32_t t2, uint32_t t3) { return (t1 - 1u) + 6u * t2 + 42u * t3; }   // t1 in 1..6
// N levels: (t1 - 1) + 6 (t2 + 7 (t3 + 7 (..))); 6 * 7^(N-1) rows
VPT_HD uint32_t type_row_count(int levels) { uint32_t n = 6; for (int i = 1; i < levels; ++i) n *= 7u; return n; }

// geometry of the row of a pattern of n symbols under window W (see the header comment)
VPT_HD int row_lo(int n, int W) { return (n - 1 - W) < -1 ? (n - 1 - W) : -1; }
VPT_HD int row_hi(int n, int W) { return (W - 1) > (n - 1) ? (W - 1) : (n - 1); }
VPT_HD int row_len(int n, int W) { return row_hi(n, W) - row_lo(n, W) + 1; }

// ---- packed node geometry by row window wl = 3 .. 8 (header comment, "ROW WINDOW")
#if defined(__HIPCC__)
#define VPT_HDC __host__ __device__ constexpr
#else
#define VPT_HDC constexpr
#endif
VPT_HDC int pk_uni_fields(int wl) { return 2 * wl; }         // boundaries s - wl     .. s + wl - 1
VPT_HDC int pk_bi_fields(int wl) { return 2 * wl - 1; }      //            s - wl + 1 .. s + wl - 1
VPT_HDC int pk_tri_fields(int wl) { return 2 * wl - 2; }     //            s - wl + 2 .. s + wl - 1
VPT_HDC int pk_uni_dw(int wl) { return wl <= 3 ? 4 : wl <= 6 ? 8 : 16; }     // 36 wl + 20 bits
VPT_HDC int pk_bi_dw(int wl) { return wl <= 3 ? 8 : 16; }
VPT_HDC int pk_tri_dw(int wl) { return wl <= 3 ? 4 : wl <= 7 ? 8 : 16; }
VPT_HDC int pk_trow_dw(int wl) { return wl <= 3 ? 4 : wl <= 7 ? 8 : 12; }    // LDS type rows: 36 wl bits
VPT_HDC int pk_trow_global_dw(int wl) { return (2 * wl + 3) & ~3; }          // global type rows: i32 fields
VPT_HDC int pk_uni_base_bit(int wl) { return kUniFieldBits * 2 * wl; }       // B1 (19 bits), then the wide flag
VPT_HDC int pk_bi_key_dw(int) { return 0; }
VPT_HDC int pk_bi_row_dw(int wl) { return wl <= 3 ? 1 : 4; }                 // first dword of the row's bit string
VPT_HDC int pk_bi_base_dw(int wl) { return wl <= 3 ? 4 : 1; }
VPT_HDC int pk_bi_filter_dw(int wl) { return wl <= 3 ? 5 : 2; }
VPT_HDC int pk_bi_wide_bit(int wl) { return kBiFieldBits * (2 * wl - 1); }   // bit of the row's bit string
VPT_HDC int pk_bi_used_dw(int wl) { return wl <= 3 ? 8 : 4 + (kBiFieldBits * (2 * wl - 1) + 1 + 31) / 32; }   // dwords a lookup reads
VPT_HDC int pk_tri_w_dw(int wl) { return wl <= 3 ? 1 : 2; }                  // first dword of the 16-bit weights
VPT_HDC int pk_tri_kids_dw(int wl) { return wl <= 3 ? 3 : 1; }
VPT_HDC int pk_tri_used_dw(int wl) { return wl <= 3 ? 4 : 2 + (wl - 1); }
VPT_HDC int pk_row_window(int wc, int wt) { return wc > wt ? (wc > 3 ? wc : 3) : (wt > 3 ? wt : 3); }
// the specialised kernel's geometry per row window: separator slots between sentences (>= wl - 1 so that no row reaches the
// next sentence, = wl so that s - wl never leaves the tile), flat positions per tile, workgroups per CU it is built for
VPT_HDC int pk_pad(int wl) { return wl; }
constexpr int kMaxTypeRowLevels = 6;
constexpr uint32_t kTypeRowsNone = 0, kTypeRowsLds = 1, kTypeRowsGlobal = 2;   // PackedView::trow_mode

// `n` bits from bit `bit` of a little-endian dword string, signed (fields) / unsigned (bases)
VPT_HD int32_t bits_signed(const uint32_t* d, int bit, int n) {
    const int q = bit >> 5, r = bit & 31;
    const uint64_t v = (uint64_t(d[q]) | (r + n > 32 ? uint64_t(d[q + 1]) << 32 : 0ull)) >> r;
    return int32_t(uint32_t(v) << (32 - n)) >> (32 - n);
}
VPT_HD uint32_t bits_unsigned(const uint32_t* d, int bit, int n) {
    const int q = bit >> 5, r = bit & 31;
    const uint64_t v = (uint64_t(d[q]) | (r + n > 32 ? uint64_t(d[q + 1]) << 32 : 0ull)) >> r;
    return uint32_t(v) & ((n >= 32) ? 0xFFFFFFFFu : ((1u << n) - 1u));
}

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
    uint32_t off_xcid;              // the alphabet's chars outside the BMP (0: there are none)
    uint32_t n_uni;                 // unigram nodes (ids above n_uni - 1 read the last, all-zero one)
    uint32_t n_tri;                 // trigram nodes (a filter false positive may point past them)
    uint32_t bi_shift;              // bigram slot = (B1 << bi_shift) + id2
    uint32_t present;
    uint32_t wl;                    // the row window the nodes are laid out for (3 .. 8)
    uint32_t trow_mode;             // kTypeRowsNone / kTypeRowsLds / kTypeRowsGlobal
    uint32_t trow_levels;           // type symbols a row is indexed by (3 for LDS rows, 3 .. 6 for global ones)
};

}  // namespace vpt
