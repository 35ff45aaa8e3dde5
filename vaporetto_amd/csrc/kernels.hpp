// Kernel-side parameter block and launchers (kernels.hip), used by the C ABI (capi.cpp).
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>

#include "layout.h"
#include "tables.hpp"

namespace vpt {

constexpr int kThreads = 256;                    // 4 waves per workgroup
constexpr int kCap = 2048;                       // flat positions (chars + separators) one LDS tile can hold
constexpr int kTileFlat = 1024;                  // flat positions a tile is cut at (a tile ends with the sentence
                                                 // that crosses the cut, so it needs kCap - kTileFlat of slack)
constexpr int kMargin = 8;                       // zeroed slack past the tile for the s+1, s+2 look-ahead
// The specialised kernel's tile per ROW WINDOW wl (layout.h): fast_cap(wl) flat positions, fast_wg(wl) workgroups per CU.  Window 3
// (every distributed model): 1280 / 8, the best of the geometries measured on MI355X (profiles/r02_c5_ab*.jsonl, r03_d_ab_m1.jsonl;
// LDS-resident caches in front of the char table and the unigram nodes, a dense matrix for the bigram nodes of frequent second
// chars and a stage-major walk were measured in round 3 and are gone: profiles/r03_d/f/h_ab_*.jsonl).  Wider windows have wider
// nodes in flight and wider type rows in LDS: 1024 positions at 6 (5 for window 8) workgroups per CU.  Round 4 measured one more LDS table
// -- the symbol words of ASCII, U+3000..30FF and U+FF00..FFEF, direct-indexed, in the idle W queues during classify -- 2-3 % SLOWER on
// every workload (profiles/r04_f_ctab_*.jsonl; the fill and the range tests cost more than the gathers they save) and tiles of 960 /
// 640 positions (more rounds on a 100 K-sentence batch) 2 % / 13 % slower: out again.
// (-D overrides: A/B builds of tools/build_variants.sh.)
#ifndef VPT_FAST_CAP
#define VPT_FAST_CAP 1280
#endif
#ifndef VPT_FAST_WG
#define VPT_FAST_WG 8
#endif
#ifndef VPT_FAST_CAP_WIDE
#define VPT_FAST_CAP_WIDE 1024
#endif
constexpr int fast_cap(int wl) { return wl <= 3 ? VPT_FAST_CAP : VPT_FAST_CAP_WIDE; }
constexpr int fast_wg(int wl) { return wl <= 3 ? VPT_FAST_WG : wl <= 7 ? 6 : 5; }
constexpr int kFastCapMax = VPT_FAST_CAP > VPT_FAST_CAP_WIDE ? VPT_FAST_CAP : VPT_FAST_CAP_WIDE;
// Two ways of cutting a batch into tiles ("Tiles"):
//   whole sentences   a tile = the sentences whose flat start lies in its range; needs every sentence to fit beside tile_flat
//   cut anywhere      a tile = a range of flat positions plus a halo of max(pattern length) chars on either side, so a sentence
//                     of ANY length is scored by as many tiles as it spans (the reference scores any length in one loop,
//                     char_scorer/boundary_scorer.rs:93-113; its tantivy adapter passes whole documents as one Sentence)
constexpr int fast_whole_max_chars(int wl) { return fast_cap(wl) / 2 - 2 * pk_pad(wl); }   // batches whose longest sentence has more chars are cut
                                                 // anywhere (a whole-sentence tile holds tile_flat >= cap / 2 positions plus the sentence that crosses its end)
constexpr int kCutBlockShift = 8;                // cut tiles find their text through lead-byte counts per 256-byte block
constexpr int kFastStageSlack = 136;             // flat positions a cut tile leaves unused so that its text always fits the staging area
                                                 // (a tile of n chars stages at most 4 n + 2 * 256 bytes; the area holds 4 * cap + 32)
constexpr int kBitmapWords = 4 * kCap / 32 + 4;  // one bit per text byte of a tile (UTF-8: <= 4 bytes per char)

// device status word (OR of bits)
constexpr uint32_t kErrEmptySentence = 1u;    // "text: must contain at least one character" (sentence.rs:181-186)
constexpr uint32_t kErrNulChar = 2u;          // "text: must not contain NULL"               (sentence.rs:174-179)
constexpr uint32_t kErrBadOffsets = 4u;       // out_offsets do not match the text (or invalid UTF-8)
constexpr uint32_t kErrScratchTooSmall = 8u;  // max_sentence_bytes was understated
constexpr uint32_t kErrUnknownLabel = 16u;    // token emission: a label that is neither 0 nor 1
constexpr uint32_t kErrOutputTooSmall = 32u;  // token emission: text_capacity is smaller than the tokenized text


struct ScoreParams {
    PatternTableView ct;        // characters: n-grams + dictionary words
    PatternTableView tt;        // character types, when type_kind == kTypePatternTable
    PackedView pk;              // characters again, as the double-array trie of the specialised kernel (if eligible)
    const int32_t* type_table;  // 8^(2W) window scores, when type_kind == kTypeWindowTable
    const uint8_t* ctype;       // CharacterType of every BMP scalar value (65536 bytes)
    const uint32_t* cinfo;      // only with VPT_FLAG_KYTEA_FULLWIDTH, else nullptr: per BMP scalar value the char it is
                                // scored as (KyteaFullwidthFilter's image) | CharacterType of that char << 16
    const uint32_t* cid;        // specialised kernel: per BMP scalar value id | CharacterType << 16 | linebreak << 19 of the char it is
                                // scored as (layout.h, "ids"; the plain table or the one through KyteaFullwidthFilter)
    int32_t type_window;
    int32_t type_kind;
    int32_t bias;
    int32_t pad;
    const uint8_t* text;
    const uint64_t* boff;       // [S+1] byte offsets
    const uint64_t* ooff;       // [S+1] output (boundary) offsets
    const uint32_t* tile_first; // [n_tiles+1] (general kernel)
    const struct TileDesc* tiles; // [n_tiles] (specialised kernel, cut tiles; nullptr: whole-sentence tiles from tile_first)
    uint64_t n_sent;
    uint32_t tile_flat, n_tiles;
    int32_t* scores;
    uint8_t* labels;
    uint32_t* status;
    uint32_t* slow_list;        // tiles deferred to the global-scratch path
    uint32_t* slow_count;
    unsigned char* scratch;     // slabs for score_slow_kernel
    uint64_t scratch_stride;    // bytes per workgroup slab
    uint32_t scratch_cap;       // flat positions per slab
    uint32_t* cps_out;          // specialised kernel, optional: the batch's chars flat (char g of sentence i at ooff[i] + i + g) as scored
                                // scalar value | CharacterType << 24 -- what decode_chars_kernel makes for the tag kernel
    uint64_t total_chars;       // what cps_out holds: total boundaries + sentences of the call
    uint32_t post;              // label post-filters: bits 1..6 KyteaWsConstFilter per CharacterType, bit 7 SplitLinebreaksFilter
    uint32_t force_window_table;// experiment knob (read when the predictor is made): the 8^(2W) type table although type rows exist
    uint32_t debug;             // profiling ablation bits (VPT_DEBUG_ABLATE env, 0 in production)
    uint64_t* prof;             // per-phase shader-cycle counters (VPT_PROFILE_PHASES env), else nullptr
};

// What a workgroup of the specialised kernel needs to know of its tile, made by the assign kernels (kernels_fast.hip).  The tile's
// text is bytes [byte0, byte0 + nbytes); lead number ci of it (0-based) in tile-local sentence si sits at LDS position
// c_off + ci + pad * si, si = sib0 + (sentence starts at or before its byte); positions outside [pad, flat_len) are not kept.
struct TileDesc {
    uint32_t i0, nsent;         // sentences [i0, i0 + nsent): every sentence with a byte in the staged text
    uint32_t byte0_lo, byte0_hi;
    uint32_t nbytes;
    int32_t c_off;
    int32_t sib0;               // -1: byte0 starts sentence i0; 0: it lies inside sentence i0
    uint32_t own_lo, own_hi;    // the boundaries (and chars) at LDS positions [own_lo, own_hi) are this tile's to write
    uint32_t flat_len;
    uint32_t g0_lo, g0_hi;      // global char number (out_offsets[i] + i + char index) of the first staged lead
    uint32_t expect_chars;      // leads the staged text must hold (whole-sentence tiles), 0xFFFFFFFF: not checked
    uint32_t strict_end;        // a lead at or past flat_len is an error (the text ends with the tile)
    uint32_t rsv0, rsv1;
};
static_assert(sizeof(TileDesc) == 64, "four 16-byte scalar loads");

// tag prediction (kernels_tags.hip); table layouts: HostTagTables in tables.hpp
struct TagParams {
    const uint32_t *tok_tab, *models, *mfilt, *ngrams, *nrec, *syms, *slots;
    const int32_t* weights;
    const uint32_t* cinfo;      // as in ScoreParams
    uint32_t tok_bits, n_tags, use_char, use_type;
    const uint32_t* cps;        // the batch's chars, flat (decode_chars_kernel): scored scalar value | CharacterType << 24
    const uint64_t* ooff;       // [S+1]
    const uint8_t* labels;      // [total boundaries] CharacterBoundary values (0, 1, 2 = Unknown)
    uint64_t n_sent;
    uint64_t total_chars;       // total boundaries + S: what `cps` holds (below 2^32 - 256: capi.cpp)
    const uint32_t *slot_str, *str_off;   // the tag strings' lengths (HostTagTables): what a token's tags take in the tokenized text
    uint32_t n_strings;
    // What fill_tags LEAVES (round 6): one RECORD per token that can have a tag model -- the reference holds None for every other char
    // (predictor.rs:558-573) and so nothing is stored for them -- sorted by the token's last char:
    //   records[k]  = {flat index of the token's last char (2 dwords), its tok_model word (layout.h: tag model + 1 | the bytes its tags take in
    //                  the tokenized text << 24; 0: the token table's filter let the token through but it has no tag model: an empty record),
    //                  slots of its "/tag" suffix (up to the last Some)}
    //                  (between the lookups and the passes: {last char, context clip, tag model + 1 | record form << 31, 0});
    //   rec_tags[k * n_tags + j] = the candidate chosen for slot j, -1 = None;   rec_str[k * n_tags + j] = {where that candidate's string starts in
    //                  the predictor's str_bytes, its length}: the writer copies bytes, it looks nothing up
    // The front end walks the batch in RUNS of `run_sent` consecutive sentences, a wave per run, in order, and numbers the run's CANDIDATES (the
    // token ends the filter lets through) as it meets them: candidate i of run r is record run_pref[r] + i.  run_pref [n_runs + 1]: zero in front of
    // the launches; the front end leaves run r's candidate count in entry r + 1; a chained scan turns it into the exclusive prefix (entry
    // n_runs: all records).  cands [total_chars]: run r's candidates at [first char of r, + count): {char in the run, token length, chars
    // before | after << 8 inside the sentence (clipped to the context), 0} -- plain stores, nothing the front end waits for.
    uint4* records;
    int32_t* rec_tags;
    uint2* rec_str;
    uint64_t* run_pref;
    uint64_t* scan_state;       // scan_part_entries(n_runs) words of that scan, zero in front of the launches
    uint64_t n_runs;
    uint32_t run_sent;
    uint4* cands;
    // the dense arrays of the C ABI, all optional (nullptr: not wanted), indexed by char; the caller has set them to None (-1) / left them alone:
    int32_t* tags;              // [(total boundaries + S) * n_tags] candidate index per slot: written at the last char of a token with a tag model
    int32_t* scores_out;        // [(total boundaries + S) * score_stride]: Predictor::store_tag_scores (predictor.rs:510-514,599-601): there, entries
                                // [0, bias.len()) = the scores the reference keeps in sentence.tag_scores[i]
    int32_t* model_out;         // [total boundaries + S]: there, the index of the tag model (Model::tag_models order)
    uint32_t score_stride;
    uint32_t n_cus;             // the device's CUs: the grids are what it holds at a time
    const uint32_t* summary;    // 2^(kTagSumLog2 - 5) words for the summary of the token filter (tag_filter_summary_kernel writes it, the front end reads it)
};
// sentences of a front-end run for a batch of this shape (about VPT_TAG_RUN_CHARS chars, at most 256 sentences); words of TagParams::summary
uint32_t tag_run_sentences(uint64_t n_sent, uint64_t total_chars);
size_t tag_summary_words();
// `fullwidth`: cinfo is the table that looks through KyteaFullwidthFilter (else the identity: no word of it is read)
hipError_t launch_decode_chars(const uint8_t* text, const uint64_t* boff, const uint64_t* ooff, uint64_t n_sent, uint64_t total_chars,
                               const uint32_t* cinfo, uint32_t* cps, uint8_t* types, uint32_t* status, hipStream_t stream, bool fullwidth);
hipError_t launch_tag_tokens(const TagParams& P, hipStream_t stream);

// token emission (kernels_emit.hip): Sentence::write_tokenized_text, boundary part
struct EmitParams {
    const uint8_t* text;
    const uint64_t* boff;       // [S+1]
    const uint64_t* ooff;       // [S+1]
    const uint8_t* labels;      // [total boundaries] 0 / 1
    uint64_t n_sent, total_boundaries;
    uint8_t* out_text;          // [capacity]
    uint64_t* out_offsets;      // [S+1] byte range of every sentence's tokenized text in out_text
    uint64_t capacity;
    uint32_t* status;
    // "/tag" suffixes (sentence.rs:866-881), from the records the fill_tags call on this workspace left (TagParams); records == nullptr: none
    const uint4* records;
    const uint2* rec_str;       // [records * n_tags]: the strings of a record's suffix
    const uint64_t* run_pref;   // [n_runs + 1]: records in front of run r of `run_sent` sentences
    uint64_t n_runs;
    uint32_t run_sent;
    uint32_t n_tags;
    const uint8_t* str_bytes;
};
// scan_part: workspace of scan_part_entries(n_sent) uint64 (the prefix sum's per-workgroup partials)
size_t scan_part_entries(uint64_t n);
// The writer's launch (emit_flat_kernel): a WORKGROUP per run of `per_block` consecutive sentences (1 .. kEmitFlatMaxBlock)
constexpr uint32_t kEmitFlatMaxBlock = 512;   // (two sentences a thread)
struct EmitFuse {
    uint64_t* state;        // n_blocks + 1 words, ZERO when the kernel starts: the blocks' sizes / positions and the ticket
    uint64_t* clear;        // the state words of the NEXT call (the other of two arrays), zeroed by this one: [0, clear_n)
    uint64_t clear_n, n_blocks;
    uint32_t per_block;
    uint64_t* total_out;    // optional device-writable HOST address that receives the output's total size
    // calls enqueued one after the other on one stream write ONE contiguous text (vpt_tokenize_batch's chunks): a device word that holds the
    // output position in front of this call's text / receives the position behind it; nullptr: the text starts at 0
    const uint64_t* chain_in;
    uint64_t* chain_out;
};
// inclusive prefix sum over offsets[1 .. n] in place, offsets[0] = 0 (a chained scan, one launch; part: scan_part_entries(n) zero words)
hipError_t launch_scan(uint64_t* offsets, uint64_t n, uint64_t* part, uint64_t capacity, uint32_t* status, uint64_t* total_out, hipStream_t stream);
// vpt_expand_tags_batch_device: the dense tags array from the records (None everywhere else)
hipError_t launch_expand_tags(const uint4* records, const int32_t* rec_tags, const uint64_t* n_records, uint32_t n_tags, uint64_t total_chars, int32_t* tags,
                              uint32_t n_cus, hipStream_t stream);
hipError_t launch_emit_tokenized(const EmitParams& P, const EmitFuse& F, hipStream_t stream);
// vpt_count_boundaries on the device: ooff_out[S+1]; *max_chars (atomicMax) = the longest sentence in chars; text_bytes_hint: the batch's text
// bytes when the host knows them (0: not), which sizes the workgroups' shares
hipError_t launch_count_boundaries(const uint8_t* text, const uint64_t* boff, uint64_t n_sent, uint64_t* ooff_out, uint64_t* scan_part, uint32_t* status,
                                   uint32_t* max_chars, uint64_t text_bytes_hint, hipStream_t stream);

size_t score_tiles_lds_bytes();
hipError_t launch_assign_tiles(const uint64_t* ooff, uint64_t n_sent, int pad, uint32_t tile_flat, uint32_t n_tiles,
                               uint32_t* tile_first, uint32_t* ctrl, hipStream_t stream);
// specialised kernel (kernels_fast.hip): packed tables (windows up to 8, BMP, i16), type rows, the type window table (row window 3) or none
bool fast_path_supported(const ScoreParams& P);
int fast_path_cap(const ScoreParams& P);   // flat positions per tile / workgroups per CU of the instance that scores P
int fast_path_wg(const ScoreParams& P);
size_t score_tiles_fast_lds_bytes(const ScoreParams& P);
hipError_t launch_score_tiles_fast(const ScoreParams& P, uint32_t n_tiles, hipStream_t stream);
// its tiles: whole sentences (launch_assign_tiles, as for the general kernel; ScoreParams::tiles stays nullptr) or cut anywhere: lead-byte counts per 256-byte block of the text (cut_local: exclusive inside a superblock of 256 blocks,
// cut_super: exclusive over the superblocks; sized from cut_index_entries), then the tiles
struct CutGeometry { uint32_t tile_flat, halo_left, halo_right, cap_eff, mis /* text pointer & 15 */, pad /* separator slots */, cap /* the kernel's tile */; };
void cut_index_entries(uint64_t total_chars_bound, size_t* n_local, size_t* n_super);
hipError_t launch_assign_tiles_cut(const ScoreParams& P, const CutGeometry& G, uint32_t n_tiles, uint64_t total_chars_bound, uint32_t* cut_local,
                                   uint64_t* cut_super, TileDesc* tiles, uint32_t* ctrl, hipStream_t stream);
hipError_t launch_score_tiles(const ScoreParams& P, int chunks, uint32_t n_tiles, hipStream_t stream);
hipError_t launch_score_slow(const ScoreParams& P, int chunks, uint32_t n_blocks, hipStream_t stream);

}  // namespace vpt
