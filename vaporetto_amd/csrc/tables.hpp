// Host-side table compiler: what Predictor::new does in the reference (predictor.rs:450-508), producing the
// position-parallel tables of layout.h instead of automata.
#pragma once
#include <cstdint>
#include <vector>

#include "layout.h"
#include "model.hpp"

namespace vpt {

struct HostPatternTable {
    bool present = false;
    int window = 0;
    int lo[3] = {0, 0, 0}, len[3] = {0, 0, 0};
    uint32_t slots = 0;       // weight slots per short entry (incl. the level-3 extension slot)
    uint32_t stride_dw = 4, uni_dw = 4, uni_n = 0, ext_slot = 0;
    uint32_t short_bits = 4, edge_bits = 4;
    bool has_long = false;
    std::vector<uint32_t> short_tab, uni, edges;
    std::vector<int32_t> wdata;
    // statistics
    uint32_t n_short = 0, n_long_nodes = 0, max_pattern = 0, max_probe_short = 0, max_probe_edge = 0;
    uint32_t n_displaced_short = 0;

    uint64_t bytes() const {
        return 4ull * (short_tab.size() + uni.size() + edges.size() + wdata.size());
    }
};

// Packed tables of the specialised kernel (layout.h, "PACKED TABLES": a double-array trie over the first three symbols).
struct HostPackedTable {
    bool present = false;          // false: the model is not eligible (see build notes in tables.cpp)
    int wl = 3;                    // the row window the nodes are laid out for (layout.h, "ROW WINDOW"): max(3, W_c, W_t)
    std::vector<uint32_t> uni;     // (n_alpha + 2) unigram nodes x pk_uni_dw(wl) dwords, indexed by id
    std::vector<uint32_t> bi;      // pk_bi_dw(wl) dwords per bigram node
    std::vector<uint32_t> tri;     // pk_tri_dw(wl) dwords per trigram node
    std::vector<uint32_t> deep;    // 16 dwords per entry
    std::vector<int32_t> xrows;    // external i32 rows
    std::vector<uint32_t> trow;    // type rows (layout.h, "TYPE ROWS"), empty when the type n-grams do not fit either form
    uint32_t trow_mode = kTypeRowsNone, trow_levels = 0;
    std::vector<uint32_t> cpid;    // n_alpha + 2: id -> code point
    std::vector<uint16_t> id_of;   // 65536: BMP code point -> id (kNoId: no pattern contains it)
    std::vector<uint32_t> xcid;    // the alphabet's chars outside the BMP (layout.h, "xcid"); empty when there are none
    std::vector<uint32_t> hot;     // 0x110000: how many pattern symbols are this code point (ids go by it)
    uint32_t id_for(uint32_t cp) const { return cp < 0x10000u ? uint32_t(id_of[cp]) : (xcid.empty() ? kNoId : xcid_find(xcid.data(), cp)); }
    uint32_t n_alpha = 0, bi_shift = 2;
    // statistics
    uint32_t n_bi = 0, n_tri = 0, n_deep = 0, n_wide = 0;
    uint64_t bytes() const { return 4ull * (uni.size() + bi.size() + tri.size() + deep.size() + xrows.size() + trow.size() + cpid.size() + xcid.size()); }
};

// Tag prediction tables (Predictor::predict_tags, predictor.rs:546-637), see kernels_tags.hip.
//   tok_tab   open addressing over token surfaces, 4 dwords per slot: model index + 1 (0 = empty), len | fast << 30 | inline << 31,
//             then the hashed key (layout.h, tag_token_hash_key): the low 16 bits of the first four chars.  inline = at most 4
//             symbols, all below 0xFFFF: the key IS the surface; any other candidate is verified against `syms` through its
//             model record.  fast = the model fits the record form below.  The LAST model of a repeated token wins, like
//             HashMap::insert in predictor.rs:466-478
//   models    12 dwords per tag model: sym_off, sym_len, char-ngram first/count, type-ngram first/count, bias_off, zlen,
//             slot first/count, 0, 0
//   ngrams    4 dwords per (tag n-gram, rel_position): sym_off, len | rel << 24, w_off, wlen
//   nrec      the same entries again as self-contained 32-byte records for the kernel's fast path (a lane checks a whole
//             n-gram from ONE record and the text window in LDS): dword 0 = len | rel << 8 | kind << 16 (0 char, 1 type) |
//             compact << 17 | min(wlen, 255) << 24, dword 1 = w_off, dwords 2..7 = up to 12 symbols, 16 bits each; compact =
//             the n-gram has at most 12 symbols, all below 0xFFFF.  models[10] bit 0 = every entry of the model is compact,
//             its char and type entries are adjacent, it has at most 16 scores and at most 3 slots: the fast path may take its
//             tokens; models[11] then packs the slots: (candidates | score offset << 5) << (9 * slot).
//   mfilt     32 dwords per tag model, all the fast path reads of a model: the char entries of a model are ordered by rel_position;
//             dwords 2r, 2r+1 = a 64-bit filter over packed_filter_bit(last symbol) of the n-grams with rel_position r (0..3), dword 8 =
//             the four group sizes, 8 bits each -- a token whose text has no candidate last char at offset r skips the whole group;
//             for a model in the record form also dword 9 = its first record, 10 = type entries | scores << 8 | slots << 16,
//             11 = models[11], 12..27 = the bias, zero padded, 28..30 = slot_str of its slots (the first candidate string of each).
//   slots     2 dwords per tag slot: candidate count, offset of its scores in z (slots with >= 2 candidates)
//   slot_str  per tag slot: index of its first candidate in str_off; str_off[k] .. str_off[k+1] = the bytes of candidate
//             string k in str_bytes, ALREADY escaped the way Sentence::write_tokenized_text writes a tag (sentence.rs:871-880)
struct HostTagTables {
    bool present = false;
    uint32_t n_tags = 0, n_models = 0, tok_bits = 4, max_zlen = 0;
    bool use_char = false, use_type = false;   // the scorers exist (char_scorer.rs:98-100, type_scorer.rs:109-111)
    std::vector<uint32_t> tok_tab, models, mfilt, ngrams, nrec, syms, slots, slot_str, str_off;
    std::vector<uint8_t> str_bytes;
    std::vector<int32_t> weights;
};
constexpr uint32_t kTagMaxZ = 1024;   // tag scores per token the kernel keeps in LDS
constexpr uint32_t kTagFastZ = 16;    // ... per token on the fast path
constexpr uint32_t kTagFastSyms = 12; // symbols of an n-gram a 32-byte record holds
constexpr uint32_t kTagFastMaxRel = 4; // chars past a token's last one that the fast path's context window holds (kernels_tags.hip: kCtx - 1 - kCtxBack)

enum TypeKind : int { kTypeNone = 0, kTypeWindowTable = 1, kTypePatternTable = 2 };

struct CompiledModel {
    int32_t bias = 0;
    int pad = 1;                       // zero symbols placed between sentences inside a tile
    HostPatternTable chars;            // char n-grams + dictionary words
    HostPackedTable packed;            // the same patterns in the specialised kernel's layout, when eligible
    int type_kind = kTypeNone;
    int type_window = 0;
    int char_window = 0;               // the model's own char window (chars.window is the one its rows are laid out for: >= 3)
    std::vector<int32_t> type_table;   // 8^(2W) scores indexed by the 3-bit packed type window (cache variant)
    HostPatternTable types;            // used when type_kind == kTypePatternTable
    // counts for vpt_model_info
    uint32_t n_char_ngrams = 0, n_type_ngrams = 0, n_dict_words = 0, n_tag_models = 0;
    bool predict_tags = false;
    HostTagTables tags;                // only with predict_tags and tag models
};

// hash of a token surface (code points) for HostTagTables::tok_tab; the kernel computes the same
uint32_t tag_token_hash(const uint32_t* cps, size_t n);

// KyteaFullwidthFilter (vaporetto_rules/src/string_filters/kytea_fullwidth.rs:13-117): the char a BMP char maps to
uint32_t kytea_fullwidth_host(uint32_t c);

// CharacterType::get_type (sentence.rs:50-67) on the host; used to build the device's BMP class table.
uint8_t char_type_host(uint32_t c);

// Throws ModelError with the reference's message where it defines one.
CompiledModel compile_model(const ModelData& m, bool predict_tags);

}  // namespace vpt
