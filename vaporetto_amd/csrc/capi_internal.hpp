// What the files of the C ABI share (capi.cpp: predictors and workspaces; capi_device.cpp: the device-resident entry points; capi_host.cpp:
// the host-buffer pipelines): the handles' structs, the knobs, and the small helpers -- those in an unnamed namespace, a copy per file.
#pragma once
#include "../../include/vaporetto_hip.h"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "kernels.hpp"
#include "model.hpp"
#include "tables.hpp"

extern thread_local std::string vpt_g_last_error;   // (capi.cpp) the calling thread's message: vpt_last_error

namespace {

vpt_status fail(vpt_status st, const std::string& msg) {
    vpt_g_last_error = msg;
    return st;
}

#define VPT_HIP(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(VPT_RUNTIME_ERROR, std::string("HIP error: ") + hipGetErrorString(e_) + " at " #expr); \
    } while (0)

// Experiment / test knobs from the environment (tools/README.md).  Read ONCE -- the table-level ones when a predictor is made
// (bind_predictor), together with the launch-level ones of the workspaces its host entry points make for themselves (a pooled
// workspace is created lazily, inside predict / fill_tags / tokenize, possibly from several host threads at once: it copies the
// predictor's snapshot); a workspace the CALLER makes (vpt_batch_create) reads the launch-level ones then -- never on the launch
// path: getenv is not thread-safe against setenv, and a drop-in library must not change behaviour under a running host.
struct PredictorKnobs {
    bool force_window_table = false;    // VPT_FORCE_WINDOW_TABLE: the 8^(2W) type table instead of the type rows
    int pipe_lanes = -1;                // VPT_PIPE_LANES (-1: the size rule)
    uint64_t chunk_chars = 0;           // VPT_CHUNK_CHARS (0: the size rule)
    uint64_t tokenize_chunk_bytes = uint64_t(256) << 20;   // VPT_TOKENIZE_CHUNK_BYTES (the tagged pipeline's default; the fused one: an eighth of the batch, at least 4 MB)
    bool tokenize_chunk_bytes_set = false;
};
struct BatchKnobs {
    bool force_generic = false;         // VPT_FORCE_GENERIC
    int force_cut = 0;                  // VPT_FORCE_CUT_TILES: 1 = cut tiles for every batch, -1 = whole-sentence tiles whenever they fit (tests, A/B)
    uint32_t tile_flat = 0;             // VPT_TILE_FLAT: flat positions per tile (tests: cuts at many places; never above what fits)
    uint32_t debug_ablate = 0;          // VPT_DEBUG_ABLATE
    bool profile_phases = false;        // VPT_PROFILE_PHASES
    uint32_t emit_per_block = 0;        // VPT_EMIT_PER_BLOCK: sentences a workgroup of the writer takes (1..512; 0: from the mean sentence length) -- tests: runs of any size
};
PredictorKnobs read_predictor_knobs() {
    PredictorKnobs k;
    k.force_window_table = std::getenv("VPT_FORCE_WINDOW_TABLE") != nullptr;
    if (const char* v = std::getenv("VPT_PIPE_LANES")) k.pipe_lanes = std::max(0, std::atoi(v));
    if (const char* v = std::getenv("VPT_CHUNK_CHARS")) { const long long n = std::atoll(v); if (n > 0) k.chunk_chars = uint64_t(n); }
    if (const char* v = std::getenv("VPT_TOKENIZE_CHUNK_BYTES")) { const long long n = std::atoll(v); if (n > 0) { k.tokenize_chunk_bytes = uint64_t(n); k.tokenize_chunk_bytes_set = true; } }
    return k;
}
BatchKnobs read_batch_knobs() {
    BatchKnobs k;
    k.force_generic = std::getenv("VPT_FORCE_GENERIC") != nullptr;
    if (const char* v = std::getenv("VPT_FORCE_CUT_TILES")) k.force_cut = std::atoi(v);
    if (const char* v = std::getenv("VPT_TILE_FLAT")) k.tile_flat = uint32_t(std::max(0, std::atoi(v)));
    if (const char* v = std::getenv("VPT_EMIT_PER_BLOCK")) k.emit_per_block = uint32_t(std::min(512, std::max(0, std::atoi(v))));
    if (const char* v = std::getenv("VPT_DEBUG_ABLATE")) k.debug_ablate = uint32_t(std::atoi(v));
    k.profile_phases = std::getenv("VPT_PROFILE_PHASES") != nullptr;
    return k;
}

constexpr size_t kTimingRing = 256;     // timed launches remembered per vpt_batch
constexpr size_t kTablePadBytes = 256;  // probes read whole 16-byte chunks; keep the tail of every table readable

// Every table of a predictor lives in ONE device allocation (the arena), each in a section of its own, 256-byte aligned
// and followed by kTablePadBytes of zeros.  With the fixed-size description below (PredictorMeta) the arena IS the compiled
// predictor: it can be written out and read back (vpt_predictor_save / _load: the analogue of Predictor::serialize_to_vec /
// deserialize_from_slice_unchecked, predictor.rs:640-664, in a format of our own) and copied to another GPU device to
// device (vpt_predictor_clone_to_device) without compiling the model again.
enum Section : int {
    kSecCShort, kSecCUni, kSecCEdges, kSecCWdata,              // general char tables
    kSecTShort, kSecTUni, kSecTEdges, kSecTWdata,              // general type tables (type_kind == pattern tables)
    kSecPUni, kSecPBi, kSecPTri, kSecPDeep, kSecPXrows, kSecPTrow, kSecPCpid, kSecPXcid,   // packed tables: contiguous, addressed from kSecPUni
    kSecTypeTable, kSecCtype, kSecCinfo, kSecCid,
    kSecTagTokTab, kSecTagModels, kSecTagMfilt, kSecTagNgrams, kSecTagNrec, kSecTagSyms, kSecTagSlots, kSecTagWeights, kSecTagSlotStr, kSecTagStrOff, kSecTagStrBytes,
    kSectionCount
};
struct TableGeom {
    uint32_t present, short_bits, edge_bits, stride_dw, uni_dw, uni_n, ext_slot, has_long;
    int32_t window, lo[3], len[3];
};
constexpr char kCompiledMagic[16] = "VaporettoHIP-C\x01";   // 15 chars + NUL
constexpr uint32_t kCompiledVersion = 12;                    // bump whenever layout.h or a kernel's reading of it changes
struct PredictorMeta {                                      // plain data: written and read as is (little-endian hosts)
    char magic[16];
    uint32_t version, meta_bytes;
    uint64_t arena_bytes, checksum;                         // checksum: of the arena's bytes (0 in a description: vpt_predictor_describe)
    uint64_t meta_checksum;                                 // of this block with both checksum fields zero
    uint64_t sec_off[kSectionCount], sec_bytes[kSectionCount];
    int32_t bias, pad, type_kind, type_window, chunks;
    uint32_t predict_tags, has_tags, n_tags, tok_bits, max_tag_suffix, tag_use_char, tag_use_type, n_tag_models, n_tag_strings, max_tag_scores;
    TableGeom geom[2];                                      // chars, types
    uint32_t pk_present, pk_n_uni, pk_n_tri, pk_bi_shift, pk_wl, pk_trow_mode, pk_trow_levels, pk_xcid_bits;
    vpt_model_info info;
};

uint64_t arena_checksum(const unsigned char* p, size_t n) {   // n is a multiple of 256
    const uint64_t* w = reinterpret_cast<const uint64_t*>(p);
    uint64_t a = 0x9E3779B97F4A7C15ull, b = 0xC2B2AE3D27D4EB4Full, c = 0x165667B19E3779F9ull, d = 0x27D4EB2F165667C5ull;
    for (size_t i = 0; i + 4 <= n / 8; i += 4) {   // four independent lanes: memory-bound, not multiply-bound
        a = (a ^ w[i]) * 0x100000001B3ull; b = (b ^ w[i + 1]) * 0x100000001B3ull;
        c = (c ^ w[i + 2]) * 0x100000001B3ull; d = (d ^ w[i + 3]) * 0x100000001B3ull;
    }
    return a ^ (b << 1 | b >> 63) ^ (c << 2 | c >> 62) ^ (d << 3 | d >> 61) ^ uint64_t(n);
}

// Field by field: hashing the struct's raw bytes would take in its padding (vpt_model_info has 4 bytes of it in front of
// device_table_bytes and 4 at its tail), which only a memcpy of the whole block preserves.  The two checksum fields are left out.
struct MetaHash {
    uint64_t h = 0xCBF29CE484222325ull;
    template <typename T>
    void add(const T& v) {
        static_assert(std::is_arithmetic<T>::value, "scalars only: no padding inside");
        const unsigned char* p = reinterpret_cast<const unsigned char*>(&v);
        for (size_t i = 0; i < sizeof(T); ++i) h = (h ^ p[i]) * 0x100000001B3ull;
    }
    template <typename T, size_t N>
    void add(const T (&a)[N]) { for (const T& v : a) add(v); }
};
uint64_t meta_checksum_of(const PredictorMeta& m) {
    MetaHash f;
    f.add(m.magic); f.add(m.version); f.add(m.meta_bytes); f.add(m.arena_bytes);
    f.add(m.sec_off); f.add(m.sec_bytes);
    f.add(m.bias); f.add(m.pad); f.add(m.type_kind); f.add(m.type_window); f.add(m.chunks);
    f.add(m.predict_tags); f.add(m.has_tags); f.add(m.n_tags); f.add(m.tok_bits); f.add(m.max_tag_suffix); f.add(m.tag_use_char); f.add(m.tag_use_type);
    f.add(m.n_tag_models); f.add(m.n_tag_strings); f.add(m.max_tag_scores);
    for (const TableGeom& g : m.geom) {
        f.add(g.present); f.add(g.short_bits); f.add(g.edge_bits); f.add(g.stride_dw); f.add(g.uni_dw); f.add(g.uni_n); f.add(g.ext_slot); f.add(g.has_long);
        f.add(g.window); f.add(g.lo); f.add(g.len);
    }
    f.add(m.pk_present); f.add(m.pk_n_uni); f.add(m.pk_n_tri); f.add(m.pk_bi_shift); f.add(m.pk_wl); f.add(m.pk_trow_mode); f.add(m.pk_trow_levels); f.add(m.pk_xcid_bits);
    const vpt_model_info& i = m.info;
    f.add(i.n_char_ngrams); f.add(i.n_type_ngrams); f.add(i.n_dict_words); f.add(i.n_tag_models); f.add(i.bias); f.add(i.char_window); f.add(i.type_window);
    f.add(i.max_pattern_chars); f.add(i.n_short_entries); f.add(i.n_long_nodes); f.add(i.type_kind); f.add(i.device_table_bytes); f.add(i.hot_table_bytes);
    f.add(i.packed); f.add(i.n_displaced); f.add(i.type_rows); f.add(i.n_overflow_children); f.add(i.predict_tags);
    return f.h ^ 0x5A17EDull;
}
// (a field added to PredictorMeta / TableGeom / vpt_model_info must be added above: these sizes are the reminder)
static_assert(sizeof(TableGeom) == 60 && sizeof(vpt_model_info) == 88, "meta_checksum_of lists every field");

// What the kernels assume of the scalars and section sizes of a compiled predictor that did not come from compile_model in
// this process (vpt_predictor_load, vpt_predictor_adopt_device): a flipped bit in the description must not become an
// out-of-bounds device read or a silently different score.  The arena's CONTENT is covered by its own checksum.
const char* validate_meta(const PredictorMeta& m) {
    auto sz = [&](int sec) { return m.sec_bytes[sec]; };
    if (m.pad < 1 || m.pad > vpt::kMaxWindow || m.chunks < 1 || m.chunks > 16) return "pad / chunks";
    if (m.type_kind < 0 || m.type_kind > 2 || m.type_window < 0 || m.type_window > vpt::kMaxWindow) return "type scorer";
    if (sz(kSecCtype) != 65536 || sz(kSecCinfo) != 2ull * 65536 * 4) return "char class tables";
    if (sz(kSecCid) != (m.pk_present ? 2ull * 65536 * 4 : 0ull)) return "char id table";
    if (m.type_kind == vpt::kTypeWindowTable && (m.type_window > 3 || sz(kSecTypeTable) < (4ull << (6 * m.type_window)))) return "type window table";
    for (int t = 0; t < 2; ++t) {
        const TableGeom& g = m.geom[t];
        if (!g.present) continue;
        const int first = t == 0 ? kSecCShort : kSecTShort;
        if (g.short_bits < 2 || g.short_bits > 30 || g.edge_bits < 3 || g.edge_bits > 30) return "table geometry (bits)";
        if (g.stride_dw < 4 || g.stride_dw > 64 || g.stride_dw % 4 || g.uni_dw < 4 || g.uni_dw > 64 || g.uni_dw % 4) return "table geometry (strides)";
        if (g.window < 0 || g.window > vpt::kMaxWindow || g.ext_slot + 2 > g.stride_dw) return "table geometry (window)";
        if (g.uni_n > (t == 0 ? vpt::kUniDirectChars : vpt::kUniDirectTypes)) return "table geometry (direct rows)";
        for (int i = 0; i < 3; ++i)
            if (g.len[i] < 0 || g.len[i] > 2 * vpt::kMaxWindow + 2 || g.lo[i] > 0 || g.lo[i] < -vpt::kMaxWindow - 1 || uint32_t(g.len[i]) + 2 > g.stride_dw || uint32_t(g.len[i]) > g.uni_dw + (i ? 64u : 0u)) return "table geometry (rows)";
        if (sz(first) < (4ull << g.short_bits) * g.stride_dw || sz(first + 1) < 4ull * g.uni_n * g.uni_dw || sz(first + 2) < (16ull << g.edge_bits)) return "table geometry (sections)";
    }
    if (m.type_kind == vpt::kTypePatternTable && !m.geom[1].present) return "type pattern tables";
    if (m.pk_present) {
        if (m.pk_wl < 3 || m.pk_wl > uint32_t(vpt::kMaxWindow)) return "packed tables (row window)";
        const int wl = int(m.pk_wl);
        if (m.pk_n_uni < 2 || sz(kSecPUni) < 4ull * vpt::pk_uni_dw(wl) * m.pk_n_uni || sz(kSecPCpid) < 4ull * m.pk_n_uni || sz(kSecPTri) < 4ull * vpt::pk_tri_dw(wl) * m.pk_n_tri) return "packed tables";
        if (m.pk_bi_shift > 16 || sz(kSecPBi) < 4ull * vpt::pk_bi_dw(wl) || sz(kSecPDeep) < 64) return "packed tables (bigram level)";
        if (m.pk_trow_mode > vpt::kTypeRowsGlobal) return "type rows";
        if (m.pk_trow_mode == vpt::kTypeRowsLds && (m.pk_trow_levels != 3 || sz(kSecPTrow) < 4ull * vpt::pk_trow_dw(wl) * vpt::kTypeRowCount)) return "type rows";
        if (m.pk_trow_mode == vpt::kTypeRowsGlobal && (m.pk_trow_levels < 3 || m.pk_trow_levels > uint32_t(vpt::kMaxTypeRowLevels) ||
                                                      sz(kSecPTrow) < 4ull * vpt::pk_trow_global_dw(wl) * vpt::type_row_count(int(m.pk_trow_levels)))) return "type rows";
        if (m.pk_xcid_bits > 20 || sz(kSecPXcid) != (m.pk_xcid_bits ? 8ull + (8ull << m.pk_xcid_bits) : 0ull)) return "packed tables (chars outside the BMP)";
        if (m.sec_off[kSecPXcid] + sz(kSecPXcid) - m.sec_off[kSecPUni] >= (1ull << 32)) return "packed tables (32-bit offsets)";
    }
    if (m.has_tags) {
        if (m.n_tags == 0 || m.n_tags > 4096 || m.tok_bits < 2 || m.tok_bits > 30 || m.max_tag_scores > vpt::kTagMaxZ) return "tag tables";
        if (sz(kSecTagTokTab) < (16ull << m.tok_bits) || sz(kSecTagModels) < 48ull * m.n_tag_models || sz(kSecTagMfilt) < 4ull * vpt::kTagFiltStride * m.n_tag_models) return "tag tables (sections)";
        if (sz(kSecTagStrOff) < 4ull * (uint64_t(m.n_tag_strings) + 1)) return "tag strings";
    } else if (m.predict_tags > 1) return "predict_tags";
    return nullptr;
}

TableGeom geom_of(const vpt::HostPatternTable& h) {
    TableGeom g{};
    g.present = h.present ? 1u : 0u;
    if (!h.present) return g;
    g.short_bits = h.short_bits; g.edge_bits = h.edge_bits; g.stride_dw = h.stride_dw; g.uni_dw = h.uni_dw; g.uni_n = h.uni_n;
    g.ext_slot = h.ext_slot; g.has_long = h.has_long ? 1u : 0u; g.window = h.window;
    for (int i = 0; i < 3; ++i) { g.lo[i] = h.lo[i]; g.len[i] = h.len[i]; }
    return g;
}

void fill_info(const vpt::CompiledModel& c, vpt_model_info* info) {
    std::memset(info, 0, sizeof(*info));
    info->n_char_ngrams = c.n_char_ngrams; info->n_type_ngrams = c.n_type_ngrams;
    info->n_dict_words = c.n_dict_words; info->n_tag_models = c.n_tag_models;
    info->bias = c.bias;
    info->char_window = c.chars.present ? uint32_t(c.char_window) : 0;
    info->type_window = uint32_t(c.type_window);
    info->max_pattern_chars = c.chars.max_pattern;
    info->n_short_entries = c.chars.n_short; info->n_long_nodes = c.chars.n_long_nodes;
    info->type_kind = uint32_t(c.type_kind);
    info->packed = c.packed.present ? 1u : 0u;
    info->n_displaced = c.packed.present ? 0u : c.chars.n_displaced_short;
    info->type_rows = c.packed.present ? c.packed.trow_mode : 0u;
    info->n_overflow_children = 0u;
    // the specialised kernel reads only the packed tables; the general ones stay resident for oversized sentences
    info->device_table_bytes = (c.chars.present ? c.chars.bytes() : 0) + (c.types.present ? c.types.bytes() : 0) +
                               4ull * c.type_table.size() + (c.packed.present ? c.packed.bytes() : 0);
    info->hot_table_bytes = c.packed.present ? c.packed.bytes() + (c.packed.trow_mode == vpt::kTypeRowsNone ? 4ull * c.type_table.size() : 0) : info->device_table_bytes;
}

}  // namespace

struct vpt_batch {
    const vpt_predictor* pred = nullptr;
    int device = 0;
    BatchKnobs knobs;                  // read once, when the workspace was made
    // per-call device tables
    uint32_t* d_tile_first = nullptr; size_t tile_cap = 0;
    vpt::TileDesc* d_tiles = nullptr; size_t tiles_cap = 0;          // the specialised kernel's tiles
    uint32_t* d_cut_local = nullptr; size_t cut_local_cap = 0;       // ... and, for cut tiles, the lead-byte index of the text
    uint64_t* d_cut_super = nullptr; size_t cut_super_cap = 0;
    uint32_t* d_slow_list = nullptr;
    uint32_t* d_ctrl = nullptr;        // [0] status bits, [1] slow tile count
    uint64_t* d_prof = nullptr;        // 8 per-phase cycle counters + 8 node-read counters (only with VPT_PROFILE_PHASES set)
    unsigned char* d_scratch = nullptr; size_t scratch_bytes = 0;
    uint32_t* d_cps = nullptr; size_t cps_cap = 0;   // decoded scalar values for vpt_fill_tags_batch_device
    // the batch whose chars d_cps holds because the scoring kernel of a predict call on this workspace wrote them (all 0: none)
    const void* cps_text = nullptr; const void* cps_ooff = nullptr; size_t cps_sentences = 0; uint64_t cps_boundaries = 0; unsigned cps_flags = 0;
    uint64_t max_chars = 0;            // caller's bound on chars per sentence (0 = unknown)
    unsigned flags = 0;                // VPT_FLAG_*
    // (vpt_tokenize_batch) the scoring launch of the next predict call goes on split_stream, behind split_event recorded after the tile search:
    // the caller's next chunk can then have its chars counted and its tiles found (other workspace, the call's stream) while this one is scored
    hipStream_t split_stream = nullptr; hipEvent_t split_event = nullptr;
    // timing
    bool timing = false;
    std::vector<hipEvent_t> ev;        // ring of (start, stop) pairs around the scoring kernel
    size_t ev_calls = 0;               // timed calls since the last vpt_batch_kernel_ms
    uint32_t last_tiles = 0, last_tile_flat = 0, last_plan = 0;   // last_plan: 0 general kernels, 1 whole-sentence tiles, 2 cut tiles
    hipStream_t last_stream = nullptr; bool pending = false;
    // staging for the host-buffer entry points
    hipStream_t own_stream = nullptr;
    uint8_t* d_text = nullptr; size_t text_cap = 0;
    uint64_t *d_boff = nullptr, *d_ooff = nullptr; size_t off_cap = 0;
    int32_t* d_scores = nullptr; uint8_t* d_labels = nullptr; size_t out_cap = 0;
    int32_t* d_tags = nullptr; size_t tags_cap = 0;                 // vpt_fill_tags_batch
    int32_t* d_tag_scores = nullptr; size_t tag_scores_cap = 0;     // vpt_fill_tags_scores_batch
    int32_t* d_tag_models = nullptr; size_t tag_models_cap = 0;
    uint8_t* d_tok = nullptr; size_t tok_cap = 0;                   // vpt_write_tokenized_batch
    uint8_t* d_tlab = nullptr; size_t tlab_cap = 0;                 // vpt_tokenize_batch: the labels of the whole batch (no scores are kept)
    uint64_t* d_toff = nullptr; size_t toff_cap = 0;
    uint64_t* d_chain = nullptr; size_t chain_cap = 0;             // vpt_tokenize_batch: where a chunk's tokenized text starts (EmitOut::chain_in / chain_out)
    // what the last fill_tags on this workspace left (TagParams, kernels.hpp): a record per token that has a tag model, sorted by position
    uint4* d_tag_records = nullptr; size_t tag_records_cap = 0;
    int32_t* d_rec_tags = nullptr; size_t rec_tags_cap = 0;
    uint64_t* d_tag_ctl = nullptr; size_t tag_ctl_cap = 0;          // the scan's state, run_pref [n_runs + 1]: zeroed as one range per call
    uint64_t* d_run_pref = nullptr;                                 // (inside d_tag_ctl)
    uint2* d_rec_str = nullptr; size_t rec_str_cap = 0;
    uint4* d_tag_cands = nullptr; size_t tag_cands_cap = 0;
    uint32_t* d_tag_summary = nullptr;
    uint64_t tag_chars = 0, tag_sentences = 0, tag_runs = 0;        // the batch those records belong to (0 chars: none)
    uint32_t tag_run_sent = 0;
    std::vector<uint64_t> h_boff, h_ooff;                           // rebased offsets of the call in flight (copied asynchronously)
    uint8_t* d_types = nullptr; size_t types_cap = 0;               // vpt_char_types_batch
    uint64_t* d_scan_part = nullptr; size_t scan_part_cap = 0;      // per-workgroup partials of the prefix sums (kernels_emit.hip)
    // the writer's state words (EmitFuse): two arrays of emit_state_cap words, used in turn; a call zeroes what the call before it
    // left in the other one (emit_dirty = how many words that is)
    uint64_t* d_emit_state = nullptr; size_t emit_state_cap = 0; size_t emit_dirty[2] = {0, 0}; int emit_flip = 0;
    // the pipelined host-buffer path (predict_pipelined): two sets of device buffers, copy streams, pinned offset staging
    struct PipeSet {
        uint8_t* text = nullptr; size_t text_cap = 0;
        uint64_t* off = nullptr; size_t off_cap = 0;      // byte offsets, then boundary offsets: one copy
        int32_t* scores = nullptr; uint8_t* labels = nullptr; size_t scores_cap = 0, labels_cap = 0;
        hipEvent_t ev_in = nullptr, ev_k = nullptr, ev_out = nullptr;   // chunk copied in / scored / copied out
    } pipe[2];   // (four sets, i.e. the host running further ahead, measured no faster at 2 M-char chunks and slower at 1 M: profiles/r02_c7_e2e.txt)
    // ONE copy stream per direction: a single hipMemcpyAsync stream moves 56 GB/s each way and 84 GB/s both ways at once on
    // this link; two streams per direction were slower (profiles/r02_c6_pcie_microbench.txt)
    hipStream_t s_in = nullptr, s_out = nullptr;
    hipStream_t s_tok_in = nullptr, s_tok_out = nullptr;            // vpt_tokenize_batch's copy streams (the fused path)
    uint64_t* h_off = nullptr; size_t h_off_cap = 0;                // pinned: the rebased offsets of every chunk of the call in flight
    std::vector<hipEvent_t> chunk_ev;                               // vpt_tokenize_batch: one per chunk in flight
};

struct DeviceTags {   // views into the arena
    const uint32_t *tok_tab = nullptr, *models = nullptr, *mfilt = nullptr, *ngrams = nullptr, *nrec = nullptr, *syms = nullptr, *slots = nullptr, *slot_str = nullptr, *str_off = nullptr;
    const uint8_t* str_bytes = nullptr;
    const int32_t* weights = nullptr;
    uint32_t n_models = 0, n_strings = 0;
};

struct vpt_predictor {
    int device = 0;
    PredictorKnobs knobs;              // read once, when the predictor was made (bind_predictor)
    BatchKnobs pool_knobs;             // ... and what the workspaces of its pool are made with
    unsigned char* arena = nullptr;    // the one device allocation that holds every table
    PredictorMeta meta{};
    // what the launches use, bound from meta + arena (bind_predictor)
    bool predict_tags = false;
    bool has_tags = false;
    uint32_t n_tags = 0, tok_bits = 0, max_tag_suffix = 0, max_tag_scores = 0;
    bool tag_use_char = false, tag_use_type = false;
    DeviceTags dtag;
    vpt_model_info info{};
    int32_t bias = 0; int pad = 1; int type_kind = 0; int type_window = 0; int chunks = 2;
    uint32_t tile_slots = 0;           // workgroups of the scoring kernel the device runs at a time (0 = unknown)
    uint32_t n_cus = 0;                // compute units of the device (0 = unknown)
    vpt::PackedView pk{};
    const int32_t* d_type_table = nullptr;
    const uint8_t* d_ctype = nullptr;
    const uint32_t* d_cinfo = nullptr; // [0, 65536): plain; [65536, 131072): through KyteaFullwidthFilter
    const uint32_t* d_cid = nullptr;   // the same two tables for the specialised kernel: id | type << 16 | linebreak << 19
    vpt::PatternTableView ct{}, tt{};
    mutable std::mutex pool_mu;
    mutable std::vector<vpt_batch*> pool;  // idle workspaces for the host-buffer entry points
};

namespace {

vpt_status compile(const uint8_t* bytes, size_t len, int predict_tags, vpt::CompiledModel* out) {
    if (!bytes) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: model_bytes: must not be NULL");
    try {
        vpt::ModelData m = vpt::parse_model(bytes, len, nullptr);
        *out = vpt::compile_model(m, predict_tags != 0);
    } catch (const vpt::ModelError& e) {
        return fail(VPT_INVALID_MODEL, e.what());
    } catch (const std::bad_alloc&) {
        return fail(VPT_RUNTIME_ERROR, "out of host memory while compiling the model");
    }
    return VPT_OK;
}

void batch_release(vpt_batch* b) {
    if (!b) return;
    (void)hipSetDevice(b->device);
    (void)hipFree(b->d_scan_part);
    (void)hipFree(b->d_emit_state);
    (void)hipFree(b->d_chain);
    (void)hipFree(b->d_tiles); (void)hipFree(b->d_cut_local); (void)hipFree(b->d_cut_super);
    (void)hipFree(b->d_tile_first); (void)hipFree(b->d_slow_list); (void)hipFree(b->d_ctrl); (void)hipFree(b->d_scratch);
    (void)hipFree(b->d_prof); (void)hipFree(b->d_cps);
    (void)hipFree(b->d_text); (void)hipFree(b->d_boff); (void)hipFree(b->d_ooff); (void)hipFree(b->d_scores); (void)hipFree(b->d_labels);
    (void)hipFree(b->d_tags); (void)hipFree(b->d_tag_scores); (void)hipFree(b->d_tag_models); (void)hipFree(b->d_tok); (void)hipFree(b->d_tlab); (void)hipFree(b->d_toff); (void)hipFree(b->d_tag_records); (void)hipFree(b->d_rec_tags); (void)hipFree(b->d_tag_ctl); (void)hipFree(b->d_rec_str); (void)hipFree(b->d_tag_cands); (void)hipFree(b->d_tag_summary);
    (void)hipFree(b->d_types);
    for (auto& ps : b->pipe) {
        (void)hipFree(ps.text); (void)hipFree(ps.off); (void)hipFree(ps.scores); (void)hipFree(ps.labels);
        if (ps.ev_in) (void)hipEventDestroy(ps.ev_in);
        if (ps.ev_k) (void)hipEventDestroy(ps.ev_k);
        if (ps.ev_out) (void)hipEventDestroy(ps.ev_out);
    }
    if (b->h_off) (void)hipHostFree(b->h_off);
    for (hipEvent_t e : b->chunk_ev) (void)hipEventDestroy(e);
    if (b->s_in) (void)hipStreamDestroy(b->s_in);
    if (b->s_tok_in) (void)hipStreamDestroy(b->s_tok_in);
    if (b->s_tok_out) (void)hipStreamDestroy(b->s_tok_out);
    if (b->s_out) (void)hipStreamDestroy(b->s_out);
    for (hipEvent_t e : b->ev) (void)hipEventDestroy(e);
    if (b->own_stream) (void)hipStreamDestroy(b->own_stream);
    delete b;
}

template <typename T>
vpt_status grow(T** ptr, size_t* cap, size_t need) {
    if (need <= *cap && *ptr) return VPT_OK;
    size_t ncap = std::max<size_t>(need, *cap + *cap / 2);
    ncap = std::max<size_t>(ncap, 64);
    (void)hipFree(*ptr);
    *ptr = nullptr; *cap = 0;
    VPT_HIP(hipMalloc(reinterpret_cast<void**>(ptr), ncap * sizeof(T) + 64));
    *cap = ncap;
    return VPT_OK;
}

vpt_status status_from_bits(uint32_t bits) {
    if (bits == 0) return VPT_OK;
    if (bits & vpt::kErrEmptySentence)
        return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: text: must contain at least one character");
    if (bits & vpt::kErrNulChar) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: text: must not contain NULL");
    if (bits & vpt::kErrBadOffsets)
        return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: out_offsets: do not match the text (or the text is not valid UTF-8)");
    if (bits & vpt::kErrUnknownLabel)
        return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: labels: only NotWordBoundary (0) and WordBoundary (1) can be written as tokenized text");
    if (bits & vpt::kErrOutputTooSmall)
        return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: text_capacity: smaller than the tokenized text");
    return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: max_sentence_bytes / max_sentence_chars: smaller than the longest sentence");
}

// An idle workspace of the predictor's pool for one host-buffer call (created, with a stream of its own, when the
// pool is empty); goes back to the pool when the guard dies.
struct Workspace {
    const vpt_predictor* p = nullptr;
    vpt_batch* b = nullptr;
    ~Workspace() {
        if (!b) return;
        (void)hipStreamSynchronize(b->own_stream);   // an error return may leave copies from the caller's buffers in flight
        if (b->s_in) (void)hipStreamSynchronize(b->s_in);
        if (b->s_tok_in) (void)hipStreamSynchronize(b->s_tok_in);
        if (b->s_tok_out) (void)hipStreamSynchronize(b->s_tok_out);
        if (b->s_out) (void)hipStreamSynchronize(b->s_out);
        std::lock_guard<std::mutex> g(p->pool_mu);
        p->pool.push_back(b);
    }
};
vpt_status batch_create_with(const vpt_predictor* p, const BatchKnobs& knobs, vpt_batch** out) {
    *out = nullptr;
    VPT_HIP(hipSetDevice(p->device));
    vpt_batch* b = new (std::nothrow) vpt_batch();
    if (!b) return fail(VPT_RUNTIME_ERROR, "out of host memory");
    b->pred = p; b->device = p->device;
    b->knobs = knobs;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&b->d_ctrl), 64);
    if (e == hipSuccess) e = hipMemset(b->d_ctrl, 0, 64);
    if (e == hipSuccess && b->knobs.profile_phases) {
        e = hipMalloc(reinterpret_cast<void**>(&b->d_prof), 128);
        if (e == hipSuccess) e = hipMemset(b->d_prof, 0, 128);
    }
    if (e != hipSuccess) { batch_release(b); return fail(VPT_RUNTIME_ERROR, std::string("HIP error: ") + hipGetErrorString(e)); }
    *out = b;
    return VPT_OK;
}

vpt_status acquire(const vpt_predictor* p, Workspace* w) {
    w->p = p;
    {
        std::lock_guard<std::mutex> g(p->pool_mu);
        if (!p->pool.empty()) { w->b = p->pool.back(); p->pool.pop_back(); }
    }
    if (w->b) { w->b->cps_text = nullptr; return VPT_OK; }   // a pooled workspace remembers nothing of the call before
    vpt_batch* b = nullptr;
    vpt_status st = batch_create_with(p, p->pool_knobs, &b);   // (the knobs the predictor was made under: no getenv from a host thread's first call)
    if (st != VPT_OK) return st;
    if (hipStreamCreateWithFlags(&b->own_stream, hipStreamNonBlocking) != hipSuccess) {
        batch_release(b);
        return fail(VPT_RUNTIME_ERROR, "HIP error: cannot create a stream");
    }
    w->b = b;
    return VPT_OK;
}

// The caller's batch -> the workspace's staging buffers, on its stream: text, offsets rebased so that the device sees
// text and outputs starting at 0, and (when given) the labels.  `max_bytes` / `max_chars`: the longest sentence.
vpt_status stage(vpt_batch* b, const uint8_t* utf8, const uint64_t* byte_offsets, const uint64_t* out_offsets, size_t n_sentences,
                 const uint8_t* labels, uint64_t* total_b_out, uint64_t* max_bytes_out, uint64_t* max_chars_out) {
    b->cps_text = nullptr;   // d_text is about to be rewritten: whatever chars a predict call left decoded are another batch's
    const uint64_t t0 = byte_offsets[0], t1 = byte_offsets[n_sentences];
    if (t1 < t0) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: byte_offsets: must be non-decreasing");
    const size_t nbytes = size_t(t1 - t0);
    const uint64_t total_b = out_offsets[n_sentences] - out_offsets[0];
    uint64_t max_bytes = 0, max_chars = 0;
    for (size_t i = 0; i < n_sentences; ++i) {
        if (byte_offsets[i + 1] <= byte_offsets[i])
            return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: text: must contain at least one character");
        const uint64_t nb = byte_offsets[i + 1] - byte_offsets[i];
        // n chars take between n and 4n bytes: anything else cannot have come from vpt_count_boundaries (checked
        // before any buffer is sized from these numbers)
        if (out_offsets[i + 1] < out_offsets[i] || out_offsets[i + 1] - out_offsets[i] + 1 > nb)
            return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: out_offsets: do not match the text (or the text is not valid UTF-8)");
        max_bytes = std::max<uint64_t>(max_bytes, nb);
        max_chars = std::max<uint64_t>(max_chars, out_offsets[i + 1] - out_offsets[i] + 1);
    }
    vpt_status st;
    if ((st = grow(&b->d_text, &b->text_cap, nbytes + 32)) != VPT_OK) return st;
    {
        size_t cap = b->off_cap;
        if ((st = grow(&b->d_boff, &cap, n_sentences + 1)) != VPT_OK) return st;
        size_t cap2 = b->off_cap;
        if ((st = grow(&b->d_ooff, &cap2, n_sentences + 1)) != VPT_OK) return st;
        b->off_cap = std::min(cap, cap2);
    }
    {
        size_t cap = b->out_cap;
        if ((st = grow(&b->d_scores, &cap, size_t(total_b) + 1)) != VPT_OK) return st;
        size_t cap2 = b->out_cap;
        if ((st = grow(&b->d_labels, &cap2, size_t(total_b) + 1)) != VPT_OK) return st;
        b->out_cap = std::min(cap, cap2);
    }
    std::vector<uint64_t>&boff = b->h_boff, &ooff = b->h_ooff;   // they outlive the asynchronous copies: the call ends with a sync
    boff.resize(n_sentences + 1); ooff.resize(n_sentences + 1);
    for (size_t i = 0; i <= n_sentences; ++i) { boff[i] = byte_offsets[i] - t0; ooff[i] = out_offsets[i] - out_offsets[0]; }
    hipStream_t s = b->own_stream;
    VPT_HIP(hipMemcpyAsync(b->d_text, utf8 + t0, nbytes, hipMemcpyHostToDevice, s));
    VPT_HIP(hipMemcpyAsync(b->d_boff, boff.data(), 8 * (n_sentences + 1), hipMemcpyHostToDevice, s));
    VPT_HIP(hipMemcpyAsync(b->d_ooff, ooff.data(), 8 * (n_sentences + 1), hipMemcpyHostToDevice, s));
    if (labels && total_b) VPT_HIP(hipMemcpyAsync(b->d_labels, labels + out_offsets[0], size_t(total_b), hipMemcpyHostToDevice, s));
    *total_b_out = total_b;
    if (max_bytes_out) *max_bytes_out = max_bytes;
    if (max_chars_out) *max_chars_out = max_chars;
    return VPT_OK;
}

}  // namespace

// the pieces of the device-resident path that the host pipelines drive as well (capi_device.cpp)
namespace vptc {
vpt_status emit_device(const vpt_predictor* p, vpt_batch* b, const uint8_t* d_utf8, const uint64_t* d_byte_offsets,
                       const uint64_t* d_out_offsets, size_t n_sentences, uint64_t total_boundaries, const uint8_t* d_labels,
                       bool tagged, uint8_t* d_text_out, uint64_t text_capacity, uint64_t* d_text_offsets_out,
                       hipStream_t stream, uint64_t* total_out = nullptr, const uint64_t* chain_in = nullptr, uint64_t* chain_out = nullptr);
vpt_status predict_device_impl(const vpt_predictor* p, vpt_batch* b, const uint8_t* d_utf8, const uint64_t* d_byte_offsets,
                               const uint64_t* d_out_offsets, size_t n_sentences, uint64_t total_boundaries,
                               uint64_t max_sentence_bytes, int32_t* d_scores, uint8_t* d_labels, void* hip_stream);
vpt_status count_boundaries_impl(const vpt_predictor* p, vpt_batch* b, const uint8_t* d_utf8, const uint64_t* d_byte_offsets,
                                 size_t n_sentences, uint64_t* d_out_offsets, void* hip_stream, uint64_t text_bytes_hint);
}  // namespace vptc
using namespace vptc;
