// C ABI of libvaporetto_hip.so (include/vaporetto_hip.h).  Host side of the drop-in boundary:
// model loading + table upload (Predictor::new), batch staging and kernel launches (Predictor::predict).
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

namespace {

thread_local std::string g_last_error;

vpt_status fail(vpt_status st, const std::string& msg) {
    g_last_error = msg;
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
    uint32_t emit_per_block = 0;        // VPT_EMIT_PER_BLOCK: sentences a workgroup of the writer takes (1..256; 0: from the mean sentence length) -- tests: runs of any size
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
    if (const char* v = std::getenv("VPT_EMIT_PER_BLOCK")) k.emit_per_block = uint32_t(std::min(256, std::max(0, std::atoi(v))));
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

extern "C" {

const char* vpt_last_error(void) { return g_last_error.c_str(); }
const char* vpt_version(void) { return "vaporetto_hip 0.1.0 (gfx950)"; }

vpt_status vpt_model_read_len(const uint8_t* model_bytes, size_t len, size_t* consumed) {
    if (!model_bytes || !consumed) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: NULL argument");
    try {
        (void)vpt::parse_model(model_bytes, len, consumed);
    } catch (const vpt::ModelError& e) {
        return fail(VPT_INVALID_MODEL, e.what());
    } catch (const std::bad_alloc&) {
        return fail(VPT_RUNTIME_ERROR, "out of host memory while decoding the model");
    }
    return VPT_OK;
}

vpt_status vpt_model_inspect(const uint8_t* model_bytes, size_t len, int predict_tags, vpt_model_info* info) {
    if (!info) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: info: must not be NULL");
    vpt::CompiledModel c;
    vpt_status st = compile(model_bytes, len, predict_tags, &c);
    if (st != VPT_OK) return st;
    fill_info(c, info);
    return VPT_OK;
}

namespace {

struct SectionSrc { const void* ptr; size_t bytes; };

// Views and scalars of a predictor from its meta block and arena; the occupancy figure comes from the device.
void bind_predictor(vpt_predictor* p) {
    const PredictorMeta& m = p->meta;
    p->knobs = read_predictor_knobs();
    p->pool_knobs = read_batch_knobs();
    auto at = [&](int sec) { return p->arena + m.sec_off[sec]; };
    auto view = [&](const TableGeom& g, int first) {
        vpt::PatternTableView v{};
        v.present = g.present;
        if (!g.present) return v;
        v.short_tab = reinterpret_cast<const uint32_t*>(at(first)); v.uni = reinterpret_cast<const uint32_t*>(at(first + 1));
        v.edges = reinterpret_cast<const uint32_t*>(at(first + 2)); v.wdata = reinterpret_cast<const int32_t*>(at(first + 3));
        v.short_shift = 32 - (g.short_bits - 1); v.short_mask = (1u << (g.short_bits - 1)) - 1;
        v.edge_shift = 32 - (g.edge_bits - 2); v.edge_mask = (1u << (g.edge_bits - 2)) - 1;
        v.stride_dw = g.stride_dw; v.uni_dw = g.uni_dw; v.uni_n = g.uni_n; v.ext_slot = g.ext_slot;
        v.window = g.window;
        for (int i = 0; i < 3; ++i) { v.lo[i] = g.lo[i]; v.len[i] = g.len[i]; }
        v.has_long = g.has_long;
        return v;
    };
    p->ct = view(m.geom[0], kSecCShort);
    p->tt = view(m.geom[1], kSecTShort);
    p->pk = vpt::PackedView{};
    p->pk.present = m.pk_present;
    if (m.pk_present) {
        p->pk.base = at(kSecPUni);
        auto rel = [&](int sec) { return uint32_t(m.sec_off[sec] - m.sec_off[kSecPUni]); };
        p->pk.off_uni = 0; p->pk.off_bi = rel(kSecPBi); p->pk.off_tri = rel(kSecPTri); p->pk.off_deep = rel(kSecPDeep);
        p->pk.off_xrows = rel(kSecPXrows); p->pk.off_trow = rel(kSecPTrow); p->pk.off_cpid = rel(kSecPCpid);
        p->pk.off_xcid = m.pk_xcid_bits ? rel(kSecPXcid) : 0u;
        p->pk.n_uni = m.pk_n_uni; p->pk.n_tri = m.pk_n_tri; p->pk.bi_shift = m.pk_bi_shift;
        p->pk.wl = m.pk_wl; p->pk.trow_mode = m.pk_trow_mode; p->pk.trow_levels = m.pk_trow_levels;
    }
    p->d_type_table = m.sec_bytes[kSecTypeTable] ? reinterpret_cast<const int32_t*>(at(kSecTypeTable)) : nullptr;
    p->d_ctype = at(kSecCtype);
    p->d_cinfo = reinterpret_cast<const uint32_t*>(at(kSecCinfo));
    p->d_cid = m.sec_bytes[kSecCid] ? reinterpret_cast<const uint32_t*>(at(kSecCid)) : nullptr;
    p->predict_tags = m.predict_tags != 0; p->has_tags = m.has_tags != 0;
    p->n_tags = m.n_tags; p->tok_bits = m.tok_bits; p->max_tag_suffix = m.max_tag_suffix; p->max_tag_scores = m.max_tag_scores;
    p->tag_use_char = m.tag_use_char != 0; p->tag_use_type = m.tag_use_type != 0;
    if (m.has_tags) {
        p->dtag.tok_tab = reinterpret_cast<const uint32_t*>(at(kSecTagTokTab)); p->dtag.models = reinterpret_cast<const uint32_t*>(at(kSecTagModels));
        p->dtag.mfilt = reinterpret_cast<const uint32_t*>(at(kSecTagMfilt));
        p->dtag.ngrams = reinterpret_cast<const uint32_t*>(at(kSecTagNgrams)); p->dtag.nrec = reinterpret_cast<const uint32_t*>(at(kSecTagNrec));
        p->dtag.syms = reinterpret_cast<const uint32_t*>(at(kSecTagSyms));
        p->dtag.slots = reinterpret_cast<const uint32_t*>(at(kSecTagSlots)); p->dtag.weights = reinterpret_cast<const int32_t*>(at(kSecTagWeights));
        p->dtag.slot_str = reinterpret_cast<const uint32_t*>(at(kSecTagSlotStr)); p->dtag.str_off = reinterpret_cast<const uint32_t*>(at(kSecTagStrOff));
        p->dtag.str_bytes = at(kSecTagStrBytes);
        p->dtag.n_models = m.n_tag_models; p->dtag.n_strings = m.n_tag_strings;
    }
    p->info = m.info;
    p->bias = m.bias; p->pad = m.pad; p->type_kind = m.type_kind; p->type_window = m.type_window; p->chunks = m.chunks;
    p->tile_slots = 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, p->device) == hipSuccess && prop.multiProcessorCount > 0) {
        p->n_cus = uint32_t(prop.multiProcessorCount);
        vpt::ScoreParams probe{};
        probe.ct = p->ct; probe.pk = p->pk; probe.pad = p->pad; probe.ctype = p->d_ctype; probe.cid = p->d_cid; probe.type_kind = p->type_kind;
        probe.type_window = p->type_window; probe.force_window_table = p->knobs.force_window_table ? 1u : 0u;
        const bool fast = vpt::fast_path_supported(probe);
        auto slots_for = [&](size_t lds, size_t built_for) {
            const size_t granules = (lds + 1279) / 1280;   // gfx950 hands out its 160 KB of LDS in 1280-byte granules
            return uint32_t(prop.multiProcessorCount) * uint32_t(std::min<size_t>(built_for, std::max<size_t>(1, 128 / std::max<size_t>(granules, 1))));
        };
        if (fast) {
            p->tile_slots = slots_for(vpt::score_tiles_fast_lds_bytes(probe), size_t(vpt::fast_path_wg(probe)));
        } else {
            p->tile_slots = slots_for(vpt::score_tiles_lds_bytes(), 8);   // the general kernel is built for 8 workgroups per CU
        }
    }
}

vpt_status check_device(int device_id) {
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return fail(VPT_RUNTIME_ERROR, "no HIP device available (this library has no CPU fallback)");
    if (device_id < 0 || device_id >= n_dev) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: device_id: no such HIP device");
    return VPT_OK;
}

// the section layout of an arena: 256-byte aligned starts, kTablePadBytes of zeros behind every table
uint64_t layout_sections(const size_t bytes[kSectionCount], uint64_t off[kSectionCount]) {
    uint64_t total = 0;
    for (int i = 0; i < kSectionCount; ++i) { off[i] = total; total += (uint64_t(bytes[i]) + kTablePadBytes + 255) & ~uint64_t(255); }
    return total;
}

}  // namespace

vpt_status vpt_predictor_create(const uint8_t* model_bytes, size_t len, int predict_tags, int device_id, vpt_predictor** out) {
    if (!out) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: out: must not be NULL");
    *out = nullptr;
    vpt::CompiledModel c;
    vpt_status st = compile(model_bytes, len, predict_tags, &c);
    if (st != VPT_OK) return st;
    if ((st = check_device(device_id)) != VPT_OK) return st;
    VPT_HIP(hipSetDevice(device_id));

    // ---- the small tables made here: CharacterType / KyteaFullwidthFilter images of the BMP, ids for the packed tables
    std::vector<uint32_t> cinfo(2 * 65536);   // the char a BMP char is scored as | its CharacterType << 16
    std::vector<uint8_t> ctype(65536);
    for (uint32_t cp = 0; cp < 65536; ++cp) {
        cinfo[cp] = cp | (uint32_t(vpt::char_type_host(cp)) << 16);
        const uint32_t fw = vpt::kytea_fullwidth_host(cp);
        cinfo[65536 + cp] = fw | (uint32_t(vpt::char_type_host(fw)) << 16);
        ctype[cp] = vpt::char_type_host(cp);
    }
    std::vector<uint32_t> cid;
    if (c.packed.present) {
        cid.resize(2 * 65536);
        for (uint32_t cp = 0; cp < 65536; ++cp)
            for (int mode = 0; mode < 2; ++mode) {
                const uint32_t scored = mode ? vpt::kytea_fullwidth_host(cp) : cp;   // 1:1 on the BMP
                cid[size_t(mode) * 65536 + cp] = c.packed.id_for(scored) | (uint32_t(vpt::char_type_host(scored)) << 16) |
                                                 ((scored == 0x0Au || scored == 0x0Du) ? vpt::kCinfoLinebreak : 0u);
            }
    }

    // ---- sections
    SectionSrc src[kSectionCount] = {};
    auto put = [&](int sec, const auto& v) { src[sec] = {v.data(), v.size() * sizeof(v[0])}; };
    if (c.chars.present) { put(kSecCShort, c.chars.short_tab); put(kSecCUni, c.chars.uni); put(kSecCEdges, c.chars.edges); put(kSecCWdata, c.chars.wdata); }
    if (c.types.present) { put(kSecTShort, c.types.short_tab); put(kSecTUni, c.types.uni); put(kSecTEdges, c.types.edges); put(kSecTWdata, c.types.wdata); }
    bool packed_ok = c.packed.present;
    if (packed_ok) {
        put(kSecPUni, c.packed.uni); put(kSecPBi, c.packed.bi); put(kSecPTri, c.packed.tri); put(kSecPDeep, c.packed.deep);
        put(kSecPXrows, c.packed.xrows); put(kSecPTrow, c.packed.trow); put(kSecPCpid, c.packed.cpid); put(kSecPXcid, c.packed.xcid);
        size_t packed_total = 0;
        for (int i = kSecPUni; i <= kSecPXcid; ++i) packed_total += (src[i].bytes + kTablePadBytes + 255) & ~size_t(255);
        if (packed_total >= (size_t(1) << 32)) {   // the specialised kernel addresses them with 32-bit offsets: the general tables serve
            packed_ok = false;
            for (int i = kSecPUni; i <= kSecPXcid; ++i) src[i] = {nullptr, 0};
        }
    }
    if (c.type_kind == vpt::kTypeWindowTable) put(kSecTypeTable, c.type_table);
    put(kSecCtype, ctype); put(kSecCinfo, cinfo);
    if (packed_ok) put(kSecCid, cid);
    if (c.tags.present) {
        put(kSecTagTokTab, c.tags.tok_tab); put(kSecTagModels, c.tags.models); put(kSecTagMfilt, c.tags.mfilt); put(kSecTagNgrams, c.tags.ngrams); put(kSecTagNrec, c.tags.nrec); put(kSecTagSyms, c.tags.syms);
        put(kSecTagSlots, c.tags.slots); put(kSecTagWeights, c.tags.weights); put(kSecTagSlotStr, c.tags.slot_str);
        put(kSecTagStrOff, c.tags.str_off); put(kSecTagStrBytes, c.tags.str_bytes);
    }

    vpt_predictor* p = new (std::nothrow) vpt_predictor();
    if (!p) return fail(VPT_RUNTIME_ERROR, "out of host memory");
    p->device = device_id;
    PredictorMeta& m = p->meta;
    std::memset(&m, 0, sizeof(m));
    std::memcpy(m.magic, kCompiledMagic, sizeof(m.magic));
    m.version = kCompiledVersion; m.meta_bytes = uint32_t(sizeof(PredictorMeta));
    size_t bytes[kSectionCount];
    for (int i = 0; i < kSectionCount; ++i) { bytes[i] = src[i].bytes; m.sec_bytes[i] = src[i].bytes; }
    m.arena_bytes = layout_sections(bytes, m.sec_off);
    m.bias = c.bias; m.pad = c.pad; m.type_kind = c.type_kind; m.type_window = c.type_window;
    {
        uint32_t stride = 4;
        if (c.chars.present) stride = std::max(stride, c.chars.stride_dw);
        if (c.types.present) stride = std::max(stride, c.types.stride_dw);
        m.chunks = int32_t(std::max<uint32_t>(2, stride / 4));
    }
    m.predict_tags = predict_tags != 0;
    if (c.tags.present) {
        m.has_tags = 1; m.n_tags = c.tags.n_tags; m.tok_bits = c.tags.tok_bits;
        m.tag_use_char = c.tags.use_char; m.tag_use_type = c.tags.use_type;
        m.n_tag_models = c.tags.n_models; m.n_tag_strings = uint32_t(c.tags.str_off.size() - 1); m.max_tag_scores = c.tags.max_zlen;
        for (uint32_t mi = 0; mi < c.tags.n_models; ++mi) {   // the longest "/tag/tag.." a token can get
            const uint32_t* mr = &c.tags.models[size_t(mi) * 12];
            uint32_t worst = 0;
            for (uint32_t j = 0; j < mr[9]; ++j) {
                const uint32_t first = c.tags.slot_str[mr[8] + j], cnt = c.tags.slots[size_t(mr[8] + j) * 2];
                uint32_t longest = 0;
                for (uint32_t k = 0; k < cnt; ++k) longest = std::max(longest, c.tags.str_off[first + k + 1] - c.tags.str_off[first + k]);
                worst += 1 + longest;
            }
            m.max_tag_suffix = std::max(m.max_tag_suffix, worst);
        }
    }
    m.geom[0] = geom_of(c.chars); m.geom[1] = geom_of(c.types);
    m.pk_present = packed_ok ? 1u : 0u;
    if (packed_ok) {
        m.pk_wl = uint32_t(c.packed.wl);
        m.pk_n_uni = uint32_t(c.packed.uni.size() / size_t(vpt::pk_uni_dw(c.packed.wl))); m.pk_n_tri = uint32_t(c.packed.tri.size() / size_t(vpt::pk_tri_dw(c.packed.wl)));
        m.pk_bi_shift = c.packed.bi_shift; m.pk_trow_mode = c.packed.trow_mode; m.pk_trow_levels = c.packed.trow_levels;
        m.pk_xcid_bits = c.packed.xcid.empty() ? 0u : c.packed.xcid[0];
    }
    fill_info(c, &m.info);
    m.info.predict_tags = predict_tags != 0;
    if (!packed_ok) { m.info.packed = 0; m.info.type_rows = 0; }

    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p->arena), m.arena_bytes);
    if (e == hipSuccess) e = hipMemset(p->arena, 0, m.arena_bytes);
    for (int i = 0; i < kSectionCount && e == hipSuccess; ++i)
        if (src[i].bytes) e = hipMemcpy(p->arena + m.sec_off[i], src[i].ptr, src[i].bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        std::string msg = std::string("HIP error while uploading the tables: ") + hipGetErrorString(e);
        vpt_predictor_destroy(p);
        return fail(VPT_RUNTIME_ERROR, msg);
    }
    bind_predictor(p);
    *out = p;
    return VPT_OK;
}

void vpt_predictor_destroy(vpt_predictor* p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    for (vpt_batch* b : p->pool) batch_release(b);
    (void)hipFree(p->arena);
    delete p;
}

// ---- the compiled form: PredictorMeta + the arena's bytes
vpt_status vpt_predictor_save(const vpt_predictor* p, uint8_t* out, size_t capacity, size_t* needed) {
    if (!p || !needed) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: NULL argument");
    const size_t total = sizeof(PredictorMeta) + size_t(p->meta.arena_bytes);
    *needed = total;
    if (!out || capacity < total) return out ? fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: capacity: smaller than the compiled predictor") : VPT_OK;
    VPT_HIP(hipSetDevice(p->device));
    VPT_HIP(hipMemcpy(out + sizeof(PredictorMeta), p->arena, size_t(p->meta.arena_bytes), hipMemcpyDeviceToHost));
    PredictorMeta m = p->meta;
    m.checksum = arena_checksum(out + sizeof(PredictorMeta), size_t(m.arena_bytes));
    m.meta_checksum = meta_checksum_of(m);
    std::memcpy(out, &m, sizeof(m));
    return VPT_OK;
}

vpt_status vpt_predictor_load(const uint8_t* blob, size_t len, int device_id, vpt_predictor** out) {
    if (!out) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: out: must not be NULL");
    *out = nullptr;
    if (!blob) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: blob: must not be NULL");
    PredictorMeta m;
    if (len < sizeof(m)) return fail(VPT_INVALID_MODEL, "InvalidModelError: not a compiled predictor (too short)");
    std::memcpy(&m, blob, sizeof(m));
    if (std::memcmp(m.magic, kCompiledMagic, sizeof(m.magic)) != 0) return fail(VPT_INVALID_MODEL, "InvalidModelError: not a compiled predictor");
    if (m.version != kCompiledVersion || m.meta_bytes != sizeof(PredictorMeta))
        return fail(VPT_INVALID_MODEL, "InvalidModelError: compiled predictor version mismatch (compile the model again with this library)");
    if (meta_checksum_of(m) != m.meta_checksum) return fail(VPT_INVALID_MODEL, "InvalidModelError: compiled predictor is corrupt (checksum of the description)");
    if (m.arena_bytes % 256 != 0 || len != sizeof(m) + m.arena_bytes) return fail(VPT_INVALID_MODEL, "InvalidModelError: compiled predictor is truncated");
    {   // the sections must be the layout this library would make of their sizes
        size_t bytes[kSectionCount];
        uint64_t off[kSectionCount];
        for (int i = 0; i < kSectionCount; ++i) bytes[i] = size_t(m.sec_bytes[i]);
        if (layout_sections(bytes, off) != m.arena_bytes || std::memcmp(off, m.sec_off, sizeof(off)) != 0)
            return fail(VPT_INVALID_MODEL, "InvalidModelError: compiled predictor has an inconsistent section table");
    }
    if (arena_checksum(blob + sizeof(m), size_t(m.arena_bytes)) != m.checksum)
        return fail(VPT_INVALID_MODEL, "InvalidModelError: compiled predictor is corrupt (checksum)");
    if (const char* what = validate_meta(m)) return fail(VPT_INVALID_MODEL, std::string("InvalidModelError: compiled predictor has an inconsistent description (") + what + ")");
    vpt_status st = check_device(device_id);
    if (st != VPT_OK) return st;
    VPT_HIP(hipSetDevice(device_id));
    vpt_predictor* p = new (std::nothrow) vpt_predictor();
    if (!p) return fail(VPT_RUNTIME_ERROR, "out of host memory");
    p->device = device_id;
    p->meta = m;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p->arena), m.arena_bytes);
    if (e == hipSuccess) e = hipMemcpy(p->arena, blob + sizeof(m), size_t(m.arena_bytes), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        std::string msg = std::string("HIP error while uploading the tables: ") + hipGetErrorString(e);
        vpt_predictor_destroy(p);
        return fail(VPT_RUNTIME_ERROR, msg);
    }
    bind_predictor(p);
    *out = p;
    return VPT_OK;
}

// The two halves of the compiled form for callers that move the tables themselves (one process per GPU: the arena goes from
// rank 0 to the others with one RCCL broadcast over xGMI, device to device, and every rank adopts the bytes it received).
vpt_status vpt_predictor_describe(const vpt_predictor* p, uint8_t* meta_out, size_t capacity, size_t* meta_bytes, const void** d_arena,
                                  size_t* arena_bytes) {
    if (!p || !meta_bytes) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: NULL argument");
    *meta_bytes = sizeof(PredictorMeta);
    if (d_arena) *d_arena = p->arena;
    if (arena_bytes) *arena_bytes = size_t(p->meta.arena_bytes);
    if (meta_out) {
        if (capacity < sizeof(PredictorMeta)) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: capacity: smaller than the description");
        PredictorMeta m = p->meta;
        m.checksum = 0;   // the device bytes are not read back for it; a transport that can corrupt them has its own checks
        m.meta_checksum = meta_checksum_of(m);
        std::memcpy(meta_out, &m, sizeof(m));
    }
    return VPT_OK;
}

vpt_status vpt_predictor_adopt_device(const uint8_t* meta, size_t meta_len, const void* d_arena, size_t arena_bytes, int device_id,
                                      vpt_predictor** out) {
    if (!out) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: out: must not be NULL");
    *out = nullptr;
    if (!meta || !d_arena) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: NULL argument");
    PredictorMeta m;
    if (meta_len != sizeof(m)) return fail(VPT_INVALID_MODEL, "InvalidModelError: not a compiled predictor description");
    std::memcpy(&m, meta, sizeof(m));
    if (std::memcmp(m.magic, kCompiledMagic, sizeof(m.magic)) != 0 || m.version != kCompiledVersion || m.meta_bytes != sizeof(PredictorMeta))
        return fail(VPT_INVALID_MODEL, "InvalidModelError: compiled predictor version mismatch (compile the model again with this library)");
    if (meta_checksum_of(m) != m.meta_checksum) return fail(VPT_INVALID_MODEL, "InvalidModelError: compiled predictor is corrupt (checksum of the description)");
    if (m.arena_bytes != arena_bytes) return fail(VPT_INVALID_MODEL, "InvalidModelError: compiled predictor is truncated");
    if (const char* what = validate_meta(m)) return fail(VPT_INVALID_MODEL, std::string("InvalidModelError: compiled predictor has an inconsistent description (") + what + ")");
    {
        size_t bytes[kSectionCount];
        uint64_t off[kSectionCount];
        for (int i = 0; i < kSectionCount; ++i) bytes[i] = size_t(m.sec_bytes[i]);
        if (layout_sections(bytes, off) != m.arena_bytes || std::memcmp(off, m.sec_off, sizeof(off)) != 0)
            return fail(VPT_INVALID_MODEL, "InvalidModelError: compiled predictor has an inconsistent section table");
    }
    vpt_status st = check_device(device_id);
    if (st != VPT_OK) return st;
    VPT_HIP(hipSetDevice(device_id));
    vpt_predictor* p = new (std::nothrow) vpt_predictor();
    if (!p) return fail(VPT_RUNTIME_ERROR, "out of host memory");
    p->device = device_id;
    p->meta = m;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p->arena), m.arena_bytes);
    if (e == hipSuccess) e = hipMemcpy(p->arena, d_arena, size_t(m.arena_bytes), hipMemcpyDeviceToDevice);
    if (e != hipSuccess) {
        std::string msg = std::string("HIP error while copying the tables: ") + hipGetErrorString(e);
        vpt_predictor_destroy(p);
        return fail(VPT_RUNTIME_ERROR, msg);
    }
    bind_predictor(p);
    *out = p;
    return VPT_OK;
}

vpt_status vpt_predictor_clone_to_device(const vpt_predictor* src, int device_id, vpt_predictor** out) {
    if (!out) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: out: must not be NULL");
    *out = nullptr;
    if (!src) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: predictor: must not be NULL");
    vpt_status st = check_device(device_id);
    if (st != VPT_OK) return st;
    VPT_HIP(hipSetDevice(device_id));
    vpt_predictor* p = new (std::nothrow) vpt_predictor();
    if (!p) return fail(VPT_RUNTIME_ERROR, "out of host memory");
    p->device = device_id;
    p->meta = src->meta;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p->arena), src->meta.arena_bytes);
    // device to device: over xGMI between two GPUs of a node (hipMemcpyPeer falls back to staging through the host when
    // peer access is not possible), a plain copy on the same device
    if (e == hipSuccess)
        e = device_id == src->device ? hipMemcpy(p->arena, src->arena, size_t(src->meta.arena_bytes), hipMemcpyDeviceToDevice)
                                     : hipMemcpyPeer(p->arena, device_id, src->arena, src->device, size_t(src->meta.arena_bytes));
    if (e != hipSuccess) {
        std::string msg = std::string("HIP error while copying the tables: ") + hipGetErrorString(e);
        vpt_predictor_destroy(p);
        return fail(VPT_RUNTIME_ERROR, msg);
    }
    bind_predictor(p);
    *out = p;
    return VPT_OK;
}

vpt_status vpt_predictor_info(const vpt_predictor* p, vpt_model_info* info) {
    if (!p || !info) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: NULL argument");
    *info = p->info;
    return VPT_OK;
}

vpt_status vpt_count_boundaries(const uint8_t* utf8, const uint64_t* byte_offsets, size_t n_sentences, uint64_t* out_offsets) {
    if ((!utf8 && n_sentences) || !byte_offsets || !out_offsets) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: NULL argument");
    uint64_t acc = 0;
    for (size_t i = 0; i < n_sentences; ++i) {
        out_offsets[i] = acc;
        const uint64_t b0 = byte_offsets[i], b1 = byte_offsets[i + 1];
        if (b1 <= b0) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: text: must contain at least one character");
        const uint8_t* s = utf8 + b0;
        const size_t n = size_t(b1 - b0);
        if (std::memchr(s, 0, n)) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: text: must not contain NULL");
        uint64_t chars = 0;
        for (size_t k = 0; k < n; ++k) chars += (s[k] & 0xC0) != 0x80;
        if (chars == 0) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: text: must contain at least one character");
        acc += chars - 1;
    }
    out_offsets[n_sentences] = acc;
    return VPT_OK;
}

vpt_status vpt_batch_create(const vpt_predictor* p, vpt_batch** out) {
    if (!p || !out) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: NULL argument");
    return batch_create_with(p, read_batch_knobs(), out);
}
void vpt_batch_destroy(vpt_batch* b) { batch_release(b); }

vpt_status vpt_batch_set_timing(vpt_batch* b, int enabled) {
    if (!b) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: NULL argument");
    VPT_HIP(hipSetDevice(b->device));
    if (enabled && b->ev.empty()) {
        for (size_t i = 0; i < 2 * kTimingRing; ++i) {
            hipEvent_t e = nullptr;
            VPT_HIP(hipEventCreate(&e));
            b->ev.push_back(e);
        }
    }
    b->timing = enabled != 0;
    b->ev_calls = 0;
    return VPT_OK;
}

vpt_status vpt_batch_kernel_ms(vpt_batch* b, float* score_kernel_ms, uint32_t* n_tiles) {
    if (!b) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: NULL argument");
    if (n_tiles) *n_tiles = b->last_tiles;
    if (score_kernel_ms) {
        *score_kernel_ms = 0.f;
        if (b->pending) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: batch: call vpt_batch_sync first");
        const size_t n = std::min(b->ev_calls, kTimingRing);
        double sum = 0;
        for (size_t k = 0; k < n; ++k) {
            const size_t slot = (b->ev_calls - 1 - k) % kTimingRing;
            float ms = 0.f;
            VPT_HIP(hipEventElapsedTime(&ms, b->ev[2 * slot], b->ev[2 * slot + 1]));
            sum += ms;
        }
        if (n) *score_kernel_ms = float(sum / double(n));
        b->ev_calls = 0;
    }
    return VPT_OK;
}

vpt_status vpt_batch_kernel_times(vpt_batch* b, float* ms_out, size_t capacity, size_t* n_out) {
    if (!b || !n_out) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: NULL argument");
    if (b->pending) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: batch: call vpt_batch_sync first");
    const size_t n = std::min(std::min(b->ev_calls, kTimingRing), ms_out ? capacity : size_t(0));
    for (size_t k = 0; k < n; ++k) {   // oldest first
        const size_t slot = (b->ev_calls - n + k) % kTimingRing;
        VPT_HIP(hipEventElapsedTime(&ms_out[k], b->ev[2 * slot], b->ev[2 * slot + 1]));
    }
    *n_out = ms_out ? n : std::min(b->ev_calls, kTimingRing);
    return VPT_OK;
}

vpt_status vpt_batch_set_max_sentence_chars(vpt_batch* b, uint64_t max_sentence_chars) {
    if (!b) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: NULL argument");
    b->max_chars = max_sentence_chars;
    return VPT_OK;
}

vpt_status vpt_batch_set_flags(vpt_batch* b, unsigned flags) {
    if (!b) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: NULL argument");
    if (flags & ~unsigned(VPT_FLAG_ALL)) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: flags: unknown bit");
    b->flags = flags;
    return VPT_OK;
}

vpt_status vpt_batch_phase_cycles(vpt_batch* b, uint64_t cycles[8]) {
    if (!b || !cycles) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: NULL argument");
    std::memset(cycles, 0, 64);
    if (!b->d_prof) return VPT_OK;
    VPT_HIP(hipSetDevice(b->device));
    VPT_HIP(hipDeviceSynchronize());
    VPT_HIP(hipMemcpy(cycles, b->d_prof, 64, hipMemcpyDeviceToHost));
    VPT_HIP(hipMemset(b->d_prof, 0, 64));
    return VPT_OK;
}

vpt_status vpt_batch_node_reads(vpt_batch* b, uint64_t reads[8]) {
    if (!b || !reads) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: NULL argument");
    std::memset(reads, 0, 64);
    if (!b->d_prof) return VPT_OK;
    VPT_HIP(hipSetDevice(b->device));
    VPT_HIP(hipDeviceSynchronize());
    VPT_HIP(hipMemcpy(reads, b->d_prof + 8, 64, hipMemcpyDeviceToHost));
    VPT_HIP(hipMemset(b->d_prof + 8, 0, 64));
    return VPT_OK;
}

namespace {
vpt_status emit_device(const vpt_predictor* p, vpt_batch* b, const uint8_t* d_utf8, const uint64_t* d_byte_offsets,
                       const uint64_t* d_out_offsets, size_t n_sentences, uint64_t total_boundaries, const uint8_t* d_labels,
                       bool tagged, uint8_t* d_text_out, uint64_t text_capacity, uint64_t* d_text_offsets_out,
                       hipStream_t stream, uint64_t* total_out = nullptr, const uint64_t* chain_in = nullptr, uint64_t* chain_out = nullptr);

vpt_status predict_device_impl(const vpt_predictor* p, vpt_batch* b, const uint8_t* d_utf8, const uint64_t* d_byte_offsets,
                               const uint64_t* d_out_offsets, size_t n_sentences, uint64_t total_boundaries,
                               uint64_t max_sentence_bytes, int32_t* d_scores, uint8_t* d_labels, void* hip_stream) {
    if (!p || !b || b->pred != p) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: batch: does not belong to this predictor");
    if (n_sentences == 0) { b->last_tiles = 0; return VPT_OK; }   // nothing enqueued; earlier work stays pending
    if (!d_utf8 || !d_byte_offsets || !d_out_offsets) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: NULL device pointer");
    if (n_sentences >= 0xFFFFFFFFull) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: n_sentences: at most 2^32-2 per call");
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    VPT_HIP(hipSetDevice(p->device));
    vpt::ScoreParams P{};
    P.ct = p->ct; P.tt = p->tt; P.pk = p->pk; P.type_table = p->d_type_table;
    P.ctype = p->d_ctype;
    P.cinfo = (b->flags & VPT_FLAG_KYTEA_FULLWIDTH) ? p->d_cinfo + 65536 : nullptr;
    P.cid = p->d_cid ? p->d_cid + ((b->flags & VPT_FLAG_KYTEA_FULLWIDTH) ? 65536 : 0) : nullptr;
    P.post = b->flags & 0xFEu;
    P.type_window = p->type_window; P.type_kind = p->type_kind; P.bias = p->bias; P.pad = p->pad;
    P.force_window_table = p->knobs.force_window_table ? 1u : 0u;
    // flat positions of the longest sentence: its chars, bounded by the caller's hint or else by its bytes
    const uint64_t max_chars = (b->max_chars && b->max_chars < max_sentence_bytes) ? b->max_chars : max_sentence_bytes;
    const uint64_t total_flat = total_boundaries + uint64_t(n_sentences) * uint64_t(1 + p->pad);
    const uint64_t total_chars = total_boundaries + n_sentences;
    // The specialised kernel takes whole-sentence tiles while every sentence is short, and tiles cut at any flat position (with a
    // halo of the longest pattern on either side) otherwise -- a sentence of any length is scored there.  Only a model whose longest
    // pattern leaves no room for a tile between its halos sends long sentences to the general kernels.
    bool fast = vpt::fast_path_supported(P) && !b->knobs.force_generic;
    const uint64_t fast_cap = fast ? uint64_t(vpt::fast_path_cap(P)) : 0;   // flat positions per tile of the instance that scores this predictor
    vpt::CutGeometry cut{};
    bool cut_tiles = false;
    if (fast) {
        const uint32_t lmax = std::max<uint32_t>(p->info.max_pattern_chars, 3);
        // A pattern of m chars that starts at s touches the boundaries s - wl .. s + max(wl, m) - 1 (layout.h, row_lo / row_hi): a
        // boundary needs the start positions within max(lmax, wl) - 1 to its left and within wl to its right, and those need their
        // chars -- lmax - 1 further on -- and the types their row is indexed by.
        const uint32_t wl = p->pk.wl;
        const uint32_t levels = p->pk.trow_mode == vpt::kTypeRowsGlobal ? p->pk.trow_levels : 3u;
        cut.halo_left = std::max<uint32_t>(lmax - 1, wl); cut.halo_right = wl + std::max<uint32_t>(std::max<uint32_t>(lmax - 1, wl), levels);
        cut.pad = uint32_t(p->pad); cut.cap = uint32_t(fast_cap);
        cut.cap_eff = uint32_t(fast_cap - vpt::kFastStageSlack);
        const int64_t room = int64_t(cut.cap_eff) - int64_t(p->pad) - int64_t(cut.halo_left) - int64_t(cut.halo_right);
        const bool can_cut = room >= 256;
        cut.tile_flat = can_cut ? uint32_t(room) : 0u;
        cut.mis = uint32_t(reinterpret_cast<uintptr_t>(d_utf8) & 15u);
        const bool fits_whole = max_chars + 2 * uint64_t(p->pad) <= fast_cap / 2;
        const bool whole_possible = max_chars + 2 * uint64_t(p->pad) + fast_cap / 2 <= fast_cap;
        // Whole-sentence tiles are cut every (capacity - longest sentence) positions: the longer the longest sentence, the emptier
        // they run, while a tile cut anywhere is always full -- at the price of the index of the text (a pass over it).  Measured on
        // configs[4] (8 .. 512 chars, profiles/r03_u_cut_vs_whole.txt): kernel 2.14 -> 1.99 ms, step 3.80 -> 3.76; on configs[1] (64
        // chars) the kernel is the same and the index costs 19 us of 117.  So: cut above a quarter of the capacity.
        const bool prefer_cut = max_chars > fast_cap / 4;
        cut_tiles = can_cut && (b->knobs.force_cut > 0 || (b->knobs.force_cut == 0 ? (prefer_cut || !fits_whole) : !whole_possible));
        if (!cut_tiles && max_chars + 2 * uint64_t(p->pad) + fast_cap / 2 > fast_cap) fast = false;   // neither kind of tile holds the batch
    }
    const uint64_t cap = fast ? fast_cap : vpt::kCap;
    // Whole-sentence tiles are cut every `tile_flat` flat positions (chars + separators) and end with the sentence that crosses
    // the cut, so a tile holds < tile_flat + longest sentence: pick tile_flat to fill the kernel's LDS capacity.
    uint64_t tile_flat = cap / 2;
    if (cut_tiles) tile_flat = cut.tile_flat;
    else if (max_chars + 2 * uint64_t(p->pad) + cap / 2 <= cap) tile_flat = cap - 2 * uint64_t(p->pad) - max_chars;
    // Whole rounds: the chip runs `slots` tiles at a time; cutting the batch into a multiple of that many tiles (by
    // shrinking the tiles a little) avoids a last round that leaves most CUs idle.
    if (p->tile_slots > 0) {
        const uint64_t n_min = (total_flat + tile_flat - 1) / tile_flat;
        const uint64_t rounds = (n_min + p->tile_slots - 1) / p->tile_slots;
        const uint64_t even = (total_flat + rounds * p->tile_slots - 1) / (rounds * p->tile_slots);
        if (even < tile_flat) tile_flat = std::max<uint64_t>(even, 256);
    }
    if (b->knobs.tile_flat && b->knobs.tile_flat < tile_flat) tile_flat = std::max<uint64_t>(b->knobs.tile_flat, 16);
    if (cut_tiles) {   // the window a cut tile decodes and walks: its own positions and the two halos
        cut.tile_flat = uint32_t(tile_flat);
        cut.cap_eff = uint32_t(p->pad) + cut.halo_left + cut.tile_flat + cut.halo_right;
        // what the tile planner promises the scoring kernel (kernels_fast.hip: a tile of n positions stages at most 4 n + 2 * 256 bytes,
        // the staging area holds 4 * cap + 15): an internal inconsistency, not the caller's offsets
        if (uint64_t(cut.cap_eff) * 4 + 2 * 256 > cap * 4 + 15 || cut.cap_eff > cap)
            return fail(VPT_RUNTIME_ERROR, "internal error: the tile planner produced a cut tile that does not fit the kernel's staging area");
    }
    const uint64_t n_tiles64 = (total_flat + tile_flat - 1) / tile_flat;
    if (n_tiles64 >= 0x7FFFFFFFull) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: batch too large for one call");
    const uint32_t n_tiles = uint32_t(n_tiles64);
    vpt_status st;
    bool need_slow = false;
    uint32_t slow_blocks = 0, scratch_cap = 0;
    uint64_t slab = 0;
    if (fast && cut_tiles) {
        if ((st = grow(&b->d_tiles, &b->tiles_cap, size_t(n_tiles) + 1)) != VPT_OK) return st;
        size_t n_local = 0, n_super = 0;
        vpt::cut_index_entries(total_chars, &n_local, &n_super);
        if ((st = grow(&b->d_cut_local, &b->cut_local_cap, n_local + 16)) != VPT_OK) return st;
        if ((st = grow(&b->d_cut_super, &b->cut_super_cap, n_super + 16)) != VPT_OK) return st;
    } else {
        if (size_t(n_tiles) + 1 > b->tile_cap || !b->d_tile_first) {
            (void)hipFree(b->d_slow_list); b->d_slow_list = nullptr;
            size_t tcap = b->tile_cap;
            if ((st = grow(&b->d_tile_first, &tcap, size_t(n_tiles) + 1)) != VPT_OK) return st;
            b->tile_cap = tcap;
            VPT_HIP(hipMalloc(reinterpret_cast<void**>(&b->d_slow_list), tcap * sizeof(uint32_t) + 64));
        }
        // long-sentence scratch of the general kernels (only when a sentence might not fit the LDS tile)
        need_slow = !fast && max_chars + 2 * uint64_t(p->pad) + tile_flat > cap;
        if (need_slow) {
            const uint64_t cap64 = max_chars + 2 * uint64_t(p->pad) + vpt::kMargin + 8;
            if (cap64 >= 0x7FFFFFF0ull) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: max_sentence_bytes: too large");
            scratch_cap = uint32_t((cap64 + 15) & ~15ull);
            slab = (uint64_t(scratch_cap) * 9 + 255) & ~255ull;
            slow_blocks = uint32_t(std::max<uint64_t>(1, std::min<uint64_t>(64, (8ull << 30) / slab)));
            const size_t need = size_t(slab) * slow_blocks;
            if (need > b->scratch_bytes) {
                (void)hipFree(b->d_scratch); b->d_scratch = nullptr; b->scratch_bytes = 0;
                VPT_HIP(hipMalloc(reinterpret_cast<void**>(&b->d_scratch), need));
                b->scratch_bytes = need;
            }
        }
    }
    P.text = d_utf8; P.boff = d_byte_offsets; P.ooff = d_out_offsets; P.tile_first = b->d_tile_first; P.tiles = (fast && cut_tiles) ? b->d_tiles : nullptr;
    P.scores = d_scores; P.labels = d_labels; P.status = b->d_ctrl; P.slow_list = b->d_slow_list; P.slow_count = b->d_ctrl + 1;
    P.scratch = b->d_scratch; P.scratch_stride = slab; P.scratch_cap = scratch_cap;
    P.prof = b->d_prof;
    P.total_chars = total_chars;
    // A predictor with tag models: the specialised kernel also leaves the decoded chars behind, and a vpt_fill_tags_batch_device
    // call for the same buffers on this workspace (Sentence::fill_tags follows Predictor::predict on the same sentence,
    // predictor.rs:542) skips its own decode pass.
    b->cps_text = nullptr;
    if (p->has_tags && p->predict_tags && fast) {
        vpt_status st2 = grow(&b->d_cps, &b->cps_cap, size_t(total_chars) + 16);
        if (st2 != VPT_OK) return st2;
        P.cps_out = b->d_cps;
        b->cps_text = d_utf8; b->cps_ooff = d_out_offsets; b->cps_sentences = n_sentences; b->cps_boundaries = total_boundaries;
        b->cps_flags = b->flags & VPT_FLAG_KYTEA_FULLWIDTH;
    }
    if (b->knobs.debug_ablate) { P.debug = b->knobs.debug_ablate; P.ct.debug = P.debug; P.tt.debug = P.debug; }

    P.n_sent = n_sentences; P.tile_flat = uint32_t(tile_flat); P.n_tiles = n_tiles;
    // the tiles (a kernel of its own: finding them at the head of every workgroup measured slower, profiles/r02_c1_ab.jsonl); for
    // cut tiles preceded by the lead-byte index of the text
    if (fast && cut_tiles) VPT_HIP(vpt::launch_assign_tiles_cut(P, cut, n_tiles, total_chars, b->d_cut_local, b->d_cut_super, b->d_tiles, b->d_ctrl, stream));
    else VPT_HIP(vpt::launch_assign_tiles(d_out_offsets, n_sentences, p->pad, uint32_t(tile_flat), n_tiles, b->d_tile_first, b->d_ctrl, stream));
    if (b->split_stream) {
        VPT_HIP(hipEventRecord(b->split_event, stream));
        VPT_HIP(hipStreamWaitEvent(b->split_stream, b->split_event, 0));
        stream = b->split_stream;
    }
    const size_t slot = b->ev_calls % kTimingRing;
    if (b->timing) VPT_HIP(hipEventRecord(b->ev[2 * slot], stream));
    if (fast) VPT_HIP(vpt::launch_score_tiles_fast(P, n_tiles, stream));
    else VPT_HIP(vpt::launch_score_tiles(P, p->chunks, n_tiles, stream));
    if (need_slow) VPT_HIP(vpt::launch_score_slow(P, p->chunks, slow_blocks, stream));   // inside the timed events: it is part of the scoring
    if (b->timing) { VPT_HIP(hipEventRecord(b->ev[2 * slot + 1], stream)); ++b->ev_calls; }
    b->last_tiles = n_tiles; b->last_stream = stream; b->pending = true;
    b->last_tile_flat = uint32_t(tile_flat); b->last_plan = !fast ? 0u : cut_tiles ? 2u : 1u;
    return VPT_OK;
}
}  // namespace

vpt_status vpt_predict_batch_device(const vpt_predictor* p, vpt_batch* b, const uint8_t* d_utf8, const uint64_t* d_byte_offsets,
                                    const uint64_t* d_out_offsets, size_t n_sentences, uint64_t total_boundaries,
                                    uint64_t max_sentence_bytes, int32_t* d_scores, uint8_t* d_labels, void* hip_stream) {
    return predict_device_impl(p, b, d_utf8, d_byte_offsets, d_out_offsets, n_sentences, total_boundaries, max_sentence_bytes, d_scores, d_labels, hip_stream);
}

// Predictor::predict + Sentence::write_tokenized_text (no tags) for a batch as ONE call: the scoring launch and the writer's, back to back on the
// stream; scores and labels are optional outputs (no d_labels: the labels stay in the workspace).  (Rounds 4 - 5 fused the writer into the scoring
// kernel as a fourth phase: slower than the two launches on every batch size and, chunk by chunk, in vpt_tokenize_batch too -- HISTORY.md.)
vpt_status vpt_predict_write_batch_device(const vpt_predictor* p, vpt_batch* b, const uint8_t* d_utf8, const uint64_t* d_byte_offsets,
                                          const uint64_t* d_out_offsets, size_t n_sentences, uint64_t total_boundaries,
                                          uint64_t max_sentence_bytes, int32_t* d_scores, uint8_t* d_labels, uint8_t* d_text_out,
                                          uint64_t text_capacity, uint64_t* d_text_offsets_out, void* hip_stream) {
    if (!p || !b || b->pred != p) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: batch: does not belong to this predictor");
    if (!d_text_offsets_out || (text_capacity && !d_text_out)) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: NULL device pointer");
    if (n_sentences == 0) {
        VPT_HIP(hipSetDevice(p->device));
        VPT_HIP(hipMemsetAsync(d_text_offsets_out, 0, sizeof(uint64_t), static_cast<hipStream_t>(hip_stream)));
        b->last_stream = static_cast<hipStream_t>(hip_stream); b->pending = true; b->cps_text = nullptr;
        return VPT_OK;
    }
    if (!d_utf8 || !d_byte_offsets || !d_out_offsets) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: NULL device pointer");
    uint8_t* labels = d_labels;
    if (!labels) {
        const vpt_status st = grow(&b->d_tlab, &b->tlab_cap, size_t(total_boundaries) + 16);
        if (st != VPT_OK) return st;
        labels = b->d_tlab;
    }
    const vpt_status st = predict_device_impl(p, b, d_utf8, d_byte_offsets, d_out_offsets, n_sentences, total_boundaries, max_sentence_bytes, d_scores, labels, hip_stream);
    if (st != VPT_OK) return st;
    return emit_device(p, b, d_utf8, d_byte_offsets, d_out_offsets, n_sentences, total_boundaries, labels, false, d_text_out, text_capacity, d_text_offsets_out,
                       static_cast<hipStream_t>(hip_stream));
}

vpt_status vpt_batch_last_plan(const vpt_batch* b, uint32_t* n_tiles, uint32_t* tile_flat, uint32_t* kind) {
    if (!b) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: NULL argument");
    if (n_tiles) *n_tiles = b->last_tiles;
    if (tile_flat) *tile_flat = b->last_tile_flat;
    if (kind) *kind = b->last_plan;
    return VPT_OK;
}

vpt_status vpt_batch_sync(vpt_batch* b) {
    if (!b) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: NULL argument");
    // A caller that waits for the device may rewrite its buffers afterwards: the chars a predict call left decoded for the fill_tags
    // call behind it (matched by buffer address and shape only) are good for a fill_tags enqueued BEFORE the next sync, no longer.
    b->cps_text = nullptr;
    if (!b->pending) return VPT_OK;
    VPT_HIP(hipSetDevice(b->device));
    uint32_t ctrl[2] = {0, 0};
    VPT_HIP(hipMemcpyAsync(ctrl, b->d_ctrl, sizeof(ctrl), hipMemcpyDeviceToHost, b->last_stream));
    VPT_HIP(hipStreamSynchronize(b->last_stream));
    b->pending = false;
    if (ctrl[0]) VPT_HIP(hipMemset(b->d_ctrl, 0, sizeof(uint32_t)));   // reported once; accumulates over every call enqueued since the last sync
    return status_from_bits(ctrl[0]);
}

namespace {

// How vpt_predict_batch takes a host batch of `total_chars` (measured on MI355X behind PCIe 5 x16, profiles/r02_g_e2e.txt):
//   up to 1.5 M chars            one copy in, the kernels, one copy out
//   up to 16 M chars             predict_lanes: 4 lanes, 512 K-char chunks.  The batch is over in about a millisecond, so what counts is
//                                how soon the first copy out starts and that nothing waits on the host: 0.85 ms per 6.4 M chars against
//                                0.98 through the events (3 or 5 lanes, or 256 K / 1 M-char chunks: 0.93 .. 1.05)
//   more                         predict_pipelined: three streams and events, 4 M-char chunks: 6.9 .. 8.1 ms per 64 M chars, steadier over
//                                many chunks than the lanes, whose copies contend (7.3 .. 10 ms)
// VPT_CHUNK_CHARS / VPT_PIPE_LANES (0 = the event pipeline) override; the tests use them to cut small batches into many chunks.
struct PipePlan { int lanes; uint64_t chunk; bool pipelined; };
PipePlan pipeline_plan(const PredictorKnobs& knobs, uint64_t total_chars) {
    PipePlan plan = total_chars <= (uint64_t(16) << 20) ? PipePlan{4, uint64_t(512) << 10, false} : PipePlan{0, uint64_t(4) << 20, false};
    if (knobs.pipe_lanes >= 0) plan.lanes = knobs.pipe_lanes;
    if (knobs.chunk_chars) plan.chunk = knobs.chunk_chars;
    plan.pipelined = total_chars > (plan.lanes > 0 ? 3 * plan.chunk : plan.chunk + plan.chunk / 2);
    return plan;
}

// vpt_predict_batch for a LARGE batch: the copy in of chunk k + 1, the kernels of chunk k and the copy out of chunk k - 1 run at
// the same time on three streams over two sets of device buffers, so that a caller with PINNED buffers (vpt_host_alloc) gets
// both directions of the PCIe link busy at once instead of a copy-launch-copy sequence; pageable buffers go through the
// runtime's staging and still overlap with the kernels.  The host walks the sentences ONCE, chunk by chunk -- validating them,
// rebasing their offsets into pinned staging, finding the cut -- while the chunks before are in flight; a chunk costs eleven
// runtime calls (two copies in, two launches, two copies out, events) and a wait of the host for the chunk two before it.  The
// runtime performs this stream's copies out as blit kernels, which the next chunk's kernels queue behind (timeline in
// profiles/r02_g_e2e.txt): copy out and kernels take turns, 0.24 ms per 2 M chars -- 15 % above what the link does both ways.
vpt_status predict_pipelined(const vpt_predictor* p, vpt_batch* b, const uint8_t* utf8, const uint64_t* byte_offsets, size_t n_sentences,
                             int32_t* scores_out, uint8_t* labels_out, const uint64_t* out_offsets, uint64_t chunk_chars) {
    if (!b->s_in) {
        VPT_HIP(hipStreamCreateWithFlags(&b->s_in, hipStreamNonBlocking));
        VPT_HIP(hipStreamCreateWithFlags(&b->s_out, hipStreamNonBlocking));
        for (auto& ps : b->pipe) {
            VPT_HIP(hipEventCreateWithFlags(&ps.ev_in, hipEventDisableTiming));
            VPT_HIP(hipEventCreateWithFlags(&ps.ev_k, hipEventDisableTiming));
            VPT_HIP(hipEventCreateWithFlags(&ps.ev_out, hipEventDisableTiming));
        }
    }
    const uint64_t total_chars = out_offsets[n_sentences] - out_offsets[0] + n_sentences;
    const size_t max_chunks = std::min<size_t>(n_sentences, size_t(total_chars / chunk_chars) + 2);   // every chunk holds a sentence
    const size_t need_off = 2 * (n_sentences + max_chunks);   // a chunk of n sentences stages 2 (n + 1) offsets
    if (need_off > b->h_off_cap) {
        if (b->h_off) (void)hipHostFree(b->h_off);
        b->h_off = nullptr; b->h_off_cap = 0;
        VPT_HIP(hipHostMalloc(reinterpret_cast<void**>(&b->h_off), (need_off + need_off / 2) * sizeof(uint64_t), hipHostMallocDefault));
        b->h_off_cap = need_off + need_off / 2;
    }
    vpt_status st;
    size_t i = 0, staged = 0;
    for (size_t k = 0; i < n_sentences; ++k) {
        vpt_batch::PipeSet& ps = b->pipe[k & 1];
        // ---- walk the chunk's sentences: validate, rebase, cut
        const size_t a = i;
        const uint64_t t0 = byte_offsets[a], o0 = out_offsets[a];
        uint64_t* hb = b->h_off + staged;             // n + 1 byte offsets, then n + 1 boundary offsets, filled below
        uint64_t chars = 0, max_bytes = 0, max_chars = 0;
        while (i < n_sentences && chars < chunk_chars) {
            if (byte_offsets[i + 1] <= byte_offsets[i]) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: text: must contain at least one character");
            const uint64_t nby = byte_offsets[i + 1] - byte_offsets[i];
            if (out_offsets[i + 1] < out_offsets[i] || out_offsets[i + 1] - out_offsets[i] + 1 > nby)
                return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: out_offsets: do not match the text (or the text is not valid UTF-8)");
            const uint64_t nch = out_offsets[i + 1] - out_offsets[i] + 1;
            max_bytes = std::max(max_bytes, nby); max_chars = std::max(max_chars, nch);
            chars += nch;
            ++i;
        }
        const size_t n = i - a;
        uint64_t* ho = hb + (n + 1);
        for (size_t j = 0; j <= n; ++j) { hb[j] = byte_offsets[a + j] - t0; ho[j] = out_offsets[a + j] - o0; }
        staged += 2 * (n + 1);
        const uint64_t nbytes = byte_offsets[i] - t0, nb = out_offsets[i] - o0;
        // ---- this set's buffers are free once chunk k - 2 has been scored (inputs) and copied out (outputs)
        if (k >= 2) { VPT_HIP(hipEventSynchronize(ps.ev_out)); }   // also bounds how far the host runs ahead; growing a buffer below is then safe
        if ((st = grow(&ps.text, &ps.text_cap, size_t(nbytes) + 32)) != VPT_OK) return st;
        if ((st = grow(&ps.off, &ps.off_cap, 2 * (n + 1))) != VPT_OK) return st;
        if (scores_out && (st = grow(&ps.scores, &ps.scores_cap, size_t(nb) + 1)) != VPT_OK) return st;
        if (labels_out && (st = grow(&ps.labels, &ps.labels_cap, size_t(nb) + 1)) != VPT_OK) return st;
        VPT_HIP(hipMemcpyAsync(ps.text, utf8 + t0, size_t(nbytes), hipMemcpyHostToDevice, b->s_in));
        VPT_HIP(hipMemcpyAsync(ps.off, hb, 16 * (n + 1), hipMemcpyHostToDevice, b->s_in));
        VPT_HIP(hipEventRecord(ps.ev_in, b->s_in));
        VPT_HIP(hipStreamWaitEvent(b->own_stream, ps.ev_in, 0));
        b->max_chars = max_chars;
        st = vpt_predict_batch_device(p, b, ps.text, ps.off, ps.off + (n + 1), n, nb, max_bytes, scores_out ? ps.scores : nullptr,
                                      labels_out ? ps.labels : nullptr, b->own_stream);
        if (st != VPT_OK) return st;
        VPT_HIP(hipEventRecord(ps.ev_k, b->own_stream));
        VPT_HIP(hipStreamWaitEvent(b->s_out, ps.ev_k, 0));
        if (scores_out && nb) VPT_HIP(hipMemcpyAsync(scores_out + o0, ps.scores, 4 * size_t(nb), hipMemcpyDeviceToHost, b->s_out));
        if (labels_out && nb) VPT_HIP(hipMemcpyAsync(labels_out + o0, ps.labels, size_t(nb), hipMemcpyDeviceToHost, b->s_out));
        VPT_HIP(hipEventRecord(ps.ev_out, b->s_out));
    }
    VPT_HIP(hipStreamSynchronize(b->s_in));
    VPT_HIP(hipStreamSynchronize(b->s_out));
    return vpt_batch_sync(b);   // the device's verdict over every chunk (the status word accumulates)
}

// vpt_predict_batch for a batch of a few chunks: every chunk's copy in, kernels and copy out are enqueued IN ORDER on one stream,
// and the chunks alternate over `n_lanes` streams (each with a workspace and one set of device buffers of its own).  The
// overlap is between the lanes -- one copies out while the next copies in and scores -- and nothing crosses streams: no events,
// no host synchronisation before the end (a set's reuse is ordered by its own stream), and the copies out go through SDMA.
vpt_status predict_lanes(const vpt_predictor* p, vpt_batch* b, const uint8_t* utf8, const uint64_t* byte_offsets, size_t n_sentences,
                         int32_t* scores_out, uint8_t* labels_out, const uint64_t* out_offsets, uint64_t chunk_chars, int n_lanes) {
    constexpr int kMaxLanes = 8;
    n_lanes = std::max(1, std::min(n_lanes, kMaxLanes));
    Workspace extra[kMaxLanes - 1];
    vpt_batch* lane[kMaxLanes] = {b};
    vpt_status st;
    for (int l = 1; l < n_lanes; ++l) {
        if ((st = acquire(p, &extra[l - 1])) != VPT_OK) return st;
        lane[l] = extra[l - 1].b;
        lane[l]->flags = b->flags;
    }
    const uint64_t total_chars = out_offsets[n_sentences] - out_offsets[0] + n_sentences;
    const size_t max_chunks = std::min<size_t>(n_sentences, size_t(total_chars / chunk_chars) + 2);
    const size_t need_off = 2 * (n_sentences + max_chunks);   // a chunk of n sentences stages 2 (n + 1) offsets
    if (need_off > b->h_off_cap) {
        if (b->h_off) (void)hipHostFree(b->h_off);
        b->h_off = nullptr; b->h_off_cap = 0;
        VPT_HIP(hipHostMalloc(reinterpret_cast<void**>(&b->h_off), (need_off + need_off / 2) * sizeof(uint64_t), hipHostMallocDefault));
        b->h_off_cap = need_off + need_off / 2;
    }
    size_t i = 0, staged = 0;
    for (size_t k = 0; i < n_sentences; ++k) {
        vpt_batch* bb = lane[k % size_t(n_lanes)];
        vpt_batch::PipeSet& ps = bb->pipe[0];
        hipStream_t s = bb->own_stream;
        const size_t a = i;
        const uint64_t t0 = byte_offsets[a], o0 = out_offsets[a];
        uint64_t* hb = b->h_off + staged;
        uint64_t chars = 0, max_bytes = 0, max_chars = 0;
        while (i < n_sentences && chars < chunk_chars) {
            if (byte_offsets[i + 1] <= byte_offsets[i]) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: text: must contain at least one character");
            const uint64_t nby = byte_offsets[i + 1] - byte_offsets[i];
            if (out_offsets[i + 1] < out_offsets[i] || out_offsets[i + 1] - out_offsets[i] + 1 > nby)
                return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: out_offsets: do not match the text (or the text is not valid UTF-8)");
            const uint64_t nch = out_offsets[i + 1] - out_offsets[i] + 1;
            max_bytes = std::max(max_bytes, nby); max_chars = std::max(max_chars, nch);
            chars += nch;
            ++i;
        }
        const size_t n = i - a;
        uint64_t* ho = hb + (n + 1);
        for (size_t j = 0; j <= n; ++j) { hb[j] = byte_offsets[a + j] - t0; ho[j] = out_offsets[a + j] - o0; }
        staged += 2 * (n + 1);
        const uint64_t nbytes = byte_offsets[i] - t0, nb = out_offsets[i] - o0;
        // a buffer that has to grow may still be read or written by the lane's earlier chunk
        if (size_t(nbytes) + 32 > ps.text_cap || 2 * (n + 1) > ps.off_cap || (scores_out && size_t(nb) + 1 > ps.scores_cap) ||
            (labels_out && size_t(nb) + 1 > ps.labels_cap))
            VPT_HIP(hipStreamSynchronize(s));
        if ((st = grow(&ps.text, &ps.text_cap, size_t(nbytes) + 32)) != VPT_OK) return st;
        if ((st = grow(&ps.off, &ps.off_cap, 2 * (n + 1))) != VPT_OK) return st;
        if (scores_out && (st = grow(&ps.scores, &ps.scores_cap, size_t(nb) + 1)) != VPT_OK) return st;
        if (labels_out && (st = grow(&ps.labels, &ps.labels_cap, size_t(nb) + 1)) != VPT_OK) return st;
        VPT_HIP(hipMemcpyAsync(ps.text, utf8 + t0, size_t(nbytes), hipMemcpyHostToDevice, s));
        VPT_HIP(hipMemcpyAsync(ps.off, hb, 16 * (n + 1), hipMemcpyHostToDevice, s));
        bb->max_chars = max_chars;
        st = vpt_predict_batch_device(p, bb, ps.text, ps.off, ps.off + (n + 1), n, nb, max_bytes, scores_out ? ps.scores : nullptr,
                                      labels_out ? ps.labels : nullptr, s);
        if (st != VPT_OK) return st;
        if (scores_out && nb) VPT_HIP(hipMemcpyAsync(scores_out + o0, ps.scores, 4 * size_t(nb), hipMemcpyDeviceToHost, s));
        if (labels_out && nb) VPT_HIP(hipMemcpyAsync(labels_out + o0, ps.labels, size_t(nb), hipMemcpyDeviceToHost, s));
    }
    // the device's verdict over every chunk (a lane's status word accumulates): the lanes' words are fetched together, one
    // round trip for all of them instead of one each
    uint32_t ctrl[kMaxLanes][2] = {};
    for (int l = 0; l < n_lanes; ++l)
        if (lane[l]->pending) VPT_HIP(hipMemcpyAsync(ctrl[l], lane[l]->d_ctrl, sizeof(ctrl[l]), hipMemcpyDeviceToHost, lane[l]->last_stream));
    st = VPT_OK;
    for (int l = 0; l < n_lanes; ++l) {
        if (!lane[l]->pending) continue;
        VPT_HIP(hipStreamSynchronize(lane[l]->last_stream));
        lane[l]->pending = false;
        if (ctrl[l][0]) {
            VPT_HIP(hipMemset(lane[l]->d_ctrl, 0, sizeof(uint32_t)));
            if (st == VPT_OK) st = status_from_bits(ctrl[l][0]);
        }
    }
    return st;
}

}  // namespace

vpt_status vpt_predict_batch(const vpt_predictor* p, const uint8_t* utf8, const uint64_t* byte_offsets, size_t n_sentences,
                             int32_t* scores_out, uint8_t* labels_out, const uint64_t* out_offsets) {
    return vpt_predict_batch_flags(p, utf8, byte_offsets, n_sentences, scores_out, labels_out, out_offsets, 0u);
}

vpt_status vpt_predict_batch_flags(const vpt_predictor* p, const uint8_t* utf8, const uint64_t* byte_offsets, size_t n_sentences,
                                   int32_t* scores_out, uint8_t* labels_out, const uint64_t* out_offsets, unsigned flags) {
    if (flags & ~unsigned(VPT_FLAG_ALL)) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: flags: unknown bit");
    if (!p) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: predictor: must not be NULL");
    if (n_sentences == 0) return VPT_OK;
    if (!utf8 || !byte_offsets || !out_offsets) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: NULL argument");
    VPT_HIP(hipSetDevice(p->device));
    Workspace w;
    vpt_status st = acquire(p, &w);
    if (st != VPT_OK) return st;
    vpt_batch* b = w.b;
    b->flags = flags;
    if (byte_offsets[n_sentences] >= byte_offsets[0] && out_offsets[n_sentences] >= out_offsets[0]) {   // a batch of several chunks goes through a copy/compute pipeline
        const PipePlan plan = pipeline_plan(p->knobs, out_offsets[n_sentences] - out_offsets[0] + n_sentences);
        if (plan.pipelined && plan.lanes > 0) return predict_lanes(p, b, utf8, byte_offsets, n_sentences, scores_out, labels_out, out_offsets, plan.chunk, plan.lanes);
        if (plan.pipelined) return predict_pipelined(p, b, utf8, byte_offsets, n_sentences, scores_out, labels_out, out_offsets, plan.chunk);
    }
    uint64_t total_b = 0, max_bytes = 0, max_chars = 0;
    if ((st = stage(b, utf8, byte_offsets, out_offsets, n_sentences, nullptr, &total_b, &max_bytes, &max_chars)) != VPT_OK) return st;
    b->max_chars = max_chars;
    hipStream_t s = b->own_stream;
    st = vpt_predict_batch_device(p, b, b->d_text, b->d_boff, b->d_ooff, n_sentences, total_b, max_bytes,
                                  scores_out ? b->d_scores : nullptr, labels_out ? b->d_labels : nullptr, s);
    if (st != VPT_OK) return st;
    if (scores_out && total_b) VPT_HIP(hipMemcpyAsync(scores_out + out_offsets[0], b->d_scores, 4 * total_b, hipMemcpyDeviceToHost, s));
    if (labels_out && total_b) VPT_HIP(hipMemcpyAsync(labels_out + out_offsets[0], b->d_labels, total_b, hipMemcpyDeviceToHost, s));
    return vpt_batch_sync(b);
}

vpt_status vpt_predictor_n_tags(const vpt_predictor* p, uint32_t* n_tags) {
    if (!p || !n_tags) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: NULL argument");
    *n_tags = p->n_tags;
    return VPT_OK;
}

vpt_status vpt_predictor_max_tag_suffix(const vpt_predictor* p, uint32_t* n_bytes) {
    if (!p || !n_bytes) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: NULL argument");
    *n_bytes = p->max_tag_suffix;
    return VPT_OK;
}

vpt_status vpt_predictor_tag_score_stride(const vpt_predictor* p, uint32_t* stride) {
    if (!p || !stride) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: NULL argument");
    *stride = p->max_tag_scores;
    return VPT_OK;
}

vpt_status vpt_fill_tags_batch(const vpt_predictor* p, const uint8_t* utf8, const uint64_t* byte_offsets, size_t n_sentences,
                               const uint64_t* out_offsets, const uint8_t* labels, int32_t* tags_out) {
    return vpt_fill_tags_batch_flags(p, utf8, byte_offsets, n_sentences, out_offsets, labels, tags_out, 0u);
}

vpt_status vpt_fill_tags_batch_flags(const vpt_predictor* p, const uint8_t* utf8, const uint64_t* byte_offsets, size_t n_sentences,
                                     const uint64_t* out_offsets, const uint8_t* labels, int32_t* tags_out, unsigned flags) {
    return vpt_fill_tags_scores_batch(p, utf8, byte_offsets, n_sentences, out_offsets, labels, flags, tags_out, nullptr, nullptr);
}

vpt_status vpt_fill_tags_scores_batch(const vpt_predictor* p, const uint8_t* utf8, const uint64_t* byte_offsets, size_t n_sentences,
                                      const uint64_t* out_offsets, const uint8_t* labels, unsigned flags, int32_t* tags_out,
                                      int32_t* tag_scores_out, int32_t* tag_models_out) {
    if (flags & ~unsigned(VPT_FLAG_KYTEA_FULLWIDTH)) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: flags: unknown bit");
    if (!p) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: predictor: must not be NULL");
    if (!p->predict_tags) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: this predictor is created with predict_tags = false");
    if (n_sentences == 0 || p->n_tags == 0) return VPT_OK;   // predictor.rs:553-555
    if (!utf8 || !byte_offsets || !out_offsets || !tags_out) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: NULL argument");
    if (out_offsets[n_sentences] != out_offsets[0] && !labels) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: labels: must not be NULL");
    VPT_HIP(hipSetDevice(p->device));
    Workspace w;
    vpt_status st = acquire(p, &w);
    if (st != VPT_OK) return st;
    vpt_batch* b = w.b;
    uint64_t total_b = 0;
    if ((st = stage(b, utf8, byte_offsets, out_offsets, n_sentences, labels, &total_b, nullptr, nullptr)) != VPT_OK) return st;
    const size_t n_rows = size_t(total_b + n_sentences), n_tag_words = n_rows * p->n_tags, n_score_words = n_rows * p->max_tag_scores;
    if ((st = grow(&b->d_tags, &b->tags_cap, n_tag_words + 16)) != VPT_OK) return st;
    if (tag_scores_out && (st = grow(&b->d_tag_scores, &b->tag_scores_cap, n_score_words + 16)) != VPT_OK) return st;
    if (tag_models_out && (st = grow(&b->d_tag_models, &b->tag_models_cap, n_rows + 16)) != VPT_OK) return st;
    // rows that end no token with a tag model are not written by the kernel: they read 0 (the reference holds None there)
    if (tag_scores_out && n_score_words) VPT_HIP(hipMemsetAsync(b->d_tag_scores, 0, n_score_words * sizeof(int32_t), b->own_stream));
    b->flags = flags;
    st = vpt_fill_tags_scores_batch_device(p, b, b->d_text, b->d_boff, b->d_ooff, n_sentences, total_b, b->d_labels, b->d_tags,
                                           tag_scores_out ? b->d_tag_scores : nullptr, tag_models_out ? b->d_tag_models : nullptr, b->own_stream);
    if (st != VPT_OK) return st;
    if ((st = vpt_batch_sync(b)) != VPT_OK) return st;
    const size_t row0 = size_t(out_offsets[0]);   // (the caller's rows start at out_offsets[0] + 0)
    VPT_HIP(hipMemcpy(tags_out + row0 * p->n_tags, b->d_tags, n_tag_words * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (tag_scores_out && n_score_words) VPT_HIP(hipMemcpy(tag_scores_out + row0 * p->max_tag_scores, b->d_tag_scores, n_score_words * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (tag_models_out) VPT_HIP(hipMemcpy(tag_models_out + row0, b->d_tag_models, n_rows * sizeof(int32_t), hipMemcpyDeviceToHost));
    return VPT_OK;
}

vpt_status vpt_fill_tags_batch_device(const vpt_predictor* p, vpt_batch* b, const uint8_t* d_utf8, const uint64_t* d_byte_offsets,
                                      const uint64_t* d_out_offsets, size_t n_sentences, uint64_t total_boundaries,
                                      const uint8_t* d_labels, int32_t* d_tags_out, void* hip_stream) {
    return vpt_fill_tags_scores_batch_device(p, b, d_utf8, d_byte_offsets, d_out_offsets, n_sentences, total_boundaries, d_labels, d_tags_out,
                                             nullptr, nullptr, hip_stream);
}

vpt_status vpt_fill_tags_scores_batch_device(const vpt_predictor* p, vpt_batch* b, const uint8_t* d_utf8, const uint64_t* d_byte_offsets,
                                             const uint64_t* d_out_offsets, size_t n_sentences, uint64_t total_boundaries,
                                             const uint8_t* d_labels, int32_t* d_tags_out, int32_t* d_tag_scores_out,
                                             int32_t* d_tag_models_out, void* hip_stream) {
    if (!p || !b || b->pred != p) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: batch: does not belong to this predictor");
    if (!p->predict_tags) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: this predictor is created with predict_tags = false");
    if (n_sentences == 0 || p->n_tags == 0) return VPT_OK;
    if (!d_utf8 || !d_byte_offsets || !d_out_offsets || (total_boundaries && !d_labels))
        return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: NULL device pointer");
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    VPT_HIP(hipSetDevice(p->device));
    const uint64_t total_c = total_boundaries + n_sentences;
    // record numbers and queue places are 32-bit (one per char at most)
    if (total_c >= 0xFFFFFF00ull) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: batch: fill_tags takes fewer than 2^32 - 256 chars per call");
    vpt_status st = grow(&b->d_cps, &b->cps_cap, size_t(total_c) + 16);
    if (st != VPT_OK) return st;
    b->tag_chars = 0;   // (until the launches are enqueued: a failure below leaves no records behind)
    // What the call leaves is ONE RECORD PER TOKEN THAT HAS A TAG MODEL (kernels.hpp, TagParams): the reference holds None for every other
    // char (predictor.rs:558-573).  Everything is sized for the worst case -- a tagged token per char, which a real tag model comes close
    // to (most tokens of real text have one; the synthetic M3's one token in thirty-five is the other end) -- so nothing can overflow and
    // there is no second path: records 16 + 12 n_tags bytes, the candidates between the launches 16.
    const uint32_t run_sent = vpt::tag_run_sentences(n_sentences, total_c);
    const uint64_t n_runs = (uint64_t(n_sentences) + run_sent - 1) / run_sent;
    const size_t n_state = vpt::scan_part_entries(n_runs), ctl_words = n_state + size_t(n_runs) + 2;
    if ((st = grow(&b->d_tag_records, &b->tag_records_cap, size_t(total_c) + 16)) != VPT_OK) return st;
    if ((st = grow(&b->d_rec_tags, &b->rec_tags_cap, size_t(total_c) * p->n_tags + 16)) != VPT_OK) return st;
    if ((st = grow(&b->d_rec_str, &b->rec_str_cap, size_t(total_c) * p->n_tags + 16)) != VPT_OK) return st;
    if ((st = grow(&b->d_tag_cands, &b->tag_cands_cap, size_t(total_c) + 16)) != VPT_OK) return st;
    if ((st = grow(&b->d_tag_ctl, &b->tag_ctl_cap, ctl_words)) != VPT_OK) return st;
    if (!b->d_tag_summary) VPT_HIP(hipMalloc(reinterpret_cast<void**>(&b->d_tag_summary), vpt::tag_summary_words() * sizeof(uint32_t)));
    VPT_HIP(hipMemsetAsync(b->d_tag_ctl, 0, ctl_words * sizeof(uint64_t), stream));   // the scan's state, the runs' counts
    // the dense arrays of the C ABI, for the callers that want them: None everywhere (what `resize(n_tags * len, None)` leaves, predictor.rs:556-557);
    // the passes write the entries of the tokens that have a model
    if (d_tags_out) VPT_HIP(hipMemsetAsync(d_tags_out, 0xFF, size_t(total_c) * p->n_tags * sizeof(int32_t), stream));
    if (d_tag_models_out) VPT_HIP(hipMemsetAsync(d_tag_models_out, 0xFF, size_t(total_c) * sizeof(int32_t), stream));
    const uint32_t* cinfo = p->d_cinfo + ((b->flags & VPT_FLAG_KYTEA_FULLWIDTH) ? 65536 : 0);
    const bool have_cps = b->cps_text == d_utf8 && b->cps_ooff == d_out_offsets && b->cps_sentences == n_sentences &&
                          b->cps_boundaries == total_boundaries && b->cps_flags == (b->flags & VPT_FLAG_KYTEA_FULLWIDTH) && b->last_stream == stream;
    b->cps_text = nullptr;   // one shot: only the fill_tags call that FOLLOWS the predict call takes its chars (predictor.rs:542)
    if (!have_cps) {
        VPT_HIP(vpt::launch_decode_chars(d_utf8, d_byte_offsets, d_out_offsets, n_sentences, total_c, cinfo, b->d_cps, nullptr, b->d_ctrl, stream));
    }
    vpt::TagParams T{};
    T.tok_tab = p->dtag.tok_tab; T.models = p->dtag.models; T.mfilt = p->dtag.mfilt; T.ngrams = p->dtag.ngrams; T.nrec = p->dtag.nrec; T.syms = p->dtag.syms; T.slots = p->dtag.slots;
    T.weights = p->dtag.weights; T.cinfo = cinfo; T.tok_bits = p->tok_bits; T.n_tags = p->n_tags;
    T.use_char = p->tag_use_char ? 1u : 0u; T.use_type = p->tag_use_type ? 1u : 0u;
    T.cps = b->d_cps; T.ooff = d_out_offsets; T.labels = d_labels; T.n_sent = n_sentences; T.total_chars = total_c; T.tags = d_tags_out;
    T.slot_str = p->dtag.slot_str; T.str_off = p->dtag.str_off; T.n_strings = p->dtag.n_strings;
    T.scores_out = p->max_tag_scores ? d_tag_scores_out : nullptr; T.model_out = d_tag_models_out; T.score_stride = p->max_tag_scores;
    T.n_cus = p->n_cus;
    T.records = b->d_tag_records; T.rec_tags = b->d_rec_tags; T.rec_str = b->d_rec_str; T.cands = b->d_tag_cands;
    T.scan_state = b->d_tag_ctl; T.run_pref = b->d_tag_ctl + n_state;
    T.n_runs = n_runs; T.run_sent = run_sent;
    T.summary = b->d_tag_summary;
    VPT_HIP(vpt::launch_tag_tokens(T, stream));
    b->d_run_pref = T.run_pref; b->tag_chars = total_c; b->tag_sentences = n_sentences; b->tag_runs = n_runs; b->tag_run_sent = run_sent;
    b->last_stream = stream; b->pending = true;
    return VPT_OK;
}

vpt_status vpt_expand_tags_batch_device(const vpt_predictor* p, vpt_batch* b, size_t n_sentences, uint64_t total_boundaries, int32_t* d_tags_out, void* hip_stream) {
    if (!p || !b || b->pred != p) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: batch: does not belong to this predictor");
    if (!p->predict_tags) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: this predictor is created with predict_tags = false");
    if (n_sentences == 0 || p->n_tags == 0) return VPT_OK;
    if (!d_tags_out) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: NULL device pointer");
    if (b->tag_chars != total_boundaries + n_sentences || b->tag_sentences != n_sentences || !b->d_tag_records)
        return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: batch: call vpt_fill_tags_batch_device on this workspace for this batch first");
    VPT_HIP(hipSetDevice(p->device));
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    VPT_HIP(vpt::launch_expand_tags(b->d_tag_records, b->d_rec_tags, b->d_run_pref + b->tag_runs, p->n_tags, b->tag_chars, d_tags_out, p->n_cus, stream));
    b->last_stream = stream; b->pending = true;
    return VPT_OK;
}

namespace {
vpt_status emit_device(const vpt_predictor* p, vpt_batch* b, const uint8_t* d_utf8, const uint64_t* d_byte_offsets,
                       const uint64_t* d_out_offsets, size_t n_sentences, uint64_t total_boundaries, const uint8_t* d_labels,
                       bool tagged, uint8_t* d_text_out, uint64_t text_capacity, uint64_t* d_text_offsets_out,
                       hipStream_t stream, uint64_t* total_out, const uint64_t* chain_in, uint64_t* chain_out) {
    if (!p || !b || b->pred != p) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: batch: does not belong to this predictor");
    if (!d_text_offsets_out) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: NULL device pointer");
    VPT_HIP(hipSetDevice(p->device));
    if (n_sentences == 0) {
        VPT_HIP(hipMemsetAsync(d_text_offsets_out, 0, sizeof(uint64_t), stream));
        b->last_stream = stream; b->pending = true; b->cps_text = nullptr;
        return VPT_OK;
    }
    if (!d_utf8 || !d_byte_offsets || !d_out_offsets || (total_boundaries && !d_labels) || (text_capacity && !d_text_out))
        return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: NULL device pointer");
    vpt::EmitParams E{};
    E.text = d_utf8; E.boff = d_byte_offsets; E.ooff = d_out_offsets; E.labels = d_labels; E.n_sent = n_sentences;
    E.total_boundaries = total_boundaries; E.out_text = d_text_out; E.out_offsets = d_text_offsets_out; E.capacity = text_capacity;
    E.status = b->d_ctrl;
    if (tagged && p->n_tags > 0) {   // "/tag" suffixes: from the records the fill_tags call on this workspace left for this batch
        if (!p->predict_tags) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: this predictor is created with predict_tags = false");
        if (b->tag_chars != total_boundaries + n_sentences || b->tag_sentences != n_sentences || !b->d_tag_records)
            return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: batch: call vpt_fill_tags_batch_device on this workspace for this batch first");
        E.records = b->d_tag_records; E.rec_str = b->d_rec_str; E.run_pref = b->d_run_pref; E.n_runs = b->tag_runs; E.run_sent = b->tag_run_sent;
        E.n_tags = p->n_tags; E.str_bytes = p->dtag.str_bytes;
    }
    // A WORKGROUP per run of sentences (emit_flat_kernel, round 5; a wave per block of 2 K chars before): 5 K chars when the batch is small (the
    // chip wants a thousand workgroups and more), up to 20 K on a big one -- fewer look-backs and size passes per byte (measured,
    // profiles/r05_h_*, r05_k_*: configs[1] 5 K 0.057 ms / 10 K 0.060 / 20 K 0.068; configs[2] 2.56 / 2.28 / 2.15; tagged configs[4] 2.39 / 2.10 /
    // 2.04); at most 256 sentences.  With tags: a whole multiple of fill_tags' runs, so that a workgroup's records are run_pref[a] .. run_pref[b].
    vpt::EmitFuse F{};
    {
        const uint64_t chars = total_boundaries + n_sentences;
        const uint64_t auto_run = std::min<uint64_t>(std::max<uint64_t>(chars / (uint64_t(16) * std::max<uint32_t>(p->n_cus, 64)), 5120), 20480);
        const uint64_t target = auto_run;
        uint64_t per = std::min<uint64_t>(std::max<uint64_t>((target * n_sentences + chars / 2) / chars, 1), vpt::kEmitFlatMaxBlock);   // round(target / mean chars per sentence)
        if (E.records && E.run_sent <= vpt::kEmitFlatMaxBlock)
            per = std::min<uint64_t>(std::max<uint64_t>((per + E.run_sent / 2) / E.run_sent, 1) * E.run_sent, (vpt::kEmitFlatMaxBlock / E.run_sent) * E.run_sent);
        F.per_block = uint32_t(per);
        if (b->knobs.emit_per_block) F.per_block = std::min<uint32_t>(b->knobs.emit_per_block, vpt::kEmitFlatMaxBlock);
        F.n_blocks = (n_sentences + F.per_block - 1) / F.per_block;
    }
    const size_t words = size_t(F.n_blocks) + 1;
    if (words > b->emit_state_cap) {
        const size_t cap = std::max(words + words / 2, size_t(4096));
        (void)hipFree(b->d_emit_state);   // (waits for the device)
        b->d_emit_state = nullptr; b->emit_state_cap = 0;
        VPT_HIP(hipMalloc(reinterpret_cast<void**>(&b->d_emit_state), 2 * cap * sizeof(uint64_t)));
        VPT_HIP(hipMemsetAsync(b->d_emit_state, 0, 2 * cap * sizeof(uint64_t), stream));   // (in front of the kernel on ITS stream: a plain hipMemset is not ordered with a non-blocking stream)
        b->emit_state_cap = cap; b->emit_dirty[0] = b->emit_dirty[1] = 0; b->emit_flip = 0;
    }
    F.state = b->d_emit_state + size_t(b->emit_flip) * b->emit_state_cap;
    F.clear = b->d_emit_state + size_t(b->emit_flip ^ 1) * b->emit_state_cap;
    F.clear_n = b->emit_dirty[b->emit_flip ^ 1];
    F.total_out = total_out; F.chain_in = chain_in; F.chain_out = chain_out;
    b->emit_dirty[b->emit_flip ^ 1] = 0; b->emit_dirty[b->emit_flip] = words;
    b->emit_flip ^= 1;
    VPT_HIP(vpt::launch_emit_tokenized(E, F, stream));
    b->last_stream = stream; b->pending = true; b->cps_text = nullptr;
    return VPT_OK;
}
}  // namespace

vpt_status vpt_write_tokenized_batch_device(const vpt_predictor* p, vpt_batch* b, const uint8_t* d_utf8, const uint64_t* d_byte_offsets,
                                            const uint64_t* d_out_offsets, size_t n_sentences, uint64_t total_boundaries,
                                            const uint8_t* d_labels, uint8_t* d_text_out, uint64_t text_capacity,
                                            uint64_t* d_text_offsets_out, void* hip_stream) {
    return emit_device(p, b, d_utf8, d_byte_offsets, d_out_offsets, n_sentences, total_boundaries, d_labels, false, d_text_out, text_capacity,
                       d_text_offsets_out, static_cast<hipStream_t>(hip_stream));
}

vpt_status vpt_write_tagged_batch_device(const vpt_predictor* p, vpt_batch* b, const uint8_t* d_utf8, const uint64_t* d_byte_offsets,
                                         const uint64_t* d_out_offsets, size_t n_sentences, uint64_t total_boundaries,
                                         const uint8_t* d_labels, const int32_t* d_tags, uint8_t* d_text_out, uint64_t text_capacity,
                                         uint64_t* d_text_offsets_out, void* hip_stream) {
    (void)d_tags;   // (until round 6: the dense array of fill_tags; the tags are the workspace's records of that call now -- NULL is fine)
    return emit_device(p, b, d_utf8, d_byte_offsets, d_out_offsets, n_sentences, total_boundaries, d_labels, true, d_text_out, text_capacity,
                       d_text_offsets_out, static_cast<hipStream_t>(hip_stream));
}

static vpt_status emit_host(const vpt_predictor* p, const uint8_t* utf8, const uint64_t* byte_offsets, size_t n_sentences,
                            const uint64_t* out_offsets, const uint8_t* labels, bool tagged, unsigned flags, uint8_t* text_out,
                            uint64_t text_capacity, uint64_t* text_offsets_out) {
    if (!p) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: predictor: must not be NULL");
    if (!text_offsets_out) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: NULL argument");
    if (flags & ~unsigned(VPT_FLAG_KYTEA_FULLWIDTH)) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: flags: unknown bit");
    if (tagged && !p->predict_tags) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: this predictor is created with predict_tags = false");
    text_offsets_out[0] = 0;
    if (n_sentences == 0) return VPT_OK;
    if (!utf8 || !byte_offsets || !out_offsets || (text_capacity && !text_out)) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: NULL argument");
    if (out_offsets[n_sentences] != out_offsets[0] && !labels) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: labels: must not be NULL");
    VPT_HIP(hipSetDevice(p->device));
    Workspace w;
    vpt_status st = acquire(p, &w);
    if (st != VPT_OK) return st;
    vpt_batch* b = w.b;
    uint64_t total_b = 0;
    if ((st = stage(b, utf8, byte_offsets, out_offsets, n_sentences, labels, &total_b, nullptr, nullptr)) != VPT_OK) return st;
    if ((st = grow(&b->d_tok, &b->tok_cap, size_t(text_capacity) + 16)) != VPT_OK) return st;
    if ((st = grow(&b->d_toff, &b->toff_cap, n_sentences + 1)) != VPT_OK) return st;
    const bool with_tags = tagged && p->n_tags > 0;
    if (with_tags) {   // Sentence::fill_tags, then the writer, as the CLI does (predict/src/main.rs:156-176); no dense array: the records are the tags
        b->flags = flags;
        st = vpt_fill_tags_batch_device(p, b, b->d_text, b->d_boff, b->d_ooff, n_sentences, total_b, b->d_labels, nullptr, b->own_stream);
        if (st != VPT_OK) return st;
    }
    st = emit_device(p, b, b->d_text, b->d_boff, b->d_ooff, n_sentences, total_b, b->d_labels, with_tags, b->d_tok,
                     text_capacity, b->d_toff, b->own_stream);
    if (st != VPT_OK) return st;
    if ((st = vpt_batch_sync(b)) != VPT_OK) return st;
    VPT_HIP(hipMemcpy(text_offsets_out, b->d_toff, 8 * (n_sentences + 1), hipMemcpyDeviceToHost));
    const uint64_t total = text_offsets_out[n_sentences];
    if (total) VPT_HIP(hipMemcpy(text_out, b->d_tok, size_t(total), hipMemcpyDeviceToHost));
    return VPT_OK;
}

vpt_status vpt_write_tokenized_batch(const vpt_predictor* p, const uint8_t* utf8, const uint64_t* byte_offsets, size_t n_sentences,
                                     const uint64_t* out_offsets, const uint8_t* labels, uint8_t* text_out, uint64_t text_capacity,
                                     uint64_t* text_offsets_out) {
    return emit_host(p, utf8, byte_offsets, n_sentences, out_offsets, labels, false, 0u, text_out, text_capacity, text_offsets_out);
}

vpt_status vpt_write_tagged_batch(const vpt_predictor* p, const uint8_t* utf8, const uint64_t* byte_offsets, size_t n_sentences,
                                  const uint64_t* out_offsets, const uint8_t* labels, unsigned flags, uint8_t* text_out,
                                  uint64_t text_capacity, uint64_t* text_offsets_out) {
    return emit_host(p, utf8, byte_offsets, n_sentences, out_offsets, labels, true, flags, text_out, text_capacity, text_offsets_out);
}

namespace {
vpt_status count_boundaries_impl(const vpt_predictor* p, vpt_batch* b, const uint8_t* d_utf8, const uint64_t* d_byte_offsets,
                                 size_t n_sentences, uint64_t* d_out_offsets, void* hip_stream, uint64_t text_bytes_hint) {
    if (!p || !b || b->pred != p) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: batch: does not belong to this predictor");
    if (!d_out_offsets || (n_sentences && (!d_utf8 || !d_byte_offsets))) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: NULL device pointer");
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    VPT_HIP(hipSetDevice(p->device));
    if (n_sentences == 0) VPT_HIP(hipMemsetAsync(d_out_offsets, 0, sizeof(uint64_t), stream));
    else {
        const vpt_status st = grow(&b->d_scan_part, &b->scan_part_cap, vpt::scan_part_entries(n_sentences));
        if (st != VPT_OK) return st;
        VPT_HIP(vpt::launch_count_boundaries(d_utf8, d_byte_offsets, n_sentences, d_out_offsets, b->d_scan_part, b->d_ctrl, nullptr /* nobody reads the longest sentence: no launch to clear it */, text_bytes_hint, stream));
    }
    b->last_stream = stream; b->pending = true; b->cps_text = nullptr;
    return VPT_OK;
}
}  // namespace

vpt_status vpt_count_boundaries_device(const vpt_predictor* p, vpt_batch* b, const uint8_t* d_utf8, const uint64_t* d_byte_offsets,
                                       size_t n_sentences, uint64_t* d_out_offsets, void* hip_stream) {
    return count_boundaries_impl(p, b, d_utf8, d_byte_offsets, n_sentences, d_out_offsets, hip_stream, 0);
}

namespace {
// vpt_tokenize_batch without tags, in chunks: a chunk of lines is a copy in (text, offsets), the char count, the tile search, the scoring
// launch (labels into the workspace) and the writer's launch.  Every chunk's text follows the one before it -- the writers hand the output
// position on through a chain of device words (EmitFuse::chain_in / chain_out) -- so the batch's output is one piece however it is cut.
// Four stages overlap:
//   copy in of chunk k + 2            its own stream
//   char count + tile search of k + 1 the PREPARING stream, with the scratch of one of two workspaces (k + 1 & 1)
//   scoring + writer of chunk k       the SCORING stream: the launches follow each other with an event wait that has long fired in between
//                                     (profiles/r04_e_tokenize_timeline.txt: 33 us of small launches in front of every scoring launch before)
//   copy out of chunk k - 1           issued by the host as soon as the chunk's event has fired; a pinned word says where its text ends
// and the host is a fifth: a chunk is some fifteen runtime calls, and with everything enqueued before the first wait the copies out only
// started when the LAST chunk was enqueued (r04_i_tokenize_timeline.txt).  So the loop cuts and rebases a chunk's offsets when its copy in
// is due, keeps the copies in two chunks ahead of the kernels, and after every chunk looks whether an earlier one can leave.
// Round 6: rounds 4 - 5 scored a chunk with the writer FUSED into the scoring kernel (one launch, 88 us of device time per 3.2 MB chunk); with
// the flat writer (round 5) the two launches take 28 + 12 us and the whole call 0.83 ms per 100 K lines against 0.88, 5.75 against 5.83 per
// million (profiles/r06_k_tokenize.jsonl) -- the fused phase, a second instance of every scoring kernel, is gone.
// Measured and dropped earlier (profiles/r04_{e,g,k,m,n}_tokenize*): a small first and last chunk (0.96 ms against 0.88); two independent lanes,
// each chunk's text placed by an upper bound and closed up by the copies out (1.07); kernels storing STRAIGHT into a pinned caller buffer
// (their stores cross PCIe at 27 .. 34 GB/s against the copy engine's 56: 1.15 - 1.29); every kernel of every chunk on one stream (0.97).
// `text` = the batch's first byte; byte_offsets are the caller's (relative to byte_offsets[0]).
vpt_status tokenize_chunked(const vpt_predictor* p, vpt_batch* b, const uint8_t* text, const uint64_t* byte_offsets, size_t n_sentences, unsigned flags,
                          uint64_t max_bytes, uint8_t* text_out, uint64_t text_capacity, uint64_t* text_offsets_out) {
    (void)max_bytes;
    const uint64_t t0 = byte_offsets[0];
    const size_t nbytes = size_t(byte_offsets[n_sentences] - t0);
    // 1/6 of the batch, at least 2 MB (a scoring launch costs 50 us + 1.3 us per 1000 lines) and at most 8 (1 M lines: 5.8 ms in 8 MB chunks, 6.4 in 32)
    const uint64_t chunk_bytes = p->knobs.tokenize_chunk_bytes_set ? std::max<uint64_t>(p->knobs.tokenize_chunk_bytes, 1)
                                                                   : std::min<uint64_t>(uint64_t(8) << 20, std::max<uint64_t>(uint64_t(2) << 20, (uint64_t(nbytes) + 5) / 6));
    const size_t n_cuts = size_t((uint64_t(nbytes) + chunk_bytes - 1) / chunk_bytes);   // chunks, unless sentences longer than one swallow some
    const size_t max_chunks = std::min<size_t>(n_sentences, n_cuts) + 1;
    auto cut_end = [&](size_t k) -> uint64_t { return std::min<uint64_t>(uint64_t(nbytes), (k + 1) * chunk_bytes); };   // where chunk k of n_cuts should end
    vpt_status st;
    Workspace second;   // the scratch (tiles, partial sums, the writer's words, labels, status) of every other chunk
    if ((st = acquire(p, &second)) != VPT_OK) return st;
    // Kernels enqueued on THIS workspace's streams look back over the second one's scratch (its writer words, its status): on any return --
    // an error one in the middle of the chunks included -- those streams are drained BEFORE `second` goes back to the pool (a guard declared
    // behind it is destroyed in front of it), or another host thread could take and clear what a chunk in flight still reads
    struct DrainFirst {
        vpt_batch* b;
        ~DrainFirst() {
            (void)hipStreamSynchronize(b->own_stream);
            if (b->s_tok_in) (void)hipStreamSynchronize(b->s_tok_in);
            if (b->s_tok_out) (void)hipStreamSynchronize(b->s_tok_out);
        }
    } drain_first{b};
    vpt_batch* const ws[2] = {b, second.b};
    if (!b->s_tok_in) VPT_HIP(hipStreamCreateWithFlags(&b->s_tok_in, hipStreamNonBlocking));
    if (!b->s_tok_out) VPT_HIP(hipStreamCreateWithFlags(&b->s_tok_out, hipStreamNonBlocking));
    while (b->chunk_ev.size() < 3 * max_chunks) {   // per chunk: copied in, tiles found, scored
        hipEvent_t e;
        VPT_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        b->chunk_ev.push_back(e);
    }
    if ((st = grow(&b->d_text, &b->text_cap, nbytes + 32)) != VPT_OK) return st;
    {
        size_t cap = b->off_cap;
        if ((st = grow(&b->d_boff, &cap, n_sentences + max_chunks + 1)) != VPT_OK) return st;
        size_t cap2 = b->off_cap;
        if ((st = grow(&b->d_ooff, &cap2, n_sentences + max_chunks + 1)) != VPT_OK) return st;
        b->off_cap = std::min(cap, cap2);
    }
    if ((st = grow(&b->d_chain, &b->chain_cap, max_chunks + 2)) != VPT_OK) return st;
    const uint64_t out_cap = uint64_t(nbytes) * 3 + 16;
    if ((st = grow(&b->d_tok, &b->tok_cap, size_t(out_cap) + 16)) != VPT_OK) return st;
    if ((st = grow(&b->d_toff, &b->toff_cap, n_sentences + 2)) != VPT_OK) return st;
    uint8_t* const d_out = b->d_tok;
    uint64_t* const d_off_out = b->d_toff;
    const size_t need_off = n_sentences + 1 + max_chunks + 1 + 2;   // pinned: the offsets relative to the batch's text, where every chunk's text ends, the workspaces' status words
    if (need_off > b->h_off_cap) {
        if (b->h_off) (void)hipHostFree(b->h_off);
        b->h_off = nullptr; b->h_off_cap = 0;
        VPT_HIP(hipHostMalloc(reinterpret_cast<void**>(&b->h_off), (need_off + need_off / 2) * sizeof(uint64_t), hipHostMallocDefault));
        b->h_off_cap = need_off + need_off / 2;
    }
    uint64_t* const h_boff = b->h_off;
    uint64_t* const h_end = b->h_off + n_sentences + 1;
    uint64_t* const h_ctrl = h_end + max_chunks + 1;
    hipStream_t s = b->own_stream, s_in = b->s_tok_in, s_out = b->s_tok_out;
    hipStream_t s_prep = second.b->own_stream;
    for (vpt_batch* w : ws) {
        w->flags = flags;
        w->max_chars = 0;   // unknown on the host (no round trip for it): the scoring kernel takes the geometry that fits any sentence
    }
    VPT_HIP(hipMemsetAsync(b->d_chain, 0, sizeof(uint64_t), s));
    struct Chunk { size_t a, n; uint64_t tb, nby, mb; };
    std::vector<Chunk> chunks;
    chunks.reserve(max_chunks);
    size_t cut_at = 0, cut_k = 0;
    h_boff[0] = 0;
    auto copy_in_next = [&]() -> vpt_status {   // the next chunk of lines: offsets rebased into pinned memory, text and offsets on their way
        if (cut_at >= n_sentences) return VPT_OK;
        const size_t a = cut_at, k = chunks.size();
        uint64_t want = cut_end(cut_k++);
        while (want <= h_boff[a] && cut_k < n_cuts) want = cut_end(cut_k++);   // (a sentence longer than a chunk took these)
        size_t i = a;
        uint64_t mb = 0;
        do {
            h_boff[i + 1] = byte_offsets[i + 1] - t0;
            mb = std::max<uint64_t>(mb, h_boff[i + 1] - h_boff[i]);
            ++i;
        } while (i < n_sentences && (h_boff[i] < want || chunks.size() + 1 >= max_chunks));
        const size_t n = i - a;
        const uint64_t tb = h_boff[a], nby = h_boff[i] - tb;
        VPT_HIP(hipMemcpyAsync(b->d_text + tb, text + tb, size_t(nby), hipMemcpyHostToDevice, s_in));
        VPT_HIP(hipMemcpyAsync(b->d_boff + a + k, h_boff + a, 8 * (n + 1), hipMemcpyHostToDevice, s_in));   // n + 1 entries per chunk; offsets into the WHOLE text: no rebasing
        VPT_HIP(hipEventRecord(b->chunk_ev[3 * k], s_in));
        chunks.push_back({a, n, tb, nby, mb});
        cut_at = i;
        return VPT_OK;
    };
    uint64_t at = 0;
    size_t next_out = 0;
    bool out_of_range = false, too_small = false;
    auto copy_out = [&](size_t k) -> vpt_status {   // chunk k has been scored: its text and offsets leave
        const uint64_t end = h_end[k];
        if (end > out_cap || end < at) { out_of_range = true; return VPT_OK; }   // the device found the inputs inconsistent and says so below
        if (end > text_capacity) { too_small = true; return VPT_OK; }
        if (end > at) VPT_HIP(hipMemcpyAsync(text_out + at, d_out + at, size_t(end - at), hipMemcpyDeviceToHost, s_out));
        VPT_HIP(hipMemcpyAsync(text_offsets_out + chunks[k].a, d_off_out + chunks[k].a, 8 * (chunks[k].n + 1), hipMemcpyDeviceToHost, s_out));
        at = end;
        return VPT_OK;
    };
    if ((st = copy_in_next()) != VPT_OK || (st = copy_in_next()) != VPT_OK) return st;
    for (size_t k = 0; k < chunks.size(); ++k) {
        const Chunk c = chunks[k];
        vpt_batch* const w = ws[k & 1];
        uint64_t* d_boff_k = b->d_boff + c.a + k;
        uint64_t* d_ooff_k = b->d_ooff + c.a + k;   // n + 1 entries per chunk, chunk-relative
        VPT_HIP(hipStreamWaitEvent(s_prep, b->chunk_ev[3 * k], 0));
        if (k >= 2) VPT_HIP(hipStreamWaitEvent(s_prep, b->chunk_ev[3 * (k - 2) + 2], 0));   // this workspace's scratch: the chunk before the last is through with it
        if ((st = count_boundaries_impl(p, w, b->d_text, d_boff_k, c.n, d_ooff_k, s_prep, c.nby)) != VPT_OK) return st;
        h_end[k] = ~uint64_t(0);
        if ((st = grow(&w->d_tlab, &w->tlab_cap, size_t(c.nby) + 16)) != VPT_OK) return st;
        w->split_stream = s; w->split_event = b->chunk_ev[3 * k + 1];   // the scoring launch: on the scoring stream, behind the tile search
        st = predict_device_impl(p, w, b->d_text, d_boff_k, d_ooff_k, c.n, c.nby - c.n /* boundaries of the chunk, at most */, c.mb, nullptr, w->d_tlab, s_prep);
        w->split_stream = nullptr; w->split_event = nullptr;
        if (st != VPT_OK) return st;
        st = emit_device(p, w, b->d_text, d_boff_k, d_ooff_k, c.n, c.nby - c.n, w->d_tlab, false, d_out, out_cap, d_off_out + c.a, s, h_end + k, b->d_chain + k, b->d_chain + k + 1);
        if (st != VPT_OK) return st;
        VPT_HIP(hipEventRecord(b->chunk_ev[3 * k + 2], s));
        if ((st = copy_in_next()) != VPT_OK) return st;
        while (!out_of_range && !too_small && next_out < k && hipEventQuery(b->chunk_ev[3 * next_out + 2]) == hipSuccess)
            if ((st = copy_out(next_out++)) != VPT_OK) return st;
    }
    // ---- collect what is still on the device
    for (; next_out < chunks.size() && !out_of_range && !too_small; ++next_out) {
        VPT_HIP(hipEventSynchronize(b->chunk_ev[3 * next_out + 2]));
        if ((st = copy_out(next_out)) != VPT_OK) return st;
    }
    // ---- the device's verdict over every chunk (into pinned words: a copy to the stack is staged and waited for, 25 us each)
    h_ctrl[0] = h_ctrl[1] = 0;
    VPT_HIP(hipStreamSynchronize(s_prep));
    VPT_HIP(hipMemcpyAsync(h_ctrl, b->d_ctrl, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    VPT_HIP(hipMemcpyAsync(h_ctrl + 1, second.b->d_ctrl, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    VPT_HIP(hipStreamSynchronize(s));
    VPT_HIP(hipStreamSynchronize(s_out));
    for (vpt_batch* w : ws) w->pending = false;
    const uint32_t bits = uint32_t(h_ctrl[0]) | uint32_t(h_ctrl[1]);
    if (bits) {
        for (vpt_batch* w : ws) VPT_HIP(hipMemset(w->d_ctrl, 0, sizeof(uint32_t)));
        return status_from_bits(bits);
    }
    if (too_small) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: text_capacity: smaller than the tokenized text");
    if (out_of_range || h_end[chunks.size() - 1] > out_cap) return fail(VPT_RUNTIME_ERROR, "vpt_tokenize_batch: the output size is out of range");
    return VPT_OK;
}
}  // namespace

// Lines in, tokenized lines out: Sentence::from_raw -> [KyteaFullwidthFilter] -> Predictor::predict -> [post-filters]
// -> [fill_tags] -> write_tokenized_text for a whole batch (the loop of predict/src/main.rs:122-176), with only the
// text crossing PCIe: char counting, scoring, tagging and the writer all run on the device.
vpt_status vpt_tokenize_batch(const vpt_predictor* p, const uint8_t* utf8, const uint64_t* byte_offsets, size_t n_sentences, unsigned flags,
                              int tagged, uint8_t* text_out, uint64_t text_capacity, uint64_t* text_offsets_out) {
    if (!p) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: predictor: must not be NULL");
    if (!text_offsets_out) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: NULL argument");
    if (flags & ~unsigned(VPT_FLAG_ALL)) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: flags: unknown bit");
    if (tagged && !p->predict_tags) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: this predictor is created with predict_tags = false");
    text_offsets_out[0] = 0;
    if (n_sentences == 0) return VPT_OK;
    if (!utf8 || !byte_offsets || (text_capacity && !text_out)) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: NULL argument");
    const uint64_t t0 = byte_offsets[0], t1 = byte_offsets[n_sentences];
    if (t1 < t0) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: byte_offsets: must be non-decreasing");
    uint64_t max_bytes = 0;
    for (size_t i = 0; i < n_sentences; ++i) {
        if (byte_offsets[i + 1] <= byte_offsets[i])
            return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: text: must contain at least one character");
        max_bytes = std::max<uint64_t>(max_bytes, byte_offsets[i + 1] - byte_offsets[i]);
    }
    VPT_HIP(hipSetDevice(p->device));
    Workspace w;
    vpt_status st = acquire(p, &w);
    if (st != VPT_OK) return st;
    vpt_batch* b = w.b;
    const size_t nbytes = size_t(t1 - t0);
    const bool with_tags = tagged && p->n_tags > 0;
    if (!with_tags) return tokenize_chunked(p, b, utf8 + t0, byte_offsets, n_sentences, flags, max_bytes, text_out, text_capacity, text_offsets_out);
    // NOTHING on the way needs a number from the device: the device buffers hold the whole batch and a slice of an output starts
    // where an upper bound puts it (a char is at least one byte: boundaries and chars in front of a slice <= text bytes in front
    // of it; tokenized text <= 3 bytes per text byte + the longest tag suffix per char), so copy in, char count, scoring, tagging
    // and writer are enqueued back to back; the host then waits for the event behind them, reads the size of the text from pinned
    // memory (the prefix sum's last kernel wrote it there) and copies text and offsets to where they belong.  The code can cut the
    // batch into chunks that alternate over a few lanes (streams with a workspace each, as in predict_lanes) and collect them in
    // order -- but a chunk is fifteen runtime calls, and measured on MI355X (profiles/r02_j_tokenize.txt) that enqueueing costs
    // more than the overlap returns: 100 K sentences 1.55 ms as one chunk, 1.73 .. 2.4 in 7 .. 25; a million 11.0 ms as one, 10.6 ..
    // 22.7 in 4 .. 125.  So a chunk is 256 MB of text -- one, unless the batch is larger than that (VPT_TOKENIZE_CHUNK_BYTES
    // overrides; the tests use it).
    constexpr int kMaxLanes = 4;
    const uint64_t chunk_bytes = p->knobs.tokenize_chunk_bytes;
    const uint64_t per_byte = 3 + (with_tags ? uint64_t(p->max_tag_suffix) : 0);   // tokenized bytes per text byte, at most
    const size_t max_chunks = std::min<size_t>(n_sentences, size_t(nbytes / chunk_bytes) + 2);
    const int n_lanes = int(std::min<size_t>(kMaxLanes, max_chunks));
    Workspace extra[kMaxLanes - 1];
    vpt_batch* lane[kMaxLanes] = {b};
    for (int l = 1; l < n_lanes; ++l) {
        if ((st = acquire(p, &extra[l - 1])) != VPT_OK) return st;
        lane[l] = extra[l - 1].b;
    }
    if (!b->s_out) VPT_HIP(hipStreamCreateWithFlags(&b->s_out, hipStreamNonBlocking));
    while (b->chunk_ev.size() < max_chunks) {
        hipEvent_t e;
        VPT_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        b->chunk_ev.push_back(e);
    }
    // whole-batch device buffers (lane 0's workspace owns them); the other lanes only lend their scratch
    if ((st = grow(&b->d_text, &b->text_cap, nbytes + 32)) != VPT_OK) return st;
    {
        size_t cap = b->off_cap;
        if ((st = grow(&b->d_boff, &cap, n_sentences + max_chunks + 1)) != VPT_OK) return st;
        size_t cap2 = b->off_cap;
        if ((st = grow(&b->d_ooff, &cap2, n_sentences + max_chunks + 1)) != VPT_OK) return st;
        b->off_cap = std::min(cap, cap2);
    }
    if ((st = grow(&b->d_tlab, &b->tlab_cap, nbytes + 1)) != VPT_OK) return st;
    if ((st = grow(&b->d_tok, &b->tok_cap, size_t(nbytes * per_byte) + 16)) != VPT_OK) return st;
    if ((st = grow(&b->d_toff, &b->toff_cap, n_sentences + max_chunks + 1)) != VPT_OK) return st;
    const size_t need_off = n_sentences + 1 + max_chunks;   // pinned: the offsets relative to the batch's text, then one total per chunk
    if (need_off > b->h_off_cap) {
        if (b->h_off) (void)hipHostFree(b->h_off);
        b->h_off = nullptr; b->h_off_cap = 0;
        VPT_HIP(hipHostMalloc(reinterpret_cast<void**>(&b->h_off), (need_off + need_off / 2) * sizeof(uint64_t), hipHostMallocDefault));
        b->h_off_cap = need_off + need_off / 2;
    }
    uint64_t* const h_boff = b->h_off;
    uint64_t* const h_total = b->h_off + n_sentences + 1;
    for (size_t i = 0; i <= n_sentences; ++i) h_boff[i] = byte_offsets[i] - t0;
    struct Chunk { size_t a, n; uint64_t tb; };   // first sentence, sentences, first text byte (relative to the batch's)
    std::vector<Chunk> chunks;
    for (size_t i = 0; i < n_sentences;) {
        const size_t a = i;
        uint64_t mb = 0;
        while (i < n_sentences && h_boff[i] - h_boff[a] < chunk_bytes) { mb = std::max<uint64_t>(mb, h_boff[i + 1] - h_boff[i]); ++i; }
        const size_t k = chunks.size(), n = i - a;
        const uint64_t tb = h_boff[a], nby = h_boff[i] - tb;
        vpt_batch* bb = lane[k % size_t(n_lanes)];
        hipStream_t s = bb->own_stream;
        bb->flags = flags;
        bb->max_chars = 0;   // unknown on the host (no round trip for it): the scoring kernel takes the geometry that fits any sentence
        uint64_t* d_boff_k = b->d_boff + a + k;                   // n + 1 entries per chunk; offsets into the WHOLE text: no rebasing
        uint64_t* d_ooff_k = b->d_ooff + a + k;                   // n + 1 entries per chunk, chunk-relative
        uint8_t* d_labels_k = b->d_tlab + tb;
        VPT_HIP(hipMemcpyAsync(b->d_text + tb, utf8 + t0 + tb, size_t(nby), hipMemcpyHostToDevice, s));
        VPT_HIP(hipMemcpyAsync(d_boff_k, h_boff + a, 8 * (n + 1), hipMemcpyHostToDevice, s));
        if ((st = count_boundaries_impl(p, bb, b->d_text, d_boff_k, n, d_ooff_k, s, nby)) != VPT_OK) return st;
        const uint64_t tb_bound = nby - n;                        // boundaries of the chunk, at most
        st = vpt_predict_batch_device(p, bb, b->d_text, d_boff_k, d_ooff_k, n, tb_bound, mb, nullptr, d_labels_k, s);
        if (st != VPT_OK) return st;
        if (with_tags) {   // (no dense array: the writer takes the records)
            bb->flags = flags & VPT_FLAG_KYTEA_FULLWIDTH;
            st = vpt_fill_tags_batch_device(p, bb, b->d_text, d_boff_k, d_ooff_k, n, tb_bound, d_labels_k, nullptr, s);
            if (st != VPT_OK) return st;
        }
        h_total[k] = 0;
        st = emit_device(p, bb, b->d_text, d_boff_k, d_ooff_k, n, tb_bound, d_labels_k, with_tags, b->d_tok + tb * per_byte, nby * per_byte,
                         b->d_toff + a + k, s, h_total + k);
        if (st != VPT_OK) return st;
        VPT_HIP(hipEventRecord(b->chunk_ev[k], s));
        chunks.push_back({a, n, tb});
    }
    // ---- collect: chunk by chunk, in order
    uint64_t at = 0;   // tokenized bytes in front of the chunk
    std::vector<uint64_t> base(chunks.size());
    bool incomplete = false;
    st = VPT_OK;
    for (size_t k = 0; k < chunks.size() && st == VPT_OK; ++k) {
        const Chunk& c = chunks[k];
        VPT_HIP(hipEventSynchronize(b->chunk_ev[k]));
        const uint64_t total = h_total[k];
        if (total > (h_boff[c.a + c.n] - c.tb) * per_byte) { incomplete = true; break; }   // the device found the inputs inconsistent and says so below
        if (total > text_capacity || at > text_capacity - total) { st = fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: text_capacity: smaller than the tokenized text"); break; }
        base[k] = at;
        if (total) VPT_HIP(hipMemcpyAsync(text_out + at, b->d_tok + c.tb * per_byte, size_t(total), hipMemcpyDeviceToHost, b->s_out));
        VPT_HIP(hipMemcpyAsync(text_offsets_out + c.a + 1, b->d_toff + c.a + k + 1, 8 * c.n, hipMemcpyDeviceToHost, b->s_out));
        at += total;
    }
    VPT_HIP(hipStreamSynchronize(b->s_out));
    // the device's verdict over every chunk (a lane's status word accumulates), fetched together
    uint32_t ctrl[kMaxLanes][2] = {};
    for (int l = 0; l < n_lanes; ++l)
        if (lane[l]->pending) VPT_HIP(hipMemcpyAsync(ctrl[l], lane[l]->d_ctrl, sizeof(ctrl[l]), hipMemcpyDeviceToHost, lane[l]->last_stream));
    for (int l = 0; l < n_lanes; ++l) {
        if (!lane[l]->pending) continue;
        VPT_HIP(hipStreamSynchronize(lane[l]->last_stream));
        lane[l]->pending = false;
        if (ctrl[l][0]) {
            VPT_HIP(hipMemset(lane[l]->d_ctrl, 0, sizeof(uint32_t)));
            if (st == VPT_OK) st = status_from_bits(ctrl[l][0]);
        }
    }
    if (st != VPT_OK) return st;
    if (incomplete) return fail(VPT_RUNTIME_ERROR, "vpt_tokenize_batch: a chunk's output size is out of range");
    // the offsets came back relative to their chunk's text
    text_offsets_out[0] = 0;
    for (size_t k = 1; k < chunks.size(); ++k)
        for (size_t j = 1; j <= chunks[k].n; ++j) text_offsets_out[chunks[k].a + j] += base[k];
    return VPT_OK;
}

// ---- pinned host memory for callers that want the PCIe link at full rate
vpt_status vpt_host_alloc(size_t bytes, void** out) {
    if (!out) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: out: must not be NULL");
    *out = nullptr;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return fail(VPT_RUNTIME_ERROR, "no HIP device available (this library has no CPU fallback)");
    VPT_HIP(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return VPT_OK;
}
void vpt_host_free(void* ptr) { if (ptr) (void)hipHostFree(ptr); }

// Contiguous sentence ranges with about the same number of CHARACTERS each (what the scoring costs; a sentence of n chars
// has out_offsets[i+1] - out_offsets[i] + 1 of them): bounds[r] .. bounds[r+1] is shard r.
vpt_status vpt_shard_bounds(const uint64_t* out_offsets, size_t n_sentences, size_t n_shards, uint64_t* bounds) {
    if (!out_offsets || !bounds || n_shards == 0) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: NULL argument or n_shards = 0");
    const uint64_t base = out_offsets[0];
    auto chars_before = [&](size_t i) { return out_offsets[i] - base + i; };
    const uint64_t total = chars_before(n_sentences);
    bounds[0] = 0;
    for (size_t r = 1; r < n_shards; ++r) {
        const uint64_t target = (unsigned __int128)(total) * r / n_shards;
        size_t lo = size_t(bounds[r - 1]), hi = n_sentences;      // first sentence with chars_before >= target
        while (lo < hi) {
            const size_t mid = lo + (hi - lo) / 2;
            if (chars_before(mid) >= target) hi = mid; else lo = mid + 1;
        }
        bounds[r] = lo;
    }
    bounds[n_shards] = n_sentences;
    return VPT_OK;
}

// Predictor::predict over one batch on SEVERAL GPUs: shard r (vpt_shard_bounds) is scored by preds[r] -- normally
// vpt_predictor_clone_to_device copies of one predictor, one per GPU of the node -- on a host thread of its own through the
// pipelined host-buffer path, straight into its slice of the caller's outputs.  No exchange between the devices.
vpt_status vpt_predict_batch_sharded(const vpt_predictor* const* preds, size_t n_preds, const uint8_t* utf8, const uint64_t* byte_offsets,
                                     size_t n_sentences, int32_t* scores_out, uint8_t* labels_out, const uint64_t* out_offsets, unsigned flags) {
    if (!preds || n_preds == 0) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: preds: must name at least one predictor");
    for (size_t r = 0; r < n_preds; ++r)
        if (!preds[r]) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: predictor: must not be NULL");
    if (n_sentences == 0) return VPT_OK;
    if (!utf8 || !byte_offsets || !out_offsets) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: NULL argument");
    if (n_preds == 1) return vpt_predict_batch_flags(preds[0], utf8, byte_offsets, n_sentences, scores_out, labels_out, out_offsets, flags);
    std::vector<uint64_t> bounds(n_preds + 1);
    vpt_status st = vpt_shard_bounds(out_offsets, n_sentences, n_preds, bounds.data());
    if (st != VPT_OK) return st;
    std::vector<vpt_status> rc(n_preds, VPT_OK);
    std::vector<std::string> msg(n_preds);
    std::vector<std::thread> th;
    for (size_t r = 0; r < n_preds; ++r)
        th.emplace_back([&, r] {
            const size_t a = size_t(bounds[r]), n = size_t(bounds[r + 1] - bounds[r]);
            if (n == 0) return;
            // the offsets stay absolute: shard r reads utf8[byte_offsets[a] ..) and writes scores_out[out_offsets[a] ..)
            rc[r] = vpt_predict_batch_flags(preds[r], utf8, byte_offsets + a, n, scores_out, labels_out, out_offsets + a, flags);
            if (rc[r] != VPT_OK) msg[r] = g_last_error;   // thread-local: carried back to the caller's thread
        });
    for (std::thread& t : th) t.join();
    for (size_t r = 0; r < n_preds; ++r)
        if (rc[r] != VPT_OK) return fail(rc[r], msg[r]);
    return VPT_OK;
}

// Sentence::char_types for a batch (sentence.rs:1016; CharacterType::get_type, sentence.rs:50-67): one u8 per char,
// char c of sentence i at types_out[out_offsets[i] + i + c]; with VPT_FLAG_KYTEA_FULLWIDTH the types of the normalised text
// (what the CLI's Sentence holds, predict/src/main.rs:126-129).
vpt_status vpt_char_types_batch_device(const vpt_predictor* p, vpt_batch* b, const uint8_t* d_utf8, const uint64_t* d_byte_offsets,
                                       const uint64_t* d_out_offsets, size_t n_sentences, uint64_t total_boundaries, uint8_t* d_types_out,
                                       void* hip_stream) {
    if (!p || !b || b->pred != p) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: batch: does not belong to this predictor");
    if (n_sentences == 0) return VPT_OK;
    if (!d_utf8 || !d_byte_offsets || !d_out_offsets || !d_types_out) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: NULL device pointer");
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    VPT_HIP(hipSetDevice(p->device));
    const uint32_t* cinfo = p->d_cinfo + ((b->flags & VPT_FLAG_KYTEA_FULLWIDTH) ? 65536 : 0);
    VPT_HIP(vpt::launch_decode_chars(d_utf8, d_byte_offsets, d_out_offsets, n_sentences, total_boundaries + n_sentences, cinfo, nullptr, d_types_out,
                                     b->d_ctrl, stream));
    b->last_stream = stream; b->pending = true; b->cps_text = nullptr;
    return VPT_OK;
}

vpt_status vpt_char_types_batch(const vpt_predictor* p, const uint8_t* utf8, const uint64_t* byte_offsets, size_t n_sentences,
                                const uint64_t* out_offsets, unsigned flags, uint8_t* types_out) {
    if (flags & ~unsigned(VPT_FLAG_KYTEA_FULLWIDTH)) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: flags: unknown bit");
    if (!p) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: predictor: must not be NULL");
    if (n_sentences == 0) return VPT_OK;
    if (!utf8 || !byte_offsets || !out_offsets || !types_out) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: NULL argument");
    VPT_HIP(hipSetDevice(p->device));
    Workspace w;
    vpt_status st = acquire(p, &w);
    if (st != VPT_OK) return st;
    vpt_batch* b = w.b;
    uint64_t total_b = 0;
    if ((st = stage(b, utf8, byte_offsets, out_offsets, n_sentences, nullptr, &total_b, nullptr, nullptr)) != VPT_OK) return st;
    const size_t total_c = size_t(total_b) + n_sentences;
    if ((st = grow(&b->d_types, &b->types_cap, total_c + 16)) != VPT_OK) return st;
    b->flags = flags;
    if ((st = vpt_char_types_batch_device(p, b, b->d_text, b->d_boff, b->d_ooff, n_sentences, total_b, b->d_types, b->own_stream)) != VPT_OK) return st;
    if ((st = vpt_batch_sync(b)) != VPT_OK) return st;
    VPT_HIP(hipMemcpy(types_out + size_t(out_offsets[0]) + 0, b->d_types, total_c, hipMemcpyDeviceToHost));
    return VPT_OK;
}

vpt_status vpt_predict_one(const vpt_predictor* p, const uint8_t* utf8, size_t len, int32_t* scores, uint8_t* labels, size_t* n_boundaries) {
    uint64_t boff[2] = {0, uint64_t(len)}, ooff[2] = {0, 0};
    if (len == 0) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: text: must contain at least one character");
    vpt_status st = vpt_count_boundaries(utf8, boff, 1, ooff);
    if (st != VPT_OK) return st;
    if (n_boundaries) *n_boundaries = size_t(ooff[1]);
    return vpt_predict_batch(p, utf8, boff, 1, scores, labels, ooff);
}

}  // extern "C"
