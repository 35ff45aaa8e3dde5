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

struct DevicePacked {
    unsigned char* base = nullptr;
    uint32_t off[7] = {0, 0, 0, 0, 0, 0, 0};   // uni, bi, tri, deep, xrows, trow, cpid
    void release() { (void)hipFree(base); base = nullptr; }
};

struct DeviceTable {
    uint32_t *short_tab = nullptr, *uni = nullptr, *edges = nullptr;
    int32_t* wdata = nullptr;
    void release() {
        (void)hipFree(short_tab); (void)hipFree(uni); (void)hipFree(edges); (void)hipFree(wdata);
        short_tab = uni = edges = nullptr; wdata = nullptr;
    }
};

constexpr size_t kTimingRing = 256;     // timed launches remembered per vpt_batch
constexpr size_t kTablePadBytes = 256;  // probes read whole 16-byte chunks; keep the tail readable

template <typename T>
hipError_t upload(const std::vector<T>& v, T** out) {
    *out = nullptr;
    size_t bytes = v.size() * sizeof(T);
    hipError_t e = hipMalloc(reinterpret_cast<void**>(out), bytes + kTablePadBytes);
    if (e != hipSuccess) return e;
    e = hipMemset(*out, 0, bytes + kTablePadBytes);
    if (e != hipSuccess) return e;
    if (bytes) e = hipMemcpy(*out, v.data(), bytes, hipMemcpyHostToDevice);
    return e;
}

vpt::PatternTableView make_view(const vpt::HostPatternTable& h, const DeviceTable& d) {
    vpt::PatternTableView v{};
    v.present = h.present ? 1u : 0u;
    if (!h.present) return v;
    v.short_tab = d.short_tab; v.uni = d.uni; v.edges = d.edges; v.wdata = d.wdata;
    v.short_shift = 32 - (h.short_bits - 1); v.short_mask = (1u << (h.short_bits - 1)) - 1;
    v.edge_shift = 32 - (h.edge_bits - 2); v.edge_mask = (1u << (h.edge_bits - 2)) - 1;
    v.stride_dw = h.stride_dw; v.uni_dw = h.uni_dw; v.uni_n = h.uni_n; v.ext_slot = h.ext_slot;
    v.window = h.window;
    for (int i = 0; i < 3; ++i) { v.lo[i] = h.lo[i]; v.len[i] = h.len[i]; }
    v.has_long = h.has_long ? 1u : 0u;
    return v;
}

vpt::PackedView make_packed_view(const vpt::HostPackedTable& h, const DevicePacked& d) {
    vpt::PackedView v{};
    v.present = h.present ? 1u : 0u;
    if (!h.present) return v;
    v.base = d.base;
    v.off_uni = d.off[0]; v.off_bi = d.off[1]; v.off_tri = d.off[2]; v.off_deep = d.off[3];
    v.off_xrows = d.off[4]; v.off_trow = d.off[5]; v.off_cpid = d.off[6];
    v.n_uni = uint32_t(h.uni.size() / 4); v.n_tri = uint32_t(h.tri.size() / 4); v.bi_shift = h.bi_shift;
    v.has_trow = h.trow.empty() ? 0u : 1u;
    return v;
}

// all packed arrays in one allocation (PackedView); false if they do not fit 32-bit offsets
hipError_t upload_packed(const vpt::HostPackedTable& h, DevicePacked* d, bool* fits) {
    const void* src[7] = {h.uni.data(), h.bi.data(), h.tri.data(), h.deep.data(), h.xrows.data(), h.trow.data(), h.cpid.data()};
    const size_t bytes[7] = {4 * h.uni.size(), 4 * h.bi.size(), 4 * h.tri.size(), 4 * h.deep.size(), 4 * h.xrows.size(),
                             4 * h.trow.size(), 4 * h.cpid.size()};
    size_t total = 0;
    size_t off[7];
    for (int i = 0; i < 7; ++i) { off[i] = total; total += (bytes[i] + kTablePadBytes + 255) & ~size_t(255); }
    *fits = total < (size_t(1) << 32);
    if (!*fits) return hipSuccess;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d->base), total);
    if (e != hipSuccess) return e;
    e = hipMemset(d->base, 0, total);
    for (int i = 0; i < 7 && e == hipSuccess; ++i) {
        d->off[i] = uint32_t(off[i]);
        if (bytes[i]) e = hipMemcpy(d->base + off[i], src[i], bytes[i], hipMemcpyHostToDevice);
    }
    return e;
}

void fill_info(const vpt::CompiledModel& c, vpt_model_info* info) {
    std::memset(info, 0, sizeof(*info));
    info->n_char_ngrams = c.n_char_ngrams; info->n_type_ngrams = c.n_type_ngrams;
    info->n_dict_words = c.n_dict_words; info->n_tag_models = c.n_tag_models;
    info->bias = c.bias;
    info->char_window = c.chars.present ? uint32_t(c.chars.window) : 0;
    info->type_window = uint32_t(c.type_window);
    info->max_pattern_chars = c.chars.max_pattern;
    info->n_short_entries = c.chars.n_short; info->n_long_nodes = c.chars.n_long_nodes;
    info->type_kind = uint32_t(c.type_kind);
    info->packed = c.packed.present ? 1u : 0u;
    info->n_displaced = c.packed.present ? 0u : c.chars.n_displaced_short;
    info->type_rows = c.packed.present && !c.packed.trow.empty() ? 1u : 0u;
    info->n_overflow_children = 0u;
    // the specialised kernel reads only the packed tables; the general ones stay resident for oversized sentences
    info->device_table_bytes = (c.chars.present ? c.chars.bytes() : 0) + (c.types.present ? c.types.bytes() : 0) +
                               4ull * c.type_table.size() + (c.packed.present ? c.packed.bytes() : 0);
    info->hot_table_bytes = c.packed.present ? c.packed.bytes() + (c.packed.trow.empty() ? 4ull * c.type_table.size() : 0) : info->device_table_bytes;
}

}  // namespace

struct vpt_batch {
    const vpt_predictor* pred = nullptr;
    int device = 0;
    // per-call device tables
    uint32_t* d_tile_first = nullptr; size_t tile_cap = 0;
    uint32_t* d_slow_list = nullptr;
    uint32_t* d_ctrl = nullptr;        // [0] status bits, [1] slow tile count
    uint64_t* d_prof = nullptr;        // 8 per-phase cycle counters (only with VPT_PROFILE_PHASES set)
    unsigned char* d_scratch = nullptr; size_t scratch_bytes = 0;
    uint32_t* d_cps = nullptr; size_t cps_cap = 0;   // decoded scalar values for vpt_fill_tags_batch_device
    uint64_t max_chars = 0;            // caller's bound on chars per sentence (0 = unknown)
    unsigned flags = 0;                // VPT_FLAG_*
    // timing
    bool timing = false;
    std::vector<hipEvent_t> ev;        // ring of (start, stop) pairs around the scoring kernel
    size_t ev_calls = 0;               // timed calls since the last vpt_batch_kernel_ms
    uint32_t last_tiles = 0;
    hipStream_t last_stream = nullptr; bool pending = false;
    // staging for the host-buffer entry points
    hipStream_t own_stream = nullptr;
    uint8_t* d_text = nullptr; size_t text_cap = 0;
    uint64_t *d_boff = nullptr, *d_ooff = nullptr; size_t off_cap = 0;
    int32_t* d_scores = nullptr; uint8_t* d_labels = nullptr; size_t out_cap = 0;
    int32_t* d_tags = nullptr; size_t tags_cap = 0;                 // vpt_fill_tags_batch
    uint8_t* d_tok = nullptr; size_t tok_cap = 0;                   // vpt_write_tokenized_batch
    uint64_t* d_toff = nullptr; size_t toff_cap = 0;
    int32_t* d_tok_model = nullptr; size_t tok_model_cap = 0;      // tag model of every token, from the last fill_tags on this workspace
    uint64_t tok_model_chars = 0;                                   // ... which covered this many chars
    std::vector<uint64_t> h_boff, h_ooff;                           // rebased offsets of the call in flight (copied asynchronously)
};

struct DeviceTags {
    uint32_t *tok_tab = nullptr, *models = nullptr, *ngrams = nullptr, *syms = nullptr, *slots = nullptr, *slot_str = nullptr, *str_off = nullptr;
    uint8_t* str_bytes = nullptr;
    int32_t* weights = nullptr;
    uint32_t n_models = 0, n_strings = 0;
    void release() {
        (void)hipFree(tok_tab); (void)hipFree(models); (void)hipFree(ngrams); (void)hipFree(syms); (void)hipFree(slots); (void)hipFree(weights);
        (void)hipFree(slot_str); (void)hipFree(str_off); (void)hipFree(str_bytes);
        tok_tab = models = ngrams = syms = slots = slot_str = str_off = nullptr; weights = nullptr; str_bytes = nullptr;
    }
};

struct vpt_predictor {
    int device = 0;
    bool predict_tags = false;
    bool has_tags = false;
    uint32_t n_tags = 0, tok_bits = 0, max_tag_suffix = 0;
    bool tag_use_char = false, tag_use_type = false;
    DeviceTags dtag;
    vpt_model_info info{};
    int32_t bias = 0; int pad = 1; int type_kind = 0; int type_window = 0; int chunks = 2;
    uint32_t tile_slots = 0;           // workgroups of the scoring kernel the device runs at a time (0 = unknown)
    DeviceTable dc, dt;
    DevicePacked dp;
    vpt::PackedView pk{};
    int32_t* d_type_table = nullptr;
    uint8_t* d_ctype = nullptr;
    uint32_t* d_cinfo = nullptr;       // [0, 65536): plain; [65536, 131072): through KyteaFullwidthFilter
    uint32_t* d_cid = nullptr;         // the same two tables for the specialised kernel: id | type << 16 | linebreak << 19
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
    (void)hipFree(b->d_tile_first); (void)hipFree(b->d_slow_list); (void)hipFree(b->d_ctrl); (void)hipFree(b->d_scratch);
    (void)hipFree(b->d_prof); (void)hipFree(b->d_cps);
    (void)hipFree(b->d_text); (void)hipFree(b->d_boff); (void)hipFree(b->d_ooff); (void)hipFree(b->d_scores); (void)hipFree(b->d_labels);
    (void)hipFree(b->d_tags); (void)hipFree(b->d_tok); (void)hipFree(b->d_toff); (void)hipFree(b->d_tok_model);
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
        std::lock_guard<std::mutex> g(p->pool_mu);
        p->pool.push_back(b);
    }
};
vpt_status acquire(const vpt_predictor* p, Workspace* w) {
    w->p = p;
    {
        std::lock_guard<std::mutex> g(p->pool_mu);
        if (!p->pool.empty()) { w->b = p->pool.back(); p->pool.pop_back(); }
    }
    if (w->b) return VPT_OK;
    vpt_batch* b = nullptr;
    vpt_status st = vpt_batch_create(p, &b);
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

vpt_status vpt_model_inspect(const uint8_t* model_bytes, size_t len, int predict_tags, vpt_model_info* info) {
    if (!info) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: info: must not be NULL");
    vpt::CompiledModel c;
    vpt_status st = compile(model_bytes, len, predict_tags, &c);
    if (st != VPT_OK) return st;
    fill_info(c, info);
    return VPT_OK;
}

vpt_status vpt_predictor_create(const uint8_t* model_bytes, size_t len, int predict_tags, int device_id, vpt_predictor** out) {
    if (!out) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: out: must not be NULL");
    *out = nullptr;
    vpt::CompiledModel c;
    vpt_status st = compile(model_bytes, len, predict_tags, &c);
    if (st != VPT_OK) return st;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return fail(VPT_RUNTIME_ERROR, "no HIP device available (this library has no CPU fallback)");
    if (device_id < 0 || device_id >= n_dev) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: device_id: no such HIP device");
    VPT_HIP(hipSetDevice(device_id));
    vpt_predictor* p = new (std::nothrow) vpt_predictor();
    if (!p) return fail(VPT_RUNTIME_ERROR, "out of host memory");
    p->device = device_id;
    fill_info(c, &p->info);
    p->bias = c.bias; p->pad = c.pad; p->type_kind = c.type_kind; p->type_window = c.type_window;
    hipError_t e = hipSuccess;
    auto up = [&](const vpt::HostPatternTable& h, DeviceTable& d) {
        if (!h.present) return;
        if (e == hipSuccess) e = upload(h.short_tab, &d.short_tab);
        if (e == hipSuccess) e = upload(h.uni, &d.uni);
        if (e == hipSuccess) e = upload(h.edges, &d.edges);
        if (e == hipSuccess) e = upload(h.wdata, &d.wdata);
    };
    up(c.chars, p->dc);
    up(c.types, p->dt);
    p->predict_tags = predict_tags != 0;
    if (c.tags.present) {
        p->has_tags = true; p->n_tags = c.tags.n_tags; p->tok_bits = c.tags.tok_bits;
        p->tag_use_char = c.tags.use_char; p->tag_use_type = c.tags.use_type;
        if (e == hipSuccess) e = upload(c.tags.tok_tab, &p->dtag.tok_tab);
        if (e == hipSuccess) e = upload(c.tags.models, &p->dtag.models);
        if (e == hipSuccess) e = upload(c.tags.ngrams, &p->dtag.ngrams);
        if (e == hipSuccess) e = upload(c.tags.syms, &p->dtag.syms);
        if (e == hipSuccess) e = upload(c.tags.slots, &p->dtag.slots);
        if (e == hipSuccess) e = upload(c.tags.weights, &p->dtag.weights);
        if (e == hipSuccess) e = upload(c.tags.slot_str, &p->dtag.slot_str);
        if (e == hipSuccess) e = upload(c.tags.str_off, &p->dtag.str_off);
        if (e == hipSuccess) e = upload(c.tags.str_bytes, &p->dtag.str_bytes);
        p->dtag.n_models = c.tags.n_models; p->dtag.n_strings = uint32_t(c.tags.str_off.size() - 1);
        for (uint32_t mi = 0; mi < c.tags.n_models; ++mi) {   // the longest "/tag/tag.." a token can get
            const uint32_t* mr = &c.tags.models[size_t(mi) * 12];
            uint32_t worst = 0;
            for (uint32_t j = 0; j < mr[9]; ++j) {
                const uint32_t first = c.tags.slot_str[mr[8] + j], cnt = c.tags.slots[size_t(mr[8] + j) * 2];
                uint32_t longest = 0;
                for (uint32_t k = 0; k < cnt; ++k) longest = std::max(longest, c.tags.str_off[first + k + 1] - c.tags.str_off[first + k]);
                worst += 1 + longest;
            }
            p->max_tag_suffix = std::max(p->max_tag_suffix, worst);
        }
    }
    bool packed_ok = c.packed.present;
    if (c.packed.present && e == hipSuccess) e = upload_packed(c.packed, &p->dp, &packed_ok);
    if (e == hipSuccess && c.type_kind == vpt::kTypeWindowTable) e = upload(c.type_table, &p->d_type_table);
    if (e == hipSuccess) {
        std::vector<uint32_t> cinfo(2 * 65536);   // the char a BMP char is scored as | its CharacterType << 16
        for (uint32_t cp = 0; cp < 65536; ++cp) {
            cinfo[cp] = cp | (uint32_t(vpt::char_type_host(cp)) << 16);
            const uint32_t fw = vpt::kytea_fullwidth_host(cp);
            cinfo[65536 + cp] = fw | (uint32_t(vpt::char_type_host(fw)) << 16);
        }
        e = upload(cinfo, &p->d_cinfo);
        if (e == hipSuccess && c.packed.present) {
            std::vector<uint32_t> cid(2 * 65536);
            for (uint32_t cp = 0; cp < 65536; ++cp)
                for (int mode = 0; mode < 2; ++mode) {
                    const uint32_t scored = mode ? vpt::kytea_fullwidth_host(cp) : cp;   // 1:1 on the BMP
                    cid[size_t(mode) * 65536 + cp] = uint32_t(c.packed.id_of[scored]) | (uint32_t(vpt::char_type_host(scored)) << 16) |
                                                     ((scored == 0x0Au || scored == 0x0Du) ? vpt::kCinfoLinebreak : 0u);
                }
            e = upload(cid, &p->d_cid);
        }
        std::vector<uint8_t> ctype(65536);
        for (uint32_t cp = 0; cp < 65536; ++cp) ctype[cp] = vpt::char_type_host(cp);
        if (e == hipSuccess) e = upload(ctype, &p->d_ctype);
    }
    if (e != hipSuccess) {
        std::string msg = std::string("HIP error while uploading the tables: ") + hipGetErrorString(e);
        vpt_predictor_destroy(p);
        return fail(VPT_RUNTIME_ERROR, msg);
    }
    p->ct = make_view(c.chars, p->dc);
    p->tt = make_view(c.types, p->dt);
    p->pk = make_packed_view(c.packed, p->dp);
    if (!packed_ok) { p->pk.present = 0; p->info.packed = 0; p->info.type_rows = 0; }
    uint32_t stride = 4;
    if (c.chars.present) stride = std::max(stride, c.chars.stride_dw);
    if (c.types.present) stride = std::max(stride, c.types.stride_dw);
    p->chunks = int(std::max<uint32_t>(2, stride / 4));
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0) {
            vpt::ScoreParams probe{};
            probe.ct = p->ct; probe.pk = p->pk; probe.pad = p->pad; probe.ctype = p->d_ctype; probe.cid = p->d_cid; probe.type_kind = p->type_kind;
            probe.type_window = p->type_window;
            const bool fast = vpt::fast_path_supported(probe);
            size_t lds = fast ? vpt::score_tiles_fast_lds_bytes(probe) : vpt::score_tiles_lds_bytes();
            if (const char* padv = std::getenv("VPT_DEBUG_LDS_PAD")) lds += size_t(std::atoi(padv));   // occupancy experiments (kernels_fast.hip)
            const size_t granules = (lds + 1279) / 1280;   // gfx950 hands out its 160 KB of LDS in 1280-byte granules
            // the specialised kernel is built for 6 workgroups of 4 waves per CU (<= 80 VGPRs), the general one for 8
            const uint32_t per_cu = uint32_t(std::min<size_t>(fast ? 6 : 8, std::max<size_t>(1, 128 / std::max<size_t>(granules, 1))));
            p->tile_slots = uint32_t(prop.multiProcessorCount) * per_cu;
        }
    }
    *out = p;
    return VPT_OK;
}

void vpt_predictor_destroy(vpt_predictor* p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    for (vpt_batch* b : p->pool) batch_release(b);
    p->dc.release(); p->dt.release(); p->dp.release(); p->dtag.release();
    (void)hipFree(p->d_type_table);
    (void)hipFree(p->d_cinfo); (void)hipFree(p->d_ctype); (void)hipFree(p->d_cid);
    delete p;
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
    *out = nullptr;
    VPT_HIP(hipSetDevice(p->device));
    vpt_batch* b = new (std::nothrow) vpt_batch();
    if (!b) return fail(VPT_RUNTIME_ERROR, "out of host memory");
    b->pred = p; b->device = p->device;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&b->d_ctrl), 64);
    if (e == hipSuccess) e = hipMemset(b->d_ctrl, 0, 64);
    if (e == hipSuccess && std::getenv("VPT_PROFILE_PHASES")) {
        e = hipMalloc(reinterpret_cast<void**>(&b->d_prof), 64);
        if (e == hipSuccess) e = hipMemset(b->d_prof, 0, 64);
    }
    if (e != hipSuccess) { batch_release(b); return fail(VPT_RUNTIME_ERROR, std::string("HIP error: ") + hipGetErrorString(e)); }
    *out = b;
    return VPT_OK;
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

vpt_status vpt_predict_batch_device(const vpt_predictor* p, vpt_batch* b, const uint8_t* d_utf8, const uint64_t* d_byte_offsets,
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
    // Tiles are cut every `tile_flat` flat positions (chars + separators) and end with the sentence that crosses
    // the cut, so a tile holds < tile_flat + longest sentence: pick tile_flat to fill the kernel's LDS capacity.
    const bool fast = vpt::fast_path_supported(P) && !std::getenv("VPT_FORCE_GENERIC");
    const uint64_t cap = fast ? vpt::kFastCap : vpt::kCap;
    uint64_t tile_flat = cap / 2;
    // flat positions of the longest sentence: its chars, bounded by the caller's hint or else by its bytes
    const uint64_t max_chars = (b->max_chars && b->max_chars < max_sentence_bytes) ? b->max_chars : max_sentence_bytes;
    if (max_chars + 2 * uint64_t(p->pad) + cap / 2 <= cap) tile_flat = cap - 2 * uint64_t(p->pad) - max_chars;
    const uint64_t total_flat = total_boundaries + uint64_t(n_sentences) * uint64_t(1 + p->pad);
    // Whole rounds: the chip runs `slots` tiles at a time; cutting the batch into a multiple of that many tiles (by
    // shrinking the tiles a little) avoids a last round that leaves most CUs idle.
    if (p->tile_slots > 0) {
        const uint64_t n_min = (total_flat + tile_flat - 1) / tile_flat;
        const uint64_t rounds = (n_min + p->tile_slots - 1) / p->tile_slots;
        const uint64_t even = (total_flat + rounds * p->tile_slots - 1) / (rounds * p->tile_slots);
        if (even < tile_flat) tile_flat = std::max<uint64_t>(even, 256);
    }
    const uint64_t n_tiles64 = (total_flat + tile_flat - 1) / tile_flat;
    if (n_tiles64 >= 0x7FFFFFFFull) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: batch too large for one call");
    const uint32_t n_tiles = uint32_t(n_tiles64);
    if (size_t(n_tiles) + 1 > b->tile_cap || !b->d_tile_first) {
        (void)hipFree(b->d_slow_list); b->d_slow_list = nullptr;
        size_t cap = b->tile_cap;
        vpt_status st = grow(&b->d_tile_first, &cap, size_t(n_tiles) + 1);
        if (st != VPT_OK) return st;
        b->tile_cap = cap;
        VPT_HIP(hipMalloc(reinterpret_cast<void**>(&b->d_slow_list), cap * sizeof(uint32_t) + 64));
    }
    // long-sentence scratch (only when a sentence might not fit the LDS tile)
    const bool need_slow = max_chars + 2 * uint64_t(p->pad) + tile_flat > cap;
    uint32_t slow_blocks = 0, scratch_cap = 0;
    uint64_t slab = 0;
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
    P.text = d_utf8; P.boff = d_byte_offsets; P.ooff = d_out_offsets; P.tile_first = b->d_tile_first;
    P.scores = d_scores; P.labels = d_labels; P.status = b->d_ctrl; P.slow_list = b->d_slow_list; P.slow_count = b->d_ctrl + 1;
    P.scratch = b->d_scratch; P.scratch_stride = slab; P.scratch_cap = scratch_cap;
    P.prof = b->d_prof;
    if (const char* dbg = std::getenv("VPT_DEBUG_ABLATE")) { P.debug = uint32_t(std::atoi(dbg)); P.ct.debug = P.debug; P.tt.debug = P.debug; }

    P.n_sent = n_sentences; P.tile_flat = uint32_t(tile_flat); P.n_tiles = n_tiles;
    // The specialised kernel can find its tiles itself (one launch per step).  Measured on MI355X (profiles/r02_c1_ab.jsonl):
    // the search at the head of every workgroup costs the scoring kernel more (+4 us) than the separate 5 us kernel and its
    // launch gap cost the step, so the separate kernel stays the default.
    const bool inline_assign = fast && !need_slow && std::getenv("VPT_INLINE_ASSIGN");
    if (inline_assign) P.tile_first = nullptr;
    else VPT_HIP(vpt::launch_assign_tiles(d_out_offsets, n_sentences, p->pad, uint32_t(tile_flat), n_tiles, b->d_tile_first, b->d_ctrl, stream));
    const size_t slot = b->ev_calls % kTimingRing;
    if (b->timing) VPT_HIP(hipEventRecord(b->ev[2 * slot], stream));
    if (fast) VPT_HIP(vpt::launch_score_tiles_fast(P, n_tiles, stream));
    else VPT_HIP(vpt::launch_score_tiles(P, p->chunks, n_tiles, stream));
    if (b->timing) { VPT_HIP(hipEventRecord(b->ev[2 * slot + 1], stream)); ++b->ev_calls; }
    if (need_slow) VPT_HIP(vpt::launch_score_slow(P, p->chunks, slow_blocks, stream));
    b->last_tiles = n_tiles; b->last_stream = stream; b->pending = true;
    return VPT_OK;
}

vpt_status vpt_batch_sync(vpt_batch* b) {
    if (!b) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: NULL argument");
    if (!b->pending) return VPT_OK;
    VPT_HIP(hipSetDevice(b->device));
    uint32_t ctrl[2] = {0, 0};
    VPT_HIP(hipMemcpyAsync(ctrl, b->d_ctrl, sizeof(ctrl), hipMemcpyDeviceToHost, b->last_stream));
    VPT_HIP(hipStreamSynchronize(b->last_stream));
    b->pending = false;
    if (ctrl[0]) VPT_HIP(hipMemset(b->d_ctrl, 0, sizeof(uint32_t)));   // reported once; accumulates over every call enqueued since the last sync
    return status_from_bits(ctrl[0]);
}

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
    uint64_t total_b = 0, max_bytes = 0, max_chars = 0;
    if ((st = stage(b, utf8, byte_offsets, out_offsets, n_sentences, nullptr, &total_b, &max_bytes, &max_chars)) != VPT_OK) return st;
    b->max_chars = max_chars;
    b->flags = flags;
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

vpt_status vpt_fill_tags_batch(const vpt_predictor* p, const uint8_t* utf8, const uint64_t* byte_offsets, size_t n_sentences,
                               const uint64_t* out_offsets, const uint8_t* labels, int32_t* tags_out) {
    return vpt_fill_tags_batch_flags(p, utf8, byte_offsets, n_sentences, out_offsets, labels, tags_out, 0u);
}

vpt_status vpt_fill_tags_batch_flags(const vpt_predictor* p, const uint8_t* utf8, const uint64_t* byte_offsets, size_t n_sentences,
                                     const uint64_t* out_offsets, const uint8_t* labels, int32_t* tags_out, unsigned flags) {
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
    const size_t n_tag_words = size_t(total_b + n_sentences) * p->n_tags;
    if ((st = grow(&b->d_tags, &b->tags_cap, n_tag_words + 16)) != VPT_OK) return st;
    b->flags = flags;
    st = vpt_fill_tags_batch_device(p, b, b->d_text, b->d_boff, b->d_ooff, n_sentences, total_b, b->d_labels, b->d_tags, b->own_stream);
    if (st != VPT_OK) return st;
    if ((st = vpt_batch_sync(b)) != VPT_OK) return st;
    VPT_HIP(hipMemcpy(tags_out + size_t(out_offsets[0]) * p->n_tags, b->d_tags, n_tag_words * sizeof(int32_t), hipMemcpyDeviceToHost));
    return VPT_OK;
}

vpt_status vpt_fill_tags_batch_device(const vpt_predictor* p, vpt_batch* b, const uint8_t* d_utf8, const uint64_t* d_byte_offsets,
                                      const uint64_t* d_out_offsets, size_t n_sentences, uint64_t total_boundaries,
                                      const uint8_t* d_labels, int32_t* d_tags_out, void* hip_stream) {
    if (!p || !b || b->pred != p) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: batch: does not belong to this predictor");
    if (!p->predict_tags) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: this predictor is created with predict_tags = false");
    if (n_sentences == 0 || p->n_tags == 0) return VPT_OK;
    if (!d_utf8 || !d_byte_offsets || !d_out_offsets || !d_tags_out || (total_boundaries && !d_labels))
        return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: NULL device pointer");
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    VPT_HIP(hipSetDevice(p->device));
    const uint64_t total_c = total_boundaries + n_sentences;
    vpt_status st = grow(&b->d_cps, &b->cps_cap, size_t(total_c) + 16);
    if (st != VPT_OK) return st;
    if ((st = grow(&b->d_tok_model, &b->tok_model_cap, size_t(total_c) + 16)) != VPT_OK) return st;
    b->tok_model_chars = total_c;
    const uint32_t* cinfo = p->d_cinfo + ((b->flags & VPT_FLAG_KYTEA_FULLWIDTH) ? 65536 : 0);
    VPT_HIP(hipMemsetAsync(d_tags_out, 0xFF, size_t(total_c) * p->n_tags * sizeof(int32_t), stream));   // -1 = None
    VPT_HIP(hipMemsetAsync(b->d_tok_model, 0, size_t(total_c) * sizeof(int32_t), stream));
    VPT_HIP(vpt::launch_decode_chars(d_utf8, d_byte_offsets, d_out_offsets, n_sentences, total_c, cinfo, b->d_cps, b->d_ctrl, stream));
    vpt::TagParams T{};
    T.tok_tab = p->dtag.tok_tab; T.models = p->dtag.models; T.ngrams = p->dtag.ngrams; T.syms = p->dtag.syms; T.slots = p->dtag.slots;
    T.weights = p->dtag.weights; T.cinfo = cinfo; T.tok_bits = p->tok_bits; T.n_tags = p->n_tags;
    T.use_char = p->tag_use_char ? 1u : 0u; T.use_type = p->tag_use_type ? 1u : 0u;
    T.cps = b->d_cps; T.ooff = d_out_offsets; T.labels = d_labels; T.n_sent = n_sentences; T.total_chars = total_c; T.tags = d_tags_out;
    T.tok_model = b->d_tok_model;
    VPT_HIP(vpt::launch_tag_tokens(T, stream));
    b->last_stream = stream; b->pending = true;
    return VPT_OK;
}

static vpt_status emit_device(const vpt_predictor* p, vpt_batch* b, const uint8_t* d_utf8, const uint64_t* d_byte_offsets,
                              const uint64_t* d_out_offsets, size_t n_sentences, uint64_t total_boundaries, const uint8_t* d_labels,
                              const int32_t* d_tags, uint8_t* d_text_out, uint64_t text_capacity, uint64_t* d_text_offsets_out,
                              hipStream_t stream) {
    if (!p || !b || b->pred != p) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: batch: does not belong to this predictor");
    if (!d_text_offsets_out) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: NULL device pointer");
    VPT_HIP(hipSetDevice(p->device));
    if (n_sentences == 0) {
        VPT_HIP(hipMemsetAsync(d_text_offsets_out, 0, sizeof(uint64_t), stream));
        b->last_stream = stream; b->pending = true;
        return VPT_OK;
    }
    if (!d_utf8 || !d_byte_offsets || !d_out_offsets || (total_boundaries && !d_labels) || (text_capacity && !d_text_out))
        return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: NULL device pointer");
    vpt::EmitParams E{};
    E.text = d_utf8; E.boff = d_byte_offsets; E.ooff = d_out_offsets; E.labels = d_labels; E.n_sent = n_sentences;
    E.total_boundaries = total_boundaries; E.out_text = d_text_out; E.out_offsets = d_text_offsets_out; E.capacity = text_capacity;
    E.status = b->d_ctrl;
    if (d_tags && p->n_tags > 0) {   // "/tag" suffixes: the indices of fill_tags + the tag model it found for every token
        if (!p->predict_tags) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: this predictor is created with predict_tags = false");
        if (b->tok_model_chars != total_boundaries + n_sentences || !b->d_tok_model)
            return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: batch: call vpt_fill_tags_batch_device on this workspace for this batch first");
        E.tags = d_tags; E.tok_model = b->d_tok_model; E.n_tags = p->n_tags; E.n_models = p->dtag.n_models; E.n_strings = p->dtag.n_strings;
        E.models = p->dtag.models; E.slot_str = p->dtag.slot_str; E.str_off = p->dtag.str_off; E.str_bytes = p->dtag.str_bytes;
    }
    VPT_HIP(vpt::launch_emit_tokenized(E, stream));
    b->last_stream = stream; b->pending = true;
    return VPT_OK;
}

vpt_status vpt_write_tokenized_batch_device(const vpt_predictor* p, vpt_batch* b, const uint8_t* d_utf8, const uint64_t* d_byte_offsets,
                                            const uint64_t* d_out_offsets, size_t n_sentences, uint64_t total_boundaries,
                                            const uint8_t* d_labels, uint8_t* d_text_out, uint64_t text_capacity,
                                            uint64_t* d_text_offsets_out, void* hip_stream) {
    return emit_device(p, b, d_utf8, d_byte_offsets, d_out_offsets, n_sentences, total_boundaries, d_labels, nullptr, d_text_out, text_capacity,
                       d_text_offsets_out, static_cast<hipStream_t>(hip_stream));
}

vpt_status vpt_write_tagged_batch_device(const vpt_predictor* p, vpt_batch* b, const uint8_t* d_utf8, const uint64_t* d_byte_offsets,
                                         const uint64_t* d_out_offsets, size_t n_sentences, uint64_t total_boundaries,
                                         const uint8_t* d_labels, const int32_t* d_tags, uint8_t* d_text_out, uint64_t text_capacity,
                                         uint64_t* d_text_offsets_out, void* hip_stream) {
    if (p && p->n_tags > 0 && !d_tags) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: NULL device pointer");
    return emit_device(p, b, d_utf8, d_byte_offsets, d_out_offsets, n_sentences, total_boundaries, d_labels, d_tags, d_text_out, text_capacity,
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
    if (with_tags) {   // Sentence::fill_tags, then the writer, as the CLI does (predict/src/main.rs:156-176)
        if ((st = grow(&b->d_tags, &b->tags_cap, size_t(total_b + n_sentences) * p->n_tags + 16)) != VPT_OK) return st;
        b->flags = flags;
        st = vpt_fill_tags_batch_device(p, b, b->d_text, b->d_boff, b->d_ooff, n_sentences, total_b, b->d_labels, b->d_tags, b->own_stream);
        if (st != VPT_OK) return st;
    }
    st = emit_device(p, b, b->d_text, b->d_boff, b->d_ooff, n_sentences, total_b, b->d_labels, with_tags ? b->d_tags : nullptr, b->d_tok,
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

vpt_status vpt_count_boundaries_device(const vpt_predictor* p, vpt_batch* b, const uint8_t* d_utf8, const uint64_t* d_byte_offsets,
                                       size_t n_sentences, uint64_t* d_out_offsets, void* hip_stream) {
    if (!p || !b || b->pred != p) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: batch: does not belong to this predictor");
    if (!d_out_offsets || (n_sentences && (!d_utf8 || !d_byte_offsets))) return fail(VPT_INVALID_ARGUMENT, "InvalidArgumentError: NULL device pointer");
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    VPT_HIP(hipSetDevice(p->device));
    VPT_HIP(hipMemsetAsync(b->d_ctrl + 2, 0, sizeof(uint32_t), stream));   // [2]: the longest sentence in chars
    if (n_sentences == 0) VPT_HIP(hipMemsetAsync(d_out_offsets, 0, sizeof(uint64_t), stream));
    else VPT_HIP(vpt::launch_count_boundaries(d_utf8, d_byte_offsets, n_sentences, d_out_offsets, b->d_ctrl, b->d_ctrl + 2, stream));
    b->last_stream = stream; b->pending = true;
    return VPT_OK;
}

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
    hipStream_t s = b->own_stream;
    const size_t nbytes = size_t(t1 - t0);
    if ((st = grow(&b->d_text, &b->text_cap, nbytes + 32)) != VPT_OK) return st;
    {
        size_t cap = b->off_cap;
        if ((st = grow(&b->d_boff, &cap, n_sentences + 1)) != VPT_OK) return st;
        size_t cap2 = b->off_cap;
        if ((st = grow(&b->d_ooff, &cap2, n_sentences + 1)) != VPT_OK) return st;
        b->off_cap = std::min(cap, cap2);
    }
    std::vector<uint64_t>& boff = b->h_boff;
    boff.resize(n_sentences + 1);
    for (size_t i = 0; i <= n_sentences; ++i) boff[i] = byte_offsets[i] - t0;
    VPT_HIP(hipMemcpyAsync(b->d_text, utf8 + t0, nbytes, hipMemcpyHostToDevice, s));
    VPT_HIP(hipMemcpyAsync(b->d_boff, boff.data(), 8 * (n_sentences + 1), hipMemcpyHostToDevice, s));
    // chars per sentence on the device; the totals come back to size the outputs (the one mid-pipeline sync)
    if ((st = vpt_count_boundaries_device(p, b, b->d_text, b->d_boff, n_sentences, b->d_ooff, s)) != VPT_OK) return st;
    uint64_t total_b = 0;
    uint32_t max_chars = 0;
    VPT_HIP(hipMemcpyAsync(&total_b, b->d_ooff + n_sentences, sizeof(total_b), hipMemcpyDeviceToHost, s));
    VPT_HIP(hipMemcpyAsync(&max_chars, b->d_ctrl + 2, sizeof(max_chars), hipMemcpyDeviceToHost, s));
    if ((st = vpt_batch_sync(b)) != VPT_OK) return st;   // NUL / empty sentences are reported here
    {
        size_t cap = b->out_cap;
        if ((st = grow(&b->d_scores, &cap, size_t(total_b) + 1)) != VPT_OK) return st;
        size_t cap2 = b->out_cap;
        if ((st = grow(&b->d_labels, &cap2, size_t(total_b) + 1)) != VPT_OK) return st;
        b->out_cap = std::min(cap, cap2);
    }
    if ((st = grow(&b->d_tok, &b->tok_cap, size_t(text_capacity) + 16)) != VPT_OK) return st;
    if ((st = grow(&b->d_toff, &b->toff_cap, n_sentences + 1)) != VPT_OK) return st;
    b->max_chars = max_chars;
    b->flags = flags;
    st = vpt_predict_batch_device(p, b, b->d_text, b->d_boff, b->d_ooff, n_sentences, total_b, max_bytes, nullptr, b->d_labels, s);
    if (st != VPT_OK) return st;
    const bool with_tags = tagged && p->n_tags > 0;
    if (with_tags) {
        if ((st = grow(&b->d_tags, &b->tags_cap, size_t(total_b + n_sentences) * p->n_tags + 16)) != VPT_OK) return st;
        b->flags = flags & VPT_FLAG_KYTEA_FULLWIDTH;
        st = vpt_fill_tags_batch_device(p, b, b->d_text, b->d_boff, b->d_ooff, n_sentences, total_b, b->d_labels, b->d_tags, s);
        if (st != VPT_OK) return st;
    }
    st = emit_device(p, b, b->d_text, b->d_boff, b->d_ooff, n_sentences, total_b, b->d_labels, with_tags ? b->d_tags : nullptr, b->d_tok,
                     text_capacity, b->d_toff, s);
    if (st != VPT_OK) return st;
    if ((st = vpt_batch_sync(b)) != VPT_OK) return st;
    VPT_HIP(hipMemcpy(text_offsets_out, b->d_toff, 8 * (n_sentences + 1), hipMemcpyDeviceToHost));
    const uint64_t total = text_offsets_out[n_sentences];
    if (total) VPT_HIP(hipMemcpy(text_out, b->d_tok, size_t(total), hipMemcpyDeviceToHost));
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
