// C ABI of libvaporetto_hip.so (include/vaporetto_hip.h), first of three files: models, predictors (Predictor::new: the table compiler's output in one
// device arena; save / load / describe / adopt / clone) and workspaces.  capi_device.cpp: the device-resident entry points (Predictor::predict,
// Sentence::fill_tags, write_tokenized_text on device buffers); capi_host.cpp: the host-buffer pipelines over them.
#include "capi_internal.hpp"

thread_local std::string vpt_g_last_error;

extern "C" {

const char* vpt_last_error(void) { return vpt_g_last_error.c_str(); }
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
        // (decode_chars_kernel reads the table only where the filter can differ from the identity: every char it rewrites lies in these ranges)
        if (fw != cp && !(cp < 0x80u || (cp - 0x2000u) < 0x600u || (cp - 0xFF00u) < 0xF0u))
            return fail(VPT_RUNTIME_ERROR, "internal error: KyteaFullwidthFilter rewrites a char outside the ranges the decode kernel looks up");
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

}  // extern "C"
