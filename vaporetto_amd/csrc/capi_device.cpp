// C ABI of libvaporetto_hip.so, the device-resident entry points: Predictor::predict (tile planning + the scoring launch), Sentence::fill_tags (the tag
// records), write_tokenized_text (the flat writer), the char count, Sentence::char_types -- all on device buffers, asynchronous on the caller's stream.
#include "capi_internal.hpp"

// (the entry points have C linkage from their declarations in include/vaporetto_hip.h)

namespace vptc {

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
}  // namespace vptc

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
        VPT_HIP(vpt::launch_decode_chars(d_utf8, d_byte_offsets, d_out_offsets, n_sentences, total_c, cinfo, b->d_cps, nullptr, b->d_ctrl, stream, (b->flags & VPT_FLAG_KYTEA_FULLWIDTH) != 0));
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

namespace vptc {
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
    // profiles/r05_h_*, r05_k_*, r06_t_*: configs[2] 2.03 ms at 128 sentences, 1.80 at 256; tagged configs[4] 2.39 / 2.10 / 2.04 at 5 K / 10 K / 20 K
    // chars; without tags up to 32 K chars: 512 sentences of 64, `r06_zv_*`); at most 512 sentences (two a thread).  With tags: a whole multiple of fill_tags' runs, so that a workgroup's records are run_pref[a] .. run_pref[b].
    vpt::EmitFuse F{};
    {
        const uint64_t chars = total_boundaries + n_sentences;
        const uint64_t auto_run = std::min<uint64_t>(std::max<uint64_t>(chars / (uint64_t(16) * std::max<uint32_t>(p->n_cus, 64)), 5120), E.records ? 20480 : 32768);
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
}  // namespace vptc

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

namespace vptc {
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
}  // namespace vptc

vpt_status vpt_count_boundaries_device(const vpt_predictor* p, vpt_batch* b, const uint8_t* d_utf8, const uint64_t* d_byte_offsets,
                                       size_t n_sentences, uint64_t* d_out_offsets, void* hip_stream) {
    return count_boundaries_impl(p, b, d_utf8, d_byte_offsets, n_sentences, d_out_offsets, hip_stream, 0);
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
                                     b->d_ctrl, stream, (b->flags & VPT_FLAG_KYTEA_FULLWIDTH) != 0));
    b->last_stream = stream; b->pending = true; b->cps_text = nullptr;
    return VPT_OK;
}
