// C ABI of libvaporetto_hip.so, the host-buffer entry points: the pipelines that move a caller's batch over PCIe around the device-resident calls
// (vpt_predict_batch: chunks over lanes or three streams; vpt_tokenize_batch: lines in, tokenized lines out), the sharded call, single sentences.
#include "capi_internal.hpp"

// (the entry points have C linkage from their declarations in include/vaporetto_hip.h)

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
            if (rc[r] != VPT_OK) msg[r] = vpt_g_last_error;   // thread-local: carried back to the caller's thread
        });
    for (std::thread& t : th) t.join();
    for (size_t r = 0; r < n_preds; ++r)
        if (rc[r] != VPT_OK) return fail(rc[r], msg[r]);
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
