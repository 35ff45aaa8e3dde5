// Token emission on the device: Sentence::write_tokenized_text for a batch (sentence.rs:850-886), boundary part:
// the tokens of a sentence are the runs between WordBoundary labels (sentence.rs:1270-1300), written in order with
// one ' ' between them and a '\' in front of every ' ', '\' and '/' byte of a surface; with tags (the indices
// vpt_fill_tags_batch wrote and the tag model it found for every token) each token is followed by "/tag" for its tag
// slots up to the last Some, an empty string for a None in between (sentence.rs:866-881).  (Unknown boundaries only
// come from partially annotated corpora, never from predict: they are rejected here, kErrUnknownLabel.)
//
// Output size is data dependent, so three launches on one stream:
//   emit_count_kernel   one wave per sentence: bytes this sentence will take -> offsets[i + 1]
//   emit_scan_kernel    one workgroup: exclusive prefix sum over the sentences, in place (offsets[0] = 0)
//   emit_write_kernel   one wave per sentence: 64 text bytes per step, a wave prefix sum places every byte
// A sentence's bytes are independent of the other sentences', its position is not: that is the scan.
#include <hip/hip_runtime.h>

#include "device_common.h"
#include "kernels.hpp"

namespace vpt {
namespace {

constexpr int kEmitThreads = 256;
constexpr int kEmitWaves = kEmitThreads / 64;
constexpr int kScanThreads = 1024;

__device__ __forceinline__ uint32_t below(uint64_t mask, int lane) { return uint32_t(__popcll(mask & ((uint64_t(1) << lane) - 1))); }

// what one text byte turns into: [' '] ['\'] byte
struct ByteOut {
    uint32_t n;        // 1..3 bytes (0 outside the sentence)
    bool space, esc;
};

// The per-chunk state machine shared by the counting and the writing kernel.  `chars_before` = chars of the sentence
// in front of this chunk; returns what this lane's byte becomes and, through `leads`, the chunk's lead-byte mask.
__device__ __forceinline__ ByteOut classify_byte(const uint8_t* __restrict__ text, uint64_t at, uint64_t b1, const uint8_t* __restrict__ lab,
                                                 uint64_t n_labels, uint64_t chars_before, int lane, uint64_t& leads, uint32_t& err) {
    const bool in = at < b1;
    const uint32_t byte = in ? text[at] : 0x80u;
    const bool lead = in && (byte & 0xC0u) != 0x80u;
    leads = __ballot(lead);
    ByteOut o;
    o.esc = in && (byte == 0x20u || byte == 0x5Cu || byte == 0x2Fu);
    o.space = false;
    if (lead) {
        const uint64_t ci = chars_before + below(leads, lane);   // index of this char in the sentence
        if (ci >= 1) {
            if (ci - 1 < n_labels) {
                const uint32_t l = lab[ci - 1];
                o.space = l == 1u;
                if (l > 1u) err |= kErrUnknownLabel;
            } else {
                err |= kErrBadOffsets;   // more chars than out_offsets promise
            }
        }
    }
    o.n = in ? 1u + (o.esc ? 1u : 0u) + (o.space ? 1u : 0u) : 0u;
    return o;
}

// "/tag/tag.." of the token whose last char is char `c` (batch-flat index): bytes it takes; written to `dst` when given
__device__ __forceinline__ uint32_t tag_suffix(const EmitParams& P, uint64_t c, uint8_t* dst) {
    if (!P.tags) return 0;
    const int32_t model = P.tok_model[c];
    if (model <= 0 || uint32_t(model) > P.n_models) return 0;   // no tag model for this surface (or not our array)
    const uint32_t* mr = P.models + size_t(model - 1) * 12;
    const int32_t* tg = P.tags + c * P.n_tags;
    const uint32_t n_slots = mr[9] < P.n_tags ? mr[9] : P.n_tags;
    uint32_t last = 0;   // slots to write: up to the last Some
    for (uint32_t j = 0; j < n_slots; ++j)
        if (tg[j] >= 0) last = j + 1;
    uint32_t n = 0;
    for (uint32_t j = 0; j < last; ++j) {
        if (dst) dst[n] = 0x2Fu;
        ++n;
        if (tg[j] < 0) continue;
        const uint32_t k = P.slot_str[mr[8] + j] + uint32_t(tg[j]);
        if (k >= P.n_strings) continue;                            // an index fill_tags cannot have written
        const uint32_t a = P.str_off[k], b = P.str_off[k + 1];
        if (dst) for (uint32_t q = a; q < b; ++q) dst[n + (q - a)] = P.str_bytes[q];
        n += b - a;
    }
    return n;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t x) {   // total over the 64 lanes, in every lane
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x += uint32_t(__shfl_xor(int(x), d));
    return x;
}
__device__ __forceinline__ uint32_t wave_exclusive(uint32_t x, int lane) {   // sum of the lanes below this one
    uint32_t incl = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = uint32_t(__shfl_up(int(incl), unsigned(d)));
        if (lane >= d) incl += t;
    }
    return incl - x;
}

__global__ __launch_bounds__(kEmitThreads) void emit_count_kernel(const EmitParams P) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = uint64_t(blockIdx.x) * kEmitWaves + (threadIdx.x >> 6);
    const uint64_t n_waves = uint64_t(gridDim.x) * kEmitWaves;
    uint32_t err = 0;
    for (uint64_t i = wave; i < P.n_sent; i += n_waves) {
        const uint64_t b0 = P.boff[i], b1 = P.boff[i + 1], o0 = P.ooff[i], o1 = P.ooff[i + 1];
        const bool sane = b1 > b0 && o1 >= o0 && o1 <= P.total_boundaries;
        const uint64_t n_labels = sane ? o1 - o0 : 0;
        uint64_t chars = 0, bytes_out = 0;
        for (uint64_t pos = b0; sane && pos < b1; pos += 64) {
            uint64_t leads;
            const ByteOut o = classify_byte(P.text, pos + uint64_t(lane), b1, P.labels + o0, n_labels, chars, lane, leads, err);
            // a token's tag suffix goes in front of the space that starts the next token: char index of its last char
            const uint32_t sfx = o.space ? tag_suffix(P, o0 + i + chars + below(leads, lane) - 1, nullptr) : 0u;
            bytes_out += wave_sum(o.n + sfx);
            chars += uint64_t(__popcll(leads));
        }
        if (!sane) err |= b1 > b0 ? kErrBadOffsets : kErrEmptySentence;
        else if (chars != n_labels + 1) err |= kErrBadOffsets;
        else bytes_out += tag_suffix(P, o1 + i, nullptr);   // the last token's (every lane computes the same)
        if (lane == 0) P.out_offsets[i + 1] = bytes_out;
    }
    if (err) atomicOr(P.status, err);
}

// exclusive scan over offsets[1 .. n] in place (offsets[k] = sum of the lengths of sentences 0 .. k-1), one workgroup
__global__ __launch_bounds__(kScanThreads) void emit_scan_kernel(uint64_t* __restrict__ offsets, uint64_t n, uint64_t capacity,
                                                                 uint32_t* __restrict__ status) {
    __shared__ uint64_t part[kScanThreads];
    const uint32_t tid = threadIdx.x;
    const uint64_t per = (n + kScanThreads - 1) / kScanThreads;     // consecutive sentences per thread
    const uint64_t lo = uint64_t(tid) * per, hi = lo + per < n ? lo + per : n;
    uint64_t sum = 0;
    for (uint64_t k = lo; k < hi; ++k) sum += offsets[k + 1];
    part[tid] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < uint32_t(kScanThreads); d <<= 1) {        // Hillis-Steele over the per-thread totals
        const uint64_t v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint64_t run = tid == 0 ? 0 : part[tid - 1];
    for (uint64_t k = lo; k < hi; ++k) {                               // lengths -> end offsets, in place
        run += offsets[k + 1];
        offsets[k + 1] = run;
    }
    if (tid == 0) {
        offsets[0] = 0;
        if (part[kScanThreads - 1] > capacity) atomicOr(status, kErrOutputTooSmall);
    }
}

__global__ __launch_bounds__(kEmitThreads) void emit_write_kernel(const EmitParams P) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = uint64_t(blockIdx.x) * kEmitWaves + (threadIdx.x >> 6);
    const uint64_t n_waves = uint64_t(gridDim.x) * kEmitWaves;
    for (uint64_t i = wave; i < P.n_sent; i += n_waves) {
        const uint64_t b0 = P.boff[i], b1 = P.boff[i + 1], o0 = P.ooff[i], o1 = P.ooff[i + 1];
        if (!(b1 > b0 && o1 >= o0 && o1 <= P.total_boundaries)) continue;   // reported by emit_count_kernel
        const uint64_t n_labels = o1 - o0;
        const uint64_t end = P.out_offsets[i + 1];
        uint64_t at_out = P.out_offsets[i], chars = 0;
        if (end > P.capacity) continue;                                    // kErrOutputTooSmall
        uint32_t err = 0;
        for (uint64_t pos = b0; pos < b1; pos += 64) {
            uint64_t leads;
            const uint64_t at = pos + uint64_t(lane);
            const ByteOut o = classify_byte(P.text, at, b1, P.labels + o0, n_labels, chars, lane, leads, err);
            const uint64_t prev = o0 + i + chars + below(leads, lane) - 1;   // last char of the token in front of a space
            const uint32_t sfx = o.space ? tag_suffix(P, prev, nullptr) : 0u;
            uint64_t w = at_out + wave_exclusive(o.n + sfx, lane);
            if (o.n != 0 && w + o.n + sfx <= end) {                        // `end` only binds when the inputs changed under us
                if (sfx) w += tag_suffix(P, prev, P.out_text + w);
                if (o.space) P.out_text[w++] = 0x20u;
                if (o.esc) P.out_text[w++] = 0x5Cu;
                P.out_text[w] = P.text[at];
            }
            at_out += wave_sum(o.n + sfx);
            chars += uint64_t(__popcll(leads));
        }
        if (lane == 0 && chars == n_labels + 1) {                          // the last token's tags
            const uint32_t sfx = tag_suffix(P, o1 + i, nullptr);
            if (sfx && at_out + sfx <= end) tag_suffix(P, o1 + i, P.out_text + at_out);
        }
    }
}

// vpt_count_boundaries on the device: chars - 1 of every sentence -> offsets[i + 1] (the scan follows), the same
// validation as Sentence::from_raw (sentence.rs:160-196), the longest sentence (in chars) -> *max_chars
__global__ __launch_bounds__(kEmitThreads) void count_chars_kernel(const uint8_t* __restrict__ text, const uint64_t* __restrict__ boff,
                                                                   uint64_t n_sent, uint64_t* __restrict__ offsets, uint32_t* __restrict__ status,
                                                                   uint32_t* __restrict__ max_chars) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = uint64_t(blockIdx.x) * kEmitWaves + (threadIdx.x >> 6);
    const uint64_t n_waves = uint64_t(gridDim.x) * kEmitWaves;
    uint32_t err = 0, longest = 0;
    for (uint64_t i = wave; i < n_sent; i += n_waves) {
        const uint64_t b0 = boff[i], b1 = boff[i + 1];
        uint64_t chars = 0;
        bool nul = false;
        for (uint64_t pos = b0; pos < b1; pos += 64) {
            const uint64_t at = pos + uint64_t(lane);
            const bool in = at < b1;
            const uint32_t byte = in ? text[at] : 0x80u;
            nul = nul || (in && byte == 0);
            chars += uint64_t(__popcll(__ballot(in && (byte & 0xC0u) != 0x80u)));
        }
        if (__ballot(nul) != 0) err |= kErrNulChar;
        if (b1 <= b0 || chars == 0) err |= kErrEmptySentence;
        if (chars > 0xFFFFFFFFull) err |= kErrBadOffsets;
        longest = chars > longest ? uint32_t(chars) : longest;
        if (lane == 0) offsets[i + 1] = chars > 0 ? chars - 1 : 0;
    }
    if (err) atomicOr(status, err);
    if (lane == 0 && longest) atomicMax(max_chars, longest);
}

}  // namespace

hipError_t launch_count_boundaries(const uint8_t* text, const uint64_t* boff, uint64_t n_sent, uint64_t* ooff_out, uint32_t* status,
                                   uint32_t* max_chars, hipStream_t stream) {
    const uint64_t want = (n_sent + kEmitWaves - 1) / kEmitWaves;
    const uint32_t blocks = uint32_t(want < 1 ? 1 : want > 65536 ? 65536 : want);
    hipLaunchKernelGGL(count_chars_kernel, dim3(blocks), dim3(kEmitThreads), 0, stream, text, boff, n_sent, ooff_out, status, max_chars);
    hipLaunchKernelGGL(emit_scan_kernel, dim3(1), dim3(kScanThreads), 0, stream, ooff_out, n_sent, ~uint64_t(0), status);
    return hipGetLastError();
}

hipError_t launch_emit_tokenized(const EmitParams& P, hipStream_t stream) {
    const uint64_t want = (P.n_sent + kEmitWaves - 1) / kEmitWaves;
    const uint32_t blocks = uint32_t(want < 1 ? 1 : want > 65536 ? 65536 : want);
    hipLaunchKernelGGL(emit_count_kernel, dim3(blocks), dim3(kEmitThreads), 0, stream, P);
    hipLaunchKernelGGL(emit_scan_kernel, dim3(1), dim3(kScanThreads), 0, stream, P.out_offsets, P.n_sent, P.capacity, P.status);
    hipLaunchKernelGGL(emit_write_kernel, dim3(blocks), dim3(kEmitThreads), 0, stream, P);
    return hipGetLastError();
}

}  // namespace vpt
