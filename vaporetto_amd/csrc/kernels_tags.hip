// Tag prediction on the device: Sentence::fill_tags -> Predictor::predict_tags (sentence.rs:1144-1148,
// predictor.rs:546-637), all-matches form ("Tag scoring"):
//
//   for every token (maximal run between WordBoundary labels; tokens touching an Unknown label are skipped) whose
//   surface is the token of a tag model M:   z = M.bias
//     + w of every char tag n-gram (g, rel, w) of M with  text[p + rel + 1 - |g| .. p + rel + 1) == g   (p = last char)
//     + w of every type tag n-gram likewise on the character types
//   slot j with >= 2 candidates: argmax of its run of z, first maximum wins; 1 candidate: that one; 0: none.
//
// The reference reaches the same sums through the per-position longest-match pattern ids it records during predict
// (char_scorer/boundary_tag_scorer.rs:62-174, type_scorer/boundary_tag_scorer.rs:51-143) with suffix-merged tag
// weights; integer adds are order-free, so the sums are bit-identical.
//
// Two kernels, one wave per sentence each (a first, simple mapping: the tag path is not the measured hot path):
//   decode_chars_kernel  UTF-8 -> scalar values, flat per batch (char g of sentence i at out_offsets[i] + i + g)
//   tag_tokens_kernel    token walk over the labels, token lookup, z accumulation in LDS, argmax per slot
#include <hip/hip_runtime.h>

#include "device_common.h"
#include "kernels.hpp"

namespace vpt {
namespace {

constexpr int kTagThreads = 256;
constexpr int kTagWaves = kTagThreads / 64;

__device__ __forceinline__ uint32_t lanes_below(uint64_t mask, int lane) {
    return uint32_t(__popcll(mask & ((uint64_t(1) << lane) - 1)));
}

// `total_chars` = chars the caller's offsets promise for the whole batch (the size of `cps`): offsets that do not match
// the text raise kErrBadOffsets instead of writing outside it
__global__ __launch_bounds__(kTagThreads) void decode_chars_kernel(const uint8_t* __restrict__ text, const uint64_t* __restrict__ boff,
                                                                   const uint64_t* __restrict__ ooff, uint64_t n_sent, uint64_t total_chars,
                                                                   const uint32_t* __restrict__ cinfo, uint32_t* __restrict__ cps,
                                                                   uint8_t* __restrict__ types, uint32_t* __restrict__ status) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = uint64_t(blockIdx.x) * kTagWaves + (threadIdx.x >> 6);
    const uint64_t n_waves = uint64_t(gridDim.x) * kTagWaves;
    for (uint64_t i = wave; i < n_sent; i += n_waves) {
        const uint64_t b0 = boff[i], b1 = boff[i + 1];
        const uint64_t o0 = ooff[i], o1 = ooff[i + 1];
        const uint64_t g = o0 + i;  // first char of the sentence in the flat char array
        const bool sane = o1 >= o0 && b1 >= b0 && o1 + i + 1 <= total_chars;
        const uint64_t want = sane ? o1 - o0 + 1 : 0;   // chars of this sentence
        uint64_t seen = 0;
        for (uint64_t pos = b0; sane && pos < b1; pos += 64) {
            const uint64_t at = pos + uint64_t(lane);
            const bool in = at < b1;
            const uint32_t byte0 = in ? text[at] : 0x80u;
            const bool lead = in && (byte0 & 0xC0u) != 0x80u;
            const uint64_t m = __ballot(lead);
            const uint64_t idx = seen + lanes_below(m, lane);
            if (lead && idx < want) {
                uint32_t b4 = byte0;
                if (byte0 >= 0xC0u && at + 1 < b1) b4 |= uint32_t(text[at + 1]) << 8;
                if (byte0 >= 0xE0u && at + 2 < b1) b4 |= uint32_t(text[at + 2]) << 16;
                if (byte0 >= 0xF0u && at + 3 < b1) b4 |= uint32_t(text[at + 3]) << 24;
                const uint32_t cp = utf8_scalar(b4);
                const uint32_t info = cp < 0x10000u ? cinfo[cp] : 0u;   // the scored char | its CharacterType << 16 (BMP only)
                if (cps) cps[g + idx] = cp < 0x10000u ? info & 0xFFFFu : cp;
                if (types) types[g + idx] = uint8_t(cp < 0x10000u ? info >> 16 : char_type(cp));   // Sentence::char_types (sentence.rs:1016)
            }
            seen += uint64_t(__popcll(m));
        }
        if (lane == 0 && (!sane || seen != want)) atomicOr(status, kErrBadOffsets);   // an empty sentence too (want >= 1)
    }
}

struct TagLds {
    int32_t z[kTagWaves][1024];   // kTagMaxZ scores per token, one buffer per wave
};

__device__ __forceinline__ uint32_t type_of(const uint32_t* cinfo, uint32_t cp) {
    return cp < 0x10000u ? cinfo[cp] >> 16 : char_type(cp);   // cp is already the scored char: its image is itself
}

constexpr uint32_t kLaneZ = 16;   // tag scores a token may have to be handled by ONE lane (z interleaved in the wave's LDS buffer)

// the tag model (index + 1) whose token is cps[s0 .. e], or 0: one lane on its own
__device__ __forceinline__ uint32_t find_tag_model(const TagParams& P, const uint32_t* cps, int64_t s0, int64_t e) {
    const int64_t len = e - s0 + 1;
    uint32_t h = 0x811C9DC5u;
    for (int64_t j = 0; j < len; ++j) h = (h ^ cps[s0 + j]) * 0x01000193u;
    h ^= h >> 15;
    h *= kHashMulLo;
    const uint32_t tok_mask = (1u << P.tok_bits) - 1u;
    uint32_t slot = h >> (32 - P.tok_bits);
    for (;;) {
        const uint32_t cur = P.tok_tab[slot];
        if (cur == 0) return 0;
        const uint32_t* mr = P.models + size_t(cur - 1) * 12;
        if (int64_t(mr[1]) == len) {
            bool same = true;
            for (int64_t j = 0; j < len && same; ++j) same = P.syms[mr[0] + j] == cps[s0 + j];
            if (same) return cur;
        }
        slot = (slot + 1) & tok_mask;
    }
}

// One token handled by one lane: z lives at zl[i * 64] (i < kLaneZ), so that the lanes of a wave never share a bank.
__device__ __forceinline__ void tag_token_by_lane(const TagParams& P, const uint32_t* cps, int64_t n, int64_t e, uint64_t g0, uint32_t model,
                                                  volatile int32_t* zl) {
    const uint32_t* mr = P.models + size_t(model - 1) * 12;
    const uint32_t zlen = mr[7];
    for (uint32_t i = 0; i < zlen; ++i) zl[i * 64] = P.weights[mr[6] + i];
    for (int kind = 0; kind < 2; ++kind) {
        if (kind == 0 ? !P.use_char : !P.use_type) continue;
        const uint32_t first = mr[kind == 0 ? 2 : 4], count = mr[kind == 0 ? 3 : 5];
        for (uint32_t q = 0; q < count; ++q) {
            const uint32_t* nr = P.ngrams + size_t(first + q) * 4;
            const int64_t glen = int64_t(nr[1] & 0xFFFFFFu), rel = int64_t(nr[1] >> 24);
            const int64_t endp = e + rel + 1, beg = endp - glen;
            if (beg < 0 || endp > n) continue;
            bool same = true;
            for (int64_t j = 0; j < glen && same; ++j) {
                const uint32_t c = cps[beg + j];
                same = P.syms[nr[0] + j] == (kind == 0 ? c : type_of(P.cinfo, c));
            }
            if (!same) continue;
            const uint32_t wl = nr[3] < zlen ? nr[3] : zlen;   // zip: the shorter of the two (predictor.rs:82-89)
            for (uint32_t i = 0; i < wl; ++i) zl[i * 64] = int32_t(uint32_t(zl[i * 64]) + uint32_t(P.weights[nr[2] + i]));
        }
    }
    const uint32_t n_slots = mr[9] < P.n_tags ? mr[9] : P.n_tags;
    for (uint32_t j = 0; j < n_slots; ++j) {   // argmax per slot (TagPredictor::predict, predictor.rs:286-304)
        const uint32_t cnt = P.slots[size_t(mr[8] + j) * 2], off = P.slots[size_t(mr[8] + j) * 2 + 1];
        int32_t tag = cnt == 1 ? 0 : -1;
        if (cnt >= 2) {
            int32_t best = INT32_MIN;
            tag = 0;
            for (uint32_t c = 0; c < cnt && off + c < zlen; ++c) {
                const int32_t v = zl[(off + c) * 64];
                if (v > best) { best = v; tag = int32_t(c); }
            }
        }
        P.tags[(g0 + uint64_t(e)) * P.n_tags + j] = tag;
    }
}

// One token handled by the whole wave (tokens with more than kLaneZ tag scores): lanes over z entries, n-gram chars, slots.
__device__ __forceinline__ void tag_token_by_wave(const TagParams& P, const uint32_t* cps, int64_t n, int64_t e, uint64_t g0, uint32_t model,
                                                  volatile int32_t* z, int lane) {
    const uint32_t* mr = P.models + size_t(model - 1) * 12;
    const uint32_t zlen = mr[7];
    for (uint32_t i = lane; i < zlen; i += 64) z[i] = P.weights[mr[6] + i];
    for (int kind = 0; kind < 2; ++kind) {
        if (kind == 0 ? !P.use_char : !P.use_type) continue;
        const uint32_t first = mr[kind == 0 ? 2 : 4], count = mr[kind == 0 ? 3 : 5];
        for (uint32_t q = 0; q < count; ++q) {
            const uint32_t* nr = P.ngrams + size_t(first + q) * 4;
            const int64_t glen = int64_t(nr[1] & 0xFFFFFFu), rel = int64_t(nr[1] >> 24);
            const int64_t endp = e + rel + 1, beg = endp - glen;
            if (beg < 0 || endp > n) continue;
            bool same = true;
            for (int64_t j0 = 0; j0 < glen; j0 += 64) {
                const int64_t j = j0 + lane;
                bool ne = false;
                if (j < glen) {
                    const uint32_t c = cps[beg + j];
                    ne = P.syms[nr[0] + j] != (kind == 0 ? c : type_of(P.cinfo, c));
                }
                if (__ballot(ne) != 0) { same = false; break; }
            }
            if (!same) continue;
            const uint32_t wl = nr[3] < zlen ? nr[3] : zlen;
            for (uint32_t i = lane; i < wl; i += 64) z[i] = int32_t(uint32_t(z[i]) + uint32_t(P.weights[nr[2] + i]));
        }
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t n_slots = mr[9] < P.n_tags ? mr[9] : P.n_tags;
    for (uint32_t j = lane; j < n_slots; j += 64) {
        const uint32_t cnt = P.slots[size_t(mr[8] + j) * 2], off = P.slots[size_t(mr[8] + j) * 2 + 1];
        int32_t tag = cnt == 1 ? 0 : -1;
        if (cnt >= 2) {
            int32_t best = INT32_MIN;
            tag = 0;
            for (uint32_t c = 0; c < cnt && off + c < zlen; ++c) {
                const int32_t v = z[off + c];
                if (v > best) { best = v; tag = int32_t(c); }
            }
        }
        P.tags[(g0 + uint64_t(e)) * P.n_tags + j] = tag;
    }
    __builtin_amdgcn_wave_barrier();
}

// One wave per sentence, 64 chars per step.  Every lane whose char ends a token (its label is WordBoundary, or it is the
// last char) owns that token: it finds the token's start from the boundary masks of the step, looks the surface up and,
// when the tag model has at most kLaneZ scores (the usual case: a few candidates per slot), scores and tags the token
// on its own -- up to 64 tokens in flight per wave instead of one.  The rare bigger models go through the whole-wave
// routine afterwards, one token at a time.
__global__ __launch_bounds__(kTagThreads) void tag_tokens_kernel(const TagParams P) {
    __shared__ TagLds L;
    const int lane = threadIdx.x & 63;
    volatile int32_t* z = L.z[threadIdx.x >> 6];
    const uint64_t wave = uint64_t(blockIdx.x) * kTagWaves + (threadIdx.x >> 6);
    const uint64_t n_waves = uint64_t(gridDim.x) * kTagWaves;
    for (uint64_t si = wave; si < P.n_sent; si += n_waves) {
        const uint64_t g0 = P.ooff[si] + si;                     // flat index of the sentence's first char
        if (P.ooff[si + 1] < P.ooff[si] || P.ooff[si + 1] + si + 1 > P.total_chars) continue;   // reported by decode_chars_kernel
        const int64_t n = int64_t(P.ooff[si + 1] - P.ooff[si]) + 1;  // chars
        const uint32_t* cps = P.cps + g0;
        const uint8_t* lab = P.labels + P.ooff[si];              // n - 1 labels
        int64_t start = 0;          // where the token that is open at the beginning of this step started
        bool have_start = true;     // ... and no Unknown boundary has been seen inside it (predictor.rs:566-567)
        for (int64_t base = 0; base < n; base += 64) {
            const int64_t p = base + lane;
            const uint32_t b = p < n ? (p == n - 1 ? 1u : uint32_t(lab[p])) : 0u;
            const uint64_t ends = __ballot(b == 1u), unk = __ballot(b == 2u);
            // this lane's token, if its char ends one: [s0, p], valid when no Unknown lies inside
            const uint64_t below_me = (uint64_t(1) << lane) - 1;
            const uint64_t prev_ends = ends & below_me;
            const int prev = prev_ends ? 63 - __clzll((long long)prev_ends) : -1;
            const int64_t s0 = prev >= 0 ? base + prev + 1 : start;
            const uint64_t after_prev = prev >= 0 ? ~((uint64_t(2) << prev) - 1) : ~uint64_t(0);
            const bool valid = b == 1u && (unk & below_me & after_prev) == 0 && (prev >= 0 || have_start);
            uint32_t model = valid ? find_tag_model(P, cps, s0, p) : 0u;
            if (valid && P.tok_model) P.tok_model[g0 + uint64_t(p)] = int32_t(model);   // 0: no tag model for this surface
            const bool big = model != 0 && (P.models + size_t(model - 1) * 12)[7] > kLaneZ;
            if (model != 0 && !big) tag_token_by_lane(P, cps, n, p, g0, model, z + lane);
            uint64_t todo = __ballot(big);
            __builtin_amdgcn_wave_barrier();
            while (todo) {   // wave-uniform
                const int k = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const uint32_t mk = uint32_t(__shfl(int(model), k));
                tag_token_by_wave(P, cps, n, base + k, g0, mk, z, lane);
            }
            // the token that stays open into the next step
            if (ends) {
                const int last = 63 - __clzll((long long)ends);
                start = base + last + 1;
                have_start = last == 63 || (unk >> (last + 1)) == 0;
            } else if (unk) {
                have_start = false;
            }
        }
    }
}

}  // namespace

hipError_t launch_decode_chars(const uint8_t* text, const uint64_t* boff, const uint64_t* ooff, uint64_t n_sent, uint64_t total_chars,
                               const uint32_t* cinfo, uint32_t* cps, uint8_t* types, uint32_t* status, hipStream_t stream) {
    const uint64_t want = (n_sent + kTagWaves - 1) / kTagWaves;
    const uint32_t blocks = uint32_t(want < 1 ? 1 : want > 65536 ? 65536 : want);
    hipLaunchKernelGGL(decode_chars_kernel, dim3(blocks), dim3(kTagThreads), 0, stream, text, boff, ooff, n_sent, total_chars, cinfo, cps,
                       types, status);
    return hipGetLastError();
}

hipError_t launch_tag_tokens(const TagParams& P, hipStream_t stream) {
    const uint64_t want = (P.n_sent + kTagWaves - 1) / kTagWaves;
    const uint32_t blocks = uint32_t(want < 1 ? 1 : want > 65536 ? 65536 : want);
    hipLaunchKernelGGL(tag_tokens_kernel, dim3(blocks), dim3(kTagThreads), 0, stream, P);
    return hipGetLastError();
}

}  // namespace vpt
