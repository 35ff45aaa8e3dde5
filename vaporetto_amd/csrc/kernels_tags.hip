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
// What fill_tags leaves (round 6): the reference stores None for every char that does not end a token with a tag model
// (predictor.rs:558-573) -- here NOTHING is stored for them.  The device-side result is one RECORD per token that can have a tag model
// (last char, tag model, chosen candidates, where the strings of its "/tag" suffix are), sorted by position (TagParams, kernels.hpp); the writer
// reads the records, and the dense (chars x n_tags) array of the C ABI is a scatter of them over a memset for the callers that ask for it.
//
// Kernels:
//   decode_chars_kernel  UTF-8 -> flat per batch (char g of sentence i at out_offsets[i] + i + g): the scored scalar value
//                        | CharacterType << 24; optionally Sentence::char_types on their own
//   tag_filter_summary_kernel + tag_front_flat_kernel
//                        The FRONT END, flat over the batch's chars: a wave takes a run of consecutive sentences (about 2 K chars) and walks it
//                        in 128-char steps with every lane busy; sentence starts and ends come from one bitmap of marks per step built from the
//                        run's offsets.  Every lane whose char ends a token owns it; the token table's FILTER -- keyed like the table by the
//                        length and the first four chars, which the lane reads from the ring in LDS, so there is no loop over the token --
//                        says whether the surface can be a tag model's at all (a summary of it sits in LDS: five of six token ends need no
//                        load).  The CANDIDATES it lets through are stored, in order, at the run's own places in HBM and counted: a candidate's
//                        number in its run names its record.  The kernel stores nothing else and waits for nothing it stores (until round 6 it
//                        stored the None entries of every char -- 1.07 of configs[4]'s 1.6 GB of writes -- and did the lookups itself, a wave
//                        waiting on each chain of trips: 0.53 of its 0.65 ms).
//   launch_scan          the runs' candidate counts -> the runs' first records (a chained scan, kernels_emit.hip)
//   tag_resolve_kernel   The LOOKUPS, a lane per candidate, as many waves as the device holds: which tag model has this surface?  The answer goes
//                        into the candidate's record (none: an empty record).
//   tag_pass_kernel      The PASSES over the records whose token has a model, 16 tokens at a time: their context chars (p - 11 .. p + 4), model records and bias arrive in
//                        one trip; a model's char n-grams come in groups by rel_position with a 64-bit filter over the chars they END with,
//                        so a token only enumerates the groups its text can match; the wave's lanes then take (token, tag n-gram) PAIRS,
//                        64 per round: a lane checks one whole n-gram from one 32-byte record against the token's context in LDS; the
//                        matches are collected and lanes over (match, score) pairs add their weights to the token's scores in LDS (which
//                        start as the model's bias); lanes over (token, slot) pairs take the argmax (first maximum, predictor.rs:286-304).
//                        Models that do not fit the record form (an n-gram over 12 symbols or outside the BMP, more than 16 scores or 3
//                        slots, rel_position above 3) go through a whole-wave routine, one token at a time.
//   (Until round 6 there was a one-launch kernel for small batches and for a queue that overflowed -- tag_tokens_kernel, a wave per
//   sentence -- and an A/B of the front end by sentence; the queue now holds a token per char of the batch and cannot overflow: HISTORY.md.)
//   With predict_tags the scoring kernel of the preceding vpt_predict_batch_device call leaves the decoded chars behind and
//   decode_chars_kernel is skipped (capi.cpp).
#include <hip/hip_runtime.h>

#include "device_common.h"
#include "kernels.hpp"
#include <algorithm>
#include <atomic>
#include <type_traits>
#include <cstdlib>
#include <cstdio>

namespace vpt {
namespace {

constexpr int kTagThreads = 256;
constexpr int kTagWaves = kTagThreads / 64;

__device__ __forceinline__ uint64_t wave_sum64(uint64_t x) {   // total over the 64 lanes, in every lane
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t lo = uint32_t(__shfl_xor(int(uint32_t(x)), d)), hi = uint32_t(__shfl_xor(int(uint32_t(x >> 32)), d));
        x += uint64_t(lo) | (uint64_t(hi) << 32);
    }
    return x;
}
__device__ __forceinline__ uint32_t lanes_below(uint64_t mask, int lane) {
    return uint32_t(__popcll(mask & ((uint64_t(1) << lane) - 1)));
}

// `total_chars` = chars the caller's offsets promise for the whole batch (the size of `cps`): offsets that do not match
// the text raise kErrBadOffsets instead of writing outside it.
//
// FLAT over the text (round 4; a wave per sentence -- its offsets, then its text, then the char table: three dependent trips per
// sentence -- took 1.23 ms on configs[4], longer than fill_tags itself): a workgroup takes `per_block` consecutive sentences and
// streams their bytes in pieces of 4 KB, sixteen bytes per thread; a block-wide prefix sum over the threads' lead counts numbers the
// chars -- char number c of the run goes to flat position (ooff[s0] + s0) + c, which IS ooff[i] + i + g for char g of sentence i when
// the offsets match the text, and that is checked per sentence (the same two LDS reads per sentence as count_chars_kernel).
constexpr uint32_t kDecodePiece = kTagThreads * 16;
__global__ __launch_bounds__(kTagThreads) void decode_chars_kernel(const uint8_t* __restrict__ text, const uint64_t* __restrict__ boff,
                                                                   const uint64_t* __restrict__ ooff, uint64_t n_sent, uint64_t total_chars,
                                                                   const uint32_t* __restrict__ cinfo, uint32_t* __restrict__ cps,
                                                                   uint8_t* __restrict__ types, uint32_t* __restrict__ status, uint32_t per_block, uint32_t fullwidth) {
    __shared__ uint16_t masks[kTagThreads];   // lead mask of every 16-byte chunk of the piece
    __shared__ uint32_t pfx[kTagThreads];     // leads of the piece in front of the chunk
    __shared__ uint32_t wtot[kTagWaves];
    __shared__ uint32_t tlut[80];             // CharacterType nibbles of ASCII, U+30xx and U+FFxx (below)
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
    const uint64_t s0 = uint64_t(blockIdx.x) * per_block;
    if (s0 >= n_sent) return;
    const uint32_t ns = uint32_t(n_sent - s0 < per_block ? n_sent - s0 : per_block);
    const uint64_t B0 = boff[s0], B1 = boff[s0 + ns];
    const uint64_t G0 = ooff[s0] + s0;        // flat position of the run's first char
    uint64_t my_b = 0, my_e = 0, my_want = 0;
    const bool mine = tid < ns;
    bool sane = false;
    if (mine) {
        my_b = boff[s0 + tid]; my_e = boff[s0 + tid + 1];
        const uint64_t o0 = ooff[s0 + tid], o1 = ooff[s0 + tid + 1];
        // the sentence sits where the run's char count puts it, and the whole of it inside the batch's arrays
        sane = my_e > my_b && my_b >= B0 && my_e <= B1 && o1 >= o0 && o1 + (s0 + tid) + 1 <= total_chars;
        my_want = sane ? o1 - o0 + 1 : 0;
    }
    uint32_t err = (mine && !sane) ? kErrBadOffsets : 0u;
    if (B1 <= B0 || G0 > total_chars || B1 - B0 >= 0xFFFFFF00ull) { if (tid == 0) atomicOr(status, kErrBadOffsets); return; }   // (a run of 4 GB: not in the chars' index width below)
    // CharacterType of the chars a Japanese text is made of, as nibbles in LDS: ASCII, U+3000 .. 30FF (punctuation, hiragana, katakana), U+FF00 .. FFFF
    // (fullwidth and halfwidth forms); the main block of kanji is one compare; anything else is rare and takes char_type's twenty range tests -- which
    // every char of a wave took before: 45 of the 80 vector instructions a char cost, in a kernel that ran at the vector ALU's issue rate (round 6)
    if (tid < 80u) {
        const uint32_t c0 = tid < 16u ? 8u * tid : tid < 48u ? 0x3000u + 8u * (tid - 16u) : 0xFF00u + 8u * (tid - 48u);
        uint32_t w = 0;
        for (uint32_t j = 0; j < 8u; ++j) w |= char_type(c0 + j) << (4u * j);
        tlut[tid] = w;
    }
    __syncthreads();
    uint64_t p_start = 0, p_end = 0, carry = 0;
    const uintptr_t t_lo = reinterpret_cast<uintptr_t>(text) + B0, t_hi = reinterpret_cast<uintptr_t>(text) + B1;
    // (the run's chars from its first: 32 bits -- a run is a few sentences -- against what the batch's arrays hold behind it)
    const uint32_t room = total_chars - G0 < 0xFFFFFFFFull ? uint32_t(total_chars - G0) : 0xFFFFFFFFu;
    uint32_t* const cps_run = cps ? cps + G0 : nullptr;
    uint8_t* const types_run = types ? types + G0 : nullptr;
    for (uintptr_t piece = t_lo & ~uintptr_t(15); piece < t_hi; piece += kDecodePiece) {
        const uintptr_t a = piece + 16u * tid;
        uint32_t d[5] = {0, 0, 0, 0, 0};
        if (a + 16 > t_lo && a < t_hi) {
            const uint4 v = *reinterpret_cast<const uint4*>(a);
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            if (a + 16 < t_hi) d[4] = *reinterpret_cast<const uint32_t*>(a + 16);   // the tail of a char that starts in the chunk's last bytes
        }
        const uint32_t lo = t_lo > a ? (t_lo - a < 16 ? uint32_t(t_lo - a) : 16u) : 0u;
        const uint32_t hi = t_hi > a ? (t_hi - a < 16 ? uint32_t(t_hi - a) : 16u) : 0u;
        const uint32_t vm = hi > lo ? ((1u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;
        const uint32_t lm = flag_bytes_to_mask16(lead_flags(d[0]), lead_flags(d[1]), lead_flags(d[2]), lead_flags(d[3])) & vm;
        const uint32_t cnt = uint32_t(__popc(lm));
        masks[tid] = uint16_t(lm);
        const uint32_t incl = wave_inclusive_scan(cnt);
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        uint32_t woff = 0, total = 0;
#pragma unroll
        for (uint32_t k = 0; k < uint32_t(kTagWaves); ++k) {
            const uint32_t u = wtot[k];
            if (k < wave) woff += u;
            total += u;
        }
        const uint32_t excl = woff + incl - cnt;
        pfx[tid] = excl;
        __syncthreads();
        // ---- this chunk's chars, one after the other (a wave walks as many as its fullest chunk holds: six of CJK text), decoded from registers
        uint32_t dest = uint32_t(carry) + excl;   // (carry < 2^32: the run's chars so far)
        for (uint32_t rem = lm; rem != 0; rem &= rem - 1u) {
            const uint32_t k = uint32_t(__builtin_ctz(rem)), q = k >> 2;
            const uint32_t w0 = q == 0 ? d[0] : q == 1 ? d[1] : q == 2 ? d[2] : d[3], w1 = q == 0 ? d[1] : q == 1 ? d[2] : q == 2 ? d[3] : d[4];
            const uint32_t cp = utf8_scalar(__builtin_amdgcn_alignbyte(w1, w0, k));
            // the char it is scored as | its CharacterType: a word of the char table -- asked for only where the table can differ from the
            // identity (with KyteaFullwidthFilter: ASCII, U+2000 .. 25FF, U+FF00 .. FFEF hold every char the filter rewrites,
            // kytea_fullwidth.rs:13-117); anything else is itself and its type is arithmetic.  A gather per char -- one miss of the vector
            // L1 each -- was what the launch ran at (0.47 ms on configs[4], profiles/r06_o_*).
            const bool mapped = fullwidth && cp < 0x10000u && (cp < 0x80u || (cp - 0x2000u) < 0x600u || (cp - 0xFF00u) < 0xF0u);
            uint32_t scored = cp, ty;
            if (mapped) { const uint32_t info = cinfo[cp]; scored = info & 0xFFFFu; ty = info >> 16; }
            else if (cp - 0x4E00u <= 0x9FFFu - 0x4E00u) ty = 5u;
            else {
                const uint32_t page = cp >> 8;
                const uint32_t idx = cp < 0x80u ? cp : page == 0x30u ? 128u + (cp & 0xFFu) : page == 0xFFu ? 384u + (cp & 0xFFu) : ~0u;
                if (idx != ~0u) ty = (tlut[idx >> 3] >> (4u * (idx & 7u))) & 15u;
                else ty = char_type(cp);
            }
            if (dest < room) {
                if (cps_run) cps_run[dest] = scored | (ty << 24);
                if (types_run) types_run[dest] = uint8_t(ty);   // Sentence::char_types (sentence.rs:1016)
            }
            ++dest;
        }
        // ---- the sentences' char counts against their offsets
        auto before = [&](uintptr_t x) -> uint32_t {   // leads of the piece in front of byte x (piece <= x <= piece + kDecodePiece)
            const uint32_t r = uint32_t(x - piece);
            if (r >= kDecodePiece) return total;
            return pfx[r >> 4] + uint32_t(__popc(uint32_t(masks[r >> 4]) & ((1u << (r & 15u)) - 1u)));
        };
        if (sane) {
            const uintptr_t xs = reinterpret_cast<uintptr_t>(text) + my_b, xe = reinterpret_cast<uintptr_t>(text) + my_e;
            if (xs >= piece && xs - piece < kDecodePiece) p_start = carry + before(xs);
            if (xe > piece && xe - piece <= kDecodePiece) p_end = carry + before(xe);
        }
        carry += total;
        __syncthreads();   // the next piece rewrites masks / pfx / wtot
    }
    // a sentence holds the chars its offsets say, and starts at the run's char its offsets say (an empty sentence too: want >= 1)
    if (sane && (p_end - p_start != my_want || G0 + p_start != ooff[s0 + tid] + (s0 + tid))) err |= kErrBadOffsets;
    if (err) atomicOr(status, err);
}

constexpr uint32_t kCharMask = 0x1FFFFFu;   // a cps word: scored scalar value | CharacterType << 24
constexpr int kRing = 256;                   // the sentence's cps words in LDS: char q at txt[q & 255]: the last 192 (one launch) / 128 (front-end launch: steps of two half-steps) chars in front of a step and the step itself
constexpr int kTagPass = 16;                 // queued tokens a pass takes
constexpr int kCtx = 16, kCtxBack = 11;      // the text a queued token's n-grams can touch: chars p - 11 .. p + 4 around its last char p
static_assert(kCtx - 1 - kCtxBack == int(kTagFastMaxRel), "tables.cpp keeps models with a tag n-gram further past the token off the fast path");
constexpr int kMatchCap = 128;               // matched (token, n-gram) pairs collected before their weights are added

struct TagWaveLds {
    union {
        int32_t z[kTagMaxZ];                 // the whole-wave routine: one token's scores
        struct {
            uint32_t tok[kTagPass][12];             // the queue: tag model + 1, flat char index (2), chars before | after << 8 inside the sentence (clipped
                                                   // to the context); a pass adds: first record, scores | slots << 8 | type entries << 16 | active
                                                   // groups << 24, packed slots, the four char group sizes, slot_str of its slots (3); [11]: the token's record
            uint32_t ctx[kTagPass][kCtx];          // cps words p - 11 .. p + 4 of every token, 0 outside its sentence
            int32_t zt[kTagPass][kTagFastZ + 1];   // the scores (rows padded against bank conflicts)
            uint32_t pref[kTagPass + 1];           // records before token t (exclusive prefix of the counts)
            uint32_t mlist[kMatchCap][2];          // matches: weight offset, token | scores to add << 8
        } f;
    };
};
struct alignas(16) TagFrontLds {             // what the step loop keeps per wave
    uint32_t txt[kRing + 4];                 // (the ring keeps words 0 .. 2 once more at kRing ..: a token's first four chars are consecutive words)
};
constexpr int kTagPairOcc = 8;               // workgroups per CU of the passes (57 VGPRs)
static_assert(sizeof(TagWaveLds) * kTagWaves <= 160 * 1024 / kTagPairOcc, "workgroups per CU");

// The tag model (index + 1) whose token is the `len` chars at `tc` (flat cps words), or 0: the table is keyed by the length and the
// first four chars; a surface of up to four BMP chars is verified by the slot itself, a longer candidate against `syms`.
__device__ __forceinline__ uint32_t find_tag_model(const TagParams& P, const uint32_t* tc, uint32_t len, bool* fast) {
    uint32_t c[4];
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) c[j] = j < len ? tc[j] & kCharMask : 0u;
    const uint32_t lo = (c[0] & 0xFFFFu) | (c[1] << 16), hi = (c[2] & 0xFFFFu) | (c[3] << 16);
    const bool bmp = ((c[0] | c[1] | c[2] | c[3]) >> 16) == 0;
    const uint32_t tok_mask = (1u << P.tok_bits) - 1u;
    uint32_t slot = tag_token_hash_key(lo, hi, len) >> (32 - P.tok_bits);
    for (;;) {
        const uint4 t = reinterpret_cast<const uint4*>(P.tok_tab)[slot];
        if (t.x == 0) return 0;
        if ((t.y & kTagTokLenMask) == len && t.z == lo && t.w == hi) {
            bool same = bmp;
            if (!(t.y & kTagTokInline)) {   // a longer token (or one outside the BMP): its symbols, four to a trip
                const uint32_t so = P.models[size_t(t.x - 1) * 12];
                same = true;
                for (uint32_t j = 0; j < len && same; j += 4) {
                    uint32_t a[4], b[4];
#pragma unroll
                    for (uint32_t q = 0; q < 4; ++q) { const bool in = j + q < len; a[q] = in ? P.syms[so + j + q] : 0u; b[q] = in ? tc[j + q] & kCharMask : 0u; }
                    same = a[0] == b[0] && a[1] == b[1] && a[2] == b[2] && a[3] == b[3];
                }
            }
            if (same) { *fast = (t.y & kTagTokFast) != 0; return t.x; }
        }
        slot = (slot + 1) & tok_mask;
    }
}

// One token handled by the whole wave (models outside the record form): lanes over z entries, n-gram chars, slots.
__device__ __forceinline__ void tag_token_by_wave(const TagParams& P, const uint32_t* cps, int64_t n, int64_t e, uint64_t g0, uint32_t model,
                                                  VPT_LDS_PTR(volatile int32_t) z, int lane, uint64_t slot) {
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
                    ne = P.syms[nr[0] + j] != (kind == 0 ? (c & kCharMask) : (c >> 24));
                }
                if (__ballot(ne) != 0) { same = false; break; }
            }
            if (!same) continue;
            const uint32_t wl = nr[3] < zlen ? nr[3] : zlen;
            for (uint32_t i = lane; i < wl; i += 64) z[i] = int32_t(uint32_t(z[i]) + uint32_t(P.weights[nr[2] + i]));
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (P.scores_out) {   // Predictor::store_tag_scores: the token's whole score vector (predictor.rs:599-601)
        for (uint32_t i = lane; i < zlen && i < P.score_stride; i += 64) P.scores_out[(g0 + uint64_t(e)) * P.score_stride + i] = z[i];
    }
    const uint32_t n_slots = mr[9] < P.n_tags ? mr[9] : P.n_tags;
    uint32_t str_bytes = 0, last_some = 0;   // what the token's tags take in the tokenized text (layout.h, tok_model words)
    for (uint32_t j = lane; j < P.n_tags; j += 64) {   // the slots past the model's own are None
        const uint32_t cnt = j < n_slots ? P.slots[size_t(mr[8] + j) * 2] : 0u, off = j < n_slots ? P.slots[size_t(mr[8] + j) * 2 + 1] : 0u;
        int32_t tag = cnt == 1 ? 0 : -1;
        if (cnt >= 2) {
            int32_t best = INT32_MIN;
            tag = 0;
            for (uint32_t c = 0; c < cnt && off + c < zlen; ++c) {
                const int32_t v = z[off + c];
                if (v > best) { best = v; tag = int32_t(c); }
            }
        }
        P.rec_tags[slot * P.n_tags + j] = tag;
        if (P.tags) P.tags[(g0 + uint64_t(e)) * P.n_tags + j] = tag;
        uint2 str = make_uint2(0u, 0u);
        if (tag >= 0) {
            const uint32_t k = P.slot_str[mr[8] + j] + uint32_t(tag);
            if (k < P.n_strings) str = make_uint2(P.str_off[k], P.str_off[k + 1] - P.str_off[k]);
            const uint32_t sl = str.y;
            str_bytes += sl < 0x10000u ? sl : 0x10000u;
            last_some = j + 1;
        }
        P.rec_str[slot * P.n_tags + j] = str;
    }
    {
        const uint32_t last = wave_max(last_some);
        const uint32_t bytes = uint32_t(wave_sum64(str_bytes < 0x10000u ? str_bytes : 0x10000u)) + last;
        const uint64_t gp = g0 + uint64_t(e);
        if (lane == 0) {
            P.records[slot] = make_uint4(uint32_t(gp), uint32_t(gp >> 32), model | ((bytes < kTokSuffixLong ? bytes : kTokSuffixLong) << kTokSuffixShift), last);
            if (P.model_out) P.model_out[gp] = int32_t(model) - 1;
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// add the weights of the collected matches: lanes over (match, score) pairs, four matches per 64 lanes, the loads of sixteen
// matches in flight together
__device__ __forceinline__ void tag_add_matches(const TagParams& P, TagWaveLds& L, uint32_t n_match, int lane) {
    __builtin_amdgcn_wave_barrier();
    for (uint32_t k0 = 0; k0 < n_match; k0 += 16) {
        int32_t w[4];
        uint32_t info[4];
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            const uint32_t k = k0 + q * 4u + (uint32_t(lane) >> 4), i = uint32_t(lane) & 15u;
            info[q] = k < n_match ? L.f.mlist[k][1] : 0u;
            w[q] = (k < n_match && i < (info[q] >> 8)) ? P.weights[L.f.mlist[k][0] + i] : 0;
        }
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            const uint32_t i = uint32_t(lane) & 15u;
            if (i < (info[q] >> 8)) atomicAdd(&L.f.zt[info[q] & 0xFFu][i], w[q]);   // several n-grams of one token may match
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// A pass over the `nq` queued tokens (they come from any of the wave's sentences): their context chars, model records and
// bias are fetched in ONE trip -- lanes over (token, char), tokens, (token, score) -- then the n-grams of ALL of them are
// checked in rounds of 64 (token, n-gram) pairs, the weights of the matches added, the argmax taken.
__device__ __forceinline__ void tag_pass(const TagParams& P, TagWaveLds& L, uint32_t nq, int lane) {
    constexpr uint32_t dbg = 0;
    const uint32_t nt = P.n_tags;
    const uint64_t below_me = (uint64_t(1) << lane) - 1;
    __builtin_amdgcn_wave_barrier();
    uint32_t cw[kTagPass * kCtx / 64];
    int32_t bias[kTagPass * kTagFastZ / 64];
    static_assert(kCtx == 16 && kTagFastZ == 16, "lanes over (token, 16 entries)");
#pragma unroll
    for (int q0 = 0; q0 < kTagPass * kCtx / 64; ++q0) {
        const uint32_t t = uint32_t(q0) * 4u + (uint32_t(lane) >> 4);
        const int k = (lane & 15) - kCtxBack;
        const uint32_t clip = L.f.tok[t][3];
        const bool have = t < nq && k >= -int(clip & 0xFFu) && k <= int(clip >> 8);
        const uint64_t gp = uint64_t(L.f.tok[t][1]) | (uint64_t(L.f.tok[t][2]) << 32);
        cw[q0] = have ? P.cps[gp + uint64_t(int64_t(k))] : 0u;
        bias[q0] = (t < nq && !(dbg & 32u)) ? int32_t(P.mfilt[size_t(L.f.tok[t][0] - 1) * kTagFiltStride + 12u + (uint32_t(lane) & 15u)]) : 0;
    }
    uint4 f0 = make_uint4(0, 0, 0, 0), f1 = f0, f2 = f0;
    if (uint32_t(lane) < nq) {
        const uint4* fr = reinterpret_cast<const uint4*>(P.mfilt + size_t(L.f.tok[lane][0] - 1) * kTagFiltStride);
        f0 = fr[0]; f1 = fr[1]; f2 = fr[2];
        const uint4 f7 = fr[7];   // slot_str of the model's (up to three) slots: where the strings of their candidates start
        L.f.tok[lane][8] = f7.x; L.f.tok[lane][9] = f7.y; L.f.tok[lane][10] = f7.z;
    }
#pragma unroll
    for (int q0 = 0; q0 < kTagPass * kCtx / 64; ++q0) {
        const uint32_t t = uint32_t(q0) * 4u + (uint32_t(lane) >> 4);
        L.f.ctx[t][lane & 15] = cw[q0];
        L.f.zt[t][lane & 15] = bias[q0];
    }
    __builtin_amdgcn_wave_barrier();
    // the char entries of a model come in groups by rel_position r: the n-grams of group r END at char p + r, and the group's
    // filter says which chars they end with -- a group the text cannot match is not enumerated at all
    uint32_t m_count = 0;
    if (uint32_t(lane) < nq) {
        const uint32_t flo[4] = {f0.x, f0.z, f1.x, f1.z}, fhi[4] = {f0.y, f0.w, f1.y, f1.w};
        uint32_t act = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t c = L.f.ctx[lane][kCtxBack + r] & kCharMask;   // 0 past the sentence's end
            const uint32_t bit = packed_filter_bit(c);
            const bool on = P.use_char != 0 && c != 0 && c < 0xFFFFu && (((bit < 32 ? flo[r] >> bit : fhi[r] >> (bit - 32)) & 1u) != 0);
            if (on) { act |= 1u << r; m_count += (f2.x >> (8 * r)) & 0xFFu; }
        }
        const uint32_t tc = P.use_type != 0 ? (f2.z & 0xFFu) : 0u;
        m_count += tc;
        const uint32_t n_slots = ((f2.z >> 16) & 0xFFu) < nt ? ((f2.z >> 16) & 0xFFu) : nt;
        L.f.tok[lane][4] = f2.y; L.f.tok[lane][5] = ((f2.z >> 8) & 0xFFu) | (n_slots << 8) | (tc << 16) | (act << 24);
        L.f.tok[lane][6] = f2.w; L.f.tok[lane][7] = f2.x;
    }
    const uint32_t incl = wave_inclusive_scan(m_count);
    const uint32_t total = (dbg & 4u) ? 0u : uint32_t(__builtin_amdgcn_readlane(int(incl), 63));
    if (uint32_t(lane) < nq) L.f.pref[lane] = incl - m_count;
    if (uint32_t(lane) == nq) L.f.pref[nq] = total;
    __builtin_amdgcn_wave_barrier();
    uint32_t n_acc = 0;   // matches waiting in mlist
    // (the rounds software-pipelined -- the records of round r + 1 asked for before round r is compared -- changed nothing: 0.2551 against 0.2569 ms
    // on configs[4], profiles/r06_p_*; by its instruction counts the launch is nearer its vector ALUs' limit than its trips')
    for (uint32_t r0 = 0; r0 < total; r0 += 64) {
        const uint32_t pi = r0 + uint32_t(lane);
        const bool have = pi < total;
        uint32_t t = 0, ri = 0;
        if (have) {   // the token of this pair: the last t with pref[t] <= pi
            uint32_t lo = 0, hi = nq;   // pref[lo] <= pi < pref[hi]
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (L.f.pref[mid] <= pi) lo = mid; else hi = mid; }
            t = lo;
            // the k-th entry of the token's ACTIVE groups (then its type entries) -> record index
            uint32_t k = pi - L.f.pref[t], off = L.f.tok[t][4];
            const uint32_t groups = L.f.tok[t][7], act = L.f.tok[t][5] >> 24;
            bool placed = false;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t c = (groups >> (8 * r)) & 0xFFu;
                if (!placed && ((act >> r) & 1u)) { if (k < c) { ri = off + k; placed = true; } else k -= c; }
                off += c;
            }
            if (!placed) ri = off + k;   // a type entry
        }
        const uint4 ra = have ? reinterpret_cast<const uint4*>(P.nrec)[size_t(ri) * 2] : make_uint4(0, 0, 0, 0);
        const uint4 rb = have ? reinterpret_cast<const uint4*>(P.nrec)[size_t(ri) * 2 + 1] : make_uint4(0, 0, 0, 0);
        // the longest n-gram of this round bounds the compare loop (wave-uniform: no lane runs 12 steps for 3-char n-grams)
        uint32_t gmax = wave_max(ra.x & 0xFFu);
        gmax = gmax < kTagFastSyms ? gmax : kTagFastSyms;
        const uint32_t glen = ra.x & 0xFFu, rel = (ra.x >> 8) & 0xFFu, kind = (ra.x >> 16) & 1u;
        const uint32_t clip = have ? L.f.tok[t][3] : 0u;
        // chars [p + rel + 1 - glen, p + rel + 1) of the sentence: inside it, and inside the context (glen <= 12, rel <= 3)
        bool ok = have && glen != 0 && glen <= rel + 1u + (clip & 0xFFu) && rel <= (clip >> 8) && (kind == 0 ? P.use_char != 0 : P.use_type != 0);
        // the record's symbols as a 192-bit shift register: the next one is always the low 16 bits
        uint32_t s0w = ra.z, s1w = ra.w, s2w = rb.x, s3w = rb.y, s4w = rb.z, s5w = rb.w;
        const uint32_t wbeg = (uint32_t(kCtxBack) + 1u + rel - glen) & uint32_t(kCtx - 1);
        for (uint32_t j = 0; j < gmax; ++j) {
            if (j < glen && ok) {
                const uint32_t c = L.f.ctx[t][(wbeg + j) & uint32_t(kCtx - 1)];
                ok = (s0w & 0xFFFFu) == (kind == 0 ? (c & kCharMask) : (c >> 24));
            }
            s0w = (s0w >> 16) | (s1w << 16); s1w = (s1w >> 16) | (s2w << 16); s2w = (s2w >> 16) | (s3w << 16);
            s3w = (s3w >> 16) | (s4w << 16); s4w = (s4w >> 16) | (s5w << 16); s5w >>= 16;
        }
        const uint64_t m0 = __ballot(ok);
        if (ok) {
            const uint32_t k = n_acc + uint32_t(__popcll(m0 & below_me));
            const uint32_t zlen = L.f.tok[t][5] & 0xFFu;
            const uint32_t wl = (ra.x >> 24) < zlen ? (ra.x >> 24) : zlen;   // zip: the shorter of the two (predictor.rs:82-89)
            L.f.mlist[k][0] = ra.y; L.f.mlist[k][1] = t | (wl << 8);
        }
        n_acc += uint32_t(__popcll(m0));
        if (n_acc > uint32_t(kMatchCap - 64)) {   // the next round may not fit
            if (!(dbg & 8u)) tag_add_matches(P, L, n_acc, lane);
            n_acc = 0;
        }
    }
    if (n_acc != 0 && !(dbg & 8u)) tag_add_matches(P, L, n_acc, lane);
    __builtin_amdgcn_wave_barrier();
    if (P.scores_out) {   // wave-uniform.  Predictor::store_tag_scores: lanes over (token, score), 4 tokens per 64 lanes
#pragma unroll
        for (int q0 = 0; q0 < kTagPass * int(kTagFastZ) / 64; ++q0) {
            const uint32_t t = uint32_t(q0) * 4u + (uint32_t(lane) >> 4), i = uint32_t(lane) & 15u;
            if (t < nq && i < (L.f.tok[t][5] & 0xFFu) && i < P.score_stride) {
                const uint64_t gp = uint64_t(L.f.tok[t][1]) | (uint64_t(L.f.tok[t][2]) << 32);
                P.scores_out[gp * P.score_stride + i] = L.f.zt[t][i];
            }
        }
    }
    // argmax per (token, slot) (TagPredictor::predict, predictor.rs:286-304); the bytes the token's tags take in the tokenized text
    // ("/tag" per slot up to the last Some, sentence.rs:866-881) are summed on the way for the writer (layout.h, tok_model words)
    if (uint32_t(lane) < nq) { L.f.pref[lane] = 0; L.f.zt[lane][kTagFastZ] = 0; }   // bytes of the Some slots' strings / last Some + 1
    __builtin_amdgcn_wave_barrier();
    for (uint32_t a0 = 0; a0 < ((dbg & 16u) ? 0u : nq * nt); a0 += 64) {
        const uint32_t a = a0 + uint32_t(lane);
        if (a < nq * nt) {
            const uint32_t t = a / nt, j = a - t * nt;
            const uint32_t info = L.f.tok[t][5], zlen = info & 0xFFu, n_slots = (info >> 8) & 0xFFu;
            const uint32_t ps = j < n_slots ? L.f.tok[t][6] >> (9 * j) : 0u, cnt = ps & 31u, off = (ps >> 5) & 15u;
            int32_t tag = cnt == 1 ? 0 : -1;   // None: no candidates, or a slot this model does not have
            if (cnt >= 2) {
                int32_t best = INT32_MIN;
                tag = 0;
                for (uint32_t c = 0; c < cnt && off + c < zlen; ++c) {
                    const int32_t v = L.f.zt[t][off + c];
                    if (v > best) { best = v; tag = int32_t(c); }
                }
            }
            P.rec_tags[size_t(L.f.tok[t][11]) * nt + j] = tag;
            if (P.tags) {
                const uint64_t gp = uint64_t(L.f.tok[t][1]) | (uint64_t(L.f.tok[t][2]) << 32);
                P.tags[gp * nt + j] = tag;
            }
            uint2 str = make_uint2(0u, 0u);
            if (tag >= 0) {
                const uint32_t k = L.f.tok[t][8 + j] + uint32_t(tag);   // (j < 3: the record form)
                if (k < P.n_strings) str = make_uint2(P.str_off[k], P.str_off[k + 1] - P.str_off[k]);
                atomicAdd(&L.f.pref[t], str.y < 0x10000u ? str.y : 0x10000u);
                atomicMax(reinterpret_cast<uint32_t*>(&L.f.zt[t][kTagFastZ]), j + 1u);
            }
            P.rec_str[size_t(L.f.tok[t][11]) * nt + j] = str;
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (uint32_t(lane) < nq) {   // the token's record: where the writer finds its tags
        const uint32_t bytes = L.f.pref[lane] + uint32_t(L.f.zt[lane][kTagFastZ]);
        P.records[L.f.tok[lane][11]] = make_uint4(L.f.tok[lane][1], L.f.tok[lane][2],
                                                  L.f.tok[lane][0] | ((bytes < kTokSuffixLong ? bytes : kTokSuffixLong) << kTokSuffixShift), uint32_t(L.f.zt[lane][kTagFastZ]));
        if (P.model_out) P.model_out[uint64_t(L.f.tok[lane][1]) | (uint64_t(L.f.tok[lane][2]) << 32)] = int32_t(L.f.tok[lane][0]) - 1;
    }
    __builtin_amdgcn_wave_barrier();
}

// The LOOKUPS, a launch of their own (round 6; the front end used to do them itself, 64 candidates at a time, and spent most of its time
// waiting for them: a lookup is a chain of dependent trips -- the token's chars, the table slot, a longer token's symbols -- in ONE wave
// with nothing else to run; profiles/r06_c_tag_ablations.jsonl: the front end without its candidates took 0.12 of its 0.65 ms).  Here a
// wave takes a run's candidates, a lane each, and there are as many waves as the device holds: the trips overlap.  A candidate that has a
// tag model joins the queue of the passes -- record-form models from its front, the others from its back, a wave's share with ONE 64-bit
// atomic -- with the record it will fill; one that has none gets its (empty) record here.
constexpr uint32_t kResolveRuns = 8;   // runs a wave takes together: their candidates, back to back, keep its lanes busy (a run has some forty)
__global__ __launch_bounds__(kTagThreads) void tag_resolve_kernel(const TagParams P) {
    __shared__ uint64_t PREF[kTagWaves][kResolveRuns + 1], RUN0[kTagWaves][kResolveRuns];
    const int lane = threadIdx.x & 63;
    const uint32_t wid = wave_uniform(threadIdx.x >> 6);
    const uint64_t wave = uint64_t(blockIdx.x) * kTagWaves + wid, n_waves = uint64_t(gridDim.x) * kTagWaves;
    for (uint64_t r_first = wave * kResolveRuns; r_first < P.n_runs; r_first += n_waves * kResolveRuns) {
        const uint32_t nr = uint32_t(P.n_runs - r_first < kResolveRuns ? P.n_runs - r_first : kResolveRuns);
        __builtin_amdgcn_wave_barrier();
        // the runs' first records (= their candidates' numbers: candidate c IS record c) and first chars, one trip for all of them
        if (uint32_t(lane) <= nr) PREF[wid][lane] = P.run_pref[r_first + uint32_t(lane)];
        if (uint32_t(lane) < nr) { const uint64_t i_a = (r_first + uint32_t(lane)) * P.run_sent; RUN0[wid][lane] = P.ooff[i_a] + i_a; }
        __builtin_amdgcn_wave_barrier();
        const uint64_t c_lo = PREF[wid][0], c_hi = PREF[wid][nr];
        for (uint64_t c0 = c_lo; c0 < c_hi; c0 += 64) {
            const uint64_t c = c0 + uint32_t(lane);
            const bool have = c < c_hi;
            uint32_t j = 0;   // the candidate's run: the last one whose first record is not behind it
#pragma unroll
            for (uint32_t q = 1; q < kResolveRuns; ++q) j += (q < nr && PREF[wid][q] <= c) ? 1u : 0u;
            const uint64_t run0 = RUN0[wid][j];
            const uint4 e = have ? P.cands[run0 + (c - PREF[wid][j])] : make_uint4(0, 1, 0, 0);
            const uint64_t gp = run0 + e.x;
            bool fast = false;
            const uint32_t model = have ? find_tag_model(P, P.cps + (gp + 1 - e.y), e.y, &fast) : 0u;
            // the record, for now: {last char, context clip, tag model + 1 | record form << 31, 0}; the passes make it the token's record.
            // No tag model: that IS the (empty) record.  (A queue of the tokens that have one, filled with an atomic per 64 candidates,
            // stood here first: 44 K atomics on one word took most of the launch's 0.48 ms, profiles/r06_e_*.)
            if (have) P.records[c] = make_uint4(uint32_t(gp), model ? e.z : 0u, model | (fast ? 0x80000000u : 0u), 0u);
        }
    }
}

// The passes: the waves stride over the RECORDS the lookups left, 64 at a time; those whose token has a tag model of the record form are
// gathered, in order, into the wave's queue in LDS -- 16 of them are a pass --, the others' tokens go through the whole-wave routine one at a
// time.  Nothing is allocated, nothing is counted: a record is overwritten in place by what its token's tags turn out to be.  The grid
// is what the device holds.
__global__ __launch_bounds__(kTagThreads, kTagPairOcc) void tag_pass_kernel(const TagParams P) {
    __shared__ TagWaveLds LDS[kTagWaves];
    const int lane = threadIdx.x & 63;
    const uint32_t wid = wave_uniform(threadIdx.x >> 6);
    TagWaveLds& L = LDS[wid];
    const uint64_t below_me = (uint64_t(1) << lane) - 1;
    const uint64_t wave = uint64_t(blockIdx.x) * kTagWaves + wid, n_waves = uint64_t(gridDim.x) * kTagWaves;
    const uint64_t n_rec = wave_uniform64(P.run_pref[P.n_runs]);
    uint32_t nq = 0;
    for (uint64_t c0 = wave * 64; c0 < n_rec; c0 += n_waves * 64) {
        const uint64_t c = c0 + uint32_t(lane);
        const uint4 e = c < n_rec ? P.records[c] : make_uint4(0, 0, 0, 0);
        const uint32_t model = e.z & 0x7FFFFFFFu;
        const bool fast = (e.z >> 31) != 0;
        const uint64_t qmask = __ballot(model != 0 && fast);
        uint64_t todo = __ballot(model != 0 && !fast);
        if (qmask != 0) {
            const uint32_t rank = uint32_t(__popcll(qmask & below_me));
            uint32_t remaining = uint32_t(__popcll(qmask)), done = 0;
            for (;;) {
                const uint32_t room = uint32_t(kTagPass) - nq, take = remaining < room ? remaining : room;
                if (model != 0 && fast && rank >= done && rank < done + take) {
                    const uint32_t row = nq + rank - done;
                    L.f.tok[row][0] = model; L.f.tok[row][1] = e.x; L.f.tok[row][2] = 0u; L.f.tok[row][3] = e.y; L.f.tok[row][11] = uint32_t(c);
                }
                nq += take; done += take; remaining -= take;
                if (nq < uint32_t(kTagPass)) break;
                tag_pass(P, L, nq, lane);
                nq = 0;
                if (!remaining) break;
            }
        }
        while (todo) {   // wave-uniform: the whole wave, one token at a time; its sentence is looked up in the offsets (rare models)
            const int k = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const uint32_t mk = uint32_t(__builtin_amdgcn_readlane(int(model), k));
            const uint64_t gk = uint64_t(uint32_t(__builtin_amdgcn_readlane(int(e.x), k)));
            uint64_t lo = 0, hi = P.n_sent;   // the last sentence i with ooff[i] + i <= gk
            while (hi - lo > 1) {
                const uint64_t mid = (lo + hi) >> 1;
                if (P.ooff[mid] + mid <= gk) lo = mid; else hi = mid;
            }
            const uint64_t g0 = P.ooff[lo] + lo;
            // (the routine's scratch is the union's other member: the queue's rows stay as they are only while it is empty -- so run the waiting pass first)
            if (nq != 0) { tag_pass(P, L, nq, lane); nq = 0; }
            tag_token_by_wave(P, P.cps + g0, int64_t(P.ooff[lo + 1] - P.ooff[lo]) + 1, int64_t(gk - g0), g0, mk, VPT_TO_LDS_PTR(volatile int32_t, L.z), lane, c0 + uint32_t(k));
        }
    }
    if (nq != 0) tag_pass(P, L, nq, lane);
}

// The front end, FLAT over the batch's chars (round 4; a wave per sentence ran its steps of 128 chars half empty at a sentence's end --
// configs[4]'s sentences of 8 .. 512 chars: 2.5 steps where 2.0 would do -- and started every sentence with its own trips for offsets
// and first chars).  A wave takes a RUN of `per` consecutive sentences, which are consecutive chars (char q of sentence i sits at
// ooff[i] + i + q), and walks them in steps of two 64-char half-steps with every lane busy up to the run's last step.  What the per-sentence loop knew from its loop variables comes from two bitmaps per
// step, built a step ahead from the run's offsets (64 sentences' worth in the lanes at a time): SM, the chars that start a sentence, and EM,
// the chars that end one -- a sentence's last char ends a token (predictor.rs:563-570), the label of any other char q of sentence i is
// labels[flat(q) - i], and a candidate's context stops at its sentence's ends.  The token logic (ends, Unknown, filter, candidates,
// lookups, queue in HBM) works in run-relative positions.
//
// Round 5, first the same steps in fewer vector instructions (358 of them per step, two thirds of the SIMDs' issue slots by the counters:
// profiles/r05_z2_c4_summary.txt; SQ_INSTS_VALU 372 M -> 162 M per launch of configs[4] -- which bought 4 %: the instructions were not the
// bound, see kSum below) --
//   * every global access of a step is a wave-uniform pointer (advanced per step in scalar registers) plus a lane offset that does
//     not change: no 64-bit address arithmetic in the lanes;
//   * ONE bitmap of marks per step (sentence starts and the run's end; EM is SM shifted down by one), set by a scalar loop over the few
//     window entries that fall into the step (LDS atomics only when there are more than six);
//   * which token ends are valid (no Unknown label inside, predictor.rs:566-567) is scalar bit arithmetic: adding the positions just
//     above the Unknown labels to the mask of unlabelled positions carries each one up to the next label -- the token ends reached
//     that way are the tainted ones;
//   * counts of mask bits below a lane through v_mbcnt, lane predicates straight from scalar masks;
//   * the ring keeps its first three words again behind its end, so a token's first four chars are four consecutive LDS words;
//   * the one or two candidates of a half-step are put into their rows by scalar code (context clip from the bitmaps, length from the lane).
// Round 6: the kernel STORES NOTHING but the queue entries of the tokens that have a tag model (it used to write the None entries of every
// char -- two thirds of its stores, 0.30 of its 0.75 ms by the ablations -- and the writer's token word per char: the output is the records now).
#ifndef VPT_TAG_FLAT_OCC
#define VPT_TAG_FLAT_OCC 6     // waves per SIMD the flat front end is compiled for (A/B builds: -D); 62 VGPRs either way, the scalar registers decide
#endif
#ifndef VPT_TAG_RUN_CHARS
#define VPT_TAG_RUN_CHARS 2048 // chars of a run of sentences
#endif
// -DVPT_TAG_PROFILE (diagnostic builds, tools/build_variants.sh): where a wave of the flat front end spends its time -- shader-clock ticks per
// part of a step, summed over the waves (printed by the NEXT launch: g_tag_prof is read back before it is cleared)
#ifndef VPT_TAG_ABLATE
#define VPT_TAG_ABLATE 0   // timing ablations of the flat front end (wrong results): 1 no filter loads, 2 no candidates, 4 no None stores, 8 no label loads, 16 no ring / keys, 32 filter words loaded but no candidates, 64 candidates queued but never looked up
#endif
#ifdef VPT_TAG_PROFILE
__device__ unsigned long long g_tag_prof[16];
#define VPT_TP_DECL unsigned long long tp_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long tp_t = __builtin_amdgcn_s_memtime(); const unsigned long long tp_t00 = tp_t
#define VPT_TP(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tp_[k] += t_ - tp_t; tp_t = t_; } while (0)
#define VPT_TP_COUNT(k) (tp_[k] += 1)
#define VPT_TP_FLUSH() do { tp_[11] = __builtin_amdgcn_s_memtime() - tp_t00; if (lane == 0) for (int k_ = 0; k_ < 12; ++k_) atomicAdd(&g_tag_prof[k_], tp_[k_]); } while (0)
#else
#define VPT_TP_DECL ((void)0)
#define VPT_TP(k) ((void)0)
#define VPT_TP_COUNT(k) ((void)0)
#define VPT_TP_FLUSH() ((void)0)
#endif
__device__ __forceinline__ void read_tag_params(TagParams& Q, VPT_KARG(TagParams) R) { __builtin_memcpy(&Q, R, sizeof(Q)); }   // (for the lookups: the whole block)
__device__ __forceinline__ uint32_t mask_bits_below_lane(uint64_t mask, uint32_t plus) {   // of a wave-uniform mask: v_mbcnt_lo / _hi
    return __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), plus));
}
// kSum: a SUMMARY of the token table's filter sits in LDS (2^18 bits, one for every 2^(tok_bits + 5 - 18) of the filter's: set when any of
// them is; built by tag_filter_summary_kernel in front of this launch, into TagParams::summary).  The filter word of a token
// end is a random 4-byte load that hits the L2 -- 25 of them per half-step, 52 M per launch of configs[4], a third of the kernel's time at the
// rate the vector L1 takes random lanes (profiles/r05_y_tag_front_ablations.txt: keys + filter loads 0.32 ms, None stores 0.30 ms, the rest
// 0.09 + 0.09, and they ADD) -- and 96 of 100 answers are "no".  With 50 000 tag models the summary is a sixth full: five of six of those
// loads are not issued.  Workgroups of 8 waves share one copy (32 KB; 3 workgroups per CU).
constexpr int kFlatThreads = 512, kFlatWaves = kFlatThreads / 64;
#ifndef VPT_TAG_SUM_LOG2
#define VPT_TAG_SUM_LOG2 18   // (tests build a small one: a summary bit then stands for many filter bits with the test models' small tables, too)
#endif
constexpr uint32_t kSumLog2 = VPT_TAG_SUM_LOG2;
static_assert(kSumLog2 >= 5 && kSumLog2 <= 18, "the summary's words in LDS");
__global__ __launch_bounds__(256) void tag_filter_summary_kernel(const uint32_t* __restrict__ filt, const uint32_t big_log2, const uint32_t sum_log2,
                                                                   uint32_t* __restrict__ out) {
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;   // a word of the summary
    if (w >= (1u << (sum_log2 - 5u))) return;
    const uint32_t c = big_log2 - sum_log2;   // a summary bit stands for 2^c filter bits
    uint32_t r = 0;
    for (uint32_t j = 0; j < 32u; ++j) {
        const uint32_t sb = 32u * w + j;
        bool any = false;
        if (c >= 5u) {
            for (uint32_t k = sb << (c - 5u); k < (sb + 1u) << (c - 5u); ++k) any = any || filt[k] != 0;
        } else {
            const uint32_t b0 = sb << c;   // (2^c <= 16 bits that start at a multiple of 2^c: inside one word)
            any = ((filt[b0 >> 5] >> (b0 & 31u)) & ((1u << (1u << c)) - 1u)) != 0;
        }
        r |= uint32_t(any) << j;
    }
    out[w] = r;
}
template <bool kSum>
__global__ __launch_bounds__(kFlatThreads, VPT_TAG_FLAT_OCC) void tag_front_flat_kernel(const TagParams P_in, const uint32_t sum_log2) {
    VPT_KARG(TagParams) P = VPT_KARG_PTR(TagParams, P_in);   // (read where it is used, like the scoring kernel's block: device_common.h)
    __shared__ TagFrontLds FR[kFlatWaves];
    __shared__ uint32_t BITS[kFlatWaves][8];   // per wave: the marks of the step being prepared, when there are many (5 words)
    __shared__ uint32_t SUM[kSum ? (1u << (kSumLog2 - 5u)) : 1u];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wid = wave_uniform(threadIdx.x >> 6);
    TagFrontLds& L = FR[wid];
    uint32_t* const bits = BITS[wid];
    const uint64_t wave = uint64_t(blockIdx.x) * kFlatWaves + wid, n_waves = uint64_t(gridDim.x) * kFlatWaves;
    if constexpr (kSum) {
        const uint32_t* const summary = P->summary;
        for (uint32_t i = threadIdx.x; i < (1u << (sum_log2 - 5u)); i += uint32_t(kFlatThreads)) SUM[i] = summary[i];
        __syncthreads();
    }
    const uint32_t sshift = 32u - sum_log2;
    const uint64_t below_me = (uint64_t(1) << lane) - 1;
    const uint64_t total_b = P->total_chars - P->n_sent;   // labels of the batch
    const uint32_t per = P->run_sent;
    const uint64_t n_runs = P->n_runs;
    VPT_TP_DECL;
    for (uint64_t run = wave; run < n_runs; run += n_waves) {
        VPT_TP(0);
        VPT_KARG_FENCE(P);
        const uint64_t i_a = run * per, i_b = i_a + per < P->n_sent ? i_a + per : P->n_sent;
        const uint64_t o_a = wave_uniform64(P->ooff[i_a]);
        const uint64_t run0 = o_a + i_a, run1 = wave_uniform64(P->ooff[i_b]) + i_b;   // flat chars [run0, run1)
        // offsets that do not fit the batch are reported by decode_chars_kernel / the scoring kernel; such a run (and one of 2^31 chars:
        // not in this kernel's index width) is left alone
        if (run1 < run0 || run1 > P->total_chars || run1 - run0 >= 0x7FFFFF00ull) continue;
        const uint32_t n = uint32_t(run1 - run0);
        // the run's arrays (kept in scalar registers, or spilled to a lane: either is cheaper in a step than asking the constant cache again --
        // a step is a chain of waits, not of instructions)
        const uint32_t* const cps = P->cps + run0;
        uint4* const cands = P->cands + run0;   // the run's candidates: as many places as it has chars
        uint32_t n_cand = 0;                    // (wave-uniform)
        const uint8_t* const lab_run = P->labels + o_a;
        const uint32_t tok_bits = P->tok_bits, fshift = 32u - tok_bits - kTagFilterLog2;
        const uint32_t* const filt = P->tok_tab + (size_t(4) << tok_bits);
        // the label of char q of the run, in sentence i_a + k: labels[o_a + q - k]; nothing past the batch's last label is read whatever
        // the offsets say (d_max: the furthest q - k may go)
        const bool labels_here = total_b != 0 && o_a < total_b;
        const uint64_t d_max = labels_here ? total_b - 1 - o_a : 0;
        // ---- the run's sentence starts, run-relative, 64 at a time in the lanes: entry k of the window is sentence i_load + k; the entry of
        // i_b is the run's end; entries past it are "never"
        constexpr uint32_t kNever = 0x7FFFFFFFu;
        uint64_t i_load = i_a;
        auto load_window = [&]() -> uint32_t {
            const uint64_t i = i_load + uint64_t(lane);
            if (i > i_b) return kNever;
            const uint64_t f = P->ooff[i] + i;
            return (f >= run0 && f <= run1) ? uint32_t(f - run0) : kNever;   // (offsets out of order: reported elsewhere; nothing is marked)
        };
        uint32_t win = load_window();
        // MARKS of the chars [base, base + 128]: a sentence starts there, or the run ends there.  (base + 128 is the next step's first
        // char: its mark is the end of this step's last one.)
        uint64_t mk0 = 0, mk1 = 0;
        uint32_t mk2 = 0;
        auto prepare = [&](uint32_t base) {
            mk0 = 0; mk1 = 0; mk2 = 0;
            for (;;) {
                const uint32_t d = win - base;   // an entry in front of the step, or "never": far above 128
                uint64_t hit = __ballot(d <= 128u);
                if (hit != 0) {
                    if (__popcll(hit) <= 6) {
                        do {
                            const int k = __ffsll((long long)hit) - 1;
                            hit &= hit - 1;
                            const uint32_t dk = uint32_t(__builtin_amdgcn_readlane(int(d), k));
                            if (dk < 64u) mk0 |= uint64_t(1) << dk;
                            else if (dk < 128u) mk1 |= uint64_t(1) << (dk - 64u);
                            else mk2 = 1u;
                        } while (hit != 0);
                    } else {
                        if (lane < 5) bits[lane] = 0;
                        __builtin_amdgcn_wave_barrier();
                        if (d <= 128u) atomicOr(&bits[d >> 5], 1u << (d & 31u));
                        __builtin_amdgcn_wave_barrier();
                        mk0 |= uint64_t(wave_uniform(bits[0])) | (uint64_t(wave_uniform(bits[1])) << 32);
                        mk1 |= uint64_t(wave_uniform(bits[2])) | (uint64_t(wave_uniform(bits[3])) << 32);
                        mk2 |= wave_uniform(bits[4]) & 1u;
                        __builtin_amdgcn_wave_barrier();
                    }
                }
                // the window is used up when its last entry lies in front of the next step (an entry AT the next step's first char has
                // marked the end in front of it here and marks its start there) -- as long as sentences are left
                const uint32_t last = uint32_t(__builtin_amdgcn_readlane(int(win), 63));
                if (last == kNever || last >= base + 128u || i_load + 64 > i_b) break;
                i_load += 64;
                win = load_window();
            }
        };
        uint32_t cs = 0;   // marks of the run in front of the step being prepared (= sentence starts: the end's mark is the last one)
        uint32_t c_next[2], b_next[2];
        auto fetch = [&](uint32_t base) {   // the prepared step's chars and labels (1 where a sentence ends, 0 past the run)
            // SM, the chars that start a sentence (and the run's end), are the marks; EM, the chars that end one, the marks one down
            const uint64_t sm_n[2] = {mk0, mk1}, em_n[2] = {(mk0 >> 1) | (mk1 << 63), (mk1 >> 1) | (uint64_t(mk2) << 63)};
            const uint32_t rem = n - base;
            const uint32_t* const cstep = cps + base;
            // char q of the step sits in sentence i_a + cs - 1 + (marks of the step at or in front of q): its label is
            // labels[o_a + (base - cs) + q + 1 - those]
            const uint64_t ahead = uint64_t(base - cs);   // (every mark in front of the step is a char in front of it: cs <= base)
            const bool lab = labels_here && d_max >= ahead;
            const uint64_t room = lab ? d_max - ahead : 0;
            const uint32_t lim = room < 255u ? uint32_t(room) : 255u;
            const uint8_t* const lstep = lab_run + ahead;
#pragma unroll
            for (uint32_t h = 0; h < 2; ++h) {
                const uint32_t q = lane + 64u * h;
                const bool in_run = q < rem;
                // marks at or in front of the lane = bit 0 + the bits of (mask >> 1) below the lane
                const uint32_t u = 64u * h + 1u - uint32_t(sm_n[h] & 1u) - (h ? uint32_t(__popcll(sm_n[0])) : 0u);
                uint32_t voff = lane + u - mask_bits_below_lane(sm_n[h] >> 1, 0u);
                voff = voff < lim ? voff : lim;
                const bool last = __builtin_amdgcn_inverse_ballot_w64(em_n[h]);
                c_next[h] = in_run ? cstep[q] : 0u;
                b_next[h] = !in_run ? 0u : (last || !lab) ? 1u : (VPT_TAG_ABLATE & 8) ? (voff & 3u) == 0 : uint32_t(lstep[voff]);
            }
            cs += uint32_t(__popcll(mk0)) + uint32_t(__popcll(mk1));
        };
        prepare(0);
        fetch(0);
        // (the first step's chars and labels are waited for HERE: at the loop's head the wait would be merged with the back edge's, where
        // only stores are outstanding, into a wait for everything -- a step would start with a wait for the write acknowledgements)
        VPT_PIN(c_next[0]); VPT_PIN(c_next[1]); VPT_PIN(b_next[0]); VPT_PIN(b_next[1]);
        VPT_TP(1);   // a run's start: its offsets, the first window, the first step's chars and labels
        VPT_TP_COUNT(9);
        int start = 0;              // where the token that is open at the beginning of this half-step started
        uint32_t tainted = 0;       // 1: an Unknown boundary has been seen inside it (predictor.rs:566-567)
        uint64_t sm_prev = 0;       // SM of the half-step in front of the current step
        uint32_t base = 0;
        auto step = [&](auto full_step) {   // the chars [base, base + 128) of the run; full_step: all of them are (every step but a run's last)
            constexpr bool kFull = decltype(full_step)::value;
            const uint32_t c[2] = {c_next[0], c_next[1]}, b[2] = {b_next[0], b_next[1]};
            const uint64_t sm[2] = {mk0, mk1};   // this step's marks (the next step's are prepared below)
            const uint32_t sm_top = mk2;         // ... and the mark of the char behind it
            {   // the ring: char x at txt[x & 255], and chars = 0, 1, 2 (mod 256) once more behind its end
                const uint32_t r = (base & 128u) + lane;
                L.txt[r] = c[0];
                L.txt[r + 64u] = c[1];
                if ((base & 128u) == 0 && lane < 3u) L.txt[kRing + lane] = c[0];
            }
            const bool more = kFull && base + 128u < n;
            VPT_TP(2);   // (what is left of the step before, the ring)
            if (more) { prepare(base + 128u); fetch(base + 128u); }
            VPT_TP(3);   // the next step's marks; its loads issued
            __builtin_amdgcn_wave_barrier();
            // ---- (1) this lane's tokens, if its chars end one: [s0, p], valid when no Unknown lies inside.  Could they have a tag model?
            uint32_t len[2], fbit[2], fword[2];
            uint64_t valid[2];
            bool want[2];
#pragma unroll
            for (uint32_t h = 0; h < 2; ++h) {
                const uint32_t hb = base + 64u * h;
                const uint64_t E = __ballot(b[h] == 1u), U = __ballot(b[h] == 2u);
                // a carry started just above an Unknown label (and at the half-step's first char when the open token has one already)
                // runs up the unlabelled positions into the next label: the token ends reached are the tainted ones
                const uint64_t M = ~(E | U), seed = (U << 1) | tainted;
                valid[h] = E & ~((M + seed) ^ M);
                const uint64_t prev_ends = E & below_me;
                const int s0 = prev_ends ? int(hb) + 64 - __clzll((long long)prev_ends) : start;
                len[h] = hb + lane + 1u - uint32_t(s0);
                const uint32_t* const r = &L.txt[uint32_t(s0) & (kRing - 1)];
                uint32_t c0 = r[0], c1 = r[1], c2 = r[2], c3 = r[3];   // always an LDS read ...
                if (VPT_TAG_ABLATE & 16) { c0 = c[h]; c1 = c2 = c3 = uint32_t(s0); }
                const bool mine = __builtin_amdgcn_inverse_ballot_w64(valid[h]);
                if (mine && s0 < int(base) - 128) {   // ... and for a token that began before the ring (rare) the chars themselves
                    c0 = cps[s0];
                    c1 = len[h] > 1u ? cps[s0 + 1] : 0u; c2 = len[h] > 2u ? cps[s0 + 2] : 0u; c3 = len[h] > 3u ? cps[s0 + 3] : 0u;
                }
                uint32_t lo = (c0 & 0xFFFFu) | (c1 << 16), hi = (c2 & 0xFFFFu) | (c3 << 16);
                if (len[h] < 2u) lo &= 0xFFFFu;
                if (len[h] < 4u) hi &= 0xFFFFu;
                if (len[h] < 3u) hi = 0u;
                const uint32_t key = tag_token_hash_key(lo, hi, len[h]);
                fbit[h] = key >> fshift;
                bool ask = mine;   // is the filter word asked for?  Only where the summary does not say no already
                if constexpr (kSum) {
                    const uint32_t sb = key >> sshift;
                    ask = ask && ((SUM[sb >> 5] >> (sb & 31u)) & 1u) != 0;
                }
                want[h] = ask;
                // the token that stays open into the next half-step
                if (E) {
                    const int last = 63 - __clzll((long long)E);
                    start = int(hb) + last + 1;
                    tainted = ((U >> last) >> 1) != 0 ? 1u : 0u;   // (an Unknown label above the last end)
                } else if (U) {
                    tainted = 1u;
                }
            }
#pragma unroll
            for (uint32_t h = 0; h < 2; ++h) fword[h] = (!(VPT_TAG_ABLATE & 1) && want[h]) ? filt[fbit[h] >> 5] : 0u;
            VPT_TP(4);   // token ends, keys, filter loads issued
            // ---- (2) the candidates -- one or two in a half-step (BASELINE's configs[4]) -- go to the run's places in HBM, in order, for the
            // lookups' launch: the char, the token's length, and its context clip from the bitmaps (chars of the token's sentence in front
            // of / behind its last char: up to the nearest sentence start at or in front of the char, up to the nearest sentence end at
            // or behind it).  Plain stores: nothing here waits for them.  (Finding a step's candidates a step later, so that one wait serves
            // the filter words and the next chars, changed nothing -- 0.479 / 0.487 ms, profiles/r06_e_*: the kernel is bound by the scalar
            // instructions of its mask arithmetic, 235 a step.)
#pragma unroll
            for (uint32_t h = 0; h < 2; ++h) {
                const bool cand = ((fword[h] >> (fbit[h] & 31u)) & 1u) != 0;   // (no filter word where no valid token ends)
                const uint64_t cmask = (VPT_TAG_ABLATE & 2) ? 0 : __ballot(cand);
                if (cmask != 0) {
                    // EM: the marks one down
                    const uint64_t em_h = h ? (sm[1] >> 1) | (uint64_t(sm_top) << 63) : (sm[0] >> 1) | (sm[1] << 63);
                    const uint64_t pm = h ? sm[0] : sm_prev;
                    const uint64_t nm = h ? (more ? (mk0 >> 1) | (mk1 << 63) : 0) : (sm[1] >> 1) | (uint64_t(sm_top) << 63);
                    const uint64_t upto = (uint64_t(2) << lane) - 1;   // bits 0 .. lane
                    const uint64_t at_or_before = sm[h] & upto, at_or_after = em_h & ~(upto >> 1);
                    const uint32_t back_full = at_or_before ? lane - uint32_t(63 - __clzll((long long)at_or_before))
                                                            : pm ? lane + 1u + uint32_t(__clzll((long long)pm)) : 0xFFu;
                    const uint32_t fwd_full = at_or_after ? uint32_t(__ffsll((long long)at_or_after)) - 1u - lane
                                                          : nm ? 64u - lane + uint32_t(__ffsll((long long)nm)) - 1u : 0xFFu;
                    const uint32_t back = back_full < uint32_t(kCtxBack) ? back_full : uint32_t(kCtxBack);
                    const uint32_t fwd = fwd_full < uint32_t(kCtx - 1 - kCtxBack) ? fwd_full : uint32_t(kCtx - 1 - kCtxBack);
                    if (cand) cands[n_cand + uint32_t(__popcll(cmask & below_me))] = make_uint4(base + 64u * h + lane, len[h], back | (fwd << 8), 0u);
                    n_cand += uint32_t(__popcll(cmask));
                }
            }
            sm_prev = sm[1];
            __builtin_amdgcn_wave_barrier();   // the ring is written by the next step
            VPT_TP(7);
        };
        for (; n - base >= 128u; base += 128u) step(std::true_type{});
        if (base < n) step(std::false_type{});
        if (lane == 0) P->run_pref[run + 1] = n_cand;   // (this wave's word alone; the scan behind this launch reads it)
    }
    VPT_TP(0);
    VPT_TP_FLUSH();
}

}  // namespace

hipError_t launch_decode_chars(const uint8_t* text, const uint64_t* boff, const uint64_t* ooff, uint64_t n_sent, uint64_t total_chars,
                               const uint32_t* cinfo, uint32_t* cps, uint8_t* types, uint32_t* status, hipStream_t stream, bool fullwidth) {
    // sentences per workgroup: about 16 K chars (4 pieces of CJK text and more; a char is at least a byte, so the chars bound the bytes from
    // below), at most one per thread
    // (4 K / 8 K / 32 K chars measured the same, profiles/r06_o_*: the launch runs at the rate of its char-table gathers, one L1 miss a char)
    const uint64_t per = std::min<uint64_t>(std::max<uint64_t>((uint64_t(16384) * n_sent + total_chars / 2) / std::max<uint64_t>(total_chars, 1), 1), kTagThreads);
    const uint64_t blocks = (n_sent + per - 1) / per;
    hipLaunchKernelGGL(decode_chars_kernel, dim3(uint32_t(blocks)), dim3(kTagThreads), 0, stream, text, boff, ooff, n_sent, total_chars, cinfo, cps,
                       types, status, uint32_t(per), fullwidth ? 1u : 0u);
    return hipGetLastError();
}

uint32_t tag_run_sentences(uint64_t n_sent, uint64_t total_chars) {
    // runs of about 2 K chars (16 steps), at least a sentence: enough runs for every wave of the grid to even out, a window of offsets every
    // few steps (measured on configs[4], profiles/r04_o_tag_front.jsonl: runs of 2 K / 4 K / 16 K chars 1.761 / 1.781 / 1.970 ms stand-alone);
    // at most 256 sentences (the writer's runs are whole multiples of these: kEmitFlatMaxBlock)
    const uint64_t per = (uint64_t(VPT_TAG_RUN_CHARS) * n_sent + total_chars / 2) / std::max<uint64_t>(total_chars, 1);
    return uint32_t(std::min<uint64_t>(std::max<uint64_t>(per, 1), 256));
}
size_t tag_summary_words() { return size_t(1) << (kSumLog2 - 5u); }

// fill_tags: [summary of the token filter] -> front end (candidates) -> scan (the runs' first records) -> lookups (the records' models) ->
// passes (the records).  The caller has zeroed run_pref and scan_state and set the dense arrays it wants to None.
hipError_t launch_tag_tokens(const TagParams& P, hipStream_t stream) {
#ifdef VPT_TAG_PROFILE
    {
        unsigned long long h[16] = {0};
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_tag_prof), sizeof(h));
        if (h[11]) {
            static const char* names[9] = {"between runs", "run start", "step: ring", "step: prepare + fetch", "step: ends, keys", "step: stores", "step: WAIT filter words", "step: candidates", "lookups"};
            std::fprintf(stderr, "[tag front profile] %llu runs, %llu steps, wave ticks %llu:", h[9], h[10], h[11]);
            for (int k = 0; k < 9; ++k) std::fprintf(stderr, " %s %.1f%%", names[k], 100.0 * double(h[k]) / double(h[11]));
            std::fprintf(stderr, "\n");
        }
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tag_prof), z, sizeof(z));
    }
#endif
    const uint32_t cus = P.n_cus ? P.n_cus : 256u;
    // the filter's summary for the workgroups' LDS
    const uint32_t big_log2 = P.tok_bits + kTagFilterLog2, sum_log2 = std::min(big_log2, kSumLog2);
    const uint32_t sum_words = 1u << (sum_log2 - 5u);
#ifdef VPT_TAG_NO_SUMMARY   // (A/B builds: the same geometry without the summary.  A build switch, not an environment variable: nothing on the launch path calls getenv)
    constexpr bool sum = false;
#else
    constexpr bool sum = true;
    hipLaunchKernelGGL(tag_filter_summary_kernel, dim3((sum_words + 255u) / 256u), dim3(256), 0, stream, P.tok_tab + (size_t(4) << P.tok_bits), big_log2, sum_log2,
                       const_cast<uint32_t*>(P.summary));
#endif
    // the grid: what the device holds at a time and no more -- 3 workgroups per CU (6 waves per SIMD; the summary's 32 KB each), their waves
    // striding over the runs.  A workgroup loads its summary once; a second generation of workgroups would load it again and leave the
    // CUs unevenly filled at the end (measured on configs[4]: 768 / 1024 / 1536 / 2048 / 4096 workgroups 0.69 / 0.83 / 0.74 / 0.76 /
    // 0.80 ms, profiles/r05_zc_tag_front_grid.txt).
    const uint64_t want_w = (P.n_runs + kFlatWaves - 1) / kFlatWaves, cap_w = uint64_t(cus) * (VPT_TAG_FLAT_OCC / 2);
    const dim3 grid(uint32_t(want_w < 1 ? 1 : want_w > cap_w ? cap_w : want_w)), block(kFlatThreads);
    hipLaunchKernelGGL((tag_front_flat_kernel<sum>), grid, block, 0, stream, P, sum_log2);
    hipError_t e = launch_scan(P.run_pref, P.n_runs, P.scan_state, ~uint64_t(0), nullptr, nullptr, stream);
    if (e != hipSuccess) return e;
    // the lookups and the passes: what the device holds (8 workgroups of 4 waves per CU), never more waves than there are runs / than a
    // wave per 16 chars of the batch would need
    const uint64_t cap_p = uint64_t(cus) * kTagPairOcc;
    const uint64_t want_r = (P.n_runs + uint64_t(kTagWaves) * kResolveRuns - 1) / (uint64_t(kTagWaves) * kResolveRuns);
    hipLaunchKernelGGL(tag_resolve_kernel, dim3(uint32_t(want_r < 1 ? 1 : want_r > cap_p ? cap_p : want_r)), dim3(kTagThreads), 0, stream, P);
    const uint64_t want_p = (P.total_chars + uint64_t(64) * kTagWaves - 1) / (uint64_t(64) * kTagWaves);
    hipLaunchKernelGGL(tag_pass_kernel, dim3(uint32_t(want_p < 1 ? 1 : want_p > cap_p ? cap_p : want_p)), dim3(kTagThreads), 0, stream, P);
    return hipGetLastError();
}

// vpt_expand_tags_batch_device: the dense array of the C ABI from the records -- None everywhere (the reference's Vec<Option<..>> after
// `resize(n_tags * len, None)`, predictor.rs:556-557), then the records' tags at their tokens' last chars
__global__ __launch_bounds__(256) void expand_tags_kernel(const uint4* __restrict__ records, const int32_t* __restrict__ rec_tags, const uint64_t* __restrict__ n_records,
                                                          const uint32_t n_tags, const uint64_t total_chars, int32_t* __restrict__ tags) {
    const uint64_t n = *n_records, n_items = n * n_tags;
    for (uint64_t i = uint64_t(blockIdx.x) * 256u + threadIdx.x; i < n_items; i += uint64_t(gridDim.x) * 256u) {
        const uint64_t k = i / n_tags, j = i - k * n_tags;
        const uint4 r = records[k];
        const uint64_t gp = uint64_t(r.x) | (uint64_t(r.y) << 32);
        if (gp < total_chars && (r.z & kTokModelMask) != 0) tags[gp * n_tags + j] = rec_tags[i];   // (an empty record: a candidate without a tag model)
    }
}
hipError_t launch_expand_tags(const uint4* records, const int32_t* rec_tags, const uint64_t* n_records, uint32_t n_tags, uint64_t total_chars, int32_t* tags,
                              uint32_t n_cus, hipStream_t stream) {
    const hipError_t e = hipMemsetAsync(tags, 0xFF, size_t(total_chars) * n_tags * sizeof(int32_t), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(expand_tags_kernel, dim3((n_cus ? n_cus : 256u) * 8u), dim3(256), 0, stream, records, rec_tags, n_records, n_tags, total_chars, tags);
    return hipGetLastError();
}

}  // namespace vpt
