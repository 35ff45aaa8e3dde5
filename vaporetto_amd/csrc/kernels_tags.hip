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
// Two kernels, one wave per sentence each:
//   decode_chars_kernel  UTF-8 -> flat per batch (char g of sentence i at out_offsets[i] + i + g): the scored scalar value
//                        | CharacterType << 24; optionally Sentence::char_types on their own
//   tag_tokens_kernel    64 chars per step.  (1) Every lane whose char ends a token owns it: start from the step's boundary
//                        masks, surface hashed from the step's text window in LDS, looked up in the token table (tokens of up
//                        to 4 chars are verified from the 16-byte slot itself).  (2) The tokens that found a tag model are
//                        compacted; a model's char n-grams come in groups by rel_position with a 64-bit filter over the chars
//                        they END with, so a token only enumerates the groups the text can match; the wave's lanes then take
//                        (token, tag n-gram) PAIRS, 64 per round: a lane checks one whole n-gram from one 32-byte record
//                        against the text window in LDS; the matches are compacted and lanes over (match, score) pairs add
//                        their weights to the token's scores in LDS (which start as the model's bias).  The n-grams of ALL the
//                        step's tokens are checked in a few rounds of independent loads instead of token after token, n-gram
//                        after n-gram, symbol after symbol.  (3) Lanes over (token, slot) pairs take the argmax.  Models that
//                        do not fit the record form (an n-gram over 12 symbols or outside the BMP, more than 16 scores or 3
//                        slots, rel_position above 3) go through a whole-wave routine, one token at a time.
//   With predict_tags the scoring kernel of the preceding vpt_predict_batch_device call leaves the decoded chars behind and
//   decode_chars_kernel is skipped (capi.cpp).
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
// the text raise kErrBadOffsets instead of writing outside it.  A lane takes FOUR text bytes per step (one dword, and the
// next one for the tail of a char that starts in its own), counts its lead bytes, and a wave prefix sum places its chars.
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
        for (uint64_t pos = b0; sane && pos < b1; pos += 256) {
            const uint64_t at = pos + 4 * uint64_t(lane);
            // eight bytes from `at`, never reading at or past b1 (the text may end where its allocation does)
            uint64_t x = 0;
            if (at + 8 <= b1) {
                uint32_t lo, hi;
                __builtin_memcpy(&lo, text + at, 4);
                __builtin_memcpy(&hi, text + at + 4, 4);
                x = uint64_t(lo) | (uint64_t(hi) << 32);
            } else {
                for (uint64_t k = 0; k < 8 && at + k < b1; ++k) x |= uint64_t(text[at + k]) << (8 * k);
            }
            const uint32_t mine = at < b1 ? uint32_t(b1 - at < 4 ? b1 - at : 4) : 0u;          // bytes of this lane's own dword inside the sentence
            const uint32_t lm = lead_nibble(uint32_t(x)) & ((1u << mine) - 1u);              // which of them start a char
            const uint32_t incl = wave_inclusive_scan(uint32_t(__popc(lm)));
            const uint32_t total = uint32_t(__builtin_amdgcn_readlane(int(incl), 63));
            uint64_t idx = seen + incl - uint32_t(__popc(lm));
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) {
                if ((lm >> k) & 1u) {
                    if (idx < want) {
                        const uint32_t cp = utf8_scalar(uint32_t(x >> (8 * k)));
                        const uint32_t info = cp < 0x10000u ? cinfo[cp] : 0u;   // the scored char | its CharacterType << 16 (BMP only)
                        const uint32_t ty = cp < 0x10000u ? info >> 16 : char_type(cp);
                        if (cps) cps[g + idx] = (cp < 0x10000u ? info & 0xFFFFu : cp) | (ty << 24);
                        if (types) types[g + idx] = uint8_t(ty);   // Sentence::char_types (sentence.rs:1016)
                    }
                    ++idx;
                }
            }
            seen += total;
        }
        if (lane == 0 && (!sane || seen != want)) atomicOr(status, kErrBadOffsets);   // an empty sentence too (want >= 1)
    }
}

constexpr uint32_t kCharMask = 0x1FFFFFu;   // a cps word: scored scalar value | CharacterType << 24
constexpr int kWin = 128, kWinBack = 32;     // the text window of a step in LDS: positions base - 32 .. base + 95
constexpr int kPairs = 1;                    // (token, record) pairs a lane takes per round: two need 78 VGPRs (6 waves per SIMD), one 64 (8)
constexpr int kTagPass = 32;                 // tokens with a model the fast path takes per pass (a step has at most 64)

struct TagWaveLds {
    union {
        int32_t z[kTagMaxZ];                 // the whole-wave routine: one token's scores
        struct {
            int32_t zt[kTagPass][kTagFastZ + 1];   // the fast path: what the n-grams add to every token of the pass (rows padded against bank conflicts)
            uint32_t tok[kTagPass][6];             // per token: first record, position in the step | zlen << 8 | n_slots << 16, bias offset, packed slots,
                                                   // the four char group sizes, type entries | active groups << 8
            uint32_t pref[kTagPass + 1];           // records before token t (exclusive prefix of the counts)
            uint32_t mlist[128][2];                // the matches of a round: weight offset, token | scores to add << 8
        } f;
    };
    uint32_t txt[kWin];                      // cps words of the window, 0 outside the sentence
};
struct TagLds { TagWaveLds w[kTagWaves]; };
static_assert(sizeof(TagLds) <= 20 * 1024, "8 workgroups per CU");

// the tag model (index + 1) whose token is the chars [s0, e] of the sentence, or 0: one lane on its own.  The chars come
// from the step's LDS window when the token starts inside it (nearly always), a token of up to 4 BMP chars is verified
// from its table slot alone.
__device__ __forceinline__ uint32_t find_tag_model(const TagParams& P, const uint32_t* cps, const uint32_t* txt, int base, int s0, int e) {
    const int len = e - s0 + 1;
    const bool in_win = s0 >= base - kWinBack;
    auto ch = [&](int q) { return (in_win ? txt[q - base + kWinBack] : cps[q]) & kCharMask; };
    uint32_t h = 0x811C9DC5u;
    for (int j = 0; j < len; ++j) h = (h ^ ch(s0 + j)) * 0x01000193u;
    h ^= h >> 15;
    h *= kHashMulLo;
    const uint32_t tok_mask = (1u << P.tok_bits) - 1u;
    uint32_t slot = h >> (32 - P.tok_bits);
    for (;;) {
        const uint4 t = reinterpret_cast<const uint4*>(P.tok_tab)[slot];
        if (t.x == 0) return 0;
        if (int(t.y & 0x7FFFFFFFu) == len) {
            bool same = true;
            if (t.y & 0x80000000u) {
                const uint32_t sy[2] = {t.z, t.w};
                for (int j = 0; j < len && same; ++j) same = ((sy[j >> 1] >> (16 * (j & 1))) & 0xFFFFu) == ch(s0 + j);
            } else {
                const uint32_t* mr = P.models + size_t(t.x - 1) * 12;
                for (int j = 0; j < len && same; ++j) same = P.syms[mr[0] + j] == ch(s0 + j);
            }
            if (same) return t.x;
        }
        slot = (slot + 1) & tok_mask;
    }
}

// One token handled by the whole wave (models outside the record form): lanes over z entries, n-gram chars, slots.
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

__global__ __launch_bounds__(kTagThreads, 8) void tag_tokens_kernel(const TagParams P) {
    __shared__ TagLds LDS;
    const int lane = threadIdx.x & 63;
    TagWaveLds& L = LDS.w[threadIdx.x >> 6];
    volatile int32_t* z = L.z;
    const uint64_t wave = uint64_t(blockIdx.x) * kTagWaves + (threadIdx.x >> 6);
    const uint64_t n_waves = uint64_t(gridDim.x) * kTagWaves;
    const uint32_t nt = P.n_tags;
    for (uint64_t si = wave; si < P.n_sent; si += n_waves) {
        const uint64_t g0 = P.ooff[si] + si;                     // flat index of the sentence's first char
        if (P.ooff[si + 1] < P.ooff[si] || P.ooff[si + 1] + si + 1 > P.total_chars) continue;   // reported by decode_chars_kernel
        if (P.ooff[si + 1] - P.ooff[si] >= 0x7FFFFF00ull) continue;   // (a sentence of 2^31 chars: not in this kernel's index width)
        const int n = int(P.ooff[si + 1] - P.ooff[si]) + 1;  // chars
        const uint32_t* cps = P.cps + g0;
        const uint8_t* lab = P.labels + P.ooff[si];              // n - 1 labels
        int start = 0;              // where the token that is open at the beginning of this step started
        bool have_start = true;     // ... and no Unknown boundary has been seen inside it (predictor.rs:566-567)
        for (int base = 0; base < n; base += 64) {
            const int p = base + lane;
            const uint32_t b = p < n ? (p == n - 1 ? 1u : uint32_t(lab[p])) : 0u;
            // the text window of this step
            for (int w = lane; w < kWin; w += 64) {
                const int q = base - kWinBack + w;
                L.txt[w] = (q >= 0 && q < n) ? cps[q] : 0u;
            }
            const uint64_t ends = __ballot(b == 1u), unk = __ballot(b == 2u);
            __builtin_amdgcn_wave_barrier();
            // ---- (1) this lane's token, if its char ends one: [s0, p], valid when no Unknown lies inside
            const uint64_t below_me = (uint64_t(1) << lane) - 1;
            const uint64_t prev_ends = ends & below_me;
            const int prev = prev_ends ? 63 - __clzll((long long)prev_ends) : -1;
            const int s0 = prev >= 0 ? base + prev + 1 : start;
            const uint64_t after_prev = prev >= 0 ? ~((uint64_t(2) << prev) - 1) : ~uint64_t(0);
            const bool valid = b == 1u && (unk & below_me & after_prev) == 0 && (prev >= 0 || have_start);
            const uint32_t model = valid ? find_tag_model(P, cps, L.txt, base, s0, p) : 0u;
            if (valid && P.tok_model) P.tok_model[g0 + uint64_t(p)] = int32_t(model);   // 0: no tag model for this surface
            // the model record: first record, counts, bias offset, zlen, slots (one trip: three 16-byte loads)
            uint32_t m_first = 0, m_count = 0, m_zlen = 0, m_bias = 0, m_pslots = 0, m_nslots = 0, m_groups = 0, m_tc = 0, m_act = 0;
            bool fast = false;
            if (model != 0) {
                const uint4* mr = reinterpret_cast<const uint4*>(P.models + size_t(model - 1) * 12);
                const uint4 r0 = mr[0], r1 = mr[1], r2 = mr[2];   // dwords 0..3, 4..7, 8..11
                fast = (r2.z & 1u) != 0;
                m_first = r0.z; m_bias = r1.z; m_zlen = r1.w; m_pslots = r2.w;
                m_nslots = r2.y < nt ? r2.y : nt;
                if (fast) {
                    // the char entries come in groups by rel_position r: the n-grams of group r END at char p + r, and the group's
                    // filter says which chars they end with -- a group the text cannot match is not enumerated at all
                    const uint4* fr = reinterpret_cast<const uint4*>(P.mfilt + size_t(model - 1) * 12);
                    const uint4 f0 = fr[0], f1 = fr[1], f2 = fr[2];
                    const uint32_t flo[4] = {f0.x, f0.z, f1.x, f1.z}, fhi[4] = {f0.y, f0.w, f1.y, f1.w};
                    m_groups = f2.x;
                    uint32_t act = 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const uint32_t c = p + r < n ? (L.txt[p + r - base + kWinBack] & kCharMask) : 0u;
                        const uint32_t bit = packed_filter_bit(c);
                        const bool on = P.use_char != 0 && c != 0 && c < 0xFFFFu && (((bit < 32 ? flo[r] >> bit : fhi[r] >> (bit - 32)) & 1u) != 0);
                        if (on) { act |= 1u << r; m_count += (m_groups >> (8 * r)) & 0xFFu; }
                    }
                    m_tc = P.use_type != 0 ? r1.y : 0u;
                    m_count += m_tc;
                    m_act = act;
                }
            }
            const uint64_t fmask = __ballot(fast);
            const uint32_t n_fast = uint32_t(__popcll(fmask));
            const uint32_t rank = uint32_t(__popcll(fmask & below_me));
            // ---- (2) the tokens whose model fits the record form, kTagPass at a time
            for (uint32_t t0 = 0; t0 < n_fast; t0 += kTagPass) {
                const bool mine = fast && rank >= t0 && rank < t0 + kTagPass;
                const uint32_t t_me = rank - t0;
                const uint32_t n_pass = n_fast - t0 < uint32_t(kTagPass) ? n_fast - t0 : uint32_t(kTagPass);
                // exclusive prefix of the record counts over the pass's tokens (wave scan; the other lanes add 0)
                const uint32_t incl = wave_inclusive_scan(mine ? m_count : 0u);
                const uint32_t total = uint32_t(__builtin_amdgcn_readlane(int(incl), 63));
                if (mine) {
                    L.f.tok[t_me][0] = m_first; L.f.tok[t_me][1] = uint32_t(p - base) | (m_zlen << 8) | (m_nslots << 16);
                    L.f.tok[t_me][2] = m_bias; L.f.tok[t_me][3] = m_pslots; L.f.tok[t_me][4] = m_groups; L.f.tok[t_me][5] = m_tc | (m_act << 8);
                    L.f.pref[t_me] = incl - m_count;
                }
                if (lane == 0) L.f.pref[n_pass] = total;
                __builtin_amdgcn_wave_barrier();
                {   // the scores start as the models' bias: lanes over (token, score) pairs, every load of the pass in flight together
                    int32_t bias[kTagPass * kTagFastZ / 64];
#pragma unroll
                    for (int q0 = 0; q0 < kTagPass * int(kTagFastZ) / 64; ++q0) {
                        const uint32_t q = uint32_t(q0) * 64u + uint32_t(lane), t = q / kTagFastZ, i = q % kTagFastZ;
                        bias[q0] = (t < n_pass && i < ((L.f.tok[t][1] >> 8) & 0xFFu)) ? P.weights[L.f.tok[t][2] + i] : 0;
                    }
#pragma unroll
                    for (int q0 = 0; q0 < kTagPass * int(kTagFastZ) / 64; ++q0) {
                        const uint32_t q = uint32_t(q0) * 64u + uint32_t(lane);
                        L.f.zt[q / kTagFastZ][q % kTagFastZ] = bias[q0];
                    }
                }
                __builtin_amdgcn_wave_barrier();
                // rounds of 64 * kPairs (token, record) pairs: every record load of a round is in flight together
                for (uint32_t r0 = 0; r0 < total; r0 += 64 * kPairs) {
                    uint32_t tk[kPairs], hdr[kPairs], woff[kPairs];
                    bool same[kPairs];
                    uint4 ra[kPairs], rb[kPairs];
#pragma unroll
                    for (int u = 0; u < kPairs; ++u) {
                        const uint32_t pi = r0 + uint32_t(u) * 64u + uint32_t(lane);
                        const bool have = pi < total;
                        uint32_t t = 0;
                        if (have) {   // the token of this pair: the last t with pref[t] <= pi
                            uint32_t lo = 0, hi = n_pass;   // pref[lo] <= pi < pref[hi]
                            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (L.f.pref[mid] <= pi) lo = mid; else hi = mid; }
                            t = lo;
                        }
                        tk[u] = t;
                        same[u] = have;
                        uint32_t ri = 0;
                        if (have) {   // the k-th entry of the token's ACTIVE groups (then its type entries) -> record index
                            uint32_t k = pi - L.f.pref[t], off = L.f.tok[t][0];
                            const uint32_t groups = L.f.tok[t][4], act = L.f.tok[t][5] >> 8;
                            bool placed = false;
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const uint32_t c = (groups >> (8 * r)) & 0xFFu;
                                if (!placed && ((act >> r) & 1u)) { if (k < c) { ri = off + k; placed = true; } else k -= c; }
                                off += c;
                            }
                            if (!placed) ri = off + k;   // a type entry
                        }
                        ra[u] = have ? reinterpret_cast<const uint4*>(P.nrec)[size_t(ri) * 2] : make_uint4(0, 0, 0, 0);
                        rb[u] = have ? reinterpret_cast<const uint4*>(P.nrec)[size_t(ri) * 2 + 1] : make_uint4(0, 0, 0, 0);
                    }
                    // the longest n-gram of this round bounds the compare loops (wave-uniform: no lane runs 12 steps for 3-char n-grams)
                    uint32_t gmax = wave_max(kPairs == 2 ? ((ra[0].x & 0xFFu) > (ra[kPairs - 1].x & 0xFFu) ? (ra[0].x & 0xFFu) : (ra[kPairs - 1].x & 0xFFu)) : (ra[0].x & 0xFFu));
                    gmax = gmax < kTagFastSyms ? gmax : kTagFastSyms;
#pragma unroll
                    for (int u = 0; u < kPairs; ++u) {
                        const uint32_t glen = ra[u].x & 0xFFu, rel = (ra[u].x >> 8) & 0xFFu, kind = (ra[u].x >> 16) & 1u;
                        const int e = base + int(same[u] ? L.f.tok[tk[u]][1] & 0xFFu : 0u);
                        const int endp = e + int(rel) + 1, beg = endp - int(glen);
                        bool ok = same[u] && glen != 0 && beg >= 0 && endp <= n && (kind == 0 ? P.use_char != 0 : P.use_type != 0);
                        // beg >= base - 11 and endp <= base + 72 (glen <= 12, rel <= the window <= 8): inside the LDS window
                        // the record's symbols as a 192-bit shift register: the next one is always the low 16 bits
                        uint32_t s0w = ra[u].z, s1w = ra[u].w, s2w = rb[u].x, s3w = rb[u].y, s4w = rb[u].z, s5w = rb[u].w;
                        const int wbeg = beg - base + kWinBack;
                        for (uint32_t j = 0; j < gmax; ++j) {
                            if (j < glen && ok) {
                                const uint32_t c = L.txt[(wbeg + int(j)) & (kWin - 1)];
                                ok = (s0w & 0xFFFFu) == (kind == 0 ? (c & kCharMask) : (c >> 24));
                            }
                            s0w = (s0w >> 16) | (s1w << 16); s1w = (s1w >> 16) | (s2w << 16); s2w = (s2w >> 16) | (s3w << 16);
                            s3w = (s3w >> 16) | (s4w << 16); s4w = (s4w >> 16) | (s5w << 16); s5w >>= 16;
                        }
                        same[u] = ok;
                        hdr[u] = ra[u].x; woff[u] = ra[u].y;
                    }
                    // the matches of the round, compacted; then lanes over (match, score) pairs add the weights: four matches per
                    // 64 lanes, the loads of up to sixteen matches in flight together
                    const uint64_t m0 = __ballot(same[0]), m1 = kPairs == 2 ? __ballot(same[kPairs - 1]) : 0;
                    const uint32_t n0m = uint32_t(__popcll(m0)), n_match = n0m + uint32_t(__popcll(m1));
                    if (n_match != 0) {
#pragma unroll
                        for (int u = 0; u < kPairs; ++u) {
                            if (same[u]) {
                                const uint32_t k = (u ? n0m : 0u) + uint32_t(__popcll((u ? m1 : m0) & below_me));
                                const uint32_t zlen = (L.f.tok[tk[u]][1] >> 8) & 0xFFu;
                                const uint32_t wl = (hdr[u] >> 24) < zlen ? (hdr[u] >> 24) : zlen;   // zip: the shorter of the two (predictor.rs:82-89)
                                L.f.mlist[k][0] = woff[u]; L.f.mlist[k][1] = tk[u] | (wl << 8);
                            }
                        }
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
                }
                __builtin_amdgcn_wave_barrier();
                // ---- (3) argmax per (token, slot) (TagPredictor::predict, predictor.rs:286-304)
                for (uint32_t a0 = 0; a0 < n_pass * nt; a0 += 64) {
                    const uint32_t a = a0 + uint32_t(lane);
                    if (a < n_pass * nt) {
                        const uint32_t t = a / nt, j = a - t * nt;
                        const uint32_t info = L.f.tok[t][1], zlen = (info >> 8) & 0xFFu, n_slots = info >> 16;
                        if (j < n_slots) {
                            const uint32_t ps = L.f.tok[t][3] >> (9 * j), cnt = ps & 31u, off = (ps >> 5) & 15u;
                            int32_t tag = cnt == 1 ? 0 : -1;
                            if (cnt >= 2) {
                                int32_t best = INT32_MIN;
                                tag = 0;
                                for (uint32_t c = 0; c < cnt && off + c < zlen; ++c) {
                                    const int32_t v = L.f.zt[t][off + c];
                                    if (v > best) { best = v; tag = int32_t(c); }
                                }
                            }
                            P.tags[(g0 + uint64_t(uint32_t(base) + (info & 0xFFu))) * nt + j] = tag;
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            // ---- the models outside the record form: the whole wave, one token at a time
            uint64_t todo = __ballot(model != 0 && !fast);
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
            __builtin_amdgcn_wave_barrier();   // the window is rewritten by the next step
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
