// Token emission on the device: Sentence::write_tokenized_text for a batch (sentence.rs:850-886), boundary part:
// the tokens of a sentence are the runs between WordBoundary labels (sentence.rs:1270-1300), written in order with
// one ' ' between them and a '\' in front of every ' ', '\' and '/' byte of a surface; with tags (the indices
// vpt_fill_tags_batch wrote and the tag model it found for every token) each token is followed by "/tag" for its tag
// slots up to the last Some, an empty string for a None in between (sentence.rs:866-881).  (Unknown boundaries only
// come from partially annotated corpora, never from predict: they are rejected here, kErrUnknownLabel.)
//
// Output size is data dependent: emit_fused_kernel (below) sizes, places and writes the batch in one launch (three launches --
// count, prefix sum, write, a wave per sentence -- until round 3: profiles/r02_j_emit_kernels.txt, r03_l_emit_kernel_stats.csv).
// count_chars_kernel + scan_chained_kernel are vpt_count_boundaries on the device.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "device_common.h"
#include "kernels.hpp"

namespace vpt {
namespace {

constexpr int kEmitThreads = 256;
constexpr int kEmitWaves = kEmitThreads / 64;
constexpr int kScanThreads = 256, kScanPer = 16;
constexpr uint64_t kScanBlock = uint64_t(kScanThreads) * kScanPer;   // offsets one workgroup of the scan takes

// "/tag/tag.." of the token whose last char is char `c` (batch-flat index) and whose tag model (index + 1, from the
// fill_tags call) is `model`: bytes it takes; written to `dst` when given -- never more than `limit` of them (what was reserved
// for it: a caller that changed the tags after fill_tags gets the offsets error, not a write outside the suffix's place)
__device__ __noinline__ uint32_t tag_suffix_of(const EmitParams& P, uint64_t c, int32_t model, uint8_t* dst, uint32_t limit = 0xFFFFFFFFu) {
    if (model <= 0 || uint32_t(model) > P.n_models) return 0;   // no tag model for this surface (or not our array)
    const uint32_t* mr = P.models + size_t(model - 1) * 12;
    const int32_t* tg = P.tags + c * P.n_tags;
    const uint32_t slot0 = mr[8], n_slots = mr[9] < P.n_tags ? mr[9] : P.n_tags;
    uint32_t last = 0;   // slots to write: up to the last Some (the tags are fetched with the model record, not after it)
    for (uint32_t j = 0; j < P.n_tags; ++j)
        if (tg[j] >= 0 && j < n_slots) last = j + 1;
    uint32_t n = 0;
    for (uint32_t j = 0; j < last; ++j) {
        if (dst && n < limit) dst[n] = 0x2Fu;
        ++n;
        if (tg[j] < 0) continue;
        const uint32_t k = P.slot_str[slot0 + j] + uint32_t(tg[j]);
        if (k >= P.n_strings) continue;                            // an index fill_tags cannot have written
        const uint32_t a = P.str_off[k], b = P.str_off[k + 1];
        if (dst) for (uint32_t q = a; q < b && n + (q - a) < limit; ++q) dst[n + (q - a)] = P.str_bytes[q];
        n += b - a;
    }
    return n;
}
__device__ __forceinline__ uint32_t tag_suffix(const EmitParams& P, uint64_t c, uint8_t* dst) {
    return P.tags ? tag_suffix_of(P, c, int32_t(uint32_t(P.tok_model[c]) & kTokModelMask), dst) : 0u;
}
// the bytes of the suffix of the token whose tok_model word is w (layout.h): carried from fill_tags unless it is a long one
__device__ __forceinline__ uint32_t tag_suffix_bytes(const EmitParams& P, uint64_t c, uint32_t w) {
    const uint32_t code = w >> kTokSuffixShift;
    return (w & kTokModelMask) == 0 ? 0u : code != kTokSuffixLong ? code : tag_suffix_of(P, c, int32_t(w & kTokModelMask), nullptr);
}
__device__ __forceinline__ uint64_t wave_sum64(uint64_t x) {   // total over the 64 lanes, in every lane
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t lo = uint32_t(__shfl_xor(int(uint32_t(x)), d)), hi = uint32_t(__shfl_xor(int(uint32_t(x >> 32)), d));
        x += uint64_t(lo) | (uint64_t(hi) << 32);
    }
    return x;
}

// the scan's state, zeroed by the kernel in front of it: workgroup b of that kernel clears word b (its grid is at least as large)
__device__ __forceinline__ void clear_scan_state(uint64_t* state, uint64_t n) {
    const uint64_t n_part = (n + kScanBlock - 1) / kScanBlock;
    if (threadIdx.x == 0) {
        for (uint64_t k = blockIdx.x; k <= n_part; k += gridDim.x) state[k] = 0;
    }
}

// ---- inclusive prefix sum over offsets[1 .. n] in place (offsets[k] = sum of the lengths of sentences 0 .. k-1)
__device__ __forceinline__ uint64_t block_exclusive_scan(uint64_t* lds, uint64_t v, uint32_t tid, uint32_t n_threads, uint64_t* total) {
    lds[tid] = v;
    __syncthreads();
    for (uint32_t d = 1; d < n_threads; d <<= 1) {        // Hillis-Steele
        const uint64_t t = tid >= d ? lds[tid - d] : 0;
        __syncthreads();
        lds[tid] += t;
        __syncthreads();
    }
    const uint64_t incl = lds[tid];
    *total = lds[n_threads - 1];
    __syncthreads();
    return incl - v;
}

// ONE launch: a chained scan.  A workgroup takes a ticket (so that every workgroup with a smaller number is already running: the
// look-back below cannot wait for one that has not started), sums its kScanBlock entries, publishes the sum, walks back over its
// predecessors' words until one holds an inclusive prefix, publishes its own inclusive prefix and writes its entries.  A word =
// flag << 62 | value (1: the block's sum, 2: the prefix up to and including the block; 0: nothing yet), written and read as one
// 64-bit access, so value and flag can never be seen apart.  `state` (n_part words + the ticket) is zeroed by the kernel that
// produced the lengths (the launch in front of this one on the stream).
// total_out (optional): where the grand total is left as well -- host memory the device can write (hipHostMalloc), so that a
// caller waiting on an event of the stream reads the size of the output without a copy of its own
__global__ __launch_bounds__(kScanThreads) void scan_chained_kernel(uint64_t* __restrict__ offsets, uint64_t n, uint64_t* __restrict__ state, uint64_t n_part,
                                                                    uint64_t capacity, uint32_t* __restrict__ status, uint64_t* __restrict__ total_out) {
    __shared__ uint64_t lds[kScanThreads];
    __shared__ uint64_t bcast[2];
    const uint32_t tid = threadIdx.x;
    if (tid == 0) bcast[0] = atomicAdd(reinterpret_cast<unsigned long long*>(state + n_part), 1ull);
    __syncthreads();
    const uint64_t blk = bcast[0];
    const uint64_t first = blk * kScanBlock + uint64_t(tid) * kScanPer;   // this thread's consecutive entries
    uint64_t v[kScanPer];
    uint64_t sum = 0;
#pragma unroll
    for (int j = 0; j < kScanPer; ++j) {
        v[j] = first + j < n ? offsets[first + j + 1] : 0;
        sum += v[j];
    }
    uint64_t total;
    uint64_t run = block_exclusive_scan(lds, sum, tid, kScanThreads, &total);
    if (tid == 0) {
        constexpr uint64_t kVal = (uint64_t(1) << 62) - 1;
        uint64_t base = 0;
        if (blk != 0) {
            __hip_atomic_store(state + blk, (uint64_t(1) << 62) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (uint64_t p = blk; p-- > 0;) {
                uint64_t w;
                do { w = __hip_atomic_load(state + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((w >> 62) == 0);
                base += w & kVal;
                if ((w >> 62) == 2) break;
            }
        }
        __hip_atomic_store(state + blk, (uint64_t(2) << 62) | ((base + total) & kVal), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bcast[1] = base;
        if (blk == 0) offsets[0] = 0;
        if (blk == n_part - 1) {
            if (base + total > capacity) atomicOr(status, kErrOutputTooSmall);
            if (total_out) *total_out = base + total;
        }
    }
    __syncthreads();
    run += bcast[1];
#pragma unroll
    for (int j = 0; j < kScanPer; ++j) {
        run += v[j];
        if (first + j < n) offsets[first + j + 1] = run;
    }
}

hipError_t launch_scan(uint64_t* offsets, uint64_t n, uint64_t* part, uint64_t capacity, uint32_t* status, uint64_t* total_out, hipStream_t stream) {
    const uint64_t n_part = (n + kScanBlock - 1) / kScanBlock;
    hipLaunchKernelGGL(scan_chained_kernel, dim3(uint32_t(n_part)), dim3(kScanThreads), 0, stream, offsets, n, part, n_part, capacity, status, total_out);
    return hipGetLastError();
}
// ------------------------------------------------------------------------------------------------------------
// emit_fused_kernel: the whole writer in ONE launch.
//
// The batch is FLAT for the writer: the output is the text with insertions (a ' ' in front of a char whose label in front is a
// boundary, a '\' in front of an escaped byte, "/tag.." in front of the ' ' -- or of the sentence's end -- that ends a token with
// tags), and the sentences' positions are the output positions of their first bytes.  A WAVE takes a block of consecutive
// sentences (`per_block` of them, at most 64: about a KB or three of text) and walks its bytes in steps of 1 KB, SIXTEEN bytes per
// lane, every lane busy whatever the sentences' lengths are:
//   pass A  what the block will write: its bytes + the escaped bytes + the boundary labels of its label range (plain reductions
//           over 16-byte loads; with tags: a dry run of pass B) -> published; a decoupled look-back over the earlier blocks'
//           words (64 of them per trip) gives the block's position in the output
//   pass B  per step: lead / escape / sentence-start masks of the lane's 16 bytes (the starts come from the block's byte offsets
//           through an LDS bitmap), ONE DPP prefix sum numbers the lane's chars and sentences, which names the labels of its chars
//           in the window of labels staged in LDS with the step; a second prefix sum places the lane's output, which is assembled
//           in LDS and leaves as aligned 16-byte stores.  The lane that holds a sentence's first byte writes its offset and checks
//           that the sentence starts at the char its boundary offset promises.
// Blocks take a TICKET, so that every block with a smaller number is running or done (the look-back cannot wait for one that
// has not started).  The state words of a call (one per block + the ticket) were zeroed by the call before it, which used the
// other of two arrays: no launch in front of this one.  A word = flag << 62 | value (1: the block's size, 2: the output position
// behind the block), one 64-bit access.
// ------------------------------------------------------------------------------------------------------------
constexpr uint32_t kFuseStepBytes = 1024;                    // text bytes of a wave's step
constexpr uint32_t kFuseStageBytes = 3 * kFuseStepBytes + 32; // a step's output without tag suffixes (every byte escaped, a space per char) + the alignment head
constexpr uint32_t kFuseLabDwords = (kFuseStepBytes + 48) / 4;
struct alignas(16) FuseWaveLds {
    uint32_t stage[kFuseStageBytes / 4];
    uint32_t labs[kFuseLabDwords];            // the labels a step's chars can ask for, from a 16-byte aligned address
    uint32_t starts[kFuseStepBytes / 32];     // one bit per byte of the step: a sentence starts here
    uint32_t so[kEmitFuseMaxBlock + 1];       // the block's boundary offsets, relative to its first
    uint32_t dump[64];                        // where a lane's stores of bytes that are not there go
};
struct FuseLds {
    FuseWaveLds w[kEmitWaves];
    uint64_t ticket;
};

__device__ __forceinline__ uint32_t lead16(const uint4& x) { return lead_nibble(x.x) | (lead_nibble(x.y) << 4) | (lead_nibble(x.z) << 8) | (lead_nibble(x.w) << 12); }
__device__ __forceinline__ uint32_t esc16(const uint4& x) { return esc_nibble(x.x) | (esc_nibble(x.y) << 4) | (esc_nibble(x.z) << 8) | (esc_nibble(x.w) << 12); }
__device__ __forceinline__ uint32_t one_nibble(uint32_t y) { return byte_flags_to_nibble(zero_bytes(y ^ 0x01010101u)); }   // bytes equal to 1
__device__ __forceinline__ uint32_t one16(const uint4& y) { return one_nibble(y.x) | (one_nibble(y.y) << 4) | (one_nibble(y.z) << 8) | (one_nibble(y.w) << 12); }
__device__ __forceinline__ uint32_t unk_nibble(uint32_t y) { return byte_flags_to_nibble(~zero_bytes(y & 0xFEFEFEFEu) & 0x80808080u); }   // bytes above 1
__device__ __forceinline__ uint32_t unk16(const uint4& y) { return unk_nibble(y.x) | (unk_nibble(y.y) << 4) | (unk_nibble(y.z) << 8) | (unk_nibble(y.w) << 12); }
// which of the 16 bytes at `addr` lie in [lo, hi)
__device__ __forceinline__ uint32_t in_range16(uintptr_t addr, uintptr_t lo, uintptr_t hi) {
    const uint32_t a = lo > addr ? (lo - addr < 16 ? uint32_t(lo - addr) : 16u) : 0u;
    const uint32_t b = hi > addr ? (hi - addr < 16 ? uint32_t(hi - addr) : 16u) : 0u;
    return b > a ? ((1u << b) - 1u) & ~((1u << a) - 1u) : 0u;
}
__device__ __forceinline__ uint32_t byte_of(const uint4& x, uint32_t k) {
    const uint32_t d = k < 4 ? x.x : k < 8 ? x.y : k < 12 ? x.z : x.w;
    return (d >> (8 * (k & 3u))) & 0xFFu;
}
__device__ __forceinline__ uint32_t byte_of_rt(const uint4& x, uint32_t k) {   // k not known at compile time
    const uint32_t q = k >> 2;
    const uint32_t d = q == 0 ? x.x : q == 1 ? x.y : q == 2 ? x.z : x.w;
    return (d >> (8 * (k & 3u))) & 0xFFu;
}
__device__ __forceinline__ uint64_t lane_value64(uint64_t v, int src) {
    return uint64_t(uint32_t(__builtin_amdgcn_readlane(int(uint32_t(v)), src))) | (uint64_t(uint32_t(__builtin_amdgcn_readlane(int(uint32_t(v >> 32)), src))) << 32);
}

struct FuseBlock {            // a wave's block (the same in every lane)
    uint64_t i0, B0, B1, O0, O1;
    uint32_t ns;
};

// a step's loads: the lane's 16 text bytes and its 16 (lanes 0..2: 32) bytes of the label window, which starts at label `lw`
struct FuseLoads { uint4 x, la, lb; uint32_t vm, lab_head; };
__device__ __forceinline__ FuseLoads fuse_load(uintptr_t step, uintptr_t t_lo, uintptr_t t_hi, uintptr_t l_lo, uintptr_t l_hi, uint64_t lw, int lane) {
    FuseLoads r;
    const uintptr_t addr = step + 16u * uint32_t(lane);
    r.vm = in_range16(addr, t_lo, t_hi);
    r.x = r.vm ? *reinterpret_cast<const uint4*>(addr) : make_uint4(0, 0, 0, 0);
    const uintptr_t lab_at = l_lo + lw, lab_al = lab_at & ~uintptr_t(15);
    r.lab_head = uint32_t(lab_at - lab_al);
    const uintptr_t a = lab_al + 16u * uint32_t(lane), a2 = lab_al + 16u * uint32_t(64 + lane);
    r.la = (step < t_hi && a + 16 > l_lo && a < l_hi) ? *reinterpret_cast<const uint4*>(a) : make_uint4(0, 0, 0, 0);
    r.lb = (step < t_hi && lane < 3 && a2 + 16 > l_lo && a2 < l_hi) ? *reinterpret_cast<const uint4*>(a2) : make_uint4(0, 0, 0, 0);
    return r;
}

// Pass B over a block: every byte of its sentences to its place.  `my_b`: boff[i0 + lane] for the block's sentences; `at_out`: where
// the block goes; nothing is stored past `end` (the size pass A published) and nothing at all unless `store_ok` (the output fits
// the caller's buffer) -- the sentences' offsets are written either way.  kTags: the kernel's variant with "/tag" suffixes.
template <bool kTags, bool kDbg>
__device__ __forceinline__ void fuse_walk(const EmitParams& P, const FuseBlock& K, uint64_t my_b, int lane, FuseWaveLds& L, uint64_t at_out, uint64_t end,
                                          bool store_ok, uint32_t& err, uint32_t dbg) {
    uint8_t* const sbytes = reinterpret_cast<uint8_t*>(L.stage);
    const uintptr_t t_lo = reinterpret_cast<uintptr_t>(P.text) + K.B0, t_hi = reinterpret_cast<uintptr_t>(P.text) + K.B1;
    const uintptr_t l_lo = reinterpret_cast<uintptr_t>(P.labels), l_hi = l_lo + P.total_boundaries;
    const uintptr_t my_start = reinterpret_cast<uintptr_t>(P.text) + my_b;
    uint64_t cb = 0, sb = 0;   // chars / sentence starts of the block in front of the step
    bool fits = true;
    uintptr_t step = t_lo & ~uintptr_t(15);
    // the labels the step's chars can ask for: label (O0 + cb - sb) onwards (every char but a sentence's first has one in front)
    FuseLoads nxt = fuse_load(step, t_lo, t_hi, l_lo, l_hi, K.O0, lane);
    for (; step < t_hi && fits; step += kFuseStepBytes) {
        const FuseLoads cur = nxt;
        const uint4 x = cur.x;
        const uint32_t vm = cur.vm;
        reinterpret_cast<uint4*>(L.labs)[lane] = cur.la;
        if (lane < 3) reinterpret_cast<uint4*>(L.labs)[64 + lane] = cur.lb;
        if (uint32_t(lane) < K.ns && my_start >= step && my_start - step < kFuseStepBytes) {
            const uint32_t r = uint32_t(my_start - step);
            atomicOr(&L.starts[r >> 5], 1u << (r & 31u));
        }
        __builtin_amdgcn_wave_barrier();
        uint32_t sm = (L.starts[lane >> 1] >> (16 * (lane & 1))) & 0xFFFFu;
        __builtin_amdgcn_wave_barrier();
        if (lane < int(kFuseStepBytes / 32)) L.starts[lane] = 0;
        const uint32_t lm = lead16(x) & vm, em = esc16(x) & vm;
        if (sm & ~lm) err |= kErrBadOffsets;   // a sentence that starts inside a char (or outside the block)
        sm &= lm;
        const uint32_t nl = uint32_t(__popc(lm)), nst = uint32_t(__popc(sm));
        const uint32_t incl = wave_inclusive_scan(nl | (nst << 16));
        const uint32_t tot = uint32_t(__builtin_amdgcn_readlane(int(incl), 63));
        const uint32_t c_in = (incl & 0xFFFFu) - nl, s_in = (incl >> 16) - nst;   // chars / starts of the step in front of this lane
        // the next step's text and labels are on their way while this one is placed
        nxt = fuse_load(step + kFuseStepBytes, t_lo, t_hi, l_lo, l_hi, K.O0 + (cb + (tot & 0xFFFFu)) - (sb + (tot >> 16)), lane);
        // the lane's chars that have a label in front take consecutive labels from (c_in - s_in) of the window on
        const uint32_t nm = lm & ~sm;
        const uint32_t loff = cur.lab_head + (c_in - s_in);           // byte offset in labs: <= 15 + 1024
        uint32_t spm = 0;
        if (!(kDbg && (dbg & 16u))) {
            const uint32_t d = loff >> 2, r = loff & 3u;
            uint4 y;
            y.x = __builtin_amdgcn_alignbyte(L.labs[d + 1], L.labs[d], r); y.y = __builtin_amdgcn_alignbyte(L.labs[d + 2], L.labs[d + 1], r);
            y.z = __builtin_amdgcn_alignbyte(L.labs[d + 3], L.labs[d + 2], r); y.w = __builtin_amdgcn_alignbyte(L.labs[d + 4], L.labs[d + 3], r);
            uint32_t bits = one16(y), rem = nm;
            while (rem) {   // label q of the lane onto its q-th labelled char
                const uint32_t low = rem & (0u - rem);
                if (bits & 1u) spm |= low;
                bits >>= 1;
                rem &= rem - 1u;
            }
        }
        // Tag suffixes go in front of a space and in front of a sentence's first byte (the last token of the sentence before it),
        // except the block's first (the block before this one wrote that one behind its last byte).  The lane that holds the byte
        // in FRONT of which a suffix goes owns it; at most two per lane are carried in registers (tk: the byte, tl: the length,
        // tc: the token's last char), a third sends the lane through its chars one by one.
        uint32_t tmask = 0, tk1 = 16, tl1 = 0, tk2 = 16, tl2 = 0, tc1 = 0, tc2 = 0;
        int32_t tm1 = 0, tm2 = 0;
        bool many = false;
        const uint64_t g_first = K.O0 + K.i0 + cb + c_in;     // batch-flat index of the lane's first char
        if (kTags) {
            tmask = spm | sm;
            if (sb + s_in == 0 && sm) tmask &= ~(sm & (0u - sm));
            if (kDbg && (dbg & 64u)) tmask = 0;
            if (tmask) {
                // the tag models of the chars in front of the lane's chars: tok_model[g_first - 1 + j] for its j-th char (the array has
                // zeros in front of the batch's first char and behind its last: capi.cpp); which of them are there at all
                const int32_t* tmod = P.tok_model + g_first - 1;
                uint32_t pm = 0;
                for (uint32_t j0 = 0; j0 < nl; j0 += 4) {
                    int32_t m4[4];
                    __builtin_memcpy(m4, tmod + j0, sizeof(m4));
                    pm |= (((uint32_t(m4[0]) & kTokModelMask) ? 1u : 0u) | ((uint32_t(m4[1]) & kTokModelMask) ? 2u : 0u) | ((uint32_t(m4[2]) & kTokModelMask) ? 4u : 0u) |
                           ((uint32_t(m4[3]) & kTokModelMask) ? 8u : 0u)) << j0;
                }
                pm &= (1u << nl) - 1u;
                while (pm) {   // few
                    const uint32_t j = uint32_t(__ffs(int(pm))) - 1u;
                    pm &= pm - 1u;
                    uint32_t rem = lm;
                    for (uint32_t q = 0; q < j; ++q) rem &= rem - 1u;
                    const uint32_t k = uint32_t(__ffs(int(rem))) - 1u;          // the byte of the lane's j-th char
                    if (!((tmask >> k) & 1u)) continue;                          // no token ends in front of it
                    const uint32_t word = uint32_t(tmod[j]);
                    const int32_t mdl = int32_t(word & kTokModelMask);
                    const uint32_t len = tag_suffix_bytes(P, g_first + j - 1u, word);   // (carried from fill_tags)
                    if (!len) continue;
                    if (tk1 == 16) { tk1 = k; tl1 = len; tc1 = j; tm1 = mdl; }
                    else if (tk2 == 16) { tk2 = k; tl2 = len; tc2 = j; tm2 = mdl; }
                    else many = true;
                }
            }
        }
        const bool slow = kTags && __ballot(many) != 0;        // (wave-uniform)
        uint32_t sfx_total = tl1 + tl2;
        if (slow) {
            sfx_total = 0;
            uint32_t todo = tmask;
            while (todo) {
                const uint32_t low = todo & (0u - todo);
                todo &= todo - 1u;
                sfx_total += tag_suffix(P, g_first + uint32_t(__popc(lm & (low - 1u))) - 1u, nullptr);
            }
        }
        const uint32_t t = uint32_t(__popc(vm)) + uint32_t(__popc(spm)) + uint32_t(__popc(em)) + sfx_total;
        const uint32_t incl_t = wave_inclusive_scan(t);
        const uint32_t total = uint32_t(__builtin_amdgcn_readlane(int(incl_t), 63));
        if (at_out + total > end) { fits = false; break; }
        uint8_t* const dst = P.out_text + at_out;
        const uint32_t head = uint32_t(reinterpret_cast<uintptr_t>(dst) & 15u);
        const bool staged = !kTags || head + total <= kFuseStageBytes;   // (wave-uniform) else: byte stores straight to the output
        const uint32_t w = incl_t - t;
        if (!slow) {
            // the lane's bytes in order: [tags] [' '] ['\\'] byte.  No branches: what is not there goes to a slot of the lane's own
            if (store_ok && !(kDbg && (dbg & 1u))) {
                uint8_t* const o = !kTags || staged ? sbytes + head : dst;   // (with tags: a generic pointer)
                uint8_t* const dump = reinterpret_cast<uint8_t*>(L.dump + lane);
                uint32_t pos = w;
#pragma unroll
                for (uint32_t k = 0; k < 16; ++k) {
                    const uint32_t v = (vm >> k) & 1u, sp = (spm >> k) & 1u, es = (em >> k) & 1u;
                    if (kTags) pos += (k == tk1 ? tl1 : 0u) + (k == tk2 ? tl2 : 0u);
                    *(sp ? o + pos : dump) = 0x20u; pos += sp;
                    *(es ? o + pos : dump) = 0x5Cu; pos += es;
                    *(v ? o + pos : dump) = uint8_t(byte_of(x, k)); pos += v;
                }
                if (kTags && tl1) {   // the tags themselves (few lanes)
                    const uint32_t b1 = (1u << tk1) - 1u, b2 = (1u << tk2) - 1u;
                    if (tag_suffix_of(P, g_first + tc1 - 1u, tm1, o + w + uint32_t(__popc(vm & b1)) + uint32_t(__popc(spm & b1)) + uint32_t(__popc(em & b1)), tl1) != tl1) err |= kErrBadOffsets;
                    if (tl2 && tag_suffix_of(P, g_first + tc2 - 1u, tm2, o + w + tl1 + uint32_t(__popc(vm & b2)) + uint32_t(__popc(spm & b2)) + uint32_t(__popc(em & b2)), tl2) != tl2) err |= kErrBadOffsets;
                }
            }
            uint32_t rem = sm;   // the sentences that start in the lane's bytes (few lanes, one as a rule)
            while (rem) {
                const uint32_t k = uint32_t(__ffs(int(rem))) - 1u, below = (1u << k) - 1u;
                rem &= rem - 1u;
                const uint64_t s = sb + s_in + uint32_t(__popc(sm & below));
                if (s < K.ns) {
                    P.out_offsets[K.i0 + s] = at_out + w + uint32_t(__popc(vm & below)) + uint32_t(__popc(spm & below)) + uint32_t(__popc(em & below)) +
                                              (kTags ? (tk1 <= k ? tl1 : 0u) + (tk2 <= k ? tl2 : 0u) : 0u);
                    if (cb + c_in + uint32_t(__popc(lm & below)) != uint64_t(L.so[s]) + s) err |= kErrBadOffsets;   // not the char its offset names
                } else err |= kErrBadOffsets;
            }
        } else {
            // [tags] [' '] | sentence offset | ['\\'] byte, char by char (o: LDS or the output itself)
            uint8_t* const o = !store_ok ? nullptr : staged ? sbytes + head : dst;
            uint32_t pos = w, ci = 0;
#pragma unroll 1
            for (uint32_t k = 0; k < 16; ++k) {
                if (!((vm >> k) & 1u)) continue;
                if ((tmask >> k) & 1u) pos += tag_suffix(P, g_first + ci - 1u, o ? o + pos : nullptr);
                if ((spm >> k) & 1u) { if (o) o[pos] = 0x20u; ++pos; }
                if ((sm >> k) & 1u) {
                    const uint64_t s = sb + s_in + uint32_t(__popc(sm & ((1u << k) - 1u)));
                    if (s < K.ns) {
                        P.out_offsets[K.i0 + s] = at_out + pos;
                        if (cb + c_in + ci != uint64_t(L.so[s]) + s) err |= kErrBadOffsets;
                    } else err |= kErrBadOffsets;
                }
                if ((em >> k) & 1u) { if (o) o[pos] = 0x5Cu; ++pos; }
                if (o) o[pos] = uint8_t(byte_of_rt(x, k));
                ++pos;
                ci += (lm >> k) & 1u;
            }
        }
        if (store_ok && staged && !(kDbg && (dbg & 2u))) {   // LDS byte j is output byte j - head: whole 16-byte chunks leave aligned, the two edges byte by byte
            __builtin_amdgcn_wave_barrier();
            uint8_t* const abase = dst - head;
            const uint32_t nd = (head + total + 15u) >> 4;
            for (uint32_t d = uint32_t(lane); d < nd; d += 64) {
                const uint32_t lo = d * 16u, hi = lo + 16u;
                if (lo >= head && hi <= head + total) {
                    *reinterpret_cast<uint4*>(abase + lo) = reinterpret_cast<const uint4*>(L.stage)[d];
                } else {
                    const uint32_t a = lo > head ? lo : head, b = hi < head + total ? hi : head + total;
                    for (uint32_t j = a; j < b; ++j) abase[j] = sbytes[j];
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        at_out += total;
        cb += tot & 0xFFFFu;
        sb += tot >> 16;
    }
    if (kTags && fits) {   // the tags of the block's last token
        const uint64_t g_last = K.O1 + K.i0 + K.ns - 1;
        const uint32_t s = tag_suffix(P, g_last, nullptr);   // (every lane computes the same)
        if (s && store_ok && at_out + s <= end && lane == 0) tag_suffix(P, g_last, P.out_text + at_out);
        at_out += s;
    }
    // (what was written is what pass A said: anything else means chars, labels, offsets -- or the tags' token words and the labels,
    // which must be the ones fill_tags saw -- do not belong together)
    if (!fits || at_out != end || cb != (K.O1 - K.O0) + K.ns || sb != K.ns) err |= kErrBadOffsets;
}

// Pass A with tags: the bytes of the block's tag suffixes -- a reduction over the token words of its chars (fill_tags left the bytes
// of a token's tags in the word of its last char: layout.h), sixteen chars per lane in flight.
__device__ __forceinline__ uint32_t fuse_tag_bytes(const EmitParams& P, const FuseBlock& K, int lane) {
    const uint64_t g0 = K.O0 + K.i0, n_chars = (K.O1 - K.O0) + K.ns;
    uint32_t bytes = 0;
    for (uint64_t c0 = 4 * uint64_t(lane); c0 < n_chars; c0 += 1024) {
        int32_t m[4][4];
#pragma unroll
        for (uint32_t r = 0; r < 4; ++r) {
            if (c0 + 256 * r < n_chars) __builtin_memcpy(m[r], P.tok_model + g0 + c0 + 256 * r, sizeof(m[r]));   // (the array goes on behind the batch's last char)
            else m[r][0] = m[r][1] = m[r][2] = m[r][3] = 0;
        }
#pragma unroll
        for (uint32_t r = 0; r < 4; ++r)
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) {
                const uint64_t c = c0 + 256 * r + q;
                if (c < n_chars) bytes += tag_suffix_bytes(P, g0 + c, uint32_t(m[r][q]));
            }
    }
    return bytes;
}

template <bool kTags, bool kDbg, int kOcc = 1>
__global__ __launch_bounds__(kEmitThreads, kOcc) void emit_fused_kernel(const EmitParams P, const EmitFuse F) {
    const uint32_t dbg = kDbg ? F.dbg : 0u;   // timing ablations (VPT_DEBUG_EMIT; results are wrong with any bit set)
    __shared__ FuseLds LDS;
    // the other array of state words, for the call after this one
    for (uint64_t k = uint64_t(blockIdx.x) * kEmitThreads + threadIdx.x; k < F.clear_n; k += uint64_t(gridDim.x) * kEmitThreads) F.clear[k] = 0;
    const int lane = threadIdx.x & 63;
    const uint32_t wid = wave_uniform(threadIdx.x >> 6);
    FuseWaveLds& L = LDS.w[wid];
    // a workgroup's waves take consecutive tickets with one atomic
    if (threadIdx.x == 0) LDS.ticket = atomicAdd(reinterpret_cast<unsigned long long*>(F.state + F.n_blocks), (unsigned long long)kEmitWaves);
    __syncthreads();
    const uint64_t blk = wave_uniform64(LDS.ticket) + wid;
    if (blk >= F.n_blocks) return;   // (the grid is whole workgroups)
    FuseBlock K;
    K.i0 = blk * F.per_block;
    K.ns = uint32_t(P.n_sent - K.i0 < F.per_block ? P.n_sent - K.i0 : F.per_block);
    // the block's offsets: lane j holds sentence i0 + j's; what follows the block's last in every lane
    uint64_t my_b = ~uint64_t(0), my_o = 0;
    if (uint32_t(lane) < K.ns) { my_b = P.boff[K.i0 + lane]; my_o = P.ooff[K.i0 + lane]; }
    K.B1 = wave_uniform64(P.boff[K.i0 + K.ns]); K.O1 = wave_uniform64(P.ooff[K.i0 + K.ns]);
    K.B0 = lane_value64(my_b, 0); K.O0 = lane_value64(my_o, 0);
    uint32_t err = 0;
    bool sane;
    {
        const int up = (lane + 1) & 63;
        const uint64_t nb = uint64_t(uint32_t(__shfl(int(uint32_t(my_b)), up))) | (uint64_t(uint32_t(__shfl(int(uint32_t(my_b >> 32)), up))) << 32);
        const uint64_t no = uint64_t(uint32_t(__shfl(int(uint32_t(my_o)), up))) | (uint64_t(uint32_t(__shfl(int(uint32_t(my_o >> 32)), up))) << 32);
        const bool last = uint32_t(lane) + 1 == K.ns;
        const uint64_t b_next = last ? K.B1 : nb, o_next = last ? K.O1 : no;
        const bool mine = uint32_t(lane) < K.ns;
        const bool empty = mine && b_next <= my_b, bad = mine && (o_next < my_o || o_next > P.total_boundaries);
        if (empty) err |= kErrEmptySentence;
        if (bad) err |= kErrBadOffsets;
        sane = __ballot(empty || bad) == 0 && K.O1 - K.O0 < 0xFFFF0000ull && K.B1 - K.B0 < 0xFFFF0000ull;
        if (!sane) err |= kErrBadOffsets;
    }
    if (uint32_t(lane) < K.ns) L.so[lane] = uint32_t(my_o - K.O0);
    if (lane == 0) L.so[K.ns] = uint32_t(K.O1 - K.O0);
    if (lane < int(kFuseStepBytes / 32)) L.starts[lane] = 0;
    __builtin_amdgcn_wave_barrier();

    // (a wave that has not published its size yet holds up every block behind it: it goes first on its SIMD)
    __builtin_amdgcn_s_setprio(3);
    // ---- pass A: the block's size = its bytes + the escaped bytes + the boundary labels of its label range (+ the tag suffixes)
    uint64_t size = 0;
    if (kDbg && (dbg & 8u)) size = 3 * (K.B1 - K.B0);
    else if (sane) {
        uint32_t added = 0;
        const uintptr_t l_lo = reinterpret_cast<uintptr_t>(P.labels) + K.O0, l_hi = reinterpret_cast<uintptr_t>(P.labels) + K.O1;
        const uintptr_t t_lo = reinterpret_cast<uintptr_t>(P.text) + K.B0, t_hi = reinterpret_cast<uintptr_t>(P.text) + K.B1;
        for (uintptr_t a = (t_lo & ~uintptr_t(15)) + 16u * uint32_t(lane); a < t_hi; a += 4 * kFuseStepBytes) {   // four loads in flight
            uint4 x[4];
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) x[q] = a + q * kFuseStepBytes < t_hi ? *reinterpret_cast<const uint4*>(a + q * kFuseStepBytes) : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) added += uint32_t(__popc(esc16(x[q]) & in_range16(a + q * kFuseStepBytes, t_lo, t_hi)));
        }
        for (uintptr_t a = (l_lo & ~uintptr_t(15)) + 16u * uint32_t(lane); a < l_hi; a += kFuseStepBytes) {
            const uint4 y = *reinterpret_cast<const uint4*>(a);
            const uint32_t m = in_range16(a, l_lo, l_hi);
            added += uint32_t(__popc(one16(y) & m));
            if (unk16(y) & m) err |= kErrUnknownLabel;
        }
        if (kTags && !(kDbg && (dbg & 128u))) added += fuse_tag_bytes(P, K, lane);
        size = (K.B1 - K.B0) + wave_sum64(added);
    }

    // ---- the block's position: the earlier blocks' words, 64 per trip
    constexpr uint64_t kVal = (uint64_t(1) << 62) - 1;
    if (lane == 0) __hip_atomic_store(F.state + blk, (uint64_t(1) << 62) | size, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint64_t base = 0;
    if (kDbg && (dbg & 4u)) base = blk * 3 * (K.B1 - K.B0);
    else for (uint64_t p = blk; p > 0;) {
        const bool have = uint64_t(lane) < p;
        uint64_t w = uint64_t(2) << 62;   // in front of block 0: position 0
        if (have) w = __hip_atomic_load(F.state + (p - 1 - uint64_t(lane)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint64_t pending = __ballot((w >> 62) == 0), prefixed = __ballot((w >> 62) == 2);
        const int first = prefixed ? __ffsll((long long)prefixed) - 1 : 64;   // the nearest block whose position is known
        const uint64_t need = first < 63 ? (uint64_t(2) << first) - 1 : ~uint64_t(0);
        if (pending & need) { __builtin_amdgcn_s_sleep(2); continue; }         // not all published yet: look again
        base += wave_sum64(lane <= first ? (w & kVal) : 0);
        if (first < 64) break;
        p -= 64;
    }
    if (lane == 0) __hip_atomic_store(F.state + blk, (uint64_t(2) << 62) | ((base + size) & kVal), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_setprio(0);
    const uint64_t end = base + size;
    const bool store_ok = end <= P.capacity;
    if (blk == F.n_blocks - 1 && lane == 0) {
        P.out_offsets[P.n_sent] = end;
        if (end > P.capacity) err |= kErrOutputTooSmall;
        if (F.total_out) *F.total_out = end;
    }

    // ---- pass B
    if (kDbg && (dbg & 32u)) return;
    if (sane) fuse_walk<kTags, kDbg>(P, K, my_b, lane, L, base, end, store_ok, err, dbg);
    else if (uint32_t(lane) < K.ns) P.out_offsets[K.i0 + lane] = base;
    if (err) atomicOr(P.status, err);
}

// ------------------------------------------------------------------------------------------------------------
// emit_flat_kernel (round 5): the writer without tags, FLAT over runs of sentences like count_chars_kernel / decode_chars_kernel.
//
// A WORKGROUP takes a run of `per_block` consecutive sentences (at most 256: about 16 KB of text); its text bytes and its labels are
// two contiguous ranges.  What it will write is a plain reduction -- the run's bytes + its escaped bytes + its boundary labels -- streamed
// with four 16-byte loads per thread in flight; ONE look-back per workgroup (wave 0, 64 words per trip) places the run; then the run is
// walked in pieces of 4 KB, sixteen bytes per thread: lead / escape / sentence-start masks, one block prefix sum numbers the threads' chars
// and sentences (which names their labels in the window of labels staged in LDS with the piece), a second one places their output, which is
// assembled in LDS and leaves as aligned 16-byte stores.  Same checks, same error bits, same output as emit_fused_kernel<false> (which stays
// as the tagged writer and as the A/B: VPT_EMIT_WAVE_BLOCKS); what changes is the shape: 256 threads share a step's prefix sums and barriers
// where a wave did them alone for 1 KB, and a workgroup looks back once where four waves did.
// ------------------------------------------------------------------------------------------------------------
constexpr uint32_t kFlatPiece = kEmitThreads * 16;                 // text bytes of a workgroup's step
constexpr uint32_t kFlatStageBytes = 3 * kFlatPiece + 32;          // its output at most (every byte escaped, a space per char) + the alignment head
struct alignas(16) FlatLds {
    uint32_t stage[kFlatStageBytes / 4];
    uint32_t labs[(kFlatPiece + 64) / 4];     // the labels a piece's chars can ask for, from a 16-byte aligned address
    uint32_t starts[kFlatPiece / 32];         // one bit per byte of the piece: a sentence starts here
    uint32_t so[kEmitFlatMaxBlock + 1];       // the run's boundary offsets, relative to its first
    uint32_t dump[kEmitThreads];              // where a thread's stores of bytes that are not there go
    uint32_t wtot[kEmitWaves];
    uint32_t flags;                           // OR of the threads' "my offsets are no offsets"
    uint64_t red[kEmitWaves];
    uint64_t bcast[4];                        // ticket, B0, O0, base
};

// exclusive prefix sum of x over the workgroup's threads (two packed 16-bit counts or one 32-bit one); *total = the sum
__device__ __forceinline__ uint32_t flat_block_scan(uint32_t x, uint32_t* wtot, uint32_t lane, uint32_t wave, uint32_t* total) {
    const uint32_t incl = wave_inclusive_scan(x);
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
#pragma unroll
    for (uint32_t k = 0; k < uint32_t(kEmitWaves); ++k) {
        const uint32_t u = wtot[k];
        if (k < wave) woff += u;
        tot += u;
    }
    __syncthreads();   // wtot is written again by the next sum
    *total = tot;
    return woff + incl - x;
}

template <bool kTags>
__global__ __launch_bounds__(kEmitThreads) void emit_flat_kernel(const EmitParams P, const EmitFuse F) {
    __shared__ FlatLds L;
    // the other array of state words, for the call after this one
    for (uint64_t k = uint64_t(blockIdx.x) * kEmitThreads + threadIdx.x; k < F.clear_n; k += uint64_t(gridDim.x) * kEmitThreads) F.clear[k] = 0;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
    if (tid == 0) { L.bcast[0] = atomicAdd(reinterpret_cast<unsigned long long*>(F.state + F.n_blocks), 1ull); L.flags = 0; }
    if (tid < kFlatPiece / 32) L.starts[tid] = 0;
    __syncthreads();
    const uint64_t blk = L.bcast[0];
    if (blk >= F.n_blocks) return;
    const uint64_t i0 = blk * F.per_block;
    const uint32_t ns = uint32_t(P.n_sent - i0 < F.per_block ? P.n_sent - i0 : F.per_block);
    // the run's offsets: thread j holds sentence i0 + j's and its successor's
    uint64_t my_b = ~uint64_t(0), my_o = 0, nx_b = 0, nx_o = 0;
    const bool mine = tid < ns;
    if (mine) { my_b = P.boff[i0 + tid]; my_o = P.ooff[i0 + tid]; nx_b = P.boff[i0 + tid + 1]; nx_o = P.ooff[i0 + tid + 1]; }
    if (tid == 0) { L.bcast[1] = my_b; L.bcast[2] = my_o; }
    if (tid == ns - 1) { L.red[0] = nx_b; L.red[1] = nx_o; }
    uint32_t err = 0;
    {
        const bool empty = mine && nx_b <= my_b, bad = mine && (nx_o < my_o || nx_o > P.total_boundaries);
        if (empty) err |= kErrEmptySentence;
        if (bad) err |= kErrBadOffsets;
        if (empty || bad) atomicOr(&L.flags, 1u);
    }
    __syncthreads();
    const uint64_t B0 = L.bcast[1], O0 = L.bcast[2], B1 = L.red[0], O1 = L.red[1];
    const bool sane = L.flags == 0 && O1 - O0 < 0xFFFF0000ull && B1 - B0 < 0xFFFF0000ull;
    if (!sane) err |= kErrBadOffsets;
    if (mine) L.so[tid] = uint32_t(my_o - O0);
    if (tid == 0) L.so[ns] = uint32_t(O1 - O0);
    __syncthreads();   // (red[] is used again below)

    // ---- the run's size = its bytes + the escaped bytes + the boundary labels of its label range
    const uintptr_t t_lo = reinterpret_cast<uintptr_t>(P.text) + B0, t_hi = reinterpret_cast<uintptr_t>(P.text) + B1;
    const uintptr_t l_all = reinterpret_cast<uintptr_t>(P.labels), l_end = l_all + P.total_boundaries;
    uint64_t size = 0;
    if (sane) {
        uint32_t added = 0;
        const uintptr_t l_lo = l_all + O0, l_hi = l_all + O1;
        for (uintptr_t a = (t_lo & ~uintptr_t(15)) + 16u * tid; a < t_hi; a += 4 * kFlatPiece) {   // four loads in flight
            uint4 x[4];
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) x[q] = a + q * kFlatPiece < t_hi ? *reinterpret_cast<const uint4*>(a + q * kFlatPiece) : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) added += uint32_t(__popc(esc16(x[q]) & in_range16(a + q * kFlatPiece, t_lo, t_hi)));
        }
        for (uintptr_t a = (l_lo & ~uintptr_t(15)) + 16u * tid; a < l_hi; a += kFlatPiece) {
            const uint4 y = *reinterpret_cast<const uint4*>(a);
            const uint32_t m = in_range16(a, l_lo, l_hi);
            added += uint32_t(__popc(one16(y) & m));
            if (unk16(y) & m) err |= kErrUnknownLabel;
        }
        if (kTags) {   // the bytes of the run's tag suffixes: fill_tags left them in the token word of every token's last char (layout.h)
            const uint64_t g0 = O0 + i0, n_chars = (O1 - O0) + ns;
            for (uint64_t c0 = 4 * uint64_t(tid); c0 < n_chars; c0 += 4 * kEmitThreads) {
                int32_t m4[4];
                __builtin_memcpy(m4, P.tok_model + g0 + c0, sizeof(m4));   // (the array goes on behind the batch's last char)
#pragma unroll
                for (uint32_t q = 0; q < 4; ++q)
                    if (c0 + q < n_chars) added += tag_suffix_bytes(P, g0 + c0 + q, uint32_t(m4[q]));
            }
        }
        const uint64_t ws = wave_sum64(added);
        if (lane == 0) L.red[wave] = ws;
        __syncthreads();
        size = B1 - B0;
#pragma unroll
        for (uint32_t k = 0; k < uint32_t(kEmitWaves); ++k) size += L.red[k];
    }
    // ---- the run's position: wave 0 looks back over the earlier runs' words, 64 per trip
    if (wave == 0) {
        constexpr uint64_t kVal = (uint64_t(1) << 62) - 1;
        if (lane == 0) __hip_atomic_store(F.state + blk, (uint64_t(1) << 62) | size, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint64_t base = 0;
        for (uint64_t p = blk; p > 0;) {
            const bool have = uint64_t(lane) < p;
            uint64_t w = uint64_t(2) << 62;   // in front of run 0: position 0
            if (have) w = __hip_atomic_load(F.state + (p - 1 - uint64_t(lane)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint64_t pending = __ballot((w >> 62) == 0), prefixed = __ballot((w >> 62) == 2);
            const int first = prefixed ? __ffsll((long long)prefixed) - 1 : 64;   // the nearest run whose position is known
            const uint64_t need = first < 63 ? (uint64_t(2) << first) - 1 : ~uint64_t(0);
            if (pending & need) { __builtin_amdgcn_s_sleep(2); continue; }         // not all published yet: look again
            base += wave_sum64(int(lane) <= first ? (w & kVal) : 0);
            if (first < 64) break;
            p -= 64;
        }
        if (lane == 0) {
            __hip_atomic_store(F.state + blk, (uint64_t(2) << 62) | ((base + size) & kVal), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            L.bcast[3] = base;
        }
    }
    __syncthreads();
    const uint64_t base = L.bcast[3], end = base + size;
    const bool store_ok = end <= P.capacity;
    if (blk == F.n_blocks - 1 && tid == 0) {
        P.out_offsets[P.n_sent] = end;
        if (end > P.capacity) err |= kErrOutputTooSmall;
        if (F.total_out) *F.total_out = end;
    }
    if (!sane) {
        if (mine) P.out_offsets[i0 + tid] = base;
        if (err) atomicOr(P.status, err);
        return;
    }

    // ---- the pieces: every byte of the run to its place
    uint8_t* const sbytes = reinterpret_cast<uint8_t*>(L.stage);
    const uintptr_t my_start = reinterpret_cast<uintptr_t>(P.text) + my_b;
    uint64_t at_out = base, cb = 0, sb = 0;   // output position, chars and sentence starts of the run in front of the piece
    bool fits = true;
    for (uintptr_t piece = t_lo & ~uintptr_t(15); piece < t_hi; piece += kFlatPiece) {
        const uintptr_t addr = piece + 16u * tid;
        const uint32_t vm = in_range16(addr, t_lo, t_hi);
        const uint4 x = vm ? *reinterpret_cast<const uint4*>(addr) : make_uint4(0, 0, 0, 0);
        // the labels the piece's chars can ask for: label (O0 + cb - sb) onwards (every char but a sentence's first has one in front)
        const uintptr_t lab_at = l_all + O0 + (cb - sb), lab_al = lab_at & ~uintptr_t(15);
        const uint32_t lab_head = uint32_t(lab_at - lab_al);
        {
            const uintptr_t a = lab_al + 16u * tid, a2 = lab_al + 16u * (uint32_t(kEmitThreads) + tid);
            reinterpret_cast<uint4*>(L.labs)[tid] = (a + 16 > l_all && a < l_end) ? *reinterpret_cast<const uint4*>(a) : make_uint4(0, 0, 0, 0);
            if (tid < 4) reinterpret_cast<uint4*>(L.labs)[kEmitThreads + tid] = (a2 + 16 > l_all && a2 < l_end) ? *reinterpret_cast<const uint4*>(a2) : make_uint4(0, 0, 0, 0);
        }
        if (mine && my_start >= piece && my_start - piece < kFlatPiece) {
            const uint32_t r = uint32_t(my_start - piece);
            atomicOr(&L.starts[r >> 5], 1u << (r & 31u));
        }
        __syncthreads();
        uint32_t sm = (L.starts[tid >> 1] >> (16 * (tid & 1))) & 0xFFFFu;
        const uint32_t lm = lead16(x) & vm, em = esc16(x) & vm;
        if (sm & ~lm) err |= kErrBadOffsets;   // a sentence that starts inside a char (or outside the run)
        sm &= lm;
        const uint32_t nl = uint32_t(__popc(lm)), nst = uint32_t(__popc(sm));
        uint32_t tot;
        const uint32_t excl = flat_block_scan(nl | (nst << 16), L.wtot, lane, wave, &tot);   // (its barriers: every thread has read its starts)
        if (tid < kFlatPiece / 32) L.starts[tid] = 0;
        const uint32_t c_in = excl & 0xFFFFu, s_in = excl >> 16;   // chars / starts of the piece in front of this thread
        // the thread's chars that have a label in front take consecutive labels from (c_in - s_in) of the window on
        const uint32_t nm = lm & ~sm;
        uint32_t spm = 0;
        {
            const uint32_t loff = lab_head + (c_in - s_in);           // byte offset in labs: <= 15 + 4096
            const uint32_t d = loff >> 2, r = loff & 3u;
            uint4 y;
            y.x = __builtin_amdgcn_alignbyte(L.labs[d + 1], L.labs[d], r); y.y = __builtin_amdgcn_alignbyte(L.labs[d + 2], L.labs[d + 1], r);
            y.z = __builtin_amdgcn_alignbyte(L.labs[d + 3], L.labs[d + 2], r); y.w = __builtin_amdgcn_alignbyte(L.labs[d + 4], L.labs[d + 3], r);
            uint32_t bits = one16(y), rem = nm;
            while (rem) {   // label q of the thread onto its q-th labelled char
                const uint32_t low = rem & (0u - rem);
                if (bits & 1u) spm |= low;
                bits >>= 1;
                rem &= rem - 1u;
            }
        }
        // Tag suffixes go in front of a space and in front of a sentence's first byte (the last token of the sentence before it), except
        // the run's first (the run before this one wrote that one behind its last byte).  The thread that holds the byte in FRONT of which
        // a suffix goes owns it; at most two per thread are carried in registers (tk: the byte, tl: the length, tc: the token's last
        // char), a third sends the thread's WAVE through its chars one by one (emit_fused_kernel's scheme).
        uint32_t tmask = 0, tk1 = 16, tl1 = 0, tk2 = 16, tl2 = 0, tc1 = 0, tc2 = 0;
        int32_t tm1 = 0, tm2 = 0;
        bool many = false;
        const uint64_t g_first = O0 + i0 + cb + c_in;     // batch-flat index of the thread's first char
        if (kTags) {
            tmask = spm | sm;
            if (sb + s_in == 0 && sm) tmask &= ~(sm & (0u - sm));
            if (tmask) {
                // the tag models of the chars in front of the thread's chars: tok_model[g_first - 1 + j] for its j-th char (the array has
                // zeros in front of the batch's first char and behind its last: capi.cpp); which of them are there at all
                const int32_t* tmod = P.tok_model + g_first - 1;
                uint32_t pm = 0;
                for (uint32_t j0 = 0; j0 < nl; j0 += 4) {
                    int32_t m4[4];
                    __builtin_memcpy(m4, tmod + j0, sizeof(m4));
                    pm |= (((uint32_t(m4[0]) & kTokModelMask) ? 1u : 0u) | ((uint32_t(m4[1]) & kTokModelMask) ? 2u : 0u) | ((uint32_t(m4[2]) & kTokModelMask) ? 4u : 0u) |
                           ((uint32_t(m4[3]) & kTokModelMask) ? 8u : 0u)) << j0;
                }
                pm &= (1u << nl) - 1u;
                while (pm) {   // few
                    const uint32_t j = uint32_t(__ffs(int(pm))) - 1u;
                    pm &= pm - 1u;
                    uint32_t remj = lm;
                    for (uint32_t q = 0; q < j; ++q) remj &= remj - 1u;
                    const uint32_t k = uint32_t(__ffs(int(remj))) - 1u;          // the byte of the thread's j-th char
                    if (!((tmask >> k) & 1u)) continue;                          // no token ends in front of it
                    const uint32_t word = uint32_t(tmod[j]);
                    const int32_t mdl = int32_t(word & kTokModelMask);
                    const uint32_t len = tag_suffix_bytes(P, g_first + j - 1u, word);   // (carried from fill_tags)
                    if (!len) continue;
                    if (tk1 == 16) { tk1 = k; tl1 = len; tc1 = j; tm1 = mdl; }
                    else if (tk2 == 16) { tk2 = k; tl2 = len; tc2 = j; tm2 = mdl; }
                    else many = true;
                }
            }
        }
        const bool slow = kTags && __ballot(many) != 0;        // (wave-uniform)
        uint32_t sfx_total = tl1 + tl2;
        if (slow) {
            sfx_total = 0;
            uint32_t todo = tmask;
            while (todo) {
                const uint32_t low = todo & (0u - todo);
                todo &= todo - 1u;
                sfx_total += tag_suffix(P, g_first + uint32_t(__popc(lm & (low - 1u))) - 1u, nullptr);
            }
        }
        const uint32_t t = uint32_t(__popc(vm)) + uint32_t(__popc(spm)) + uint32_t(__popc(em)) + sfx_total;
        uint32_t total;
        const uint32_t w = flat_block_scan(t, L.wtot, lane, wave, &total);
        if (at_out + total > end) { fits = false; break; }   // (the same in every thread)
        uint8_t* const dst = P.out_text + at_out;
        const uint32_t head = uint32_t(reinterpret_cast<uintptr_t>(dst) & 15u);
        const bool staged = !kTags || head + total <= kFlatStageBytes;   // (the same in every thread) else: byte stores straight to the output
        if (!slow) {
            if (store_ok) {   // the thread's bytes in order: [tags] [' '] ['\\'] byte.  No branches: what is not there goes to a slot of the thread's own
                uint8_t* const o = !kTags || staged ? sbytes + head : dst + 0;   // (with tags: a generic pointer)
                uint8_t* const dump = reinterpret_cast<uint8_t*>(L.dump + tid);
                uint32_t pos = w;
#pragma unroll
                for (uint32_t k = 0; k < 16; ++k) {
                    const uint32_t v = (vm >> k) & 1u, sp = (spm >> k) & 1u, es = (em >> k) & 1u;
                    if (kTags) pos += (k == tk1 ? tl1 : 0u) + (k == tk2 ? tl2 : 0u);
                    *(sp ? o + pos : dump) = 0x20u; pos += sp;
                    *(es ? o + pos : dump) = 0x5Cu; pos += es;
                    *(v ? o + pos : dump) = uint8_t(byte_of(x, k)); pos += v;
                }
                if (kTags && tl1) {   // the tags themselves (few threads)
                    const uint32_t b1 = (1u << tk1) - 1u, b2 = (1u << tk2) - 1u;
                    if (tag_suffix_of(P, g_first + tc1 - 1u, tm1, o + w + uint32_t(__popc(vm & b1)) + uint32_t(__popc(spm & b1)) + uint32_t(__popc(em & b1)), tl1) != tl1) err |= kErrBadOffsets;
                    if (tl2 && tag_suffix_of(P, g_first + tc2 - 1u, tm2, o + w + tl1 + uint32_t(__popc(vm & b2)) + uint32_t(__popc(spm & b2)) + uint32_t(__popc(em & b2)), tl2) != tl2) err |= kErrBadOffsets;
                }
            }
            uint32_t rem = sm;   // the sentences that start in the thread's bytes (few threads, one as a rule)
            while (rem) {
                const uint32_t k = uint32_t(__ffs(int(rem))) - 1u, below = (1u << k) - 1u;
                rem &= rem - 1u;
                const uint64_t s = sb + s_in + uint32_t(__popc(sm & below));
                if (s < ns) {
                    P.out_offsets[i0 + s] = at_out + w + uint32_t(__popc(vm & below)) + uint32_t(__popc(spm & below)) + uint32_t(__popc(em & below)) +
                                            (kTags ? (tk1 <= k ? tl1 : 0u) + (tk2 <= k ? tl2 : 0u) : 0u);
                    if (cb + c_in + uint32_t(__popc(lm & below)) != uint64_t(L.so[s]) + s) err |= kErrBadOffsets;   // not the char its offset names
                } else err |= kErrBadOffsets;
            }
        } else {
            // [tags] [' '] | sentence offset | ['\\'] byte, char by char (o: LDS or the output itself)
            uint8_t* const o = !store_ok ? nullptr : staged ? sbytes + head : dst + 0;
            uint32_t pos = w, ci = 0;
#pragma unroll 1
            for (uint32_t k = 0; k < 16; ++k) {
                if (!((vm >> k) & 1u)) continue;
                if ((tmask >> k) & 1u) pos += tag_suffix(P, g_first + ci - 1u, o ? o + pos : nullptr);
                if ((spm >> k) & 1u) { if (o) o[pos] = 0x20u; ++pos; }
                if ((sm >> k) & 1u) {
                    const uint64_t s = sb + s_in + uint32_t(__popc(sm & ((1u << k) - 1u)));
                    if (s < ns) {
                        P.out_offsets[i0 + s] = at_out + pos;
                        if (cb + c_in + ci != uint64_t(L.so[s]) + s) err |= kErrBadOffsets;
                    } else err |= kErrBadOffsets;
                }
                if ((em >> k) & 1u) { if (o) o[pos] = 0x5Cu; ++pos; }
                if (o) o[pos] = uint8_t(byte_of_rt(x, k));
                ++pos;
                ci += (lm >> k) & 1u;
            }
        }
        __syncthreads();
        if (store_ok && staged) {   // LDS byte j is output byte j - head: whole 16-byte chunks leave aligned, the two edges byte by byte
            uint8_t* const abase = dst - head;
            const uint32_t nd = (head + total + 15u) >> 4;
            for (uint32_t d = tid; d < nd; d += kEmitThreads) {
                const uint32_t lo = d * 16u, hi = lo + 16u;
                if (lo >= head && hi <= head + total) {
                    *reinterpret_cast<uint4*>(abase + lo) = reinterpret_cast<const uint4*>(L.stage)[d];
                } else {
                    const uint32_t a = lo > head ? lo : head, b = hi < head + total ? hi : head + total;
                    for (uint32_t j = a; j < b; ++j) abase[j] = sbytes[j];
                }
            }
        }
        __syncthreads();   // the next piece rewrites stage / labs / starts
        at_out += total;
        cb += tot & 0xFFFFu;
        sb += tot >> 16;
    }
    if (kTags && fits) {   // the tags of the run's last token
        const uint64_t g_last = O1 + i0 + ns - 1;
        const uint32_t sl = tag_suffix(P, g_last, nullptr);   // (every thread computes the same)
        if (sl && store_ok && at_out + sl <= end && tid == 0) tag_suffix(P, g_last, P.out_text + at_out);
        at_out += sl;
    }
    // (what was written is what the size pass said: anything else means chars, labels, offsets -- or the tags' token words and the labels,
    // which must be the ones fill_tags saw -- do not belong together)
    if (!fits || at_out != end || cb != (O1 - O0) + ns || sb != ns) err |= kErrBadOffsets;
    if (err) atomicOr(P.status, err);
}

// vpt_count_boundaries on the device: chars - 1 of every sentence -> offsets[i + 1] (the scan follows), the same
// validation as Sentence::from_raw (sentence.rs:160-196), the longest sentence (in chars) -> *max_chars.
//
// FLAT over the text (round 4; a wave per sentence took 0.38 ms for configs[1]'s 19 MB -- three dependent trips to memory per sentence
// -- and was the longest kernel of vpt_tokenize_batch, profiles/r04_d_tokenize_timeline.txt): a workgroup takes `per_block` consecutive
// sentences and streams their bytes in pieces of 16 KB, 64 contiguous bytes per thread; a block-wide prefix sum over the threads' lead
// counts and the chunks' lead masks in LDS turn "leads in front of byte x" into two LDS reads, which the thread of every sentence asks
// for its first byte and for the byte behind its last.
constexpr uint32_t kCountPiece = kEmitThreads * 64;   // bytes per piece
__global__ __launch_bounds__(kEmitThreads) void count_chars_kernel(const uint8_t* __restrict__ text, const uint64_t* __restrict__ boff,
                                                                   uint64_t n_sent, uint64_t* __restrict__ offsets, uint32_t* __restrict__ status,
                                                                   uint32_t* __restrict__ max_chars, uint64_t* scan_state, uint32_t per_block) {
    __shared__ uint16_t masks[kEmitThreads * 4];   // lead mask of every 16-byte chunk of the piece
    __shared__ uint32_t pfx[kEmitThreads];         // leads of the piece in front of the thread's 64 bytes
    __shared__ uint32_t wtot[kEmitWaves];
    clear_scan_state(scan_state, n_sent);
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
    const uint64_t s0 = uint64_t(blockIdx.x) * per_block;
    if (s0 >= n_sent) return;
    const uint32_t ns = uint32_t(n_sent - s0 < per_block ? n_sent - s0 : per_block);
    const uint64_t B0 = boff[s0], B1 = boff[s0 + ns];
    uint64_t my_b = 0, my_e = 0;
    if (tid < ns) { my_b = boff[s0 + tid]; my_e = boff[s0 + tid + 1]; }
    uint32_t err = 0;
    const bool mine = tid < ns;
    const bool sane = mine && my_e > my_b && my_b >= B0 && my_e <= B1;   // (offsets that are not non-decreasing: reported, nothing read for them)
    if (mine && !sane) err |= my_e <= my_b ? kErrEmptySentence : kErrBadOffsets;
    uint64_t p_start = 0, p_end = 0, carry = 0;
    bool nul = false;
    const uintptr_t t_lo = reinterpret_cast<uintptr_t>(text) + B0, t_hi = reinterpret_cast<uintptr_t>(text) + B1;
    for (uintptr_t piece = t_lo & ~uintptr_t(15); piece < t_hi; piece += kCountPiece) {
        const uintptr_t mine_at = piece + 64u * tid;
        uint4 v[4];
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            const uintptr_t a = mine_at + 16u * q;
            v[q] = (a + 16 > t_lo && a < t_hi) ? *reinterpret_cast<const uint4*>(a) : make_uint4(0, 0, 0, 0);
        }
        uint32_t cnt = 0;
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            const uintptr_t a = mine_at + 16u * q;
            const uint32_t lo = t_lo > a ? (t_lo - a < 16 ? uint32_t(t_lo - a) : 16u) : 0u;
            const uint32_t hi = t_hi > a ? (t_hi - a < 16 ? uint32_t(t_hi - a) : 16u) : 0u;
            const uint32_t vm = hi > lo ? ((1u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;
            const uint32_t lm = (lead_nibble(v[q].x) | (lead_nibble(v[q].y) << 4) | (lead_nibble(v[q].z) << 8) | (lead_nibble(v[q].w) << 12)) & vm;
            const uint32_t zm = (byte_flags_to_nibble(zero_bytes(v[q].x)) | (byte_flags_to_nibble(zero_bytes(v[q].y)) << 4) |
                                 (byte_flags_to_nibble(zero_bytes(v[q].z)) << 8) | (byte_flags_to_nibble(zero_bytes(v[q].w)) << 12)) & vm;
            nul = nul || zm != 0;
            masks[tid * 4 + q] = uint16_t(lm);
            cnt += uint32_t(__popc(lm));
        }
        const uint32_t incl = wave_inclusive_scan(cnt);
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        uint32_t woff = 0, total = 0;
#pragma unroll
        for (uint32_t k = 0; k < uint32_t(kEmitWaves); ++k) {
            const uint32_t u = wtot[k];
            if (k < wave) woff += u;
            total += u;
        }
        pfx[tid] = woff + incl - cnt;
        __syncthreads();
        // leads of the piece in front of byte x (piece <= x <= piece + kCountPiece)
        auto before = [&](uintptr_t x) -> uint32_t {
            const uint32_t r = uint32_t(x - piece);
            if (r >= kCountPiece) return total;
            const uint32_t t = r >> 6, q = (r >> 4) & 3u, bit = r & 15u;
            uint32_t c = pfx[t];
            for (uint32_t qq = 0; qq < q; ++qq) c += uint32_t(__popc(uint32_t(masks[t * 4 + qq])));
            return c + uint32_t(__popc(uint32_t(masks[t * 4 + q]) & ((1u << bit) - 1u)));
        };
        if (sane) {
            const uintptr_t xs = reinterpret_cast<uintptr_t>(text) + my_b, xe = reinterpret_cast<uintptr_t>(text) + my_e;
            if (xs >= piece && xs - piece < kCountPiece) p_start = carry + before(xs);
            if (xe > piece && xe - piece <= kCountPiece) p_end = carry + before(xe);
        }
        carry += total;
        __syncthreads();   // the next piece rewrites masks / pfx / wtot
    }
    if (__ballot(nul) != 0) err |= kErrNulChar;
    uint32_t longest = 0;
    if (mine) {
        const uint64_t chars = sane ? p_end - p_start : 0;
        if (chars == 0) err |= kErrEmptySentence;
        if (chars > 0xFFFFFFFFull) err |= kErrBadOffsets;
        longest = chars > 0xFFFFFFFFull ? 0xFFFFFFFFu : uint32_t(chars);
        offsets[s0 + tid + 1] = chars > 0 ? chars - 1 : 0;
    }
    if (err) atomicOr(status, err);
    if (max_chars) {   // (optional output; the caller clears it)
        longest = wave_max(longest);
        if (lane == 0 && longest) atomicMax(max_chars, longest);
    }
}

}  // namespace

size_t scan_part_entries(uint64_t n) { return size_t((n + kScanBlock - 1) / kScanBlock) + 2; }   // the blocks' words + the ticket

// one wave per sentence, but no more workgroups than `max_blocks` (a few generations of what the device runs at a time; 0: 65536):
// the waves then stride over the batch and the sentences' lengths even out (see launch_tag_tokens)
static uint32_t emit_blocks(uint64_t n_sent, uint32_t max_blocks) {
    const uint64_t want = (n_sent + kEmitWaves - 1) / kEmitWaves, cap = max_blocks ? max_blocks : 65536;
    return uint32_t(want < 1 ? 1 : want > cap ? cap : want);
}

hipError_t launch_count_boundaries(const uint8_t* text, const uint64_t* boff, uint64_t n_sent, uint64_t* ooff_out, uint64_t* scan_part, uint32_t* status,
                                   uint32_t* max_chars, uint64_t text_bytes_hint, hipStream_t stream) {
    // sentences per workgroup: about 64 KB of text when the caller knows how much text there is (the device entry point does not: then 32 sentences);
    // at most one per thread
    uint64_t per = 32;
    if (text_bytes_hint && n_sent) per = std::min<uint64_t>(std::max<uint64_t>((uint64_t(65536) * n_sent + text_bytes_hint / 2) / std::max<uint64_t>(text_bytes_hint, 1), 1), kEmitThreads);
    const uint64_t blocks = (n_sent + per - 1) / per;
    hipLaunchKernelGGL(count_chars_kernel, dim3(uint32_t(blocks)), dim3(kEmitThreads), 0, stream, text, boff, n_sent, ooff_out, status, max_chars, scan_part, uint32_t(per));
    return launch_scan(ooff_out, n_sent, scan_part, ~uint64_t(0), status, nullptr, stream);
}

hipError_t launch_emit_tokenized(const EmitParams& P, const EmitFuse& F, hipStream_t stream) {
    if (F.flat) {   // (no timing ablations: capi.cpp) a workgroup per run of sentences
        if (P.tags) hipLaunchKernelGGL(emit_flat_kernel<true>, dim3(uint32_t(F.n_blocks)), dim3(kEmitThreads), 0, stream, P, F);
        else hipLaunchKernelGGL(emit_flat_kernel<false>, dim3(uint32_t(F.n_blocks)), dim3(kEmitThreads), 0, stream, P, F);
        return hipGetLastError();
    }
    const dim3 grid(uint32_t((F.n_blocks + kEmitWaves - 1) / kEmitWaves));
    if (F.dbg) {
        if (P.tags) hipLaunchKernelGGL((emit_fused_kernel<true, true>), grid, dim3(kEmitThreads), 0, stream, P, F);
        else hipLaunchKernelGGL((emit_fused_kernel<false, true>), grid, dim3(kEmitThreads), 0, stream, P, F);
    } else if (P.tags) {
        // the tagged instance for 4 waves per SIMD: 128 VGPRs + 176 bytes of scratch per lane (the rare suffix routines' call frames) -- left to
        // itself the compiler takes 135 VGPRs = 3 waves per SIMD.  configs[4], 444 MB of tagged text (profiles/r04_u_tagged_writer.jsonl):
        // 3 waves 2.72 ms, 4 waves 2.40, 5 waves (96 VGPRs, 304 bytes) 2.57, 6 waves (80, 384) 2.74
        hipLaunchKernelGGL((emit_fused_kernel<true, false, 4>), grid, dim3(kEmitThreads), 0, stream, P, F);
    }
    else hipLaunchKernelGGL((emit_fused_kernel<false, false>), grid, dim3(kEmitThreads), 0, stream, P, F);
    return hipGetLastError();
}

}  // namespace vpt
