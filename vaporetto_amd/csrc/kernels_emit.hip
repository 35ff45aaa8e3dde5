// Token emission on the device: Sentence::write_tokenized_text for a batch (sentence.rs:850-886), boundary part:
// the tokens of a sentence are the runs between WordBoundary labels (sentence.rs:1270-1300), written in order with
// one ' ' between them and a '\' in front of every ' ', '\' and '/' byte of a surface; with tags (the indices
// vpt_fill_tags_batch wrote and the tag model it found for every token) each token is followed by "/tag" for its tag
// slots up to the last Some, an empty string for a None in between (sentence.rs:866-881).  (Unknown boundaries only
// come from partially annotated corpora, never from predict: they are rejected here, kErrUnknownLabel.)
//
// Output size is data dependent: emit_flat_kernel (below) sizes, places and writes the batch in one launch (three launches --
// count, prefix sum, write, a wave per sentence -- until round 3: profiles/r02_j_emit_kernels.txt, r03_l_emit_kernel_stats.csv; a wave
// per block of sentences, emit_fused_kernel, in rounds 3 - 5: HISTORY.md).  With tags the suffixes come from the RECORDS fill_tags left
// (one per token that has a tag model, sorted by position: TagParams, kernels.hpp), not from a dense array.
// count_chars_kernel + scan_chained_kernel are vpt_count_boundaries on the device.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "device_common.h"
#include "kernels.hpp"
#include <type_traits>

namespace vpt {
namespace {

constexpr int kEmitThreads = 256;
constexpr int kEmitWaves = kEmitThreads / 64;
constexpr int kScanThreads = 256, kScanPer = 16;
constexpr uint64_t kScanBlock = uint64_t(kScanThreads) * kScanPer;   // offsets one workgroup of the scan takes

// "/tag/tag.." of the token of record `ri`: `last` slots (up to the last Some, sentence.rs:866-881), each a '/' and the chosen candidate's
// string -- rec_str says where it starts in str_bytes and how long it is (an empty one for a None in between); `pre`: the first two slots'
// entries when the caller has them at hand already (LDS), else nullptr.  Returns the bytes it takes; written to `dst` when given -- never more
// than `limit` of them (what was reserved for it).
// (What the suffix routines read of the kernel's parameters, BY VALUE: tag_suffix is a call, and a reference to the parameter block made the
// compiler keep the whole block in scratch memory -- 128 bytes a lane written at the kernel's start, every P.field of the kernel a scratch load;
// found in round 6 by reading the ISA.)
struct TagStrings {
    const uint4* records;
    const uint2* rec_str;
    const uint8_t* str_bytes;
    uint32_t n_tags;
};
__device__ __forceinline__ uint32_t tag_suffix_write(const TagStrings P, uint64_t ri, uint32_t last, const uint2* pre, uint8_t* dst, uint32_t limit = 0xFFFFFFFFu) {
    const uint2* rs = P.rec_str + ri * P.n_tags;
    if (last > P.n_tags) last = P.n_tags;
    uint32_t n = 0;
    for (uint32_t j = 0; j < last; ++j) {
        const uint2 e = (pre && j < 2) ? pre[j] : rs[j];
        if (dst && n < limit) dst[n] = 0x2Fu;
        ++n;
        // (copied byte by byte: sixteen bytes a trip -- through registers, or through five dwords of LDS of the thread's own -- measured no
        // faster, 1.43 - 1.70 against 1.37 ms on configs[4], profiles/r06_h_*, r06_i_*: the strings are a few bytes and sit in the L2)
        if (dst) for (uint32_t q = 0; q < e.y && n + q < limit; ++q) dst[n + q] = P.str_bytes[e.x + q];
        n += e.y;
    }
    return n;
}
// the bytes of the suffix of the token of record ri, whose token word is w (layout.h) and whose suffix has `last` slots: carried from fill_tags
// unless it is a long one
__device__ __forceinline__ uint32_t tag_suffix_bytes(const TagStrings P, uint64_t ri, uint32_t w, uint32_t last) {
    const uint32_t code = w >> kTokSuffixShift;
    return (w & kTokModelMask) == 0 ? 0u : code != kTokSuffixLong ? code : tag_suffix_write(P, ri, last, nullptr, nullptr);
}
__device__ __noinline__ uint32_t tag_suffix(const TagStrings P, uint64_t ri, uint8_t* dst, uint32_t limit = 0xFFFFFFFFu) {   // (the rare paths: everything from HBM)
    const uint4 rec = P.records[ri];
    return (rec.z & kTokModelMask) == 0 ? 0u : tag_suffix_write(P, ri, rec.w, nullptr, dst, limit);
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
            if (base + total > capacity && status) atomicOr(status, kErrOutputTooSmall);
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

}  // namespace
hipError_t launch_scan(uint64_t* offsets, uint64_t n, uint64_t* part, uint64_t capacity, uint32_t* status, uint64_t* total_out, hipStream_t stream) {
    const uint64_t n_part = (n + kScanBlock - 1) / kScanBlock;
    hipLaunchKernelGGL(scan_chained_kernel, dim3(uint32_t(n_part)), dim3(kScanThreads), 0, stream, offsets, n, part, n_part, capacity, status, total_out);
    return hipGetLastError();
}
namespace {
// ------------------------------------------------------------------------------------------------------------
// emit_flat_kernel (round 5): the writer, FLAT over runs of sentences like count_chars_kernel / decode_chars_kernel.
//
// A WORKGROUP takes a run of `per_block` consecutive sentences (at most 256: about 16 KB of text); its text bytes and its labels are
// two contiguous ranges.  What it will write is a plain reduction -- the run's bytes + its escaped bytes + its boundary labels -- streamed
// with four 16-byte loads per thread in flight; ONE look-back per workgroup (wave 0, 64 words per trip) places the run; then the run is
// walked in pieces of 4 KB, sixteen bytes per thread: lead / escape / sentence-start masks, one block prefix sum numbers the threads' chars
// and sentences (which names their labels in the window of labels staged in LDS with the piece), a second one places their output, which is
// assembled in LDS and leaves as aligned 16-byte stores.  (Rounds 3 - 4 gave a block of sentences to a WAVE: 256 threads now share a step's
// prefix sums and barriers where a wave did them alone for 1 KB, and a workgroup looks back once where four waves did.)
// kTags: "/tag" suffixes from the records of fill_tags (round 6).  The records are sorted by position and the front end's runs are runs of
// sentences like this kernel's: the records of a workgroup's chars are ONE contiguous slice (run_pref of its first and last front-end run;
// capi.cpp makes this kernel's runs whole multiples of those, anything else costs the slice's ends a compare).  Its suffix bytes are a
// reduction over the slice; per piece the records whose chars lie in it are MARKED in LDS (a u16 per char: which record) with a coalesced
// read, and a thread asks the marks of its chars -- where rounds 4 - 5 read a dense token word per char from HBM, twice, for one
// token in thirty-five that had tags.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lead16(const uint4& x) { return lead_mask16(x); }
__device__ __forceinline__ uint32_t esc16(const uint4& x) { return esc_mask16(x); }
__device__ __forceinline__ uint32_t one_flags(uint32_t y) { return zero_bytes(y ^ 0x01010101u); }   // bytes equal to 1
__device__ __forceinline__ uint32_t one16(const uint4& y) { return flag_bytes_to_mask16(one_flags(y.x), one_flags(y.y), one_flags(y.z), one_flags(y.w)); }
__device__ __forceinline__ uint32_t unk_flags(uint32_t y) { return ~zero_bytes(y & 0xFEFEFEFEu) & 0x80808080u; }   // bytes above 1
__device__ __forceinline__ uint32_t unk16(const uint4& y) { return flag_bytes_to_mask16(unk_flags(y.x), unk_flags(y.y), unk_flags(y.z), unk_flags(y.w)); }
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
// which of the 16 bytes at offset `off` lie in [lo, hi) (offsets from the same base)
__device__ __forceinline__ uint32_t in_range16_rel(uint32_t off, uint32_t lo, uint32_t hi) {
    const uint32_t a = lo > off ? (lo - off < 16u ? lo - off : 16u) : 0u;
    const uint32_t b = hi > off ? (hi - off < 16u ? hi - off : 16u) : 0u;
    return ((1u << b) - 1u) & ~((1u << a) - 1u);   // (b < a: nothing)
}
constexpr uint32_t kFlatPiece = kEmitThreads * 16;                 // text bytes of a workgroup's step
constexpr uint32_t kFlatStageBytes = 3 * kFlatPiece + 32;          // its output at most (every byte escaped, a space per char) + the alignment head
struct alignas(16) FlatLds {
    uint32_t stage[kFlatStageBytes / 4];
    uint32_t labs[(kFlatPiece + 64) / 4];     // the labels a piece's chars can ask for, from a 16-byte aligned address
    uint32_t starts[kFlatPiece / 32];         // one bit per byte of the piece: a sentence starts here
    uint32_t so[kEmitFlatMaxBlock + 1];       // the run's boundary offsets, relative to its first
    uint32_t dump[kEmitWaves];                // where a wave's stores of bytes that are not there go (nobody reads it)
    uint32_t wtot[kEmitWaves], wtot1[kEmitWaves], wtot2[kEmitWaves];   // the waves' sums: of the sums that come in loops, of a piece's first and of its second
    uint32_t flags;                           // OR of the threads' "my offsets are no offsets"
    uint64_t red[kEmitWaves];
    uint64_t bcast[4];                        // ticket, B0, O0, base
};
// with tags: which record (its number in the piece's records + 1; 0: none; kMarkCarry: the record carried over from the piece before) ends a
// token at the piece's char k - 1, in marks[k]; marks[0]: the last char of the piece before
constexpr uint32_t kMarkCarry = 0xFFFFu;
#ifndef VPT_EMIT_ABLATE
#define VPT_EMIT_ABLATE 0   // timing ablations of the tagged writer (A/B builds, wrong output): 1 the suffixes' bytes are not written, 2 no suffixes at all (the marks stay), 4 no marks either
#endif
constexpr uint32_t kStash = 256;             // records of a piece whose words and first strings wait in LDS (entry kStash: the carried record's)
struct alignas(16) FlatMarks {
    uint16_t m[kFlatPiece + 16];
    uint32_t word[kStash + 1], last[kStash + 1];
    uint2 str[kStash + 1][2];
};

// exclusive prefix sum of x over the workgroup's threads (two packed 16-bit counts or one 32-bit one); *total = the sum.  kAgain: the same wtot
// serves the next sum at once (a loop of sums) -- a second barrier; the sums of a piece that come once have words of their own and take one
template <bool kAgain = true>
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
    if (kAgain) __syncthreads();   // wtot is written again by the next sum
    *total = tot;
    return woff + incl - x;
}

// The chain of the runs' positions (one word per run: flag << 62 | value; 1: the run's size, 2: the position behind it).  A run's size is
// published as soon as it is known; the WAVE that calls place_run walks back over the earlier runs' words, 64 per trip, until one holds a
// position, publishes the run's own and returns where the run starts.
__device__ __forceinline__ void publish_run_size(const EmitFuse& F, uint64_t blk, uint64_t size) {
    __hip_atomic_store(F.state + blk, (uint64_t(1) << 62) | size, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t place_run(const EmitFuse& F, uint64_t blk, uint64_t size, uint32_t lane) {
    constexpr uint64_t kVal = (uint64_t(1) << 62) - 1;
    const uint64_t start = (F.chain_in ? *F.chain_in : 0ull) & kVal;   // where the call's text starts (a call chained behind another: EmitFuse)
    uint64_t base = 0;
    bool anchored = false;   // the sum has reached a run whose position is known (or the front's sentinel): it holds `start`
    for (uint64_t p = blk; p > 0;) {
        const bool have = uint64_t(lane) < p;
        uint64_t w = (uint64_t(2) << 62) | start;   // in front of run 0
        if (have) w = __hip_atomic_load(F.state + (p - 1 - uint64_t(lane)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint64_t pending = __ballot((w >> 62) == 0), prefixed = __ballot((w >> 62) == 2);
        const int first = prefixed ? __ffsll((long long)prefixed) - 1 : 64;   // the nearest run whose position is known
        const uint64_t need = first < 63 ? (uint64_t(2) << first) - 1 : ~uint64_t(0);
        if (pending & need) { __builtin_amdgcn_s_sleep(2); continue; }         // not all published yet: look again
        base += wave_sum64(int(lane) <= first ? (w & kVal) : 0);
        if (first < 64) { anchored = true; break; }
        p -= 64;
    }
    // run 0, or a walk that ran off the front exactly at a multiple of 64 runs with none of them placed yet (then no lane held the sentinel:
    // found on MI355X by the chained chunks of vpt_tokenize_batch with one-sentence runs -- a misplaced run's text landed in an earlier chunk's)
    if (!anchored) base += start;
    if (lane == 0) {
#ifndef VPT_EMIT_NO_PREFIX   // (test builds, tests/test_kernel_emu.py: the runs publish their sizes only, so every look-back walks to the launch's front)
        __hip_atomic_store(F.state + blk, (uint64_t(2) << 62) | ((base + size) & kVal), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    }
    return base;
}

#ifndef VPT_EMIT_TAG_OCC
#define VPT_EMIT_TAG_OCC 4   // waves per SIMD the tagged instance is compiled for at least (A/B builds: -D; 1: what the compiler takes by itself -- 140 VGPRs, 3 waves: 1.54 ms on configs[4] against 1.37 at 4, profiles/r06_g_*)
#endif
#ifndef VPT_EMIT_OCC
#define VPT_EMIT_OCC 8    // ... and the instance without tags
#endif
template <bool kTags>
__global__ __launch_bounds__(kEmitThreads, kTags ? VPT_EMIT_TAG_OCC : VPT_EMIT_OCC) void emit_flat_kernel(const EmitParams P, const EmitFuse F) {
    __shared__ FlatLds L;
    __shared__ FlatMarks MK[1];   // (with tags; the instance without never touches it and the compiler drops it)
    const TagStrings TS{P.records, P.rec_str, P.str_bytes, P.n_tags};
    // the other array of state words, for the call after this one
    for (uint64_t k = uint64_t(blockIdx.x) * kEmitThreads + threadIdx.x; k < F.clear_n; k += uint64_t(gridDim.x) * kEmitThreads) F.clear[k] = 0;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
    if (kTags) {
        reinterpret_cast<uint4*>(MK[0].m)[tid] = make_uint4(0, 0, 0, 0);
        reinterpret_cast<uint4*>(MK[0].m)[kEmitThreads + tid] = make_uint4(0, 0, 0, 0);
        if (tid < 2) reinterpret_cast<uint4*>(MK[0].m)[2 * kEmitThreads + tid] = make_uint4(0, 0, 0, 0);
    }
    if (tid == 0) { L.bcast[0] = atomicAdd(reinterpret_cast<unsigned long long*>(F.state + F.n_blocks), 1ull); L.flags = 0; }
    if (tid < kFlatPiece / 32) L.starts[tid] = 0;
    __syncthreads();
    const uint64_t blk = L.bcast[0];
    if (blk >= F.n_blocks) return;
    const uint64_t i0 = blk * F.per_block;
    const uint32_t ns = uint32_t(P.n_sent - i0 < F.per_block ? P.n_sent - i0 : F.per_block);
    // the run's offsets: thread j holds sentences i0 + j and i0 + kEmitThreads + j (and their successors')
    constexpr uint32_t kMine = kEmitFlatMaxBlock / kEmitThreads;
    static_assert(kMine * kEmitThreads == kEmitFlatMaxBlock, "sentences per thread");
    uint64_t my_b[kMine], my_o[kMine];
    bool mine[kMine];
    uint32_t err = 0;
#pragma unroll
    for (uint32_t j = 0; j < kMine; ++j) {
        const uint32_t s = tid + j * kEmitThreads;
        mine[j] = s < ns;
        uint64_t nx_b = 0, nx_o = 0;
        my_b[j] = ~uint64_t(0); my_o[j] = 0;
        if (mine[j]) { my_b[j] = P.boff[i0 + s]; my_o[j] = P.ooff[i0 + s]; nx_b = P.boff[i0 + s + 1]; nx_o = P.ooff[i0 + s + 1]; }
        if (s == 0) { L.bcast[1] = my_b[j]; L.bcast[2] = my_o[j]; }
        if (s == ns - 1) { L.red[0] = nx_b; L.red[1] = nx_o; }
        const bool empty = mine[j] && nx_b <= my_b[j], bad = mine[j] && (nx_o < my_o[j] || nx_o > P.total_boundaries);
        if (empty) err |= kErrEmptySentence;
        if (bad) err |= kErrBadOffsets;
        if (empty || bad) atomicOr(&L.flags, 1u);
    }
    __syncthreads();
    const uint64_t B0 = L.bcast[1], O0 = L.bcast[2], B1 = L.red[0], O1 = L.red[1];
    const bool sane = L.flags == 0 && O1 - O0 < 0xFFFF0000ull && B1 - B0 < 0xFFFF0000ull;
    if (!sane) err |= kErrBadOffsets;
#pragma unroll
    for (uint32_t j = 0; j < kMine; ++j) if (mine[j]) L.so[tid + j * kEmitThreads] = uint32_t(my_o[j] - O0);
    if (tid == 0) L.so[ns] = uint32_t(O1 - O0);
    __syncthreads();   // (red[] is used again below)

    // ---- the run's size = its bytes + the escaped bytes + the boundary labels of its label range
    const uintptr_t t_lo = reinterpret_cast<uintptr_t>(P.text) + B0, t_hi = reinterpret_cast<uintptr_t>(P.text) + B1;
    const uintptr_t l_all = reinterpret_cast<uintptr_t>(P.labels), l_end = l_all + P.total_boundaries;
    // (the run's bytes from a 16-byte aligned base, in 32 bits: the run is shorter than 4 GB -- `sane`)
    const uintptr_t tb = t_lo & ~uintptr_t(15);
    const uint32_t lo_rel = uint32_t(t_lo - tb), span = sane ? uint32_t(t_hi - tb) : 0u;
    // with tags: the records of the run's chars [g0, g1) are the slice [r_lo, r_hi) of the sorted records
    const uint64_t g0 = O0 + i0, g1 = O1 + i0 + ns;
    uint64_t r_lo = 0, r_hi = 0;
    if (kTags && sane) {
        const uint64_t ra = i0 / P.run_sent, rb = (i0 + ns + P.run_sent - 1) / P.run_sent;
        r_lo = wave_uniform64(P.run_pref[ra < P.n_runs ? ra : P.n_runs]);
        r_hi = wave_uniform64(P.run_pref[rb < P.n_runs ? rb : P.n_runs]);
        if (r_hi < r_lo) r_hi = r_lo;
        if (i0 != ra * P.run_sent) {   // (the same in every thread) a run that starts inside a front-end run: its records begin further on
            for (;;) {
                bool below = false;
                if (r_lo + tid < r_hi) { const uint4 rec = P.records[r_lo + tid]; below = (uint64_t(rec.x) | (uint64_t(rec.y) << 32)) < g0; }
                uint32_t n_below;
                flat_block_scan(below ? 1u : 0u, L.wtot, lane, wave, &n_below);
                r_lo += n_below;
                if (n_below < uint32_t(kEmitThreads)) break;
            }
        }
    }
    uint64_t size = 0;
    if (sane) {
        uint32_t added = 0;
        const uintptr_t l_lo = l_all + O0, l_hi = l_all + O1;
        // (one loop for text and labels, six loads in flight: a run of 48 KB and its 16 KB of labels are three trips to memory -- as two loops of
        // four and one loads they were seven, and the workgroup does nothing else while it waits for its size: profiles/r06_w_*)
        constexpr uint32_t kT = 4, kY = 2;
        const uintptr_t lb = l_lo & ~uintptr_t(15);
        const uint32_t llo = uint32_t(l_lo - lb), lspan = l_hi > l_lo ? uint32_t(l_hi - lb) : 0u;   // (O1 - O0 < 4 GB: `sane`)
        for (uint32_t toff = 16u * tid, loff = 16u * tid; toff < span || loff < lspan; toff += kT * kFlatPiece, loff += kY * kFlatPiece) {
            uint4 x[kT], y[kY];
#pragma unroll
            for (uint32_t q = 0; q < kT; ++q) x[q] = toff + q * kFlatPiece < span ? *reinterpret_cast<const uint4*>(tb + toff + q * kFlatPiece) : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (uint32_t q = 0; q < kY; ++q) y[q] = loff + q * kFlatPiece < lspan ? *reinterpret_cast<const uint4*>(lb + loff + q * kFlatPiece) : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (uint32_t q = 0; q < kT; ++q) added += uint32_t(__popc(esc16(x[q]) & in_range16_rel(toff + q * kFlatPiece, lo_rel, span)));
#pragma unroll
            for (uint32_t q = 0; q < kY; ++q) {
                const uint32_t m = in_range16_rel(loff + q * kFlatPiece, llo, lspan);
                added += uint32_t(__popc(one16(y[q]) & m));
                if (unk16(y[q]) & m) err |= kErrUnknownLabel;
            }
        }
        if (kTags) {   // the bytes of the run's tag suffixes: fill_tags left them in the records' token words (layout.h)
            for (uint64_t r = r_lo + tid; r < r_hi; r += kEmitThreads) {
                const uint4 rec = P.records[r];
                const uint64_t pos = uint64_t(rec.x) | (uint64_t(rec.y) << 32);
                if (pos >= g0 && pos < g1 && !(VPT_EMIT_ABLATE & 6)) added += tag_suffix_bytes(TS, r, rec.z, rec.w);
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
        if (lane == 0) publish_run_size(F, blk, size);
        const uint64_t base = place_run(F, blk, size, lane);
        if (lane == 0) L.bcast[3] = base;
    }
    __syncthreads();
    const uint64_t base = L.bcast[3], end = base + size;
    const bool store_ok = end <= P.capacity;
    if (blk == F.n_blocks - 1 && tid == 0) {
        P.out_offsets[P.n_sent] = end;
        if (end > P.capacity) err |= kErrOutputTooSmall;
        if (F.total_out) *F.total_out = end;
        if (F.chain_out) *F.chain_out = end;
    }
    if (!sane) {
#pragma unroll
        for (uint32_t j = 0; j < kMine; ++j) if (mine[j]) P.out_offsets[i0 + tid + j * kEmitThreads] = base;
        if (err) atomicOr(P.status, err);
        return;
    }

    // ---- the pieces: every byte of the run to its place
    uint8_t* const sbytes = reinterpret_cast<uint8_t*>(L.stage);
    uint32_t my_rel[kMine];   // where the thread's sentences start, from tb (far away: none)
#pragma unroll
    for (uint32_t j = 0; j < kMine; ++j) my_rel[j] = mine[j] ? uint32_t(my_b[j] - B0) + lo_rel : 0xFFFFFFFFu;
    uint64_t at_out = base, cb = 0, sb = 0;   // output position, chars and sentence starts of the run in front of the piece
    bool fits = true;
    uint64_t rp = r_lo;                       // with tags: the first record not yet behind the pieces done (the same in every thread) ...
    uint64_t carry_rec = ~uint64_t(0);        // ... and the record, if any, of the last char in front of the piece
    for (uint32_t p_off = 0; p_off < span; p_off += kFlatPiece) {
        const uint32_t vm = in_range16_rel(p_off + 16u * tid, lo_rel, span);
        const uint4 x = vm ? *reinterpret_cast<const uint4*>(tb + p_off + 16u * tid) : make_uint4(0, 0, 0, 0);
        // the labels the piece's chars can ask for: label (O0 + cb - sb) onwards (every char but a sentence's first has one in front)
        const uintptr_t lab_at = l_all + O0 + (cb - sb), lab_al = lab_at & ~uintptr_t(15);
        const uint32_t lab_head = uint32_t(lab_at - lab_al);
        {
            // (lab_al + 16 > l_all: the window begins at a label of the run; how much of it there is, in 32 bits -- the workgroup's value)
            const uint32_t lim = l_end > lab_al ? (l_end - lab_al < 0xFFFFFFFFull ? uint32_t(l_end - lab_al) : 0xFFFFFFFFu) : 0u;
            const uint32_t a = 16u * tid, a2 = 16u * (uint32_t(kEmitThreads) + tid);
            reinterpret_cast<uint4*>(L.labs)[tid] = a < lim ? *reinterpret_cast<const uint4*>(lab_al + a) : make_uint4(0, 0, 0, 0);
            if (tid < 4) reinterpret_cast<uint4*>(L.labs)[kEmitThreads + tid] = a2 < lim ? *reinterpret_cast<const uint4*>(lab_al + a2) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (uint32_t j = 0; j < kMine; ++j) {
            if (mine[j] && my_rel[j] - p_off < kFlatPiece) {   // (unsigned: a start in front of the piece is far behind it)
                const uint32_t r = my_rel[j] - p_off;
                atomicOr(&L.starts[r >> 5], 1u << (r & 31u));
            }
        }
        __syncthreads();
        uint32_t sm = (L.starts[tid >> 1] >> (16 * (tid & 1))) & 0xFFFFu;
        const uint32_t lm = lead16(x) & vm, em = esc16(x) & vm;
        if (sm & ~lm) err |= kErrBadOffsets;   // a sentence that starts inside a char (or outside the run)
        sm &= lm;
        const uint32_t nl = uint32_t(__popc(lm)), nst = uint32_t(__popc(sm));
        uint32_t tot;
        const uint32_t excl = flat_block_scan<false>(nl | (nst << 16), L.wtot1, lane, wave, &tot);   // (its barrier: every thread has read its starts)
        if (tid < kFlatPiece / 32) L.starts[tid] = 0;
        const uint32_t c_in = excl & 0xFFFFu, s_in = excl >> 16;   // chars / starts of the piece in front of this thread
        // the thread's chars that have a label in front take consecutive labels from (c_in - s_in) of the window on: label q of the thread = bit q of lab
        const uint32_t nm = lm & ~sm;
        uint32_t spm = 0, lab, n_sp;
        {
            const uint32_t loff = lab_head + (c_in - s_in);           // byte offset in labs: <= 15 + 4096
            const uint32_t d = loff >> 2, r = loff & 3u;
            uint4 y;
            y.x = __builtin_amdgcn_alignbyte(L.labs[d + 1], L.labs[d], r); y.y = __builtin_amdgcn_alignbyte(L.labs[d + 2], L.labs[d + 1], r);
            y.z = __builtin_amdgcn_alignbyte(L.labs[d + 3], L.labs[d + 2], r); y.w = __builtin_amdgcn_alignbyte(L.labs[d + 4], L.labs[d + 3], r);
            lab = one16(y);
            if (kTags) {   // (the tags' owners ask for the spaces byte by byte)
                uint32_t bits = lab, rem = nm;
                while (rem) {   // label q of the thread onto its q-th labelled char
                    const uint32_t low = rem & (0u - rem);
                    if (bits & 1u) spm |= low;
                    bits >>= 1;
                    rem &= rem - 1u;
                }
                n_sp = uint32_t(__popc(spm));
            } else {
                // without tags nothing asks where the spaces are before the bytes go out, one after the other, each labelled char taking the next
                // label (below): their number is enough here -- no loop over the chars, which ran as long as the longest of the wave's lanes said
                n_sp = uint32_t(__popc(lab & ((1u << uint32_t(__popc(nm))) - 1u)));
            }
        }
        // spaces in front of the thread's byte k (below = the bits under k)
        const auto spaces_below = [&](uint32_t below) -> uint32_t {
            return kTags ? uint32_t(__popc(spm & below)) : uint32_t(__popc(lab & ((1u << uint32_t(__popc(nm & below))) - 1u)));
        };
        // Tag suffixes go in front of a space and in front of a sentence's first byte (the last token of the sentence before it), except
        // the run's first (the run before this one wrote that one behind its last byte).  The thread that holds the byte in FRONT of which
        // a suffix goes owns it; at most two per thread are carried in registers (tk: the byte, tl: the length, tc: the token's last
        // char), a third sends the thread's WAVE through its chars one by one (emit_fused_kernel's scheme).
        uint32_t tmask = 0, tk1 = 16, tl1 = 0, tk2 = 16, tl2 = 0, ts1 = 0, ts2 = 0;   // (ts: the record's place in LDS)
        uint64_t tr1 = 0, tr2 = 0;
        bool many = false;
        const uint32_t piece_chars = tot & 0xFFFFu;
        uint32_t n_rec = 0;                       // records of the piece (the same in every thread)
        uint16_t* const marks = kTags ? MK[0].m : nullptr;
        FlatMarks& SK = MK[0];
        // the record that ends a token at the piece's char k - 1 (k = 0: the char in front of the piece), ~0: none; *si: where its word and
        // strings wait in LDS (kStash: the carried record's), ~0: nowhere
        auto rec_at = [&](uint32_t k, uint32_t* si) -> uint64_t {
            const uint32_t m = marks[k];
            *si = m == 0 ? ~0u : m == kMarkCarry ? kStash : m - 1u < kStash ? m - 1u : ~0u;
            return m == 0 ? ~uint64_t(0) : m == kMarkCarry ? carry_rec : rp + (m - 1u);
        };
        if (kTags) {
            // mark the piece's records: sorted, so they are the next ones -- one coalesced read of their positions (a quarter of the workgroup
            // first: a piece of CJK text has some twenty), their words and first strings on the same trip into LDS
            const uint64_t p_lo = g0 + cb, p_hi = p_lo + piece_chars;
            uint32_t width = 64;
            for (; !(VPT_EMIT_ABLATE & 4);) {
                const uint64_t r = rp + n_rec + tid;
                bool in = false;
                if (tid < width && r < r_hi) {
                    const uint4 rec = P.records[r];
                    const uint32_t si = n_rec + tid;
                    uint2 s0 = make_uint2(0u, 0u), s1 = s0;
                    if (si < kStash) { s0 = P.rec_str[r * P.n_tags]; if (P.n_tags > 1) s1 = P.rec_str[r * P.n_tags + 1]; }
                    const uint64_t pos = uint64_t(rec.x) | (uint64_t(rec.y) << 32);
                    in = pos < p_hi;
                    if (in && pos >= p_lo) {
                        marks[uint32_t(pos - p_lo) + 1u] = uint16_t(si + 1u);
                        if (si < kStash) { SK.word[si] = rec.z; SK.last[si] = rec.w; SK.str[si][0] = s0; SK.str[si][1] = s1; }
                    }
                }
                uint32_t n_in;
                flat_block_scan(in ? 1u : 0u, L.wtot, lane, wave, &n_in);   // (its barriers: the marks are written)
                n_rec += n_in;
                if (n_in < width) break;
                width = uint32_t(kEmitThreads);
            }
            tmask = spm | sm;
            if (sb + s_in == 0 && sm) tmask &= ~(sm & (0u - sm));
            if (tmask && !(VPT_EMIT_ABLATE & 6)) {
                // the records of the chars in front of the thread's chars: marks[c_in + j] for its j-th char; which of them are there at all
                uint32_t pm = 0;
                for (uint32_t j = 0; j < nl; ++j) pm |= (marks[c_in + j] != 0 ? 1u : 0u) << j;
                while (pm) {   // few
                    const uint32_t j = uint32_t(__ffs(int(pm))) - 1u;
                    pm &= pm - 1u;
                    uint32_t remj = lm;
                    for (uint32_t q = 0; q < j; ++q) remj &= remj - 1u;
                    const uint32_t k = uint32_t(__ffs(int(remj))) - 1u;          // the byte of the thread's j-th char
                    if (!((tmask >> k) & 1u)) continue;                          // no token ends in front of it
                    uint32_t si;
                    const uint64_t ri = rec_at(c_in + j, &si);
                    if (ri == ~uint64_t(0)) continue;
                    uint32_t word, last;
                    if (si != ~0u) { word = SK.word[si]; last = SK.last[si]; }
                    else { const uint4 rec = P.records[ri]; word = rec.z; last = rec.w; }
                    const uint32_t len = tag_suffix_bytes(TS, ri, word, last);   // (carried from fill_tags)
                    if (!len) continue;
                    if (tk1 == 16) { tk1 = k; tl1 = len; tr1 = ri; ts1 = si; }
                    else if (tk2 == 16) { tk2 = k; tl2 = len; tr2 = ri; ts2 = si; }
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
                uint32_t si;
                const uint64_t ri = rec_at(c_in + uint32_t(__popc(lm & (low - 1u))), &si);
                if (ri != ~uint64_t(0)) sfx_total += tag_suffix(TS, ri, nullptr);
            }
        }
        const uint32_t t = uint32_t(__popc(vm)) + n_sp + uint32_t(__popc(em)) + sfx_total;
        uint32_t total;
        const uint32_t w = flat_block_scan<false>(t, L.wtot2, lane, wave, &total);
        if (at_out + total > end) { fits = false; break; }   // (the same in every thread)
        uint8_t* const dst = P.out_text + at_out;
        const uint32_t head = uint32_t(reinterpret_cast<uintptr_t>(dst) & 15u);
        const bool staged = !kTags || head + total <= kFlatStageBytes;   // (the same in every thread) else: byte stores straight to the output
        if (!slow) {
            // (wave-uniform) every lane's sixteen bytes are the run's and the piece is assembled in LDS: the bytes go out unconditionally
            const bool whole = (!kTags || staged) && __ballot(vm != 0xFFFFu) == 0;
            uint8_t* const o = !kTags || staged ? sbytes + head : dst + 0;   // (with tags: a generic pointer)
            if (store_ok && whole) {
                // The thread's bytes in order, [tags] [' '] ['\\'] byte, with no selects: a ' ' (and a '\\') is written where it WOULD stand and the
                // position moves on only if it does -- what follows overwrites it otherwise (the LDS takes a wave's stores in the order they were
                // issued, and the last store of a thread is a byte of its text; the tags' bytes, written below, lie where none of these stores
                // goes).  Four issue slots a byte where the selects of the general loop below took fifteen: the writer runs at the vector ALU's
                // issue rate (profiles/r06_r_*).  The variants are the wave's: no '\\' stores without an escaped byte, no tag lengths without a tag.
                // (Measured and not kept, profiles/r06_v_*: a dword's bytes and spaces picked by two v_perm_b32 with selectors from a table in LDS and stored as
                // one unaligned ds_write_b64 -- 10 % fewer vector instructions, 7 % slower: the LDS splits the unaligned stores.)
                uint8_t* const ob = sbytes + head + w;
                const auto bytes_out = [&](auto esc_c, auto tag_c) {
                    constexpr bool kEsc = decltype(esc_c)::value, kTag = decltype(tag_c)::value;
                    uint32_t ins = 0, bits = lab, mk = kTags ? spm : nm, ek = em;
#pragma unroll
                    for (uint32_t k = 0; k < 16; ++k) {
                        // (a dword at a time, its masks taken from copies nothing else reads: computed for all sixteen bytes ahead of the branch, as the
                        // compiler would, the bits and the bytes hold thirty registers and the kernel runs five waves per SIMD where it had eight)
                        if ((k & 3u) == 0) VPT_OPAQUE3(mk, ek, ins);
                        uint32_t sp;
                        if (kTags) sp = (mk >> k) & 1u;
                        else { const uint32_t nmk = (mk >> k) & 1u; sp = nmk & bits; bits >>= nmk; }   // (a labelled char takes the next label)
                        if (kTag) ins += (k == tk1 ? tl1 : 0u) + (k == tk2 ? tl2 : 0u);
                        ob[ins + k] = 0x20u; ins += sp;
                        if (kEsc) { ob[ins + k] = 0x5Cu; ins += (ek >> k) & 1u; }
                        ob[ins + k] = uint8_t(byte_of(x, k));
                    }
                };
                const bool any_esc = __ballot(em != 0) != 0, any_tag = kTags && __ballot(tl1 != 0) != 0;
                if (!any_esc && !any_tag) bytes_out(std::false_type{}, std::false_type{});
                else if (!any_tag) bytes_out(std::true_type{}, std::false_type{});
                else if (!any_esc) bytes_out(std::false_type{}, std::true_type{});
                else bytes_out(std::true_type{}, std::true_type{});
            } else if (store_ok && __ballot(vm != 0) != 0) {   // the same with a select per store: what is not there goes to a slot of the thread's own
                uint8_t* const dump = reinterpret_cast<uint8_t*>(L.dump + wave);
                uint32_t pos = w, bits = lab;
#pragma unroll
                for (uint32_t k = 0; k < 16; ++k) {
                    const uint32_t nmk = (nm >> k) & 1u;
                    const uint32_t v = (vm >> k) & 1u, sp = kTags ? (spm >> k) & 1u : nmk & bits, es = (em >> k) & 1u;
                    bits >>= nmk;
                    if (kTags) pos += (k == tk1 ? tl1 : 0u) + (k == tk2 ? tl2 : 0u);
                    *(sp ? o + pos : dump) = 0x20u; pos += sp;
                    *(es ? o + pos : dump) = 0x5Cu; pos += es;
                    *(v ? o + pos : dump) = uint8_t(byte_of(x, k)); pos += v;
                }
            }
            if (kTags && store_ok && tl1 && !(VPT_EMIT_ABLATE & 1)) {   // the tags themselves (few threads)
                const uint32_t b1 = (1u << tk1) - 1u, b2 = (1u << tk2) - 1u;
                // (their strings' places wait in LDS as a rule: the bytes are one trip away)
                const auto put = [&](uint64_t ri, uint32_t si, uint8_t* at, uint32_t len) -> bool {
                    return si != ~0u ? tag_suffix_write(TS, ri, SK.last[si], SK.str[si], at, len) == len : tag_suffix(TS, ri, at, len) == len;
                };
                if (!put(tr1, ts1, o + w + uint32_t(__popc(vm & b1)) + uint32_t(__popc(spm & b1)) + uint32_t(__popc(em & b1)), tl1)) err |= kErrBadOffsets;
                if (tl2 && !put(tr2, ts2, o + w + tl1 + uint32_t(__popc(vm & b2)) + uint32_t(__popc(spm & b2)) + uint32_t(__popc(em & b2)), tl2)) err |= kErrBadOffsets;
            }
            uint32_t rem = sm;   // the sentences that start in the thread's bytes (few threads, one as a rule)
            while (rem) {
                const uint32_t k = uint32_t(__ffs(int(rem))) - 1u, below = (1u << k) - 1u;
                rem &= rem - 1u;
                const uint64_t s = sb + s_in + uint32_t(__popc(sm & below));
                if (s < ns) {
                    P.out_offsets[i0 + s] = at_out + w + uint32_t(__popc(vm & below)) + spaces_below(below) + uint32_t(__popc(em & below)) +
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
                if ((tmask >> k) & 1u) { uint32_t si; const uint64_t ri = rec_at(c_in + ci, &si); if (ri != ~uint64_t(0)) pos += tag_suffix(TS, ri, o ? o + pos : nullptr); }
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
        uint64_t next_carry = carry_rec;   // the record of the piece's last char goes on to the next piece (every thread reads the same mark)
        uint32_t next_si = kStash;
        if (kTags && piece_chars) next_carry = rec_at(piece_chars, &next_si);
        __syncthreads();
        if (kTags) {   // the marks are done with: clear them for the next piece, whose marks[0] is this piece's last char
            reinterpret_cast<uint4*>(marks)[tid] = make_uint4(0, 0, 0, 0);
            reinterpret_cast<uint4*>(marks)[kEmitThreads + tid] = make_uint4(0, 0, 0, 0);
            if (tid < 2) reinterpret_cast<uint4*>(marks)[2 * kEmitThreads + tid] = make_uint4(0, 0, 0, 0);
            carry_rec = next_carry;
            rp += n_rec;
            if (tid == 0 && carry_rec != ~uint64_t(0)) {   // ... with its word and strings in the carried record's place
                marks[0] = uint16_t(kMarkCarry);
                if (next_si != kStash) {
                    if (next_si != ~0u) { SK.word[kStash] = SK.word[next_si]; SK.last[kStash] = SK.last[next_si]; SK.str[kStash][0] = SK.str[next_si][0]; SK.str[kStash][1] = SK.str[next_si][1]; }
                    else {
                        const uint4 rec = P.records[carry_rec];
                        SK.word[kStash] = rec.z; SK.last[kStash] = rec.w;
                        SK.str[kStash][0] = P.rec_str[carry_rec * P.n_tags]; SK.str[kStash][1] = P.n_tags > 1 ? P.rec_str[carry_rec * P.n_tags + 1] : make_uint2(0u, 0u);
                    }
                }
            }
        }
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
    if (kTags && fits && carry_rec != ~uint64_t(0) && !(VPT_EMIT_ABLATE & 6)) {   // the tags of the run's last token: the record, if any, of its last char
        const uint4 rec = P.records[carry_rec];
        const uint32_t sl = (uint64_t(rec.x) | (uint64_t(rec.y) << 32)) == g1 - 1 ? tag_suffix(TS, carry_rec, nullptr) : 0u;   // (every thread computes the same)
        if (sl && store_ok && at_out + sl <= end && tid == 0) tag_suffix(TS, carry_rec, P.out_text + at_out);
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
            const uint32_t lm = lead_mask16(v[q]) & vm;
            const uint32_t zm = zero_mask16(v[q]) & vm;
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

hipError_t launch_emit_tokenized(const EmitParams& P, const EmitFuse& F, hipStream_t stream) {   // a workgroup per run of sentences
    if (P.records) hipLaunchKernelGGL(emit_flat_kernel<true>, dim3(uint32_t(F.n_blocks)), dim3(kEmitThreads), 0, stream, P, F);
    else hipLaunchKernelGGL(emit_flat_kernel<false>, dim3(uint32_t(F.n_blocks)), dim3(kEmitThreads), 0, stream, P, F);
    return hipGetLastError();
}

}  // namespace vpt
