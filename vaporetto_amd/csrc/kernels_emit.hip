// Token emission on the device: Sentence::write_tokenized_text for a batch (sentence.rs:850-886), boundary part:
// the tokens of a sentence are the runs between WordBoundary labels (sentence.rs:1270-1300), written in order with
// one ' ' between them and a '\' in front of every ' ', '\' and '/' byte of a surface; with tags (the indices
// vpt_fill_tags_batch wrote and the tag model it found for every token) each token is followed by "/tag" for its tag
// slots up to the last Some, an empty string for a None in between (sentence.rs:866-881).  (Unknown boundaries only
// come from partially annotated corpora, never from predict: they are rejected here, kErrUnknownLabel.)
//
// Output size is data dependent, so:
//   emit_count_kernel   one wave per sentence: bytes this sentence will take -> offsets[i + 1].  The count needs no
//                       byte <-> char correspondence: text bytes + escaped bytes (a pass over the text, four bytes per lane)
//                       + WordBoundary labels (a pass over the labels, four per lane) + the tag suffixes of the tokens,
//                       which hang on label positions
//   scan_chained_kernel inclusive prefix sum over the sentences, in place (offsets[0] = 0): one launch, every workgroup looks back
//                       over its predecessors' published sums (three launches before: profiles/r02_j_emit_kernels.txt)
//   emit_write_kernel   one wave per sentence, 256 text bytes per step (a dword per lane): lead and escape bits from the
//                       dword, the labels of the lane's chars in one unaligned load, a DPP prefix sum places every lane's
//                       output, which is assembled in LDS and leaves as aligned dword stores
// A sentence's bytes are independent of the other sentences', its position is not: that is the scan.
#include <hip/hip_runtime.h>

#include "device_common.h"
#include "kernels.hpp"

namespace vpt {
namespace {

constexpr int kEmitThreads = 256;
constexpr int kEmitWaves = kEmitThreads / 64;
constexpr int kScanThreads = 256, kScanPer = 16;
constexpr uint64_t kScanBlock = uint64_t(kScanThreads) * kScanPer;   // offsets one workgroup of the scan takes
constexpr uint32_t kStageBytes = 1024;   // a wave's output of one step in LDS: <= 3 + 3 * 256 bytes without tag suffixes

// 0x80 in every byte of v that is zero (exact: no carries between the bytes)
__device__ __forceinline__ uint32_t zero_bytes(uint32_t v) { return ~(((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v | 0x7F7F7F7Fu); }
// bits 7, 15, 23, 31 -> bits 0..3
__device__ __forceinline__ uint32_t byte_flags_to_nibble(uint32_t m) { return (((m >> 7) * 0x00204081u) >> 21) & 0xFu; }
// 4-bit mask of the bytes of x that write_tokenized_text escapes: ' ', '\', '/'
__device__ __forceinline__ uint32_t esc_nibble(uint32_t x) {
    return byte_flags_to_nibble(zero_bytes(x ^ 0x20202020u) | zero_bytes(x ^ 0x5C5C5C5Cu) | zero_bytes(x ^ 0x2F2F2F2Fu));
}
// "/tag/tag.." of the token whose last char is char `c` (batch-flat index) and whose tag model (index + 1, from the
// fill_tags call) is `model`: bytes it takes; written to `dst` when given
__device__ __forceinline__ uint32_t tag_suffix_of(const EmitParams& P, uint64_t c, int32_t model, uint8_t* dst) {
    if (model <= 0 || uint32_t(model) > P.n_models) return 0;   // no tag model for this surface (or not our array)
    const uint32_t* mr = P.models + size_t(model - 1) * 12;
    const int32_t* tg = P.tags + c * P.n_tags;
    const uint32_t slot0 = mr[8], n_slots = mr[9] < P.n_tags ? mr[9] : P.n_tags;
    uint32_t last = 0;   // slots to write: up to the last Some (the tags are fetched with the model record, not after it)
    for (uint32_t j = 0; j < P.n_tags; ++j)
        if (tg[j] >= 0 && j < n_slots) last = j + 1;
    uint32_t n = 0;
    for (uint32_t j = 0; j < last; ++j) {
        if (dst) dst[n] = 0x2Fu;
        ++n;
        if (tg[j] < 0) continue;
        const uint32_t k = P.slot_str[slot0 + j] + uint32_t(tg[j]);
        if (k >= P.n_strings) continue;                            // an index fill_tags cannot have written
        const uint32_t a = P.str_off[k], b = P.str_off[k + 1];
        if (dst) for (uint32_t q = a; q < b; ++q) dst[n + (q - a)] = P.str_bytes[q];
        n += b - a;
    }
    return n;
}
__device__ __forceinline__ uint32_t tag_suffix(const EmitParams& P, uint64_t c, uint8_t* dst) {
    return P.tags ? tag_suffix_of(P, c, P.tok_model[c], dst) : 0u;
}
// the tag models of the four chars from `c` on in one load (the array is padded past the batch's chars: capi.cpp)
struct Models4 { int32_t m[4]; };
__device__ __forceinline__ Models4 load_models4(const EmitParams& P, uint64_t c) {
    Models4 r;
    __builtin_memcpy(&r, P.tok_model + c, sizeof(r));
    return r;
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

// the four offsets of sentence i (the same in every lane), loaded one sentence ahead of their use: a wave's sentences are n_waves
// apart, so every one of them starts with a trip to memory that nothing else of the sentence can overlap with
struct SentOff { uint64_t b0, b1, o0, o1; };
__device__ __forceinline__ SentOff load_sent_off(const EmitParams& P, uint64_t i) {
    SentOff r{0, 0, 0, 0};
    if (i < P.n_sent) { r.b0 = P.boff[i]; r.b1 = P.boff[i + 1]; r.o0 = P.ooff[i]; r.o1 = P.ooff[i + 1]; }
    return r;
}

__global__ __launch_bounds__(kEmitThreads) void emit_count_kernel(const EmitParams P, uint64_t* scan_state) {
    clear_scan_state(scan_state, P.n_sent);
    const int lane = threadIdx.x & 63;
    const uint64_t wave = uint64_t(blockIdx.x) * kEmitWaves + wave_uniform(threadIdx.x >> 6);
    const uint64_t n_waves = uint64_t(gridDim.x) * kEmitWaves;
    uint32_t err = 0;
    SentOff nxt = load_sent_off(P, wave);
    for (uint64_t i = wave; i < P.n_sent; i += n_waves) {
        const uint64_t b0 = wave_uniform64(nxt.b0), b1 = wave_uniform64(nxt.b1), o0 = wave_uniform64(nxt.o0), o1 = wave_uniform64(nxt.o1);
        nxt = load_sent_off(P, i + n_waves);
        const bool sane = b1 > b0 && o1 >= o0 && o1 <= P.total_boundaries;
        const uint64_t n_labels = sane ? o1 - o0 : 0;
        uint64_t mine = 0, leads = 0;   // this lane's share of the added bytes / of the chars
        if (sane) {
            const uint8_t* lab = P.labels + o0;
            // the first 256 text bytes and the first 256 labels -- all there is of an ordinary sentence -- are asked for together
            const uint32_t x0 = load4(P.text, b0 + 4 * uint64_t(lane), b1);
            uint32_t y0 = load4(lab, 4 * uint64_t(lane), n_labels);
            for (uint64_t pos = b0; pos < b1; pos += 256) {
                const uint64_t at = pos + 4 * uint64_t(lane);
                const uint32_t x = pos == b0 ? x0 : load4(P.text, at, b1);
                const uint32_t nv = at < b1 ? uint32_t(b1 - at < 4 ? b1 - at : 4) : 0u, vm = (1u << nv) - 1u;
                leads += uint32_t(__popc(lead_nibble(x) & vm));
                mine += uint32_t(__popc(esc_nibble(x) & vm));
            }
            for (uint64_t k0 = 0; k0 < n_labels; k0 += 256) {
                const uint64_t k = k0 + 4 * uint64_t(lane);
                const uint32_t y = k0 == 0 ? y0 : load4(lab, k, n_labels);
                if (y & 0xFEFEFEFEu) err |= kErrUnknownLabel;
                uint32_t om = byte_flags_to_nibble(zero_bytes(y ^ 0x01010101u));   // (a byte past n_labels reads 0: not a boundary)
                mine += uint32_t(__popc(om));
                if (P.tags && om) {   // a token's tag suffix hangs on the label that ends it: char k + q is its last char
                    const Models4 tm = load_models4(P, o0 + i + k);
                    uint32_t todo = om & ((tm.m[0] > 0 ? 1u : 0u) | (tm.m[1] > 0 ? 2u : 0u) | (tm.m[2] > 0 ? 4u : 0u) | (tm.m[3] > 0 ? 8u : 0u));
                    while (todo) {   // the few tokens with a tag model (one call site: the routine is long)
                        const uint32_t q = uint32_t(__ffs(int(todo))) - 1u;
                        todo &= todo - 1u;
                        mine += tag_suffix_of(P, o0 + i + k + q, q == 0 ? tm.m[0] : q == 1 ? tm.m[1] : q == 2 ? tm.m[2] : tm.m[3], nullptr);
                    }
                }
            }
        }
        // one reduction for both sums: chars in the high half (a sentence of 2^31 chars is not in this kernel's index width anyway)
        uint64_t chars, added;
        if (b1 - b0 < (uint64_t(1) << 30)) {   // one reduction for both sums (wave-uniform; neither can reach 2^32 then)
            const uint64_t both = wave_sum64(mine | (leads << 32));
            chars = both >> 32; added = both & 0xFFFFFFFFull;
        } else { chars = wave_sum64(leads); added = wave_sum64(mine); }
        uint64_t bytes_out = (b1 - b0) + added;
        if (!sane) { err |= b1 > b0 ? kErrBadOffsets : kErrEmptySentence; bytes_out = 0; }
        else if (chars != n_labels + 1) err |= kErrBadOffsets;
        else bytes_out += tag_suffix(P, o1 + i, nullptr);   // the last token's (every lane computes the same)
        if (lane == 0) P.out_offsets[i + 1] = bytes_out;
    }
    if (err) atomicOr(P.status, err);
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
struct EmitLds {
    uint32_t stage[kEmitWaves][kStageBytes / 4];
    uint32_t labs[kEmitWaves][68];   // the labels a step's chars can ask for: 256 + 3 bytes, fetched with the step's text
};

__global__ __launch_bounds__(kEmitThreads) void emit_write_kernel(const EmitParams P) {
    __shared__ EmitLds LDS;
    const int lane = threadIdx.x & 63;
    const uint32_t wid = wave_uniform(threadIdx.x >> 6);
    uint32_t* const stage = LDS.stage[wid];
    uint8_t* const sb = reinterpret_cast<uint8_t*>(stage);
    uint32_t* const labs = LDS.labs[wid];
    const uint64_t wave = uint64_t(blockIdx.x) * kEmitWaves + wid;
    const uint64_t n_waves = uint64_t(gridDim.x) * kEmitWaves;
    // a sentence's offsets are loaded one sentence ahead (see emit_count_kernel)
    SentOff nxt = load_sent_off(P, wave);
    uint64_t nxt_a = wave < P.n_sent ? P.out_offsets[wave] : 0, nxt_e = wave < P.n_sent ? P.out_offsets[wave + 1] : 0;
    for (uint64_t i = wave; i < P.n_sent; i += n_waves) {
        const uint64_t b0 = wave_uniform64(nxt.b0), b1 = wave_uniform64(nxt.b1), o0 = wave_uniform64(nxt.o0), o1 = wave_uniform64(nxt.o1);
        const uint64_t end = wave_uniform64(nxt_e);
        uint64_t at_out = wave_uniform64(nxt_a), chars = 0;
        nxt = load_sent_off(P, i + n_waves);
        if (i + n_waves < P.n_sent) { nxt_a = P.out_offsets[i + n_waves]; nxt_e = P.out_offsets[i + n_waves + 1]; }
        if (!(b1 > b0 && o1 >= o0 && o1 <= P.total_boundaries)) continue;   // reported by emit_count_kernel
        const uint64_t n_labels = o1 - o0;
        const uint8_t* lab = P.labels + o0;
        if (end > P.capacity || end < at_out) continue;                    // kErrOutputTooSmall
        bool fits = true;                                                  // `end` only binds when the inputs changed under us
        for (uint64_t pos = b0; pos < b1 && fits; pos += 256) {
            const uint64_t at = pos + 4 * uint64_t(lane);
            // the step's text and the labels its chars can ask for (label[chars - 1] onwards: at most 256 chars start in 256 bytes) leave
            // in ONE trip to memory; a lane then finds the labels of its own chars in LDS (which ones it learns from the text)
            const uint64_t lbase = chars ? chars - 1 : 0;
            const uint32_t x = load4(P.text, at, b1);
            labs[lane] = load4(lab, lbase + 4 * uint64_t(lane), n_labels);
            if (lane < 4) labs[64 + lane] = lane == 0 ? load4(lab, lbase + 256, n_labels) : 0u;
            const uint32_t nv = at < b1 ? uint32_t(b1 - at < 4 ? b1 - at : 4) : 0u, vm = (1u << nv) - 1u;
            const uint32_t lm = lead_nibble(x) & vm, em = esc_nibble(x) & vm;
            const uint32_t nl = uint32_t(__popc(lm));
            const uint32_t incl_l = wave_inclusive_scan(nl);
            const uint64_t ci0 = chars + (incl_l - nl);                    // index in the sentence of this lane's first char
            __builtin_amdgcn_wave_barrier();
            // the labels in front of this lane's chars: label[ci0 - 1 + q] for its q-th char (none in front of char 0)
            const uint32_t loff = uint32_t((ci0 ? ci0 - 1 : 0) - lbase);   // <= 256
            uint32_t y = nl ? __builtin_amdgcn_alignbyte(labs[(loff >> 2) + 1], labs[loff >> 2], loff & 3u) : 0u;
            if (ci0 == 0) y <<= 8;
            // which of the lane's chars have a space in front (bit q: its q-th char), moved onto the chars' lead bytes (bit k: byte k)
            const uint32_t spq = byte_flags_to_nibble(zero_bytes(y ^ 0x01010101u)) & ((1u << nl) - 1u);
            uint32_t spm = 0;
            {
                uint32_t rem = lm;
#pragma unroll
                for (uint32_t q = 0; q < 4; ++q) {
                    const uint32_t low = rem & (0u - rem);
                    if ((spq >> q) & 1u) spm |= low;
                    rem &= rem - 1u;
                }
            }
            // the token in front of a space ends at the char before it and its tags go in front of the space: the tag models of
            // chars ci0 - 1 .. ci0 + 2 in one load, the suffix lengths of the few that have one (one call site: the routine is long)
            uint32_t sfx[4] = {0, 0, 0, 0};
            Models4 tm = {{0, 0, 0, 0}};
            if (P.tags && spq) {
                tm = load_models4(P, o0 + i + (ci0 ? ci0 - 1 : 0));
                if (ci0 == 0) { tm.m[3] = tm.m[2]; tm.m[2] = tm.m[1]; tm.m[1] = tm.m[0]; tm.m[0] = 0; }
                uint32_t todo = spq & ((tm.m[0] > 0 ? 1u : 0u) | (tm.m[1] > 0 ? 2u : 0u) | (tm.m[2] > 0 ? 4u : 0u) | (tm.m[3] > 0 ? 8u : 0u));
                while (todo) {
                    const uint32_t q = uint32_t(__ffs(int(todo))) - 1u;
                    todo &= todo - 1u;
                    const uint32_t len = tag_suffix_of(P, o0 + i + ci0 + q - 1, q == 0 ? tm.m[0] : q == 1 ? tm.m[1] : q == 2 ? tm.m[2] : tm.m[3], nullptr);
                    // char q's lead byte: the q-th set bit of lm
                    uint32_t rem = lm;
                    for (uint32_t r = 0; r < q; ++r) rem &= rem - 1u;
                    const uint32_t k = uint32_t(__ffs(int(rem))) - 1u;
                    sfx[0] += k == 0 ? len : 0u; sfx[1] += k == 1 ? len : 0u; sfx[2] += k == 2 ? len : 0u; sfx[3] += k == 3 ? len : 0u;
                }
            }
            const uint32_t t = nv + uint32_t(__popc(spm)) + uint32_t(__popc(em)) + sfx[0] + sfx[1] + sfx[2] + sfx[3];
            const uint32_t incl_t = wave_inclusive_scan(t);
            const uint32_t total = uint32_t(__builtin_amdgcn_readlane(int(incl_t), 63));
            const uint32_t w = incl_t - t;                                 // where this lane's output starts in the step's
            if (at_out + total > end) { fits = false; break; }
            uint8_t* const dst = P.out_text + at_out;
            const uint32_t head = uint32_t(reinterpret_cast<uintptr_t>(dst) & 3u);
            const bool staged = head + total <= kStageBytes;               // (wave-uniform) else: byte stores straight to the output
            auto put = [&](uint8_t* const o) {   // byte k of the lane goes to w + k + what is inserted up to it: [tags] [' '] ['\\'] byte
                uint32_t ins = w;
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) {
                    const uint32_t sp = (spm >> k) & 1u, es = (em >> k) & 1u;
                    ins += sfx[k] + sp + es;
                    if (k < nv) {
                        o[ins + k] = uint8_t(x >> (8 * k));
                        if (es) o[ins + k - 1u] = 0x5Cu;
                        if (sp) o[ins + k - 1u - es] = 0x20u;
                    }
                }
                uint32_t todo = (sfx[0] ? 1u : 0u) | (sfx[1] ? 2u : 0u) | (sfx[2] ? 4u : 0u) | (sfx[3] ? 8u : 0u);
                while (todo) {   // rare: the tags themselves, in front of the space of lead byte k
                    const uint32_t k = uint32_t(__ffs(int(todo))) - 1u;
                    todo &= todo - 1u;
                    uint32_t at_k = w + k, q = 0;   // bytes of the lane in front of byte k's own insertions; k is the lane's q-th char
                    for (uint32_t j = 0; j < k; ++j) { at_k += sfx[j] + ((spm >> j) & 1u) + ((em >> j) & 1u); q += (lm >> j) & 1u; }
                    (void)tag_suffix_of(P, o0 + i + ci0 + q - 1, q == 0 ? tm.m[0] : q == 1 ? tm.m[1] : q == 2 ? tm.m[2] : tm.m[3], o + at_k);
                }
            };
            if (staged) put(sb + head);   // (two calls: one writes LDS, one global memory -- not one through a generic pointer)
            else put(dst);
            if (staged) {   // LDS byte j is output byte j - head: whole dwords leave aligned, the two edges byte by byte
                __builtin_amdgcn_wave_barrier();
                uint8_t* const abase = dst - head;
                const uint32_t nd = (head + total + 3u) >> 2;
                for (uint32_t d = uint32_t(lane); d < nd; d += 64) {
                    const uint32_t lo = d * 4u, hi = lo + 4u;
                    if (lo >= head && hi <= head + total) {
                        *reinterpret_cast<uint32_t*>(abase + lo) = stage[d];
                    } else {
                        const uint32_t a = lo > head ? lo : head, b = hi < head + total ? hi : head + total;
                        for (uint32_t j = a; j < b; ++j) abase[j] = sb[j];
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            at_out += total;
            chars += uint32_t(__builtin_amdgcn_readlane(int(incl_l), 63));
        }
        if (fits && lane == 0 && chars == n_labels + 1) {                  // the last token's tags
            const uint32_t s = tag_suffix(P, o1 + i, nullptr);
            if (s && at_out + s <= end) tag_suffix(P, o1 + i, P.out_text + at_out);
        }
    }
}

// vpt_count_boundaries on the device: chars - 1 of every sentence -> offsets[i + 1] (the scan follows), the same
// validation as Sentence::from_raw (sentence.rs:160-196), the longest sentence (in chars) -> *max_chars
__global__ __launch_bounds__(kEmitThreads) void count_chars_kernel(const uint8_t* __restrict__ text, const uint64_t* __restrict__ boff,
                                                                   uint64_t n_sent, uint64_t* __restrict__ offsets, uint32_t* __restrict__ status,
                                                                   uint32_t* __restrict__ max_chars, uint64_t* scan_state) {
    clear_scan_state(scan_state, n_sent);
    const int lane = threadIdx.x & 63;
    const uint64_t wave = uint64_t(blockIdx.x) * kEmitWaves + wave_uniform(threadIdx.x >> 6);
    const uint64_t n_waves = uint64_t(gridDim.x) * kEmitWaves;
    uint32_t err = 0, longest = 0;
    for (uint64_t i = wave; i < n_sent; i += n_waves) {
        const uint64_t b0 = wave_uniform64(boff[i]), b1 = wave_uniform64(boff[i + 1]);
        uint64_t mine = 0;
        bool nul = false;
        for (uint64_t pos = b0; pos < b1; pos += 256) {
            const uint64_t at = pos + 4 * uint64_t(lane);
            const uint32_t x = load4(text, at, b1);
            const uint32_t nv = at < b1 ? uint32_t(b1 - at < 4 ? b1 - at : 4) : 0u, vm = (1u << nv) - 1u;
            nul = nul || (byte_flags_to_nibble(zero_bytes(x)) & vm) != 0;
            mine += uint32_t(__popc(lead_nibble(x) & vm));
        }
        const uint64_t chars = wave_sum64(mine);
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

size_t scan_part_entries(uint64_t n) { return size_t((n + kScanBlock - 1) / kScanBlock) + 2; }   // the blocks' words + the ticket

// one wave per sentence, but no more workgroups than `max_blocks` (a few generations of what the device runs at a time; 0: 65536):
// the waves then stride over the batch and the sentences' lengths even out (see launch_tag_tokens)
static uint32_t emit_blocks(uint64_t n_sent, uint32_t max_blocks) {
    const uint64_t want = (n_sent + kEmitWaves - 1) / kEmitWaves, cap = max_blocks ? max_blocks : 65536;
    return uint32_t(want < 1 ? 1 : want > cap ? cap : want);
}

hipError_t launch_count_boundaries(const uint8_t* text, const uint64_t* boff, uint64_t n_sent, uint64_t* ooff_out, uint64_t* scan_part, uint32_t* status,
                                   uint32_t* max_chars, uint32_t max_blocks, hipStream_t stream) {
    const uint32_t blocks = emit_blocks(n_sent, max_blocks);
    hipLaunchKernelGGL(count_chars_kernel, dim3(blocks), dim3(kEmitThreads), 0, stream, text, boff, n_sent, ooff_out, status, max_chars, scan_part);
    return launch_scan(ooff_out, n_sent, scan_part, ~uint64_t(0), status, nullptr, stream);
}

hipError_t launch_emit_tokenized(const EmitParams& P, uint64_t* scan_part, uint32_t max_blocks, uint64_t* total_out, hipStream_t stream) {
    const uint32_t blocks = emit_blocks(P.n_sent, max_blocks);
    hipLaunchKernelGGL(emit_count_kernel, dim3(blocks), dim3(kEmitThreads), 0, stream, P, scan_part);
    const hipError_t e = launch_scan(P.out_offsets, P.n_sent, scan_part, P.capacity, P.status, total_out, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(emit_write_kernel, dim3(blocks), dim3(kEmitThreads), 0, stream, P);
    return hipGetLastError();
}

}  // namespace vpt
