// HIP kernels for gfx950 (MI355X): batched Predictor::predict (predictor.rs:518-543).
//
// One workgroup = one TILE of consecutive whole sentences (sentences never interact: every n-gram, dictionary
// word and type window lives inside one sentence).  Per tile, entirely in LDS:
//   A. UTF-8 -> scalar values + character types (Sentence::parse_raw, sentence.rs:160-196; get_type :50-67),
//      laid out FLAT with `pad` zero symbols between sentences so that no pattern can match across a sentence
//      break and the type window sees the reference's "outside the sentence" code 0.
//   B. every flat position = a pattern START: probe the pattern table for the 1-, 2-, 3-symbol strings starting
//      there (and walk the trie for longer dictionary words), add each hit's weight row into the LDS score
//      array with ds_add (integer => order-free => bit-exact).
//   C. every flat position = a BOUNDARY: bias + LDS score + type-window table lookup, write i32 score and the
//      sign label, coalesced.
// HBM traffic per boundary: the text once (~3 B), score + label once (5 B); everything else is table gathers
// (L2 / Infinity Cache / HBM), which is what bounds the kernel.
#include <hip/hip_runtime.h>

#include "device_common.h"
#include "kernels.hpp"

namespace vpt {

namespace {

constexpr int kWaves = kThreads / 64;

// ------------------------------------------------------------------------------------------------------------
// device error word bits (read back by vpt_batch_sync)
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void raise(uint32_t* status, uint32_t bit) { atomicOr(status, bit); }

// tile memory: LDS on the fast path, a global scratch slab for sentences that do not fit in LDS
template <bool kLds>
struct TileMem {
    uint32_t* sym;     // scalar value per flat position, 0 = separator
    int32_t* score;    // accumulated pattern weights per boundary (boundary p lies between flat p and p+1)
    uint8_t* typ;      // character type per flat position, 0 = separator
    uint16_t* sidx;    // tile-local sentence index per flat position (LDS path only)
};

template <bool kLds>
__device__ __forceinline__ void score_add(int32_t* p, int32_t v) {
    if (kLds) atomicAdd(p, v);                       // ds_add_u32, no return
    else __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool kLds>
__device__ __forceinline__ int32_t score_get(const int32_t* p) {
    if (kLds) return *p;
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // L2, where the atomics landed
}

// ------------------------------------------------------------------------------------------------------------
// B. pattern scoring for one start position
// ------------------------------------------------------------------------------------------------------------
template <int kChunks>
__device__ __forceinline__ bool probe_short(const PatternTableView& T, uint64_t key, uint32_t (&e)[4 * kChunks]) {
    uint32_t b = hash_slot(key, T.short_shift);
    const uint32_t klo = uint32_t(key), khi = uint32_t(key >> 32);
    bool home = true;
    for (;;) {  // buckets of two entries (layout.h)
        bool free_slot = false, displaced = false;
#pragma unroll
        for (uint32_t j = 0; j < kShortBucket; ++j) {
            const uint4* p = reinterpret_cast<const uint4*>(T.short_tab + (size_t(b) * kShortBucket + j) * T.stride_dw);
#pragma unroll
            for (int q = 0; q < kChunks; ++q) {
                uint4 v = p[q];
                e[4 * q] = v.x; e[4 * q + 1] = v.y; e[4 * q + 2] = v.z; e[4 * q + 3] = v.w;
            }
            if (e[0] == klo && (e[1] & ~kDisplacedBit) == khi) return true;
            if ((e[0] | e[1]) == 0) free_slot = true;
            if (j == 0 && (e[1] & kDisplacedBit)) displaced = true;
        }
        if (free_slot || (home && !displaced)) return false;
        home = false;
        b = (b + 1) & T.short_mask;
    }
}

template <bool kLds, int kChunks, typename SymT>
__device__ __forceinline__ void score_start(const PatternTableView& T, const SymT* sym, int32_t* score, int s) {
    const uint32_t c1 = sym[s];
    if (c1 == 0) return;
    uint32_t c2 = sym[s + 1];
    uint32_t c3 = c2 ? uint32_t(sym[s + 2]) : 0u;
    if (T.debug & 2) c2 = 0;        // ablation: unigram rows only
    else if (T.debug & 4) c3 = 0;   // ablation: no trigram probes
    uint32_t e[4 * kChunks];

    // level 1
    if (c1 < T.uni_n) {
        const uint4* p = reinterpret_cast<const uint4*>(T.uni + size_t(c1) * T.uni_dw);
#pragma unroll
        for (int q = 0; q < kChunks; ++q) {
            if (uint32_t(4 * q) < T.uni_dw) {
                uint4 v = p[q];
                e[4 * q] = v.x; e[4 * q + 1] = v.y; e[4 * q + 2] = v.z; e[4 * q + 3] = v.w;
            }
        }
#pragma unroll
        for (int j = 0; j < 4 * kChunks - 2; ++j)
            if (j < T.len[0] && e[j] != 0) score_add<kLds>(score + s + T.lo[0] + j, int32_t(e[j]));
    } else if (probe_short<kChunks>(T, short_key(c1, 0, 0), e)) {
#pragma unroll
        for (int j = 0; j < 4 * kChunks - 2; ++j)
            if (j < T.len[0] && e[2 + j] != 0) score_add<kLds>(score + s + T.lo[0] + j, int32_t(e[2 + j]));
    }
    if (c2 == 0) return;
    // level 2
    if (probe_short<kChunks>(T, short_key(c1, c2, 0), e)) {
#pragma unroll
        for (int j = 0; j < 4 * kChunks - 2; ++j)
            if (j < T.len[1] && e[2 + j] != 0) score_add<kLds>(score + s + T.lo[1] + j, int32_t(e[2 + j]));
    }
    if (c3 == 0) return;
    // level 3 (+ continuation into the trie of longer strings)
    if (!probe_short<kChunks>(T, short_key(c1, c2, c3), e)) return;
    uint32_t node = 0;
#pragma unroll
    for (int j = 0; j < 4 * kChunks - 2; ++j) {
        if (j < T.len[2] && e[2 + j] != 0) score_add<kLds>(score + s + T.lo[2] + j, int32_t(e[2 + j]));
        if (uint32_t(j) == T.ext_slot) node = e[2 + j];
    }
    int n = 3;
    while (node != 0) {
        const uint32_t c = sym[s + n];
        if (c == 0) break;  // sentence end (separator) -- also bounds the walk
        const uint64_t key = edge_key(node, c);
        uint32_t b = hash_slot(key, T.edge_shift);
        uint4 ed = make_uint4(0, 0, 0, kNoRow);
        bool home = true;
        for (bool done = false; !done;) {  // buckets of four edges (layout.h)
            bool free_slot = false, displaced = false;
            for (uint32_t j = 0; j < kEdgeBucket && !done; ++j) {
                const uint4 c4 = reinterpret_cast<const uint4*>(T.edges)[size_t(b) * kEdgeBucket + j];
                if (c4.x == uint32_t(key) && (c4.y & ~kDisplacedBit) == uint32_t(key >> 32)) { ed = c4; done = true; }
                if ((c4.x | c4.y) == 0) free_slot = true;
                if (j == 0 && (c4.y & kDisplacedBit)) displaced = true;
            }
            if (free_slot || (home && !displaced)) done = true;
            home = false;
            b = (b + 1) & T.edge_mask;
        }
        node = ed.z & ~kHasKidsBit;
        ++n;
        if (ed.w != kNoRow) {
            const int lo = row_lo(n, T.window), len = row_len(n, T.window);
            const int32_t* w = T.wdata + ed.w;
            for (int j = 0; j < len; ++j) {
                int32_t v = w[j];
                if (v != 0) score_add<kLds>(score + s + lo + j, v);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// one tile: sentences [i0, i1)
// ------------------------------------------------------------------------------------------------------------
template <bool kLds, int kChunks>
__device__ __forceinline__ void process_tile(const ScoreParams& P, const TileMem<kLds>& M, uint32_t* bitmap,
                                             uint32_t* wtot, uint64_t i0, uint64_t i1, uint32_t flat_len) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pad = P.pad;
    const uint64_t B0 = P.boff[i0], B1 = P.boff[i1];
    const uint64_t O0 = P.ooff[i0];
    const uint32_t nsent = uint32_t(i1 - i0);
    const uint32_t expect_chars = uint32_t((P.ooff[i1] + i1) - (O0 + i0));

    // ---- zero the tile
    const uint32_t zlen = flat_len + kMargin;
    for (uint32_t i = tid; i < zlen; i += kThreads) {
        M.sym[i] = 0;
        M.typ[i] = 0;
        if (kLds) M.score[i] = 0;
        else __hip_atomic_store(M.score + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const uint8_t* tbase = P.text + B0;
    const uintptr_t a0 = reinterpret_cast<uintptr_t>(tbase) & ~uintptr_t(15);  // aligned loads never cross a page
    const uint32_t head = uint32_t(reinterpret_cast<uintptr_t>(tbase) - a0);
    const uint64_t nbytes_al = head + (B1 - B0);
    const uint64_t nchunks = (nbytes_al + 15) >> 4;
    if (kLds) {
        for (uint32_t i = tid; i < uint32_t((nbytes_al + 31) >> 5) + 1; i += kThreads) bitmap[i] = 0;
    }
    __syncthreads();
    // ---- sentence starts (LDS path: many sentences per tile; slow path: exactly one sentence)
    if (kLds) {
        for (uint32_t j = tid; j < nsent; j += kThreads) {
            const uint64_t b = P.boff[i0 + j], bn = P.boff[i0 + j + 1];
            if (bn <= b) raise(P.status, kErrEmptySentence);
            const uint32_t pos = head + uint32_t(b - B0);
            atomicOr(&bitmap[pos >> 5], 1u << (pos & 31));
        }
        __syncthreads();
    } else if (tid == 0 && B1 <= B0) raise(P.status, kErrEmptySentence);

    // ---- A. decode
    uint32_t base_leads = 0, base_starts = 0;
    for (uint64_t c0 = 0; c0 < nchunks; c0 += kThreads) {
        const uint64_t c = c0 + tid;
        uint32_t w[5] = {0, 0, 0, 0, 0};
        uint32_t lm = 0, sm = 0;
        if (c < nchunks) {
            const uint4 v = reinterpret_cast<const uint4*>(a0)[c];
            w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
            if ((c + 1) * 16 < nbytes_al) w[4] = reinterpret_cast<const uint32_t*>(a0)[(c + 1) * 4];
            const uint64_t pos0 = c * 16;
            const uint32_t lo = pos0 < head ? head - uint32_t(pos0) : 0u;
            const uint64_t rem = nbytes_al - pos0;
            const uint32_t hi = rem < 16 ? uint32_t(rem) : 16u;
            const uint32_t vm = ((1u << hi) - 1u) & ~((1u << lo) - 1u);
            lm = flag_bytes_to_mask16(lead_flags(w[0]), lead_flags(w[1]), lead_flags(w[2]), lead_flags(w[3])) & vm;
            if (kLds) sm = (bitmap[pos0 >> 5] >> (pos0 & 31)) & 0xFFFFu;
            else sm = (pos0 <= head && head < pos0 + 16) ? (1u << (head - uint32_t(pos0))) : 0u;
        }
        // block-wide exclusive scan of (lead count | start count << 16) in chunk order
        const uint32_t mine = __popc(lm) | (__popc(sm) << 16);
        uint32_t incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        uint32_t woff = 0, total = 0;
#pragma unroll
        for (int k = 0; k < kWaves; ++k) {
            const uint32_t t = wtot[k];
            if (k < wave) woff += t;
            total += t;
        }
        const uint32_t excl = woff + incl - mine;
        const uint32_t ci0 = base_leads + (excl & 0xFFFFu);
        const uint32_t si0 = base_starts + (excl >> 16);
        base_leads += total & 0xFFFFu;
        base_starts += total >> 16;
        __syncthreads();  // wtot is rewritten next iteration

#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (!((lm >> k) & 1u)) continue;
            const uint32_t below = (1u << k) - 1u;
            const uint32_t ci = ci0 + __popc(lm & below);
            const uint32_t si = si0 + __popc(sm & ((2u << k) - 1u)) - 1u;  // starts at or before this byte
            // bytes k .. k+3 of the 20-byte window
            const uint32_t q = k >> 2, r = (k & 3) * 8;
            const uint64_t two = (uint64_t(w[q + 1]) << 32) | w[q];
            const uint32_t b4 = uint32_t(two >> r);
            uint32_t cp = utf8_scalar(b4);
            if (cp == 0) raise(P.status, kErrNulChar);           // sentence.rs:174-179
            if (P.cinfo && cp < 0x10000u) cp = P.cinfo[cp] & 0xFFFFu;   // KyteaFullwidthFilter's image
            const uint32_t flat = uint32_t(pad) + ci + uint32_t(pad) * si;
            if (flat < flat_len) {
                M.sym[flat] = cp;
                M.typ[flat] = uint8_t(char_type(cp));
                if (kLds) M.sidx[flat] = uint16_t(si);
            } else raise(P.status, kErrBadOffsets);
        }
    }
    if (tid == 0 && base_leads != expect_chars) raise(P.status, kErrBadOffsets);
    __syncthreads();

    // ---- B. patterns
    if (P.ct.present && !(P.debug & 1)) {
        for (uint32_t s = tid; s < flat_len; s += kThreads) score_start<kLds, kChunks, uint32_t>(P.ct, M.sym, M.score, int(s));
    }
    if (P.type_kind == kTypePatternTable) {
        for (uint32_t s = tid; s < flat_len; s += kThreads) score_start<kLds, kChunks, uint8_t>(P.tt, M.typ, M.score, int(s));
    }
    __syncthreads();

    // ---- C. boundaries: bias + patterns + type window, sign threshold (predictor.rs:520-541)
    const int wt = P.type_window;
    for (uint32_t p = uint32_t(pad) + tid; p + 1 < flat_len; p += kThreads) {
        if (P.debug & 8) break;     // ablation: no output phase
        if (M.sym[p] == 0 || M.sym[p + 1] == 0) continue;
        int32_t y = P.bias + score_get<kLds>(M.score + p);
        if (P.type_kind == kTypeWindowTable) {
            uint32_t id = 0;  // window t[b-W+1 .. b+W], 3 bits each (boundary_scorer_cache.rs:59-81)
            for (int i = 1 - wt; i <= wt; ++i) id = (id << 3) | M.typ[int(p) + i];
            y += P.type_table[id];
        }
        const uint32_t si = kLds ? uint32_t(M.sidx[p]) : 0u;
        const uint64_t o = O0 + (p - uint32_t(pad)) - uint64_t(pad + 1) * si;
        if (o >= P.ooff[i1]) { raise(P.status, kErrBadOffsets); continue; }  // only with offsets that do not match the text
        if (P.scores) P.scores[o] = y;
        if (P.labels) {
            uint32_t label = y > 0 ? 1u : 0u;
            if (P.post) {   // KyteaWsConstFilter / SplitLinebreaksFilter on the label
                const uint32_t t1 = M.typ[p], t2 = M.typ[p + 1], ca = M.sym[p], cb = M.sym[p + 1];
                if (t1 == t2 && ((P.post >> t1) & 1u) && t1 >= 1 && t1 <= 6) label = 0;
                if ((P.post & 0x80u) && (ca == 0x0Au || ca == 0x0Du || cb == 0x0Au || cb == 0x0Du)) label = 1;
            }
            P.labels[o] = uint8_t(label);
        }
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------------------

// tile_first[t] = first sentence whose flat start F(i) = ooff[i] + i*(1+pad) is >= t*tile_flat  (t = 0..n_tiles)
// Also clears the batch's deferred-tile count for the scoring kernels that follow on the same stream -- one launch
// less than a separate memset.  The status word (ctrl[0]) is NOT cleared here: it accumulates over every call
// enqueued on the workspace until vpt_batch_sync reads and clears it, so an error of any of them is reported.
__global__ void assign_tiles_kernel(const uint64_t* __restrict__ ooff, uint64_t n_sent, int pad, uint32_t tile_flat,
                                    uint32_t n_tiles, uint32_t* __restrict__ tile_first, uint32_t* __restrict__ ctrl) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 1) ctrl[1] = 0;
    if (t > n_tiles) return;
    const uint64_t lo = first_sentence_at(ooff, n_sent, uint64_t(1 + pad), uint64_t(t) * tile_flat);
    tile_first[t] = t == n_tiles ? uint32_t(n_sent) : uint32_t(lo);
}

template <int kChunks>
__global__ __launch_bounds__(kThreads) void score_tiles_kernel(const ScoreParams P) {
    VPT_DYNAMIC_LDS(smem);
    TileMem<true> M;
    M.sym = reinterpret_cast<uint32_t*>(smem);
    M.score = reinterpret_cast<int32_t*>(M.sym + (kCap + kMargin));
    uint32_t* bitmap = reinterpret_cast<uint32_t*>(M.score + (kCap + kMargin));
    uint32_t* wtot = bitmap + kBitmapWords;
    M.sidx = reinterpret_cast<uint16_t*>(wtot + 8);
    M.typ = reinterpret_cast<uint8_t*>(M.sidx + (kCap + kMargin));

    const uint32_t t = blockIdx.x;
    const uint64_t i0 = P.tile_first[t], i1 = P.tile_first[t + 1];
    if (i0 >= i1) return;
    const uint64_t F0 = P.ooff[i0] + i0 * uint64_t(1 + P.pad), F1 = P.ooff[i1] + i1 * uint64_t(1 + P.pad);
    const uint64_t flat_len = uint64_t(P.pad) + (F1 - F0);
    const uint64_t nbytes = P.boff[i1] - P.boff[i0];
    if (flat_len > kCap || nbytes + 16 > uint64_t(kBitmapWords - 2) * 32) {  // does not fit in LDS: defer
        if (threadIdx.x == 0) {
            if (P.scratch_cap == 0) atomicOr(P.status, kErrScratchTooSmall);   // the caller's length bounds were understated
            else P.slow_list[atomicAdd(P.slow_count, 1u)] = t;
        }
        return;
    }
    process_tile<true, kChunks>(P, M, bitmap, wtot, i0, i1, uint32_t(flat_len));
}

// Sentences that do not fit the LDS tile: one sentence at a time per workgroup, tile arrays in a global slab.
template <int kChunks>
__global__ __launch_bounds__(kThreads) void score_slow_kernel(const ScoreParams P) {
    __shared__ uint32_t wtot[8];
    const uint32_t n_slow = *P.slow_count;
    unsigned char* slab = P.scratch + size_t(blockIdx.x) * P.scratch_stride;
    TileMem<false> M;
    M.sym = reinterpret_cast<uint32_t*>(slab);
    M.score = reinterpret_cast<int32_t*>(M.sym + P.scratch_cap);
    M.typ = reinterpret_cast<uint8_t*>(M.score + P.scratch_cap);
    M.sidx = nullptr;
    for (uint32_t k = blockIdx.x; k < n_slow; k += gridDim.x) {
        const uint32_t t = P.slow_list[k];
        const uint64_t i0 = P.tile_first[t], i1 = P.tile_first[t + 1];
        for (uint64_t i = i0; i < i1; ++i) {
            const uint64_t n_chars = P.ooff[i + 1] - P.ooff[i] + 1;
            const uint64_t flat_len = n_chars + 2 * uint64_t(P.pad);
            if (flat_len + kMargin > P.scratch_cap) {
                if (threadIdx.x == 0) raise(P.status, kErrScratchTooSmall);
                continue;
            }
            __syncthreads();
            process_tile<false, kChunks>(P, M, nullptr, wtot, i, i + 1, uint32_t(flat_len));
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------------------
size_t score_tiles_lds_bytes() {
    return size_t(kCap + kMargin) * (4 + 4 + 2 + 1) + size_t(kBitmapWords + 8) * 4;
}

hipError_t launch_assign_tiles(const uint64_t* ooff, uint64_t n_sent, int pad, uint32_t tile_flat, uint32_t n_tiles,
                               uint32_t* tile_first, uint32_t* ctrl, hipStream_t stream) {
    const uint32_t threads = 256, blocks = (n_tiles + 1 + threads - 1) / threads;
    hipLaunchKernelGGL(assign_tiles_kernel, dim3(blocks), dim3(threads), 0, stream, ooff, n_sent, pad, tile_flat, n_tiles, tile_first, ctrl);
    return hipGetLastError();
}

template <int kChunks>
static hipError_t launch_score_t(const ScoreParams& P, uint32_t n_tiles, hipStream_t stream) {
    hipLaunchKernelGGL(score_tiles_kernel<kChunks>, dim3(n_tiles), dim3(kThreads), score_tiles_lds_bytes(), stream, P);
    return hipGetLastError();
}
template <int kChunks>
static hipError_t launch_slow_t(const ScoreParams& P, uint32_t n_blocks, hipStream_t stream) {
    hipLaunchKernelGGL(score_slow_kernel<kChunks>, dim3(n_blocks), dim3(kThreads), 0, stream, P);
    return hipGetLastError();
}

hipError_t launch_score_tiles(const ScoreParams& P, int chunks, uint32_t n_tiles, hipStream_t stream) {
    switch (chunks) {
        case 2: return launch_score_t<2>(P, n_tiles, stream);
        case 3: return launch_score_t<3>(P, n_tiles, stream);
        case 4: return launch_score_t<4>(P, n_tiles, stream);
        case 5: return launch_score_t<5>(P, n_tiles, stream);
        default: return hipErrorInvalidValue;
    }
}
hipError_t launch_score_slow(const ScoreParams& P, int chunks, uint32_t n_blocks, hipStream_t stream) {
    switch (chunks) {
        case 2: return launch_slow_t<2>(P, n_blocks, stream);
        case 3: return launch_slow_t<3>(P, n_blocks, stream);
        case 4: return launch_slow_t<4>(P, n_blocks, stream);
        case 5: return launch_slow_t<5>(P, n_blocks, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace vpt
