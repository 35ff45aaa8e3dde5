// Device helpers shared by the general (kernels.hip) and specialised (kernels_fast.hip) tile kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

// the workgroup's dynamic LDS as a byte array (the CPU emulator of tests/native/hipemu supplies its own definition)
#ifndef VPT_DYNAMIC_LDS
#define VPT_DYNAMIC_LDS(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#endif

// The lanes of a wave execute in lock step, so an LDS write by one lane is visible to the next instruction of
// every other lane of that wave.  Where the kernels rely on that WITHOUT a cross-lane operation in between, they say
// so with this marker: nothing on the GPU, a wave rendezvous in the emulator (whose lanes are only ordered by
// cross-lane operations).
#ifndef VPT_WAVE_LOCKSTEP
#define VPT_WAVE_LOCKSTEP() ((void)0)
#endif

// Have a value computed HERE: the compiler otherwise sinks a chain of selects to its first use and keeps every input
// register and every compare mask alive until then (VGPR and SGPR pressure in the main loop of the specialised kernel).
#ifndef VPT_PIN
#define VPT_PIN(x) __asm__ volatile("" : "+v"(x))
#endif

// A pointer that is KNOWN to point into LDS (a function argument; a plain pointer there is a flat one).  hipcc 7.2 fails on tag_token_by_wave's
// flat scratch pointer next to its global stores ("Illegal instruction detected: Operand has incorrect register class. V_CMP_NE_U32_e32 0,
// $src_shared_base"); with the address space in the type there is no flat access to lower.  The CPU emulator of the tests has one address space.
#if defined(VPT_HIPEMU)
#define VPT_LDS_PTR(T) T*
#define VPT_TO_LDS_PTR(T, p) (p)
#else
#define VPT_LDS_PTR(T) __attribute__((address_space(3))) T*
#define VPT_TO_LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))
#endif

// The kernel's parameter block read WHERE IT IS USED.  A by-value kernel argument is loaded from the kernarg segment at the kernel's entry
// and kept in scalar registers to its last use; the specialised scoring kernel has 70-odd such words and 72 SGPRs at 8 waves per SIMD
// (800 per SIMD, 16 of every wave's allocation reserved for the trap handler), so the compiler spilled 84 of them to VGPR lanes.  The
// kernel reads its block through a pointer into the kernarg segment instead (scalar loads out of the constant cache, a phase's fields
// asked for at the top of the phase) and puts a FENCE between phases, behind which nothing loaded earlier is assumed still at hand.
// VPT_UNDEF4: a uint4 whose value does not matter until a guarded load has filled it (no zeroing moves).
#ifndef VPT_KARG
#define VPT_KARG(T) const __attribute__((address_space(4))) T*
#define VPT_KARG_PTR(T, arg) ((VPT_KARG(T))__builtin_amdgcn_kernarg_segment_ptr())
#if defined(VPT_AB_NO_KARG_FENCE)   /* A/B builds (tools/build_variants.sh) */
#define VPT_KARG_FENCE(p) ((void)0)
#else
#define VPT_KARG_FENCE(p) __asm__ volatile("" : "+s"(p))
#endif
#if defined(VPT_AB_ZERO_NODES)
#define VPT_UNDEF4(v) ((v) = make_uint4(0, 0, 0, 0))
#else
#define VPT_UNDEF4(v) __asm__ volatile("" : "=v"((v).x), "=v"((v).y), "=v"((v).z), "=v"((v).w))
#endif
#endif

namespace vpt {

// A value that every lane of the wave holds alike, moved to a scalar register.  The hardware gains nothing; the
// COMPILER learns that branches and loop exits depending on it are wave-uniform, so it keeps counters in SGPRs and
// emits scalar branches instead of exec-mask loops (thread ids and LDS loads are divergent as far as it can tell).
__device__ __forceinline__ uint32_t wave_uniform(uint32_t x) { return uint32_t(__builtin_amdgcn_readfirstlane(int(x))); }
__device__ __forceinline__ uint64_t wave_uniform64(uint64_t x) { return uint64_t(wave_uniform(uint32_t(x))) | (uint64_t(wave_uniform(uint32_t(x >> 32))) << 32); }

// First sentence i in [0, n_sent] whose flat start F(i) = ooff[i] + i * (1 + pad) is >= target (F is non-decreasing and
// F(n_sent) is the total).  Sentences of similar length make F nearly linear: start from the interpolated position
// and gallop outwards (two or three dependent loads instead of log2(n_sent)), then bisect.
__device__ __forceinline__ uint64_t first_sentence_at(const uint64_t* __restrict__ ooff, uint64_t n_sent, uint64_t step, uint64_t target) {
    auto F = [&](uint64_t i) { return ooff[i] + i * step; };
    const uint64_t total = F(n_sent);
    uint64_t g = total ? uint64_t((unsigned __int128)(target) * n_sent / total) : 0;
    if (g > n_sent) g = n_sent;
    uint64_t lo, hi;
    if (F(g) >= target) {          // answer <= g: gallop down to an i with F(i) < target (or 0)
        hi = g;
        uint64_t w = 1;
        lo = 0;
        while (hi - lo > 0) {
            const uint64_t p = g >= w ? g - w : 0;
            if (F(p) < target) { lo = p + 1; break; }
            hi = p;
            if (p == 0) { lo = 0; break; }
            w <<= 2;
        }
    } else {                       // answer > g: gallop up to an i with F(i) >= target (n_sent at the latest)
        lo = g + 1;
        uint64_t w = 1;
        hi = n_sent;
        for (;;) {
            const uint64_t p = g + w < n_sent ? g + w : n_sent;
            if (F(p) >= target) { hi = p; break; }
            lo = p + 1;
            if (p == n_sent) { hi = n_sent; break; }
            w <<= 2;
        }
    }
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (F(mid) >= target) hi = mid; else lo = mid + 1;
    }
    return lo < n_sent ? lo : n_sent;   // a target past the total (a caller that only knows an upper bound of it): no sentence starts there
}

// inclusive prefix sum over the 64 lanes of a wave: DPP row shifts, then the two row broadcasts (no LDS round trips)
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t x) {
    x += uint32_t(__builtin_amdgcn_update_dpp(0, int(x), 0x111, 0xF, 0xF, false));  // row_shr:1
    x += uint32_t(__builtin_amdgcn_update_dpp(0, int(x), 0x112, 0xF, 0xF, false));  // row_shr:2
    x += uint32_t(__builtin_amdgcn_update_dpp(0, int(x), 0x114, 0xF, 0xF, false));  // row_shr:4
    x += uint32_t(__builtin_amdgcn_update_dpp(0, int(x), 0x118, 0xF, 0xF, false));  // row_shr:8
    x += uint32_t(__builtin_amdgcn_update_dpp(0, int(x), 0x142, 0xA, 0xF, false));  // row_bcast:15 -> rows 1, 3
    x += uint32_t(__builtin_amdgcn_update_dpp(0, int(x), 0x143, 0xC, 0xF, false));  // row_bcast:31 -> rows 2, 3
    return x;
}
// maximum over the 64 lanes of a wave, in every lane: the same DPP steps (row_bcast feeds the rows above; the last lane holds the
// total, which readlane 63 hands to everybody)
__device__ __forceinline__ uint32_t wave_max(uint32_t x) {
    auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
    x = mx(x, uint32_t(__builtin_amdgcn_update_dpp(0, int(x), 0x111, 0xF, 0xF, false)));  // row_shr:1
    x = mx(x, uint32_t(__builtin_amdgcn_update_dpp(0, int(x), 0x112, 0xF, 0xF, false)));  // row_shr:2
    x = mx(x, uint32_t(__builtin_amdgcn_update_dpp(0, int(x), 0x114, 0xF, 0xF, false)));  // row_shr:4
    x = mx(x, uint32_t(__builtin_amdgcn_update_dpp(0, int(x), 0x118, 0xF, 0xF, false)));  // row_shr:8
    x = mx(x, uint32_t(__builtin_amdgcn_update_dpp(0, int(x), 0x142, 0xA, 0xF, false)));  // row_bcast:15 -> rows 1, 3
    x = mx(x, uint32_t(__builtin_amdgcn_update_dpp(0, int(x), 0x143, 0xC, 0xF, false)));  // row_bcast:31 -> rows 2, 3
    return uint32_t(__builtin_amdgcn_readlane(int(x), 63));
}

// CharacterType::get_type (sentence.rs:50-67): 1 Digit, 2 Roman, 3 Hiragana, 4 Katakana, 5 Kanji, 6 Other
__device__ __forceinline__ uint32_t char_type(uint32_t c) {
    if ((c - 0x30u) <= 9u || (c - 0xFF10u) <= 9u) return 1;
    if ((c - 0x41u) <= 25u || (c - 0x61u) <= 25u || (c - 0xFF21u) <= 25u || (c - 0xFF41u) <= 25u) return 2;
    if ((c - 0x3040u) <= (0x3096u - 0x3040u)) return 3;
    if ((c - 0x30A0u) <= (0x30FAu - 0x30A0u) || (c - 0x30FCu) <= 3u || (c - 0xFF66u) <= (0xFF9Fu - 0xFF66u))
        return 4;
    if ((c - 0x3400u) <= (0x4DBFu - 0x3400u) || (c - 0x4E00u) <= (0x9FFFu - 0x4E00u) ||
        (c - 0xF900u) <= (0xFAFFu - 0xF900u) || (c - 0x20000u) <= (0x2A6DFu - 0x20000u) ||
        (c - 0x2A700u) <= (0x2B73Fu - 0x2A700u) || (c - 0x2B740u) <= (0x2B81Fu - 0x2B740u) ||
        (c - 0x2B820u) <= (0x2CEAFu - 0x2B820u) || (c - 0x2F800u) <= (0x2FA1Fu - 0x2F800u))
        return 5;
    return 6;
}

// bytes [at, at + 4) of p as a little-endian dword, zero at and past `end`: the one or two ALIGNED dwords that hold the wanted
// bytes, both loads issued together -- no byte loop at a sentence's end (the whole wave would wait three trips for its
// last lane), and nothing is read outside the aligned words the wanted bytes live in
__device__ __forceinline__ uint32_t load4(const uint8_t* __restrict__ p, uint64_t at, uint64_t end) {
    if (at >= end) return 0;
    const uint32_t nv = end - at < 4 ? uint32_t(end - at) : 4u;
    const uintptr_t addr = reinterpret_cast<uintptr_t>(p + at);
    const uint32_t sh = uint32_t(addr & 3u);
    const uint32_t* a0 = reinterpret_cast<const uint32_t*>(addr - sh);
    const uint32_t d0 = a0[0];
    const uint32_t d1 = sh + nv > 4u ? a0[1] : 0u;
    const uint32_t x = __builtin_amdgcn_alignbyte(d1, d0, sh);
    return nv == 4u ? x : x & ((1u << (8u * nv)) - 1u);
}

// values the compiler must take as new at this point (no common subexpressions with what was computed from them before, nothing hoisted across)
#ifdef VPT_HIPEMU
#define VPT_OPAQUE3(a, b, c) ((void)0)
#else
#define VPT_OPAQUE3(a, b, c) asm volatile("" : "+v"(a), "+v"(b), "+v"(c))
#endif

// bits 7, 15, 23, 31 -> bits 0..3.  Bits 7, 15, 23 travel to bits 21, 22, 23 of a 24-bit product (0x4081 = bits 0, 7, 14: the nine
// partial products fall on nine different bits) -- a full-rate v_mul_u32_u24, where the 32-bit multiply this used to be (v_mul_lo_u32)
// takes four issue slots and every chunk of text needs eight of them -- and bit 31 is placed by hand.
__device__ __forceinline__ uint32_t byte_flags_to_nibble(uint32_t m) {
    return ((((m & 0x00808080u) * 0x4081u) >> 21) & 7u) | ((m >> 31) << 3);
}
// 4-bit mask of the bytes of x that are NOT UTF-8 continuation bytes (10xxxxxx): bit 7 clear or bit 6 set
__device__ __forceinline__ uint32_t lead_flags(uint32_t x) { return (~x | (x << 1)) & 0x80808080u; }   // 0x80 in every such byte
__device__ __forceinline__ uint32_t lead_nibble(uint32_t x) { return byte_flags_to_nibble(lead_flags(x)); }
// Sixteen byte flags (0x80 or 0 in every byte of four dwords) -> a 16-bit mask, byte k of the sixteen -> bit k.  v_dot4_u32_u8 weighs a dword's four
// bytes in one issue slot (round 6: the writer was found to run at the vector ALU's issue rate, and a third of its instructions gathered flag bits
// nibble by nibble -- byte_flags_to_nibble above, five slots a dword and three more to join the nibbles); a flag byte adds 0x80 x its weight.
__device__ __forceinline__ uint32_t flag_bytes_to_mask16(uint32_t f0, uint32_t f1, uint32_t f2, uint32_t f3) {
    uint32_t lo = __builtin_amdgcn_udot4(f0, 0x08040201u, 0u, false), hi = __builtin_amdgcn_udot4(f2, 0x08040201u, 0u, false);
    lo = __builtin_amdgcn_udot4(f1, 0x80402010u, lo, false);
    hi = __builtin_amdgcn_udot4(f3, 0x80402010u, hi, false);
    return (lo + (hi << 8)) >> 7;
}
// 16-bit mask over the 16 bytes of v
__device__ __forceinline__ uint32_t lead_mask16(const uint4& v) { return flag_bytes_to_mask16(lead_flags(v.x), lead_flags(v.y), lead_flags(v.z), lead_flags(v.w)); }

// 0x80 in every byte of v that is zero (exact: no carries between the bytes)
__device__ __forceinline__ uint32_t zero_bytes(uint32_t v) { return ~(((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v | 0x7F7F7F7Fu); }
__device__ __forceinline__ uint32_t zero_mask16(const uint4& v) { return flag_bytes_to_mask16(zero_bytes(v.x), zero_bytes(v.y), zero_bytes(v.z), zero_bytes(v.w)); }
// 0x80 in the bytes of x that Sentence::write_tokenized_text escapes: ' ', '\', '/' (sentence.rs:850-886).  All three are below 0x80: the low
// seven bits are compared (x7 ^ c is at most 0x7F: adding 0x7F sets bit 7 unless it is zero, and carries nowhere), bit 7 of x itself must be clear
__device__ __forceinline__ uint32_t esc_flags(uint32_t x) {
    const uint32_t x7 = x & 0x7F7F7F7Fu;
    const uint32_t a = (x7 ^ 0x20202020u) + 0x7F7F7F7Fu, b = (x7 ^ 0x5C5C5C5Cu) + 0x7F7F7F7Fu, c = (x7 ^ 0x2F2F2F2Fu) + 0x7F7F7F7Fu;
    return ~((a & b & c) | x) & 0x80808080u;
}
__device__ __forceinline__ uint32_t esc_nibble(uint32_t x) { return byte_flags_to_nibble(esc_flags(x)); }
__device__ __forceinline__ uint32_t esc_mask16(const uint4& v) { return flag_bytes_to_mask16(esc_flags(v.x), esc_flags(v.y), esc_flags(v.z), esc_flags(v.w)); }

// scalar value of the UTF-8 sequence whose four bytes (lead first) are packed little-endian in b4
__device__ __forceinline__ uint32_t utf8_scalar(uint32_t b4) {
    const uint32_t b0 = b4 & 0xFF, b1 = (b4 >> 8) & 0x3F, b2 = (b4 >> 16) & 0x3F, b3 = (b4 >> 24) & 0x3F;
    if (b0 < 0x80) return b0;
    if (b0 < 0xE0) return ((b0 & 0x1F) << 6) | b1;
    if (b0 < 0xF0) return ((b0 & 0x0F) << 12) | (b1 << 6) | b2;
    return ((b0 & 0x07) << 18) | (b1 << 12) | (b2 << 6) | b3;
}

}  // namespace vpt
