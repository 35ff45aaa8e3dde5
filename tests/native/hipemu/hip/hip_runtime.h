// TEST INFRASTRUCTURE ONLY -- never part of the product, never loaded by vaporetto_amd.
//
// A minimal stand-in for <hip/hip_runtime.h> that lets g++ compile the kernel SOURCES of vaporetto_amd/csrc
// (kernels*.hip) and the host side of the C ABI for the CPU, so that the `-m "not gpu"` tests can run the very
// code the GPU runs -- the LDS tile layout, the wave stacks, the DPP exchanges -- against the oracle without a GPU
// (tests/test_kernel_emu.py).  It is an interpreter of the execution MODEL, not a second implementation:
//
//   * a workgroup is `blockDim.x` fibers (hipemu.cpp) on one OS thread; a fiber runs until it reaches a
//     cross-lane operation (__ballot, DPP, __shfl_up, wave barrier, __syncthreads), where it waits for the other
//     lanes of its wave / workgroup.  Nothing else orders the lanes: code that silently relies on lock-step
//     execution between two such points fails here (a stricter model than the hardware's);
//   * "device memory" is host memory with red zones (checked on hipFree) filled with a poison pattern, and the
//     dynamic LDS of every workgroup starts out as garbage;
//   * workgroups run one after the other; streams are synchronous; events measure nothing.
//
// Only what vaporetto_amd/csrc uses is provided; anything else fails to compile or aborts with a message.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>

#define VPT_HIPEMU 1

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define __shared__ static   /* static LDS: one workgroup runs at a time */
// dynamic LDS: kernels declare it through VPT_DYNAMIC_LDS (device_common.h)
#define VPT_DYNAMIC_LDS(name) unsigned char* const name = ::hipemu::dynamic_lds()
// lock-step marker of the kernels (device_common.h): the lanes of the wave meet here
#define VPT_WAVE_LOCKSTEP() ::hipemu::wave_sync()
#define VPT_PIN(x) ((void)0)   /* a code-generation hint on the GPU */
// the kernel's parameter block through a pointer (device_common.h): here simply the by-value argument
#define VPT_KARG(T) const T*
#define VPT_KARG_PTR(T, arg) (&(arg))
#define VPT_KARG_FENCE(p) ((void)0)
#define VPT_UNDEF4(v) ((v) = make_uint4(0xDEADBEEFu, 0xDEADBEEFu, 0xDEADBEEFu, 0xDEADBEEFu))   /* poison: a use before the guarded load shows */

struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; } __attribute__((aligned(16)));
struct int2 { int32_t x, y; } __attribute__((aligned(8)));
struct int4 { int32_t x, y, z, w; } __attribute__((aligned(16)));
static inline int2 make_int2(int32_t x, int32_t y) { return int2{x, y}; }
static inline int4 make_int4(int32_t x, int32_t y, int32_t z, int32_t w) { return int4{x, y, z, w}; }
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

// ---------------------------------------------------------------------------------------------- runtime API
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
enum { hipStreamNonBlocking = 1, hipHostMallocDefault = 0, hipEventDisableTiming = 2 };
struct hipemu_stream;
struct hipemu_event;
typedef hipemu_stream* hipStream_t;
typedef hipemu_event* hipEvent_t;
struct hipDeviceProp_t { int multiProcessorCount; };

hipError_t hipGetDeviceCount(int* n);
hipError_t hipSetDevice(int dev);
hipError_t hipGetDevice(int* dev);
hipError_t hipGetDeviceProperties(hipDeviceProp_t* prop, int dev);
hipError_t hipMalloc(void** p, size_t bytes);
hipError_t hipFree(void* p);
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind);
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s);
hipError_t hipMemcpyPeer(void* dst, int dst_device, const void* src, int src_device, size_t bytes);
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned flags);
hipError_t hipHostFree(void* p);
enum hipMemoryType { hipMemoryTypeUnregistered = 0, hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2, hipMemoryTypeManaged = 3 };
struct hipPointerAttribute_t { hipMemoryType type; int device; void* devicePointer; void* hostPointer; };
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* attr, const void* ptr);
hipError_t hipHostGetDevicePointer(void** dptr, void* hptr, unsigned flags);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventQuery(hipEvent_t e);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipMemset(void* dst, int value, size_t bytes);
hipError_t hipMemsetAsync(void* dst, int value, size_t bytes, hipStream_t s);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipDeviceSynchronize();
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipGetLastError();
const char* hipGetErrorString(hipError_t e);

// ---------------------------------------------------------------------------------------------- execution model
namespace hipemu {
struct Idx { unsigned x, y, z; };
extern Idx g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;
unsigned char* dynamic_lds();
unsigned lane();                       // threadIdx.x & 63
void run_grid(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()>& body);
// collectives: every live lane of the wave (workgroup) must call them in the same order
uint64_t ballot(bool pred);
uint32_t dpp(uint32_t old, uint32_t src, unsigned ctrl, unsigned row_mask, unsigned bank_mask, bool bound_ctrl);
uint32_t shfl_up(uint32_t v, unsigned delta);
uint32_t shfl_xor(uint32_t v, unsigned mask);
uint32_t readfirstlane(uint32_t v);
void wave_sync();
void block_sync();
uint64_t clock();
}  // namespace hipemu

#define threadIdx (::hipemu::g_threadIdx)
#define blockIdx (::hipemu::g_blockIdx)
#define blockDim (::hipemu::g_blockDim)
#define gridDim (::hipemu::g_gridDim)

#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) \
    ::hipemu::run_grid((grid), (block), (lds), [=]() { kernel(__VA_ARGS__); })

static inline void __syncthreads() { ::hipemu::block_sync(); }
static inline uint64_t __ballot(int pred) { return ::hipemu::ballot(pred != 0); }
static inline int __shfl_up(int v, unsigned d) { return int(::hipemu::shfl_up(uint32_t(v), d)); }
static inline unsigned __shfl_up(unsigned v, unsigned d) { return ::hipemu::shfl_up(v, d); }
static inline int __shfl(int v, int src) { return int(::hipemu::shfl_xor(uint32_t(v), unsigned(src) ^ ::hipemu::lane())); }
static inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
static inline int __shfl_xor(int v, int m) { return int(::hipemu::shfl_xor(uint32_t(v), unsigned(m))); }
static inline int __popc(uint32_t x) { return __builtin_popcount(x); }
static inline int __popcll(uint64_t x) { return __builtin_popcountll(x); }
static inline int __ffs(int x) { return __builtin_ffs(x); }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
static inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }

// atomics: one fiber runs at a time, so plain read-modify-write is atomic
static inline int atomicAdd(int* p, int v) { int o = *p; *p = int(uint32_t(o) + uint32_t(v)); return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p = o + v; return o; }
static inline unsigned atomicOr(unsigned* p, unsigned v) { unsigned o = *p; *p = o | v; return o; }
static inline unsigned atomicMax(unsigned* p, unsigned v) { unsigned o = *p; *p = o > v ? o : v; return o; }
static inline unsigned atomicMin(unsigned* p, unsigned v) { unsigned o = *p; *p = o < v ? o : v; return o; }
#define __HIP_MEMORY_SCOPE_AGENT 0
template <typename T, typename V> static inline T __hip_atomic_fetch_add(T* p, V v, int, int) {
    T o = *p;
    *p = T(uint32_t(o) + uint32_t(v));
    return o;
}
template <typename T> static inline T __hip_atomic_load(const T* p, int, int) { return *p; }
template <typename T, typename V> static inline void __hip_atomic_store(T* p, V v, int, int) { *p = T(v); }

// the gfx950 builtins the kernels use
#define __builtin_amdgcn_mov_dpp(src, ctrl, rm, bm, bc) int(::hipemu::dpp(0xDEADBEEFu, uint32_t(src), (ctrl), (rm), (bm), (bc)))
#define __builtin_amdgcn_readlane(v, l) int(::hipemu::shfl_xor(uint32_t(v), unsigned(l) ^ ::hipemu::lane()))
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) int(::hipemu::dpp(uint32_t(old), uint32_t(src), (ctrl), (rm), (bm), (bc)))
#define __builtin_amdgcn_readfirstlane(x) int(::hipemu::readfirstlane(uint32_t(x)))
#define __builtin_amdgcn_wave_barrier() ::hipemu::wave_sync()
#define __builtin_amdgcn_s_memtime() ::hipemu::clock()
#define __builtin_amdgcn_s_sleep(n) ((void)0)
#define __builtin_amdgcn_s_setprio(n) ((void)0)
#define __builtin_amdgcn_sched_barrier(m) ((void)0)
static inline uint32_t hipemu_mbcnt_lo(uint32_t mask, uint32_t base) {
    const unsigned l = ::hipemu::lane();
    return base + uint32_t(__builtin_popcount(l >= 32 ? mask : mask & ((1u << l) - 1u)));
}
static inline uint32_t hipemu_mbcnt_hi(uint32_t mask, uint32_t base) {
    const unsigned l = ::hipemu::lane();
    return base + uint32_t(l <= 32 ? 0 : __builtin_popcount(mask & ((1u << (l - 32)) - 1u)));
}
#define __builtin_amdgcn_mbcnt_lo(mask, base) hipemu_mbcnt_lo((mask), (base))
#define __builtin_amdgcn_mbcnt_hi(mask, base) hipemu_mbcnt_hi((mask), (base))
static inline uint32_t hipemu_alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) {
    return uint32_t(((uint64_t(hi) << 32) | lo) >> (8 * (sh & 3u)));
}
static inline uint32_t hipemu_alignbit(uint32_t hi, uint32_t lo, uint32_t sh) {
    return uint32_t(((uint64_t(hi) << 32) | lo) >> (sh & 31u));
}
static inline uint32_t hipemu_udot4(uint32_t a, uint32_t b, uint32_t c) {   // v_dot4_u32_u8 (no clamp)
    for (int k = 0; k < 4; ++k) c += ((a >> (8 * k)) & 0xFFu) * ((b >> (8 * k)) & 0xFFu);
    return c;
}
#define __builtin_amdgcn_udot4(a, b, c, clamp) hipemu_udot4((a), (b), (c))
#define __builtin_amdgcn_alignbyte(hi, lo, sh) hipemu_alignbyte((hi), (lo), (sh))
#define __builtin_amdgcn_alignbit(hi, lo, sh) hipemu_alignbit((hi), (lo), (sh))
// a wave-uniform 64-bit mask as a lane predicate (the mask goes to EXEC / VCC as it is)
#define __builtin_amdgcn_inverse_ballot_w64(m) (((uint64_t(m) >> ::hipemu::lane()) & 1u) != 0)
