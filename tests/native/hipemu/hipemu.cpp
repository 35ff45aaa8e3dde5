// TEST INFRASTRUCTURE ONLY -- the fiber scheduler and the fake runtime behind tests/native/hipemu/hip/hip_runtime.h.
// Never linked into libvaporetto_hip.so and never loaded by the vaporetto_amd package.
#include <map>
#include <mutex>
#if !defined(__x86_64__)
#include <ucontext.h>
#endif

#include <cstdio>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <vector>

#include "hip/hip_runtime.h"

namespace hipemu {

Idx g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;
bool g_reverse = std::getenv("HIPEMU_REVERSE") != nullptr;   // lane order of the scheduler; also hipemu_set_lane_order()

namespace {

constexpr size_t kStackBytes = 256 << 10;
constexpr size_t kRedZone = 256;          // bytes of poison on either side of every allocation
constexpr unsigned char kPoison = 0xCB;
constexpr size_t kMaxLds = 160 << 10;

// A fiber switch is the callee-saved registers and the stack pointer: swapcontext also saves the signal mask -- two system calls per
// switch, a third of the emulated suite's time -- so on x86-64 the switch is these fourteen instructions (ucontext elsewhere).
#if defined(__x86_64__)
struct Context { void* sp = nullptr; };
__attribute__((naked, noinline)) void switch_context(void** /* save the running stack pointer here: rdi */, void* /* and continue on this one: rsi */) {
    asm volatile(
        "pushq %rbp\n\tpushq %rbx\n\tpushq %r12\n\tpushq %r13\n\tpushq %r14\n\tpushq %r15\n\t"
        "movq %rsp, (%rdi)\n\t"
        "movq %rsi, %rsp\n\t"
        "popq %r15\n\tpopq %r14\n\tpopq %r13\n\tpopq %r12\n\tpopq %rbx\n\tpopq %rbp\n\t"
        "ret\n\t");
}
#else
struct Context { ucontext_t uc; };
#endif

struct Fiber {
    Context ctx;
    unsigned tid = 0;
    bool done = false;
};
struct Group {             // a wave or the workgroup
    uint64_t gen = 0;
    unsigned arrived = 0, live = 0;
};
struct Wave : Group {
    uint32_t buf[2][64];        // values deposited for the collective of generation g live in buf[g & 1]
    uint64_t present[2] = {0, 0};   // ... by these lanes (a lane may END before the others read its value)
    uint64_t stamp[2] = {0, 0};
    uint32_t* deposit(unsigned lane, uint32_t v, const uint64_t** who) {
        const unsigned i = unsigned(gen & 1);
        if (stamp[i] != gen + 1) { stamp[i] = gen + 1; present[i] = 0; }
        present[i] |= uint64_t(1) << lane;
        buf[i][lane] = v;
        *who = &present[i];
        return buf[i];
    }
};

std::recursive_mutex g_mu;        // one grid at a time
Context g_sched;
std::vector<Fiber> g_fibers;
std::unique_ptr<char[]> g_stacks;   // not touched until used
std::vector<Wave> g_waves;
Group g_block;
Fiber* g_cur = nullptr;
const std::function<void()>* g_body = nullptr;
alignas(16) unsigned char g_lds[kMaxLds];
uint64_t g_progress = 0;          // bumps whenever a group is released or a fiber ends
uint64_t g_clock = 0;
uint32_t g_garbage = 0x9E3779B9u;


[[noreturn]] void die(const char* msg) {
    std::fprintf(stderr, "hipemu: %s (block %u, thread %u)\n", msg, g_blockIdx.x, g_threadIdx.x);
    std::abort();
}

void switch_to(Context& from, Context& to) {
#if defined(__x86_64__)
    switch_context(&from.sp, to.sp);
#else
    swapcontext(&from.uc, &to.uc);
#endif
}

void yield() {
    Fiber* me = g_cur;
    switch_to(me->ctx, g_sched);
}

void release_if_complete(Group& g) {
    if (g.live > 0 && g.arrived == g.live) {
        g.arrived = 0;
        ++g.gen;
        ++g_progress;
    }
}

void arrive_and_wait(Group& g) {
    const uint64_t my = g.gen;
    ++g.arrived;
    release_if_complete(g);
    while (g.gen == my) yield();
}

Wave& my_wave() { return g_waves[g_cur->tid >> 6]; }

void fiber_main() {
    (*g_body)();
    Fiber* me = g_cur;
    me->done = true;
    ++g_progress;
    Wave& w = g_waves[me->tid >> 6];
    --w.live;
    release_if_complete(w);       // the hardware barrier counts only the waves (lanes) that are still running
    --g_block.live;
    release_if_complete(g_block);
#if defined(__x86_64__)
    switch_to(me->ctx, g_sched);   // for good: a finished fiber is not resumed
    std::abort();
#endif
    // (ucontext) returning switches to uc_link = the scheduler
}

void prepare_fiber(Fiber& f, char* stack, size_t bytes) {
#if defined(__x86_64__)
    // what switch_context pops on the first switch: six registers, then `ret` into fiber_main with the stack as after a call
    void** top = reinterpret_cast<void**>((reinterpret_cast<uintptr_t>(stack) + bytes) & ~uintptr_t(15)) - 2;   // 16-byte aligned slot of the entry address
    top[0] = reinterpret_cast<void*>(&fiber_main);
    top[1] = nullptr;                                                                                            // fiber_main's return address: never used
    for (int k = 1; k <= 6; ++k) top[-k] = nullptr;
    f.ctx.sp = top - 6;
#else
    getcontext(&f.ctx.uc);
    f.ctx.uc.uc_stack.ss_sp = stack;
    f.ctx.uc.uc_stack.ss_size = bytes;
    f.ctx.uc.uc_link = &g_sched.uc;
    makecontext(&f.ctx.uc, fiber_main, 0);
#endif
}

}  // namespace

unsigned char* dynamic_lds() { return g_lds; }
unsigned lane() { return g_cur->tid & 63u; }
uint64_t clock() { return ++g_clock; }

void wave_sync() { arrive_and_wait(my_wave()); }
void block_sync() { arrive_and_wait(g_block); }

uint64_t ballot(bool pred) {
    Wave& w = my_wave();
    const uint64_t* who;
    const uint32_t* b = w.deposit(lane(), pred ? 1u : 0u, &who);
    arrive_and_wait(w);
    uint64_t m = 0;
    for (unsigned l = 0; l < 64; ++l)
        if (((*who >> l) & 1u) && b[l]) m |= uint64_t(1) << l;
    return m;
}

uint32_t shfl_up(uint32_t v, unsigned delta) {
    Wave& w = my_wave();
    const unsigned l = lane();
    const uint64_t* who;
    const uint32_t* b = w.deposit(l, v, &who);
    arrive_and_wait(w);
    return (l >= delta && ((*who >> (l - delta)) & 1u)) ? b[l - delta] : v;
}

uint32_t shfl_xor(uint32_t v, unsigned mask) {
    Wave& w = my_wave();
    const unsigned l = lane(), from = (l ^ mask) & 63u;
    const uint64_t* who;
    const uint32_t* b = w.deposit(l, v, &who);
    arrive_and_wait(w);
    return ((*who >> from) & 1u) ? b[from] : v;
}

uint32_t readfirstlane(uint32_t v) {   // the value of the lowest lane that takes part
    Wave& w = my_wave();
    const uint64_t* who;
    const uint32_t* b = w.deposit(lane(), v, &who);
    arrive_and_wait(w);
    return b[__builtin_ctzll(*who)];
}

// DPP controls as in the CDNA ISA manual ("DPP_CTRL"); a lane whose source is invalid or whose row/bank is masked
// keeps `old` (or reads 0 with bound_ctrl when only the source is invalid)
uint32_t dpp(uint32_t old, uint32_t src, unsigned ctrl, unsigned row_mask, unsigned bank_mask, bool bound_ctrl) {
    Wave& w = my_wave();
    const unsigned l = lane();
    const uint64_t* who;
    const uint32_t* b = w.deposit(l, src, &who);
    arrive_and_wait(w);
    const unsigned row = l >> 4, in_row = l & 15u;
    int from = -1;   // source lane, -1 = invalid
    if (ctrl <= 0xFF) from = int((l & ~3u) | ((ctrl >> (2 * (l & 3u))) & 3u));                       // quad_perm
    else if (ctrl >= 0x101 && ctrl <= 0x10F) { const unsigned n = ctrl & 15u; if (in_row + n <= 15) from = int(l + n); }   // row_shl
    else if (ctrl >= 0x111 && ctrl <= 0x11F) { const unsigned n = ctrl & 15u; if (in_row >= n) from = int(l - n); }        // row_shr
    else if (ctrl >= 0x121 && ctrl <= 0x12F) { const unsigned n = ctrl & 15u; from = int((row << 4) | ((in_row - n) & 15u)); }  // row_ror
    else if (ctrl == 0x140) from = int((row << 4) | (15u - in_row));                                   // row_mirror
    else if (ctrl == 0x141) from = int((l & ~7u) | (7u - (l & 7u)));                                   // row_half_mirror
    else if (ctrl == 0x142) { if (row >= 1) from = int(((row - 1) << 4) | 15u); }                      // row_bcast:15
    else if (ctrl == 0x143) { if (row >= 2) from = 31; }                                               // row_bcast:31
    else die("unsupported DPP control");
    if (!((row_mask >> row) & 1u) || !((bank_mask >> (in_row >> 2)) & 1u)) return old;
    if (from >= 0 && !((*who >> unsigned(from)) & 1u)) from = -1;   // inactive source lane
    if (from < 0) return bound_ctrl ? 0u : old;
    return b[from];
}

void run_grid(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()>& body) {
    std::lock_guard<std::recursive_mutex> lock(g_mu);
    if (g_cur) die("nested launch");
    if (block.y != 1 || block.z != 1 || grid.y != 1 || grid.z != 1) die("only 1-D launches are emulated");
    if (lds_bytes > kMaxLds) die("dynamic LDS beyond 160 KB");
    if (block.x == 0 || block.x > 1024) die("bad block size");
    const unsigned n = block.x;
    if (g_fibers.size() != n) {
        g_fibers.assign(n, Fiber());
        g_stacks.reset(new char[size_t(n) * kStackBytes]);
    }
    g_waves.assign((n + 63) / 64, Wave());
    g_blockDim = Idx{n, 1, 1};
    g_gridDim = Idx{grid.x, 1, 1};
    g_body = &body;
    for (unsigned bx = 0; bx < grid.x; ++bx) {
        g_blockIdx = Idx{bx, 0, 0};
        for (size_t i = 0; i < kMaxLds / 4; ++i) {   // LDS is not zeroed between workgroups
            g_garbage = g_garbage * 1664525u + 1013904223u;
            reinterpret_cast<uint32_t*>(g_lds)[i] = g_garbage;
        }
        g_block = Group();
        g_block.live = n;
        for (size_t wv = 0; wv < g_waves.size(); ++wv) {
            g_waves[wv] = Wave();
            g_waves[wv].live = unsigned(n - wv * 64 < 64 ? n - wv * 64 : 64);
        }
        for (unsigned t = 0; t < n; ++t) {
            Fiber& f = g_fibers[t];
            f.tid = t;
            f.done = false;
            prepare_fiber(f, g_stacks.get() + size_t(t) * kStackBytes, kStackBytes);
        }
        unsigned left = n;
        while (left > 0) {
            const uint64_t before = g_progress;
            left = 0;
            for (unsigned k = 0; k < n; ++k) {
                const unsigned t = g_reverse ? n - 1 - k : k;   // HIPEMU_REVERSE=1: the other lane order must give the same results
                Fiber& f = g_fibers[t];
                if (f.done) continue;
                g_cur = &f;
                g_threadIdx = Idx{t, 0, 0};
                switch_to(g_sched, f.ctx);
                if (!f.done) ++left;
            }
            g_cur = nullptr;
            if (left > 0 && g_progress == before) die("deadlock: the lanes of a wave (workgroup) disagree on a cross-lane operation");
        }
    }
    g_body = nullptr;
}

}  // namespace hipemu

// The order in which the scheduler visits the lanes must not matter (0: ascending, 1: descending).
extern "C" void hipemu_set_lane_order(int descending) { hipemu::g_reverse = descending != 0; }

// ------------------------------------------------------------------------------------------------ fake runtime
namespace {
struct Header { size_t bytes; uint64_t magic; };
constexpr uint64_t kMagic = 0x48495045'4D554C21ull;
using hipemu::kPoison;
using hipemu::kRedZone;
}  // namespace

struct hipemu_stream { int unused; };
struct hipemu_event { int unused; };

// two pretend devices (one address space), so that the multi-device entry points run on the emulator too
hipError_t hipGetDeviceCount(int* n) { *n = 2; return hipSuccess; }
static thread_local int g_current_device = 0;
hipError_t hipSetDevice(int dev) { if (dev != 0 && dev != 1) return hipErrorInvalidValue; g_current_device = dev; return hipSuccess; }
hipError_t hipGetDevice(int* dev) { *dev = g_current_device; return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* prop, int) { prop->multiProcessorCount = 2; return hipSuccess; }

hipError_t hipMalloc(void** p, size_t bytes) {
    unsigned char* raw = static_cast<unsigned char*>(std::malloc(bytes + 2 * kRedZone + 64));
    if (!raw) { *p = nullptr; return hipErrorOutOfMemory; }
    std::memset(raw, kPoison, bytes + 2 * kRedZone + 64);   // device memory is not zeroed either
    Header h{bytes, kMagic};
    std::memcpy(raw, &h, sizeof(h));
    *p = raw + kRedZone;
    return hipSuccess;
}
hipError_t hipFree(void* p) {
    if (!p) return hipSuccess;
    unsigned char* user = static_cast<unsigned char*>(p);
    unsigned char* raw = user - kRedZone;
    Header h;
    std::memcpy(&h, raw, sizeof(h));
    if (h.magic != kMagic) { std::fprintf(stderr, "hipemu: hipFree of a pointer hipMalloc did not return\n"); std::abort(); }
    for (size_t i = sizeof(h); i < kRedZone; ++i)
        if (raw[i] != kPoison) { std::fprintf(stderr, "hipemu: write BEFORE a device allocation of %zu bytes\n", h.bytes); std::abort(); }
    for (size_t i = 0; i < kRedZone; ++i)
        if (user[h.bytes + i] != kPoison) { std::fprintf(stderr, "hipemu: write PAST a device allocation of %zu bytes (+%zu)\n", h.bytes, i); std::abort(); }
    std::free(raw);
    return hipSuccess;
}
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind) { std::memcpy(dst, src, bytes); return hipSuccess; }
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind, hipStream_t) { std::memcpy(dst, src, bytes); return hipSuccess; }
hipError_t hipMemcpyPeer(void* dst, int, const void* src, int, size_t bytes) { std::memcpy(dst, src, bytes); return hipSuccess; }
// pinned host memory: plain malloc, remembered so that hipPointerGetAttributes can tell it from pageable memory (the product writes
// tokenized text straight into a pinned output buffer and copies into a pageable one)
namespace { std::mutex g_pinned_mu; std::map<uintptr_t, size_t> g_pinned; }
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned) {
    *p = std::malloc(bytes ? bytes : 1);
    if (!*p) return hipErrorOutOfMemory;
    std::lock_guard<std::mutex> g(g_pinned_mu);
    g_pinned[reinterpret_cast<uintptr_t>(*p)] = bytes ? bytes : 1;
    return hipSuccess;
}
hipError_t hipHostFree(void* p) {
    { std::lock_guard<std::mutex> g(g_pinned_mu); g_pinned.erase(reinterpret_cast<uintptr_t>(p)); }
    std::free(p);
    return hipSuccess;
}
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* attr, const void* ptr) {
    attr->type = hipMemoryTypeUnregistered; attr->device = 0; attr->devicePointer = nullptr; attr->hostPointer = const_cast<void*>(ptr);
    std::lock_guard<std::mutex> g(g_pinned_mu);
    const uintptr_t a = reinterpret_cast<uintptr_t>(ptr);
    auto it = g_pinned.upper_bound(a);
    if (it != g_pinned.begin()) {
        --it;
        if (a >= it->first && a < it->first + it->second) { attr->type = hipMemoryTypeHost; attr->devicePointer = const_cast<void*>(ptr); }
    }
    return hipSuccess;
}
hipError_t hipHostGetDevicePointer(void** dptr, void* hptr, unsigned) { *dptr = hptr; return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }   // (work is done when it is enqueued)
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new hipemu_event(); return hipSuccess; }
hipError_t hipMemset(void* dst, int value, size_t bytes) { std::memset(dst, value, bytes); return hipSuccess; }
hipError_t hipMemsetAsync(void* dst, int value, size_t bytes, hipStream_t) { std::memset(dst, value, bytes); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = new hipemu_stream(); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipemu_event(); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : e == hipErrorOutOfMemory ? "hipErrorOutOfMemory" : "hipErrorInvalidValue"; }
