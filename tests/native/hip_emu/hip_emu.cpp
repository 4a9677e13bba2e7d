// TEST INFRASTRUCTURE ONLY -- the runtime behind tests/native/hip_emu/hip/hip_runtime.h: fibers (one per GPU thread of
// the workgroup being run), lock-step wave collectives and workgroup barriers, and trivial stand-ins for the HIP runtime
// calls the library makes (memory is host memory, streams are synchronous).
#include <hip/hip_runtime.h>

#include <cassert>
#include <vector>

namespace hipemu {

Idx g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;
uint32_t g_lane;

namespace {

extern "C" void hipemu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size hipemu_switch,.-hipemu_switch
)");

constexpr size_t kStackBytes = 256 * 1024;

constexpr int kSnapshots = 16;

struct Wave {
    uint64_t val[64];                 // value each arrived lane deposited
    const void* site[64];             // ... and the call site it waits at
    uint64_t arrived_mask = 0, alive_mask = 0;
    int alive = 0;
    // a completed collective: the values and the mask of the lanes that took part, kept until its members have read them
    uint64_t snap_val[kSnapshots][64];
    uint64_t snap_live[kSnapshots];
    int next_snap = 0;
    int lane_snap[64];                // snapshot a released lane reads
    uint32_t lane_gen[64];            // bumped when the lane is released
};

struct Fiber {
    void* sp = nullptr;
    char* stack = nullptr;
    uint32_t tid = 0, lane = 0, wave = 0;
    bool done = false;
    const uint32_t* wait_ptr = nullptr;  // resumable once *wait_ptr != wait_val
    uint32_t wait_val = 0;
};

std::vector<Fiber> g_fibers;
std::vector<Wave> g_waves;
std::vector<char*> g_stack_pool;
Fiber* g_cur = nullptr;
void* g_sched_sp = nullptr;
const std::function<void()>* g_body = nullptr;
uint64_t g_events = 0;            // collectives / barriers completed and fibers finished: progress of the workgroup
int g_bar_arrived = 0, g_block_alive = 0;
uint32_t g_bar_gen = 0;
std::vector<char> g_dyn_lds;

void yield_to_scheduler() {
    Fiber* f = g_cur;
    hipemu_switch(&f->sp, g_sched_sp);
}

void wait_until_changed(const uint32_t* p, uint32_t v) {
    while (*p == v) {
        g_cur->wait_ptr = p;
        g_cur->wait_val = v;
        yield_to_scheduler();
    }
    g_cur->wait_ptr = nullptr;
}

// releases the lanes in `members` (all waiting at one call site) with the values they deposited
void release(Wave& w, uint64_t members) {
    const int sn = w.next_snap;
    w.next_snap = (w.next_snap + 1) % kSnapshots;
    std::memcpy(w.snap_val[sn], w.val, sizeof(w.val));
    w.snap_live[sn] = members;
    for (int l = 0; l < 64; ++l)
        if ((members >> l) & 1u) {
            w.lane_snap[l] = sn;
            ++w.lane_gen[l];
        }
    w.arrived_mask &= ~members;
    ++g_events;
}

// every live lane has arrived at the same call site: the ordinary, convergent case
bool try_complete(Wave& w) {
    if (!w.arrived_mask || w.arrived_mask != w.alive_mask) return false;
    const void* s0 = w.site[__builtin_ctzll(w.arrived_mask)];
    for (int l = 0; l < 64; ++l)
        if (((w.arrived_mask >> l) & 1u) && w.site[l] != s0) return false;
    release(w, w.arrived_mask);
    return true;
}

// Nothing can run: collectives inside divergent code.  On the GPU the lanes that took the branch execute the
// instruction among themselves (EXEC masks the others out); here those are the lanes waiting at one call site while the
// rest of their wave waits somewhere else (another call site, the workgroup barrier).  Release the group of the lowest
// waiting lane of every such wave.
bool resolve_divergent() {
    bool any = false;
    for (Wave& w : g_waves) {
        if (!w.arrived_mask) continue;
        const void* s0 = w.site[__builtin_ctzll(w.arrived_mask)];
        uint64_t members = 0;
        for (int l = 0; l < 64; ++l)
            if (((w.arrived_mask >> l) & 1u) && w.site[l] == s0) members |= 1ull << l;
        // HIPEMU_STRICT=1: a kernel whose collectives all sit in wave-uniform code never gets here with EVERY live lane
        // waiting in a collective -- unless the host compiler cloned a call site behind a per-lane branch, after which
        // the lanes no longer meet and the emulation is no longer the GPU's lock step (README: rules)
        if (w.arrived_mask == w.alive_mask) {
            static const bool strict = getenv("HIPEMU_STRICT") != nullptr;
            if (strict) {
                std::fprintf(stderr, "hip_emu: the lanes of a wave wait in collectives at different call sites (%p ...)\n", s0);
                std::abort();
            }
        }
        release(w, members);
        any = true;
    }
    return any;
}

void fiber_exit() {
    Fiber* f = g_cur;
    f->done = true;
    Wave& w = g_waves[f->wave];
    w.alive_mask &= ~(1ull << f->lane);
    --w.alive;
    --g_block_alive;
    ++g_events;
    // lanes that left no longer take part: a collective / barrier the others wait in may be complete now
    try_complete(w);
    if (g_block_alive > 0 && g_bar_arrived == g_block_alive) {
        g_bar_arrived = 0;
        ++g_bar_gen;
    }
    hipemu_switch(&f->sp, g_sched_sp);
    std::abort();  // a finished fiber is never resumed
}

void fiber_main() {
    (*g_body)();
    fiber_exit();
}

}  // namespace

const uint64_t* wave_exchange(uint64_t v, uint64_t* live) {
    Fiber* f = g_cur;
    Wave& w = g_waves[f->wave];
    const uint32_t l = f->lane;
    w.val[l] = v;
    w.site[l] = __builtin_return_address(0);
    w.arrived_mask |= 1ull << l;
    const uint32_t g = w.lane_gen[l];
    if (!try_complete(w)) wait_until_changed(&w.lane_gen[l], g);
    const int sn = w.lane_snap[l];
    if (live) *live = w.snap_live[sn];
    return w.snap_val[sn];
}

void yield_now() { yield_to_scheduler(); }

void block_barrier() {
    const uint32_t g = g_bar_gen;
    if (++g_bar_arrived == g_block_alive) {
        g_bar_arrived = 0;
        ++g_bar_gen;
        ++g_events;
    } else {
        wait_until_changed(&g_bar_gen, g);
    }
}

void* dynamic_lds() { return g_dyn_lds.data(); }

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
    if (g_cur) {
        std::fprintf(stderr, "hip_emu: nested kernel launch\n");
        std::abort();
    }
    const uint32_t nthreads = block.x * block.y * block.z;
    if (block.y != 1 || block.z != 1 || grid.y != 1 || grid.z != 1 || nthreads == 0 || nthreads > 1024) {
        std::fprintf(stderr, "hip_emu: only 1-D launches of up to 1024 threads are modelled\n");
        std::abort();
    }
    const uint32_t nwaves = (nthreads + 63) / 64;
    g_body = &body;
    g_blockDim = Idx{block.x, 1, 1};
    g_gridDim = Idx{grid.x, 1, 1};
    if (g_dyn_lds.size() < shmem + 64) g_dyn_lds.resize(shmem + 64);
    while (g_stack_pool.size() < nthreads) g_stack_pool.push_back(static_cast<char*>(std::malloc(kStackBytes)));
    g_fibers.resize(nthreads);
    g_waves.resize(nwaves);
    for (uint32_t b = 0; b < grid.x; ++b) {
        g_blockIdx = Idx{b, 0, 0};
        std::memset(g_dyn_lds.data(), 0xCD, g_dyn_lds.size());  // LDS is not zero on entry
        for (uint32_t w = 0; w < nwaves; ++w) {
            Wave& wv = g_waves[w];
            const uint32_t lanes = std::min<uint32_t>(64, nthreads - w * 64);
            wv.alive = static_cast<int>(lanes);
            wv.alive_mask = lanes == 64 ? ~0ull : ((1ull << lanes) - 1ull);
            wv.arrived_mask = 0;
            wv.next_snap = 0;
            std::memset(wv.lane_gen, 0, sizeof(wv.lane_gen));
        }
        g_bar_arrived = 0;
        g_bar_gen = 0;
        g_block_alive = static_cast<int>(nthreads);
        for (uint32_t t = 0; t < nthreads; ++t) {
            Fiber& f = g_fibers[t];
            f.tid = t;
            f.lane = t & 63;
            f.wave = t >> 6;
            f.done = false;
            f.wait_ptr = nullptr;
            f.stack = g_stack_pool[t];
            // initial frame for hipemu_switch: six callee-saved registers, then the entry point as return address,
            // placed so that the entry point sees the stack alignment of a called function
            uintptr_t top = (reinterpret_cast<uintptr_t>(f.stack) + kStackBytes) & ~static_cast<uintptr_t>(15);
            void** sp = reinterpret_cast<void**>(top - 16);
            *sp = reinterpret_cast<void*>(&fiber_main);
            sp -= 6;
            for (int k = 0; k < 6; ++k) sp[k] = nullptr;
            f.sp = sp;
        }
        uint32_t remaining = nthreads;
        while (remaining) {
            const uint64_t before = g_events;
            bool ran = false;
            for (uint32_t t = 0; t < nthreads; ++t) {
                Fiber& f = g_fibers[t];
                if (f.done) continue;
                if (f.wait_ptr && *f.wait_ptr == f.wait_val) continue;
                g_cur = &f;
                g_threadIdx = Idx{t, 0, 0};
                g_lane = f.lane;
                hipemu_switch(&g_sched_sp, f.sp);
                g_cur = nullptr;
                ran = true;
                if (f.done) --remaining;
            }
            if (!ran || (g_events == before && remaining)) {
                // nobody could run, or a whole round went by without a collective completing or a thread finishing
                bool any_runnable = false;
                for (uint32_t t = 0; t < nthreads; ++t)
                    if (!g_fibers[t].done && !(g_fibers[t].wait_ptr && *g_fibers[t].wait_ptr == g_fibers[t].wait_val))
                        any_runnable = true;
                if (!any_runnable && resolve_divergent()) continue;
                if (!any_runnable) {
                    std::fprintf(stderr, "hip_emu: deadlock in workgroup %u: a wave collective or __syncthreads is not "
                                         "reached by every live lane\n", b);
                    for (uint32_t w = 0; w < nwaves; ++w)
                        std::fprintf(stderr, "  wave %u: %d alive, waiting lanes %016llx\n", w, g_waves[w].alive,
                                     static_cast<unsigned long long>(g_waves[w].arrived_mask));
                    std::fprintf(stderr, "  barrier: %d of %d arrived\n", g_bar_arrived, g_block_alive);
                    std::abort();
                }
            }
        }
    }
    g_body = nullptr;
}

}  // namespace hipemu

// ---------------------------------------------------------------------------------------------------- runtime stand-ins
struct hipemu_stream {
    int dummy;
};
struct hipemu_event {
    int dummy;
};
static hipemu_stream g_the_stream;
static hipemu_event g_the_event;

const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "hip_emu error"; }
hipError_t hipGetDeviceCount(int* n) {
    *n = 1;
    return hipSuccess;
}
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevice(int* d) {
    *d = 0;
    return hipSuccess;
}
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned flags, int) { return hipStreamCreateWithFlags(s, flags); }
hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) {
    if (least) *least = 1;
    if (greatest) *greatest = -1;
    return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) {
    *s = &g_the_stream;
    return hipSuccess;
}
hipError_t hipStreamCreate(hipStream_t* s) {
    *s = &g_the_stream;
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipMalloc(void** p, size_t n) {
    // device memory is not zero either: poison it so that reads of never-written words show up in the results
    void* q = nullptr;
    if (posix_memalign(&q, 256, n ? n : 1) != 0) return hipErrorOutOfMemory;
    std::memset(q, 0xCD, n);
    *p = q;
    return hipSuccess;
}
hipError_t hipFree(void* p) {
    std::free(p);
    return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { return hipMalloc(p, n); }
hipError_t hipHostFree(void* p) { return hipFree(p); }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) {
    std::memmove(d, s, n);
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) {
    std::memmove(d, s, n);
    return hipSuccess;
}
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) {
    std::memset(d, v, n);
    return hipSuccess;
}
hipError_t hipMemset(void* d, int v, size_t n) {
    std::memset(d, v, n);
    return hipSuccess;
}
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) {
    *e = &g_the_event;
    return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) {
    *ms = 0.001f;
    return hipSuccess;
}
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
