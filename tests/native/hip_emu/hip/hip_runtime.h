// TEST INFRASTRUCTURE ONLY -- a host stand-in for <hip/hip_runtime.h> so that the product's .hip sources compile with
// g++ and run on the CPU of the build container (which has no GPU): one fiber per GPU thread, scheduled in lock step
// at every wave collective (ballot, DPP, readlane, shuffles) and workgroup barrier.  It exists to find logic errors in
// the kernels (indexing, lane masks, control flow) before a run on a real MI355X is spent on them; it models neither
// timing nor the memory system.  Nothing under slimm_amd/ includes this file: the test build puts this directory in
// front of the include path (tests/native/Makefile), the product build uses ROCm's header.
//
// Discipline the emulation relies on (and checks where it can): wave collectives are reached by every live lane of a
// wave in the same order (wave-uniform control flow around them); __syncthreads by every live thread of a workgroup.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <tuple>
#include <utility>

#define SLIMM_HIP_EMU 1
#define __HIPCC__ 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __shared__ static
#define __launch_bounds__(...)
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __HIP_MEMORY_SCOPE_WORKGROUP 0
#define __HIP_MEMORY_SCOPE_SYSTEM 0

// ------------------------------------------------------------------------------------------------ vector types
struct uint2 {
    uint32_t x, y;
};
struct uint4 {
    uint32_t x, y, z, w;
};
struct int2 {
    int32_t x, y;
};
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
struct dim3 {
    uint32_t x, y, z;
    dim3(uint32_t x_ = 1, uint32_t y_ = 1, uint32_t z_ = 1) : x(x_), y(y_), z(z_) {}
};

// ------------------------------------------------------------------------------------------------ runtime API subset
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNoDevice = 100 };
typedef struct hipemu_stream* hipStream_t;
typedef struct hipemu_event* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipStreamNonBlocking = 1, hipHostMallocDefault = 0, hipEventDisableTiming = 2 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };

const char* hipGetErrorString(hipError_t e);
hipError_t hipGetDeviceCount(int* n);
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int* d);
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 };
static inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) {
    *v = 3;   // (a few persistent workgroups: grid-stride loops get their second trip)
    return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned flags, int priority);
hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest);
hipError_t hipStreamCreate(hipStream_t* s);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamQuery(hipStream_t s);
hipError_t hipDeviceSynchronize();
static inline hipError_t hipDeviceReset() { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipMalloc(void** p, size_t n);
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t n, unsigned flags);
hipError_t hipHostFree(void* p);
static inline hipError_t hipHostRegister(void*, size_t, unsigned) { return hipSuccess; }
static inline hipError_t hipMemGetInfo(size_t* fr, size_t* tot) {
    *fr = *tot = size_t(1) << 40;
    return hipSuccess;
}
static inline hipError_t hipHostUnregister(void*) { return hipSuccess; }
#define hipHostRegisterDefault 0
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t st);
static inline hipError_t hipMemcpyPeerAsync(void* dst, int, const void* src, int, size_t n, hipStream_t s) {
    return hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, s);
}
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st);
hipError_t hipMemset(void* d, int v, size_t n);
hipError_t hipGetLastError();
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventQuery(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipFuncSetAttribute(const void* f, hipFuncAttribute a, int v);
template <typename T>
static inline hipError_t hipMalloc(T** p, size_t n) {
    return hipMalloc(reinterpret_cast<void**>(p), n);
}
template <typename T>
static inline hipError_t hipHostMalloc(T** p, size_t n, unsigned flags = 0) {
    return hipHostMalloc(reinterpret_cast<void**>(p), n, flags);
}

// ------------------------------------------------------------------------------------------------ fibers
namespace hipemu {
struct Idx {
    uint32_t x, y, z;
};
extern Idx g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;
extern uint32_t g_lane;  // lane of the running fiber inside its wave
// every live lane of the caller's wave deposits v; returns the 64 deposited values (valid until the wave's next
// exchange) and, through *live, the mask of lanes that took part
const uint64_t* wave_exchange(uint64_t v, uint64_t* live = nullptr);
void block_barrier();
void yield_now();   // the running fiber steps aside and stays runnable: what a spin-wait does between two looks (s_sleep)
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
void* dynamic_lds();
}  // namespace hipemu
#define threadIdx (::hipemu::g_threadIdx)
#define blockIdx (::hipemu::g_blockIdx)
#define blockDim (::hipemu::g_blockDim)
#define gridDim (::hipemu::g_gridDim)
constexpr int warpSize = 64;

// kernel launch: arguments are evaluated once, the kernel runs to completion before the macro returns
// (the launch with its own time stamp events: the events are ignored here)
#define hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, e0, e1, flags, ...) \
    hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__)
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)                                   \
    do {                                                                                               \
        auto hipemu_args_ = std::make_tuple(__VA_ARGS__);                                              \
        ::hipemu::launch((grid), (block), (shmem), [&]() { std::apply((kernel), hipemu_args_); });     \
    } while (0)

static inline void __syncthreads() { ::hipemu::block_barrier(); }
static inline void __threadfence() {}
static inline void __threadfence_block() {}
static inline void __builtin_amdgcn_fence(int, const char*) {}
static inline void __builtin_amdgcn_sched_barrier(int) {}
static inline void __builtin_amdgcn_s_waitcnt(int) {}

// ------------------------------------------------------------------------------------------------ wave collectives
static inline uint64_t __builtin_amdgcn_ballot_w64(bool p) {
    uint64_t live = 0;
    const uint64_t* v = ::hipemu::wave_exchange(p ? 1u : 0u, &live);
    uint64_t m = 0;
    for (int l = 0; l < 64; ++l)
        if (((live >> l) & 1u) && v[l]) m |= 1ull << l;
    return m;
}
static inline uint64_t __ballot(int p) { return __builtin_amdgcn_ballot_w64(p != 0); }
// (on the GPU a scheduling barrier and no instruction: the lanes of a wave run in lockstep; here the lanes are fibers and
// this is where they wait for each other -- e.g. before reading LDS words another lane has written)
static inline void __builtin_amdgcn_wave_barrier() {
    uint64_t live = 0;
    (void)::hipemu::wave_exchange(0u, &live);
}
static inline int __any(int p) { return __ballot(p) != 0ull; }
static inline int __all(int p) {
    uint64_t live = 0;
    const uint64_t* v = ::hipemu::wave_exchange(p ? 1u : 0u, &live);
    for (int l = 0; l < 64; ++l)
        if (((live >> l) & 1u) && !v[l]) return 0;
    return 1;
}
static inline uint32_t __builtin_amdgcn_mbcnt_lo(uint32_t mask, uint32_t add) {
    const uint32_t l = ::hipemu::g_lane;
    const uint32_t below = l >= 32 ? 0xffffffffu : ((1u << l) - 1u);
    return add + static_cast<uint32_t>(__builtin_popcount(mask & below));
}
static inline uint32_t __builtin_amdgcn_mbcnt_hi(uint32_t mask, uint32_t add) {
    const uint32_t l = ::hipemu::g_lane;
    const uint32_t below = l <= 32 ? 0u : ((1u << (l - 32)) - 1u);
    return add + static_cast<uint32_t>(__builtin_popcount(mask & below));
}
// the lane's bit of a wave-uniform mask: a v_cndmask on the GPU, no exchange between lanes (so it may sit inside
// divergent code, which the lock-step model cannot order against collectives outside the branch)
static inline bool __builtin_amdgcn_inverse_ballot_w64(uint64_t mask) { return (mask >> ::hipemu::g_lane) & 1u; }
static inline uint32_t __builtin_bitreverse32(uint32_t v) {
    uint32_t r = 0;
    for (int i = 0; i < 32; ++i) r |= ((v >> i) & 1u) << (31 - i);
    return r;
}
static inline uint64_t __builtin_bitreverse64(uint64_t v) {
    uint64_t r = 0;
    for (int i = 0; i < 64; ++i) r |= ((v >> i) & 1ull) << (63 - i);
    return r;
}
static inline uint32_t __builtin_amdgcn_readlane(uint32_t v, uint32_t lane) {
    const uint64_t* a = ::hipemu::wave_exchange(v);
    return static_cast<uint32_t>(a[lane & 63u]);
}
// v_writelane: lane `lane` of the result is `value` (wave-uniform), the other lanes keep `old`; no exchange
static inline uint32_t __builtin_amdgcn_writelane(uint32_t value, uint32_t lane, uint32_t old) {
    return ::hipemu::g_lane == (lane & 63u) ? value : old;
}
static inline uint32_t __builtin_amdgcn_readfirstlane(uint32_t v) {
    uint64_t live = 0;
    const uint64_t* a = ::hipemu::wave_exchange(v, &live);
    return static_cast<uint32_t>(a[__builtin_ctzll(live)]);
}
template <typename T>
static inline T hipemu_shfl_from(T v, int src, bool ok) {
    static_assert(sizeof(T) <= 8, "shuffle of at most 8 bytes");
    uint64_t bits = 0;
    std::memcpy(&bits, &v, sizeof(T));
    uint64_t live = 0;
    const uint64_t* a = ::hipemu::wave_exchange(bits, &live);
    if (!ok || src < 0 || src > 63 || !((live >> src) & 1u)) return v;
    T out;
    std::memcpy(&out, &a[src], sizeof(T));
    return out;
}
template <typename T>
static inline T __shfl(T v, int lane, int width = 64) {
    const int l = static_cast<int>(::hipemu::g_lane);
    return hipemu_shfl_from(v, (l & ~(width - 1)) + (lane & (width - 1)), true);
}
template <typename T>
static inline T __shfl_up(T v, unsigned d, int width = 64) {
    const int l = static_cast<int>(::hipemu::g_lane);
    const int src = l - static_cast<int>(d);
    return hipemu_shfl_from(v, src, src >= (l & ~(width - 1)));
}
template <typename T>
static inline T __shfl_down(T v, unsigned d, int width = 64) {
    const int l = static_cast<int>(::hipemu::g_lane);
    const int src = l + static_cast<int>(d);
    return hipemu_shfl_from(v, src, src < (l & ~(width - 1)) + width);
}
template <typename T>
static inline T __shfl_xor(T v, int m, int width = 64) {
    (void)width;
    return hipemu_shfl_from(v, static_cast<int>(::hipemu::g_lane) ^ m, true);
}
// DPP move: dst = (source lane valid and enabled) ? src[source lane] : (bound_ctrl ? 0 : old)
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    uint64_t live = 0;
    const uint64_t* a = ::hipemu::wave_exchange(static_cast<uint32_t>(src), &live);
    const int l = static_cast<int>(::hipemu::g_lane);
    const int row = l >> 4, in_row = l & 15;
    if (!((row_mask >> row) & 1) || !((bank_mask >> (in_row >> 2)) & 1)) return old;
    int s = -1;
    if (ctrl >= 0x000 && ctrl <= 0x0ff) {  // quad_perm
        s = (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);
    } else if (ctrl >= 0x101 && ctrl <= 0x10f) {  // row_shl:n
        const int n = ctrl & 15;
        s = in_row + n <= 15 ? l + n : -1;
    } else if (ctrl >= 0x111 && ctrl <= 0x11f) {  // row_shr:n
        const int n = ctrl & 15;
        s = in_row - n >= 0 ? l - n : -1;
    } else if (ctrl >= 0x121 && ctrl <= 0x12f) {  // row_ror:n
        const int n = ctrl & 15;
        s = (l & ~15) | ((in_row - n) & 15);
    } else if (ctrl == 0x130) {  // wave_shl:1
        s = l + 1 <= 63 ? l + 1 : -1;
    } else if (ctrl == 0x134) {  // wave_rol:1
        s = (l + 1) & 63;
    } else if (ctrl == 0x138) {  // wave_shr:1
        s = l - 1;
    } else if (ctrl == 0x13c) {  // wave_ror:1
        s = (l + 63) & 63;
    } else if (ctrl == 0x140) {  // row_mirror
        s = (l & ~15) | (15 - in_row);
    } else if (ctrl == 0x141) {  // row_half_mirror
        s = (l & ~7) | (7 - (l & 7));
    } else if (ctrl == 0x142) {  // row_bcast:15
        s = row > 0 ? 16 * row - 1 : -1;
    } else if (ctrl == 0x143) {  // row_bcast:31
        s = l >= 32 ? 31 : -1;
    } else {
        std::fprintf(stderr, "hip_emu: DPP control 0x%x not modelled\n", ctrl);
        std::abort();
    }
    if (s < 0 || !((live >> s) & 1u)) return bound_ctrl ? 0 : old;
    return static_cast<int>(static_cast<uint32_t>(a[s]));
}
static inline void __builtin_amdgcn_s_sleep(int) { ::hipemu::yield_now(); }
static inline uint32_t __builtin_amdgcn_ds_bpermute(int byte_addr, uint32_t v) {
    uint64_t live = 0;
    const uint64_t* a = ::hipemu::wave_exchange(v, &live);
    const int s = (byte_addr >> 2) & 63;
    return ((live >> s) & 1u) ? static_cast<uint32_t>(a[s]) : 0u;
}

// ------------------------------------------------------------------------------------------------ arithmetic, atomics
// v_alignbit_b32: the low 32 bits of {hi, lo} >> (shift & 31)
static inline uint32_t __builtin_amdgcn_alignbit(uint32_t hi, uint32_t lo, uint32_t shift) {
    return static_cast<uint32_t>(((static_cast<uint64_t>(hi) << 32) | lo) >> (shift & 31u));
}
static inline uint32_t __umulhi(uint32_t a, uint32_t b) { return static_cast<uint32_t>((static_cast<uint64_t>(a) * b) >> 32); }
static inline int __popc(uint32_t v) { return __builtin_popcount(v); }
static inline int __popcll(uint64_t v) { return __builtin_popcountll(v); }
static inline int __clz(uint32_t v) { return v ? __builtin_clz(v) : 32; }
static inline int __clzll(uint64_t v) { return v ? __builtin_clzll(v) : 64; }
static inline int __ffs(uint32_t v) { return __builtin_ffs(static_cast<int>(v)); }
static inline int __ffsll(uint64_t v) { return __builtin_ffsll(static_cast<long long>(v)); }
using std::max;
using std::min;
static inline uint32_t min(uint32_t a, int b) { return a < static_cast<uint32_t>(b) ? a : static_cast<uint32_t>(b); }
static inline uint32_t min(int a, uint32_t b) { return static_cast<uint32_t>(a) < b ? static_cast<uint32_t>(a) : b; }
static inline uint32_t max(uint32_t a, int b) { return a > static_cast<uint32_t>(b) ? a : static_cast<uint32_t>(b); }
static inline uint32_t max(int a, uint32_t b) { return static_cast<uint32_t>(a) > b ? static_cast<uint32_t>(a) : b; }

#define HIPEMU_ATOMIC(name, expr)                                   \
    template <typename T, typename V>                               \
    static inline T name(T* p, V v_) {                              \
        const T o = *p;                                             \
        const T v = static_cast<T>(v_);                             \
        *p = (expr);                                                \
        return o;                                                   \
    }
HIPEMU_ATOMIC(atomicAdd, o + v)
HIPEMU_ATOMIC(atomicSub, o - v)
HIPEMU_ATOMIC(atomicOr, o | v)
HIPEMU_ATOMIC(atomicAnd, o& v)
HIPEMU_ATOMIC(atomicXor, o ^ v)
HIPEMU_ATOMIC(atomicMin, v < o ? v : o)
HIPEMU_ATOMIC(atomicMax, v > o ? v : o)
HIPEMU_ATOMIC(atomicExch, v)
#undef HIPEMU_ATOMIC
template <typename T, typename C, typename V>
static inline T atomicCAS(T* p, C cmp, V v) {
    const T o = *p;
    if (o == static_cast<T>(cmp)) *p = static_cast<T>(v);
    return o;
}
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))
#define __hip_atomic_fetch_add(p, v, order, scope) atomicAdd((p), (v))
#define HIP_DYNAMIC_SHARED(type, name) type* name = static_cast<type*>(::hipemu::dynamic_lds());
