// Stable least-significant-digit radix sort of the compacted record stream by read identity (gfx950, wave64).
//
// Only used when the caller declares record_order = SLIMM_ORDER_ANY: the reference groups records of a read through
// a string-keyed hash map (src/slimm.hpp:204-211) and therefore accepts any record order.  A STABLE sort by identity
// makes every read a contiguous run while keeping file order inside the run, which is what the first-bin rule (Q1)
// needs; after it the stream goes through the same kernels as name-grouped input.
//
// 8 passes of 8 bits.  Per pass: per-tile digit histograms -> one exclusive scan in digit-major order -> scatter with
// stable in-tile ranks (per-wave match masks from 8 ballots, LDS prefix across waves and chunks).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace slimm {

constexpr int kSBlock = 256;
constexpr int kSItems = 8;
constexpr int kSTile = kSBlock * kSItems;
constexpr int kSWaves = kSBlock / 64;

__device__ __forceinline__ uint32_t rs_mask_rank(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32),
                                     __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
}

__global__ __launch_bounds__(kSBlock) void k_rs_hist(const uint64_t* __restrict__ ident,
                                                     const uint32_t* __restrict__ counters, uint32_t shift,
                                                     uint32_t ntiles, uint32_t* __restrict__ hist) {
    __shared__ uint32_t s_h[256];
    const uint32_t V = counters[CNT_V];
    const uint32_t base = blockIdx.x * kSTile;
    s_h[threadIdx.x] = 0;
    __syncthreads();
    if (base < V) {
#pragma unroll
        for (int k = 0; k < kSItems; ++k) {
            uint32_t i = base + k * kSBlock + threadIdx.x;
            if (i < V) atomicAdd(&s_h[static_cast<uint32_t>(ident[i] >> shift) & 255u], 1u);
        }
    }
    __syncthreads();
    hist[threadIdx.x * ntiles + blockIdx.x] = s_h[threadIdx.x];
}

// exclusive scan of a uint32 array by one workgroup
__global__ __launch_bounds__(1024) void k_rs_scan(uint32_t* __restrict__ a, uint32_t n) {
    __shared__ uint32_t s_part[1024];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (n + 1023) / 1024;
    const uint32_t lo = min(tid * per, n), hi = min(lo + per, n);
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; ++i) sum += a[i];
    s_part[tid] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        uint32_t add = (tid >= off) ? s_part[tid - off] : 0u;
        __syncthreads();
        s_part[tid] += add;
        __syncthreads();
    }
    uint32_t run = s_part[tid] - sum;
    for (uint32_t i = lo; i < hi; ++i) {
        uint32_t v = a[i];
        a[i] = run;
        run += v;
    }
}

__global__ __launch_bounds__(kSBlock) void k_rs_scatter(const uint64_t* __restrict__ ident_in,
                                                        const uint32_t* __restrict__ ref_in,
                                                        const uint32_t* __restrict__ gbin_in,
                                                        const uint32_t* __restrict__ counters, uint32_t shift,
                                                        uint32_t ntiles, const uint32_t* __restrict__ hist,
                                                        uint64_t* __restrict__ ident_out, uint32_t* __restrict__ ref_out,
                                                        uint32_t* __restrict__ gbin_out) {
    __shared__ uint32_t s_goff[256];             // global offset of (digit, this tile) + items of earlier chunks
    __shared__ uint32_t s_wcnt[kSWaves][256];    // per wave digit counts of the current chunk
    const uint32_t V = counters[CNT_V];
    const uint32_t base = blockIdx.x * kSTile;
    if (base >= V) return;
    const uint32_t tid = threadIdx.x, wave = tid >> 6;
    s_goff[tid] = hist[tid * ntiles + blockIdx.x];
#pragma unroll
    for (int w = 0; w < kSWaves; ++w) s_wcnt[w][tid] = 0;
    __syncthreads();
    for (int k = 0; k < kSItems; ++k) {
        uint32_t i = base + k * kSBlock + tid;
        bool live = i < V;
        uint64_t id = live ? ident_in[i] : 0;
        uint32_t d = static_cast<uint32_t>(id >> shift) & 255u;
        // lanes of this wave holding the same digit (live lanes only)
        uint64_t peers = __ballot(live);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            uint64_t bm = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? bm : ~bm;
        }
        uint32_t rank = rs_mask_rank(peers);
        if (live && rank == 0) s_wcnt[wave][d] = __popcll(peers);
        __syncthreads();
        if (live) {
            uint32_t o = s_goff[d] + rank;
#pragma unroll
            for (int w = 0; w < kSWaves; ++w)
                if (w < static_cast<int>(wave)) o += s_wcnt[w][d];
            ident_out[o] = id;
            ref_out[o] = ref_in[i];
            gbin_out[o] = gbin_in[i];
        }
        __syncthreads();
        uint32_t add = 0;
#pragma unroll
        for (int w = 0; w < kSWaves; ++w) {
            add += s_wcnt[w][tid];
            s_wcnt[w][tid] = 0;
        }
        s_goff[tid] += add;
        __syncthreads();
    }
}

void launch_sort_by_ident(hipStream_t st, uint32_t n_upper, const uint32_t* counters, uint64_t* ident, uint32_t* cref,
                          uint32_t* cgbin, uint64_t* ident_tmp, uint32_t* cref_tmp, uint32_t* cgbin_tmp, uint32_t* hist) {
    const uint32_t nt = (n_upper + kSTile - 1) / kSTile;
    if (nt == 0) return;
    uint64_t* ki = ident;
    uint32_t* ri = cref;
    uint32_t* gi = cgbin;
    uint64_t* ko = ident_tmp;
    uint32_t* ro = cref_tmp;
    uint32_t* go = cgbin_tmp;
    for (uint32_t pass = 0; pass < 8; ++pass) {
        const uint32_t shift = pass * 8;
        hipLaunchKernelGGL(k_rs_hist, dim3(nt), dim3(kSBlock), 0, st, ki, counters, shift, nt, hist);
        hipLaunchKernelGGL(k_rs_scan, dim3(1), dim3(1024), 0, st, hist, 256u * nt);
        hipLaunchKernelGGL(k_rs_scatter, dim3(nt), dim3(kSBlock), 0, st, ki, ri, gi, counters, shift, nt, hist, ko, ro, go);
        uint64_t* tk = ki; ki = ko; ko = tk;
        uint32_t* tr = ri; ri = ro; ro = tr;
        uint32_t* tg = gi; gi = go; go = tg;
    }
    // 8 passes: the result is back in (ident, cref, cgbin)
}

}  // namespace slimm
