// LDS-privatised coverage histograms (gfx950, wave64): cov / uniq_cov without a single global atomic on a bin.
//
// Why: a device-scope atomic on a random 4-byte bin is a separate memory-side request; MI355X retires ~20 G of them per
// second (measured: k_hist, 8.2 M + 1.1 M atomics = 408 us at config 2 -- 4 % of the HBM roofline).  The targets carry
// no locality (reads land anywhere on any genome), so they are first bucketed by bin TILE (8192 consecutive bins), then
// one workgroup per tile accumulates its bucket in LDS and writes the finished tile with coalesced 16-byte stores.
// The tile write-back also replaces the zero-fill of cov / uniq_cov.
//
//   k_tile_count    persistent grid; per-workgroup LDS histogram of tile ids over its slice of the targets, merged into
//                   tile_count[] with one (contiguous, non-returning) global atomic per non-empty tile
//   k_tile_scan     exclusive scan tile_count -> tile_base (one workgroup; <= 36 K tiles)
//   k_tile_scatter  same slices; reserves a range per (workgroup, tile) with one returning atomic on tile_cursor[],
//                   then writes each target as a 16-bit word (13-bit bin-in-tile | unique bit) into its bucket
//   k_tile_hist     one workgroup per tile: LDS cov[8192] + uniq_cov[8192], bucket in, finished tile out
//
// Reference semantics: src/slimm.hpp:219-257 (cov[bin]++ per target; uniq_cov[bin]++ when the read has one target).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace slimm {

constexpr int kTBlock = 256;
constexpr uint32_t kTileMask = kTileBins - 1;

// slice of the targets owned by workgroup b of g: [lo, hi), 256-aligned so loads stay coalesced
__device__ __forceinline__ void slice_of(uint32_t P, uint32_t b, uint32_t g, uint32_t& lo, uint32_t& hi) {
    uint32_t chunk = (P + g - 1) / g;
    chunk = (chunk + 255u) & ~255u;
    uint64_t l = static_cast<uint64_t>(b) * chunk;
    uint64_t h = l + chunk;
    lo = l < P ? static_cast<uint32_t>(l) : P;
    hi = h < P ? static_cast<uint32_t>(h) : P;
}

__global__ __launch_bounds__(kTBlock) void k_tile_count(const uint32_t* __restrict__ tgt_gbin,
                                                        const uint32_t* __restrict__ counters, uint32_t ntiles,
                                                        uint32_t* __restrict__ tile_count) {
    extern __shared__ uint32_t s_hist[];
    const uint32_t P = counters[CNT_P];
    for (uint32_t i = threadIdx.x; i < ntiles; i += kTBlock) s_hist[i] = 0;
    __syncthreads();
    uint32_t lo, hi;
    slice_of(P, blockIdx.x, gridDim.x, lo, hi);
    for (uint32_t t = lo + threadIdx.x; t < hi; t += kTBlock) atomicAdd(&s_hist[tgt_gbin[t] >> kTileShift], 1u);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < ntiles; i += kTBlock) {
        uint32_t h = s_hist[i];
        if (h) atomicAdd(&tile_count[i], h);
    }
}

// exclusive scan of tile_count[0..ntiles) -> tile_base[0..ntiles], tile_base[ntiles] = total; zeroes tile_cursor
__global__ __launch_bounds__(1024) void k_tile_scan(const uint32_t* __restrict__ tile_count, uint32_t ntiles,
                                                    uint32_t* __restrict__ tile_base, uint32_t* __restrict__ tile_cursor) {
    __shared__ uint32_t s_part[1024];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (ntiles + 1023) / 1024;
    const uint32_t lo = min(tid * per, ntiles), hi = min(lo + per, ntiles);
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; ++i) sum += tile_count[i];
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
        uint32_t v = tile_count[i];
        tile_base[i] = run;
        tile_cursor[i] = 0;
        run += v;
    }
    if (tid == 1023) tile_base[ntiles] = s_part[1023];
}

__global__ __launch_bounds__(kTBlock) void k_tile_scatter(const uint32_t* __restrict__ tgt_ref,
                                                          const uint32_t* __restrict__ tgt_gbin,
                                                          const uint32_t* __restrict__ counters, uint32_t ntiles,
                                                          const uint32_t* __restrict__ tile_base,
                                                          uint32_t* __restrict__ tile_cursor,
                                                          uint16_t* __restrict__ bucket) {
    extern __shared__ uint32_t s_hist[];
    const uint32_t P = counters[CNT_P];
    for (uint32_t i = threadIdx.x; i < ntiles; i += kTBlock) s_hist[i] = 0;
    __syncthreads();
    uint32_t lo, hi;
    slice_of(P, blockIdx.x, gridDim.x, lo, hi);
    for (uint32_t t = lo + threadIdx.x; t < hi; t += kTBlock) atomicAdd(&s_hist[tgt_gbin[t] >> kTileShift], 1u);
    __syncthreads();
    // reserve [base, base + h) of each non-empty tile's bucket for this workgroup; s_hist becomes the write cursor
    for (uint32_t i = threadIdx.x; i < ntiles; i += kTBlock) {
        uint32_t h = s_hist[i];
        if (h) s_hist[i] = tile_base[i] + atomicAdd(&tile_cursor[i], h);
    }
    __syncthreads();
    for (uint32_t t = lo + threadIdx.x; t < hi; t += kTBlock) {
        uint32_t g = tgt_gbin[t];
        bool start = tgt_ref[t] >> 31;
        bool next_start = (t + 1 == P) || (tgt_ref[t + 1] >> 31);
        uint32_t pos = atomicAdd(&s_hist[g >> kTileShift], 1u);
        bucket[pos] = static_cast<uint16_t>((g & kTileMask) | ((start && next_start) ? kTileBins : 0u));
    }
}

__global__ __launch_bounds__(512) void k_tile_hist(const uint16_t* __restrict__ bucket,
                                                   const uint32_t* __restrict__ tile_base, uint32_t* __restrict__ cov,
                                                   uint32_t* __restrict__ ucov) {
    __shared__ uint32_t s_cov[kTileBins];
    __shared__ uint32_t s_ucov[kTileBins];
    const uint32_t tile = blockIdx.x;
    for (uint32_t i = threadIdx.x; i < kTileBins; i += 512) {
        s_cov[i] = 0;
        s_ucov[i] = 0;
    }
    __syncthreads();
    const uint32_t lo = tile_base[tile], hi = tile_base[tile + 1];
    for (uint32_t e = lo + threadIdx.x; e < hi; e += 512) {
        uint32_t v = bucket[e];
        atomicAdd(&s_cov[v & kTileMask], 1u);
        if (v & kTileBins) atomicAdd(&s_ucov[v & kTileMask], 1u);
    }
    __syncthreads();
    uint4* oc = reinterpret_cast<uint4*>(cov + static_cast<size_t>(tile) * kTileBins);
    uint4* ou = reinterpret_cast<uint4*>(ucov + static_cast<size_t>(tile) * kTileBins);
    const uint4* sc = reinterpret_cast<const uint4*>(s_cov);
    const uint4* su = reinterpret_cast<const uint4*>(s_ucov);
    for (uint32_t i = threadIdx.x; i < kTileBins / 4; i += 512) {
        oc[i] = sc[i];
        ou[i] = su[i];
    }
}

int tile_hist_setup(uint32_t ntiles) {
    // dynamic LDS above 64 KiB has to be opted into
    size_t bytes = static_cast<size_t>(ntiles) * 4;
    if (bytes > kTileLdsMax) return -1;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile_count), hipFuncAttributeMaxDynamicSharedMemorySize,
                            static_cast<int>(bytes)) != hipSuccess)
        return -1;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile_scatter), hipFuncAttributeMaxDynamicSharedMemorySize,
                            static_cast<int>(bytes)) != hipSuccess)
        return -1;
    return 0;
}

void launch_tile_count(hipStream_t st, uint32_t grid, uint32_t ntiles, const uint32_t* tgt_gbin, const uint32_t* counters,
                       uint32_t* tile_count) {
    (void)hipMemsetAsync(tile_count, 0, static_cast<size_t>(ntiles) * 4, st);
    hipLaunchKernelGGL(k_tile_count, dim3(grid), dim3(kTBlock), static_cast<size_t>(ntiles) * 4, st, tgt_gbin, counters,
                       ntiles, tile_count);
}

void launch_tile_scan(hipStream_t st, uint32_t ntiles, const uint32_t* tile_count, uint32_t* tile_base,
                      uint32_t* tile_cursor) {
    hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, st, tile_count, ntiles, tile_base, tile_cursor);
}

void launch_tile_scatter(hipStream_t st, uint32_t grid, uint32_t ntiles, const uint32_t* tgt_ref, const uint32_t* tgt_gbin,
                         const uint32_t* counters, const uint32_t* tile_base, uint32_t* tile_cursor, uint16_t* bucket) {
    hipLaunchKernelGGL(k_tile_scatter, dim3(grid), dim3(kTBlock), static_cast<size_t>(ntiles) * 4, st, tgt_ref, tgt_gbin,
                       counters, ntiles, tile_base, tile_cursor, bucket);
}

void launch_tile_hist(hipStream_t st, uint32_t ntiles, const uint16_t* bucket, const uint32_t* tile_base, uint32_t* cov,
                      uint32_t* ucov) {
    hipLaunchKernelGGL(k_tile_hist, dim3(ntiles), dim3(512), 0, st, bucket, tile_base, cov, ucov);
}

}  // namespace slimm
