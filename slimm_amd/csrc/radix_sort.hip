// Stable least-significant-digit radix sort of the compacted record stream by read identity (gfx950, wave64).
//
// Only used when the caller declares record_order = SLIMM_ORDER_ANY: the reference groups records of a read through
// a string-keyed hash map (src/slimm.hpp:204-211) and therefore accepts any record order.  A STABLE sort by identity
// makes every read a contiguous run while keeping file order inside the run, which is what the first-bin rule (Q1)
// needs; after it the stream goes through the same kernels as name-grouped input.
//
// 8 passes of 8 bits over 16-byte records (ident u64 | ref u32 | gbin u32).  Per pass:
//   k_rs_hist     per-tile digit histogram (LDS), stored digit-major: hist[digit][tile]
//   k_rs_rowscan  one workgroup per digit: exclusive scan of its row over the tiles (coalesced, running carry) and the
//                 digit's total
//   k_rs_scatter  per tile: stable local ranks (per-wave match masks from 8 ballots, LDS prefix across waves and
//                 chunks), records reordered by digit in LDS, then written out so that consecutive lanes store to
//                 consecutive addresses of each digit's run
// (The first version scanned the 256 x tiles matrix with ONE workgroup and scattered straight from registers:
//  22.9 ms for 10 M records; this one is bounded by 40 bytes of traffic per record and pass.)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace slimm {

constexpr int kSBlock = 256;
constexpr int kSItems = 8;
constexpr int kSTile = kSBlock * kSItems;  // 2048 records per workgroup
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
        uint64_t k[kSItems];
#pragma unroll
        for (int u = 0; u < kSItems; ++u) {
            const uint32_t i = base + u * kSBlock + threadIdx.x;
            k[u] = (i < V) ? ident[i] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < kSItems; ++u) {
            const uint32_t i = base + u * kSBlock + threadIdx.x;
            if (i < V) atomicAdd(&s_h[static_cast<uint32_t>(k[u] >> shift) & 255u], 1u);
        }
    }
    __syncthreads();
    hist[threadIdx.x * ntiles + blockIdx.x] = s_h[threadIdx.x];
}

// grid = 256 (one workgroup per digit): exclusive scan of hist[digit][0..ntiles) in place, total -> totals[digit]
__global__ __launch_bounds__(1024) void k_rs_rowscan(uint32_t* __restrict__ hist, uint32_t ntiles,
                                                     uint32_t* __restrict__ totals) {
    __shared__ uint32_t s_wave[16];
    uint32_t* row = hist + static_cast<size_t>(blockIdx.x) * ntiles;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t carry = 0;
    for (uint32_t c0 = 0; c0 < ntiles; c0 += 1024) {
        const uint32_t i = c0 + tid;
        const uint32_t v = (i < ntiles) ? row[i] : 0u;
        uint32_t inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t a = __shfl_up(inc, o, 64);
            if (lane >= static_cast<uint32_t>(o)) inc += a;
        }
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        uint32_t before = carry, total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const uint32_t t = s_wave[w];
            if (w < static_cast<int>(wave)) before += t;
            total += t;
        }
        if (i < ntiles) row[i] = before + inc - v;
        carry += total;
        __syncthreads();
    }
    if (tid == 0) totals[blockIdx.x] = carry;
}

// kChk: a third payload word travels along (the check word of slimm_push_records_checked)
template <bool kChk>
__global__ __launch_bounds__(kSBlock) void k_rs_scatter(const uint64_t* __restrict__ ident_in,
                                                        const uint32_t* __restrict__ ref_in,
                                                        const uint32_t* __restrict__ gbin_in,
                                                        const uint32_t* __restrict__ chk_in,
                                                        const uint32_t* __restrict__ counters, uint32_t shift,
                                                        uint32_t ntiles, const uint32_t* __restrict__ hist,
                                                        const uint32_t* __restrict__ totals,
                                                        uint64_t* __restrict__ ident_out, uint32_t* __restrict__ ref_out,
                                                        uint32_t* __restrict__ gbin_out, uint32_t* __restrict__ chk_out) {
    __shared__ uint64_t s_key[kSTile];           // records reordered by digit
    __shared__ uint2 s_pay[kSTile];
    __shared__ uint32_t s_chk[kChk ? kSTile : 1];
    __shared__ uint32_t s_goff[256];             // global position of this tile's first record of each digit
    __shared__ uint32_t s_lstart[256];           // local position of the digit's first record / running cursor
    __shared__ uint32_t s_cnt[256];              // records of each digit in this tile
    __shared__ uint32_t s_wcnt[kSWaves][256];    // per wave digit counts of the current chunk
    __shared__ uint32_t s_wscan[kSWaves];
    const uint32_t V = counters[CNT_V];
    const uint32_t base = blockIdx.x * kSTile;
    if (base >= V) return;
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const uint32_t n_here = min(static_cast<uint32_t>(kSTile), V - base);

    // all loads of the tile first
    uint64_t k[kSItems];
    uint2 pay[kSItems];
    uint32_t chk[kSItems];
#pragma unroll
    for (int u = 0; u < kSItems; ++u) {
        const uint32_t i = base + u * kSBlock + tid;
        const bool live = i < V;
        k[u] = live ? ident_in[i] : 0ull;
        pay[u] = live ? make_uint2(ref_in[i], gbin_in[i]) : make_uint2(0u, 0u);
        chk[u] = (kChk && live) ? chk_in[i] : 0u;
    }
    s_cnt[tid] = 0;
#pragma unroll
    for (int w = 0; w < kSWaves; ++w) s_wcnt[w][tid] = 0;
    // digit bases: exclusive scan of the 256 digit totals (one value per thread)
    {
        const uint32_t t = totals[tid];
        uint32_t inc = t;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t a = __shfl_up(inc, o, 64);
            if (lane >= static_cast<uint32_t>(o)) inc += a;
        }
        if (lane == 63) s_wscan[wave] = inc;
        __syncthreads();
        uint32_t before = 0;
        for (uint32_t w = 0; w < wave; ++w) before += s_wscan[w];
        s_goff[tid] = before + inc - t + hist[tid * ntiles + blockIdx.x];
    }
    __syncthreads();
    // digit counts of the tile
#pragma unroll
    for (int u = 0; u < kSItems; ++u) {
        const uint32_t i = base + u * kSBlock + tid;
        if (i < V) atomicAdd(&s_cnt[static_cast<uint32_t>(k[u] >> shift) & 255u], 1u);
    }
    __syncthreads();
    {   // local start of every digit: exclusive scan of s_cnt
        const uint32_t c = s_cnt[tid];
        uint32_t inc = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t a = __shfl_up(inc, o, 64);
            if (lane >= static_cast<uint32_t>(o)) inc += a;
        }
        if (lane == 63) s_wscan[wave] = inc;
        __syncthreads();
        uint32_t before = 0;
        for (uint32_t w = 0; w < wave; ++w) before += s_wscan[w];
        s_lstart[tid] = before + inc - c;
    }
    __syncthreads();
    // stable local placement, chunk by chunk (chunk u = records base + u*256 .. in file order)
#pragma unroll
    for (int u = 0; u < kSItems; ++u) {
        const uint32_t i = base + u * kSBlock + tid;
        const bool live = i < V;
        const uint32_t d = static_cast<uint32_t>(k[u] >> shift) & 255u;
        uint64_t peers = __ballot(live);  // lanes of this wave holding the same digit
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint64_t bm = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? bm : ~bm;
        }
        const uint32_t rank = rs_mask_rank(peers);
        if (live && rank == 0) s_wcnt[wave][d] = __popcll(peers);
        __syncthreads();
        if (live) {
            uint32_t o = s_lstart[d] + rank;
#pragma unroll
            for (int w = 0; w < kSWaves; ++w)
                if (w < static_cast<int>(wave)) o += s_wcnt[w][d];
            s_key[o] = k[u];
            s_pay[o] = pay[u];
            if (kChk) s_chk[o] = chk[u];
        }
        __syncthreads();
        uint32_t add = 0;
#pragma unroll
        for (int w = 0; w < kSWaves; ++w) {
            add += s_wcnt[w][tid];
            s_wcnt[w][tid] = 0;
        }
        s_lstart[tid] += add;  // becomes the cursor past the chunk's records of this digit
        __syncthreads();
    }
    // after the last chunk s_lstart[d] = local start + count; the local start is s_lstart[d] - s_cnt[d]
    for (uint32_t p = tid; p < n_here; p += kSBlock) {
        const uint64_t key = s_key[p];
        const uint32_t d = static_cast<uint32_t>(key >> shift) & 255u;
        const uint32_t dst = s_goff[d] + (p - (s_lstart[d] - s_cnt[d]));
        const uint2 py = s_pay[p];
        ident_out[dst] = key;
        ref_out[dst] = py.x;
        gbin_out[dst] = py.y;
        if (kChk) chk_out[dst] = s_chk[p];
    }
}

void launch_sort_by_ident(hipStream_t st, uint32_t n_upper, const uint32_t* counters, uint64_t* ident, uint32_t* cref,
                          uint32_t* cgbin, uint64_t* ident_tmp, uint32_t* cref_tmp, uint32_t* cgbin_tmp, uint32_t* hist,
                          uint32_t* cchk, uint32_t* cchk_tmp) {
    const uint32_t nt = (n_upper + kSTile - 1) / kSTile;
    if (nt == 0) return;
    uint32_t* totals = hist + static_cast<size_t>(256) * nt;  // 256 words behind the matrix
    uint64_t* ki = ident;
    uint32_t* ri = cref;
    uint32_t* gi = cgbin;
    uint64_t* ko = ident_tmp;
    uint32_t* ro = cref_tmp;
    uint32_t* go = cgbin_tmp;
    uint32_t* ci = cchk;
    uint32_t* co = cchk_tmp;
    for (uint32_t pass = 0; pass < 8; ++pass) {
        const uint32_t shift = pass * 8;
        hipLaunchKernelGGL(k_rs_hist, dim3(nt), dim3(kSBlock), 0, st, ki, counters, shift, nt, hist);
        hipLaunchKernelGGL(k_rs_rowscan, dim3(256), dim3(1024), 0, st, hist, nt, totals);
        if (cchk)
            hipLaunchKernelGGL(k_rs_scatter<true>, dim3(nt), dim3(kSBlock), 0, st, ki, ri, gi, ci, counters, shift, nt, hist,
                               totals, ko, ro, go, co);
        else
            hipLaunchKernelGGL(k_rs_scatter<false>, dim3(nt), dim3(kSBlock), 0, st, ki, ri, gi, ci, counters, shift, nt, hist,
                               totals, ko, ro, go, co);
        uint64_t* tk = ki; ki = ko; ko = tk;
        uint32_t* tr = ri; ri = ro; ro = tr;
        uint32_t* tg = gi; gi = go; go = tg;
        uint32_t* tc = ci; ci = co; co = tc;
    }
    // 8 passes: the result is back in (ident, cref, cgbin)
}

}  // namespace slimm
