// LDS-privatised coverage histograms (gfx950, wave64): cov / uniq_cov without a single global atomic on a bin.
//
// Why: a device-scope atomic on a random 4-byte bin is a separate memory-side request; MI355X retires ~20 G of them per
// second (measured: k_hist, 8.2 M + 1.1 M atomics = 408 us at config 2 -- 4 % of the HBM roofline).  The targets carry
// no locality (reads land anywhere on any genome), so they are first bucketed by bin TILE (8192 or 16384 consecutive bins:
// this file is compiled once per tile size, see kernels.h), then one workgroup per tile accumulates its bucket in LDS
// and writes the finished tile with coalesced 16-byte stores.  The tile write-back also replaces the zero-fill of cov /
// uniq_cov.
//
//   k_tile_count    persistent grid; per-workgroup LDS histogram of tile ids over its slots of targets (front.hip), added
//                   to one of 8 copies of tile_count[] with one non-returning global atomic per non-empty tile
//   k_tile_scan     one workgroup: sums the copies, exclusive scan -> tile_base, turns every copy into the start of its
//                   stretch inside the buckets, cuts buckets into work items of <= 16 K entries, lists the split tiles
//   k_tile_scatter  same slots, one level: rounds of 8 K values held in registers; LDS count, one returning atomic per
//                   tile and round on the copy's cursor, 16-bit entries (bin-in-tile | unique bit) out -- up to
//                   4096 tiles (k_tile_scatter_fused) ordered by tile in LDS first, so that a tile's run leaves as
//                   consecutive stores.  Also zeroes the tiles that k_tile_hist will accumulate with atomics.
//   k_part_super    two levels, level 1: targets go to their SUPER tile (64 tiles) as 32-bit words
//                   (bin-in-super | unique bit): few destinations per workgroup, long runs
//   k_part_tile     level 2: work items of <= 32 K entries of one super tile are split into its 64 tiles
//   k_tile_hist     one workgroup per work item: LDS cov[tile] + uniq_cov[tile], bucket in, finished tile out, plus the
//                   per-reference {sum, non-zero} statistics of the tile and (multi-GPU) its 'bin != 0' bitmaps
//   k_pack          small result arrays behind the statistics; non-zero counts / bitmaps of the split tiles
//
// Reference semantics: src/slimm.hpp:219-257 (cov[bin]++ per target; uniq_cov[bin]++ when the read has one target).
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>

#include <algorithm>

#include "force.h"
#include "kernels.h"

// compiled once per tile size (Makefile: -DSLIMM_TILE_SHIFT=13 -> namespace tiles13, =14 -> tiles14; kernels.h)
#ifndef SLIMM_TILE_SHIFT
#error "tile_hist.hip is compiled with -DSLIMM_TILE_SHIFT=13 or 14"
#endif
#if SLIMM_TILE_SHIFT == 13
#define SLIMM_TILE_NS tiles13
#elif SLIMM_TILE_SHIFT == 14
#define SLIMM_TILE_NS tiles14
#else
#error "a bucket entry holds 14 bits of bin and the unique bit"
#endif

namespace slimm {
namespace SLIMM_TILE_NS {

constexpr uint32_t kTileShift = SLIMM_TILE_SHIFT;
constexpr uint32_t kTileBins = 1u << kTileShift;        // bins per tile
constexpr uint32_t kSuperShift = kTileShift + 6;        // bins per super tile (kSuperTiles = 64 tiles)
constexpr uint32_t kSuperMask = (1u << kSuperShift) - 1;

#if defined(EXP) && EXP == 8
__device__ unsigned long long g_prof_t[8 * 4096];  // [workgroup-wave][phase]
#define TPROF_T(x) const unsigned long long x = __builtin_readcyclecounter()
#define TPROF_ADD(slot, a, b) if ((threadIdx.x & 63) == 0) g_prof_t[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + slot] += (b) - (a)
#else
#define TPROF_T(x)
#define TPROF_ADD(slot, a, b)
#endif
#if defined(EXP) && EXP == 9  // cycle split of k_tile_hist: thread 0 of every workgroup (scripts/tprof_tiles.py)
__device__ unsigned long long g_prof_t[8 * 4096];
#define HPROF_T(x) const unsigned long long x = __builtin_readcyclecounter()
#define HPROF_ADD(slot, a, b) if (threadIdx.x == 0) g_prof_t[(blockIdx.x & 4095u) * 8 + slot] += (b) - (a)
#else
#define HPROF_T(x)
#define HPROF_ADD(slot, a, b)
#endif

constexpr int kTBlock = 512;
constexpr uint32_t kTileMask = kTileBins - 1;
constexpr uint32_t kScanStaged = 16384;  // tiles whose counts k_tile_scan stages in LDS (the one-level bucketing range)

// The bucketing kernels read values that lie in slots (front.hip): slot s holds its entries compacted at
// [slots[s].x, slots[s].x + n) with n = slots[s].y (one value per target) or slots[s].z (one value per read).
// Workgroup b of g owns a contiguous range of slots -- the SAME range in the count and in the scatter kernel, whose
// counter copies pair up by workgroup -- and its waves take the slots of that range round robin, 64 entries at a time.
struct SlotWalk {  // all wave-uniform
    const uint4* slots;       // nullptr: the dense form -- [base, d_end) are this wave's values still to come, in pieces
                              // of the caller's size, step of them apart (the workgroup's waves take its stretch's
                              // pieces round robin)
    uint32_t d_end, d_wave;
    bool d_first;
    uint32_t s, s_end;        // next slot of this wave, end of the workgroup's range
    uint32_t step;            // waves of the workgroup
    uint32_t base, left;      // entries of the current slot not yet handed out: [base, base + left)
    uint32_t sum_f, sum_h, sum_v;  // targets, reads and mapped records of the slots taken so far
    bool per_read;
    uint4 nd;                 // slots[s], asked for when the slot before it was taken up (a scalar load per slot with
                              // its wait right behind it is a round trip per ~600 values in every kernel that walks)
};

constexpr uint32_t kPiece = 256;
constexpr uint32_t kPieceMax = 256;  // the largest piece any kernel asks slot_next for
// fold: this workgroup stands for `fold` workgroups of the bucketing grid (k_tile_count: fewer, larger workgroups flush
// fewer LDS histograms with global atomics)
// wg: the workgroup's index among the owners of slot ranges (blockIdx.x, or a permutation of it: xcd_logical_id)
__device__ __forceinline__ SlotWalk slot_walk(const uint4* slots, uint32_t nslots, bool per_read, uint32_t fold = 1,
                                              uint32_t wg = blockIdx.x) {
    SlotWalk w;
    w.slots = slots;
    w.step = blockDim.x >> 6;
    w.d_end = 0;
    w.d_wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    w.d_first = true;
    w.nd = make_uint4(0u, 0u, 0u, 0u);
    if (!slots) {  // dense: nslots VALUES; every unit of the bucketing grid a stretch of whole pieces
        const uint32_t units = gridDim.x * fold;
        const uint32_t per_unit = ((nslots + units - 1u) / units + kPieceMax - 1u) & ~(kPieceMax - 1u);
        const uint64_t lo64 = static_cast<uint64_t>(wg) * per_unit * fold;
        const uint32_t lo = static_cast<uint32_t>(min<uint64_t>(lo64, nslots));
        w.base = lo;
        w.d_end = static_cast<uint32_t>(min<uint64_t>(lo64 + static_cast<uint64_t>(per_unit) * fold, nslots));
        w.left = 0;
        w.s = w.s_end = 0;
        w.sum_f = w.sum_h = w.sum_v = 0;
        w.per_read = per_read;
        return w;
    }
    const uint32_t per_wg = (nslots + gridDim.x * fold - 1) / (gridDim.x * fold) * fold;
    const uint32_t lo = min(wg * per_wg, nslots);
    w.s = lo + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    w.s_end = min(lo + per_wg, nslots);
    w.nd = w.s < w.s_end ? slots[w.s] : make_uint4(0u, 0u, 0u, 0u);
    w.base = 0;
    w.left = 0;
    w.sum_f = w.sum_h = w.sum_v = 0;
    w.per_read = per_read;
    return w;
}

// Which slot range (and with it which copy of the tile counters and cursors) a bucketing workgroup takes.  Workgroups
// whose blockIdx agree mod 8 share an XCD, i.e. an L2; a counter copy's bucket frontiers -- the cache lines its
// workgroups append 2-byte entries to -- are written back whole only if ONE L2 collects the entries, otherwise every
// XCD writes its partial line through (PMC at config 4: 4.2 GB written for 1.23 GB of entries).  The logical id is a
// permutation of blockIdx with copy(logical) = (logical / kCountFold) % kTileReps == blockIdx % 8: all workgroups of a
// copy on one XCD.  (The counting workgroup c = logical / kCountFold adds to copy c % 8 and sits on XCD label c % 8 as
// well.)  Grids that are no multiple of 16: the identity.
__device__ __forceinline__ uint32_t xcd_logical_id() {
    static_assert(kTileReps == 8 && kCountFold == 2, "the permutation below is written for 8 copies, 2 halves");
    const uint32_t b = blockIdx.x;
    if (gridDim.x % 16u) return b;
    const uint32_t q = b >> 3, x = b & 7u;
    return 16u * (q >> 1) + 2u * x + (q & 1u);
}

// the next (up to) 256 entries: returns their count (0: the wave has no more), *base = index of the first
__device__ __forceinline__ uint32_t slot_next(SlotWalk& w, uint32_t* base, uint32_t cap = kPiece) {
    if (!w.slots) {
        if (w.d_first) {
            w.base += w.d_wave * cap;
            w.d_first = false;
        }
        if (w.base >= w.d_end) return 0u;
        const uint32_t n = min(cap, w.d_end - w.base);
        *base = w.base;
        w.base += w.step * cap;
        return n;
    }
    while (w.left == 0u) {
        if (w.s >= w.s_end) return 0u;
        const uint4 d = w.nd;
        w.s += w.step;
        if (w.s < w.s_end) w.nd = w.slots[w.s];  // (a scalar load: one address for the wave)
        w.base = d.x;
        w.left = w.per_read ? d.z : d.y;
        w.sum_f += d.y;
        w.sum_h += d.z;
        w.sum_v += d.w;
    }
    const uint32_t n = min(w.left, cap);
    *base = w.base;
    w.base += n;
    w.left -= n;
    return n;
}

// A piece of n <= 256 values at vals[base ...): four consecutive values per lane in ONE 16-byte load (a 4-byte load per
// lane is a load instruction per 64 values; the slots' values start anywhere, so the load is only dword-aligned, which
// global loads take).  Values behind the piece read as 0xffffffff = nothing; the load itself may reach up to three
// entries behind the piece -- other slots' values or the arrays' padding (ensure_work_buffers).
struct __attribute__((packed, aligned(4))) Quad {
    uint32_t a, b, c, d;
};
__device__ __forceinline__ void piece_load(const uint32_t* __restrict__ vals, uint32_t base, uint32_t n, uint32_t lane,
                                           uint32_t* v) {
    const uint32_t i = 4u * lane;
    Quad q{0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    if (i < n) q = *reinterpret_cast<const Quad*>(vals + base + i);
    v[0] = i < n ? q.a : 0xffffffffu;
    v[1] = i + 1u < n ? q.b : 0xffffffffu;
    v[2] = i + 2u < n ? q.c : 0xffffffffu;
    v[3] = i + 3u < n ? q.d : 0xffffffffu;
}

// The wave's next (up to) 256 values with every lane at work: a slot holds ~470 values at 0.6 targets per record, so its
// second piece is short -- a fifth of the lanes of every round had nothing to count or place.  Lanes behind the quads of a
// short piece take the first quads of the NEXT slot's values (one more piece of the walk, as long as lanes are left;
// a slot with fewer values than that leaves the rest idle).  Returns whether the wave got any value.
__device__ __forceinline__ bool piece_load_packed(SlotWalk& w, const uint32_t* __restrict__ vals, uint32_t lane, uint32_t* v) {
    uint32_t base_a = 0, base_b = 0;
    const uint32_t n_a = slot_next(w, &base_a);
    const uint32_t q_a = (n_a + 3u) >> 2;                       // lanes the first piece takes
    const uint32_t n_b = (n_a != 0u && q_a < 64u) ? slot_next(w, &base_b, 4u * (64u - q_a)) : 0u;
    const bool second = lane >= q_a;
    const uint32_t i = 4u * (second ? lane - q_a : lane), n = second ? n_b : n_a;
    Quad q{0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    if (i < n) q = *reinterpret_cast<const Quad*>(vals + (second ? base_b : base_a) + i);
    v[0] = i < n ? q.a : 0xffffffffu;
    v[1] = i + 1u < n ? q.b : 0xffffffffu;
    v[2] = i + 2u < n ? q.c : 0xffffffffu;
    v[3] = i + 3u < n ? q.d : 0xffffffffu;
    return n_a != 0u;
}

// tile of a value: bit 31 (unique read) is no part of the bin index; 0xffffffff = nothing to count (a read without
// a selector)
__device__ __forceinline__ uint32_t tile_of(uint32_t v) { return (v & 0x7fffffffu) >> kTileShift; }

__global__ __launch_bounds__(kTBlock * kCountFold) void k_tile_count(const uint32_t* __restrict__ vals, const uint4* __restrict__ slots,
                                                        uint32_t nslots, int per_read, uint32_t ntiles,
                                                        uint32_t* __restrict__ tile_count_all, uint32_t reps,
                                                        uint32_t rep_stride, uint4* __restrict__ part,
                                                        uint32_t* __restrict__ matrix) {
    HIP_DYNAMIC_SHARED(uint32_t, s_hist)
    __shared__ uint32_t s_sum[3];
    if (threadIdx.x < 3) s_sum[threadIdx.x] = 0;
    // 512 workgroups adding to the same 2.4 K counters serialise in the memory-side atomic units: every workgroup adds to
    // one of `reps` copies instead, and k_tile_scan sums the copies
    uint32_t* __restrict__ tile_count = tile_count_all + static_cast<size_t>(blockIdx.x % reps) * rep_stride;
    for (uint32_t i = threadIdx.x; i < ntiles; i += kTBlock * kCountFold) s_hist[i] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    SlotWalk w = slot_walk(slots, nslots, per_read != 0, kCountFold);
    while (true) {  // four pieces of 256 values per trip, their loads in flight together
        uint32_t v[16];
        bool any = false;
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // (piece_load_packed: 460 -> 499 us here -- this kernel waits for memory, not for lanes)
            uint32_t base = 0;
            const uint32_t n = slot_next(w, &base);
            any = any || n != 0u;
            piece_load(vals, base, n, lane, v + 4 * u);
        }
        if (!any) break;
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (v[u] != 0xffffffffu) atomicAdd(&s_hist[tile_of(v[u])], 1u);
    }
    if (part && lane == 0u) {  // the totals of this workgroup's slots (the front end leaves them to its first consumer)
        atomicAdd(&s_sum[0], w.sum_v);
        atomicAdd(&s_sum[1], w.sum_h);
        atomicAdd(&s_sum[2], w.sum_f);
    }
    __syncthreads();
    if (part && threadIdx.x == 0) part[blockIdx.x] = make_uint4(s_sum[0], s_sum[1], s_sum[2], 0u);
    if (matrix) {  // the workgroup's own row of the count matrix: plain coalesced stores, zeros included
        uint32_t* __restrict__ row = matrix + static_cast<size_t>(blockIdx.x) * rep_stride;
        for (uint32_t i = threadIdx.x; i < ntiles; i += kTBlock * kCountFold) row[i] = s_hist[i];
        return;
    }
    for (uint32_t i = threadIdx.x; i < ntiles; i += kTBlock * kCountFold) {
        uint32_t h = s_hist[i];
        if (h) atomicAdd(&tile_count[i], h);
    }
}

// Matrix bucketing (layouts of more than kFusedScanTiles tiles): every counting workgroup keeps a ROW of its own in a
// count matrix [workgroup][tile]; this kernel turns every column into its exclusive prefix over the rows -- where each
// workgroup's stretch starts inside the tile's bucket -- and leaves the column sums in total[] for k_tile_scan.  The
// scatter then needs no global atomic and no rounds at all: a workgroup's place for a value is tile_base[tile] + its
// row's prefix + a running LDS count (k_tile_scatter_matrix).  The direct rounds paid one returning global atomic per
// touched tile and round of 16 K values -- 0.48 per value at config 3 -- and four barriers per round.
__global__ __launch_bounds__(256) void k_matrix_prefix(uint32_t* __restrict__ matrix, uint32_t nrows, uint32_t ntiles,
                                                       uint32_t row_stride, uint32_t* __restrict__ total) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= ntiles) return;
    uint32_t run = 0;
    uint32_t r = 0;
    for (; r + 16u <= nrows; r += 16u) {  // sixteen rows' loads in flight together (the addresses do not depend on the values)
        uint32_t v[16];
#pragma unroll
        for (uint32_t u = 0; u < 16u; ++u) v[u] = matrix[static_cast<size_t>(r + u) * row_stride + t];
#pragma unroll
        for (uint32_t u = 0; u < 16u; ++u) {
            matrix[static_cast<size_t>(r + u) * row_stride + t] = run;
            run += v[u];
        }
    }
    for (; r < nrows; ++r) {
        const uint32_t v = matrix[static_cast<size_t>(r) * row_stride + t];
        matrix[static_cast<size_t>(r) * row_stride + t] = run;
        run += v;
    }
    total[t] = run;
}

// the totals {mapped records, reads, targets} of the stream from the per-workgroup sums k_tile_count left: into the
// counter block (from where they reach the host) and the scalars that travel with the bins (multi-GPU)
__device__ __forceinline__ void publish_totals(const uint4* __restrict__ part, uint32_t nparts, uint32_t* __restrict__ counters,
                                               uint32_t* __restrict__ tail, uint32_t* s_tot) {
    if (threadIdx.x < 3) s_tot[threadIdx.x] = 0;
    __syncthreads();
    uint32_t v = 0, h = 0, f = 0;
    for (uint32_t i = threadIdx.x; i < nparts; i += blockDim.x) {
        const uint4 p = part[i];
        v += p.x;
        h += p.y;
        f += p.z;
    }
    v = wave_sum_dpp(v);  // (one LDS add per wave and total, not one per thread on three addresses)
    h = wave_sum_dpp(h);
    f = wave_sum_dpp(f);
    if ((threadIdx.x & 63u) == 63u) {
        if (v) atomicAdd(&s_tot[0], v);
        if (h) atomicAdd(&s_tot[1], h);
        if (f) atomicAdd(&s_tot[2], f);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        counters[CNT_V] = s_tot[0];
        counters[CNT_M] = s_tot[1];
        counters[CNT_P] = s_tot[2];
        if (tail) {
            tail[0] = s_tot[0];
            tail[1] = s_tot[1];
            tail[2] = s_tot[2];
        }
    }
}

// One workgroup: exclusive scan tile_count -> tile_base (tile_base[ntiles] = total), zero tile_cursor, and cut every
// tile's bucket into work items of at most kTileSub entries for k_tile_hist: items[k] = {tile, lo, hi, pieces of tile}.
// A tile with no entry still gets one item (its finished tile is all zeros and has to be written).
__global__ __launch_bounds__(1024) void k_tile_scan(uint32_t* __restrict__ tile_count, uint32_t ntiles,
                                                    uint32_t* __restrict__ tile_base, uint32_t* __restrict__ tile_cursor,
                                                    uint4* __restrict__ items, uint32_t* __restrict__ counters,
                                                    uint4* __restrict__ items2, uint32_t* __restrict__ sup_cursor,
                                                    uint32_t* __restrict__ split_tiles, uint32_t reps,
                                                    uint32_t rep_stride, int two_level, const uint4* __restrict__ part,
                                                    uint32_t nparts, uint32_t* __restrict__ tail, uint32_t tile_sub) {
    __shared__ uint2 s_part[1024];
    __shared__ uint32_t s_nsplit;
    __shared__ uint32_t s_tot[3];
    if (part) publish_totals(part, nparts, counters, tail, s_tot);
    __shared__ uint32_t s_cnt[kScanStaged];  // a stretch of tiles: a tile's total over the copies, then its base
    __shared__ uint2 s_wave[16];
    if (threadIdx.x == 0) s_nsplit = 0;
    const uint32_t tid = threadIdx.x;
    const bool kept = ntiles <= 4096 && reps == kTileReps;  // the copies of <= 4 tiles per thread stay in registers
    uint32_t keep[4][kTileReps];
    uint2 carry = make_uint2(0u, 0u);  // entries and work items of the stretches before this one
    // stretches of kScanStaged tiles (one for the one-level layouts; the two-level ones, reps == 1, have up to 64 K tiles:
    // reading them in place, every thread its own consecutive tiles, was 85 us for the 45 K tiles of 200 k references)
    for (uint32_t c0 = 0; c0 < ntiles; c0 += kScanStaged) {
        const uint32_t n = min(kScanStaged, ntiles - c0);
        uint32_t* tc = tile_count + c0;
        uint32_t* tcur = tile_cursor + c0;
        if (kept) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {  // coalesced, the copies' loads independent of each other
                const uint32_t i = tid + q * 1024;
                if (i >= n) break;
                uint32_t c = 0;
#pragma unroll
                for (uint32_t rep = 0; rep < kTileReps; ++rep) keep[q][rep] = tc[static_cast<size_t>(rep) * rep_stride + i];
#pragma unroll
                for (uint32_t rep = 0; rep < kTileReps; ++rep) c += keep[q][rep];
                s_cnt[i] = c;
            }
        } else {
            for (uint32_t i = tid; i < n; i += 1024) {
                uint32_t c = 0;
                if (reps == kTileReps) {
                    uint32_t v[kTileReps];
#pragma unroll
                    for (uint32_t rep = 0; rep < kTileReps; ++rep) v[rep] = tc[static_cast<size_t>(rep) * rep_stride + i];
#pragma unroll
                    for (uint32_t rep = 0; rep < kTileReps; ++rep) c += v[rep];
                } else {
                    for (uint32_t rep = 0; rep < reps; ++rep) c += tc[static_cast<size_t>(rep) * rep_stride + i];
                }
                s_cnt[i] = c;
            }
        }
        __syncthreads();
        const uint32_t per = (n + 1023) / 1024;
        const uint32_t lo = min(tid * per, n), hi = min(lo + per, n);
        uint2 sum = make_uint2(0u, 0u);
        for (uint32_t i = lo; i < hi; ++i) {
            const uint32_t c = s_cnt[i];
            sum.x += c;
            sum.y += c ? (c + tile_sub - 1) / tile_sub : 1u;
        }
        {   // inclusive scan of the threads' sums: inside the waves by shuffles, the 16 wave totals through LDS (one barrier
            // instead of the twenty of a scan by doubling over the whole workgroup)
            const uint32_t lane = tid & 63u, wave = tid >> 6;
            uint2 inc = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t ax = __shfl_up(inc.x, o, 64), ay = __shfl_up(inc.y, o, 64);
                if (lane >= static_cast<uint32_t>(o)) {
                    inc.x += ax;
                    inc.y += ay;
                }
            }
            if (lane == 63) s_wave[wave] = inc;
            __syncthreads();
            uint2 before = carry;
#pragma unroll
            for (uint32_t w = 0; w < 16; ++w) {
                const uint2 t = s_wave[w];
                if (w < wave) {
                    before.x += t.x;
                    before.y += t.y;
                }
            }
            s_part[tid] = make_uint2(before.x + inc.x, before.y + inc.y);
        }
        uint2 run = make_uint2(s_part[tid].x - sum.x, s_part[tid].y - sum.y);
        for (uint32_t i = lo; i < hi; ++i) {
            const uint32_t c = s_cnt[i];
            const uint32_t pieces = c ? (c + tile_sub - 1) / tile_sub : 1u;
            s_cnt[i] = run.x;
            if (pieces > 1) split_tiles[atomicAdd(&s_nsplit, 1u)] = c0 + i;
            for (uint32_t k = 0; k < pieces; ++k) {
                uint32_t a = run.x + k * tile_sub;
                uint32_t b = min(a + tile_sub, run.x + c);
                items[run.y + k] = make_uint4(c0 + i, a, b, pieces);
            }
            run.x += c;
            run.y += pieces;
        }
        __syncthreads();
        // the copies of a tile's count become the start of every copy's stretch inside the tile's bucket
        if (kept) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t i = tid + q * 1024;
                if (i >= n) break;
                uint32_t at = s_cnt[i];
                tile_base[c0 + i] = at;
#pragma unroll
                for (uint32_t rep = 0; rep < kTileReps; ++rep) {
                    const size_t k = static_cast<size_t>(rep) * rep_stride + i;
                    tc[k] = at;
                    tcur[k] = 0;
                    at += keep[q][rep];
                }
            }
        } else {
            for (uint32_t i = tid; i < n; i += 1024) {
                uint32_t at = s_cnt[i];
                tile_base[c0 + i] = at;
                if (reps == kTileReps) {
                    uint32_t v[kTileReps];
#pragma unroll
                    for (uint32_t rep = 0; rep < kTileReps; ++rep) v[rep] = tc[static_cast<size_t>(rep) * rep_stride + i];
#pragma unroll
                    for (uint32_t rep = 0; rep < kTileReps; ++rep) {
                        const size_t k = static_cast<size_t>(rep) * rep_stride + i;
                        tc[k] = at;
                        tcur[k] = 0;
                        at += v[rep];
                    }
                } else {
                    for (uint32_t rep = 0; rep < reps; ++rep) {
                        const size_t k = static_cast<size_t>(rep) * rep_stride + i;
                        const uint32_t cr = tc[k];
                        tc[k] = at;
                        tcur[k] = 0;
                        at += cr;
                    }
                }
            }
        }
        carry = s_part[1023];
        __syncthreads();  // (s_cnt, s_part and s_wave are the next stretch's)
    }
    if (tid == 1023) {
        tile_base[ntiles] = carry.x;
        counters[CNT_ITEMS] = carry.y;
    }
    __shared__ uint32_t s_n2;
    if (tid == 0) s_n2 = 0;
    __threadfence_block();
    __syncthreads();  // tile_base is complete (same workgroup, same CU)
    // work items of k_part_tile: <= kPartSub entries of one super tile each (their order does not matter)
    const uint32_t nsup = two_level ? (ntiles + kSuperTiles - 1) / kSuperTiles : 0u;  // (one-level bucketing: none)
    for (uint32_t sp = tid; sp < nsup; sp += 1024) {
        sup_cursor[sp] = 0;
        const uint32_t a = tile_base[sp * kSuperTiles];
        const uint32_t b = tile_base[min((sp + 1) * kSuperTiles, ntiles)];
        const uint32_t pieces = (b - a + kPartSub - 1) / kPartSub;
        if (pieces) {
            uint32_t slot = atomicAdd(&s_n2, pieces);
            for (uint32_t q = a; q < b; q += kPartSub) items2[slot++] = make_uint4(sp, q, min(q + kPartSub, b), 0u);
        }
    }
    __syncthreads();
    if (tid == 0) {
        counters[CNT_ITEMS2] = s_n2;
        counters[CNT_SPLIT] = s_nsplit;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Bucketing rounds.  In one round every wave of the workgroup takes kRoundPieces / 4 pieces of 256 values from its slots
// into registers, four consecutive values per lane (at most kRoundCap values per workgroup and round).
// ---------------------------------------------------------------------------------------------------------
constexpr int kRoundPieces = 16;
constexpr int kDirectPieces = 32;  // ... of the direct rounds (no stage in LDS): one returning atomic per touched tile and round
constexpr uint32_t kTileSlots = 4096;  // entries of the ordered round's tile tables (= kFusedTiles)
constexpr uint32_t kRoundCap = (kTBlock / 64) * kRoundPieces * 64;  // 8192 values per round

// loads the round's values (0xffffffff where there is none); returns whether this wave got any.  kWide: pieces of 256
// values, four consecutive ones per lane in a 16-byte load -- for the rounds that order their values in LDS before
// they leave.  The direct rounds keep pieces of 64 values, one per lane: their stores go out straight from the
// registers, and consecutive targets (one read's neighbouring references) often share a tile and so get consecutive
// bucket places -- with consecutive LANES holding them a store instruction's lanes fall into few cache lines, with
// four values per lane they are four targets apart (measured: config 3 362 -> 392 us with the wide pieces).
template <int kPieces, bool kWide>
__device__ __forceinline__ bool round_load(SlotWalk& w, const uint32_t* __restrict__ vals, uint32_t lane,
                                           uint32_t (&v)[kPieces]) {
    static_assert(kPieces % 4 == 0, "values come four to a lane");
    bool any = false;
    if (kWide) {
#pragma unroll
        for (int k = 0; k < kPieces; k += 4) any = piece_load_packed(w, vals, lane, v + k) || any;
    } else {
#pragma unroll
        for (int k = 0; k < kPieces; ++k) {
            uint32_t base = 0;
            const uint32_t n = slot_next(w, &base, 64u);
            any = any || n != 0u;
            v[k] = lane < n ? vals[base + lane] : 0xffffffffu;
        }
    }
    return any;
}

// bucket entry of a value: 13-bit bin inside its tile | the unique bit
constexpr uint32_t kEntryBits = kTileShift + 1;  // a bucket entry: bin in tile | unique bit
static_assert(kEntryBits <= 16, "bucket entries are 16-bit");
__device__ __forceinline__ uint32_t entry_of(uint32_t v) { return (v & kTileMask) | ((v >> 31) << kTileShift); }

// One round of the one-level bucketing with the values ORDERED BY TILE in LDS before they leave (<= 4096 tiles).
// Scattered 2-byte stores straight from registers were one L2 write request per value (8.2 M requests per file, ~half of
// the old kernel's time); here a value's place inside the round is rank-inside-its-tile (the returned count of the LDS
// histogram add) + the tile's offset (a scan of that histogram), the values are staged in that order, and consecutive
// lanes then store consecutive bucket positions of the same tile: one request per (tile, round) run.
//   s_cnt   [4096]  values per tile in this round; then: global position of the tile's run minus its place in the stage
//   s_loff  [4096]  exclusive scan of s_cnt
//   s_stage [8192]  tile << kEntryBits | entry, in tile order
// mine[t] = start of this workgroup's counter copy inside tile t's bucket, cursor[t] = the copy's fill (global atomics).
__device__ __forceinline__ uint32_t block_excl_scan_4096(uint32_t* s, uint32_t* s_wtot);

__device__ __forceinline__ void scatter_round_ordered(const uint32_t (&v)[kRoundPieces], uint32_t ntiles,
                                                      const uint32_t* s_mine, uint32_t* __restrict__ tile_cursor,
                                                      uint16_t* __restrict__ bucket, uint32_t* s_cnt, uint32_t* s_loff,
                                                      uint32_t* s_stage, uint32_t* s_wtot) {
    const uint32_t tid = threadIdx.x;
    TPROF_T(p0);
    for (uint32_t i = tid; i < kTileSlots; i += kTBlock) s_cnt[i] = 0;
    __syncthreads();
    TPROF_T(p1);
    TPROF_ADD(1, p0, p1);
    uint32_t r[kRoundPieces];
#pragma unroll
    for (int k = 0; k < kRoundPieces; ++k) r[k] = v[k] != 0xffffffffu ? atomicAdd(&s_cnt[tile_of(v[k])], 1u) : 0u;
    __syncthreads();
    TPROF_T(p2);
    TPROF_ADD(2, p1, p2);
    // this round's stretch of every touched tile's bucket: one returning atomic per tile, all of a thread's issued
    // before the scan so that they are back when it is done
    uint32_t got[kTileSlots / kTBlock];
#pragma unroll
    for (uint32_t j = 0; j < kTileSlots / kTBlock; ++j) {
        const uint32_t i = j * kTBlock + tid;
        const uint32_t h = s_cnt[i];
        s_loff[i] = h;
        // (nothing but the atomic inside the condition, its result not used before the scan is done: with "+ s_mine[i]"
        // in here every reservation was a branch with a wait at its end -- eight round trips in a row per round)
        got[j] = 0u;
        if (h && i < ntiles) got[j] = atomicAdd(&tile_cursor[i], h);
    }
    __syncthreads();
    TPROF_T(p3);
    TPROF_ADD(3, p2, p3);
    const uint32_t total = block_excl_scan_4096(s_loff, s_wtot);
#pragma unroll
    for (uint32_t j = 0; j < kTileSlots / kTBlock; ++j) {
        const uint32_t i = j * kTBlock + tid;
        s_cnt[i] = got[j] + s_mine[i] - s_loff[i];  // (entries behind the last tile: never read)
    }
    TPROF_T(p4);
    TPROF_ADD(4, p3, p4);
#pragma unroll
    for (int k = 0; k < kRoundPieces; ++k) {
        if (v[k] == 0xffffffffu) continue;
        const uint32_t t = tile_of(v[k]);
        s_stage[s_loff[t] + r[k]] = (t << kEntryBits) | entry_of(v[k]);
    }
    __syncthreads();
    TPROF_T(p5);
    TPROF_ADD(5, p4, p5);
    for (uint32_t j = tid; j < total; j += kTBlock) {
        const uint32_t e = s_stage[j];
        bucket[s_cnt[e >> kEntryBits] + j] = static_cast<uint16_t>(e & ((1u << kEntryBits) - 1u));
    }
    __syncthreads();  // the next round clears s_cnt and refills the stage
    TPROF_T(p6);
    TPROF_ADD(6, p5, p6);
}

// One round of the one-level bucketing straight from registers (more than 4096 tiles: the tile tables of the ordered
// form no longer fit LDS beside a stage).  s_hist[ntiles]: the round's histogram, then the write cursors.
template <int kPieces>
__device__ __forceinline__ void scatter_round_direct(const uint32_t (&v)[kPieces], uint32_t ntiles,
                                                     const uint32_t* __restrict__ tile_base,
                                                     uint32_t* __restrict__ tile_cursor, uint16_t* __restrict__ bucket,
                                                     uint32_t* s_hist) {
    constexpr int kMaxTilesPerThread = 8;  // tiles per thread whose reservations are in flight together
    for (uint32_t i = threadIdx.x; i < ntiles; i += kTBlock) s_hist[i] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kPieces; ++k)
        if (v[k] != 0xffffffffu) atomicAdd(&s_hist[tile_of(v[k])], 1u);
    __syncthreads();
    for (uint32_t i0 = 0; i0 < ntiles; i0 += kMaxTilesPerThread * kTBlock) {
        uint32_t got[kMaxTilesPerThread];
#pragma unroll
        for (int j = 0; j < kMaxTilesPerThread; ++j) {
            const uint32_t i = i0 + j * kTBlock + threadIdx.x;
            const uint32_t h = i < ntiles ? s_hist[i] : 0u;
            got[j] = 0u;
            if (h) got[j] = atomicAdd(&tile_cursor[i], h);  // tiles this round does not touch: no atomic.  (Nothing but
                                                            // the atomic in here: a use of its result is a wait per tile)
        }
#pragma unroll
        for (int j = 0; j < kMaxTilesPerThread; ++j) {
            const uint32_t i = i0 + j * kTBlock + threadIdx.x;
            if (i < ntiles) s_hist[i] = got[j] + tile_base[i];
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kPieces; ++k) {
        if (v[k] == 0xffffffffu) continue;
        const uint32_t pos = atomicAdd(&s_hist[tile_of(v[k])], 1u);
        bucket[pos] = static_cast<uint16_t>(entry_of(v[k]));
    }
    __syncthreads();  // s_hist is cleared by the next round
}

// rounds until no wave of the workgroup has values left (s_more: one flag per wave)
template <int kPieces = kRoundPieces, bool kWide = true, typename Body>
__device__ __forceinline__ void bucketing_rounds(SlotWalk& walk, const uint32_t* __restrict__ vals, uint32_t* s_more,
                                                 Body body) {
    while (true) {
        uint32_t v[kPieces];
        const bool mine = round_load<kPieces, kWide>(walk, vals, threadIdx.x & 63u, v);
        if ((threadIdx.x & 63u) == 0) s_more[threadIdx.x >> 6] = mine ? 1u : 0u;
        __syncthreads();
        uint32_t any = 0;
#pragma unroll
        for (int w = 0; w < kTBlock / 64; ++w) any |= s_more[w];
        __syncthreads();
        if (!any) break;
        body(v);
    }
}

// tiles cut into several work items are accumulated by k_tile_hist with (contiguous) global atomics: zero them first
__device__ __forceinline__ void zero_split_tiles(const uint32_t* tile_base, uint32_t ntiles, uint32_t* __restrict__ cov,
                                                 uint32_t* __restrict__ ucov, uint32_t tile_sub) {
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        if (tile_base[tile + 1] - tile_base[tile] <= tile_sub) continue;
        uint4* oc = reinterpret_cast<uint4*>(cov + static_cast<size_t>(tile) * kTileBins);
        uint4* ou = reinterpret_cast<uint4*>((ucov ? ucov : cov) + static_cast<size_t>(tile) * kTileBins);
        const uint4 z = make_uint4(0, 0, 0, 0);
        for (uint32_t i = threadIdx.x; i < kTileBins / 4; i += kTBlock) {
            oc[i] = z;
            if (ucov) ou[i] = z;
        }
    }
}

__global__ __launch_bounds__(kTBlock) void k_tile_scatter(const uint32_t* __restrict__ vals, const uint4* __restrict__ slots,
                                                          uint32_t nslots, int per_read, uint32_t ntiles,
                                                          const uint32_t* __restrict__ tile_base,
                                                          uint32_t* __restrict__ tile_cursor_all,
                                                          uint16_t* __restrict__ bucket, uint32_t* __restrict__ cov,
                                                          uint32_t* __restrict__ ucov,
                                                          const uint32_t* __restrict__ rep_base_all, uint32_t reps,
                                                          uint32_t rep_stride, uint32_t tile_sub) {
    HIP_DYNAMIC_SHARED(uint32_t, s_hist)
    __shared__ uint32_t s_more[kTBlock / 64];
    const uint32_t lid = xcd_logical_id();
    const size_t rep_off = static_cast<size_t>((lid / kCountFold) % reps) * rep_stride;
    uint32_t* __restrict__ tile_cursor = tile_cursor_all + rep_off;
    const uint32_t* __restrict__ rep_base = rep_base_all + rep_off;
    zero_split_tiles(tile_base, ntiles, cov, ucov, tile_sub);
    SlotWalk walk = slot_walk(slots, nslots, per_read != 0, 1, lid);
    bucketing_rounds<kDirectPieces, false>(walk, vals, s_more, [&](const uint32_t (&v)[kDirectPieces]) {
        scatter_round_direct(v, ntiles, rep_base, tile_cursor, bucket, s_hist);
    });
}

// ---------------------------------------------------------------------------------------------------------
// One-level bucketing above kFusedScanTiles tiles with the values ORDERED BY TILE in LDS before they leave.
// The direct rounds (k_tile_scatter) store every 2-byte entry from a register: one L2 write request per value, 617 M at
// 1 B records -- about what the L2's channels take in the kernel's 2.4 ms -- beside one returning atomic per touched
// tile and round of 16 K values.  Here a workgroup of kBigBlock threads (the ONE of its CU: the LDS is its own) takes
// rounds of kBigBlock x kBigPieces values, and ONE table of a word per tile serves in turn as the round's histogram,
// as the tiles' cursors inside the stage (placement: a returning add), and as "global position of the tile's
// run minus its offset in the stage" (write-out: consecutive lanes store consecutive bucket positions of one
// tile -- a request per run, not per value).  Thread t owns tiles t, t + kBigBlock, ...: their counts, their reservations
// (in registers between the steps) and their runs' places in the stage -- thread-major, so that the scan is over threads.
// Same slot ranges and counter copies as k_tile_count (same grid, same fold): copy = blockIdx.x % reps, and the
// workgroups of a copy share an XCD (blockIdx.x % 8) like the frontiers they append to.
// ---------------------------------------------------------------------------------------------------------
#ifndef SLIMM_BIG_PIECES
#define SLIMM_BIG_PIECES 24
#endif
constexpr int kBigBlock = kTBlock * kCountFold;                  // 1024 threads
constexpr int kBigPieces = SLIMM_BIG_PIECES;                     // values per thread and round (four to a 16-byte load)
constexpr uint32_t kBigRound = kBigBlock * kBigPieces;           // 24 576 values per round
static_assert(kBigPieces % 8 == 0, "values come four to a lane; the rounds' LDS operations in batches of eight");
template <int kE>  // tiles per thread: ntiles <= kE * kBigBlock
__global__ __launch_bounds__(kBigBlock) void k_tile_scatter_big(const uint32_t* __restrict__ vals, const uint4* __restrict__ slots,
                                                                uint32_t nslots, int per_read, uint32_t ntiles,
                                                                const uint32_t* __restrict__ tile_base,
                                                                uint32_t* __restrict__ tile_cursor_all,
                                                                uint16_t* __restrict__ bucket, uint32_t* __restrict__ cov,
                                                                uint32_t* __restrict__ ucov,
                                                                const uint32_t* __restrict__ rep_base_all, uint32_t reps,
                                                                uint32_t rep_stride, uint32_t tile_sub) {
    // s_tab[kE * kBigBlock] | 64 more entries, nobody's tiles: where a lane's "no value" counts | s_stage[kBigRound] | 64
    // more words, where a lane's "no value" is put -- the rounds' LDS operations are issued unconditionally, batch after
    // batch (a value behind `if (there is one)` is a branch, and a returning LDS atomic inside a branch is waited for
    // inside it).  An entry per LANE: lanes without a value (behind a slot's short last piece: up to a fifth of a round's
    // before piece_load_packed) would meet in one shared entry and take it one after the other -- measured: 3 x slower.
    HIP_DYNAMIC_SHARED(uint32_t, s_dyn)
    __shared__ uint32_t s_more[kBigBlock / 64], s_wtot[kBigBlock / 64];
    constexpr uint32_t kNoTile = kE * kBigBlock;
    uint32_t* const s_tab = s_dyn;
    uint32_t* const s_stage = s_dyn + kNoTile + 64;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid < 64u) s_tab[kNoTile + tid] = 0u;
    const uint32_t no_tile = kNoTile + lane;
    const size_t rep_off = static_cast<size_t>(blockIdx.x % reps) * rep_stride;
    uint32_t* __restrict__ tile_cursor = tile_cursor_all + rep_off;
    const uint32_t* __restrict__ rep_base = rep_base_all + rep_off;
    TPROF_T(q0);
    if (tid < kTBlock) zero_split_tiles(tile_base, ntiles, cov, ucov, tile_sub);
    SlotWalk walk = slot_walk(slots, nslots, per_read != 0, kCountFold);
    // (a round's values are asked for while the round before it leaves: step 3 is the last to read them)
    uint32_t v[kBigPieces];
    bool mine = round_load<kBigPieces, true>(walk, vals, lane, v);
    TPROF_T(q1);
    if (wave < 8u) TPROF_ADD(0, q0, q1);
    while (true) {
        TPROF_T(p0);
        if (lane == 0u) s_more[wave] = mine ? 1u : 0u;
#pragma unroll
        for (int u = 0; u < kE; ++u) s_tab[u * kBigBlock + tid] = 0u;
        __syncthreads();
        uint32_t any = 0;
#pragma unroll
        for (int w = 0; w < kBigBlock / 64; ++w) any |= s_more[w];
        if (!any) break;
        TPROF_T(p1);
        if (wave < 8u) TPROF_ADD(1, p0, p1);
        // 1. the round's histogram
#pragma unroll
        for (int k = 0; k < kBigPieces; ++k) atomicAdd(&s_tab[v[k] != 0xffffffffu ? tile_of(v[k]) : no_tile], 1u);
        __syncthreads();
        TPROF_T(p2);
        if (wave < 8u) TPROF_ADD(2, p1, p2);
        // 2. my tiles: their stretch of the bucket for this round (one returning atomic per touched tile, all of a
        // thread's under way during the scan), their offsets in the stage
        // (Nothing but the atomic inside its condition, and no use of the result before step 4: "h ? atomicAdd() + base : 0"
        // is a branch with a wait at its end per tile -- the thread's reservations one round trip after the other.)
        uint32_t got[kE], at[kE], sum = 0;
        {
            uint32_t h[kE];
#pragma unroll
            for (int u = 0; u < kE; ++u) {
                const uint32_t i = u * kBigBlock + tid;
                at[u] = rep_base[i < ntiles ? i : 0u];
            }
#pragma unroll
            for (int u = 0; u < kE; ++u) {
                const uint32_t i = u * kBigBlock + tid;
                h[u] = s_tab[i];
                got[u] = 0u;
                if (h[u] && i < ntiles) got[u] = atomicAdd(&tile_cursor[i], h[u]);
                sum += h[u];
            }
            uint32_t inc = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t a = __shfl_up(inc, o, 64);
                if (lane >= static_cast<uint32_t>(o)) inc += a;
            }
            if (lane == 63u) s_wtot[wave] = inc;
            sum = inc - sum;  // (what the wave's lanes before mine hold)
            __syncthreads();
            uint32_t run = sum;
#pragma unroll
            for (int w = 0; w < kBigBlock / 64; ++w)
                if (w < static_cast<int>(wave)) run += s_wtot[w];
            sum = run;  // my first tile's offset in the stage
#pragma unroll
            for (int u = 0; u < kE; ++u) {
                s_tab[u * kBigBlock + tid] = run;
                run += h[u];
            }
        }
        uint32_t total = 0;
#pragma unroll
        for (int w = 0; w < kBigBlock / 64; ++w) total += s_wtot[w];
        __syncthreads();
        TPROF_T(p3);
        if (wave < 8u) TPROF_ADD(3, p2, p3);
        // 3. the values into the stage, tile after tile: a value's place is what its tile's offset was when it came by
        // (the order inside a tile's run is of no consequence to a histogram)
#pragma unroll
        for (int k0 = 0; k0 < kBigPieces; k0 += 8) {
            uint32_t at_[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) at_[k] = atomicAdd(&s_tab[v[k0 + k] != 0xffffffffu ? tile_of(v[k0 + k]) : no_tile], 1u);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t t = tile_of(v[k0 + k]);
                s_stage[v[k0 + k] != 0xffffffffu ? at_[k] : kBigRound + lane] = (t << kEntryBits) | entry_of(v[k0 + k]);
            }
        }
        __syncthreads();
        TPROF_T(p4);
        if (wave < 8u) TPROF_ADD(4, p3, p4);
        // 4. the table turns into "where the tile's run goes, minus where it lies in the stage" (an entry holds the END
        // of its run now = the start of my next tile's)
        {
            uint32_t off = sum;
#pragma unroll
            for (int u = 0; u < kE; ++u) {
                const uint32_t end = s_tab[u * kBigBlock + tid];
                s_tab[u * kBigBlock + tid] = got[u] + at[u] - off;
                off = end;
            }
        }
        mine = round_load<kBigPieces, true>(walk, vals, lane, v);  // (behind step 4: a wait for the reservations would
                                                                    // be a wait for these loads too)
        __syncthreads();
        TPROF_T(p5);
        if (wave < 8u) TPROF_ADD(5, p4, p5);
        // 5. out: consecutive lanes, consecutive entries
#pragma unroll
        for (int k0 = 0; k0 < kBigPieces; k0 += 4) {  // (four stage reads, then four table reads, then the stores)
            if (static_cast<uint32_t>(k0) * kBigBlock >= total) break;
            uint32_t e[4], d[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) e[k] = s_stage[(k0 + k) * kBigBlock + tid];
#pragma unroll
            for (int k = 0; k < 4; ++k) d[k] = s_tab[min(e[k] >> kEntryBits, kNoTile)];  // (behind `total`: whatever the stage holds)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t j = (k0 + k) * kBigBlock + tid;
                if (j < total) bucket[d[k] + j] = static_cast<uint16_t>(e[k] & ((1u << kEntryBits) - 1u));
            }
        }
        __syncthreads();  // (the next round clears the table and refills the stage)
        TPROF_T(p6);
        if (wave < 8u) TPROF_ADD(6, p5, p6);
    }
    TPROF_T(q2);
    if (wave < 8u) TPROF_ADD(7, q0, q2);
}
constexpr size_t big_lds_bytes(int e) { return (static_cast<size_t>(e) * kBigBlock + kBigRound + 128u) * 4u; }
// (kBigE = 12, for the 12 183 tiles of config 5 -- two values per tile and round --, measured slower than the direct
// rounds there: 236 against 201 us)
constexpr int kBigE = 6;
static_assert(big_lds_bytes(kBigE) <= 160u * 1024u - 256u, "the table and the stage share a CU's LDS");
static_assert(kBigE * kBigBlock == kBigRoundTiles, "kernels.h tells the context which layouts take these rounds");
static_assert(kTileShift + 1 + 14 <= 32, "a stage entry holds the tile and the bucket entry");

// Matrix bucketing, the scatter: the workgroup that counted these slots (same grid, same slot ranges as k_tile_count)
// writes them out.  s_cur[tile] = tile_base[tile] + this row's prefix; a value's place is the returned LDS count.  One
// pass over the slots, one barrier, no global atomic.
__global__ __launch_bounds__(kTBlock * kCountFold) void k_tile_scatter_matrix(
    const uint32_t* __restrict__ vals, const uint4* __restrict__ slots, uint32_t nslots, int per_read, uint32_t ntiles,
    const uint32_t* __restrict__ tile_base, const uint32_t* __restrict__ matrix, uint32_t row_stride,
    uint16_t* __restrict__ bucket, uint32_t* __restrict__ cov, uint32_t* __restrict__ ucov, uint32_t tile_sub) {
    HIP_DYNAMIC_SHARED(uint32_t, s_cur)
    const uint32_t* __restrict__ row = matrix + static_cast<size_t>(blockIdx.x) * row_stride;
    for (uint32_t i = threadIdx.x; i < ntiles; i += kTBlock * kCountFold) s_cur[i] = tile_base[i] + row[i];
    if (threadIdx.x < kTBlock) zero_split_tiles(tile_base, ntiles, cov, ucov, tile_sub);
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    SlotWalk w = slot_walk(slots, nslots, per_read != 0, kCountFold);
    while (true) {  // two pieces of 256 values per trip, their loads in flight together
        uint32_t v[8];
        bool any = false;
#pragma unroll
        for (int u = 0; u < 8; u += 4) {
            uint32_t base = 0;
            const uint32_t n = slot_next(w, &base);
            any = any || n != 0u;
            piece_load(vals, base, n, lane, v + u);
        }
        if (!any) break;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (v[u] == 0xffffffffu) continue;
            const uint32_t pos = atomicAdd(&s_cur[tile_of(v[u])], 1u);
            bucket[pos] = static_cast<uint16_t>(entry_of(v[u]));
        }
    }
}

// Exclusive scan of s[0 .. kFusedTiles) in place by the 512 threads of a workgroup (8 consecutive elements per thread,
// wave prefix by DPP-free shuffles, 8 wave totals through s_wtot); returns the grand total to every thread.
constexpr uint32_t kFusedTiles = 4096;  // slots of the fused scan: 8 per thread (kFusedScanTiles of them may hold tiles)
__device__ __forceinline__ uint32_t block_excl_scan_4096(uint32_t* s, uint32_t* s_wtot) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t v[8], sum = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        v[u] = s[tid * 8 + u];
        sum += v[u];
    }
    uint32_t inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t a = __shfl_up(inc, o, 64);
        if (lane >= static_cast<uint32_t>(o)) inc += a;
    }
    if (lane == 63) s_wtot[wave] = inc;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        const uint32_t t = s_wtot[w];
        if (w < static_cast<int>(wave)) before += t;
        total += t;
    }
    uint32_t run = before + inc - sum;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        s[tid * 8 + u] = run;
        run += v[u];
    }
    __syncthreads();
    return total;
}

// k_tile_scan folded into the one-level bucketing kernel (<= 4096 tiles): every workgroup sums the copies of the tile
// counts and scans them itself (78 KB of L2 reads and ~2 us per workgroup) instead of waiting for a single-workgroup
// kernel (10 us per launch, twice per file); workgroup 0 also cuts the buckets into k_tile_hist's work items and lists
// the split tiles.  tile_cursor must be zero on entry (k_zero).
__global__ __launch_bounds__(kTBlock) void k_tile_scatter_fused(const uint32_t* __restrict__ vals,
                                                                const uint4* __restrict__ slots, uint32_t nslots,
                                                                int per_read, uint32_t* __restrict__ counters,
                                                                uint32_t ntiles, const uint32_t* __restrict__ tile_count_all,
                                                                uint32_t* __restrict__ tile_cursor_all,
                                                                uint16_t* __restrict__ bucket, uint32_t* __restrict__ cov,
                                                                uint32_t* __restrict__ ucov, uint32_t rep_stride,
                                                                uint4* __restrict__ items, uint32_t* __restrict__ split_tiles,
                                                                const uint4* __restrict__ part, uint32_t nparts,
                                                                uint32_t* __restrict__ tail, uint32_t tile_sub) {
    // 80 KiB of LDS to the byte, so that two workgroups share a CU: the three small arrays live in the tail of s_mine,
    // whose entries from ntiles on are never read (the launcher admits at most kFusedScanTiles = 4064 tiles)
    __shared__ uint32_t s_stage[kRoundCap];          // the ordered round's stage; before the rounds: the tile totals and
                                                     // their exclusive scan (= tile_base), kFusedTiles + 1 words
    __shared__ uint32_t s_mine[kFusedTiles];         // start of this workgroup's copy inside every tile's bucket
    __shared__ uint32_t s_cnt[kFusedTiles];          // (scatter_round_ordered) / pieces per tile (workgroup 0's work items)
    __shared__ uint32_t s_loff[kFusedTiles];
    static_assert(kRoundCap >= kFusedTiles + 1, "the stage doubles as the tile_base table");
    static_assert(kFusedScanTiles + 32 <= kFusedTiles, "room for the small arrays behind the last tile");
    uint32_t* const s_wtot = s_mine + kFusedTiles - 32;       // [8]
    uint32_t* const s_more = s_mine + kFusedTiles - 24;       // [kTBlock / 64]
    uint32_t& s_nsplit = s_mine[kFusedTiles - 16];
    uint32_t* const s_base = s_stage;
    const uint32_t tid = threadIdx.x;
    const uint32_t lid = xcd_logical_id();
    const uint32_t my_rep = (lid / kCountFold) % kTileReps;  // (the copy the counting workgroup of these slots added to)
    TPROF_T(q0);
    if (tid == 0) s_nsplit = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {  // coalesced, the copies' loads independent of each other
        const uint32_t i = q * kTBlock + tid;
        uint32_t c = 0, mine = 0;
        if (i < ntiles) {
            uint32_t v[kTileReps];
#pragma unroll
            for (uint32_t rep = 0; rep < kTileReps; ++rep) v[rep] = tile_count_all[static_cast<size_t>(rep) * rep_stride + i];
#pragma unroll
            for (uint32_t rep = 0; rep < kTileReps; ++rep) {
                if (rep < my_rep) mine += v[rep];
                c += v[rep];
            }
        }
        s_base[i] = c;
        if (i < ntiles) s_mine[i] = mine;
    }
    __syncthreads();
    const uint32_t total = block_excl_scan_4096(s_base, s_wtot);
    if (tid == 0) s_base[kFusedTiles] = total;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const uint32_t i = q * kTBlock + tid;
        if (i < ntiles) s_mine[i] += s_base[i];
    }
    __syncthreads();
    // (tiles beyond ntiles have count 0: their base is the grand total, so base[i + 1] - base[i] is right for every i < ntiles)
    if (blockIdx.x == 0) {  // work items of k_tile_hist: <= kTileSub entries of one tile each; an empty tile still gets one
        if (part) publish_totals(part, nparts, counters, tail, s_loff);  // (s_loff is free until the first round)
        uint32_t* s_piece = s_cnt;  // (free until the first round) pieces per tile, then their exclusive scan
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint32_t i = q * kTBlock + tid;
            const uint32_t c = i < ntiles ? s_base[i + 1] - s_base[i] : 0u;
            s_piece[i] = i < ntiles ? (c ? (c + tile_sub - 1) / tile_sub : 1u) : 0u;
        }
        __syncthreads();
        const uint32_t n_items = block_excl_scan_4096(s_piece, s_wtot);
        for (uint32_t i = tid; i < ntiles; i += kTBlock) {
            const uint32_t b0 = s_base[i], c = s_base[i + 1] - b0;
            const uint32_t pieces = c ? (c + tile_sub - 1) / tile_sub : 1u;
            if (pieces > 1) split_tiles[atomicAdd(&s_nsplit, 1u)] = i;
            for (uint32_t k = 0; k < pieces; ++k) {
                const uint32_t lo = b0 + k * tile_sub;
                items[s_piece[i] + k] = make_uint4(i, lo, min(lo + tile_sub, b0 + c), pieces);
            }
        }
        __syncthreads();
        if (tid == 0) {
            counters[CNT_ITEMS] = n_items;
            counters[CNT_ITEMS2] = 0;
            counters[CNT_SPLIT] = s_nsplit;
        }
    }
    zero_split_tiles(s_base, ntiles, cov, ucov, tile_sub);
    __syncthreads();  // the stage (s_base) and s_cnt are handed to the rounds
    uint32_t* __restrict__ tile_cursor = tile_cursor_all + static_cast<size_t>(my_rep) * rep_stride;
    SlotWalk walk = slot_walk(slots, nslots, per_read != 0, 1, lid);
    TPROF_T(q1);
    TPROF_ADD(0, q0, q1);
    bucketing_rounds(walk, vals, s_more, [&](const uint32_t (&v)[kRoundPieces]) {
        scatter_round_ordered(v, ntiles, s_mine, tile_cursor, bucket, s_cnt, s_loff, s_stage, s_wtot);
    });
    TPROF_T(q2);
    TPROF_ADD(7, q0, q2);
}

// two levels, level 1: values go to their SUPER tile (64 tiles = 512 K bins) as 32-bit words (19-bit bin-in-super |
// unique bit): few destinations per workgroup, long runs
__global__ __launch_bounds__(kTBlock) void k_part_super(const uint32_t* __restrict__ vals, const uint4* __restrict__ slots,
                                                        uint32_t nslots, int per_read, uint32_t ntiles,
                                                        const uint32_t* __restrict__ tile_base,
                                                        uint32_t* __restrict__ sup_cursor, uint32_t* __restrict__ mid,
                                                        uint32_t* __restrict__ cov, uint32_t* __restrict__ ucov,
                                                        uint32_t tile_sub) {
    __shared__ uint32_t s_cur[kMaxSuper];
    __shared__ uint32_t s_more[kTBlock / 64];
    const uint32_t nsup = (ntiles + kSuperTiles - 1) / kSuperTiles;
    zero_split_tiles(tile_base, ntiles, cov, ucov, tile_sub);
    SlotWalk walk = slot_walk(slots, nslots, per_read != 0);
    bucketing_rounds(walk, vals, s_more, [&](const uint32_t (&v)[kRoundPieces]) {
        for (uint32_t i = threadIdx.x; i < nsup; i += kTBlock) s_cur[i] = 0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kRoundPieces; ++k)
            if (v[k] != 0xffffffffu) atomicAdd(&s_cur[(v[k] & 0x7fffffffu) >> kSuperShift], 1u);
        __syncthreads();
        // reserve [base, base + h) of each non-empty super tile's range for this round; s_cur becomes the write cursor
        for (uint32_t i = threadIdx.x; i < nsup; i += kTBlock) {
            const uint32_t h = s_cur[i];
            if (h) s_cur[i] = tile_base[i * kSuperTiles] + atomicAdd(&sup_cursor[i], h);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kRoundPieces; ++k) {
            if (v[k] == 0xffffffffu) continue;
            const uint32_t g = v[k] & 0x7fffffffu;
            const uint32_t pos = atomicAdd(&s_cur[g >> kSuperShift], 1u);
            mid[pos] = (g & kSuperMask) | ((v[k] >> 31) << kSuperShift);
        }
        __syncthreads();
    });
}

__global__ __launch_bounds__(kTBlock) void k_part_tile(const uint32_t* __restrict__ mid, const uint4* __restrict__ items2,
                                                       const uint32_t* __restrict__ counters,
                                                       const uint32_t* __restrict__ tile_base,
                                                       uint32_t* __restrict__ tile_cursor, uint32_t ntiles,
                                                       uint16_t* __restrict__ bucket) {
    __shared__ uint32_t s_cur[kSuperTiles];
    if (blockIdx.x >= counters[CNT_ITEMS2]) return;
    const uint4 it = items2[blockIdx.x];
    const uint32_t sup = it.x, lo = it.y, hi = it.z;
    const uint32_t tile0 = sup * kSuperTiles;
    if (threadIdx.x < kSuperTiles) s_cur[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t e0 = lo; e0 < hi; e0 += 4 * kTBlock) {
        uint32_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            uint32_t e = e0 + u * kTBlock + threadIdx.x;
            v[u] = (e < hi) ? mid[e] : 0xffffffffu;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (v[u] != 0xffffffffu) atomicAdd(&s_cur[(v[u] & kSuperMask) >> kTileShift], 1u);
    }
    __syncthreads();
    if (threadIdx.x < kSuperTiles) {
        const uint32_t h = s_cur[threadIdx.x], tile = tile0 + threadIdx.x;
        if (h && tile < ntiles) s_cur[threadIdx.x] = tile_base[tile] + atomicAdd(&tile_cursor[tile], h);
    }
    __syncthreads();
    for (uint32_t e0 = lo; e0 < hi; e0 += 4 * kTBlock) {
        uint32_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            uint32_t e = e0 + u * kTBlock + threadIdx.x;
            v[u] = (e < hi) ? mid[e] : 0xffffffffu;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (v[u] == 0xffffffffu) continue;
            const uint32_t local = v[u] & kSuperMask;
            uint32_t pos = atomicAdd(&s_cur[local >> kTileShift], 1u);
            bucket[pos] = static_cast<uint16_t>((local & kTileMask) | (((v[u] >> kSuperShift) & 1u) ? kTileBins : 0u));
        }
    }
}

// Per-reference statistics of the tile held in LDS (src/reference_contig.hpp:84-91 and the sums reads_count /
// uniq_reads_count are): every wave owns one eighth of the tile, walks the references overlapping it (their offsets
// are staged in LDS) and adds its partial results to stats[ref * 4 + {0: sum a, 1: non-zero a, 2: sum b, 3: non-zero b}].
// A few dozen atomics per tile replace a kernel that streamed both arrays again (31 + 22 us at config 2).
// 'bin != 0' bitmap words of the tile held in LDS (multi-GPU coverage summary): one ballot per 64 bins.  The bitmap
// region is laid out for the exchange: the tiles are cut into slices of `tps` tiles (one slice per rank for the
// all-to-all form, a single slice otherwise) and slice j holds [array 0 bits | array 1 bits] of its tiles.
// A tile in LDS: one 32-bit word per bin (k_pack, which holds finished tiles), or PACKED, two 16-bit counts per word --
// bin i in half (i >> 12) of word (i & 4095), so that four consecutive words hold four consecutive bins twice over.
// k_tile_hist<true> (cov + uniq_cov) counts packed: a work item has at most kTileSub (< 65536) entries, no count can
// carry into its neighbour, and at 32 KiB per workgroup a CU holds four workgroups instead of two (the kernel is a
// chain of latencies: work item -> bucket -> LDS -> statistics -> stores).  The single-array instance has its four
// workgroups per CU with plain words and keeps them (packing only adds the unpacking: config 3, 77 -> 87 us).
constexpr uint32_t kPackWords = kTileBins / 2;
static_assert(kTileSub < 65536, "packed 16-bit counts in k_tile_hist");
template <bool kPacked>
__device__ __forceinline__ uint4 tile_load4(const uint32_t* s, uint32_t bin) {  // bins bin .. bin + 3 (bin % 4 == 0)
    if (!kPacked) return *reinterpret_cast<const uint4*>(s + bin);
    const uint4 w = *reinterpret_cast<const uint4*>(s + (bin & (kPackWords - 1)));
    return bin & kPackWords ? make_uint4(w.x >> 16, w.y >> 16, w.z >> 16, w.w >> 16)
                            : make_uint4(w.x & 0xffffu, w.y & 0xffffu, w.z & 0xffffu, w.w & 0xffffu);
}
template <bool kPacked>
__device__ __forceinline__ uint32_t tile_load1(const uint32_t* s, uint32_t bin) {
    if (!kPacked) return s[bin];
    const uint32_t w = s[bin & (kPackWords - 1)];
    return bin & kPackWords ? w >> 16 : w & 0xffffu;
}
// one more in bin `bin` (13 bits) of a tile
template <bool kPacked>
__device__ __forceinline__ void tile_count(uint32_t* s, uint32_t bin) {
    if (kPacked)
        atomicAdd(&s[bin & (kPackWords - 1)], bin & kPackWords ? 0x10000u : 1u);
    else
        atomicAdd(&s[bin], 1u);
}

template <bool kPacked, uint32_t kWaves>
__device__ __forceinline__ void tile_nonzero_bits(const uint32_t* s_a, uint32_t tile, const BitsLayout& bl, uint32_t array) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint64_t* dst = bl.base + (static_cast<uint64_t>(tile / bl.tps) * 2 + array) * bl.slice_w64 +
                    static_cast<uint64_t>(tile % bl.tps) * (kTileBins / 64);
    constexpr uint32_t kPart = kTileBins / kWaves;  // every wave its part of the tile
#pragma unroll
    for (uint32_t j = 0; j < kPart / 64; ++j) {
        const uint32_t i = wave * kPart + j * 64;
        const uint64_t m = __ballot(tile_load1<kPacked>(s_a, i + lane) != 0u);
        if (lane == 0) dst[i >> 6] = m;
    }
}

constexpr uint32_t kStatRefs = 128;  // reference offsets staged in LDS per tile (more: read from global memory)

template <bool kTwo, bool kPacked, uint32_t kWaves>
__device__ __forceinline__ void tile_ref_stats(const uint32_t* s_a, const uint32_t* s_b, uint32_t tile,
                                               const uint32_t* s_off, uint32_t r0,
                                               const uint32_t* __restrict__ bin_off, uint32_t n_refs,
                                               uint32_t* __restrict__ stats, bool want_sum, bool want_nz) {
    // every wave owns its part of the tile (1024 or 2048 bins) and walks the references overlapping it
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t tile0 = tile * kTileBins;
    const uint32_t t0 = tile0 + wave * (kTileBins / kWaves), t1 = t0 + kTileBins / kWaves;
    static_assert(kTileBins / kWaves <= 32768, "the two non-zero counts of a wave's part share one reduction");
    if (r0 >= n_refs) return;
    // first reference whose stretch reaches into my part: the last one starting at or before t0
    uint32_t k = 0;
    {
        const uint32_t staged = min(kStatRefs, n_refs - r0);  // s_off[0 .. staged] are valid
        uint32_t lo = 0, hi = staged;                         // invariant: s_off[lo] <= t0 (s_off[0] <= tile0 <= t0)
        while (lo + 1 < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s_off[mid] <= t0)
                lo = mid;
            else
                hi = mid;
        }
        k = lo;
        if (k + 1 == staged && staged == kStatRefs)  // more references than staged: continue in global memory
            while (r0 + k + 1 < n_refs && bin_off[r0 + k + 1] <= t0) ++k;
    }
    // statistics of reference r, whose bins [s, e) overlap my part
    auto one_ref = [&](uint32_t r, uint32_t s, uint32_t e) {
        const uint32_t a0 = max(s, t0) - tile0, a1 = min(e, t1) - tile0;  // multiples of 4 (offsets are 16-byte aligned)
        uint32_t sa = 0, za = 0, sb = 0, zb = 0;
        for (uint32_t i = a0 + lane * 4; i < a1; i += 256) {
            const uint4 v = tile_load4<kPacked>(s_a, i);
            sa += v.x + v.y + v.z + v.w;
            za += (v.x != 0) + (v.y != 0) + (v.z != 0) + (v.w != 0);
            if (kTwo) {
                const uint4 w = tile_load4<kPacked>(s_b, i);
                sb += w.x + w.y + w.z + w.w;
                zb += (w.x != 0) + (w.y != 0) + (w.z != 0) + (w.w != 0);
            }
        }
        sa = wave_sum_dpp(sa);
        za = wave_sum_dpp(kTwo ? za | (zb << 16) : za);  // non-zero counts are at most the part's bins each: one reduction for both
        if (kTwo) {
            sb = wave_sum_dpp(sb);
            zb = za >> 16;
            za &= 0xffffu;
        }
        if (lane == 0) {
            uint32_t* o4 = stats + static_cast<size_t>(r) * 4;
            if (want_sum && sa) atomicAdd(o4 + 0, sa);
            if (want_nz && za) atomicAdd(o4 + 1, za);
            if (kTwo && want_sum && sb) atomicAdd(o4 + 2, sb);
            if (kTwo && want_nz && zb) atomicAdd(o4 + 3, zb);
        }
    };
    // Two loops, one over the offsets staged in LDS and one (rare: more than kStatRefs references in a tile) over global
    // memory: `k < kStatRefs ? s_off[k] : bin_off[r]` in ONE loop selects between an LDS and a global pointer and
    // compiles to flat loads on the critical path of every iteration.
    const uint32_t staged = min(kStatRefs, n_refs - r0);
    for (; k < staged; ++k) {
        const uint32_t s = s_off[k];
        if (s >= t1) return;
        const uint32_t e = s_off[k + 1];
        if (e > t0) one_ref(r0 + k, s, e);
    }
    for (; r0 + k < n_refs; ++k) {
        const uint32_t s = bin_off[r0 + k];
        if (s >= t1) break;
        const uint32_t e = bin_off[r0 + k + 1];
        if (e > t0) one_ref(r0 + k, s, e);
    }
}

// kPacked = false with two arrays: the WIDE form for layouts whose tiles hold far more than kTileSub entries each (1 B
// records on 20 k references: 63 k per tile): 32-bit counts, 64 KB of LDS, work items of up to kTileSubWide entries -- a
// tile is then ONE item again instead of four pieces that each add 16 K words to global memory with atomics.
// Workgroup size and occupancy: LDS decides how many workgroups a CU holds (160 KB: 32 KB -> 4, 64 KB -> 2, 128 KB -> 1);
// the wide form over the large tiles is alone on its CU and runs 16 waves to have as many loads in flight as two
// workgroups of eight (1 B records on 20 k references: 669 -> 480 us; for the 64 KB forms 16 waves measured the same as 8).
template <bool kTwo, bool kPacked>
constexpr uint32_t hist_lds_bytes() { return (kPacked ? kTileBins / 2 : kTileBins) * 4u * (kTwo ? 2u : 1u); }
template <bool kTwo, bool kPacked>
constexpr uint32_t hist_block() { return hist_lds_bytes<kTwo, kPacked>() > 64u * 1024u ? 1024u : 512u; }
template <bool kTwo, bool kPacked>
constexpr uint32_t hist_waves_per_simd() {
    constexpr uint32_t wgs = std::min<uint32_t>(4u, (160u * 1024u) / (hist_lds_bytes<kTwo, kPacked>() + 1024u));
    return wgs * (hist_block<kTwo, kPacked>() / 64u) / 4u;
}
template <bool kTwo, bool kPacked>
__global__ __launch_bounds__((hist_block<kTwo, kPacked>()), (hist_waves_per_simd<kTwo, kPacked>())) void k_tile_hist(const uint16_t* __restrict__ bucket, const uint4* __restrict__ items,
                                                   const uint32_t* __restrict__ counters, uint32_t* __restrict__ cov,
                                                   uint32_t* __restrict__ ucov, const uint32_t* __restrict__ bin_off,
                                                   uint32_t n_refs, const uint32_t* __restrict__ tile_ref0,
                                                   uint32_t* __restrict__ stats, const BitsLayout bits, uint32_t store_from) {
    constexpr uint32_t kWords = kPacked ? kPackWords : kTileBins;   // (packed: two 16-bit counts per word, tile_load4)
    constexpr uint32_t kHB = hist_block<kTwo, kPacked>();
    constexpr uint32_t kWaves = kHB / 64;
    __shared__ uint32_t s_cov[kWords];
    __shared__ uint32_t s_ucov[kTwo ? kWords : 4];
    __shared__ uint32_t s_off[kStatRefs + 1];  // bin offsets of the references overlapping this tile (and one more)
    HPROF_T(h0);
    if (blockIdx.x >= counters[CNT_ITEMS]) return;
    const uint4 it = items[blockIdx.x];
    const uint32_t tile = it.x, lo = it.y, hi = it.z;
    const bool whole = it.w == 1;
    if (whole && lo == hi) {  // a tile without entries: zeros out, nothing for any reference's statistics, no LDS at all
        if (tile >= store_from) {
            uint4* oc = reinterpret_cast<uint4*>(cov + static_cast<size_t>(tile) * kTileBins);
            uint4* ou = reinterpret_cast<uint4*>(ucov + static_cast<size_t>(tile) * kTileBins);
            const uint4 z = make_uint4(0, 0, 0, 0);
            for (uint32_t i = threadIdx.x; i < kTileBins / 4; i += kHB) {
                oc[i] = z;
                if (kTwo) ou[i] = z;
            }
        }
        if (bits.base && threadIdx.x < kTileBins / 64) {
            for (uint32_t array = 0; array < (kTwo ? 2u : 1u); ++array)
                bits.base[(static_cast<uint64_t>(tile / bits.tps) * 2 + array) * bits.slice_w64 +
                          static_cast<uint64_t>(tile % bits.tps) * (kTileBins / 64) + threadIdx.x] = 0ull;
        }
        return;
    }
    // first reference overlapping the tile (host table; n_refs for tiles behind the last reference), then its and its
    // successors' offsets: issued now, needed after the accumulation
    const uint32_t r0 = stats ? tile_ref0[tile] : n_refs;
    const bool stage_off = threadIdx.x <= kStatRefs && r0 < n_refs && r0 + threadIdx.x <= n_refs;
    const uint32_t off_reg = stage_off ? bin_off[r0 + threadIdx.x] : 0u;  // lands in LDS after the accumulation
    {
        uint4* zc = reinterpret_cast<uint4*>(s_cov);
        uint4* zu = reinterpret_cast<uint4*>(s_ucov);
        const uint4 z = make_uint4(0, 0, 0, 0);
        for (uint32_t i = threadIdx.x; i < kWords / 4; i += kHB) {
            zc[i] = z;
            if (kTwo) zu[i] = z;
        }
    }
    __syncthreads();
    HPROF_T(h2);
    HPROF_ADD(1, h0, h2);
    {
        // The item's entries, eight to a 16-byte load where the bucket is aligned (a 2-byte load per lane is one load
        // instruction per 64 entries: 9.6 M of them per launch at 1 B records); the ragged ends one by one.
        auto one = [&](uint32_t v) {
            tile_count<kPacked>(s_cov, v & kTileMask);
            if (kTwo && (v & kTileBins)) tile_count<kPacked>(s_ucov, v & kTileMask);
        };
        const uint32_t a0 = min(hi, (lo + 7u) & ~7u);          // first entry on a 16-byte boundary
        const uint32_t n8 = (hi - a0) >> 3;                     // whole groups of eight behind it
        const uint32_t a1 = a0 + (n8 << 3);
        if (lo + threadIdx.x < a0) one(bucket[lo + threadIdx.x]);
        if (a1 + threadIdx.x < hi) one(bucket[a1 + threadIdx.x]);
        const uint4* __restrict__ b8 = reinterpret_cast<const uint4*>(bucket + a0);
        for (uint32_t i0 = 0; i0 < n8; i0 += 2 * kHB) {
            uint4 q[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const uint32_t i = i0 + u * kHB + threadIdx.x;
                q[u] = i < n8 ? b8[i] : make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (i0 + u * kHB + threadIdx.x >= n8) continue;
                const uint32_t w[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    one(w[k] & 0xffffu);
                    one(w[k] >> 16);
                }
            }
        }
    }
    HPROF_T(h3);
    HPROF_ADD(2, h2, h3);
    if (stage_off) s_off[threadIdx.x] = off_reg;
    __syncthreads();
    HPROF_T(h4);
    HPROF_ADD(3, h3, h4);
    uint32_t* gc = cov + static_cast<size_t>(tile) * kTileBins;
    uint32_t* gu = ucov + static_cast<size_t>(tile) * kTileBins;
    if (whole) {
        uint4* oc = reinterpret_cast<uint4*>(gc);
        uint4* ou = reinterpret_cast<uint4*>(gu);
        const uint4* sc = reinterpret_cast<const uint4*>(s_cov);
        const uint4* su = reinterpret_cast<const uint4*>(s_ucov);
        // the statistics' atomics first: they come back from the memory side in ~2 us, and a workgroup retires only when
        // they have -- issued before the 64 KB of tile stores they are back by the time those are
        if (stats) tile_ref_stats<kTwo, kPacked, kWaves>(s_cov, s_ucov, tile, s_off, r0, bin_off, n_refs, stats, true, true);
        HPROF_T(h5);
        HPROF_ADD(4, h4, h5);
        // tiles below store_from: the caller only wants what is derived from the finished tile while it is in LDS
        // (statistics, bit maps); the coverage arrays themselves are not materialised (tiles cut into pieces still are:
        // they are summed in global memory)
        if (tile >= store_from && !kPacked) {
            for (uint32_t i = threadIdx.x; i < kTileBins / 4; i += kHB) {
                oc[i] = sc[i];
                if (kTwo) ou[i] = su[i];
            }
        } else if (tile >= store_from) {  // four packed words = bins i .. i + 3 and i + 4096 .. i + 4099
            for (uint32_t i = threadIdx.x; i < kPackWords / 4; i += kHB) {
                const uint4 w = sc[i];
                oc[i] = make_uint4(w.x & 0xffffu, w.y & 0xffffu, w.z & 0xffffu, w.w & 0xffffu);
                oc[i + kPackWords / 4] = make_uint4(w.x >> 16, w.y >> 16, w.z >> 16, w.w >> 16);
                if (kTwo) {
                    const uint4 x = su[i];
                    ou[i] = make_uint4(x.x & 0xffffu, x.y & 0xffffu, x.z & 0xffffu, x.w & 0xffffu);
                    ou[i + kPackWords / 4] = make_uint4(x.x >> 16, x.y >> 16, x.z >> 16, x.w >> 16);
                }
            }
        }
        if (bits.base) {
            tile_nonzero_bits<kPacked, kWaves>(s_cov, tile, bits, 0);
            if (kTwo) tile_nonzero_bits<kPacked, kWaves>(s_ucov, tile, bits, 1);
        }
        HPROF_T(h6);
        HPROF_ADD(5, h5, h6);
        HPROF_ADD(7, h0, h6);
        return;
    }
    for (uint32_t i = threadIdx.x; !kPacked && i < kTileBins; i += kHB) {
        const uint32_t a = s_cov[i];
        if (a) atomicAdd(&gc[i], a);
        if (kTwo) {
            const uint32_t b2 = s_ucov[i];
            if (b2) atomicAdd(&gu[i], b2);
        }
    }
    for (uint32_t i = threadIdx.x; kPacked && i < kPackWords; i += kHB) {
        const uint32_t a = s_cov[i];
        if (a & 0xffffu) atomicAdd(&gc[i], a & 0xffffu);
        if (a >> 16) atomicAdd(&gc[i + kPackWords], a >> 16);
        if (kTwo) {
            const uint32_t b2 = s_ucov[i];
            if (b2 & 0xffffu) atomicAdd(&gu[i], b2 & 0xffffu);
            if (b2 >> 16) atomicAdd(&gu[i + kPackWords], b2 >> 16);
        }
    }
    // a tile cut into pieces: the sums are additive over the pieces; the non-zero counts need the finished tile and are
    // added by k_pack, the next kernel on the stream
    if (stats) tile_ref_stats<kTwo, kPacked, kWaves>(s_cov, s_ucov, tile, s_off, r0, bin_off, n_refs, stats, true, false);
}

// small arrays gathered behind the statistics so that ONE copy brings everything to the host; the same launch finishes
// the non-zero bin counts of the tiles that k_tile_hist accumulated in pieces (it could only add their sums)
template <bool kTwo>
__global__ __launch_bounds__(512) void k_pack(uint32_t* __restrict__ dst, const PackArgs pack,
                                              const uint32_t* __restrict__ split_tiles,
                                              const uint32_t* __restrict__ counters, const uint32_t* __restrict__ a,
                                              const uint32_t* __restrict__ b, const uint32_t* __restrict__ bin_off,
                                              uint32_t n_refs, const uint32_t* __restrict__ tile_ref0,
                                              uint32_t* __restrict__ stats, const BitsLayout bits) {
    __shared__ uint32_t s_a[kTileBins];
    __shared__ uint32_t s_b[kTwo ? kTileBins : 4];
    __shared__ uint32_t s_off[kStatRefs + 1];
    {
        const uint32_t gid = blockIdx.x * 512 + threadIdx.x, gsz = gridDim.x * 512;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            for (uint32_t i = gid; i < pack.n[k]; i += gsz) dst[i] = packed_word(pack, k, i);
            dst += pack.n[k];
        }
    }
    if (!stats) return;
    const uint32_t nsplit = counters[CNT_SPLIT];
    for (uint32_t q = blockIdx.x; q < nsplit; q += gridDim.x) {
        const uint32_t tile = split_tiles[q];
        const uint32_t r0 = tile_ref0[tile];
        __syncthreads();  // LDS of the previous tile is no longer read
        if (threadIdx.x <= kStatRefs && r0 < n_refs && r0 + threadIdx.x <= n_refs) s_off[threadIdx.x] = bin_off[r0 + threadIdx.x];
        const uint4* ga = reinterpret_cast<const uint4*>(a + static_cast<size_t>(tile) * kTileBins);
        const uint4* gb = reinterpret_cast<const uint4*>(b + static_cast<size_t>(tile) * kTileBins);
        for (uint32_t i = threadIdx.x; i < kTileBins / 4; i += 512) {
            reinterpret_cast<uint4*>(s_a)[i] = ga[i];
            if (kTwo) reinterpret_cast<uint4*>(s_b)[i] = gb[i];
        }
        __syncthreads();
        tile_ref_stats<kTwo, false, 8>(s_a, s_b, tile, s_off, r0, bin_off, n_refs, stats, false, true);
        if (bits.base) {
            tile_nonzero_bits<false, 8>(s_a, tile, bits, 0);
            if (kTwo) tile_nonzero_bits<false, 8>(s_b, tile, bits, 1);
        }
    }
}

void launch_pack(hipStream_t st, uint32_t* dst, const PackArgs& pack, const uint32_t* split_tiles, const uint32_t* counters,
                 const uint32_t* a, const uint32_t* b, const uint32_t* bin_off, uint32_t n_refs, const uint32_t* tile_ref0,
                 uint32_t* stats, const BitsLayout& bits) {
    const uint32_t blocks = 256;
    if (b)
        hipLaunchKernelGGL(k_pack<true>, dim3(blocks), dim3(512), 0, st, dst, pack, split_tiles, counters, a, b, bin_off,
                           n_refs, tile_ref0, stats, bits);
    else
        hipLaunchKernelGGL(k_pack<false>, dim3(blocks), dim3(512), 0, st, dst, pack, split_tiles, counters, a, a, bin_off,
                           n_refs, tile_ref0, stats, bits);
}

uint32_t tile_count_grid(uint32_t grid) { return (grid + kCountFold - 1) / kCountFold; }

int tile_hist_setup(uint32_t ntiles) {
    // dynamic LDS above 64 KiB has to be opted into
    size_t bytes = static_cast<size_t>(ntiles) * 4;
    if (bytes > kTileLdsMax) return -1;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile_count), hipFuncAttributeMaxDynamicSharedMemorySize,
                            static_cast<int>(bytes)) != hipSuccess)
        return -1;
    if ((ntiles + kSuperTiles - 1) / kSuperTiles > kMaxSuper) return -1;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile_scatter), hipFuncAttributeMaxDynamicSharedMemorySize,
                            static_cast<int>(bytes)) != hipSuccess)
        return -1;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile_scatter_matrix), hipFuncAttributeMaxDynamicSharedMemorySize,
                            static_cast<int>(bytes)) != hipSuccess)
        return -1;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile_scatter_big<kBigE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            static_cast<int>(big_lds_bytes(kBigE))) != hipSuccess)
        return -1;
    return 0;
}

void launch_tile_count(hipStream_t st, uint32_t grid, uint32_t ntiles, const SlotValues& in, uint4* part,
                       uint32_t* tile_count, uint32_t reps, uint32_t rep_stride, uint32_t* matrix) {
    hipLaunchKernelGGL(k_tile_count, dim3(tile_count_grid(grid)), dim3(kTBlock * kCountFold), static_cast<size_t>(ntiles) * 4, st, in.vals, in.slots,
                       in.nslots, in.per_read ? 1 : 0, ntiles, tile_count, reps, rep_stride, part, matrix);
}

void launch_matrix_prefix(hipStream_t st, uint32_t grid, uint32_t ntiles, uint32_t* matrix, uint32_t row_stride, uint32_t* total) {
    hipLaunchKernelGGL(k_matrix_prefix, dim3((ntiles + 255u) / 256u), dim3(256), 0, st, matrix, tile_count_grid(grid), ntiles,
                       row_stride, total);
}

void launch_tile_scatter_matrix(hipStream_t st, uint32_t grid, uint32_t ntiles, const SlotValues& in, const uint32_t* tile_base,
                                const uint32_t* matrix, uint32_t row_stride, uint16_t* bucket, uint32_t* cov, uint32_t* ucov,
                                uint32_t tile_sub) {
    hipLaunchKernelGGL(k_tile_scatter_matrix, dim3(tile_count_grid(grid)), dim3(kTBlock * kCountFold),
                       static_cast<size_t>(ntiles) * 4, st, in.vals, in.slots, in.nslots, in.per_read ? 1 : 0, ntiles, tile_base,
                       matrix, row_stride, bucket, cov, ucov, tile_sub);
}

void launch_tile_scan(hipStream_t st, uint32_t ntiles, uint32_t* tile_count, uint32_t* tile_base,
                      uint32_t* tile_cursor, uint4* items, uint32_t* counters, uint4* items2, uint32_t* sup_cursor,
                      uint32_t* split_tiles, uint32_t reps, uint32_t rep_stride, bool two_level, const Totals& tot,
                      uint32_t tile_sub) {
    hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, st, tile_count, ntiles, tile_base, tile_cursor, items, counters,
                       items2, sup_cursor, split_tiles, reps, rep_stride, two_level ? 1 : 0, tot.part, tot.nparts, tot.tail,
                       tile_sub);
}

uint32_t part_items_upper(uint32_t ntiles, uint32_t n_upper) {
    return (ntiles + kSuperTiles - 1) / kSuperTiles + n_upper / kPartSub + 1;
}

void launch_tile_scatter(hipStream_t st, uint32_t grid, uint32_t ntiles, uint32_t n_upper, const SlotValues& in,
                         const uint32_t* counters, const uint32_t* tile_base, uint32_t* tile_cursor, uint32_t* sup_cursor,
                         const uint4* items2, uint32_t* mid, uint16_t* bucket, uint32_t* cov, uint32_t* ucov, bool two_level,
                         const uint32_t* rep_base, uint32_t reps, uint32_t rep_stride, uint32_t tile_sub) {
    long big = 1;
    (void)forced("scatter_big", &big);  // (SLIMM_FORCE, force.h: tests) 0: the direct rounds
    const bool big_rounds = big != 0;
    // (the copies' workgroups are the count's: tile_count_grid(grid) of them)
    if (!two_level && big_rounds && reps > 1 && ntiles <= static_cast<uint32_t>(kBigE) * kBigBlock) {
        hipLaunchKernelGGL(k_tile_scatter_big<kBigE>, dim3(tile_count_grid(grid)), dim3(kBigBlock), big_lds_bytes(kBigE), st,
                           in.vals, in.slots, in.nslots, in.per_read ? 1 : 0, ntiles, tile_base, tile_cursor, bucket, cov,
                           ucov, rep_base, reps, rep_stride, tile_sub);
        return;
    }
    if (!two_level) {
        const size_t lds = static_cast<size_t>(ntiles) * 4;
        hipLaunchKernelGGL(k_tile_scatter, dim3(grid), dim3(kTBlock), lds, st, in.vals, in.slots, in.nslots,
                           in.per_read ? 1 : 0, ntiles, tile_base, tile_cursor, bucket, cov, ucov, rep_base, reps, rep_stride,
                           tile_sub);
        return;
    }
    hipLaunchKernelGGL(k_part_super, dim3(grid), dim3(kTBlock), 0, st, in.vals, in.slots, in.nslots, in.per_read ? 1 : 0,
                       ntiles, tile_base, sup_cursor, mid, cov, ucov, tile_sub);
    hipLaunchKernelGGL(k_part_tile, dim3(part_items_upper(ntiles, n_upper)), dim3(kTBlock), 0, st, mid, items2, counters,
                       tile_base, tile_cursor, ntiles, bucket);
}

// count -> scan -> scatter of the one-level bucketing with the scan inside the scatter kernel (<= 4096 tiles, copies of
// the counters / cursors in use); tile_cursor must be zero.  counters is written (work item and split-tile counts).
void launch_tile_scatter_fused(hipStream_t st, uint32_t grid, uint32_t ntiles, const SlotValues& in, uint32_t* counters,
                               const uint32_t* tile_count, uint32_t* tile_cursor, uint16_t* bucket, uint32_t* cov,
                               uint32_t* ucov, uint32_t rep_stride, uint4* items, uint32_t* split_tiles,
                               const Totals& tot, uint32_t tile_sub) {
    hipLaunchKernelGGL(k_tile_scatter_fused, dim3(grid), dim3(kTBlock), 0, st, in.vals, in.slots, in.nslots,
                       in.per_read ? 1 : 0, counters, ntiles, tile_count, tile_cursor, bucket, cov, ucov, rep_stride, items,
                       split_tiles, tot.part, tot.nparts, tot.tail, tile_sub);
}

uint32_t tile_items_upper(uint32_t ntiles, uint32_t n_upper) { return ntiles + n_upper / std::min(kTileSub, kTileSubB) + 1; }

// cov + ucov (ucov != nullptr: bit kTileShift -- 13 or 14 -- of a bucket entry selects uniq_cov as well) or a single array
// stats != nullptr: also accumulate the per-reference statistics (zeroed by the caller) of the finished arrays
void launch_tile_hist(hipStream_t st, uint32_t ntiles, uint32_t n_upper, const uint16_t* bucket, const uint32_t* tile_base,
                      const uint4* items, const uint32_t* counters, uint32_t* cov, uint32_t* ucov, const uint32_t* bin_off,
                      uint32_t n_refs, const uint32_t* tile_ref0, uint32_t* stats, const BitsLayout& bits, uint32_t store_from,
                      bool wide) {
    const uint32_t grid = tile_items_upper(ntiles, n_upper);
    if (ucov && wide)
        hipLaunchKernelGGL((k_tile_hist<true, false>), dim3(grid), dim3(hist_block<true, false>()), 0, st, bucket, items, counters, cov, ucov, bin_off,
                           n_refs, tile_ref0, stats, bits, store_from);
    else if (ucov)
        hipLaunchKernelGGL((k_tile_hist<true, true>), dim3(grid), dim3(hist_block<true, true>()), 0, st, bucket, items, counters, cov, ucov, bin_off,
                           n_refs, tile_ref0, stats, bits, store_from);
    else
        hipLaunchKernelGGL((k_tile_hist<false, false>), dim3(grid), dim3(hist_block<false, false>()), 0, st, bucket, items, counters, cov, cov, bin_off,
                           n_refs, tile_ref0, stats, bits, store_from);
}

}  // namespace SLIMM_TILE_NS
}  // namespace slimm

#ifndef TPROF_SHIFT
#define TPROF_SHIFT 13  // (the cycle probes read the small-tile build unless told otherwise)
#endif
#if defined(EXP) && (EXP == 8 || EXP == 9) && SLIMM_TILE_SHIFT == TPROF_SHIFT
extern "C" int slimm_debug_prof_tiles(unsigned long long* out, int n, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(slimm::SLIMM_TILE_NS::g_prof_t), sizeof(unsigned long long) * n);
    if (reset) {
        static unsigned long long z[8 * 4096];
        (void)hipMemcpyToSymbol(HIP_SYMBOL(slimm::SLIMM_TILE_NS::g_prof_t), z, sizeof(z));
    }
    return e == hipSuccess ? 0 : -1;
}
#endif
