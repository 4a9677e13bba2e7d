// HIP kernels of the alignment-to-profile path for gfx950 (MI355X, CDNA4, wave64).
//
// The whole path is integer scatter / segmented count / small-table lookup: bounded by HBM bandwidth
// (and by the L2/fabric atomic rate where bins are hit at random), never by arithmetic -- there is no MFMA work here.
// Layout (all SoA, 32-bit indices; one context handles < 2^31 records):
//   records   key u64 | ref i32 | pos i32 | flag u16               (what the reference reads per BamAlignmentRecord)
//   compact   ident u64 | ref u32 | gbin u32 | fl u8               (mapped records only, file order kept)
//   targets   tgt_ref u32 (bit31 = first target of its read) | tgt_gbin u32   (CSR over reads: read_off u32[M+1])
//   bins      cov[Bp] | uniq_cov[Bp] | tail[64] | uniq_cov2[Bp]    (u32; each reference padded to a multiple of 4 bins)
//
// Reference semantics implemented here:
//   a3  record filter + bin + read identity      src/slimm.hpp:194-213        k_runs / k_emit on the raw records;
//                                                                              k_valid_count, k_compact feed the sort path
//   a4  first bin per distinct (read, ref)       src/read_stat.hpp:116-135    k_runs, k_emit (runs.hip)
//   a5  cov / uniq_cov histograms                src/slimm.hpp:219-257        tile_hist.hip (k_hist = fallback)
//   a7  non-zero bin counts (+ per-ref sums)     src/reference_contig.hpp:84-91   k_tile_hist (tile_hist.hip); k_ref_stats
//                                                                              after a bins merge / on the fallback path
//   a10 per-read filter, uniq_cov2               src/slimm.hpp:380-391, read_stat.hpp:98-114   k_filter_lca
//   a11 level-scan LCA                           src/slimm.hpp:516-531        k_filter_lca
//   a12 step 1: per-taxon counts + children      src/slimm.hpp:536-557        k_filter_lca
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>

#include "kernels.h"

namespace slimm {

constexpr int kBlock = 256;
constexpr int kItems = 8;
constexpr int kTile = kBlock * kItems;  // records per workgroup
constexpr int kWaves = kBlock / 64;

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// number of set bits of `mask` below this lane
__device__ __forceinline__ uint32_t mask_rank(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32),
                                     __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ bool record_is_mapped(uint16_t flag, int32_t ref) {
    return !(flag & 0x4) && ref != -1;  // src/slimm.hpp:197
}

// ---------------------------------------------------------------------------------------------------------
// k_valid_count: mapped records per tile (pass 1 of the compaction).  Flags records naming a reference >= n_refs.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_valid_count(const uint16_t* __restrict__ flag, const int32_t* __restrict__ ref,
                                                        uint32_t n, uint32_t n_refs, uint2* __restrict__ tile_cnt,
                                                        uint32_t* __restrict__ counters) {
    __shared__ uint32_t s_w[kWaves];
    const uint32_t base = blockIdx.x * kTile;
    uint32_t cnt = 0;
    bool bad = false;
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
        uint32_t i = base + k * kBlock + threadIdx.x;
        if (i < n) {
            int32_t r = ref[i];
            bool v = record_is_mapped(flag[i], r);
            if (v && static_cast<uint32_t>(r) >= n_refs) {
                bad = true;
                v = false;
            }
            cnt += v;
        }
    }
    cnt = wave_sum(cnt);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = cnt;
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(&counters[CNT_ERR], ERR_REF_RANGE);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) t += s_w[w];
        tile_cnt[blockIdx.x] = make_uint2(t, 0u);
    }
}

// ---------------------------------------------------------------------------------------------------------
// k_scan_tiles: exclusive scan of the per-tile (x, y) counts; totals go to counters[slot_x/slot_y] and to
// tile_cnt[ntiles].  Up to 16 K tiles (32 M records) ONE workgroup does it in one pass over registers; beyond that the tiles are cut
// into chunks of kScanChunk, k_scan_sums reduces every chunk, and the workgroups of k_scan_tiles start from the sum of
// the chunks before theirs (one workgroup needed 0.86 ms for the 488 K tiles of 10^9 records).
// When read_off != nullptr also writes the CSR sentinel read_off[total_x] = total_y.
// ---------------------------------------------------------------------------------------------------------
constexpr uint32_t kScanRegs = 16;    // entries per thread of the one-pass scan
constexpr uint32_t kScanChunk = 8192;  // tiles per workgroup of the chunked scan (SLIMM_SCAN_CHUNK overrides, for tests)

__global__ __launch_bounds__(1024) void k_scan_sums(const uint2* __restrict__ tile_cnt, uint32_t ntiles,
                                                    const uint32_t* __restrict__ extra, uint4* __restrict__ sums,
                                                    uint32_t chunk) {
    __shared__ uint4 s_w[16];
    const uint32_t lo = blockIdx.x * chunk, hi = min(lo + chunk, ntiles);
    uint32_t x = 0, y = 0, e = 0;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += 1024) {
        const uint2 v = tile_cnt[i];
        x += v.x;
        y += v.y;
        if (extra) e += extra[i];
    }
    x = wave_sum(x);
    y = wave_sum(y);
    e = wave_sum(e);
    if ((threadIdx.x & 63u) == 0) s_w[threadIdx.x >> 6] = make_uint4(x, y, e, 0u);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint4 t = make_uint4(0u, 0u, 0u, 0u);
        for (int w = 0; w < 16; ++w) {
            t.x += s_w[w].x;
            t.y += s_w[w].y;
            t.z += s_w[w].z;
        }
        sums[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(1024) void k_scan_tiles(uint2* __restrict__ tile_cnt, uint32_t ntiles,
                                                     uint32_t* __restrict__ counters, int slot_x, int slot_y,
                                                     uint32_t* __restrict__ read_off, const uint32_t* __restrict__ extra,
                                                     int slot_extra, uint32_t* __restrict__ tail,
                                                     const uint4* __restrict__ sums, uint32_t chunk) {
    // coalesced chunks of 1024 entries with a running carry: wave scan by shuffles, wave totals through LDS
    __shared__ uint2 s_wave[16];
    __shared__ uint32_t s_extra[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    // one workgroup: the whole range; several: my chunk, starting from the totals of the chunks before it
    const uint32_t lo = gridDim.x > 1 ? blockIdx.x * chunk : 0u;
    const uint32_t hi = gridDim.x > 1 ? min(lo + chunk, ntiles) : ntiles;
    uint2 carry = make_uint2(0u, 0u);
    uint32_t ex = 0, ex_before = 0;
    if (gridDim.x > 1) {
        for (uint32_t g = 0; g < blockIdx.x; ++g) {  // (uniform loads, at most a few dozen chunks)
            const uint4 t = sums[g];
            carry.x += t.x;
            carry.y += t.y;
            ex_before += t.z;
        }
    }
    const bool one_pass = gridDim.x == 1 && ntiles > 0 && ntiles <= 1024u * kScanRegs;
    if (one_pass) {
        // The usual case (up to 16 K tiles = 16 M records): every thread takes `per` consecutive entries into registers
        // with all its loads in flight together, scans them, and ONE workgroup scan of the thread totals follows --
        // a loop of 1024-entry chunks pays a load round trip and two barriers per chunk (10 us at 9.8 K tiles).
        const uint32_t per = (ntiles + 1023u) >> 10, base = tid * per;
        uint2 v[kScanRegs];
        uint2 mine = make_uint2(0u, 0u);
#pragma unroll
        for (uint32_t k = 0; k < kScanRegs; ++k) {
            const uint32_t i = base + k;
            const bool in = k < per && i < ntiles;
            const uint32_t ic = min(i, ntiles - 1u);  // clamped: a load behind a per-element branch is a round trip of its own
            const uint2 t = tile_cnt[ic];
            const uint32_t e = extra ? extra[ic] : 0u;
            v[k] = in ? t : make_uint2(0u, 0u);
            if (in) ex += e;
            mine.x += v[k].x;
            mine.y += v[k].y;
        }
        uint2 inc = mine;
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
        uint2 run = make_uint2(inc.x - mine.x, inc.y - mine.y);
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const uint2 t = s_wave[w];
            if (w < static_cast<int>(wave)) {
                run.x += t.x;
                run.y += t.y;
            }
            carry.x += t.x;
            carry.y += t.y;
        }
#pragma unroll
        for (uint32_t k = 0; k < kScanRegs; ++k) {
            const uint32_t i = base + k;
            if (k < per && i < ntiles) tile_cnt[i] = run;
            run.x += v[k].x;
            run.y += v[k].y;
        }
        __syncthreads();
    }
    for (uint32_t c0 = lo; c0 < hi && !one_pass; c0 += 1024) {
        const uint32_t i = c0 + tid;
        uint2 v = (i < hi) ? tile_cnt[i] : make_uint2(0u, 0u);
        if (extra && i < hi) ex += extra[i];
        uint2 inc = v;  // inclusive scan inside the wave
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
        uint2 before = carry, total = make_uint2(0u, 0u);
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const uint2 t = s_wave[w];
            if (w < static_cast<int>(wave)) {
                before.x += t.x;
                before.y += t.y;
            }
            total.x += t.x;
            total.y += t.y;
        }
        if (i < hi) tile_cnt[i] = make_uint2(before.x + inc.x - v.x, before.y + inc.y - v.y);
        carry.x += total.x;
        carry.y += total.y;
        __syncthreads();
    }
    if (blockIdx.x + 1 != gridDim.x) return;  // the workgroup of the last chunk publishes the totals
    ex = wave_sum(ex);
    if (lane == 0) s_extra[wave] = ex;
    __syncthreads();
    if (tid == 0) {
        const uint2 tot = carry;
        tile_cnt[ntiles] = tot;
        counters[slot_x] = tot.x;
        if (slot_y >= 0) counters[slot_y] = tot.y;
        if (read_off) read_off[tot.x] = tot.y;
        uint32_t v = counters[CNT_V];
        if (extra) {
            uint32_t e = ex_before;
            for (int w = 0; w < 16; ++w) e += s_extra[w];
            counters[slot_extra] = e;
            if (slot_extra == CNT_V) v = e;
        }
        if (tail) {  // the additive scalars that travel with the bins through the multi-GPU exchange
            tail[0] = v;       // hits
            tail[1] = tot.x;   // matches (reads)
            tail[2] = tot.y;   // targets
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// k_compact: pass 2 of the compaction.  Drops unmapped records, folds the mate number into the identity
// (src/slimm.hpp:204-208: qName + ".1" / ".2"), computes the bin of the record (src/slimm.hpp:200-201) and
// stores it as a global bin index (bin_off[ref] + bin).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_compact(const uint64_t* __restrict__ key, const int32_t* __restrict__ ref,
                                                    const int32_t* __restrict__ pos, const uint16_t* __restrict__ flag,
                                                    uint32_t n, uint32_t n_refs, const uint2* __restrict__ tile_off,
                                                    const uint32_t* __restrict__ ref_len,
                                                    const uint32_t* __restrict__ bin_off, uint32_t half_read,
                                                    uint32_t bin_width, uint64_t* __restrict__ ident,
                                                    uint32_t* __restrict__ cref, uint32_t* __restrict__ cgbin) {
    __shared__ uint32_t s_w[2][kWaves];
    const uint32_t base = blockIdx.x * kTile;
    const uint32_t wave = threadIdx.x >> 6;
    uint32_t running = tile_off[blockIdx.x].x;
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
        uint32_t i = base + k * kBlock + threadIdx.x;
        bool v = false;
        int32_t r = -1;
        uint16_t f = 0;
        if (i < n) {
            r = ref[i];
            f = flag[i];
            v = record_is_mapped(f, r) && static_cast<uint32_t>(r) < n_refs;
        }
        uint64_t m = __ballot(v);
        uint32_t rank = mask_rank(m);
        if ((threadIdx.x & 63) == 0) s_w[k & 1][wave] = __popcll(m);
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            uint32_t c = s_w[k & 1][w];
            if (w < static_cast<int>(wave)) before += c;
            total += c;
        }
        if (v) {
            uint32_t o = running + before + rank;
            uint32_t mate = (f & 0x40) ? 1u : ((f & 0x80) ? 2u : 0u);
            ident[o] = (key[i] << 2) | mate;
            // uint32 wrap-around of int32 + uint32, then clamp to the contig length (Q3)
            uint32_t center = min(static_cast<uint32_t>(pos[i]) + half_read, ref_len[r]);
            cref[o] = static_cast<uint32_t>(r);
            cgbin[o] = bin_off[r] + center / bin_width;
        }
        running += total;
    }
}

// ---------------------------------------------------------------------------------------------------------
// k_hist: cov[g]++ for every target; uniq_cov[g]++ when the target is the only one of its read
// (src/slimm.hpp:219-257).  reads_count / uniq_reads_count are NOT counted here: each target adds exactly one to
// both reads_count[ref] and one bin of ref, so they are the per-reference bin sums (k_tile_hist / k_ref_stats).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_hist(const uint32_t* __restrict__ tgt_ref, const uint32_t* __restrict__ tgt_gbin,
                                                 const uint32_t* __restrict__ counters, uint32_t* __restrict__ cov,
                                                 uint32_t* __restrict__ ucov) {
    const uint32_t P = counters[CNT_P];
    const uint32_t stride = gridDim.x * kBlock;
    for (uint32_t t = blockIdx.x * kBlock + threadIdx.x; t < P; t += stride) {
        uint32_t g = tgt_gbin[t];
        bool start = tgt_ref[t] >> 31;
        bool next_start = (t + 1 == P) || (tgt_ref[t + 1] >> 31);
        atomicAdd(&cov[g], 1u);
        if (start && next_start) atomicAdd(&ucov[g], 1u);
    }
}

// ---------------------------------------------------------------------------------------------------------
// k_ref_stats: one wave per reference streams its (16-byte aligned, zero padded) bin range of up to two arrays and
// reduces {sum, non-zero count} for each.  out[ref*4 + {0,1,2,3}] = sum_a, nz_a, sum_b, nz_b.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_ref_stats(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                                      const uint32_t* __restrict__ bin_off, uint32_t n_refs,
                                                      uint32_t* __restrict__ out, const PackArgs pack) {
    // ride-along: gather the small result arrays behind the statistics so that ONE copy brings everything to the host
    {
        const uint32_t gid = blockIdx.x * kBlock + threadIdx.x, gsz = gridDim.x * kBlock;
        uint32_t* dst = out + 4ull * n_refs;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            for (uint32_t i = gid; i < pack.n[k]; i += gsz) dst[i] = packed_word(pack, k, i);
            dst += pack.n[k];
        }
    }
    const uint32_t ref = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    if (ref >= n_refs) return;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t s = bin_off[ref], e = bin_off[ref + 1];
    uint32_t sa = 0, za = 0, sb = 0, zb = 0;
    for (uint32_t i = s + lane * 4; i < e; i += 256) {
        uint4 v = *reinterpret_cast<const uint4*>(a + i);
        sa += v.x + v.y + v.z + v.w;
        za += (v.x != 0) + (v.y != 0) + (v.z != 0) + (v.w != 0);
        if (b) {
            uint4 w = *reinterpret_cast<const uint4*>(b + i);
            sb += w.x + w.y + w.z + w.w;
            zb += (w.x != 0) + (w.y != 0) + (w.z != 0) + (w.w != 0);
        }
    }
    sa = wave_sum(sa);
    za = wave_sum(za);
    sb = wave_sum(sb);
    zb = wave_sum(zb);
    if (lane == 0) *reinterpret_cast<uint4*>(out + static_cast<size_t>(ref) * 4) = make_uint4(sa, za, sb, zb);
}

// ---------------------------------------------------------------------------------------------------------
// k_filter_lca: one thread per read.
//   * keep the targets whose reference is valid (read_stat::update, read_stat.hpp:98-114)
//   * exactly one left  -> uniq_cov2[g]++   (src/slimm.hpp:383-390)
//   * more than one     -> level-scan LCA over the dense lineage rows (src/slimm.hpp:516-531): the first level at
//     which all rows agree (a shared 0 "hole" agrees, Q5); if none, the level-7 entry of the LARGEST reference id --
//     the reference iterates a std::set and returns the last value it read (Q4).  Then lca_count[t]++ and
//     children[t] gets every kept reference (src/slimm.hpp:552-555): as a (ref, level) mark bit when a level agrees
//     (t is lineage[ref][level] for each of them), as a (t, ref) pair in a hash set otherwise.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}

__device__ void pair_insert(uint64_t key, uint64_t* __restrict__ tab, uint64_t* __restrict__ list, uint32_t mask,
                            uint32_t* __restrict__ counters) {
    uint32_t slot = static_cast<uint32_t>(mix64(key)) & mask;
    for (uint32_t probe = 0; probe <= mask; ++probe) {
        uint64_t cur = __hip_atomic_load(&tab[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == key) return;
        if (cur == ~0ull) {
            unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&tab[slot]), ~0ull,
                                               static_cast<unsigned long long>(key));
            if (old == ~0ull) {
                uint32_t k = atomicAdd(&counters[CNT_PAIRS], 1u);
                if (k <= (mask >> 1))
                    list[k] = key;
                else
                    atomicOr(&counters[CNT_ERR], ERR_PAIR_OVERFLOW);
                return;
            }
            if (old == key) return;
        }
        slot = (slot + 1) & mask;
    }
    atomicOr(&counters[CNT_ERR], ERR_PAIR_OVERFLOW);
}

__device__ __forceinline__ uint32_t row_eq_mask(const uint4& a, const uint4& b, const uint4& a0, const uint4& b0) {
    return (a.x == a0.x ? 1u : 0u) | (a.y == a0.y ? 2u : 0u) | (a.z == a0.z ? 4u : 0u) | (a.w == a0.w ? 8u : 0u) |
           (b.x == b0.x ? 16u : 0u) | (b.y == b0.y ? 32u : 0u) | (b.z == b0.z ? 64u : 0u) | (b.w == b0.w ? 128u : 0u);
}

__global__ __launch_bounds__(kBlock) void k_filter_lca(const uint32_t* __restrict__ read_off,
                                                       const uint32_t* __restrict__ tgt_ref,
                                                       const uint32_t* __restrict__ tgt_gbin,
                                                       uint32_t* __restrict__ counters, const uint8_t* __restrict__ valid,
                                                       const uint4* __restrict__ lin4, uint32_t* __restrict__ ucov2,
                                                       uint32_t* __restrict__ uniq_gbin,
                                                       uint32_t* __restrict__ lca_count, uint32_t* __restrict__ marks,
                                                       uint64_t* __restrict__ pair_tab, uint64_t* __restrict__ pair_list,
                                                       uint32_t pair_mask, uint32_t taxon_base) {
    const uint32_t M = counters[CNT_M];
    const uint32_t m = blockIdx.x * kBlock + threadIdx.x;
    if (m >= M) return;
    const uint32_t s = read_off[m], e = read_off[m + 1];
    const bool multi = (e - s) > 1;
    uint32_t nv = 0, first_t = 0, first_ref = 0, max_ref = 0, eq = 0xffu;
    uint4 a0 = make_uint4(0, 0, 0, 0), b0 = a0;
    // four targets per trip: their ref / valid / lineage-row loads are independent, so they overlap in flight
    for (uint32_t c = s; c < e; c += 4) {
        uint32_t r[4];
        bool ok[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = (c + k < e) ? (tgt_ref[c + k] & 0x7fffffffu) : 0xffffffffu;
#pragma unroll
        for (int k = 0; k < 4; ++k) ok[k] = (r[k] != 0xffffffffu) && valid[r[k]];
        uint4 ra[4], rb[4];
        if (multi) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (ok[k]) {
                    ra[k] = lin4[2 * r[k]];
                    rb[k] = lin4[2 * r[k] + 1];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!ok[k]) continue;
            if (nv == 0) {
                first_t = c + k;
                first_ref = r[k];
                if (multi) {
                    a0 = ra[k];
                    b0 = rb[k];
                }
            } else {
                eq &= row_eq_mask(ra[k], rb[k], a0, b0);
            }
            max_ref = max(max_ref, r[k]);
            ++nv;
        }
    }
    uint32_t sel = 0xffffffffu;  // what this read adds one to: a uniq_cov2 bin, an LCA taxon counter, or nothing
    if (nv == 1) {
        sel = tgt_gbin[first_t];
        if (ucov2) atomicAdd(&ucov2[sel], 1u);
    } else if (nv > 1) {
        const uint32_t* lin = reinterpret_cast<const uint32_t*>(lin4);
        uint32_t taxon;
        if (eq) {
            const uint32_t lv = __builtin_ctz(eq);
            taxon = lin[static_cast<size_t>(first_ref) * 8 + lv];
            for (uint32_t t = s; t < e; ++t) {
                const uint32_t r = tgt_ref[t] & 0x7fffffffu;
                if (valid[r] && !((marks[r] >> lv) & 1u)) atomicOr(&marks[r], 1u << lv);
            }
        } else {
            taxon = lin[static_cast<size_t>(max_ref) * 8 + 7];
            for (uint32_t t = s; t < e; ++t) {
                const uint32_t r = tgt_ref[t] & 0x7fffffffu;
                if (valid[r]) pair_insert((static_cast<uint64_t>(taxon) << 32) | r, pair_tab, pair_list, pair_mask, counters);
            }
        }
        if (uniq_gbin)
            sel = taxon_base + taxon;  // counted by the tile histogram: hot taxa make global atomics serialise
        else
            atomicAdd(&lca_count[taxon], 1u);
    }
    if (uniq_gbin) uniq_gbin[m] = sel;
}

// ---------------------------------------------------------------------------------------------------------
// k_filter_lca16: the same per-read work on 16-byte lineage rows: eight per-level 16-bit indices, with the run's valid
// flag in the top bit of the level-7 half-word.  One 16-byte gather per target replaces a byte gather (valid[]) plus
// two 16-byte gathers (32-byte rows): the kernel is bound by the number of distinct cache lines its gathers touch.
//   row.x = lvl0 | lvl1 << 16, row.y = lvl2 | lvl3 << 16, row.z = lvl4 | lvl5 << 16, row.w = lvl6 | (valid << 15 | lvl7) << 16
// level_taxon[level_off[lv] + idx] gives the dense taxon of a (level, index).
// ---------------------------------------------------------------------------------------------------------
#if defined(EXP) && EXP == 7
__device__ unsigned long long g_prof_f[8];
#define FPROF_T(x) const unsigned long long x = __builtin_readcyclecounter()
#define FPROF_ADD(slot, a, b) if ((threadIdx.x & 63) == 0 && (blockIdx.x & 31) == 0) atomicAdd(&g_prof_f[slot], (b) - (a))
#else
#define FPROF_T(x)
#define FPROF_ADD(slot, a, b)
#endif
struct LevelOffsets {
    uint32_t off[8];
};

__device__ __forceinline__ uint32_t row16_eq_mask(const uint4& a, const uint4& b) {
    const uint32_t dx = a.x ^ b.x, dy = a.y ^ b.y, dz = a.z ^ b.z, dw = a.w ^ b.w;
    return ((dx & 0xffffu) ? 0u : 1u) | ((dx >> 16) ? 0u : 2u) | ((dy & 0xffffu) ? 0u : 4u) | ((dy >> 16) ? 0u : 8u) |
           ((dz & 0xffffu) ? 0u : 16u) | ((dz >> 16) ? 0u : 32u) | ((dw & 0xffffu) ? 0u : 64u) | ((dw >> 16) ? 0u : 128u);
}

__device__ __forceinline__ uint32_t row16_level(const uint4& a, uint32_t lv) {
    const uint32_t w = (lv < 2) ? a.x : (lv < 4) ? a.y : (lv < 6) ? a.z : a.w;
    const uint32_t h = (lv & 1u) ? (w >> 16) : (w & 0xffffu);
    return (lv == 7) ? (h & 0x7fffu) : h;
}

__global__ __launch_bounds__(kBlock) void k_filter_lca16(const uint32_t* __restrict__ read_off,
                                                         const uint32_t* __restrict__ tgt_ref,
                                                         const uint32_t* __restrict__ tgt_gbin,
                                                         uint32_t* __restrict__ counters, const uint4* __restrict__ rows16,
                                                         const uint32_t* __restrict__ level_taxon, const LevelOffsets lo,
                                                         uint32_t* __restrict__ ucov2, uint32_t* __restrict__ uniq_gbin,
                                                         uint32_t* __restrict__ lca_count, uint32_t* __restrict__ marks_all,
                                                         uint64_t* __restrict__ pair_tab, uint64_t* __restrict__ pair_list,
                                                         uint32_t pair_mask, uint32_t taxon_base, uint32_t n_refs) {
    const uint32_t M = counters[CNT_M];
    const uint32_t m = blockIdx.x * kBlock + threadIdx.x;
    if (m >= M) return;
    // Level marks: one BYTE per (reference, level), set by a plain store.  Most reads hit the same few references; bits
    // in a word meant atomicOr (memory-side: thousands of lanes on the same words while a bit is not yet visible, and a
    // workgroup retires only when its atomics have come back) or, to avoid those, a read of the word first -- a fourth
    // dependent round trip per read.  A byte store of 1 is idempotent, needs no read, and leaves no atomic outstanding;
    // k_pack folds the 8 bytes of a reference into its mark word.
    uint8_t* __restrict__ marks = reinterpret_cast<uint8_t*>(marks_all);
    FPROF_T(q0);
    const uint32_t s = read_off[m], e = read_off[m + 1];
    uint32_t nv = 0, first_g = 0, max_ref = 0, w_max = 0, eq = 0xffu;
    uint4 a0 = make_uint4(0, 0, 0, 0);
    constexpr int kChunk = 4;  // targets per trip (8 measured slower: more predicated loads than longer reads save)
    uint32_t r0[kChunk], vmask0 = 0;  // the first kChunk targets (all of them for most reads) stay in registers
    for (uint32_t c = s; c < e; c += kChunk) {
        uint32_t r[kChunk], g[kChunk];
        uint4 row[kChunk];
        // the bins travel with the reference ids (same addresses, no extra dependent load at the end)
#pragma unroll
        for (int k = 0; k < kChunk; ++k) {  // clamped instead of bounds-tested: the loads of a trip go out together
            const uint32_t t = min(c + k, e - 1u);
            r[k] = tgt_ref[t] & 0x7fffffffu;
            g[k] = tgt_gbin[t];
        }
        // slots past the read's last target all gather row 0: lanes with the same address cost the texture addresser one
        // line together, a clamped index (the read's own last row again) one line per lane -- and with 2.6 targets per
        // read half the slots are such
#pragma unroll
        for (int k = 0; k < kChunk; ++k) row[k] = rows16[(c + k < e) ? r[k] : 0u];
#pragma unroll
        for (int k = 0; k < kChunk; ++k) {
            const bool ok = (c + k < e) && (row[k].w >> 31);  // inside the read and a valid reference
            if (c == s) {
                r0[k] = r[k];
                vmask0 |= ok ? (1u << k) : 0u;
            }
            if (!ok) continue;
            if (nv == 0) {
                first_g = g[k];
                a0 = row[k];
            } else {
                eq &= row16_eq_mask(row[k], a0);
            }
            if (r[k] >= max_ref) {
                max_ref = r[k];
                w_max = row[k].w;
            }
            ++nv;
        }
    }
    FPROF_T(q1);
    uint32_t sel = 0xffffffffu;  // what this read adds one to: a uniq_cov2 bin, an LCA taxon counter, or nothing
    if (nv == 1) {
        sel = first_g;
        if (ucov2) atomicAdd(&ucov2[sel], 1u);
    } else if (nv > 1) {
        uint32_t taxon;
        if (eq) {
            const uint32_t lv = __builtin_ctz(eq);
            taxon = level_taxon[lo.off[lv] + row16_level(a0, lv)];
#pragma unroll
            for (int k = 0; k < kChunk; ++k)
                if ((vmask0 >> k) & 1u) marks[r0[k] * 8u + lv] = 1;
            for (uint32_t t = s + kChunk; t < e; ++t) {
                const uint32_t r = tgt_ref[t] & 0x7fffffffu;
                if (rows16[r].w >> 31) marks[r * 8u + lv] = 1;
            }
        } else {
            taxon = level_taxon[lo.off[7] + ((w_max >> 16) & 0x7fffu)];
#pragma unroll
            for (int k = 0; k < kChunk; ++k)
                if ((vmask0 >> k) & 1u)
                    pair_insert((static_cast<uint64_t>(taxon) << 32) | r0[k], pair_tab, pair_list, pair_mask, counters);
            for (uint32_t t = s + kChunk; t < e; ++t) {
                const uint32_t r = tgt_ref[t] & 0x7fffffffu;
                if (rows16[r].w >> 31)
                    pair_insert((static_cast<uint64_t>(taxon) << 32) | r, pair_tab, pair_list, pair_mask, counters);
            }
        }
        if (uniq_gbin)
            sel = taxon_base + taxon;  // counted by the tile histogram: hot taxa make global atomics serialise
        else
            atomicAdd(&lca_count[taxon], 1u);
    }
    FPROF_T(q2);
    if (uniq_gbin) uniq_gbin[m] = sel;
    FPROF_T(q3);
    FPROF_ADD(0, q0, q1);
    FPROF_ADD(1, q1, q2);
    FPROF_ADD(2, q2, q3);
    FPROF_ADD(3, q0, q0 + 1);
}

// ---------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------
static inline uint32_t tiles_for(uint32_t n) { return (n + kTile - 1) / kTile; }

uint32_t num_tiles(uint32_t n) { return tiles_for(n); }

void launch_valid_count(hipStream_t st, const DeviceRecords& in, uint32_t n_refs, uint2* tile_cnt, uint32_t* counters) {
    uint32_t nt = tiles_for(in.n);
    if (nt) hipLaunchKernelGGL(k_valid_count, dim3(nt), dim3(kBlock), 0, st, in.flag, in.ref, in.n, n_refs, tile_cnt, counters);
}

void launch_scan_tiles(hipStream_t st, uint2* tile_cnt, uint32_t ntiles, uint32_t* counters, int slot_x, int slot_y,
                       uint32_t* read_off, const uint32_t* extra, int slot_extra, uint32_t* tail, uint4* sums) {
    // one workgroup up to the range of its one-pass register path; its loop over 1024-entry pieces beyond that costs
    // ~5 us per piece (93 us at 16 385 tiles, 161 us at 24 K) against ~15 us for the two launches of the chunked scan
    uint32_t grid = 1, chunk = kScanChunk, above = 1024u * kScanRegs;
    if (const char* e = getenv("SLIMM_SCAN_CHUNK")) {  // tests: chunked scan on small inputs
        chunk = std::max<uint32_t>(1u, static_cast<uint32_t>(atol(e)));
        above = chunk;
    }
    chunk = std::max(chunk, (ntiles + kScanMaxChunks - 1) / kScanMaxChunks);  // sums holds kScanMaxChunks entries
    if (sums && ntiles > above) {
        grid = (ntiles + chunk - 1) / chunk;
        hipLaunchKernelGGL(k_scan_sums, dim3(grid), dim3(1024), 0, st, tile_cnt, ntiles, extra, sums, chunk);
    }
    hipLaunchKernelGGL(k_scan_tiles, dim3(grid), dim3(1024), 0, st, tile_cnt, ntiles, counters, slot_x, slot_y, read_off, extra,
                       slot_extra, tail, sums, chunk);
}

void launch_compact(hipStream_t st, const DeviceRecords& in, uint32_t n_refs, const uint2* tile_off, const uint32_t* ref_len,
                    const uint32_t* bin_off, uint32_t half_read, uint32_t bin_width, uint64_t* ident, uint32_t* cref,
                    uint32_t* cgbin) {
    uint32_t nt = tiles_for(in.n);
    if (nt)
        hipLaunchKernelGGL(k_compact, dim3(nt), dim3(kBlock), 0, st, in.key, in.ref, in.pos, in.flag, in.n, n_refs, tile_off,
                           ref_len, bin_off, half_read, bin_width, ident, cref, cgbin);
}

void launch_hist(hipStream_t st, uint32_t n_upper, const uint32_t* tgt_ref, const uint32_t* tgt_gbin, const uint32_t* counters,
                 uint32_t* cov, uint32_t* ucov) {
    uint32_t blocks = (n_upper + kBlock - 1) / kBlock;
    if (blocks > 256u * 16u) blocks = 256u * 16u;  // grid-stride beyond 16 workgroups per CU
    if (blocks) hipLaunchKernelGGL(k_hist, dim3(blocks), dim3(kBlock), 0, st, tgt_ref, tgt_gbin, counters, cov, ucov);
}

void launch_ref_stats(hipStream_t st, const uint32_t* a, const uint32_t* b, const uint32_t* bin_off, uint32_t n_refs,
                      uint32_t* out, const PackArgs* pack) {
    uint32_t blocks = (n_refs + kWaves - 1) / kWaves;
    PackArgs none;
    if (blocks)
        hipLaunchKernelGGL(k_ref_stats, dim3(blocks), dim3(kBlock), 0, st, a, b, bin_off, n_refs, out, pack ? *pack : none);
}

// Result block to pinned host memory by a kernel: a hipMemcpyAsync goes through the DMA engine, whose start-up latency
// (11 us between the last kernel and the copy, rocprofv3 trace) is longer than this whole kernel for a few hundred KB.
__global__ __launch_bounds__(256) void k_copy_out(uint32_t* __restrict__ dst_host, const uint32_t* __restrict__ src,
                                                  uint32_t n) {
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x, gsz = gridDim.x * 256;
    const uint32_t n4 = n >> 2;
    for (uint32_t i = gid; i < n4; i += gsz)
        reinterpret_cast<uint4*>(dst_host)[i] = reinterpret_cast<const uint4*>(src)[i];
    for (uint32_t i = (n4 << 2) + gid; i < n; i += gsz) dst_host[i] = src[i];
}

void launch_copy_out(hipStream_t st, uint32_t* dst_host, const uint32_t* src, uint32_t n) {
    const uint32_t blocks = std::min<uint32_t>(64u, (n / 4 + 255u) / 256u + 1u);
    hipLaunchKernelGGL(k_copy_out, dim3(blocks), dim3(256), 0, st, dst_host, src, n);
}

// several small arrays cleared by one launch (each hipMemsetAsync is a launch of its own)
__global__ __launch_bounds__(256) void k_zero(const ZeroArgs z) {
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x, gsz = gridDim.x * 256;
#pragma unroll
    for (int k = 0; k < 5; ++k)
        for (uint32_t i = gid; i < z.n[k]; i += gsz) z.p[k][i] = 0u;
    for (uint32_t i = gid; i < z.n64; i += gsz) z.p64[i] = ~0ull;
    const uint32_t n4 = z.cp_n >> 2;
    for (uint32_t i = gid; i < n4; i += gsz) reinterpret_cast<uint4*>(z.cp_dst)[i] = reinterpret_cast<const uint4*>(z.cp_src)[i];
    for (uint32_t i = (n4 << 2) + gid; i < z.cp_n; i += gsz) z.cp_dst[i] = z.cp_src[i];
}

void launch_zero(hipStream_t st, const ZeroArgs& z) {
    uint32_t most = std::max(z.n64, z.cp_n / 4);
    for (int k = 0; k < 5; ++k) most = most > z.n[k] ? most : z.n[k];
    uint32_t blocks = (most + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks) hipLaunchKernelGGL(k_zero, dim3(blocks), dim3(256), 0, st, z);
}

// Multi-GPU: this rank's additive partial results as one int32 buffer that ranks SUM in place:
//   [R uniq_reads_count2 | T per-taxon LCA counts | 2R level marks, one 8-bit field per level | 1 number of pairs]
// (a sum over at most 255 ranks cannot carry between the fields, so "field != 0" afterwards is the OR of the marks)
__global__ __launch_bounds__(256) void k_partials_pack(const uint32_t* __restrict__ block_b, uint32_t R, uint32_t T,
                                                       uint32_t* __restrict__ out) {
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x, gsz = gridDim.x * 256;
    const uint32_t* cnt = block_b + 4ull * R;
    const uint32_t* marks = cnt + 32;
    const uint32_t* lca = marks + R;
    for (uint32_t r = gid; r < R; r += gsz) {
        out[r] = block_b[4ull * r];
        const uint32_t m = marks[r];
        out[R + T + 2 * r] = (m & 1u) | ((m & 2u) << 7) | ((m & 4u) << 14) | ((m & 8u) << 21);
        out[R + T + 2 * r + 1] = ((m >> 4) & 1u) | ((m & 32u) << 3) | ((m & 64u) << 10) | ((m & 128u) << 17);
    }
    for (uint32_t t = gid; t < T; t += gsz) out[R + t] = lca[t];
    if (gid == 0) out[3ull * R + T] = cnt[CNT_PAIRS];
}

void launch_partials_pack(hipStream_t st, const uint32_t* block_b, uint32_t R, uint32_t T, uint32_t* out) {
    const uint32_t n = R > T ? R : T;
    hipLaunchKernelGGL(k_partials_pack, dim3(std::min<uint32_t>(256u, (n + 255u) / 256u + 1u)), dim3(256), 0, st, block_b, R, T,
                       out);
}

void launch_filter_lca(hipStream_t st, uint32_t n_upper, const uint32_t* read_off, const uint32_t* tgt_ref,
                       const uint32_t* tgt_gbin, uint32_t* counters, const uint8_t* valid, const uint32_t* lin_dense,
                       uint32_t* ucov2, uint32_t* uniq_gbin, uint32_t* lca_count, uint32_t* marks, uint64_t* pair_tab,
                       uint64_t* pair_list, uint32_t pair_mask, uint32_t taxon_base) {
    uint32_t blocks = (n_upper + kBlock - 1) / kBlock;
    if (blocks)
        hipLaunchKernelGGL(k_filter_lca, dim3(blocks), dim3(kBlock), 0, st, read_off, tgt_ref, tgt_gbin, counters, valid,
                           reinterpret_cast<const uint4*>(lin_dense), ucov2, uniq_gbin, lca_count, marks, pair_tab, pair_list,
                           pair_mask, taxon_base);
}

void launch_filter_lca16(hipStream_t st, uint32_t n_upper, const uint32_t* read_off, const uint32_t* tgt_ref,
                         const uint32_t* tgt_gbin, uint32_t* counters, const void* rows16, const uint32_t* level_taxon,
                         const uint32_t* level_off, uint32_t* ucov2, uint32_t* uniq_gbin, uint32_t* lca_count,
                         uint32_t* marks, uint64_t* pair_tab, uint64_t* pair_list, uint32_t pair_mask,
                         uint32_t taxon_base, uint32_t n_refs) {
    uint32_t blocks = (n_upper + kBlock - 1) / kBlock;
    LevelOffsets lo;
    for (int i = 0; i < 8; ++i) lo.off[i] = level_off[i];
    if (blocks)
        hipLaunchKernelGGL(k_filter_lca16, dim3(blocks), dim3(kBlock), 0, st, read_off, tgt_ref, tgt_gbin, counters,
                           reinterpret_cast<const uint4*>(rows16), level_taxon, lo, ucov2, uniq_gbin, lca_count, marks,
                           pair_tab, pair_list, pair_mask, taxon_base, n_refs);
}

}  // namespace slimm

// =========================================================================================================
// Multi-GPU coverage summary (no reference counterpart: the reference is one process).
// What the cut-offs need from the other ranks is small: per-reference SUMS of cov / uniq_cov (additive) and per-reference
// counts of NON-ZERO bins, i.e. popcounts of the OR of every rank's "bin != 0" bitmap.  One bit per bin travels instead
// of one 32-bit word: 1/32 of the all-reduce volume, and an all-gather instead of a ring all-reduce.
//   summary = [ per-ref {sum_cov, -, sum_ucov, -} (4R words) | tail (16 words) | cov bits (Bp/32) | uniq_cov bits (Bp/32) ]
// =========================================================================================================
namespace slimm {

__global__ __launch_bounds__(256) void k_nonzero_bits(const uint32_t* __restrict__ bins, uint64_t n_words64,
                                                      uint64_t* __restrict__ bits) {
    // one wave per 64 bins: the ballot of (bin != 0) IS the bitmap word
    const uint64_t w = (static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t stride = (static_cast<uint64_t>(gridDim.x) * 256) >> 6;
    for (uint64_t k = w; k < n_words64; k += stride) {
        const uint64_t m = __ballot(bins[k * 64 + lane] != 0u);
        if (lane == 0) bits[k] = m;
    }
}

__device__ __forceinline__ uint32_t or_over_ranks(const uint32_t* __restrict__ gathered, uint64_t rank_stride,
                                                  uint32_t n_ranks, uint64_t word) {
    uint32_t v = 0;
    for (uint32_t k = 0; k < n_ranks; ++k) v |= gathered[k * rank_stride + word];
    return v;
}

// one wave per reference: sums over ranks, popcount of the OR-ed bitmaps over the reference's (padded) bin range
__global__ __launch_bounds__(256) void k_merge_summary(const uint32_t* __restrict__ gathered, uint64_t rank_stride,
                                                       uint32_t n_ranks, const uint32_t* __restrict__ bin_off,
                                                       uint32_t n_refs, uint64_t bits_off_cov, uint64_t bits_off_ucov,
                                                       uint32_t* __restrict__ out_stats,
                                                       const uint32_t* __restrict__ counters) {
    const uint32_t ref = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t* const out_tail = out_stats + 4ull * n_refs + 32;       // packed block A: [4R stats | 32 counters | 16 tail]
    if (ref == 0 && lane < 32) out_stats[4ull * n_refs + lane] = counters[lane];
    if (ref == 0 && lane < 16) {  // additive scalars; slot 3 holds error bits and is OR-ed
        uint32_t s = 0, o = 0;
        for (uint32_t k = 0; k < n_ranks; ++k) {
            const uint32_t v = gathered[k * rank_stride + 4ull * n_refs + lane];
            s += v;
            o |= v;
        }
        out_tail[lane] = (lane == 3) ? o : s;
    }
    if (ref >= n_refs) return;
    const uint32_t s = bin_off[ref], e = bin_off[ref + 1];
    uint32_t nz_a = 0, nz_b = 0;
    if (e > s) {
        const uint32_t w0 = s >> 5, w1 = (e - 1) >> 5;
        for (uint32_t w = w0 + lane; w <= w1; w += 64) {
            uint32_t mask = 0xffffffffu;
            if (w == w0) mask &= 0xffffffffu << (s & 31u);
            if (w == w1) mask &= 0xffffffffu >> (31u - ((e - 1) & 31u));
            nz_a += __popc(or_over_ranks(gathered, rank_stride, n_ranks, bits_off_cov + w) & mask);
            nz_b += __popc(or_over_ranks(gathered, rank_stride, n_ranks, bits_off_ucov + w) & mask);
        }
    }
    nz_a = wave_sum(nz_a);
    nz_b = wave_sum(nz_b);
    if (lane == 0) {
        uint32_t sa = 0, sb = 0;
        for (uint32_t k = 0; k < n_ranks; ++k) {
            sa += gathered[k * rank_stride + 4ull * ref + 0];
            sb += gathered[k * rank_stride + 4ull * ref + 2];
        }
        *reinterpret_cast<uint4*>(out_stats + 4ull * ref) = make_uint4(sa, nz_a, sb, nz_b);
    }
}

// one wave per reference; see launch_merge_slices in kernels.h
__global__ __launch_bounds__(256) void k_merge_slices(const uint32_t* __restrict__ recv, uint32_t n_ranks,
                                                      uint64_t slice_words, uint32_t lo_bin, uint32_t hi_bin,
                                                      const uint32_t* __restrict__ bin_off, uint32_t n_refs,
                                                      const uint32_t* __restrict__ own_stats,
                                                      const uint32_t* __restrict__ own_tail, uint32_t* __restrict__ vec) {
    const uint32_t ref = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    if (ref == 0 && lane < 16) vec[4ull * n_refs + lane] = own_tail[lane];
    if (ref >= n_refs) return;
    const uint32_t s = max(bin_off[ref], lo_bin), e = min(bin_off[ref + 1], hi_bin);
    uint32_t nz_a = 0, nz_b = 0;
    if (e > s) {
        const uint32_t w0 = s >> 5, w1 = (e - 1) >> 5, wl = lo_bin >> 5;  // lo_bin is a multiple of the tile size
        for (uint32_t w = w0 + lane; w <= w1; w += 64) {
            uint32_t mask = 0xffffffffu;
            if (w == w0) mask &= 0xffffffffu << (s & 31u);
            if (w == w1) mask &= 0xffffffffu >> (31u - ((e - 1) & 31u));
            uint32_t a = 0, b = 0;
            for (uint32_t k = 0; k < n_ranks; ++k) {
                const uint32_t* chunk = recv + static_cast<uint64_t>(k) * 2 * slice_words;
                a |= chunk[w - wl];
                b |= chunk[slice_words + (w - wl)];
            }
            nz_a += __popc(a & mask);
            nz_b += __popc(b & mask);
        }
    }
    nz_a = wave_sum(nz_a);
    nz_b = wave_sum(nz_b);
    if (lane == 0)
        *reinterpret_cast<uint4*>(vec + 4ull * ref) =
            make_uint4(own_stats[4ull * ref + 0], nz_a, own_stats[4ull * ref + 2], nz_b);
}

void launch_merge_slices(hipStream_t st, const uint32_t* recv, uint32_t n_ranks, uint64_t slice_words, uint32_t lo_bin,
                         uint32_t hi_bin, const uint32_t* bin_off, uint32_t n_refs, const uint32_t* own_stats,
                         const uint32_t* own_tail, uint32_t* vec) {
    const uint32_t blocks = (n_refs + 3) / 4;
    if (blocks)
        hipLaunchKernelGGL(k_merge_slices, dim3(blocks), dim3(256), 0, st, recv, n_ranks, slice_words, lo_bin, hi_bin, bin_off,
                           n_refs, own_stats, own_tail, vec);
}

void launch_nonzero_bits(hipStream_t st, const uint32_t* bins, uint64_t n_bins, uint32_t* bits) {
    const uint64_t n64 = n_bins / 64;  // n_bins is a multiple of the tile size
    uint32_t blocks = static_cast<uint32_t>(std::min<uint64_t>((n64 + 3) / 4, 4096));
    if (blocks)
        hipLaunchKernelGGL(k_nonzero_bits, dim3(blocks), dim3(256), 0, st, bins, n64, reinterpret_cast<uint64_t*>(bits));
}

void launch_merge_summary(hipStream_t st, const uint32_t* gathered, uint64_t rank_stride, uint32_t n_ranks,
                          const uint32_t* bin_off, uint32_t n_refs, uint64_t bits_off_cov, uint64_t bits_off_ucov,
                          uint32_t* out_stats, const uint32_t* counters) {
    uint32_t blocks = (n_refs + 3) / 4;
    if (blocks)
        hipLaunchKernelGGL(k_merge_summary, dim3(blocks), dim3(256), 0, st, gathered, rank_stride, n_ranks, bin_off, n_refs,
                           bits_off_cov, bits_off_ucov, out_stats, counters);
}

}  // namespace slimm

#if defined(EXP) && EXP == 7
extern "C" int slimm_debug_prof_filter(unsigned long long* out, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(slimm::g_prof_f), sizeof(unsigned long long) * 8);
    if (reset) {
        unsigned long long z[8] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(slimm::g_prof_f), z, sizeof(z));
    }
    return e == hipSuccess ? 0 : -1;
}
#endif
