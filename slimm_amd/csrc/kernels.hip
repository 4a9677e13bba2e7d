// HIP kernels of the alignment-to-profile path for gfx950 (MI355X, CDNA4, wave64).
//
// The whole path is integer scatter / segmented count / small-table lookup: bounded by HBM bandwidth
// (and by the L2/fabric atomic rate where bins are hit at random), never by arithmetic -- there is no MFMA work here.
// Layout (all SoA, 32-bit indices; one context handles < 2^31 records):
//   records   key u64 | ref i32 | pos i32 | flag u16               (what the reference reads per BamAlignmentRecord)
//   grouped   ident u64 | {ref u32, gbin u32}                      (record_order = ANY: mapped records, a read's adjacent)
//   targets   tgt_ref u32 (bit31 = first target of its read) | tgt_gbin u32   (CSR over reads: read_off u32[M+1])
//   bins      cov[Bp] | uniq_cov[Bp] | tail[64] | uniq_cov2[Bp]    (u32; each reference padded to a multiple of 4 bins)
//
// Reference semantics implemented here:
//   a3  record filter + bin + read identity      src/slimm.hpp:194-213        k_front on the raw records (front.hip); the first
//                                                                              pass of group_by_ident.hip for record_order = ANY
//   a4  first bin per distinct (read, ref)       src/read_stat.hpp:116-135    k_runs, k_emit (runs.hip)
//   a5  cov / uniq_cov histograms                src/slimm.hpp:219-257        tile_hist.hip (k_hist = fallback)
//   a7  non-zero bin counts (+ per-ref sums)     src/reference_contig.hpp:84-91   k_tile_hist (tile_hist.hip); k_ref_stats
//                                                                              after a bins merge / on the fallback path
//   a10 per-read filter, uniq_cov2               src/slimm.hpp:380-391, read_stat.hpp:98-114   k_filter_lca
//   a11 level-scan LCA                           src/slimm.hpp:516-531        k_filter_lca
//   a12 step 1: per-taxon counts + children      src/slimm.hpp:536-557        k_filter_lca
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>

#include "kernels.h"

namespace slimm {

constexpr int kBlock = 256;
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

// ---------------------------------------------------------------------------------------------------------
// k_hist: cov[g]++ for every target; uniq_cov[g]++ when the target is the only one of its read (bit 31 of tgt_gbin)
// (src/slimm.hpp:219-257).  The direct-atomics fallback for bin counts too large for the LDS tile tables.
// reads_count / uniq_reads_count are NOT counted here: each target adds exactly one to both reads_count[ref] and one bin
// of ref, so they are the per-reference bin sums (k_tile_hist / k_ref_stats).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_hist(const uint32_t* __restrict__ tgt_gbin, const uint4* __restrict__ slots,
                                                 uint32_t nslots, uint32_t* __restrict__ counters,
                                                 uint32_t* __restrict__ tail, uint32_t* __restrict__ cov,
                                                 uint32_t* __restrict__ ucov, int count_mapped) {
    __shared__ uint32_t s_tot[3];
    if (threadIdx.x < 3) s_tot[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * kBlock + threadIdx.x) >> 6, n_waves = (gridDim.x * kBlock) >> 6;
    uint32_t sum_f = 0, sum_h = 0, sum_v = 0;
    for (uint32_t s = wave; s < nslots; s += n_waves) {
        const uint4 d = slots[s];
        sum_f += d.y;
        sum_h += d.z;
        sum_v += d.w;
        for (uint32_t o = lane; o < d.y; o += 64u) {
            const uint32_t g = tgt_gbin[d.x + o];
            atomicAdd(&cov[g & 0x7fffffffu], 1u);
            if (g >> 31) atomicAdd(&ucov[g & 0x7fffffffu], 1u);
        }
    }
    // the stream's totals (the front end leaves them to the consumer of its slots): one set of atomics per workgroup
    if (lane == 0u) {
        atomicAdd(&s_tot[0], sum_v);
        atomicAdd(&s_tot[1], sum_h);
        atomicAdd(&s_tot[2], sum_f);
    }
    __syncthreads();
    if (threadIdx.x < 3 && s_tot[threadIdx.x]) {
        const int slot = threadIdx.x == 0 ? CNT_V : threadIdx.x == 1 ? CNT_M : CNT_P;
        if (threadIdx.x != 0 || count_mapped) atomicAdd(&counters[slot], s_tot[threadIdx.x]);
        if (tail) atomicAdd(&tail[threadIdx.x], s_tot[threadIdx.x]);
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
// Phase B + C(1) per read (k_filter_compact + k_filter_walk, further down), ONE LANE PER TARGET:
//   * keep the targets whose reference is valid (read_stat::update, read_stat.hpp:98-114)
//   * exactly one left  -> the read's selector is that target's bin: uniq_cov2[g]++   (src/slimm.hpp:383-390)
//   * more than one     -> level-scan LCA over the lineage rows (src/slimm.hpp:516-531): the first level at which all
//     rows agree (a shared 0 "hole" agrees, Q5); if none, the level-7 entry of the LARGEST reference id -- the reference
//     iterates a std::set and returns the last value it read (Q4).  The selector is the taxon (counted by the second tile
//     histogram: lca_count[t]++), and children[t] gets every kept reference (src/slimm.hpp:552-555): as a (reference,
//     level) mark when a level agrees (t is lineage[ref][level] for each of them), as a (t, ref) pair in a hash set
//     otherwise.
// What follows first are the pieces both kernels use: the pair set, the two row forms, the lane-mask helpers, and the
// window-by-window walk (filter_window / filter_q4 / filter_span: windows of up to 64 targets that are whole reads, first /
// second valid lane of every read by carry chains, reads of 64 targets and more in chunks) that k_filter_walk runs on the few
// slots k_filter_compact leaves to it.
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
                    atomicOr(&counters[CNT_ERR], static_cast<uint32_t>(ERR_PAIR_OVERFLOW));
                return;
            }
            if (old == key) return;
        }
        slot = (slot + 1) & mask;
    }
    atomicOr(&counters[CNT_ERR], static_cast<uint32_t>(ERR_PAIR_OVERFLOW));
}

// 16-byte lineage rows: eight per-level 16-bit indices, the reference's valid flag in the top bit of the level-7
// half-word; taxon_flat[(idx << 3) | lv] is the dense taxon of a (level, index)
struct Rows16 {
    const uint4* rows;
    const uint32_t* taxon_flat;
    uint32_t shift;
    struct Row {
        uint4 q;
    };
    __device__ Row load(uint32_t ref) const { return Row{rows[ref]}; }
    __device__ static bool valid(const Row& r) { return (r.q.w >> 31) != 0u; }
    __device__ static void fields(const Row& r, uint32_t (&f)[8]) {
        f[0] = r.q.x & 0xffffu;
        f[1] = r.q.x >> 16;
        f[2] = r.q.y & 0xffffu;
        f[3] = r.q.y >> 16;
        f[4] = r.q.z & 0xffffu;
        f[5] = r.q.z >> 16;
        f[6] = r.q.w & 0xffffu;
        f[7] = (r.q.w >> 16) & 0x7fffu;
    }
    // the row lane `byte_addr / 4` holds
    __device__ static Row from_lane(const Row& r, uint32_t byte_addr) {
        return Row{make_uint4(__builtin_amdgcn_ds_bpermute(byte_addr, r.q.x), __builtin_amdgcn_ds_bpermute(byte_addr, r.q.y),
                              __builtin_amdgcn_ds_bpermute(byte_addr, r.q.z), __builtin_amdgcn_ds_bpermute(byte_addr, r.q.w))};
    }
    // the levels at which two rows of VALID references differ
    __device__ static void differ(const Row& a, const Row& b, bool (&ne)[8]) {
        const uint32_t d[4] = {a.q.x ^ b.q.x, a.q.y ^ b.q.y, a.q.z ^ b.q.z, a.q.w ^ b.q.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            ne[2 * k] = static_cast<uint16_t>(d[k]) != 0;
            ne[2 * k + 1] = d[k] > 0xffffu;
        }
    }
    // the field of level lv (a per-lane value below 8)
    __device__ static uint32_t field(const Row& r, uint32_t lv) {
        const uint64_t lo = (static_cast<uint64_t>(r.q.y) << 32) | r.q.x;
        const uint64_t hi = (static_cast<uint64_t>(r.q.w & 0x7fffffffu) << 32) | r.q.z;
        return static_cast<uint32_t>((lv < 4u ? lo : hi) >> ((lv & 3u) * 16u)) & 0xffffu;
    }
    // what a selector names an LCA by: (level, index in the level) -- the second tile histogram counts per such entry
    // and k_pack sums the entries of a taxon (PackArgs::sum_k); taxon_at, the dense taxon, is looked up only where a
    // (taxon, reference) pair is stored (no level agrees: rare)
    // (index-major: the reads of a file agree mostly at ONE level, and level-major entries would put all their selectors
    // into one or two tiles of the second histogram, every piece of those adding to the same few thousand words)
    __device__ uint32_t taxon_index(uint32_t lv, uint32_t field) const { return (field << 3) | lv; }
    __device__ uint32_t taxon_at(uint32_t index) const { return taxon_flat[index]; }
};
// 32-byte rows (databases with more than 65535 taxa on one level): eight dense taxon indices, validity in a byte array
struct Rows32 {
    const uint4* lin4;
    const uint8_t* valid_of;
    struct Row {
        uint4 a, b;
        bool ok;
    };
    __device__ Row load(uint32_t ref) const { return Row{lin4[2 * ref], lin4[2 * ref + 1], valid_of[ref] != 0}; }
    __device__ static bool valid(const Row& r) { return r.ok; }
    __device__ static void fields(const Row& r, uint32_t (&f)[8]) {
        f[0] = r.a.x;
        f[1] = r.a.y;
        f[2] = r.a.z;
        f[3] = r.a.w;
        f[4] = r.b.x;
        f[5] = r.b.y;
        f[6] = r.b.z;
        f[7] = r.b.w;
    }
    __device__ static Row from_lane(const Row& r, uint32_t byte_addr) {
        Row o;
        o.a = make_uint4(__builtin_amdgcn_ds_bpermute(byte_addr, r.a.x), __builtin_amdgcn_ds_bpermute(byte_addr, r.a.y),
                         __builtin_amdgcn_ds_bpermute(byte_addr, r.a.z), __builtin_amdgcn_ds_bpermute(byte_addr, r.a.w));
        o.b = make_uint4(__builtin_amdgcn_ds_bpermute(byte_addr, r.b.x), __builtin_amdgcn_ds_bpermute(byte_addr, r.b.y),
                         __builtin_amdgcn_ds_bpermute(byte_addr, r.b.z), __builtin_amdgcn_ds_bpermute(byte_addr, r.b.w));
        o.ok = true;
        return o;
    }
    __device__ static void differ(const Row& p, const Row& q, bool (&ne)[8]) {
        ne[0] = p.a.x != q.a.x;
        ne[1] = p.a.y != q.a.y;
        ne[2] = p.a.z != q.a.z;
        ne[3] = p.a.w != q.a.w;
        ne[4] = p.b.x != q.b.x;
        ne[5] = p.b.y != q.b.y;
        ne[6] = p.b.z != q.b.z;
        ne[7] = p.b.w != q.b.w;
    }
    __device__ static uint32_t field(const Row& r, uint32_t lv) {
        uint32_t f[8];
        fields(r, f);
        uint32_t v = f[0];
#pragma unroll
        for (uint32_t l = 1; l < 8; ++l) v = lv == l ? f[l] : v;
        return v;
    }
    __device__ uint32_t taxon_index(uint32_t, uint32_t field) const { return field; }
    __device__ uint32_t taxon_at(uint32_t index) const { return index; }
};

namespace {

__device__ __forceinline__ uint64_t k_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool k_bit(uint64_t wave_uniform_mask) { return __builtin_amdgcn_inverse_ballot_w64(wave_uniform_mask); }
__device__ __forceinline__ uint64_t k_below(uint32_t n) { return n >= 64u ? ~0ull : ((1ull << n) - 1ull); }
// first bit of `bits` at or after every bit of `starts`, looking no further than the next stop (see front.hip)
__device__ __forceinline__ uint64_t k_first_after(uint64_t starts, uint64_t bits, uint64_t stops) {
    const uint64_t ones = ~bits & ~stops;
    return (ones + starts) & ~ones & bits;
}
// the bit of `heads` at or below every bit of `bits` (every bit has one): the same carry chain on the reversed masks
__device__ __forceinline__ uint64_t k_head_of(uint64_t bits, uint64_t heads) {
    const uint64_t br = __builtin_bitreverse64(bits), hr = __builtin_bitreverse64(heads);
    return __builtin_bitreverse64((~hr + br) & hr);
}

struct FilterOut {
    uint32_t* sel;                 // one selector per read: a uniq_cov2 bin, taxon_base + taxon, or 0xffffffff -- dense: the
    const uint32_t* rbase;         // reads of slot s start at rbase[s] + bbase[s >> 10] (launch_slot_read_prefix)
    const uint32_t* bbase;
    uint8_t* marks;                // one byte per (reference, level)
    uint64_t* pair_tab;
    uint64_t* pair_list;
    uint32_t pair_mask, taxon_base;
    uint32_t* counters;
};

// what is known about one read's valid targets so far (all wave-uniform)
struct ReadAcc {
    uint32_t nv;        // valid targets
    uint32_t first_g;   // bin word of the first of them
    uint32_t a0[8];     // its row
    uint32_t eq;        // bit lv: every valid row so far agrees with a0 at level lv
    uint32_t max_ref, max_f7;  // largest valid reference and its level-7 field (Q4)
};

__device__ __forceinline__ void read_clear(ReadAcc& acc) {
    acc.nv = 0;
    acc.first_g = 0;
    acc.eq = 0xffu;
    acc.max_ref = 0;
    acc.max_f7 = 0;
}

// the largest reference among the lanes `Vs` and its level-7 field (needed only when no level agrees, Q4)
__device__ __forceinline__ void read_max(ReadAcc& acc, uint64_t Vs, uint32_t ref, uint32_t f7) {
    uint64_t rest = Vs;
    while (rest) {
        const uint32_t l = static_cast<uint32_t>(__builtin_ctzll(rest));
        rest &= rest - 1ull;
        const uint32_t r = __builtin_amdgcn_readlane(ref, l);
        if (r >= acc.max_ref) {
            acc.max_ref = r;
            acc.max_f7 = __builtin_amdgcn_readlane(f7, l);
        }
    }
}

// adds the valid lanes `Vs` (of one read) of the wave's current 64 targets
__device__ __forceinline__ void read_add(ReadAcc& acc, uint64_t Vs, uint32_t g, const uint32_t (&f)[8]) {
    if (!Vs) return;
    if (acc.nv == 0u) {
        const uint32_t fl = static_cast<uint32_t>(__builtin_ctzll(Vs));
        acc.first_g = __builtin_amdgcn_readlane(g, fl);
#pragma unroll
        for (int l = 0; l < 8; ++l) acc.a0[l] = __builtin_amdgcn_readlane(f[l], fl);
    }
    acc.nv += static_cast<uint32_t>(__popcll(Vs));
#pragma unroll
    for (int l = 0; l < 8; ++l)
        if (k_ballot(f[l] != acc.a0[l]) & Vs) acc.eq &= ~(1u << l);
}

// LCA of a read with several valid targets; *lv = the agreeing level (8: none, Q4)
template <typename Rows>
__device__ __forceinline__ uint32_t read_taxon(const Rows& rows, const ReadAcc& acc, uint32_t* lv, uint32_t* index) {
    if (acc.eq) {
        *lv = static_cast<uint32_t>(__builtin_ctz(acc.eq));
        *index = rows.taxon_index(*lv, acc.a0[*lv]);
        return rows.taxon_at(*index);
    }
    *lv = 8u;
    *index = rows.taxon_index(7u, acc.max_f7);
    return rows.taxon_at(*index);
}

// children[taxon] gets the valid lanes' references: a level mark, or a (taxon, reference) pair when no level agreed
__device__ __forceinline__ void read_children(const FilterOut& out, bool mine, uint32_t ref, uint32_t lv, uint32_t taxon) {
    if (!mine) return;
    if (lv < 8u)
        out.marks[ref * kMarkBytes + lv] = 1;  // a plain, idempotent byte store (no read of the word, no atomic)
    else
        pair_insert((static_cast<uint64_t>(taxon) << 32) | ref, out.pair_tab, out.pair_list, out.pair_mask, out.counters);
}

// One window of up to 64 targets that are whole reads: lanes [0, X); w / g = the target words, row = the lineage rows
// of the lanes' references, sel_base = index of the window's first read among the selectors.
// Everything per read is lane-mask arithmetic; the reads that keep several targets (8 % at config 2, most of them at
// config 5) are worked on all at once as well: every valid lane fetches the row of its read's FIRST valid lane (a lane
// permute), per level one ballot says which lanes differ from it, and the first valid lane of each read -- its owner --
// finds the first level at which no lane between itself and the next read's head does.  The owners' taxa are ONE gather
// per window, left to the caller (Lookup) so that the gathers of a batch of windows are in flight together.
struct Lookup {
    uint64_t q4;     // first valid lanes of the reads on whose targets no level agrees (quirk Q4): left to filter_q4
};

template <typename Rows>
__device__ __forceinline__ Lookup filter_window(const Rows& rows, const FilterOut& out, uint32_t lane, uint32_t X, uint32_t w,
                                                uint32_t g, const typename Rows::Row& row, uint64_t valid_lanes,
                                                uint32_t sel_base) {
    const uint64_t PR = k_below(X);
    const uint64_t H = k_ballot((w >> 31) != 0u) & PR;  // (bit 0 is set: a window starts at a read's first target)
    const uint32_t ref = w & 0x7fffffffu;
    const uint64_t VB = valid_lanes & PR;               // valid_lanes = k_ballot(Rows::valid(row)), taken by the caller
    const uint64_t stops = (H >> 1) | (1ull << (X - 1u));
    const uint64_t FV = k_first_after(H, VB, stops);              // first valid target of every read
    // Owners: the first valid target of the reads that keep several.  Every LATER valid target points at one: the
    // nearest first-valid lane below a valid lane is its own read's (another read's would lie above it or below its
    // head), so "the head at or below each bit" with FV as the heads finds them in one carry chain.
    const uint64_t OW = k_head_of(VB & ~FV, FV);
    const uint64_t single = FV & ~OW;                             // the one valid target of the reads with one
    const uint64_t empty = H & ~k_head_of(FV, H);                 // reads that lost every target
    // index of this lane's read among the selectors
    const uint32_t ridx = sel_base + mask_rank(H) + (k_bit(H) ? 1u : 0u) - 1u;
    if (k_bit(single)) out.sel[ridx] = g & 0x7fffffffu;
    if (k_bit(empty)) out.sel[ridx] = 0xffffffffu;
    Lookup lk{0ull};
    if (OW) {
        const uint64_t le = (2ull << lane) - 1ull;  // the lanes up to and including this one
        // the first valid lane of this lane's read (for the valid lanes): the highest bit of FV at or below it
        const uint32_t fvl = 63u - static_cast<uint32_t>(__builtin_clzll((FV & le) | 1ull));
        const typename Rows::Row first = Rows::from_lane(row, fvl << 2);
        bool ne[8];
        Rows::differ(row, first, ne);
        // Per level: D = the valid lanes that differ from their read's first valid one; the nearest first-valid lane at
        // or below a D lane is its own read's, so the OWNERS with a differing lane come out of one carry chain on the
        // scalar unit (k_head_of) and a lane learns its level from one select per level -- not, per lane and level, from
        // 64-bit masks of "the lanes above me in my read" (two vector instructions per 64-bit operation: the kernel is
        // bound by its vector instructions per window).
        const uint64_t fvr = __builtin_bitreverse64(FV), nfvr = ~fvr;
        auto owners_of = [&](uint64_t D) { return __builtin_bitreverse64((nfvr + __builtin_bitreverse64(D)) & fvr); };
        uint32_t lv = 8u;  // src/slimm.hpp:516-531: the first level (from the leaves) on which all valid targets agree
        // (most reads agree within a few levels of the leaves: the four upper levels are looked at only when some read of
        // the window has found none among the four lower ones -- config 2 -5 %, config 5 -12 %; two / two / four: no better)
#pragma unroll
        for (int l = 3; l >= 0; --l) {
            const uint64_t bad = owners_of(k_ballot(ne[l]) & VB);
            lv = k_bit(bad) ? lv : static_cast<uint32_t>(l);
        }
        if (k_ballot(lv == 8u) & OW) {
            uint32_t up = 8u;
#pragma unroll
            for (int l = 7; l >= 4; --l) {
                const uint64_t bad = owners_of(k_ballot(ne[l]) & VB);
                up = k_bit(bad) ? up : static_cast<uint32_t>(l);
            }
            lv = lv == 8u ? up : lv;
        }
        const uint64_t Q4 = k_ballot(lv == 8u) & OW;  // no level agrees (quirk Q4): rare, one read at a time below
        // the owners' selectors: the taxon part is (level, index) as the row holds it -- no look-up, no third round trip
        if (k_bit(OW & ~Q4)) out.sel[ridx] = out.taxon_base + rows.taxon_index(lv & 7u, Rows::field(row, lv & 7u));
        // children[taxon] gets the valid targets' references (src/slimm.hpp:536-557): a (reference, level) mark
        const uint32_t lv_read = __builtin_amdgcn_ds_bpermute(fvl << 2, lv);
        if (k_bit(VB & ~single) && lv_read < 8u) out.marks[ref * kMarkBytes + lv_read] = 1;  // plain, idempotent byte store
        lk.q4 = Q4;
    }
    return lk;
}

// The reads of a window on whose valid targets no level agrees (quirk Q4: rare), one at a time: the taxon is level 7 of
// the largest valid reference, the children are (taxon, reference) pairs in the hash set.  Kept OUT of filter_window:
// with this path's loads and loops inside it, the compiler no longer knows how many memory operations are under way
// where the paths join and makes every window wait for all of them (s_waitcnt vmcnt(0): the stores of the window
// before, the loads asked for ahead).
template <typename Rows>
__device__ __forceinline__ void filter_q4(const Rows& rows, const FilterOut& out, uint32_t lane, uint32_t X, uint32_t w,
                                          const typename Rows::Row& row, uint64_t valid_lanes, uint32_t sel_base, uint64_t Q4) {
    const uint64_t PR = k_below(X);
    const uint64_t H = k_ballot((w >> 31) != 0u) & PR;
    const uint32_t ref = w & 0x7fffffffu;
    const uint64_t VB = valid_lanes & PR;
    uint32_t f[8];
    Rows::fields(row, f);
    uint64_t todo = Q4;
    while (todo) {
        const uint32_t o = static_cast<uint32_t>(__builtin_ctzll(todo));
        todo &= todo - 1ull;
        const uint64_t later = H & ~k_below(o + 1u);
        const uint32_t nxt = later ? static_cast<uint32_t>(__builtin_ctzll(later)) : X;
        const uint64_t Vs = VB & k_below(nxt) & ~k_below(o);
        ReadAcc acc;
        read_clear(acc);
        read_max(acc, Vs, ref, f[7]);
        const uint32_t index = rows.taxon_index(7u, acc.max_f7);
        const uint32_t taxon = rows.taxon_at(index);
        read_children(out, k_bit(Vs), ref, 8u, taxon);
        if (lane == 0u) out.sel[sel_base + static_cast<uint32_t>(__popcll(H & k_below(o + 1u))) - 1u] = out.taxon_base + index;
    }
}

// ... and the same with the owners' taxa looked up at once (the windows of filter_span, one at a time)
template <typename Rows>
__device__ __forceinline__ void filter_window_now(const Rows& rows, const FilterOut& out, uint32_t lane, uint32_t X, uint32_t w,
                                                  uint32_t g, const typename Rows::Row& row, uint32_t sel_base) {
    const uint64_t vb = k_ballot(Rows::valid(row));
    const Lookup lk = filter_window(rows, out, lane, X, w, g, row, vb, sel_base);
    if (lk.q4) filter_q4(rows, out, lane, X, w, row, vb, sel_base, lk.q4);
}

// The targets [pos, endp) -- whole reads, any number of them, reads of 64 targets and more among them -- one window
// after the other (what a run of 64 records or more leaves behind, front.hip).
template <typename Rows>
__device__ __forceinline__ void filter_span(const Rows& rows, const FilterOut& out, const uint32_t* __restrict__ tgt_ref,
                                         const uint32_t* __restrict__ tgt_gbin, uint32_t lane, uint32_t pos, uint32_t endp,
                                         uint32_t sel_base) {
    while (pos < endp) {
        const uint32_t n_live = min(64u, endp - pos);
        const bool live = lane < n_live;
        const uint32_t w = live ? tgt_ref[pos + lane] : 0u;
        const uint32_t g = live ? tgt_gbin[pos + lane] : 0u;
        const uint64_t HB = k_ballot((w >> 31) != 0u) & k_below(n_live);
        const uint32_t X = pos + 64u >= endp ? n_live : 63u - static_cast<uint32_t>(__builtin_clzll(HB | 1ull));
        if (X != 0u) {
            const typename Rows::Row row = rows.load(lane < X ? (w & 0x7fffffffu) : 0u);
            filter_window_now(rows, out, lane, X, w, g, row, sel_base);
            sel_base += static_cast<uint32_t>(__popcll(HB & k_below(X)));
            pos += X;
            continue;
        }
        // ---- a read with 64 targets or more: walk it in chunks, then once more for its children
        ReadAcc acc;
        read_clear(acc);
        uint32_t e = pos;  // end of the read
        while (true) {
            const bool in = e + lane < endp;
            const uint32_t ww = in ? tgt_ref[e + lane] : 0x80000000u;
            const uint32_t gg = in ? tgt_gbin[e + lane] : 0u;
            uint64_t heads = k_ballot((ww >> 31) != 0u);
            if (e == pos) heads &= ~1ull;
            const uint32_t n_in = heads ? static_cast<uint32_t>(__builtin_ctzll(heads)) : 64u;
            const uint32_t rr = ww & 0x7fffffffu;
            const typename Rows::Row row = rows.load(lane < n_in ? rr : 0u);
            uint32_t f[8];
            Rows::fields(row, f);
            const uint64_t Vs = k_ballot(Rows::valid(row)) & k_below(n_in);
            read_add(acc, Vs, gg, f);
            read_max(acc, Vs, rr, f[7]);
            e += n_in;
            if (n_in < 64u) break;
        }
        uint32_t sel = 0xffffffffu;
        if (acc.nv == 1u) {
            sel = acc.first_g & 0x7fffffffu;
        } else if (acc.nv > 1u) {
            uint32_t lv, index;
            const uint32_t taxon = read_taxon(rows, acc, &lv, &index);
            sel = out.taxon_base + index;
            for (uint32_t c = pos; c < e; c += 64u) {
                const bool in = c + lane < e;
                const uint32_t rr = in ? (tgt_ref[c + lane] & 0x7fffffffu) : 0u;
                const typename Rows::Row row = rows.load(rr);
                read_children(out, in && Rows::valid(row), rr, lv, taxon);
            }
        }
        if (lane == 0u) out.sel[sel_base] = sel;
        sel_base += 1u;
        pos = e;
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------
// k_filter_compact (round 6): the same phase B + C(1), COMPACTED FIRST.
// k_filter above works through every window of <= 64 targets with ~85 scalar + ~85 vector instructions of lane-mask
// arithmetic -- three rounds at 2.4 ms per 1 B records, bound by instruction issue, while only a fifth of the targets name
// a valid reference (728 of 20 000 references at 1 B records).  Here a wave streams through its slot's targets in plain
// chunks of 64 (no window cuts: read boundaries do not matter yet): one bit per reference says "valid" (a 2.5 KB bitmap: an
// L1 hit instead of the 16-byte row gather), a ballot and two lane counts give every valid target its place in a ring of
// 128 entries in LDS -- {reference, bin, index of its read in the slot} -- and the number of its read among the slot's.
// Whenever the ring holds 64 entries (or the slot ends) the oldest ones that are whole reads are worked on as ONE window
// in which every lane is valid: heads = "my read differs from the lane's before" (one DPP move), a read with one entry is
// unique (its selector is its bin), a read with several gets the level scan of k_filter's window -- on a quarter of the
// windows.  Reads that keep no target never show up: the slot's selectors live in LDS, start as "nothing" and leave in one
// coalesced pass at the slot's end.  A read with 64 valid targets or more does not fit a window: its slot goes through
// filter_span (the one-window-at-a-time walk that takes any read length) instead -- same selectors, marks set twice.
// ---------------------------------------------------------------------------------------------------------
constexpr uint32_t kFcRing = 128;                  // (a power of two; at most 63 left over + 64 new entries)
constexpr uint32_t kFcReads = kSlotRecs + 8;       // reads of a slot: its records' runs START in it; the last run may add two
constexpr uint32_t kFcBatch = 4;                   // chunks of 64 targets per trip of a wave

template <typename Rows>
__device__ __forceinline__ void compact_window(const Rows& rows, const FilterOut& out, uint32_t lane, uint32_t X, uint64_t Hc,
                                               uint32_t cref, uint32_t cg, uint32_t cr, const typename Rows::Row& row,
                                               uint32_t* __restrict__ s_sel) {
    const uint64_t PR = k_below(X);
    const uint64_t H = Hc & PR;                                    // (bit 0 is set)
    const uint64_t stops = (H >> 1) | (1ull << (X - 1u));          // the last entry of every read
    const uint64_t single = H & stops, OW = H & ~stops;            // reads of one entry; first entries of the others
    if (k_bit(single)) s_sel[cr] = cg & 0x7fffffffu;
    if (OW) {
        const uint64_t le = (2ull << lane) - 1ull;
        const uint32_t fvl = 63u - static_cast<uint32_t>(__builtin_clzll((H & le) | 1ull));   // the first entry of this lane's read
        const typename Rows::Row first = Rows::from_lane(row, fvl << 2);
        bool ne[8];
        Rows::differ(row, first, ne);
        const uint64_t fvr = __builtin_bitreverse64(H), nfvr = ~fvr;
        auto owners_of = [&](uint64_t D) { return __builtin_bitreverse64((nfvr + __builtin_bitreverse64(D)) & fvr); };
        uint32_t lv = 8u;  // src/slimm.hpp:516-531: the first level (from the leaves) on which all kept targets agree
#pragma unroll
        for (int l = 3; l >= 0; --l) {
            const uint64_t bad = owners_of(k_ballot(ne[l]) & PR);
            lv = k_bit(bad) ? lv : static_cast<uint32_t>(l);
        }
        if (k_ballot(lv == 8u) & OW) {
            uint32_t up = 8u;
#pragma unroll
            for (int l = 7; l >= 4; --l) {
                const uint64_t bad = owners_of(k_ballot(ne[l]) & PR);
                up = k_bit(bad) ? up : static_cast<uint32_t>(l);
            }
            lv = lv == 8u ? up : lv;
        }
        const uint64_t Q4 = k_ballot(lv == 8u) & OW;  // no level agrees (quirk Q4): rare, one read at a time below
        if (k_bit(OW & ~Q4)) s_sel[cr] = out.taxon_base + rows.taxon_index(lv & 7u, Rows::field(row, lv & 7u));
        const uint32_t lv_read = __builtin_amdgcn_ds_bpermute(fvl << 2, lv);
        if (k_bit(PR & ~single) && lv_read < 8u) out.marks[cref * kMarkBytes + lv_read] = 1;  // plain, idempotent byte store
        if (Q4) {
            uint32_t f[8];
            Rows::fields(row, f);
            uint64_t todo = Q4;
            while (todo) {
                const uint32_t o = static_cast<uint32_t>(__builtin_ctzll(todo));
                todo &= todo - 1ull;
                const uint64_t later = H & ~k_below(o + 1u);
                const uint32_t nxt = later ? static_cast<uint32_t>(__builtin_ctzll(later)) : X;
                const uint64_t Vs = k_below(nxt) & ~k_below(o);
                ReadAcc acc;
                read_clear(acc);
                read_max(acc, Vs, cref, f[7]);
                const uint32_t index = rows.taxon_index(7u, acc.max_f7);
                const uint32_t taxon = rows.taxon_at(index);
                read_children(out, k_bit(Vs), cref, 8u, taxon);
                const uint32_t rl = static_cast<uint32_t>(__builtin_amdgcn_readlane(cr, o));
                if (lane == 0u) s_sel[rl] = out.taxon_base + index;
            }
        }
    }
}

template <typename Rows>
__global__ __launch_bounds__(64, 8) void k_filter_compact(const uint32_t* __restrict__ tgt_ref, const uint32_t* __restrict__ tgt_gbin,
                                                          const uint4* __restrict__ slots, uint32_t nslots,
                                                          const uint32_t* __restrict__ valid_bits, uint32_t* __restrict__ redo,
                                                          const Rows rows, const FilterOut out) {
    __shared__ uint32_t s_ref[kFcRing], s_g[kFcRing], s_r[kFcRing];
    __shared__ uint32_t s_sel[kFcReads];
    const uint32_t lane = lane_id();
    for (uint32_t r = lane; r < kFcReads; r += 64u) s_sel[r] = 0xffffffffu;
    __builtin_amdgcn_wave_barrier();
    for (uint32_t slot = blockIdx.x; slot < nslots; slot += gridDim.x) {
        const uint4 d = slots[slot];
        const uint32_t rb = out.rbase[slot] + out.bbase[slot >> 10];
        const uint32_t t0 = d.x, tend = d.x + d.y;
        bool whole_reads_walk = d.z > kFcReads;   // (cannot happen with kSlotRecs records per slot; the walk takes anything)
        uint32_t rcount = 0, ccount = 0, cdone = 0;
        // kFcBatch chunks of 64 targets per trip: their words were asked for a trip ahead, their bitmap words are asked for
        // together -- one dependent round trip per 256 targets (with one chunk per trip the kernel waited for that gather:
        // 2.15 ms at 1 B records, 12 % under the window-by-window kernel)
        uint32_t wn[kFcBatch], gn[kFcBatch];
        auto ask = [&](uint32_t t) {
#pragma unroll
            for (uint32_t u = 0; u < kFcBatch; ++u) {
                const uint32_t i = min(t + 64u * u + lane, tend - 1u);   // (lanes behind the slot's end: its last target, not used)
                wn[u] = tgt_ref[i];
                gn[u] = tgt_gbin[i];
            }
        };
        if (!whole_reads_walk && t0 < tend) ask(t0);
        for (uint32_t t = t0; t < tend && !whole_reads_walk; t += 64u * kFcBatch) {
            uint32_t w[kFcBatch], g[kFcBatch], bits[kFcBatch];
#pragma unroll
            for (uint32_t u = 0; u < kFcBatch; ++u) {
                w[u] = wn[u];
                g[u] = gn[u];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (uint32_t u = 0; u < kFcBatch; ++u) bits[u] = valid_bits[(w[u] & 0x7fffffffu) >> 5];
            __builtin_amdgcn_sched_barrier(0);
            if (t + 64u * kFcBatch < tend) ask(t + 64u * kFcBatch);
            __builtin_amdgcn_sched_barrier(0);
            // (one chunk after the other through ONE copy of the code below: the chunk's words picked out of the batch's
            // registers by a few selects -- unrolled, the window code would stand in the kernel kFcBatch times)
#pragma unroll 1
            for (uint32_t u = 0; u < kFcBatch; ++u) {
                const uint32_t tu = t + 64u * u;
                if (tu >= tend) break;
                uint32_t wu = w[0], gu = g[0], bu = bits[0];
#pragma unroll
                for (uint32_t k = 1; k < kFcBatch; ++k) {
                    wu = u == k ? w[k] : wu;
                    gu = u == k ? g[k] : gu;
                    bu = u == k ? bits[k] : bu;
                }
                const bool live = tu + lane < tend;
                const uint32_t ref = wu & 0x7fffffffu;
                const bool valid = live && ((bu >> (ref & 31u)) & 1u) != 0u;
                const bool head = live && (wu >> 31) != 0u;
                const uint64_t H = k_ballot(head), VB = k_ballot(valid);
                if (valid) {
                    const uint32_t p = (ccount + mask_rank(VB)) & (kFcRing - 1u);
                    s_ref[p] = ref;
                    s_g[p] = gu;
                    s_r[p] = rcount + mask_rank(H) + (head ? 1u : 0u) - 1u;
                }
                rcount += static_cast<uint32_t>(__popcll(H));
                ccount += static_cast<uint32_t>(__popcll(VB));
                const bool last = tu + 64u >= tend;
                while (ccount - cdone >= 64u || (last && ccount != cdone)) {
                    __builtin_amdgcn_wave_barrier();   // (the ring's words: written above, read here)
                    const uint32_t n_live = min(64u, ccount - cdone);
                    const uint32_t e = (cdone + lane) & (kFcRing - 1u);
                    const uint32_t cref = lane < n_live ? s_ref[e] : 0u;
                    const uint32_t cg = s_g[e], cr = lane < n_live ? s_r[e] : 0xffffffffu;
                    typename Rows::Row row = rows.load(cref);
                    const uint32_t before = __builtin_amdgcn_update_dpp(0xffffffffu, cr, 0x138, 0xf, 0xf, false);   // wave_shr:1 (lane 0: none)
                    const uint64_t Hc = k_ballot(lane < n_live && (lane == 0u || cr != before));
                    // whole reads only: the last read of a full window may go on in the entries to come
                    const bool final_window = last && ccount - cdone <= 64u;
                    const uint32_t X = final_window ? n_live : 63u - static_cast<uint32_t>(__builtin_clzll(Hc));
                    __builtin_amdgcn_wave_barrier();   // (every lane has read its entry before the ring is written again)
                    if (X == 0u) {   // 64 valid targets of one read (or more): the slot goes through the walk that takes any length
                        whole_reads_walk = true;
                        break;
                    }
                    compact_window(rows, out, lane, X, Hc, cref, cg, cr, row, s_sel);
                    cdone += X;
                }
                if (whole_reads_walk) break;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (whole_reads_walk) {   // (rare: left to k_filter_walk, a kernel of its own -- inlined, its registers would be this kernel's)
            if (lane == 0u) redo[atomicAdd(&out.counters[CNT_REDO], 1u)] = slot;
            for (uint32_t r = lane; r < kFcReads; r += 64u) s_sel[r] = 0xffffffffu;
        } else {
            for (uint32_t r = lane; r < d.z; r += 64u) {   // the slot's selectors, reads without a valid target included
                out.sel[rb + r] = s_sel[r];
                s_sel[r] = 0xffffffffu;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// the slots k_filter_compact could not take: one window at a time, reads of any length (filter_span)
template <typename Rows>
__global__ __launch_bounds__(64) void k_filter_walk(const uint32_t* __restrict__ tgt_ref, const uint32_t* __restrict__ tgt_gbin,
                                                    const uint4* __restrict__ slots, const uint32_t* __restrict__ redo, const Rows rows,
                                                    const FilterOut out) {
    const uint32_t lane = lane_id();
    const uint32_t n = out.counters[CNT_REDO];
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        const uint32_t slot = redo[i];
        const uint4 d = slots[slot];
        filter_span(rows, out, tgt_ref, tgt_gbin, lane, d.x, d.x + d.y, out.rbase[slot] + out.bbase[slot >> 10]);
    }
}

// ---------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------
// slimm_check_grouping: is a stream that was DECLARED grouped really grouped?  Every record that starts a qName run (its
// identity differs from the record before) puts its identity into an open-addressing set; an identity that is already
// there starts a second run -- the name re-appears non-adjacently, and the single-pass front end would make two reads of
// what the reference's hash map (src/slimm.hpp:204-211) merges into one.  *n_split counts such run starts.
__global__ __launch_bounds__(256) void k_check_grouping(const uint64_t* __restrict__ key, uint32_t n, uint64_t id_mask,
                                                        uint64_t* __restrict__ tab, uint32_t tab_mask,
                                                        uint32_t* __restrict__ n_split) {
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const uint64_t k = key[i] & id_mask;
        if (i != 0u && (key[i - 1u] & id_mask) == k) continue;  // inside a run
        uint32_t slot = static_cast<uint32_t>(mix64(k)) & tab_mask;
        for (uint32_t probe = 0; probe <= tab_mask; ++probe) {
            const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&tab[slot]), ~0ull,
                                                     static_cast<unsigned long long>(k));
            if (old == ~0ull) break;
            if (old == k) {
                atomicAdd(n_split, 1u);
                break;
            }
            slot = (slot + 1u) & tab_mask;
        }
    }
}
void launch_check_grouping(hipStream_t st, const uint64_t* key, uint32_t n, uint64_t id_mask, uint64_t* tab, uint32_t tab_mask,
                           uint32_t* n_split) {
    if (n) hipLaunchKernelGGL(k_check_grouping, dim3(std::min<uint32_t>((n + 255u) / 256u, 8192u)), dim3(256), 0, st, key, n,
                              id_mask, tab, tab_mask, n_split);
}

void launch_hist(hipStream_t st, const uint32_t* tgt_gbin, const uint4* slots, uint32_t nslots, uint32_t* counters,
                 uint32_t* tail, uint32_t* cov, uint32_t* ucov, bool count_mapped) {
    uint32_t blocks = std::max(1u, std::min((nslots + kWaves - 1) / kWaves, 256u * 16u));
    hipLaunchKernelGGL(k_hist, dim3(blocks), dim3(kBlock), 0, st, tgt_gbin, slots, nslots, counters, tail, cov, ucov,
                       count_mapped ? 1 : 0);
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
    for (uint32_t i = gid; i < z.cp2_n; i += gsz) z.cp2_dst[i] = z.cp2_src[i];
}

void launch_zero(hipStream_t st, const ZeroArgs& z) {
    uint32_t most = std::max(std::max(z.n64, z.cp_n / 4), z.cp2_n);
    for (int k = 0; k < 5; ++k) most = most > z.n[k] ? most : z.n[k];
    uint32_t blocks = (most + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks) hipLaunchKernelGGL(k_zero, dim3(blocks), dim3(256), 0, st, z);
}

// Multi-GPU: this rank's additive partial results as one int32 buffer that ranks SUM in place:
//   [R uniq_reads_count2 | T per-taxon LCA counts | 2R level marks, one 8-bit field per level | number of pairs |
//    pair-set overflow (0 / 1)]
// (a sum over at most 255 ranks cannot carry between the fields, so "field != 0" afterwards is the OR of the marks),
// followed by this rank's own {number of pairs, error flags}, which stay out of the exchange
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
    if (gid == 0) {
        out[3ull * R + T] = cnt[CNT_PAIRS];
        out[3ull * R + T + 1] = (cnt[CNT_ERR] & ERR_PAIR_OVERFLOW) ? 1u : 0u;
        out[3ull * R + T + 2] = cnt[CNT_PAIRS];
        out[3ull * R + T + 3] = cnt[CNT_ERR];
    }
}

void launch_partials_pack(hipStream_t st, const uint32_t* block_b, uint32_t R, uint32_t T, uint32_t* out) {
    const uint32_t n = R > T ? R : T;
    hipLaunchKernelGGL(k_partials_pack, dim3(std::min<uint32_t>(256u, (n + 255u) / 256u + 1u)), dim3(256), 0, st, block_b, R, T,
                       out);
}

// Where a slot's reads start among ALL reads of the file: the exclusive prefix of slots[s].z, as rbase[s] (inside the slot's
// block of 1024 slots) + bbase[s >> 10] (the blocks before).  k_filter writes the selectors there, so that phase B's
// bucketing kernels read ONE dense array instead of walking the slots -- a slot holds 90 reads at 1 B records on 20 k
// references, 18 on 50 k strain-level ones, and a piece of 256 values per slot was 35 % / 7 % full.  Two small launches on
// a side stream while phase A's tile kernels run.
constexpr uint32_t kPrefixBlock = 1024;
__global__ __launch_bounds__(kPrefixBlock) void k_slot_prefix_blocks(const uint4* __restrict__ slots, uint32_t nslots,
                                                                     uint32_t* __restrict__ rbase, uint32_t* __restrict__ bsum) {
    __shared__ uint32_t s_wave[kPrefixBlock / 64];
    const uint32_t s = blockIdx.x * kPrefixBlock + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t c = s < nslots ? slots[s].z : 0u;
    uint32_t inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t a = __shfl_up(inc, o, 64);
        if (lane >= static_cast<uint32_t>(o)) inc += a;
    }
    if (lane == 63u) s_wave[wave] = inc;
    __syncthreads();
    uint32_t before = 0;
#pragma unroll
    for (uint32_t w = 0; w < kPrefixBlock / 64; ++w) before += w < wave ? s_wave[w] : 0u;
    if (s < nslots) rbase[s] = before + inc - c;
    if (threadIdx.x == kPrefixBlock - 1u) bsum[blockIdx.x] = before + inc;
}
// ... and the blocks' sums into the blocks' bases, in place (one workgroup)
__global__ __launch_bounds__(kPrefixBlock) void k_slot_prefix_bases(uint32_t* __restrict__ bsum, uint32_t nblocks) {
    __shared__ uint32_t s_wave[kPrefixBlock / 64];
    __shared__ uint32_t s_carry;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    for (uint32_t b0 = 0; b0 < nblocks; b0 += kPrefixBlock) {
        const uint32_t b = b0 + threadIdx.x;
        const uint32_t c = b < nblocks ? bsum[b] : 0u;
        uint32_t inc = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t a = __shfl_up(inc, o, 64);
            if (lane >= static_cast<uint32_t>(o)) inc += a;
        }
        if (lane == 63u) s_wave[wave] = inc;
        __syncthreads();
        uint32_t before = s_carry;
#pragma unroll
        for (uint32_t w = 0; w < kPrefixBlock / 64; ++w) before += w < wave ? s_wave[w] : 0u;
        if (b < nblocks) bsum[b] = before + inc - c;
        __syncthreads();
        if (threadIdx.x == kPrefixBlock - 1u) s_carry = before + inc;
        __syncthreads();
    }
}
void launch_slot_read_prefix(hipStream_t st, const uint4* slots, uint32_t nslots, uint32_t* rbase, uint32_t* bbase) {
    if (!nslots) return;
    const uint32_t nb = (nslots + kPrefixBlock - 1u) / kPrefixBlock;
    hipLaunchKernelGGL(k_slot_prefix_blocks, dim3(nb), dim3(kPrefixBlock), 0, st, slots, nslots, rbase, bbase);
    hipLaunchKernelGGL(k_slot_prefix_bases, dim3(1), dim3(kPrefixBlock), 0, st, bbase, nb);
}

// the selectors counted with global atomics (direct-atomics fallback: no tile histogram): uniq_cov2[g]++ per read that
// kept one target, lca_count[t]++ per read counted at its LCA
__global__ __launch_bounds__(kBlock) void k_sel_atomics(const uint32_t* __restrict__ sel, uint32_t n_reads, uint32_t taxon_base,
                                                        uint32_t* __restrict__ ucov2, uint32_t* __restrict__ lca_count) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n_reads; i += gridDim.x * kBlock) {
        const uint32_t v = sel[i];
        if (v == 0xffffffffu) continue;
        if (v < taxon_base)
            atomicAdd(&ucov2[v], 1u);
        else
            atomicAdd(&lca_count[v - taxon_base], 1u);
    }
}

void launch_sel_atomics(hipStream_t st, const uint32_t* sel, uint32_t n_reads, uint32_t taxon_base, uint32_t* ucov2,
                        uint32_t* lca_count) {
    uint32_t blocks = std::max(1u, std::min((n_reads + kBlock - 1) / kBlock, 256u * 16u));
    hipLaunchKernelGGL(k_sel_atomics, dim3(blocks), dim3(kBlock), 0, st, sel, n_reads, taxon_base, ucov2, lca_count);
}

void launch_filter(hipStream_t st, const FilterArgs& a, hipEvent_t t0, hipEvent_t t1) {
    if (!a.nslots) return;
    FilterOut out;
    out.sel = a.sel;
    out.rbase = a.slot_rbase;
    out.bbase = a.slot_bbase;
    out.marks = reinterpret_cast<uint8_t*>(a.marks);
    out.pair_tab = a.pair_tab;
    out.pair_list = a.pair_list;
    out.pair_mask = a.pair_mask;
    out.taxon_base = a.taxon_base;
    out.counters = a.counters;
    // one wave per workgroup, one slot at a time, however many workgroups that makes: the dispatcher backfills wave by wave.
    // (Measured and dropped in round 6, profiles/round6/03_filter_variants.txt: persistent waves walking their slots as one
    // stream of batches with the next slot's first batch asked for ahead -- 3 to 4 % slower, and with the descriptors two slots
    // ahead the registers spill; 2 / 8 chunks per trip: the same; row gathers or the target words past the L1: slower.)
    const uint32_t grid = a.nslots;
    if (a.rows16) {
        Rows16 r;
        r.rows = reinterpret_cast<const uint4*>(a.rows16);
        r.taxon_flat = a.taxon_flat;
        r.shift = a.taxon_shift;
        hipExtLaunchKernelGGL(k_filter_compact<Rows16>, dim3(grid), dim3(64), 0, st, t0, t1, 0, a.tgt_ref, a.tgt_gbin, a.slots, a.nslots,
                              a.valid_bits, a.redo, r, out);
        hipLaunchKernelGGL(k_filter_walk<Rows16>, dim3(std::min(a.nslots, 256u)), dim3(64), 0, st, a.tgt_ref, a.tgt_gbin, a.slots, a.redo, r, out);
    } else {
        Rows32 r;
        r.lin4 = reinterpret_cast<const uint4*>(a.lin_dense);
        r.valid_of = a.valid;
        hipExtLaunchKernelGGL(k_filter_compact<Rows32>, dim3(grid), dim3(64), 0, st, t0, t1, 0, a.tgt_ref, a.tgt_gbin, a.slots, a.nslots,
                              a.valid_bits, a.redo, r, out);
        hipLaunchKernelGGL(k_filter_walk<Rows32>, dim3(std::min(a.nslots, 256u)), dim3(64), 0, st, a.tgt_ref, a.tgt_gbin, a.slots, a.redo, r, out);
    }
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

