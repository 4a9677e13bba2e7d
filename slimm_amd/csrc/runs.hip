// Front end of phase A (gfx950, wave64): from the record stream to the per-read target lists (CSR).
//
//   k_runs   classifies every record inside its qName run: head (first record of its read), first (first record of
//            its (read, ref) pair), mate, run start.  One workgroup stages the 32-bit {mapped, run start, mate, ref}
//            word of its 2048 records plus a 128-record halo in LDS, so the look-back over a read's earlier records is
//            an LDS walk; only runs reaching further back than the halo touch global memory again.
//   (k_scan_tiles turns the per-tile head / first counts into offsets)
//   k_emit   scatters `first` records into the target arrays and `head` records into read_off, correcting positions so
//            that the reads of one qName run (mates interleave in mapper output) are laid out contiguously by mate.
//
// Both kernels are templates over the record source: RawRecords works on the caller's record arrays directly -- no
// compaction pass; an unmapped record just never becomes a head or a first -- and SortedRecords on the compacted,
// identity-sorted stream of the record_order = ANY path.
//
// Reference semantics: src/slimm.hpp:194-213 (record filter, bin, read identity = qName + mate),
// src/read_stat.hpp:116-135 (add_target keeps the bin of the FIRST record of a (read, ref) pair: Q1).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>

#include "kernels.h"
#ifndef EXP
#define EXP 0
#endif

namespace slimm {

#if EXP == 9
__device__ unsigned long long g_prof[8 * 32768];  // [tile-wave][phase]
#define PROF_T(x) const unsigned long long x = __builtin_readcyclecounter()
#define PROF_ADD(slot, a, b) if (lane == 0) g_prof[(tile * 4 + (threadIdx.x >> 6)) * 8 + slot] = (b) - (a)
#else
#define PROF_T(x)
#define PROF_ADD(slot, a, b)
#endif

constexpr int kRBlock = 256;
constexpr int kRItems = 8;
constexpr int kRTile = kRBlock * kRItems;
constexpr int kRWaves = kRBlock / 64;
constexpr uint32_t kHalo = 128;
constexpr uint32_t kLookBackMax = 4096;   // records one look-back may walk (SLIMM_E_RUN_LENGTH beyond)
constexpr uint32_t kRunWalkMax = 1u << 20;

// meta word: bit31 mapped | bit30 first record of its qName run | bits 29-28 mate | bits 27-0 reference id
constexpr uint32_t M_VALID = 0x80000000u, M_RUN = 0x40000000u, M_IDENT = 0x3fffffffu;
// flag byte: bit0 head | bit1 first | bits 2-3 mate | bit4 run start | bit5 an earlier record of the run has a larger mate
enum { FL_HEAD = 1, FL_FIRST = 2, FL_MATE_SHIFT = 2, FL_RUN_START = 16, FL_GREATER_BEFORE = 32 };

__device__ __forceinline__ uint32_t r_mask_rank(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32),
                                     __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
}
// Lane mask of a predicate.  HIP's __ballot(int) takes an integer: a bool expression is first turned into 0 / 1 in a VGPR
// and then compared with zero again (two vector instructions per ballot that the compare producing the bool already paid
// for); the builtin takes the i1 directly.
__device__ __forceinline__ uint64_t r_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ uint32_t r_wave_sum(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

struct RawRecords {
    const uint64_t* key;
    const int32_t* ref;
    const int32_t* pos;
    const uint16_t* flag;
    uint32_t n, n_refs;
    const uint32_t* ref_len;
    const uint32_t* bin_off;
    const uint2* geo_tab;  // {contig length, first bin} per reference (k_emit)
    uint32_t half_read, bin_width;
    uint32_t bw_magic;  // floor((2^32 - 1) / bin_width), see div_bin_width
    static constexpr bool kCountsMapped = true;
    // n / bin_width for a divisor known on the host: the quotient estimate mulhi(n, floor((2^32 - 1) / d)) is the true
    // quotient or one less (n * M / 2^32 lies in (n / d - 1, n / d]), so one correction makes it exact.  The compiler's
    // general u32 division refines a float reciprocal and corrects twice: ~30 full-rate-equivalent vector instructions
    // per record (five of them quarter-rate multiplies), a quarter of k_emit's vector work.
    __device__ uint32_t div_bin_width(uint32_t n) const {
        const uint32_t q = __umulhi(n, bw_magic);
        const uint32_t r = n - q * bin_width;
        return q + (r >= bin_width ? 1u : 0u);
    }
    __device__ uint32_t count(const uint32_t*) const { return n; }
    __device__ uint64_t key_of(uint32_t i) const { return key[i] << 2; }  // only the low 62 bits are significant
    __device__ uint32_t meta_of(uint32_t i, bool& bad) const {
        const uint32_t f = flag[i];
        const int32_t r = ref[i];
        const uint32_t mate = (f & 0x40) ? 1u : ((f & 0x80) ? 2u : 0u);   // src/slimm.hpp:205-208
        bool mapped = !(f & 0x4) && r != -1;                             // src/slimm.hpp:197
        if (mapped && static_cast<uint32_t>(r) >= n_refs) {
            bad = true;
            mapped = false;
        }
        return mapped ? (M_VALID | (mate << 28) | static_cast<uint32_t>(r)) : (mate << 28);
    }
    // the loads of one record without any arithmetic or branch in between (so that several records' loads can be in
    // flight at once) and the meta word from them
    struct Raw {
        uint64_t key;
        uint32_t flag;
        int32_t ref;
    };
    __device__ Raw load(uint32_t i) const { return Raw{key[i], flag[i], ref[i]}; }
    __device__ uint64_t raw_key(uint32_t i) const { return key[i]; }
    __device__ static bool same_run(uint64_t a, uint64_t b) { return ((a ^ b) << 2) == 0ull; }
    __device__ static uint64_t key_bits(const Raw& r) { return r.key; }
    __device__ uint32_t meta(const Raw& r, bool& bad) const {
        const uint32_t mate = (r.flag & 0x40) ? 1u : ((r.flag & 0x80) ? 2u : 0u);
        bool mapped = !(r.flag & 0x4) && r.ref != -1;
        if (mapped && static_cast<uint32_t>(r.ref) >= n_refs) {
            bad = true;
            mapped = false;
        }
        return mapped ? (M_VALID | (mate << 28) | static_cast<uint32_t>(r.ref)) : (mate << 28);
    }
    // k_emit: what a `first` record contributes, in two load levels that can each be batched over several records
    struct Hit {
        int32_t ref, pos;
    };
    __device__ Hit hit(uint32_t i) const { return Hit{ref[i], pos[i]}; }
    __device__ uint32_t hit_ref(const Hit& h) const { return min(static_cast<uint32_t>(h.ref), n_refs - 1u); }
    struct Geo {
        uint32_t len, off;
    };
    __device__ Geo geo(uint32_t r) const {  // one 8-byte gather: divergent loads are paid per lane, not per byte
        const uint2 g = geo_tab[r];
        return Geo{g.x, g.y};
    }
    __device__ uint32_t gbin(const Hit& h, const Geo& g) const {
        // uint32 wrap-around of int32 + uint32, then clamp to the contig length (src/slimm.hpp:200-201, Q3)
        return g.off + div_bin_width(min(static_cast<uint32_t>(h.pos) + half_read, g.len));
    }
    __device__ uint32_t gbin_of(uint32_t i, uint32_t r) const {
        // uint32 wrap-around of int32 + uint32, then clamp to the contig length (src/slimm.hpp:200-201, Q3)
        const uint32_t center = min(static_cast<uint32_t>(pos[i]) + half_read, ref_len[r]);
        return bin_off[r] + div_bin_width(center);
    }
};

struct SortedRecords {
    const uint64_t* ident;
    const uint32_t* cref;
    const uint32_t* cgbin;
    static constexpr bool kCountsMapped = false;
    __device__ uint32_t count(const uint32_t* counters) const { return counters[CNT_V]; }
    __device__ uint64_t key_of(uint32_t i) const { return ident[i] >> 2; }
    __device__ uint32_t meta_of(uint32_t i, bool&) const {
        return M_VALID | ((static_cast<uint32_t>(ident[i]) & 3u) << 28) | cref[i];
    }
    struct Raw {
        uint64_t key;
        uint32_t ref;
    };
    __device__ Raw load(uint32_t i) const { return Raw{ident[i], cref[i]}; }
    __device__ uint64_t raw_key(uint32_t i) const { return ident[i]; }
    __device__ static bool same_run(uint64_t a, uint64_t b) { return ((a ^ b) >> 2) == 0ull; }
    __device__ static uint64_t key_bits(const Raw& r) { return r.key; }
    __device__ uint32_t meta(const Raw& r, bool&) const {
        return M_VALID | ((static_cast<uint32_t>(r.key) & 3u) << 28) | r.ref;
    }
    struct Hit {
        uint32_t ref, gbin;
    };
    __device__ Hit hit(uint32_t i) const { return Hit{cref[i], cgbin[i]}; }
    __device__ uint32_t hit_ref(const Hit& h) const { return h.ref; }
    struct Geo {};
    __device__ Geo geo(uint32_t) const { return Geo{}; }
    __device__ uint32_t gbin(const Hit& h, const Geo&) const { return h.gbin; }
    __device__ uint32_t gbin_of(uint32_t i, uint32_t) const { return cgbin[i]; }
};

template <typename Acc>
__device__ __forceinline__ uint32_t full_meta(const Acc& acc, uint32_t i, bool& bad) {
    uint32_t m = acc.meta_of(i, bad);
    if (i == 0 || acc.key_of(i) != acc.key_of(i - 1)) m |= M_RUN;
    return m;
}

// ---------------------------------------------------------------------------------------------------------
// Pieces shared by k_runs and k_runs_hash.  The staged window is cut into 64-record segments (one wave instruction
// each).  Per segment, what a record right behind it inherits:
//   bits 0-2   mates of the mapped records between the segment's last run start (or its first record) and its end
//   bits 8-27  number of records in that stretch
//   bit 31     the segment holds a run start (otherwise the stretch continues into the segment before)
// ---------------------------------------------------------------------------------------------------------
constexpr uint32_t SEG_START = 0x80000000u;

// summary word of the segment whose 64 meta words the wave holds (m = 0 for slots past the window)
__device__ __forceinline__ uint32_t segment_summary(uint32_t m) {
    const uint64_t starts = r_ballot(((m >> 30) & 1u) != 0u);
    // "mapped and mate == k" as ONE compare each ({mapped, mate} = bits 31, 29-28): the ballot of a compare is the
    // compare; the ballot of `a && b` is both compares, an and, a 0/1 select and a third compare
    const uint32_t vm = m & (M_VALID | (3u << 28));
    const uint64_t tail = starts ? ~((1ull << (63 - __builtin_clzll(starts))) - 1ull) : ~0ull;
    const uint32_t seen = ((r_ballot(vm == M_VALID) & tail) ? 1u : 0u) |
                          ((r_ballot(vm == (M_VALID | (1u << 28))) & tail) ? 2u : 0u) |
                          ((r_ballot(vm == (M_VALID | (2u << 28))) & tail) ? 4u : 0u);
    return seen | (static_cast<uint32_t>(__popcll(tail)) << 8) | (starts ? SEG_START : 0u);
}

// The previous record's key for every lane of a wave holding 64 consecutive keys: the neighbouring lane's (DPP wave
// shift), lane 0 gets carry_key (the last key of the segment before).  (Loading key[i - 1] as well -- 512-byte wave
// loads 8 bytes off their alignment -- cost 35 of k_runs' 90 us.)
__device__ __forceinline__ uint64_t previous_key(uint64_t mine, uint64_t carry_key) {
    const uint32_t plo = __builtin_amdgcn_update_dpp(static_cast<uint32_t>(carry_key), static_cast<uint32_t>(mine), 0x138,
                                                     0xf, 0xf, false);
    const uint32_t phi = __builtin_amdgcn_update_dpp(static_cast<uint32_t>(carry_key >> 32),
                                                     static_cast<uint32_t>(mine >> 32), 0x138, 0xf, 0xf, false);
    return (static_cast<uint64_t>(phi) << 32) | plo;
}

// Turns the segment summaries s_seg[0 .. nseg) into the carry INTO each segment (mates seen and records since the run
// start, looking back over whole segments; bit 31: the run starts before the window).  All threads of the workgroup call.
__device__ __forceinline__ void segment_carries(uint32_t* s_seg, uint32_t nseg, bool stream_start) {
    uint32_t carry = 0;
    if (threadIdx.x < nseg) {
        uint32_t seen = 0, len = 0, open = 1;
        for (int p = static_cast<int>(threadIdx.x) - 1; p >= 0; --p) {
            const uint32_t sm = s_seg[p];
            seen |= sm & 7u;
            len += (sm >> 8) & 0xfffffu;
            if (sm & SEG_START) {
                open = 0;
                break;
            }
        }
        if (threadIdx.x == 0 && stream_start) open = 0;  // nothing before the first record of the stream
        carry = seen | (len << 8) | (open << 31);
    }
    __syncthreads();
    if (threadIdx.x < nseg) s_seg[threadIdx.x] = carry;
    __syncthreads();
}

// Per record, from the ballots of its own segment plus the carry: is it the first mapped record of its (run, mate)
// (head), did a mapped record with a larger mate precede it in the run (gb), how many records precede it in the run
// (len), and does the run start before the window (open).  `head` and `gb` are "was there a mapped record with mate m
// since the run start": three ballots, a clz and a 64-bit compare instead of a walk.
struct RunScan {
    uint32_t head, gb, len, open;
};
__device__ __forceinline__ RunScan run_scan(uint32_t me, uint32_t seg_carry, bool use_carry, uint32_t lane, uint64_t lt_mask,
                                            uint64_t le_mask) {
    const uint32_t valid = me >> 31, my_mate = (me >> 28) & 3u;
    const uint64_t starts = r_ballot(((me >> 30) & 1u) != 0u);
    const uint32_t vm = me & (M_VALID | (3u << 28));  // (one compare per ballot: see segment_summary)
    const uint64_t v0 = r_ballot(vm == M_VALID), v1 = r_ballot(vm == (M_VALID | (1u << 28))),
                   v2 = r_ballot(vm == (M_VALID | (2u << 28)));
    const uint64_t ps = starts & le_mask;  // run starts at or before me in this segment
    const uint32_t from = ps ? 63u - static_cast<uint32_t>(__builtin_clzll(ps)) : 0u;
    const uint64_t bit_from = 1ull << from;
    // a mapped record with mate m in [from, lane)  <=>  (v_m & lanes before me) >= bit(from)
    uint32_t seen = ((v0 & lt_mask) >= bit_from ? 1u : 0u) | ((v1 & lt_mask) >= bit_from ? 2u : 0u) |
                    ((v2 & lt_mask) >= bit_from ? 4u : 0u);
    RunScan r;
    r.len = lane - from;
    r.open = 0u;
    if (!ps && use_carry) {  // my run starts in an earlier segment
        seen |= seg_carry & 7u;
        r.len += (seg_carry >> 8) & 0xfffffu;
        r.open = (seg_carry >> 31) & valid;
    }
    r.head = valid & (((seen >> my_mate) & 1u) ^ 1u);
    r.gb = valid & ((seen >> (my_mate + 1u)) != 0u ? 1u : 0u);
    return r;
}

// Tile offsets without a scan kernel (streams of up to kFusedEmitTiles tiles): the per-tile (heads, firsts, mapped) counts
// are summed by chunk of 64 - 256 tiles, and a k_emit workgroup gets its offset as the chunk sums before its chunk plus
// the tile counts before it inside the chunk: at most 256 + 63 loads, all L2 hits.  Of the two classification kernels one
// returns at once; when that is k_runs_hash (launched second) its idle launch computes the chunk sums from k_runs'
// counts, and when k_runs_hash does the classifying it adds each tile's counts to its chunk with three atomics.
// (Atomics at the end of k_runs' one-tile workgroups held every workgroup ~2 us longer: 81 -> 96 us.  k_scan_tiles
// as a launch of its own was 9 us of a 520 us file, half of it the dependent-dispatch overhead every launch pays.)
// (chunks of 64, 128 or 256 tiles -- 1 << chunk_shift -- so that there are at most 256 of them up to 64 K tiles)
template <typename Acc>
__device__ __forceinline__ void add_chunk_sums(uint32_t* chunk_acc, uint32_t chunk_shift, uint32_t tile, uint2 t, uint32_t v) {
    uint32_t* a = chunk_acc + (tile >> chunk_shift) * 4u;
    atomicAdd(a, t.x);
    atomicAdd(a + 1, t.y);
    if (Acc::kCountsMapped) atomicAdd(a + 2, v);
}

// k_runs works through a tile in passes of kQTile records (each with its own halo): half the registers of a
// whole-tile pass, so more workgroups are resident and their load and classify phases overlap
#ifndef SLIMM_RUNS_MINBLOCKS
#define SLIMM_RUNS_MINBLOCKS 3  // workgroups per CU the register allocation aims for
#endif
#ifndef SLIMM_Q_ITEMS
#define SLIMM_Q_ITEMS 4
#endif
constexpr int kQItems = SLIMM_Q_ITEMS;
constexpr int kQTile = kRBlock * kQItems;
constexpr uint32_t kSegs = (kQTile + kHalo) / 64;

template <typename Acc>
__device__ __forceinline__ void runs_walk(const Acc& acc, uint32_t ntiles, uint32_t* __restrict__ counters,
                                          uint8_t* __restrict__ fl, uint2* __restrict__ tile_cnt,
                                          uint32_t* __restrict__ tile_valid) {
    __shared__ uint32_t s_meta[kQTile + kHalo];
    __shared__ uint32_t s_flw[kQTile / 4];  // the pass's flag bytes, written out coalesced
    __shared__ uint32_t s_seg[kSegs];       // segment summaries, then (in place) the carry into each segment
    __shared__ uint64_t s_last[kSegs + 2];  // key of the last record of each segment
    __shared__ uint2 s_w[kRWaves];
    __shared__ uint32_t s_v[kRWaves];
    uint8_t* s_fl = reinterpret_cast<uint8_t*>(s_flw);
    const uint32_t N = acc.count(counters);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t lt_mask = (1ull << lane) - 1ull;          // lanes before mine
    const uint64_t le_mask = lt_mask | (1ull << lane);       // ... and mine
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {  // grid-stride over tiles: the launch stays
    uint32_t nh = 0, nf = 0, nv = 0, any_gb = 0;                          // small, so not being chosen costs nothing
    bool bad = false, too_long = false;
    for (uint32_t base = tile * kRTile; base < min((tile + 1u) * kRTile, N); base += kQTile) {
        const uint32_t lds_lo = base >= kHalo ? base - kHalo : 0u;
        const uint32_t lds_hi = min(base + static_cast<uint32_t>(kQTile), N);
        const uint32_t off = base - lds_lo, n_here = lds_hi - base, W = lds_hi - lds_lo;
        const uint32_t nseg = (W + 63u) >> 6;
        // 1. stage {mapped, run start, mate, ref} of the window (tile + halo before it) and summarise each segment.
        //    All loads of a thread (9 records x {key, previous key, flag, ref}) go out back to back, with clamped
        //    indices instead of bounds branches: a branch per record made every load round trip a serial step
        //    (two per record, 18 per tile), and that latency chain -- not bandwidth -- set the kernel's time.
        PROF_T(t0);
        {
            typename Acc::Raw raw[kQItems + 1];
#pragma unroll
            for (int k = 0; k <= kQItems; ++k) raw[k] = acc.load(min(lds_lo + k * kRBlock + threadIdx.x, N - 1u));
            // the key of the record before the window (every thread the same address: one request)
            const uint64_t before = acc.raw_key(lds_lo ? lds_lo - 1u : 0u);
            // The previous record's key comes from the neighbouring lane (DPP shift), for lane 0 from the last lane
            // of the wave before through LDS.  (Loading key[i - 1] as well -- 512-byte wave loads 8 bytes off their
            // alignment -- cost 35 of the kernel's 90 us.)
            if (lane == 63) {
#pragma unroll
                for (int k = 0; k <= kQItems; ++k) s_last[k * kRWaves + wave] = Acc::key_bits(raw[k]);
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k <= kQItems; ++k) {
                const uint32_t j = k * kRBlock + threadIdx.x;   // window index; segment j >> 6 is this wave's, whole
                if ((j & ~63u) < W) {                           // wave-uniform
                    const uint64_t mine = Acc::key_bits(raw[k]);
                    const uint32_t seg = j >> 6;
                    const uint64_t prev = previous_key(mine, seg ? s_last[seg - 1u] : before);  // s_last: by segment
                    uint32_t m = 0u;
                    if (j < W) {
                        m = acc.meta(raw[k], bad);
                        if (lds_lo + j == 0u || !Acc::same_run(mine, prev)) m |= M_RUN;
                        s_meta[j] = m;
                    }
                    const uint32_t summary = segment_summary(m);
                    if (lane == 0) s_seg[seg] = summary;
                }
            }
        }
        PROF_T(t1);
        __syncthreads();
        PROF_T(t2);
        // 2. carry into each segment: mates seen and records since the run start, looking back over whole segments
        segment_carries(s_seg, nseg, lds_lo == 0);
        PROF_T(t3);
        // 3. classify.  head / larger-mate-before come from the ballots of the record's own segment plus the carry;
        //    only the duplicate test (same read AND same reference earlier in the run: Q1) walks back through LDS.
        //    The walk handles the thread's 8 records in ONE loop over the distance -- 8 independent LDS reads per
        //    trip, so their latency overlaps -- and the trip count is known up front (distance to the run start).
        //    (The first version walked per record for all three answers, ~30 vector instructions per trip behind a
        //    serial LDS read each, and was bound by exactly that.)
        uint32_t me_[kQItems], wl_[kQItems], st_[kQItems];  // meta word, walk length, {head, gb, open} bits
#pragma unroll
        for (int k = 0; k < kQItems; ++k) {
            const uint32_t lpos = k * kRBlock + threadIdx.x;
            const uint32_t li = off + lpos;
            const uint32_t me = lpos < n_here ? s_meta[li] : M_RUN;  // past the end: not mapped, a run of its own
            const RunScan rs = run_scan(me, s_seg[li >> 6], true, lane, lt_mask, le_mask);
            me_[k] = me;
            wl_[k] = (me >> 31) ? min(rs.len, li) : 0u;  // compare with this many records before me (all staged unless open)
            st_[k] = rs.head | (rs.gb << 1) | (rs.open << 2);
        }
        PROF_T(t4);
        uint32_t dup_[kQItems];  // minimum over the walked records of (their word ^ mine) & compared bits: 0 <=> duplicate
        {
            uint32_t longest = 0u;
            uint32_t a4[kQItems], w4[kQItems];  // byte offsets: my slot in s_meta, walk length
#pragma unroll
            for (int k = 0; k < kQItems; ++k) {
                longest = max(longest, wl_[k]);
                dup_[k] = 0xffffffffu;
                a4[k] = (off + k * kRBlock + threadIdx.x) * 4u;
                w4[k] = wl_[k] * 4u;
            }
            const char* meta_bytes = reinterpret_cast<const char*>(s_meta);
            for (uint32_t d = 1; r_ballot(d <= longest) != 0ull; ++d) {
#pragma unroll
                for (int k = 0; k < kQItems; ++k) {
                    // past my run: the same record (the run start) again, harmless
                    const uint32_t m = *reinterpret_cast<const uint32_t*>(meta_bytes + (a4[k] - min(d * 4u, w4[k])));
                    dup_[k] = min(dup_[k], (m ^ me_[k]) & (M_VALID | M_IDENT));
                }
            }
        }
        PROF_T(t5);
#pragma unroll
        for (int k = 0; k < kQItems; ++k) {
            const uint32_t lpos = k * kRBlock + threadIdx.x;
            const uint32_t me = me_[k], valid = me >> 31, my_mate = (me >> 28) & 3u;
            uint32_t head = st_[k] & 1u, gb = (st_[k] >> 1) & 1u;
            uint32_t first = valid & ((wl_[k] != 0u && dup_[k] == 0u) ? 0u : 1u);
            if (st_[k] & 4u) {  // the run reaches back beyond the halo: go on in global memory (rare)
                const uint32_t my_ident = me & M_IDENT;
                uint32_t j = lds_lo, steps = 0;
                while (j > 0 && first) {
                    --j;
                    bool dummy = false;
                    const uint32_t mg = full_meta(acc, j, dummy);
                    if (mg & M_VALID) {
                        const uint32_t mt = (mg >> 28) & 3u;
                        if ((mg & M_IDENT) == my_ident) {
                            head = 0u;
                            first = 0u;
                            break;
                        }
                        if (mt == my_mate) head = 0u;
                        if (mt > my_mate) gb = 1u;
                    }
                    if (mg & M_RUN) break;
                    if (++steps > kLookBackMax) {
                        too_long = true;
                        break;
                    }
                }
            }
            head &= first;  // a duplicate of an earlier record is never a head (that record has its mate)
            gb &= first;    // k_emit reads the flag of `first` records only
            if (lpos < n_here)
                s_fl[lpos] = static_cast<uint8_t>((my_mate << FL_MATE_SHIFT) | (((me >> 30) & 1u) << 4) | head |
                                                  (first << 1) | (gb << 5));
            nh += head;
            nf += first;
            nv += valid;
            any_gb |= gb;
        }
        PROF_T(t6);
        PROF_ADD(0, t0, t1);
        PROF_ADD(1, t1, t2);
        PROF_ADD(2, t2, t3);
        PROF_ADD(3, t3, t4);
        PROF_ADD(4, t4, t5);
        PROF_ADD(5, t5, t6);
        PROF_ADD(6, t0, t0 + 1);
        PROF_ADD(7, 0, t0);
        __syncthreads();
        if (n_here == static_cast<uint32_t>(kQTile)) {  // kQItems flag bytes per thread in one store
            if (kQItems == 8)
                reinterpret_cast<uint2*>(fl + base)[threadIdx.x] = reinterpret_cast<const uint2*>(s_flw)[threadIdx.x];
            else
                reinterpret_cast<uint32_t*>(fl + base)[threadIdx.x] = s_flw[threadIdx.x];
        } else {
            for (uint32_t j = threadIdx.x; j < n_here; j += kRBlock) fl[base + j] = s_fl[j];
        }
        __syncthreads();  // LDS is reused by the next pass
    }
    nh = r_wave_sum(nh);
    nf = r_wave_sum(nf);
    nv = r_wave_sum(nv);
    const uint32_t err = (__any(bad) ? ERR_REF_RANGE : 0u) | (__any(too_long) ? ERR_RUN_LENGTH : 0u);
    const bool wave_gb = __any(any_gb != 0u);
    if ((threadIdx.x & 63) == 0) {
        s_w[threadIdx.x >> 6] = make_uint2(nh, nf);
        s_v[threadIdx.x >> 6] = nv;
        if (err) atomicOr(&counters[CNT_ERR], err);
        if (wave_gb && counters[CNT_ANYGB] == 0u) atomicOr(&counters[CNT_ANYGB], 1u);  // mates interleave in this stream
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint2 t = make_uint2(0u, 0u);
        uint32_t v = 0;
#pragma unroll
        for (int w = 0; w < kRWaves; ++w) {
            t.x += s_w[w].x;
            t.y += s_w[w].y;
            v += s_v[w];
        }
        tile_cnt[tile] = t;
        if (Acc::kCountsMapped) tile_valid[tile] = v;  // summed into hits_count by k_scan_tiles (src/slimm.hpp:212)
    }
    __syncthreads();  // LDS is reused by the next tile
    }
}

// ---------------------------------------------------------------------------------------------------------
// runs_tagged: the body of k_runs for databases of fewer than 2^18 references (every BASELINE config).
//
// The plain walk (runs_walk) costs 23 vector instructions per step for a thread's four records (clamped address, LDS
// read, xor, mask, min) and runs to the longest run of the wave: 250 of 800 instructions per pass at 3 hits per read, 700
// of 1200 at 8.  Here the LDS word a walk compares carries the window position of the record's RUN START above {mapped,
// mate, reference}:
//     bits 31-21 run start | bit 20 mapped | bits 19-18 mate | bits 17-0 reference
// so "same read and same reference earlier in my run" is plain equality with my own word: records before my run's start
// belong to runs that start earlier, the walk needs no clamp (it may read past the window's first word into a zeroed
// guard, which never equals a mapped word) and no mask -- 8 instructions per step.  The run start is known only after
// the segment carries, so the tagged words are written in the second stage, which therefore also covers the halo (a
// fifth record for half the waves) behind one more barrier.  Both stages use ONE record-to-thread mapping (window index
// off + k * 256 + thread, plus the halo for the first 128 threads), so the second stage takes the record's word from
// registers and ballots again (a first version handed the ballots over through LDS and lost: 112 vs 83 us).  Measured
// (scripts/exp_walk_crossover.py, 10 M records): 72 vs 81 us at 3 hits per read, 88 vs 116 at 8, 125 vs 191 at 20.
// Everything else (flags, slow path for runs reaching back beyond the halo, outputs) is runs_walk's.
// ---------------------------------------------------------------------------------------------------------
constexpr uint32_t kTagRefBits = 18;
constexpr uint32_t kTagGuard = kQTile + kHalo + 4;  // zero words below the window: the unclamped walk may read them

__device__ __forceinline__ uint32_t tagged_word(uint32_t m, uint32_t run_start) {
    return (m & ((1u << kTagRefBits) - 1u)) | ((m >> 10) & 0xc0000u) | ((m >> 11) & 0x100000u) | (run_start << 21);
}

template <typename Acc>
__device__ __forceinline__ void runs_tagged(const Acc& acc, uint32_t ntiles, uint32_t* __restrict__ counters,
                                            uint8_t* __restrict__ fl, uint2* __restrict__ tile_cnt,
                                            uint32_t* __restrict__ tile_valid) {
    __shared__ uint32_t s_tagw[kTagGuard + kQTile + kHalo];
    __shared__ uint32_t s_flw[kQTile / 4];
    __shared__ uint32_t s_seg[kSegs];
    __shared__ uint64_t s_last[kSegs + 2];
    __shared__ uint2 s_w[kRWaves];
    __shared__ uint32_t s_v[kRWaves];
    uint8_t* s_fl = reinterpret_cast<uint8_t*>(s_flw);
    uint32_t* const s_tag = s_tagw + kTagGuard;
    for (uint32_t i = threadIdx.x; i < kTagGuard; i += kRBlock) s_tagw[i] = 0u;  // (first read after several barriers)
    const uint32_t N = acc.count(counters);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    const uint64_t le_mask = lt_mask | (1ull << lane);
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    uint32_t nh = 0, nf = 0, nv = 0, any_gb = 0;
    bool bad = false, too_long = false;
    for (uint32_t base = tile * kRTile; base < min((tile + 1u) * kRTile, N); base += kQTile) {
        const uint32_t lds_lo = base >= kHalo ? base - kHalo : 0u;
        const uint32_t lds_hi = min(base + static_cast<uint32_t>(kQTile), N);
        const uint32_t off = base - lds_lo, n_here = lds_hi - base, W = lds_hi - lds_lo;
        const uint32_t nseg = (W + 63u) >> 6;
        const bool halo_mine = threadIdx.x < off;  // off is 0 or 128: waves 0 and 1 also take the halo's two segments
        // 1. meta word of every record of the window and its segment's summary (slot kQItems = the halo)
        uint32_t m_[kQItems + 1];
        {
            typename Acc::Raw raw[kQItems + 1];
#pragma unroll
            for (int k = 0; k < kQItems; ++k) raw[k] = acc.load(min(base + k * kRBlock + threadIdx.x, N - 1u));
            raw[kQItems] = acc.load(min(lds_lo + threadIdx.x, N - 1u));
            const uint64_t before = acc.raw_key(lds_lo ? lds_lo - 1u : 0u);
            if (lane == 63) {
#pragma unroll
                for (int k = 0; k < kQItems; ++k) s_last[(off >> 6) + k * kRWaves + wave] = Acc::key_bits(raw[k]);
                if (halo_mine) s_last[wave] = Acc::key_bits(raw[kQItems]);
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k <= kQItems; ++k) {
                const uint32_t j = k < kQItems ? off + k * kRBlock + threadIdx.x : threadIdx.x;  // window index
                const bool active = k < kQItems ? (j & ~63u) < W : halo_mine;                      // wave-uniform
                uint32_t m = 0u;
                if (active) {
                    const uint64_t mine = Acc::key_bits(raw[k]);
                    const uint32_t seg = j >> 6;
                    const uint64_t prev = previous_key(mine, seg ? s_last[seg - 1u] : before);
                    if (j < W) {
                        m = acc.meta(raw[k], bad);
                        if (lds_lo + j == 0u || !Acc::same_run(mine, prev)) m |= M_RUN;
                    }
                    const uint32_t summary = segment_summary(m);
                    if (lane == 0) s_seg[seg] = summary;
                }
                m_[k] = m;
            }
        }
        __syncthreads();
        // 2. carry into each segment
        segment_carries(s_seg, nseg, lds_lo == 0);
        // 3. head / larger-mate-before / distance to the run start from the ballots and the carry; the tagged word
        uint32_t me_[kQItems], wl_[kQItems], st_[kQItems];
#pragma unroll
        for (int k = 0; k <= kQItems; ++k) {
            const uint32_t j = k < kQItems ? off + k * kRBlock + threadIdx.x : threadIdx.x;
            const bool active = k < kQItems ? (j & ~63u) < W : halo_mine;
            uint32_t t = 0u, wl = 0u, st = 0u;
            if (active) {
                const uint32_t me = m_[k];
                const RunScan rs = run_scan(me, s_seg[j >> 6], true, lane, lt_mask, le_mask);
                t = tagged_word(me, j - rs.len);  // (a run reaching back beyond the window: position 0)
                if (j < W) s_tag[j] = t;
                wl = (me >> 31) ? min(rs.len, j) : 0u;
                st = rs.head | (rs.gb << 1) | (rs.open << 2);
            }
            if (k < kQItems) {
                me_[k] = t;
                wl_[k] = wl;
                st_[k] = st;
            }
        }
        __syncthreads();
        // 4. duplicate walk: four steps per trip, equality with my own tagged word
        uint32_t dup_[kQItems];
        {
            uint32_t longest = 0u;
            uint32_t b_[kQItems];  // byte offset into s_tagw (an offset, not a pointer: pointer arrays decay to flat addressing)
#pragma unroll
            for (int k = 0; k < kQItems; ++k) {
                longest = max(longest, wl_[k]);
                dup_[k] = 0xffffffffu;
                b_[k] = (kTagGuard + off + k * kRBlock + threadIdx.x) * 4u;
            }
            const char* tag_bytes = reinterpret_cast<const char*>(s_tagw);
            for (uint32_t d = 0; r_ballot(d < longest) != 0ull; d += 4) {
#pragma unroll
                for (int k = 0; k < kQItems; ++k) {
                    b_[k] -= 16u;
                    const uint32_t* q = reinterpret_cast<const uint32_t*>(tag_bytes + b_[k]);
                    const uint32_t x0 = q[3] ^ me_[k], x1 = q[2] ^ me_[k], x2 = q[1] ^ me_[k], x3 = q[0] ^ me_[k];
                    dup_[k] = min(min(dup_[k], min(x0, x1)), min(x2, x3));
                }
            }
        }
#pragma unroll
        for (int k = 0; k < kQItems; ++k) {
            const uint32_t lpos = k * kRBlock + threadIdx.x;
            const uint32_t me = m_[k], valid = me >> 31, my_mate = (me >> 28) & 3u;
            uint32_t head = st_[k] & 1u, gb = (st_[k] >> 1) & 1u;
            uint32_t first = valid & (dup_[k] != 0u ? 1u : 0u);  // (no walk: dup_ stayed all ones)
            if (st_[k] & 4u) {  // the run reaches back beyond the halo: go on in global memory (rare)
                const uint32_t my_ident = me & M_IDENT;
                uint32_t j = lds_lo, steps = 0;
                while (j > 0 && first) {
                    --j;
                    bool dummy = false;
                    const uint32_t mg = full_meta(acc, j, dummy);
                    if (mg & M_VALID) {
                        const uint32_t mt = (mg >> 28) & 3u;
                        if ((mg & M_IDENT) == my_ident) {
                            head = 0u;
                            first = 0u;
                            break;
                        }
                        if (mt == my_mate) head = 0u;
                        if (mt > my_mate) gb = 1u;
                    }
                    if (mg & M_RUN) break;
                    if (++steps > kLookBackMax) {
                        too_long = true;
                        break;
                    }
                }
            }
            head &= first;
            gb &= first;
            if (lpos < n_here)
                s_fl[lpos] = static_cast<uint8_t>((my_mate << FL_MATE_SHIFT) | (((me >> 30) & 1u) << 4) | head |
                                                  (first << 1) | (gb << 5));
            nh += head;
            nf += first;
            nv += valid;
            any_gb |= gb;
        }
        __syncthreads();
        if (n_here == static_cast<uint32_t>(kQTile)) {
            if (kQItems == 8)
                reinterpret_cast<uint2*>(fl + base)[threadIdx.x] = reinterpret_cast<const uint2*>(s_flw)[threadIdx.x];
            else
                reinterpret_cast<uint32_t*>(fl + base)[threadIdx.x] = s_flw[threadIdx.x];
        } else {
            for (uint32_t j = threadIdx.x; j < n_here; j += kRBlock) fl[base + j] = s_fl[j];
        }
        __syncthreads();  // LDS is reused by the next pass
    }
    nh = r_wave_sum(nh);
    nf = r_wave_sum(nf);
    nv = r_wave_sum(nv);
    const uint32_t err = (__any(bad) ? ERR_REF_RANGE : 0u) | (__any(too_long) ? ERR_RUN_LENGTH : 0u);
    const bool wave_gb = __any(any_gb != 0u);
    if ((threadIdx.x & 63) == 0) {
        s_w[threadIdx.x >> 6] = make_uint2(nh, nf);
        s_v[threadIdx.x >> 6] = nv;
        if (err) atomicOr(&counters[CNT_ERR], err);
        if (wave_gb && counters[CNT_ANYGB] == 0u) atomicOr(&counters[CNT_ANYGB], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint2 t = make_uint2(0u, 0u);
        uint32_t v = 0;
#pragma unroll
        for (int w = 0; w < kRWaves; ++w) {
            t.x += s_w[w].x;
            t.y += s_w[w].y;
            v += s_v[w];
        }
        tile_cnt[tile] = t;
        if (Acc::kCountsMapped) tile_valid[tile] = v;
    }
    __syncthreads();
    }
}

// The look-back classification launch: the plain walk (mode 0) or the tagged-word walk (mode 2), as picked on the device.
template <typename Acc>
__global__ __launch_bounds__(kRBlock, SLIMM_RUNS_MINBLOCKS) void k_runs(const Acc acc, uint32_t ntiles,
                                                                         uint32_t* __restrict__ counters,
                                                                         uint8_t* __restrict__ fl, uint2* __restrict__ tile_cnt,
                                                                         uint32_t* __restrict__ tile_valid) {
    const uint32_t mode = counters[CNT_MODE];
    if (mode == 0u)
        runs_walk(acc, ntiles, counters, fl, tile_cnt, tile_valid);
    else if (mode == 2u)
        runs_tagged(acc, ntiles, counters, fl, tile_cnt, tile_valid);
}

// ---------------------------------------------------------------------------------------------------------
// k_runs_hash: the same classification with an O(1) duplicate test.  The duplicate walk of k_runs costs O(records of
// the read) per record and a whole wave waits for its longest run -- fine at 3-8 hits per read, the dominant cost at 40
// (BASELINE config 5).  Here `head` and `larger mate before` come from the same segment ballots + carries as in k_runs,
// and every mapped record of the staged window puts ONE key, (run start, mate, ref), into an LDS hash table with
// atomicMin(window index): `first` <=> the minimum of its key is the record itself.  Records whose run starts before the
// staged window (512 records of back halo) fall back to the global look-back walk.
// One 64-bit LDS word per entry: (run start 12 bits | mate 2 | ref 28) << 12 | window index.
// ---------------------------------------------------------------------------------------------------------
constexpr int kHBlock = 512;
constexpr uint32_t kHHalo = 512;
constexpr uint32_t kHWin = kRTile + kHHalo;        // 2560 staged records
constexpr uint32_t kHSlots = 8192;                 // one entry per mapped staged record: load <= 0.32
constexpr uint64_t kEmptySlot = ~0ull;

__device__ __forceinline__ uint32_t hash_slot(uint64_t key) {
    key ^= key >> 23;
    key *= 0x2127599bf4325c37ULL;
    key ^= key >> 29;
    return static_cast<uint32_t>(key) & (kHSlots - 1);
}

// insert (key, idx) keeping the minimum idx; returns the slot
__device__ __forceinline__ uint32_t hash_put_min(uint64_t* tab, uint64_t key, uint32_t idx) {
    const uint64_t val = (key << 12) | idx;
    uint32_t s = hash_slot(key);
    while (true) {
        uint64_t cur = tab[s];
        if (cur == kEmptySlot) {
            const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&tab[s]),
                                                     static_cast<unsigned long long>(kEmptySlot),
                                                     static_cast<unsigned long long>(val));
            if (old == kEmptySlot) return s;
            cur = old;
        }
        if ((cur >> 12) == key) {
            atomicMin(reinterpret_cast<unsigned long long*>(&tab[s]), static_cast<unsigned long long>(val));
            return s;
        }
        s = (s + 1) & (kHSlots - 1);
    }
}

// minimum idx stored for key, or 0xffffffff when the key is absent
__device__ __forceinline__ uint32_t hash_get_min(const uint64_t* tab, uint64_t key) {
    uint32_t s = hash_slot(key);
    while (true) {
        const uint64_t cur = tab[s];
        if (cur == kEmptySlot) return 0xffffffffu;
        if ((cur >> 12) == key) return static_cast<uint32_t>(cur) & 0xfffu;
        s = (s + 1) & (kHSlots - 1);
    }
}

template <typename Acc>
__global__ __launch_bounds__(kHBlock) void k_runs_hash(const Acc acc, uint32_t ntiles, uint32_t* __restrict__ counters,
                                                       uint8_t* __restrict__ fl, uint2* __restrict__ tile_cnt,
                                                       uint32_t* __restrict__ tile_valid, uint32_t* __restrict__ chunk_acc,
                                                       uint32_t chunk_shift) {
    constexpr int kSlots = kHWin / kHBlock;       // 5 staged records per thread
    constexpr uint32_t kHSegs = kHWin / 64;       // 40 segments
    constexpr int kHWaves = kHBlock / 64;
    __shared__ uint64_t s_tab[kHSlots];
    __shared__ uint32_t s_meta[kHWin];
    __shared__ uint32_t s_seg[kHSegs];            // segment summaries, then the carry into each segment (as in k_runs)
    __shared__ uint64_t s_last[kHSegs];           // key of the last record of each segment (staging)
    __shared__ uint2 s_w[kHWaves];
    __shared__ uint32_t s_v[kHWaves];
    if (counters[CNT_MODE] != 1u) {  // k_pick_runs chose the look-back kernel for this stream: k_runs has classified it
        // ... and this launch, which would otherwise return at once, sums its per-tile counts by chunk for k_emit
        const uint32_t chunk = 1u << chunk_shift;  // <= 256 tiles: the first four waves, one tile per thread
        if (chunk_acc && (blockIdx.x << chunk_shift) < ntiles) {
            const uint32_t t = (blockIdx.x << chunk_shift) + threadIdx.x, tc = min(t, ntiles - 1u);
            uint2 c = tile_cnt[tc];
            uint32_t v = Acc::kCountsMapped ? tile_valid[tc] : 0u;
            if (t >= ntiles || threadIdx.x >= chunk) {
                c = make_uint2(0u, 0u);
                v = 0u;
            }
            c.x = r_wave_sum(c.x);
            c.y = r_wave_sum(c.y);
            v = r_wave_sum(v);
            if ((threadIdx.x & 63u) == 0u) {
                s_w[threadIdx.x >> 6] = c;
                s_v[threadIdx.x >> 6] = v;
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                uint4 sum = make_uint4(0u, 0u, 0u, 0u);
                for (uint32_t w = 0; w < (chunk + 63u) / 64u; ++w) {
                    sum.x += s_w[w].x;
                    sum.y += s_w[w].y;
                    sum.z += s_v[w];
                }
                *reinterpret_cast<uint4*>(chunk_acc + blockIdx.x * 4u) = sum;
            }
        }
        return;
    }
    const uint32_t N = acc.count(counters);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    const uint64_t le_mask = lt_mask | (1ull << lane);
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const uint32_t base = tile * kRTile;
    uint32_t nh = 0, nf = 0, nv = 0, any_gb = 0;
    bool bad = false, too_long = false;
    if (base < N) {
        const uint32_t lds_lo = base >= kHHalo ? base - kHHalo : 0u;
        const uint32_t lds_hi = min(base + static_cast<uint32_t>(kRTile), N);
        const uint32_t off = base - lds_lo, wn = lds_hi - lds_lo;  // staged records
        const uint32_t nseg = (wn + 63u) >> 6;
        for (uint32_t i = tid; i < kHSlots; i += kHBlock) s_tab[i] = kEmptySlot;
        // 1. stage the window like k_runs (all loads up front with clamped indices, previous key by DPP lane shift) and
        //    summarise every 64-record segment
        {
            typename Acc::Raw raw[kSlots];
#pragma unroll
            for (int k = 0; k < kSlots; ++k) raw[k] = acc.load(min(lds_lo + k * kHBlock + tid, N - 1u));
            const uint64_t before = acc.raw_key(lds_lo ? lds_lo - 1u : 0u);
            if (lane == 63) {
#pragma unroll
                for (int k = 0; k < kSlots; ++k) s_last[k * kHWaves + wave] = Acc::key_bits(raw[k]);
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < kSlots; ++k) {
                const uint32_t j = k * kHBlock + tid;
                if ((j & ~63u) < wn) {  // wave-uniform
                    const uint64_t mine = Acc::key_bits(raw[k]);
                    const uint32_t seg = j >> 6;
                    const uint64_t prev = previous_key(mine, seg ? s_last[seg - 1u] : before);
                    uint32_t m = 0u;
                    if (j < wn) {
                        m = acc.meta(raw[k], bad);
                        if (lds_lo + j == 0u || !Acc::same_run(mine, prev)) m |= M_RUN;
                        s_meta[j] = m;
                    }
                    const uint32_t summary = segment_summary(m);
                    if (lane == 0) s_seg[seg] = summary;
                }
            }
        }
        __syncthreads();
        // 2. carry into each segment (mates seen and records since the run start)
        segment_carries(s_seg, nseg, lds_lo == 0);
        // 3. head / larger-mate-before from the ballots of the record's segment plus the carry (k_runs); every mapped
        //    staged record whose run starts inside the window enters the table once, keyed (run start, mate, ref), with
        //    atomicMin(window index): `first` <=> the minimum of the key is the record itself.  (The first version also
        //    kept (run, mate, ANY) entries to answer the head / larger-mate questions: three table operations more per
        //    record, the bulk of its time.)
        uint32_t me_[kSlots], st_[kSlots], rs_[kSlots];  // meta word; {head, gb, open} bits; window index of the run start
#pragma unroll
        for (int k = 0; k < kSlots; ++k) {
            const uint32_t j = k * kHBlock + tid;
            const bool live = j < wn;
            const uint32_t me = live ? s_meta[j] : M_RUN;
            const RunScan rs = run_scan(me, live ? s_seg[j >> 6] : 0u, live, lane, lt_mask, le_mask);
            const uint32_t valid = me >> 31, open = rs.open;
            me_[k] = me;
            rs_[k] = j - min(rs.len, j);
            st_[k] = rs.head | (rs.gb << 1) | (rs.open << 2);
            if (valid && !open)
                hash_put_min(s_tab, (static_cast<uint64_t>(rs_[k]) << 30) | (me & M_IDENT), j);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kSlots; ++k) {
            const uint32_t j = k * kHBlock + tid;
            if (j < off || j >= wn) continue;  // halo records belong to the tile before; slots past the end
            const uint32_t i = lds_lo + j;
            const uint32_t me = me_[k], valid = me >> 31, my_mate = (me >> 28) & 3u, my_ident = me & M_IDENT;
            uint32_t head = st_[k] & 1u, gb = (st_[k] >> 1) & 1u, first = valid;
            if (valid && !(me & M_RUN)) {
                if (!(st_[k] & 4u)) {
                    first = hash_get_min(s_tab, (static_cast<uint64_t>(rs_[k]) << 30) | my_ident) == j ? 1u : 0u;
                } else {  // the run starts before the staged window: walk back through global memory (rare)
                    head = 1u;
                    gb = 0u;
                    uint32_t q = i, steps = 0;
                    while (q > 0) {
                        --q;
                        bool dummy = false;
                        const uint32_t m = (q >= lds_lo) ? s_meta[q - lds_lo] : full_meta(acc, q, dummy);
                        if (m & M_VALID) {
                            const uint32_t mt = (m >> 28) & 3u;
                            if ((m & M_IDENT) == my_ident) {
                                head = 0u;
                                first = 0u;
                                break;
                            }
                            if (mt == my_mate) head = 0u;
                            if (mt > my_mate) gb = 1u;
                        }
                        if (m & M_RUN) break;
                        if (++steps > kLookBackMax + kHHalo) {
                            too_long = true;
                            break;
                        }
                    }
                }
            }
            head &= first;
            gb &= first;
            const uint32_t f = (my_mate << FL_MATE_SHIFT) | ((me & M_RUN) ? FL_RUN_START : 0u) | (head ? FL_HEAD : 0u) |
                               (first ? FL_FIRST : 0u) | (gb ? FL_GREATER_BEFORE : 0u);
            nh += head;
            nf += first;
            nv += valid;
            any_gb |= gb;
            fl[i] = static_cast<uint8_t>(f);
        }
    }
    nh = r_wave_sum(nh);
    nf = r_wave_sum(nf);
    nv = r_wave_sum(nv);
    const uint32_t err = (__any(bad) ? ERR_REF_RANGE : 0u) | (__any(too_long) ? ERR_RUN_LENGTH : 0u);
    const bool wave_gb = __any(any_gb != 0u);
    if (lane == 0) {
        s_w[wave] = make_uint2(nh, nf);
        s_v[wave] = nv;
        if (err) atomicOr(&counters[CNT_ERR], err);
        if (wave_gb && counters[CNT_ANYGB] == 0u) atomicOr(&counters[CNT_ANYGB], 1u);
    }
    __syncthreads();
    if (tid == 0) {
        uint2 t = make_uint2(0u, 0u);
        uint32_t v = 0;
#pragma unroll
        for (int w = 0; w < kHWaves; ++w) {
            t.x += s_w[w].x;
            t.y += s_w[w].y;
            v += s_v[w];
        }
        tile_cnt[tile] = t;
        if (Acc::kCountsMapped) tile_valid[tile] = v;
        if (chunk_acc) add_chunk_sums<Acc>(chunk_acc, chunk_shift, tile, t, v);
    }
    __syncthreads();  // LDS is reused by the next tile
    }
}

// ---------------------------------------------------------------------------------------------------------
// k_pick_runs: which classification kernel suits this stream?  Samples the first records, measures records per qName
// run, and writes the choice to counters[CNT_MODE]; both kernels are launched and the one not chosen returns at once.
// Measured per 10 M records: look-back 82 us at 3 records/run, 100 at 10, 249 at 42; hash table ~125 at any depth.
// ---------------------------------------------------------------------------------------------------------
constexpr uint32_t kPickSample = 2048;  // one trip of k_zero_pick (8 records per thread); 8192 cost 3 us more per file
constexpr uint32_t kPickHashAbove = 16;  // records per qName run above which the hash-table kernel beats the plain walk
constexpr uint32_t kPickHashAboveTagged = 28;  // ... and the tagged-word walk (125 vs 148 us at 24 per run, level at 29)
constexpr uint32_t kPickTagAbove = 0;    // ... and the tagged-word walk: it wins at every depth (scripts/exp_walk_crossover.py:
                                         // 72 vs 81 us at 3 hits per read, 100 vs 144 at 12); the plain walk serves >= 2^18 references
__device__ __forceinline__ uint32_t pick_mode(uint32_t sample, uint32_t runs, uint32_t tag_ok, int force) {
    uint32_t mode = 0u;
    if (runs != 0u) {
        const uint32_t per_run = sample / runs;
        if (tag_ok && per_run >= kPickTagAbove)
            mode = per_run > kPickHashAboveTagged ? 1u : 2u;
        else
            mode = per_run > kPickHashAbove ? 1u : 0u;
    }
    if (force >= 0) mode = (force == 2 && !tag_ok) ? 0u : static_cast<uint32_t>(force);
    return mode;
}

template <typename Acc>
__global__ __launch_bounds__(1024) void k_pick_runs(const Acc acc, uint32_t* __restrict__ counters, int force,
                                                    uint32_t tag_ok) {
    __shared__ uint32_t s_runs[16];
    const uint32_t N = acc.count(counters);
    const uint32_t S = min(N, kPickSample);
    uint32_t runs = 0;
    for (uint32_t i = threadIdx.x; i < S; i += 1024) runs += (i == 0 || acc.key_of(i) != acc.key_of(i - 1)) ? 1u : 0u;
    runs = r_wave_sum(runs);
    if ((threadIdx.x & 63) == 0) s_runs[threadIdx.x >> 6] = runs;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < 16; ++w) t += s_runs[w];
        counters[CNT_MODE] = pick_mode(S, t, tag_ok, force);
    }
}

// k_zero + k_pick_runs in one launch: every workgroup clears its share of the arrays; workgroup 0 clears the counters,
// then samples the first records and writes the choice.  (Two single-purpose launches cost ~6 us each in launch and
// drain latency.)
template <typename Acc>
__global__ __launch_bounds__(256) void k_zero_pick(const ZeroArgs z, const Acc acc, uint32_t* __restrict__ counters,
                                                   int force, uint32_t tag_ok) {
    __shared__ uint32_t s_runs[4];
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x, gsz = gridDim.x * 256;
#pragma unroll
    for (int k = 0; k < 5; ++k)
        for (uint32_t i = gid; i < z.n[k]; i += gsz) z.p[k][i] = 0u;
    for (uint32_t i = gid; i < z.n64; i += gsz) z.p64[i] = ~0ull;
    if (blockIdx.x != 0) return;
    if (threadIdx.x < CNT_WORDS) counters[threadIdx.x] = 0u;
    const uint32_t N = acc.count(counters);  // (raw records: a kernel argument, not a counter)
    const uint32_t S = min(N, kPickSample);
    if (S == 0) {
        if (threadIdx.x == 0) counters[CNT_MODE] = pick_mode(0u, 0u, tag_ok, force);
        return;
    }
    uint32_t runs = 0;
    for (uint32_t i0 = 0; i0 < S; i0 += 8 * 256) {  // 8 records per thread and trip, their loads issued together
        uint64_t a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t i = min(i0 + u * 256 + threadIdx.x, S - 1u);
            a[u] = acc.key_of(i);
            b[u] = acc.key_of(i ? i - 1u : 0u);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t i = i0 + u * 256 + threadIdx.x;
            runs += (i < S && (i == 0 || a[u] != b[u])) ? 1u : 0u;
        }
    }
    runs = r_wave_sum(runs);
    if ((threadIdx.x & 63) == 0) s_runs[threadIdx.x >> 6] = runs;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = s_runs[0] + s_runs[1] + s_runs[2] + s_runs[3];
        counters[CNT_MODE] = pick_mode(S, t, tag_ok, force);
    }
}

template <typename Acc>
__global__ __launch_bounds__(kRBlock) void k_emit(const Acc acc, const uint8_t* __restrict__ fl,
                                                  uint32_t* __restrict__ counters, const uint2* __restrict__ tile_off,
                                                  uint32_t* __restrict__ tgt_ref, uint32_t* __restrict__ tgt_gbin,
                                                  uint32_t* __restrict__ read_off, const uint32_t* __restrict__ chunk_acc,
                                                  uint32_t* __restrict__ tail, uint32_t chunk_shift) {
    __shared__ uint8_t s_fl[kHalo + kRTile + kHalo];
    __shared__ uint2 s_w[kRItems][kRWaves];
    __shared__ uint2 s_pre[kRWaves];
    const uint32_t N = acc.count(counters);
    const uint32_t base = blockIdx.x * kRTile;
    if (base >= N) return;
    const uint32_t wave = threadIdx.x >> 6;
    if (chunk_acc) {  // my offset from the chunk sums and my chunk's tile counts (tile_off holds COUNTS here, not offsets)
        const uint32_t c = blockIdx.x >> chunk_shift, in_chunk = blockIdx.x - (c << chunk_shift);  // both < kRBlock
        uint2 p = make_uint2(0u, 0u);
        const uint2 cs = *reinterpret_cast<const uint2*>(chunk_acc + min(threadIdx.x, c ? c - 1u : 0u) * 4u);
        const uint2 ts = tile_off[(c << chunk_shift) + min(threadIdx.x, in_chunk ? in_chunk - 1u : 0u)];
        if (threadIdx.x < c) p = cs;
        if (threadIdx.x < in_chunk) {
            p.x += ts.x;
            p.y += ts.y;
        }
        p.x = r_wave_sum(p.x);
        p.y = r_wave_sum(p.y);
        if ((threadIdx.x & 63) == 0) s_pre[wave] = p;
        if (blockIdx.x == 0) {  // totals for the kernels and the host after this one
            const uint32_t nchunks = (gridDim.x + (1u << chunk_shift) - 1u) >> chunk_shift;  // <= kRBlock
            uint4 q = *reinterpret_cast<const uint4*>(chunk_acc + min(threadIdx.x, nchunks - 1u) * 4u);
            if (threadIdx.x >= nchunks) q = make_uint4(0u, 0u, 0u, 0u);
            __shared__ uint4 s_tot[kRWaves];
            q.x = r_wave_sum(q.x);
            q.y = r_wave_sum(q.y);
            q.z = r_wave_sum(q.z);
            if ((threadIdx.x & 63) == 0) s_tot[wave] = q;
            __syncthreads();
            if (threadIdx.x == 0) {
                uint4 t = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
                for (int w = 0; w < kRWaves; ++w) {
                    t.x += s_tot[w].x;
                    t.y += s_tot[w].y;
                    t.z += s_tot[w].z;
                }
                counters[CNT_M] = t.x;
                counters[CNT_P] = t.y;
                if (Acc::kCountsMapped) counters[CNT_V] = t.z;
                read_off[t.x] = t.y;  // the CSR sentinel
                if (tail) {           // the additive scalars that travel with the bins through the multi-GPU exchange
                    tail[0] = Acc::kCountsMapped ? t.z : counters[CNT_V];
                    tail[1] = t.x;
                    tail[2] = t.y;
                }
            }
        }
    }
    const uint32_t lds_lo = base >= kHalo ? base - kHalo : 0u;
    const uint32_t lds_hi = min(base + static_cast<uint32_t>(kRTile) + kHalo, N);
    // All global loads of the tile go out up front, with clamped indices instead of bounds branches (a load inside a
    // per-record branch is a serial round trip; see k_runs): the flag bytes of the window, then reference / position of
    // the thread's 8 records, then the contig geometry those references point at.
    {
        uint8_t fb[kRItems + 1];
#pragma unroll
        for (int k = 0; k <= kRItems; ++k) fb[k] = fl[min(lds_lo + k * kRBlock + threadIdx.x, N - 1u)];
#pragma unroll
        for (int k = 0; k <= kRItems; ++k) {
            const uint32_t j = k * kRBlock + threadIdx.x;
            if (j < lds_hi - lds_lo) s_fl[j] = fb[k];
        }
    }
    typename Acc::Hit hit[kRItems];
#pragma unroll
    for (int k = 0; k < kRItems; ++k) hit[k] = acc.hit(min(base + k * kRBlock + threadIdx.x, N - 1u));
    typename Acc::Geo geo[kRItems];
#pragma unroll
    for (int k = 0; k < kRItems; ++k) geo[k] = acc.geo(acc.hit_ref(hit[k]));
    __syncthreads();
    uint2 running;
    if (chunk_acc) {
        running = make_uint2(0u, 0u);
#pragma unroll
        for (int w = 0; w < kRWaves; ++w) {
            running.x += s_pre[w].x;
            running.y += s_pre[w].y;
        }
    } else {
        running = tile_off[blockIdx.x];
    }
    bool too_long = false;
    // A later record with a smaller mate exists only where some record carries the larger-mate-before flag; when no
    // record of the whole stream does (unpaired data, or mates not interleaved), the forward look is skipped.
    const bool mates_interleave = counters[CNT_ANYGB] != 0u;
    // heads / firsts of every 64-record segment (one barrier for the whole tile, then per-thread running sums)
#pragma unroll
    for (int k = 0; k < kRItems; ++k) {
        const uint32_t i = base + k * kRBlock + threadIdx.x;
        const uint32_t f = (i < N) ? s_fl[i - lds_lo] : 0u;
        const uint64_t mh = r_ballot((f & FL_HEAD) != 0u), mf = r_ballot((f & FL_FIRST) != 0u);
        if ((threadIdx.x & 63) == 0) s_w[k][wave] = make_uint2(__popcll(mh), __popcll(mf));
    }
    __syncthreads();
    // exclusive scan of the 32 (item, wave) counts by 32 lanes, once (every thread adding up its item's four counts with
    // selects cost 16 vector instructions per record slot: this kernel's vector ALUs are busy 2/3 of the time)
    if (threadIdx.x < kRItems * kRWaves) {
        const uint2 c = s_w[threadIdx.x / kRWaves][threadIdx.x % kRWaves];
        uint2 inc = c;
#pragma unroll
        for (int o = 1; o < kRItems * kRWaves; o <<= 1) {
            const uint32_t ax = __shfl_up(inc.x, o, 64), ay = __shfl_up(inc.y, o, 64);
            if (threadIdx.x >= static_cast<uint32_t>(o)) {
                inc.x += ax;
                inc.y += ay;
            }
        }
        s_w[threadIdx.x / kRWaves][threadIdx.x % kRWaves] = make_uint2(inc.x - c.x, inc.y - c.y);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kRItems; ++k) {
        const uint32_t i = base + k * kRBlock + threadIdx.x;
        const uint32_t f = (i < N) ? s_fl[i - lds_lo] : 0u;
        const bool head = f & FL_HEAD, first = f & FL_FIRST;
        const uint64_t mh = r_ballot(head), mf = r_ballot(first);
        const uint32_t rh = r_mask_rank(mh), rf = r_mask_rank(mf);
        const uint2 before = s_w[k][wave];  // heads / firsts of the tile before this wave's 64 records of item k
        const uint32_t mate = (f >> FL_MATE_SHIFT) & 3u;
        const uint32_t li = (i < N) ? i - lds_lo : 0u;
        uint32_t t = running.y + before.y + rf;
        uint32_t m = running.x + before.x + rh;
        // Reads of one qName run are laid out by ascending mate.  Both corrections are wave-uniform loops over the
        // distance with predicated bodies; runs leaving the staged window finish in global memory (rare).
        // (integer 0/1 lane state in VGPRs instead of bools: see k_runs)
        {   // earlier records of the run with a larger mate sit AFTER this one in the target order
            uint32_t act = (first && (f & FL_GREATER_BEFORE) && !(f & FL_RUN_START)) ? 1u : 0u;
            uint32_t openb = 0u;
            for (uint32_t d = 1; r_ballot(act != 0u) != 0ull; ++d) {
                const uint32_t diff = li - d;
                const uint32_t inn = act & ((~diff) >> 31);
                const uint32_t g = s_fl[diff & (0u - inn)];
                const uint32_t bigger = inn & ((mate - ((g >> FL_MATE_SHIFT) & 3u)) >> 31);  // its mate > mine
                t -= bigger & (g >> 1);   // FL_FIRST is bit 1
                m -= bigger & g;          // FL_HEAD is bit 0
                openb |= act & (inn ^ 1u);
                act = inn & (((g >> 4) & 1u) ^ 1u);  // stop at the run start (bit 4)
            }
            if (openb & 1u) {
                uint32_t j = lds_lo, steps = 0;
                while (j > 0) {
                    --j;
                    const uint32_t g = fl[j];
                    if (((g >> FL_MATE_SHIFT) & 3u) > mate) {
                        t -= (g & FL_FIRST) ? 1u : 0u;
                        m -= (g & FL_HEAD) ? 1u : 0u;
                    }
                    if (g & FL_RUN_START) break;
                    if (++steps > kRunWalkMax) {
                        too_long = true;
                        break;
                    }
                }
            }
        }
        {   // later records of the run with a smaller mate sit BEFORE this one
            const uint32_t room = lds_hi - lds_lo;  // staged records
            uint32_t act = (first && mate > 0 && mates_interleave) ? 1u : 0u;
            uint32_t openb = 0u;
            for (uint32_t d = 1; r_ballot(act != 0u) != 0ull; ++d) {
                const uint32_t j = li + d;
                const uint32_t inn = act & ((j - room) >> 31);           // j < room
                const uint32_t g = s_fl[j & (0u - inn)];
                const uint32_t go = inn & (((g >> 4) & 1u) ^ 1u);        // not a run start: still my run
                const uint32_t smaller = go & ((((g >> FL_MATE_SHIFT) & 3u) - mate) >> 31);  // its mate < mine
                t += smaller & (g >> 1);
                m += smaller & g;
                openb |= act & (inn ^ 1u) & (((lds_lo + j) - N) >> 31);  // window ended, the stream did not
                act = go;
            }
            if (openb & 1u) {
                uint32_t steps = 0;
                for (uint32_t j = lds_hi; j < N; ++j) {
                    const uint32_t g = fl[j];
                    if (g & FL_RUN_START) break;
                    if (((g >> FL_MATE_SHIFT) & 3u) < mate) {
                        t += (g & FL_FIRST) ? 1u : 0u;
                        m += (g & FL_HEAD) ? 1u : 0u;
                    }
                    if (++steps > kRunWalkMax) {
                        too_long = true;
                        break;
                    }
                }
            }
        }
        if (first) {  // a first record is mapped, so its reference is in range (k_runs checked)
            tgt_ref[t] = acc.hit_ref(hit[k]) | (head ? 0x80000000u : 0u);
            tgt_gbin[t] = acc.gbin(hit[k], geo[k]);
            if (head) read_off[m] = t;
        }
    }
    const bool any_long = __any(too_long);
    if (any_long && (threadIdx.x & 63) == 0) atomicOr(&counters[CNT_ERR], ERR_RUN_LENGTH);
}

static inline uint32_t rtiles(uint32_t n) { return (n + kRTile - 1) / kRTile; }

// SLIMM_RUNS_KERNEL=walk|hash overrides the choice k_pick_runs makes on the device (for tests and A/B timing)
// the tagged-word walk needs reference ids below 2^18; SLIMM_RUNS_TAGGED=0 takes it out of the choice
static uint32_t tagged_ok(uint32_t n_refs) {
    if (n_refs >= (1u << kTagRefBits)) return 0u;
    const char* e = getenv("SLIMM_RUNS_TAGGED");
    return (e && e[0] == '0') ? 0u : 1u;
}

static int forced_runs_mode() {
    const char* e = getenv("SLIMM_RUNS_KERNEL");
    if (!e) return -1;
    return e[0] == 'h' ? 1 : (e[0] == 'w' ? 0 : (e[0] == 't' ? 2 : -1));
}

// grid of k_runs (grid-stride over tiles beyond it)
static uint32_t runs_grid() {
    const char* e = getenv("SLIMM_RUNS_GRID");
    const long v = e ? atol(e) : 0;
    return v > 0 ? static_cast<uint32_t>(v) : 16384u;
}

static RawRecords make_raw(const DeviceRecords& in, uint32_t n_refs, const uint32_t* ref_len, const uint32_t* bin_off,
                           uint32_t half_read, uint32_t bin_width, const uint2* geo = nullptr) {
    RawRecords a;
    a.key = in.key;
    a.ref = in.ref;
    a.pos = in.pos;
    a.flag = in.flag;
    a.n = in.n;
    a.n_refs = n_refs;
    a.ref_len = ref_len;
    a.bin_off = bin_off;
    a.geo_tab = geo;
    a.half_read = half_read;
    a.bin_width = bin_width;
    a.bw_magic = bin_width ? 0xffffffffu / bin_width : 0u;
    return a;
}

void launch_zero_pick_raw(hipStream_t st, const ZeroArgs& z, const DeviceRecords& in, uint32_t* counters, uint32_t n_refs) {
    uint32_t most = z.n64;
    for (int k = 0; k < 5; ++k) most = most > z.n[k] ? most : z.n[k];
    const uint32_t blocks = std::min<uint32_t>(1024u, (most + 255u) / 256u + 1u);
    const RawRecords a = make_raw(in, 0, nullptr, nullptr, 0, 1);
    hipLaunchKernelGGL(k_zero_pick<RawRecords>, dim3(blocks), dim3(256), 0, st, z, a, counters, forced_runs_mode(),
                       tagged_ok(n_refs));
}

void launch_runs_raw(hipStream_t st, const DeviceRecords& in, uint32_t n_refs, const uint32_t* ref_len,
                     const uint32_t* bin_off, uint32_t half_read, uint32_t bin_width, uint32_t* counters, uint8_t* fl,
                     uint2* tile_cnt, uint32_t* tile_valid, int part, uint32_t* chunk_acc, uint32_t chunk_shift) {
    const uint32_t nt = rtiles(in.n);
    if (!nt) return;
    const RawRecords a = make_raw(in, n_refs, ref_len, bin_off, half_read, bin_width);
    if (part == 0)
        hipLaunchKernelGGL(k_pick_runs<RawRecords>, dim3(1), dim3(1024), 0, st, a, counters, forced_runs_mode(),
                           tagged_ok(n_refs));
    else if (part == 1)
        hipLaunchKernelGGL(k_runs<RawRecords>, dim3(std::min(nt, runs_grid())), dim3(kRBlock), 0, st, a, nt, counters, fl,
                           tile_cnt, tile_valid);
    else
        hipLaunchKernelGGL(k_runs_hash<RawRecords>, dim3(std::min(nt, 512u)), dim3(kHBlock), 0, st, a, nt, counters, fl,
                           tile_cnt, tile_valid, chunk_acc, chunk_shift);
}

void launch_emit_raw(hipStream_t st, const DeviceRecords& in, uint32_t n_refs, const uint32_t* ref_len,
                     const uint32_t* bin_off, const uint2* geo, uint32_t half_read, uint32_t bin_width, const uint8_t* fl,
                     uint32_t* counters,
                     const uint2* tile_off, uint32_t* tgt_ref, uint32_t* tgt_gbin, uint32_t* read_off,
                     const uint32_t* chunk_acc, uint32_t* tail, uint32_t chunk_shift) {
    const uint32_t nt = rtiles(in.n);
    if (!nt) return;
    hipLaunchKernelGGL(k_emit<RawRecords>, dim3(nt), dim3(kRBlock), 0, st,
                       make_raw(in, n_refs, ref_len, bin_off, half_read, bin_width, geo), fl, counters, tile_off, tgt_ref,
                       tgt_gbin, read_off, chunk_acc, tail, chunk_shift);
}

void launch_runs_sorted(hipStream_t st, uint32_t n_upper, const uint64_t* ident, const uint32_t* cref, const uint32_t* cgbin,
                        uint32_t* counters, uint8_t* fl, uint2* tile_cnt, int part, uint32_t n_refs) {
    const uint32_t nt = rtiles(n_upper);
    if (!nt) return;
    SortedRecords a{ident, cref, cgbin};
    if (part == 0)
        hipLaunchKernelGGL(k_pick_runs<SortedRecords>, dim3(1), dim3(1024), 0, st, a, counters, forced_runs_mode(),
                           tagged_ok(n_refs));
    else if (part == 1)
        hipLaunchKernelGGL(k_runs<SortedRecords>, dim3(std::min(nt, runs_grid())), dim3(kRBlock), 0, st, a, nt, counters, fl,
                           tile_cnt, static_cast<uint32_t*>(nullptr));
    else
        hipLaunchKernelGGL(k_runs_hash<SortedRecords>, dim3(std::min(nt, 512u)), dim3(kHBlock), 0, st, a, nt, counters, fl,
                           tile_cnt, static_cast<uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr), 6u);
}

void launch_emit_sorted(hipStream_t st, uint32_t n_upper, const uint64_t* ident, const uint32_t* cref, const uint32_t* cgbin,
                        const uint8_t* fl, uint32_t* counters, const uint2* tile_off, uint32_t* tgt_ref, uint32_t* tgt_gbin,
                        uint32_t* read_off) {
    const uint32_t nt = rtiles(n_upper);
    if (!nt) return;
    SortedRecords a{ident, cref, cgbin};
    hipLaunchKernelGGL(k_emit<SortedRecords>, dim3(nt), dim3(kRBlock), 0, st, a, fl, counters, tile_off, tgt_ref, tgt_gbin,
                       read_off, static_cast<const uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr), 6u);
}

}  // namespace slimm

#if EXP == 9
extern "C" int slimm_debug_prof(unsigned long long* out, int n) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(slimm::g_prof), sizeof(unsigned long long) * n);
    return e == hipSuccess ? 0 : -1;
}
#endif
