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

namespace slimm {

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
    uint32_t half_read, bin_width;
    static constexpr bool kCountsMapped = true;
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
    __device__ uint32_t gbin_of(uint32_t i, uint32_t r) const {
        // uint32 wrap-around of int32 + uint32, then clamp to the contig length (src/slimm.hpp:200-201, Q3)
        const uint32_t center = min(static_cast<uint32_t>(pos[i]) + half_read, ref_len[r]);
        return bin_off[r] + center / bin_width;
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
    __device__ uint32_t gbin_of(uint32_t i, uint32_t) const { return cgbin[i]; }
};

template <typename Acc>
__device__ __forceinline__ uint32_t full_meta(const Acc& acc, uint32_t i, bool& bad) {
    uint32_t m = acc.meta_of(i, bad);
    if (i == 0 || acc.key_of(i) != acc.key_of(i - 1)) m |= M_RUN;
    return m;
}

template <typename Acc>
__global__ __launch_bounds__(kRBlock) void k_runs(const Acc acc, uint32_t ntiles, uint32_t* __restrict__ counters,
                                                  uint8_t* __restrict__ fl, uint2* __restrict__ tile_cnt,
                                                  uint32_t* __restrict__ tile_valid) {
    __shared__ uint32_t s_meta[kRTile + kHalo];
    __shared__ uint2 s_w[kRWaves];
    __shared__ uint32_t s_v[kRWaves];
    if (counters[CNT_MODE] != 0u) return;  // k_pick_runs chose the hash-table kernel for this stream
    const uint32_t N = acc.count(counters);
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {  // grid-stride over tiles: the launch stays
    const uint32_t base = tile * kRTile;                                  // small, so not being chosen costs nothing
    uint32_t nh = 0, nf = 0, nv = 0;
    bool bad = false, too_long = false, any_gb = false;
    if (base < N) {
        const uint32_t lds_lo = base >= kHalo ? base - kHalo : 0u;
        const uint32_t lds_hi = min(base + static_cast<uint32_t>(kRTile), N);
        for (uint32_t i = lds_lo + threadIdx.x; i < lds_hi; i += kRBlock) s_meta[i - lds_lo] = full_meta(acc, i, bad);
        __syncthreads();
        for (int k = 0; k < kRItems; ++k) {
            const uint32_t i = base + k * kRBlock + threadIdx.x;
            const bool live = i < N;
            const uint32_t li = live ? i - lds_lo : 0u;
            const uint32_t me = live ? s_meta[li] : M_RUN;  // dead lanes: not mapped, never walk
            const bool valid = me & M_VALID;
            const uint32_t my_ident = me & M_IDENT, my_mate = (me >> 28) & 3u;
            // Look-back through LDS, one distance per trip for the whole wave.  The per-lane state lives in VGPRs as
            // 0/1 integers and is updated with plain integer arithmetic: C++ bools would become 64-bit lane masks in
            // SGPRs, and the CU's single scalar ALU -- shared by all four SIMDs -- was the measured bottleneck
            // (62 M SALU instructions per launch at config 2).
            uint32_t act = (valid && !(me & M_RUN)) ? 1u : 0u;
            uint32_t headb = 1u, firstb = 1u, gbb = 0u, openb = 0u;
            for (uint32_t d = 1; __ballot(act != 0u) != 0ull; ++d) {
                const uint32_t diff = li - d;                         // negative (bit 31) once d > li
                const uint32_t inn = act & ((~diff) >> 31);           // still inside the staged window
                const uint32_t m = s_meta[diff & (0u - inn)];
                const uint32_t v = inn & (m >> 31);                   // a mapped record of this run
                const uint32_t x = (m ^ me) & M_IDENT;                // 0 <=> same read and same reference
                const uint32_t same = v & (((x | (0u - x)) >> 31) ^ 1u);
                const uint32_t xm = x >> 28;                          // 0 <=> same mate
                const uint32_t eqm = ((xm | (0u - xm)) >> 31) ^ 1u;
                const uint32_t gt = (my_mate - ((m >> 28) & 3u)) >> 31;  // its mate is larger than mine
                headb &= ~(v & eqm);
                firstb &= ~same;
                gbb |= v & gt;
                openb |= act & (inn ^ 1u);
                act = inn & (same ^ 1u) & (((m >> 30) & 1u) ^ 1u);    // stop at a duplicate or at the run start
            }
            bool head = headb & 1u, first = firstb & 1u, greater_before = gbb & 1u;
            const bool open = openb & 1u;  // ran out of staged records before the run start
            if (open) {  // the run reaches back beyond the halo: continue in global memory (rare)
                uint32_t j = lds_lo, steps = 0;
                while (j > 0) {
                    --j;
                    bool dummy = false;
                    const uint32_t m = full_meta(acc, j, dummy);
                    if (m & M_VALID) {
                        const uint32_t mt = (m >> 28) & 3u;
                        if ((m & M_IDENT) == my_ident) {
                            head = false;
                            first = false;
                            break;
                        }
                        head = head && (mt != my_mate);
                        greater_before = greater_before || (mt > my_mate);
                    }
                    if (m & M_RUN) break;
                    if (++steps > kLookBackMax) {
                        too_long = true;
                        break;
                    }
                }
            }
            head = head && valid;
            first = first && valid;
            any_gb = any_gb || (valid && greater_before);
            uint32_t f = (my_mate << FL_MATE_SHIFT) | ((me & M_RUN) ? FL_RUN_START : 0u) | (head ? FL_HEAD : 0u) |
                         (first ? FL_FIRST : 0u) | ((valid && greater_before) ? FL_GREATER_BEFORE : 0u);
            nh += head;
            nf += first;
            nv += valid;
            if (live) fl[i] = static_cast<uint8_t>(f);
        }
    }
    nh = r_wave_sum(nh);
    nf = r_wave_sum(nf);
    nv = r_wave_sum(nv);
    const uint32_t err = (__any(bad) ? ERR_REF_RANGE : 0u) | (__any(too_long) ? ERR_RUN_LENGTH : 0u);
    const bool wave_gb = __any(any_gb);
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
// k_runs_hash: the same classification in O(1) per record.  The look-back of k_runs costs O(records of the read) per
// record and a whole wave waits for its longest run -- fine at 3 hits per read, the dominant cost at 40 (BASELINE
// config 5: 4.2 ms of a 9.7 ms file).  Here every mapped record of the staged window puts
//     (run, mate, ref)  and  (run, mate, ANY)
// into an LDS hash table with atomicMin(record index); afterwards `first` <=> the minimum of its (run, mate, ref) entry
// is the record itself, `head` <=> likewise for (run, mate, ANY), and "an earlier record of the run has a larger mate"
// <=> the minimum of a larger mate's ANY entry is smaller than the record's index.  `run` is the window index of the
// run's first record (block-wide max-scan of the run-start flags).  Records whose run starts before the staged window
// (512 records of back halo) fall back to the global look-back walk.
// One 64-bit LDS word per entry: (run 12 bits | mate 2 | ref 28) << 12 | window index.
// ---------------------------------------------------------------------------------------------------------
constexpr int kHBlock = 512;
constexpr uint32_t kHHalo = 512;
constexpr uint32_t kHWin = kRTile + kHHalo;        // 2560 staged records
constexpr uint32_t kHSlots = 8192;                 // >= 2 entries per staged record at load <= 0.63
constexpr uint32_t kRefAny = 0x0fffffffu;          // not a reference id: slimm_create keeps n_refs below 2^28 - 1
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
                                                       uint32_t* __restrict__ tile_valid) {
    __shared__ uint64_t s_tab[kHSlots];
    __shared__ uint32_t s_meta[kHWin];
    __shared__ uint16_t s_rs[kHWin];     // window index of the run's first record + 1; 0 = starts before the window
    __shared__ uint32_t s_wmax[kHBlock / 64];
    __shared__ uint2 s_w[kHBlock / 64];
    __shared__ uint32_t s_v[kHBlock / 64];
    if (counters[CNT_MODE] != 1u) return;  // k_pick_runs chose the look-back kernel for this stream
    const uint32_t N = acc.count(counters);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const uint32_t base = tile * kRTile;
    uint32_t nh = 0, nf = 0, nv = 0;
    bool bad = false, too_long = false, any_gb = false;
    if (base < N) {
        const uint32_t lds_lo = base >= kHHalo ? base - kHHalo : 0u;
        const uint32_t lds_hi = min(base + static_cast<uint32_t>(kRTile), N);
        const uint32_t wn = lds_hi - lds_lo;  // staged records
        for (uint32_t i = tid; i < kHSlots; i += kHBlock) s_tab[i] = kEmptySlot;
        for (uint32_t i = lds_lo + tid; i < lds_hi; i += kHBlock) s_meta[i - lds_lo] = full_meta(acc, i, bad);
        __syncthreads();
        // run start of every staged record: inclusive max-scan of (run-start ? index + 1 : 0), 512 records per trip
        uint32_t carry = 0;
        for (uint32_t c0 = 0; c0 < wn; c0 += kHBlock) {
            const uint32_t j = c0 + tid;
            uint32_t v = (j < wn && (s_meta[j] & M_RUN)) ? j + 1 : 0u;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t a = __shfl_up(v, o, 64);
                if (lane >= static_cast<uint32_t>(o)) v = max(v, a);
            }
            if (lane == 63) s_wmax[wave] = v;
            __syncthreads();
            uint32_t before = carry, total = carry;
#pragma unroll
            for (int w = 0; w < kHBlock / 64; ++w) {
                const uint32_t t = s_wmax[w];
                if (w < static_cast<int>(wave)) before = max(before, t);
                total = max(total, t);
            }
            if (j < wn) s_rs[j] = static_cast<uint16_t>(max(v, before));
            carry = total;
            __syncthreads();
        }
        // every mapped staged record (halo included: it holds the predecessors) enters the table: always with its
        // (run, mate, ref); with (run, mate, ANY) only when the record before it is not a mapped record of the same
        // run and mate -- the first record of a (run, mate) always qualifies, and a read's dozens of consecutive
        // records no longer hammer one LDS word
        bool paired = false;
        for (uint32_t j = tid; j < wn; j += kHBlock) {
            const uint32_t me = s_meta[j], rs = s_rs[j];
            if ((me & M_VALID) && rs) {
                const uint64_t run = static_cast<uint64_t>(rs - 1) << 30;
                hash_put_min(s_tab, run | (me & M_IDENT), j);
                bool lead = true;
                if (j > 0 && !(me & M_RUN)) {
                    const uint32_t pm = s_meta[j - 1];
                    lead = !((pm & M_VALID) && ((pm ^ me) & 0x30000000u) == 0u);
                }
                if (lead) hash_put_min(s_tab, run | (me & 0x30000000u) | kRefAny, j);
                paired = paired || (me & 0x30000000u);
            }
        }
        const bool any_paired = __syncthreads_or(paired);  // no mate numbers in the window: nothing can precede with a larger one
        for (int k = 0; k < kRTile / kHBlock; ++k) {
            const uint32_t i = base + k * kHBlock + tid;
            if (i >= N) continue;
            const uint32_t j = i - lds_lo;
            const uint32_t me = s_meta[j], rs = s_rs[j];
            const bool valid = me & M_VALID;
            const uint32_t my_ident = me & M_IDENT, my_mate = (me >> 28) & 3u;
            bool head = valid, first = valid, greater_before = false;
            if (valid && !(me & M_RUN)) {
                if (rs) {
                    const uint64_t run = static_cast<uint64_t>(rs - 1) << 30;
                    first = hash_get_min(s_tab, run | my_ident) == j;
                    head = hash_get_min(s_tab, run | (me & 0x30000000u) | kRefAny) == j;
                    if (any_paired)
                        for (uint32_t m2 = my_mate + 1; m2 < 3; ++m2)
                            greater_before = greater_before || hash_get_min(s_tab, run | (m2 << 28) | kRefAny) < j;
                } else {  // the run starts before the staged window: walk back through global memory (rare)
                    uint32_t q = i, steps = 0;
                    while (q > 0) {
                        --q;
                        bool dummy = false;
                        const uint32_t m = (q >= lds_lo) ? s_meta[q - lds_lo] : full_meta(acc, q, dummy);
                        if (m & M_VALID) {
                            const uint32_t mt = (m >> 28) & 3u;
                            if ((m & M_IDENT) == my_ident) {
                                head = false;
                                first = false;
                                break;
                            }
                            head = head && (mt != my_mate);
                            greater_before = greater_before || (mt > my_mate);
                        }
                        if (m & M_RUN) break;
                        if (++steps > kLookBackMax + kHHalo) {
                            too_long = true;
                            break;
                        }
                    }
                }
            }
            const uint32_t f = (my_mate << FL_MATE_SHIFT) | ((me & M_RUN) ? FL_RUN_START : 0u) | (head ? FL_HEAD : 0u) |
                               (first ? FL_FIRST : 0u) | ((valid && greater_before) ? FL_GREATER_BEFORE : 0u);
            nh += head;
            nf += first;
            nv += valid;
            any_gb = any_gb || (valid && greater_before);
            fl[i] = static_cast<uint8_t>(f);
        }
    }
    nh = r_wave_sum(nh);
    nf = r_wave_sum(nf);
    nv = r_wave_sum(nv);
    const uint32_t err = (__any(bad) ? ERR_REF_RANGE : 0u) | (__any(too_long) ? ERR_RUN_LENGTH : 0u);
    const bool wave_gb = __any(any_gb);
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
        for (int w = 0; w < kHBlock / 64; ++w) {
            t.x += s_w[w].x;
            t.y += s_w[w].y;
            v += s_v[w];
        }
        tile_cnt[tile] = t;
        if (Acc::kCountsMapped) tile_valid[tile] = v;
    }
    __syncthreads();  // LDS is reused by the next tile
    }
}

// ---------------------------------------------------------------------------------------------------------
// k_pick_runs: which classification kernel suits this stream?  Samples the first records, measures records per qName
// run, and writes the choice to counters[CNT_MODE]; both kernels are launched and the one not chosen returns at once.
// Measured per 10 M records: look-back 100 us at 3 records/run, 152 at 8, 417 at 42; hash table 318 / - / 222.
// ---------------------------------------------------------------------------------------------------------
constexpr uint32_t kPickSample = 8192;
constexpr uint32_t kPickHashAbove = 24;  // records per run

template <typename Acc>
__global__ __launch_bounds__(1024) void k_pick_runs(const Acc acc, uint32_t* __restrict__ counters, int force) {
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
        uint32_t mode = (t != 0 && S / t > kPickHashAbove) ? 1u : 0u;
        if (force >= 0) mode = static_cast<uint32_t>(force);
        counters[CNT_MODE] = mode;
    }
}

template <typename Acc>
__global__ __launch_bounds__(kRBlock) void k_emit(const Acc acc, const uint8_t* __restrict__ fl,
                                                  uint32_t* __restrict__ counters, const uint2* __restrict__ tile_off,
                                                  uint32_t* __restrict__ tgt_ref, uint32_t* __restrict__ tgt_gbin,
                                                  uint32_t* __restrict__ read_off) {
    __shared__ uint8_t s_fl[kHalo + kRTile + kHalo];
    __shared__ uint2 s_w[2][kRWaves];
    const uint32_t N = acc.count(counters);
    const uint32_t base = blockIdx.x * kRTile;
    if (base >= N) return;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t lds_lo = base >= kHalo ? base - kHalo : 0u;
    const uint32_t lds_hi = min(base + static_cast<uint32_t>(kRTile) + kHalo, N);
    for (uint32_t i = lds_lo + threadIdx.x; i < lds_hi; i += kRBlock) s_fl[i - lds_lo] = fl[i];
    __syncthreads();
    uint2 running = tile_off[blockIdx.x];
    bool too_long = false;
    // A later record with a smaller mate exists only where some record carries the larger-mate-before flag; when no
    // record of the whole stream does (unpaired data, or mates not interleaved), the forward look is skipped.
    const bool mates_interleave = counters[CNT_ANYGB] != 0u;
#pragma unroll
    for (int k = 0; k < kRItems; ++k) {
        const uint32_t i = base + k * kRBlock + threadIdx.x;
        const uint32_t f = (i < N) ? s_fl[i - lds_lo] : 0u;
        const bool head = f & FL_HEAD, first = f & FL_FIRST;
        const uint64_t mh = __ballot(head), mf = __ballot(first);
        const uint32_t rh = r_mask_rank(mh), rf = r_mask_rank(mf);
        if ((threadIdx.x & 63) == 0) s_w[k & 1][wave] = make_uint2(__popcll(mh), __popcll(mf));
        __syncthreads();
        uint2 before = make_uint2(0u, 0u), total = make_uint2(0u, 0u);
#pragma unroll
        for (int w = 0; w < kRWaves; ++w) {
            const uint2 c = s_w[k & 1][w];
            if (w < static_cast<int>(wave)) {
                before.x += c.x;
                before.y += c.y;
            }
            total.x += c.x;
            total.y += c.y;
        }
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
            for (uint32_t d = 1; __ballot(act != 0u) != 0ull; ++d) {
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
            for (uint32_t d = 1; __ballot(act != 0u) != 0ull; ++d) {
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
        if (first) {
            bool dummy = false;
            const uint32_t r = acc.meta_of(i, dummy) & 0x0fffffffu;
            tgt_ref[t] = r | (head ? 0x80000000u : 0u);
            tgt_gbin[t] = acc.gbin_of(i, r);
            if (head) read_off[m] = t;
        }
        running.x += total.x;
        running.y += total.y;
    }
    const bool any_long = __any(too_long);
    if (any_long && (threadIdx.x & 63) == 0) atomicOr(&counters[CNT_ERR], ERR_RUN_LENGTH);
}

static inline uint32_t rtiles(uint32_t n) { return (n + kRTile - 1) / kRTile; }

// SLIMM_RUNS_KERNEL=walk|hash overrides the choice k_pick_runs makes on the device (for tests and A/B timing)
static int forced_runs_mode() {
    const char* e = getenv("SLIMM_RUNS_KERNEL");
    if (!e) return -1;
    return e[0] == 'h' ? 1 : (e[0] == 'w' ? 0 : -1);
}

static RawRecords make_raw(const DeviceRecords& in, uint32_t n_refs, const uint32_t* ref_len, const uint32_t* bin_off,
                           uint32_t half_read, uint32_t bin_width) {
    RawRecords a;
    a.key = in.key;
    a.ref = in.ref;
    a.pos = in.pos;
    a.flag = in.flag;
    a.n = in.n;
    a.n_refs = n_refs;
    a.ref_len = ref_len;
    a.bin_off = bin_off;
    a.half_read = half_read;
    a.bin_width = bin_width;
    return a;
}

void launch_runs_raw(hipStream_t st, const DeviceRecords& in, uint32_t n_refs, const uint32_t* ref_len,
                     const uint32_t* bin_off, uint32_t half_read, uint32_t bin_width, uint32_t* counters, uint8_t* fl,
                     uint2* tile_cnt, uint32_t* tile_valid, int part) {
    const uint32_t nt = rtiles(in.n);
    if (!nt) return;
    const RawRecords a = make_raw(in, n_refs, ref_len, bin_off, half_read, bin_width);
    if (part == 0)
        hipLaunchKernelGGL(k_pick_runs<RawRecords>, dim3(1), dim3(1024), 0, st, a, counters, forced_runs_mode());
    else if (part == 1)
        hipLaunchKernelGGL(k_runs<RawRecords>, dim3(std::min(nt, 16384u)), dim3(kRBlock), 0, st, a, nt, counters, fl,
                           tile_cnt, tile_valid);
    else
        hipLaunchKernelGGL(k_runs_hash<RawRecords>, dim3(std::min(nt, 512u)), dim3(kHBlock), 0, st, a, nt, counters, fl,
                           tile_cnt, tile_valid);
}

void launch_emit_raw(hipStream_t st, const DeviceRecords& in, uint32_t n_refs, const uint32_t* ref_len,
                     const uint32_t* bin_off, uint32_t half_read, uint32_t bin_width, const uint8_t* fl, uint32_t* counters,
                     const uint2* tile_off, uint32_t* tgt_ref, uint32_t* tgt_gbin, uint32_t* read_off) {
    const uint32_t nt = rtiles(in.n);
    if (!nt) return;
    hipLaunchKernelGGL(k_emit<RawRecords>, dim3(nt), dim3(kRBlock), 0, st,
                       make_raw(in, n_refs, ref_len, bin_off, half_read, bin_width), fl, counters, tile_off, tgt_ref,
                       tgt_gbin, read_off);
}

void launch_runs_sorted(hipStream_t st, uint32_t n_upper, const uint64_t* ident, const uint32_t* cref, const uint32_t* cgbin,
                        uint32_t* counters, uint8_t* fl, uint2* tile_cnt, int part) {
    const uint32_t nt = rtiles(n_upper);
    if (!nt) return;
    SortedRecords a{ident, cref, cgbin};
    if (part == 0)
        hipLaunchKernelGGL(k_pick_runs<SortedRecords>, dim3(1), dim3(1024), 0, st, a, counters, forced_runs_mode());
    else if (part == 1)
        hipLaunchKernelGGL(k_runs<SortedRecords>, dim3(std::min(nt, 16384u)), dim3(kRBlock), 0, st, a, nt, counters, fl,
                           tile_cnt, static_cast<uint32_t*>(nullptr));
    else
        hipLaunchKernelGGL(k_runs_hash<SortedRecords>, dim3(std::min(nt, 512u)), dim3(kHBlock), 0, st, a, nt, counters, fl,
                           tile_cnt, static_cast<uint32_t*>(nullptr));
}

void launch_emit_sorted(hipStream_t st, uint32_t n_upper, const uint64_t* ident, const uint32_t* cref, const uint32_t* cgbin,
                        const uint8_t* fl, uint32_t* counters, const uint2* tile_off, uint32_t* tgt_ref, uint32_t* tgt_gbin,
                        uint32_t* read_off) {
    const uint32_t nt = rtiles(n_upper);
    if (!nt) return;
    SortedRecords a{ident, cref, cgbin};
    hipLaunchKernelGGL(k_emit<SortedRecords>, dim3(nt), dim3(kRBlock), 0, st, a, fl, counters, tile_off, tgt_ref, tgt_gbin,
                       read_off);
}

}  // namespace slimm
