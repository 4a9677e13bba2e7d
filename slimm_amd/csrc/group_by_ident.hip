// Grouping of the record stream by read identity for record_order = SLIMM_ORDER_ANY (gfx950, wave64).
//
// The reference groups the records of a read through a string-keyed hash map (src/slimm.hpp:204-211) and therefore takes
// them in any order; the first-bin rule (src/read_stat.hpp:116-135, Q1) needs the FILE ORDER of a read's records.  What
// the single-pass front end (front.hip) needs of the stream is exactly that and nothing more: the records of one
// identity adjacent, in file order -- not a total order.  Round 1 - 3 made one anyway (an 8-pass LSD radix sort of the
// 64-bit identity behind a separate compaction: 10 passes over the records).  Here:
//
//   1. PARTITION by b bits of a HASH of the qName key, b = log2(records / 4) or so: P = ceil(b / W) stable counting
//      passes of W bits each, least significant digit first.  The first pass reads the caller's records themselves --
//      the record filter (src/slimm.hpp:197), the read identity (qName key << 2 | mate, :204-208) and the bin (:200-201)
//      happen on the way: no compaction passes.  After the last pass the stream is ordered by the b hash bits, file
//      order kept inside equal bits: a BUCKET of a few records, almost always of one identity.
//   2. FINISH: a wave walks the buckets that start in its stretch of the stream in windows of 64 records.  A bucket
//      whose identities are in non-decreasing order already (nearly all) is left alone -- 8 bytes read per record,
//      nothing written.  The others are put in order by identity, file order kept among equal identities: buckets of
//      fewer than 64 records by a rank computed with lane shifts (as many steps as the longest such bucket), in place;
//      longer ones (a read of thousands of records that shares its hash bits with another) by selection, one distinct
//      identity per sweep, through the scratch arrays.  Any bucket size, any number of identities per bucket.
//
// A counting pass (k_gb_count -> k_gb_scan -> k_gb_scatter) is STABLE without a histogram per tile of the stream:
// G persistent workgroups own G contiguous stretches; the count matrix is [digit][workgroup], its prefix in that order
// gives every workgroup the start of its own piece of every digit's run, and the workgroup advances those G x 2^W
// cursors in LDS round by round (rounds of 4096 records; inside a round: per-wave match masks from W ballots, the
// waves' counts prefixed in LDS -- a wave owns 512 consecutive records of the round, so (wave, chunk, lane) IS file
// order).  Records go out straight from the registers; a digit's records of one round are consecutive at the digit's
// cursor, and the L2 of the workgroup's XCD merges them into whole lines.  Per pass: 8 bytes read by the count (the
// identities only: the payload is a separate array), 16 read + 16 written by the scatter.
//
// Layout between the passes and into k_front<FrontSorted>:  ident u64 (key << 2 | mate) | pay uint2 {reference, global
// bin} (| chk u32, the optional check word of slimm_push_records_checked).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>

#include "force.h"
#include "kernels.h"

namespace slimm {

namespace {

#if defined(EXP) && EXP == 10  // cycle split of k_gb_scatter: wave 0's lane 0 of every workgroup (scripts/tprof_group.py)
__device__ unsigned long long g_prof_g[8 * 1024];
#define GPROF_T(x) const unsigned long long x = __builtin_readcyclecounter()
#define GPROF_ADD(slot, a, b) if (threadIdx.x == 0) g_prof_g[(blockIdx.x & 1023u) * 8 + slot] += (b) - (a)
#else
#define GPROF_T(x)
#define GPROF_ADD(slot, a, b)
#endif
constexpr int kGBlock = 512;
constexpr int kGWaves = kGBlock / 64;
constexpr int kGItems = 8;
constexpr uint32_t kGRound = kGBlock * kGItems;   // records per workgroup and round
constexpr uint32_t kGWaveRecs = 64 * kGItems;     // consecutive records of a round one wave owns
constexpr uint32_t kGMaxDigits = 1u << kGroupMaxBits;

__device__ __forceinline__ uint32_t g_lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ uint32_t g_rank(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
}
__device__ __forceinline__ uint64_t g_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// The hash the digits are cut from: 32 bits of the qName key (identity >> 2: both mates of a name share a bucket, as they
// share a run in the front end).  Two 32-bit multiplicative hashes added up; the TOP bits depend on every bit of the key
// (consecutive integers, keys that differ in their top bits only: tests/test_gpu_group_by_ident.py).  32-bit multiplies
// on purpose: a 64-bit product is four of them at a quarter of the vector rate, per record, pass and kernel.
__device__ __forceinline__ uint32_t gb_mix(uint64_t ident) {
    const uint64_t key = ident >> 2;
    return static_cast<uint32_t>(key) * 0x9E3779B1u + static_cast<uint32_t>(key >> 32) * 0x85EBCA77u;
}

// A load of the stream the scatter reads once: a plain load.  (Marked non-temporal -- so that the lines it brings in do not
// push the half-written lines at the G x 2^W output frontiers out of the L2 -- it measured even at 10 M records and 11 %
// SLOWER at 100 M: scatter 886 -> 987 us, count 225 -> 267.)
template <typename T>
__device__ __forceinline__ T gb_stream_load(const T* p) {
    return *p;
}

// ---- record sources -------------------------------------------------------------------------------------------------
// the caller's records (first pass).  kPacked: 16-byte records, the flag bits in the key's top three bits.
template <bool kPacked>
struct GbRaw {
    static constexpr bool kRaw = true;
    const uint64_t* key;
    const int32_t* ref;
    const int32_t* pos;
    const uint16_t* flag;
    const uint32_t* check;
    const uint2* geo;  // {contig length, first bin} per reference
    uint32_t n, n_refs, half_read, bin_width, bw_magic;
    __device__ uint32_t count(const uint32_t*) const { return n; }
    // identity and "is a mapped record of a known reference" from the words the count reads (src/slimm.hpp:197, :204-208)
    __device__ bool ident_of(uint64_t k, uint32_t r, uint32_t f, uint64_t& ident, bool& bad) const {
        bool mapped;
        uint32_t mate;
        if (kPacked) {
            mapped = static_cast<int64_t>(k) >= 0 && r != 0xffffffffu;
            mate = static_cast<uint32_t>(k >> 61) & 3u;
        } else {
            mapped = !(f & 0x4u) && r != 0xffffffffu;
            mate = (f & 0x40u) ? 1u : ((f & 0x80u) ? 2u : 0u);
        }
        if (mapped && r >= n_refs) {
            bad = true;
            mapped = false;
        }
        ident = (k << (kPacked ? 3 : 2) >> (kPacked ? 1 : 0)) | mate;  // packed: 61 identity bits, the four arrays: 62
        return mapped;
    }
    __device__ uint32_t div_bin_width(uint32_t v) const {
        const uint32_t q = __umulhi(v, bw_magic);
        const uint32_t r = v - q * bin_width;
        return q + (r >= bin_width ? 1u : 0u);
    }
};
// the stream between the passes
struct GbIdent {
    static constexpr bool kRaw = false;
    const uint64_t* ident;
    const uint2* pay;
    const uint32_t* chk;
    __device__ uint32_t count(const uint32_t* counters) const { return counters[CNT_V]; }
};

// the workgroup's stretch of the stream
__device__ __forceinline__ void gb_stretch(uint32_t n, uint32_t& lo, uint32_t& hi) {
    const uint32_t per = (n + gridDim.x - 1u) / gridDim.x;
    lo = min(n, blockIdx.x * per);
    hi = min(n, lo + per);
}

// ---------------------------------------------------------------------------------------------------------
// k_gb_count: matrix[digit * G + workgroup] = records of the workgroup's stretch with that digit
// ---------------------------------------------------------------------------------------------------------
template <typename Src>
__global__ __launch_bounds__(kGBlock) void k_gb_count(const Src src, uint32_t* __restrict__ counters, uint32_t shift,
                                                      uint32_t bits, uint32_t* __restrict__ matrix) {
    __shared__ uint32_t s_h[kGMaxDigits];
    const uint32_t D = 1u << bits, tid = threadIdx.x;
    for (uint32_t d = tid; d < D; d += kGBlock) s_h[d] = 0u;
    __syncthreads();
    uint32_t lo, hi;
    gb_stretch(src.count(counters), lo, hi);
    bool bad = false;
    for (uint32_t r0 = lo; r0 < hi; r0 += kGRound) {
        uint64_t k[kGItems];
        uint32_t r[kGItems], f[kGItems];
#pragma unroll
        for (int u = 0; u < kGItems; ++u) {
            const uint32_t i = min(r0 + u * kGBlock + tid, hi - 1u);  // (clamped: every load of the round in flight at once)
            if constexpr (Src::kRaw) {
                k[u] = src.key[i];
                r[u] = static_cast<uint32_t>(src.ref[i]);
                f[u] = src.flag ? src.flag[i] : 0u;
            } else {
                k[u] = src.ident[i];
                r[u] = f[u] = 0u;
            }
        }
#pragma unroll
        for (int u = 0; u < kGItems; ++u) {
            bool live = r0 + u * kGBlock + tid < hi;
            uint64_t ident = k[u];
            if constexpr (Src::kRaw) live = src.ident_of(k[u], r[u], f[u], ident, bad) && live;
            if (live) atomicAdd(&s_h[(gb_mix(ident) >> shift) & (D - 1u)], 1u);
        }
    }
    __syncthreads();
    for (uint32_t d = tid; d < D; d += kGBlock) matrix[static_cast<size_t>(d) * gridDim.x + blockIdx.x] = s_h[d];
    if (Src::kRaw && __any(bad) && (tid & 63u) == 0u) atomicOr(&counters[CNT_ERR], static_cast<uint32_t>(ERR_REF_RANGE));
}

// ---------------------------------------------------------------------------------------------------------
// k_gb_scan: one workgroup per digit: exclusive prefix of its row of the matrix over the workgroups, the row's total
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kGroupMaxGrid) void k_gb_scan(uint32_t* __restrict__ matrix, uint32_t G, uint32_t* __restrict__ totals) {
    __shared__ uint32_t s_w[kGroupMaxGrid / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t* row = matrix + static_cast<size_t>(blockIdx.x) * G;
    const uint32_t v = tid < G ? row[tid] : 0u;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t a = __shfl_up(inc, o, 64);
        if (lane >= static_cast<uint32_t>(o)) inc += a;
    }
    if (lane == 63u) s_w[wave] = inc;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < kGroupMaxGrid / 64; ++w) {
        const uint32_t t = s_w[w];
        before += w < wave ? t : 0u;
        total += t;
    }
    if (tid < G) row[tid] = before + inc - v;
    if (tid == 0u) totals[blockIdx.x] = total;
}

// ---------------------------------------------------------------------------------------------------------
// k_gb_scatter: the stable scatter of one pass (header)
// ---------------------------------------------------------------------------------------------------------
// dynamic LDS of k_gb_scatter, in 32-bit words: cursors + per-wave counts (16-bit, a spare per wave), rounded to 8 bytes
__host__ __device__ constexpr uint32_t gb_lds_tables(uint32_t D) { return (D + (kGWaves * (D + 1u) + 1u) / 2u + 1u) & ~1u; }
constexpr uint32_t kGMatchBits = 10;  // digits up to this width find their peers through a table of lane masks in LDS
                                       // (10 bits, staged: 156 KB with the tables -- the stage alone makes it one workgroup per CU)
constexpr uint32_t kGLdsWords = (160u * 1024u - 256u) / 4u;  // what a workgroup may ask for (a CU's LDS less the static arrays)
// the lane-mask tables: up to kGMatchBits, where they fit beside the rest
__host__ __device__ constexpr bool gb_match(uint32_t D, bool staged) {
    return D <= (1u << kGMatchBits) && gb_lds_tables(D) + (staged ? 2u * D + 4u * kGRound : 0u) + 2u * kGWaves * D <= kGLdsWords;
}
__host__ __device__ constexpr uint32_t gb_lds_words(uint32_t D, bool staged) {
    return gb_lds_tables(D) + (staged ? 2u * D + 4u * kGRound : 0u) + (gb_match(D, staged) ? 2u * kGWaves * D : 0u);
}

// kStaged: the round's records go out ORDERED BY DIGIT through LDS -- a digit's records of the round are one run of
// consecutive lanes and consecutive addresses (16 records = 128 bytes of identities at 8-bit digits) instead of 64 lanes
// storing 8 bytes each to some fifty places: measured at 10 M records, the same kernel with its stores made consecutive
// runs 68 instead of 98 us.  Costs 64 KB of LDS per workgroup (the digits wider than 9 bits, whose tables need the room,
// and streams with check words keep the direct stores).
template <typename Src, bool kChk, bool kStaged>
__global__ __launch_bounds__(kGBlock) void k_gb_scatter(const Src src, uint32_t* __restrict__ counters, uint32_t shift,
                                                        uint32_t bits, const uint32_t* __restrict__ matrix,
                                                        const uint32_t* __restrict__ totals, uint64_t* __restrict__ ident_out,
                                                        uint2* __restrict__ pay_out, uint32_t* __restrict__ chk_out) {
    HIP_DYNAMIC_SHARED(uint32_t, s_dyn)
    // s_cursor[D] (where the workgroup's next record of each digit goes) | s_wcnt[kGWaves][D] (16-bit: per wave and digit,
    // first the records of the round seen so far, then the wave's offset inside the digit's records of the round)
    __shared__ uint32_t s_ws[kGWaves];
    const uint32_t D = 1u << bits, tid = threadIdx.x, lane = g_lane();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint32_t* const s_cursor = s_dyn;
    uint16_t* const s_wcnt = reinterpret_cast<uint16_t*>(s_dyn + D);
    uint16_t* const my_cnt = s_wcnt + wave * D;
    uint16_t* const spare = s_wcnt + kGWaves * D + wave;  // a word nobody reads
    // (kStaged) behind the tables: s_dstart[D] = where a digit's records start in the ordered round, s_gdelta[D] = global
    // place minus place in the round, then the round itself: identities, payloads
    // (bits <= kGMatchBits) behind those: s_match[kGWaves][D] 64-bit lane masks
    const bool match = gb_match(D, kStaged);
    unsigned long long* const my_match = reinterpret_cast<unsigned long long*>(s_dyn + gb_lds_tables(D) + (kStaged ? 2u * D + 4u * kGRound : 0u)) + wave * D;
    uint32_t* const s_dstart = s_dyn + gb_lds_tables(D);
    uint32_t* const s_gdelta = s_dstart + D;
    uint64_t* const s_rid = reinterpret_cast<uint64_t*>(s_gdelta + D);
    uint2* const s_rpay = reinterpret_cast<uint2*>(s_rid + kGRound);
    // ---- digit bases: exclusive prefix of the digit totals (every workgroup computes it), plus this workgroup's offset
    {
        const uint32_t per = (D + kGBlock - 1u) / kGBlock;  // digits per thread: consecutive ones
        uint32_t t[kGMaxDigits / kGBlock > 0 ? kGMaxDigits / kGBlock : 1];
        uint32_t mine = 0;
#pragma unroll
        for (uint32_t k = 0; k < kGMaxDigits / kGBlock; ++k) {
            const uint32_t d = tid * per + k;
            t[k] = (k < per && d < D) ? totals[d] : 0u;
            mine += t[k];
        }
        uint32_t inc = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t a = __shfl_up(inc, o, 64);
            if (lane >= static_cast<uint32_t>(o)) inc += a;
        }
        if (lane == 63u) s_ws[wave] = inc;
        __syncthreads();
        uint32_t run = inc - mine, all = 0;
#pragma unroll
        for (uint32_t w = 0; w < kGWaves; ++w) {
            const uint32_t x = s_ws[w];
            run += w < wave ? x : 0u;
            all += x;
        }
#pragma unroll
        for (uint32_t k = 0; k < kGMaxDigits / kGBlock; ++k) {
            const uint32_t d = tid * per + k;
            if (k < per && d < D) s_cursor[d] = run + matrix[static_cast<size_t>(d) * gridDim.x + blockIdx.x];
            run += t[k];
        }
        // hits_count (src/slimm.hpp:212): the first pass has just counted the mapped records
        if (Src::kRaw && blockIdx.x == 0u && tid == 0u) counters[CNT_V] = all;
    }
    for (uint32_t d = tid; d < kGWaves * D; d += kGBlock) s_wcnt[d] = 0;
    if (match)
        for (uint32_t d = lane; d < D; d += 64u) my_match[d] = 0ull;
    __syncthreads();
    uint32_t lo, hi;
    gb_stretch(src.count(counters), lo, hi);
    // ---- a round's records: the wave's 512 consecutive ones in chunks of 64, all loads in flight together.  (Asking for
    // the NEXT round's behind the placement, so that they arrive while the round is written out: measured, no change --
    // the kernel waits for the LDS pipe, scripts/tprof_group.py: peers + ranks 37 %, placement 30 %, write-out 20 %.)
    uint64_t ident[kGItems];
    uint2 pay[kGItems];
    uint32_t chk[kGItems], aux[kGItems];
    auto load_round = [&](uint32_t r0) {
        const uint32_t w0 = r0 + wave * kGWaveRecs;
#pragma unroll
        for (int u = 0; u < kGItems; ++u) {
            const uint32_t i = min(w0 + u * 64u + lane, hi - 1u);
            if constexpr (Src::kRaw) {
                ident[u] = gb_stream_load(src.key + i);
                pay[u].x = static_cast<uint32_t>(gb_stream_load(src.ref + i));
                pay[u].y = static_cast<uint32_t>(gb_stream_load(src.pos + i));
                aux[u] = src.flag ? src.flag[i] : 0u;
                chk[u] = kChk ? src.check[i] : 0u;
            } else {
                ident[u] = gb_stream_load(src.ident + i);
                const uint64_t q = gb_stream_load(reinterpret_cast<const uint64_t*>(src.pay + i));
                pay[u] = make_uint2(static_cast<uint32_t>(q), static_cast<uint32_t>(q >> 32));
                aux[u] = 0u;
                chk[u] = kChk ? src.chk[i] : 0u;
            }
        }
    };
    for (uint32_t r0 = lo; r0 < hi; r0 += kGRound) {
        GPROF_T(g0);
        load_round(r0);
        bool live[kGItems];
        const uint32_t w0 = r0 + wave * kGWaveRecs;
        if constexpr (Src::kRaw) {
            // filter, identity, and the bin: one 8-byte gather of the contig's geometry per record (rows of unmapped
            // records gather row 0), uint32 wrap-around and clamp as src/slimm.hpp:200-201 (Q3)
            bool bad = false;
            uint2 geo[kGItems];
#pragma unroll
            for (int u = 0; u < kGItems; ++u) {
                uint64_t id;
                live[u] = src.ident_of(ident[u], pay[u].x, aux[u], id, bad) && (w0 + u * 64u + lane < hi);
                ident[u] = id;
                geo[u] = src.geo[live[u] ? pay[u].x : 0u];
            }
#pragma unroll
            for (int u = 0; u < kGItems; ++u)
                pay[u].y = geo[u].y + src.div_bin_width(min(pay[u].y + src.half_read, geo[u].x));
            (void)bad;  // (the count has flagged it)
        } else {
#pragma unroll
            for (int u = 0; u < kGItems; ++u) live[u] = w0 + u * 64u + lane < hi;
        }
        // ---- place inside the wave's records of the round, per digit, in file order
        GPROF_T(g1);
        GPROF_ADD(0, g0, g1);
        uint32_t dig[kGItems], place[kGItems];
#pragma unroll
        for (int u = 0; u < kGItems; ++u) {
            const uint32_t d = (gb_mix(ident[u]) >> shift) & (D - 1u);
            // the lanes of this chunk with my digit: up to 10-bit digits through the wave's own table of lane masks in LDS
            // (everybody ORs its bit into its digit's entry, reads the entry back and clears it again: three LDS
            // operations whatever the digits), wider ones by a ballot per bit (6 vector + 2 scalar instructions each)
            uint64_t peers;
            if (match) {
                unsigned long long* const e = my_match + d;
                atomicOr(e, live[u] ? 1ull << lane : 0ull);  // (every lane issues it: no branch next to the wave barriers)
                __builtin_amdgcn_wave_barrier();
                peers = *e;
                __builtin_amdgcn_wave_barrier();
                *e = 0ull;
            } else {
                peers = g_ballot(live[u]);
                for (uint32_t b = 0; b < bits; ++b) {
                    const bool bit = (d >> b) & 1u;
                    const uint64_t bm = g_ballot(bit);
                    peers &= bit ? bm : ~bm;
                }
            }
            const uint32_t rank = g_rank(peers);
            // (no branch on per-lane state next to the wave barriers: every lane loads, every lane stores -- the first
            // lane of a digit the new count, the others into the wave's spare word.  Cheaper than exec-mask bookkeeping on
            // the GPU, and the host emulator of tests/native tells collectives apart by call site, which a compiler that
            // clones the code behind an `if` would double.)
            const uint32_t before = my_cnt[d];
            __builtin_amdgcn_wave_barrier();  // (every lane has read the count before the first lane of a digit raises it)
            const bool lead = live[u] & (rank == 0u);
            uint16_t* const to = lead ? my_cnt + d : spare;
            *to = static_cast<uint16_t>(before + static_cast<uint32_t>(__popcll(peers)));
            __builtin_amdgcn_wave_barrier();
            dig[u] = d;
            place[u] = before + rank;
        }
        GPROF_T(g2);
        GPROF_ADD(1, g1, g2);
        __syncthreads();
        GPROF_T(g3);
        GPROF_ADD(2, g2, g3);
        if constexpr (!kStaged) {
            // ---- the waves' counts of every digit -> each wave's offset inside the digit's records of the round
            uint32_t tot[kGMaxDigits / kGBlock > 0 ? kGMaxDigits / kGBlock : 1];
#pragma unroll
            for (uint32_t k = 0; k < kGMaxDigits / kGBlock; ++k) {
                const uint32_t d = tid + k * kGBlock;
                uint32_t run = 0;
                if (d < D) {
#pragma unroll
                    for (uint32_t w = 0; w < kGWaves; ++w) {
                        const uint32_t c = s_wcnt[w * D + d];
                        s_wcnt[w * D + d] = static_cast<uint16_t>(run);
                        run += c;
                    }
                }
                tot[k] = run;
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < kGItems; ++u) {
                if (live[u]) {
                    const uint32_t dst = s_cursor[dig[u]] + my_cnt[dig[u]] + place[u];
                    ident_out[dst] = ident[u];
                    pay_out[dst] = pay[u];
                    if (kChk) chk_out[dst] = chk[u];
                }
            }
            __syncthreads();
#pragma unroll
            for (uint32_t k = 0; k < kGMaxDigits / kGBlock; ++k) {
                const uint32_t d = tid + k * kGBlock;
                if (d < D) {
                    s_cursor[d] += tot[k];
#pragma unroll
                    for (uint32_t w = 0; w < kGWaves; ++w) s_wcnt[w * D + d] = 0;
                }
            }
            __syncthreads();
        } else {
            // ---- the waves' counts of every digit -> each wave's offset inside the digit's records of the round, the
            // digit's start in the ordered round (a prefix over the digits: consecutive digits per thread), and what
            // turns a place in the round into a place in the stream
            constexpr uint32_t kPer = kGMaxDigits / kGBlock > 0 ? kGMaxDigits / kGBlock : 1;
            const uint32_t per = D > static_cast<uint32_t>(kGBlock) ? D / kGBlock : 1u;
            uint32_t tot[kPer];
            uint32_t mine = 0;
#pragma unroll
            for (uint32_t k = 0; k < kPer; ++k) {
                const uint32_t d = tid * per + k;
                uint32_t run = 0;
                if (k < per && d < D) {
#pragma unroll
                    for (uint32_t w = 0; w < kGWaves; ++w) {
                        const uint32_t c = s_wcnt[w * D + d];
                        s_wcnt[w * D + d] = static_cast<uint16_t>(run);
                        run += c;
                    }
                }
                tot[k] = run;
                mine += run;
            }
            uint32_t inc = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t x = __shfl_up(inc, o, 64);
                if (lane >= static_cast<uint32_t>(o)) inc += x;
            }
            if (lane == 63u) s_ws[wave] = inc;
            __syncthreads();
            uint32_t run = inc - mine, n_round = 0;
#pragma unroll
            for (uint32_t w = 0; w < kGWaves; ++w) {
                const uint32_t x = s_ws[w];
                run += w < wave ? x : 0u;
                n_round += x;
            }
#pragma unroll
            for (uint32_t k = 0; k < kPer; ++k) {
                const uint32_t d = tid * per + k;
                if (k < per && d < D) {
                    s_dstart[d] = run;
                    s_gdelta[d] = s_cursor[d] - run;
                    s_cursor[d] += tot[k];
                }
                run += tot[k];
            }
            __syncthreads();
            GPROF_T(g4);
            GPROF_ADD(3, g3, g4);
            // ---- the records into the ordered round
#pragma unroll
            for (int u = 0; u < kGItems; ++u) {
                if (live[u]) {
                    const uint32_t at = s_dstart[dig[u]] + my_cnt[dig[u]] + place[u];
                    s_rid[at] = ident[u];
                    s_rpay[at] = pay[u];
                }
            }
            __syncthreads();
            GPROF_T(g5);
            GPROF_ADD(4, g4, g5);
            // ---- and out: consecutive lanes take consecutive places (the digit again from the identity)
#pragma unroll
            for (int u = 0; u < kGItems; ++u) {
                const uint32_t at = u * kGBlock + tid;
                if (at < n_round) {
                    const uint64_t id = s_rid[at];
                    const uint32_t dst = at + s_gdelta[(gb_mix(id) >> shift) & (D - 1u)];
                    ident_out[dst] = id;
                    pay_out[dst] = s_rpay[at];
                }
            }
            GPROF_T(g6);
            GPROF_ADD(5, g5, g6);
#pragma unroll
            for (uint32_t k = 0; k < kPer; ++k) {
                const uint32_t d = tid * per + k;
                if (k < per && d < D) {
#pragma unroll
                    for (uint32_t w = 0; w < kGWaves; ++w) s_wcnt[w * D + d] = 0;
                }
            }
            __syncthreads();
            GPROF_T(g7);
            GPROF_ADD(6, g6, g7);
            GPROF_ADD(7, g0, g7);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// k_gb_finish (header): identities in order inside every bucket of equal hash bits, file order among equal identities
// ---------------------------------------------------------------------------------------------------------
#ifndef SLIMM_FINISH_RECS
#define SLIMM_FINISH_RECS 512
#endif
constexpr uint32_t kFinishRecs = SLIMM_FINISH_RECS;  // records per wave: it handles the buckets that START among them
constexpr uint32_t kFinishStage = kFinishRecs + 128u;  // ... and keeps this many identities in LDS, from the record in front of them on

__device__ __forceinline__ uint32_t g_shr1(uint32_t v, uint32_t lane0) {  // the value of the lane before
    return __builtin_amdgcn_update_dpp(lane0, v, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t g_shl1(uint32_t v, uint32_t lane63) {  // the value of the lane behind
    return __builtin_amdgcn_update_dpp(lane63, v, 0x130, 0xf, 0xf, false);
}
__device__ __forceinline__ uint64_t g_u64(uint32_t lo, uint32_t hi) { return (static_cast<uint64_t>(hi) << 32) | lo; }
__device__ __forceinline__ uint64_t g_readlane64(uint64_t v, uint32_t l) {
    return g_u64(__builtin_amdgcn_readlane(static_cast<uint32_t>(v), l), __builtin_amdgcn_readlane(static_cast<uint32_t>(v >> 32), l));
}
__device__ __forceinline__ uint64_t g_wave_min64(uint64_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint64_t a = g_u64(__shfl_xor(static_cast<uint32_t>(v), o, 64), __shfl_xor(static_cast<uint32_t>(v >> 32), o, 64));
        v = a < v ? a : v;
    }
    return v;
}

// A bucket of 64 records or more (or one that the window code could not see the end of), starting at record p.
// Returns the index behind it.
template <bool kChk>
__device__ __forceinline__ uint32_t finish_long(uint64_t* __restrict__ ident, uint2* __restrict__ pay, uint32_t* __restrict__ chk,
                                                uint64_t* __restrict__ t_ident, uint2* __restrict__ t_pay,
                                                uint32_t* __restrict__ t_chk, uint32_t p, uint32_t V, uint32_t hshift,
                                                uint32_t lane) {
    const uint32_t h0 = gb_mix(ident[p]) >> hshift;
    // 1. where it ends; whether its identities are in order already
    uint32_t end = p;
    bool sorted = true;
    uint64_t last = 0;
    while (true) {
        const uint32_t i = end + lane;
        const bool in = i < V;
        const uint64_t id = ident[in ? i : V - 1u];
        const uint64_t same = g_ballot(in && (gb_mix(id) >> hshift) == h0);
        const uint32_t n_in = (~same) == 0ull ? 64u : static_cast<uint32_t>(__builtin_ctzll(~same));
        const uint64_t prev = g_u64(g_shr1(static_cast<uint32_t>(id), static_cast<uint32_t>(last)),
                                    g_shr1(static_cast<uint32_t>(id >> 32), static_cast<uint32_t>(last >> 32)));
        const uint64_t in_mask = n_in >= 64u ? ~0ull : ((1ull << n_in) - 1ull);
        sorted = sorted & ((g_ballot(id < prev) & in_mask) == 0ull);
        if (n_in) last = g_readlane64(id, n_in - 1u);
        end += n_in;
        if (n_in < 64u) break;
    }
    if (sorted) return end;
    // 2. selection: the smallest identity not placed yet, all its records in file order, and again
    uint32_t out = p;
    uint64_t bound = 0;
    while (out < end) {
        uint64_t m = ~0ull;
        for (uint32_t cb = p; cb < end; cb += 64u) {
            const uint32_t i = cb + lane;
            const uint64_t id = ident[i < end ? i : end - 1u];
            const uint64_t cand = (i < end && id >= bound) ? id : ~0ull;
            m = cand < m ? cand : m;
        }
        m = g_wave_min64(m);
        for (uint32_t cb = p; cb < end; cb += 64u) {
            const uint32_t i = cb + lane;
            const bool in = i < end;
            const uint64_t id = ident[in ? i : end - 1u];
            const bool hit = in && id == m;
            const uint64_t hm = g_ballot(hit);
            if (hit) {
                const uint32_t dst = out + g_rank(hm);
                t_ident[dst] = id;
                t_pay[dst] = pay[i];
                if (kChk) t_chk[dst] = chk[i];
            }
            out += static_cast<uint32_t>(__popcll(hm));
        }
        bound = m + 1ull;
    }
    // 3. back from the scratch arrays (this wave's own stores, read back past the vector cache)
    __threadfence();
    __builtin_amdgcn_wave_barrier();  // (lock step on the GPU; where the host emulator's lanes wait for each other's stores)
    for (uint32_t cb = p; cb < end; cb += 64u) {
        const uint32_t i = cb + lane;
        if (i < end) {
            ident[i] = __hip_atomic_load(&t_ident[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint64_t q = __hip_atomic_load(reinterpret_cast<const uint64_t*>(&t_pay[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pay[i] = make_uint2(static_cast<uint32_t>(q), static_cast<uint32_t>(q >> 32));
            if (kChk) chk[i] = __hip_atomic_load(&t_chk[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    return end;
}

template <bool kChk>
__global__ __launch_bounds__(64) void k_gb_finish(uint64_t* __restrict__ ident, uint2* __restrict__ pay, uint32_t* __restrict__ chk,
                                                  uint64_t* __restrict__ t_ident, uint2* __restrict__ t_pay,
                                                  uint32_t* __restrict__ t_chk, const uint32_t* __restrict__ counters,
                                                  uint32_t hshift) {
    // The wave's stretch, the record in front of it and a window's worth behind it, staged in LDS with every load in flight
    // at once: the windows below would otherwise be a chain of dependent loads, one round trip each (60 us at 10 M
    // records, when reading the identities takes 20).
    __shared__ uint64_t s_id[kFinishStage];
    __shared__ uint16_t s_mv[kFinishStage + 64];  // where a record of the stretch moves to (0xffff: it stays); + a spare word per lane
    const uint32_t V = counters[CNT_V];
    const uint32_t lane = g_lane();
    const uint32_t a = blockIdx.x * kFinishRecs;
    if (a >= V) return;
    const uint32_t e = min(V, a + kFinishRecs);
    const uint32_t base = a > 0u ? a - 1u : 0u;
    {
        uint64_t v[kFinishStage / 64];
#pragma unroll
        for (uint32_t k = 0; k < kFinishStage / 64u; ++k) v[k] = ident[min(base + 64u * k + lane, V - 1u)];
#pragma unroll
        for (uint32_t k = 0; k < kFinishStage / 64u; ++k) {
            s_id[64u * k + lane] = v[k];
            s_mv[64u * k + lane] = 0xffffu;
        }
        __builtin_amdgcn_wave_barrier();
    }
    // the lane's identity of the window of 64 records at q (records behind the stream's end: the last one again)
    auto window = [&](uint32_t q) -> uint64_t {
        if (q - base + 64u <= kFinishStage) return s_id[q - base + lane];
        return ident[min(q + lane, V - 1u)];
    };
    // the first bucket that starts at or behind a
    uint32_t p = a;
    if (a > 0u) {
        const uint32_t hprev = gb_mix(s_id[0]) >> hshift;
        while (true) {
            const uint32_t i = p + lane;
            const bool in = i < V;
            const uint64_t id = window(p);
            const uint64_t same = g_ballot(in && (gb_mix(id) >> hshift) == hprev);
            const uint32_t n_in = (~same) == 0ull ? 64u : static_cast<uint32_t>(__builtin_ctzll(~same));
            p += n_in;
            if (n_in < 64u) break;
        }
    }
    while (p < e) {  // p: a bucket starts here, p < V
        const uint32_t i = p + lane;
        const bool in = i < V;
        const uint64_t id = window(p);
        const uint32_t h = gb_mix(id) >> hshift;
        const bool start = in & (g_shr1(h, ~h) != h);  // (lane 0 starts a bucket: p is a bucket start)
        const uint64_t S = g_ballot(start);
        // whole buckets: up to the last bucket start of the window -- or to the end of the stream
        uint32_t X = p + 64u >= V ? V - p : 63u - static_cast<uint32_t>(__builtin_clzll(S));
        if (e - p < 64u) {  // buckets that start at or behind e are the next wave's
            const uint64_t beyond = S & ~((1ull << (e - p)) - 1ull);
            if (beyond) X = min(X, static_cast<uint32_t>(__builtin_ctzll(beyond)));
        }
        if (X == 0u) {
            p = finish_long<kChk>(ident, pay, chk, t_ident, t_pay, t_chk, p, V, hshift, lane);
            continue;
        }
        const uint64_t PR = X >= 64u ? ~0ull : ((1ull << X) - 1ull);
        const uint32_t idl = static_cast<uint32_t>(id), idh = static_cast<uint32_t>(id >> 32);
        const uint64_t prev = g_u64(g_shr1(idl, 0u), g_shr1(idh, 0u));
        const uint64_t dis = g_ballot(id < prev) & ~S & PR;  // a record in front of which a larger identity of its bucket lies
        if (dis) {
            // my bucket: lanes [bs, be)
            const uint64_t Sx = S & PR;
            const uint64_t le = lane >= 63u ? ~0ull : ((2ull << lane) - 1ull);
            const uint32_t bs = 63u - static_cast<uint32_t>(__builtin_clzll((Sx & le) | 1ull));
            const uint64_t above = Sx & ~le;
            const uint32_t be = above ? static_cast<uint32_t>(__builtin_ctzll(above)) : X;
            const bool mine = lane < X && ((dis >> bs) & (((be - bs) >= 64u) ? ~0ull : ((1ull << (be - bs)) - 1ull))) != 0ull;
            const uint32_t back = mine ? lane - bs : 0u, fwd = mine ? be - 1u - lane : 0u;
            // rank inside the bucket: records in front with an identity <= mine, records behind with one < mine
            uint32_t rank = 0;
            uint32_t bl = idl, bh = idh, fl = idl, fh = idh;
            for (uint32_t d = 1; g_ballot(back >= d || fwd >= d) != 0ull; ++d) {
                bl = g_shr1(bl, 0u);
                bh = g_shr1(bh, 0u);
                fl = g_shl1(fl, 0u);
                fh = g_shl1(fh, 0u);
                rank += (back >= d && g_u64(bl, bh) <= id) ? 1u : 0u;
                rank += (fwd >= d && g_u64(fl, fh) < id) ? 1u : 0u;
            }
            // (the move itself waits for the end of the stretch: every window's loads of the payload would be a round
            // trip of its own -- 39 of this kernel's 65 us at 10 M records)
            const uint32_t at = p - base + lane;
            s_mv[(mine && bs + rank != lane) ? at : kFinishStage + lane] = static_cast<uint16_t>(p - base + bs + rank);
        }
        p += X;
    }
    // the moves of the whole stretch: all payload loads in flight together, then the stores (a record's new place lies in
    // its own bucket, which only this wave touches)
    __builtin_amdgcn_wave_barrier();
    uint32_t mv[kFinishStage / 64];
    uint2 py[kFinishStage / 64];
    uint32_t ck[kFinishStage / 64];
#pragma unroll
    for (uint32_t k = 0; k < kFinishStage / 64u; ++k) {
        mv[k] = s_mv[64u * k + lane];
        py[k] = make_uint2(0u, 0u);
        ck[k] = 0u;
        if (g_ballot(mv[k] != 0xffffu) != 0ull) {  // (wave-uniform: every lane loads, its own record if it has no move)
            const uint32_t i = min(base + 64u * k + lane, V - 1u);
            py[k] = pay[i];
            if (kChk) ck[k] = chk[i];
        }
    }
    __builtin_amdgcn_wave_barrier();  // (every lane has loaded before any lane stores: lock step on the GPU, a meeting
                                      // point for the host emulator's lanes)
#pragma unroll
    for (uint32_t k = 0; k < kFinishStage / 64u; ++k) {
        if (mv[k] != 0xffffu) {
            const uint32_t dst = base + mv[k];
            ident[dst] = s_id[64u * k + lane];
            pay[dst] = py[k];
            if (kChk) chk[dst] = ck[k];
        }
    }
}

// test / tuning knobs (SLIMM_FORCE, force.h; read at every call: a test varies them inside one process): group_bits = hash
// bits of a bucket, group_width = widest digit of a pass, group_passes = number of passes (with the other two), group_grid =
// persistent workgroups of a pass.  (Two passes instead of three, measured at 10 M records, scripts/group_bits.sh: 2 x 11 bits
// 255 us per scatter pass + 107 us of finish, 2 x 10 bits 156 + 211, against 3 x 8 bits 96 + 30.)
uint32_t g_env_u32(const char* key) {
    long v = 0;
    return forced(key, &v) && v > 0 ? static_cast<uint32_t>(v) : 0u;
}

}  // namespace

GroupPlan group_plan(uint32_t n_records) {
    GroupPlan g;
    // As many buckets as records: nearly every bucket then holds ONE read, and the finish only reads the identities.  With
    // a quarter of that (1 B records of 8 hits per read in 2^28 buckets: half the buckets hold two reads) the finish
    // moves half the stream once more: 9.6 ms against 2.0 ms with 2^30 buckets, for 1.9 ms more in the passes.
    uint32_t b = 8;
    while (b < 32u && (1ull << b) < n_records) ++b;
    const uint32_t forced_bits = g_env_u32("group_bits");
    const uint32_t forced_width = g_env_u32("group_width");
    // A pass costs the same up to 8 bits per digit and more beyond (a round's records of one digit get fewer, the open
    // write frontiers more): measured per pass at 100 M / 1 B records with the reads interleaved at random, relative to
    // 8 bits: 9 bits 1.12 - 1.17, 10 bits 1.23 (inside the long bench process) - 1.41 (a fresh one), 11 bits ~1.9; the
    // count + scan of a pass 0.21.  One bit more than the records need halves the finish's moves (100 M records: 600 ->
    // 206 us).  The cheapest split of b or b + 1 bits into passes of 6 .. 11 bits, as even as it goes: 8 + 8 + 8 at 10 M
    // records, 9 + 9 + 9 at 100 M, 10 + 10 + 10 at 1 B -- where three wide passes and four narrow ones come out within a
    // few per cent of each other (6.44 - 6.80 ms at 100 M; 57.4 - 58.2 ms at 1 B in a fresh process, 59.3 against 64.6 ms
    // in the bench's).
    static const float kCost[12] = {0, 1, 1, 1, 1, 1, 1, 1, 1, 1.12f, 1.33f, 1.9f};
    float best = 1e30f;
    for (uint32_t bits = (forced_bits ? std::min(forced_bits, 32u) : b); bits <= (forced_bits ? std::min(forced_bits, 32u) : std::min(b + 1u, 32u));
         ++bits) {
        const uint32_t wcap = forced_width ? std::min(forced_width, kGroupMaxBits) : kGroupMaxBits;
        const uint32_t forced_passes = g_env_u32("group_passes");
        for (uint32_t P = (bits + wcap - 1u) / wcap; P <= kGroupMaxPasses; ++P) {
            const uint32_t lo = bits / P, hi = lo + (bits % P ? 1u : 0u);
            if (hi > wcap) continue;
            if (forced_passes && P != forced_passes && P < kGroupMaxPasses) continue;
            // (the finish: cheap once there are 1.6 buckets per record -- 10 M records in 2^24 buckets: 25 us = 0.27 of a pass)
            const bool roomy = forced_bits || (1ull << bits) * 10ull >= static_cast<uint64_t>(n_records) * 16ull;
            float cost = 0.21f * static_cast<float>(P) + (roomy ? 0.22f : 0.40f);
            for (uint32_t p = 0; p < P; ++p) cost += kCost[p < bits % P ? hi : std::max(lo, 1u)];
            if (forced_width && !forced_bits && P != (bits + wcap - 1u) / wcap) break;  // (a forced width: the fewest passes it allows)
            if (cost < best) {
                best = cost;
                g.passes = P;
                g.bits = bits;
                g.width = hi;
                for (uint32_t p = 0; p < kGroupMaxPasses; ++p) g.widths[p] = p < P ? (p < bits % P ? hi : lo) : 0u;
            }
            if (lo <= 6u) break;  // (narrower passes only add passes)
        }
    }
    g.grid = kGroupMaxGrid;
    if (const uint32_t o = g_env_u32("group_grid")) g.grid = std::min(o, kGroupMaxGrid);
    return g;
}

size_t group_hist_words(const GroupPlan& g) { return (static_cast<size_t>(g.grid) + 1u) << g.width; }

// where pass `pass` writes: the arrays alternate and the last pass ends in job.a
static const GroupArrays& gb_dest(const GroupJob& j, uint32_t pass) { return ((j.plan.passes - 1u - pass) & 1u) ? j.t : j.a; }
static uint32_t gb_shift(const GroupJob& j, uint32_t pass) {
    uint32_t below = 0;
    for (uint32_t p = 0; p < pass; ++p) below += j.plan.widths[p];
    return 32u - j.plan.bits + below;
}
template <bool kPacked>
static GbRaw<kPacked> gb_raw(const GroupJob& j) {
    GbRaw<kPacked> src;
    src.key = j.in.key, src.ref = j.in.ref, src.pos = j.in.pos, src.flag = j.in.flag, src.check = j.in.check, src.geo = j.geo;
    src.n = j.in.n, src.n_refs = j.n_refs, src.half_read = j.half_read, src.bin_width = j.bin_width;
    src.bw_magic = j.bin_width ? 0xffffffffu / j.bin_width : 0u;
    return src;
}

void launch_group_count(hipStream_t st, const GroupJob& j, uint32_t pass) {
    if (j.in.n == 0) return;
    const uint32_t G = j.plan.grid, W = j.plan.widths[pass], shift = gb_shift(j, pass);
    if (pass == 0) {
        if (j.in.packed)
            hipLaunchKernelGGL((k_gb_count<GbRaw<true>>), dim3(G), dim3(kGBlock), 0, st, gb_raw<true>(j), j.counters, shift, W, j.hist);
        else
            hipLaunchKernelGGL((k_gb_count<GbRaw<false>>), dim3(G), dim3(kGBlock), 0, st, gb_raw<false>(j), j.counters, shift, W, j.hist);
    } else {
        const GroupArrays& s = gb_dest(j, pass - 1u);
        hipLaunchKernelGGL((k_gb_count<GbIdent>), dim3(G), dim3(kGBlock), 0, st, GbIdent{s.ident, s.pay, s.chk}, j.counters, shift, W,
                           j.hist);
    }
}

void launch_group_scan(hipStream_t st, const GroupJob& j, uint32_t pass) {
    if (j.in.n == 0) return;
    const uint32_t W = j.plan.widths[pass];
    hipLaunchKernelGGL(k_gb_scan, dim3(1u << W), dim3(kGroupMaxGrid), 0, st, j.hist, j.plan.grid,
                       j.hist + (static_cast<size_t>(j.plan.grid) << W));
}

// ordered rounds through LDS (k_gb_scatter<.., kStaged>): up to 10-bit digits without check words (same box, 8-bit digits:
// 10 M records 94.5 -> 91.7 us per pass, 100 M records 886 -> 818); SLIMM_FORCE group_staged=0 / 1 forces
static bool gb_staged(uint32_t width, bool has_chk) {
    if (has_chk) return false;
    long st = 0;
    if (forced("group_staged", &st)) return st == 1;
    return width <= 10u && gb_lds_words(1u << width, true) <= kGLdsWords;
}

void launch_group_scatter(hipStream_t st, const GroupJob& j, uint32_t pass) {
    if (j.in.n == 0) return;
    const uint32_t G = j.plan.grid, W = j.plan.widths[pass], D = 1u << W, shift = gb_shift(j, pass);
    const uint32_t* totals = j.hist + (static_cast<size_t>(G) << W);
    const bool has_chk = j.in.check != nullptr;
    const bool staged = gb_staged(W, has_chk);
    const size_t lds = static_cast<size_t>(gb_lds_words(D, staged)) * 4u;
    const GroupArrays& o = gb_dest(j, pass);
    auto go = [&](auto src) {
        using Src = decltype(src);
        if (has_chk)
            hipLaunchKernelGGL((k_gb_scatter<Src, true, false>), dim3(G), dim3(kGBlock), lds, st, src, j.counters, shift, W, j.hist,
                               totals, o.ident, o.pay, o.chk);
        else if (staged)
            hipLaunchKernelGGL((k_gb_scatter<Src, false, true>), dim3(G), dim3(kGBlock), lds, st, src, j.counters, shift, W, j.hist,
                               totals, o.ident, o.pay, o.chk);
        else
            hipLaunchKernelGGL((k_gb_scatter<Src, false, false>), dim3(G), dim3(kGBlock), lds, st, src, j.counters, shift, W, j.hist,
                               totals, o.ident, o.pay, o.chk);
    };
    if (pass == 0) {
        if (j.in.packed)
            go(gb_raw<true>(j));
        else
            go(gb_raw<false>(j));
    } else {
        const GroupArrays& s = gb_dest(j, pass - 1u);
        go(GbIdent{s.ident, s.pay, s.chk});
    }
}

void launch_group_finish(hipStream_t st, const GroupJob& j) {
    if (j.in.n == 0) return;
    const uint32_t fgrid = (j.in.n + kFinishRecs - 1u) / kFinishRecs;
    if (j.in.check)
        hipLaunchKernelGGL((k_gb_finish<true>), dim3(fgrid), dim3(64), 0, st, j.a.ident, j.a.pay, j.a.chk, j.t.ident, j.t.pay, j.t.chk,
                           j.counters, 32u - j.plan.bits);
    else
        hipLaunchKernelGGL((k_gb_finish<false>), dim3(fgrid), dim3(64), 0, st, j.a.ident, j.a.pay, j.a.chk, j.t.ident, j.t.pay, j.t.chk,
                           j.counters, 32u - j.plan.bits);
}

int group_init() {
    // dynamic LDS beyond 64 KB needs the attribute
    uint32_t words = 0;  // the largest any width asks for (the lane-mask tables stop at kGMatchBits)
    for (uint32_t w = 1; w <= kGroupMaxBits; ++w) {
        words = std::max(words, gb_lds_words(1u << w, false));
        if (w <= 10u && gb_lds_words(1u << w, true) <= kGLdsWords) words = std::max(words, gb_lds_words(1u << w, true));  // (gb_staged)
    }
    const int lds = static_cast<int>(words * 4u);
    hipError_t e = hipSuccess;
    int nth = 0;
    auto set = [&](const void* f) {
        if (e != hipSuccess) return;
        e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess)
            fprintf(stderr, "slimm_hip: group_init: %d bytes of dynamic LDS for k_gb_scatter variant %d: %s\n", lds, nth, hipGetErrorString(e));
        ++nth;
    };
    set(reinterpret_cast<const void*>(k_gb_scatter<GbRaw<true>, true, false>));
    set(reinterpret_cast<const void*>(k_gb_scatter<GbRaw<true>, false, false>));
    set(reinterpret_cast<const void*>(k_gb_scatter<GbRaw<true>, false, true>));
    set(reinterpret_cast<const void*>(k_gb_scatter<GbRaw<false>, true, false>));
    set(reinterpret_cast<const void*>(k_gb_scatter<GbRaw<false>, false, false>));
    set(reinterpret_cast<const void*>(k_gb_scatter<GbRaw<false>, false, true>));
    set(reinterpret_cast<const void*>(k_gb_scatter<GbIdent, true, false>));
    set(reinterpret_cast<const void*>(k_gb_scatter<GbIdent, false, false>));
    set(reinterpret_cast<const void*>(k_gb_scatter<GbIdent, false, true>));
    return e == hipSuccess ? 0 : 1;
}

}  // namespace slimm

#if defined(EXP) && EXP == 10
extern "C" int slimm_debug_prof_group(unsigned long long* out, int n, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(slimm::g_prof_g), sizeof(unsigned long long) * n);
    if (reset) {
        static unsigned long long z[8 * 1024];
        (void)hipMemcpyToSymbol(HIP_SYMBOL(slimm::g_prof_g), z, sizeof(z));
    }
    return e == hipSuccess ? 0 : -1;
}
#endif
