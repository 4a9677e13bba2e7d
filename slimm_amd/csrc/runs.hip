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
__global__ __launch_bounds__(kRBlock) void k_runs(const Acc acc, uint32_t* __restrict__ counters, uint8_t* __restrict__ fl,
                                                  uint2* __restrict__ tile_cnt, uint32_t* __restrict__ tile_valid) {
    __shared__ uint32_t s_meta[kRTile + kHalo + kRBlock];
    __shared__ uint2 s_w[kRWaves];
    __shared__ uint32_t s_v[kRWaves];
    const uint32_t N = acc.count(counters);
    const uint32_t base = blockIdx.x * kRTile;
    uint32_t nh = 0, nf = 0, nv = 0;
    bool bad = false, too_long = false;
    if (base < N) {
        const uint32_t lds_lo = base >= kHalo ? base - kHalo : 0u;
        const uint32_t lds_hi = min(base + static_cast<uint32_t>(kRTile), N);
        const uint32_t lane = threadIdx.x & 63u;
        // ---- stage the meta words: all global loads of the tile are issued before the first one is consumed ----
        constexpr int kFill = (kRTile + kHalo + kRBlock - 1) / kRBlock;  // 9 slices of 256 records
        uint64_t key[kFill], kprev0[kFill];
        uint32_t meta[kFill];
#pragma unroll
        for (int k = 0; k < kFill; ++k) {
            const uint32_t i = lds_lo + k * kRBlock + threadIdx.x;
            const bool live = i < lds_hi;
            key[k] = live ? acc.key_of(i) : 0ull;
            meta[k] = live ? acc.meta_of(i, bad) : 0u;
            // the predecessor's key comes from the neighbouring lane; lane 0 of each wave fetches its own
            kprev0[k] = (live && lane == 0 && i > 0) ? acc.key_of(i - 1) : 0ull;
        }
#pragma unroll
        for (int k = 0; k < kFill; ++k) {
            const uint32_t i = lds_lo + k * kRBlock + threadIdx.x;
            uint64_t prev = __shfl_up(key[k], 1, 64);
            if (lane == 0) prev = kprev0[k];
            if (i == 0 || key[k] != prev) meta[k] |= M_RUN;
            if (i < lds_hi) s_meta[i - lds_lo] = meta[k];
        }
        __syncthreads();
        // ---- classify: look-back through LDS, one distance per trip for the whole wave (wave-uniform loop with a
        // predicated body), four 256-record slices interleaved so four independent LDS reads are in flight per trip ----
        constexpr int kIlp = 4;
#pragma unroll 1
        for (int k0 = 0; k0 < kRItems; k0 += kIlp) {
            uint32_t li[kIlp], me[kIlp], my_ident[kIlp], my_mate[kIlp];
            bool live[kIlp], valid[kIlp], head[kIlp], first[kIlp], gb[kIlp], active[kIlp], open[kIlp];
            bool any_active = false;
#pragma unroll
            for (int u = 0; u < kIlp; ++u) {
                const uint32_t i = base + (k0 + u) * kRBlock + threadIdx.x;
                live[u] = i < N;
                li[u] = live[u] ? i - lds_lo : 0u;
                me[u] = live[u] ? s_meta[li[u]] : M_RUN;  // dead lanes: not mapped, never walk
                valid[u] = me[u] & M_VALID;
                my_ident[u] = me[u] & M_IDENT;
                my_mate[u] = (me[u] >> 28) & 3u;
                head[u] = first[u] = true;
                gb[u] = open[u] = false;
                active[u] = valid[u] && !(me[u] & M_RUN);
                any_active = any_active || active[u];
            }
            for (uint32_t d = 1; __ballot(any_active) != 0ull; ++d) {
                uint32_t m[kIlp];
                bool in[kIlp];
#pragma unroll
                for (int u = 0; u < kIlp; ++u) {
                    in[u] = active[u] && d <= li[u];
                    m[u] = s_meta[in[u] ? li[u] - d : 0u];
                }
                any_active = false;
#pragma unroll
                for (int u = 0; u < kIlp; ++u) {
                    const bool v = in[u] && (m[u] & M_VALID);
                    const uint32_t mt = (m[u] >> 28) & 3u;
                    const bool same = v && ((m[u] & M_IDENT) == my_ident[u]);
                    head[u] = head[u] && !(v && mt == my_mate[u]);
                    first[u] = first[u] && !same;
                    gb[u] = gb[u] || (v && mt > my_mate[u]);
                    open[u] = open[u] || (active[u] && !in[u]);
                    active[u] = in[u] && !same && !(m[u] & M_RUN);
                    any_active = any_active || active[u];
                }
            }
#pragma unroll
            for (int u = 0; u < kIlp; ++u) {
                if (open[u]) {  // the run reaches back beyond the halo: continue in global memory (rare)
                    uint32_t j = lds_lo, steps = 0;
                    while (j > 0) {
                        --j;
                        bool dummy = false;
                        const uint32_t m = full_meta(acc, j, dummy);
                        if (m & M_VALID) {
                            const uint32_t mt = (m >> 28) & 3u;
                            if ((m & M_IDENT) == my_ident[u]) {
                                head[u] = false;
                                first[u] = false;
                                break;
                            }
                            head[u] = head[u] && (mt != my_mate[u]);
                            gb[u] = gb[u] || (mt > my_mate[u]);
                        }
                        if (m & M_RUN) break;
                        if (++steps > kLookBackMax) {
                            too_long = true;
                            break;
                        }
                    }
                }
                const bool h = head[u] && valid[u], f1 = first[u] && valid[u];
                const uint32_t f = (my_mate[u] << FL_MATE_SHIFT) | ((me[u] & M_RUN) ? FL_RUN_START : 0u) |
                                   (h ? FL_HEAD : 0u) | (f1 ? FL_FIRST : 0u) |
                                   ((valid[u] && gb[u]) ? FL_GREATER_BEFORE : 0u);
                nh += h;
                nf += f1;
                nv += valid[u];
                if (live[u]) fl[base + (k0 + u) * kRBlock + threadIdx.x] = static_cast<uint8_t>(f);
            }
        }
    }
    nh = r_wave_sum(nh);
    nf = r_wave_sum(nf);
    nv = r_wave_sum(nv);
    const uint32_t err = (__any(bad) ? ERR_REF_RANGE : 0u) | (__any(too_long) ? ERR_RUN_LENGTH : 0u);
    if ((threadIdx.x & 63) == 0) {
        s_w[threadIdx.x >> 6] = make_uint2(nh, nf);
        s_v[threadIdx.x >> 6] = nv;
        if (err) atomicOr(&counters[CNT_ERR], err);
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
        tile_cnt[blockIdx.x] = t;
        if (Acc::kCountsMapped) tile_valid[blockIdx.x] = v;  // summed into hits_count by k_scan_tiles (src/slimm.hpp:212)
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
        {   // earlier records of the run with a larger mate sit AFTER this one in the target order
            bool act = first && (f & FL_GREATER_BEFORE) && !(f & FL_RUN_START);
            bool open = false;
            for (uint32_t d = 1; __ballot(act) != 0ull; ++d) {
                const bool in = act && d <= li;
                const uint32_t g = s_fl[in ? li - d : 0u];
                const bool bigger = in && ((g >> FL_MATE_SHIFT) & 3u) > mate;
                t -= (bigger && (g & FL_FIRST)) ? 1u : 0u;
                m -= (bigger && (g & FL_HEAD)) ? 1u : 0u;
                open = open || (act && !in);
                act = in && !(g & FL_RUN_START);
            }
            if (open) {
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
            bool act = first && mate > 0;
            bool open = false;
            for (uint32_t d = 1; __ballot(act) != 0ull; ++d) {
                const bool in = act && (i + d) < lds_hi;
                const uint32_t g = s_fl[in ? li + d : 0u];
                const bool stop = in && (g & FL_RUN_START);
                const bool smaller = in && !stop && ((g >> FL_MATE_SHIFT) & 3u) < mate;
                t += (smaller && (g & FL_FIRST)) ? 1u : 0u;
                m += (smaller && (g & FL_HEAD)) ? 1u : 0u;
                open = open || (act && !in && (i + d) < N);
                act = in && !stop;
            }
            if (open) {
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
                     uint2* tile_cnt, uint32_t* tile_valid) {
    const uint32_t nt = rtiles(in.n);
    if (!nt) return;
    hipLaunchKernelGGL(k_runs<RawRecords>, dim3(nt), dim3(kRBlock), 0, st,
                       make_raw(in, n_refs, ref_len, bin_off, half_read, bin_width), counters, fl, tile_cnt, tile_valid);
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
                        uint32_t* counters, uint8_t* fl, uint2* tile_cnt) {
    const uint32_t nt = rtiles(n_upper);
    if (!nt) return;
    SortedRecords a{ident, cref, cgbin};
    hipLaunchKernelGGL(k_runs<SortedRecords>, dim3(nt), dim3(kRBlock), 0, st, a, counters, fl, tile_cnt,
                       static_cast<uint32_t*>(nullptr));
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
