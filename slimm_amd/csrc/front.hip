// Front end of phase A (gfx950, wave64): ONE pass over the record stream, from the records to the per-read target lists.
//
// Reference semantics: src/slimm.hpp:194-213 (record filter, bin, read identity = qName + mate number) and
// src/read_stat.hpp:116-135 (add_target keeps the bin of the FIRST record of a (read, reference) pair: quirk Q1);
// the unique bit is the `reads.size() == 1` test of src/slimm.hpp:224-237.
//
// Shape.  A wave owns slots of kSlotRecs consecutive records.  Per slot:
//   1. STAGE: the slot's records are loaded block by block (64 records, six blocks in flight together), the bin of every
//      record is computed (one 8-byte gather of its contig's geometry) and two words per record -- {reference, mate,
//      mapped} and the bin -- go to the wave's own stretch of LDS; comparing every key with the key of the lane before
//      (DPP wave shift, one ballot per block) gives the slot's run-start bitmap.  This is the only part of the kernel
//      that waits for memory, and everything it loads is in flight at once.
//   2. CUT: the slot is cut into WINDOWS of up to 64 records that start at a qName run start and end behind the last run
//      that is complete inside them -- scalar arithmetic on the bitmap.  Every window holds whole runs only: nothing
//      about a read is ever carried from one window, wave or workgroup to the next -- no halo, no barrier, no scan over
//      tiles.
//   3. CLASSIFY: per window the staged words come back from LDS and everything is lane-mask arithmetic:
//        segment starts       run starts | "mate differs from the lane before"
//        first of (read, ref) Q1: a tagged word {segment start, reference} is shifted along the lanes one step at a time
//                             and compared with the lane's own; the trip count is the window's longest segment
//        head of a read       "first mapped lane at or after each segment start": ONE 64-bit scalar add -- the carry
//                             of ~mapped + starts ripples from every start to the first mapped lane behind it
//        unique reads         a head is unique iff the next target is a head again: the same carry trick on the
//                             bit-reversed masks finds the target in front of every non-head target
//      and the targets are written compacted behind the slot's first run start (rank = popcount of the lanes below).
// A slot's targets therefore sit at [start, start + nf) with start = index of its first run start and nf <= the records
// the slot is responsible for: the consumers walk slots, no prefix sum over the stream is ever needed.
//
// Three paths per window, chosen by wave-uniform tests:
//   fast     the mate numbers of every run are non-decreasing (mapper output "all of mate 1, then all of mate 2",
//            unpaired data): a read is a contiguous segment
//   general  mates interleave inside a run: per-lane walks over the run (lane shuffles), targets of a run reordered by
//            mate so that reads stay contiguous
//   long     a run of 64 records or more: chunks of 64, every chunk compared with all chunks before it by lane
//            rotation (quadratic in the run length, without any limit on it), one pass per mate number when the run's
//            mates interleave
//
// Output (all 32-bit):
//   tgt_ref [p]  reference id | bit 31: first target of its read
//   tgt_gbin[p]  global bin (bin_off[ref] + bin) | bit 31: the read has exactly one target (src/slimm.hpp:224)
//   slots   [s]  {start, targets, reads, mapped records} of slot s
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>

#include "kernels.h"

namespace slimm {

#if defined(EXP) && EXP == 11  // cycle split of k_front: lane 0 of every (one-wave) workgroup (scripts/tprof_front.py)
__device__ unsigned long long g_prof_f[4 * 8192];
#define FPROF_T(x) const unsigned long long x = __builtin_readcyclecounter()
#define FPROF_ADD(slot, a, b) if (threadIdx.x == 0) atomicAdd(&g_prof_f[(blockIdx.x & 8191u) * 4 + slot], (b) - (a))
#else
#define FPROF_T(x)
#define FPROF_ADD(slot, a, b)
#endif

namespace {

constexpr uint32_t kTagShift = 26;                       // a staged record's low 26 bits: reference + 1 (slimm_create
constexpr uint32_t kRefField = (1u << kTagShift) - 1u;   // keeps n_refs <= 2^26 - 2), all ones for an unmapped record
constexpr uint32_t kNoMatch = 0xffffffffu;               // shifted in at lane 0: equals no tagged word
constexpr uint32_t kStMateShift = 26;                    // ... bits 26-27: mate number
constexpr uint32_t kStRunStart = 1u << 31;               // ... bit 31: the record starts a qName run
constexpr uint32_t kSlotBlocks = kSlotRecs / 64 + 1;     // blocks of 64 records staged per slot: one more than the slot,
                                                         // so that a window starting in the slot's last record is covered
static_assert(kMaxRefs < kRefField, "reference + 1 must stay below the unmapped marker");

__device__ __forceinline__ uint32_t f_lane() {
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}
__device__ __forceinline__ uint32_t f_rank(uint64_t mask) {  // set bits of `mask` below this lane
    return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32),
                                     __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
}
__device__ __forceinline__ uint64_t f_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool f_bit(uint64_t wave_uniform_mask) {  // this lane's bit: one v_cndmask on the SGPR pair
    return __builtin_amdgcn_inverse_ballot_w64(wave_uniform_mask);
}
// the value of the lane before (lane 0 gets `lane0`)
__device__ __forceinline__ uint32_t f_shr1(uint32_t v, uint32_t lane0) {
    return __builtin_amdgcn_update_dpp(lane0, v, 0x138, 0xf, 0xf, false);
}
// the value of the lane before, zero at lane 0 (the form the compiler folds into the instruction that uses it)
__device__ __forceinline__ uint32_t f_shr1z(uint32_t v) {
    return __builtin_amdgcn_update_dpp(0u, v, 0x138, 0xf, 0xf, true);
}
__device__ __forceinline__ uint32_t f_ror1(uint32_t v) {  // lane i gets lane i - 1, lane 0 gets lane 63
    return __builtin_amdgcn_update_dpp(v, v, 0x13c, 0xf, 0xf, false);
}
__device__ __forceinline__ uint64_t f_below_nz(uint32_t n) {  // lanes 0 .. n-1 for 1 <= n <= 64: two scalar instructions
    return ~0ull >> (64u - n);
}
__device__ __forceinline__ uint64_t f_below(uint32_t n) {  // lanes 0 .. n-1 (n <= 64)
    return n >= 64u ? ~0ull : ((1ull << n) - 1ull);
}
// "first bit of `bits` at or after every bit of `starts`, looking no further than the next stop": adding `starts` to a
// word that is one where nothing is to be found lets the carry ripple from each start up to the first zero, which is the
// lane looked for (or the stop lane, which `& bits` removes again when it is no hit)
__device__ __forceinline__ uint64_t f_first_after(uint64_t starts, uint64_t bits, uint64_t stops) {
    const uint64_t ones = ~bits & ~stops;
    return (ones + starts) & ~ones & bits;
}


// ---------------------------------------------------------------------------------------------------------
// First occurrence of a key among the lanes of a wave -- and among everything the table has seen since it was cleared --
// in a handful of LDS operations, whatever the number of equal keys: the duplicate test of Q1 (src/read_stat.hpp:125-131)
// for segments too long for the lane-shift walk (one step per record of the longest segment) and for runs of 64 records
// or more (where the walk is quadratic in the run length).  The wave's own table of kHashSlots {key, first position}
// entries in LDS: a lane claims the entry of its key by compare-and-swap from "empty" (linear probing), every lane of a
// key then lowers the entry's position to its own (ds_min), and the one lane that reads its own position back is the
// first record of that (read, reference) pair in file order.  No barrier: one wave, and the LDS executes a wave's
// operations in order.  (The ballots between the steps are what the host emulator of tests/native synchronises its
// lanes on; on the GPU they are the loop test and a compare.)
#ifndef SLIMM_HASH_BITS
#define SLIMM_HASH_BITS 7
#endif
constexpr uint32_t kHashBits = SLIMM_HASH_BITS;
constexpr uint32_t kHashSlots = 1u << kHashBits;   // 128 entries of 8 bytes: 1 KB per wave, half full at 64 keys
static_assert(kHashBits >= 7, "hash_clear writes 128 entries at a time");
__device__ __forceinline__ void hash_clear(uint32_t* tab, uint32_t lane) {
#pragma unroll
    for (uint32_t k = 0; k < kHashSlots / 128u; ++k)  // 64 lanes x 2 entries
        reinterpret_cast<uint4*>(tab)[64u * k + lane] = make_uint4(0u, 0xffffffffu, 0u, 0xffffffffu);
}
// key != 0; `pos` = the lane's position in file order among everything inserted since hash_clear.  Returns whether the
// lane is the first of its key; overflow: the table is full (more than kHashSlots distinct keys) -- the caller falls
// back to the comparison walk.
__device__ __forceinline__ bool hash_first(uint32_t* tab, uint32_t key, uint32_t pos, bool active, bool& overflow) {
    // Straight-line on purpose (the scalar unit is this kernel's bottleneck, and every `if (pending)` around an LDS
    // atomic is a dozen scalar instructions of exec-mask bookkeeping per trip): EVERY lane issues every operation.  A
    // lane that has its entry finds its own key there again, a lane with nothing to insert swaps nothing into nothing
    // (key 0 for "empty"), and lowers no position (0xffffffff).
    uint2* const e = reinterpret_cast<uint2*>(tab);
#ifdef SLIMM_HASH_LINEAR
    uint32_t slot = (key + (key >> 7) * 37u + (key >> 26) * 11u) & (kHashSlots - 1u);
#else
    uint32_t slot = (key * 0x9E3779B1u) >> (32u - kHashBits);
#endif
    const uint32_t mine = active ? key : 0u;
    bool pending = active;
    uint32_t probes = 0;
    while (f_ballot(pending) != 0ull) {  // (tested at the top: one call site for the collective, see tests/native/README.md)
        const uint32_t old = atomicCAS(&e[slot].x, 0u, mine);
        pending = pending & (old != 0u) & (old != key);          // neither claimed nor found: the next entry
        slot = pending ? ((slot + 1u) & (kHashSlots - 1u)) : slot;
        probes += pending ? 1u : 0u;
        const bool full = probes >= kHashSlots;                   // every entry holds another key
        overflow = overflow | full;
        pending = pending & !full;
    }
    const bool placed = active & !overflow;
    atomicMin(&e[slot].y, placed ? pos : 0xffffffffu);
    const bool lowered = f_ballot(placed) != 0ull;  // (every lane's minimum is in before any lane reads)
    const uint32_t got = lowered ? __hip_atomic_load(&e[slot].y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0xffffffffu;
    return placed && got == pos;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------
// record sources: the caller's arrays (grouped input) or the compacted, identity-sorted stream (record_order = ANY)
// ---------------------------------------------------------------------------------------------------------
struct FrontRec {  // what a window needs of a record besides its place in a run
    uint32_t mate, ref, aux;
    bool mapped;
};
struct FrontRaw3 {  // the three words a window loads per record, as they come from memory
    uint32_t a, b, c;
};
struct FrontLoaded {  // a record as the staging loop loads it: key halves and the three words
    uint32_t klo, khi, a, b, c;
};
// element `byte_off / sizeof(T)` behind a wave-uniform pointer: a scalar base and ONE 32-bit lane offset per load
// instead of a 64-bit address computed per lane and array
template <typename T>
__device__ __forceinline__ T f_load_at(const T* uniform_base, uint32_t byte_off) {
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(uniform_base) + byte_off);
}

// kPacked: 16 bytes per record -- the flag bits the path reads ride in the key's top three bits (bit 63: unmapped,
// src/slimm.hpp:197; bits 62-61: mate number 0 / 1 / 2, src/slimm.hpp:205-208), the qName identity is the low 61 bits and
// there is no flag array (slimm_push_records_packed).
template <bool kPacked>
struct FrontRawT {
    static constexpr uint32_t kHiMask = kPacked ? 0x1fffffffu : 0x3fffffffu;  // identity bits of a key's high word
    const uint64_t* key;
    const int32_t* ref;
    const int32_t* pos;
    const uint16_t* flag;
    const uint32_t* check;  // optional: a second hash of the read name (null: none)
    const uint2* geo;  // {contig length, first bin} per reference
    uint32_t n, n_refs, half_read, bin_width, bw_magic;
    static constexpr bool kCountsMapped = true;   // hits_count (src/slimm.hpp:212) is counted here
    static constexpr bool kMarked = false;        // run starts come from comparing adjacent keys
    __device__ uint32_t count(const uint32_t*) const { return n; }
    __device__ bool same_run(uint32_t i, uint32_t, uint32_t klo0, uint32_t khi0) const {
        uint32_t lo, hi;
        key_at(i, lo, hi);
        return lo == klo0 && hi == khi0;
    }
    // qName identity (the top two bits of the key are not significant)
    __device__ void key_at(uint32_t i, uint32_t& lo, uint32_t& hi) const {
        const uint64_t k = key[i];
        lo = static_cast<uint32_t>(k);
        hi = static_cast<uint32_t>(k >> 32) & kHiMask;
    }
    __device__ FrontRaw3 raw(uint32_t i) const {
        if (kPacked)
            return FrontRaw3{reinterpret_cast<const uint32_t*>(key)[2 * static_cast<size_t>(i) + 1], static_cast<uint32_t>(ref[i]),
                             static_cast<uint32_t>(pos[i])};
        return FrontRaw3{flag[i], static_cast<uint32_t>(ref[i]), static_cast<uint32_t>(pos[i])};
    }
    // ---- the staging loop's view: record base + rel (base wave-uniform)
    __device__ void load(uint32_t base, uint32_t rel, FrontLoaded& o) const {
        const uint2 k = f_load_at(reinterpret_cast<const uint2*>(key) + base, rel * 8u);
        o.klo = k.x;
        o.khi = k.y;  // (nothing but loads here: key_fix below, once the whole group is on its way)
        o.a = kPacked ? k.y : static_cast<uint32_t>(f_load_at(flag + base, rel * 2u));
        o.b = static_cast<uint32_t>(f_load_at(ref + base, rel * 4u));
        o.c = static_cast<uint32_t>(f_load_at(pos + base, rel * 4u));
    }
    __device__ void load_key(uint32_t base, uint32_t rel, uint32_t& lo, uint32_t& hi) const {
        const uint2 k = f_load_at(reinterpret_cast<const uint2*>(key) + base, rel * 8u);
        lo = k.x;
        hi = k.y;
    }
    __device__ static void key_fix(uint32_t&, uint32_t& hi) { hi &= kHiMask; }  // qName identity: 62 (61) bits
    __device__ uint32_t load_check(uint32_t base, uint32_t rel) const { return f_load_at(check + base, rel * 4u); }
    __device__ static void no_key(uint32_t& lo, uint32_t& hi) {  // differs from every key (no identity bit)
        lo = 0u;
        hi = kHiMask + 1u;
    }
    __device__ uint2 geo_at(const FrontLoaded& w) const { return f_load_at(geo, (w.b < n_refs ? w.b : 0u) * 8u); }
    // field = reference + 1 of a mapped record (src/slimm.hpp:197), kRefField otherwise; mate: src/slimm.hpp:205-208
    // worst: the largest reference + 1 of an aligned record seen so far (above n_refs: neither -1 nor a reference -- the
    // caller's test, once per slot; a flag kept per record is four vector instructions each)
    __device__ void fields(const FrontLoaded& w, uint32_t& field, uint32_t& mate, uint32_t& worst) const {
        const bool aligned = kPacked ? (static_cast<int32_t>(w.a) >= 0) : ((w.a & 0x4u) == 0u);
        const uint32_t t = aligned ? w.b + 1u : 0u;  // (reference -1 wraps to 0 as well)
        worst = max(worst, t);
        field = (t - 1u < n_refs) ? t : kRefField;
        if (kPacked) {
            mate = (w.a >> 29) & 3u;
        } else {
            const uint32_t c = (w.a >> 6) & 3u;     // first-in-pair wins over last-in-pair
            mate = c == 3u ? 1u : c;
        }
    }
    __device__ uint32_t gbin_of(const FrontLoaded& w, const uint2& g) const {
        return g.y + div_bin_width(min(w.c + half_read, g.x));
    }
    __device__ bool out_of_range(uint32_t worst) const { return worst > n_refs; }
    __device__ FrontRec decode(const FrontRaw3& w, bool& bad) const {
        FrontRec o;
        if (kPacked) {
            o.mate = (w.a >> 29) & 3u;
            o.mapped = static_cast<int32_t>(w.a) >= 0 && w.b != 0xffffffffu;
        } else {
            o.mate = (w.a & 0x40u) ? 1u : ((w.a & 0x80u) ? 2u : 0u);            // src/slimm.hpp:205-208
            o.mapped = !(w.a & 0x4u) && w.b != 0xffffffffu;                      // src/slimm.hpp:197
        }
        if (o.mapped && w.b >= n_refs) {
            bad = true;
            o.mapped = false;
        }
        o.ref = w.b;
        o.aux = w.c;
        return o;
    }
    __device__ FrontRec rec(uint32_t i, bool& bad) const { return decode(raw(i), bad); }
    // the contig's {length, first bin}: one 8-byte gather (lanes whose reference is no index of the table gather row 0)
    __device__ uint2 geo_of(const FrontRaw3& w) const { return geo[w.b < n_refs ? w.b : 0u]; }
    // n / bin_width with a host-computed reciprocal: mulhi(n, floor((2^32 - 1) / d)) is the quotient or one less
    __device__ uint32_t div_bin_width(uint32_t v) const {
        const uint32_t q = __umulhi(v, bw_magic);
        const uint32_t r = v - q * bin_width;
        return q + (r >= bin_width ? 1u : 0u);
    }
    // bin of the record: uint32 wrap-around of int32 + uint32, then the clamp to the contig length
    // (src/slimm.hpp:200-201, quirk Q3)
    __device__ uint32_t gbin(const FrontRec& r, const uint2& g) const {
        return g.y + div_bin_width(min(r.aux + half_read, g.x));
    }
};
using FrontRaw = FrontRawT<false>;
using FrontPacked = FrontRawT<true>;

struct FrontSorted {
    const uint64_t* ident;  // key << 2 | mate
    const uint2* pay;       // {reference, global bin}
    const uint32_t* check;  // optional, grouped along with the records
    static constexpr bool kCountsMapped = false;  // the compaction counted the mapped records
    static constexpr bool kMarked = false;
    __device__ uint32_t count(const uint32_t* counters) const { return counters[CNT_V]; }
    __device__ bool same_run(uint32_t i, uint32_t, uint32_t klo0, uint32_t khi0) const {
        uint32_t lo, hi;
        key_at(i, lo, hi);
        return lo == klo0 && hi == khi0;
    }
    __device__ void key_at(uint32_t i, uint32_t& lo, uint32_t& hi) const {
        const uint64_t k = ident[i];
        lo = static_cast<uint32_t>(k) & ~3u;
        hi = static_cast<uint32_t>(k >> 32);
    }
    __device__ FrontRaw3 raw(uint32_t i) const {
        const uint2 q = pay[i];
        return FrontRaw3{reinterpret_cast<const uint32_t*>(ident)[2 * static_cast<size_t>(i)], q.x, q.y};
    }
    __device__ FrontRec decode(const FrontRaw3& w, bool&) const { return FrontRec{w.a & 3u, w.b, w.c, true}; }
    __device__ FrontRec rec(uint32_t i, bool& bad) const { return decode(raw(i), bad); }
    __device__ uint2 geo_of(const FrontRaw3&) const { return make_uint2(0u, 0u); }
    __device__ uint32_t gbin(const FrontRec& r, const uint2&) const { return r.aux; }
    __device__ void load(uint32_t base, uint32_t rel, FrontLoaded& o) const {
        const uint2 k = f_load_at(reinterpret_cast<const uint2*>(ident) + base, rel * 8u);
        o.klo = k.x;
        o.khi = k.y;
        o.a = k.x;
        const uint2 q = f_load_at(pay + base, rel * 8u);
        o.b = q.x;
        o.c = q.y;
    }
    __device__ void load_key(uint32_t base, uint32_t rel, uint32_t& lo, uint32_t& hi) const {
        const uint2 k = f_load_at(reinterpret_cast<const uint2*>(ident) + base, rel * 8u);
        lo = k.x;
        hi = k.y;
    }
    __device__ static void key_fix(uint32_t& lo, uint32_t&) { lo &= ~3u; }  // (the mate number rides in the low bits)
    __device__ uint32_t load_check(uint32_t base, uint32_t rel) const { return f_load_at(check + base, rel * 4u); }
    __device__ static void no_key(uint32_t& lo, uint32_t& hi) {  // (the low two bits of every key's low word are clear)
        lo = 1u;
        hi = 0u;
    }
    __device__ uint2 geo_at(const FrontLoaded&) const { return make_uint2(0u, 0u); }
    __device__ void fields(const FrontLoaded& w, uint32_t& field, uint32_t& mate, uint32_t&) const {
        field = w.b + 1u;
        mate = w.a & 3u;
    }
    __device__ bool out_of_range(uint32_t) const { return false; }
    __device__ uint32_t gbin_of(const FrontLoaded& w, const uint2&) const { return w.c; }
};

// Run-marked records, 8 bytes each (slimm_push_records_marked): for input that is grouped by qName the device never
// needed the names -- only where a run of equal names starts.  word = reference + 1 (0: not mapped: the unmapped flag or
// reference -1, src/slimm.hpp:197) | mate number << 29 (src/slimm.hpp:205-208) | "this record starts a qName run" << 31;
// no key array crosses the bus or is read by the front end.
struct FrontMarked {
    const uint32_t* word;
    const int32_t* pos;
    const uint32_t* check = nullptr;  // (none in this form)
    const uint2* geo;  // {contig length, first bin} per reference
    uint32_t n, n_refs, half_read, bin_width, bw_magic;
    static constexpr bool kCountsMapped = true;
    static constexpr bool kMarked = true;
    static constexpr uint32_t kRefBits = 0x1fffffffu;
    __device__ uint32_t count(const uint32_t*) const { return n; }
    __device__ void key_at(uint32_t, uint32_t& lo, uint32_t& hi) const { lo = hi = 0u; }
    // record i belongs to the run that starts at record `first`
    __device__ bool same_run(uint32_t i, uint32_t first, uint32_t, uint32_t) const {
        return i == first || static_cast<int32_t>(word[i]) >= 0;
    }
    __device__ FrontRaw3 raw(uint32_t i) const { return FrontRaw3{word[i], (word[i] & kRefBits) - 1u, static_cast<uint32_t>(pos[i])}; }
    __device__ void load(uint32_t base, uint32_t rel, FrontLoaded& o) const {
        o.klo = o.khi = 0u;
        o.a = f_load_at(word + base, rel * 4u);
        o.b = (o.a & kRefBits) - 1u;  // the reference (0xffffffff: none)
        o.c = static_cast<uint32_t>(f_load_at(pos + base, rel * 4u));
    }
    __device__ void load_key(uint32_t, uint32_t, uint32_t& lo, uint32_t& hi) const { lo = hi = 0u; }
    __device__ static void key_fix(uint32_t&, uint32_t&) {}
    __device__ uint32_t load_check(uint32_t, uint32_t) const { return 0u; }
    __device__ static void no_key(uint32_t& lo, uint32_t& hi) { lo = hi = 1u; }
    __device__ uint2 geo_at(const FrontLoaded& w) const { return f_load_at(geo, (w.b < n_refs ? w.b : 0u) * 8u); }
    __device__ void fields(const FrontLoaded& w, uint32_t& field, uint32_t& mate, uint32_t& worst) const {
        const uint32_t t = w.a & kRefBits;  // reference + 1
        worst = max(worst, t);
        field = (t - 1u < n_refs) ? t : kRefField;
        mate = (w.a >> 29) & 3u;
    }
    __device__ static bool starts_run(const FrontLoaded& w) { return static_cast<int32_t>(w.a) < 0; }
    __device__ bool out_of_range(uint32_t worst) const { return worst > n_refs; }
    __device__ uint32_t div_bin_width(uint32_t v) const {
        const uint32_t q = __umulhi(v, bw_magic);
        const uint32_t r = v - q * bin_width;
        return q + (r >= bin_width ? 1u : 0u);
    }
    __device__ uint32_t gbin_of(const FrontLoaded& w, const uint2& g) const {
        return g.y + div_bin_width(min(w.c + half_read, g.x));
    }
    __device__ FrontRec decode(const FrontRaw3& w, bool& bad) const {
        FrontRec o;
        o.mate = (w.a >> 29) & 3u;
        o.mapped = w.b != 0xffffffffu;
        if (o.mapped && w.b >= n_refs) {
            bad = true;
            o.mapped = false;
        }
        o.ref = w.b;
        o.aux = w.c;
        return o;
    }
    __device__ FrontRec rec(uint32_t i, bool& bad) const { return decode(raw(i), bad); }
    __device__ uint2 geo_of(const FrontRaw3& w) const { return geo[w.b < n_refs ? w.b : 0u]; }
    __device__ uint32_t gbin(const FrontRec& r, const uint2& g) const { return g.y + div_bin_width(min(r.aux + half_read, g.x)); }
};

namespace {

struct WinOut {  // where the slot's targets go and what it has counted so far (all wave-uniform)
    uint32_t base;            // index of the slot's first run start = position of its first target
    uint32_t nf, nh, nv;      // targets, reads, mapped records
};

// a record as the general path sees it
struct Staged {
    uint32_t mate, ref, gbin;
    bool mapped;
};

// Segments longer than this take the hash table in window_fast (a walk step is two vector and two scalar instructions,
// the table about fifty instructions and four LDS round trips whatever the segments look like)
#ifndef SLIMM_HASH_WALK
#define SLIMM_HASH_WALK 32
#endif
constexpr uint32_t kHashWalk = SLIMM_HASH_WALK;
// D = N0 & (N0 << 1) marks the second and later lanes of every stretch of set bits of N0 (non-start lanes: a segment of
// L records is a stretch of L - 1).  Is there a stretch of D of at least n bits (a segment of more than n + 1 records)?
__device__ __forceinline__ bool has_run_of(uint64_t D, uint32_t n) {
    // stretches of >= 2^k bits by doubling, then the remainder
    uint64_t x = D;
    uint32_t have = 1;
    while (have * 2u <= n) {
        x &= x << have;
        have *= 2u;
    }
    if (have < n) x &= x << (n - have);
    return x != 0ull;
}

// ---------------------------------------------------------------------------------------------------------
// fast path: lanes [0, X) of the window hold whole runs whose mates never decrease.  field = the staged reference field,
// SS = segment starts (run starts and mate changes), V = mapped lanes, both inside [0, X).
// ---------------------------------------------------------------------------------------------------------
struct WinMasks {  // what the caller needs to cut the filter's windows: the window's targets and read heads
    uint64_t F, H;
};
__device__ __forceinline__ WinMasks window_fast(uint32_t field, uint32_t gbin, uint32_t lane, uint64_t SS, uint64_t V, uint32_t X,
                                                WinOut& so, uint32_t* __restrict__ tgt_ref, uint32_t* __restrict__ tgt_gbin,
                                                uint32_t* tab) {
    // Q1: an earlier lane of my segment with my reference?  T = {segment number, field} is never zero and equal only
    // inside a segment; x_d[i] = T[i - d] ^ T[i] (T of the lanes in front of lane 0 taken as zero) follows from
    // x_{d+1}[i] = x_d[i - 1] ^ x_1[i]: ONE xor with a lane shift per step.  A lane is a duplicate iff some x_d is zero.
    // Steps beyond a lane's own segment compare it with other segments' words -- never equal, so nothing guards them;
    // the trip count follows the longest segment of THIS window (scalar: D = lanes at least d behind their start).
    // A window with a segment of more than kHashWalk records takes the hash table instead: the walk's trip count is the
    // longest segment, the table's cost is the same whatever the segments look like (config 5, 40 hits per read: every
    // window; config 3, 8 hits per read: one window in three).
    const uint32_t T = (f_rank(SS >> 1) << kTagShift) | field;
    const uint64_t N0 = ~SS & f_below_nz(X);  // (a window holds a record)
    uint64_t D = N0 & (N0 << 1);
    uint64_t F;
    if (has_run_of(D, kHashWalk - 1u)) {
        hash_clear(tab, lane);
        bool overflow = false;  // (never: 64 keys at most)
        F = f_ballot(hash_first(tab, T, lane, f_bit(V), overflow));
    } else {
        const uint32_t x1 = f_shr1z(T) ^ T;
        uint32_t x = x1, differ = x1;
        // (the scalar unit is what a CU has one of: segments of up to six records -- four steps -- update D step by step,
        // longer ones go on eight steps at a time with D advanced by doubling: one scalar instruction per step, not two)
        if (D) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                x = f_shr1z(x) ^ x1;
                differ = min(differ, x);
                D &= D << 1;
            }
        }
        while (D) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                x = f_shr1z(x) ^ x1;
                differ = min(differ, x);
            }
            const uint64_t t2 = D & (D << 1), t4 = t2 & (t2 << 2), t8 = t4 & (t4 << 4);
            D = t8 & (D << 8);
        }
        F = f_ballot(differ != 0u) & V;
    }
    // heads: the first mapped lane of every segment; the last lane of a segment stops the carry of its start
    const uint64_t H = f_first_after(SS, V, (SS >> 1) | (1ull << (X - 1u)));
    // unique reads: a head whose NEXT target is a head again (or there is none in the window: the next run's first
    // target is a head).  In bit-reversed order "the target in front of g" is "the first target above g".
    const uint64_t Fr = __builtin_bitreverse64(F), Gr = __builtin_bitreverse64(F & ~H);
    const uint64_t U = H & ~__builtin_bitreverse64((~Fr + (Gr << 1)) & Fr);
    const bool uniq = f_bit(U);
    if (f_bit(F)) {
        // (the index in a vector register: the scalar unit is the kernel's bottleneck, and a scalar base per array costs
        // it two 64-bit adds and a shift per window)
        const uint32_t p = so.base + so.nf + f_rank(F);
        tgt_ref[p] = (field - 1u) | (f_bit(H) ? 0x80000000u : 0u);
        tgt_gbin[p] = gbin | (uniq ? 0x80000000u : 0u);
    }
    so.nf += static_cast<uint32_t>(__popcll(F));
    so.nh += static_cast<uint32_t>(__popcll(H));
    so.nv += static_cast<uint32_t>(__popcll(V));
    return WinMasks{F, H};
}

// ---------------------------------------------------------------------------------------------------------
// general path: mates interleave inside some run of the window.  Per-lane walks over the run by lane shuffles; the
// targets of a run are written ordered by (mate, file order), so a read's targets stay contiguous.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void window_general(const Staged& rec, uint32_t lane, uint64_t RS, uint64_t V, uint32_t X,
                                               WinOut& so, uint32_t* __restrict__ tgt_ref,
                                               uint32_t* __restrict__ tgt_gbin) {
    const bool in_pr = lane < X;
    const bool use = f_bit(V);
    const uint64_t le = (2ull << lane) - 1ull;
    const uint32_t run_from = 63u - static_cast<uint32_t>(__builtin_clzll((RS & le) | 1ull));
    const uint64_t above = RS & ~le;  // run starts behind this lane
    const uint32_t run_to = above ? static_cast<uint32_t>(__builtin_ctzll(above)) : X;  // one past my run's last lane
    const uint32_t W = use ? ((rec.mate << 28) | rec.ref) : (0xc0000000u | lane);  // mate 3: equals nobody's
    // backward over the run: first of its (mate, ref)?  head of its mate?
    const uint32_t back = in_pr ? lane - run_from : 0u;
    bool dup = false, mate_seen = false;
    for (uint32_t d = 1; f_ballot(back >= d) != 0ull; ++d) {
        const uint32_t o = __shfl(W, static_cast<int>((lane - d) & 63u), 64);
        const bool mine = back >= d;
        dup = dup || (mine && o == W);
        mate_seen = mate_seen || (mine && (o >> 28) == (W >> 28));
    }
    const bool first = use && !dup;
    const bool head = use && !mate_seen;
    // per target: targets of my run with a smaller mate (they go in front of me), targets of my own mate before me,
    // targets of my own mate at all (unique <=> 1)
    const uint32_t WF = first ? (W >> 28) : 3u;  // mate of a target, 3 for lanes that are none
    const uint32_t span = in_pr ? run_to - run_from : 0u;
    uint32_t smaller = 0, same_before = 0, same_total = 0;
    for (uint32_t d = 0; f_ballot(span > d) != 0ull; ++d) {
        const uint32_t j = run_from + d;  // lane d of my run
        const uint32_t o = __shfl(WF, static_cast<int>(j & 63u), 64);
        const bool mine = span > d && o != 3u && first;
        smaller += (mine && o < WF) ? 1u : 0u;
        same_total += (mine && o == WF) ? 1u : 0u;
        same_before += (mine && o == WF && j < lane) ? 1u : 0u;
    }
    const uint64_t F = f_ballot(first);
    const uint64_t H = f_ballot(head);
    if (first) {
        // targets of the runs before mine in this window, then my place inside my run
        const uint32_t before_run = static_cast<uint32_t>(__popcll(F & ((1ull << run_from) - 1ull)));
        const uint32_t p = so.base + so.nf + before_run + smaller + same_before;
        tgt_ref[p] = rec.ref | (head ? 0x80000000u : 0u);
        tgt_gbin[p] = rec.gbin | ((head && same_total == 1u) ? 0x80000000u : 0u);
    }
    so.nf += static_cast<uint32_t>(__popcll(F));
    so.nh += static_cast<uint32_t>(__popcll(H));
    so.nv += static_cast<uint32_t>(__popcll(V));
}

// ---------------------------------------------------------------------------------------------------------
// long path: the run starting at `pos` has 64 records or more.  Chunks of 64 records; a chunk's records are compared
// with the chunk's earlier lanes (shift) and with every earlier chunk of the run (64 rotations each).  Returns the index
// of the record behind the run.
// ---------------------------------------------------------------------------------------------------------
template <bool kChk, typename Acc>
__device__ __forceinline__ uint32_t long_run(const Acc& acc, uint32_t pos, uint32_t N, uint32_t lane, WinOut& so,
                                          uint32_t* __restrict__ tgt_ref, uint32_t* __restrict__ tgt_gbin, bool& bad,
                                          bool& collide, uint32_t* tab) {
    // 1. where the run ends, whether its mates ever decrease
    uint32_t klo0, khi0;
    acc.key_at(pos, klo0, khi0);
    uint32_t end = pos;
    bool decreasing = false;
    uint32_t last_mate = 0;
    while (true) {
        const uint32_t i = end + lane;
        const bool live = i < N;
        bool b = false;
        const bool mine = acc.same_run(live ? i : N - 1u, pos, klo0, khi0);
        const FrontRec r = acc.rec(live ? i : N - 1u, b);
        const uint64_t same = f_ballot(live && mine);
        if (kChk) collide = collide | (live && mine && acc.check[i] != acc.check[pos]);
        const uint32_t n_same = static_cast<uint32_t>(__builtin_ctzll(~same | (1ull << 63)));  // lanes before the first other key
        const bool whole = (~same) == 0ull;
        const uint32_t n_in = whole ? 64u : n_same;
        const uint32_t mprev = f_shr1(r.mate, last_mate);
        decreasing = decreasing || (f_ballot(r.mate < mprev) & f_below(n_in)) != 0ull;
        if (n_in) last_mate = __builtin_amdgcn_readlane(r.mate, n_in - 1u);
        end += n_in;
        if (!whole) break;
    }
    // 2. one pass for all mates when they never decrease (reads are contiguous), one pass per mate otherwise.
    // First of its (read, reference)?  Through the wave's hash table (the run's keys {mate, reference} stay in it from
    // chunk to chunk: constant work per chunk); a run with more distinct pairs than the table holds starts again with
    // the comparison walk -- every chunk against its own earlier lanes (shift) and all earlier chunks (64 rotations
    // each), quadratic in the run length but without any limit on it.
    const WinOut so_at_run = so;
    bool hashed = true;
    uint32_t nv_run;
    bool again;
    do {
        again = false;
        so = so_at_run;
        nv_run = 0;
        for (uint32_t pass = 0; pass < (decreasing ? 3u : 1u) && !again; ++pass) {
            uint32_t head_seen = 0;                        // bit m: a mapped record with mate m came by
            uint32_t n_first[3] = {0u, 0u, 0u};            // targets per mate
            uint32_t head_p[3] = {0u, 0u, 0u}, head_g[3] = {0u, 0u, 0u};  // where each mate's head went, its bin word
            if (hashed) hash_clear(tab, lane);
            for (uint32_t cb = pos; cb < end; cb += 64u) {
                const uint32_t i = cb + lane;
                const bool in_run = i < end;
                const FrontRec r = acc.rec(in_run ? i : end - 1u, bad);
                const bool use = in_run && r.mapped && (!decreasing || r.mate == pass);
                if (pass == 0) nv_run += static_cast<uint32_t>(__popcll(f_ballot(in_run && r.mapped)));
                bool first;
                if (hashed) {
                    bool overflow = false;
                    first = hash_first(tab, (r.mate << 28) | (r.ref + 1u), i - pos, use, overflow);
                    if (f_ballot(overflow) != 0ull) {  // more distinct (mate, reference) pairs than the table holds
                        hashed = false;
                        again = true;
                        break;
                    }
                } else {
                    const uint32_t T = use ? ((r.mate << 28) | r.ref) : (0xc0000000u | lane);
                    uint32_t differ = kNoMatch;
                    {   // earlier lanes of this chunk
                        uint32_t Ts = T;
                        for (uint32_t d = 1; d < 64u; ++d) {
                            Ts = f_shr1(Ts, kNoMatch);
                            differ = min(differ, Ts ^ T);
                        }
                    }
                    for (uint32_t eb = pos; eb < cb; eb += 64u) {  // every earlier chunk, all rotations
                        bool b = false;
                        const FrontRec e = acc.rec(eb + lane, b);
                        uint32_t Te = e.mapped ? ((e.mate << 28) | e.ref) : 0xe0000000u;
                        for (uint32_t d = 0; d < 64u; ++d) {
                            differ = min(differ, Te ^ T);
                            Te = f_ror1(Te);
                        }
                    }
                    first = use && differ != 0u;
                }
                const uint64_t F = f_ballot(first);
                // heads: per mate the first mapped lane, unless an earlier chunk had one
                uint64_t H = 0;
#pragma unroll
                for (uint32_t m = 0; m < 3u; ++m) {
                    const uint64_t Vm = f_ballot(use && r.mate == m);
                    if (Vm && !((head_seen >> m) & 1u)) {
                        H |= 1ull << __builtin_ctzll(Vm);
                        head_seen |= 1u << m;
                    }
                    n_first[m] += static_cast<uint32_t>(__popcll(f_ballot(first && r.mate == m)));
                }
                const bool head = f_bit(H);
                const uint32_t p = so.base + so.nf + f_rank(F);
                const FrontRaw3 rw{0u, first ? r.ref : 0u, 0u};
                const uint32_t g = first ? acc.gbin(r, acc.geo_of(rw)) : 0u;
                if (first) {
                    tgt_ref[p] = r.ref | (head ? 0x80000000u : 0u);
                    if (!head) tgt_gbin[p] = g;  // a head's bin word waits for the end of the run (unique or not)
                }
                uint64_t Hm = H;
                while (Hm) {
                    const uint32_t hl = static_cast<uint32_t>(__builtin_ctzll(Hm));
                    Hm &= Hm - 1ull;
                    const uint32_t m = __builtin_amdgcn_readlane(r.mate, hl);
                    head_p[m] = __builtin_amdgcn_readlane(p, hl);
                    head_g[m] = __builtin_amdgcn_readlane(g, hl);
                }
                so.nf += static_cast<uint32_t>(__popcll(F));
                so.nh += static_cast<uint32_t>(__popcll(H));
            }
            if (!again) {
#pragma unroll
                for (uint32_t m = 0; m < 3u; ++m)
                    if (((head_seen >> m) & 1u) && lane == 0u)
                        tgt_gbin[head_p[m]] = head_g[m] | (n_first[m] == 1u ? 0x80000000u : 0u);
            }
        }
    } while (again);
    so.nv += nv_run;
    return end;
}

constexpr uint32_t kStageRecs = 64u * kSlotBlocks;   // records staged per slot (the slot and 64 more)
constexpr uint32_t kStageGroup = 6;                  // blocks whose loads are in flight together

// first run start among the staged records at or behind `from`; kStageRecs when there is none
__device__ __forceinline__ uint32_t next_run_start(const uint32_t* st1, uint32_t lane, uint32_t from) {
    for (uint32_t o = from; o < kStageRecs; o += 64u) {
        const uint32_t i = o + lane;
        const uint32_t w = i < kStageRecs ? st1[i] : 0u;
        const uint64_t m = f_ballot(static_cast<int32_t>(w) < 0);
        if (m) return o + static_cast<uint32_t>(__builtin_ctzll(m));
    }
    return kStageRecs;
}

// ---------------------------------------------------------------------------------------------------------
// long path, staged: a run of 64 records or more that lies inside the slot's staged stretch, records [off, end_off) of
// st1 / st2 -- nothing is loaded from global memory again, the bins are the staged ones, and first-of-(read, reference)
// goes through the wave's hash table chunk after chunk (constant work per chunk).  Returns false when the run holds
// more distinct (mate, reference) pairs than the table does: the caller then takes long_run, which can fall back to
// the comparison walk.  `so` is unchanged in that case.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool long_run_staged(const uint32_t* st1, const uint32_t* st2, uint32_t off, uint32_t end_off,
                                                uint32_t lane, WinOut& so, uint32_t* __restrict__ tgt_ref,
                                                uint32_t* __restrict__ tgt_gbin, uint32_t* tab) {
    // 1. mapper output: the run's mates never decrease (all of mate 1, then all of mate 2), a read is a stretch of the run.
    // ONE pass that says so on the way; heads and the targets per mate by mask arithmetic on two ballots.
    const WinOut so_at_run = so;
    {
        uint32_t head_seen = 0, last_mate = 0, nv = 0;
        uint32_t n_first[3] = {0u, 0u, 0u}, head_p[3] = {0u, 0u, 0u}, head_g[3] = {0u, 0u, 0u};
        bool falls = false;
        hash_clear(tab, lane);
        for (uint32_t o = off; o < end_off; o += 64u) {
            const uint32_t i = o + lane;
            const uint32_t n_in = min(64u, end_off - o);
            const uint32_t at = min(i, end_off - 1u);
            const uint32_t w1 = st1[at], g = st2[at];
            const uint32_t field = w1 & kRefField, mate = (w1 >> kStMateShift) & 3u;
            const uint64_t IN = f_below_nz(n_in);
            if ((f_ballot(mate < f_shr1(mate, last_mate)) & IN) != 0ull) {
                falls = true;
                break;
            }
            last_mate = __builtin_amdgcn_readlane(mate, n_in - 1u);
            const uint64_t V = f_ballot(field != kRefField) & IN;
            bool overflow = false;
            const bool first = hash_first(tab, (mate << 28) | field, i - off, f_bit(V), overflow);
            if (f_ballot(overflow) != 0ull) {
                so = so_at_run;
                return false;
            }
            const uint64_t F = f_ballot(first);
            const uint64_t M1 = f_ballot(mate == 1u), M2 = f_ballot(mate == 2u);
            const uint64_t Mm[3] = {~(M1 | M2), M1, M2};
            uint64_t H = 0;
#pragma unroll
            for (uint32_t m = 0; m < 3u; ++m) {
                const uint64_t Vm = V & Mm[m];
                if (Vm && !((head_seen >> m) & 1u)) {
                    H |= Vm & (0ull - Vm);
                    head_seen |= 1u << m;
                }
                n_first[m] += static_cast<uint32_t>(__popcll(F & Mm[m]));
            }
            const bool head = f_bit(H);
            const uint32_t p = so.base + so.nf + f_rank(F);
            if (first) {
                tgt_ref[p] = (field - 1u) | (head ? 0x80000000u : 0u);
                if (!head) tgt_gbin[p] = g;  // a head's bin word waits for the end of the run (unique or not)
            }
            uint64_t Hm = H;
            while (Hm) {
                const uint32_t hl = static_cast<uint32_t>(__builtin_ctzll(Hm));
                Hm &= Hm - 1ull;
                const uint32_t m = __builtin_amdgcn_readlane(mate, hl);
                head_p[m] = __builtin_amdgcn_readlane(p, hl);
                head_g[m] = __builtin_amdgcn_readlane(g, hl);
            }
            nv += static_cast<uint32_t>(__popcll(V));
            so.nf += static_cast<uint32_t>(__popcll(F));
            so.nh += static_cast<uint32_t>(__popcll(H));
        }
        if (!falls) {
#pragma unroll
            for (uint32_t m = 0; m < 3u; ++m)
                if (((head_seen >> m) & 1u) && lane == 0u) tgt_gbin[head_p[m]] = head_g[m] | (n_first[m] == 1u ? 0x80000000u : 0u);
            so.nv += nv;
            return true;
        }
        so = so_at_run;
    }
    // 2. the mates interleave: one pass per mate number, so that a read's targets stay contiguous
    const bool decreasing = true;
    uint32_t nv_run = 0;
    for (uint32_t pass = 0; pass < 3u; ++pass) {
        uint32_t head_seen = 0;                        // bit m: a mapped record with mate m came by
        uint32_t n_first[3] = {0u, 0u, 0u};            // targets per mate
        uint32_t head_p[3] = {0u, 0u, 0u}, head_g[3] = {0u, 0u, 0u};  // where each mate's head went, its bin word
        hash_clear(tab, lane);
        for (uint32_t o = off; o < end_off; o += 64u) {
            const uint32_t i = o + lane;
            const bool in_run = i < end_off;
            const uint32_t w1 = st1[in_run ? i : end_off - 1u], g = st2[in_run ? i : end_off - 1u];
            const uint32_t field = w1 & kRefField, mate = (w1 >> kStMateShift) & 3u;
            const bool mapped = in_run && field != kRefField;
            const bool use = mapped && (!decreasing || mate == pass);
            if (pass == 0) nv_run += static_cast<uint32_t>(__popcll(f_ballot(mapped)));
            bool overflow = false;
            const bool first = hash_first(tab, (mate << 28) | field, i - off, use, overflow);
            if (f_ballot(overflow) != 0ull) {
                so = so_at_run;
                return false;
            }
            const uint64_t F = f_ballot(first);
            uint64_t H = 0;  // heads: per mate the first mapped lane, unless an earlier chunk had one
#pragma unroll
            for (uint32_t m = 0; m < 3u; ++m) {
                const uint64_t Vm = f_ballot(use && mate == m);
                if (Vm && !((head_seen >> m) & 1u)) {
                    H |= 1ull << __builtin_ctzll(Vm);
                    head_seen |= 1u << m;
                }
                n_first[m] += static_cast<uint32_t>(__popcll(f_ballot(first && mate == m)));
            }
            const bool head = f_bit(H);
            const uint32_t p = so.base + so.nf + f_rank(F);
            if (first) {
                tgt_ref[p] = (field - 1u) | (head ? 0x80000000u : 0u);
                if (!head) tgt_gbin[p] = g;  // a head's bin word waits for the end of the run (unique or not)
            }
            uint64_t Hm = H;
            while (Hm) {
                const uint32_t hl = static_cast<uint32_t>(__builtin_ctzll(Hm));
                Hm &= Hm - 1ull;
                const uint32_t m = __builtin_amdgcn_readlane(mate, hl);
                head_p[m] = __builtin_amdgcn_readlane(p, hl);
                head_g[m] = __builtin_amdgcn_readlane(g, hl);
            }
            so.nf += static_cast<uint32_t>(__popcll(F));
            so.nh += static_cast<uint32_t>(__popcll(H));
        }
#pragma unroll
        for (uint32_t m = 0; m < 3u; ++m)
            if (((head_seen >> m) & 1u) && lane == 0u) tgt_gbin[head_p[m]] = head_g[m] | (n_first[m] == 1u ? 0x80000000u : 0u);
    }
    so.nv += nv_run;
    return true;
}

// STAGE: records [B, B + kStageRecs) of the stream as two words each in the wave's stretch of LDS --
//   st1: reference + 1 (kRefField: not mapped) | mate << 26 | run start << 31,  st2: global bin.
// kClamp: the stream ends inside the stretch (its last slots only): lanes behind the end load the last record again and
// stage "not mapped, no run start".
// kChk: the records carry check words (a second hash of the read name): two adjacent records with one key and two
// check words are two names that collide in the key -> `collide`.
template <bool kClamp, bool kChk, typename Acc>
__device__ __forceinline__ void stage_slot(const Acc& acc, uint32_t B, uint32_t N, uint32_t lane, uint32_t* st1, uint32_t* st2,
                                           bool& bad, bool& collide) {
    uint32_t worst = 0;
    // The key in front of every record: the lane before's (DPP), for lane 0 the last key of the block before -- and for
    // the stretch's first block a second load of the keys, one record down, issued with the others.  (The key of record
    // B - 1 alone, loaded up front, would be a round trip of its own per slot.)
    uint32_t plo = 0, phi = 0, q0lo = 0, q0hi = 0, q0chk = 0, pchk = 0;
    if (!Acc::kMarked) {
        const uint32_t pb = B > 0u ? B - 1u : 0u;                        // B == 0: lane 0 is given no_key below,
        uint32_t rel = (B == 0u && lane != 0u) ? lane - 1u : lane;       // lane i gets key[i - 1]
        if (kClamp) rel = min(rel, N - 1u - pb);
        acc.load_key(pb, rel, q0lo, q0hi);
        if (kChk) q0chk = acc.load_check(pb, rel);
    }
#pragma unroll
    for (uint32_t j0 = 0; j0 < kSlotBlocks; j0 += kStageGroup) {
        FrontLoaded rec[kStageGroup];
        uint2 geo[kStageGroup];
        uint32_t chk[kStageGroup];
#pragma unroll
        for (uint32_t u = 0; u < kStageGroup; ++u) {
            if (j0 + u >= kSlotBlocks) break;
            // (the block's first record as the scalar base, the lane as the offset: the same three offset registers
            // serve every block of every slot)
            const uint32_t blk = kClamp ? min(B + 64u * (j0 + u), N - 1u) : B + 64u * (j0 + u);
            acc.load(blk, kClamp ? min(lane, N - 1u - blk) : lane, rec[u]);
            chk[u] = kChk ? acc.load_check(blk, kClamp ? min(lane, N - 1u - blk) : lane) : 0u;
        }
        __builtin_amdgcn_sched_barrier(0);  // (all the group's loads before the first use of one: left to itself the
                                            // scheduler pairs every load with its use, a round trip each)
#pragma unroll
        for (uint32_t u = 0; u < kStageGroup; ++u) {
            if (j0 + u >= kSlotBlocks) break;
            geo[u] = acc.geo_at(rec[u]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (uint32_t u = 0; u < kStageGroup; ++u) {
            const uint32_t j = j0 + u;
            if (j >= kSlotBlocks) break;
            uint32_t field, mate;
            acc.fields(rec[u], field, mate, worst);
            Acc::key_fix(rec[u].klo, rec[u].khi);
            const uint32_t g = acc.gbin_of(rec[u], geo[u]);
            const uint64_t k = (static_cast<uint64_t>(rec[u].khi) << 32) | rec[u].klo;
            uint32_t qlo, qhi, qchk;
            if (Acc::kMarked) {
                qlo = qhi = qchk = 0u;
            } else if (j == 0u) {
                qlo = q0lo;
                qhi = q0hi;
                qchk = q0chk;
                Acc::key_fix(qlo, qhi);
                if (B == 0u && lane == 0u) Acc::no_key(qlo, qhi);  // the first record of the stream starts a run
            } else {
                qlo = f_shr1(rec[u].klo, plo);
                qhi = f_shr1(rec[u].khi, phi);
                qchk = kChk ? f_shr1(chk[u], pchk) : 0u;
            }
            const uint64_t q = (static_cast<uint64_t>(qhi) << 32) | qlo;
            if (kChk) {
                // (no short-circuit operators on per-lane state here: the host emulator of tests/native identifies a wave
                // collective by its call site, and a compiler that clones the code behind `collide ||` gives the lanes
                // of one wave two sites for the same collective)
                bool here = (k == q) & (chk[u] != qchk);
                if (kClamp) here = here & (64u * j + lane < N - B);
                collide = collide | here;
                pchk = static_cast<uint32_t>(__builtin_amdgcn_readlane(chk[u], 63));
            }
            bool starts;
            if constexpr (Acc::kMarked)
                starts = Acc::starts_run(rec[u]) || (B == 0u && j == 0u && lane == 0u);  // (the stream's first record does)
            else
                starts = k != q;
            uint32_t w = field | (mate << kStMateShift) | (starts ? kStRunStart : 0u);
            if (kClamp && 64u * j + lane >= N - B) w = kRefField;
            st1[64u * j + lane] = w;
            st2[64u * j + lane] = g;
            if (!Acc::kMarked) {
                plo = static_cast<uint32_t>(__builtin_amdgcn_readlane(rec[u].klo, 63));
                phi = static_cast<uint32_t>(__builtin_amdgcn_readlane(rec[u].khi, 63));
            }
        }
    }
    bad = bad | acc.out_of_range(worst);
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------
// k_front
// ---------------------------------------------------------------------------------------------------------
template <typename Acc, bool kChk>
__global__ __launch_bounds__(kFrontBlock, 4) void k_front(const Acc acc, uint32_t nslots, uint32_t* __restrict__ counters,
                                                       uint32_t* __restrict__ tgt_ref, uint32_t* __restrict__ tgt_gbin,
                                                       uint4* __restrict__ slots) {
    __shared__ uint32_t s_stage[kFrontBlock / 64][2][kStageRecs];
    __shared__ __attribute__((aligned(16))) uint32_t s_tab[kFrontBlock / 64][2 * kHashSlots];  // hash_first's table, one per wave
    const uint32_t N = acc.count(counters);
    const uint32_t lane = f_lane();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t n_waves = gridDim.x * (kFrontBlock / 64);
    uint32_t* const st1 = s_stage[wave][0];
    uint32_t* const st2 = s_stage[wave][1];
    uint32_t* const tab = s_tab[wave];
    bool bad = false, collide = false;
    for (uint32_t slot = blockIdx.x * (kFrontBlock / 64) + wave; slot < nslots; slot += n_waves) {
        const uint32_t B = slot * kSlotRecs;
        WinOut so{B, 0u, 0u, 0u};
        FPROF_T(f0);
        if (B < N) {
            // ---- 1. stage (the only part that waits for memory; everything a group loads is in flight at once)
            if (N - B >= kStageRecs)
                stage_slot<false, kChk>(acc, B, N, lane, st1, st2, bad, collide);
            else
                stage_slot<true, kChk>(acc, B, N, lane, st1, st2, bad, collide);
            // ---- 2 + 3. windows: cut at the staged run starts, classified from the staged words.  (Every load has
            // landed by now; said aloud, so that the compiler does not make each window wait for the stores of the
            // window before it on behalf of a register some load of the staging loop once wrote.)
            __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
            FPROF_T(f1);
            FPROF_ADD(0, f0, f1);
            uint32_t off = next_run_start(st1, lane, 0u);
            so.base = B + min(off, kSlotRecs);
            while (off < kSlotRecs && B + off < N) {  // (the stream may end inside the slot)
                const uint32_t w1 = st1[off + lane], w2 = st2[off + lane];
                const uint64_t RSw = f_ballot(static_cast<int32_t>(w1) < 0);  // (bit 0 is set: a window starts a run)
                const uint32_t pos = B + off;
                // complete runs end at the last run start of the window -- or at the end of the stream
                uint32_t X = pos + 64u >= N ? N - pos : 63u - static_cast<uint32_t>(__builtin_clzll(RSw | 1ull));
                if (kSlotRecs - off < 64u) {  // runs starting at or behind the slot's end are the next slot's
                    const uint64_t beyond = RSw & ~f_below(kSlotRecs - off);
                    if (beyond) X = min(X, static_cast<uint32_t>(__builtin_ctzll(beyond)));
                }
                if (X) {
                    const uint32_t field = w1 & kRefField, mate = (w1 >> kStMateShift) & 3u;
                    const uint64_t PR = f_below_nz(X);
                    const uint64_t RS = RSw & PR;
                    const uint32_t mprev = f_shr1z(mate);
                    const uint64_t V = f_ballot(field != kRefField) & PR;
                    if ((f_ballot(mate < mprev) & ~RS & PR) == 0ull) {
                        (void)window_fast(field, w2, lane, (RS | f_ballot(mate != mprev)) & PR, V, X, so, tgt_ref, tgt_gbin, tab);
                    } else {  // (mates interleave: the window's targets are not in lane order)
                        window_general(Staged{mate, field - 1u, w2, f_bit(V)}, lane, RS, V, X, so, tgt_ref, tgt_gbin);
                    }
                    off += X;
                } else {
                    // a run of 64 records or more.  Inside the staged stretch (its end is the next staged run start): from
                    // the staged words; running on beyond it, or with more distinct references than the hash table
                    // holds: from global memory, at its own pace
                    FPROF_T(l0);
                    const uint32_t end_off = next_run_start(st1, lane, off + 64u);
                    if (end_off < kStageRecs && long_run_staged(st1, st2, off, end_off, lane, so, tgt_ref, tgt_gbin, tab)) {
                        off = end_off;
                    } else {
                        const uint32_t end = long_run<kChk>(acc, pos, N, lane, so, tgt_ref, tgt_gbin, bad, collide, tab);
                        __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): this path's loads stay out of the other windows' waits
                        off = end - B < kStageRecs ? next_run_start(st1, lane, end - B) : kStageRecs;
                    }
                    FPROF_T(l1);
                    FPROF_ADD(3, l0, l1);  // (runs of 64 records or more)
                }
            }
        }
        FPROF_T(f2);
        FPROF_ADD(1, f0, f2);
        FPROF_ADD(2, f2, f2 + 1ull);  // (slots)
        if (lane == 0u) slots[slot] = make_uint4(so.base, so.nf, so.nh, so.nv);
    }
    // (No totals here: thousands of waves adding to the same three counters are as many memory-side atomics in a row,
    // ~40 ns each -- 0.3 ms at the end of a 0.1 ms kernel.  The first consumer of the slots sums their counts.)
    if (__any(bad) && lane == 0u) atomicOr(&counters[CNT_ERR], static_cast<uint32_t>(ERR_REF_RANGE));
    if (kChk && __any(collide) && lane == 0u) atomicOr(&counters[CNT_ERR], static_cast<uint32_t>(ERR_KEY_COLLISION));
}

// ---------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------
uint32_t front_slots(uint32_t n_records) { return (n_records + kSlotRecs - 1u) / kSlotRecs; }

static uint32_t front_grid(uint32_t nslots) {
    // One slot per wave, however many workgroups that makes: the dispatcher hands the next workgroup to whichever CU
    // retires one.  (A grid capped at what is resident at once, its waves striding over the slots, gives some waves
    // one slot more than others -- at 2.4 slots per resident wave that is 3 rounds for the price of 2.4.)
    const uint32_t blocks = (nslots + (kFrontBlock / 64) - 1u) / (kFrontBlock / 64);
    return std::max(1u, blocks);
}

void launch_front_raw(hipStream_t st, const DeviceRecords& in, uint32_t n_refs, const uint2* geo, uint32_t half_read,
                      uint32_t bin_width, uint32_t* counters, uint32_t* tgt_ref, uint32_t* tgt_gbin, uint4* slots,
                      hipEvent_t t0, hipEvent_t t1) {
    const uint32_t ns = front_slots(in.n);
    if (!ns) return;
    FrontRawT<false> a;
    a.key = in.key;
    a.ref = in.ref;
    a.pos = in.pos;
    a.flag = in.flag;
    a.check = in.check;
    a.geo = geo;
    a.n = in.n;
    a.n_refs = n_refs;
    a.half_read = half_read;
    a.bin_width = bin_width;
    a.bw_magic = bin_width ? 0xffffffffu / bin_width : 0u;
    // (t0 / t1, when given: the dispatch's own start and end time stamps -- what rocprofv3 reports as the kernel's
    // duration; events recorded around the launch add the 4 - 6 us it takes a dependent dispatch to start)
    if (in.marked) {
        FrontMarked m;
        m.word = reinterpret_cast<const uint32_t*>(in.ref);
        m.pos = in.pos;
        m.geo = geo;
        m.n = in.n;
        m.n_refs = n_refs;
        m.half_read = half_read;
        m.bin_width = bin_width;
        m.bw_magic = a.bw_magic;
        hipExtLaunchKernelGGL((k_front<FrontMarked, false>), dim3(front_grid(ns)), dim3(kFrontBlock), 0, st, t0, t1, 0, m, ns,
                              counters, tgt_ref, tgt_gbin, slots);
    } else if (in.packed) {  // (same members, other accessors)
        FrontPacked b;
        b.key = a.key, b.ref = a.ref, b.pos = a.pos, b.flag = nullptr, b.check = a.check, b.geo = a.geo, b.n = a.n;
        b.n_refs = a.n_refs, b.half_read = a.half_read, b.bin_width = a.bin_width, b.bw_magic = a.bw_magic;
        if (in.check)
            hipExtLaunchKernelGGL((k_front<FrontPacked, true>), dim3(front_grid(ns)), dim3(kFrontBlock), 0, st, t0, t1, 0, b, ns,
                                  counters, tgt_ref, tgt_gbin, slots);
        else
            hipExtLaunchKernelGGL((k_front<FrontPacked, false>), dim3(front_grid(ns)), dim3(kFrontBlock), 0, st, t0, t1, 0, b, ns,
                                  counters, tgt_ref, tgt_gbin, slots);
    } else if (in.check)
        hipExtLaunchKernelGGL((k_front<FrontRaw, true>), dim3(front_grid(ns)), dim3(kFrontBlock), 0, st, t0, t1, 0, a, ns,
                              counters, tgt_ref, tgt_gbin, slots);
    else
        hipExtLaunchKernelGGL((k_front<FrontRaw, false>), dim3(front_grid(ns)), dim3(kFrontBlock), 0, st, t0, t1, 0, a, ns,
                              counters, tgt_ref, tgt_gbin, slots);
}

void launch_front_sorted(hipStream_t st, uint32_t n_upper, const uint64_t* ident, const uint2* pay,
                         uint32_t* counters, uint32_t* tgt_ref, uint32_t* tgt_gbin, uint4* slots,
                         const uint32_t* cchk) {
    const uint32_t ns = front_slots(n_upper);
    if (!ns) return;
    FrontSorted a{ident, pay, cchk};
    if (cchk)
        hipLaunchKernelGGL((k_front<FrontSorted, true>), dim3(front_grid(ns)), dim3(kFrontBlock), 0, st, a, ns, counters, tgt_ref,
                           tgt_gbin, slots);
    else
        hipLaunchKernelGGL((k_front<FrontSorted, false>), dim3(front_grid(ns)), dim3(kFrontBlock), 0, st, a, ns, counters,
                           tgt_ref, tgt_gbin, slots);
}

}  // namespace slimm

#if defined(EXP) && EXP == 11
extern "C" int slimm_debug_prof_front(unsigned long long* out, int n, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(slimm::g_prof_f), sizeof(unsigned long long) * n);
    if (reset) {
        static unsigned long long z[4 * 8192];
        (void)hipMemcpyToSymbol(HIP_SYMBOL(slimm::g_prof_f), z, sizeof(z));
    }
    return e == hipSuccess ? 0 : -1;
}
#endif
