// BGZF blocks inflated on the device in TWO phases (gfx950, wave64) -- the fast path in front of bgzf_inflate.hip.
//
// Where it stands on the path: the reference reads its input through seqan::BamFileIn (call sites src/misc.hpp:498-522,
// src/slimm.hpp:194-208), which inflates the BGZF blocks of a BAM file.  bgzf_inflate.hip does that with a LANE per block
// that decodes AND copies: every trip of a wave's symbol loop waits for the slowest lane's match copy (19 GB/s of inflated
// bytes on a BAM that compresses 3-fold, profiles/round5/00_realistic_cli_before.txt).  Here the two jobs are apart:
//
//   k_inflate_decode   a LANE per block, Huffman decoding only, in wave-uniform steps (no data-dependent branch inside a
//                      step: every lane decodes one literal/length symbol and, speculatively, the distance symbol behind
//                      it).  Canonical decoding without first-level tables: the next 15 stream bits, bit-reversed, are
//                      compared with the 15 left-justified code-length limits held in REGISTERS (a subtraction and an
//                      and-or per pair of lengths), which leaves LDS only the sorted symbols (9 + 5 bits) and one base per
//                      length: 420 B per lane, 27 KB per wave -> five of them per CU where the table-driven kernel fits two
//                      (what construction needs per length lives in the bases' bytes while a header is read; the
//                      code-length code stays in registers).
//                      TWO waves per 64 blocks: the DECODER wave does the above and hands one word per lane and step
//                      (literals / a match / the end) through a ring in LDS to the WRITER wave, which puts literals
//                      straight to their final place in the output (four at a time), turns matches into 4-byte TOKENS
//                      {literals since the last token, length, distance} and keeps the bounds -- what is left of the block
//                      afterwards are holes.  What bounds this kernel is the latency of a lane's chain of dependent
//                      instructions at one decoder wave per SIMD (LDS capacity); the writer's share of that chain runs on
//                      another SIMD now, the decoder never waits for a store and the writer never for a load.  The input
//                      comes through a 64-bit reservoir per lane that takes 8 unaligned bytes per step, asked for one step
//                      ahead.
//   k_inflate_resolve  a WORKGROUP of 512 threads per block, the block's 64 KB in LDS: the holes are filled in chunks of <= 4 KB
//                      of output.  A token leaves two 16-bit MARKERS in a table of the chunk's bytes (where its literals
//                      begin, where its match begins: the distance); every thread carries the markers forward over 8
//                      consecutive bytes (registers, a DPP scan over the wave, one word per wave through LDS) and turns them
//                      into a pointer per byte -- no loop over a token's bytes, whose longest the lanes of a wave would
//                      wait for.  Pointer JUMPING, two jumps per barrier, brings every pointer to a literal or to a byte in
//                      front of the chunk (a pointer carries a bit that says so: no round that only confirms; log4 of the
//                      longest chain of matches copying matches), one gather finishes the chunk.  No byte waits for another
//                      lane's copy loop.  Then the CRC32 of the gzip trailer (512 segments in parallel; crc32_combine is
//                      linear: every segment's CRC times x^(8 * the bytes behind it) from a table, XORed up) and the block
//                      written out in 16-byte stores.
//
// What the fast path does not do it hands to bgzf_inflate.hip's kernel block by block (info.flag): stored DEFLATE blocks,
// and ANY irregularity -- a bad code, a distance beyond the output, a CRC that does not match.  That kernel decides what is
// an error and which; a valid stream is never rejected here, a corrupt one never accepted.  status[2] counts the blocks
// that went that way.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "kernels.h"

namespace slimm {

namespace {

// ------------------------------------------------------------------------------------------------ phase 1: decode
// LDS bytes of a lane, element-interleaved over the 64 lanes: byte i of a table at offset T lies at [(T + i) * 64 + lane], a
// 16-bit element at byte [T * 64 + i * 128 + 2 lane], a word at [T * 64 + i * 256 + 4 lane] -- one shift-and-add per access
// (a layout interleaved by whole words kept every lane in its own bank and cost three more instructions per access: the
// kernel is bound by its vector instructions, not by the LDS).
constexpr uint32_t kLsymLo = 0;      // 288 B: low 8 bits of the 288 literal/length symbols in canonical order
constexpr uint32_t kLsymHi = 288;    //  36 B: their bit 8 (9 words)
constexpr uint32_t kDsym = 324;      //  32 B: the 30 distance symbols in canonical order
constexpr uint32_t kLbase = 356;     //  32 B: per code length, (symbols with shorter codes) - (first code of the length), 16 bits
constexpr uint32_t kDbase = 388;     //  32 B: the same for the distance code
// While a header is read the two base tables are not in use, and what construction needs per length -- counts, then the next
// free place -- lives in THEIR bytes; the bases follow from where the places have got to when every symbol is in
// (bases_from_ends).  The code-length code's 19 symbols and 7 bases stay in registers (ClCode).  420 B per lane instead of
// 520: 26.9 KB of tables + 4 KB of ring per workgroup, FIVE decoder waves per CU instead of four.
constexpr uint32_t kTmpB = kLbase;   //  counts of the distance code, then the literal/length code's next free places
constexpr uint32_t kTmpA = kDbase;   //  counts of the literal/length code, then the distance code's next free places
constexpr uint32_t kLaneBytes = 420;

// (the same LDS bytes are read and written as bytes, halves and words: types that may alias)
typedef uint16_t __attribute__((may_alias)) u16a;
typedef uint32_t __attribute__((may_alias)) u32a;
typedef int16_t __attribute__((may_alias)) i16a;
struct __attribute__((may_alias, aligned(8))) uint2a {
    uint32_t x, y;
};

struct Lds {
    uint8_t* p;      // the workgroup's table memory
    uint32_t lane;
    __device__ __forceinline__ uint8_t& b8(uint32_t base, uint32_t i) const { return p[(base + i) * 64u + lane]; }
    __device__ __forceinline__ u16a& b16(uint32_t base, uint32_t i) const { return reinterpret_cast<u16a*>(p)[base * 32u + i * 64u + lane]; }
    __device__ __forceinline__ u32a& b32(uint32_t base, uint32_t i) const { return reinterpret_cast<u32a*>(p)[base * 16u + i * 64u + lane]; }
};

__device__ __forceinline__ uint64_t ld64u(const uint8_t* p) {
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
__device__ __forceinline__ void st32u(uint8_t* p, uint32_t v) { __builtin_memcpy(p, &v, 4); }

// The input of a lane: up to 64 bits of the stream, low bits first; `rp` = the stream byte the next refill takes from.
// refill(the 8 bytes at rp): afterwards 56 .. 63 bits are there (a step takes at most 48).
struct Bits {
    uint64_t lo;
    uint32_t avail, rp;
    __device__ __forceinline__ void start(uint32_t at) {
        lo = 0;
        avail = 0;
        rp = at;
    }
    __device__ __forceinline__ void refill(uint64_t next8) {
        lo |= next8 << avail;   // (bits beyond the whole bytes counted below are the stream's own: the next refill repeats them)
        const uint32_t adv = (63u - avail) >> 3;
        rp += adv;
        avail += adv << 3;
    }
    __device__ __forceinline__ void drop(uint32_t c) {  // c <= avail
        lo >>= c;
        avail -= c;
    }
    // stream position, in bits, of the next unread bit
    __device__ __forceinline__ uint64_t at_bit() const { return static_cast<uint64_t>(rp) * 8u - avail; }
};

// 15 left-justified limits of a canonical code: a 15-bit pattern x (first stream bit on top) has a code of length
// 1 + #{l : x >= lim[l]}; 16 = no code.  Two limits to a register, each as lim + 0x7fff: the 32-bit difference with x in both
// halves never borrows across them, and bit 15 / 31 of it says x < lim.  A subtraction, a shift and an and-or per PAIR, a
// population count at the end -- no compare, no carry chain.  (The sixteenth slot is a limit nothing reaches.)
struct Limits {
    uint32_t v[8];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = 0xffffffffu;
    }
    __device__ __forceinline__ void set(uint32_t l, uint32_t limit) {   // l = 0 .. 14, a constant where it matters
        const uint32_t sh = (l & 1u) * 16u;
        v[l >> 1] = (v[l >> 1] & ~(0xffffu << sh)) | ((limit + 0x7fffu) << sh);
    }
    __device__ __forceinline__ uint32_t length_of(uint32_t x) const {
        const uint32_t xx = x | (x << 16);
        uint32_t acc = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) acc = (acc >> 1) | ((v[k] - xx) & 0x80008000u);
        return 17u - static_cast<uint32_t>(__builtin_popcount(acc));
    }
};

__device__ __forceinline__ uint32_t top15(uint32_t w) { return __builtin_bitreverse32(w) >> 17; }

// From the counts per code length (LDS, 16 bits each, tmp A): the limits, the base per length (-> `base_at`), the first
// free place per length (-> tmp B).  Returns the code's slack: 0 complete, > 0 incomplete, < 0 over-subscribed.
// `bases8`: the bases of a code of at most 7 bits (the code-length code: 19 symbols, codes below 128), one signed byte per length.
template <uint32_t kMaxLen>
__device__ int code_from_counts(const Lds& L, Limits& lim, uint64_t* bases8) {
    uint32_t code = 0, offs = 0;
    int left = 1;
    uint64_t b8 = 0;
#pragma unroll
    for (uint32_t l = 1; l <= 15u; ++l) {
        const uint32_t c = l <= kMaxLen ? L.b16(kTmpA, l) : 0u;
        left = (left << 1) - static_cast<int>(c);
        lim.set(l - 1, (code + c) << (15u - l));
        if (l <= kMaxLen) {
            if (kMaxLen <= 7u) b8 |= static_cast<uint64_t>((offs - code) & 0xffu) << (8u * l);
            L.b16(kTmpB, l) = static_cast<uint16_t>(offs);
        }
        offs += c;
        code = (code + c) << 1;
    }
    if (bases8) *bases8 = b8;
    return left;
}

// Every symbol of a code is in its place: slot l of `at` holds where the symbols of length l END (= where those of length
// l + 1 begin).  It becomes the length's base, (symbols with shorter codes) - (first code of the length).
__device__ __forceinline__ void bases_from_ends(const Lds& L, uint32_t at) {
    uint32_t code = 0, offs = 0;
#pragma unroll
    for (uint32_t l = 1; l <= 15u; ++l) {
        const uint32_t end = L.b16(at, l);
        const uint32_t c = end - offs;
        L.b16(at, l) = static_cast<uint16_t>(offs - code);
        offs = end;
        code = (code + c) << 1;
    }
}

__device__ __forceinline__ uint32_t symbol_at(const Lds& L, uint32_t base_at, uint32_t x, uint32_t len) {
    const uint32_t l = len > 15u ? 15u : len;
    return (L.b16(base_at, l) + (x >> (15u - l))) & 0xffffu;
}

// The code-length code of a dynamic block, in registers: limits, a signed byte of base per length (1 .. 7), the 19 symbols in
// canonical order at 5 bits each (twelve to a word).
struct ClCode {
    Limits lim;
    uint64_t bases8, sym_a, sym_b;
    __device__ __forceinline__ void put(uint32_t place, uint32_t s) {   // place < 19
        if (place < 12u) sym_a |= static_cast<uint64_t>(s) << (5u * place);
        else sym_b |= static_cast<uint64_t>(s) << (5u * (place - 12u));
    }
    __device__ __forceinline__ uint32_t index_of(uint32_t x, uint32_t len) const {   // len = 1 .. 7
        const uint32_t base = static_cast<uint32_t>(static_cast<int32_t>(static_cast<int8_t>(bases8 >> (8u * len))));
        return (base + (x >> (15u - len))) & 0xffffu;
    }
    __device__ __forceinline__ uint32_t symbol(uint32_t idx) const {   // idx < 19
        return static_cast<uint32_t>(idx < 12u ? sym_a >> (5u * idx) : sym_b >> (5u * (idx - 12u))) & 31u;
    }
};

enum : uint32_t { kModeHeader = 0, kModeDecode = 1, kModeDone = 2, kModeHandOver = 3 };

#if defined(EXP) && EXP == 12  // cycle split of k_inflate_decode's steps: lane 0 of every wave (scripts/tprof_decode.py)
__device__ unsigned long long g_prof_d[8 * 1024];
#define DPROF_T(x) const unsigned long long x = __builtin_readcyclecounter()
#define DPROF_WAIT_VM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define DPROF_WAIT_ALL() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#define DPROF_ADD(slot, a, b) if (threadIdx.x == 0) g_prof_d[(blockIdx.x & 1023u) * 8 + slot] += (b) - (a)
#else
#define DPROF_T(x)
#define DPROF_WAIT_VM()
#define DPROF_WAIT_ALL()
#define DPROF_ADD(slot, a, b)
#endif

// One lane's view of its block while headers are read: plain bit taking with the load waited for on the spot (a header is
// a few hundred bits per ~16 K symbols).
struct HeaderBits {
    Bits& b;
    const uint8_t* in;
    __device__ __forceinline__ uint32_t take(uint32_t n) {  // n <= 16
        if (b.avail < 32u) b.refill(ld64u(in + b.rp));
        const uint32_t v = static_cast<uint32_t>(b.lo) & ((1u << n) - 1u);
        b.drop(n);
        return v;
    }
    __device__ __forceinline__ uint32_t peek15() {
        if (b.avail < 32u) b.refill(ld64u(in + b.rp));
        return top15(static_cast<uint32_t>(b.lo));
    }
};

// The code lengths of a dynamic block as runs {length value, repeat}: decoded from the stream with the code-length code
// (`cl`: ClCode).  prev = the length before (for symbol 16).  Returns false for a bad code.
__device__ __forceinline__ bool next_run(HeaderBits& hb, const ClCode& cl, uint32_t& prev, uint32_t& val, uint32_t& rep, bool first) {
    const uint32_t x = hb.peek15();
    const uint32_t len = cl.lim.length_of(x);
    if (len > 7u) return false;
    const uint32_t idx = cl.index_of(x, len);
    if (idx >= 19u) return false;
    const uint32_t sym = cl.symbol(idx);
    hb.b.drop(len);
    if (sym < 16u) {
        val = sym;
        rep = 1;
        prev = sym;
    } else if (sym == 16u) {
        if (first) return false;
        val = prev;
        rep = 3u + hb.take(2);
    } else if (sym == 17u) {
        val = 0;
        rep = 3u + hb.take(3);
        prev = 0;
    } else {
        val = 0;
        rep = 11u + hb.take(7);
        prev = 0;
    }
    return true;
}

__device__ const uint8_t kClOrderD[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// The header of the next DEFLATE block of a lane and the two codes behind it.  Returns the lane's new mode.
// (Executed by the lanes that wait for a header, the others masked: trip counts differ by lane, nothing else.)
// The code lengths of a dynamic block are decoded TWICE from the stream -- once for the counts per length, once to put the
// symbols in their places -- so no array of 316 lengths exists anywhere.
__device__ uint32_t read_header(Bits& bits, const uint8_t* in, uint32_t csize, const Lds& L, Limits& LL, Limits& DL, uint32_t& last) {
    HeaderBits hb{bits, in};
    last = hb.take(1);
    const uint32_t type = hb.take(2);
    if (type != 1u && type != 2u) return kModeHandOver;  // stored blocks (and type 3) are the other kernel's
    uint32_t nlen = 288, ndist = 30;
    ClCode cl;
    cl.lim.clear();
    cl.bases8 = cl.sym_a = cl.sym_b = 0;
    for (uint32_t l = 0; l < 16u; ++l) L.b16(kTmpA, l) = 0;   // (a table is only ever used through ONE element size: the lanes' elements interleave by it)
    if (type == 2u) {
        nlen = hb.take(5) + 257u;
        ndist = hb.take(5) + 1u;
        const uint32_t ncode = hb.take(4) + 4u;
        if (nlen > 286u || ndist > 30u) return kModeHandOver;
        // the code-length code: 19 lengths of 3 bits -> counts, limits, bases, sorted symbols
        uint32_t cll[3] = {0, 0, 0};
        for (uint32_t i = 0; i < ncode; ++i) {
            const uint32_t v = hb.take(3);
            const uint32_t s = kClOrderD[i];
            cll[s >> 3] |= v << (3u * (s & 7u));
            if (v) ++L.b16(kTmpA, v);
        }
        uint32_t n_cl = 0;
        for (uint32_t l = 1; l <= 7u; ++l) n_cl += L.b16(kTmpA, l);
        const int slack = code_from_counts<7>(L, cl.lim, &cl.bases8);
        if (slack < 0 || (slack > 0 && n_cl != 1u) || n_cl == 0u) return kModeHandOver;
        for (uint32_t s = 0; s < 19u; ++s) {
            const uint32_t v = (cll[s >> 3] >> (3u * (s & 7u))) & 7u;
            if (v) cl.put(L.b16(kTmpB, v)++, s);
        }
    }
    // pass 1: how many codes of each length -- literal/length code in tmp A, distance code in tmp B
    const Bits saved = bits;
    for (uint32_t l = 0; l < 16u; ++l) L.b16(kTmpA, l) = 0;
    for (uint32_t l = 0; l < 16u; ++l) L.b16(kTmpB, l) = 0;
    if (type == 1u) {
        L.b16(kTmpA, 7) = 24;
        L.b16(kTmpA, 8) = 152;
        L.b16(kTmpA, 9) = 112;
        L.b16(kTmpB, 5) = 30;
    } else {
        uint32_t index = 0, prev = 0;
        while (index < nlen + ndist) {
            uint32_t val, rep;
            if (!next_run(hb, cl, prev, val, rep, index == 0u)) return kModeHandOver;
            if (index + rep > nlen + ndist) return kModeHandOver;
            if (val) {
                const uint32_t in_l = index < nlen ? min(rep, nlen - index) : 0u;
                L.b16(kTmpA, val) = static_cast<uint16_t>(L.b16(kTmpA, val) + in_l);
                L.b16(kTmpB, val) = static_cast<uint16_t>(L.b16(kTmpB, val) + (rep - in_l));
            }
            index += rep;
        }
        if (bits.at_bit() > static_cast<uint64_t>(csize) * 8u) return kModeHandOver;
    }
    // the distance counts out of the way, the literal/length code from tmp A (its places -> tmp B), then the distance code
    // from its counts (back in tmp A; its places take their place)
    uint32_t dcnt[16];
#pragma unroll
    for (uint32_t l = 0; l < 16u; ++l) dcnt[l] = L.b16(kTmpB, l);
    uint32_t n_l = 0;
    for (uint32_t l = 1; l <= 15u; ++l) n_l += L.b16(kTmpA, l);
    const int lslack = code_from_counts<15>(L, LL, nullptr);
    if (lslack < 0 || (lslack > 0 && n_l != 1u) || n_l == 0u) return kModeHandOver;
#pragma unroll
    for (uint32_t l = 0; l < 16u; ++l) L.b16(kTmpA, l) = static_cast<uint16_t>(dcnt[l]);
    {
        uint32_t code = 0, offs = 0, n_d = 0;
        int left = 1;
        uint32_t places[16];
#pragma unroll
        for (uint32_t l = 1; l <= 15u; ++l) {
            const uint32_t c = L.b16(kTmpA, l);
            n_d += c;
            left = (left << 1) - static_cast<int>(c);
            DL.set(l - 1, (code + c) << (15u - l));
            places[l] = offs;
            offs += c;
            code = (code + c) << 1;
        }
#pragma unroll
        for (uint32_t l = 1; l <= 15u; ++l) L.b16(kTmpA, l) = static_cast<uint16_t>(places[l]);
        // (a distance code may be incomplete with one code, or absent altogether in a block of literals only; the fixed
        // code has 30 of its 32 codes)
        if (type == 2u && (left < 0 || (left > 0 && n_d > 1u))) return kModeHandOver;
    }
    // pass 2: the symbols into their places
    for (uint32_t i = 0; i < 9u; ++i) L.b32(kLsymHi, i) = 0;
    if (type == 1u) {
        for (uint32_t s = 0; s < 288u; ++s) {
            const uint32_t l = s < 144u ? 8u : (s < 256u ? 9u : (s < 280u ? 7u : 8u));
            const uint32_t at = L.b16(kTmpB, l)++;
            L.b8(kLsymLo, at) = static_cast<uint8_t>(s);
            if (s >= 256u) L.b32(kLsymHi, at >> 5) |= 1u << (at & 31u);
        }
        for (uint32_t s = 0; s < 30u; ++s) L.b8(kDsym, L.b16(kTmpA, 5)++) = static_cast<uint8_t>(s);
    } else {
        const Bits end1 = bits;
        bits = saved;
        uint32_t index = 0, prev = 0;
        while (index < nlen + ndist) {
            uint32_t val = 0, rep = 1;
            if (!next_run(hb, cl, prev, val, rep, index == 0u)) return kModeHandOver;  // (cannot happen: pass 1 took these runs)
            if (val) {
                for (uint32_t k = 0; k < rep; ++k) {
                    const uint32_t s = index + k;
                    if (s < nlen) {
                        const uint32_t at = L.b16(kTmpB, val)++;
                        L.b8(kLsymLo, at) = static_cast<uint8_t>(s);
                        if (s >= 256u) L.b32(kLsymHi, at >> 5) |= 1u << (at & 31u);
                    } else {
                        L.b8(kDsym, L.b16(kTmpA, val)++) = static_cast<uint8_t>(s - nlen);
                    }
                }
            }
            index += rep;
        }
        (void)end1;
    }
    // the places have run to the ends of their lengths: the bases take their bytes
    bases_from_ends(L, kLbase);
    bases_from_ends(L, kDbase);
    return kModeDecode;
}

}  // namespace

// What the decoder wave hands the writer wave per lane and step: one word.
//   kind (bits 30-31): 0 nothing, 1 literals {bits 0-7, 8-15: the bytes, bit 16: two of them}, 2 a match {bits 0-7: length - 3,
//   bits 8-22: distance - 1}, 3 the lane's stream ends here or went wrong (the decoder's own verdict travels in s_final).
constexpr uint32_t kRing = 16;    // steps of records between the two waves (a burst is up to kBurst of them)
constexpr uint32_t kBurst = 8;
constexpr int kWriterNap = 16, kDecoderNap = 4;   // s_sleep units of 64 cycles between two looks at the other wave's counter

// The hand-over words between the two waves: an ACQUIRE load / a RELEASE store at workgroup scope, so that the ring's plain
// stores stay in front of the counter that announces them and its plain loads behind the counter that admits them (ADVICE
// round 5: `volatile` orders only against other volatile accesses; on gfx950 the fences cost an s_waitcnt lgkmcnt).
__device__ __forceinline__ uint32_t lds_now(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_set(uint32_t* p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// TWO waves per 64 blocks.  Wave 0, the DECODER: headers, tables, the symbol steps -- and nothing of what becomes of a symbol.
// Wave 1, the WRITER: the literals gathered four to a store, the match tokens, the bounds of the output, every store.  A
// lane's step had been both in one chain of ~410 dependent instructions at one wave per SIMD (what bounds this kernel is
// the latency of that chain: CHANGELOG, round 5); apart they are ~250 and ~160 on two SIMDs side by side, the decoder
// never waits for a store and the writer never for a load.  Between them a ring of one word per lane and step in LDS,
// handed over burst by burst through two counters (wave-uniform: every lane takes a step's slot, "nothing" included).
__global__ __launch_bounds__(128) void k_inflate_decode(const uint8_t* __restrict__ comp, const BgzfBlock* __restrict__ blocks, uint32_t n_blocks,
                                                        uint8_t* __restrict__ out, uint32_t* __restrict__ tok, InflateInfo* __restrict__ info) {
    __shared__ uint32_t s_lds[kLaneBytes * 16u];
    __shared__ uint32_t s_ring[kRing * 64u];
    __shared__ uint32_t s_sync[4];     // [0] steps the decoder has handed over, [1] steps the writer has taken, [2] the decoder is through
    __shared__ uint32_t s_stop[2];     // lanes the writer found wrong: the decoder stops decoding them
    __shared__ uint32_t s_final[64];   // the decoder's verdict per lane: 1 = its stream ended where it must
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t b = blockIdx.x * 64u + lane;
    const bool have = b < n_blocks;
    BgzfBlock d;
    d.src = d.dst = 0;
    d.csize = d.isize = 0;
    d.tok = 0;
    if (have) d = blocks[b];
    const uint32_t csize = d.csize, isize = d.isize;
    if (threadIdx.x < 4u) s_sync[threadIdx.x] = 0;
    if (threadIdx.x < 2u) s_stop[threadIdx.x] = 0;
    __syncthreads();

    if (wave == 0) {
        // ------------------------------------------------------------------------------------------------ the decoder
        Lds L{reinterpret_cast<uint8_t*>(s_lds), lane};
        const uint8_t* in = comp + d.src;
        Bits bits;
        bits.start(0);
        Limits LL, DL;
        LL.clear();
        DL.clear();
        uint32_t mode = have ? kModeHeader : kModeDone;
        uint32_t last = 0, st = 0;   // st: steps handed over so far (wave-uniform)
        uint64_t ahead = 0;          // the 8 input bytes the lane's next refill takes: asked for a step ahead, ACROSS bursts too
        bool have_ahead = false;
        DPROF_T(d_start);
        for (;;) {
            // lanes the writer has given up: no use decoding them on
            {
                const uint64_t stop = static_cast<uint64_t>(lds_now(&s_stop[0])) | (static_cast<uint64_t>(lds_now(&s_stop[1])) << 32);
                if ((stop >> lane) & 1u) {
                    if (mode == kModeHeader || mode == kModeDecode) mode = kModeHandOver;
                }
            }
            const uint64_t want = __ballot(mode == kModeHeader);
            const uint64_t going = __ballot(mode == kModeDecode);
            if (!want && !going) break;
            if (want && (!going || __popcll(want) >= 8)) {
                DPROF_T(h0);
                if (mode == kModeHeader) {
                    mode = read_header(bits, in, csize, L, LL, DL, last);
                    have_ahead = false;
                }
                DPROF_T(h1);
                DPROF_ADD(5, h0, h1);
                continue;
            }
            // room for a burst: the writer has taken all but the last kRing - kBurst steps
            while (static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(lds_now(&s_sync[1]))) + (kRing - kBurst) < st) __builtin_amdgcn_s_sleep(kDecoderNap);
            // a burst of uniform steps
            if (mode == kModeDecode && !have_ahead) ahead = ld64u(in + bits.rp);   // (the first step behind a header)
            for (uint32_t it = 0; it < kBurst; ++it) {
                const bool run = mode == kModeDecode;
                DPROF_T(s0);
                DPROF_WAIT_VM();
                DPROF_T(s1);
                DPROF_ADD(0, s0, s1);
                // the input asked for a step ago
                if (run) {
                    bits.refill(ahead);
                    ahead = ld64u(in + bits.rp);
                    have_ahead = true;
                }
                if (!__any(run)) break;
                // ---- one literal/length symbol, then ONE more symbol of whichever code comes next: behind a literal the
                // literal/length code again (a second literal is taken along; anything else waits for the next step), behind a
                // length the distance code.  The second chain's limits, base table and symbol table are selected per lane.
                const uint32_t w1 = static_cast<uint32_t>(bits.lo);
                const uint32_t x = top15(w1);
                const uint32_t len = LL.length_of(x);
                const uint32_t idx = min(symbol_at(L, kLbase, x, len), 287u);
                const uint32_t sym = L.b8(kLsymLo, idx) | (((L.b32(kLsymHi, idx >> 5) >> (idx & 31u)) & 1u) << 8);
                bool bad = len > 15u;
                uint32_t c = len;
                DPROF_WAIT_ALL();
                DPROF_T(s2);
                DPROF_ADD(1, s1, s2);
                const bool is_lit = sym < 256u, is_eob = sym == 256u, is_len = sym > 256u;
                const uint32_t ls = is_len ? sym - 257u : 0u;
                bad = bad | (ls > 28u);
                const uint32_t le = (ls >= 8u && ls < 28u) ? (ls >> 2) - 1u : 0u;
                const uint32_t lb = ls < 8u ? ls + 3u : (ls >= 28u ? 258u : 3u + ((4u + (ls & 3u)) << le));
                const uint32_t mlen = lb + ((w1 >> (len > 15u ? 15u : len)) & ((1u << le) - 1u));
                // (32 bits behind the first symbol and its extra bits: 15 for the second code + 13 extra bits of a distance)
                const uint32_t w2 = static_cast<uint32_t>(bits.lo >> ((len > 15u ? 15u : len) + le));   // (le = 0 behind a literal)
                Limits SL;
#pragma unroll
                for (int k = 0; k < 8; ++k) SL.v[k] = is_lit ? LL.v[k] : DL.v[k];
                const uint32_t y = top15(w2);
                const uint32_t l2 = SL.length_of(y);
                const uint32_t i2 = min(symbol_at(L, is_lit ? kLbase : kDbase, y, l2), is_lit ? 287u : 31u);
                const uint32_t s2lo = L.b8(is_lit ? kLsymLo : kDsym, i2);
                const uint32_t s2hi = (L.b32(kLsymHi, (i2 >> 5) & 15u) >> (i2 & 31u)) & 1u;
                DPROF_WAIT_ALL();
                DPROF_T(s3);
                DPROF_ADD(2, s2, s3);
                // behind a literal: a second literal?
                const bool lit2 = is_lit & (l2 <= 15u) & (s2hi == 0u);
                // behind a length: the distance
                const uint32_t ds = s2lo;
                const uint32_t de = ds >= 4u ? (ds >> 1) - 1u : 0u;
                const uint32_t dbv = ds < 4u ? ds + 1u : 1u + ((2u + (ds & 1u)) << de);
                const uint32_t dist = dbv + ((w2 >> (l2 > 15u ? 15u : l2)) & ((1u << de) - 1u));
                if (is_len) {
                    bad = bad | (l2 > 15u) | (ds > 29u);
                    c += le + (l2 > 15u ? 0u : l2) + de;
                }
                if (lit2) c += l2;
                uint32_t rec = 0;
                if (run) {
                    bits.drop(c);
                    if (bits.at_bit() > static_cast<uint64_t>(csize) * 8u) bad = true;
                    if (bad) {
                        mode = kModeHandOver;
                        rec = 3u << 30;
                    } else if (is_eob) {
                        mode = last ? kModeDone : kModeHeader;
                        rec = last ? 3u << 30 : 0u;   // (between two DEFLATE blocks of a BGZF block nothing happens to the output)
                    } else if (is_lit) {
                        rec = (1u << 30) | sym | (lit2 ? (s2lo << 8) | (1u << 16) : 0u);
                    } else {
                        rec = (2u << 30) | (mlen - 3u) | ((dist - 1u) << 8);
                    }
                }
                s_ring[(st % kRing) * 64u + lane] = rec;
                ++st;
                DPROF_T(s4);
                DPROF_ADD(3, s3, s4);
                DPROF_ADD(4, s4, s4 + 1);
                DPROF_ADD(7, s4, s4 + ((run & lit2) ? 1u : 0u));   // (lane 0's literal pairs)
            }
            if (lane == 0) lds_set(&s_sync[0], st);   // (behind the burst's records: LDS takes a wave's operations in order)
        }
        DPROF_T(d_end);
        DPROF_ADD(6, d_start, d_end);
        // the stream must end inside the payload (that it ends at ISIZE bytes is the writer's to say)
        s_final[lane] = (mode == kModeDone && (bits.at_bit() + 7u) / 8u <= csize) ? 1u : 0u;
        if (lane == 0) lds_set(&s_sync[2], 1u);
    } else {
        // ------------------------------------------------------------------------------------------------ the writer
        uint8_t* o_base = out + d.dst;
        uint32_t* t_base = tok + d.tok;
        const uint32_t tok_room = bgzf_token_room(isize);
        uint32_t o = 0, acc_n = 0, litrun = 0, ntok = 0, wst = 0;
        uint64_t acc = 0;   // literals waiting to be stored: up to 3 from the steps before + 2 of this one
        bool bad = false, ended = !have;
        for (;;) {
            const uint32_t through = __builtin_amdgcn_readfirstlane(lds_now(&s_sync[2]));
            const uint32_t upto = __builtin_amdgcn_readfirstlane(lds_now(&s_sync[0]));
            if (upto == wst) {
                if (through) break;   // (read before the count: nothing more comes)
                __builtin_amdgcn_s_sleep(kWriterNap);   // (a burst takes the decoder ~10 us: a look every ~0.5 us, not every 30 ns on its SIMD)
                continue;
            }
            uint32_t rec_next = s_ring[(wst % kRing) * 64u + lane];
            for (; wst != upto; ++wst) {
                const uint32_t rec = rec_next;
                rec_next = s_ring[((wst + 1u) % kRing) * 64u + lane];   // (behind the last one: a slot not yet written, not used)
                const uint32_t kind = rec >> 30;
                const bool is_lit = kind == 1u, is_len = kind == 2u;
                if (kind == 0u || ended || bad) continue;
                bool flush = false;
                if (is_lit) {
                    const uint32_t n_lit = 1u + ((rec >> 16) & 1u);
                    if (o + n_lit > isize) bad = true;
                    acc |= static_cast<uint64_t>(rec & 0xffffu) << (8u * acc_n);
                    acc_n += n_lit;
                    o += n_lit;
                    litrun += n_lit;
                    flush = acc_n >= 4u;
                } else {
                    flush = acc_n != 0u;
                }
                if (flush & !bad) {   // (four bytes when there are four, else what there is: a match or the block's end follows)
                    const uint32_t n_out = acc_n >= 4u ? 4u : acc_n;
                    const uint32_t at = o - acc_n, v = static_cast<uint32_t>(acc);
                    if (at + 4u <= isize) {
                        st32u(o_base + at, v);
                    } else {
                        for (uint32_t k = 0; k < n_out; ++k) o_base[at + k] = static_cast<uint8_t>(v >> (8u * k));
                    }
                    acc = n_out == 4u ? acc >> 32 : 0ull;
                    acc_n -= n_out;
                }
                if (is_len & !bad) {
                    const uint32_t mlen = (rec & 0xffu) + 3u, dist = ((rec >> 8) & 0x7fffu) + 1u;
                    if (dist > o || o + mlen > isize || ntok + 2u > tok_room) {
                        bad = true;
                    } else {
                        if (litrun > 255u) {
                            t_base[ntok] = 0x80000000u | litrun;
                            t_base[ntok + 1u] = ((mlen - 3u) << 15) | (dist - 1u);
                            ntok += 2u;
                        } else {
                            t_base[ntok] = (litrun << 23) | ((mlen - 3u) << 15) | (dist - 1u);
                            ntok += 1u;
                        }
                        litrun = 0;
                        o += mlen;
                    }
                }
                if (kind == 3u) ended = true;
            }
            if (lane == 0) lds_set(&s_sync[1], wst);
            {   // the lanes that went wrong here: the decoder need not go on with them
                const uint64_t wrong = __ballot(bad);
                if (lane == 0 && wrong) {
                    atomicOr(&s_stop[0], static_cast<uint32_t>(wrong));
                    atomicOr(&s_stop[1], static_cast<uint32_t>(wrong >> 32));
                }
            }
        }
        // (between the decoder's verdicts and this wave's reading them: the barrier below)
        s_ring[lane] = bad ? 1u : 0u;   // (the ring is done with)
        s_ring[64u + lane] = o;
        s_ring[128u + lane] = ntok;
    }
    __syncthreads();
    if (wave == 1 && have) {
        // the stream must have ended, inside the payload, at exactly ISIZE bytes
        const bool ok = s_final[lane] != 0u && s_ring[lane] == 0u && s_ring[64u + lane] == isize;
        InflateInfo r;
        r.n_tok = s_ring[128u + lane];
        r.flag = ok ? 0u : 1u;
        info[b] = r;
    }
}

// ------------------------------------------------------------------------------------------------ phase 2: resolve
namespace {

#if defined(EXP) && EXP == 11  // cycle split of k_inflate_resolve: thread 0 of every workgroup (scripts/tprof_resolve.py)
__device__ unsigned long long g_prof_r[16 * 1024];
#define RPROF_T(x) const unsigned long long x = __builtin_readcyclecounter()
#define RPROF_ADD(slot, a, b) if (threadIdx.x == 0) g_prof_r[(blockIdx.x & 1023u) * 16 + slot] += (b) - (a)
#define RPROF_INC(slot, n) if (threadIdx.x == 0) g_prof_r[(blockIdx.x & 1023u) * 16 + slot] += (n)
#else
#define RPROF_T(x)
#define RPROF_ADD(slot, a, b)
#define RPROF_INC(slot, n)
#endif
constexpr uint32_t kRThreads = 512;                  // threads of a resolve workgroup = the tokens a chunk can take
constexpr uint32_t kRWaves = kRThreads / 64;
constexpr uint32_t kPer = 8;                          // output bytes of a chunk per thread (a multiple of 4)
constexpr uint32_t kJumps = 2;                        // pointer jumps per round and barrier
constexpr uint32_t kChunk = kRThreads * kPer;         // output bytes of a chunk (a 16-bit pointer each); 64 KB + 8 KB: two workgroups per CU
constexpr uint32_t kRLog2 = 9;
constexpr uint32_t kCrcRow = kRThreads;               // words of the block a CRC row covers: one per thread
constexpr uint32_t kWinLoads = 65536 / 16 / kRThreads;
// A pointer of the chunk as 16 bits, told apart by RANGE (one compare): [0, kChunk) a byte of the chunk that is itself copied
// from somewhere (the pointer still moves); [kChunk, 2 kChunk) = kChunk + a LITERAL of the chunk (final; where it points the
// table holds the same value: a literal's own pointer); [0x8000, 0xffff] negative, a byte in front of the chunk (final).
constexpr uint32_t kLit = kChunk;
constexpr uint32_t kLitMarker = (1u - kLit) & 0xffffu;   // markers: 0 none, dist + 1 a match from here on, this a run of literals
static_assert(2u * kChunk <= 0x8000u && (kChunk & (kChunk - 1u)) == 0u && kPer % 4 == 0 && (1u << kRLog2) == kRThreads, "");

// inclusive scans over the 64 lanes of a wave with DPP moves (no LDS round trips as with ds_bpermute shuffles)
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t v) {
    v += __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0u, v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0u, v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0u, v, 0x118, 0xf, 0xf, true);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0u, v, 0x142, 0xa, 0xf, true);   // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0u, v, 0x143, 0xc, 0xf, true);   // row_bcast:31 into rows 2 and 3
    return v;
}
// "the nearest non-zero value at or in front of the lane" (0: none).  (The choice as arithmetic, not as `v ? v : u`: the host
// emulator's compiler clones the next move behind such a branch, and the lanes of a wave no longer meet at one call site.)
__device__ __forceinline__ uint32_t keep_or(uint32_t v, uint32_t u) { return v | (u & (0u - static_cast<uint32_t>(v == 0u))); }
__device__ __forceinline__ uint32_t wave_scan_last(uint32_t v) {
    v = keep_or(v, __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xf, 0xf, true));
    v = keep_or(v, __builtin_amdgcn_update_dpp(0u, v, 0x112, 0xf, 0xf, true));
    v = keep_or(v, __builtin_amdgcn_update_dpp(0u, v, 0x114, 0xf, 0xf, true));
    v = keep_or(v, __builtin_amdgcn_update_dpp(0u, v, 0x118, 0xf, 0xf, true));
    v = keep_or(v, __builtin_amdgcn_update_dpp(0u, v, 0x142, 0xa, 0xf, true));
    v = keep_or(v, __builtin_amdgcn_update_dpp(0u, v, 0x143, 0xc, 0xf, true));
    return v;
}
__device__ __forceinline__ uint32_t wave_xor(uint32_t v) {   // lane 63: the XOR over the wave
    v ^= __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xf, 0xf, true);
    v ^= __builtin_amdgcn_update_dpp(0u, v, 0x112, 0xf, 0xf, true);
    v ^= __builtin_amdgcn_update_dpp(0u, v, 0x114, 0xf, 0xf, true);
    v ^= __builtin_amdgcn_update_dpp(0u, v, 0x118, 0xf, 0xf, true);
    v ^= __builtin_amdgcn_update_dpp(0u, v, 0x142, 0xa, 0xf, true);
    v ^= __builtin_amdgcn_update_dpp(0u, v, 0x143, 0xc, 0xf, true);
    return v;
}
constexpr uint32_t kPoly = 0xedb88320u;

// zlib's multmodp: a(x) * b(x) mod P(x), reflected representation (bit 31 = x^0)
__host__ __device__ constexpr uint32_t multmodp(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (uint32_t m = 1u << 31; m; m >>= 1) {
        if (a & m) p ^= b;
        b = (b & 1u) ? (b >> 1) ^ kPoly : b >> 1;
    }
    return p;
}
struct CrcPowers {
    uint32_t v[32];  // x^(2^k) mod P
};
__host__ __device__ constexpr CrcPowers crc_powers() {
    CrcPowers t{};
    uint32_t p = 1u << 30;  // x^1
    t.v[0] = p;
    for (int k = 1; k < 32; ++k) t.v[k] = p = multmodp(p, p);
    return t;
}
struct CrcWordPowers {
    uint32_t v[kCrcRow + 1];  // x^(32 m) mod P, m = 0 .. kCrcRow
};
__host__ __device__ constexpr CrcWordPowers crc_word_powers() {
    CrcWordPowers t{};
    const uint32_t step = crc_powers().v[5];  // x^32
    uint32_t p = 1u << 31;  // x^0
    for (uint32_t m = 0; m <= kCrcRow; ++m) {
        t.v[m] = p;
        p = multmodp(p, step);
    }
    return t;
}
struct CrcStrideTables {
    uint32_t v[1024];  // [k * 256 + b]: the register with byte k = b and nothing else, times x^(32 * kCrcRow)
};
__host__ __device__ constexpr CrcStrideTables crc_stride_tables(uint32_t log2_bits) {
    CrcStrideTables t{};
    const uint32_t shift = crc_powers().v[log2_bits];
    for (uint32_t k = 0; k < 4u; ++k)
        for (uint32_t b = 0; b < 256u; ++b) t.v[k * 256u + b] = multmodp(shift, b << (8u * k));
    return t;
}
__device__ const CrcWordPowers kX32 = crc_word_powers();
__device__ const CrcStrideTables kStride = crc_stride_tables(kRLog2 + 5u);   // a row of kRThreads words = 2^(kRLog2 + 5) bits

}  // namespace

__global__ __launch_bounds__(kRThreads) void k_inflate_resolve(const BgzfBlock* __restrict__ blocks, uint32_t n_blocks, uint8_t* __restrict__ out,
                                                               const uint32_t* __restrict__ tok, InflateInfo* __restrict__ info) {
    __shared__ uint4 s_win4[65536 / 16];
    __shared__ uint4 s_ptr4[kChunk * 2 / 16];  // 16-bit markers, then pointers; the CRC tables afterwards
    __shared__ uint32_t s_scan[4 * kRWaves + 1];
    __shared__ uint32_t s_flag[4];
    uint8_t* const win = reinterpret_cast<uint8_t*>(s_win4);
    u16a* const mark = reinterpret_cast<u16a*>(s_ptr4);
    u32a* const ptr32 = reinterpret_cast<u32a*>(s_ptr4);
    uint2a* const mine8 = reinterpret_cast<uint2a*>(s_ptr4) + threadIdx.x * (kPer / 4u);   // this thread's kPer consecutive pointers
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const uint32_t b = blockIdx.x;
    if (b >= n_blocks) return;
    const InflateInfo nf = info[b];
    if (nf.flag) return;  // (uniform: the other kernel's block)
    const BgzfBlock d = blocks[b];
    const uint32_t isize = d.isize;
    uint8_t* const o_base = out + d.dst;
    const uint32_t* const t_base = tok + d.tok;
    RPROF_T(c_start);
    {   // the block as phase 1 left it (literals in place, holes where matches go): all of a thread's loads in flight together
        uint4 w[kWinLoads];
#pragma unroll
        for (uint32_t k = 0; k < kWinLoads; ++k) {
            const uint32_t i = (tid + k * kRThreads) * 16u;
            w[k] = make_uint4(0u, 0u, 0u, 0u);
            if (i < isize) __builtin_memcpy(&w[k], o_base + i, 16);  // (up to 15 bytes behind the block: the next block's, or the buffer's slack)
        }
#pragma unroll
        for (uint32_t k = 0; k < kWinLoads; ++k) {
            const uint32_t i = (tid + k * kRThreads) * 16u;
            if (i < isize) s_win4[i >> 4] = w[k];
        }
    }
    if (tid < 4u) s_flag[tid] = 0;   // [0 .. 2]: "some pointer moved" of the jumping rounds, in turn; [3]: something is wrong
    __syncthreads();
    uint32_t base = 0, t0 = 0, rr = 0;
    bool wrong = false;
    RPROF_T(c_loaded);
    RPROF_ADD(0, c_start, c_loaded);
    uint32_t tv_next = tid < nf.n_tok ? t_base[tid] : 0x80000000u;   // (the next chunk's token is asked for while this one is filled)
    while (t0 < nf.n_tok) {
        // the table empty for this chunk's markers (the barrier at the end of a chunk stands between every thread's loading
        // its pointers and this; the markers come behind the next barrier)
#pragma unroll
        for (uint32_t k = 0; k < kPer / 4u; ++k) mine8[k] = uint2a{0u, 0u};
        // a token per thread, the spans' running sum
        RPROF_T(c0);
        const uint32_t t = t0 + tid;
        const uint32_t tv = tv_next;
        const bool skip = (tv >> 31) != 0u;
        const uint32_t litrun = skip ? tv & 0x7fffffffu : (tv >> 23) & 0xffu;
        const uint32_t mlen = skip ? 0u : ((tv >> 15) & 0xffu) + 3u;
        const uint32_t dist = (tv & 0x7fffu) + 1u;
        const uint32_t span = litrun + mlen;
        uint32_t incl = wave_scan_add(span);
        if (lane == 63u) s_scan[wv] = incl;
        if (tid == 0) s_scan[4 * kRWaves] = span;
        __syncthreads();
        RPROF_T(c1);
        RPROF_ADD(1, c0, c1);
        uint32_t before = 0;   // (all eight words asked for at once: a loop up to this wave's number waits for each in turn)
#pragma unroll
        for (uint32_t k = 0; k + 1u < kRWaves; ++k) before += k < wv ? s_scan[k] : 0u;
        incl += before;
        // the chunk: the tokens whose spans end inside kChunk bytes (a prefix: spans are sums)
        const bool in_chunk = t < nf.n_tok && incl <= kChunk;
        const uint64_t mine = __ballot(in_chunk);
        const uint32_t cnt = static_cast<uint32_t>(__popcll(mine));
        const uint32_t top = __builtin_amdgcn_readlane(incl, cnt ? cnt - 1u : 0u);   // (the count is the wave's: a scalar)
        if (lane == 0) {
            s_scan[kRWaves + wv] = cnt;
            s_scan[2 * kRWaves + wv] = cnt ? top : 0u;
        }
        // a token leaves MARKERS where its literals and its match begin: 1 = literals from here on, dist + 1 = bytes copied from
        // `dist` in front of themselves from here on.  (Every byte of the chunk lies behind a marker: a token that has neither
        // literals nor a match covers no byte.)  No loop over a token's bytes: the lanes of a wave would wait for its longest.
        if (in_chunk) {
            const uint32_t e0 = incl - span, d0 = incl - mlen;
            if (litrun) mark[e0] = static_cast<uint16_t>(kLitMarker);
            if (mlen) {
                mark[d0] = static_cast<uint16_t>(dist + 1u);
                if (base + d0 < dist) s_flag[3] = 1;   // a distance beyond the start of the output
            }
        }
        __syncthreads();
        RPROF_T(c2);
        RPROF_ADD(2, c1, c2);
        RPROF_INC(8, 1);
        uint32_t n_act = 0, S = 0;
#pragma unroll
        for (uint32_t k = 0; k < kRWaves; ++k) {
            n_act += s_scan[kRWaves + k];
            S = max(S, s_scan[2 * kRWaves + k]);
        }
        {
            const uint32_t tn = t0 + (n_act ? n_act : 1u) + tid;
            tv_next = tn < nf.n_tok ? t_base[tn] : 0x80000000u;
        }
        if (n_act == 0) {  // the first token is a run of literals longer than a chunk: nothing to fill (and no marker was left)
            const uint32_t first_span = s_scan[4 * kRWaves];
            __syncthreads();
            base += first_span;
            t0 += 1;
            if (base > isize) {
                wrong = true;
                break;
            }
            continue;
        }
        if (base + S > isize || s_flag[3] != 0u) {   // (a distance beyond the start of the output: the other kernel says what it is)
            wrong = true;
            break;
        }
        // the markers carried forward, a thread over its kPer consecutive bytes: the last marker at or in front of each byte.
        // Inside the thread in registers, across the wave's threads by a scan of "the right one unless it is empty", across the
        // waves through LDS.  Then every byte's pointer: a literal to itself + kLit, a byte of a match to the byte
        // `dist` in front of it (relative to the chunk: negative = an earlier chunk's byte); behind the chunk's end: kLit.
        bool moving = false;
        {
            uint32_t m[kPer];
#pragma unroll
            for (uint32_t k = 0; k < kPer / 4u; ++k) {
                const uint2a v = mine8[k];
                m[4 * k] = v.x & 0xffffu;
                m[4 * k + 1] = v.x >> 16;
                m[4 * k + 2] = v.y & 0xffffu;
                m[4 * k + 3] = v.y >> 16;
            }
            uint32_t last = 0;
#pragma unroll
            for (uint32_t k = 0; k < kPer; ++k) {
                last = keep_or(m[k], last);
                m[k] = last;
            }
            const uint32_t carry = wave_scan_last(last);   // inclusive over the wave's threads
            if (lane == 63u) s_scan[3 * kRWaves + wv] = carry;
            uint32_t in = __builtin_amdgcn_update_dpp(0u, carry, 0x138, 0xf, 0xf, true);   // wave_shr:1 (lane 0: none)
            __syncthreads();
            {
                uint32_t far = 0;   // (the nearest earlier wave that saw a marker; all words asked for at once)
#pragma unroll
                for (uint32_t k = 0; k + 1u < kRWaves; ++k) far = keep_or(k < wv ? s_scan[3 * kRWaves + k] : 0u, far);
                in = keep_or(in, far);
            }
            uint32_t pw[kPer];
#pragma unroll
            for (uint32_t k = 0; k < kPer; ++k) {
                const uint32_t j = tid * kPer + k;
                const uint32_t v = m[k] ? m[k] : in;
                const uint32_t pp = j + 1u - v;   // (a match: j - dist; literals: j + kLit -- the marker is made for it)
                pw[k] = (j < S ? pp : kLit) & 0xffffu;   // (behind the chunk's end: final, and a place the gather may read)
            }
#pragma unroll
            for (uint32_t k = 0; k < kPer / 4u; ++k)
                mine8[k] = uint2a{pw[4 * k] | (pw[4 * k + 1] << 16), pw[4 * k + 2] | (pw[4 * k + 3] << 16)};
#pragma unroll
            for (uint32_t k = 0; k < kPer; ++k) moving = moving | (pw[k] < kLit);   // (these bytes; all threads' together: the chunk)
        }
        // (the barrier behind the pointers is the first round's too: whether any pointer moves at all is known here)
        bool go;
        {
            const uint32_t slot = rr % 3u;
            if (moving) s_flag[slot] = 1;
            if (tid == 0) s_flag[(rr + 1u) % 3u] = 0;
            __syncthreads();
            ++rr;
            go = s_flag[slot] != 0u;
        }
        // pointer jumping: until every pointer is FINAL -- at a literal or in front of the chunk (the ranges above).  A thread
        // keeps the pointers of its kPer bytes (j = tid + kRThreads k) in registers; a round replaces a pointer by the pointer
        // found where it points (one LDS read, all of a thread's in flight together) and learns from that word's own bits
        // whether it is final now: no round that only confirms.  Reads and writes are not ordered against each other: whatever
        // a pointer's place holds is a pointer to a byte of the same value.  The rounds' "some pointer is not final" words take
        // turns (three: the one a round sets was cleared two barriers ago)
        uint32_t pj[kPer];
#pragma unroll
        for (uint32_t k = 0; k < kPer; ++k) pj[k] = mark[tid + kRThreads * k];   // (behind the chunk's end the table holds a final pointer)
        RPROF_T(c3);
        RPROF_ADD(3, c2, c3);
        while (go) {
            RPROF_INC(9, 1);
            // kJumps jumps per round and barrier, every lane the same instructions (a final pointer reads some word and keeps itself)
            uint32_t cur[kPer];
#pragma unroll
            for (uint32_t k = 0; k < kPer; ++k) cur[k] = pj[k];
#pragma unroll
            for (uint32_t jump = 0; jump < kJumps; ++jump) {
                uint32_t q[kPer];
#pragma unroll
                for (uint32_t k = 0; k < kPer; ++k) q[k] = mark[cur[k] & (kLit - 1u)];
#pragma unroll
                for (uint32_t k = 0; k < kPer; ++k) cur[k] = cur[k] < kLit ? q[k] : cur[k];
            }
            moving = false;
#pragma unroll
            for (uint32_t k = 0; k < kPer; ++k) {
                if (pj[k] < kLit) mark[tid + kRThreads * k] = static_cast<uint16_t>(cur[k]);
                pj[k] = cur[k];
                moving = moving | (cur[k] < kLit);
            }
            const uint32_t slot = rr % 3u;
            if (moving) s_flag[slot] = 1;
            if (tid == 0) s_flag[(rr + 1u) % 3u] = 0;
            __syncthreads();
            ++rr;
            go = s_flag[slot] != 0u;
        }
        RPROF_T(c4);
        RPROF_ADD(4, c3, c4);
        {
            // every byte of the chunk from where its pointer ends (a literal takes its own value again)
            uint8_t v[kPer];
#pragma unroll
            for (uint32_t k = 0; k < kPer; ++k) {
                const int32_t t = static_cast<int32_t>(pj[k] << 16) >> 16;
                const int32_t off = t & (static_cast<int32_t>(kLit - 1u) | (t >> 31));   // negative: as it is; else the literal's place
                v[k] = win[static_cast<int32_t>(base) + off];   // (never below 0: such a block left above; behind S: not stored)
            }
#pragma unroll
            for (uint32_t k = 0; k < kPer; ++k) {
                const uint32_t j = tid + kRThreads * k;
                if (j < S) win[base + j] = v[k];
            }
        }
        __syncthreads();
        RPROF_T(c5);
        RPROF_ADD(5, c4, c5);
        base += S;
        t0 += n_act;
    }
    __syncthreads();
    RPROF_T(c_crc);
    if (wrong || s_flag[3]) {
        if (tid == 0) info[b].flag = 1;
        return;
    }
    // ---- the gzip trailer's CRC32.  A thread takes the words t, t + 512, t + 1024 ... of the block (the threads of a wave read
    // consecutive words: a segment of consecutive bytes per thread puts all 64 lanes on one LDS bank): its register is
    // c = (c ^ word) * x^(8 * 2048) per row -- four look-ups in tables made for that power, like slice-by-4's for x^32 --,
    // then times x^(32 * the words from its last one on) from a table, and the threads' registers XORed up (the CRC is
    // linear; the initial ~0 goes in with word 0).  The bytes behind the last full word and the final complement: thread 0.
    u32a* const st = ptr32;
    for (uint32_t i = tid; i < 1024u; i += kRThreads) st[i] = kStride.v[i];
    __syncthreads();
    const uint32_t nwords = isize >> 2;
    uint32_t part = 0;
    if (tid < nwords) {
        const u32a* w32 = reinterpret_cast<const u32a*>(win);
        uint32_t c = tid == 0 ? 0xffffffffu : 0u, i = tid;
        for (;;) {
            c ^= w32[i];
            i += kRThreads;
            if (i >= nwords) break;
            c = st[c & 0xffu] ^ st[256u + ((c >> 8) & 0xffu)] ^ st[512u + ((c >> 16) & 0xffu)] ^ st[768u + (c >> 24)];
        }
        part = multmodp(kX32.v[nwords - (i - kRThreads)], c);   // (its last word and the words behind it)
    }
    part = wave_xor(part);
    if (lane == 63u) s_scan[wv] = part;
    __syncthreads();
    if (tid == 0) {
        uint32_t c = nwords ? 0u : 0xffffffffu;
        for (uint32_t k = 0; k < kRWaves; ++k) c ^= s_scan[k];
        for (uint32_t i = nwords << 2; i < isize; ++i) {
            c ^= win[i];
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? kPoly ^ (c >> 1) : c >> 1;
        }
        s_flag[3] = ~c == d.crc ? 0u : 1u;
    }
    __syncthreads();
    if (s_flag[3]) {
        if (tid == 0) info[b].flag = 1;
        return;
    }
    RPROF_T(c_out);
    RPROF_ADD(6, c_crc, c_out);
    for (uint32_t i = tid * 16u; i < isize; i += kRThreads * 16u) {
        if (i + 16u <= isize) {
            const uint4 v = s_win4[i >> 4];
            __builtin_memcpy(o_base + i, &v, 16);
        } else {
            for (uint32_t k = i; k < isize; ++k) o_base[k] = win[k];
        }
    }
    RPROF_T(c_end);
    RPROF_ADD(7, c_out, c_end);
    RPROF_ADD(10, c_start, c_end);
}

size_t bgzf_inflate_scratch_bytes(uint32_t n_blocks, uint64_t token_words) {
    return bgzf_lanes_scratch_bytes(kBgzfMaxGrid) + ((static_cast<size_t>(n_blocks) * sizeof(InflateInfo) + 255u) & ~static_cast<size_t>(255u)) +
           token_words * 4u + 256u;
}

// status[0] = the largest error code met (0: every block inflated to its ISIZE and CRC), status[1] = the first bad block
// (preset ~0), status[2] = blocks that went through the lane-per-block kernel (preset 0)
void launch_bgzf_inflate(hipStream_t st, const uint8_t* comp, const BgzfBlock* blocks, uint32_t n_blocks, uint8_t* out, void* scratch,
                         uint32_t* status) {
    if (!n_blocks) return;
    uint8_t* s = static_cast<uint8_t*>(scratch);
    void* lanes_scratch = s;
    s += bgzf_lanes_scratch_bytes(kBgzfMaxGrid);
    InflateInfo* info = reinterpret_cast<InflateInfo*>(s);
    s += (static_cast<size_t>(n_blocks) * sizeof(InflateInfo) + 255u) & ~static_cast<size_t>(255u);
    uint32_t* tok = reinterpret_cast<uint32_t*>(s);
    hipLaunchKernelGGL(k_inflate_decode, dim3((n_blocks + 63u) / 64u), dim3(128), 0, st, comp, blocks, n_blocks, out, tok, info);
    hipLaunchKernelGGL(k_inflate_resolve, dim3(n_blocks), dim3(kRThreads), 0, st, blocks, n_blocks, out, tok, info);
    launch_bgzf_inflate_lanes(st, comp, blocks, n_blocks, out, lanes_scratch, bgzf_inflate_grid(n_blocks), status, info);
}

}  // namespace slimm

#if defined(EXP) && EXP == 12
extern "C" int slimm_debug_prof_decode(unsigned long long* out, int n, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(slimm::g_prof_d), sizeof(unsigned long long) * n);
    if (reset) {
        static unsigned long long z[8 * 1024];
        (void)hipMemcpyToSymbol(HIP_SYMBOL(slimm::g_prof_d), z, sizeof(z));
    }
    return e == hipSuccess ? 0 : -1;
}
#endif
#if defined(EXP) && EXP == 11
extern "C" int slimm_debug_prof_resolve(unsigned long long* out, int n, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(slimm::g_prof_r), sizeof(unsigned long long) * n);
    if (reset) {
        static unsigned long long z[16 * 1024];
        (void)hipMemcpyToSymbol(HIP_SYMBOL(slimm::g_prof_r), z, sizeof(z));
    }
    return e == hipSuccess ? 0 : -1;
}
#endif
