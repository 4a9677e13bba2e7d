// BGZF blocks inflated on the device (gfx950, wave64): DEFLATE (RFC 1951) streams of at most 64 KB, independent of each other.
//
// Where it stands on the path: the reference reads its input through seqan::BamFileIn (call sites src/misc.hpp:498-522,
// src/slimm.hpp:194-208), which inflates the BGZF blocks of a BAM file and decodes the records.  This library already finds
// and decodes the records on the device (bam_decode.hip); with this kernel the inflate happens there too, and what crosses
// PCIe is the compressed file (a seventeenth to a fifth of the inflated bytes).
//
// Shape: a LANE per block.  A block's symbols depend on each other (every Huffman code starts where the one before ended,
// every match copies what was written before), so a block is a sequential job; a file has hundreds of thousands of them.
// Every lane keeps its bit buffer and pointers in registers and the two first-level decode tables of its current DEFLATE
// block in LDS (8 bits for the literal / length code, 6 bits for the distance code; entry = symbol << 4 | code length,
// lane-interleaved so that the lanes of a wave looking up different entries meet in different banks more often than not;
// 70 KB per wave with the code lengths and counts of the block being set up: two waves per CU).  Codes longer than the first
// level -- rare symbols -- take the canonical walk (count per length + symbols sorted by code: puff's decode); the sorted
// literal / length symbols are the one array in global memory (1 KB of scratch per lane).  Literals leave as byte stores;
// a match is ONE round of loads from the lane's own output (its period, at most 258 bytes, into registers) and 16-byte
// stores.  Measured (scripts/inflate_rate.py, 25 K blocks of a BAM file = 78 % of the lanes busy): 28 - 32 GB/s of inflated
// bytes, whatever the compression ratio -- every trip of a wave's symbol loop pays one round trip to the L2 (some lane
// always has a match), ~4 us per symbol under load; 16 host cores inflate 30 (names in random order) to 55 GB/s.
//
// Checked: block type, stored-length complement, over-subscribed / incomplete codes, distances beyond the output so far,
// output beyond ISIZE, input beyond the payload, that a block ends exactly at ISIZE bytes, and the CRC32 of the gzip trailer
// over what was written (a second pass of the lane over its own output, four table look-ups in LDS per four bytes).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "kernels.h"

namespace slimm {

namespace {

constexpr uint32_t kLitBits = 8, kDistBits = 6;
constexpr uint32_t kLitTab = 1u << kLitBits, kDistTab = 1u << kDistBits;

// A lane's working set.  In LDS, lane-interleaved (element i of a lane at [i * 64]): the two first-level tables, the code
// lengths of the block being set up, the counts per code length of its two codes and the sorted symbols of the distance code
// (which also serves the code-length code of a dynamic block's header).  In global memory (1 KB per resident lane): the
// sorted symbols of the literal / length code -- written once per block, read only by codes longer than the first level.
struct LaneState {
    uint16_t *ltab, *dtab;      // LDS, stride 64
    uint8_t* len;               // LDS, stride 64: 288 literal / length + 32 distance code lengths
    uint16_t *lcount, *dcount;  // LDS, stride 64: 16 each
    uint16_t* dsym;             // LDS, stride 64: 32
    uint16_t* offs;             // LDS, stride 64: 16 (construction)
    uint16_t* lsym;             // global, stride 1: 288
};
constexpr uint32_t kScratchPerLane = 1024;  // bytes of global scratch per resident lane (lsym: 576 used)

__device__ __forceinline__ uint32_t ld32(const uint8_t* p) {
    uint32_t w;
    __builtin_memcpy(&w, p, 4);
    return w;
}
__device__ __forceinline__ uint64_t ld64(const uint8_t* p) {
    uint64_t w;
    __builtin_memcpy(&w, p, 8);
    return w;
}
__device__ __forceinline__ void st64(uint8_t* p, uint64_t w) { __builtin_memcpy(p, &w, 8); }

struct BitReader {
    const uint8_t* in;    // the next four bytes the buffer will take: already loaded, in `ahead`
    const uint8_t* end;   // behind the payload
    uint64_t buf;
    uint32_t cnt;
    uint32_t ahead;       // (asked for one refill early: a refill waits for no load, the load has three symbols' time)
    __device__ __forceinline__ void start(const uint8_t* p) {
        in = p;
        buf = 0;
        cnt = 0;
        ahead = ld32(p);
    }
    __device__ __forceinline__ void refill() {  // at least 33 bits afterwards (the buffer behind `end` is padded)
        if (cnt <= 32u) {
            buf |= static_cast<uint64_t>(ahead) << cnt;
            in += 4;
            cnt += 32u;
            ahead = ld32(in);
        }
    }
    __device__ __forceinline__ uint32_t peek(uint32_t n) const { return static_cast<uint32_t>(buf) & ((1u << n) - 1u); }
    __device__ __forceinline__ void drop(uint32_t n) {
        buf >>= n;
        cnt -= n;
    }
    __device__ __forceinline__ uint32_t take(uint32_t n) {  // n <= 16, refilled before
        const uint32_t v = peek(n);
        drop(n);
        return v;
    }
    // bytes of the payload used so far (whole bytes still in the buffer are not)
    __device__ __forceinline__ int64_t used(const uint8_t* start_) const { return (in - start_) - static_cast<int64_t>(cnt >> 3); }
};

// puff's construct(): count[len] and the symbols in canonical order from the code lengths.  Returns 0 for a complete code,
// > 0 for an incomplete one, < 0 for an over-subscribed one.  count / length / offs: LDS with stride 64; sym: stride SS.
template <uint32_t SS>
__device__ int huff_construct(uint16_t* count, uint16_t* sym, const uint8_t* length, uint32_t n, uint16_t* offs) {
    for (uint32_t l = 0; l <= 15u; ++l) count[l * 64u] = 0;
    for (uint32_t s = 0; s < n; ++s) ++count[length[s * 64u] * 64u];
    if (count[0] == n) return 0;  // no codes: complete, but decoding will fail
    int left = 1;
    for (uint32_t l = 1; l <= 15u; ++l) {
        left <<= 1;
        left -= static_cast<int>(count[l * 64u]);
        if (left < 0) return left;
    }
    offs[1 * 64u] = 0;
    for (uint32_t l = 1; l < 15u; ++l) offs[(l + 1u) * 64u] = static_cast<uint16_t>(offs[l * 64u] + count[l * 64u]);
    for (uint32_t s = 0; s < n; ++s) {
        const uint32_t l = length[s * 64u];
        if (l != 0) sym[(offs[l * 64u]++) * SS] = static_cast<uint16_t>(s);
    }
    return left;
}

// the first-level table of a code: every `bits`-bit pattern whose low bits are a code of at most `bits` bits -> symbol << 4 |
// length; patterns that start a longer code stay 0.  Straight from the code lengths: the codes of one length are consecutive
// in symbol order, the first code of a length follows from the counts (offs: the next code per length).
__device__ void huff_table(uint16_t* tab, uint32_t bits, const uint16_t* count, const uint8_t* length, uint32_t n, uint16_t* next) {
    const uint32_t size = 1u << bits;
    for (uint32_t i = 0; i < size; ++i) tab[i * 64u] = 0;
    uint32_t code = 0;
    for (uint32_t l = 1; l <= bits; ++l) {
        next[l * 64u] = static_cast<uint16_t>(code);
        code = (code + count[l * 64u]) << 1;
    }
    for (uint32_t s = 0; s < n; ++s) {
        const uint32_t l = length[s * 64u];
        if (l == 0 || l > bits) continue;
        const uint32_t c = next[l * 64u]++;
        // (codes are packed starting from their most significant bit: the pattern in the stream is the code reversed)
        const uint32_t rev = __builtin_bitreverse32(c) >> (32u - l);
        const uint16_t e = static_cast<uint16_t>((s << 4) | l);
        for (uint32_t i = rev; i < size; i += 1u << l) tab[i * 64u] = e;
    }
}

// puff's decode(): a symbol of a canonical code, one bit at a time (codes longer than the first-level table; the codes of a
// dynamic block's header)
template <uint32_t SS>
__device__ int huff_walk(BitReader& br, const uint16_t* count, const uint16_t* sym) {
    int code = 0, first = 0, index = 0;
    for (uint32_t l = 1; l <= 15u; ++l) {
        code |= static_cast<int>(br.take(1));
        const int c = count[l * 64u];
        if (code - c < first) return sym[static_cast<uint32_t>(index + (code - first)) * SS];
        index += c;
        first += c;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}

template <uint32_t SS>
__device__ __forceinline__ int huff_symbol(BitReader& br, const uint16_t* tab, uint32_t bits, const uint16_t* count, const uint16_t* sym) {
    const uint32_t e = tab[br.peek(bits) * 64u];
    if (e) {
        br.drop(e & 15u);
        return static_cast<int>(e >> 4);
    }
    return huff_walk<SS>(br, count, sym);
}

__device__ const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__device__ const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__device__ const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__device__ const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__device__ const uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

enum : uint32_t {
    kInfOk = 0, kInfBadType = 1, kInfBadStored = 2, kInfBadCode = 3, kInfBadSymbol = 4, kInfBadDistance = 5, kInfTooLong = 6,
    kInfInputEnd = 7, kInfShort = 8, kInfCrc = 9
};

// one DEFLATE stream: payload [src, src + csize) -> out[0, isize).  Returns kInf*.
__device__ uint32_t inflate_block(const uint8_t* src, uint32_t csize, uint8_t* out, uint32_t isize, const LaneState& L) {
    uint16_t* const ltab = L.ltab;
    uint16_t* const dtab = L.dtab;
    uint8_t* const len_ = L.len;
    BitReader br;
    br.end = src + csize;
    br.start(src);
    uint32_t o = 0;
    for (;;) {
        br.refill();
        const uint32_t last = br.take(1), type = br.take(2);
        if (type == 0u) {  // stored: to the next byte boundary, LEN, ~LEN, the bytes
            br.drop(br.cnt & 7u);
            br.refill();
            const uint32_t len = br.take(16);
            br.refill();
            const uint32_t nlen = br.take(16);
            if ((len ^ nlen) != 0xffffu) return kInfBadStored;
            // the bit buffer holds whole bytes now: hand them back
            const uint8_t* p = br.in - (br.cnt >> 3);
            if (p + len > br.end) return kInfInputEnd;
            if (o + len > isize) return kInfTooLong;
            for (uint32_t i = 0; i < len; ++i) out[o + i] = p[i];
            o += len;
            br.start(p + len);
        } else if (type == 1u || type == 2u) {
            uint32_t nlen = 288, ndist = 30;
            if (type == 1u) {  // the fixed code
                for (uint32_t s = 0; s < 144u; ++s) len_[s * 64u] = 8;
                for (uint32_t s = 144u; s < 256u; ++s) len_[s * 64u] = 9;
                for (uint32_t s = 256u; s < 280u; ++s) len_[s * 64u] = 7;
                for (uint32_t s = 280u; s < 288u; ++s) len_[s * 64u] = 8;
                for (uint32_t s = 0; s < 30u; ++s) len_[(288u + s) * 64u] = 5;
            } else {
                br.refill();
                nlen = br.take(5) + 257u;
                ndist = br.take(5) + 1u;
                const uint32_t ncode = br.take(4) + 4u;
                if (nlen > 286u || ndist > 30u) return kInfBadCode;
                // the code length code: its 19 lengths in the (still unused) first-level table's space, its canonical form in
                // the distance code's arrays, which are set up after it
                uint8_t* const cl = reinterpret_cast<uint8_t*>(ltab - threadIdx.x) + threadIdx.x;  // (byte lane offset)
                for (uint32_t i = 0; i < 19u; ++i) cl[i * 64u] = 0;
                for (uint32_t i = 0; i < ncode; ++i) {
                    br.refill();
                    cl[kClOrder[i] * 64u] = static_cast<uint8_t>(br.take(3));
                }
                if (huff_construct<64>(L.dcount, L.dsym, cl, 19, L.offs) != 0) return kInfBadCode;
                uint32_t index = 0;
                while (index < nlen + ndist) {
                    br.refill();
                    const int sym = huff_walk<64>(br, L.dcount, L.dsym);
                    if (sym < 0) return kInfBadSymbol;
                    if (sym < 16) {
                        len_[(index++) * 64u] = static_cast<uint8_t>(sym);
                    } else {
                        uint32_t rep, val = 0;
                        br.refill();
                        if (sym == 16) {
                            if (index == 0) return kInfBadCode;
                            val = len_[(index - 1u) * 64u];
                            rep = 3u + br.take(2);
                        } else if (sym == 17) {
                            rep = 3u + br.take(3);
                        } else {
                            rep = 11u + br.take(7);
                        }
                        if (index + rep > nlen + ndist) return kInfBadCode;
                        while (rep--) len_[(index++) * 64u] = static_cast<uint8_t>(val);
                    }
                }
                // the distance lengths behind entry 288 (where the fixed code has them), nothing between the two codes
                if (nlen < 288u) {
                    for (uint32_t s = ndist; s-- > 0;) len_[(288u + s) * 64u] = len_[(nlen + s) * 64u];
                    for (uint32_t s = nlen; s < 288u; ++s) len_[s * 64u] = 0;
                }
                if (len_[256u * 64u] == 0) return kInfBadCode;  // no end-of-block code
                nlen = 288u;
            }
            // (an incomplete code is only allowed when it has a single code; the fixed distance code -- 30 of 32 five-bit
            // codes -- is what it is)
            const uint32_t nl = type == 1u ? 288u : nlen;
            int err = huff_construct<1>(L.lcount, L.lsym, len_, nl, L.offs);
            if (err < 0 || (err > 0 && nl - L.lcount[0] != 1u)) return kInfBadCode;
            err = huff_construct<64>(L.dcount, L.dsym, len_ + 288u * 64u, ndist, L.offs);
            if (type == 2u && (err < 0 || (err > 0 && ndist - L.dcount[0] != 1u))) return kInfBadCode;
            huff_table(ltab, kLitBits, L.lcount, len_, nl, L.offs);
            huff_table(dtab, kDistBits, L.dcount, len_ + 288u * 64u, ndist, L.offs);
            for (;;) {
                br.refill();
                if (br.in > br.end + 8) return kInfInputEnd;
                int sym = huff_symbol<1>(br, ltab, kLitBits, L.lcount, L.lsym);
                if (sym < 0) return kInfBadSymbol;
                if (sym < 256) {
                    if (o >= isize) return kInfTooLong;
                    out[o++] = static_cast<uint8_t>(sym);
                    continue;
                }
                if (sym == 256) break;
                sym -= 257;
                if (sym >= 29) return kInfBadSymbol;
                const uint32_t len = kLenBase[sym] + br.take(kLenExtra[sym]);
                br.refill();
                const int ds = huff_symbol<64>(br, dtab, kDistBits, L.dcount, L.dsym);
                if (ds < 0 || ds >= 30) return kInfBadSymbol;
                br.refill();
                const uint32_t dist = kDistBase[ds] + br.take(kDistExtra[ds]);
                if (dist > o) return kInfBadDistance;
                if (o + len > isize) return kInfTooLong;
                // The copy.  A byte loop waits for one load per byte (the store of a byte may be the source of the next, so the
                // loads cannot be asked for together) -- 1.6 us per byte measured; and on this hardware a wait for a load
                // is a wait for every store issued before it as well.  So: ONE round of loads per match.  What a match
                // writes is periodic with the period min(distance, length); that much of the source (at most 258 bytes: 17
                // registers of 16 bytes) is loaded at once and stored as many times as the match is long -- a store of 16 bytes
                // may write garbage behind the period's or the match's end, which the next store of this match or the
                // symbols that follow overwrite (never beyond the block's own output: its last bytes go the slow way).
                // Distances below eight: the pattern is read once and repeated from two registers.
                const uint8_t* from = out + o - dist;
                uint8_t* to = out + o;
                if (o + len + 16u <= isize) {
                    if (dist >= 8u) {
                        const uint32_t period = dist < len ? dist : len;
                        uint4 r[17];
#pragma unroll
                        for (uint32_t j = 0; j < 17u; ++j)
                            if (j * 16u < period) __builtin_memcpy(&r[j], from + j * 16u, 16);
                        for (uint32_t k = 0; k < len; k += period) {
                            const uint32_t n = len - k < period ? len - k : period;
#pragma unroll
                            for (uint32_t j = 0; j < 17u; ++j)
                                if (j * 16u < n) __builtin_memcpy(to + k + j * 16u, &r[j], 16);
                        }
                    } else {
                        // sixteen bytes of the repeating pattern in two registers; the chunk at pattern position `phase` is a
                        // funnel shift of them, and `phase` advances by 8 mod dist per chunk (no division anywhere)
                        uint64_t p0 = ld64(from) & ((1ull << (8u * dist)) - 1ull);
                        for (uint32_t k = dist; k < 8u; k += dist) p0 |= p0 << (8u * k);   // (what does not fit the 8 bytes falls off)
                        uint32_t step = 8u;
                        while (step >= dist) step -= dist;
                        uint64_t p1 = 0;
                        for (uint32_t k = 0, idx = step; k < 8u; ++k) {
                            p1 |= ((p0 >> (8u * idx)) & 0xffull) << (8u * k);
                            if (++idx == dist) idx = 0;
                        }
                        uint32_t phase = 0;
                        for (uint32_t i = 0; i < len; i += 8u) {
                            st64(to + i, phase ? (p0 >> (8u * phase)) | (p1 << (64u - 8u * phase)) : p0);
                            phase += step;
                            if (phase >= dist) phase -= dist;
                        }
                    }
                } else {
                    for (uint32_t i = 0; i < len; ++i) to[i] = from[i];  // (the last bytes of a block)
                }
                o += len;
            }
        } else {
            return kInfBadType;
        }
        if (last) break;
    }
    if (o != isize) return kInfShort;
    if (br.used(src) > static_cast<int64_t>(csize)) return kInfInputEnd;
    return kInfOk;
}

}  // namespace

// CRC32 (the gzip one: reflected 0xEDB88320) of out[0, n), four bytes per step through four tables of 256 words in LDS
// (shared by the lanes of the workgroup); the words are asked for eight at a time -- the lane reads back what it wrote.
__device__ uint32_t crc32_of(const uint8_t* p, uint32_t n, const uint32_t* t) {
    uint32_t crc = 0xffffffffu, i = 0;
    for (; i + 32u <= n; i += 32u) {
        uint32_t w[8];
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k) w[k] = ld32(p + i + 4u * k);
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k) {
            crc ^= w[k];
            crc = t[768u + (crc & 0xffu)] ^ t[512u + ((crc >> 8) & 0xffu)] ^ t[256u + ((crc >> 16) & 0xffu)] ^ t[crc >> 24];
        }
    }
    for (; i < n; ++i) crc = t[(crc ^ p[i]) & 0xffu] ^ (crc >> 8);
    return ~crc;
}

__global__ __launch_bounds__(64) void k_bgzf_inflate(const uint8_t* __restrict__ comp, const BgzfBlock* __restrict__ blocks, uint32_t n_blocks,
                                                     uint8_t* __restrict__ out, uint8_t* __restrict__ scratch, uint32_t* __restrict__ status,
                                                     const InflateInfo* __restrict__ info) {
    __shared__ uint16_t s_ltab[kLitTab * 64u];
    __shared__ uint16_t s_dtab[kDistTab * 64u];
    __shared__ uint16_t s_small[(16u + 16u + 32u + 16u) * 64u];  // lcount, dcount, dsym, offs
    __shared__ uint8_t s_len[320u * 64u];
    __shared__ uint32_t s_crc[4u * 256u];
    const uint32_t lane = threadIdx.x;
    const uint32_t slot = blockIdx.x * 64u + lane, n_slots = gridDim.x * 64u;
    for (uint32_t i = lane; i < 256u; i += 64u) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xedb88320u ^ (c >> 1) : c >> 1;
        s_crc[i] = c;
    }
    __syncthreads();
    for (uint32_t i = lane; i < 256u; i += 64u) {
        uint32_t c = s_crc[i];
        for (uint32_t k = 1; k < 4u; ++k) {
            c = s_crc[c & 0xffu] ^ (c >> 8);
            s_crc[k * 256u + i] = c;
        }
    }
    __syncthreads();
    LaneState L;
    L.ltab = s_ltab + lane;
    L.dtab = s_dtab + lane;
    L.len = s_len + lane;
    L.lcount = s_small + lane;
    L.dcount = s_small + 16u * 64u + lane;
    L.dsym = s_small + 32u * 64u + lane;
    L.offs = s_small + 64u * 64u + lane;
    L.lsym = reinterpret_cast<uint16_t*>(scratch + static_cast<size_t>(slot) * kScratchPerLane);
    uint32_t bad = 0, first_bad = 0xffffffffu, mine = 0;
    for (uint32_t b = slot; b < n_blocks; b += n_slots) {
        if (info && !info[b].flag) continue;  // (the two-phase kernels of bgzf_tokens.hip have inflated it)
        ++mine;
        const BgzfBlock d = blocks[b];
        uint32_t rc = inflate_block(comp + d.src, d.csize, out + d.dst, d.isize, L);
        if (rc == kInfOk && crc32_of(out + d.dst, d.isize, s_crc) != d.crc) rc = kInfCrc;
        if (rc != kInfOk && !bad) {
            bad = rc;
            first_bad = b;
        }
    }
    if (bad) {
        atomicMin(&status[1], first_bad);
        atomicMax(&status[0], bad);
    }
    if (info && mine) atomicAdd(&status[2], mine);
}

uint32_t bgzf_inflate_grid(uint32_t n_blocks) { return std::max(1u, std::min(n_blocks / 64u + 1u, kBgzfMaxGrid)); }
size_t bgzf_lanes_scratch_bytes(uint32_t grid) { return static_cast<size_t>(grid) * 64u * kScratchPerLane; }

void launch_bgzf_inflate_lanes(hipStream_t st, const uint8_t* comp, const BgzfBlock* blocks, uint32_t n_blocks, uint8_t* out, void* scratch,
                               uint32_t grid, uint32_t* status, const InflateInfo* info) {
    if (!n_blocks) return;
    hipLaunchKernelGGL(k_bgzf_inflate, dim3(grid), dim3(64), 0, st, comp, blocks, n_blocks, out, static_cast<uint8_t*>(scratch),
                       status, info);
}

}  // namespace slimm

namespace slimm {
// Host side: the BGZF blocks of `bytes` (whole blocks, one behind the other) as descriptors; payloads are addressed relative
// to `bytes`, outputs one behind the other from `dst0` on.  Returns false + why for anything that is not a BGZF block.
bool bgzf_parse_blocks(const uint8_t* bytes, uint64_t n_bytes, uint64_t dst0, std::vector<BgzfBlock>& out, uint64_t& inflated,
                       std::string& err) {
    uint64_t p = 0, dst = dst0;
    uint32_t tok = 0;
    inflated = 0;
    while (p < n_bytes) {
        if (n_bytes - p < 28) {
            err = "truncated BGZF block";
            return false;
        }
        const uint8_t* h = bytes + p;
        if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) {
            err = "not a BGZF block (gzip magic / FEXTRA)";
            return false;
        }
        const uint32_t xlen = h[10] | (static_cast<uint32_t>(h[11]) << 8);
        if (n_bytes - p < 12ull + xlen + 8ull) {
            err = "truncated BGZF block";
            return false;
        }
        uint32_t bsize = 0;
        for (uint32_t q = 0; q + 4 <= xlen;) {  // the BC subfield
            const uint8_t* f = h + 12 + q;
            const uint32_t slen = f[2] | (static_cast<uint32_t>(f[3]) << 8);
            if (f[0] == 'B' && f[1] == 'C' && slen == 2 && q + 6 <= xlen) bsize = (f[4] | (static_cast<uint32_t>(f[5]) << 8)) + 1u;
            q += 4 + slen;
        }
        if (bsize < 12u + xlen + 8u || bsize > n_bytes - p) {
            err = bsize ? "truncated BGZF block" : "BGZF block without a BC field";
            return false;
        }
        const uint8_t* t = h + bsize - 4;
        const uint32_t isize = t[0] | (static_cast<uint32_t>(t[1]) << 8) | (static_cast<uint32_t>(t[2]) << 16) | (static_cast<uint32_t>(t[3]) << 24);
        if (isize > 65536u) {
            err = "BGZF block of more than 64 KB";
            return false;
        }
        const uint8_t* tc = h + bsize - 8;
        BgzfBlock d;
        d.crc = tc[0] | (static_cast<uint32_t>(tc[1]) << 8) | (static_cast<uint32_t>(tc[2]) << 16) | (static_cast<uint32_t>(tc[3]) << 24);
        d.tok = tok;
        d.src = p + 12u + xlen;
        d.csize = bsize - 12u - xlen - 8u;
        d.isize = isize;
        d.dst = dst;
        // (the empty block at the end of a file -- a fixed-code block that holds its end-of-block code only -- has nothing to
        // inflate; any other payload that claims ISIZE 0 is inflated all the same: it must give no byte and the CRC of none)
        const bool eof_block = isize == 0 && d.csize == 2 && h[12u + xlen] == 0x03 && h[13u + xlen] == 0x00 && d.crc == 0;
        if (!eof_block) {
            out.push_back(d);
            tok += bgzf_token_room(isize);
        }
        dst += isize;
        inflated += isize;
        p += bsize;
    }
    return true;
}

}  // namespace slimm

// ---------------------------------------------------------------------------------------------------------
// A whole buffer of BGZF blocks in host memory -> their inflated bytes in host memory, through the device: the building
// block by itself (tests, throughput measurements; include/slimm_hip.h).  kernel_ms: the inflate kernel alone.
// ---------------------------------------------------------------------------------------------------------
extern "C" int slimm_bgzf_inflate_with(int device, const uint8_t* blocks, uint64_t n_bytes, uint8_t* out, uint64_t out_cap,
                                       uint64_t* out_bytes, double* kernel_ms, char* err, uint64_t err_cap, uint32_t how,
                                       uint32_t* lane_blocks);
extern "C" int slimm_bgzf_inflate(int device, const uint8_t* blocks, uint64_t n_bytes, uint8_t* out, uint64_t out_cap, uint64_t* out_bytes,
                                  double* kernel_ms, char* err, uint64_t err_cap) {
    return slimm_bgzf_inflate_with(device, blocks, n_bytes, out, out_cap, out_bytes, kernel_ms, err, err_cap, 0u, nullptr);
}
extern "C" int slimm_bgzf_inflate_with(int device, const uint8_t* blocks, uint64_t n_bytes, uint8_t* out, uint64_t out_cap,
                                       uint64_t* out_bytes, double* kernel_ms, char* err, uint64_t err_cap, uint32_t how,
                                       uint32_t* lane_blocks) {
    if (lane_blocks) *lane_blocks = 0;
    auto fail = [&](int code, const std::string& why) {
        if (err && err_cap) {
            const size_t k = std::min<size_t>(why.size(), err_cap - 1);
            memcpy(err, why.data(), k);
            err[k] = 0;
        }
        return code;
    };
    if (out_bytes) *out_bytes = 0;
    if (kernel_ms) *kernel_ms = 0;
    if ((n_bytes && !blocks) || !out_bytes) return fail(-1, "null argument");
    std::vector<slimm::BgzfBlock> desc;
    uint64_t inflated = 0;
    std::string why;
    if (!slimm::bgzf_parse_blocks(blocks, n_bytes, 0, desc, inflated, why)) return fail(-1, why);
    *out_bytes = inflated;
    if (inflated > out_cap || (inflated && !out)) return fail(-1, "output buffer too small");
    if (desc.empty()) return 0;
    if (desc.size() >= (1ull << 32) || inflated >= (8ull << 30)) return fail(-1, "too many blocks in one call (8 GB of inflated bytes at most)");
    if (hipSetDevice(device) != hipSuccess) return fail(-2, "hipSetDevice failed");
    const uint32_t n = static_cast<uint32_t>(desc.size()), grid = slimm::bgzf_inflate_grid(n);
    uint8_t *d_comp = nullptr, *d_out = nullptr;
    slimm::BgzfBlock* d_desc = nullptr;
    void* d_scratch = nullptr;
    uint32_t* d_status = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = 0;
    std::string msg;
    auto ok = [&](hipError_t e, const char* what) {
        if (e == hipSuccess || rc) return;
        rc = -2;
        msg = std::string(what) + ": " + hipGetErrorString(e);
    };
    ok(hipMalloc(&d_comp, n_bytes + slimm::kBgzfTail), "hipMalloc");
    ok(hipMalloc(&d_out, inflated + 16), "hipMalloc");
    ok(hipMalloc(&d_desc, desc.size() * sizeof(slimm::BgzfBlock)), "hipMalloc");
    ok(hipMalloc(&d_scratch, how == 1u ? slimm::bgzf_lanes_scratch_bytes(grid) : slimm::bgzf_inflate_scratch_bytes(n, desc.back().tok + slimm::bgzf_token_room(desc.back().isize))), "hipMalloc");
    ok(hipMalloc(&d_status, 16), "hipMalloc");
    if (!rc) {
        const uint32_t st0[4] = {0u, 0xffffffffu, 0u, 0u};
        ok(hipMemcpy(d_comp, blocks, n_bytes, hipMemcpyHostToDevice), "hipMemcpy");
        ok(hipMemset(d_comp + n_bytes, 0, slimm::kBgzfTail), "hipMemset");
        ok(hipMemcpy(d_desc, desc.data(), desc.size() * sizeof(slimm::BgzfBlock), hipMemcpyHostToDevice), "hipMemcpy");
        ok(hipMemcpy(d_status, st0, 16, hipMemcpyHostToDevice), "hipMemcpy");
        ok(hipEventCreate(&e0), "hipEventCreate");
        ok(hipEventCreate(&e1), "hipEventCreate");
    }
    if (!rc) {
        ok(hipEventRecord(e0, nullptr), "hipEventRecord");
        if (how == 1u)
            slimm::launch_bgzf_inflate_lanes(nullptr, d_comp, d_desc, n, d_out, d_scratch, grid, d_status, nullptr);
        else
            slimm::launch_bgzf_inflate(nullptr, d_comp, d_desc, n, d_out, d_scratch, d_status);
        ok(hipEventRecord(e1, nullptr), "hipEventRecord");
        ok(hipDeviceSynchronize(), "hipDeviceSynchronize");
        float ms = 0;
        if (!rc && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && kernel_ms) *kernel_ms = ms;
        uint32_t st1[4] = {0, 0, 0, 0};
        ok(hipMemcpy(st1, d_status, 16, hipMemcpyDeviceToHost), "hipMemcpy");
        if (lane_blocks) *lane_blocks = how == 1u ? n : st1[2];
        if (!rc && st1[0]) {
            rc = -1;
            msg = "corrupt BGZF block (device inflate: error " + std::to_string(st1[0]) + " in block " + std::to_string(st1[1]) + ")";
        }
        if (!rc) ok(hipMemcpy(out, d_out, inflated, hipMemcpyDeviceToHost), "hipMemcpy");
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(d_comp);
    (void)hipFree(d_out);
    (void)hipFree(d_desc);
    (void)hipFree(d_scratch);
    (void)hipFree(d_status);
    return rc ? fail(rc, msg) : 0;
}
