// Device-side decoding of BAM alignment records (gfx950, wave64): from the BGZF-inflated bytes of a BAM file to the record
// stream of the alignment-to-profile path.
//
// Replaces, for the `slimm` command on MI355X, what the reference does through seqan::BamFileIn record by record
// (call sites src/misc.hpp:498-522, src/slimm.hpp:194-208): of every alignment record the path reads refID, the 0-based
// position, the flag word and the read name -- 10 bytes and a name of a 200-byte record.  The host walked every inflated
// byte three times for that (record boundaries, field decode + name hash, adjacent-name check: 1.4 s of a 1.9 s run on
// 100 M records under a 16-core quota); the inflated windows cross PCIe in 0.4 s, overlapped with the inflate, and the
// three walks become four small kernels per window:
//
//   k_bam_pieces   the window is cut into pieces of 16 KB, a LANE per piece: where does the first record of the piece
//                  start?  -- GUESSED from the bytes (a plausible header whose two successors are plausible too, like the
//                  host reader's find_records), then the piece is walked record by record (a chain of dependent 4-byte
//                  loads -- thousands of chains at once) and the record offsets are noted in the piece's own slots
//   k_bam_verify   a guess is right iff the piece before it ends exactly there; a wrong or missing guess (a record
//                  longer than a piece, a header look-alike in a sequence) is walked again from the true start, serially
//                  -- rare; the all-pieces-right case is one compare per piece
//   k_bam_scan     records in front of every piece, the window's totals, where its last complete record ends
//   k_bam_decode   a WAVE per piece, a lane per record: fields out; input grouped by name: the name compared with the name
//                  of the record before (across pieces and windows) -> run-marked 8-byte records (slimm_mark_word); any
//                  other order: the 62-bit name hash and the check word of the host reader (host/alignment_file.cpp:
//                  hash_read_name, check_read_name -- the same functions, so both readers give a file the same keys)
//
// The bytes behind the last complete record of a window (an incomplete record) are carried in front of the next window.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstring>

#include "kernels.h"

namespace slimm {

namespace {

__device__ __forceinline__ uint32_t ld_u32(const uint8_t* p) {
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
__device__ __forceinline__ uint64_t ld_u64(const uint8_t* p) {
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
__device__ __forceinline__ uint32_t ld_u16(const uint8_t* p) { return static_cast<uint32_t>(p[0]) | (static_cast<uint32_t>(p[1]) << 8); }

// Does a BAM record plausibly start at b[o]?  (host/alignment_file.cpp: plausible_record -- the same tests)
__device__ bool bam_plausible(const uint8_t* b, uint64_t o, uint64_t end, uint32_t n_refs, int depth) {
    for (;;) {
        if (o + 36 > end) return false;
        const uint8_t* r = b + o;
        const uint32_t bs = ld_u32(r);
        const int32_t ref = static_cast<int32_t>(ld_u32(r + 4)), pos = static_cast<int32_t>(ld_u32(r + 8));
        const uint32_t l_name = r[12], n_cigar = ld_u16(r + 16), l_seq = ld_u32(r + 20);
        const int32_t nref = static_cast<int32_t>(ld_u32(r + 24)), npos = static_cast<int32_t>(ld_u32(r + 28));
        const int32_t nrefs = static_cast<int32_t>(n_refs);
        if (bs < 32 || bs > (1u << 24) || ref < -1 || ref >= nrefs || pos < -1 || l_name == 0 || l_seq > (1u << 28)) return false;
        if (nref < -1 || nref >= nrefs || npos < -1) return false;
        if (32ull + l_name + 4ull * n_cigar + (static_cast<uint64_t>(l_seq) + 1) / 2 + l_seq > bs) return false;
        if (o + 36 + l_name <= end && b[o + 36 + l_name - 1] != 0) return false;  // the name ends with NUL
        if (depth <= 0 || o + 4 + bs + 36 > end) return true;
        o += 4 + static_cast<uint64_t>(bs);
        --depth;
    }
}

// Walks the records whose START lies in [from, hi), noting up to `slots` offsets (relative to the window's base); stops at
// the first record that does not fit in front of `end`.  Returns where it stopped; bad: a malformed record.
__device__ uint64_t bam_walk(const uint8_t* b, uint64_t from, uint64_t hi, uint64_t end, uint32_t* offs, uint32_t slots, uint32_t& count,
                             bool& bad) {
    uint64_t p = from;
    uint32_t n = 0;
    bad = false;
    while (p < hi) {
        if (end - p < 4) break;
        const uint32_t bs = ld_u32(b + p);
        if (bs < 32) {
            bad = true;
            break;
        }
        if (end - p < 4 + static_cast<uint64_t>(bs)) break;
        if (32u + b[p + 12] > bs || n >= slots) {
            bad = true;
            break;
        }
        if (offs) offs[n] = static_cast<uint32_t>(p);
        ++n;
        p += 4 + static_cast<uint64_t>(bs);
    }
    count = n;
    return p;
}

// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_bam_pieces(const uint8_t* __restrict__ b, uint64_t lo, uint64_t end, uint32_t n_pieces,
                                                   uint32_t n_refs, BamPiece* __restrict__ pieces, uint32_t* __restrict__ offs) {
    const uint32_t c = blockIdx.x * 64u + threadIdx.x;
    if (c >= n_pieces) return;
    const uint64_t plo = lo + static_cast<uint64_t>(c) * kBamPiece;
    const uint64_t phi = (c + 1 == n_pieces) ? end : plo + kBamPiece;
    BamPiece pc;
    pc.guess = 0xffffffffu;
    pc.count = 0;
    pc.stop = static_cast<uint32_t>(plo);
    pc.flags = 0;
    uint64_t from = plo;
    bool guessed = c == 0;
    if (c != 0) {
        for (uint64_t o = plo; o < phi; ++o)
            if (bam_plausible(b, o, end, n_refs, 2)) {
                guessed = true;
                from = o;
                break;
            }
    }
    if (guessed) {
        bool bad;
        uint32_t n;
        const uint64_t stop = bam_walk(b, from, phi, end, offs + static_cast<size_t>(c) * kBamSlots, kBamSlots, n, bad);
        pc.guess = static_cast<uint32_t>(from);
        pc.count = n;
        pc.stop = static_cast<uint32_t>(stop);
        pc.flags = bad ? kBamPieceBad : 0u;
    }
    pieces[c] = pc;
}

// one workgroup: every guess against the end of the piece before it; what does not fit is walked again from there
__global__ __launch_bounds__(1024) void k_bam_verify(const uint8_t* __restrict__ b, uint64_t lo, uint64_t end, uint32_t n_pieces,
                                                     BamPiece* __restrict__ pieces, uint32_t* __restrict__ offs) {
    __shared__ uint32_t s_wrong;
    if (threadIdx.x == 0) s_wrong = 0;
    __syncthreads();
    uint32_t wrong = 0;
    for (uint32_t c = 1 + threadIdx.x; c < n_pieces; c += 1024u) wrong += pieces[c].guess != pieces[c - 1].stop ? 1u : 0u;
    if (wrong) atomicAdd(&s_wrong, wrong);
    __syncthreads();
    if (s_wrong == 0 || threadIdx.x >= 64u) return;
    // The other path, ONE WAVE: the chain of true starts from the first piece on.  The wave takes 64 pieces at a time (one
    // coalesced load instead of a round trip per piece) and goes through them in order with wave-uniform state: a piece
    // whose guess is where the chain arrives is left as it is; any other is walked again from there (every lane walks,
    // lane 0 notes the offsets), or -- the chain being behind the piece already: a record longer than a piece -- simply
    // holds no record.  Files of long reads, where that is every piece, therefore cost a few scalar instructions per piece.
    const uint32_t lane = threadIdx.x;
    uint32_t cur = pieces[0].stop;
    bool stuck = (pieces[0].flags & kBamPieceBad) != 0;  // (nothing behind a malformed record can be trusted)
    bool ended = !stuck && n_pieces > 1 && cur < lo + kBamPiece;  // the window's incomplete last record starts in piece 0
    for (uint32_t c0 = 1; c0 < n_pieces; c0 += 64u) {
        const uint32_t c = c0 + lane;
        BamPiece pc = pieces[min(c, n_pieces - 1u)];
        bool touched = false;
        const uint32_t n_here = min(64u, n_pieces - c0);
        for (uint32_t l = 0; l < n_here; ++l) {
            const uint32_t ci = c0 + l;
            const uint32_t guess = __builtin_amdgcn_readlane(pc.guess, l), stop = __builtin_amdgcn_readlane(pc.stop, l);
            const uint32_t flags = __builtin_amdgcn_readlane(pc.flags, l);
            const uint64_t plo = lo + static_cast<uint64_t>(ci) * kBamPiece;
            const uint64_t phi = (ci + 1 == n_pieces) ? end : plo + kBamPiece;
            uint32_t n_guess = guess, n_stop = stop, n_count = __builtin_amdgcn_readlane(pc.count, l), n_flags = flags;
            bool change = false;
            if (ended) {
                n_count = 0, n_stop = cur, n_guess = cur, n_flags = 0, change = true;
            } else if (stuck) {
                n_count = 0, n_stop = cur, n_guess = cur, n_flags = kBamPieceBad, change = true;
            } else if (guess != cur) {
                n_guess = cur, n_flags = 0, change = true;
                if (cur < phi) {
                    bool bad;
                    uint32_t n;
                    // (all lanes walk the same chain; the offsets are noted once)
                    uint32_t* po = offs + static_cast<size_t>(ci) * kBamSlots;
                    n_stop = static_cast<uint32_t>(bam_walk(b, cur, phi, end, lane == 0u ? po : nullptr, kBamSlots, n, bad));
                    n_count = n;
                    if (bad) n_flags = kBamPieceBad;
                } else {  // (a record that runs over the whole piece)
                    n_count = 0, n_stop = cur;
                }
            }
            if (change) {
                pc.guess = lane == l ? n_guess : pc.guess;
                pc.stop = lane == l ? n_stop : pc.stop;
                pc.count = lane == l ? n_count : pc.count;
                pc.flags = lane == l ? n_flags : pc.flags;
                touched = touched | (lane == l);
            }
            if (n_flags & kBamPieceBad) stuck = true;
            cur = n_stop;
            // a walk that stopped inside its piece met the window's incomplete last record: nothing behind it is a record
            if (!stuck && !ended && cur < phi) ended = true;
        }
        if (touched && c < n_pieces) pieces[c] = pc;
    }
}

// one workgroup: records in front of every piece (pieces[c].base), the window's result
__global__ __launch_bounds__(1024) void k_bam_scan(BamPiece* __restrict__ pieces, uint32_t n_pieces, uint64_t end,
                                                   BamWindowResult* __restrict__ out, uint32_t sam_lo) {
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_bad;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0) s_bad = 0;
    __syncthreads();
    uint32_t carry = 0;
    for (uint32_t c0 = 0; c0 < n_pieces; c0 += 1024u) {
        const uint32_t c = c0 + tid;
        const uint32_t v = c < n_pieces ? pieces[c].count : 0u;
        if (c < n_pieces && (pieces[c].flags & (kBamPieceBad | kSamPieceSkip))) atomicOr(&s_bad, pieces[c].flags & (kBamPieceBad | kSamPieceSkip));
        uint32_t inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t a = __shfl_up(inc, o, 64);
            if (lane >= static_cast<uint32_t>(o)) inc += a;
        }
        if (lane == 63u) s_w[wave] = inc;
        __syncthreads();
        uint32_t before = carry, total = 0;
#pragma unroll
        for (uint32_t w = 0; w < 16; ++w) {
            const uint32_t t = s_w[w];
            before += w < wave ? t : 0u;
            total += t;
        }
        if (c < n_pieces) pieces[c].base = before + inc - v;
        carry += total;
        __syncthreads();
    }
    if (tid == 0) {
        // the last piece holding a record: its last record is the one the next window's first name is compared with
        uint32_t lastc = 0xffffffffu;
        for (uint32_t c = n_pieces; c-- > 0;)
            if (pieces[c].count) {
                lastc = c;
                break;
            }
        out->n_records = carry;
        if (sam_lo != 0xffffffffu)   // (SAM text: where the last complete line ends)
            out->stop = lastc != 0xffffffffu ? pieces[lastc].stop : sam_lo;
        else
            out->stop = n_pieces ? pieces[n_pieces - 1].stop : static_cast<uint32_t>(end);
        out->bad = s_bad;
        out->last_piece = lastc;
    }
}

// host/alignment_file.cpp: hash_read_name / check_read_name, on the device
__device__ uint64_t bam_hash_name(const uint8_t* s, uint32_t n) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (static_cast<uint64_t>(n) * 0xff51afd7ed558ccdULL);
    uint32_t i = 0;
    for (; i + 8 <= n; i += 8) {
        h ^= ld_u64(s + i);
        h *= 0xff51afd7ed558ccdULL;
        h ^= h >> 32;
    }
    uint64_t tail = 0;
    for (uint32_t k = 0; i + k < n; ++k) tail |= static_cast<uint64_t>(s[i + k]) << (8 * k);
    h ^= tail;
    h *= 0xc4ceb9fe1a85ec53ULL;
    h ^= h >> 29;
    h *= 0xff51afd7ed558ccdULL;
    h ^= h >> 32;
    return h >> 2;
}
__device__ uint32_t bam_check_name(const uint8_t* s, uint32_t n) {
    uint64_t h = 0xcbf29ce484222325ull;
    for (uint32_t i = 0; i < n; ++i) {
        h ^= s[i];
        h *= 0x100000001b3ull;
    }
    return static_cast<uint32_t>(h ^ (h >> 32));
}
__device__ bool bam_same_name(const uint8_t* a, uint32_t la, const uint8_t* b, uint32_t lb) {
    if (la != lb) return false;
    uint32_t i = 0;
    for (; i + 8 <= la; i += 8)
        if (ld_u64(a + i) != ld_u64(b + i)) return false;
    for (; i < la; ++i)
        if (a[i] != b[i]) return false;
    return true;
}

// Q18 (read_identity.h: canonical_read, on the device): the reference keys a read by the STRING qName + ".1" / ".2" / nothing
// (src/slimm.hpp:204-208); the identity is the canonical (base, mate) of that string.  Shortens nlen to the base and
// returns the flag with the mate bit the base carries.
__device__ __forceinline__ uint32_t bam_canonical(const uint8_t* name, uint32_t& nlen, uint32_t fl) {
    if (fl & 0xC0u) return fl;
    if (nlen >= 2u && name[nlen - 2u] == '.') {
        const uint32_t d = name[nlen - 1u];
        if (d == '1' || d == '2') {
            nlen -= 2u;
            return fl | (d == '1' ? 0x40u : 0x80u);
        }
    }
    return fl;
}

// a wave per piece, a lane per record
template <bool kMarked>
__global__ __launch_bounds__(64) void k_bam_decode(const uint8_t* __restrict__ b, const BamPiece* __restrict__ pieces,
                                                   const uint32_t* __restrict__ offs, uint32_t n_pieces, BamCarry* __restrict__ carry,
                                                   const BamWindowResult* __restrict__ res, uint64_t out_at, uint64_t* __restrict__ key,
                                                   int32_t* __restrict__ ref, int32_t* __restrict__ pos, uint16_t* __restrict__ flag,
                                                   uint32_t* __restrict__ check) {
    const uint32_t c = blockIdx.x;
    const BamPiece pc = pieces[c];
    const uint32_t* po = offs + static_cast<size_t>(c) * kBamSlots;
    // the record in front of this piece's first one: the last record of the nearest piece before it that has any; none in
    // this window: the name carried over from the window before (carry->have == 0: the file's first record)
    uint32_t prev0 = 0xffffffffu;
    if (kMarked && pc.count) {
        for (uint32_t d = c; d-- > 0;) {
            const uint32_t n = pieces[d].count;
            if (n) {
                prev0 = offs[static_cast<size_t>(d) * kBamSlots + n - 1];
                break;
            }
        }
    }
    uint32_t n_short_starts = 0, n_short_to_plain = 0;
    for (uint32_t k = threadIdx.x; k < pc.count; k += 64u) {
        const uint8_t* r = b + po[k] + 4;
        const int32_t rid = static_cast<int32_t>(ld_u32(r));
        const int32_t rpos = static_cast<int32_t>(ld_u32(r + 4));
        const uint32_t l_name = r[8];
        const uint8_t* name = r + 32;
        uint32_t nlen = l_name ? l_name - 1u : 0u;
        const uint32_t fl0 = ld_u16(r + 14);
        const uint32_t fl = bam_canonical(name, nlen, fl0);
        const uint64_t at = out_at + pc.base + k;
        if (kMarked) {
            bool starts, prev_short;
            const uint32_t po_prev = k ? po[k - 1] : prev0;
            if (po_prev != 0xffffffffu) {
                const uint8_t* q = b + po_prev + 4;
                uint32_t qlen = q[8] ? q[8] - 1u : 0u;
                const uint32_t qfl0 = ld_u16(q + 14);
                prev_short = bam_canonical(q + 32, qlen, qfl0) != qfl0;
                starts = !bam_same_name(name, nlen, q + 32, qlen);
            } else {
                starts = !(carry->have && bam_same_name(name, nlen, carry->name, carry->len));
                prev_short = carry->last_short != 0u;
            }
            // Q18 on a grouped stream (kernels.h: BamCarry): runs that start with a shortened name, shortened -> plain steps
            // inside a run -- counted per lane, added up per piece behind the loop
            const bool is_short = fl != fl0;
            n_short_starts += (is_short & starts) ? 1u : 0u;
            n_short_to_plain += (!is_short & !starts & prev_short) ? 1u : 0u;
            // slimm_mark_word (context.hip): reference + 1 (0: not mapped) | mate << 29 | starts a qName run << 31
            const uint32_t mate = (fl & 0x40u) ? 1u : ((fl & 0x80u) ? 2u : 0u);
            const bool mapped = !(fl & 0x4u) && rid != -1;
            const uint32_t r1 = mapped ? min(static_cast<uint32_t>(rid) + 1u, 0x1fffffffu) : 0u;
            reinterpret_cast<uint32_t*>(ref)[at] = r1 | (mate << 29) | (starts ? 0x80000000u : 0u);
            pos[at] = rpos;
        } else {
            key[at] = bam_hash_name(name, nlen);
            ref[at] = rid;
            pos[at] = rpos;
            flag[at] = static_cast<uint16_t>(fl);
            check[at] = bam_check_name(name, nlen);
        }
    }
    if (kMarked && __any(static_cast<int>(n_short_starts | n_short_to_plain))) {   // (a file without such names: one vote per piece)
        for (uint32_t d = 32u; d; d >>= 1) {
            n_short_starts += static_cast<uint32_t>(__shfl_xor(static_cast<int>(n_short_starts), static_cast<int>(d)));
            n_short_to_plain += static_cast<uint32_t>(__shfl_xor(static_cast<int>(n_short_to_plain), static_cast<int>(d)));
        }
        if (threadIdx.x == 0) {
            if (n_short_starts) atomicAdd(&carry->short_starts, n_short_starts);
            if (n_short_to_plain) atomicAdd(&carry->short_to_plain, n_short_to_plain);
        }
    }
}

// the window's last record's name -> the carry (after k_bam_decode of the same window has read the old one)
__global__ __launch_bounds__(64) void k_bam_carry(const uint8_t* __restrict__ b, const BamPiece* __restrict__ pieces,
                                                  const uint32_t* __restrict__ offs, const BamWindowResult* __restrict__ res,
                                                  BamCarry* __restrict__ carry) {
    const uint32_t c = res->last_piece;
    if (c == 0xffffffffu) return;  // (no record in this window: the carried name stays)
    const uint32_t o = offs[static_cast<size_t>(c) * kBamSlots + pieces[c].count - 1u];
    const uint8_t* r = b + o + 4;
    uint32_t nlen = r[8] ? r[8] - 1u : 0u;
    const uint32_t fl0 = ld_u16(r + 14);
    const uint32_t fl = bam_canonical(r + 32, nlen, fl0);  // the carried name is the canonical base
    for (uint32_t i = threadIdx.x; i < nlen; i += 64u) carry->name[i] = r[32 + i];
    if (threadIdx.x == 0) {
        carry->len = nlen;
        carry->have = 1;
        carry->last_short = fl != fl0 ? 1u : 0u;
    }
}

}  // namespace

uint32_t bam_pieces(uint64_t n_bytes) { return static_cast<uint32_t>((n_bytes + kBamPiece - 1) / kBamPiece); }

void launch_bam_find(hipStream_t st, const uint8_t* bytes, uint64_t lo, uint64_t end, uint32_t n_refs, BamPiece* pieces, uint32_t* offs,
                     BamWindowResult* result) {
    const uint32_t np = bam_pieces(end - lo);
    if (np) {
        hipLaunchKernelGGL(k_bam_pieces, dim3((np + 63u) / 64u), dim3(64), 0, st, bytes, lo, end, np, n_refs, pieces, offs);
        hipLaunchKernelGGL(k_bam_verify, dim3(1), dim3(1024), 0, st, bytes, lo, end, np, pieces, offs);
    }
    hipLaunchKernelGGL(k_bam_scan, dim3(1), dim3(1024), 0, st, pieces, np, end, result, 0xffffffffu);
}

void launch_bam_scan(hipStream_t st, BamPiece* pieces, uint32_t n_pieces, uint64_t end, BamWindowResult* result, uint32_t sam_lo) {
    hipLaunchKernelGGL(k_bam_scan, dim3(1), dim3(1024), 0, st, pieces, n_pieces, end, result, sam_lo);
}

void launch_bam_decode(hipStream_t st, const uint8_t* bytes, uint64_t lo, uint64_t end, const BamPiece* pieces, const uint32_t* offs,
                       BamCarry* carry, const BamWindowResult* result, bool marked, uint64_t out_at, uint64_t* key, int32_t* ref,
                       int32_t* pos, uint16_t* flag, uint32_t* check) {
    const uint32_t np = bam_pieces(end - lo);
    if (!np) return;
    if (marked)
        hipLaunchKernelGGL(k_bam_decode<true>, dim3(np), dim3(64), 0, st, bytes, pieces, offs, np, carry, result, out_at, key, ref, pos,
                           flag, check);
    else
        hipLaunchKernelGGL(k_bam_decode<false>, dim3(np), dim3(64), 0, st, bytes, pieces, offs, np, carry, result, out_at, key, ref, pos,
                           flag, check);
    hipLaunchKernelGGL(k_bam_carry, dim3(1), dim3(64), 0, st, bytes, pieces, offs, result, carry);
}

}  // namespace slimm
