// SAM text decoded on the device: what the reference gets from seqan::readRecord on a SAM file (call sites
// src/slimm.hpp:194-208; src/file_helper.hpp:73-75 takes .sam and .bam alike) for the four fields its record loop reads --
// QNAME, FLAG, RNAME (-> the header's reference index), POS -- from windows of the file's text behind the header.
// The counterpart of bam_decode.hip; the window pipeline around it (copies, the incomplete last line carried in front of the
// next window, the record arrays) is the same (context.hip: bam_push_window).
//
//   k_sam_pieces   a LANE per 8 KB piece: the lines that START in the piece -- 16 bytes at a time, newline and tab bytes
//                  found by word arithmetic --, their offsets (bit 31: the line has fewer than ten fields, is empty or starts
//                  with '@': an error of the push, like the host reader's), and where the last of them ends.  A newline is an
//                  exact boundary: nothing is guessed, nothing verified.
//   k_bam_scan     (bam_decode.hip) lines in front of every piece, the window's totals, where its last complete line ends
//   k_sam_decode   a WAVE per piece, a lane per line: the fields; RNAME through a hash table of the header's names (exact:
//                  the bytes are compared); input grouped by name: QNAME compared with the line before (its canonical base:
//                  read_identity.h, Q18) -> run-marked 8-byte records; any other order: the host reader's name hash + check
//                  word.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstring>

#include "kernels.h"

namespace slimm {

namespace {

__device__ __forceinline__ uint64_t ld_u64(const uint8_t* p) {
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
// bytes of w that equal c: bit 7 of every such byte (exact: no false positives with the borrow-free form)
__device__ __forceinline__ uint64_t bytes_equal(uint64_t w, uint8_t c) {
    const uint64_t x = w ^ (0x0101010101010101ull * c);
    const uint64_t lo7 = (x & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full;
    return ~(lo7 | x | 0x7f7f7f7f7f7f7f7full);
}

// A lane per piece.  offs[c * kBamSlots + k] = start of the piece's k-th line (| bit 31: not a record line); pieces[c].stop
// = behind the newline of its last COMPLETE line (a line without its newline in front of `end` is the window's tail);
// pieces[c].count = complete lines.
__global__ __launch_bounds__(64) void k_sam_pieces(const uint8_t* __restrict__ b, uint64_t lo, uint64_t end, uint32_t n_pieces,
                                                   BamPiece* __restrict__ pieces, uint32_t* __restrict__ offs) {
    const uint32_t c = blockIdx.x * 64u + threadIdx.x;
    if (c >= n_pieces) return;
    const uint64_t plo = lo + static_cast<uint64_t>(c) * kSamPiece;
    const uint64_t phi = (c + 1 == n_pieces) ? end : plo + kSamPiece;
    uint32_t* po = offs + static_cast<size_t>(c) * kBamSlots;
    BamPiece pc;
    pc.guess = static_cast<uint32_t>(plo);
    pc.count = 0;
    pc.stop = static_cast<uint32_t>(plo);
    pc.flags = 0;
    // the first line that starts here: at plo when the byte in front is a newline (or plo is the window's first byte)
    uint64_t p = plo;
    bool in_line = false;       // a line that started in this piece is open
    uint32_t start = 0, tabs = 0, n = 0;
    if (c == 0 || b[plo - 1] == '\n') {
        in_line = true;
        start = static_cast<uint32_t>(plo);
    }
    // through the piece -- and beyond it, to the newline of the last line that started in it
    while (p < end && (p < phi || in_line)) {
        const uint64_t left = end - p;
        uint64_t w = ld_u64(b + p);   // (up to 7 bytes behind `end`: the window buffer's slack)
        const uint32_t nb = left < 8 ? static_cast<uint32_t>(left) : 8u;
        uint64_t nl = bytes_equal(w, '\n'), tb = bytes_equal(w, '\t');
        if (nb < 8) {
            const uint64_t keep = (1ull << (8 * nb)) - 1ull;
            nl &= keep;
            tb &= keep;
        }
        if (!nl) {
            tabs += static_cast<uint32_t>(__builtin_popcountll(tb));
            p += nb;
            continue;
        }
        // byte by byte through a word that holds a newline
        for (uint32_t k = 0; k < nb; ++k) {
            const uint64_t at = p + k;
            const uint32_t ch = static_cast<uint32_t>(w >> (8 * k)) & 0xffu;
            if (ch == '\t') {
                ++tabs;
            } else if (ch == '\n') {
                if (in_line) {
                    const uint32_t first = b[start];
                    // (a line that is empty but for a CR -- a blank line of a CR LF file: the host reader strips the CR and skips
                    // the line, so it is the host decoder's like any empty line; a CR behind a record stays in its last field, which
                    // nobody reads.  ADVICE round 5)
                    const bool blank = at == start || (at == start + 1u && first == '\r');
                    const bool record = !blank && first != '@' && tabs >= 9u;   // (the host reader: < 10 fields is an error)
                    if (n < kBamSlots) po[n] = start | (record ? 0u : 0x80000000u);
                    if (!record) pc.flags |= (!blank && first != '@') ? kBamPieceBad : kSamPieceSkip;
                    ++n;
                    pc.stop = static_cast<uint32_t>(at + 1);
                }
                in_line = at + 1 < phi;   // the next line starts in this piece, or in the next
                start = static_cast<uint32_t>(at + 1);
                tabs = 0;
                if (!in_line && at + 1 >= phi) {
                    p = end;   // done
                    break;
                }
            }
        }
        if (p != end) p += nb;
    }
    pc.count = n < kBamSlots ? n : kBamSlots;
    if (n > kBamSlots) pc.flags = kSamPieceSkip;   // (more lines than a piece of records can hold: lines of < 18 bytes, the host reader's to judge)
    pieces[c] = pc;
}

// FNV-1a, 64 bits: the reference-name table's hash (host: sam_name_hash)
__device__ __forceinline__ uint64_t fnv64(const uint8_t* s, uint32_t n) {
    uint64_t h = 0xcbf29ce484222325ull;
    for (uint32_t i = 0; i < n; ++i) {
        h ^= s[i];
        h *= 0x100000001b3ull;
    }
    return h;
}

// host/alignment_file.cpp: hash_read_name / check_read_name, on the device (as in bam_decode.hip)
__device__ uint64_t sam_hash_name(const uint8_t* s, uint32_t n) {
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
__device__ uint32_t sam_check_name(const uint8_t* s, uint32_t n) {
    uint64_t h = 0xcbf29ce484222325ull;
    for (uint32_t i = 0; i < n; ++i) {
        h ^= s[i];
        h *= 0x100000001b3ull;
    }
    return static_cast<uint32_t>(h ^ (h >> 32));
}
__device__ bool sam_same(const uint8_t* a, uint32_t la, const uint8_t* b, uint32_t lb) {
    if (la != lb) return false;
    for (uint32_t i = 0; i < la; ++i)
        if (a[i] != b[i]) return false;
    return true;
}

// the field that starts at p: its length (up to the tab / newline / `end`)
__device__ __forceinline__ uint32_t field_len(const uint8_t* b, uint64_t p, uint64_t end) {
    uint32_t n = 0;
    while (p + n < end && b[p + n] != '\t' && b[p + n] != '\n') ++n;
    return n;
}
// strtol / strtoul base 10 of a field, as the host reader applies them (optional sign, digits; anything else ends it)
__device__ __forceinline__ int64_t field_number(const uint8_t* s, uint32_t n) {
    uint32_t i = 0;
    bool neg = false;
    while (i < n && (s[i] == ' ')) ++i;
    if (i < n && (s[i] == '-' || s[i] == '+')) neg = s[i++] == '-';
    int64_t v = 0;
    for (; i < n && s[i] >= '0' && s[i] <= '9'; ++i) v = v * 10 + (s[i] - '0');
    return neg ? -v : v;
}
// (name, flag) of the line at o: QNAME's canonical base (read_identity.h, Q18) and the flag with the base's mate bit
__device__ __forceinline__ void line_identity(const uint8_t* b, uint64_t o, uint64_t end, const uint8_t*& name, uint32_t& nlen, uint32_t& fl,
                                              uint64_t& after_flag, bool* shortened = nullptr) {
    name = b + o;
    const uint32_t qn = field_len(b, o, end);
    const uint64_t f0 = o + qn + 1;
    const uint32_t fn = field_len(b, f0, end);
    fl = static_cast<uint32_t>(static_cast<uint64_t>(field_number(b + f0, fn))) & 0xffffu;
    nlen = qn;
    if (!(fl & 0xC0u) && nlen >= 2u && name[nlen - 2u] == '.' && (name[nlen - 1u] == '1' || name[nlen - 1u] == '2')) {
        fl |= name[nlen - 1u] == '1' ? 0x40u : 0x80u;
        nlen -= 2u;
        if (shortened) *shortened = true;
    } else if (shortened) {
        *shortened = false;
    }
    after_flag = f0 + fn + 1;
}

// a wave per piece, a lane per line
template <bool kMarked>
__global__ __launch_bounds__(64) void k_sam_decode(const uint8_t* __restrict__ b, uint64_t end, const BamPiece* __restrict__ pieces,
                                                   const uint32_t* __restrict__ offs, uint32_t n_pieces, BamCarry* __restrict__ carry,
                                                   uint64_t out_at, uint64_t* __restrict__ key, int32_t* __restrict__ ref,
                                                   int32_t* __restrict__ pos, uint16_t* __restrict__ flag, uint32_t* __restrict__ check,
                                                   const SamRefEntry* __restrict__ table, uint32_t table_mask, const uint8_t* __restrict__ names) {
    const uint32_t c = blockIdx.x;
    const BamPiece pc = pieces[c];
    const uint32_t* po = offs + static_cast<size_t>(c) * kBamSlots;
    uint32_t prev0 = 0xffffffffu;   // the line in front of this piece's first one
    if (kMarked && pc.count) {
        for (uint32_t d = c; d-- > 0;) {
            const uint32_t n = pieces[d].count;
            if (n) {
                prev0 = offs[static_cast<size_t>(d) * kBamSlots + n - 1] & 0x7fffffffu;
                break;
            }
        }
    }
    uint32_t n_short_starts = 0, n_short_to_plain = 0;
    for (uint32_t k = threadIdx.x; k < pc.count; k += 64u) {
        const uint64_t o = po[k] & 0x7fffffffu;
        const uint8_t* name;
        uint32_t nlen, fl;
        uint64_t p;
        bool is_short;
        line_identity(b, o, end, name, nlen, fl, p, &is_short);
        // RNAME -> the header's index (the host reader: "*" and names the header does not have are -1)
        const uint32_t rn = field_len(b, p, end);
        int32_t rid = -1;
        if (!(rn == 1u && b[p] == '*') && table_mask) {
            const uint64_t h = fnv64(b + p, rn);
            for (uint32_t slot = static_cast<uint32_t>(h) & table_mask;; slot = (slot + 1u) & table_mask) {
                const SamRefEntry e = table[slot];
                if (e.ref < 0) break;
                if (e.hash == h && sam_same(b + p, rn, names + e.name_off, e.name_len)) {
                    rid = e.ref;
                    break;
                }
            }
        }
        p += rn + 1;
        const uint32_t pn = field_len(b, p, end);
        const int32_t rpos = static_cast<int32_t>(field_number(b + p, pn) - 1);   // SAM POS is 1-based; 0 ("unavailable") becomes -1
        const uint64_t at = out_at + pc.base + k;
        if (kMarked) {
            bool starts, prev_short;
            const uint32_t po_prev = k ? (po[k - 1] & 0x7fffffffu) : prev0;
            if (po_prev != 0xffffffffu) {
                const uint8_t* qname;
                uint32_t qlen, qfl;
                uint64_t unused;
                line_identity(b, po_prev, end, qname, qlen, qfl, unused, &prev_short);
                starts = !sam_same(name, nlen, qname, qlen);
            } else {
                starts = !(carry->have && sam_same(name, nlen, carry->name, carry->len));
                prev_short = carry->last_short != 0u;
            }
            // Q18 on a grouped stream (kernels.h: BamCarry), as in k_bam_decode
            n_short_starts += (is_short & starts) ? 1u : 0u;
            n_short_to_plain += (!is_short & !starts & prev_short) ? 1u : 0u;
            const uint32_t mate = (fl & 0x40u) ? 1u : ((fl & 0x80u) ? 2u : 0u);
            const bool mapped = !(fl & 0x4u) && rid != -1;
            const uint32_t r1 = mapped ? min(static_cast<uint32_t>(rid) + 1u, 0x1fffffffu) : 0u;
            reinterpret_cast<uint32_t*>(ref)[at] = r1 | (mate << 29) | (starts ? 0x80000000u : 0u);
            pos[at] = rpos;
        } else {
            key[at] = sam_hash_name(name, nlen);
            ref[at] = rid;
            pos[at] = rpos;
            flag[at] = static_cast<uint16_t>(fl);
            check[at] = sam_check_name(name, nlen);
        }
    }
    if (kMarked && __any(static_cast<int>(n_short_starts | n_short_to_plain))) {
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

// the window's last line's name (its canonical base) -> the carry
__global__ __launch_bounds__(64) void k_sam_carry(const uint8_t* __restrict__ b, uint64_t end, const BamPiece* __restrict__ pieces,
                                                  const uint32_t* __restrict__ offs, const BamWindowResult* __restrict__ res,
                                                  BamCarry* __restrict__ carry) {
    const uint32_t c = res->last_piece;
    if (c == 0xffffffffu) return;
    const uint64_t o = offs[static_cast<size_t>(c) * kBamSlots + pieces[c].count - 1u] & 0x7fffffffu;
    const uint8_t* name;
    uint32_t nlen, fl;
    uint64_t unused;
    bool is_short;
    line_identity(b, o, end, name, nlen, fl, unused, &is_short);
    if (nlen > 255u) nlen = 255u;   // (QNAME is at most 254 characters)
    for (uint32_t i = threadIdx.x; i < nlen; i += 64u) carry->name[i] = name[i];
    if (threadIdx.x == 0) {
        carry->len = nlen;
        carry->have = 1;
        carry->last_short = is_short ? 1u : 0u;
    }
}

}  // namespace

uint32_t sam_pieces(uint64_t n_bytes) { return static_cast<uint32_t>((n_bytes + kSamPiece - 1) / kSamPiece); }

uint64_t sam_name_hash(const char* s, size_t n) {
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < n; ++i) {
        h ^= static_cast<unsigned char>(s[i]);
        h *= 0x100000001b3ull;
    }
    return h;
}

void launch_sam_find(hipStream_t st, const uint8_t* bytes, uint64_t lo, uint64_t end, BamPiece* pieces, uint32_t* offs, BamWindowResult* result) {
    const uint32_t np = sam_pieces(end - lo);
    if (np) hipLaunchKernelGGL(k_sam_pieces, dim3((np + 63u) / 64u), dim3(64), 0, st, bytes, lo, end, np, pieces, offs);
    launch_bam_scan(st, pieces, np, end, result, static_cast<uint32_t>(lo));
}

void launch_sam_decode(hipStream_t st, const uint8_t* bytes, uint64_t lo, uint64_t end, const BamPiece* pieces, const uint32_t* offs,
                       BamCarry* carry, const BamWindowResult* result, bool marked, uint64_t out_at, uint64_t* key, int32_t* ref, int32_t* pos,
                       uint16_t* flag, uint32_t* check, const SamRefEntry* table, uint32_t table_mask, const uint8_t* names) {
    const uint32_t np = sam_pieces(end - lo);
    if (!np) return;
    if (marked)
        hipLaunchKernelGGL(k_sam_decode<true>, dim3(np), dim3(64), 0, st, bytes, end, pieces, offs, np, carry, out_at, key, ref, pos, flag, check,
                           table, table_mask, names);
    else
        hipLaunchKernelGGL(k_sam_decode<false>, dim3(np), dim3(64), 0, st, bytes, end, pieces, offs, np, carry, out_at, key, ref, pos, flag, check,
                           table, table_mask, names);
    hipLaunchKernelGGL(k_sam_carry, dim3(1), dim3(64), 0, st, bytes, end, pieces, offs, result, carry);
}

}  // namespace slimm
