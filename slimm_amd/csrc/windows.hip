// The window pipeline (include/slimm_hip.h: slimm_push_bam_bytes, slimm_push_bgzf_blocks, slimm_push_sam_bytes): a file's
// bytes -- inflated BAM records, whole BGZF blocks, SAM text -- cross the bus in windows; the device inflates (bgzf_tokens.hip,
// bgzf_inflate.hip), finds the records (bam_decode.hip, sam_decode.hip) and appends them to the context's record stream.
// Replaces seqan::BamFileIn + readRecord of the reference (src/misc.hpp:498-522, src/slimm.hpp:194-208).
#include "context.h"

namespace slimm {
// the Q18 run counts of the device decoders (every window launched so far), from the carry block
int bam_fetch_q18(slimm_ctx* c) {
    slimm_ctx::BamDecode& B = c->bam;
    if (!B.carry.p) return SLIMM_OK;
    (void)hipSetDevice(c->device);
    uint32_t w[2] = {0, 0};
    HIP_TRY(c, hipMemcpyAsync(w, &B.carry.p->short_starts, sizeof(w), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    B.q18_starts = w[0];
    B.q18_plain = w[1];
    return SLIMM_OK;
}
}  // namespace slimm

extern "C" {

// A caller's host buffer page-locked for the life of the context: the DMA engine then reads it directly (slimm_push_bam_bytes
// and the *_async pushes take any host memory, at the speed of the runtime's own staging when it is pageable)
namespace {
void push_trace(const char* fmt, ...);
}
int slimm_pin_host_buffer(slimm_ctx* c, const void* p, uint64_t n_bytes) {
    if (!c || !p || !n_bytes) return SLIMM_E_INVALID;
    if (c->device < 0) return fail(c, SLIMM_E_INVALID, "host-only context");
    (void)hipSetDevice(c->device);
    const uint8_t* b = static_cast<const uint8_t*>(p);
    for (auto& r : c->bam.registered)
        if (b >= r.first && b + n_bytes <= r.first + r.second) return SLIMM_OK;
    const auto t0 = std::chrono::steady_clock::now();
    if (hipHostRegister(const_cast<uint8_t*>(b), n_bytes, hipHostRegisterDefault) != hipSuccess) {
        (void)hipGetLastError();
        return fail(c, SLIMM_E_HIP, "hipHostRegister failed");
    }
    push_trace("page-locked %.0f MB in %.2f ms", n_bytes / 1e6, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    c->bam.registered.emplace_back(b, static_cast<size_t>(n_bytes));
    return SLIMM_OK;
}

// BAM alignment records decoded on the device (include/slimm_hip.h; kernels: bam_decode.hip).
// A window is copied when it is pushed and WORKED ON when the next one is pushed (or at once, when it is the last): its
// host-to-device copy then runs beside the kernels and the host's bookkeeping of the window before it -- the copies are
// what bounds this path (192 MB at 54 GB/s: 3.6 ms; kernels + one synchronisation per window: 0.8 ms).
namespace {
// window j (n bytes, in buffer j % kBamRing) -> records appended; the incomplete record at its end goes in front of window j + 1
int bam_finish_window(slimm_ctx* c, uint64_t j, uint64_t n_bytes, bool is_last, uint64_t& n_rec_out) {
    slimm_ctx::BamDecode& B = c->bam;
    hipStream_t st = c->stream;
    const bool marked = c->order == SLIMM_ORDER_GROUPED;
    const uint32_t b = static_cast<uint32_t>(j % slimm_ctx::kBamRing), nb = static_cast<uint32_t>((j + 1u) % slimm_ctx::kBamRing);
    const uint64_t lo = kBamSlack - B.carry_bytes, end = kBamSlack + n_bytes;
    const uint32_t np = B.sam ? sam_pieces(end - lo) : bam_pieces(end - lo);
    // (with room to spare and without a hipFree: windows differ by a few pieces, and a hipFree waits for the inflate kernels
    // of the windows behind this one)
    if (B.pieces.cap < static_cast<size_t>(np) + 1) HIP_TRY(c, B.pieces.ensure_later(static_cast<size_t>(np) + (np >> 2) + 64, B.outgrown));
    if (B.offs.cap < static_cast<size_t>(np + 1) * kBamSlots)
        HIP_TRY(c, B.offs.ensure_later((static_cast<size_t>(np) + (np >> 2) + 64) * kBamSlots, B.outgrown));
    if (n_bytes) HIP_TRY(c, hipStreamWaitEvent(st, B.copied[b], 0));
    const bool inflated_here = n_bytes && B.inflated[b];
    if (inflated_here)
        HIP_TRY(c, hipMemcpyAsync(B.h_inflate_status.p, B.inflate_status.p + 4u * b, 16, hipMemcpyDeviceToHost, st));
    if (B.sam)
        launch_sam_find(st, B.bytes[b].p, lo, end, B.pieces.p, B.offs.p, B.result.p);
    else
        launch_bam_find(st, B.bytes[b].p, lo, end, c->R, B.pieces.p, B.offs.p, B.result.p);
    HIP_TRY(c, hipStreamSynchronize(st));  // the window is on the device and counted
    if (inflated_here && B.h_inflate_status.p[0])
        return fail(c, SLIMM_E_INVALID, "corrupt BGZF block (device inflate: error %u in block %u of the window)", B.h_inflate_status.p[0],
                    B.h_inflate_status.p[1]);
    const BamWindowResult res = *B.result.p;
    if (res.bad && B.sam)
        return fail(c, SLIMM_E_INVALID, (res.bad & kBamPieceBad) ? "SAM line with fewer than 10 fields"
                                                                : "a header line or an empty line among the alignment lines: decode this file on the host");
    if (res.bad) return fail(c, SLIMM_E_INVALID, "bad BAM record");
    const uint64_t n_rec = res.n_records, stop = np ? res.stop : end;
    const uint64_t tail = end - stop;
    if (tail > kBamSlack)
        return fail(c, SLIMM_E_INVALID, B.sam ? "a SAM line longer than 16 MiB: decode this file on the host"
                                              : "a BAM record longer than 16 MiB: decode this file on the host");
    if (is_last && tail) return fail(c, SLIMM_E_INVALID, "truncated BAM record");
    if (c->n_pushed + n_rec >= 0x7fffffffull) return fail(c, SLIMM_E_INVALID, "a context handles fewer than 2^31 records; shard the stream");
    int rc = slimm_reserve(c, c->n_pushed + n_rec);
    if (rc != SLIMM_OK) return rc;
    if (!marked) {  // (the four-array form's flag and check arrays appear at a file's first window)
        const uint64_t want = c->n_pushed + n_rec;
        if (c->in_flag.cap < want) {
            if (c->n_pushed) return fail(c, SLIMM_E_HIP, "record arrays out of step");
            HIP_TRY(c, c->in_flag.ensure_later(std::max<uint64_t>(want, c->in_ref.cap), B.outgrown));
        }
        if (c->in_check.cap < want) {
            if (c->n_pushed) return fail(c, SLIMM_E_HIP, "record arrays out of step");
            HIP_TRY(c, c->in_check.ensure_later(std::max<uint64_t>(want, c->in_ref.cap), B.outgrown));
        }
    }
    if (B.sam)
        launch_sam_decode(st, B.bytes[b].p, lo, end, B.pieces.p, B.offs.p, B.carry.p, B.result.p, marked, c->n_pushed, c->in_key.p,
                          c->in_ref.p, c->in_pos.p, c->in_flag.p, c->in_check.p, B.sam_table.p, B.sam_mask, B.sam_names.p);
    else
        launch_bam_decode(st, B.bytes[b].p, lo, end, B.pieces.p, B.offs.p, B.carry.p, B.result.p, marked, c->n_pushed, c->in_key.p,
                          c->in_ref.p, c->in_pos.p, c->in_flag.p, c->in_check.p);
    if (tail) {  // the incomplete record goes in front of the next window (whose own bytes may be on their way already)
        if (B.bytes[nb].cap < kBamSlack + 64) HIP_TRY(c, B.bytes[nb].ensure(kBamSlack + 64));
        HIP_TRY(c, hipMemcpyAsync(B.bytes[nb].p + kBamSlack - tail, B.bytes[b].p + stop, tail, hipMemcpyDeviceToDevice, st));
    }
    HIP_TRY(c, hipGetLastError());
    B.carry_bytes = tail;
    c->n_pushed += n_rec;
    c->rec = DeviceRecords();
    c->rec.ref = c->in_ref.p;
    c->rec.pos = c->in_pos.p;
    c->rec.n = static_cast<uint32_t>(c->n_pushed);
    if (marked) {
        c->rec.marked = true;
    } else {
        c->rec.key = c->in_key.p;
        c->rec.flag = c->in_flag.p;
        c->rec.check = c->in_check.p;
    }
    n_rec_out = n_rec;
    return SLIMM_OK;
}
}  // namespace

namespace {
enum { kFormatBam = 0, kFormatBgzf = 1, kFormatSam = 2 };
int bam_push_window(slimm_ctx* c, const uint8_t* bytes, uint64_t n_bytes, int format, uint32_t skip, int last, uint64_t* n_records);
}
int slimm_set_input_size_hint(slimm_ctx* c, uint64_t compressed_bytes) {
    if (!c) return SLIMM_E_INVALID;
    if (c->bam.active) return fail(c, SLIMM_E_INVALID, "slimm_set_input_size_hint: before the file's first window");
    c->bam.size_hint = compressed_bytes;
    return SLIMM_OK;
}
int slimm_window_memory(slimm_ctx* c, uint64_t* device_bytes) {
    if (!c || !device_bytes) return SLIMM_E_INVALID;
    *device_bytes = c->bam.held_bytes();
    return SLIMM_OK;
}
int slimm_device_memory(slimm_ctx* c, uint64_t* used_bytes, uint64_t* total_bytes) {
    if (!c || !used_bytes || !total_bytes) return SLIMM_E_INVALID;
    if (c->device < 0) return fail(c, SLIMM_E_INVALID, "host-only context");
    (void)hipSetDevice(c->device);
    size_t fr = 0, tot = 0;
    HIP_TRY(c, hipMemGetInfo(&fr, &tot));
    *used_bytes = tot - fr;
    *total_bytes = tot;
    return SLIMM_OK;
}
int slimm_push_bam_bytes(slimm_ctx* c, const uint8_t* bytes, uint64_t n_bytes, int last, uint64_t* n_records) {
    return bam_push_window(c, bytes, n_bytes, kFormatBam, 0u, last, n_records);
}
int slimm_push_bgzf_blocks(slimm_ctx* c, const uint8_t* blocks, uint64_t n_bytes, uint32_t skip, int last, uint64_t* n_records) {
    return bam_push_window(c, blocks, n_bytes, kFormatBgzf, skip, last, n_records);
}
int slimm_push_sam_bytes(slimm_ctx* c, const uint8_t* text, uint64_t n_bytes, int last, uint64_t* n_records) {
    return bam_push_window(c, text, n_bytes, kFormatSam, 0u, last, n_records);
}
// The header's reference names (@SQ SN, index = the reference id) for slimm_push_sam_bytes: a hash table on the device.
int slimm_set_reference_names(slimm_ctx* c, const char* const* names) {
    if (!c || !names) return SLIMM_E_INVALID;
    if (c->device < 0) return fail(c, SLIMM_E_INVALID, "host-only context has no record stream");
    (void)hipSetDevice(c->device);
    slimm_ctx::BamDecode& B = c->bam;
    uint32_t cap = 16;
    while (cap < 2u * c->R + 2u) cap <<= 1;
    std::vector<SamRefEntry> tab(cap);
    for (auto& e : tab) {
        e.hash = 0;
        e.ref = -1;
        e.name_off = e.name_len = e.pad = 0;
    }
    std::vector<uint8_t> blob;
    for (uint32_t r = 0; r < c->R; ++r) {
        const char* nm = names[r] ? names[r] : "";
        const size_t n = strlen(nm);
        const uint64_t h = sam_name_hash(nm, n);
        bool dup = false;
        uint32_t slot = static_cast<uint32_t>(h) & (cap - 1u);
        for (;; slot = (slot + 1u) & (cap - 1u)) {
            if (tab[slot].ref < 0) break;
            if (tab[slot].hash == h && tab[slot].name_len == n && memcmp(blob.data() + tab[slot].name_off, nm, n) == 0) {
                dup = true;   // (two header lines with one name: the first one's index, like the host reader's map)
                break;
            }
        }
        if (dup) continue;
        tab[slot].hash = h;
        tab[slot].ref = static_cast<int32_t>(r);
        tab[slot].name_off = static_cast<uint32_t>(blob.size());
        tab[slot].name_len = static_cast<uint32_t>(n);
        blob.insert(blob.end(), nm, nm + n);
    }
    blob.resize(blob.size() + 16, 0);
    HIP_TRY(c, B.sam_table.ensure(cap));
    HIP_TRY(c, B.sam_names.ensure(blob.size()));
    HIP_TRY(c, hipMemcpy(B.sam_table.p, tab.data(), cap * sizeof(SamRefEntry), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(B.sam_names.p, blob.data(), blob.size(), hipMemcpyHostToDevice));
    B.sam_mask = cap - 1u;
    return SLIMM_OK;
}
namespace {
// SLIMM_TRACE=push: what the window pipeline does and when (stderr; milliseconds since the first line)
void push_trace(const char* fmt, ...) {
    static const bool on = traced("push");
    if (!on) return;
    static const auto t0 = std::chrono::steady_clock::now();
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    va_list ap;
    va_start(ap, fmt);
    fprintf(stderr, "[push %9.3f] ", ms);
    vfprintf(stderr, fmt, ap);
    fputc('\n', stderr);
    va_end(ap);
}
// the window buffer of window `windows`, large enough for n_bytes behind its slack; what the slack holds is kept
int bam_window_buffer(slimm_ctx* c, uint64_t n_bytes, bool gathered = false) {
    slimm_ctx::BamDecode& B = c->bam;
    const uint32_t b = static_cast<uint32_t>(B.windows % slimm_ctx::kBamRing);
    // (a gathered window gets the room of the largest one at once: a buffer that grows is a hipFree, and a hipFree waits for
    // the inflate kernels of the windows before)
    // (planned from the caller's hint; not told: a quarter more than this window needs, up to the largest window there is)
    const uint64_t room = !gathered    ? n_bytes
                          : B.win_cap ? std::max<uint64_t>(n_bytes, B.win_cap)
                                      : std::max<uint64_t>(n_bytes, std::min<uint64_t>(slimm_ctx::kBamGather, n_bytes + (n_bytes >> 2)));
    const uint64_t need = kBamSlack + room + 64;
    if (B.bytes[b].cap >= need) return SLIMM_OK;
    // (what the buffer held -- the window a ring's length back -- is done with: it was finished before this one was let in.
    // Only the carried bytes in its slack matter, and only when the window before this one is finished already: otherwise
    // its end will put them there later)
    if (B.head == B.windows && B.carry_bytes) {
        DevBuf<uint8_t> nb;
        HIP_TRY(c, nb.ensure(need + (need >> 3)));
        HIP_TRY(c, hipMemcpyAsync(nb.p + kBamSlack - B.carry_bytes, B.bytes[b].p + kBamSlack - B.carry_bytes, B.carry_bytes,
                                  hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        std::swap(B.bytes[b].p, nb.p);
        std::swap(B.bytes[b].cap, nb.cap);
        if (nb.p) B.outgrown.push_back(nb.p);
        nb.p = nullptr;
        nb.cap = 0;
    } else {
        HIP_TRY(c, B.bytes[b].ensure_later(need + (gathered ? 0u : need >> 3), B.outgrown));
    }
    return SLIMM_OK;
}

void bam_free_outgrown(slimm_ctx* c) {
    for (void* p : c->bam.outgrown) (void)hipFree(p);
    c->bam.outgrown.clear();
}

// The BGZF blocks gathered so far become window `windows`: descriptors over, the inflate launched behind the copies on the
// inflate stream whose turn it is.
int bam_launch_gathered(slimm_ctx* c) {
    slimm_ctx::BamDecode& B = c->bam;
    if (!B.acc_open) return SLIMM_OK;
    const uint32_t b = static_cast<uint32_t>(B.windows % slimm_ctx::kBamRing), si = static_cast<uint32_t>(B.windows & 1u);
    const uint32_t nblk = static_cast<uint32_t>(B.desc_host[b].size());
    const uint64_t n_bytes = B.acc_dst - B.acc_skip;
    B.acc_open = false;
    if (!nblk || !n_bytes) return SLIMM_OK;
    const int rc = bam_window_buffer(c, n_bytes, true);
    if (rc != SLIMM_OK) return rc;
    // (room to spare only when a buffer has to GROW: a request "with a quarter more" that exceeded a buffer reserved for exactly
    // this kind of window re-allocated 3 GB of scratch in the middle of a file -- a hipMalloc that waits for the kernels in
    // flight and, now and then, 1.4 s for the memory itself: profiles/round6/06_cli_stall_hunt.txt)
    if (B.desc[b].cap < static_cast<size_t>(nblk) + 1)
        HIP_TRY(c, B.desc[b].ensure_later(static_cast<size_t>(nblk) + (nblk >> 1) + 1, B.outgrown));
    if (B.inflate_scratch[si].cap < bgzf_inflate_scratch_bytes(nblk, B.acc_tok)) {
        push_trace("inflate scratch %u grows: %.0f MB held, %.0f MB needed", si, B.inflate_scratch[si].cap / 1e6,
                   bgzf_inflate_scratch_bytes(nblk, B.acc_tok) / 1e6);
        HIP_TRY(c, B.inflate_scratch[si].ensure_later(bgzf_inflate_scratch_bytes(nblk + (nblk >> 2), B.acc_tok + (B.acc_tok >> 2)), B.outgrown));
    }
    HIP_TRY(c, B.inflate_status.ensure(4u * slimm_ctx::kBamRing));
    HIP_TRY(c, B.h_inflate_status.ensure(4));
    HIP_TRY(c, hipMemsetAsync(B.comp[b].p + B.acc_src, 0, kBgzfTail, c->copy_stream));
    HIP_TRY(c, hipMemcpyAsync(B.desc[b].p, B.desc_host[b].data(), static_cast<size_t>(nblk) * sizeof(BgzfBlock), hipMemcpyHostToDevice,
                              c->copy_stream));
    HIP_TRY(c, hipMemsetAsync(B.inflate_status.p + 4u * b, 0, 16, c->copy_stream));
    HIP_TRY(c, hipMemsetAsync(B.inflate_status.p + 4u * b + 1u, 0xff, 4, c->copy_stream));
    HIP_TRY(c, need_stream(B.inflate_stream[si], kStreamLow));
    if (!B.comp_copied) HIP_TRY(c, hipEventCreateWithFlags(&B.comp_copied, hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(B.comp_copied, c->copy_stream));
    HIP_TRY(c, hipStreamWaitEvent(B.inflate_stream[si], B.comp_copied, 0));
    launch_bgzf_inflate(B.inflate_stream[si], B.comp[b].p, B.desc[b].p, nblk, B.bytes[b].p + kBamSlack - B.acc_skip, B.inflate_scratch[si].p,
                        B.inflate_status.p + 4u * b);
    HIP_TRY(c, hipEventRecord(B.copied[b], B.inflate_stream[si]));
    B.inflated[b] = true;
    B.win_bytes[b] = n_bytes;
    ++B.windows;
    push_trace("window %llu launched: %u blocks, %.0f MB -> %.0f MB, inflate stream %u", (unsigned long long)(B.windows - 1), nblk,
               B.acc_src / 1e6, n_bytes / 1e6, si);
    return SLIMM_OK;
}

// A window of a BAM file's alignment-record bytes: inflated already (`bytes` are the records' bytes) or as whole BGZF blocks
// (`bytes` are compressed; the first `skip` inflated bytes are not records).  src_bytes = what crosses the bus.
// BGZF pushes are GATHERED: their compressed bytes are copied behind each other into the next window's buffer, and the window
// is launched -- inflate, then the record kernels -- once it holds kBamGatherGoal inflated bytes (or the file ends, or a
// push of the other kind comes): the inflate's first phase is a lane per block and wants tens of thousands of them, whatever
// size the caller's buffers have.
int bam_push_window(slimm_ctx* c, const uint8_t* bytes, uint64_t src_bytes, int format, uint32_t skip, int last, uint64_t* n_records) {
    if (!c) return SLIMM_E_INVALID;
    if (n_records) *n_records = 0;
    bool compressed = format == kFormatBgzf;
    const bool sam = format == kFormatSam;
    if (sam && !c->bam.sam_mask) return fail(c, SLIMM_E_INVALID, "slimm_set_reference_names first: SAM text names its references");
    if (c->bam.active && c->bam.sam != sam) return fail(c, SLIMM_E_INVALID, "SAM text and BAM bytes do not mix within a file");
    uint64_t n_bytes = src_bytes;  // the push's record bytes
    std::vector<BgzfBlock> dh;
    uint64_t inflated = 0;
    if (compressed && src_bytes) {
        if (!bytes) return fail(c, SLIMM_E_INVALID, "null byte buffer");
        std::string why;
        if (!bgzf_parse_blocks(bytes, src_bytes, 0, dh, inflated, why)) return fail(c, SLIMM_E_INVALID, "%s", why.c_str());
        if (skip > inflated || (skip && c->bam.active && (c->bam.windows > 0 || c->bam.acc_open)))
            return fail(c, SLIMM_E_INVALID, "skip: only in front of a file's first records");
        if (dh.size() >= (1ull << 31)) return fail(c, SLIMM_E_INVALID, "too many blocks in one window");
        n_bytes = inflated - skip;
        // blocks that lie wholly inside the skipped bytes (a BAM header of any size) are not inflated at all; what is left to
        // skip is less than one block, so the inflater's first byte stays inside the window buffer's slack
        size_t drop = 0;
        while (drop < dh.size() && dh[drop].dst + dh[drop].isize <= skip) ++drop;
        if (drop) {
            const uint64_t d0 = drop < dh.size() ? dh[drop].dst : inflated;
            dh.erase(dh.begin(), dh.begin() + static_cast<long>(drop));
            for (BgzfBlock& d : dh) d.dst -= d0;
            skip -= static_cast<uint32_t>(d0);
            inflated -= d0;
        }
        if (skip >= 65536u) return fail(c, SLIMM_E_INVALID, "skip: past the first block that holds a record byte");
        if (n_bytes == 0) {  // (blocks without a record byte: nothing to inflate, nothing to decode)
            compressed = false;
            src_bytes = 0;
        }
    } else if (compressed) {
        compressed = false;
    }
    if (c->device < 0) return fail(c, SLIMM_E_INVALID, "host-only context has no record stream");
    if (n_bytes && !bytes) return fail(c, SLIMM_E_INVALID, "null byte buffer");
    if (c->analyzed) return fail(c, SLIMM_E_INVALID, "records already analysed; reset first");
    if (c->borrowed) return fail(c, SLIMM_E_INVALID, "records are borrowed device arrays; reset first");
    if (n_bytes >= (1ull << 31)) return fail(c, SLIMM_E_INVALID, "a window of BAM bytes is less than 2 GiB");
    const bool marked = c->order == SLIMM_ORDER_GROUPED;
    if (c->n_pushed && !c->bam.active)
        return fail(c, SLIMM_E_INVALID, "earlier batches were decoded records: the forms do not mix within a file");
    (void)hipSetDevice(c->device);
    HIP_TRY(c, need_stream(c->copy_stream, kStreamHigh));
    slimm_ctx::BamDecode& B = c->bam;
    hipStream_t st = c->stream;
    if (!B.active) {  // a file's first window
        B.active = true;
        B.windows = 0;
        B.head = 0;
        B.carry_bytes = 0;
        B.pushes = 0;
        B.acc_open = false;
        B.sam = sam;
        B.planned = false;
        B.sam_last_byte = '\n';
        c->marked = marked;
        c->has_check = !marked;
        c->packed = false;
        HIP_TRY(c, B.carry.ensure(1));
        HIP_TRY(c, B.result.ensure(1));
        for (auto& e : B.copied)
            if (!e) HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (auto& e : B.h2d_done)
            if (!e) HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        HIP_TRY(c, hipMemsetAsync(B.carry.p, 0, sizeof(BamCarry), st));
    }
    if (B.closed) return fail(c, SLIMM_E_INVALID, "the file's last window has been pushed; reset first");
    uint64_t total = 0;
    bool copy_started = false;
    push_trace("push: %.0f MB %s -> %.0f MB%s", src_bytes / 1e6, compressed ? "of blocks" : "inflated", n_bytes / 1e6, last ? " (last)" : "");
    if (n_bytes && compressed) {
        // behind what is gathered already -- unless the window would grow past its size: that one goes first
        if (B.acc_open && B.acc_dst + inflated > slimm_ctx::kBamGather) {
            const int rc = bam_launch_gathered(c);
            if (rc != SLIMM_OK) return rc;
        }
        const uint32_t b = static_cast<uint32_t>(B.windows % slimm_ctx::kBamRing);
        if (!B.acc_open) {
            B.acc_open = true;
            B.acc_src = B.acc_dst = 0;
            B.acc_skip = skip;
            B.acc_tok = 0;
            B.desc_host[b].clear();
            // A file's first window: the buffers its windows will take in turn, now -- an allocation (like a hipFree) made
            // while inflate kernels are in flight waits for them, and the pushes that should overlap them stand still.  How
            // many and how large comes from the caller's hint (slimm_set_input_size_hint: the file's compressed bytes) and
            // this push's own ratios: a 70 MB file gets one window of its size, not four of 1.9 GB (17 GB
            // per context whatever the file).  Without a hint nothing is reserved ahead: buffers appear as windows need them.
            if (!B.planned) {   // (the file's first compressed push: windows the host inflated may have gone before it)
                B.planned = true;
                B.win_cap = 0;
                if (B.size_hint && src_bytes && inflated) {
                    // (those windows are finished first -- their buffers are about to be replaced; the bytes they carry over
                    // move with the buffer of the window that is next)
                    while (B.head < B.windows) {
                        uint64_t got = 0;
                        const uint64_t j = B.head;
                        const int frc = bam_finish_window(c, j, B.win_bytes[j % slimm_ctx::kBamRing], false, got);
                        ++B.head;
                        if (frc != SLIMM_OK) return frc;
                        total += got;
                    }
                    const double ratio = static_cast<double>(inflated) / static_cast<double>(src_bytes);
                    const uint64_t left = B.size_hint > src_bytes ? B.size_hint - src_bytes : 0;
                    const uint64_t est = inflated + static_cast<uint64_t>(static_cast<double>(left) * ratio * 1.08) + (16ull << 20);
                    const uint64_t nwin = est <= slimm_ctx::kBamGather ? 1u : (est + slimm_ctx::kBamGatherGoal - 1) / slimm_ctx::kBamGatherGoal;
                    const uint32_t nbuf = static_cast<uint32_t>(std::min<uint64_t>(slimm_ctx::kBamRing, nwin));
                    B.win_cap = nwin == 1 ? est : slimm_ctx::kBamGather;
                    // (a window's compressed bytes: what inflates to the goal, plus the push that crosses it)
                    const uint64_t comp_cap = (nwin == 1 ? B.size_hint + (B.size_hint >> 4)
                                                         : static_cast<uint64_t>(static_cast<double>(slimm_ctx::kBamGatherGoal) / ratio * 1.15) + src_bytes) +
                                              kBgzfTail + (1ull << 20);
                    const double blocks_per_byte = static_cast<double>(dh.size()) / static_cast<double>(inflated);
                    const uint32_t blocks_max = static_cast<uint32_t>(static_cast<double>(B.win_cap) * blocks_per_byte * 1.25) + 1024u;
                    const uint64_t tok_max = B.win_cap / 3u + B.win_cap / 256u + 8ull * blocks_max;
                    for (uint32_t k = 0; k < nbuf; ++k) {
                        if (k == b && B.carry_bytes) {   // (the carried bytes lie in this buffer's slack: bam_window_buffer keeps them)
                            const int wrc = bam_window_buffer(c, B.win_cap, true);
                            if (wrc != SLIMM_OK) return wrc;
                        } else {
                            HIP_TRY(c, B.bytes[k].ensure_later(kBamSlack + B.win_cap + 64, B.outgrown));
                        }
                        HIP_TRY(c, B.comp[k].ensure_later(comp_cap, B.outgrown));
                        HIP_TRY(c, B.desc[k].ensure_later(blocks_max, B.outgrown));
                    }
                    for (uint32_t k = 0; k < std::min<uint32_t>(2u, nbuf); ++k)
                        HIP_TRY(c, B.inflate_scratch[k].ensure_later(bgzf_inflate_scratch_bytes(blocks_max, tok_max), B.outgrown));
                    const size_t np_max = bam_pieces(B.win_cap + kBamSlack) + 64;
                    HIP_TRY(c, B.pieces.ensure_later(np_max, B.outgrown));
                    HIP_TRY(c, B.offs.ensure_later(np_max * kBamSlots, B.outgrown));
                    push_trace("buffers reserved");
                    push_trace("planned %llu window(s) of <= %.0f MB in %u buffer(s), %.0f MB of compressed bytes each: %.2f GB held",
                               (unsigned long long)nwin, B.win_cap / 1e6, nbuf, comp_cap / 1e6, B.held_bytes() / 1e9);
                }
                HIP_TRY(c, B.inflate_status.ensure(4u * slimm_ctx::kBamRing));
                HIP_TRY(c, B.h_inflate_status.ensure(4));
            }
        }
        const uint64_t need = B.acc_src + src_bytes + kBgzfTail + 64;
        if (B.comp[b].cap < need) {  // (grown with what earlier pushes of this window have put there)
            DevBuf<uint8_t> nb;
            HIP_TRY(c, nb.ensure(need + (need >> 1)));
            if (B.acc_src) HIP_TRY(c, hipMemcpyAsync(nb.p, B.comp[b].p, B.acc_src, hipMemcpyDeviceToDevice, c->copy_stream));
            std::swap(B.comp[b].p, nb.p);
            std::swap(B.comp[b].cap, nb.cap);
            if (nb.p) B.outgrown.push_back(nb.p);   // (no hipFree here: it would wait for the inflate kernels in flight)
            nb.p = nullptr;
            nb.cap = 0;
        }
        HIP_TRY(c, hipMemcpyAsync(B.comp[b].p + B.acc_src, bytes, src_bytes, hipMemcpyHostToDevice, c->copy_stream));
        HIP_TRY(c, hipEventRecord(B.h2d_done[B.pushes % 4u], c->copy_stream));
        ++B.pushes;
        copy_started = true;
        const uint32_t tok0 = dh.empty() ? 0u : dh.front().tok;   // (blocks dropped in front of a file's first record)
        for (BgzfBlock& d : dh) {
            d.src += B.acc_src;
            d.dst += B.acc_dst;
            d.tok = d.tok - tok0 + B.acc_tok;
        }
        if (!dh.empty()) B.acc_tok = dh.back().tok + bgzf_token_room(dh.back().isize);
        B.desc_host[b].insert(B.desc_host[b].end(), dh.begin(), dh.end());
        B.acc_src += src_bytes;
        B.acc_dst += inflated;
        if (B.acc_dst >= slimm_ctx::kBamGatherGoal) {
            const int rc = bam_launch_gathered(c);
            if (rc != SLIMM_OK) return rc;
        }
    } else if (n_bytes || (sam && last && B.sam_last_byte != '\n')) {  // inflated bytes / text: a window of their own, behind what was gathered
        int rc = bam_launch_gathered(c);
        if (rc != SLIMM_OK) return rc;
        // (SAM text whose last line has no newline gets one: a line ends where its newline is)
        if (sam && n_bytes) B.sam_last_byte = bytes[n_bytes - 1];
        const bool add_newline = sam && last && B.sam_last_byte != '\n';
        rc = bam_window_buffer(c, n_bytes + (add_newline ? 1u : 0u));
        if (rc != SLIMM_OK) return rc;
        const uint32_t b = static_cast<uint32_t>(B.windows % slimm_ctx::kBamRing);
        B.inflated[b] = false;
        if (n_bytes) HIP_TRY(c, hipMemcpyAsync(B.bytes[b].p + kBamSlack, bytes, n_bytes, hipMemcpyHostToDevice, c->copy_stream));
        if (add_newline) {
            HIP_TRY(c, hipMemsetAsync(B.bytes[b].p + kBamSlack + n_bytes, '\n', 1, c->copy_stream));
            n_bytes += 1;
            B.sam_last_byte = '\n';
        }
        HIP_TRY(c, hipEventRecord(B.copied[b], c->copy_stream));
        HIP_TRY(c, hipEventRecord(B.h2d_done[B.pushes % 4u], c->copy_stream));
        ++B.pushes;
        copy_started = true;
        B.win_bytes[b] = n_bytes;
        ++B.windows;
    }
    if (last) {
        const int rc = bam_launch_gathered(c);
        if (rc != SLIMM_OK) return rc;
    }
    // ... while the windows before are worked on: the oldest are finished (found, counted, decoded) once more than kBamLag
    // windows or kBamInFlight bytes are in flight -- all of them when this is the file's end
    const bool had_any = B.head < B.windows;
    for (;;) {
        if (B.head >= B.windows) break;
        uint64_t in_flight = 0;
        for (uint64_t j = B.head; j < B.windows; ++j) in_flight += B.win_bytes[j % slimm_ctx::kBamRing];
        if (!last && B.windows - B.head <= slimm_ctx::kBamLag && in_flight <= slimm_ctx::kBamInFlight) break;
        uint64_t got = 0;
        const uint64_t j = B.head;
        push_trace("finishing window %llu (%llu .. %llu in flight, %.0f MB)", (unsigned long long)j, (unsigned long long)B.head,
                   (unsigned long long)B.windows, in_flight / 1e6);
        const int rc = bam_finish_window(c, j, B.win_bytes[j % slimm_ctx::kBamRing], last && j + 1 == B.windows, got);
        push_trace("finished window %llu: %llu records", (unsigned long long)j, (unsigned long long)got);
        ++B.head;
        if (rc != SLIMM_OK) return rc;
        total += got;
    }
    if (last) {
        if (!had_any && B.carry_bytes) return fail(c, SLIMM_E_INVALID, "truncated BAM record");
        B.closed = true;
        bam_free_outgrown(c);
    } else {
        // the caller's buffer of the call BEFORE this one has been read (it may be reused once this call returns): the most
        // recent push whose copy this call did not start itself (a call without record bytes starts none)
        const uint64_t mine = copy_started ? 1u : 0u;
        if (B.pushes > mine) HIP_TRY(c, hipEventSynchronize(B.h2d_done[(B.pushes - 1u - mine) % 4u]));
        push_trace("push returns");
    }
    if (n_records) *n_records = total;
    return SLIMM_OK;
}
}  // namespace


}  // extern "C"
