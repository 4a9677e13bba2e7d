// Internal to the library: the context behind the C ABI (include/slimm_hip.h) and what its translation units share --
// context.hip (create / reset, phases A and B, the getters), records.hip (decoded records in), windows.hip (BAM / BGZF /
// SAM windows decoded on the device).
#pragma once
#include <cerrno>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/slimm_hip.h"
#include "host_profile.hpp"
#include "force.h"
#include "kernels.h"
#include "read_identity.h"

namespace slimm {



extern thread_local std::string g_create_error;

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;  // elements
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    hipError_t ensure(size_t n) {
        if (n <= cap) return hipSuccess;
        release();
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), n * sizeof(T));
        if (e == hipSuccess) cap = n;
        return e;
    }
    // the same without hipFree (which waits for every kernel in flight on the device): what the buffer was goes to `old`,
    // whose owner frees it when the device has nothing to do anyway
    hipError_t ensure_later(size_t n, std::vector<void*>& old) {
        if (n <= cap) return hipSuccess;
        if (p) old.push_back(p);
        p = nullptr;
        cap = 0;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), n * sizeof(T));
        if (e == hipSuccess) cap = n;
        return e;
    }
};

template <typename T>
struct PinBuf {
    T* p = nullptr;
    size_t cap = 0;
    ~PinBuf() {
        if (p) (void)hipHostFree(p);
    }
    hipError_t ensure(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&p), n * sizeof(T), hipHostMallocDefault);
        if (e == hipSuccess) cap = n;
        return e;
    }
};

enum KernelId {
    K_MEMSET = 0, K_GROUP_COUNT, K_GROUP_SCAN, K_GROUP_SCATTER, K_GROUP_FINISH, K_FRONT, K_HIST, K_REF_STATS, K_FILTER,
    K_REF_STATS2, K_TILE_COUNT, K_TILE_SCAN, K_TILE_SCATTER, K_TILE_HIST, K_TILE_COUNT2, K_TILE_SCAN2, K_TILE_SCATTER2,
    K_TILE_HIST2, K_PACK, K_PACK2, K_COUNT
};
extern const char* kKernelNames[K_COUNT];

constexpr uint32_t kTailWords = 64;

}  // namespace slimm

using namespace slimm;   // (library-internal header)

struct slimm_ctx {
    std::unique_ptr<HostProfile> host;
    std::string err;
    int device = -1;  // -1: host-only context
    int order = SLIMM_ORDER_GROUPED;
    hipStream_t stream = nullptr;
    // streamed ingest (slimm_push_records_async): host -> device copies on a stream of their own, ordered before phase A
    // by an event (never by the host); two page-locked staging sets for callers that produce records piecemeal
    // Streams and HARDWARE QUEUES.  A process has few hardware queues per device (four of each priority class unless
    // GPU_MAX_HW_QUEUES says otherwise); the runtime deals streams to them as they are made, and streams that share one wait
    // for each other.  Which of the window pipeline's streams shared decided whether `slimm DB 100M.bam` spent 110-210 or
    // 320-360 ms in slimm_push_bgzf_blocks (every odd window's inflate 18 ms late), and differently for one context and for a
    // group's two members on one device (profiles/round6/11_*).  So the streams are told apart by PRIORITY CLASS, which the
    // runtime keeps in separate queue pools: the copy stream high (its work is the DMA engine's; what it needs is to be
    // seen at once), the two inflate streams low (long kernels that fill what is free), the main stream and the side stream
    // normal -- no class holds more than four streams even with two members on one device.  The copy stream, the side
    // stream and the inflaters' streams are made when first used (need_stream below).
    hipStream_t copy_stream = nullptr;
    hipEvent_t copy_done = nullptr;
    bool copy_pending = false;
    bool filter_pending = false;  // slimm_filter_alignments_launch ran; slimm_install_merged_partials completes it
    bool stream_ordered = false;  // slimm_set_stream_ordered: the caller enqueues its collectives on `stream`
    struct Staging {
        PinBuf<uint64_t> key;
        PinBuf<int32_t> ref, pos;
        PinBuf<uint16_t> flag;
        hipEvent_t done = nullptr;
        bool pending = false;
    } staging[2];

    uint32_t R = 0, T = 0;
    uint64_t Bp = 0;                   // padded bins per coverage array (multiple of 64)
    std::vector<uint32_t> bin_off_h;   // [R+1] padded offsets

    // static tables
    DevBuf<uint32_t> d_ref_len, d_bin_off, d_lin_dense;
    DevBuf<uint32_t> d_tile_ref0;     // per bin tile: first reference overlapping it (fused statistics)
    DevBuf<uint2> d_geo;              // {contig length, first bin} per reference: one gather in k_emit
    DevBuf<uint8_t> d_valid;
    DevBuf<uint4> d_rows16;           // per run: 16-byte lineage rows with the valid bit
    DevBuf<uint32_t> d_level_taxon;    // [(index << 3) | level] -> dense taxon (8 << taxon_shift entries)
    DevBuf<uint32_t> d_taxon_off, d_taxon_idx;  // ... and back: the (level, index) entries of dense taxon t (CSR)
    uint32_t taxon_shift = 0;
    uint32_t Tsel = 0;                 // size of the selectors' taxon space: 8 << taxon_shift (16-byte rows) or T
    PinBuf<uint4> h_rows16;
    DevBuf<uint32_t> d_valid_bits;     // one bit per reference: what k_filter_compact asks before anything else
    PinBuf<uint32_t> h_valid_bits;
    std::vector<uint32_t> valid_bits_prev;   // references whose bit is set in h_valid_bits
    bool rows16_base_ready = false;    // h_rows16 holds the static part of every row
    std::vector<uint32_t> rows16_prev; // references whose valid bit is set in h_rows16
    bool use_rows16 = false;
    // records
    DevBuf<uint64_t> in_key;
    DevBuf<int32_t> in_ref, in_pos;
    DevBuf<uint16_t> in_flag;
    DevBuf<uint32_t> in_check;     // slimm_push_records_checked: a second hash of every record's read name
    bool has_check = false;        // ... all pushed batches carry one (checked and unchecked pushes do not mix)
    bool packed = false;           // slimm_push_records_packed: 16 bytes per record, no flag array (forms do not mix)
    bool marked = false;           // slimm_push_records_marked: 8 bytes per record, no key array (grouped input only)
    // slimm_push_bam_bytes: BAM records decoded on the device (bam_decode.hip).  Two byte buffers [slack | window] take the
    // windows in turn; the incomplete record at a window's end is copied in front of the next window
    // (the ring of window buffers: a window is copied -- or inflated -- into one while older ones are still on their way or
    // being decoded: at most kBamLag of them, and at most kBamInFlight bytes -- windows that arrive as BGZF blocks are
    // gathered into device windows of up to kBamGather inflated bytes, three of which keep both inflate streams busy; a ring
    // of 16 buffers of that size was 20 - 25 GB of HBM per context)
    static constexpr uint32_t kBamRing = 4, kBamLag = kBamRing - 2;
    static constexpr uint64_t kBamInFlight = 4ull << 30;    // finish the oldest window when more than this is in flight
    static constexpr uint64_t kBamGather = 1900ull << 20;   // inflated bytes of a gathered device window (a window is < 2 GiB)
    static constexpr uint64_t kBamGatherGoal = 1400ull << 20;  // ... which is launched once it holds this much
    static constexpr uint64_t kBamKeepAcrossFiles = 4ull << 30; // slimm_reset gives the pipeline's buffers back above this
    struct BamDecode {
        DevBuf<uint8_t> bytes[kBamRing];
        DevBuf<BamPiece> pieces;
        DevBuf<uint32_t> offs;
        DevBuf<BamCarry> carry;
        PinBuf<BamWindowResult> result;     // written by k_bam_scan straight into page-locked host memory
        std::vector<std::pair<const uint8_t*, size_t>> registered;  // caller buffers page-locked by hipHostRegister
        uint64_t windows = 0;               // of this file, handed over so far
        uint64_t head = 0;                  // ... of which [head, windows) are not finished yet (copied / inflating / waiting)
        uint64_t win_bytes[kBamRing] = {};  // record bytes of the windows in flight
        uint64_t carry_bytes = 0;
        bool active = false;                // this file's records come from slimm_push_bam_bytes / slimm_push_bgzf_blocks
        bool closed = false;                // the file's last window went in
        hipEvent_t copied[kBamRing] = {};   // the window's bytes are in its buffer (behind the copy, or behind the inflate)
        hipEvent_t h2d_done[4] = {};        // the caller's buffer of a push has been read (the pushes' events, in turn)
        uint64_t pushes = 0;                // pushes of this file that started a copy
        // windows that arrive as BGZF blocks (slimm_push_bgzf_blocks): compressed bytes + block descriptors per buffer, the
        // inflater's scratch and {error code, first bad block} per buffer; inflated[b]: that window was inflated here
        DevBuf<uint8_t> comp[kBamRing];
        DevBuf<BgzfBlock> desc[kBamRing];
        DevBuf<uint8_t> inflate_scratch[2];
        DevBuf<uint32_t> inflate_status;       // 4 words per buffer
        PinBuf<uint32_t> h_inflate_status;     // ... fetched with the window's other results
        bool inflated[kBamRing] = {};
        // the inflate kernels' own streams, taken in turn by the device windows: the copies of other windows go on beside
        // them, and the Huffman phase of one window (a lane per block: 30 K blocks are half the lanes) beside the other's
        hipStream_t inflate_stream[2] = {nullptr, nullptr};
        hipEvent_t comp_copied = nullptr;
        // BGZF pushes gathered for the next device window (buffer windows % kBamRing): compressed bytes so far, inflated
        // bytes so far, the inflated bytes in front of the file's first record
        bool acc_open = false;
        uint64_t acc_src = 0, acc_dst = 0;
        uint32_t acc_skip = 0, acc_tok = 0;   // (acc_tok: words of token room of the gathered blocks)
        std::vector<void*> outgrown;   // device buffers replaced by larger ones while kernels were in flight: freed at the file's end
        // SAM text (slimm_push_sam_bytes, sam_decode.hip): this file's windows are text; the header's reference names as a
        // hash table on the device (slimm_set_reference_names); the last byte pushed (a last line without its newline gets one)
        bool sam = false;
        DevBuf<SamRefEntry> sam_table;
        DevBuf<uint8_t> sam_names;
        uint32_t sam_mask = 0;
        uint8_t sam_last_byte = '\n';
        std::vector<BgzfBlock> desc_host[kBamRing];   // (a buffer's descriptors stay until the buffer's turn comes again: the copy reads them)
        // Q18 on a grouped stream (kernels.h: BamCarry): the decoders' two counts of the windows finished so far
        uint64_t q18_starts = 0, q18_plain = 0;
        // what the file's gathered windows are sized for: slimm_set_input_size_hint (the file's compressed bytes; 0 = not
        // told) and, from it and the first push's ratio, the inflated bytes a gathered window's buffer gets (0 = kBamGather)
        uint64_t size_hint = 0, win_cap = 0;
        bool planned = false;                // the file's first COMPRESSED push has reserved its buffers (or found no hint)
        uint64_t held_bytes() const {   // device memory of the window pipeline
            uint64_t n = pieces.cap * sizeof(BamPiece) + offs.cap * 4ull;
            for (uint32_t k = 0; k < kBamRing; ++k) n += bytes[k].cap + comp[k].cap + desc[k].cap * sizeof(BgzfBlock);
            for (auto& sc : inflate_scratch) n += sc.cap;
            return n;
        }
    } bam;
    DeviceRecords rec;      // what analyze reads (owned buffers or borrowed pointers)
    bool borrowed = false;
    uint64_t n_pushed = 0;
    // work arrays
    // record_order = ANY (group_by_ident.hip): the grouped stream {identity, {reference, bin}, check word} + scratch
    DevBuf<uint64_t> c_ident, s_ident;
    DevBuf<uint2> c_pay, s_pay;
    DevBuf<uint32_t> group_hist, c_chk, s_chk;
    DevBuf<uint32_t> tgt_ref, tgt_gbin;  // targets (bit 31: first of its read / the read has one target), in slots
    DevBuf<uint4> slots;                 // per kSlotRecs records: {first target, targets, reads, mapped records}
    DevBuf<uint4> tot_part;              // per workgroup of k_tile_count: totals of its slots (kernels.h: Totals)
    DevBuf<uint16_t> bucket;                            // targets bucketed by bin tile (13-bit bin | unique bit)
    DevBuf<uint32_t> tile_count, tile_base, tile_cursor, split_tiles;
    bool keep_bins = true;       // materialise cov / uniq_cov / uniq_cov2 in HBM (slimm_keep_bins)
    bool binsA_stored = false, binsB_stored = false;
    DevBuf<uint4> tile_items, part_items;
    DevBuf<uint32_t> mid, sup_cursor;                   // level-1 buckets (by super tile) and their cursors
    DevBuf<uint32_t> sel;                               // per read, dense: its uniq_cov2 bin, Bp + LCA taxon, or 0xffffffff
    DevBuf<uint32_t> filter_redo;                       // slots k_filter_compact leaves to k_filter_walk
    DevBuf<uint32_t> slot_rbase, slot_bbase;            // reads in front of a slot = rbase[s] + bbase[s >> 10] (side stream)
    hipStream_t side_stream = nullptr;
    hipEvent_t front_done = nullptr, prefix_done = nullptr;
    bool prefix_pending = false;  // prefix_done has been recorded and not been waited for by the main stream yet
    uint32_t tile_shift = kTileShiftSmall;              // log2 of the bins per tile: which build of tile_hist.hip runs (kernels.h)
    uint32_t tile_bins() const { return 1u << tile_shift; }
    uint32_t ntiles = 0;
    uint32_t Tpad = 0, ntiles2 = 0;                     // taxa padded to whole tiles; tiles of [uniq_cov2 | taxa]
    bool fused_scan = false;                            // k_tile_scan runs inside the one-level bucketing kernel
    uint32_t treps = 1, tstride = 0;                    // copies of the tile counters / cursors and their stride
    bool two_level = false;   // bucket through super tiles first (many tiles: one-level scatter stores are too scattered)
    bool matrix = false;      // phase B may bucket through a count matrix (one row per counting workgroup, no atomics)
    bool matrix_always = false;
    int wide_tiles = -1;      // SLIMM_FORCE wide_tiles: -1 = by the file's size, 0 / 1 = never / always (tests)
    // far more entries per tile than a packed work item holds (1 B records on 20 k references): work items of up to
    // kTileSubWide entries with 32-bit counts, so that a tile is one item again (kernels.h)
    bool wide_for(uint32_t n_records) const {
        return wide_tiles >= 0 ? wide_tiles != 0 : (ntiles && n_records / ntiles > 32768u);
    }
    DevBuf<uint32_t> tile_matrix;
    // multi-GPU coverage summary [4R sums | 16 scalars | bitmaps]: n_slices = 0: not announced (bitmaps by extra kernels),
    // 1: bitmaps written by k_tile_hist as [cov | uniq_cov], n > 1: in n slices of tiles for the all-to-all exchange
    uint32_t summary_slices = 0;
    bool summary_has_bits = false;   // the bitmaps of this analysis are in `summary` already
    uint32_t summary_layout = 0xffffffffu;  // n_slices the buffer was last zeroed for
    DevBuf<uint32_t> d_sum_vec;      // all-to-all form: [4R | 16] additive vector, all-reduced in place
    uint32_t slice_tiles() const { return summary_slices > 1 ? (ntiles + summary_slices - 1) / summary_slices : ntiles; }
    uint64_t slice_words() const { return static_cast<uint64_t>(slice_tiles()) * (tile_bins() / 32); }  // per array
    uint64_t summary_words() const { return 4ull * R + 16 + 2ull * std::max<uint32_t>(summary_slices, 1u) * slice_words(); }
    BitsLayout bits_layout() {
        BitsLayout b;
        if (summary_has_bits) {
            b.base = reinterpret_cast<uint64_t*>(summary.p + 4ull * R + 16);
            b.tps = slice_tiles();
            b.slice_w64 = slice_words() / 2;
        }
        return b;
    }
    bool statsA_final = false;  // k_pack has added the non-zero counts of the split tiles to the fused statistics
    bool bins_exposed = false;  // the caller holds the coverage buffer (may have merged other ranks' bins into it)
    bool use_tiles = false;   // LDS-privatised histograms (default) vs direct global atomics (too many tiles for LDS)
    DevBuf<uint32_t> bins;       // cov | uniq_cov | tail | uniq_cov2
    DevBuf<uint32_t> counters;   // CNT_WORDS
    DevBuf<uint32_t> ref_stats;  // [R*4] then [R*4]
    DevBuf<uint32_t> summary;    // multi-GPU: [4R sums | 16 scalars | cov bits | uniq_cov bits]
    DevBuf<uint32_t> lca_count, marks;
    DevBuf<uint32_t> d_partials;  // multi-GPU: the additive partial results, summed across ranks in place
    PinBuf<uint32_t> h_partials;
    DevBuf<uint64_t> pair_tab, pair_list;
    uint32_t pair_cap = 0;  // power of two
    // pinned staging
    PinBuf<uint32_t> h_stats, h_small, h_lca, h_marks;
    PinBuf<uint64_t> h_pairs;

    // per-run state
    bool analyzed = false, covered = false, filtered = false, counted = false, no_hits = false;
    uint32_t local_V = 0, local_M = 0, local_P = 0;
    uint32_t n_pairs = 0;
    std::vector<uint32_t> nz_ucov2;
    // partials handed out / installed
    std::vector<uint32_t> part_u2, part_lca, part_marks;
    std::vector<uint64_t> part_pairs;

    // kernel timing
    bool timing = false;
    int timing_only = -1;  // >= 0: bracket only this kernel id (keeps the event overhead out of the other launches)
    struct Ev {
        hipEvent_t a, b;
        int id;
    };
    std::vector<Ev> ev_used, ev_free;
    double k_ms[K_COUNT] = {0};
    uint32_t k_n[K_COUNT] = {0};

    // packed result blocks: A = [4R stats | 32 counters | 16 tail], B = [4R stats2 | 32 counters | R marks | T lca]
    size_t statsA_words() const { return 4ull * R + 64; }
    size_t statsB_words() const { return 5ull * R + 32 + T; }
    bool pair_clean = false;  // the (taxon, ref) hash set holds only empty slots
    uint32_t* cov() { return bins.p; }
    uint32_t* ucov() { return bins.p + Bp; }
    uint32_t* tail() { return bins.p + 2 * Bp; }
    uint32_t* ucov2() { return bins.p + 2 * Bp + kTailWords; }
    uint32_t* lca_tiles() { return bins.p + 3 * Bp + kTailWords; }  // [Tpad] right behind uniq_cov2: one index space
};

namespace slimm {

// SLIMM_TRACE=host: wall-clock marks of the host steps between the two device phases, on stderr
struct HostTrace {
    bool on;
    std::chrono::steady_clock::time_point t0;
    const char* what;
    explicit HostTrace(const char* w) : on(false), what(w) {
        static const bool enabled = traced("host");
        on = enabled;
        if (on) t0 = std::chrono::steady_clock::now();
    }
    void mark(const char* step) {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[host] %s: %s %.1f us\n", what, step, std::chrono::duration<double, std::micro>(t1 - t0).count());
        t0 = t1;
    }
};

int fail(slimm_ctx* c, int code, const char* fmt, ...);

#define HIP_TRY(c, expr)                                                                              \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) return fail((c), SLIMM_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

struct KernelTimer {  // brackets one launch (or a group) with events when timing is on
    slimm_ctx* c;
    slimm_ctx::Ev ev{};
    bool on;
    bool dispatch;  // the events are handed to the launch itself (hipExtLaunchKernelGGL: the dispatch's own time stamps)
    KernelTimer(slimm_ctx* ctx, int id, bool of_dispatch = false)
        : c(ctx), on(ctx->timing && (ctx->timing_only < 0 || ctx->timing_only == id)), dispatch(of_dispatch) {
        if (!on) return;
        if (!c->ev_free.empty()) {
            ev = c->ev_free.back();
            c->ev_free.pop_back();
        } else {
            (void)hipEventCreate(&ev.a);
            (void)hipEventCreate(&ev.b);
        }
        ev.id = id;
        if (!dispatch) (void)hipEventRecord(ev.a, c->stream);
    }
    hipEvent_t t0() const { return on ? ev.a : nullptr; }
    hipEvent_t t1() const { return on ? ev.b : nullptr; }
    ~KernelTimer() {
        if (!on) return;
        if (!dispatch) (void)hipEventRecord(ev.b, c->stream);
        c->ev_used.push_back(ev);
    }
};

void drain_events(slimm_ctx* c);
int ensure_work_buffers(slimm_ctx* c, uint32_t n);
int ensure_pair_table(slimm_ctx* c, uint32_t cap);
int check_device_errors(slimm_ctx* c, uint32_t err);
int bam_fetch_q18(slimm_ctx* c);   // windows.hip: the Q18 run counts of the device decoders so far

// Grows one record array to `cap` elements, keeping the `used` elements pushed so far (when the array holds them at all:
// an array the file's record form does not use is neither allocated nor copied).
// later != nullptr: no hipFree now (it waits for every kernel in flight -- the inflate of the windows behind this one):
// what the array was goes there and is freed when the file has ended
enum StreamClass { kStreamHigh = 0, kStreamNormal = 1, kStreamLow = 2 };
inline hipError_t need_stream(hipStream_t& s, StreamClass cls = kStreamNormal) {
    if (s) return hipSuccess;
    int least = 0, greatest = 0;  // (numerically: greatest priority = the smallest number)
    if (cls == kStreamNormal || hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || least == greatest)
        return hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    return hipStreamCreateWithPriority(&s, hipStreamNonBlocking, cls == kStreamHigh ? greatest : least);
}

template <typename T>
hipError_t grow_record_array(DevBuf<T>& buf, uint64_t cap, uint64_t used, hipStream_t st, std::vector<void*>* later = nullptr) {
    if (cap <= buf.cap) return hipSuccess;
    if (used == 0 || buf.cap < used) return later ? buf.ensure_later(cap, *later) : buf.ensure(cap);  // nothing of this file in it
    DevBuf<T> nb;
    hipError_t e = nb.ensure(cap);
    if (e != hipSuccess) return e;
    e = hipMemcpyAsync(nb.p, buf.p, used * sizeof(T), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(st);
    if (e != hipSuccess) return e;
    std::swap(buf.p, nb.p);
    std::swap(buf.cap, nb.cap);
    if (later && nb.p) {
        later->push_back(nb.p);
        nb.p = nullptr;
        nb.cap = 0;
    }
    return hipSuccess;
}

}  // namespace slimm
