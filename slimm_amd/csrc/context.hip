// Context, pipeline orchestration and the C ABI (include/slimm_hip.h) of the MI355X SLIMM path.
//
// One context = one `slimm` object working on one input file on one GPU (reference src/slimm.hpp:92-165).
// Device work is queued on a private stream; the only host<->device round trips of a run are
//   finish_coverage : per-reference {sum, non-zero} of cov / uniq_cov (16 bytes per reference)  -> host cut-offs
//   filter          : lineage rows with the valid bit up (16 bytes per reference); uniq2 / LCA counts / child marks down
// both through pinned host memory that a copy kernel reads / writes (no DMA-engine start-up latency).
// Multi-GPU entry points (coverage summary in all-gather or all-to-all form, device-side partials merge) are further down.
#include "context.h"

namespace slimm {

thread_local std::string g_create_error;  // (of the calling thread: a group creates its members side by side)
const char* kKernelNames[K_COUNT] = {"memset_bins", "k_group_count", "k_group_scan", "k_group_scatter", "k_group_finish",
                                     "k_front", "k_hist", "k_ref_stats", "k_filter", "k_ref_stats2",
                                     "k_tile_count", "k_tile_scan", "k_tile_scatter", "k_tile_hist",
                                     "k_tile_count2", "k_tile_scan2", "k_tile_scatter2", "k_tile_hist2",
                                     "k_pack", "k_pack2"};

int fail(slimm_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c)
        c->err = buf;
    else
        g_create_error = buf;
    return code;
}



void drain_events(slimm_ctx* c) {
    bool unrecorded = false;
    for (auto& e : c->ev_used) {
        float ms = 0.f;
        if (hipEventSynchronize(e.b) == hipSuccess && hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
            c->k_ms[e.id] += ms;
            c->k_n[e.id] += 1;
        } else {
            unrecorded = true;
        }
        c->ev_free.push_back(e);
    }
    c->ev_used.clear();
    // A pair that was never recorded (a launcher that had nothing to launch: a rank without records) makes
    // hipEventElapsedTime fail with "invalid resource handle" -- and leaves that as the thread's last error, which the next
    // slimm_analyze_alignments would report as its own (bench.py --gpus 2 with one chunk: the empty rank's second step).
    if (unrecorded) (void)hipGetLastError();
}

int ensure_work_buffers(slimm_ctx* c, uint32_t n) {
    HIP_TRY(c, c->tgt_ref.ensure(n + 1));
    HIP_TRY(c, c->tgt_gbin.ensure(n + 8));  // (+ the reach of the bucketing kernels' 16-byte loads, tile_hist.hip: piece_load)
    HIP_TRY(c, c->slots.ensure(front_slots(n) + 1));
    HIP_TRY(c, c->tot_part.ensure(512));
    HIP_TRY(c, c->sel.ensure(n + 8));
    HIP_TRY(c, c->slot_rbase.ensure(front_slots(n) + 8));
    HIP_TRY(c, c->slot_bbase.ensure(front_slots(n) / 1024 + 8));
    HIP_TRY(c, c->filter_redo.ensure(front_slots(n) + 1));
    if (c->use_tiles) {
        HIP_TRY(c, c->bucket.ensure(n + 1));
        HIP_TRY(c, c->tile_items.ensure(TILES(c->tile_shift, tile_items_upper(c->ntiles2, n)) + 1));
        HIP_TRY(c, c->mid.ensure(n + 1));
        HIP_TRY(c, c->part_items.ensure(TILES(c->tile_shift, part_items_upper(c->ntiles2, n)) + 1));
        HIP_TRY(c, c->sup_cursor.ensure(kMaxSuper));
    }
    if (c->order == SLIMM_ORDER_ANY) {
        HIP_TRY(c, c->c_ident.ensure(n + 1));
        HIP_TRY(c, c->c_pay.ensure(n + 1));
        HIP_TRY(c, c->s_ident.ensure(n + 1));
        HIP_TRY(c, c->s_pay.ensure(n + 1));
        HIP_TRY(c, c->group_hist.ensure(group_hist_words(group_plan(n))));
        if (c->rec.check) {
            HIP_TRY(c, c->c_chk.ensure(n + 1));
            HIP_TRY(c, c->s_chk.ensure(n + 1));
        }
    }
    return SLIMM_OK;
}

int ensure_pair_table(slimm_ctx* c, uint32_t cap) {
    if (cap <= c->pair_cap) return SLIMM_OK;
    HIP_TRY(c, c->pair_tab.ensure(cap));
    HIP_TRY(c, c->pair_list.ensure(cap / 2 + 1));
    HIP_TRY(c, c->h_pairs.ensure(cap / 2 + 1));
    c->pair_cap = cap;
    return SLIMM_OK;
}



int check_device_errors(slimm_ctx* c, uint32_t err) {
    if (err & ERR_REF_RANGE) return fail(c, SLIMM_E_REF_RANGE, "a record names a reference id >= n_refs");
    if (err & ERR_KEY_COLLISION)
        return fail(c, SLIMM_E_KEY_COLLISION, "two records with one read key carry different check words: two read names collide in the key");
    return SLIMM_OK;
}

}  // namespace slimm

extern "C" {

// The first HIP call of a process pays for the runtime's start-up (0.1 - 0.3 s); a host that has other work to do first
// (loading its database, opening its input) calls this from a thread of its own meanwhile.
int slimm_warm_up(int device) {
    if (device < 0) return SLIMM_OK;
    if (hipSetDevice(device) != hipSuccess) return SLIMM_E_HIP;
    if (hipFree(nullptr) != hipSuccess) return SLIMM_E_HIP;
    return SLIMM_OK;
}

// For a process about to leave: hipDeviceReset() of every device the calling process used -- queues and memory go back in
// one step instead of at the kernel driver's pace after exit (a host that ends with _exit measures which is faster).
int slimm_shutdown(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return SLIMM_E_HIP;
    int rc = SLIMM_OK;
    for (int d = 0; d < n; ++d)
        if (hipSetDevice(d) != hipSuccess || hipDeviceReset() != hipSuccess) rc = SLIMM_E_HIP;
    return rc;
}

const char* slimm_version(void) { return "slimm_hip 0.1 (gfx950)"; }

const char* slimm_last_error(const slimm_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int slimm_create(const slimm_config* cfg, slimm_ctx** out) {
    if (!cfg || !out) return fail(nullptr, SLIMM_E_INVALID, "null argument");
    *out = nullptr;
    if (cfg->n_refs == 0 || !cfg->ref_len || !cfg->lineage) return fail(nullptr, SLIMM_E_INVALID, "no references");
    if (cfg->n_refs > kMaxRefs) return fail(nullptr, SLIMM_E_INVALID, "too many references (limit 2^26 - 1)");
    if (cfg->bin_width == 0 && cfg->avg_read_len == 0)
        return fail(nullptr, SLIMM_E_INVALID, "bin_width and avg_read_len are both 0 (the reference divides by zero)");
    HostConfig hc;
    hc.n_refs = cfg->n_refs;
    hc.ref_len.assign(cfg->ref_len, cfg->ref_len + cfg->n_refs);
    hc.lineage.assign(cfg->lineage, cfg->lineage + static_cast<size_t>(cfg->n_refs) * 8);
    hc.bin_width = cfg->bin_width;
    hc.avg_read_len = cfg->avg_read_len;
    hc.min_reads = cfg->min_reads;
    hc.cov_cut_off = cfg->cov_cut_off;
    hc.abundance_cut_off = cfg->abundance_cut_off;
    hc.rank = cfg->rank ? cfg->rank : "species";
    {
        uint32_t rk = rank_from_string(hc.rank);
        if (rk < 1 || rk > 6)  // "strains", "superkingdom" and "all" are broken in the reference (Q14)
            return fail(nullptr, SLIMM_E_INVALID, "rank must be one of species, genus, family, order, class, phylum");
    }
    if (cfg->n_taxa) {
        if (!cfg->tax_id || !cfg->tax_rank || !cfg->tax_name) return fail(nullptr, SLIMM_E_INVALID, "taxa arrays missing");
        hc.tax_id.assign(cfg->tax_id, cfg->tax_id + cfg->n_taxa);
        hc.tax_rank.assign(cfg->tax_rank, cfg->tax_rank + cfg->n_taxa);
        hc.tax_name.reserve(cfg->n_taxa);
        for (uint32_t i = 0; i < cfg->n_taxa; ++i) hc.tax_name.emplace_back(cfg->tax_name[i] ? cfg->tax_name[i] : "");
    }
    std::unique_ptr<slimm_ctx> c(new slimm_ctx());
    c->host.reset(new HostProfile(hc));
    c->R = cfg->n_refs;
    c->T = c->host->n_taxa_dense();
    c->device = cfg->device;
    c->order = cfg->record_order;
    // padded bin layout: every reference starts on a 16-byte boundary
    c->bin_off_h.resize(c->R + 1);
    uint64_t off = 0;
    for (uint32_t r = 0; r < c->R; ++r) {
        c->bin_off_h[r] = static_cast<uint32_t>(off);
        off += (static_cast<uint64_t>(c->host->nbins()[r]) + 3) & ~3ull;
        if (off >= kMaxBins) return fail(nullptr, SLIMM_E_INVALID, "more than 2^31 coverage bins; use a larger bin width");
    }
    c->bin_off_h[c->R] = static_cast<uint32_t>(off);
    // The taxon part of a selector: with 16-byte lineage rows k_filter names an LCA by (level, index in the level) -- what
    // the rows hold -- and the second tile histogram counts per (level << taxon_shift | index); k_pack sums those counts
    // into per-taxon ones (kernels.h: PackArgs::sum_k).  No taxon look-up, one dependent round trip less, on k_filter's
    // path.  With the 32-byte rows the selector is the dense taxon itself.
    {
        c->use_rows16 = c->host->rows16_ok() && !forced("wide_rows");
        c->Tsel = c->T;
        if (c->use_rows16) {
            const uint32_t* loff = c->host->level_offset();
            uint32_t widest = 1;
            for (int l = 0; l < 8; ++l) widest = std::max(widest, loff[l + 1] - loff[l]);
            c->taxon_shift = 0;
            while ((1u << c->taxon_shift) < widest) ++c->taxon_shift;
            c->Tsel = 8u << c->taxon_shift;
        }
    }
    // Tile size by layout: the small tiles while the whole bucketing fits the fused kernel's tile tables (their histograms
    // run at twice the occupancy), the large ones beyond -- where the scatter's direct rounds pay a returning atomic per
    // run of one tile among neighbouring targets and half as many tiles make those runs longer (1 B records over 20 k
    // references: scatter 3.03 -> 2.3 ms, histograms 0.41 -> 0.67 ms; kernels.h)
    {
        const uint64_t small = (off + (1ull << kTileShiftSmall) - 1) >> kTileShiftSmall;
        const uint64_t small2 = small + ((static_cast<uint64_t>(c->Tsel) + (1ull << kTileShiftSmall) - 1) >> kTileShiftSmall);
        c->tile_shift = small2 > kFusedScanTiles ? kTileShiftLarge : kTileShiftSmall;
        long ts = 0;
        if (forced("tile_shift", &ts)) {
            if (ts == static_cast<long>(kTileShiftSmall)) c->tile_shift = kTileShiftSmall;
            if (ts == static_cast<long>(kTileShiftLarge)) c->tile_shift = kTileShiftLarge;
        }
    }
    const uint32_t tile_bins = c->tile_bins();
    c->Bp = (off + tile_bins - 1) & ~static_cast<uint64_t>(tile_bins - 1);
    c->ntiles = static_cast<uint32_t>(c->Bp >> c->tile_shift);
    c->Tpad = (c->Tsel + tile_bins - 1) & ~(tile_bins - 1);
    c->ntiles2 = c->ntiles + (c->Tpad >> c->tile_shift);
    if (c->Bp + c->Tpad >= kMaxBins) return fail(nullptr, SLIMM_E_INVALID, "more than 2^31 coverage bins; use a larger bin width");

    if (c->device >= 0) {
        slimm_ctx* cc = c.get();
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev == 0)
            return fail(nullptr, SLIMM_E_HIP, "no HIP device available (%s)", hipGetErrorString(e));
        if (c->device >= ndev) return fail(nullptr, SLIMM_E_INVALID, "device %d out of range (%d devices)", c->device, ndev);
#define HIP_TRY0(expr)                                                                                   \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) return fail(nullptr, SLIMM_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)
        HostTrace trc("slimm_create");
        HIP_TRY0(hipSetDevice(c->device));
        HIP_TRY0(hipStreamCreateWithFlags(&cc->stream, hipStreamNonBlocking));
        HIP_TRY0(hipEventCreateWithFlags(&cc->front_done, hipEventDisableTiming));
        HIP_TRY0(hipEventCreateWithFlags(&cc->prefix_done, hipEventDisableTiming));
        HIP_TRY0(hipEventCreateWithFlags(&cc->copy_done, hipEventDisableTiming));
        for (auto& sg : cc->staging) HIP_TRY0(hipEventCreateWithFlags(&sg.done, hipEventDisableTiming));
        trc.mark("streams + events");
        HIP_TRY0(cc->d_ref_len.ensure(c->R));
        HIP_TRY0(cc->d_bin_off.ensure(c->R + 1));
        HIP_TRY0(cc->d_lin_dense.ensure(static_cast<size_t>(c->R) * 8));
        HIP_TRY0(cc->d_valid.ensure(c->R));
        HIP_TRY0(cc->d_valid_bits.ensure(c->R / 32 + 2));
        HIP_TRY0(cc->h_valid_bits.ensure(c->R / 32 + 2));
        memset(cc->h_valid_bits.p, 0, (c->R / 32 + 2) * 4);
        HIP_TRY0(cc->bins.ensure(3 * c->Bp + kTailWords + c->Tpad));
        HIP_TRY0(cc->counters.ensure(CNT_WORDS));
        HIP_TRY0(cc->ref_stats.ensure(c->statsA_words() + c->statsB_words() + 64));
        HIP_TRY0(cc->lca_count.ensure(c->Tsel));
        HIP_TRY0(cc->marks.ensure(static_cast<size_t>(c->R) * (kMarkBytes / 4)));
        trc.mark("device arrays");
        HIP_TRY0(cc->h_stats.ensure(c->statsA_words() + c->statsB_words() + 64));
        HIP_TRY0(cc->h_small.ensure(CNT_WORDS + kTailWords));
        HIP_TRY0(cc->h_lca.ensure(c->T));
        HIP_TRY0(cc->h_marks.ensure(c->R));
        trc.mark("page-locked host arrays");
        HIP_TRY0(hipMemcpy(cc->d_ref_len.p, hc.ref_len.data(), c->R * 4, hipMemcpyHostToDevice));
        HIP_TRY0(hipMemcpy(cc->d_bin_off.p, c->bin_off_h.data(), (c->R + 1) * 4, hipMemcpyHostToDevice));
        {
            std::vector<uint2> geo(c->R);
            for (uint32_t i = 0; i < c->R; ++i) geo[i] = make_uint2(hc.ref_len[i], static_cast<uint32_t>(c->bin_off_h[i]));
            HIP_TRY0(cc->d_geo.ensure(c->R));
            HIP_TRY0(hipMemcpy(cc->d_geo.p, geo.data(), c->R * sizeof(uint2), hipMemcpyHostToDevice));
        }
        HIP_TRY0(hipMemcpy(cc->d_lin_dense.p, c->host->lineage_dense().data(), static_cast<size_t>(c->R) * 32,
                           hipMemcpyHostToDevice));
#undef HIP_TRY0
        if (cc->use_rows16) {
            // the dense taxon of a (level, index): one table with a power-of-two stride per level, so that a lane turns
            // its (level, index) into an address with one shift-or (only the rare no-level-agrees path looks it up on the
            // device) -- and, for k_pack, its inverse: the (level, index) entries of every dense taxon (a taxid may stand on
            // several levels of a lineage, e.g. a species-level accession)
            const std::vector<uint32_t>& lt = c->host->level_taxon();
            const uint32_t* off = c->host->level_offset();
            std::vector<uint32_t> flat(static_cast<size_t>(8) << cc->taxon_shift, 0u);
            std::vector<uint32_t> inv_off(c->T + 1, 0u), inv_idx;
            for (int l = 0; l < 8; ++l)
                for (uint32_t i = off[l]; i < off[l + 1]; ++i) {
                    flat[(static_cast<size_t>(i - off[l]) << 3) | static_cast<size_t>(l)] = lt[i];
                    if (lt[i] < c->T) ++inv_off[lt[i] + 1];
                }
            for (uint32_t t = 0; t < c->T; ++t) inv_off[t + 1] += inv_off[t];
            inv_idx.resize(inv_off[c->T] + 1);
            {
                std::vector<uint32_t> at(inv_off.begin(), inv_off.end() - 1);
                for (int l = 0; l < 8; ++l)
                    for (uint32_t i = off[l]; i < off[l + 1]; ++i)
                        if (lt[i] < c->T) inv_idx[at[lt[i]]++] = ((i - off[l]) << 3) | static_cast<uint32_t>(l);
            }
            if (cc->d_rows16.ensure(c->R) != hipSuccess || cc->h_rows16.ensure(c->R) != hipSuccess ||
                cc->d_level_taxon.ensure(flat.size()) != hipSuccess ||
                hipMemcpy(cc->d_level_taxon.p, flat.data(), flat.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
                cc->d_taxon_off.ensure(inv_off.size()) != hipSuccess || cc->d_taxon_idx.ensure(inv_idx.size()) != hipSuccess ||
                hipMemcpy(cc->d_taxon_off.p, inv_off.data(), inv_off.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(cc->d_taxon_idx.p, inv_idx.data(), inv_idx.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
                return fail(nullptr, SLIMM_E_HIP, "out of device memory for lineage rows");
        }
        trc.mark("tables to the device");
        cc->use_tiles = !forced("direct_atomics") && TILES(c->tile_shift, tile_hist_setup(c->ntiles2)) == 0;
        if (cc->order == SLIMM_ORDER_ANY && group_init() != 0) {
            *out = nullptr;
            return fail(nullptr, SLIMM_E_HIP, "group_init: hipFuncSetAttribute failed");
        }
        trc.mark("kernel attributes (tile_hist_setup, group_init: the code objects are loaded here)");
        {
            // default by size: with the register-resident scatter chunks one level wins up to ~10 K tiles (config 3:
            // 494 vs 601 us) and is level with two at 24 K (config 5: 430 vs 396 us)
            long tl = 0;
            cc->two_level = forced("two_level", &tl) ? tl == 1 : (c->ntiles2 > 16384);
        }
        if (cc->use_tiles) {
            cc->treps = (cc->two_level || c->ntiles2 > 16384) ? 1u : kTileReps;  // (k_tile_scan stages the copies of <= 16 K tiles)
            cc->tstride = c->ntiles2 + 1;
            {
                long fs = 1;
                (void)forced("fused_scan", &fs);
                cc->fused_scan = !cc->two_level && cc->treps == kTileReps && c->ntiles2 <= kFusedScanTiles && fs != 0;
            }
            {
                // Layouts beyond the fused kernel's 4064 tiles, PHASE B only, when a tile gets few selectors (decided per
                // file from the number of reads): a count matrix instead of counter copies and cursors -- one row of tile
                // counts per counting workgroup, no global atomics and no rounds in the scatter.  (Phase A stays on the
                // rounds: with hundreds of thousands of values per tile, appending at ONE frontier per tile and counter
                // copy keeps the bucket's open cache lines in L2, while 256 private frontiers per tile -- 2.5 M partially
                // written lines at config 3 -- turn every 2-byte store into a partial-line write: measured 834 vs 362 us.)
                // Since round 5 only beyond kBigRoundTiles tiles: up to there the rounds ordered by tile in LDS
                // (k_tile_scatter_big) are faster for both phases (config 3, phase B: count 28 -> 17 us, scatter 52 -> 42).
                long mx = 1;
                (void)forced("matrix", &mx);
                cc->matrix = !cc->fused_scan && mx != 0 && (c->ntiles2 > kBigRoundTiles || mx == 2);
                if (cc->matrix &&
                    cc->tile_matrix.ensure(static_cast<size_t>(TILES(c->tile_shift, tile_count_grid(512))) * cc->tstride) != hipSuccess)
                    return fail(nullptr, SLIMM_E_HIP, "out of device memory for the tile count matrix");
                if (mx == 2) cc->matrix_always = true;  // (tests: small layouts, whatever the number of reads)
                long wt = 0;
                if (forced("wide_tiles", &wt)) cc->wide_tiles = wt == 1 ? 1 : 0;
            }
            const size_t rep_words = static_cast<size_t>(cc->treps) * cc->tstride;
            if (cc->tile_count.ensure(rep_words) != hipSuccess || cc->tile_base.ensure(c->ntiles2 + 1) != hipSuccess ||
                cc->tile_cursor.ensure(rep_words) != hipSuccess || cc->split_tiles.ensure(c->ntiles2 + 1) != hipSuccess)
                return fail(nullptr, SLIMM_E_HIP, "out of device memory for tile tables");
            std::vector<uint32_t> ref0(c->ntiles2 + 1, c->R);
            uint32_t r = 0;
            for (uint32_t t = 0; t < c->ntiles2; ++t) {
                const uint64_t t0 = static_cast<uint64_t>(t) * tile_bins;
                if (t0 >= c->bin_off_h[c->R]) break;
                while (r + 1 < c->R && c->bin_off_h[r + 1] <= t0) ++r;
                ref0[t] = r;
            }
            if (cc->d_tile_ref0.ensure(ref0.size()) != hipSuccess ||
                hipMemcpy(cc->d_tile_ref0.p, ref0.data(), ref0.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
                return fail(nullptr, SLIMM_E_HIP, "out of device memory for tile tables");
        }
        uint32_t cap = 1u << 16;
        while (cap < 8ull * c->R && cap < (1u << 30)) cap <<= 1;
        long pc = 0;
        if (forced("pair_cap", &pc)) {  // tests: start small, so that the overflow -> grow -> retry path runs
            cap = 16;
            while (cap < static_cast<uint32_t>(pc) && cap < (1u << 30)) cap <<= 1;
        }
        int rc = ensure_pair_table(cc, cap);
        if (rc != SLIMM_OK) {
            g_create_error = cc->err;
            return rc;
        }
    }
    *out = c.release();
    return SLIMM_OK;
}

void slimm_destroy(slimm_ctx* c) {
    if (!c) return;
    if (c->device >= 0) {
        (void)hipSetDevice(c->device);
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        for (auto& e : c->ev_used) {
            (void)hipEventDestroy(e.a);
            (void)hipEventDestroy(e.b);
        }
        for (auto& e : c->ev_free) {
            (void)hipEventDestroy(e.a);
            (void)hipEventDestroy(e.b);
        }
        if (c->side_stream) {
            (void)hipStreamSynchronize(c->side_stream);
            (void)hipStreamDestroy(c->side_stream);
        }
        for (auto& is : c->bam.inflate_stream)
            if (is) {
                (void)hipStreamSynchronize(is);
                (void)hipStreamDestroy(is);
            }
        if (c->bam.comp_copied) (void)hipEventDestroy(c->bam.comp_copied);
        if (c->front_done) (void)hipEventDestroy(c->front_done);
        if (c->prefix_done) (void)hipEventDestroy(c->prefix_done);
        if (c->copy_stream) {
            (void)hipStreamSynchronize(c->copy_stream);
            (void)hipStreamDestroy(c->copy_stream);
        }
        if (c->copy_done) (void)hipEventDestroy(c->copy_done);
        for (auto& r : c->bam.registered) (void)hipHostUnregister(const_cast<uint8_t*>(r.first));
        for (void* p : c->bam.outgrown) (void)hipFree(p);
        for (auto& e : c->bam.copied)
            if (e) (void)hipEventDestroy(e);
        for (auto& e : c->bam.h2d_done)
            if (e) (void)hipEventDestroy(e);
        for (auto& sg : c->staging)
            if (sg.done) (void)hipEventDestroy(sg.done);
        if (c->stream) (void)hipStreamDestroy(c->stream);
    }
    delete c;
}

int slimm_reset(slimm_ctx* c) {
    if (!c) return SLIMM_E_INVALID;
    if (c->copy_pending) {  // copies still on their way would land in the next file's records
        (void)hipSetDevice(c->device);
        HIP_TRY(c, hipStreamSynchronize(c->copy_stream));
        c->copy_pending = false;
        for (auto& sg : c->staging) sg.pending = false;
    }
    if (c->bam.active && c->bam.head < c->bam.windows) {  // a file abandoned with windows in flight (an error, a caller's
        (void)hipSetDevice(c->device);                    // change of mind): their copies and inflates must not land in the
        if (c->copy_stream) HIP_TRY(c, hipStreamSynchronize(c->copy_stream));  // next file's buffers
        for (auto& is : c->bam.inflate_stream)
            if (is) HIP_TRY(c, hipStreamSynchronize(is));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    if (c->bam.acc_open) {  // (a gathered window that was never launched: its copies)
        (void)hipSetDevice(c->device);
        if (c->copy_stream) HIP_TRY(c, hipStreamSynchronize(c->copy_stream));
    }
    c->bam.acc_open = false;
    c->bam.acc_src = c->bam.acc_dst = 0;
    c->bam.acc_skip = 0;
    c->bam.pushes = 0;
    for (void* p : c->bam.outgrown) (void)hipFree(p);
    c->bam.outgrown.clear();
    c->host->reset();
    c->analyzed = c->covered = c->filtered = c->counted = c->no_hits = false;
    c->n_pushed = 0;
    c->bam.windows = 0;
    c->bam.carry_bytes = 0;
    c->bam.active = false;
    c->bam.head = 0;
    c->bam.closed = false;
    c->bam.q18_starts = c->bam.q18_plain = 0;
    c->bam.size_hint = c->bam.win_cap = 0;
    c->bam.planned = false;
    if (c->bam.held_bytes() > slimm_ctx::kBamKeepAcrossFiles) {   // (a large file's windows: the next file sizes its own)
        (void)hipSetDevice(c->device);
        for (uint32_t k = 0; k < slimm_ctx::kBamRing; ++k) {
            c->bam.bytes[k].release();
            c->bam.comp[k].release();
            c->bam.desc[k].release();
        }
        for (auto& sc : c->bam.inflate_scratch) sc.release();
        c->bam.pieces.release();
        c->bam.offs.release();
    }
    c->has_check = false;
    c->packed = false;
    c->marked = false;
    c->borrowed = false;
    c->rec = DeviceRecords();
    c->local_V = c->local_M = c->local_P = 0;
    c->n_pairs = 0;
    return SLIMM_OK;
}

int slimm_reset_cutoffs(slimm_ctx* c) {
    if (!c) return SLIMM_E_INVALID;
    c->host->reset_cutoffs();
    return SLIMM_OK;
}

int slimm_get_cutoff_cache(slimm_ctx* c, float* cc, float* ucc) {
    if (!c || !cc || !ucc) return SLIMM_E_INVALID;
    c->host->get_cutoff_cache(*cc, *ucc);
    return SLIMM_OK;
}

int slimm_set_cutoff_cache(slimm_ctx* c, float cc, float ucc) {
    if (!c) return SLIMM_E_INVALID;
    c->host->set_cutoff_cache(cc, ucc);
    return SLIMM_OK;
}

int slimm_set_min_reads(slimm_ctx* c, uint32_t min_reads) {
    if (!c) return SLIMM_E_INVALID;
    c->host->min_reads = min_reads;
    return SLIMM_OK;
}

// Is the stream really grouped by read name?  See k_check_grouping (kernels.hip).
int slimm_check_grouping(slimm_ctx* c, uint64_t* n_split_names) {
    if (!c || !n_split_names) return SLIMM_E_INVALID;
    if (c->device < 0) return fail(c, SLIMM_E_INVALID, "host-only context has no record stream");
    *n_split_names = 0;
    const uint32_t n = c->rec.n;
    if (n == 0) return SLIMM_OK;
    if (c->rec.marked) return fail(c, SLIMM_E_INVALID, "run-marked records carry no read names to check");
    (void)hipSetDevice(c->device);
    if (c->copy_pending) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->copy_done, 0));
    uint64_t cap = 1024;
    while (cap < 2ull * n) cap <<= 1;  // (every record may start a run)
    DevBuf<uint64_t> tab;
    DevBuf<uint32_t> cnt;
    HIP_TRY(c, tab.ensure(cap));
    HIP_TRY(c, cnt.ensure(1));
    HIP_TRY(c, hipMemsetAsync(tab.p, 0xff, cap * 8, c->stream));
    HIP_TRY(c, hipMemsetAsync(cnt.p, 0, 4, c->stream));
    launch_check_grouping(c->stream, c->rec.key, n, c->rec.packed ? (1ull << 61) - 1ull : (1ull << 62) - 1ull, tab.p,
                          static_cast<uint32_t>(cap - 1), cnt.p);
    uint32_t h = 0;
    HIP_TRY(c, hipMemcpyAsync(&h, cnt.p, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *n_split_names = h;
    return SLIMM_OK;
}

// ------------------------------------------------------------------------------------------------ phase A
int slimm_analyze_alignments(slimm_ctx* c) {
    if (!c) return SLIMM_E_INVALID;
    if (c->device < 0) return fail(c, SLIMM_E_INVALID, "host-only context: install results with slimm_set_coverage_columns");
    if (c->analyzed) return fail(c, SLIMM_E_INVALID, "already analysed; reset first");
    HostTrace tr("analyze_alignments");
    (void)hipSetDevice(c->device);
    (void)hipGetLastError();  // (the launches below answer for themselves: what another user of the runtime left in this thread is not theirs)
    if (c->bam.active) {
        // (windows pushed so far may still be gathered or in flight: analysing now would profile a truncated record stream)
        if (!c->bam.closed)
            return fail(c, SLIMM_E_INVALID, "the file's last window has not been pushed (slimm_push_bam_bytes / _bgzf_blocks / _sam_bytes with last != 0)");
        if (c->order == SLIMM_ORDER_GROUPED) {  // Q18: a run of shortened names only may have its flagged namesakes elsewhere
            const int qrc = bam_fetch_q18(c);
            if (qrc != SLIMM_OK) return qrc;
            if (c->bam.q18_starts != c->bam.q18_plain)
                return fail(c, SLIMM_E_REGROUP,
                            "read names ending in .1 / .2 without a mate flag, apart from the flagged records of the shortened name: "
                            "push this file to a context created with SLIMM_ORDER_ANY");
        }
    }
    const uint32_t n = c->rec.n;
    int rc = ensure_work_buffers(c, n);
    if (rc != SLIMM_OK) return rc;
    tr.mark("set device + buffers");
    hipStream_t st = c->stream;
    if (c->copy_pending) HIP_TRY(c, hipStreamWaitEvent(st, c->copy_done, 0));  // streamed ingest: device-side ordering
    if (c->prefix_pending) {  // the file before left its slot prefix on the side stream: k_front rewrites the slots
        HIP_TRY(c, hipStreamWaitEvent(st, c->prefix_done, 0));
        c->prefix_pending = false;
    }
    {
        KernelTimer t(c, K_MEMSET);
        if (!c->use_tiles)  // (the tile kernels write every cov / uniq_cov word themselves)
            HIP_TRY(c, hipMemsetAsync(c->bins.p, 0, 2 * c->Bp * sizeof(uint32_t), st));
        ZeroArgs z;
        z.p[0] = c->counters.p;
        z.n[0] = CNT_WORDS;
        z.p[1] = c->tail();
        z.n[1] = kTailWords;
        if (c->use_tiles) {
            z.p[2] = c->tile_count.p;
            z.n[2] = c->treps * c->tstride;
            z.p[3] = c->ref_stats.p;  // per-reference statistics accumulated by k_tile_hist
            z.n[3] = 4 * c->R;
            if (c->fused_scan) {      // (otherwise k_tile_scan clears the cursors)
                z.p[4] = c->tile_cursor.p;
                z.n[4] = c->treps * c->tstride;
            }
        }
        launch_zero(st, z);
    }
    c->bins_exposed = false;
    c->statsA_final = false;
    const HostConfig& hc = c->host->config();
    const uint32_t half_read = hc.avg_read_len / 2;
    const uint32_t nslots = front_slots(n);
    if (c->order == SLIMM_ORDER_ANY) {
        // the records of every read identity adjacent, file order kept among them (group_by_ident.hip): counting passes
        // over a few hash bits -- the first one straight over the caller's records, filter and bin on the way --, a
        // finish inside the small buckets, then the same front end as for grouped input
        GroupJob j;
        j.plan = group_plan(n);
        j.in = c->rec;
        j.n_refs = c->R;
        j.geo = c->d_geo.p;
        j.half_read = half_read;
        j.bin_width = hc.bin_width;
        j.counters = c->counters.p;
        j.a = GroupArrays{c->c_ident.p, c->c_pay.p, c->rec.check ? c->c_chk.p : nullptr};
        j.t = GroupArrays{c->s_ident.p, c->s_pay.p, c->rec.check ? c->s_chk.p : nullptr};
        j.hist = c->group_hist.p;
        for (uint32_t pass = 0; pass < j.plan.passes; ++pass) {
            {
                KernelTimer t(c, K_GROUP_COUNT);
                launch_group_count(st, j, pass);
            }
            {
                KernelTimer t(c, K_GROUP_SCAN);
                launch_group_scan(st, j, pass);
            }
            {
                KernelTimer t(c, K_GROUP_SCATTER);
                launch_group_scatter(st, j, pass);
            }
        }
        {
            KernelTimer t(c, K_GROUP_FINISH);
            launch_group_finish(st, j);
        }
        {
            KernelTimer t(c, K_FRONT);
            launch_front_sorted(st, n, c->c_ident.p, c->c_pay.p, c->counters.p, c->tgt_ref.p, c->tgt_gbin.p, c->slots.p,
                                c->rec.check ? c->c_chk.p : nullptr);
        }
    } else {
        // grouped input: one pass straight over the caller's record arrays
        KernelTimer t(c, K_FRONT, true);
        launch_front_raw(st, c->rec, c->R, c->d_geo.p, half_read, hc.bin_width, c->counters.p, c->tgt_ref.p, c->tgt_gbin.p,
                         c->slots.p, t.t0(), t.t1());
    }
    // Where every slot's reads start among all reads (for k_filter's dense selectors): two small launches on a side
    // stream -- beside k_tile_hist, NOT beside the count and the scatter: those run one persistent workgroup per CU
    // (half CU), and a CU that is busy with a prefix workgroup when they are placed makes another CU take two of them
    // (k_tile_count 456 -> 610-630 us at 1 B records in five runs of six).  The histogram's workgroups are many and short.
    auto slot_prefix_beside_what_follows = [&]() -> int {
        HIP_TRY(c, need_stream(c->side_stream));
        HIP_TRY(c, hipEventRecord(c->front_done, st));
        HIP_TRY(c, hipStreamWaitEvent(c->side_stream, c->front_done, 0));
        launch_slot_read_prefix(c->side_stream, c->slots.p, nslots, c->slot_rbase.p, c->slot_bbase.p);
        HIP_TRY(c, hipEventRecord(c->prefix_done, c->side_stream));
        return SLIMM_OK;
    };
    SlotValues targets;
    targets.vals = c->tgt_gbin.p;
    targets.slots = c->slots.p;
    targets.nslots = nslots;
    targets.per_read = false;
    const bool wide = c->wide_for(n);
    const uint32_t tile_sub = wide ? kTileSubWide : kTileSub;
    if (c->use_tiles) {
        const uint32_t grid = 512;  // two persistent workgroups per CU
        {
            KernelTimer t(c, K_TILE_COUNT);
            TILES(c->tile_shift, launch_tile_count(st, grid, c->ntiles, targets, c->tot_part.p, c->tile_count.p, c->treps, c->tstride));
        }
        Totals tot;
        tot.part = c->tot_part.p;
        tot.nparts = TILES(c->tile_shift, tile_count_grid(grid));
        tot.tail = c->tail();
        if (c->fused_scan) {
            KernelTimer t(c, K_TILE_SCATTER);
            TILES(c->tile_shift, launch_tile_scatter_fused(st, grid, c->ntiles, targets, c->counters.p, c->tile_count.p, c->tile_cursor.p,
                                      c->bucket.p, c->cov(), c->ucov(), c->tstride, c->tile_items.p, c->split_tiles.p, tot,
                                      tile_sub));
        } else {
            {
                KernelTimer t(c, K_TILE_SCAN);
                TILES(c->tile_shift, launch_tile_scan(st, c->ntiles, c->tile_count.p, c->tile_base.p, c->tile_cursor.p, c->tile_items.p,
                                 c->counters.p, c->part_items.p, c->sup_cursor.p, c->split_tiles.p, c->treps, c->tstride,
                                 c->two_level, tot, tile_sub));
            }
            {
                KernelTimer t(c, K_TILE_SCATTER);
                TILES(c->tile_shift, launch_tile_scatter(st, grid, c->ntiles, n, targets, c->counters.p, c->tile_base.p, c->tile_cursor.p,
                                    c->sup_cursor.p, c->part_items.p, c->mid.p, c->bucket.p, c->cov(), c->ucov(),
                                    c->two_level, c->tile_count.p, c->treps, c->tstride, tile_sub));
            }
        }
        if (int rc = slot_prefix_beside_what_follows()) return rc;
        {
            KernelTimer t(c, K_TILE_HIST);
            c->summary_has_bits = false;
            if (c->summary_slices) {
                const bool fresh = c->summary.cap < c->summary_words() || c->summary_layout != c->summary_slices;
                HIP_TRY(c, c->summary.ensure(c->summary_words()));
                if (fresh)  // the padding behind the last tile of the last slice is never written: zero it once per layout
                    HIP_TRY(c, hipMemsetAsync(c->summary.p, 0, c->summary_words() * 4, st));
                c->summary_layout = c->summary_slices;
                c->summary_has_bits = true;
            }
            TILES(c->tile_shift, launch_tile_hist(st, c->ntiles, n, c->bucket.p, c->tile_base.p, c->tile_items.p, c->counters.p, c->cov(),
                             c->ucov(), c->d_bin_off.p, c->R, c->d_tile_ref0.p, c->ref_stats.p, c->bits_layout(),
                             c->keep_bins ? 0u : c->ntiles, wide));
            c->binsA_stored = c->keep_bins;
        }
    } else {
        if (int rc = slot_prefix_beside_what_follows()) return rc;
        KernelTimer t(c, K_HIST);
        launch_hist(st, c->tgt_gbin.p, c->slots.p, nslots, c->counters.p, c->tail(), c->cov(), c->ucov(),
                    c->order != SLIMM_ORDER_ANY);
        c->binsA_stored = true;
    }
    // (the side stream's prefix kernels are joined by phase B in front of k_filter and -- for a file that ends without one:
    // no hits, an analyse-only caller -- by the NEXT file's phase A in front of its first launch: a wait here would put a
    // cross-queue barrier of 11 - 13 us between k_tile_hist and k_pack for an event that is long done)
    c->prefix_pending = true;
    HIP_TRY(c, hipGetLastError());
    c->analyzed = true;
    tr.mark("phase A launches");
    return SLIMM_OK;
}

int slimm_get_stream(slimm_ctx* c, void** hip_stream) {
    if (!c || !hip_stream) return SLIMM_E_INVALID;
    if (c->device < 0) return fail(c, SLIMM_E_INVALID, "host-only context has no stream");
    *hip_stream = c->stream;
    return SLIMM_OK;
}

int slimm_set_stream_ordered(slimm_ctx* c, int on) {
    if (!c) return SLIMM_E_INVALID;
    c->stream_ordered = on != 0;
    return SLIMM_OK;
}

int slimm_coverage_buffer(slimm_ctx* c, void** d_ptr, uint64_t* n_words) {
    if (!c || !d_ptr || !n_words) return SLIMM_E_INVALID;
    if (!c->analyzed) return fail(c, SLIMM_E_INVALID, "call slimm_analyze_alignments first");
    if (!c->binsA_stored) return fail(c, SLIMM_E_INVALID, "the coverage arrays were not kept (slimm_keep_bins)");
    (void)hipSetDevice(c->device);
    if (!c->stream_ordered) HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->bins_exposed = true;  // the statistics k_tile_hist accumulated describe the local bins only
    *d_ptr = c->bins.p;
    *n_words = 2 * c->Bp + 16;
    return SLIMM_OK;
}

int slimm_uniq_cov2_buffer(slimm_ctx* c, void** d_ptr, uint64_t* n_words) {
    if (!c || !d_ptr || !n_words) return SLIMM_E_INVALID;
    if (c->device < 0) return fail(c, SLIMM_E_INVALID, "host-only context has no coverage arrays");
    if (!c->filtered) return fail(c, SLIMM_E_INVALID, "call slimm_filter_alignments first");
    if (!c->binsB_stored) return fail(c, SLIMM_E_INVALID, "the coverage arrays were not kept (slimm_keep_bins)");
    (void)hipSetDevice(c->device);
    if (!c->stream_ordered) HIP_TRY(c, hipStreamSynchronize(c->stream));
    *d_ptr = c->ucov2();
    *n_words = c->Bp;
    return SLIMM_OK;
}

namespace {
// shared end of phase A: the packed block A = [4R: {reads_count, nz_cov, uniq_reads_count, nz_uniq_cov} | 32 counters |
// 16 tail (hits, matches, targets, err)] is final on the device; one copy brings it to the host
int finish_from_device_stats(slimm_ctx* c) {
    hipStream_t st = c->stream;
    const size_t R4 = 4ull * c->R;
    HostTrace tr("finish_coverage");
    launch_copy_out(st, c->h_stats.p, c->ref_stats.p, static_cast<uint32_t>(R4 + 48));  // pinned memory is host-mapped
    tr.mark("copy-out launch");
    HIP_TRY(c, hipStreamSynchronize(st));
    tr.mark("stream sync (device phase A + copy)");
    const uint32_t* cnt = c->h_stats.p + R4;
    const uint32_t* tl = c->h_stats.p + R4 + 32;
    int rc = check_device_errors(c, cnt[CNT_ERR] | tl[3]);
    if (rc != SLIMM_OK) return rc;
    c->local_V = cnt[CNT_V];
    c->local_M = cnt[CNT_M];
    c->local_P = cnt[CNT_P];
    // packed rows {reads_count = sum of cov bins, non-zero cov, uniq_reads_count = sum of uniq_cov bins, non-zero uniq_cov}
    const uint32_t* st4 = c->h_stats.p;
    c->host->set_coverage_strided(st4 + 0, st4 + 2, st4 + 1, st4 + 3, 4, tl[0], tl[1]);
    tr.mark("columns + set_coverage");
    c->covered = true;
    c->no_hits = (tl[0] == 0);
    return c->no_hits ? SLIMM_E_NO_HITS : SLIMM_OK;
}
}  // namespace

int slimm_finish_coverage(slimm_ctx* c) {
    if (!c) return SLIMM_E_INVALID;
    if (!c->analyzed) return fail(c, SLIMM_E_INVALID, "call slimm_analyze_alignments first");
    (void)hipSetDevice(c->device);
    PackArgs pk;
    pk.src[0] = c->counters.p;
    pk.n[0] = 32;
    pk.src[1] = c->tail();
    pk.n[1] = 16;
    if (c->use_tiles && !c->bins_exposed) {  // k_tile_hist left the per-reference statistics in place
        KernelTimer t(c, K_PACK);
        TILES(c->tile_shift, launch_pack(c->stream, c->ref_stats.p + 4ull * c->R, pk, c->split_tiles.p, c->counters.p, c->cov(), c->ucov(),
                    c->d_bin_off.p, c->R, c->d_tile_ref0.p, c->statsA_final ? nullptr : c->ref_stats.p, c->bits_layout()));
        c->statsA_final = true;
    } else {
        KernelTimer t(c, K_REF_STATS);
        launch_ref_stats(c->stream, c->cov(), c->ucov(), c->d_bin_off.p, c->R, c->ref_stats.p, &pk);
    }
    return finish_from_device_stats(c);
}

int slimm_keep_bins(slimm_ctx* c, int on) {
    if (!c) return SLIMM_E_INVALID;
    if (c->analyzed) return fail(c, SLIMM_E_INVALID, "slimm_keep_bins: before slimm_analyze_alignments");
    c->keep_bins = on != 0;
    return SLIMM_OK;
}

int slimm_prepare_summary(slimm_ctx* c, uint32_t n_slices) {
    if (!c) return SLIMM_E_INVALID;
    if (n_slices > 1 && !c->use_tiles) return fail(c, SLIMM_E_INVALID, "sliced summaries need the tile histogram path");
    c->summary_slices = (c->device >= 0 && c->use_tiles) ? n_slices : 0;
    return SLIMM_OK;
}

int slimm_coverage_summary(slimm_ctx* c, void** d_ptr, uint64_t* n_words) {
    if (!c || !d_ptr || !n_words) return SLIMM_E_INVALID;
    if (!c->analyzed) return fail(c, SLIMM_E_INVALID, "call slimm_analyze_alignments first");
    (void)hipSetDevice(c->device);
    if (c->summary_slices > 1 && (!c->summary_has_bits || c->bins_exposed))
        return fail(c, SLIMM_E_INVALID, "sliced summary: call slimm_prepare_summary before slimm_analyze_alignments");
    const uint64_t bits_words = c->Bp / 32;
    const uint64_t W = c->summary_words();
    HIP_TRY(c, c->summary.ensure(W));
    hipStream_t st = c->stream;
    bool have_bits = false;  // k_tile_hist / k_pack already wrote the bitmaps into the summary buffer
    if (c->use_tiles && !c->bins_exposed) {
        KernelTimer t(c, K_PACK);
        PackArgs pk;
        pk.src[0] = c->ref_stats.p;
        pk.n[0] = 4 * c->R;
        pk.src[1] = c->tail();
        pk.n[1] = 16;
        if (!c->statsA_final) {  // first finish the statistics (and bitmaps) of the split tiles in place, then copy
            TILES(c->tile_shift, launch_pack(st, c->summary.p, PackArgs(), c->split_tiles.p, c->counters.p, c->cov(), c->ucov(), c->d_bin_off.p,
                        c->R, c->d_tile_ref0.p, c->ref_stats.p, c->bits_layout()));
            c->statsA_final = true;
        }
        TILES(c->tile_shift, launch_pack(st, c->summary.p, pk));
        have_bits = c->summary_has_bits;
    } else {
        KernelTimer t(c, K_REF_STATS);
        PackArgs pk;
        pk.src[0] = c->tail();
        pk.n[0] = 16;
        launch_ref_stats(st, c->cov(), c->ucov(), c->d_bin_off.p, c->R, c->summary.p, &pk);
    }
    if (!have_bits && !c->binsA_stored)
        return fail(c, SLIMM_E_INVALID, "no bit maps and no coverage arrays: call slimm_prepare_summary or slimm_keep_bins");
    if (!have_bits) {
        launch_nonzero_bits(st, c->cov(), c->Bp, c->summary.p + 4ull * c->R + 16);
        launch_nonzero_bits(st, c->ucov(), c->Bp, c->summary.p + 4ull * c->R + 16 + bits_words);
    }
    if (!c->stream_ordered) HIP_TRY(c, hipStreamSynchronize(st));
    *d_ptr = c->summary.p;
    *n_words = W;
    return SLIMM_OK;
}

int slimm_merge_summary_slices(slimm_ctx* c, const void* d_recv, uint32_t n_ranks, uint32_t my_rank, void** d_vec,
                               uint64_t* n_words) {
    if (!c || !d_recv || !d_vec || !n_words) return SLIMM_E_INVALID;
    if (!c->analyzed) return fail(c, SLIMM_E_INVALID, "call slimm_analyze_alignments first");
    // (one rank: the unsliced summary of slimm_coverage_summary is its own single slice, however its bitmaps were made)
    if (n_ranks < 1 || n_ranks != std::max<uint32_t>(c->summary_slices, 1u) || my_rank >= n_ranks ||
        (!c->summary_has_bits && n_ranks > 1) || !c->summary.p)
        return fail(c, SLIMM_E_INVALID, "slimm_merge_summary_slices: slimm_prepare_summary(ctx, n_ranks) was not in effect");
    (void)hipSetDevice(c->device);
    const uint64_t W = 4ull * c->R + 16;
    HIP_TRY(c, c->d_sum_vec.ensure(W));
    const uint64_t lo_tile = static_cast<uint64_t>(my_rank) * c->slice_tiles();
    const uint64_t hi_tile = std::min<uint64_t>(lo_tile + c->slice_tiles(), c->ntiles);
    const uint32_t lo_bin = static_cast<uint32_t>(std::min<uint64_t>(lo_tile, c->ntiles) * c->tile_bins());
    const uint32_t hi_bin = static_cast<uint32_t>(std::max<uint64_t>(hi_tile, std::min<uint64_t>(lo_tile, c->ntiles)) * c->tile_bins());
    // own sums and scalars come from the summary buffer (slimm_coverage_summary filled it)
    launch_merge_slices(c->stream, static_cast<const uint32_t*>(d_recv), n_ranks, c->slice_words(), lo_bin, hi_bin,
                        c->d_bin_off.p, c->R, c->summary.p, c->summary.p + 4ull * c->R, c->d_sum_vec.p);
    if (!c->stream_ordered) HIP_TRY(c, hipStreamSynchronize(c->stream));
    *d_vec = c->d_sum_vec.p;
    *n_words = W;
    return SLIMM_OK;
}

int slimm_finish_coverage_reduced(slimm_ctx* c) {
    if (!c) return SLIMM_E_INVALID;
    if (!c->analyzed || !c->d_sum_vec.p) return fail(c, SLIMM_E_INVALID, "call slimm_merge_summary_slices first");
    (void)hipSetDevice(c->device);
    // packed block A = [4R merged statistics | 32 counters of this rank | 16 merged scalars]
    PackArgs pk;
    pk.src[0] = c->d_sum_vec.p;
    pk.n[0] = 4 * c->R;
    pk.src[1] = c->counters.p;
    pk.n[1] = 32;
    pk.src[2] = c->d_sum_vec.p + 4ull * c->R;
    pk.n[2] = 16;
    TILES(c->tile_shift, launch_pack(c->stream, c->ref_stats.p, pk));
    return finish_from_device_stats(c);
}

int slimm_finish_coverage_merged(slimm_ctx* c, const void* d_gathered, uint32_t n_ranks) {
    if (!c || !d_gathered || n_ranks == 0) return SLIMM_E_INVALID;
    if (!c->analyzed) return fail(c, SLIMM_E_INVALID, "call slimm_analyze_alignments first");
    (void)hipSetDevice(c->device);
    if (c->summary_slices > 1) return fail(c, SLIMM_E_INVALID, "sliced summaries are merged with slimm_merge_summary_slices");
    const uint64_t bits_words = c->Bp / 32;
    const uint64_t W = 4ull * c->R + 16 + 2 * bits_words;
    // merged statistics, this rank's counters and the merged scalars land in the packed block A
    launch_merge_summary(c->stream, static_cast<const uint32_t*>(d_gathered), W, n_ranks, c->d_bin_off.p, c->R, 4ull * c->R + 16,
                         4ull * c->R + 16 + bits_words, c->ref_stats.p, c->counters.p);
    return finish_from_device_stats(c);
}

int slimm_set_coverage_columns(slimm_ctx* c, const uint32_t* reads_count, const uint32_t* uniq_reads_count,
                               const uint32_t* nz_cov, const uint32_t* nz_uniq_cov, uint32_t hits, uint32_t matches) {
    if (!c || !reads_count || !uniq_reads_count || !nz_cov || !nz_uniq_cov) return SLIMM_E_INVALID;
    c->host->set_coverage(reads_count, uniq_reads_count, nz_cov, nz_uniq_cov, hits, matches);
    c->covered = true;
    c->no_hits = (hits == 0);
    return c->no_hits ? SLIMM_E_NO_HITS : SLIMM_OK;
}

// ------------------------------------------------------------------------------------------------ phase B + C(1)
namespace {
// slimm_filter_alignments in pieces, so that the multi-rank form (slimm_filter_alignments_launch + the exchange +
// slimm_install_merged_partials) can put its collective between the launches and the one host synchronisation.

// cut-offs + valid set on the host, lineage rows / valid bytes on their way to the device
int filter_prepare(slimm_ctx* c, bool& rows_ride_along) {
    HostProfile& h = *c->host;
    HostTrace tr("filter_alignments: prepare");
    h.compute_valid();
    tr.mark("compute_valid");
    if (c->device < 0) return SLIMM_OK;
    (void)hipSetDevice(c->device);
    hipStream_t st = c->stream;
    const uint32_t R = c->R;
    rows_ride_along = false;
    if (c->use_rows16) {
        // the rows are static except for the valid bit (bit 31 of .w): built once, then only the bits of the previous
        // file's valid references are cleared and this file's are set
        if (!c->rows16_base_ready) {
            const uint16_t* li = h.level_index().data();
            for (uint32_t r = 0; r < R; ++r) {
                const uint16_t* q = li + static_cast<size_t>(r) * 8;
                c->h_rows16.p[r] = make_uint4(q[0] | (uint32_t(q[1]) << 16), q[2] | (uint32_t(q[3]) << 16),
                                              q[4] | (uint32_t(q[5]) << 16), q[6] | (uint32_t(q[7]) << 16));
            }
            c->rows16_base_ready = true;
            c->rows16_prev.clear();
        }
        for (uint32_t r : c->rows16_prev) c->h_rows16.p[r].w &= 0x7fffffffu;
        c->rows16_prev = h.valid_list();
        for (uint32_t r : c->rows16_prev) c->h_rows16.p[r].w |= 0x80000000u;
        tr.mark("rows16 build");
        rows_ride_along = true;  // copied by the clearing kernel below (pinned host memory is device-readable)
        tr.mark("rows16 H2D call");
    } else {
        HIP_TRY(c, hipMemcpyAsync(c->d_valid.p, h.valid.data(), R, hipMemcpyHostToDevice, st));
    }
    // one bit per reference (rides along with the clearing kernel like the rows)
    for (uint32_t r : c->valid_bits_prev) c->h_valid_bits.p[r >> 5] &= ~(1u << (r & 31u));
    c->valid_bits_prev = h.valid_list();
    for (uint32_t r : c->valid_bits_prev) c->h_valid_bits.p[r >> 5] |= 1u << (r & 31u);
    tr.mark("valid bitmap");
    return SLIMM_OK;
}

// the launches of phase B / C(1): clears, k_filter, the selector histogram, the packed result block B
int filter_launch(slimm_ctx* c, bool copy_rows) {
    HostProfile& h = *c->host;
    (void)h;
    hipStream_t st = c->stream;
    const uint32_t R = c->R, T = c->T;
    uint32_t* const blockB = c->ref_stats.p + c->statsA_words();
    const bool rows_ride_along = copy_rows;
    const int attempt = 0;
    {
        KernelTimer t(c, K_MEMSET);
        if (!c->use_tiles) HIP_TRY(c, hipMemsetAsync(c->ucov2(), 0, c->Bp * sizeof(uint32_t), st));
        ZeroArgs z;
        z.p[0] = c->marks.p;
        z.n[0] = R * (kMarkBytes / 4);  // one byte per (reference, level)
        z.p[1] = c->counters.p + CNT_ERR;  // ERR, PAIRS, REDO
        z.n[1] = 3;
        if (c->use_tiles) {
            z.p[2] = c->tile_count.p;
            z.n[2] = c->treps * c->tstride;
            z.p[3] = blockB;  // per-reference statistics of uniq_cov2, accumulated by k_tile_hist
            z.n[3] = 4 * R;
            if (c->fused_scan) {
                z.p[4] = c->tile_cursor.p;
                z.n[4] = c->treps * c->tstride;
            }
        } else {
            z.p[2] = c->lca_count.p;
            z.n[2] = c->Tsel;
        }
        if (!c->pair_clean) {
            z.p64 = c->pair_tab.p;
            z.n64 = c->pair_cap;
        }
        if (rows_ride_along && attempt == 0) {
            z.cp_dst = reinterpret_cast<uint32_t*>(c->d_rows16.p);
            z.cp_src = reinterpret_cast<const uint32_t*>(c->h_rows16.p);
            z.cp_n = R * 4u;
        }
        z.cp2_dst = c->d_valid_bits.p;   // (every launch: a retry's too -- 2.5 KB at 20 000 references)
        z.cp2_src = c->h_valid_bits.p;
        z.cp2_n = R / 32u + 1u;
        launch_zero(st, z);
    }
    const uint32_t nslots = front_slots(c->rec.n);
    HIP_TRY(c, hipStreamWaitEvent(st, c->prefix_done, 0));  // (long done: recorded behind k_front)
    c->prefix_pending = false;
    {
        KernelTimer t(c, K_FILTER, true);  // (the dispatch's own time stamps, like k_front)
        FilterArgs fa;
        fa.tgt_ref = c->tgt_ref.p;
        fa.tgt_gbin = c->tgt_gbin.p;
        fa.slots = c->slots.p;
        fa.nslots = nslots;
        if (c->use_rows16) {
            fa.rows16 = c->d_rows16.p;
            fa.taxon_flat = c->d_level_taxon.p;
            fa.taxon_shift = c->taxon_shift;
        } else {
            fa.lin_dense = c->d_lin_dense.p;
            fa.valid = c->d_valid.p;
        }
        fa.valid_bits = c->d_valid_bits.p;
        fa.redo = c->filter_redo.p;
        fa.sel = c->sel.p;
        fa.slot_rbase = c->slot_rbase.p;
        fa.slot_bbase = c->slot_bbase.p;
        fa.marks = c->marks.p;
        fa.pair_tab = c->pair_tab.p;
        fa.pair_list = c->pair_list.p;
        fa.pair_mask = c->pair_cap - 1;
        fa.taxon_base = static_cast<uint32_t>(c->Bp);
        fa.counters = c->counters.p;
        launch_filter(st, fa, t.t0(), t.t1());
    }
    SlotValues selectors;  // dense: k_filter wrote every slot's selectors behind those of the slots before
    selectors.vals = c->sel.p;
    selectors.slots = nullptr;
    selectors.nslots = c->local_M;
    if (!c->use_tiles) {
        KernelTimer t(c, K_HIST);
        launch_sel_atomics(st, c->sel.p, c->local_M, static_cast<uint32_t>(c->Bp), c->ucov2(), c->lca_count.p);
    }
    if (c->use_tiles) {  // uniq_cov2 and the per-taxon LCA counts from the per-read selectors, through the LDS tile histogram
        const uint32_t grid = 512;
        // few selectors per tile (config 5: 100; config 3: 1 200): the count matrix (88 -> 61 us at config 3, 112 -> 33 us at
        // config 5); many (config 4: 12 000): the direct rounds, whose shared frontier per tile keeps the open lines in L2
        const bool matrix = c->matrix && (c->matrix_always || c->local_M / std::max(1u, c->ntiles2) < 4096u);
        const uint32_t tile_sub = c->wide_for(c->rec.n) ? kTileSubWide : kTileSubB;  // (one array: its counts are 32-bit already)
        {
            KernelTimer t(c, K_TILE_COUNT2);
            TILES(c->tile_shift, launch_tile_count(st, grid, c->ntiles2, selectors, nullptr, c->tile_count.p, c->treps, c->tstride,
                              matrix ? c->tile_matrix.p : nullptr));
            if (matrix) TILES(c->tile_shift, launch_matrix_prefix(st, grid, c->ntiles2, c->tile_matrix.p, c->tstride, c->tile_count.p));
        }
        if (c->fused_scan) {
            KernelTimer t(c, K_TILE_SCATTER2);
            TILES(c->tile_shift, launch_tile_scatter_fused(st, grid, c->ntiles2, selectors, c->counters.p, c->tile_count.p, c->tile_cursor.p,
                                      c->bucket.p, c->ucov2(), nullptr, c->tstride, c->tile_items.p, c->split_tiles.p, Totals(),
                                      tile_sub));
        } else {
            {
                KernelTimer t(c, K_TILE_SCAN2);
                TILES(c->tile_shift, launch_tile_scan(st, c->ntiles2, c->tile_count.p, c->tile_base.p, c->tile_cursor.p, c->tile_items.p,
                                 c->counters.p, c->part_items.p, c->sup_cursor.p, c->split_tiles.p, matrix ? 1u : c->treps,
                                 c->tstride, matrix ? false : c->two_level, Totals(), tile_sub));
            }
            {
                KernelTimer t(c, K_TILE_SCATTER2);
                if (matrix)
                    TILES(c->tile_shift, launch_tile_scatter_matrix(st, grid, c->ntiles2, selectors, c->tile_base.p, c->tile_matrix.p, c->tstride,
                                               c->bucket.p, c->ucov2(), nullptr, tile_sub));
                else
                    TILES(c->tile_shift, launch_tile_scatter(st, grid, c->ntiles2, c->rec.n, selectors, c->counters.p, c->tile_base.p,
                                        c->tile_cursor.p, c->sup_cursor.p, c->part_items.p, c->mid.p, c->bucket.p,
                                        c->ucov2(), nullptr, c->two_level, c->tile_count.p, c->treps, c->tstride, tile_sub));
            }
        }
        {
            KernelTimer t(c, K_TILE_HIST2);
            TILES(c->tile_shift, launch_tile_hist(st, c->ntiles2, c->rec.n, c->bucket.p, c->tile_base.p, c->tile_items.p, c->counters.p,
                             c->ucov2(), nullptr, c->d_bin_off.p, R, c->d_tile_ref0.p, blockB, BitsLayout(),
                             // (the tiles behind the bins hold the per-taxon LCA counts k_pack reads back)
                             c->keep_bins ? 0u : static_cast<uint32_t>(c->Bp / c->tile_bins())));
        }
    }
    {
        PackArgs pk;
        pk.src[0] = c->counters.p;
        pk.n[0] = 32;
        pk.src[1] = c->marks.p;
        pk.n[1] = R;
        pk.reps[1] = kPackBytes8;  // k_filter sets one byte per (reference, level)
        pk.src[2] = c->use_tiles ? c->lca_tiles() : c->lca_count.p;
        pk.n[2] = T;
        if (c->use_rows16) {  // counted per (level, index): summed into per-taxon counts on the way out
            pk.sum_k = 2;
            pk.sum_off = c->d_taxon_off.p;
            pk.sum_idx = c->d_taxon_idx.p;
        }
        if (c->use_tiles) {  // k_tile_hist left the uniq_cov2 statistics in place
            KernelTimer t(c, K_PACK2);
            TILES(c->tile_shift, launch_pack(st, blockB + 4ull * R, pk, c->split_tiles.p, c->counters.p, c->ucov2(), nullptr,
                        c->d_bin_off.p, R, c->d_tile_ref0.p, blockB));
        } else {
            KernelTimer t(c, K_REF_STATS2);
            launch_ref_stats(st, c->ucov2(), nullptr, c->d_bin_off.p, R, blockB, &pk);
        }
    }
    return SLIMM_OK;
}

// what the host learns from the counters of result block B once they are there: a pair-set overflow (the table is grown,
// `again` set) or the number of (taxon, reference) pairs, whose list is then fetched
int filter_check_counters(slimm_ctx* c, const uint32_t* h_cnt, bool& again) {
    again = false;
    if (h_cnt[CNT_ERR] & ERR_PAIR_OVERFLOW) {
        if (c->pair_cap >= (1u << 30)) return fail(c, SLIMM_E_INVALID, "(taxon, reference) pair set overflow");
        int rc = ensure_pair_table(c, c->pair_cap * 4);
        if (rc != SLIMM_OK) return rc;
        c->pair_clean = false;
        again = true;
        return SLIMM_OK;
    }
    c->n_pairs = h_cnt[CNT_PAIRS];
    c->pair_clean = (c->n_pairs == 0);
    if (c->n_pairs) {
        HIP_TRY(c, hipMemcpyAsync(c->h_pairs.p, c->pair_list.p, static_cast<size_t>(c->n_pairs) * 8, hipMemcpyDeviceToHost,
                                  c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    c->part_pairs.assign(c->h_pairs.p, c->h_pairs.p + c->n_pairs);
    std::sort(c->part_pairs.begin(), c->part_pairs.end());
    return SLIMM_OK;
}
}  // namespace

int slimm_filter_alignments(slimm_ctx* c) {
    if (!c) return SLIMM_E_INVALID;
    if (!c->covered) return fail(c, SLIMM_E_INVALID, "call slimm_finish_coverage first");
    if (c->no_hits) return SLIMM_E_NO_HITS;
    HostProfile& h = *c->host;
    HostTrace tr("filter_alignments");
    bool rows_ride_along = false;
    int rc = filter_prepare(c, rows_ride_along);
    if (rc != SLIMM_OK) return rc;
    if (c->device < 0) {  // host-only: the per-read part arrives through slimm_set_partials
        c->filtered = true;
        return SLIMM_OK;
    }
    hipStream_t st = c->stream;
    const uint32_t R = c->R;
    uint32_t* const blockB = c->ref_stats.p + c->statsA_words();
    c->filter_pending = false;
    for (int attempt = 0; attempt < 8; ++attempt) {
        rc = filter_launch(c, rows_ride_along && attempt == 0);
        if (rc != SLIMM_OK) return rc;
        uint32_t* const hB = c->h_stats.p + c->statsA_words();
        tr.mark("phase B launches");
        launch_copy_out(st, hB, blockB, static_cast<uint32_t>(c->statsB_words()));
        HIP_TRY(c, hipStreamSynchronize(st));
        tr.mark("stream sync (device phase B + copy)");
        bool again = false;
        rc = filter_check_counters(c, hB + 4ull * R, again);
        if (rc != SLIMM_OK) return rc;
        if (!again) break;
    }
    // the packed rows {uniq_reads_count2 = sum of uniq_cov2 bins, non-zero uniq_cov2 bins, -, -}, the level marks and the
    // per-taxon LCA counts go to the host profile straight from the pinned block
    const uint32_t* s2 = c->h_stats.p + c->statsA_words();
    h.set_partials_rows(s2, 4, s2 + 5ull * R + 32, s2 + 4ull * R + 32, c->part_pairs.data(), c->n_pairs);
    tr.mark("partials to host profile");
    c->binsB_stored = !c->use_tiles || c->keep_bins;
    c->filtered = true;
    return SLIMM_OK;
}

// Multi-rank form of phase B: everything slimm_filter_alignments launches, plus this rank's additive partial results
// packed for the exchange -- and no host synchronisation.  The caller sums the buffer of slimm_partials_buffer across
// ranks on the context's stream and calls slimm_install_merged_partials, which does the one copy + synchronise of the
// phase; SLIMM_E_RETRY from there (some rank's pair set overflowed: every rank has grown its table) = call this again.
int slimm_filter_alignments_launch(slimm_ctx* c) {
    if (!c) return SLIMM_E_INVALID;
    if (!c->covered) return fail(c, SLIMM_E_INVALID, "call slimm_finish_coverage first");
    if (c->no_hits) return SLIMM_E_NO_HITS;
    if (c->device < 0) return fail(c, SLIMM_E_INVALID, "host-only context: use slimm_filter_alignments + slimm_set_partials");
    bool rows_ride_along = false;
    int rc = SLIMM_OK;
    if (!c->filter_pending) {  // (a retry keeps the valid set and the rows that are on the device already)
        rc = filter_prepare(c, rows_ride_along);
        if (rc != SLIMM_OK) return rc;
    }
    rc = filter_launch(c, rows_ride_along);
    if (rc != SLIMM_OK) return rc;
    const uint64_t W = 3ull * c->R + c->T + 2;
    HIP_TRY(c, c->d_partials.ensure(W + 2));
    launch_partials_pack(c->stream, c->ref_stats.p + c->statsA_words(), c->R, c->T, c->d_partials.p);
    c->filter_pending = true;
    c->filtered = true;
    c->binsB_stored = !c->use_tiles || c->keep_bins;
    return SLIMM_OK;
}

int slimm_partials_buffer(slimm_ctx* c, void** d_ptr, uint64_t* n_words) {
    if (!c || !d_ptr || !n_words) return SLIMM_E_INVALID;
    if (!c->filtered || c->device < 0) return fail(c, SLIMM_E_INVALID, "no device partials (call slimm_filter_alignments)");
    (void)hipSetDevice(c->device);
    // [R uniq_reads_count2 | T LCA counts | 2R level marks | pairs | error flags], then this rank's own {pairs, flags}
    // (not part of the exchange)
    const uint64_t W = 3ull * c->R + c->T + 2;
    if (!c->filter_pending) {  // (slimm_filter_alignments_launch has packed them already)
        HIP_TRY(c, c->d_partials.ensure(W + 2));
        launch_partials_pack(c->stream, c->ref_stats.p + c->statsA_words(), c->R, c->T, c->d_partials.p);
    }
    if (!c->stream_ordered) HIP_TRY(c, hipStreamSynchronize(c->stream));
    *d_ptr = c->d_partials.p;
    *n_words = W;
    return SLIMM_OK;
}

int slimm_install_merged_partials(slimm_ctx* c, uint32_t* total_pairs) {
    if (!c) return SLIMM_E_INVALID;
    if (!c->filtered || c->device < 0 || !c->d_partials.p)
        return fail(c, SLIMM_E_INVALID, "call slimm_partials_buffer first");
    (void)hipSetDevice(c->device);
    const uint32_t R = c->R, T = c->T;
    const uint64_t W = 3ull * R + T + 2;
    HIP_TRY(c, c->h_partials.ensure(W + 2));
    launch_copy_out(c->stream, c->h_partials.p, c->d_partials.p, static_cast<uint32_t>(W + 2));  // (a kernel, not the DMA engine)
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const uint32_t* h = c->h_partials.p;
    if (c->filter_pending) {
        // the phase's only synchronisation has just happened: now the counters can be looked at.  Overflow flags were
        // summed with everything else, so every rank sees "some rank overflowed" and every rank goes round again.
        if (h[W - 1] != 0u) {
            if (c->pair_cap >= (1u << 30)) return fail(c, SLIMM_E_INVALID, "(taxon, reference) pair set overflow");
            int rc = ensure_pair_table(c, c->pair_cap * 4);
            if (rc != SLIMM_OK) return rc;
            c->pair_clean = false;
            return SLIMM_E_RETRY;
        }
        uint32_t local[CNT_WORDS] = {0};
        local[CNT_PAIRS] = h[W];
        local[CNT_ERR] = h[W + 1];
        bool again = false;
        int rc = filter_check_counters(c, local, again);
        if (rc != SLIMM_OK) return rc;
        c->filter_pending = false;
    }
    c->part_u2.assign(h, h + R);
    c->part_lca.assign(h + R, h + R + T);
    c->part_marks.resize(R);
    for (uint32_t r = 0; r < R; ++r) {
        const uint32_t lo = h[R + T + 2 * r], hi = h[R + T + 2 * r + 1];
        uint32_t m = 0;
        for (uint32_t lv = 0; lv < 4; ++lv) {
            if ((lo >> (8 * lv)) & 0xffu) m |= 1u << lv;
            if ((hi >> (8 * lv)) & 0xffu) m |= 1u << (4 + lv);
        }
        c->part_marks[r] = m;
    }
    if (total_pairs) *total_pairs = h[3ull * R + T];
    // the (taxon, reference) pairs of the other ranks are not in here: when the total is not this rank's own count the
    // caller gathers them and installs everything with slimm_set_partials
    c->host->set_partials(c->part_u2.data(), c->part_lca.data(), c->part_marks.data(), c->part_pairs.data(),
                          static_cast<uint32_t>(c->part_pairs.size()));
    c->counted = false;
    return SLIMM_OK;
}

int slimm_get_partials(slimm_ctx* c, slimm_partials* out) {
    if (!c || !out) return SLIMM_E_INVALID;
    if (!c->filtered || c->device < 0) return fail(c, SLIMM_E_INVALID, "no device partials (call slimm_filter_alignments)");
    out->n_refs = c->R;
    out->n_taxa_dense = c->T;
    const HostProfile& hp = *c->host;  // (what the last filter / install / set call handed to the host profile)
    out->uniq_reads_count2 = const_cast<uint32_t*>(hp.uniq_reads_count2.data());  // (read-only views; the struct is
                                                                                   // shared with slimm_set_partials)
    out->lca_count = const_cast<uint32_t*>(hp.lca_count().data());
    out->level_marks = const_cast<uint32_t*>(hp.level_marks().data());
    out->pairs = const_cast<uint64_t*>(hp.pairs().data());
    out->n_pairs = static_cast<uint32_t>(hp.pairs().size());
    out->scalars[0] = c->host->uniq_matches2;
    out->scalars[1] = out->scalars[2] = out->scalars[3] = 0;
    return SLIMM_OK;
}

int slimm_set_partials(slimm_ctx* c, const slimm_partials* in) {
    if (!c || !in) return SLIMM_E_INVALID;
    if (!c->filtered) return fail(c, SLIMM_E_INVALID, "call slimm_filter_alignments first");
    if (in->n_refs != c->R || in->n_taxa_dense != c->T) return fail(c, SLIMM_E_INVALID, "partials shape mismatch");
    if (!in->uniq_reads_count2 || !in->lca_count || !in->level_marks || (in->n_pairs && !in->pairs))
        return fail(c, SLIMM_E_INVALID, "null partial array");
    for (uint32_t k = 0; k < in->n_pairs; ++k)
        if ((in->pairs[k] >> 32) >= c->T || static_cast<uint32_t>(in->pairs[k]) >= c->R)
            return fail(c, SLIMM_E_INVALID, "pair %u out of range", k);
    c->part_u2.assign(in->uniq_reads_count2, in->uniq_reads_count2 + c->R);
    c->part_lca.assign(in->lca_count, in->lca_count + c->T);
    c->part_marks.assign(in->level_marks, in->level_marks + c->R);
    c->part_pairs.assign(in->pairs, in->pairs + in->n_pairs);
    std::sort(c->part_pairs.begin(), c->part_pairs.end());
    c->part_pairs.erase(std::unique(c->part_pairs.begin(), c->part_pairs.end()), c->part_pairs.end());
    c->host->set_partials(c->part_u2.data(), c->part_lca.data(), c->part_marks.data(), c->part_pairs.data(),
                          static_cast<uint32_t>(c->part_pairs.size()));
    c->counted = false;
    return SLIMM_OK;
}

int slimm_dense_taxa(slimm_ctx* c, uint32_t* n, const uint32_t** taxid) {
    if (!c || !n) return SLIMM_E_INVALID;
    *n = c->T;
    if (taxid) *taxid = c->host->dense_taxid().data();
    return SLIMM_OK;
}

// ------------------------------------------------------------------------------------------------ phase C(2,3) + profile
int slimm_get_reads_lca_count(slimm_ctx* c) {
    if (!c) return SLIMM_E_INVALID;
    if (c->no_hits) return SLIMM_E_NO_HITS;
    if (!c->filtered || !c->host->have_partials) return fail(c, SLIMM_E_INVALID, "call slimm_filter_alignments first");
    HostTrace tr("get_reads_lca_count");
    c->host->propagate();
    tr.mark("propagate");
    c->counted = true;
    return SLIMM_OK;
}

int slimm_get_profiles(slimm_ctx* c, const char* path) {  // src/slimm.hpp:447-489, one file on one GPU
    int rc = slimm_analyze_alignments(c);
    if (rc != SLIMM_OK) return rc;
    rc = slimm_finish_coverage(c);
    if (rc != SLIMM_OK) return rc;  // SLIMM_E_NO_HITS: "[WARNING] No mapped reads found" (:451-455), nothing written
    rc = slimm_filter_alignments(c);
    if (rc != SLIMM_OK) return rc;
    rc = slimm_get_reads_lca_count(c);
    if (rc != SLIMM_OK) return rc;
    return path ? slimm_write_abundance_file(c, path) : SLIMM_OK;
}

int slimm_write_abundance(slimm_ctx* c, const char** text, uint64_t* len) {
    if (!c || !text || !len) return SLIMM_E_INVALID;
    if (c->no_hits) return SLIMM_E_NO_HITS;
    if (!c->counted) return fail(c, SLIMM_E_INVALID, "call slimm_get_reads_lca_count first");
    const std::string& s = c->host->write_abundance();
    *text = s.data();
    *len = s.size();
    return SLIMM_OK;
}

int slimm_write_abundance_file(slimm_ctx* c, const char* path) {
    const char* text = nullptr;
    uint64_t len = 0;
    HostTrace tr("write_abundance_file");
    int rc = slimm_write_abundance(c, &text, &len);
    if (rc != SLIMM_OK) return rc;
    tr.mark("profile text");
    // (plain descriptors: three system calls, no stdio buffer to allocate and flush for a few KB written once)
    const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (fd < 0) return fail(c, SLIMM_E_INVALID, "cannot open %s for writing", path);
    uint64_t w = 0;
    while (w < len) {
        const ssize_t k = write(fd, text + w, len - w);
        if (k < 0 && errno == EINTR) continue;
        if (k <= 0) break;
        w += static_cast<uint64_t>(k);
    }
    const bool closed = close(fd) == 0;
    tr.mark("open + write + close");
    if (w != len || !closed) return fail(c, SLIMM_E_INVALID, "short write to %s", path);
    return SLIMM_OK;
}

// ------------------------------------------------------------------------------------------------ results
int slimm_get_stats(slimm_ctx* c, slimm_stats* o) {
    if (!c || !o) return SLIMM_E_INVALID;
    HostProfile& h = *c->host;
    memset(o, 0, sizeof(*o));
    o->hits_count = h.hits;
    o->matches_count = h.matches;
    o->uniq_matches_count = h.uniq_matches;
    o->uniq_hits_count = h.uniq_hits;
    o->uniq_matches_count2 = h.uniq_matches2;
    o->reference_count = h.reference_count;
    o->matched_ref_length = h.matched_ref_length;
    o->failed_by_cov = h.failed_by_cov;
    o->failed_by_uniq_cov = h.failed_by_ucov;
    o->failed_by_min_read = h.failed_by_min_read;
    o->n_valid = h.n_valid;
    o->bin_width = h.bin_width();
    o->min_reads = h.min_reads;
    o->avg_read_len = h.config().avg_read_len;
    o->profile_count = h.profile_count;
    o->profile_failed = h.profile_failed;
    if (h.have_coverage && !c->no_hits) {
        o->coverage_cut_off = h.coverage_cut_off();
        o->uniq_coverage_cut_off = h.uniq_coverage_cut_off();
        o->expected_coverage = h.matched_ref_length ? h.expected_coverage() : 0.0f;
    }
    o->n_records = c->n_pushed;
    o->n_targets = c->local_P;
    o->total_bins = h.total_bins();
    return SLIMM_OK;
}

int slimm_get_ref_columns(slimm_ctx* c, slimm_ref_columns* o) {
    if (!c || !o) return SLIMM_E_INVALID;
    HostProfile& h = *c->host;
    const uint32_t R = c->R;
    if (!h.have_coverage) return fail(c, SLIMM_E_INVALID, "no coverage results yet");
    auto cp = [&](uint32_t* dst, const std::vector<uint32_t>& src) {
        if (dst && src.size() == R) memcpy(dst, src.data(), R * 4);
        else if (dst) memset(dst, 0, R * 4);
    };
    cp(o->reads_count, h.reads_count);
    cp(o->uniq_reads_count, h.uniq_reads_count);
    cp(o->uniq_reads_count2, h.have_partials ? h.uniq_reads_count2 : std::vector<uint32_t>());
    cp(o->nbins, h.nbins());
    cp(o->nz_cov, h.nz_cov);
    cp(o->nz_uniq_cov, h.nz_ucov);
    cp(o->nz_uniq_cov2, h.nz_uniq_cov2());
    if (o->valid) {
        if (h.have_valid) memcpy(o->valid, h.valid.data(), R);
        else memset(o->valid, 0, R);
    }
    if (o->abundance || o->uniq_abundance) h.abundances();
    if (o->abundance) memcpy(o->abundance, h.abundance.data(), R * 4);
    if (o->uniq_abundance) memcpy(o->uniq_abundance, h.uniq_abundance.data(), R * 4);
    return SLIMM_OK;
}

int slimm_get_bins(slimm_ctx* c, int which, uint32_t* out) {
    if (!c || !out || which < 0 || which > 2) return SLIMM_E_INVALID;
    if (c->device < 0 || !c->analyzed) return fail(c, SLIMM_E_INVALID, "no coverage bins on this context");
    if (which == 2 && !c->filtered) return fail(c, SLIMM_E_INVALID, "uniq_cov2 needs slimm_filter_alignments");
    if (which < 2 ? !c->binsA_stored : !c->binsB_stored)
        return fail(c, SLIMM_E_INVALID, "the coverage arrays were not kept (slimm_keep_bins)");
    (void)hipSetDevice(c->device);
    const uint32_t* src = which == 0 ? c->cov() : which == 1 ? c->ucov() : c->ucov2();
    std::vector<uint32_t> padded(c->Bp);
    HIP_TRY(c, hipMemcpyAsync(padded.data(), src, c->Bp * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    uint64_t k = 0;
    for (uint32_t r = 0; r < c->R; ++r) {
        uint32_t nb = c->host->nbins()[r];
        memcpy(out + k, padded.data() + c->bin_off_h[r], static_cast<size_t>(nb) * 4);
        k += nb;
    }
    return SLIMM_OK;
}

int slimm_get_read_targets(slimm_ctx* c, uint32_t* ref, uint32_t* gbin, uint64_t cap, uint64_t* n) {
    if (!c || !n) return SLIMM_E_INVALID;
    if (c->device < 0 || !c->analyzed) return fail(c, SLIMM_E_INVALID, "no targets on this context (call slimm_analyze_alignments)");
    (void)hipSetDevice(c->device);
    const uint32_t ns = front_slots(c->rec.n);
    std::vector<uint4> sl(ns);
    if (ns) HIP_TRY(c, hipMemcpyAsync(sl.data(), c->slots.p, ns * sizeof(uint4), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    uint64_t total = 0;
    for (const uint4& d : sl) total += d.y;
    *n = total;
    if (!ref && !gbin) return SLIMM_OK;
    if (cap < total) return fail(c, SLIMM_E_INVALID, "slimm_get_read_targets: %llu entries, room for %llu",
                                 static_cast<unsigned long long>(total), static_cast<unsigned long long>(cap));
    uint64_t o = 0;
    for (const uint4& d : sl) {  // (a slot's targets are contiguous; runs of slots are too when nothing was dropped)
        if (!d.y) continue;
        if (ref) HIP_TRY(c, hipMemcpyAsync(ref + o, c->tgt_ref.p + d.x, d.y * 4ull, hipMemcpyDeviceToHost, c->stream));
        if (gbin) HIP_TRY(c, hipMemcpyAsync(gbin + o, c->tgt_gbin.p + d.x, d.y * 4ull, hipMemcpyDeviceToHost, c->stream));
        o += d.y;
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SLIMM_OK;
}

int slimm_taxon_count_size(slimm_ctx* c, int stage, uint32_t* n) {
    if (!c || !n) return SLIMM_E_INVALID;
    if (!c->counted) return fail(c, SLIMM_E_INVALID, "call slimm_get_reads_lca_count first");
    std::vector<uint32_t> t, k;
    c->host->taxon_counts(stage, t, k);
    *n = static_cast<uint32_t>(t.size());
    return SLIMM_OK;
}

int slimm_get_taxon_counts(slimm_ctx* c, int stage, uint32_t* taxid, uint32_t* count) {
    if (!c || !taxid || !count) return SLIMM_E_INVALID;
    if (!c->counted) return fail(c, SLIMM_E_INVALID, "call slimm_get_reads_lca_count first");
    std::vector<uint32_t> t, k;
    c->host->taxon_counts(stage, t, k);
    memcpy(taxid, t.data(), t.size() * 4);
    memcpy(count, k.data(), k.size() * 4);
    return SLIMM_OK;
}

int slimm_children_pairs_size(slimm_ctx* c, int stage, uint64_t* n) {
    if (!c || !n) return SLIMM_E_INVALID;
    if (!c->counted) return fail(c, SLIMM_E_INVALID, "call slimm_get_reads_lca_count first");
    std::vector<uint32_t> t, r;
    c->host->children_pairs(stage, t, r);
    *n = t.size();
    return SLIMM_OK;
}

int slimm_get_children_pairs(slimm_ctx* c, int stage, uint32_t* taxid, uint32_t* ref) {
    if (!c || !taxid || !ref) return SLIMM_E_INVALID;
    if (!c->counted) return fail(c, SLIMM_E_INVALID, "call slimm_get_reads_lca_count first");
    std::vector<uint32_t> t, r;
    c->host->children_pairs(stage, t, r);
    memcpy(taxid, t.data(), t.size() * 4);
    memcpy(ref, r.data(), r.size() * 4);
    return SLIMM_OK;
}

// ------------------------------------------------------------------------------------------------ measurement
int slimm_enable_kernel_timing(slimm_ctx* c, int on) {
    if (!c) return SLIMM_E_INVALID;
    c->timing = on != 0 && c->device >= 0;
    return SLIMM_OK;
}

int slimm_time_only_kernel(slimm_ctx* c, const char* name) {
    if (!c) return SLIMM_E_INVALID;
    if (!name || !*name) {
        c->timing_only = -1;
        return SLIMM_OK;
    }
    for (int i = 0; i < K_COUNT; ++i)
        if (std::strcmp(name, kKernelNames[i]) == 0) {
            c->timing_only = i;
            return SLIMM_OK;
        }
    return fail(c, SLIMM_E_INVALID, "slimm_time_only_kernel: unknown kernel '%s'", name);
}

int slimm_kernel_times(slimm_ctx* c, const char** names, double* ms, uint32_t* launches, uint32_t cap, uint32_t* n,
                       int reset) {
    if (!c || !n) return SLIMM_E_INVALID;
    if (c->device >= 0) {
        (void)hipSetDevice(c->device);
        drain_events(c);
    }
    uint32_t k = 0;
    for (int i = 0; i < K_COUNT && k < cap; ++i, ++k) {
        if (names) names[k] = kKernelNames[i];
        if (ms) ms[k] = c->k_ms[i];
        if (launches) launches[k] = c->k_n[i];
    }
    *n = k;
    if (reset) {
        for (int i = 0; i < K_COUNT; ++i) {
            c->k_ms[i] = 0;
            c->k_n[i] = 0;
        }
    }
    return SLIMM_OK;
}

void slimm_group_plan(uint64_t n_records, uint32_t* passes, uint32_t* width, uint32_t* bits, uint32_t* grid) {
    const GroupPlan g = group_plan(static_cast<uint32_t>(std::min<uint64_t>(n_records, 0x7ffffffeull)));
    if (passes) *passes = g.passes;
    if (width) *width = g.width;
    if (bits) *bits = g.bits;
    if (grid) *grid = g.grid;
}

int slimm_grouped_records(slimm_ctx* c, uint64_t* ident, uint32_t* ref, uint32_t* gbin, uint64_t cap, uint64_t* n) {
    if (!c || !n) return SLIMM_E_INVALID;
    if (c->device < 0 || c->order != SLIMM_ORDER_ANY) return fail(c, SLIMM_E_INVALID, "slimm_grouped_records: a device context created for SLIMM_ORDER_ANY");
    if (!c->analyzed) return fail(c, SLIMM_E_INVALID, "call slimm_analyze_alignments first");
    (void)hipSetDevice(c->device);
    uint32_t V = 0;
    HIP_TRY(c, hipMemcpyAsync(&V, c->counters.p + CNT_V, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *n = V;
    const uint64_t m = std::min<uint64_t>(cap, V);
    if (m == 0) return SLIMM_OK;
    if (ident) HIP_TRY(c, hipMemcpy(ident, c->c_ident.p, m * 8, hipMemcpyDeviceToHost));
    if (ref || gbin) {
        std::vector<uint2> pay(m);
        HIP_TRY(c, hipMemcpy(pay.data(), c->c_pay.p, m * 8, hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < m; ++i) {
            if (ref) ref[i] = pay[i].x;
            if (gbin) gbin[i] = pay[i].y;
        }
    }
    return SLIMM_OK;
}

// ------------------------------------------------------------------------------------------------ host-only helpers
uint32_t slimm_host_avg_read_length(const uint32_t* l_seq, uint64_t n, uint32_t sample_size) {  // misc.hpp:509-522
    uint32_t count = 0, total = 0;
    for (uint64_t i = 0; i < n && count < sample_size; ++i) {
        if (l_seq[i] == 0) continue;
        total += l_seq[i];
        ++count;
    }
    return count ? total / count : 0;
}

float slimm_host_quantile_cut_off(const float* v, uint32_t n, float q) {
    return slimm::quantile_cut_off(std::vector<float>(v, v + n), q);
}

uint32_t slimm_host_canonical_read_name(const char* name, uint32_t name_len, uint16_t flag, uint16_t* flag_out) {  // src/slimm.hpp:204-208 (Q18)
    size_t n = name_len;
    const uint16_t f = slimm::canonical_read(name, n, flag);
    if (flag_out) *flag_out = f;
    return static_cast<uint32_t>(n);
}
void slimm_host_q18_note(slimm_q18_runs* q, int starts_run, int shortened) {  // include/slimm_hip.h: "Q18 ON A GROUPED STREAM"
    if (!q) return;
    if (shortened && starts_run) ++q->short_starts;
    if (!shortened && !starts_run && q->last_short) ++q->short_to_plain;
    q->last_short = shortened != 0;
}
int slimm_host_q18_regroup_needed(const slimm_q18_runs* q) { return q && q->short_starts != q->short_to_plain; }
int slimm_get_q18_runs(slimm_ctx* c, uint64_t* short_starts, uint64_t* short_to_plain) {
    if (!c || !short_starts || !short_to_plain) return SLIMM_E_INVALID;
    *short_starts = *short_to_plain = 0;
    if (c->device < 0 || !c->bam.active || c->order != SLIMM_ORDER_GROUPED) return SLIMM_OK;
    const int rc = bam_fetch_q18(c);
    if (rc != SLIMM_OK) return rc;
    *short_starts = c->bam.q18_starts;
    *short_to_plain = c->bam.q18_plain;
    return SLIMM_OK;
}
uint32_t slimm_host_bin_of(int32_t begin_pos, uint32_t avg_read_len, uint32_t ref_len, uint32_t bin_width) {
    uint32_t center = std::min(static_cast<uint32_t>(begin_pos) + (avg_read_len / 2), ref_len);  // slimm.hpp:200
    return bin_width ? center / bin_width : 0;                                                   // slimm.hpp:201
}

}  // extern "C"
