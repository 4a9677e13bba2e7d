// Launch interface between the context (context.hip) and the kernels (kernels.hip, radix_sort.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

// "This value is in a vector register HERE": an empty asm statement that takes and returns it.  It costs nothing and
// gives a loaded value a use the compiler can neither move nor remove -- without one, a read-only load whose only uses
// sit in a conditional block is sunk into that block, and the loads a kernel issued together come back one at a time.
#if defined(__HIP_DEVICE_COMPILE__)
#define SLIMM_PIN_VGPR(x) asm volatile("" : "+v"(x))
#else
#define SLIMM_PIN_VGPR(x) (void)(x)
#endif

namespace slimm {

// slots of the device counter block (uint32[32])
enum {
    CNT_V = 0,      // mapped records of this context = hits_count (src/slimm.hpp:212)
    CNT_M = 1,      // reads = matches_count (src/slimm.hpp:257)
    CNT_P = 2,      // distinct (read, ref) pairs = targets
    CNT_ERR = 3,    // ERR_* bits
    CNT_PAIRS = 4,  // entries of the no-level-agrees (taxon, ref) set
    CNT_REDO = 5,   // slots k_filter_compact left to k_filter_walk (a read with 64 valid targets or more)
    CNT_ITEMS = 10, // work items of k_tile_hist
    CNT_ITEMS2 = 6, // work items of k_part_tile
    CNT_ANYGB = 8,  // some record follows a record of the same qName run with a larger mate number (mates interleave)
    CNT_MODE = 7,   // classification picked on the device: 0 = look-back walk, 2 = tagged-word walk (both k_runs), 1 = hash table (k_runs_hash)
    CNT_SPLIT = 9,  // bin tiles cut into several k_tile_hist work items (their non-zero counts are finished by k_pack)
    CNT_WORDS = 32
};
enum { ERR_REF_RANGE = 1, ERR_RUN_LENGTH = 2, ERR_PAIR_OVERFLOW = 4, ERR_KEY_COLLISION = 8 };

#if defined(__HIPCC__)
// Sum over the 64 lanes of a wave with DPP adds (no LDS round trips as with ds_bpermute shuffles); every lane gets it.
__device__ __forceinline__ uint32_t wave_sum_dpp(uint32_t v) {
    v += __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0u, v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0u, v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0u, v, 0x118, 0xf, 0xf, true);   // row_shr:8  -> lane 15 of every row: row total
    v += __builtin_amdgcn_update_dpp(0u, v, 0x142, 0xa, 0xf, true);   // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0u, v, 0x143, 0xc, 0xf, true);   // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}
#endif

struct PackArgs {  // small arrays appended behind the per-reference statistics (k_pack, k_ref_stats)
    const uint32_t* src[4] = {nullptr, nullptr, nullptr, nullptr};
    uint32_t n[4] = {0, 0, 0, 0};
    uint32_t reps[4] = {1, 1, 1, 1};  // src[k] holds reps[k] copies of n[k] words each; their bitwise OR is packed
                                      // kPackBytes8: src[k] holds 8 bytes per output word, bit l = (byte l != 0)
    // one array may be a SUM of source words instead: output word i of array sum_k = sum of src[sum_k][sum_idx[j]] for
    // j in [sum_off[i], sum_off[i + 1]) -- the per-taxon LCA counts, which k_filter counts per (level, index) of the
    // lineage rows (no taxon look-up on its path) and which become per-taxon counts here
    int sum_k = -1;
    const uint32_t* sum_off = nullptr;
    const uint32_t* sum_idx = nullptr;
};
constexpr uint32_t kPackBytes8 = 0xffffffffu;
#ifdef __HIPCC__
// word i of packed array k (k_pack, k_ref_stats)
__device__ __forceinline__ uint32_t packed_word(const PackArgs& pack, int k, uint32_t i) {
    uint32_t v = 0;
    if (k == pack.sum_k) {
        for (uint32_t j = pack.sum_off[i]; j < pack.sum_off[i + 1]; ++j) v += pack.src[k][pack.sum_idx[j]];
    } else if (pack.reps[k] == kPackBytes8) {
        const uint2 b = reinterpret_cast<const uint2*>(pack.src[k])[i];
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            v |= ((b.x >> (8 * l)) & 0xffu) ? (1u << l) : 0u;
            v |= ((b.y >> (8 * l)) & 0xffu) ? (16u << l) : 0u;
        }
    } else {
        for (uint32_t rep = 0; rep < pack.reps[k]; ++rep) v |= pack.src[k][static_cast<size_t>(rep) * pack.n[k] + i];
    }
    return v;
}
#endif
constexpr uint32_t kMarkBytes = 8;  // level marks of k_filter_lca16: one byte per (reference, level)
struct ZeroArgs {  // arrays cleared by one k_zero launch; p64 is filled with ~0 (the empty key of the pair set)
    uint32_t* p[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    uint32_t n[5] = {0, 0, 0, 0, 0};
    uint64_t* p64 = nullptr;
    uint32_t n64 = 0;
    uint32_t* cp_dst = nullptr;        // optional ride-along copy of cp_n words (16-byte aligned), e.g. pinned host -> device
    const uint32_t* cp_src = nullptr;
    uint32_t cp_n = 0;
    uint32_t* cp2_dst = nullptr;       // ... and a second, small one (word by word)
    const uint32_t* cp2_src = nullptr;
    uint32_t cp2_n = 0;
};
void launch_zero(hipStream_t st, const ZeroArgs& z);
// n words from device memory to host-mapped pinned memory (16-byte aligned both) by a kernel instead of the DMA engine
void launch_copy_out(hipStream_t st, uint32_t* dst_host, const uint32_t* src, uint32_t n);

struct DeviceRecords {
    const uint64_t* key = nullptr;
    const int32_t* ref = nullptr;
    const int32_t* pos = nullptr;
    const uint16_t* flag = nullptr;
    const uint32_t* check = nullptr;  // optional second hash of the read name: equal keys must carry equal checks
    uint32_t n = 0;
    bool packed = false;  // 16 bytes per record: flag == nullptr, the key's top three bits are {unmapped, mate number}
    bool marked = false;  // 8 bytes per record (grouped input only): key == flag == nullptr, `ref` holds the words
                          // reference + 1 | mate << 29 | starts a qName run << 31 (slimm_push_records_marked)
};


// ---- front.hip: the single-pass front end of phase A ----
// A slot = kSlotRecs consecutive records; its targets (the runs that START in it) lie compacted at
// [slot.x, slot.x + slot.y) of tgt_ref / tgt_gbin; slot.z = reads (targets with bit 31 of tgt_ref), slot.w = mapped records
// 768 records: measured, not derived.  k_front by slot size at 1 B records / config 2 / config 5 (one box, four waves per
// SIMD throughout): 1088: 5 075 / 73 / 753 us, 1024: 5 141 / 76 / 757, 896: 5 283 / 74 / 769, 832: 5 061-5 144 / 77 / 726,
// 768: 4 934-4 989 / 69 / 683, 704: 4 973 / 67 / 684, 640: 5 034 / 67 / 686, 512: 5 141 / 67 / 704 -- while every consumer
// of the slots likes them large (1 B records, 1024 -> 768 -> 640 -> 512: k_tile_count 453 -> 460 -> 543 -> 619 us, k_filter
// 2 571 -> 2 588-2 660 -> 2 616 -> 2 843, scatter 2 314 -> 2 393 -> 2 458 -> 2 503).
constexpr uint32_t kSlotRecs = 768;
constexpr int kFrontBlock = 64;   // one slot, one wave, one workgroup: the dispatcher backfills wave by wave (config 3: 594 -> 549 us)
constexpr uint32_t kMaxRefs = (1u << 26) - 2u;      // reference id + 1 fits 26 bits and is not all ones (front.hip)
constexpr uint32_t kMaxBins = 0x7ffffff0u;          // global bin indices fit 31 bits (bit 31 of tgt_gbin: unique read)
uint32_t front_slots(uint32_t n_records);
void launch_front_raw(hipStream_t st, const DeviceRecords& in, uint32_t n_refs, const uint2* geo, uint32_t half_read,
                      uint32_t bin_width, uint32_t* counters, uint32_t* tgt_ref, uint32_t* tgt_gbin, uint4* slots,
                      hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr);  // t0 / t1: the dispatch's time stamps
void launch_front_sorted(hipStream_t st, uint32_t n_upper, const uint64_t* ident, const uint2* pay,
                         uint32_t* counters, uint32_t* tgt_ref, uint32_t* tgt_gbin, uint4* slots,
                         const uint32_t* cchk = nullptr);  // pay: {reference, global bin}; cchk: check words (equal keys must carry equal ones)

// diagnostic: run starts whose identity (key & id_mask) started an earlier run too; tab: (tab_mask + 1) words of ~0
void launch_check_grouping(hipStream_t st, const uint64_t* key, uint32_t n, uint64_t id_mask, uint64_t* tab, uint32_t tab_mask,
                           uint32_t* n_split);
// the direct-atomics fallback of the coverage histograms (too many bins for the LDS tile tables)
// (also adds the stream's totals to counters[CNT_M / CNT_P] -- and [CNT_V] when count_mapped: the compaction of the
// sort path has counted the mapped records already -- and to tail[0..2]; all zero on entry)
void launch_hist(hipStream_t st, const uint32_t* tgt_gbin, const uint4* slots, uint32_t nslots, uint32_t* counters,
                 uint32_t* tail, uint32_t* cov, uint32_t* ucov, bool count_mapped);
void launch_ref_stats(hipStream_t st, const uint32_t* a, const uint32_t* b, const uint32_t* bin_off, uint32_t n_refs,
                      uint32_t* out, const PackArgs* pack = nullptr);
// phase B + C(1): one selector per read into sel[] (indexed like the slots' reads: slot.x + k): its uniq_cov2 bin,
// taxon_base + its LCA taxon, or 0xffffffff; level marks as one byte per (reference, level); (taxon, reference) pairs
// of the reads whose references agree at no level.  rows16 != nullptr: 16-byte lineage rows (level_off: host array of 8
// offsets into level_taxon); otherwise 32-byte rows lin_dense + the byte array valid.
struct FilterArgs {
    const uint32_t* tgt_ref = nullptr;
    const uint32_t* tgt_gbin = nullptr;
    const uint4* slots = nullptr;
    uint32_t nslots = 0;
    const void* rows16 = nullptr;
    const uint32_t* taxon_flat = nullptr;  // rows16: dense taxon of (level, index) at [(index << 3) | level]
    uint32_t taxon_shift = 0;
    const uint32_t* lin_dense = nullptr;
    const uint8_t* valid = nullptr;
    const uint32_t* valid_bits = nullptr;    // one bit per reference: what k_filter_compact asks before anything else
    uint32_t* redo = nullptr;                // ... and the slots it leaves to k_filter_walk (nslots words; their number: counters[CNT_REDO])
    uint32_t* sel = nullptr;                 // one selector per read, dense: slot s writes at slot_rbase[s] + slot_bbase[s >> 10]
    const uint32_t* slot_rbase = nullptr;    // (launch_slot_read_prefix)
    const uint32_t* slot_bbase = nullptr;
    uint32_t* marks = nullptr;
    uint64_t* pair_tab = nullptr;
    uint64_t* pair_list = nullptr;
    uint32_t pair_mask = 0, taxon_base = 0;
    uint32_t* counters = nullptr;
};
void launch_filter(hipStream_t st, const FilterArgs& a, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr);  // t0 / t1: as for launch_front_raw
// direct-atomics fallback: count the selectors with global atomics instead of the second tile histogram
void launch_sel_atomics(hipStream_t st, const uint32_t* sel, uint32_t n_reads, uint32_t taxon_base, uint32_t* ucov2,
                        uint32_t* lca_count);
// the exclusive prefix of the slots' read counts: rbase[s] + bbase[s >> 10] = reads in front of slot s (rbase: nslots
// words, bbase: nslots / 1024 + 1); two small launches
void launch_slot_read_prefix(hipStream_t st, const uint4* slots, uint32_t nslots, uint32_t* rbase, uint32_t* bbase);
// multi-GPU: [R uniq_reads_count2 | T LCA counts | 2R level marks in 8-bit fields | 1 pair count] from result block B
void launch_partials_pack(hipStream_t st, const uint32_t* block_b, uint32_t R, uint32_t T, uint32_t* out);
// multi-GPU, all-to-all form: this rank received every rank's bitmaps of ITS slice ([n_ranks][2][slice_words] 32-bit
// words, bins lo_bin .. hi_bin); vec = [4R: own sums in slots 0 / 2, popcount of the OR-ed bitmaps inside the slice in
// slots 1 / 3 | 16 own scalars]: additive over the ranks, so one all-reduce(SUM) completes it
void launch_merge_slices(hipStream_t st, const uint32_t* recv, uint32_t n_ranks, uint64_t slice_words, uint32_t lo_bin,
                         uint32_t hi_bin, const uint32_t* bin_off, uint32_t n_refs, const uint32_t* own_stats,
                         const uint32_t* own_tail, uint32_t* vec);
// multi-GPU coverage summary: "bin != 0" bitmaps and their merge (sums over ranks, popcount of the OR per reference)
void launch_nonzero_bits(hipStream_t st, const uint32_t* bins, uint64_t n_bins, uint32_t* bits);
void launch_merge_summary(hipStream_t st, const uint32_t* gathered, uint64_t rank_stride, uint32_t n_ranks,
                          const uint32_t* bin_off, uint32_t n_refs, uint64_t bits_off_cov, uint64_t bits_off_ucov,
                          uint32_t* out_stats, const uint32_t* counters);

// ---- LDS-privatised coverage histograms (tile_hist.hip) ----
// The bins are cut into TILES of 2^shift consecutive bins; tile_hist.hip is compiled once per tile size into its own
// namespace (tiles13: 8192 bins, tiles14: 16384 bins -- what the 16-bit bucket entries hold beside the unique bit), and a
// context picks one per layout (context.hip: tile_shift_for).  Smaller tiles: more workgroups of k_tile_hist per CU,
// faster histograms; larger tiles: half as many bucket frontiers, longer runs of one tile among neighbouring targets
// and so fewer returning atomics in the direct rounds of the scatter.
constexpr uint32_t kTileShiftSmall = 13, kTileShiftLarge = 14;
constexpr uint32_t kTileSub = 49152;                    // bucket entries per k_tile_hist work item (packed 16-bit counts: below
                                                        // 65536).  Hot tiles (a few present genomes take most of a sample's
                                                        // reads) are cut into items of this size, each adding its counts to
                                                        // global memory with atomics, and k_pack goes over them once more:
                                                        // 16384 / 32768 / 49152 at config 3: k_tile_hist 224 / 162 / 153 us,
                                                        // k_pack 46 / 27 / 17; config 5: 319 / 312 / 310 and 27 / 15 / 13
constexpr uint32_t kTileSubB = 16384;                   // ... of phase B's single-array histogram (32-bit counts): the reads
                                                        // that keep several targets meet in the few tiles of the taxon
                                                        // entries, and smaller items are more workgroups on those
constexpr uint32_t kTileSubWide = 262144;               // ... of the wide form (32-bit counts): layouts with far more than
                                                        // kTileSub entries per tile (tile_sub / wide arguments below)
constexpr size_t kTileLdsMax = 144 * 1024;              // LDS histogram of tile ids in k_tile_count / k_tile_scatter
// tile_count: `reps` copies of rep_stride words, zero on entry (k_zero); workgroup b adds to copy b % reps.
// k_tile_scan sums the copies into tile_base and turns every copy into the start of its stretch inside the buckets,
// which k_tile_scatter (same grid) then fills through the copy's own cursors: 1/reps of the same-address atomics.
#ifndef SLIMM_TILE_REPS
#define SLIMM_TILE_REPS 8
#endif
constexpr uint32_t kTileReps = SLIMM_TILE_REPS;
// the values the bucketing kernels read: one per target (tgt_gbin: bit 31 = unique read), lying in the slots front.hip
// wrote -- or, slots == nullptr, a dense array of nslots values (the selectors of phase B, one per read: 0xffffffff = none)
struct SlotValues {
    const uint32_t* vals = nullptr;
    const uint4* slots = nullptr;
    uint32_t nslots = 0;       // slots, or values of the dense form
    bool per_read = false;     // (slot form) a slot's values are its reads', not its targets'
};
// The front end leaves the totals of the stream {mapped records, reads, targets} to its consumers (thousands of waves
// adding to three counters would be as many memory-side atomics in a row): launch_tile_count with part != nullptr
// writes one partial sum per workgroup (part[grid]), and the scan that follows (launch_tile_scan or
// launch_tile_scatter_fused, given the same part) adds them up into counters[CNT_V / CNT_M / CNT_P] and tail[0..2].
struct Totals {
    const uint4* part = nullptr;
    uint32_t nparts = 0;
    uint32_t* tail = nullptr;  // may be null
};
// `grid` = the bucketing grid; the count runs tile_count_grid(grid) workgroups, kCountFold times as large, each for
// kCountFold neighbours of the bucketing grid (and writes as many entries of part[])
constexpr uint32_t kCountFold = 2;
constexpr uint32_t kSuperTiles = 64;                    // tiles per super tile (level 1 of the bucketing)
constexpr uint32_t kMaxSuper = 8192;                    // super tiles an LDS cursor array holds
constexpr uint32_t kPartSub = 32768;                    // entries per k_part_tile work item
// where k_tile_hist / k_pack put the 'bin != 0' bitmaps of the multi-GPU coverage summary: the tiles are cut into slices
// of `tps` tiles (one slice per rank for the all-to-all exchange, a single slice otherwise) and slice j holds
// [array 0 bits | array 1 bits] of its tiles back to back
struct BitsLayout {
    uint64_t* base = nullptr;   // nullptr: no bitmaps
    uint32_t tps = 1;           // tiles per slice
    uint64_t slice_w64 = 0;     // 64-bit words of one array in one slice = tps * (bins per tile / 64)
};
constexpr uint32_t kBigRoundTiles = 6144;   // tiles up to which the one-level scatter orders its rounds by tile in LDS
                                            // (k_tile_scatter_big; the direct rounds beyond)
constexpr uint32_t kFusedScanTiles = 4064;  // (4096 table entries less the room three small arrays take: tile_hist.hip)
namespace tiles13 {
#include "tile_api.inc"
}
namespace tiles14 {
#include "tile_api.inc"
}
// TILES(shift, launch_tile_count(st, ...)): the call in the namespace of the tile size
#define TILES(shift, call) ((shift) == ::slimm::kTileShiftLarge ? ::slimm::tiles14::call : ::slimm::tiles13::call)

// ---- bam_decode.hip: BAM alignment records decoded on the device (slimm_push_bam_bytes) ----
constexpr uint32_t kBamPiece = 16384;                 // bytes per piece: a lane finds and walks the records that start in it
constexpr uint32_t kBamSlots = kBamPiece / 36 + 2;    // record offsets a piece can hold (a record is at least 36 bytes)
constexpr uint32_t kBamPieceBad = 1;
constexpr uint64_t kBamSlack = 16ull << 20;           // room in front of a window for the incomplete record of the one before
struct BamPiece {
    uint32_t guess, stop, count, flags;  // where its first record starts (guessed, then verified); where its walk ended; records
    uint32_t base, pad[3];               // records of the window in front of it
};
struct BamWindowResult {
    uint32_t n_records, stop, bad, last_piece;  // complete records; offset behind the last one; malformed record met; the last piece holding a record
};
struct BamCarry {  // the name of the last record of the window before (the next window's first record is compared with it)
    uint32_t have, len;
    // Q18 on a GROUPED stream (read_identity.h): a record whose name was SHORTENED (no mate flag, the name ends in ".1" / ".2")
    // is the same read as a flagged record of the shortened name -- which a file grouped by QNAME may hold anywhere.  A run of
    // adjacent records with one canonical base is complete iff it holds an un-shortened record (the QNAME = base group is
    // contiguous, so it is THIS one); a run of shortened records only may have its flagged namesakes elsewhere.  Counted per
    // file: runs that start with a shortened record, and shortened -> plain steps inside a run (at most one per run in a file
    // grouped by QNAME); they differ iff some run holds shortened records only.  last_short: the carried record was shortened.
    uint32_t short_starts, short_to_plain, last_short, pad;
    uint8_t name[256];
};
uint32_t bam_pieces(uint64_t n_bytes);
// record boundaries of bytes[lo, end) (lo: a record starts there) -> pieces, offs (bam_pieces x kBamSlots), *result
void launch_bam_find(hipStream_t st, const uint8_t* bytes, uint64_t lo, uint64_t end, uint32_t n_refs, BamPiece* pieces, uint32_t* offs,
                     BamWindowResult* result);
// the window's records appended at out_at: marked -> ref (the words) + pos; otherwise key, ref, pos, flag, check
void launch_bam_decode(hipStream_t st, const uint8_t* bytes, uint64_t lo, uint64_t end, const BamPiece* pieces, const uint32_t* offs,
                       BamCarry* carry, const BamWindowResult* result, bool marked, uint64_t out_at, uint64_t* key, int32_t* ref,
                       int32_t* pos, uint16_t* flag, uint32_t* check);

// the scan of the pieces' counts + the window's result; sam_lo != 0xffffffff: SAM text (the window's stop is where the last
// complete line ends; sam_lo when there is none)
void launch_bam_scan(hipStream_t st, BamPiece* pieces, uint32_t n_pieces, uint64_t end, BamWindowResult* result, uint32_t sam_lo);

// ---- sam_decode.hip: SAM text decoded on the device (the same window pipeline, pieces of 8 KB: a line is >= 22 bytes) ----
constexpr uint32_t kSamPiece = 8192;
constexpr uint32_t kSamPieceSkip = 2;   // BamPiece::flags: a header line or an empty line among the alignment lines
static_assert(kSamPiece / 22 + 2 <= kBamSlots, "a SAM piece's lines fit the offsets of a BAM piece");
struct SamRefEntry {   // open addressing over the header's reference names (ref < 0: empty)
    uint64_t hash;
    int32_t ref;
    uint32_t name_off, name_len, pad;
};
uint32_t sam_pieces(uint64_t n_bytes);
uint64_t sam_name_hash(const char* s, size_t n);
void launch_sam_find(hipStream_t st, const uint8_t* bytes, uint64_t lo, uint64_t end, BamPiece* pieces, uint32_t* offs, BamWindowResult* result);
void launch_sam_decode(hipStream_t st, const uint8_t* bytes, uint64_t lo, uint64_t end, const BamPiece* pieces, const uint32_t* offs,
                       BamCarry* carry, const BamWindowResult* result, bool marked, uint64_t out_at, uint64_t* key, int32_t* ref, int32_t* pos,
                       uint16_t* flag, uint32_t* check, const SamRefEntry* table, uint32_t table_mask, const uint8_t* names);

// ---- bgzf_inflate.hip: BGZF blocks (DEFLATE streams of <= 64 KB) inflated on the device, a lane per block ----
struct BgzfBlock {
    uint64_t src, dst;     // the DEFLATE payload's offset in the compressed bytes; where its output goes
    uint32_t csize, isize; // payload bytes; inflated bytes (the gzip trailer's ISIZE)
    uint32_t crc;          // the gzip trailer's CRC32 of the inflated bytes
    uint32_t tok;          // where the block's match tokens go (words; bgzf_tokens.hip): the sum of bgzf_token_room of the blocks before
};
// room for the tokens of a block of isize bytes: at most isize / 3 matches and isize / 256 tokens for literal runs too long
// for a match token's field
__host__ __device__ inline uint32_t bgzf_token_room(uint32_t isize) { return isize / 3u + isize / 256u + 8u; }
constexpr uint32_t kBgzfTail = 1024;     // zeroed bytes behind the compressed input: a truncated table header reads into them, never past
constexpr uint32_t kBgzfMaxGrid = 512;   // waves of a launch of the lane-per-block kernel (two per CU: 70 KB of LDS each)
// bgzf_tokens.hip, the fast path: per block the match tokens of k_inflate_decode for k_inflate_resolve
struct InflateInfo {
    uint32_t n_tok;   // tokens of the block
    uint32_t flag;    // 1 = the lane-per-block kernel inflates this block (a stored block inside, or anything irregular)
};
uint32_t bgzf_inflate_grid(uint32_t n_blocks);
size_t bgzf_lanes_scratch_bytes(uint32_t grid);
size_t bgzf_inflate_scratch_bytes(uint32_t n_blocks, uint64_t token_words);   // for launch_bgzf_inflate of up to n_blocks blocks whose token rooms add up to token_words
// status[0] = the largest error code met (0: every block inflated to its ISIZE, CRC right), status[1] = the first bad block
// (preset ~0), status[2] = blocks the lane-per-block kernel inflated (preset 0; status: 4 words)
void launch_bgzf_inflate(hipStream_t st, const uint8_t* comp, const BgzfBlock* blocks, uint32_t n_blocks, uint8_t* out, void* scratch,
                         uint32_t* status);
// the lane-per-block kernel by itself: every block (info == nullptr) or the blocks with info[b].flag set
void launch_bgzf_inflate_lanes(hipStream_t st, const uint8_t* comp, const BgzfBlock* blocks, uint32_t n_blocks, uint8_t* out, void* scratch,
                               uint32_t grid, uint32_t* status, const InflateInfo* info);
// host: the descriptors of the whole BGZF blocks in bytes[0, n_bytes), outputs one behind the other from dst0 on
bool bgzf_parse_blocks(const uint8_t* bytes, uint64_t n_bytes, uint64_t dst0, std::vector<BgzfBlock>& out, uint64_t& inflated,
                       std::string& err);

// ---- group_by_ident.hip: record_order = ANY -- the records of every read identity adjacent, file order kept among them ----
// (stable counting passes over a few bits of a hash of the qName key, then a finish inside the small buckets of equal
// hash bits; the first pass reads the caller's records, so there is no compaction step)
constexpr uint32_t kGroupMaxBits = 11;      // widest digit of a pass
constexpr uint32_t kGroupDefaultBits = 8;   // ... the width a plan aims for (SLIMM_GROUP_WIDTH overrides)
constexpr uint32_t kGroupMaxGrid = 512;     // persistent workgroups of a pass = stretches of the stream = matrix columns
constexpr uint32_t kGroupMaxPasses = 6;
struct GroupPlan {
    uint32_t passes = 1, bits = 8;              // bits = the sum of the passes' widths: hash bits that make a bucket
    uint32_t widths[kGroupMaxPasses] = {8, 0, 0, 0, 0, 0};   // pass p sorts by widths[p] bits; the first pass takes the lowest
    uint32_t width = 8;                         // the widest pass (sizes the count matrix)
    uint32_t grid = kGroupMaxGrid;
};
GroupPlan group_plan(uint32_t n_records);
size_t group_hist_words(const GroupPlan& g);   // count matrix [2^width][grid] + digit totals
struct GroupArrays {
    uint64_t* ident = nullptr;  // qName key << 2 | mate number
    uint2* pay = nullptr;       // {reference, global bin}
    uint32_t* chk = nullptr;    // check words (slimm_push_records_checked), or null
};
struct GroupJob {
    GroupPlan plan;
    DeviceRecords in;           // the caller's records (four arrays or packed)
    uint32_t n_refs = 0;
    const uint2* geo = nullptr; // {contig length, first bin} per reference
    uint32_t half_read = 0, bin_width = 0;
    uint32_t* counters = nullptr;  // counters[CNT_V] = mapped records, written by the first scatter
    GroupArrays a, t;           // the result ends in `a`; `t` is scratch of the same length
    uint32_t* hist = nullptr;   // group_hist_words(plan) words
};
// pass p: count -> scan -> scatter, p = 0 .. plan.passes - 1; then the finish
void launch_group_count(hipStream_t st, const GroupJob& j, uint32_t pass);
void launch_group_scan(hipStream_t st, const GroupJob& j, uint32_t pass);
void launch_group_scatter(hipStream_t st, const GroupJob& j, uint32_t pass);
void launch_group_finish(hipStream_t st, const GroupJob& j);
int group_init();  // once per process and device: dynamic LDS attributes; 0 = ok

}  // namespace slimm
