/* ============================================================================
 * slimm_hip.h -- C ABI of the MI355X-native SLIMM alignment-to-profile path.
 *
 * The reference (seqan/slimm) has no plugin/FFI interface; the seam this library replaces is
 * the trio of `class slimm` member calls made by slimm::get_profiles()
 * (reference src/slimm.hpp:449 analyze_alignments, :464 filter_alignments,
 * :485 get_reads_lca_count, :489 write_abundance) and the public members they communicate
 * through (src/slimm.hpp:103-127).  Each entry point below names the reference code it stands for.
 *
 * Conventions: plain pointers and sizes, no C++ types, no exceptions across the boundary.
 * Every function returning int returns SLIMM_OK (0) or a negative SLIMM_E_* code;
 * slimm_last_error() gives the text.  A context is NOT thread-safe (the reference object is
 * single-threaded, one `slimm` per process: src/slimm.hpp:946-958); use one context per GPU.
 * All `const` host arrays passed in are copied before the call returns; the caller keeps
 * ownership.  Device work runs on a stream owned by the context; every call that returns
 * host-visible results has synchronised that stream.
 * ==========================================================================*/
#ifndef SLIMM_HIP_H
#define SLIMM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLIMM_LINEAGE_LEN 8 /* reference src/misc.hpp:4 LINAGE_LENGTH */

enum {
    SLIMM_OK = 0,
    SLIMM_E_INVALID = -1,   /* bad argument / call order */
    SLIMM_E_HIP = -2,       /* HIP runtime error (no GPU, out of memory, ...) */
    SLIMM_E_REF_RANGE = -3, /* a record names ref_id >= n_refs (undefined behaviour in the reference) */
    SLIMM_E_RUN_LENGTH = -4,/* (not returned any more: a read may have any number of alignment records; the value is
                               kept so that the codes after it do not move) */
    SLIMM_E_KEY_COLLISION = -5, /* records with one read key carry different check words (slimm_push_records_checked) */
    SLIMM_E_REGROUP = -6,   /* a GROUPED stream holds names ending in ".1" / ".2" without a mate flag whose flagged namesakes
                               may lie elsewhere in the file (quirk Q18): push the file again to a SLIMM_ORDER_ANY context */
    SLIMM_E_RETRY = 2,      /* not an error: slimm_install_merged_partials asks for slimm_filter_alignments_launch again */
    SLIMM_E_NO_HITS = 1     /* not an error: no mapped record (reference prints a warning and writes nothing, src/slimm.hpp:451-455) */
};

/* Order of the record stream (SAM @HD SO:/GO: tags).  GROUPED = all records of one read name are
 * contiguous (mapper output order, `GO:query` / `SO:queryname`); ANY = no assumption (the reference
 * accepts any order because it groups through a hash map, src/slimm.hpp:204-211) -- the records are
 * then stably sorted by read identity on the device first. */
enum { SLIMM_ORDER_GROUPED = 0, SLIMM_ORDER_ANY = 1 };

/* Taxonomic ranks, reference src/misc.hpp:24-35. */
enum {
    SLIMM_RANK_STRAIN = 0, SLIMM_RANK_SPECIES, SLIMM_RANK_GENUS, SLIMM_RANK_FAMILY, SLIMM_RANK_ORDER,
    SLIMM_RANK_CLASS, SLIMM_RANK_PHYLUM, SLIMM_RANK_SUPERKINGDOM, SLIMM_RANK_INTERMEDIATE
};

typedef struct slimm_ctx slimm_ctx;

/* What slimm::get_profiles() knows before the record loop starts (src/slimm.hpp:409-445) plus
 * the arg_options the path reads (src/slimm.hpp:49-87). */
typedef struct {
    uint32_t n_refs;          /* header contigs, index = BAM refID */
    const uint32_t* ref_len;  /* [n_refs] contig lengths */
    const uint32_t* lineage;  /* [n_refs*8] taxids per contig: own, species, genus, family, order, class, phylum,
                                 superkingdom = db.ac__taxid[accession]; all-zero row for an accession missing from the
                                 database (src/slimm.hpp:433-442) */
    uint32_t bin_width;       /* -w; 0 = use avg_read_len (src/slimm.hpp:412-413) */
    uint32_t avg_read_len;    /* get_avg_read_length() of the input (src/misc.hpp:509-522) */
    uint32_t min_reads;       /* -mr; 0 = derive 1+(matches-1)/10000 (src/slimm.hpp:458-459); a statistic only */
    float cov_cut_off;        /* -cc quantile (default 0.95) */
    float abundance_cut_off;  /* -ac (default 0.01) */
    const char* rank;         /* -r: "species" (default), "genus", "family", "order", "class", "phylum" */
    /* db.taxid__name (src/misc.hpp:84): rank and name per taxid; taxids not listed read as (strain, "") like the
       reference's default-constructed map entry (src/slimm.hpp:565). */
    uint32_t n_taxa;
    const uint32_t* tax_id;   /* [n_taxa] */
    const uint32_t* tax_rank; /* [n_taxa] SLIMM_RANK_* */
    const char* const* tax_name; /* [n_taxa] NUL-terminated */
    int device;               /* HIP device ordinal */
    int record_order;         /* SLIMM_ORDER_* */
} slimm_config;

/* slimm::slimm(options) + the per-file reference initialisation of get_profiles() (src/slimm.hpp:96-101, 420-445). */
int slimm_create(const slimm_config* cfg, slimm_ctx** out);
void slimm_destroy(slimm_ctx* ctx);
const char* slimm_last_error(const slimm_ctx* ctx); /* ctx may be NULL: error of the calling thread's last failed slimm_create */

/* slimm::reset() (src/slimm.hpp:167-188): forget records and results, keep configuration, allocations and --
 * like the reference -- the cached cut-offs (quirk Q8: they survive reset in -d mode). */
int slimm_reset(slimm_ctx* ctx);
/* Clears the cached cut-offs and the derived min_reads as well (a fresh `slimm` object). */
int slimm_reset_cutoffs(slimm_ctx* ctx);

/* The cached cut-offs (src/slimm.hpp:155-156).  In the reference's -d mode one `slimm` object serves all files, so
 * files 2+ reuse file 1's cut-offs (Q8); a host that creates one context per file carries them over with these. */
int slimm_get_cutoff_cache(slimm_ctx* ctx, float* coverage_cut_off, float* uniq_coverage_cut_off);
int slimm_set_cutoff_cache(slimm_ctx* ctx, float coverage_cut_off, float uniq_coverage_cut_off);
/* options.min_reads as an earlier file left it: with -mr 0 the first file with mapped reads derives it INTO the options
 * (src/slimm.hpp:458-459), where it stays for the files behind it (Q8).  slimm_reset keeps the derived value like
 * slimm::reset() does, slimm_reset_cutoffs restores the configured one; a host with one context per file carries it over
 * with this (slimm_stats.min_reads of the file before). */
int slimm_set_min_reads(slimm_ctx* ctx, uint32_t min_reads);

/* ---- record stream: what the loop of analyze_alignments() reads from each BamAlignmentRecord
 *      (src/slimm.hpp:194-211): qName identity, flag, rID, beginPos; file order. ------------------
 * read_key: identity of the qName (equal names <=> equal keys); only the low 62 bits are significant
 * (the mate number from flag 0x40/0x80 is folded into the two low bits on the device).  The reference's key is the
 * string qName + ".1" / ".2" / nothing (src/slimm.hpp:204-208): an UNFLAGGED record whose name ends in ".1" / ".2" is
 * the same read as a first / last-in-pair record of the name without that suffix (Q18).  A producer that wants that
 * input class reproduced keys such a record by the shortened name and sets the mate bit in the flag it passes:
 * slimm_host_canonical_read_name below does it, and the library's own readers do.  The library compares
 * keys, never names: "equal names <=> equal keys" is the caller's promise.  GROUPED streams only ever compare
 * ADJACENT records, so a producer that hashes names can make the promise exact by comparing every name with the
 * one before it (the slimm command's reader does: host/alignment_file.cpp, separate_adjacent_names); for ANY order
 * a 62-bit hash leaves ~n^2 / 2^63 odds of two different names meeting.
 *
 * Q18 ON A GROUPED STREAM.  A file grouped by QNAME (mapper output, @HD GO:query / SO:queryname) keeps the records of one
 * QNAME adjacent -- not the records of one reference KEY STRING: `r` (flag 0x40) and the unflagged `r.1` are one read of
 * the reference wherever the two names lie in the file (src/slimm.hpp:204-211 merges through a hash map).  Only a record
 * whose name was SHORTENED (no mate flag, ends in ".1" / ".2") can join records of another QNAME, and a run of adjacent
 * records with one canonical base is complete iff it holds an un-shortened record (the QNAME = base records are
 * contiguous, so they are these).  The library's decoders (slimm_push_bam_bytes, _bgzf_blocks, _sam_bytes) therefore count,
 * per file, the runs that consist of shortened names only; when there is one, slimm_analyze_alignments of the GROUPED
 * context returns SLIMM_E_REGROUP: the caller pushes the file again to a context created with SLIMM_ORDER_ANY, which is
 * exact whatever the order (the slimm command does that by itself).  A file without such names pays two register adds per
 * record.  A producer of decoded GROUPED records (slimm_push_records*, slimm_group_push_records*) applies the same rule with
 * slimm_host_q18_note / slimm_host_q18_regroup_needed below -- the library never sees its names.
 *
 * DECLARING A STREAM GROUPED IS A PROMISE THE LIBRARY DOES NOT CHECK BY ITSELF.  With record_order = SLIMM_ORDER_GROUPED
 * the front end compares adjacent records only: a read name that comes back after other names have been in between
 * becomes TWO reads (two unique reads where the reference, which merges through its hash map -- src/slimm.hpp:204-211 --
 * sees one multi-mapped read).  No error is raised, the profile is simply not the reference's.  Declare GROUPED only
 * what is grouped by construction (mapper output; @HD SO:queryname / GO:query, which is all the slimm command trusts);
 * everything else is SLIMM_ORDER_ANY.  slimm_check_grouping() is the diagnostic for a caller in doubt: after the
 * records are pushed it counts, on the device, the qName runs whose identity started an earlier run as well (0 = the
 * stream is grouped).  It costs a hash-set insert per run -- the set is a power of two of 8-byte entries at or above
 * twice the number of records: 16 to 32 bytes of device memory per record, up to 32 GiB near 2^31 records -- and is
 * therefore never run unasked; the slimm command runs it with --verify-grouping and warns.  Packed records are
 * compared by the 61 identity bits they carry (the four-array form: 62). */
int slimm_check_grouping(slimm_ctx* ctx, uint64_t* n_split_names);
int slimm_reserve(slimm_ctx* ctx, uint64_t n_records);
/* Append a batch from host memory (copied to the device before return). */
int slimm_push_records(slimm_ctx* ctx, const uint64_t* read_key, const int32_t* ref_id, const int32_t* begin_pos,
                       const uint16_t* flag, uint64_t n);
/* The same with a CHECK WORD per record: a second, independent hash of the read name (any 32 bits that equal names
 * share).  The library never sees names; with check words it can tell when two different names were given one key:
 * records with one key and different check words that meet -- next to each other in a GROUPED stream, after the device
 * sort for ANY order -- make slimm_finish_coverage return SLIMM_E_KEY_COLLISION instead of silently becoming one read
 * (the reference keys on the full name, src/slimm.hpp:204-211).  4 more bytes per record on the way in (and through
 * the sort).  Checked and unchecked pushes do not mix within a file. */
int slimm_push_records_checked(slimm_ctx* ctx, const uint64_t* read_key, const int32_t* ref_id, const int32_t* begin_pos,
                               const uint16_t* flag, const uint32_t* check, uint64_t n);
/* The lean form of the same stream: 16 bytes per record.  Of `flag` the record loop reads three bits -- unmapped
 * (src/slimm.hpp:197) and first / last in pair (src/slimm.hpp:205-208) -- and they ride in the key's top three bits:
 *   packed_key = (read_key & (2^61 - 1)) | mate << 61 | unmapped << 63,   mate = 1 (flag & 0x40), 2 (flag & 0x80), else 0
 * (slimm_pack_key / slimm_pack_keys do it; the identity of a qName is then its low 61 bits).  No flag array crosses the
 * bus or is read by the front end: 11 % fewer bytes on both.  Packed, unpacked and checked pushes do not mix within a
 * file.  slimm_push_records_packed_async is the streamed form (as slimm_push_records_async: the arrays must stay
 * unchanged until slimm_push_wait), slimm_set_records_device_packed the borrowed-device-arrays form. */
uint64_t slimm_pack_key(uint64_t read_key, uint16_t flag);
void slimm_pack_keys(const uint64_t* read_key, const uint16_t* flag, uint64_t n, uint64_t* packed_key);
int slimm_push_records_packed(slimm_ctx* ctx, const uint64_t* packed_key, const int32_t* ref_id, const int32_t* begin_pos,
                              uint64_t n);
int slimm_push_records_packed_async(slimm_ctx* ctx, const uint64_t* packed_key, const int32_t* ref_id,
                                    const int32_t* begin_pos, uint64_t n);
int slimm_set_records_device_packed(slimm_ctx* ctx, const uint64_t* d_packed_key, const int32_t* d_ref_id,
                                    const int32_t* d_begin_pos, uint64_t n);
/* RUN-MARKED records, 8 bytes each, for input grouped by read name (record_order = SLIMM_ORDER_GROUPED; anything else is
 * refused).  With the records of a name adjacent, the read identity the reference keys its hash map with (src/slimm.hpp:
 * 204-211) is "the qName run this record lies in" -- the device needs no name, only where a run starts.  A record is
 *   word      = reference id + 1 (0: not mapped -- the unmapped flag, src/slimm.hpp:197, or reference -1)
 *               | mate number << 29 (0 / 1 / 2, src/slimm.hpp:205-208) | (this record's qName differs from the one
 *               before it, or it is the file's first) << 31
 *   begin_pos
 * and the producer compares adjacent names instead of hashing them.  slimm_mark_word builds one word, slimm_mark_words
 * a batch from the four-array form (run starts where adjacent keys differ; *prev_key = the key in front of the batch,
 * NULL at the start of a file).  Half the bytes of the packed form cross the bus and are read by the front end.  What the
 * form gives up: there are no names to check -- slimm_check_grouping and the *_checked pushes do not apply -- and a file
 * is this form throughout.  _async / set_records_device / staged: as for the packed form (a staging set's ref_id array
 * holds the words). */
uint32_t slimm_mark_word(int32_t ref_id, uint16_t flag, int starts_run);
void slimm_mark_words(const uint64_t* read_key, const uint16_t* flag, const int32_t* ref_id, uint64_t n, const uint64_t* prev_key,
                      uint32_t* word);
int slimm_push_records_marked(slimm_ctx* ctx, const uint32_t* word, const int32_t* begin_pos, uint64_t n);
int slimm_push_records_marked_async(slimm_ctx* ctx, const uint32_t* word, const int32_t* begin_pos, uint64_t n);
int slimm_set_records_device_marked(slimm_ctx* ctx, const uint32_t* d_word, const int32_t* d_begin_pos, uint64_t n);
/* Streamed ingest.  slimm_push_records_async enqueues the copies on the context's copy stream and returns at once:
 * the arrays must stay unchanged until slimm_push_wait() returns (page-locked arrays are read by the DMA engine
 * directly; pageable ones still work, at the speed of the runtime's own staging).  slimm_analyze_alignments() is
 * ordered behind the copies on the device (an event), so the host never waits for PCIe: it decodes the next batch, or
 * drives another context's phases, meanwhile. */
int slimm_push_records_async(slimm_ctx* ctx, const uint64_t* read_key, const int32_t* ref_id, const int32_t* begin_pos,
                             const uint16_t* flag, uint64_t n);
int slimm_push_wait(slimm_ctx* ctx);
/* Two page-locked staging sets owned by the context, for producers that decode records piecemeal (the BAM reader of
 * the slimm command): fill set `which` (0 or 1; at least `capacity` records each), hand it over with
 * slimm_push_staged_async(ctx, which, n) and fill the other one meanwhile; slimm_staging_buffers / slimm_staging_wait
 * block until the set's last copy has left it. */
int slimm_staging_buffers(slimm_ctx* ctx, uint32_t which, uint64_t capacity, uint64_t** read_key, int32_t** ref_id,
                          int32_t** begin_pos, uint16_t** flag);
int slimm_push_staged_async(slimm_ctx* ctx, uint32_t which, uint64_t n);
/* The same for a set whose key array the producer filled with PACKED keys (slimm_pack_key; the set's flag array is not
 * read): 16 bytes per record over the bus. */
int slimm_push_staged_packed_async(slimm_ctx* ctx, uint32_t which, uint64_t n);
/* ... and for a set whose ref_id array holds run-marked words (slimm_mark_word; key and flag arrays are not read). */
int slimm_push_staged_marked_async(slimm_ctx* ctx, uint32_t which, uint64_t n);
int slimm_staging_wait(slimm_ctx* ctx, uint32_t which);
/* BAM ALIGNMENT RECORDS DECODED ON THE DEVICE.  Instead of decoding records on the host, a caller that holds a BAM file
 * inflates its BGZF blocks and hands over the bytes BEHIND the BAM header -- the alignment records as the SAM/BAM
 * specification lays them out (section 4.2: block_size | refID | pos | l_read_name | mapq | bin | n_cigar_op | flag | l_seq |
 * next_refID | next_pos | tlen | read_name | ...) -- in windows of any size, in file order; records may straddle windows.
 * The device finds the record boundaries and extracts what the record loop reads (src/slimm.hpp:194-208: refID, position,
 * flag, read name), appending to the context's record stream in the form its record order takes: SLIMM_ORDER_GROUPED ->
 * run-marked 8-byte records (a record starts a run where its NAME differs from the name of the record before it: exact,
 * no hash involved); SLIMM_ORDER_ANY -> key (the 62-bit hash of the name that host/alignment_file.cpp computes), refID,
 * position, flag and a check word (as slimm_push_records_checked).  `bytes`: any host memory (page-locked -- see
 * slimm_pin_host_buffer -- the DMA engine reads it directly).  A window's copy is started by the call that hands it
 * over and runs beside the device's work on the window BEFORE it: the buffer must stay unchanged until the NEXT call on
 * this context returns (a call with last != 0 finishes everything), so hand over three or more buffers in rotation.
 * *n_records (may be NULL): records appended by this call -- those of the windows it finished: a window is found, counted and
 * decoded once two later ones have been handed over or more than 4 GB of windows are in flight (they are copied, or inflated, meanwhile), all of them with last != 0.  last != 0: nothing follows (n_bytes may be 0) -- bytes of an incomplete record are then
 * an error (SLIMM_E_INVALID, "truncated BAM record"), as is a malformed record in any window.  A record longer than
 * 16 MiB is not supported in this form (SLIMM_E_INVALID: decode such a file on the host).  The forms do not mix within
 * a file.  Replaces: seqan::readRecord in src/slimm.hpp:194-208 / src/misc.hpp:509-522. */
int slimm_push_bam_bytes(slimm_ctx* ctx, const uint8_t* bytes, uint64_t n_bytes, int last, uint64_t* n_records);
/* The same with the inflate on the device as well: `blocks` = n_bytes of whole BGZF blocks of the file (compressed, as they
 * lie in it, one behind the other, in file order across calls), of whose inflated bytes the first `skip` are not alignment
 * records (the BAM header, of any size, up to its end in the file's first record-bearing block; 0 in every later call).  The
 * compressed bytes cross the bus and are inflated on the device (slimm_amd/csrc/bgzf_tokens.hip: Huffman decode a lane per
 * block, matches filled a workgroup per block; bgzf_inflate.hip behind it; ISIZE, the DEFLATE blocks' form and the CRC32 are
 * checked), and the records are found and decoded as above.  Calls of any size: the library gathers them into device windows
 * of 1.4 - 1.9 GB of inflated bytes (tens of thousands of blocks) and inflates those on two streams in turn.  Windows of this
 * form and of slimm_push_bam_bytes may alternate within a file; buffer lifetime, `last`, *n_records and the errors are those of
 * slimm_push_bam_bytes, plus
 * SLIMM_E_INVALID for anything that is not a BGZF block or does not inflate to its ISIZE. */
int slimm_push_bgzf_blocks(slimm_ctx* ctx, const uint8_t* blocks, uint64_t n_bytes, uint32_t skip, int last, uint64_t* n_records);
/* DEVICE MEMORY OF THE WINDOW PIPELINE.  A gathered window holds [16 MiB | up to 1.9 GB inflated] + its compressed bytes
 * (window / ratio + one push) + 16 B per block + the inflater's scratch (~0.36 B per inflated byte, two sets) + 456 record
 * offsets per 16 KB; up to four windows are in turn in flight.  slimm_set_input_size_hint(ctx, the file's COMPRESSED size),
 * called before a file's first window, lets the library reserve exactly what the file will use at its first push (a 70 MB
 * file: one window of ~0.3 GB; a file of many gigabytes: four windows, ~14 GB at a ratio of 3) -- reserving ahead matters
 * because an allocation made while inflate kernels run waits for them.  Without a hint nothing is reserved ahead and buffers
 * appear as windows need them (the same totals, some stalls).  slimm_reset keeps up to 4 GiB of these buffers for the next
 * file and releases them above that; slimm_window_memory reports what is held.  The hint is per file (cleared by slimm_reset). */
int slimm_set_input_size_hint(slimm_ctx* ctx, uint64_t compressed_bytes);
int slimm_window_memory(slimm_ctx* ctx, uint64_t* device_bytes);
/* hipMemGetInfo of the context's device: bytes in use (by every process and context on it) and the device's total. */
int slimm_device_memory(slimm_ctx* ctx, uint64_t* used_bytes, uint64_t* total_bytes);
/* SAM TEXT decoded on the device (slimm_amd/csrc/sam_decode.hip): the reference takes .sam and .bam alike
 * (src/file_helper.hpp:73-75; the record loop src/slimm.hpp:194-208 reads QNAME, FLAG, RNAME -> reference index, POS).
 * `text` = the file's alignment lines, everything behind the header, in windows cut ANYWHERE (the incomplete last line of a
 * window is carried in front of the next one; a last line without its newline is a line); same buffer-lifetime and window
 * rules as slimm_push_bam_bytes, the two do not mix within a file.  RNAME is looked up in the header's reference names, which
 * slimm_set_reference_names(ctx, names[n_refs]) hands over once per context (index = reference id; "*" and names the header
 * does not have give no reference, like the host reader); QNAME is compared with the line before (grouped input) or hashed
 * (any order) in its canonical form (quirk Q18).  A line with fewer than ten fields is an error like the host reader's; a
 * header line or an empty line AMONG the alignment lines (which the host reader skips) is refused with "decode this file on
 * the host". */
int slimm_set_reference_names(slimm_ctx* ctx, const char* const* names);
int slimm_push_sam_bytes(slimm_ctx* ctx, const uint8_t* text, uint64_t n_bytes, int last, uint64_t* n_records);
/* Page-locks a buffer of the caller (hipHostRegister) until the context is destroyed: copies out of it then run at the
 * speed of the bus instead of the runtime's own staging. */
int slimm_pin_host_buffer(slimm_ctx* ctx, const void* buffer, uint64_t n_bytes);
/* Use records already resident in device memory, without copying; the arrays must stay valid and unchanged
 * until slimm_reset().  Replaces anything pushed before. */
int slimm_set_records_device(slimm_ctx* ctx, const uint64_t* d_read_key, const int32_t* d_ref_id,
                             const int32_t* d_begin_pos, const uint16_t* d_flag, uint64_t n);
/* The device arrays of the records a context holds after its pushes, whatever brought them there (decoded records, or the
 * window pipeline's decoders): for a caller that hands stretches of them to other contexts (slimm_set_records_device*) -- a
 * group that lets ONE member read and decode a file does (slimm_group_get_profiles).  *form: 0 four arrays, 1 packed
 * (d_flag NULL), 2 run-marked (d_read_key and d_flag NULL, d_ref_id holds the words).  Waits for the context's copies and
 * decode kernels; the pointers stay valid until slimm_reset / slimm_destroy. */
int slimm_records_device(slimm_ctx* ctx, const uint64_t** d_read_key, const int32_t** d_ref_id, const int32_t** d_begin_pos,
                         const uint16_t** d_flag, uint64_t* n, int* form);

/* ---- phase A: slimm::analyze_alignments() (src/slimm.hpp:191-303) on this context's records:
 * grouping by read, first-bin per (read, ref), cov / uniq_cov histograms.  Local to this GPU. */
int slimm_analyze_alignments(slimm_ctx* ctx);

/* Stream-ordered exchange.  The context's kernels run on one HIP stream (slimm_get_stream: a hipStream_t).  By default
 * every function that hands out a device buffer for a collective synchronises that stream first, so that any stream
 * may read the buffer.  A caller that enqueues its collectives ON the context's stream (ncclAllGather(..., stream),
 * torch.cuda.ExternalStream) turns that off with slimm_set_stream_ordered(ctx, 1): buffers are then valid in stream
 * order only, and between slimm_analyze_alignments and the cut-offs the host never waits for the device. */
int slimm_get_stream(slimm_ctx* ctx, void** hip_stream);
int slimm_set_stream_ordered(slimm_ctx* ctx, int on);

/* Multi-GPU exchange point (no reference counterpart; reads are sharded across ranks):
 * device buffer [cov | uniq_cov | 16 scalar words] of *n_words uint32 that the caller sums across ranks in place
 * (one all-reduce) between slimm_analyze_alignments() and slimm_finish_coverage().  The stream is synchronised. */
int slimm_coverage_buffer(slimm_ctx* ctx, void** d_ptr, uint64_t* n_words);
/* The third coverage array, uniq_cov2 (reference_contig.hpp:112), after slimm_filter_alignments / slimm_install_merged_
 * partials: *n_words uint32 (every reference padded like in slimm_coverage_buffer) that the caller may sum across ranks
 * in place so that slimm_get_bins(ctx, 2, ...) returns the global array.  The stream is synchronised unless
 * slimm_set_stream_ordered is on. */
int slimm_uniq_cov2_buffer(slimm_ctx* ctx, void** d_ptr, uint64_t* n_words);

/* Multi-GPU, optional: announce before slimm_analyze_alignments that slimm_coverage_summary will be called, so that
 * the histogram kernels write the 'bin != 0' bitmaps while the finished tiles are still in LDS (otherwise
 * slimm_coverage_summary streams both coverage arrays once more to build them).  n_slices: 0 = off; 1 = one
 * [cov bits | uniq_cov bits] block (all-gather form, below); n > 1 = the bitmaps cut into n slices of bin tiles, slice j
 * = [cov bits | uniq_cov bits] of rank j's share, for the all-to-all form. */
int slimm_prepare_summary(slimm_ctx* ctx, uint32_t n_slices);

/* The coverage arrays (cov, uniq_cov, uniq_cov2: reference src/reference_contig.hpp:70-72) are intermediate results of
 * the path: the profile needs their per-reference sums and non-zero bin counts, which the tile kernels take from every
 * finished tile while it is in LDS.  on = 0 before slimm_analyze_alignments: the finished tiles are not written to HBM
 * (only what the reference's -co output, slimm_get_bins and slimm_coverage_buffer would read); those calls then fail
 * with SLIMM_E_INVALID.  Default: on = 1. */
int slimm_keep_bins(slimm_ctx* ctx, int on);
/* Leaner exchange for the same point, used by default by slimm_amd/distributed.py: the cut-offs only need per-reference
 * SUMS of cov / uniq_cov (additive) and per-reference counts of NON-ZERO bins (popcount of the OR of every rank's
 * "bin != 0" bitmap).  slimm_coverage_summary() builds [sums | 16 scalars | cov bits | uniq_cov bits] for this rank in
 * device memory (*n_words uint32, about 1/16 of the bins); the caller all-gathers the summaries of all ranks into one
 * device buffer (rank-major, contiguous) and hands it to slimm_finish_coverage_merged(), which replaces
 * slimm_finish_coverage().  cov / uniq_cov then stay per-rank partial sums (slimm_get_bins returns this rank's share).
 * Device memory the caller hands in (here and in the calls below) is read by kernels on the CONTEXT'S stream
 * (slimm_get_stream), which waits for no other stream: whatever wrote it -- a collective, a copy -- must be complete, or have
 * been enqueued on that stream. */
int slimm_coverage_summary(slimm_ctx* ctx, void** d_ptr, uint64_t* n_words);
int slimm_finish_coverage_merged(slimm_ctx* ctx, const void* d_gathered, uint32_t n_ranks);
/* All-to-all form of the same exchange for many ranks (each rank receives 1/n of every other rank's bitmaps instead of
 * all of them).  With slimm_prepare_summary(ctx, n) in effect the buffer of slimm_coverage_summary is
 * [4 n_refs + 16 words | n chunks of equal size]; send chunk j to rank j (ncclAllToAll / all_to_all_single) and hand the
 * n received chunks to slimm_merge_summary_slices: it ORs them, counts the non-zero bins of this rank's slice per
 * reference and returns a device vector of 4 n_refs + 16 words [own sums, partial non-zero counts | own scalars] that
 * the ranks sum in place (all-reduce, int32).  slimm_finish_coverage_reduced then takes the place of
 * slimm_finish_coverage. */
int slimm_merge_summary_slices(slimm_ctx* ctx, const void* d_received, uint32_t n_ranks, uint32_t my_rank, void** d_vec,
                               uint64_t* n_words);
int slimm_finish_coverage_reduced(slimm_ctx* ctx);

/* End of phase A: per-reference reads_count / uniq_reads_count / non-zero bin counts
 * (reference_contig.hpp:84-91,148-155) from the (reduced) bins, and the float statistics of src/slimm.hpp:259-302.
 * Returns SLIMM_E_NO_HITS when hits_count == 0. */
int slimm_finish_coverage(slimm_ctx* ctx);

/* Install phase-A per-reference results computed elsewhere, instead of slimm_analyze_alignments() +
 * slimm_finish_coverage(): for host-only contexts (device = -1, no GPU touched) that run just the scalar host part
 * of the path -- cut-offs, propagation, profile -- e.g. on a rank that merges results.  uniq_matches is the sum of
 * uniq_reads_count. */
int slimm_set_coverage_columns(slimm_ctx* ctx, const uint32_t* reads_count, const uint32_t* uniq_reads_count,
                               const uint32_t* nz_cov, const uint32_t* nz_uniq_cov, uint32_t hits_count,
                               uint32_t matches_count);

/* ---- phase B + C(1): slimm::filter_alignments() (src/slimm.hpp:351-392): cut-offs (coverage_cut_off :328-344,
 * uniq_coverage_cut_off :672-688, get_quantile_cut_off misc.hpp:197-216), valid set, per-read update
 * (read_stat.hpp:98-114), uniq_cov2 -- fused on the device with the per-read LCA of
 * get_reads_lca_count() step 1 (src/slimm.hpp:536-557, get_lca :516-531). Local to this GPU. */
int slimm_filter_alignments(slimm_ctx* ctx);

/* Multi-GPU form of slimm_filter_alignments with ONE host synchronisation for the whole phase: this call only
 * launches (the device work of slimm_filter_alignments plus the packing of this rank's additive partial results);
 * the caller sums the buffer of slimm_partials_buffer across ranks on the context's stream (slimm_get_stream) and
 * calls slimm_install_merged_partials, which copies the merged results to the host and synchronises.  When the
 * (taxon, reference) pair set of SOME rank overflowed (the flag travels with the sums, so every rank sees it),
 * slimm_install_merged_partials returns SLIMM_E_RETRY on every rank -- each has grown its table -- and the caller
 * goes round again: launch, sum, install. */
int slimm_filter_alignments_launch(slimm_ctx* ctx);

/* Multi-GPU: the additive partial results of phase B/C(1) of this rank, to be merged across ranks by the caller.
 *   uniq_reads_count2 [n_refs], lca_count [n_taxa_dense] (sum), level_marks [n_refs] (bitwise OR),
 *   pairs: (dense taxon << 32 | ref) of reads whose refs agree at no level (set union), scalars[4] (sum). */
typedef struct {
    uint32_t n_refs, n_taxa_dense;
    uint32_t* uniq_reads_count2;
    uint32_t* lca_count;
    uint32_t* level_marks;
    uint64_t* pairs;
    uint32_t n_pairs;
    uint32_t scalars[4]; /* [0] uniq_matches_count2 */
} slimm_partials;
/* The dense taxon index used by lca_count and pairs: ascending distinct taxids of the lineage table. */
int slimm_dense_taxa(slimm_ctx* ctx, uint32_t* n, const uint32_t** taxid);
int slimm_get_partials(slimm_ctx* ctx, slimm_partials* out);       /* pointers stay owned by ctx, valid until reset */
int slimm_set_partials(slimm_ctx* ctx, const slimm_partials* in);  /* install merged values (copied) */

/* Multi-GPU, device-side merge of the additive partial results (faster than get/set_partials through the host).
 * After slimm_filter_alignments: a device buffer of n_words 32-bit words
 *   [n_refs uniq_reads_count2 | n_taxa_dense LCA counts | 2 n_refs level marks, one 8-bit field per level | 1 pair count]
 * that the ranks sum in place (ncclAllReduce(ncclSum, ncclInt32) from C++, torch.distributed.all_reduce on a tensor
 * aliasing it); at most 255 ranks, so that no field carries into the next.  slimm_install_merged_partials then copies
 * it back and installs it.  *total_pairs = (taxon, reference) pairs over all ranks (src/slimm.hpp:551-556, the
 * no-level-agrees case): when it is not zero, gather every rank's pairs (slimm_get_partials) and install the union with
 * slimm_set_partials. */
int slimm_partials_buffer(slimm_ctx* ctx, void** d_ptr, uint64_t* n_words);
int slimm_install_merged_partials(slimm_ctx* ctx, uint32_t* total_pairs);

/* ---- phase C(2,3): the propagation part of slimm::get_reads_lca_count() (src/slimm.hpp:560-610). */
int slimm_get_reads_lca_count(slimm_ctx* ctx);

/* The per-file body of slimm::get_profiles() (src/slimm.hpp:447-489) on one GPU in one call:
 * slimm_analyze_alignments, slimm_finish_coverage, slimm_filter_alignments, slimm_get_reads_lca_count and, when path is
 * not NULL, slimm_write_abundance_file.  Returns SLIMM_E_NO_HITS (nothing written) when no record is mapped. */
int slimm_get_profiles(slimm_ctx* ctx, const char* path);

/* ---- slimm::write_abundance() (src/slimm.hpp:733-843): the final profile.  The text is identical in format to
 * the reference's <prefix>_profile.tsv; rows come in ascending taxid order (the reference's order is that of an
 * unordered_map and carries no meaning).  The buffer is owned by ctx. */
int slimm_write_abundance(slimm_ctx* ctx, const char** text, uint64_t* len);
/* Same, written to a file (path = get_tsv_file_name(...) result, computed by the caller). */
int slimm_write_abundance_file(slimm_ctx* ctx, const char* path);

/* ---- results (the public members of class slimm, src/slimm.hpp:105-127) ---- */
typedef struct {
    uint32_t hits_count, matches_count, uniq_matches_count, uniq_hits_count, uniq_matches_count2;
    uint32_t reference_count, matched_ref_length;
    uint32_t failed_by_cov, failed_by_uniq_cov, failed_by_min_read, n_valid;
    uint32_t bin_width, min_reads, avg_read_len;
    uint32_t profile_count, profile_failed;
    float coverage_cut_off, uniq_coverage_cut_off, expected_coverage;
    uint64_t n_records, n_targets; /* records pushed; distinct (read, ref) pairs */
    uint64_t total_bins;           /* sum over refs of len/bin_width + 1 */
} slimm_stats;
int slimm_get_stats(slimm_ctx* ctx, slimm_stats* out);

/* Per-reference columns; any pointer may be NULL.  All arrays [n_refs]. */
typedef struct {
    uint32_t* reads_count;
    uint32_t* uniq_reads_count;
    uint32_t* uniq_reads_count2;
    uint32_t* nbins;
    uint32_t* nz_cov;       /* cov.none_zero_bin_count() */
    uint32_t* nz_uniq_cov;
    uint32_t* nz_uniq_cov2;
    uint8_t* valid;         /* member of valid_ref_ids */
    float* abundance;       /* src/slimm.hpp:266-279 */
    float* uniq_abundance;  /* src/slimm.hpp:287-300 */
} slimm_ref_columns;
int slimm_get_ref_columns(slimm_ctx* ctx, slimm_ref_columns* out);

/* Coverage bins of every reference, concatenated without padding in refID order (total_bins words).
 * which: 0 = cov, 1 = uniq_cov, 2 = uniq_cov2 (reference_contig.hpp:110-112). */
int slimm_get_bins(slimm_ctx* ctx, int which, uint32_t* out);

/* The reads' target lists after phase A (the reference's `reads[*].targets`, src/read_stat.hpp:46-59, 61-74): one entry
 * per distinct (read, reference) pair, the targets of a read contiguous, reads in the order the device emitted them.
 *   ref[i]   reference id | bit 31: first target of its read
 *   gbin[i]  bin of the pair's first record, counted over all references (bins of the references before it included,
 *            every reference padded to a multiple of 4 bins) | bit 31: the read has this one target only
 * Call with ref = gbin = NULL to get the number of entries in *n; otherwise cap entries are available and *n are written. */
int slimm_get_read_targets(slimm_ctx* ctx, uint32_t* ref, uint32_t* gbin, uint64_t cap, uint64_t* n);

/* taxon_id__read_count (src/slimm.hpp:126). stage 0: direct LCA hits only (:536-557); 1: final (:560-610). */
int slimm_taxon_count_size(slimm_ctx* ctx, int stage, uint32_t* n);
int slimm_get_taxon_counts(slimm_ctx* ctx, int stage, uint32_t* taxid, uint32_t* count);
/* taxon_id__children (src/slimm.hpp:127) flattened to (taxid, ref) pairs, sorted. */
int slimm_children_pairs_size(slimm_ctx* ctx, int stage, uint64_t* n);
int slimm_get_children_pairs(slimm_ctx* ctx, int stage, uint32_t* taxid, uint32_t* ref);

/* ---- measurement ---- */
/* When enabled, every kernel launch is bracketed by HIP events on the context's stream. */
int slimm_enable_kernel_timing(slimm_ctx* ctx, int on);
/* Restrict the bracketing to one kernel (a name slimm_kernel_times reports); NULL or "" = all kernels again.  Events
 * cost a few microseconds of stream idle time each, so a throughput run times only the kernel it reports on. */
int slimm_time_only_kernel(slimm_ctx* ctx, const char* name);
/* Names (static strings) and accumulated milliseconds / launch counts since the last call with reset != 0. */
int slimm_kernel_times(slimm_ctx* ctx, const char** names, double* ms, uint32_t* launches, uint32_t cap,
                       uint32_t* n, int reset);

/* Diagnostic for record_order = SLIMM_ORDER_ANY: the stream as the device grouped it by read identity for the front end
 * (after slimm_analyze_alignments): per mapped record its identity (qName key << 2 | mate number, src/slimm.hpp:204-208),
 * reference and global bin, records of one identity adjacent and in file order.  Copies min(cap, *n) records to host
 * arrays (any of which may be NULL); *n = the number of mapped records. */
int slimm_grouped_records(slimm_ctx* ctx, uint64_t* ident, uint32_t* ref, uint32_t* gbin, uint64_t cap, uint64_t* n);
/* The grouping's plan for a stream of n_records (record_order = SLIMM_ORDER_ANY): counting passes over the records, bits
 * per pass, hash bits that make a bucket (= passes * width), persistent workgroups per pass. */
void slimm_group_plan(uint64_t n_records, uint32_t* passes, uint32_t* width, uint32_t* bits, uint32_t* grid);

/* ---- host-only helpers (no GPU needed; used by the host driver and testable on CPU) ---- */
/* get_avg_read_length (src/misc.hpp:509-522). Returns 0 when no record has a sequence (the reference divides by 0). */
uint32_t slimm_host_avg_read_length(const uint32_t* l_seq, uint64_t n, uint32_t sample_size);
/* get_quantile_cut_off<float> (src/misc.hpp:197-216), float32, same operation order. */
float slimm_host_quantile_cut_off(const float* v, uint32_t n, float q);
/* Bin of one record (src/slimm.hpp:200-201). */
uint32_t slimm_host_bin_of(int32_t begin_pos, uint32_t avg_read_len, uint32_t ref_len, uint32_t bin_width);
/* Canonical identity of one record's read (src/slimm.hpp:204-208, quirk Q18).  The reference keys its reads by the STRING
 * qName + ".1" (flag 0x40) / ".2" (else flag 0x80) / nothing, so a first-in-pair record of read "N" and an unflagged
 * record of a read literally named "N.1" are ONE read.  (base, mate) below is a bijection with those strings:
 *   (name, 1) if flag & 0x40;  (name, 2) else if flag & 0x80;  else (name minus its last two bytes, 1 / 2) if the name
 *   ends in ".1" / ".2";  else (name, 0).
 * Returns the length of the base and, in *flag_out, the flag with the mate bit the base carries.  A producer hashes (or
 * compares) name[0, base) and hands the library *flag_out: then "equal keys and equal mate <=> equal reference key
 * string" holds.  The library's own readers (host/alignment_file.cpp, bam_decode.hip) do exactly this. */
uint32_t slimm_host_canonical_read_name(const char* name, uint32_t name_len, uint16_t flag, uint16_t* flag_out);
/* Q18 for a producer of GROUPED records (see "Q18 ON A GROUPED STREAM" above): one slimm_q18_runs per file, zeroed;
 * slimm_host_q18_note for EVERY record in file order -- starts_run: its canonical base differs from the base of the record
 * before it (the file's first record: 1), shortened: slimm_host_canonical_read_name returned less than name_len --;
 * slimm_host_q18_regroup_needed != 0 at the end of the file: some run holds shortened names only, declare the file
 * SLIMM_ORDER_ANY.  (What the device decoders count: slimm_amd/csrc/kernels.h, BamCarry.) */
typedef struct slimm_q18_runs {
    uint64_t short_starts;    /* runs whose first record has a shortened name */
    uint64_t short_to_plain;  /* steps from a shortened to an un-shortened name inside a run */
    int last_short;           /* the record before was shortened */
} slimm_q18_runs;
void slimm_host_q18_note(slimm_q18_runs* q, int starts_run, int shortened);
int slimm_host_q18_regroup_needed(const slimm_q18_runs* q);
/* The same two counts of the file a context's device decoders have read so far (0, 0 for any other record form). */
int slimm_get_q18_runs(slimm_ctx* ctx, uint64_t* short_starts, uint64_t* short_to_plain);
/* hipDeviceReset() of the process's devices: for a host about to leave the process.  Contexts must not be used afterwards. */
int slimm_shutdown(void);
/* Library build info. */
/* ---- Several GPUs in one process (slimm_amd/csrc/group.hip): a group of contexts, one per device, used like one.
 * Records are dealt to the members by read as they are pushed (name-grouped streams: contiguous stretches of the file cut
 * at qName-run boundaries, member after member; any other order: key mod n); slimm_group_get_profiles runs the phases on
 * every member with the two exchanges of the multi-rank path in between -- ncclAllGather of the coverage summaries,
 * ncclAllReduce of the partial results, both enqueued on the members' own streams (RCCL is dlopen()ed when the group is
 * created; without it, or when one device is named several times, the same collectives are device-to-device copies and
 * a summing kernel) -- and writes the profile from member 0.  cfg->device is ignored; results (slimm_get_stats,
 * slimm_get_ref_columns, ...) are read from slimm_group_context(g, 0), whose per-reference columns, scalars and
 * per-taxon counts are the merged ones.  The coverage arrays stay per-member partial sums. */
typedef struct slimm_group slimm_group;
int slimm_group_create(const slimm_config* cfg, const int* devices, uint32_t n_devices, slimm_group** out);
void slimm_group_destroy(slimm_group* g);
const char* slimm_group_last_error(const slimm_group* g); /* g may be NULL: error of the last failed slimm_group_create */
uint32_t slimm_group_size(const slimm_group* g);
slimm_ctx* slimm_group_context(slimm_group* g, uint32_t i);
int slimm_group_uses_rccl(const slimm_group* g);
int slimm_group_reset(slimm_group* g);
int slimm_group_push_records(slimm_group* g, const uint64_t* read_key, const int32_t* ref_id, const int32_t* begin_pos,
                             const uint16_t* flag, uint64_t n);
/* Packed 16-byte records (slimm_push_records_packed; the identity of a read name is the key's low 61 bits). */
int slimm_group_push_records_packed(slimm_group* g, const uint64_t* packed_key, const int32_t* ref_id, const int32_t* begin_pos,
                                    uint64_t n);
/* Run-marked 8-byte records (slimm_push_records_marked; groups created for grouped input): whole runs go to one member. */
int slimm_group_push_records_marked(slimm_group* g, const uint32_t* word, const int32_t* begin_pos, uint64_t n);
/* With a check word per record (slimm_push_records_checked): two names that collide in the key land on the same member
 * whichever way the records are dealt, so a group reports SLIMM_E_KEY_COLLISION exactly where one context would. */
int slimm_group_push_records_checked(slimm_group* g, const uint64_t* read_key, const int32_t* ref_id, const int32_t* begin_pos,
                                     const uint16_t* flag, const uint32_t* check, uint64_t n);
/* The exchange between phase A and the cut-offs: SUMMARY = ncclAllGather of [per-reference sums | scalars | one bit per
 * bin]; SLICED = all-to-all of the bitmaps in one slice per member (ncclSend / ncclRecv in one group) + a small
 * ncclAllReduce; BINS = the ncclAllReduce(ncclSum) over the integer coverage bins themselves [cov | uniq_cov | scalars],
 * after which -- and after a second all-reduce of uniq_cov2 behind phase B -- every member's slimm_get_bins returns the
 * GLOBAL arrays (what the reference's -ro / -co outputs read, src/slimm.hpp:846-943; nz_uniq_cov2 of
 * slimm_get_ref_columns stays the member's own count: count the non-zero bins of the global array instead).
 * AUTO (default) = SUMMARY up to two members, SLICED above. */
enum { SLIMM_EXCHANGE_AUTO = 0, SLIMM_EXCHANGE_SUMMARY = 1, SLIMM_EXCHANGE_SLICED = 2, SLIMM_EXCHANGE_BINS = 3 };
int slimm_group_set_exchange(slimm_group* g, int mode);
int slimm_group_exchange(const slimm_group* g); /* the form in effect (what AUTO resolves to) */
int slimm_group_get_profiles(slimm_group* g, const char* path); /* path may be NULL; SLIMM_E_NO_HITS like slimm_get_profiles */
/* A GROUPED file may also reach a group through ONE member: push its windows to slimm_group_context(g, 0) (slimm_push_bam_bytes /
 * _bgzf_blocks / _sam_bytes: the device inflates and decodes at the single-context rate) and nothing through
 * slimm_group_push_records*; slimm_group_get_profiles then deals member 0's run-marked records to the members in contiguous
 * stretches cut at qName-run starts, device to device (8 bytes per record), before phase A.  SLIMM_E_REGROUP from member 0
 * (Q18) comes back as the group's return code: such a file goes through slimm_group_push_records* in any order. */

/* Starts the HIP runtime on `device` (what the first slimm_create of a process would otherwise pay, 0.1 - 0.3 s): for
 * hosts that call it from a thread of their own while they load their database and open their input. */
int slimm_warm_up(int device);
const char* slimm_version(void);

/* BGZF blocks inflated on the device, by themselves: `blocks` = n_bytes of whole BGZF blocks (gzip members with the BC
 * extra field, as in a .bam / .bgz file) in host memory, one behind the other; their inflated bytes, one block's behind the
 * other's, go to out[0, *out_bytes) in host memory (slimm_amd/csrc/bgzf_tokens.hip, bgzf_inflate.hip); ISIZE, the
 * well-formedness of every DEFLATE block and the CRC32 of the gzip trailer are checked.  *kernel_ms (may be null): the
 * inflate kernel alone.  Errors (-1 bad input / a corrupt block, -2 HIP) come with a message in err[0, err_cap).
 * Replaces, together with slimm_push_bam_bytes, what seqan::BamFileIn does for the reference (call sites src/misc.hpp:498-522,
 * src/slimm.hpp:194-208).  slimm_push_bgzf_blocks feeds a context's record stream the same way without the round trip. */
int slimm_bgzf_inflate(int device, const uint8_t* blocks, uint64_t n_bytes, uint8_t* out, uint64_t out_cap, uint64_t* out_bytes,
                       double* kernel_ms, char* err, uint64_t err_cap);
/* The same with the choice of kernels and a count.  how = 0: the two-phase kernels (slimm_amd/csrc/bgzf_tokens.hip: a lane per
 * block decodes the Huffman codes into literals in place + match tokens, a workgroup per block fills the matches in LDS and
 * checks the CRC), with the lane-per-block kernel behind them for what they hand over -- blocks with a stored DEFLATE block
 * inside, and anything irregular; how = 1: the lane-per-block kernel for every block.  *lane_blocks (may be null): the blocks
 * the lane-per-block kernel inflated. */
int slimm_bgzf_inflate_with(int device, const uint8_t* blocks, uint64_t n_bytes, uint8_t* out, uint64_t out_cap, uint64_t* out_bytes,
                            double* kernel_ms, char* err, uint64_t err_cap, uint32_t how, uint32_t* lane_blocks);

#ifdef __cplusplus
}
#endif
#endif /* SLIMM_HIP_H */
