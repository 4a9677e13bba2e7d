// Decoded alignment records into a context (include/slimm_hip.h: slimm_reserve, slimm_push_records* in the four-array, packed
// 16-byte and run-marked 8-byte forms, the page-locked staging sets, records borrowed from device memory): what the loop of
// analyze_alignments reads from each BamAlignmentRecord (reference src/slimm.hpp:194-211).
#include "context.h"

extern "C" {

// Room for n records of the file's form: the four-array form holds key | ref | pos | flag (| check), the packed form
// key | ref | pos, the run-marked form ref (the words) | pos.  Before the first push the form is not known yet: key, ref
// and pos are reserved and the push itself adds what its form needs beyond them.
int slimm_reserve(slimm_ctx* c, uint64_t n) {
    if (!c) return SLIMM_E_INVALID;
    if (c->device < 0) return fail(c, SLIMM_E_INVALID, "host-only context has no record stream");
    if (n >= 0x7fffffffull) return fail(c, SLIMM_E_INVALID, "a context handles fewer than 2^31 records; shard the stream");
    if (c->borrowed) return fail(c, SLIMM_E_INVALID, "records are borrowed device arrays; reset first");
    (void)hipSetDevice(c->device);
    const bool need_key = !c->marked;
    const bool need_flag = !c->packed && !c->marked && (c->n_pushed != 0 || c->in_flag.cap != 0);
    const bool need_check = c->has_check;
    const bool fits = n <= c->in_ref.cap && n <= c->in_pos.cap && (!need_key || n <= c->in_key.cap) &&
                      (!need_flag || n <= c->in_flag.cap) && (!need_check || n <= c->in_check.cap);
    if (fits) return SLIMM_OK;
    // grow, keeping what was pushed (copies on their way included)
    if (c->copy_pending) HIP_TRY(c, hipStreamSynchronize(c->copy_stream));
    uint64_t cap = n <= c->in_ref.cap ? c->in_ref.cap : std::max<uint64_t>(n, c->in_ref.cap * 2);  // (double only to grow)
    if (cap >= 0x7fffffffull) cap = 0x7ffffffeull;
    const uint64_t used = c->n_pushed;
    std::vector<void*>* later = c->bam.active ? &c->bam.outgrown : nullptr;  // (windows of a BAM file may be inflating)
    HIP_TRY(c, grow_record_array(c->in_ref, cap, used, c->stream, later));
    HIP_TRY(c, grow_record_array(c->in_pos, cap, used, c->stream, later));
    if (need_key) HIP_TRY(c, grow_record_array(c->in_key, cap, used, c->stream, later));
    if (need_flag) HIP_TRY(c, grow_record_array(c->in_flag, cap, used, c->stream, later));
    if (need_check) HIP_TRY(c, grow_record_array(c->in_check, cap, used, c->stream, later));
    return SLIMM_OK;
}

int slimm_push_records(slimm_ctx* c, const uint64_t* key, const int32_t* ref, const int32_t* pos, const uint16_t* flag,
                       uint64_t n) {
    if (!c) return SLIMM_E_INVALID;
    if (n == 0) return SLIMM_OK;
    if (!key || !ref || !pos || !flag) return fail(c, SLIMM_E_INVALID, "null record array");
    if (c->analyzed) return fail(c, SLIMM_E_INVALID, "records already analysed; reset first");
    if (c->has_check) return fail(c, SLIMM_E_INVALID, "earlier batches carried check words: push this one with slimm_push_records_checked");
    if (c->packed || c->marked)
        return fail(c, SLIMM_E_INVALID, "earlier batches were packed or run-marked records: the forms do not mix within a file");
    int rc = slimm_reserve(c, c->n_pushed + n);
    if (rc != SLIMM_OK) return rc;
    if (c->in_flag.cap < c->in_key.cap) HIP_TRY(c, c->in_flag.ensure(c->in_key.cap));  // (only ever at a file's first push)
    const uint64_t o = c->n_pushed;
    HIP_TRY(c, hipMemcpyAsync(c->in_key.p + o, key, n * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->in_ref.p + o, ref, n * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->in_pos.p + o, pos, n * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->in_flag.p + o, flag, n * 2, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));  // the caller may reuse its buffers on return
    c->n_pushed += n;
    c->rec.key = c->in_key.p;
    c->rec.ref = c->in_ref.p;
    c->rec.pos = c->in_pos.p;
    c->rec.flag = c->in_flag.p;
    c->rec.n = static_cast<uint32_t>(c->n_pushed);
    return SLIMM_OK;
}

// slimm_push_records with a check word per record: a second, independent hash of the read name.  The library compares
// keys, never names; with check words it can at least SEE when two different names share a key -- records with one key
// and two check words next to each other (grouped input) or after the sort (any order) make the run fail with
// SLIMM_E_KEY_COLLISION instead of silently becoming one read.
int slimm_push_records_checked(slimm_ctx* c, const uint64_t* key, const int32_t* ref, const int32_t* pos, const uint16_t* flag,
                               const uint32_t* check, uint64_t n) {
    if (!c) return SLIMM_E_INVALID;
    if (n == 0) return SLIMM_OK;
    if (!key || !ref || !pos || !flag || !check) return fail(c, SLIMM_E_INVALID, "null record array");
    if (c->analyzed) return fail(c, SLIMM_E_INVALID, "records already analysed; reset first");
    if (c->n_pushed && !c->has_check) return fail(c, SLIMM_E_INVALID, "earlier batches carried no check words");
    if (c->packed || c->marked)
        return fail(c, SLIMM_E_INVALID, "earlier batches were packed or run-marked records: the forms do not mix within a file");
    c->has_check = true;
    int rc = slimm_reserve(c, c->n_pushed + n);
    if (rc != SLIMM_OK) return rc;
    HIP_TRY(c, c->in_check.ensure(c->in_key.cap));
    if (c->in_flag.cap < c->in_key.cap) HIP_TRY(c, c->in_flag.ensure(c->in_key.cap));
    const uint64_t o = c->n_pushed;
    HIP_TRY(c, hipMemcpyAsync(c->in_key.p + o, key, n * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->in_ref.p + o, ref, n * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->in_pos.p + o, pos, n * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->in_flag.p + o, flag, n * 2, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->in_check.p + o, check, n * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->n_pushed += n;
    c->rec.key = c->in_key.p;
    c->rec.ref = c->in_ref.p;
    c->rec.pos = c->in_pos.p;
    c->rec.flag = c->in_flag.p;
    c->rec.check = c->in_check.p;
    c->rec.n = static_cast<uint32_t>(c->n_pushed);
    return SLIMM_OK;
}

// Streamed ingest: the copies go to a stream of their own and the call returns at once; phase A is ordered behind them
// by an event on the device, never by the host.  With page-locked arrays (the staging sets below, or the caller's own)
// the DMA engine reads them directly while the host decodes the next batch and the compute stream works on the file
// before.
int slimm_push_records_async(slimm_ctx* c, const uint64_t* key, const int32_t* ref, const int32_t* pos, const uint16_t* flag,
                             uint64_t n) {
    if (!c) return SLIMM_E_INVALID;
    if (n == 0) return SLIMM_OK;
    if (!key || !ref || !pos || !flag) return fail(c, SLIMM_E_INVALID, "null record array");
    if (c->analyzed) return fail(c, SLIMM_E_INVALID, "records already analysed; reset first");
    if (c->has_check) return fail(c, SLIMM_E_INVALID, "earlier batches carried check words: push this one with slimm_push_records_checked");
    if (c->packed || c->marked)
        return fail(c, SLIMM_E_INVALID, "earlier batches were packed or run-marked records: the forms do not mix within a file");
    int rc = slimm_reserve(c, c->n_pushed + n);
    if (rc != SLIMM_OK) return rc;
    if (c->in_flag.cap < c->in_key.cap) HIP_TRY(c, c->in_flag.ensure(c->in_key.cap));
    HIP_TRY(c, need_stream(c->copy_stream, kStreamHigh));
    const uint64_t o = c->n_pushed;
    HIP_TRY(c, hipMemcpyAsync(c->in_key.p + o, key, n * 8, hipMemcpyHostToDevice, c->copy_stream));
    HIP_TRY(c, hipMemcpyAsync(c->in_ref.p + o, ref, n * 4, hipMemcpyHostToDevice, c->copy_stream));
    HIP_TRY(c, hipMemcpyAsync(c->in_pos.p + o, pos, n * 4, hipMemcpyHostToDevice, c->copy_stream));
    HIP_TRY(c, hipMemcpyAsync(c->in_flag.p + o, flag, n * 2, hipMemcpyHostToDevice, c->copy_stream));
    HIP_TRY(c, hipEventRecord(c->copy_done, c->copy_stream));
    c->copy_pending = true;
    c->n_pushed += n;
    c->rec.key = c->in_key.p;
    c->rec.ref = c->in_ref.p;
    c->rec.pos = c->in_pos.p;
    c->rec.flag = c->in_flag.p;
    c->rec.n = static_cast<uint32_t>(c->n_pushed);
    return SLIMM_OK;
}

// 16 bytes per record: the three flag bits the record loop reads ride in the key (slimm_pack_key); no flag array crosses
// the bus or is read by the front end.  on_copy_stream: the asynchronous form (slimm_push_records_packed_async).
static int push_packed(slimm_ctx* c, const uint64_t* key, const int32_t* ref, const int32_t* pos, uint64_t n, bool on_copy_stream) {
    if (!c) return SLIMM_E_INVALID;
    if (n == 0) return SLIMM_OK;
    if (!key || !ref || !pos) return fail(c, SLIMM_E_INVALID, "null record array");
    if (c->analyzed) return fail(c, SLIMM_E_INVALID, "records already analysed; reset first");
    if (c->has_check || c->marked || (c->n_pushed && !c->packed))
        return fail(c, SLIMM_E_INVALID, "earlier batches were not packed records: the forms do not mix within a file");
    c->packed = true;
    int rc = slimm_reserve(c, c->n_pushed + n);
    if (rc != SLIMM_OK) return rc;
    const uint64_t o = c->n_pushed;
    if (on_copy_stream) HIP_TRY(c, need_stream(c->copy_stream, kStreamHigh));
    hipStream_t st = on_copy_stream ? c->copy_stream : c->stream;
    HIP_TRY(c, hipMemcpyAsync(c->in_key.p + o, key, n * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(c->in_ref.p + o, ref, n * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(c->in_pos.p + o, pos, n * 4, hipMemcpyHostToDevice, st));
    if (on_copy_stream) {
        HIP_TRY(c, hipEventRecord(c->copy_done, c->copy_stream));
        c->copy_pending = true;
    } else {
        HIP_TRY(c, hipStreamSynchronize(c->stream));  // the caller may reuse its buffers on return
    }
    c->n_pushed += n;
    c->rec.key = c->in_key.p;
    c->rec.ref = c->in_ref.p;
    c->rec.pos = c->in_pos.p;
    c->rec.flag = nullptr;
    c->rec.packed = true;
    c->rec.n = static_cast<uint32_t>(c->n_pushed);
    return SLIMM_OK;
}
int slimm_push_records_packed(slimm_ctx* c, const uint64_t* key, const int32_t* ref, const int32_t* pos, uint64_t n) {
    return push_packed(c, key, ref, pos, n, false);
}
int slimm_push_records_packed_async(slimm_ctx* c, const uint64_t* key, const int32_t* ref, const int32_t* pos, uint64_t n) {
    return push_packed(c, key, ref, pos, n, true);
}
uint64_t slimm_pack_key(uint64_t read_key, uint16_t flag) {  // src/slimm.hpp:197 (unmapped), :205-208 (mate number)
    const uint64_t mate = (flag & 0x40u) ? 1u : ((flag & 0x80u) ? 2u : 0u);
    return (read_key & ((1ull << 61) - 1ull)) | (mate << 61) | (static_cast<uint64_t>((flag & 0x4u) != 0u) << 63);
}
void slimm_pack_keys(const uint64_t* read_key, const uint16_t* flag, uint64_t n, uint64_t* packed) {
    for (uint64_t i = 0; i < n; ++i) packed[i] = slimm_pack_key(read_key[i], flag[i]);
}

// 8 bytes per record: for input grouped by read name the device never needs the names, only where a run of equal names
// starts -- the producer compares adjacent names instead of hashing them, and no key array crosses the bus or is read by
// the front end (front.hip: FrontMarked).
static int push_marked(slimm_ctx* c, const uint32_t* word, const int32_t* pos, uint64_t n, bool on_copy_stream) {
    if (!c) return SLIMM_E_INVALID;
    if (n == 0) return SLIMM_OK;
    if (!word || !pos) return fail(c, SLIMM_E_INVALID, "null record array");
    if (c->analyzed) return fail(c, SLIMM_E_INVALID, "records already analysed; reset first");
    if (c->order != SLIMM_ORDER_GROUPED)
        return fail(c, SLIMM_E_INVALID, "run-marked records carry no read identity: the context must be created for input grouped by name");
    if (c->has_check || c->packed || (c->n_pushed && !c->marked))
        return fail(c, SLIMM_E_INVALID, "earlier batches were not run-marked records: the forms do not mix within a file");
    c->marked = true;
    int rc = slimm_reserve(c, c->n_pushed + n);
    if (rc != SLIMM_OK) return rc;
    const uint64_t o = c->n_pushed;
    if (on_copy_stream) HIP_TRY(c, need_stream(c->copy_stream, kStreamHigh));
    hipStream_t st = on_copy_stream ? c->copy_stream : c->stream;
    HIP_TRY(c, hipMemcpyAsync(c->in_ref.p + o, word, n * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(c->in_pos.p + o, pos, n * 4, hipMemcpyHostToDevice, st));
    if (on_copy_stream) {
        HIP_TRY(c, hipEventRecord(c->copy_done, c->copy_stream));
        c->copy_pending = true;
    } else {
        HIP_TRY(c, hipStreamSynchronize(c->stream));  // the caller may reuse its buffers on return
    }
    c->n_pushed += n;
    c->rec = DeviceRecords();
    c->rec.ref = c->in_ref.p;
    c->rec.pos = c->in_pos.p;
    c->rec.marked = true;
    c->rec.n = static_cast<uint32_t>(c->n_pushed);
    return SLIMM_OK;
}
int slimm_push_records_marked(slimm_ctx* c, const uint32_t* word, const int32_t* pos, uint64_t n) {
    return push_marked(c, word, pos, n, false);
}
int slimm_push_records_marked_async(slimm_ctx* c, const uint32_t* word, const int32_t* pos, uint64_t n) {
    return push_marked(c, word, pos, n, true);
}
uint32_t slimm_mark_word(int32_t ref_id, uint16_t flag, int starts_run) {  // src/slimm.hpp:197 (mapped), :205-208 (mate number)
    const uint32_t mate = (flag & 0x40u) ? 1u : ((flag & 0x80u) ? 2u : 0u);
    const bool mapped = !(flag & 0x4u) && ref_id != -1;
    // (a reference that is neither -1 nor an index of the table keeps its out-of-range value: the front end reports it)
    const uint32_t r1 = mapped ? std::min<uint32_t>(static_cast<uint32_t>(ref_id) + 1u, 0x1fffffffu) : 0u;
    return r1 | (mate << 29) | (starts_run ? 0x80000000u : 0u);
}
void slimm_mark_words(const uint64_t* read_key, const uint16_t* flag, const int32_t* ref_id, uint64_t n, const uint64_t* prev_key,
                      uint32_t* word) {
    for (uint64_t i = 0; i < n; ++i) {
        const bool starts = i ? read_key[i] != read_key[i - 1] : (!prev_key || read_key[0] != *prev_key);
        word[i] = slimm_mark_word(ref_id[i], flag[i], starts ? 1 : 0);
    }
}

int slimm_push_wait(slimm_ctx* c) {
    if (!c) return SLIMM_E_INVALID;
    if (!c->copy_pending) return SLIMM_OK;
    (void)hipSetDevice(c->device);
    HIP_TRY(c, hipEventSynchronize(c->copy_done));
    c->copy_pending = false;
    for (auto& sg : c->staging) sg.pending = false;
    return SLIMM_OK;
}

int slimm_staging_buffers(slimm_ctx* c, uint32_t which, uint64_t capacity, uint64_t** key, int32_t** ref, int32_t** pos,
                          uint16_t** flag) {
    if (!c || which > 1 || !key || !ref || !pos || !flag) return SLIMM_E_INVALID;
    if (c->device < 0) return fail(c, SLIMM_E_INVALID, "host-only context has no record stream");
    (void)hipSetDevice(c->device);
    slimm_ctx::Staging& sg = c->staging[which];
    if (sg.pending) {  // the set is being read by a copy: it is the caller's again when that has finished
        HIP_TRY(c, hipEventSynchronize(sg.done));
        sg.pending = false;
    }
    HIP_TRY(c, sg.key.ensure(capacity));
    HIP_TRY(c, sg.ref.ensure(capacity));
    HIP_TRY(c, sg.pos.ensure(capacity));
    HIP_TRY(c, sg.flag.ensure(capacity));
    *key = sg.key.p;
    *ref = sg.ref.p;
    *pos = sg.pos.p;
    *flag = sg.flag.p;
    return SLIMM_OK;
}

int slimm_push_staged_async(slimm_ctx* c, uint32_t which, uint64_t n) {
    if (!c || which > 1) return SLIMM_E_INVALID;
    slimm_ctx::Staging& sg = c->staging[which];
    if (n > sg.key.cap) return fail(c, SLIMM_E_INVALID, "more records than the staging set holds");
    if (n == 0) return SLIMM_OK;
    int rc = slimm_push_records_async(c, sg.key.p, sg.ref.p, sg.pos.p, sg.flag.p, n);
    if (rc != SLIMM_OK) return rc;
    HIP_TRY(c, hipEventRecord(sg.done, c->copy_stream));
    sg.pending = true;
    return SLIMM_OK;
}

int slimm_push_staged_packed_async(slimm_ctx* c, uint32_t which, uint64_t n) {  // the set's key array holds packed keys
    if (!c || which > 1) return SLIMM_E_INVALID;
    slimm_ctx::Staging& sg = c->staging[which];
    if (n > sg.key.cap) return fail(c, SLIMM_E_INVALID, "more records than the staging set holds");
    if (n == 0) return SLIMM_OK;
    int rc = push_packed(c, sg.key.p, sg.ref.p, sg.pos.p, n, true);
    if (rc != SLIMM_OK) return rc;
    HIP_TRY(c, hipEventRecord(sg.done, c->copy_stream));
    sg.pending = true;
    return SLIMM_OK;
}

int slimm_push_staged_marked_async(slimm_ctx* c, uint32_t which, uint64_t n) {  // the set's ref array holds the words
    if (!c || which > 1) return SLIMM_E_INVALID;
    slimm_ctx::Staging& sg = c->staging[which];
    if (n > sg.key.cap) return fail(c, SLIMM_E_INVALID, "more records than the staging set holds");
    if (n == 0) return SLIMM_OK;
    int rc = push_marked(c, reinterpret_cast<const uint32_t*>(sg.ref.p), sg.pos.p, n, true);
    if (rc != SLIMM_OK) return rc;
    HIP_TRY(c, hipEventRecord(sg.done, c->copy_stream));
    sg.pending = true;
    return SLIMM_OK;
}

int slimm_staging_wait(slimm_ctx* c, uint32_t which) {
    if (!c || which > 1) return SLIMM_E_INVALID;
    slimm_ctx::Staging& sg = c->staging[which];
    if (!sg.pending) return SLIMM_OK;
    (void)hipSetDevice(c->device);
    HIP_TRY(c, hipEventSynchronize(sg.done));
    sg.pending = false;
    return SLIMM_OK;
}

int slimm_set_records_device(slimm_ctx* c, const uint64_t* key, const int32_t* ref, const int32_t* pos,
                             const uint16_t* flag, uint64_t n) {
    if (!c) return SLIMM_E_INVALID;
    if (c->device < 0) return fail(c, SLIMM_E_INVALID, "host-only context has no record stream");
    if (n >= 0x7fffffffull) return fail(c, SLIMM_E_INVALID, "a context handles fewer than 2^31 records; shard the stream");
    if (n && (!key || !ref || !pos || !flag)) return fail(c, SLIMM_E_INVALID, "null record array");
    if (c->analyzed) return fail(c, SLIMM_E_INVALID, "records already analysed; reset first");
    c->rec = DeviceRecords();  // replaces whatever was pushed or set before, in whatever form
    c->packed = c->marked = c->has_check = false;
    c->rec.key = key;
    c->rec.ref = ref;
    c->rec.pos = pos;
    c->rec.flag = flag;
    c->rec.n = static_cast<uint32_t>(n);
    c->n_pushed = n;
    c->borrowed = true;
    return SLIMM_OK;
}

int slimm_records_device(slimm_ctx* c, const uint64_t** key, const int32_t** ref, const int32_t** pos, const uint16_t** flag, uint64_t* n,
                         int* form) {
    if (!c || !key || !ref || !pos || !flag || !n || !form) return SLIMM_E_INVALID;
    if (c->device < 0) return fail(c, SLIMM_E_INVALID, "host-only context has no record stream");
    (void)hipSetDevice(c->device);
    if (c->copy_pending) HIP_TRY(c, hipStreamSynchronize(c->copy_stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *key = c->rec.key;
    *ref = c->rec.ref;
    *pos = c->rec.pos;
    *flag = c->rec.flag;
    *n = c->rec.n;
    *form = c->rec.marked ? 2 : (c->rec.packed ? 1 : 0);
    return SLIMM_OK;
}

int slimm_set_records_device_packed(slimm_ctx* c, const uint64_t* key, const int32_t* ref, const int32_t* pos, uint64_t n) {
    if (!c) return SLIMM_E_INVALID;
    if (c->device < 0) return fail(c, SLIMM_E_INVALID, "host-only context has no record stream");
    if (n >= 0x7fffffffull) return fail(c, SLIMM_E_INVALID, "a context handles fewer than 2^31 records; shard the stream");
    if (n && (!key || !ref || !pos)) return fail(c, SLIMM_E_INVALID, "null record array");
    if (c->analyzed) return fail(c, SLIMM_E_INVALID, "records already analysed; reset first");
    c->rec = DeviceRecords();
    c->rec.key = key;
    c->rec.ref = ref;
    c->rec.pos = pos;
    c->rec.packed = true;
    c->rec.n = static_cast<uint32_t>(n);
    c->n_pushed = n;
    c->packed = true;
    c->marked = c->has_check = false;
    c->borrowed = true;
    return SLIMM_OK;
}

int slimm_set_records_device_marked(slimm_ctx* c, const uint32_t* word, const int32_t* pos, uint64_t n) {
    if (!c) return SLIMM_E_INVALID;
    if (c->device < 0) return fail(c, SLIMM_E_INVALID, "host-only context has no record stream");
    if (n >= 0x7fffffffull) return fail(c, SLIMM_E_INVALID, "a context handles fewer than 2^31 records; shard the stream");
    if (n && (!word || !pos)) return fail(c, SLIMM_E_INVALID, "null record array");
    if (c->analyzed) return fail(c, SLIMM_E_INVALID, "records already analysed; reset first");
    if (c->order != SLIMM_ORDER_GROUPED)
        return fail(c, SLIMM_E_INVALID, "run-marked records carry no read identity: the context must be created for input grouped by name");
    c->rec = DeviceRecords();
    c->rec.ref = reinterpret_cast<const int32_t*>(word);
    c->rec.pos = pos;
    c->rec.marked = true;
    c->rec.n = static_cast<uint32_t>(n);
    c->n_pushed = n;
    c->marked = true;
    c->packed = c->has_check = false;
    c->borrowed = true;
    return SLIMM_OK;
}


}  // extern "C"
