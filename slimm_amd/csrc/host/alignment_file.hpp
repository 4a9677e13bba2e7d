// Minimal SAM / BAM reader for the fields SLIMM's hot path consumes.
//
// Replaces the reference's use of seqan::BamFileIn (call sites: reference src/misc.hpp:498-522,
// src/slimm.hpp:194-208, 420-424): header reference names + lengths in header order (= refID), and per record
// qName, flag, refID, 0-based position and sequence length.  CIGAR, MAPQ, qualities and tags are never looked at by
// SLIMM and are skipped.  BAM = BGZF (concatenated gzip members, inflated with libdeflate when the box has it, else zlib) carrying the binary records of the
// SAM specification; SAM = the tab-separated text form.  Written against the SAM/BAM specification -- SeqAn's source
// is not part of the reference checkout -- and cross-checked in tests against files produced by an independent
// Python writer (tests/bam_io.py).
#pragma once
#include <cstdint>
#include <cstdio>
#include <memory>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../read_identity.h"

namespace slimm {

struct RecordBatch {
    std::vector<uint64_t> read_key;  // 62-bit hash of the qName's canonical base (canonical_read below)
    std::vector<int32_t> ref_id;
    std::vector<int32_t> begin_pos;
    std::vector<uint16_t> flag;      // as in the file, plus the mate bit an unflagged "N.1" / "N.2" name stands for (Q18)
    std::vector<uint32_t> l_seq;
    std::vector<std::string> qname;  // only filled when keep_names: the name as it stands in the file ...
    std::vector<uint32_t> base_len;  // ... and the length of its canonical base
    size_t size() const { return read_key.size(); }
    void clear() {
        read_key.clear();
        ref_id.clear();
        begin_pos.clear();
        flag.clear();
        l_seq.clear();
        qname.clear();
        base_len.clear();
    }
};

enum class SortOrder { Unknown, Unsorted, QueryName, Coordinate, QueryGrouped };

// 62-bit identity of a read name.  The reader makes it exact where grouped input needs it to be: a record whose name
// differs from its predecessor's never gets the predecessor's key (separate_adjacent_names).  Names colliding far
// apart in a file that is NOT grouped by name remain possible (~0.04 % odds of any such pair among 125 M reads).
uint64_t hash_read_name(const char* s, size_t n);
uint32_t check_read_name(const char* s, size_t n);

// canonical_read (../read_identity.h, quirk Q18): what the keys, check words and name compares of this reader work on.

class AlignmentFile {
public:
    AlignmentFile();
    ~AlignmentFile();
    AlignmentFile(const AlignmentFile&) = delete;
    AlignmentFile& operator=(const AlignmentFile&) = delete;

    // How the reader works, for every AlignmentFile opened afterwards (the command's --decode-threads / --no-mmap options and
    // SLIMM_TRACE=cli): threads = 0 -- as many as the process may keep busy; no_mmap -- buffered reads instead of the mapping;
    // trace -- the stage clocks on stderr when a file is closed.
    struct Settings {
        unsigned threads = 0;
        bool no_mmap = false, trace = false;
    };
    static Settings& settings();
    // false + error() on failure (reference: "Could not open <path>!", src/misc.hpp:500-504)
    bool open(const std::string& path);
    void close();
    const std::string& error() const { return err_; }
    bool is_bam() const { return bam_; }
    // "libdeflate" when libdeflate.so.0 could be dlopen()ed (and SLIMM_INFLATE is not "zlib"), else "zlib"
    static const char* inflate_backend();

    const std::vector<std::string>& ref_names() const { return ref_names_; }
    const std::vector<uint32_t>& ref_lengths() const { return ref_len_; }
    SortOrder sort_order() const { return order_; }
    // Q18 on a file grouped by QNAME (include/slimm_hip.h, "Q18 ON A GROUPED STREAM"): among the records read_batch /
    // read_into have handed out, some run of adjacent records with one canonical base holds SHORTENED names only -- their
    // flagged namesakes may lie anywhere in the file: the file must go through the any-order path
    bool q18_regroup_needed() const { return q18_short_starts_ != q18_short_to_plain_; }

    // Appends up to max_records records to `out`; returns the number appended (0 at end of file), -1 on a format error.
    long read_batch(RecordBatch& out, size_t max_records, bool keep_names = false);
    // The same for the four fields of the hot path, written straight into the caller's arrays (room for max_records).
    // check != nullptr: also a second, independent hash of every record's name (slimm_push_records_checked)
    long read_into(uint64_t* read_key, int32_t* ref_id, int32_t* begin_pos, uint16_t* flag, size_t max_records,
                   uint32_t* check = nullptr);

    // BAM only: the inflated bytes of the alignment records -- everything behind the header, or behind the last record a
    // read_* call handed out -- window by window straight into the caller's buffer (for slimm_push_bam_bytes: the records
    // are then found and decoded on the device).  Returns the bytes written (whole BGZF blocks, at most `cap`), 0 at the
    // end of the file, -1 + error() on a format error.  Not to be mixed with read_batch / read_into afterwards.
    long read_raw(uint8_t* dst, size_t cap);
    // ... or, once read_raw has reached the part of the file it reads in place (can_read_blocks()), the next whole BGZF
    // blocks as they lie in the file, COMPRESSED (for slimm_push_bgzf_blocks: the device inflates them): at most `cap`
    // bytes that inflate to at most `max_inflated`; *inflated = what they inflate to.  Returns the bytes written, 0 at the
    // end of the file, -1 + error().  Calls of the two kinds may alternate.
    long read_blocks(uint8_t* dst, size_t cap, size_t max_inflated, size_t* inflated);
    bool can_read_blocks() const { return raw_stage_ == 2 && map_ != nullptr && !eof_; }
    // SAM only: the file's text behind the header -- the alignment lines -- window by window straight into the caller's
    // buffer, cut anywhere (for slimm_push_sam_bytes: the lines are found and decoded on the device); read by pread on
    // several threads.  Returns the bytes written, 0 at the end of the file, -1 + error().  Not to be mixed with
    // read_batch / read_into afterwards.
    long read_text(uint8_t* dst, size_t cap);
    bool can_read_text() const { return !bam_ && fp_ != nullptr; }
    bool regular_file() const;   // (read_text and the mapped reads want one; a pipe or a device goes through the buffered reads)
    // after a read_raw that returned bytes: nothing will follow them (false may also mean "not known yet")
    bool raw_exhausted() const { return raw_stage_ == 2 ? eof_ : (raw_stage_ == 1 && eof_ && raw_off_ >= spare_.size()); }

private:
    size_t raw_off_ = 0;
    const uint8_t* map_ = nullptr;    // read_raw: the file, mapped (stage 2), and the next unconsumed compressed byte
    size_t map_size_ = 0, map_pos_ = 0;
    size_t sam_buf_off_ = 0, sam_body_off_ = 0, sam_line_off_ = 0, sam_text_pos_ = 0, sam_size_ = 0;   // SAM: file offset of buf_[0]; of the first alignment line; read_text's position; the file's size
    bool sam_text_started_ = false;
    size_t blk_hint_ = 0;             // read_blocks: bytes worth reading when the inflated size ends a window before the buffer does
    int raw_stage_ = 0;               // read_raw: 0 = the decoded window at hand, 1 = the prefetched one, 2 = straight from the file
    bool fill(size_t need);           // make at least `need` decoded bytes available (BAM)
    long bam_record_starts(size_t max_records, std::vector<size_t>& offs);
    template <typename F>
    void decode_parallel(size_t cnt, F decode);
    void separate_adjacent_names(uint64_t* key, const std::vector<size_t>& offs);
    std::string last_name_;   // name and key of the last record handed out (separate_adjacent_names)
    uint64_t last_key_ = 0;
    bool have_last_ = false;
    uint64_t q18_short_starts_ = 0, q18_short_to_plain_ = 0;   // runs that start shortened; shortened -> plain steps inside a run
    bool last_short_ = false;                                   // the last record handed out had a shortened name
    // record starts in buf_[pos_, end): appended to offs, at most max_records; returns false on a malformed record
    bool find_records(size_t end, size_t max_records, std::vector<size_t>& offs, size_t& new_pos);
    bool plausible_record(size_t o, size_t end, int depth) const;
    bool read_bam_header();
    bool read_sam_header();
    bool next_sam_line(std::string& line);
    void parse_hd_line(const std::string& line);

    FILE* fp_ = nullptr;
    bool bam_ = false, eof_ = false;
    std::string err_;
    std::vector<std::string> ref_names_;
    std::vector<uint32_t> ref_len_;
    SortOrder order_ = SortOrder::Unknown;
    // decoded byte window (BAM) / raw text window (SAM)
    // byte buffers whose resize() does not zero-fill: every byte is overwritten by fread / inflate right away
    template <typename T>
    struct NoInit {
        using value_type = T;
        NoInit() = default;
        template <typename U>
        NoInit(const NoInit<U>&) {}
        T* allocate(size_t n) { return static_cast<T*>(::operator new(n * sizeof(T))); }
        void deallocate(T* p, size_t) { ::operator delete(p); }
        template <typename U, typename... A>
        void construct(U* p, A&&... a) {
            if (sizeof...(A) > 0) ::new (static_cast<void*>(p)) U(std::forward<A>(a)...);
        }
        template <typename U>
        bool operator==(const NoInit<U>&) const { return true; }
        template <typename U>
        bool operator!=(const NoInit<U>&) const { return false; }
    };
    using Bytes = std::vector<uint8_t, NoInit<uint8_t>>;
    // BAM windows are double-buffered: while the caller decodes buf_, the prefetch thread reads and inflates the next
    // stretch of the file into spare_
    bool read_inflate(Bytes& dst, size_t dst_off, size_t batch_bytes, bool& at_eof, std::string& err);
    void start_prefetch();
    void stop_prefetch();
    static constexpr size_t kSlack = 4u << 20;  // room in front of a new window for the unread tail of the one before
    Bytes buf_;
    size_t pos_ = 0;
    Bytes spare_;
    std::thread prefetch_;
    bool next_ok_ = true, next_eof_ = false, file_eof_ = false, first_batch_ = true;
    std::string next_err_;
    Bytes cbuf_;
    size_t cfill_ = 0, cdone_ = 0;  // compressed bytes waiting in cbuf_ / of them consumed by the last call
    size_t cstart_ = 0;             // read_raw: where the unconsumed compressed bytes start in cbuf_ (moved to the front only now and then)
    struct Block {
        size_t coff, clen, ooff;  // compressed bytes in cbuf_, output offset in buf_
        uint32_t isize, crc;
    };
    std::vector<Block> blocks_;
    unsigned threads_ = 1;
    double ms_read_ = 0, ms_inflate_ = 0, ms_find_ = 0, ms_decode_ = 0, ms_names_ = 0, ms_wait_ = 0;  // SLIMM_TRACE=cli
    unsigned n_windows_ = 0;
    class Workers;  // the decode threads, started once per file
    std::unique_ptr<Workers> workers_, inflaters_;
    std::string pending_line_;  // first alignment line met while reading a SAM header
    bool have_pending_ = false;
    std::unordered_map<std::string, int32_t> sam_index_;  // RNAME -> refID
};

}  // namespace slimm
