#include "alignment_file.hpp"

#include <dlfcn.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <sys/mman.h>
#include <unistd.h>
#include <sys/stat.h>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>

namespace slimm {

// wyhash-style 64-bit mix of the name bytes, folded to 62 bits
// The reader's worker threads: started once per file and handed one job after the other.  A batch of a million
// records goes through four parallel stages of a few milliseconds each (inflate, record starts, decode, name check);
// starting and joining 32 threads for every one of them cost about as much as the work.
class AlignmentFile::Workers {
public:
    explicit Workers(unsigned n_threads) {
        for (unsigned t = 1; t < n_threads; ++t) pool_.emplace_back([this] { loop(); });
    }
    ~Workers() {
        {
            std::lock_guard<std::mutex> g(mu_);
            quit_ = true;
        }
        wake_.notify_all();
        for (auto& t : pool_) t.join();
    }
    unsigned size() const { return static_cast<unsigned>(pool_.size()) + 1u; }
    // fn(i) for every i in [0, count), on the workers and the calling thread; returns when all are done
    void run(unsigned count, const std::function<void(unsigned)>& fn) {
        if (count == 0) return;
        if (count == 1 || pool_.empty()) {
            for (unsigned i = 0; i < count; ++i) fn(i);
            return;
        }
        // Every job is an object of its own with its own item counter.  A worker takes its copy of the pointer under the
        // mutex and claims items only through THAT job's counter, so a worker that woke late for job k -- and is still
        // on its way out of it when job k + 1 is published -- holds a counter that is used up: it can neither run an
        // item of the new job a second time nor touch the new job's `left`.  An item is only ever claimed while its job
        // has items left, i.e. while run() of that job is still waiting, so `fn` outlives every call of it.
        auto job = std::make_shared<Job>();
        job->fn = &fn;
        job->count = count;
        job->left = count;
        {
            std::lock_guard<std::mutex> g(mu_);
            job_ = job;
            ++generation_;
        }
        wake_.notify_all();
        work(*job);
        std::unique_lock<std::mutex> g(mu_);
        done_.wait(g, [&] { return job->left == 0; });
        job_.reset();
    }

private:
    struct Job {
        const std::function<void(unsigned)>* fn = nullptr;
        unsigned count = 0;
        std::atomic<unsigned> next{0};
        unsigned left = 0;  // items not finished yet (under mu_)
    };
    void work(Job& j) {
        unsigned finished = 0;
        for (unsigned i; (i = j.next.fetch_add(1)) < j.count;) {
            (*j.fn)(i);
            ++finished;
        }
        if (finished) {
            std::lock_guard<std::mutex> g(mu_);
            j.left -= finished;
            if (j.left == 0) done_.notify_all();
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            std::shared_ptr<Job> job;
            {
                std::unique_lock<std::mutex> g(mu_);
                wake_.wait(g, [&] { return quit_ || generation_ != seen; });
                if (quit_) return;
                seen = generation_;
                job = job_;
            }
            if (job) work(*job);
        }
    }
    std::vector<std::thread> pool_;
    std::mutex mu_;
    std::condition_variable wake_, done_;
    std::shared_ptr<Job> job_;
    uint64_t generation_ = 0;
    bool quit_ = false;
};

uint64_t hash_read_name(const char* s, size_t n) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (n * 0xff51afd7ed558ccdULL);
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t w;
        memcpy(&w, s + i, 8);
        h ^= w;
        h *= 0xff51afd7ed558ccdULL;
        h ^= h >> 32;
    }
    uint64_t tail = 0;
    memcpy(&tail, s + i, n - i);
    h ^= tail;
    h *= 0xc4ceb9fe1a85ec53ULL;
    h ^= h >> 29;
    h *= 0xff51afd7ed558ccdULL;
    h ^= h >> 32;
    return h >> 2;
}

// A second, independent hash of a read name (FNV-1a folded to 32 bits): the check word of slimm_push_records_checked.
uint32_t check_read_name(const char* s, size_t n) {
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < n; ++i) {
        h ^= static_cast<unsigned char>(s[i]);
        h *= 0x100000001b3ull;
    }
    return static_cast<uint32_t>(h ^ (h >> 32));
}

AlignmentFile::AlignmentFile() = default;

namespace {
struct StageClock {  // SLIMM_TRACE=cli: where the reader's time goes, printed when the file is closed
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double& into;
    explicit StageClock(double& acc) : into(acc) {}
    ~StageClock() { into += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};
}  // namespace
AlignmentFile::~AlignmentFile() { close(); }
bool AlignmentFile::regular_file() const {
    struct stat sb;
    return fp_ && fstat(fileno(fp_), &sb) == 0 && S_ISREG(sb.st_mode);
}
AlignmentFile::Settings& AlignmentFile::settings() {
    static Settings s;
    return s;
}

void AlignmentFile::close() {
    if (fp_ && settings().trace && (ms_read_ + ms_inflate_ + ms_find_ + ms_decode_) > 0 && n_windows_ > 1)
        fprintf(stderr, "[trace] reader: read + parse blocks %.1f ms, inflate %.1f ms, record starts %.1f ms, decode + hash %.1f ms, "
                        "name check %.1f ms; %u windows, waited %.1f ms for the prefetch thread (%u threads, inflate by %s)\n", ms_read_, ms_inflate_, ms_find_,
                ms_decode_, ms_names_, n_windows_, ms_wait_, threads_, inflate_backend());
    ms_read_ = ms_inflate_ = ms_find_ = ms_decode_ = ms_names_ = ms_wait_ = 0;
    n_windows_ = 0;
    stop_prefetch();
    if (map_) munmap(const_cast<uint8_t*>(map_), map_size_);
    map_ = nullptr;
    map_size_ = map_pos_ = 0;
    if (fp_) fclose(fp_);
    fp_ = nullptr;
    workers_.reset();
    inflaters_.reset();
}

static uint32_t rd_u32(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | (static_cast<uint32_t>(p[3]) << 24); }
static uint16_t rd_u16(const uint8_t* p) { return static_cast<uint16_t>(p[0] | (p[1] << 8)); }

bool AlignmentFile::open(const std::string& path) {
    close();
    ref_names_.clear();
    ref_len_.clear();
    sam_index_.clear();
    buf_.clear();
    pos_ = 0;
    sam_buf_off_ = sam_body_off_ = sam_line_off_ = sam_text_pos_ = sam_size_ = 0;
    sam_text_started_ = false;
    eof_ = false;
    have_pending_ = false;
    have_last_ = false;
    last_short_ = false;
    q18_short_starts_ = q18_short_to_plain_ = 0;
    {
        // as many threads as the process may keep busy: the logical CPUs, or the cgroup's CPU quota when that is less (a
        // container on a 256-thread host may be held to 16 cores' worth; twice the quota keeps them fed across the
        // stages' short waits)
        const unsigned asked = settings().threads;
        unsigned hw = std::thread::hardware_concurrency();
        hw = hw ? hw : 1u;
        if (FILE* q = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char quota[32] = {0};
            unsigned long period = 0;
            if (fscanf(q, "%31s %lu", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0) {
                const unsigned long cores = (strtoul(quota, nullptr, 10) + period - 1) / period;
                if (cores > 0) hw = std::min<unsigned>(hw, static_cast<unsigned>(2 * cores));
            }
            fclose(q);
        }
        threads_ = asked ? asked : std::max(1u, std::min(hw, 64u));
    }
    workers_.reset(new Workers(threads_));
    inflaters_.reset(new Workers(threads_));  // (the prefetch thread's own: its jobs run beside the decode jobs)
    file_eof_ = false;
    first_batch_ = true;
    raw_stage_ = 0;
    cfill_ = cdone_ = cstart_ = 0;
    spare_.clear();
    order_ = SortOrder::Unknown;
    fp_ = fopen(path.c_str(), "rb");
    if (!fp_) {
        err_ = "Could not open " + path + "!";
        return false;
    }
    unsigned char magic[2] = {0, 0};
    size_t got = fread(magic, 1, 2, fp_);
    rewind(fp_);
    bam_ = (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b);
    bool ok = bam_ ? read_bam_header() : read_sam_header();
    if (!ok && err_.empty()) err_ = "bad header in " + path;
    return ok;
}

// ---- BGZF ------------------------------------------------------------------------------------------------------
// A batch of compressed blocks is read sequentially (cheap) and inflated in parallel: BGZF blocks are independent
// gzip members, which is the point of the format.  zlib inflate runs at a few hundred MB/s per core, so the decode
// side of `slimm IN.bam` scales with the host's cores up to the file read rate.
namespace {
// libdeflate (2 - 3x zlib's inflate rate on BGZF blocks) when the box has the library: there are no headers for it in
// the image, so the four entry points used are declared here (libdeflate.h: libdeflate_alloc_decompressor,
// libdeflate_deflate_decompress -- 0 = LIBDEFLATE_SUCCESS --, libdeflate_free_decompressor, libdeflate_crc32) and bound
// with dlopen("libdeflate.so.0").  zlib is the fallback when the library is absent.
struct Deflate {
    void* (*alloc)() = nullptr;
    int (*decompress)(void*, const void*, size_t, void*, size_t, size_t*) = nullptr;
    void (*release)(void*) = nullptr;
    uint32_t (*crc)(uint32_t, const void*, size_t) = nullptr;
    bool ok = false;
    Deflate() {
        void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("libdeflate.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        alloc = reinterpret_cast<void* (*)()>(dlsym(h, "libdeflate_alloc_decompressor"));
        decompress = reinterpret_cast<int (*)(void*, const void*, size_t, void*, size_t, size_t*)>(
            dlsym(h, "libdeflate_deflate_decompress"));
        release = reinterpret_cast<void (*)(void*)>(dlsym(h, "libdeflate_free_decompressor"));
        crc = reinterpret_cast<uint32_t (*)(uint32_t, const void*, size_t)>(dlsym(h, "libdeflate_crc32"));
        ok = alloc && decompress && release && crc;
    }
};
const Deflate& deflate_lib() {
    static const Deflate d;
    return d;
}
struct ThreadDecompressor {  // one per worker thread, freed with the thread
    void* d = nullptr;
    ~ThreadDecompressor() {
        if (d) deflate_lib().release(d);
    }
};

bool inflate_one(const uint8_t* src, size_t clen, uint8_t* dst, uint32_t isize, uint32_t crc) {
    // (the end-of-file block -- a fixed-code block with its end-of-block code only -- has nothing to inflate; any other payload
    // under an ISIZE of 0 must give no byte and the CRC of none, like on the device: bgzf_parse_blocks)
    if (isize == 0 && clen == 2 && src[0] == 0x03 && src[1] == 0x00 && crc == 0) return true;
    uint8_t none[8];
    if (isize == 0) dst = none;
    const Deflate& L = deflate_lib();
    if (L.ok) {
        static thread_local ThreadDecompressor td;
        if (!td.d) td.d = L.alloc();
        if (td.d) {
            size_t got = 0;
            if (L.decompress(td.d, src, clen, dst, isize, &got) != 0 || got != isize) return false;
            return L.crc(0u, dst, isize) == crc;
        }
    }
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, -15) != Z_OK) return false;
    zs.next_in = const_cast<uint8_t*>(src);
    zs.avail_in = static_cast<uInt>(clen);
    zs.next_out = dst;
    zs.avail_out = isize;
    int rc = inflate(&zs, Z_FINISH);
    inflateEnd(&zs);
    if (rc != Z_STREAM_END || zs.avail_out != 0) return false;
    return crc32(crc32(0L, Z_NULL, 0), dst, isize) == crc;
}
}  // namespace

const char* AlignmentFile::inflate_backend() { return deflate_lib().ok ? "libdeflate" : "zlib"; }

// The next stretch of the file, inflated into dst[dst_off ...): one fread of `batch_bytes` compressed bytes (what is left of
// an incomplete block at its end waits in cbuf_ for the next call), the BGZF block headers walked in memory, the blocks
// inflated on the inflate workers.  Runs on the prefetch thread while the caller works on the window before.
bool AlignmentFile::read_inflate(Bytes& dst, size_t dst_off, size_t batch_bytes, bool& at_eof, std::string& err) {
    size_t out = dst_off;
    {
        StageClock clk(ms_read_);
        if (!file_eof_) {
            cbuf_.resize(cfill_ + batch_bytes);
            const size_t got = fread(cbuf_.data() + cfill_, 1, batch_bytes, fp_);
            cfill_ += got;
            if (got < batch_bytes) file_eof_ = true;
        }
        blocks_.clear();
        size_t p = 0;
        while (cfill_ - p >= 18) {
            const uint8_t* hdr = cbuf_.data() + p;
            if (hdr[0] != 0x1f || hdr[1] != 0x8b || hdr[2] != 8 || !(hdr[3] & 4)) {
                err = "not a BGZF block";
                return false;
            }
            const size_t xlen = rd_u16(hdr + 10);
            if (cfill_ - p < 12 + xlen) break;
            int bsize = -1;
            for (size_t o = 0; o + 4 <= xlen;) {
                const uint8_t* x = hdr + 12 + o;
                const uint16_t slen = rd_u16(x + 2);
                if (x[0] == 'B' && x[1] == 'C' && slen == 2 && o + 6 <= xlen) bsize = rd_u16(x + 4);
                o += 4 + slen;
            }
            if (bsize < 0) {
                err = "BGZF block without BC field";
                return false;
            }
            const size_t total = static_cast<size_t>(bsize) + 1;
            if (total < 12u + xlen + 8u) {
                err = "bad BGZF block size";
                return false;
            }
            if (cfill_ - p < total) break;
            Block b;
            b.coff = p + 12 + xlen;
            b.clen = total - 12 - xlen - 8;  // deflate data (crc32 and isize follow)
            b.crc = rd_u32(&cbuf_[p + total - 8]);
            b.isize = rd_u32(&cbuf_[p + total - 4]);
            if (b.isize > 65536u) {  // (the format caps a block's payload at 64 KiB; nothing is allocated on a file's word)
                err = "bad BGZF block size";
                return false;
            }
            b.ooff = out;
            out += b.isize;
            blocks_.push_back(b);
            p += total;
        }
        if (file_eof_ && blocks_.empty() && cfill_ - p != 0) {
            err = cfill_ - p < 18 ? "truncated BGZF header" : "truncated BGZF block";
            return false;
        }
        cdone_ = p;
    }
    dst.resize(out);
    if (!blocks_.empty()) {
        StageClock clk(ms_inflate_);
        std::atomic<size_t> next{0};
        std::atomic<bool> ok{true};
        inflaters_->run(std::min<unsigned>(inflaters_->size(), static_cast<unsigned>(blocks_.size())), [&](unsigned) {
            for (size_t k; (k = next.fetch_add(1)) < blocks_.size();) {
                const Block& b = blocks_[k];
                if (!inflate_one(cbuf_.data() + b.coff, b.clen, dst.data() + b.ooff, b.isize, b.crc)) ok = false;
            }
        });
        if (!ok) {
            err = "corrupt BGZF block (inflate or CRC failed)";
            return false;
        }
    }
    // what is left of the last, incomplete block moves to the front
    memmove(cbuf_.data(), cbuf_.data() + cdone_, cfill_ - cdone_);
    cfill_ -= cdone_;
    at_eof = file_eof_ && cfill_ == 0;
    return true;
}

// Starts inflating the next window into spare_ (behind kSlack free bytes, where the unread tail of the current window
// will be copied when the windows change over).
void AlignmentFile::start_prefetch() {
    const size_t batch = first_batch_ ? (1u << 20) : (16u << 20);  // (the read-length sample needs one small window)
    first_batch_ = false;
    next_ok_ = true;
    next_eof_ = false;
    next_err_.clear();
    prefetch_ = std::thread([this, batch] {
        try {
            next_ok_ = read_inflate(spare_, kSlack, batch, next_eof_, next_err_);
        } catch (const std::exception& e) {  // (an exception leaving a std::thread is std::terminate)
            next_ok_ = false;
            next_err_ = std::string("reading the alignment file: ") + e.what();
        }
    });
}

void AlignmentFile::stop_prefetch() {
    if (prefetch_.joinable()) prefetch_.join();
}

bool AlignmentFile::fill(size_t need) {
    while (buf_.size() - pos_ < need) {
        if (eof_) return false;
        if (!prefetch_.joinable()) start_prefetch();  // (the first window of the file)
        {
            StageClock clk(ms_wait_);
            prefetch_.join();
        }
        ++n_windows_;
        if (!next_ok_) {
            err_ = next_err_;
            return false;
        }
        // the unread tail of the window moves in front of the new data
        const size_t tail = buf_.size() - pos_;
        if (tail <= kSlack) {
            if (tail) memcpy(spare_.data() + kSlack - tail, buf_.data() + pos_, tail);  // (buf_ is empty at the first window)
            buf_.swap(spare_);
            pos_ = kSlack - tail;
        } else {  // (a record longer than the slack)
            Bytes big;
            big.resize(tail + spare_.size() - kSlack);
            memcpy(big.data(), buf_.data() + pos_, tail);
            memcpy(big.data() + tail, spare_.data() + kSlack, spare_.size() - kSlack);
            buf_.swap(big);
            pos_ = 0;
        }
        if (next_eof_)
            eof_ = true;
        else
            start_prefetch();
    }
    return true;
}

void AlignmentFile::parse_hd_line(const std::string& line) {
    // @HD ... SO:<order> GO:<grouping>
    size_t so = line.find("\tSO:");
    if (so != std::string::npos) {
        std::string v = line.substr(so + 4, line.find_first_of("\t\r\n", so + 4) - (so + 4));
        if (v == "queryname") order_ = SortOrder::QueryName;
        else if (v == "coordinate") order_ = SortOrder::Coordinate;
        else if (v == "unsorted") order_ = SortOrder::Unsorted;
    }
    size_t go = line.find("\tGO:");
    if (go != std::string::npos) {
        std::string v = line.substr(go + 4, line.find_first_of("\t\r\n", go + 4) - (go + 4));
        if (v == "query" && order_ != SortOrder::QueryName) order_ = SortOrder::QueryGrouped;
    }
}

bool AlignmentFile::read_bam_header() {
    if (!fill(12)) return false;
    if (memcmp(&buf_[pos_], "BAM\1", 4) != 0) {
        err_ = "missing BAM magic";
        return false;
    }
    uint32_t l_text = rd_u32(&buf_[pos_ + 4]);
    pos_ += 8;
    if (!fill(l_text + 4)) return false;
    std::string text(reinterpret_cast<const char*>(&buf_[pos_]), l_text);
    pos_ += l_text;
    if (text.compare(0, 3, "@HD") == 0) parse_hd_line(text.substr(0, text.find('\n')));
    uint32_t n_ref = rd_u32(&buf_[pos_]);
    pos_ += 4;
    for (uint32_t i = 0; i < n_ref; ++i) {
        if (!fill(4)) return false;
        uint32_t l_name = rd_u32(&buf_[pos_]);
        pos_ += 4;
        if (!fill(l_name + 4)) return false;
        ref_names_.emplace_back(reinterpret_cast<const char*>(&buf_[pos_]), l_name ? l_name - 1 : 0);
        pos_ += l_name;
        ref_len_.push_back(rd_u32(&buf_[pos_]));
        pos_ += 4;
    }
    return true;
}

// ---- SAM text --------------------------------------------------------------------------------------------------
bool AlignmentFile::next_sam_line(std::string& line) {
    line.clear();
    bool started = false;
    while (true) {
        if (pos_ >= buf_.size()) {
            sam_buf_off_ += buf_.size();
            buf_.resize(1 << 20);
            size_t got = fread(buf_.data(), 1, buf_.size(), fp_);
            buf_.resize(got);
            pos_ = 0;
            if (got == 0) {
                eof_ = true;
                return !line.empty();
            }
        }
        const uint8_t* b = buf_.data() + pos_;
        if (!started) {
            sam_line_off_ = sam_buf_off_ + pos_;
            started = true;
        }
        const uint8_t* e = static_cast<const uint8_t*>(memchr(b, '\n', buf_.size() - pos_));
        if (e) {
            line.append(reinterpret_cast<const char*>(b), e - b);
            pos_ += static_cast<size_t>(e - b) + 1;
            if (!line.empty() && line.back() == '\r') line.pop_back();
            return true;
        }
        line.append(reinterpret_cast<const char*>(b), buf_.size() - pos_);
        pos_ = buf_.size();
    }
}

bool AlignmentFile::read_sam_header() {
    std::string line;
    for (;;) {
        if (!next_sam_line(line)) {
            sam_body_off_ = sam_buf_off_ + pos_;   // (no alignment line at all)
            break;
        }
        if (line.empty()) continue;
        if (line[0] != '@') {
            sam_body_off_ = sam_line_off_;   // where the alignment lines start in the file (read_text)
            pending_line_ = line;
            have_pending_ = true;
            break;
        }
        if (line.compare(0, 3, "@HD") == 0) parse_hd_line(line);
        if (line.compare(0, 3, "@SQ") == 0) {
            std::string name;
            uint32_t len = 0;
            size_t p = 3;
            while (p < line.size()) {
                size_t q = line.find('\t', p + 1);
                if (q == std::string::npos) q = line.size();
                if (line.compare(p + 1, 3, "SN:") == 0) name = line.substr(p + 4, q - p - 4);
                if (line.compare(p + 1, 3, "LN:") == 0) len = static_cast<uint32_t>(strtoul(line.c_str() + p + 4, nullptr, 10));
                p = q;
            }
            ref_names_.push_back(name);
            ref_len_.push_back(len);
        }
    }
    return true;
}

// Does a BAM record plausibly start at buf_[o]?  Used only to GUESS where the records of a chunk of the window start so
// that the chunks can be walked in parallel; every guess is verified against the walk of the chunk before it.
bool AlignmentFile::plausible_record(size_t o, size_t end, int depth) const {
    if (o + 36 > end) return false;
    const uint8_t* r = &buf_[o];
    const uint32_t bs = rd_u32(r);
    const int32_t ref = static_cast<int32_t>(rd_u32(r + 4)), pos = static_cast<int32_t>(rd_u32(r + 8));
    const uint32_t l_name = r[12], n_cigar = rd_u16(r + 16), l_seq = rd_u32(r + 20);
    const int32_t nref = static_cast<int32_t>(rd_u32(r + 24)), npos = static_cast<int32_t>(rd_u32(r + 28));
    const int32_t nrefs = static_cast<int32_t>(ref_names_.size());
    if (bs < 32 || bs > (1u << 24) || ref < -1 || ref >= nrefs || pos < -1 || l_name == 0 || l_seq > (1u << 28)) return false;
    if (nref < -1 || nref >= nrefs || npos < -1) return false;
    if (32ull + l_name + 4ull * n_cigar + (static_cast<uint64_t>(l_seq) + 1) / 2 + l_seq > bs) return false;
    if (o + 36 + l_name <= end && buf_[o + 36 + l_name - 1] != 0) return false;  // the name ends with NUL
    if (depth > 0 && o + 4 + bs + 36 <= end) return plausible_record(o + 4 + bs, end, depth - 1);
    return true;
}

// Record starts inside buf_[pos_, end).  The walk from one record to the next is a chain of dependent loads over freshly
// inflated (cache-cold) memory -- 10 M records took 0.45 s on one core, more than the 32-thread inflate.  The window is
// therefore cut into chunks walked in parallel from GUESSED first records; a chunk's result is accepted only when the
// chunk before it ends exactly at the guess, otherwise that chunk is walked again from the true start.
bool AlignmentFile::find_records(size_t end, size_t max_records, std::vector<size_t>& offs, size_t& new_pos) {
    struct Chunk {
        size_t lo = 0, hi = 0, guess = 0, stop = 0;  // [lo, hi) byte range; first record guessed; where the walk ended
        bool guessed = false, bad = false;
        std::vector<size_t> offs;
    };
    // walk records whose START is in [from, hi); stop at the first record that does not fit in the window
    auto walk = [&](size_t from, size_t hi, std::vector<size_t>& o, size_t& stop, bool& bad, size_t limit = ~size_t(0)) {
        size_t p = from;
        bad = false;
        while (p < hi && o.size() < limit) {
            if (end - p < 4) break;
            const uint32_t bs = rd_u32(&buf_[p]);
            if (bs < 32) {
                bad = true;
                break;
            }
            if (end - p < 4 + static_cast<size_t>(bs)) break;
            if (32u + buf_[p + 4 + 8] > bs) {
                bad = true;
                break;
            }
            o.push_back(p);
            p += 4 + static_cast<size_t>(bs);
        }
        stop = p;
    };
    const size_t window = end - pos_;
    // (a caller asking for a few thousand records -- the read-length sample -- gets a bounded sequential walk)
    const unsigned nchunks =
        max_records <= 65536 ? 1u : static_cast<unsigned>(std::min<size_t>(threads_, window / (512u << 10)));
    if (nchunks <= 1) {
        size_t stop;
        bool bad;
        walk(pos_, end, offs, stop, bad, max_records);
        if (bad) {
            err_ = "bad BAM record";
            return false;
        }
        if (offs.size() > max_records) {
            stop = offs[max_records];
            offs.resize(max_records);
        }
        new_pos = stop;
        return true;
    }
    std::vector<Chunk> ch(nchunks);
    const size_t per = window / nchunks;
    for (unsigned c = 0; c < nchunks; ++c) {
        ch[c].lo = pos_ + c * per;
        ch[c].hi = (c + 1 == nchunks) ? end : pos_ + (c + 1) * per;
    }
    auto work = [&](unsigned c) {
        Chunk& k = ch[c];
        size_t from = k.lo;
        if (c == 0) {
            k.guessed = true;
            k.guess = pos_;
        } else {
            const size_t limit = std::min(k.hi, k.lo + (64u << 10));  // give up after 64 KB: the chunk is walked serially
            for (size_t o = k.lo; o < limit; ++o)
                if (plausible_record(o, end, 2)) {
                    k.guessed = true;
                    k.guess = from = o;
                    break;
                }
            if (!k.guessed) return;
        }
        k.offs.reserve((k.hi - k.lo) / 128);
        walk(from, k.hi, k.offs, k.stop, k.bad);
    };
    workers_->run(nchunks, work);
    size_t cur = pos_;  // where the next record starts, according to the verified walk so far
    bool window_done = false;
    for (unsigned c = 0; c < nchunks && !window_done; ++c) {
        Chunk& k = ch[c];
        if (!(k.guessed && k.guess == cur)) {  // wrong or missing guess (or a record spans the whole chunk): walk it now
            k.offs.clear();
            if (cur < k.hi) {
                walk(cur, k.hi, k.offs, k.stop, k.bad);
            } else {
                k.stop = cur;
                k.bad = false;
            }
        }
        if (k.bad) {
            err_ = "bad BAM record";
            return false;
        }
        offs.insert(offs.end(), k.offs.begin(), k.offs.end());
        if (k.stop < k.hi) window_done = true;  // a record that does not fit in the window: nothing behind it is complete
        cur = k.stop;
    }
    if (offs.size() > max_records) {
        cur = offs[max_records];
        offs.resize(max_records);
    }
    new_pos = cur;
    return true;
}

// BAM: the starts of the next (at most max_records) records in the decoded window; 0 at a clean end of file, -1 + err_
long AlignmentFile::bam_record_starts(size_t max_records, std::vector<size_t>& offs) {
    // the window must hold at least one complete record (or the file is at its end)
    for (;;) {
        const size_t avail = buf_.size() - pos_;
        size_t need = 4;
        if (avail >= 4) {
            const uint32_t bs = rd_u32(&buf_[pos_]);
            if (bs < 32) {
                err_ = "bad BAM record size";
                return -1;
            }
            if (avail >= 4 + static_cast<size_t>(bs)) break;
            need = 4 + static_cast<size_t>(bs);
        }
        if (!fill(need)) {
            if (!err_.empty()) return -1;
            if (buf_.size() != pos_) {
                err_ = "truncated BAM record";
                return -1;
            }
            return 0;  // clean end of file
        }
    }
    // where the records start (chunks of the window in parallel)
    offs.clear();
    size_t new_pos = pos_;
    StageClock clk(ms_find_);
    if (!find_records(buf_.size(), max_records, offs, new_pos)) return -1;
    if (offs.empty()) {
        err_ = "bad BAM record";
        return -1;
    }
    pos_ = new_pos;
    return static_cast<long>(offs.size());
}

// decode(lo, hi) over [0, cnt) on the reader's threads
template <typename F>
void AlignmentFile::decode_parallel(size_t cnt, F decode) {
    const unsigned nthreads = cnt >= 65536 ? threads_ : 1u;
    if (nthreads <= 1) {
        decode(0, cnt);
        return;
    }
    const size_t per = (cnt + nthreads - 1) / nthreads;
    workers_->run(nthreads, [&](unsigned t) { decode(std::min(cnt, t * per), std::min(cnt, (t + 1) * per)); });
}

// Exact read identity for adjacent records (what grouped input needs): a record whose name differs from the record
// before it never shares its key, whatever the hash says.  Keys are 62-bit hashes of the names; two DIFFERENT names
// with one hash next to each other in a name-grouped file would be taken for one read.  Every record whose key equals
// its predecessor's has its name compared with the predecessor's (a memcmp inside a run); on a mismatch the run gets
// the next free key.  (Names that collide far apart in an unsorted file are not seen here: 2^-62 per pair.)
void AlignmentFile::separate_adjacent_names(uint64_t* key, const std::vector<size_t>& offs) {
    const size_t cnt = offs.size();
    if (!cnt) return;
    constexpr uint64_t kMask = (1ull << 62) - 1;
    auto name_of = [&](size_t k, size_t& len) {  // the canonical base (Q18)
        const uint8_t* r = &buf_[offs[k] + 4];
        len = r[8] ? r[8] - 1u : 0u;
        canonical_read(reinterpret_cast<const char*>(r + 32), len, rd_u16(r + 14));
        return reinterpret_cast<const char*>(r + 32);
    };
    auto same_name = [&](size_t a, size_t b) {
        size_t la, lb;
        const char* pa = name_of(a, la);
        const char* pb = name_of(b, lb);
        return la == lb && memcmp(pa, pb, la) == 0;
    };
    std::vector<size_t> clash;  // records that start a new name under their predecessor's key
    std::mutex mu;
    decode_parallel(cnt, [&](size_t lo, size_t hi) {
        for (size_t k = std::max<size_t>(lo, 1); k < hi; ++k)
            if (key[k] == key[k - 1] && !same_name(k, k - 1)) {
                std::lock_guard<std::mutex> g(mu);
                clash.push_back(k);
            }
    });
    // the batch's first record against the last record of the batch before
    size_t l0;
    const char* n0 = name_of(0, l0);
    const bool continues = have_last_ && last_name_.size() == l0 && memcmp(last_name_.data(), n0, l0) == 0;
    if (have_last_ && (continues ? key[0] != last_key_ : key[0] == last_key_)) clash.push_back(0);
    std::sort(clash.begin(), clash.end());
    for (size_t q = 0; q < clash.size(); ++q) {
        const size_t k = clash[q];
        uint64_t fresh;
        if (k == 0 && continues) {
            fresh = last_key_;  // the run goes on under the key it was given in the batch before
        } else {
            const uint64_t before = k ? key[k - 1] : last_key_;
            if (key[k] != before) continue;  // (an earlier move took its predecessor out of the way)
            fresh = (key[k] + 1) & kMask;
            while (fresh == before) fresh = (fresh + 1) & kMask;
        }
        size_t e = k + 1;
        while (e < cnt && same_name(e, k)) ++e;
        for (size_t j = k; j < e; ++j) key[j] = fresh;
        // the name behind the run must not run into the fresh key either
        if (e < cnt && key[e] == fresh && (q + 1 >= clash.size() || clash[q + 1] != e)) {
            clash.insert(clash.begin() + static_cast<long>(q) + 1, e);
        }
    }
    // Q18 on a grouped file: runs (equal keys = equal canonical bases by now) that start with a shortened name, and steps
    // from a shortened to an un-shortened name inside a run
    {
        auto shortened = [&](size_t k) {
            const uint8_t* r = &buf_[offs[k] + 4];
            size_t len = r[8] ? r[8] - 1u : 0u;
            const uint16_t f0 = rd_u16(r + 14);
            return canonical_read(reinterpret_cast<const char*>(r + 32), len, f0) != f0;
        };
        bool prev_short = last_short_;
        for (size_t k = 0; k < cnt; ++k) {
            const bool starts = k ? key[k] != key[k - 1] : !(have_last_ && key[0] == last_key_);
            const bool sh = shortened(k);
            q18_short_starts_ += (sh && starts) ? 1u : 0u;
            q18_short_to_plain_ += (!sh && !starts && prev_short) ? 1u : 0u;
            prev_short = sh;
        }
        last_short_ = prev_short;
    }
    size_t ll;
    const char* ln = name_of(cnt - 1, ll);
    last_name_.assign(ln, ll);
    last_key_ = key[cnt - 1];
    have_last_ = true;
}

// The inflated record bytes window by window (read_raw in the header).  The windows the header parse left behind come
// first (what is unread of the decoded window, then the window the prefetch thread was inflating); from then on whole
// BGZF blocks are inflated straight into the caller's buffer, as many as fit.
long AlignmentFile::read_blocks(uint8_t* dst, size_t cap, size_t max_inflated, size_t* inflated) {
    if (inflated) *inflated = 0;
    if (!bam_ || !dst || !inflated || raw_stage_ != 2 || !map_) {
        err_ = "read_blocks: only behind read_raw, on a mapped BAM file";
        return -1;
    }
    if (eof_) return 0;
    StageClock clk(ms_read_);
    // The file's next bytes into the caller's buffer by pread on several threads: the page cache's copy without a page
    // table entry per 4 KB (walking the block headers in the mapping and one memcpy out of it were 60 ms per GB on one
    // thread, and the device inflates a GB in 15).  The blocks are then walked in the buffer.
    size_t want = std::min(cap, map_size_ - map_pos_);
    if (blk_hint_) want = std::min(want, std::max<size_t>(blk_hint_, 1u << 20));
    {
        const int fd = fileno(fp_);
        // (as many threads as the process has cores to run them on -- the pool holds twice that --, 12 at least: on the 16-core
        // quota of the GPU box 12 / 16 / 24 / 32 threads read a 7.8 GB file in 200-460 / 180-200 / 150-200 / 160-180 ms,
        // profiles/round6/06_pread_threads.txt)
        const unsigned pread_threads = std::max(12u, inflaters_->size() / 2u);
        const unsigned nt = static_cast<unsigned>(std::max<size_t>(1, std::min<size_t>(std::min<unsigned>(inflaters_->size(), pread_threads), want >> 22)));
        const size_t per = (want + nt - 1) / nt;
        std::atomic<bool> ok{true};
        inflaters_->run(nt, [&](unsigned t) {
            size_t lo = std::min(want, t * per);
            const size_t hi = std::min(want, lo + per);
            while (lo < hi) {
                const ssize_t k = pread(fd, dst + lo, hi - lo, static_cast<off_t>(map_pos_ + lo));
                if (k <= 0) {
                    ok = false;
                    return;
                }
                lo += static_cast<size_t>(k);
            }
        });
        if (!ok) {
            err_ = "read error";
            return -1;
        }
    }
    size_t out = 0, inf = 0;
    bool by_inflated = false;
    while (want - out >= 18) {
        const uint8_t* hdr = dst + out;
        if (hdr[0] != 0x1f || hdr[1] != 0x8b || hdr[2] != 8 || !(hdr[3] & 4)) {
            err_ = "not a BGZF block";
            return -1;
        }
        const size_t xlen = rd_u16(hdr + 10);
        if (want - out < 12 + xlen) break;
        int bsize = -1;
        for (size_t o = 0; o + 4 <= xlen;) {
            const uint8_t* x = hdr + 12 + o;
            const uint16_t slen = rd_u16(x + 2);
            if (x[0] == 'B' && x[1] == 'C' && slen == 2 && o + 6 <= xlen) bsize = rd_u16(x + 4);
            o += 4 + slen;
        }
        if (bsize < 0) {
            err_ = "BGZF block without BC field";
            return -1;
        }
        const size_t total = static_cast<size_t>(bsize) + 1;
        if (total < 12u + xlen + 8u) {
            err_ = "bad BGZF block size";
            return -1;
        }
        if (want - out < total) break;
        const uint32_t isize = rd_u32(hdr + total - 4);
        if (isize > 65536u) {
            err_ = "bad BGZF block size";
            return -1;
        }
        if (inf + isize > max_inflated) {
            by_inflated = true;
            break;
        }
        out += total;
        inf += isize;
    }
    if (out == 0 && !by_inflated && map_pos_ + want == map_size_) {
        err_ = want < 18 ? "truncated BGZF header" : "truncated BGZF block";
        return -1;
    }
    if (out == 0) {
        err_ = "a BGZF block does not fit the window";
        return -1;
    }
    // (a file that inflates to the window's limit long before the buffer is full: the next call reads about what this one used)
    blk_hint_ = by_inflated ? out + (out >> 4) + (1u << 20) : 0;
    map_pos_ += out;
    ++n_windows_;
    if (map_pos_ == map_size_) eof_ = true;
    *inflated = inf;
    return static_cast<long>(out);
}

long AlignmentFile::read_text(uint8_t* dst, size_t cap) {
    if (bam_ || !fp_ || !dst || cap < (1u << 16)) {
        err_ = "read_text: a SAM file and a buffer of at least 64 KiB";
        return -1;
    }
    if (!sam_text_started_) {
        struct stat sb;
        if (fstat(fileno(fp_), &sb) != 0 || !S_ISREG(sb.st_mode)) {
            err_ = "read_text: a regular file";
            return -1;
        }
        sam_size_ = static_cast<size_t>(sb.st_size);
        sam_text_pos_ = std::min(sam_body_off_, sam_size_);
        sam_text_started_ = true;
    }
    if (sam_text_pos_ >= sam_size_) return 0;
    StageClock clk(ms_read_);
    const size_t want = std::min(cap, sam_size_ - sam_text_pos_);
    const int fd = fileno(fp_);
    const unsigned nt = static_cast<unsigned>(std::max<size_t>(1, std::min<size_t>(std::min<unsigned>(inflaters_->size(), std::max(12u, inflaters_->size() / 2u)), want >> 22)));   // (as read_blocks)
    const size_t per = (want + nt - 1) / nt;
    std::atomic<bool> ok{true};
    inflaters_->run(nt, [&](unsigned t) {
        size_t lo = std::min(want, t * per);
        const size_t hi = std::min(want, lo + per);
        while (lo < hi) {
            const ssize_t k = pread(fd, dst + lo, hi - lo, static_cast<off_t>(sam_text_pos_ + lo));
            if (k <= 0) {
                ok = false;
                return;
            }
            lo += static_cast<size_t>(k);
        }
    });
    if (!ok) {
        err_ = "read error";
        return -1;
    }
    sam_text_pos_ += want;
    ++n_windows_;
    return static_cast<long>(want);
}

long AlignmentFile::read_raw(uint8_t* dst, size_t cap) {
    if (!bam_ || !dst || cap < (1u << 20)) {
        err_ = "read_raw: a BAM file and a buffer of at least 1 MiB";
        return -1;
    }
    if (raw_stage_ == 0) {  // what is unread of the decoded window (in pieces, when it is larger than the buffer)
        const size_t n = std::min(buf_.size() - pos_, cap);
        if (n) {
            memcpy(dst, buf_.data() + pos_, n);
            pos_ += n;
            return static_cast<long>(n);
        }
        raw_stage_ = 1;
        raw_off_ = kSlack;
        if (prefetch_.joinable()) {
            {
                StageClock clk(ms_wait_);
                prefetch_.join();
            }
            ++n_windows_;
            if (!next_ok_) {
                err_ = next_err_;
                return -1;
            }
            if (next_eof_) eof_ = true;
        } else {
            spare_.resize(0);
        }
    }
    if (raw_stage_ == 1) {  // the window the prefetch thread had inflated
        const size_t left = spare_.size() > raw_off_ ? spare_.size() - raw_off_ : 0;
        const size_t n = std::min(left, cap);
        if (n) {
            memcpy(dst, spare_.data() + raw_off_, n);
            raw_off_ += n;
            return static_cast<long>(n);
        }
        spare_.resize(0);
        raw_stage_ = 2;
        // From here on the compressed bytes are read in place, out of a mapping of the file (what the buffered reads have
        // fetched but nobody has consumed yet lies cfill_ - cstart_ bytes in front of the file position): copying 1.2 GB
        // through fread was a quarter of this thread's time.  Regular files only; anything else keeps the buffered reads.
        struct stat sb;
        const long at = ftell(fp_);
        if (at >= 0 && fstat(fileno(fp_), &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0 && !settings().no_mmap) {
            void* m = mmap(nullptr, static_cast<size_t>(sb.st_size), PROT_READ, MAP_PRIVATE, fileno(fp_), 0);
            if (m != MAP_FAILED) {
                map_ = static_cast<const uint8_t*>(m);
                map_size_ = static_cast<size_t>(sb.st_size);
                map_pos_ = static_cast<size_t>(at) - (cfill_ - cstart_);
                (void)madvise(m, map_size_, MADV_SEQUENTIAL);
            }
        }
    }
    while (!eof_ && map_) {
        size_t out = 0;
        size_t p = map_pos_;
        {
            StageClock clk(ms_read_);
            blocks_.clear();
            while (map_size_ - p >= 18) {
                const uint8_t* hdr = map_ + p;
                if (hdr[0] != 0x1f || hdr[1] != 0x8b || hdr[2] != 8 || !(hdr[3] & 4)) {
                    err_ = "not a BGZF block";
                    return -1;
                }
                const size_t xlen = rd_u16(hdr + 10);
                if (map_size_ - p < 12 + xlen) break;
                int bsize = -1;
                for (size_t o = 0; o + 4 <= xlen;) {
                    const uint8_t* x = hdr + 12 + o;
                    const uint16_t slen = rd_u16(x + 2);
                    if (x[0] == 'B' && x[1] == 'C' && slen == 2 && o + 6 <= xlen) bsize = rd_u16(x + 4);
                    o += 4 + slen;
                }
                if (bsize < 0) {
                    err_ = "BGZF block without BC field";
                    return -1;
                }
                const size_t total = static_cast<size_t>(bsize) + 1;
                if (total < 12u + xlen + 8u) {
                    err_ = "bad BGZF block size";
                    return -1;
                }
                if (map_size_ - p < total) break;
                Block b;
                b.coff = p + 12 + xlen;
                b.clen = total - 12 - xlen - 8;
                b.crc = rd_u32(map_ + p + total - 8);
                b.isize = rd_u32(map_ + p + total - 4);
                if (b.isize > 65536u) {
                    err_ = "bad BGZF block size";
                    return -1;
                }
                if (out + b.isize > cap) break;
                b.ooff = out;
                out += b.isize;
                blocks_.push_back(b);
                p += total;
            }
            if (blocks_.empty() && p != map_size_) {
                err_ = map_size_ - p < 18 ? "truncated BGZF header" : "truncated BGZF block";
                return -1;
            }
        }
        if (!blocks_.empty()) {
            StageClock clk(ms_inflate_);
            std::atomic<size_t> next{0};
            std::atomic<bool> ok{true};
            // (one more job beside the blocks: the page tables of the stretch the NEXT call will parse -- the parser is
            // one thread and would take a page fault per block header otherwise)
            const size_t ahead_lo = p & ~size_t(4095), ahead_hi = std::min(map_size_, p + (24u << 20));
            std::atomic<bool> populate{true};
            inflaters_->run(std::min<unsigned>(inflaters_->size(), static_cast<unsigned>(blocks_.size()) + 1u), [&](unsigned) {
                if (populate.exchange(false) && ahead_hi > ahead_lo)
                    (void)madvise(const_cast<uint8_t*>(map_) + ahead_lo, ahead_hi - ahead_lo, 22 /* MADV_POPULATE_READ */);
                for (size_t k; (k = next.fetch_add(1)) < blocks_.size();) {
                    const Block& b = blocks_[k];
                    if (!inflate_one(map_ + b.coff, b.clen, dst + b.ooff, b.isize, b.crc)) ok = false;
                }
            });
            if (!ok) {
                err_ = "corrupt BGZF block (inflate or CRC failed)";
                return -1;
            }
        }
        map_pos_ = p;
        ++n_windows_;
        if (map_pos_ == map_size_) eof_ = true;
        if (out) return static_cast<long>(out);
    }
    while (!eof_) {
        size_t out = 0;
        {
            StageClock clk(ms_read_);
            // top the compressed bytes up (they sit in cbuf_[cstart_, cfill_) and move to the front only when the buffer's
            // tail is used up: moving 20 MB per window was a third of this thread's time), then take the whole blocks
            // that fit into the caller's buffer
            constexpr size_t kRawBatch = 32u << 20;
            if (!file_eof_ && cfill_ - cstart_ < kRawBatch / 2) {
                if (cbuf_.size() < 4 * kRawBatch) cbuf_.resize(4 * kRawBatch);
                if (cfill_ + kRawBatch > cbuf_.size()) {
                    memmove(cbuf_.data(), cbuf_.data() + cstart_, cfill_ - cstart_);
                    cfill_ -= cstart_;
                    cstart_ = 0;
                }
                const size_t got = fread(cbuf_.data() + cfill_, 1, kRawBatch, fp_);
                cfill_ += got;
                if (got < kRawBatch) file_eof_ = true;
            }
            blocks_.clear();
            size_t p = cstart_;
            while (cfill_ - p >= 18) {
                const uint8_t* hdr = cbuf_.data() + p;
                if (hdr[0] != 0x1f || hdr[1] != 0x8b || hdr[2] != 8 || !(hdr[3] & 4)) {
                    err_ = "not a BGZF block";
                    return -1;
                }
                const size_t xlen = rd_u16(hdr + 10);
                if (cfill_ - p < 12 + xlen) break;
                int bsize = -1;
                for (size_t o = 0; o + 4 <= xlen;) {
                    const uint8_t* x = hdr + 12 + o;
                    const uint16_t slen = rd_u16(x + 2);
                    if (x[0] == 'B' && x[1] == 'C' && slen == 2 && o + 6 <= xlen) bsize = rd_u16(x + 4);
                    o += 4 + slen;
                }
                if (bsize < 0) {
                    err_ = "BGZF block without BC field";
                    return -1;
                }
                const size_t total = static_cast<size_t>(bsize) + 1;
                if (total < 12u + xlen + 8u) {
                    err_ = "bad BGZF block size";
                    return -1;
                }
                if (cfill_ - p < total) break;
                Block b;
                b.coff = p + 12 + xlen;
                b.clen = total - 12 - xlen - 8;
                b.crc = rd_u32(&cbuf_[p + total - 8]);
                b.isize = rd_u32(&cbuf_[p + total - 4]);
                if (b.isize > 65536u) {
                    err_ = "bad BGZF block size";
                    return -1;
                }
                if (out + b.isize > cap) break;  // (the buffer is full: the rest waits, compressed, for the next call)
                b.ooff = out;
                out += b.isize;
                blocks_.push_back(b);
                p += total;
            }
            if (file_eof_ && blocks_.empty() && cfill_ - p != 0 && out == 0) {
                err_ = cfill_ - p < 18 ? "truncated BGZF header" : "truncated BGZF block";
                return -1;
            }
            cdone_ = p;
        }
        if (!blocks_.empty()) {
            StageClock clk(ms_inflate_);
            std::atomic<size_t> next{0};
            std::atomic<bool> ok{true};
            inflaters_->run(std::min<unsigned>(inflaters_->size(), static_cast<unsigned>(blocks_.size())), [&](unsigned) {
                for (size_t k; (k = next.fetch_add(1)) < blocks_.size();) {
                    const Block& b = blocks_[k];
                    if (!inflate_one(cbuf_.data() + b.coff, b.clen, dst + b.ooff, b.isize, b.crc)) ok = false;
                }
            });
            if (!ok) {
                err_ = "corrupt BGZF block (inflate or CRC failed)";
                return -1;
            }
        }
        const bool progressed = cdone_ != cstart_;
        cstart_ = cdone_;
        ++n_windows_;
        if (file_eof_ && cfill_ == cstart_) eof_ = true;
        if (out) return static_cast<long>(out);
        // (only empty blocks -- the end-of-file marker -- or nothing complete yet: go round again)
        if (!progressed && file_eof_) break;
    }
    return 0;
}

// The four record fields of the hot path straight into the caller's arrays (the page-locked staging sets of
// slimm_staging_buffers: the DMA engine reads what the decode threads wrote, no copy in between).
long AlignmentFile::read_into(uint64_t* read_key, int32_t* ref_id, int32_t* begin_pos, uint16_t* flag, size_t max_records,
                              uint32_t* check) {
    if (!bam_) {  // SAM text: through a batch (parsing dominates by far)
        RecordBatch b;
        const long n = read_batch(b, max_records, check != nullptr);
        for (long i = 0; i < n; ++i) {
            read_key[i] = b.read_key[i];
            ref_id[i] = b.ref_id[i];
            begin_pos[i] = b.begin_pos[i];
            flag[i] = b.flag[i];
            if (check) check[i] = check_read_name(b.qname[i].data(), b.base_len[i]);
        }
        return n;
    }
    std::vector<size_t> offs;
    const long cnt = bam_record_starts(max_records, offs);
    if (cnt <= 0) return cnt;
    std::unique_ptr<StageClock> clk(new StageClock(ms_decode_));
    decode_parallel(static_cast<size_t>(cnt), [&](size_t lo, size_t hi) {
        for (size_t k = lo; k < hi; ++k) {
            const uint8_t* r = &buf_[offs[k] + 4];
            const uint8_t l_read_name = r[8];
            const char* name = reinterpret_cast<const char*>(r + 32);
            size_t nlen = l_read_name ? l_read_name - 1u : 0u;
            flag[k] = canonical_read(name, nlen, rd_u16(r + 14));
            read_key[k] = hash_read_name(name, nlen);
            if (check) check[k] = check_read_name(name, nlen);
            ref_id[k] = static_cast<int32_t>(rd_u32(r));
            begin_pos[k] = static_cast<int32_t>(rd_u32(r + 4));
        }
    });
    clk.reset(new StageClock(ms_names_));
    separate_adjacent_names(read_key, offs);
    return cnt;
}

long AlignmentFile::read_batch(RecordBatch& out, size_t max_records, bool keep_names) {
    long n = 0;
    if (bam_) {
        std::vector<size_t> offs;
        const long got = bam_record_starts(max_records, offs);
        if (got <= 0) return got;
        // decode + hash the names (parallel)
        const size_t cnt = offs.size(), base = out.read_key.size();
        out.read_key.resize(base + cnt);
        out.ref_id.resize(base + cnt);
        out.begin_pos.resize(base + cnt);
        out.flag.resize(base + cnt);
        out.l_seq.resize(base + cnt);
        if (keep_names) {
            out.qname.resize(base + cnt);
            out.base_len.resize(base + cnt);
        }
        decode_parallel(cnt, [&](size_t lo, size_t hi) {
            for (size_t k = lo; k < hi; ++k) {
                const uint8_t* r = &buf_[offs[k] + 4];
                const uint8_t l_read_name = r[8];
                const char* name = reinterpret_cast<const char*>(r + 32);
                const size_t nlen = l_read_name ? l_read_name - 1u : 0u;
                size_t blen = nlen;
                out.flag[base + k] = canonical_read(name, blen, rd_u16(r + 14));
                out.read_key[base + k] = hash_read_name(name, blen);
                out.ref_id[base + k] = static_cast<int32_t>(rd_u32(r));
                out.begin_pos[base + k] = static_cast<int32_t>(rd_u32(r + 4));
                out.l_seq[base + k] = rd_u32(r + 16);
                if (keep_names) {
                    out.qname[base + k].assign(name, nlen);
                    out.base_len[base + k] = static_cast<uint32_t>(blen);
                }
            }
        });
        separate_adjacent_names(out.read_key.data() + base, offs);
        return static_cast<long>(cnt);
    }
    // SAM: QNAME FLAG RNAME POS MAPQ CIGAR RNEXT PNEXT TLEN SEQ QUAL ...
    if (sam_index_.empty() && !ref_names_.empty()) {
        sam_index_.reserve(ref_names_.size() * 2);
        for (size_t i = 0; i < ref_names_.size(); ++i) sam_index_.emplace(ref_names_[i], static_cast<int32_t>(i));
    }
    const std::unordered_map<std::string, int32_t>& index = sam_index_;
    std::string line;
    while (static_cast<size_t>(n) < max_records) {
        if (have_pending_) {
            line.swap(pending_line_);
            have_pending_ = false;
        } else if (!next_sam_line(line)) {
            break;
        }
        if (line.empty() || line[0] == '@') continue;
        size_t f[11];
        size_t nf = 0, p = 0;
        f[nf++] = 0;
        while (nf < 11 && (p = line.find('\t', p)) != std::string::npos) f[nf++] = ++p;
        if (nf < 10) {
            err_ = "SAM line with fewer than 10 fields";
            return -1;
        }
        auto field = [&](size_t k) { return line.substr(f[k], (k + 1 < nf ? f[k + 1] - 1 : line.size()) - f[k]); };
        std::string qn = field(0), rn = field(2), seq = field(9);
        uint16_t flag = static_cast<uint16_t>(strtoul(line.c_str() + f[1], nullptr, 10));
        long pos1 = strtol(line.c_str() + f[3], nullptr, 10);
        int32_t ref_id = -1;
        if (rn != "*") {
            auto it = index.find(rn);
            if (it != index.end()) ref_id = it->second;
        }
        size_t blen = qn.size();
        const uint16_t flag0 = flag;
        flag = canonical_read(qn.data(), blen, flag);  // Q18: key, adjacent-name compare and mate on the canonical base
        const std::string bn = qn.substr(0, blen);
        uint64_t key = hash_read_name(bn.data(), bn.size());
        const bool starts = !(have_last_ && bn == last_name_);
        if (have_last_) {  // (separate_adjacent_names, one record at a time)
            if (!starts)
                key = last_key_;
            else if (key == last_key_)
                key = (key + 1) & ((1ull << 62) - 1);
        }
        {   // (Q18 on a grouped file: as in separate_adjacent_names)
            const bool sh = flag != flag0;
            q18_short_starts_ += (sh && starts) ? 1u : 0u;
            q18_short_to_plain_ += (!sh && !starts && last_short_) ? 1u : 0u;
            last_short_ = sh;
        }
        last_name_ = bn;
        last_key_ = key;
        have_last_ = true;
        out.read_key.push_back(key);
        out.ref_id.push_back(ref_id);
        out.begin_pos.push_back(static_cast<int32_t>(pos1 - 1));  // SAM POS is 1-based; 0 ("unavailable") becomes -1
        out.flag.push_back(flag);
        out.l_seq.push_back(seq == "*" ? 0u : static_cast<uint32_t>(seq.size()));
        if (keep_names) {
            out.qname.push_back(qn);
            out.base_len.push_back(static_cast<uint32_t>(blen));
        }
        ++n;
    }
    return n;
}

}  // namespace slimm
