// `slimm [OPTIONS] DB.sldb IN` -- the reference's command line (reference src/slimm.cpp:60-204) on top of the C ABI.
//
// Same options, defaults and output files as the reference; the per-file driver follows slimm::get_profiles()
// (reference src/slimm.hpp:395-496) step for step, with the three hot phases running on the MI355X through
// include/slimm_hip.h.  Extra options (no reference counterpart): --device N, --query-grouped, --any-order,
// --dump-records (decode only, for reader tests on machines without a GPU).
#include <dirent.h>
#include <sys/stat.h>
#include <sys/mman.h>
#include <ctime>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <memory>
#include <mutex>
#include <numeric>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/slimm_hip.h"
#include <dlfcn.h>

// ---------------------------------------------------------------------------------------------------------------
// libslimm_hip.so behind its own names, loaded late.  Linking the library the ordinary way makes the dynamic loader map
// it -- and the HIP runtime under it -- before main() starts: 0.12 s in which nothing else happens.  Here a thread of its
// own dlopen()s the library (and starts the HIP runtime, slimm_warm_up) while main() parses its options, loads the
// database, samples the read length and starts decoding; the forwarders below wait for that thread the first time the
// library is needed.  Call sites stay what they would be with the library linked in (INTEGRATION.md shows that form).
// ---------------------------------------------------------------------------------------------------------------
namespace lazy {
std::mutex mu;
std::condition_variable cv;
void* handle = nullptr;
bool done = false;
std::string error;

std::string library_path() {
    char exe[4096];
    const ssize_t n = readlink("/proc/self/exe", exe, sizeof exe - 1);
    std::string dir = n > 0 ? std::string(exe, static_cast<size_t>(n)) : std::string("./slimm");
    dir = dir.substr(0, dir.find_last_of('/') + 1);
    if (const char* e = getenv("SLIMM_HIP_LIB")) return e;
    return dir + "libslimm_hip.so";
}
void load(const std::vector<int>& devices) {  // (the devices of --devices start side by side: a thread each)
    void* h = dlopen(library_path().c_str(), RTLD_NOW | RTLD_LOCAL);
    std::string err = h ? "" : dlerror();
    if (h) {
        auto warm = reinterpret_cast<int (*)(int)>(dlsym(h, "slimm_warm_up"));
        std::vector<std::thread> others;
        for (size_t i = 1; warm && i < devices.size(); ++i)
            if (std::find(devices.begin(), devices.begin() + static_cast<long>(i), devices[i]) == devices.begin() + static_cast<long>(i))
                others.emplace_back([warm, d = devices[i]] { (void)warm(d); });
        if (warm && !devices.empty()) (void)warm(devices[0]);
        for (auto& t : others) t.join();
    }
    std::lock_guard<std::mutex> g(mu);
    handle = h;
    error = err;
    done = true;
    cv.notify_all();
}
void* symbol(const char* name) {
    std::unique_lock<std::mutex> g(mu);
    cv.wait(g, [] { return done; });
    void* f = handle ? dlsym(handle, name) : nullptr;
    if (!f) {
        std::cerr << "slimm: cannot use " << library_path() << " (" << (handle ? name : error.c_str()) << ")\n";
        _exit(1);
    }
    return f;
}
}  // namespace lazy

#define SLIMM_FORWARD(ret, name, params, args)                                       \
    extern "C" ret name params {                                                     \
        static const auto f = reinterpret_cast<ret(*) params>(lazy::symbol(#name)); \
        return f args;                                                               \
    }
SLIMM_FORWARD(int, slimm_create, (const slimm_config* a, slimm_ctx** b), (a, b))
SLIMM_FORWARD(int, slimm_set_input_size_hint, (slimm_ctx* a, uint64_t b), (a, b))
SLIMM_FORWARD(int, slimm_device_memory, (slimm_ctx* a, uint64_t* b, uint64_t* c), (a, b, c))
SLIMM_FORWARD(int, slimm_window_memory, (slimm_ctx* a, uint64_t* b), (a, b))
SLIMM_FORWARD(void, slimm_destroy, (slimm_ctx* a), (a))
SLIMM_FORWARD(const char*, slimm_last_error, (const slimm_ctx* a), (a))
SLIMM_FORWARD(int, slimm_get_cutoff_cache, (slimm_ctx* a, float* b, float* c), (a, b, c))
SLIMM_FORWARD(int, slimm_set_cutoff_cache, (slimm_ctx* a, float b, float c), (a, b, c))
SLIMM_FORWARD(int, slimm_push_records,
              (slimm_ctx* a, const uint64_t* b, const int32_t* c, const int32_t* d, const uint16_t* e, uint64_t f_), (a, b, c, d, e, f_))
SLIMM_FORWARD(int, slimm_push_records_checked,
              (slimm_ctx* a, const uint64_t* b, const int32_t* c, const int32_t* d, const uint16_t* e, const uint32_t* f_, uint64_t g),
              (a, b, c, d, e, f_, g))
SLIMM_FORWARD(int, slimm_staging_buffers,
              (slimm_ctx* a, uint32_t b, uint64_t c, uint64_t** d, int32_t** e, int32_t** f_, uint16_t** g), (a, b, c, d, e, f_, g))
SLIMM_FORWARD(int, slimm_push_staged_async, (slimm_ctx* a, uint32_t b, uint64_t c), (a, b, c))
SLIMM_FORWARD(int, slimm_push_staged_packed_async, (slimm_ctx* a, uint32_t b, uint64_t c), (a, b, c))
SLIMM_FORWARD(int, slimm_push_staged_marked_async, (slimm_ctx* a, uint32_t b, uint64_t c), (a, b, c))
SLIMM_FORWARD(int, slimm_push_records_marked, (slimm_ctx* a, const uint32_t* b, const int32_t* c, uint64_t d), (a, b, c, d))
SLIMM_FORWARD(int, slimm_group_push_records_marked, (slimm_group* a, const uint32_t* b, const int32_t* c, uint64_t d), (a, b, c, d))
SLIMM_FORWARD(int, slimm_set_reference_names, (slimm_ctx * c, const char* const* names), (c, names))
SLIMM_FORWARD(int, slimm_push_sam_bytes, (slimm_ctx * c, const uint8_t* t, uint64_t n, int last, uint64_t* got), (c, t, n, last, got))
SLIMM_FORWARD(void, slimm_mark_words,
              (const uint64_t* a, const uint16_t* b, const int32_t* c, uint64_t d, const uint64_t* e, uint32_t* f_), (a, b, c, d, e, f_))
SLIMM_FORWARD(int, slimm_push_records_packed, (slimm_ctx* a, const uint64_t* b, const int32_t* c, const int32_t* d, uint64_t e),
              (a, b, c, d, e))
SLIMM_FORWARD(int, slimm_push_bam_bytes, (slimm_ctx* a, const uint8_t* b, uint64_t c, int d, uint64_t* e), (a, b, c, d, e))
SLIMM_FORWARD(int, slimm_push_bgzf_blocks, (slimm_ctx* a, const uint8_t* b, uint64_t c, uint32_t s, int d, uint64_t* e), (a, b, c, s, d, e))
SLIMM_FORWARD(int, slimm_pin_host_buffer, (slimm_ctx* a, const void* b, uint64_t c), (a, b, c))
SLIMM_FORWARD(int, slimm_shutdown, (), ())
SLIMM_FORWARD(int, slimm_reset, (slimm_ctx* a), (a))
SLIMM_FORWARD(int, slimm_check_grouping, (slimm_ctx* a, uint64_t* b), (a, b))
SLIMM_FORWARD(int, slimm_keep_bins, (slimm_ctx* a, int b), (a, b))
SLIMM_FORWARD(int, slimm_analyze_alignments, (slimm_ctx* a), (a))
SLIMM_FORWARD(int, slimm_finish_coverage, (slimm_ctx* a), (a))
SLIMM_FORWARD(int, slimm_filter_alignments, (slimm_ctx* a), (a))
SLIMM_FORWARD(int, slimm_get_reads_lca_count, (slimm_ctx* a), (a))
SLIMM_FORWARD(int, slimm_write_abundance_file, (slimm_ctx* a, const char* b), (a, b))
SLIMM_FORWARD(int, slimm_get_stats, (slimm_ctx* a, slimm_stats* b), (a, b))
SLIMM_FORWARD(int, slimm_get_ref_columns, (slimm_ctx* a, slimm_ref_columns* b), (a, b))
SLIMM_FORWARD(int, slimm_get_bins, (slimm_ctx* a, int b, uint32_t* c), (a, b, c))
SLIMM_FORWARD(int, slimm_group_create, (const slimm_config* a, const int* b, uint32_t c, slimm_group** d), (a, b, c, d))
SLIMM_FORWARD(void, slimm_group_destroy, (slimm_group* a), (a))
SLIMM_FORWARD(const char*, slimm_group_last_error, (const slimm_group* a), (a))
SLIMM_FORWARD(slimm_ctx*, slimm_group_context, (slimm_group* a, uint32_t b), (a, b))
SLIMM_FORWARD(int, slimm_group_uses_rccl, (const slimm_group* a), (a))
SLIMM_FORWARD(int, slimm_group_set_exchange, (slimm_group* a, int b), (a, b))
SLIMM_FORWARD(int, slimm_group_push_records_checked,
              (slimm_group* a, const uint64_t* b, const int32_t* c, const int32_t* d, const uint16_t* e, const uint32_t* f_, uint64_t g),
              (a, b, c, d, e, f_, g))
SLIMM_FORWARD(int, slimm_group_push_records_packed,
              (slimm_group * g, const uint64_t* k, const int32_t* r, const int32_t* p, uint64_t n), (g, k, r, p, n))
SLIMM_FORWARD(int, slimm_group_push_records,
              (slimm_group* a, const uint64_t* b, const int32_t* c, const int32_t* d, const uint16_t* e, uint64_t f_), (a, b, c, d, e, f_))
SLIMM_FORWARD(int, slimm_group_get_profiles, (slimm_group* a, const char* b), (a, b))
SLIMM_FORWARD(int, slimm_group_reset, (slimm_group* a), (a))
#include "accession.hpp"
#include "alignment_file.hpp"
#include "sldb.hpp"

namespace {

using namespace slimm;

struct Options {  // arg_options, reference src/slimm.hpp:49-87
    float cov_cut_off = 0.95f, abundance_cut_off = 0.01f;
    uint32_t bin_width = 0, min_reads = 0;
    bool verbose = false, is_directory = false, raw_output = false, coverage_output = false;
    std::string rank = "species", input_path, output_prefix, database_path;
    // extensions
    int device = 0;
    std::vector<int> devices;  // --devices a,b,...: several GPUs, one process (slimm_group_*)
    int order = -1;  // -1: from the @HD line
    bool dump_records = false;
    bool dump_raw = false;   // --dump-raw: the inflated record bytes (AlignmentFile::read_raw), for reader tests without a GPU
    // how the records reach the device (defaults: the device inflates, finds and decodes; run-marked records for grouped files)
    bool host_decode = false;      // --host-decode: the host reader decodes the records (rounds 1 - 3's path)
    bool packed_records = false;   // --packed-records: 16-byte packed records instead of run-marked ones (host decoder)
    bool verify_grouping = false;  // --verify-grouping: count the read names that come back (slimm_check_grouping) and warn
    unsigned device_inflate = 1;   // --device-inflate K: every K-th window read in place is inflated on the device (0: none)
    unsigned window_mb = 0;        // --window-mb N: bytes per window buffer (tests make windows smaller than a record)
};
bool g_trace = false;              // SLIMM_TRACE=cli (or all): millisecond marks of the stages on stderr

const char* kRankList[] = {"strains", "species", "genus", "family", "order", "class", "phylum", "superkingdom"};

// ---- reference src/file_helper.hpp:88-123 ----
std::string get_file_name(const std::string& s) { return s.substr(s.find_last_of("/\\") + 1); }
std::string get_directory(const std::string& s) { return s.substr(0, s.find_last_of("/\\")); }
std::string get_tsv_file_name(const std::string& prefix, const std::string& input) {
    std::string dir = get_directory(prefix), file = get_file_name(prefix);
    if (file.empty()) {
        file = get_file_name(input);
        auto ends = [&](const char* ext) {
            size_t p = file.find(ext);
            return p != std::string::npos && p == file.find_last_of(".");
        };
        if (ends(".sam") || ends(".bam")) file.replace(file.find_last_of("."), 4, "");
    }
    return dir + "/" + file;  // with no '/' in the prefix `dir` is the whole prefix (Q15)
}
std::string get_tsv_file_name(const std::string& prefix, const std::string& input, const std::string& suffix) {
    return get_tsv_file_name(prefix, input) + suffix + ".tsv";
}
std::vector<std::string> get_bam_files_in_directory(const std::string& directory) {  // src/file_helper.hpp:51-86
    std::vector<std::string> out;
    DIR* dir = opendir(directory.c_str());
    if (!dir) return out;
    while (dirent* ent = readdir(dir)) {
        const std::string name = ent->d_name, full = directory + "/" + name;
        if (name.empty() || name[0] == '.') continue;
        struct stat st;
        if (stat(full.c_str(), &st) == -1 || (st.st_mode & S_IFDIR)) continue;
        if (full.find(".sam") == full.find_last_of(".") || full.find(".bam") == full.find_last_of(".")) out.push_back(full);
    }
    closedir(dir);
    return out;
}

struct Lap {  // Timer<> of src/timer.hpp: whole seconds
    std::chrono::steady_clock::time_point start = std::chrono::steady_clock::now(), lap_start = start;
    long lap() {
        auto now = std::chrono::steady_clock::now();
        long s = std::chrono::duration_cast<std::chrono::seconds>(now - lap_start).count();
        lap_start = now;
        return s;
    }
    long elapsed() const {
        return std::chrono::duration_cast<std::chrono::seconds>(std::chrono::steady_clock::now() - start).count();
    }
};

void usage() {
    std::cerr << "slimm - Species Level Identification of Microbes from Metagenomes (MI355X path)\n"
                 "usage: slimm [OPTIONS] \"DB\" \"IN\"\n"
                 "  -o,  --output-prefix PREFIX   output path prefix (default: IN)\n"
                 "  -w,  --bin-width INT          width of a single bin in nucleotides (default 0 = average read length)\n"
                 "  -mr, --min-reads INT          minimum number of matching reads to consider a reference present\n"
                 "  -r,  --rank STRING            strains|species|genus|family|order|class|phylum|superkingdom (default species)\n"
                 "  -cc, --cov-cut-off DOUBLE     quantile of coverages used as cut-off, in [0, 1] (default 0.95)\n"
                 "  -ac, --abundance-cut-off DOUBLE  do not report abundances below this, in [0, 10] (default 0.01)\n"
                 "  -d,  --directory              IN is a directory of SAM/BAM files\n"
                 "  -ro, --raw-output             write raw reference statistics\n"
                 "  -co, --coverage-output        write raw coverage statistics\n"
                 "  -v,  --verbose\n"
                 "       --device N | --devices N,M,... | --query-grouped | --any-order | --dump-records | --dump-raw\n"
                 "       --host-decode | --packed-records | --verify-grouping | --device-inflate K | --window-mb N |\n"
                 "       --decode-threads N | --no-mmap     (SLIMM_TRACE=cli: stage marks on stderr)\n";
}

// 0 ok, 1 error, 2 help
int parse(int argc, char** argv, Options& o) {
    std::vector<std::string> pos;
    bool have_prefix = false;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto value = [&](std::string& dst) {
            if (i + 1 >= argc) {
                std::cerr << "slimm: option " << a << " needs a value\n";
                return false;
            }
            dst = argv[++i];
            return true;
        };
        std::string v;
        if (a == "-h" || a == "--help") {
            usage();
            return 2;
        } else if (a == "-o" || a == "--output-prefix") {
            if (!value(o.output_prefix)) return 1;
            have_prefix = true;
        } else if (a == "-w" || a == "--bin-width") {
            if (!value(v)) return 1;
            o.bin_width = static_cast<uint32_t>(strtoul(v.c_str(), nullptr, 10));
        } else if (a == "-mr" || a == "--min-reads") {
            if (!value(v)) return 1;
            o.min_reads = static_cast<uint32_t>(strtoul(v.c_str(), nullptr, 10));
        } else if (a == "-r" || a == "--rank") {
            if (!value(o.rank)) return 1;
            if (std::find_if(std::begin(kRankList), std::end(kRankList), [&](const char* r) { return o.rank == r; }) ==
                std::end(kRankList)) {
                std::cerr << "slimm: invalid rank '" << o.rank << "'\n";
                return 1;
            }
        } else if (a == "-cc" || a == "--cov-cut-off") {
            if (!value(v)) return 1;
            double d = strtod(v.c_str(), nullptr);
            if (d < 0.0 || d > 1.0) {
                std::cerr << "slimm: cov-cut-off must be in [0, 1]\n";
                return 1;
            }
            o.cov_cut_off = static_cast<float>(d);
        } else if (a == "-ac" || a == "--abundance-cut-off") {
            if (!value(v)) return 1;
            double d = strtod(v.c_str(), nullptr);
            if (d < 0.0 || d > 10.0) {
                std::cerr << "slimm: abundance-cut-off must be in [0, 10]\n";
                return 1;
            }
            o.abundance_cut_off = static_cast<float>(d);
        } else if (a == "-d" || a == "--directory") {
            o.is_directory = true;
        } else if (a == "-ro" || a == "--raw-output") {
            o.raw_output = true;
        } else if (a == "-co" || a == "--coverage-output") {
            o.coverage_output = true;
        } else if (a == "-v" || a == "--verbose") {
            o.verbose = true;
        } else if (a == "--devices") {
            if (!value(v)) return 1;
            o.devices.clear();
            for (size_t p = 0; p <= v.size();) {
                const size_t q = std::min(v.find(',', p), v.size());
                if (q > p) o.devices.push_back(atoi(v.substr(p, q - p).c_str()));
                p = q + 1;
            }
            if (!o.devices.empty()) o.device = o.devices[0];
        } else if (a == "--device") {
            if (!value(v)) return 1;
            o.device = atoi(v.c_str());
        } else if (a == "--query-grouped") {
            o.order = SLIMM_ORDER_GROUPED;
        } else if (a == "--any-order") {
            o.order = SLIMM_ORDER_ANY;
        } else if (a == "--host-decode") {
            o.host_decode = true;
        } else if (a == "--packed-records") {
            o.packed_records = true;
        } else if (a == "--verify-grouping") {
            o.verify_grouping = true;
        } else if (a == "--device-inflate") {
            if (!value(v)) return 1;
            o.device_inflate = static_cast<unsigned>(std::max(0l, atol(v.c_str())));
        } else if (a == "--window-mb") {
            if (!value(v)) return 1;
            o.window_mb = static_cast<unsigned>(std::max(0l, atol(v.c_str())));
        } else if (a == "--decode-threads") {
            if (!value(v)) return 1;
            AlignmentFile::settings().threads = static_cast<unsigned>(std::max(1l, atol(v.c_str())));
        } else if (a == "--no-mmap") {
            AlignmentFile::settings().no_mmap = true;
        } else if (a == "--dump-records") {
            o.dump_records = true;
        } else if (a == "--dump-raw") {
            o.dump_records = o.dump_raw = true;
        } else if (!a.empty() && a[0] == '-' && a.size() > 1) {
            std::cerr << "slimm: unknown option " << a << "\n";
            return 1;
        } else {
            pos.push_back(a);
        }
    }
    if (o.dump_records && pos.size() == 1) {
        o.input_path = pos[0];
        return 0;
    }
    if (pos.size() != 2) {
        usage();
        return 1;
    }
    o.database_path = pos[0];
    o.input_path = pos[1];
    if (o.database_path.size() < 5 || o.database_path.substr(o.database_path.size() - 5) != ".sldb") {
        std::cerr << "slimm: the database must be a .sldb file\n";
        return 1;
    }
    if (!have_prefix) o.output_prefix = o.input_path;  // src/slimm.cpp:175-177
    return 0;
}

// --dump-raw: what the device decoder is fed -- the inflated bytes behind the BAM header, in windows of
// --window-mb (default 1) MiB -- to stdout, the window sizes to stderr
int dump_raw(const Options& o) {
    AlignmentFile f;
    if (!f.open(o.input_path)) {
        std::cerr << f.error() << "\n";
        return 1;
    }
    const size_t cap = static_cast<size_t>(std::max(1u, o.window_mb)) << 20;
    std::vector<uint8_t> buf(cap);
    long n;
    while ((n = f.read_raw(buf.data(), cap)) > 0) {
        std::cerr << "window\t" << n << "\t" << (f.raw_exhausted() ? "last" : "more") << "\n";
        if (fwrite(buf.data(), 1, static_cast<size_t>(n), stdout) != static_cast<size_t>(n)) return 1;
    }
    if (n < 0) {
        std::cerr << f.error() << "\n";
        return 1;
    }
    return 0;
}

int dump_records(const Options& o) {
    if (o.dump_raw) return dump_raw(o);
    AlignmentFile f;
    if (!f.open(o.input_path)) {
        std::cerr << f.error() << "\n";
        return 1;
    }
    std::cout << "#format\t" << (f.is_bam() ? "BAM" : "SAM") << "\torder\t" << static_cast<int>(f.sort_order()) << "\n";
    for (size_t i = 0; i < f.ref_names().size(); ++i) std::cout << "@\t" << f.ref_names()[i] << "\t" << f.ref_lengths()[i] << "\n";
    RecordBatch b;
    long n;
    while ((n = f.read_batch(b, 1 << 20, true)) > 0) {
        for (size_t i = 0; i < b.size(); ++i)
            std::cout << b.qname[i] << "\t" << b.flag[i] << "\t" << b.ref_id[i] << "\t" << b.begin_pos[i] << "\t" << b.l_seq[i]
                      << "\t" << b.read_key[i] << "\n";
        b.clear();
    }
    if (n < 0) {
        std::cerr << f.error() << "\n";
        return 1;
    }
    std::cerr << "#q18_regroup_needed\t" << (f.q18_regroup_needed() ? 1 : 0) << "\n";   // (reader tests)
    return 0;
}

struct Session {  // what the one `slimm` object of the reference keeps across files
    Options options;
    SlimmDatabase db;
    std::vector<std::string> input_paths;
    float cc_cache = 0.0f, ucc_cache = 0.0f;  // src/slimm.hpp:155-156: never cleared by reset() (Q8)
    uint32_t total_hits = 0;
};

#define CHECK(ctx, call)                                                            \
    do {                                                                            \
        int rc_ = (call);                                                           \
        if (rc_ < 0) {                                                              \
            std::cerr << "slimm: " << #call << ": " << slimm_last_error(ctx) << "\n"; \
            if (ctx) slimm_destroy(ctx);                                            \
            return false;                                                           \
        }                                                                           \
    } while (0)

#define CHECK_KEEP(ctx, call)                                                       \
    do {                                                                            \
        int rc_ = (call);                                                           \
        if (rc_ < 0) {                                                              \
            std::cerr << "slimm: " << #call << ": " << slimm_last_error(ctx) << "\n"; \
            return false;                                                           \
        }                                                                           \
    } while (0)

float depth_of(const uint32_t* bins, uint32_t n, uint32_t nz) {  // reference_contig.hpp:188-207 + misc.hpp:285-289
    if (nz == 0) return 0.0f;
    float s = 0.0f;
    for (uint32_t i = 0; i < n; ++i) s += float(bins[i]);
    return s / n;
}

// SLIMM_TRACE=cli: millisecond marks of the per-file stages on stderr (the reference's own timer prints whole seconds)
struct Trace {
    bool on = g_trace;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void mark(const char* what) {
        if (!on) return;
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[trace] %-34s %9.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};

// SLIMM_TRACE=cli: what the file's end holds -- device memory in use (hipMemGetInfo: everything on the device), the window
// pipeline's share of it, and the process's peak resident set
void trace_memory(slimm_ctx* ctx) {
    uint64_t used = 0, total = 0, win = 0;
    (void)slimm_device_memory(ctx, &used, &total);
    (void)slimm_window_memory(ctx, &win);
    long hwm_kb = 0;
    if (FILE* f = fopen("/proc/self/status", "r")) {
        char line[256];
        while (fgets(line, sizeof line, f))
            if (sscanf(line, "VmHWM: %ld kB", &hwm_kb) == 1) break;
        fclose(f);
    }
    fprintf(stderr, "[trace] device memory in use %.2f GB of %.0f GB (window pipeline %.2f GB); host peak resident set %.2f GB\n", used / 1e9,
            total / 1e9, win / 1e9, hwm_kb / 1e6);
}

// The record stream of one file, decoded on a thread of its own from the moment the file is open: while the main thread
// builds the lineage table and creates the context (the HIP runtime's start-up included), batches pile up in host
// memory; once the context exists they are pushed in order and the decoder switches to the context's page-locked
// staging sets, which the DMA engine reads while the next batch is decoded (slimm_push_staged_async).
struct RecordPump {
    static constexpr uint64_t kBatch = 1 << 20;   // records per batch
    static constexpr size_t kMaxQueued = 64;      // batches held in host memory before the decoder waits for the context
    struct Batch {
        std::unique_ptr<uint64_t[]> key{new uint64_t[kBatch]};
        std::unique_ptr<int32_t[]> ref{new int32_t[kBatch]}, pos{new int32_t[kBatch]};
        std::unique_ptr<uint16_t[]> flag{new uint16_t[kBatch]};
        std::unique_ptr<uint32_t[]> check;  // only for streams in no particular order (want_check)
        uint64_t n = 0;
    };
    AlignmentFile& bam;
    // want_check: the stream is in no particular order, so two read names with one key would meet after the device sort
    // unnoticed -- every record then carries a second hash of its name (slimm_push_records_checked; no staging sets)
    const bool want_check;
    // Name-grouped input (no check words) goes over the bus as RUN-MARKED 8-byte records: the reader has made the keys of
    // adjacent records equal exactly when their names are, so "this record starts a qName run" is a key comparison on the
    // host and the device never sees a key (include/slimm_hip.h, slimm_mark_word).  --verify-grouping needs the keys
    // on the device and keeps the packed 16-byte form; so does --packed-records.
    const bool marked;
    uint64_t last_key = 0;   // the key of the last record marked so far (batches are pushed in file order)
    bool have_last = false;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Batch> queued;
    slimm_ctx* ctx = nullptr;  // set by attach(): from then on the decoder pushes by itself
    slimm_group* group = nullptr;  // ... or a group of contexts (--devices): the group deals the records to its members
    bool failed = false;       // a push failed (slimm_last_error says why)
    long read_rc = 0;          // the reader's last answer: 0 = end of file, -1 = format error
    double decode_ms = 0, wait_ms = 0;
    std::thread th;

    // DEVICE DECODE (BAM files, one context): the decoder thread only inflates -- windows of BGZF-inflated record bytes
    // go into a few large host buffers, a second thread hands them to slimm_push_bam_bytes, and the device finds the
    // record boundaries, reads the fields and compares / hashes the names (slimm_amd/csrc/bam_decode.hip).  The host
    // walked every inflated byte three times for that.  --host-decode keeps the host decoder.
    const bool raw;
    // bytes per window buffer (--window-mb: tests make windows smaller than a record)
    static size_t& raw_cap_setting() {
        static size_t cap = 192u << 20;
        return cap;
    }
    static size_t raw_cap() { return raw_cap_setting(); }
    static constexpr unsigned kRawBuffers = 4;
    struct RawWindow {
        unsigned which = 0;
        long n = 0;          // bytes; 0 = end of file, -1 = the reader failed
        bool last = false;   // the reader knows that nothing follows
        bool compressed = false;  // the bytes are whole BGZF blocks as they lie in the file: the device inflates them
    };
    // The windows the reader takes straight from the file are handed over COMPRESSED (slimm_push_bgzf_blocks): 192 MB of
    // BGZF blocks at a time, read by pread on several threads; the library gathers them into device windows of 1.4 - 1.9 GB
    // of inflated bytes (its inflater's first phase is a lane per block and wants tens of thousands of blocks) and inflates
    // them in two phases on two streams in turn (slimm_amd/csrc/bgzf_tokens.hip) while the next windows cross the bus.
    // --device-inflate k: every k-th of those windows only, the others inflated by the host cores (0 = all on
    // the host: rounds 1 - 3; 6 = round 4's default, when the device inflated 19 - 38 GB/s and 16 host cores 14 - 55).
    // Measured on 100 M records that compress 3-fold (scripts/realistic_cli.py): host inflate 2.1 - 2.3 s, this 0.7 s.
    unsigned device_period = 1;
    size_t device_window = 0;  // inflated bytes of a device window
    uint64_t raw_windows_device = 0, raw_windows_host = 0;
    // (mapped with MADV_HUGEPAGE where the kernel grants it: 192 MB in 4 KB pages are 49 K page faults to fill and as many
    // pages to give back when the process leaves -- a quarter second of a one-second run over the four buffers)
    struct RawUnmap {
        void operator()(uint8_t* p) const {
            if (p) munmap(p, raw_cap());
        }
    };
    std::unique_ptr<uint8_t, RawUnmap> raw_buf[kRawBuffers];
    static uint8_t* raw_map() {
        void* p = mmap(nullptr, raw_cap(), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (p == MAP_FAILED) return nullptr;
        (void)madvise(p, raw_cap(), MADV_HUGEPAGE);
        return static_cast<uint8_t*>(p);
    }
    std::deque<RawWindow> raw_ready;   // inflated, waiting to be pushed
    unsigned raw_free = kRawBuffers;
    std::thread raw_pusher;
    uint64_t raw_records = 0;
    double raw_push_ms = 0;

    RecordPump(AlignmentFile& f, bool check_words, bool device_decode, const Options& o)
        : bam(f), want_check(check_words),
          marked(!check_words && !o.verify_grouping && !o.packed_records),
          // (SAM text from anything but a regular file -- a pipe -- goes through the host decoder's buffered reads)
          raw(device_decode && !o.verify_grouping && !o.packed_records && !o.host_decode && (f.is_bam() || f.regular_file())) {
        device_period = o.device_inflate;
        device_window = std::min<size_t>(10 * raw_cap(), 1900u << 20);
        th = std::thread([this] { raw ? run_raw() : run(); });  // (in the body: every member is initialised by now)
    }
    ~RecordPump() {
        if (th.joinable() || raw_pusher.joinable()) {
            {
                std::lock_guard<std::mutex> g(mu);
                failed = true;  // (an early return of the caller: let the decoder out of its wait)
            }
            cv.notify_all();
            if (th.joinable()) th.join();
            if (raw_pusher.joinable()) raw_pusher.join();
        }
    }
    // the inflater of the device-decode mode: fills the window buffers in turn
    void run_raw() {
        for (unsigned w = 0;; w = (w + 1) % kRawBuffers) {
            {
                std::unique_lock<std::mutex> g(mu);
                cv.wait(g, [&] { return failed || raw_free > 0; });
                if (failed) return;
                --raw_free;
            }
            if (!raw_buf[w]) raw_buf[w].reset(raw_map());
            if (!raw_buf[w]) {
                std::lock_guard<std::mutex> g(mu);
                failed = true;
                cv.notify_all();
                return;
            }
            auto t1 = std::chrono::steady_clock::now();
            // (every device_period-th of the windows read in place)
            bool compressed = false;
            long n;
            if (!bam.is_bam()) {   // SAM: the text as it lies in the file (slimm_push_sam_bytes finds and decodes the lines)
                n = bam.read_text(raw_buf[w].get(), raw_cap());
                ++raw_windows_device;
            } else if (device_period && bam.can_read_blocks() && (raw_windows_device + raw_windows_host) % device_period == device_period - 1u) {
                size_t inflated = 0;
                n = bam.read_blocks(raw_buf[w].get(), raw_cap(), device_window, &inflated);
                compressed = true;
                ++raw_windows_device;
            } else {
                n = bam.read_raw(raw_buf[w].get(), raw_cap());
                if (bam.can_read_blocks() || bam.raw_exhausted()) ++raw_windows_host;
            }
            decode_ms += ms(t1, std::chrono::steady_clock::now());
            {
                std::lock_guard<std::mutex> g(mu);
                raw_ready.push_back(RawWindow{w, n, n > 0 && bam.is_bam() && bam.raw_exhausted(), compressed});
            }
            cv.notify_all();
            if (n <= 0) {
                read_rc = n;
                return;
            }
        }
    }
    // ... and the thread that hands them to the device, from the moment the context exists
    void push_raw(slimm_ctx* c) {
        bool pinned[kRawBuffers] = {};
        bool closed = false;  // a window went out as the file's last
        bool in_flight = false;  // the window pushed last is still being copied out of its buffer
        const bool text = !bam.is_bam();
        if (text) {   // SAM text names its references: the header's names for the device's look-up
            std::vector<const char*> names;
            for (const std::string& nm : bam.ref_names()) names.push_back(nm.c_str());
            if (slimm_set_reference_names(c, names.data()) != SLIMM_OK) {
                std::lock_guard<std::mutex> g(mu);
                failed = true;
                cv.notify_all();
                return;
            }
        }
        for (;;) {
            RawWindow w;
            {
                std::unique_lock<std::mutex> g(mu);
                auto t0 = std::chrono::steady_clock::now();
                cv.wait(g, [&] { return failed || !raw_ready.empty(); });
                wait_ms += ms(t0, std::chrono::steady_clock::now());
                if (failed) return;
                w = raw_ready.front();
                raw_ready.pop_front();
            }
            if (w.n < 0) return;  // (the reader failed: read_rc says so)
            int rc = SLIMM_OK;
            uint64_t got = 0;
            auto t1 = std::chrono::steady_clock::now();
            if (w.n > 0) {
                if (!pinned[w.which]) {
                    (void)slimm_pin_host_buffer(c, raw_buf[w.which].get(), raw_cap());  // (pageable memory still works)
                    pinned[w.which] = true;
                }
                rc = text ? slimm_push_sam_bytes(c, raw_buf[w.which].get(), static_cast<uint64_t>(w.n), w.last ? 1 : 0, &got)
                     : w.compressed ? slimm_push_bgzf_blocks(c, raw_buf[w.which].get(), static_cast<uint64_t>(w.n), 0u, w.last ? 1 : 0, &got)
                                    : slimm_push_bam_bytes(c, raw_buf[w.which].get(), static_cast<uint64_t>(w.n), w.last ? 1 : 0, &got);
                closed = w.last;
            } else if (!closed) {
                rc = text ? slimm_push_sam_bytes(c, nullptr, 0, 1, &got)
                          : slimm_push_bam_bytes(c, nullptr, 0, 1, &got);  // (the end came without notice: an incomplete record is an error)
            }
            raw_push_ms += ms(t1, std::chrono::steady_clock::now());
            raw_records += got;
            {
                // (a window's buffer is the library's until the NEXT push returns: its copy runs beside the work on the
                // window before it)
                std::lock_guard<std::mutex> g(mu);
                if (in_flight) ++raw_free;
                in_flight = w.n > 0 && !closed;
                if (w.n > 0 && closed) ++raw_free;
                if (w.n == 0) ++raw_free;
                if (rc < 0) failed = true;
            }
            cv.notify_all();
            if (rc < 0 || w.n == 0) return;
        }
    }
    // 16 bytes per record over the bus: the three flag bits the path reads go into the key's top bits
    // (include/slimm_hip.h, slimm_pack_key); every unchecked push of a file is packed -- the forms do not mix
    static void pack(uint64_t* key, const uint16_t* flag, uint64_t n) {
        for (uint64_t i = 0; i < n; ++i) {
            const uint64_t f = flag[i];
            const uint64_t mate = (f & 0x40u) ? 1u : ((f & 0x80u) ? 2u : 0u);
            key[i] = (key[i] & ((1ull << 61) - 1ull)) | (mate << 61) | (((f >> 2) & 1ull) << 63);
        }
    }
    // in place: ref[] becomes the words {reference + 1 | mate << 29 | starts a run << 31}
    uint32_t* mark(const uint64_t* key, const uint16_t* flag, int32_t* ref, uint64_t n) {
        uint32_t* word = reinterpret_cast<uint32_t*>(ref);
        slimm_mark_words(key, flag, ref, n, have_last ? &last_key : nullptr, word);
        last_key = key[n - 1];
        have_last = true;
        return word;
    }
    int push(slimm_ctx* c, Batch& b) {
        if (b.check) return slimm_push_records_checked(c, b.key.get(), b.ref.get(), b.pos.get(), b.flag.get(), b.check.get(), b.n);
        if (marked) return slimm_push_records_marked(c, mark(b.key.get(), b.flag.get(), b.ref.get(), b.n), b.pos.get(), b.n);
        pack(b.key.get(), b.flag.get(), b.n);
        return slimm_push_records_packed(c, b.key.get(), b.ref.get(), b.pos.get(), b.n);
    }
    int group_push(slimm_group* g, Batch& b) {
        if (b.check)
            return slimm_group_push_records_checked(g, b.key.get(), b.ref.get(), b.pos.get(), b.flag.get(), b.check.get(), b.n);
        if (marked) return slimm_group_push_records_marked(g, mark(b.key.get(), b.flag.get(), b.ref.get(), b.n), b.pos.get(), b.n);
        pack(b.key.get(), b.flag.get(), b.n);
        return slimm_group_push_records_packed(g, b.key.get(), b.ref.get(), b.pos.get(), b.n);
    }
    static double ms(std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    }
    void run() {
        uint32_t which = 0;
        for (;;) {
            slimm_ctx* c;
            {
                std::lock_guard<std::mutex> g(mu);
                if (failed) return;
                c = (group || want_check) ? nullptr : ctx;
            }
            if (c) {  // straight into a staging set
                uint64_t* key;
                int32_t *ref, *pos;
                uint16_t* flag;
                auto t0 = std::chrono::steady_clock::now();
                if (slimm_staging_buffers(c, which, kBatch, &key, &ref, &pos, &flag) < 0) break;  // (waits for the set's last copy)
                auto t1 = std::chrono::steady_clock::now();
                const long n = bam.read_into(key, ref, pos, flag, kBatch);
                auto t2 = std::chrono::steady_clock::now();
                wait_ms += ms(t0, t1);
                decode_ms += ms(t1, t2);
                if (n <= 0) {
                    read_rc = n;
                    return;
                }
                if (marked) {  // (the set's key and flag arrays stay on the host)
                    mark(key, flag, ref, static_cast<uint64_t>(n));
                    if (slimm_push_staged_marked_async(c, which, static_cast<uint64_t>(n)) < 0) break;
                } else {
                    pack(key, flag, static_cast<uint64_t>(n));  // (the set's flag array stays on the host)
                    if (slimm_push_staged_packed_async(c, which, static_cast<uint64_t>(n)) < 0) break;
                }
                which ^= 1u;
                continue;
            }
            Batch b;
            if (want_check) b.check.reset(new uint32_t[kBatch]);
            auto t1 = std::chrono::steady_clock::now();
            const long n = bam.read_into(b.key.get(), b.ref.get(), b.pos.get(), b.flag.get(), kBatch, b.check.get());
            decode_ms += ms(t1, std::chrono::steady_clock::now());
            if (n <= 0) {
                read_rc = n;
                return;
            }
            b.n = static_cast<uint64_t>(n);
            std::unique_lock<std::mutex> g(mu);
            cv.wait(g, [&] { return ctx || group || failed || queued.size() < kMaxQueued; });
            if (failed) return;
            if (group) {
                if (group_push(group, b) < 0) break;
            } else if (ctx) {  // attached meanwhile: everything queued before has been pushed, this batch follows
                if (push(ctx, b) < 0) break;
            } else {
                queued.push_back(std::move(b));
            }
        }
        std::lock_guard<std::mutex> g(mu);
        failed = true;
    }
    // pushes what was decoded so far and hands the context to the decoder; false when a push failed
    bool attach(slimm_ctx* c) {
        if (raw) {
            raw_pusher = std::thread([this, c] { push_raw(c); });
            return true;
        }
        std::unique_lock<std::mutex> g(mu);
        for (Batch& b : queued)
            if (push(c, b) < 0) {
                failed = true;
                cv.notify_all();
                return false;
            }
        queued.clear();
        ctx = c;
        cv.notify_all();
        return true;
    }
    bool attach_group(slimm_group* grp) {
        std::unique_lock<std::mutex> g(mu);
        for (Batch& b : queued)
            if (group_push(grp, b) < 0) {
                failed = true;
                cv.notify_all();
                return false;
            }
        queued.clear();
        group = grp;
        cv.notify_all();
        return true;
    }
    // waits for the end of the file; false on a failed push
    bool finish() {
        th.join();
        if (raw_pusher.joinable()) raw_pusher.join();
        return !failed;
    }
};

// write_raw_stat (src/slimm.hpp:883-943) and write_coverage (:846-881) from a context that holds the finished columns and
// coverage arrays.  global_bins: `ctx` is member 0 of a group after the bins exchange -- its arrays are the global ones,
// but its count of non-zero uniq_cov2 bins is that of its own reads: counted here from the global array instead.
bool write_raw_and_coverage(Session& S, slimm_ctx* ctx, bool global_bins, const std::string& path, const AlignmentFile& bam,
                            const std::vector<std::string>& accession, const std::vector<uint32_t>& taxa_id,
                            const std::vector<uint32_t>& lineage, Lap& watch) {
    const Options& options = S.options;
    const uint32_t R = static_cast<uint32_t>(accession.size());
    slimm_stats st;
    slimm_get_stats(ctx, &st);
    std::vector<uint32_t> reads(R), uniq(R), uniq2(R), nbins(R), nz(R), nzu(R), nzu2(R);
    std::vector<uint8_t> valid(R);
    std::vector<float> ab(R), uab(R);
    std::vector<uint32_t> cov, ucov, ucov2;
    if (options.raw_output || options.coverage_output) {
        slimm_ref_columns cols = {reads.data(), uniq.data(), uniq2.data(), nbins.data(), nz.data(),
                                  nzu.data(),   nzu2.data(), valid.data(), ab.data(),    uab.data()};
        CHECK_KEEP(ctx, slimm_get_ref_columns(ctx, &cols));
        cov.resize(st.total_bins);
        ucov.resize(st.total_bins);
        ucov2.resize(st.total_bins);
        CHECK_KEEP(ctx, slimm_get_bins(ctx, 0, cov.data()));
        CHECK_KEEP(ctx, slimm_get_bins(ctx, 1, ucov.data()));
        CHECK_KEEP(ctx, slimm_get_bins(ctx, 2, ucov2.data()));
        if (global_bins) {  // reference_contig.hpp:84-91 over the global uniq_cov2
            uint64_t off = 0;
            for (uint32_t i = 0; i < R; ++i) {
                uint32_t c = 0;
                for (uint32_t k = 0; k < nbins[i]; ++k) c += ucov2[off + k] != 0u;
                nzu2[i] = c;
                off += nbins[i];
            }
        }
    }
    auto name_of = [&](uint32_t taxid) -> std::string {
        auto it = S.db.taxid_name.find(taxid);
        return it == S.db.taxid_name.end() ? std::string() : it->second.second;
    };
    if (options.raw_output) {  // write_raw_stat :883-943
        std::cerr << "Writing features to a file ....................... ";
        std::ofstream o(get_tsv_file_name(options.output_prefix, path, "_raw"));
        o << "accesion\ttaxaid\tname\treads_count\tabundance\tuniq1_abundance\tuniq2_abundance\tgenome_length\t"
             "uniq1_reads_count\tuniq2_reads_count\tbins_count\tbins_count(>0)\tuniq1_bins_count(>0)\t"
             "uniq2_bins_count(>0)\tcoverage_depth\tuniq1_coverage_depth\tuniq2_coverage_depth\tcoverage(%)\t"
             "uniq1_coverage(%)\tuniq2_coverage(%)\n";
        uint64_t off = 0;
        for (uint32_t i = 0; i < R; ++i) {
            std::string nm = name_of(taxa_id[i]);
            if (nm.empty()) nm = "no_name_found";
            const uint32_t nb = nbins[i];
            o << accession[i] << "\t" << taxa_id[i] << "\t" << nm << "\t" << reads[i] << "\t" << ab[i] << "\t" << uab[i] << "\t"
              << 0.0f << "\t" << bam.ref_lengths()[i] << "\t" << uniq[i] << "\t" << uniq2[i] << "\t" << nb << "\t" << nz[i] << "\t"
              << nzu[i] << "\t" << nzu2[i] << "\t" << depth_of(&cov[off], nb, nz[i]) << "\t" << depth_of(&ucov[off], nb, nzu[i])
              << "\t" << depth_of(&ucov2[off], nb, nzu2[i]) << "\t" << float(nz[i]) / nb << "\t" << float(nzu[i]) / nb << "\t"
              << float(nzu2[i]) / nb << "\n";
            off += nb;
        }
        std::cerr << "[" << watch.lap() << " secs]" << std::endl;
    }
    if (options.coverage_output) {  // write_coverage :846-881
        std::cerr << "Writing coverage profiles to a file ....................... ";
        std::ofstream a(get_tsv_file_name(options.output_prefix, path, "_coverage"));
        std::ofstream b(get_tsv_file_name(options.output_prefix, path, "_uniq_coverage"));
        std::ofstream c(get_tsv_file_name(options.output_prefix, path, "_uniq_coverage2"));
        uint64_t off = 0;
        for (uint32_t i = 0; i < R; ++i) {
            const uint32_t nb = nbins[i];
            if (valid[i]) {
                a << accession[i];
                b << accession[i];
                c << accession[i];
                for (int k = 0; k < 8; ++k) {
                    std::string nm = name_of(lineage[static_cast<size_t>(i) * 8 + k]);
                    a << "," << nm;
                    b << "," << nm;
                    c << "," << nm;
                }
                for (uint32_t k = 0; k < nb; ++k) {
                    a << "," << cov[off + k];
                    b << "," << ucov[off + k];
                    c << "," << ucov2[off + k];
                }
                a << "\n";
                b << "\n";
                c << "\n";
            }
            off += nb;
        }
        std::cerr << "[" << watch.lap() << " secs]" << std::endl;
    }

    return true;
}

// slimm::get_profiles() for one file (src/slimm.hpp:395-496)
// regroup: the file was found to need the any-order path while it was read as a grouped one (Q18, below): second reading
bool get_profiles(Session& S, size_t file_index, bool regroup = false) {
    Options& options = S.options;
    const std::string path = S.input_paths[file_index];
    Lap watch;
    Trace trace;
    if (!regroup)
        std::cerr << "\nReading " << file_index + 1 << " of " << S.input_paths.size() << " files ... (" << get_file_name(path) << ")\n"
                  << "=================================================================\n";
    // Q18 on a file grouped by QNAME (include/slimm_hip.h, "Q18 ON A GROUPED STREAM"): a read named `r.1` without a mate flag
    // is the reference's read of the first-in-pair records of `r`, wherever those lie in the file (src/slimm.hpp:204-211).
    // The readers count the runs of such shortened names that stand apart from their namesakes; a file that has one is read
    // again, in any order (a fresh context of the same process: the HIP runtime and the page cache are warm).
    auto read_again_in_any_order = [&]() {
        std::cerr << "\n(read names ending in .1 / .2 without a mate flag, apart from the flagged records of the shortened name: "
                     "reading " << get_file_name(path) << " again as a file in no particular order)\n";
        return get_profiles(S, file_index, true);
    };
    AlignmentFile bam;
    if (!bam.open(path)) {  // src/misc.hpp:500-504: message, skip the file
        std::cerr << bam.error() << "\n";
        return true;
    }
    // average read length from a sample of 100k records with a sequence (src/misc.hpp:509-522)
    uint32_t avg_read_length = 0;
    {
        RecordBatch b;
        uint32_t count = 0, total = 0;
        while (count < 100000) {
            b.clear();
            long n = bam.read_batch(b, 4096);
            if (n <= 0) break;
            for (size_t i = 0; i < b.size() && count < 100000; ++i) {
                if (b.l_seq[i] == 0) continue;
                total += b.l_seq[i];
                ++count;
            }
        }
        if (count == 0) {
            std::cerr << "[ERROR] no record with a sequence in " << path << " (the reference divides by zero here)\n";
            return false;
        }
        avg_read_length = total / count;
    }
    if (options.bin_width == 0) options.bin_width = avg_read_length;  // :412-413, persists across files
    trace.mark("open + read-length sample");
    bam.close();
    if (!bam.open(path)) return true;
    // only a header that promises name grouping is trusted; anything else is sorted on the device
    const int record_order = regroup ? SLIMM_ORDER_ANY
                             : options.order >= 0
                                 ? options.order
                                 : ((bam.sort_order() == SortOrder::QueryName || bam.sort_order() == SortOrder::QueryGrouped)
                                        ? SLIMM_ORDER_GROUPED
                                        : SLIMM_ORDER_ANY);
    // (grouped streams are exact already: the reader compares the names of adjacent records)
    const bool check_words = record_order == SLIMM_ORDER_ANY;
    // decoding starts now; the records are claimed further down, when the context exists (one context: the device decodes)
    // (a group takes a GROUPED file through member 0's device decoders and deals the records device to device afterwards:
    // slimm_group_get_profiles; any other order: the host reader deals them by key)
    RecordPump pump(bam, check_words, options.devices.size() <= 1 || record_order == SLIMM_ORDER_GROUPED, options);

    std::cerr << "Intializing coverages for all reference genome ... ";
    const uint32_t R = static_cast<uint32_t>(bam.ref_names().size());
    std::vector<std::string> accession(R);
    std::vector<uint32_t> taxa_id(R, 0), lineage(static_cast<size_t>(R) * 8, 0);
    for (uint32_t i = 0; i < R; ++i) {  // :430-445
        accession[i] = get_accession_id(bam.ref_names()[i]);
        auto it = S.db.ac_taxid.find(accession[i]);
        if (it != S.db.ac_taxid.end()) {
            taxa_id[i] = it->second.empty() ? 0 : it->second[0];
            for (size_t k = 0; k < 8 && k < it->second.size(); ++k) lineage[static_cast<size_t>(i) * 8 + k] = it->second[k];
        } else {
            S.db.ac_taxid[accession[i]] = std::vector<uint32_t>(8, 0);  // Q13
        }
    }
    std::vector<uint32_t> tax_id, tax_rank;
    std::vector<const char*> tax_name;
    tax_id.reserve(S.db.taxid_name.size());
    for (auto& kv : S.db.taxid_name) {
        tax_id.push_back(kv.first);
        tax_rank.push_back(kv.second.first);
        tax_name.push_back(kv.second.second.c_str());
    }
    slimm_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.n_refs = R;
    cfg.ref_len = bam.ref_lengths().data();
    cfg.lineage = lineage.data();
    cfg.bin_width = options.bin_width;
    cfg.avg_read_len = avg_read_length;
    cfg.min_reads = options.min_reads;
    cfg.cov_cut_off = options.cov_cut_off;
    cfg.abundance_cut_off = options.abundance_cut_off;
    cfg.rank = options.rank.c_str();
    cfg.n_taxa = static_cast<uint32_t>(tax_id.size());
    cfg.tax_id = tax_id.data();
    cfg.tax_rank = tax_rank.data();
    cfg.tax_name = tax_name.data();
    cfg.device = options.device;
    cfg.record_order = record_order;
    if (options.devices.size() > 1) {
        // ---- several GPUs, one process: the group deals the records to its members by read and runs the phases with the
        // two RCCL exchanges in between (slimm_amd/csrc/group.hip); the profile comes from member 0
        slimm_group* grp = nullptr;
        if (slimm_group_create(&cfg, options.devices.data(), static_cast<uint32_t>(options.devices.size()), &grp) != SLIMM_OK) {
            std::cerr << "slimm: " << slimm_group_last_error(nullptr) << "\n";
            return false;
        }
        const bool want_arrays = options.raw_output || options.coverage_output;
        // -ro / -co read the coverage arrays (src/slimm.hpp:846-943): the members then exchange the integer bins themselves
        // (ncclAllReduce over [cov | uniq_cov], and over uniq_cov2 behind phase B) instead of their summaries
        if (want_arrays) (void)slimm_group_set_exchange(grp, SLIMM_EXCHANGE_BINS);
        for (uint32_t i = 0; i < options.devices.size(); ++i) {
            (void)slimm_set_cutoff_cache(slimm_group_context(grp, i), S.cc_cache, S.ucc_cache);
            (void)slimm_keep_bins(slimm_group_context(grp, i), want_arrays ? 1 : 0);
        }
        slimm_ctx* c0 = slimm_group_context(grp, 0);
        trace.mark("lineage table + slimm_group_create");
        std::cerr << "[" << watch.lap() << " secs]" << std::endl;
        std::cerr << "Analysing alignments on " << options.devices.size() << " devices ("
                  << (slimm_group_uses_rccl(grp) ? "RCCL" : "copy") << " collectives) ... ";
        if (pump.raw) {   // the file's size: what member 0 sizes its window buffers by
            struct stat fst;
            if (stat(path.c_str(), &fst) == 0 && S_ISREG(fst.st_mode)) (void)slimm_set_input_size_hint(c0, static_cast<uint64_t>(fst.st_size));
        }
        bool pushed = (pump.raw ? pump.attach(c0) : pump.attach_group(grp)) && pump.finish();
        long read_rc = pump.read_rc;
        if (trace.on && pump.raw)
            fprintf(stderr, "[trace] device decode on member 0: slimm_push_bam_bytes %.2f ms for %llu records, pusher waited %.2f ms for windows\n",
                    pump.raw_push_ms, static_cast<unsigned long long>(pump.raw_records), pump.wait_ms);
        if (!pushed && pump.raw && read_rc >= 0 &&
            (strstr(slimm_last_error(c0), "decode this file on the host") || strstr(slimm_last_error(c0), "fewer than 2^31 records"))) {
            // a record longer than the device decoder's carry, or more records than ONE context takes: the host reader deals them
            std::cerr << "(" << slimm_last_error(c0) << ": decoding on the host) ";
            if (slimm_group_reset(grp) != SLIMM_OK) {
                std::cerr << "slimm: " << slimm_group_last_error(grp) << "\n";
                slimm_group_destroy(grp);
                return false;
            }
            bam.close();
            if (!bam.open(path)) {
                std::cerr << bam.error() << "\n";
                slimm_group_destroy(grp);
                return false;
            }
            RecordPump again(bam, check_words, false, options);
            pushed = again.attach_group(grp) && again.finish();
            read_rc = again.read_rc;
        }
        trace.mark("rest of read + decode + push");
        if (!pushed || read_rc < 0) {
            std::cerr << (pushed ? bam.error() : std::string("pushing records: ") + (pump.raw ? slimm_last_error(c0) : slimm_group_last_error(grp))) << "\n";
            slimm_group_destroy(grp);
            return false;
        }
        if (record_order == SLIMM_ORDER_GROUPED && bam.q18_regroup_needed()) {
            slimm_group_destroy(grp);
            return read_again_in_any_order();
        }
        const int grc = slimm_group_get_profiles(grp, get_tsv_file_name(options.output_prefix, path, "_profile").c_str());
        if (grc == SLIMM_E_REGROUP) {   // (member 0's decoders counted a run of shortened names only: Q18)
            slimm_group_destroy(grp);
            return read_again_in_any_order();
        }
        if (grc < 0) {
            std::cerr << "slimm: " << slimm_group_last_error(grp) << "\n";
            slimm_group_destroy(grp);
            return false;
        }
        trace.mark("phases + exchanges + profile");
        if (trace.on) trace_memory(c0);
        std::cerr << "[" << watch.lap() << " secs]" << std::endl;
        slimm_stats st;
        slimm_get_stats(c0, &st);
        S.total_hits += st.hits_count;
        if (grc == SLIMM_E_NO_HITS) {
            std::cerr << "[WARNING] No mapped reads found in BAM file!" << std::endl;
            slimm_group_destroy(grp);
            return true;
        }
        if (options.min_reads == 0) options.min_reads = st.min_reads;
        if (options.verbose) {
            std::cerr << "  " << st.hits_count << " records processed." << std::endl;
            std::cerr << "    " << st.matches_count << " matching reads" << std::endl;
            std::cerr << "    " << st.uniq_matches_count << " uniquily matching reads" << std::endl;
            std::cerr << "  references with reads = " << st.reference_count << std::endl;
            std::cerr << "  " << st.n_valid << " passed the threshould coverage.\n";
            std::cerr << "  uniquily matching reads increased from " << st.uniq_matches_count << " to " << st.uniq_matches_count2 << "\n";
            std::cerr << std::setw(4) << st.profile_count << std::setw(15) << (options.rank) << " (" << st.profile_failed
                      << " bellow cutoff i.e. " << options.abundance_cut_off << ")\n";
        }
        if (want_arrays && !write_raw_and_coverage(S, c0, true, path, bam, accession, taxa_id, lineage, watch)) {
            slimm_group_destroy(grp);
            return false;
        }
        std::cerr << "[Done!] File took " << watch.elapsed() << " secs to process.\n";
        (void)slimm_get_cutoff_cache(c0, &S.cc_cache, &S.ucc_cache);
        slimm_group_destroy(grp);
        return true;
    }
    slimm_ctx* ctx = nullptr;
    if (slimm_create(&cfg, &ctx) != SLIMM_OK) {
        std::cerr << "slimm: " << slimm_last_error(nullptr) << "\n";
        return false;
    }
    CHECK(ctx, slimm_set_cutoff_cache(ctx, S.cc_cache, S.ucc_cache));
    {   // the file's size: what the library sizes its window buffers by (include/slimm_hip.h, slimm_set_input_size_hint)
        struct stat fst;
        if (stat(path.c_str(), &fst) == 0 && S_ISREG(fst.st_mode)) (void)slimm_set_input_size_hint(ctx, static_cast<uint64_t>(fst.st_size));
    }
    slimm_keep_bins(ctx, (S.options.raw_output || S.options.coverage_output) ? 1 : 0);  // (only -ro / -co read the arrays back)
    trace.mark("lineage table + slimm_create");
    std::cerr << "[" << watch.lap() << " secs]" << std::endl;

    std::cerr << "Analysing alignments, reads and references ....... ";
    {
        const bool pushed = pump.attach(ctx) && pump.finish();
        if (trace.on && pump.raw)
            fprintf(stderr, "[trace] device decode: inflate %.2f ms (on its own thread, from the moment the file was open), "
                            "slimm_push_bam_bytes %.2f ms for %llu records, pusher waited %.2f ms for windows; of the windows read in "
                            "place %llu were inflated on the host, %llu on the device\n",
                    pump.decode_ms, pump.raw_push_ms, static_cast<unsigned long long>(pump.raw_records), pump.wait_ms,
                    static_cast<unsigned long long>(pump.raw_windows_host), static_cast<unsigned long long>(pump.raw_windows_device));
        else if (trace.on)
            fprintf(stderr, "[trace] decode %.2f ms (on its own thread, from the moment the file was open), waiting for staging sets %.2f ms\n",
                    pump.decode_ms, pump.wait_ms);
        trace.mark("rest of read + decode + push");
        long n = pump.read_rc;
        bool ok = pushed;
        if (!pushed && pump.raw && n >= 0 && strstr(slimm_last_error(ctx), "decode this file on the host")) {
            // a record longer than the device decoder's carry (16 MiB): this file goes through the host decoder after all
            std::cerr << "(" << slimm_last_error(ctx) << ": decoding on the host) ";
            CHECK(ctx, slimm_reset(ctx));
            bam.close();
            if (!bam.open(path)) {
                std::cerr << bam.error() << "\n";
                slimm_destroy(ctx);
                return false;
            }
            RecordPump again(bam, check_words, false, options);
            ok = again.attach(ctx) && again.finish();
            n = again.read_rc;
        }
        if (!ok) {
            std::cerr << "slimm: pushing records: " << slimm_last_error(ctx) << "\n";
            slimm_destroy(ctx);
            return false;
        }
        if (n < 0) {
            std::cerr << bam.error() << "\n";
            slimm_destroy(ctx);
            return false;
        }
    }
    if (record_order == SLIMM_ORDER_GROUPED && options.verify_grouping) {
        // the header (or --query-grouped) promises that the records of a read name are adjacent; nothing checks the promise
        // unless asked: a name that comes back later would be counted as two reads (include/slimm_hip.h, slimm_check_grouping)
        uint64_t split = 0;
        CHECK(ctx, slimm_check_grouping(ctx, &split));
        if (split)
            std::cerr << "\n[WARNING] " << split << " read name run(s) repeat a name seen earlier in " << get_file_name(path)
                      << ": the file is NOT grouped by read name although it is declared so; run with --any-order\n";
    }
    {
        // (the device decoders count inside the library: SLIMM_E_REGROUP; the host decoder counts in the reader)
        const int arc = record_order == SLIMM_ORDER_GROUPED && bam.q18_regroup_needed() ? SLIMM_E_REGROUP : slimm_analyze_alignments(ctx);
        if (arc == SLIMM_E_REGROUP) {
            slimm_destroy(ctx);
            return read_again_in_any_order();
        }
        if (arc < 0) {
            std::cerr << "slimm: slimm_analyze_alignments(ctx): " << slimm_last_error(ctx) << "\n";
            slimm_destroy(ctx);
            return false;
        }
    }
    int rc = slimm_finish_coverage(ctx);
    if (rc < 0) {
        std::cerr << "slimm: " << slimm_last_error(ctx) << "\n";
        slimm_destroy(ctx);
        return false;
    }
    trace.mark("analyze_alignments + finish_coverage");
    std::cerr << "[" << watch.lap() << " secs]" << std::endl;
    slimm_stats st;
    slimm_get_stats(ctx, &st);
    S.total_hits += st.hits_count;
    if (rc == SLIMM_E_NO_HITS) {
        std::cerr << "[WARNING] No mapped reads found in BAM file!" << std::endl;
        slimm_destroy(ctx);
        return true;
    }
    if (options.min_reads == 0) options.min_reads = st.min_reads;  // :458-459, persists across files
    if (options.verbose) {                                         // print_matches_stat :621-630
        std::cerr << "  " << st.hits_count << " records processed." << std::endl;
        std::cerr << "    " << st.matches_count << " matching reads" << std::endl;
        std::cerr << "    " << st.uniq_matches_count << " uniquily matching reads" << std::endl;
        std::cerr << "  references with reads = " << st.reference_count << std::endl;
        std::cerr << "  expected bins coverage = " << st.expected_coverage << std::endl;
        std::cerr << "  bins coverage cut-off = " << st.coverage_cut_off << " (" << options.cov_cut_off << " quantile)\n";
        std::cerr << "  uniq bins coverage cut-off = " << st.uniq_coverage_cut_off << " (" << options.cov_cut_off << " quantile)\n\n";
    }

    std::cerr << "Filtering unlikely sequences ..................... ";
    CHECK(ctx, slimm_filter_alignments(ctx));
    std::cerr << "[" << watch.lap() << " secs]" << std::endl;
    slimm_get_stats(ctx, &st);
    if (options.verbose) {  // print_filter_stat :613-619
        std::cerr << "  " << st.n_valid << " passed the threshould coverage.\n";
        std::cerr << "  " << st.failed_by_cov << " ref's couldn't pass the coverage threshould.\n";
        std::cerr << "  " << st.failed_by_uniq_cov << " ref's couldn't pass the uniq coverage threshould.\n";
        std::cerr << "  uniquily matching reads increased from " << st.uniq_matches_count << " to " << st.uniq_matches_count2 << "\n\n";
    }

    if (options.raw_output || options.coverage_output) {
        if (!write_raw_and_coverage(S, ctx, false, path, bam, accession, taxa_id, lineage, watch)) {
            slimm_destroy(ctx);
            return false;
        }
    }
    std::cerr << "Assigning reads to Least Common Ancestor (LCA) ... ";
    CHECK(ctx, slimm_get_reads_lca_count(ctx));
    std::cerr << "[" << watch.lap() << " secs]" << std::endl;

    std::cerr << "Writing taxnomic profile(s) ...................... ";
    CHECK(ctx, slimm_write_abundance_file(ctx, get_tsv_file_name(options.output_prefix, path, "_profile").c_str()));
    if (options.verbose) {
        slimm_get_stats(ctx, &st);
        std::cerr << "\n" << std::setw(4) << st.profile_count << std::setw(15) << (options.rank) << " (" << st.profile_failed
                  << " bellow cutoff i.e. " << options.abundance_cut_off << ")";
        std::cerr << "\n.................................................. ";
    }
    std::cerr << "[" << watch.lap() << " secs]" << std::endl;
    trace.mark("filter + LCA + outputs");
    if (trace.on) trace_memory(ctx);
    std::cerr << "[Done!] File took " << watch.elapsed() << " secs to process.\n";
    CHECK(ctx, slimm_get_cutoff_cache(ctx, &S.cc_cache, &S.ucc_cache));
    slimm_destroy(ctx);
    trace.mark("slimm_destroy");
    return true;
}

}  // namespace

int main(int argc, char** argv) {
    {   // SLIMM_TRACE=cli | all | 1 (a comma list; host / push are the library's: slimm_amd/csrc/force.h)
        const char* e = getenv("SLIMM_TRACE");
        for (const char* p = e; p && *p;) {
            const char* end = strchr(p, ',');
            const size_t len = end ? static_cast<size_t>(end - p) : strlen(p);
            if ((len == 3 && (!memcmp(p, "cli", 3) || !memcmp(p, "all", 3))) || (len == 1 && *p == '1')) g_trace = true;
            if (!end) break;
            p = end + 1;
        }
        AlignmentFile::settings().trace = g_trace;
    }
    if (g_trace) {
        struct timespec now;
        clock_gettime(CLOCK_REALTIME, &now);
        fprintf(stderr, "[trace] main() entered at %.6f (epoch seconds)\n", now.tv_sec + now.tv_nsec * 1e-9);
    }
    Session S;
    int pr = parse(argc, argv, S.options);
    if (pr == 2) return 0;
    if (pr != 0) return 1;
    if (S.options.window_mb && !S.options.dump_records) RecordPump::raw_cap_setting() = static_cast<size_t>(S.options.window_mb) << 20;
    if (S.options.dump_records) return dump_records(S.options);
    // the HIP runtime starts (0.1 - 0.3 s) while the database is loaded and the first file opened and sampled
    struct WarmUp {
        std::thread t;
        ~WarmUp() {
            if (t.joinable()) t.join();
        }
    } warm_up{std::thread([devices = S.options.devices.size() > 1 ? S.options.devices : std::vector<int>{S.options.device}] { lazy::load(devices); })};
    Lap watch;
    // slimm::slimm(): collect_bam_files + load_slimm_database (src/slimm.hpp:96-101, 306-326)
    if (S.options.is_directory) {
        S.input_paths = get_bam_files_in_directory(S.options.input_path);
        if (S.options.verbose)
            std::cerr << S.input_paths.size() << " SAM/BAM Files found under the directory: " << S.options.input_path << "!\n";
    } else if (access(S.options.input_path.c_str(), 0) == 0) {
        S.input_paths.push_back(S.options.input_path);
    } else {
        std::cerr << S.options.input_path << " is not a file use -d option for a directory.\n";
        return 1;
    }
    std::string err;
    Trace trace;
    if (!load_slimm_database(S.options.database_path, S.db, err)) {
        std::cerr << "slimm: " << err << "\n";
        return 1;
    }
    trace.mark("load .sldb");
    auto closing_lines = [&] {
    std::cerr << "\n*****************************************************************\n";
        std::cerr << S.total_hits << " SAM/BAM alignment records are proccessed.\n";
        std::cerr << "Taxonomic profiles are written to: \n   " << get_directory(S.options.output_prefix) << "\n";
        std::cerr << "Total time elapsed: " << watch.elapsed() << " secs\n";
        if (trace.on) {
            // since exec(): /proc/self/stat field 22 is the start time in clock ticks since boot
            double up = 0;
            if (FILE* f = fopen("/proc/uptime", "r")) {
                if (fscanf(f, "%lf", &up) != 1) up = 0;
                fclose(f);
            }
            unsigned long long start_ticks = 0;
            if (FILE* f = fopen("/proc/self/stat", "r")) {
                char buf[2048];
                if (fgets(buf, sizeof buf, f)) {
                    const char* p = strrchr(buf, ')');
                    int field = 2;
                    for (p = p ? p + 1 : buf; *p && field < 22; ++p)
                        if (*p == ' ') ++field;
                    start_ticks = strtoull(p, nullptr, 10);
                }
                fclose(f);
            }
            const double since_exec = up - static_cast<double>(start_ticks) / sysconf(_SC_CLK_TCK);
            fprintf(stderr, "[trace] main() reached its end %.0f ms after exec (10 ms resolution)\n", since_exec * 1e3);
            struct timespec now;
            clock_gettime(CLOCK_REALTIME, &now);
            fprintf(stderr, "[trace] leaving at %.6f (epoch seconds)\n", now.tv_sec + now.tv_nsec * 1e-9);
        }
        // (Every output file is written and closed.  Leaving through _exit here, without the teardown in this process, measured
        // SLOWER end to end since the window buffers are page-locked: the kernel driver then takes the process's queues, pinned
        // pages and device memory back on its own -- 0.20 - 0.23 s from _exit to the parent's wait() returning against 0.07 s of
        // orderly teardown + 0.09 s; round 4.  Round 6: _exit behind our own release of everything -- only the runtime's
        // atexit teardown skipped -- is inside the run-to-run noise of 0.64 - 0.84 s, profiles/round6/05_exit_experiment.txt.)
        std::cerr.flush();
        fflush(nullptr);
    };
    for (size_t n = 0; n < S.input_paths.size(); ++n) {
        if (!get_profiles(S, n)) return 1;
        trace.mark("get_profiles + its buffers released");
    }
    closing_lines();
    return 0;
}
