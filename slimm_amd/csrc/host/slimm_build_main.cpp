// `slimm_build -nm names.dmp -nd nodes.dmp [-o out.sldb] [-b N] [-v] FASTA ACC2TAXID [ACC2TAXID ...]`
//
// The database builder of the reference (reference src/slimm_build.cpp:54-386): offline,
// host-only, no GPU.  It defines the shape of the lineage table the alignment-to-profile path consumes:
//   accession -> [own taxid, species, genus, family, order, class, phylum, superkingdom]   (0 = rank missing)
//   taxid     -> (rank, scientific name)
// Steps, in the reference's order:
//   1. accessions of the FASTA (or FASTQ, optionally gzip'ed) ids            src/slimm_build.cpp:151-170
//   2. accession -> taxid from NCBI accession2taxid files, read in batches   src/slimm_build.cpp:175-278
//   3. parents / ranks from nodes.dmp, scientific names from names.dmp, one walk to the root per accession
//                                                                            src/slimm_build.cpp:283-346
//   4. the .sldb file (sldb.cpp)                       src/misc.hpp:178-187
// The containers and the order of insertions into them follow the reference, so that on the same C++ library the
// entries land in the file in the reference's order.  Deviations, all on malformed input only: a nodes.dmp line whose
// taxid or parent is not a number is skipped (the reference re-uses the previous line's values), a parent chain that
// never reaches taxid 1 or a missing node stops after 2^20 steps (the reference loops forever), and an unreadable input
// ends with a message and exit code 1 (the reference throws a C string nobody catches).
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

#include "accession.hpp"
#include "sldb.hpp"

namespace {

using namespace slimm;

struct Options {  // arg_options, reference src/slimm_build.cpp:54-71
    uint32_t batch = 1000000;
    bool verbose = false;
    std::string fasta_path, nodes_path, names_path, output_path = "slimm_db.sldb";
    std::vector<std::string> ac_taxid_paths;
};

void usage() {
    std::cerr << "slimm_build - reduced taxonomic information for the accessions of a multi-fasta file\n"
                 "usage: slimm_build -nm \"NAMES.dmp\" -nd \"NODES.dmp\" -o \"SLIMM.sldb\" [OPTIONS] \"FASTA\" \"ACCESSION2TAXAID\" "
                 "[ACCESSION2TAXAID_2 ...]\n"
                 "  -o,  --output-file FILE   the output file, must end in .sldb (default slimm_db.sldb)\n"
                 "  -nm, --names FILE         NCBI's names.dmp (taxid -> name)\n"
                 "  -nd, --nodes FILE         NCBI's nodes.dmp (the taxonomic tree)\n"
                 "  -b,  --batch INT          maximum number of mappings held in memory at once (default 1000000)\n"
                 "  -v,  --verbose            verbose output\n";
}

bool ends_with(const std::string& s, const char* suffix) {
    size_t n = strlen(suffix);
    return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}

// 0 = run, 1 = error, 2 = help shown
int parse_command_line(Options& o, int argc, char** argv) {
    std::vector<std::string> positional;
    bool have_names = false, have_nodes = false;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto value = [&](std::string& dst) {
            if (i + 1 >= argc) {
                std::cerr << "slimm_build: option " << a << " needs a value\n";
                return false;
            }
            dst = argv[++i];
            return true;
        };
        std::string v;
        if (a == "-h" || a == "--help") {
            usage();
            return 2;
        } else if (a == "-o" || a == "--output-file") {
            if (!value(o.output_path)) return 1;
        } else if (a == "-nm" || a == "--names") {
            if (!value(o.names_path)) return 1;
            have_names = true;
        } else if (a == "-nd" || a == "--nodes") {
            if (!value(o.nodes_path)) return 1;
            have_nodes = true;
        } else if (a == "-b" || a == "--batch") {
            if (!value(v)) return 1;
            char* end = nullptr;
            long long n = strtoll(v.c_str(), &end, 10);
            if (v.empty() || *end || n < 0 || n > 0xffffffffll) {
                std::cerr << "slimm_build: the value of --batch must be an integer\n";
                return 1;
            }
            o.batch = static_cast<uint32_t>(n);
        } else if (a == "-v" || a == "--verbose") {
            o.verbose = true;
        } else if (a.size() > 1 && a[0] == '-') {
            std::cerr << "slimm_build: unknown option " << a << "\n";
            return 1;
        } else {
            positional.push_back(a);
        }
    }
    if (!have_names || !have_nodes) {  // setRequired, reference src/slimm_build.cpp:103, 107
        std::cerr << "slimm_build: options --names and --nodes are required\n";
        return 1;
    }
    if (positional.size() < 2) {
        std::cerr << "slimm_build: a FASTA file and at least one ACCESSION2TAXAID file are needed\n";
        usage();
        return 1;
    }
    if (!ends_with(o.output_path, ".sldb")) {  // setValidValues, reference src/slimm_build.cpp:98
        std::cerr << "slimm_build: the output file must end in .sldb\n";
        return 1;
    }
    o.fasta_path = positional[0];
    o.ac_taxid_paths.assign(positional.begin() + 1, positional.end());
    return 0;
}

// ---- lines of a plain or gzip'ed text file --------------------------------------------------------------------------
class LineReader {
public:
    bool open(const std::string& path) {
        f_ = gzopen(path.c_str(), "rb");
        if (f_) gzbuffer(f_, 1 << 20);
        buf_.resize(1 << 20);
        return f_ != nullptr;
    }
    ~LineReader() {
        if (f_) gzclose(f_);
    }
    // the next line without its '\n' (a '\r' before it stays, as with std::getline); false at end of file
    bool next(const char*& p, size_t& n) {
        for (;;) {
            const char* nl = static_cast<const char*>(memchr(buf_.data() + pos_, '\n', end_ - pos_));
            if (nl) {
                p = buf_.data() + pos_;
                n = nl - p;
                pos_ += n + 1;
                return true;
            }
            if (eof_) {
                if (pos_ == end_) return false;
                p = buf_.data() + pos_;
                n = end_ - pos_;
                pos_ = end_;
                return true;
            }
            if (pos_ > 0) {
                memmove(buf_.data(), buf_.data() + pos_, end_ - pos_);
                end_ -= pos_;
                pos_ = 0;
            }
            if (end_ == buf_.size()) buf_.resize(buf_.size() * 2);
            int got = gzread(f_, buf_.data() + end_, static_cast<unsigned>(std::min<size_t>(buf_.size() - end_, 1u << 30)));
            if (got <= 0) eof_ = true;
            else end_ += got;
        }
    }

private:
    gzFile f_ = nullptr;
    std::vector<char> buf_;
    size_t pos_ = 0, end_ = 0;
    bool eof_ = false;
};

// operator>> into a uint32_t: leading white space, then digits; false (value untouched) when there is no digit
bool parse_u32(const char* p, const char* end, uint32_t& out) {
    while (p < end && (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\v' || *p == '\f')) ++p;
    if (p < end && *p == '+') ++p;
    if (p >= end || *p < '0' || *p > '9') return false;
    uint64_t v = 0;
    bool over = false;
    for (; p < end && *p >= '0' && *p <= '9'; ++p) {
        v = v * 10 + (*p - '0');
        if (v > 0xffffffffull) over = true, v = 0xffffffffull;
    }
    out = over ? 0xffffffffu : static_cast<uint32_t>(v);
    return true;
}

// `stream >> value` of a std::stringstream positioned at p (C++11 libstdc++, what the reference's
// `linestream >> taxid` does, src/slimm_build.cpp:189): white space is skipped first; when the text ends there the
// sentry fails and the value is left alone; otherwise num_get runs, and a field that does not start a number
// stores 0 (LWG 23 / C++11), an overflowing one stores the maximum.
void stream_extract_u32(const char* p, const char* end, uint32_t& out) {
    while (p < end && (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\v' || *p == '\f' || *p == '\n')) ++p;
    if (p >= end) return;
    if (!parse_u32(p, end, out)) out = 0;
}

// up to `want` tab-separated fields of a line
size_t split_tabs(const char* p, size_t n, const char** f, size_t* fl, size_t want) {
    size_t k = 0;
    const char* end = p + n;
    while (k < want) {
        const char* t = static_cast<const char*>(memchr(p, '\t', end - p));
        f[k] = p;
        fl[k] = (t ? t : end) - p;
        ++k;
        if (!t) break;
        p = t + 1;
    }
    return k;
}

// ---- step 1: get_accession_numbers, reference src/slimm_build.cpp:151-170 -------------------------------------------
bool get_accession_numbers(std::set<std::string>& accessions, const Options& o) {
    std::cerr << "[MSG] getting accessions numbers from fasta file ...\n";
    LineReader in;
    if (!in.open(o.fasta_path)) {
        std::cerr << "Unable to open contigs File: " << o.fasta_path << "\n";
        return false;
    }
    const char* p;
    size_t n;
    bool fastq = false, first = true;
    size_t seq_len = 0, qual_left = 0;
    int state = 0;  // FASTQ: 0 id expected, 1 in sequence, 2 in qualities
    while (in.next(p, n)) {
        if (n && p[n - 1] == '\r') --n;
        if (first && n) {
            fastq = p[0] == '@';
            first = false;
        }
        if (!fastq) {
            if (n && p[0] == '>') accessions.insert(get_accession_id(std::string(p + 1, n - 1)));
            continue;
        }
        if (state == 0) {
            if (n && p[0] == '@') {
                accessions.insert(get_accession_id(std::string(p + 1, n - 1)));
                state = 1;
                seq_len = 0;
            }
        } else if (state == 1) {
            if (n && p[0] == '+') {
                state = seq_len ? 2 : 0;
                qual_left = seq_len;
            } else {
                seq_len += n;
            }
        } else {  // qualities may start with '@': count characters instead of looking at them
            qual_left -= std::min(qual_left, n);
            if (qual_left == 0) state = 0;
        }
    }
    return true;
}

// ---- step 2: get_taxid_from_accession, reference src/slimm_build.cpp:175-278 ----------------------------------------
void print_missed_accessions(const std::set<std::string>& accessions, const Options& o) {  // :200-219
    std::string missed_path = o.output_path.substr(0, o.output_path.size() - 4) + "missed";
    std::ofstream missed(missed_path);
    std::cerr << "[WARNING!] " << accessions.size() << " accessions (";
    uint32_t count = 3;
    for (auto it = accessions.begin(); count > 0 && it != accessions.end(); --count, ++it) std::cerr << *it << ", ";
    std::cerr << "...) were not mapped to taxaid.\n";
    for (auto& a : accessions) missed << a << "\n";
    missed.close();
    std::cerr << "[WARNING!] Take a look at " << missed_path << " file for a complete list.\n";
    std::cerr << "[WARNING!] Try including the more ACCESSION2TAXAID MAP FILE (e.g. dead_nucl.accession2taxid)\n";
}

bool get_taxid_from_accession(SlimmDatabase& db, std::set<std::string>& accessions, const Options& o) {
    std::cerr << "[MSG] mapping accessions to taxaid ...\n";
    const size_t accessions_count = accessions.size();
    uint32_t map_file_number = 1;
    // The reference loads every batch into a hash map and then looks each outstanding accession up in it; only the
    // lines naming an outstanding accession can matter, so only those are kept (the last one of a batch wins, as with
    // the map's operator[]), and the hits enter the database in the order of the sorted accession set, as there.
    std::unordered_map<std::string, uint32_t> hits;
    for (const std::string& map_path : o.ac_taxid_paths) {
        if (accessions.empty()) return true;
        LineReader in;
        if (!in.open(map_path)) {
            std::cerr << "[WARNING!] cannot open " << map_path << "\n";  // an ifstream that failed to open reads nothing
            ++map_file_number;
            continue;
        }
        uint32_t iter_number = 1;
        bool more = true;
        while (more) {
            hits.clear();
            uint32_t taxid = 0, lines = 0;  // `taxid` survives a line that ends before its third column; a third
                                            // column that is not a number (the header line) stores 0, :180-189
            const char* p;
            size_t n;
            std::string ac;
            while ((more = in.next(p, n))) {
                const char* f[3];
                size_t fl[3];
                size_t k = split_tabs(p, n, f, fl, 3);
                if (k == 3) stream_extract_u32(f[2], p + n, taxid);
                ac.assign(f[0], fl[0]);
                if (accessions.count(ac)) hits[ac] = taxid;
                if (++lines >= o.batch) break;
            }
            if (lines == 0) break;
            if (accessions.empty()) return true;
            if (o.verbose) {
                std::cerr << "[VERBOSE MSG] mapping file: [" << map_file_number << "/" << o.ac_taxid_paths.size() << "]\t";
                std::cerr << "iter: [" << iter_number << "]\t";
                std::cerr << "accessions left: [" << accessions.size() << "/" << accessions_count << "]\n";
                ++iter_number;
            }
            std::vector<const std::string*> found;
            found.reserve(hits.size());
            for (auto& kv : hits) found.push_back(&kv.first);
            std::sort(found.begin(), found.end(), [](const std::string* a, const std::string* b) { return *a < *b; });
            for (const std::string* a : found) {
                std::vector<uint32_t> lineage(8, 0);
                lineage[0] = hits[*a];
                db.ac_taxid[*a] = std::move(lineage);
                accessions.erase(*a);
            }
        }
        ++map_file_number;
    }
    if (!accessions.empty()) print_missed_accessions(accessions, o);
    return true;
}

// ---- step 3: fill_name_taxid_linage, reference src/slimm_build.cpp:283-346 ------------------------------------------
bool fill_name_taxid_lineage(SlimmDatabase& db, const Options& o) {
    std::cerr << "[MSG] loading nodes and names mappings from files ...\n";
    std::unordered_map<uint32_t, std::pair<uint32_t, uint32_t>> taxid_parent;  // taxid -> (rank, parent)
    std::unordered_map<uint32_t, std::string> taxid_name;
    const char* p;
    size_t n;
    {
        LineReader in;
        if (!in.open(o.nodes_path)) {
            std::cerr << "slimm_build: cannot open " << o.nodes_path << "\n";
            return false;
        }
        while (in.next(p, n)) {  // taxid \t | \t parent \t | \t rank \t | ...
            const char* f[5];
            size_t fl[5];
            uint32_t taxid, parent;
            if (split_tabs(p, n, f, fl, 5) < 5) continue;
            if (!parse_u32(f[0], f[0] + fl[0], taxid) || !parse_u32(f[2], f[2] + fl[2], parent)) continue;
            taxid_parent[taxid] = std::make_pair(to_taxa_ranks(std::string(f[4], fl[4])), parent);
        }
    }
    {
        LineReader in;
        if (!in.open(o.names_path)) {
            std::cerr << "slimm_build: cannot open " << o.names_path << "\n";
            return false;
        }
        static const char kSci[] = "scientific name";
        while (in.next(p, n)) {  // taxid \t | \t name \t | \t unique name \t | \t name class \t |
            if (std::search(p, p + n, kSci, kSci + sizeof(kSci) - 1) == p + n) continue;
            const char* f[3];
            size_t fl[3];
            uint32_t taxid;
            if (split_tabs(p, n, f, fl, 3) < 3 || !parse_u32(f[0], f[0] + fl[0], taxid)) continue;
            taxid_name[taxid] = std::string(f[2], fl[2]);
        }
    }
    std::cerr << "[MSG] getting taxonomic linages and resolving names ...\n";
    for (auto& kv : db.ac_taxid) {
        uint32_t tid = kv.second[0];
        db.taxid_name[tid] = std::make_pair(0u, taxid_name[tid]);  // the accession's own node: strain_lv, :329
        for (uint32_t steps = 0; tid != 1 && steps < (1u << 20); ++steps) {
            auto it = taxid_parent.find(tid);
            if (it == taxid_parent.end()) break;
            uint32_t rank = it->second.first;
            if (rank >= 1 && rank <= 7) {  // species ... superkingdom; overwrites the strain_lv entry of a species node
                kv.second[rank] = tid;
                db.taxid_name[tid] = std::make_pair(rank, taxid_name[tid]);
            }
            tid = it->second.second;
        }
    }
    return true;
}

}  // namespace

int main(int argc, char** argv) {
    Options o;
    int r = parse_command_line(o, argc, argv);
    if (r) return r == 1;
    std::set<std::string> accessions;
    if (!get_accession_numbers(accessions, o)) return 1;
    SlimmDatabase db;
    if (!get_taxid_from_accession(db, accessions, o)) return 1;
    if (!fill_name_taxid_lineage(db, o)) return 1;
    std::string err;
    if (!save_slimm_database(o.output_path, db, err)) {
        std::cerr << "slimm_build: " << err << "\n";
        return 1;
    }
    return 0;
}
