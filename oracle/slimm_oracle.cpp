// ============================================================================
// slimm_oracle.cpp -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT.
//
// A single-threaded CPU restatement of SLIMM's alignment-to-profile path
// (reference: /root/reference/src/slimm.hpp, read_stat.hpp, reference_contig.hpp,
// misc.hpp).  It exists so the HIP path can be checked against something that
// follows the reference step by step.  Only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may load it; the product (slimm_amd/,
// libslimm_hip.so) never does.
//
// PARITY PIN STATUS: the reference repository holds no tests, golden vectors
// or fixtures for this path (reference CMakeLists.txt:69-70 has the tests dir
// commented out) and it cannot be built in this image (SeqAn 2.3.1 and cereal
// are un-vendored submodules: reference .gitmodules:1-8).  The only reference
// outputs available are the two known-answer micro-cases recorded in
// SURVEY.md Appendix C.1/C.2 (observed by the surveyor from the reference's
// own headers); this oracle is checked against those (tests/test_oracle_golden.py).
// Beyond those two cases: PARITY UNPINNED.
//
// The same containers as the reference are used on purpose
// (unordered_map<string,...> for reads, unordered_map<uint32_t,...> for taxon
// counts, std::set for children) so that iteration-order dependent behaviour
//  comes out the way libstdc++ makes it come
// out for the reference.
// ============================================================================
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <ctime>
#include <iomanip>
#include <numeric>
#include <set>
#include <sstream>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

namespace {

constexpr uint32_t kLineageLen = 8;  // misc.hpp:4 LINAGE_LENGTH

// misc.hpp:24-35 taxa_ranks
enum Rank : uint32_t { kStrain = 0, kSpecies, kGenus, kFamily, kOrder, kClass, kPhylum, kSuperkingdom, kIntermediate };

// misc.hpp:37-48
Rank rank_from_string(const std::string& s) {
    static const char* names[] = {"strain", "species", "genus", "family", "order", "class", "phylum", "superkingdom"};
    for (uint32_t i = 0; i < 8; ++i)
        if (s == names[i]) return static_cast<Rank>(i);
    return kIntermediate;
}
// misc.hpp:51-62
std::string rank_long(uint32_t r) {
    static const char* names[] = {"strain", "species", "genus", "family", "order", "class", "phylum", "superkingdom"};
    return r < 8 ? names[r] : "intermidiate";
}
// misc.hpp:64-75
std::string rank_short(uint32_t r) {
    static const char* names[] = {"r", "s", "g", "f", "o", "c", "p", "k"};
    return r < 8 ? names[r] : "i";
}

// read_stat.hpp:46-59  target_reference
struct Target {
    uint32_t ref;
    std::vector<uint32_t> bins;
    Target(uint32_t r, uint32_t b) : ref(r) {
        bins.reserve(10);
        bins.push_back(b);
    }
};

// read_stat.hpp:64-136  read_stat
struct Read {
    std::vector<Target> targets;
    uint32_t refs_length_sum = 0;

    bool is_uniq() const { return targets.size() == 1; }

    // read_stat.hpp:116-135.  The loop variable is a COPY (the reference iterates
    // `for (auto tar : targets)`), so a repeated (read, ref) pair appends its bin to
    // a temporary that is thrown away: only the first record of a pair is kept (Q1).
    void add_target(int32_t ref, uint32_t bin) {
        if (targets.empty()) {
            targets.push_back(Target(ref, bin));
            return;
        }
        for (Target tar : targets) {
            if (tar.ref == static_cast<uint32_t>(ref)) {
                tar.bins.push_back(bin);
                return;
            }
        }
        targets.push_back(Target(ref, bin));
    }

    // read_stat.hpp:98-114
    void keep_only(const std::set<uint32_t>& valid, const std::vector<uint32_t>& ref_len) {
        if (targets.empty()) return;
        std::vector<Target> kept;
        for (Target tr : targets) {
            if (valid.find(tr.ref) != valid.end())
                kept.push_back(tr);
            else
                refs_length_sum -= ref_len[tr.ref];
        }
        std::swap(targets, kept);
    }
};

// reference_contig.hpp:67-100 bins_coverage
struct Bins {
    uint32_t width = 0, n = 0;
    std::vector<uint32_t> h;
    int32_t nz_cache = -1;
    void init(uint32_t len, uint32_t w) {
        width = w;
        n = len / w + 1;  // reference_contig.hpp:80
        h.assign(n, 0);
        nz_cache = -1;
    }
    // reference_contig.hpp:84-91 (cached on first call)
    uint32_t nonzero() {
        if (nz_cache == -1) nz_cache = static_cast<int32_t>(n - std::count(h.begin(), h.end(), 0u));
        return static_cast<uint32_t>(nz_cache);
    }
};

// reference_contig.hpp:102-208 reference_contig
struct Ref {
    std::string accession;
    uint32_t taxid = 0, length = 0;
    uint32_t reads = 0, uniq = 0, uniq2 = 0;
    Bins cov, ucov, ucov2;
    float abundance = 0.f, uabundance = 0.f, uabundance2 = 0.f;
    float cov_pct() { return float(cov.nonzero()) / cov.n; }      // :148-151
    float ucov_pct() { return float(ucov.nonzero()) / ucov.n; }   // :152-155
    float ucov2_pct() { return float(ucov2.nonzero()) / ucov2.n; }
    // :188-207 (+ misc.hpp:285-289 mean)
    static float depth(Bins& b) {
        if (b.nonzero() == 0) return 0.0f;
        std::vector<float> f;
        f.reserve(b.n);
        for (uint32_t i = 0; i < b.n; ++i) f.push_back(float(b.h[i]));
        float s = std::accumulate(f.begin(), f.end(), 0.0f);
        return s / f.size();
    }
};

// misc.hpp:197-216 get_quantile_cut_off<float>
float quantile_cut_off(std::vector<float> v, float q) {
    if (v.empty()) return 0;
    float total = std::accumulate(v.begin(), v.end(), 0.0f);
    float cutoff = 0.0f, sub = 0.0f;
    std::sort(v.begin(), v.end());
    uint32_t i = static_cast<uint32_t>(v.size() - 1);
    while ((float(sub) / total) < q && i > 0.0f) {
        sub += v[i];
        --i;
    }
    cutoff = v[i];
    return cutoff;
}

// misc.hpp:415-422 get_accession_id: text before the first whitespace, '.' or '|'.
std::string accession_of(const std::string& name) {
    size_t i = 0;
    while (i < name.size()) {
        unsigned char c = static_cast<unsigned char>(name[i]);
        if (c == '.' || c == '|' || c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f') break;
        ++i;
    }
    return name.substr(0, i);
}

struct Oracle {
    // options (slimm.hpp:49-87)
    float cov_cut_off = 0.95f, abundance_cut_off = 0.01f;
    uint32_t bin_width = 0, min_reads = 0;
    std::string rank = "species";

    // database (misc.hpp:77-100)
    std::unordered_map<std::string, std::vector<uint32_t>> ac_taxid;
    std::unordered_map<uint32_t, std::tuple<uint32_t, std::string>> taxid_name;

    // state (slimm.hpp:103-127)
    uint32_t avg_read_length = 0, matched_ref_length = 0, reference_count = 0;
    uint32_t failed_by_min_read = 0, failed_by_cov = 0, failed_by_ucov = 0;
    uint32_t hits = 0, uniq_hits = 0, matches = 0, uniq_matches = 0, uniq_matches2 = 0;
    std::set<uint32_t> valid_refs;
    std::vector<uint32_t> considered;
    std::vector<Ref> refs;
    std::vector<uint32_t> ref_len;
    std::unordered_map<std::string, Read> reads;
    std::unordered_map<uint32_t, uint32_t> taxon_count;
    std::unordered_map<uint32_t, std::set<uint32_t>> taxon_children;
    // snapshot after step (1) of get_reads_lca_count (direct LCA hits only)
    std::unordered_map<uint32_t, uint32_t> lca_direct_count;
    std::unordered_map<uint32_t, std::set<uint32_t>> lca_direct_children;

    float cc_cache = 0.0f, ucc_cache = 0.0f;  // slimm.hpp:155-156 (survive reset, Q8)
    bool profile_written = false;
    std::string profile_tsv, raw_tsv, cov_csv, ucov_csv, ucov2_csv;
    uint32_t profile_count = 0, profile_failed = 0;

    // slimm.hpp:498-514
    void set_rank(const std::string& r) {
        rank = r;
        considered.clear();
        if (r == "all") {
            for (uint32_t i = 8; i > 0; --i) considered.push_back(i - 1);
        } else if (r == "superkingdom") {
            considered.push_back(rank_from_string(r));
        } else {
            considered.push_back(rank_from_string(r) + 1);
            considered.push_back(rank_from_string(r));
        }
    }

    // slimm.hpp:167-188 (cut-off caches and options are NOT cleared)
    void reset() {
        avg_read_length = matched_ref_length = reference_count = 0;
        failed_by_min_read = failed_by_cov = failed_by_ucov = 0;
        hits = uniq_hits = matches = uniq_matches = uniq_matches2 = 0;
        valid_refs.clear();
        refs.clear();
        ref_len.clear();
        reads.clear();
        taxon_count.clear();
        taxon_children.clear();
        lca_direct_count.clear();
        lca_direct_children.clear();
        profile_written = false;
        profile_tsv.clear();
        raw_tsv.clear();
        cov_csv.clear();
        ucov_csv.clear();
        ucov2_csv.clear();
    }

    // slimm.hpp:409-445: bin width default, per-contig init, unknown accession -> zero lineage (Q13)
    void init_refs(const std::vector<std::string>& names, const std::vector<uint32_t>& lens, uint32_t avg_len) {
        avg_read_length = avg_len;
        if (bin_width == 0) bin_width = avg_read_length;
        refs.resize(names.size());
        ref_len = lens;
        for (size_t i = 0; i < names.size(); ++i) {
            std::string acc = accession_of(names[i]);
            uint32_t taxid = 0;
            auto it = ac_taxid.find(acc);
            if (it != ac_taxid.end())
                taxid = it->second[0];
            else
                ac_taxid[acc] = std::vector<uint32_t>(kLineageLen, 0);
            Ref& r = refs[i];
            r = Ref();
            r.accession = acc;
            r.taxid = taxid;
            r.length = lens[i];
            r.cov.init(lens[i], bin_width);
            r.ucov = r.cov;
            r.ucov2 = r.cov;
        }
    }

    // slimm.hpp:194-213, one record
    inline void feed(const std::string& qname, uint16_t flag, int32_t rid, int32_t pos) {
        if ((flag & 0x4) || rid == -1) return;
        uint32_t center = std::min(static_cast<uint32_t>(pos) + (avg_read_length / 2), refs[rid].length);
        uint32_t bin = center / bin_width;
        std::string key = qname;
        if (flag & 0x40)
            key += ".1";
        else if (flag & 0x80)
            key += ".2";
        reads[key].add_target(rid, bin);
        ++hits;
    }

    // slimm.hpp:216-303
    void finish_analyze() {
        if (hits == 0) return;
        for (auto it = reads.begin(); it != reads.end(); ++it) {
            Read& rd = it->second;
            if (rd.is_uniq()) {
                uint32_t r = rd.targets[0].ref;
                rd.refs_length_sum += refs[r].length;
                ++uniq_matches;
                size_t pc = rd.targets[0].bins.size();
                refs[r].reads += pc;
                rd.refs_length_sum += refs[r].length;
                for (size_t j = 0; j < pc; ++j) ++refs[r].cov.h[rd.targets[0].bins[j]];
                refs[r].uniq += 1;
                uniq_hits += 1;
                ++refs[r].ucov.h[rd.targets[0].bins[0]];
            } else {
                size_t len = rd.targets.size();
                for (size_t i = 0; i < len; ++i) {
                    uint32_t r = rd.targets[i].ref;
                    rd.refs_length_sum += refs[r].length;
                    refs[r].reads += rd.targets[i].bins.size();
                    for (uint32_t b : rd.targets[i].bins) ++refs[r].cov.h[b];
                }
            }
        }
        matches = static_cast<uint32_t>(reads.size());

        float total = 0.0f;
        for (size_t i = 0; i < refs.size(); ++i) {
            if (refs[i].reads > 0) {
                ++reference_count;
                matched_ref_length += refs[i].length;
                refs[i].abundance = float(refs[i].reads * 100) / hits;  // u32 product (Q11)
                total += refs[i].abundance / refs[i].length;
            } else {
                refs[i].abundance = 0.0f;
            }
        }
        for (size_t i = 0; i < refs.size(); ++i)
            if (refs[i].reads > 0) refs[i].abundance = (refs[i].abundance * 100) / (total * refs[i].length);

        total = 0.0f;
        for (size_t i = 0; i < refs.size(); ++i) {
            if (refs[i].uniq > 0) {
                refs[i].uabundance = float(refs[i].uniq * 100) / uniq_hits;
                total += refs[i].uabundance / refs[i].length;
            } else {
                refs[i].uabundance = 0.0f;
            }
        }
        for (size_t i = 0; i < refs.size(); ++i)
            if (refs[i].uniq > 0) refs[i].uabundance = (refs[i].uabundance * 100) / (total * refs[i].length);
    }

    // slimm.hpp:328-344
    float coverage_cut_off() {
        if (cc_cache == 0.0 && cov_cut_off < 1.0) {
            std::vector<float> v;
            v.reserve(refs.size());
            for (size_t i = 0; i < refs.size(); ++i)
                if (refs[i].uniq > 0) v.push_back(refs[i].cov_pct());
            cc_cache = quantile_cut_off(v, cov_cut_off);
        }
        return cc_cache;
    }
    // slimm.hpp:672-688
    float uniq_coverage_cut_off() {
        if (ucc_cache == 0.0 && cov_cut_off < 1.0) {
            std::vector<float> v;
            v.reserve(refs.size());
            for (size_t i = 0; i < refs.size(); ++i)
                if (refs[i].uniq > 0) v.push_back(refs[i].ucov_pct());
            ucc_cache = quantile_cut_off(v, cov_cut_off);
        }
        return ucc_cache;
    }
    // slimm.hpp:346-349
    float expected_coverage() const { return float(avg_read_length * matches) / matched_ref_length; }

    // slimm.hpp:351-392
    void filter() {
        compute_valid();
        apply_valid();
    }
    // slimm.hpp:353-378
    void compute_valid() {
        uint32_t n = static_cast<uint32_t>(refs.size());
        for (uint32_t i = 0; i < n; ++i) {
            if (refs[i].reads == 0) continue;
            if (refs[i].cov_pct() >= coverage_cut_off() && refs[i].ucov_pct() >= uniq_coverage_cut_off()) {
                valid_refs.insert(i);
            } else {
                if (refs[i].ucov_pct() < uniq_coverage_cut_off()) ++failed_by_ucov;
                if (refs[i].reads < min_reads) ++failed_by_min_read;
                if (refs[i].cov_pct() < coverage_cut_off()) ++failed_by_cov;
            }
        }
    }
    // slimm.hpp:380-391
    void apply_valid() {
        for (auto it = reads.begin(); it != reads.end(); ++it) {
            it->second.keep_only(valid_refs, ref_len);
            if (it->second.is_uniq()) {
                uint32_t r = it->second.targets[0].ref;
                refs[r].uniq2 += 1;
                uniq_matches2 += 1;
                ++refs[r].ucov2.h[it->second.targets[0].bins[0]];
            }
        }
    }

    // slimm.hpp:516-531.  Level scan over refs in ascending id order; returns the last
    // value read, i.e. lineage[max ref][7] when no level agrees (Q4).
    uint32_t lca_of(const std::set<uint32_t>& ids) {
        uint32_t t = 1;
        for (uint32_t lv = 0; lv < kLineageLen; ++lv) {
            std::set<uint32_t> level;
            for (uint32_t r : ids) {
                t = ac_taxid[refs[r].accession][lv];
                level.insert(t);
            }
            if (level.size() == 1) break;
        }
        return t;
    }

    static void bump(std::unordered_map<uint32_t, uint32_t>& m, uint32_t k, uint32_t v) {  // misc.hpp:138-147
        auto p = m.find(k);
        if (p != m.end())
            p->second += v;
        else
            m[k] = v;
    }

    // slimm.hpp:533-611
    void lca_count() {
        lca_direct();
        lca_propagate();
    }
    // slimm.hpp:536-557
    void lca_direct() {
        for (auto it = reads.begin(); it != reads.end(); ++it) {
            size_t len = it->second.targets.size();
            if (len > 1) {
                std::set<uint32_t> ids;
                for (size_t i = 0; i < len; ++i) ids.insert(it->second.targets[i].ref);
                uint32_t t = lca_of(ids);
                bump(taxon_count, t, 1u);
                taxon_children[t].insert(ids.begin(), ids.end());
            }
        }
        lca_direct_count = taxon_count;
        lca_direct_children = taxon_children;
    }
    // slimm.hpp:560-610
    void lca_propagate() {
        std::unordered_map<uint32_t, uint32_t> snapshot = taxon_count;
        uint32_t receiver = 0;
        for (auto tc : snapshot) {
            uint32_t rnk = std::get<0>(taxid_name[tc.first]);  // default-constructs -> strain_lv (Q6)
            std::string first_child_acc = "";
            for (auto child : taxon_children.at(tc.first)) {
                first_child_acc = refs[child].accession;
                break;
            }
            std::vector<uint32_t> lin = ac_taxid[first_child_acc];
            std::set<uint32_t> ids = taxon_children[tc.first];
            for (uint32_t j = rnk + 1; j < kLineageLen; ++j) {
                receiver = lin[j];
                bump(taxon_count, receiver, tc.second);
                taxon_children[receiver].insert(ids.begin(), ids.end());
            }
        }

        for (uint32_t i = 0; i < refs.size(); ++i) {
            if (refs[i].uniq2 > 0) {
                std::vector<uint32_t> lin = ac_taxid[refs[i].accession];
                std::set<uint32_t> ids = taxon_children[lin[0]];
                for (uint32_t j = 1; j < kLineageLen; ++j) {  // starts at 1 (Q9)
                    receiver = lin[j];
                    auto p = taxon_count.find(receiver);
                    if (p != taxon_count.end())
                        p->second += refs[i].uniq2;
                    else
                        taxon_count[receiver] = refs[i].uniq2;
                    taxon_children[receiver].insert(i);
                    taxon_children[receiver].insert(ids.begin(), ids.end());
                }
            }
        }
    }

    // slimm.hpp:690-710
    std::string lineage_string(uint32_t rnk, const std::vector<uint32_t>& lin) {
        std::string name = std::get<1>(taxid_name[lin[rnk]]);
        if (name == "") name = "unknown_" + rank_long(rnk);
        std::string s = rank_short(rnk) + "__" + name;
        for (uint32_t i = rnk + 1; i < kLineageLen; ++i) {
            name = std::get<1>(taxid_name[lin[i]]);
            if (name == "") name = "unknown_" + rank_long(i);
            s = rank_short(i) + "__" + name + "|" + s;
        }
        return s;
    }
    // slimm.hpp:712-730
    std::string lineage_string(uint32_t rnk, uint32_t taxid) {
        std::vector<uint32_t> lin;
        if (taxid == 0) {
            lin.resize(kLineageLen, 0);
        } else {
            std::string acc = "";
            for (auto child : taxon_children.at(taxid)) {
                acc = refs[child].accession;
                break;
            }
            lin = ac_taxid[acc];
        }
        return lineage_string(rnk, lin);
    }

    // slimm.hpp:733-843
    void write_abundance() {
        std::ostringstream out;
        out << "taxa_level\ttaxa_id\tlinage\tabundance\tread_count\n";
        uint32_t rnk = considered[1];
        uint32_t parent_rnk = considered[0];

        std::unordered_map<uint32_t, float> parent_abundance;
        std::unordered_map<uint32_t, uint32_t> parent_reads;
        for (auto tc : taxon_count) {
            if (std::get<0>(taxid_name[tc.first]) == parent_rnk) {
                uint32_t glen = 0, nchild = 0;
                for (auto child : taxon_children.at(tc.first)) {
                    glen += refs[child].length;
                    ++nchild;
                }
                glen = glen / nchild;
                (void)glen;
                float ab = float(tc.second) / (matches) * 100;
                parent_abundance[tc.first] = ab;
                parent_reads[tc.first] = tc.second;
            }
        }

        uint32_t count = 0, failed = 0, sum_reads = 0;
        float sum_ab = 0.0f;
        std::unordered_map<uint32_t, float> ab_by_parent;
        std::unordered_map<uint32_t, uint32_t> reads_by_parent;

        for (auto tc : taxon_count) {
            if (std::get<0>(taxid_name[tc.first]) == rnk) {
                uint32_t glen = 0, nchild = 0;
                std::string child_acc = "";
                for (auto child : taxon_children.at(tc.first)) {
                    glen += refs[child].length;  // u32 sum (Q11)
                    child_acc = refs[child].accession;  // ends as the LAST child (Q12)
                    ++nchild;
                }
                glen = glen / nchild;
                std::vector<uint32_t> lin = ac_taxid[child_acc];
                float cov = float(tc.second * avg_read_length) / glen;  // u32 product (Q11)
                float ab = float(tc.second) / (matches) * 100;
                std::string name = std::get<1>(taxid_name[tc.first]);
                uint32_t parent = lin[parent_rnk];
                auto pa = ab_by_parent.find(parent);
                if (pa != ab_by_parent.end())
                    pa->second += ab;
                else
                    ab_by_parent[parent] = ab;
                bump(reads_by_parent, parent, tc.second);
                if (ab < abundance_cut_off || cov < coverage_cut_off() || name == "") {  // Q10
                    ++failed;
                    continue;
                }
                std::string ls = lineage_string(rnk, tc.first);
                out << rank_long(rnk) << "\t" << tc.first << "\t" << ls << "\t";
                out << ab << "\t" << tc.second << "\n";
                sum_ab += ab;
                sum_reads += tc.second;
                ++count;
            }
        }

        for (auto abp : ab_by_parent) {
            uint32_t parent = abp.first;
            float uncl_ab = parent_abundance[parent] - ab_by_parent[parent];
            uint32_t uncl_reads = parent_reads[parent] - reads_by_parent[parent];
            std::string name = std::get<1>(taxid_name[parent]) + "_unclassified";
            if (uncl_ab > abundance_cut_off && name != "_unclassified") {
                std::string ls = lineage_string(parent_rnk, parent) + "|" + rank_short(rnk) + "__" + name;
                out << rank_long(rnk) << "\t" << parent << "*\t" << ls << "\t";
                out << uncl_ab << "\t" << uncl_reads << "\n";
                sum_reads += uncl_reads;
                sum_ab += uncl_ab;
            }
        }

        std::string ls = lineage_string(rnk, 0u);
        out << rank_long(rnk) << "\t" << "0*" << "\t" << ls << "\t";
        out << 100.0 - sum_ab << "\t" << matches - sum_reads << "\n";
        profile_tsv = out.str();
        profile_count = count;
        profile_failed = failed;
        profile_written = true;
    }

    // slimm.hpp:883-943
    void write_raw() {
        std::ostringstream o;
        o << "accesion\ttaxaid\tname\treads_count\tabundance\tuniq1_abundance\tuniq2_abundance\tgenome_length\t"
             "uniq1_reads_count\tuniq2_reads_count\tbins_count\tbins_count(>0)\tuniq1_bins_count(>0)\t"
             "uniq2_bins_count(>0)\tcoverage_depth\tuniq1_coverage_depth\tuniq2_coverage_depth\tcoverage(%)\t"
             "uniq1_coverage(%)\tuniq2_coverage(%)\n";
        for (size_t i = 0; i < refs.size(); ++i) {
            Ref r = refs[i];
            std::string name = std::get<1>(taxid_name[r.taxid]);
            if (name == "") name = "no_name_found";
            o << r.accession << "\t" << r.taxid << "\t" << name << "\t" << r.reads << "\t" << r.abundance << "\t"
              << r.uabundance << "\t" << r.uabundance2 << "\t" << r.length << "\t" << r.uniq << "\t" << r.uniq2 << "\t"
              << r.cov.n << "\t" << r.cov.nonzero() << "\t" << r.ucov.nonzero() << "\t" << r.ucov2.nonzero() << "\t"
              << Ref::depth(r.cov) << "\t" << Ref::depth(r.ucov) << "\t" << Ref::depth(r.ucov2) << "\t" << r.cov_pct()
              << "\t" << r.ucov_pct() << "\t" << r.ucov2_pct() << "\n";
        }
        raw_tsv = o.str();
    }

    // slimm.hpp:846-881
    void write_coverage() {
        std::ostringstream a, b, c;
        for (auto v : valid_refs) {
            Ref r = refs[v];
            a << r.accession;
            b << r.accession;
            c << r.accession;
            for (uint32_t ti : ac_taxid[r.accession]) {
                a << "," << std::get<1>(taxid_name[ti]);
                b << "," << std::get<1>(taxid_name[ti]);
                c << "," << std::get<1>(taxid_name[ti]);
            }
            for (uint32_t k = 0; k < r.cov.n; ++k) {
                a << "," << r.cov.h[k];
                b << "," << r.ucov.h[k];
                c << "," << r.ucov2.h[k];
            }
            a << "\n";
            b << "\n";
            c << "\n";
        }
        cov_csv = a.str();
        ucov_csv = b.str();
        ucov2_csv = c.str();
    }
};

std::vector<std::string> split_blob(const char* blob, uint32_t n) {
    std::vector<std::string> out;
    out.reserve(n);
    const char* p = blob;
    for (uint32_t i = 0; i < n; ++i) {
        const char* e = std::strchr(p, '\n');
        if (!e) {
            out.emplace_back(p);
            p += std::strlen(p);
        } else {
            out.emplace_back(p, e - p);
            p = e + 1;
        }
    }
    return out;
}

}  // namespace

// ---------------------------------------------------------------------------
// Flat C API for ctypes (tests/oracle_binding.py).  Strings are '\n'-joined blobs.
// ---------------------------------------------------------------------------
extern "C" {

void* orc_create() { return new Oracle(); }
void orc_destroy(void* h) { delete static_cast<Oracle*>(h); }

void orc_set_options(void* h, uint32_t bin_width, uint32_t min_reads, float cov_cut_off, float abundance_cut_off,
                     const char* rank) {
    Oracle* o = static_cast<Oracle*>(h);
    o->bin_width = bin_width;
    o->min_reads = min_reads;
    o->cov_cut_off = cov_cut_off;
    o->abundance_cut_off = abundance_cut_off;
    o->set_rank(rank);
}

void orc_db_add_accessions(void* h, uint32_t n, const char* acc_blob, const uint32_t* lineage /*[n*8]*/) {
    Oracle* o = static_cast<Oracle*>(h);
    auto names = split_blob(acc_blob, n);
    for (uint32_t i = 0; i < n; ++i)
        o->ac_taxid[names[i]] = std::vector<uint32_t>(lineage + 8 * i, lineage + 8 * i + 8);
}

void orc_db_add_taxa(void* h, uint32_t n, const uint32_t* taxid, const uint32_t* rank, const char* name_blob) {
    Oracle* o = static_cast<Oracle*>(h);
    auto names = split_blob(name_blob, n);
    for (uint32_t i = 0; i < n; ++i) o->taxid_name[taxid[i]] = std::make_tuple(rank[i], names[i]);
}

// misc.hpp:509-522 get_avg_read_length: floor(sum l_seq / count) over the first <= sample records with l_seq > 0.
uint32_t orc_avg_read_length(const uint32_t* l_seq, uint64_t n, uint32_t sample) {
    uint32_t count = 0, total = 0;
    for (uint64_t i = 0; i < n && count < sample; ++i) {
        if (l_seq[i] == 0) continue;
        total += l_seq[i];
        ++count;
    }
    return count ? total / count : 0;  // the reference divides by zero here; callers must not pass that
}

void orc_reset(void* h) { static_cast<Oracle*>(h)->reset(); }

// One file's worth of work = slimm::get_profiles() minus I/O (slimm.hpp:395-496).
// qname_blob may be NULL: the decimal text of read_key is then the qName.
// Returns 0, or 1 when no mapped record was found (the reference prints a warning and writes nothing).
int orc_run(void* h, uint32_t n_refs, const char* ref_name_blob, const uint32_t* ref_len, uint32_t avg_read_len,
            uint64_t n_records, const char* qname_blob, const uint64_t* read_key, const uint16_t* flag,
            const int32_t* ref_id, const int32_t* begin_pos, int want_raw, int want_cov, double* phase_seconds /*[3] or NULL*/) {
    Oracle* o = static_cast<Oracle*>(h);
    o->init_refs(split_blob(ref_name_blob, n_refs), std::vector<uint32_t>(ref_len, ref_len + n_refs), avg_read_len);
    struct timespec t0, t1, t2, t3;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    if (qname_blob) {
        const char* p = qname_blob;
        for (uint64_t i = 0; i < n_records; ++i) {
            const char* e = std::strchr(p, '\n');
            size_t len = e ? static_cast<size_t>(e - p) : std::strlen(p);
            o->feed(std::string(p, len), flag[i], ref_id[i], begin_pos[i]);
            p += len + (e ? 1 : 0);
        }
    } else {
        for (uint64_t i = 0; i < n_records; ++i)
            o->feed(std::to_string(read_key[i]), flag[i], ref_id[i], begin_pos[i]);
    }
    o->finish_analyze();
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (o->hits == 0) return 1;
    if (o->min_reads == 0) o->min_reads = 1 + ((o->matches - 1) / 10000);  // slimm.hpp:458-459
    o->filter();
    clock_gettime(CLOCK_MONOTONIC, &t2);
    if (want_raw) o->write_raw();
    if (want_cov) o->write_coverage();
    o->lca_count();
    o->write_abundance();
    clock_gettime(CLOCK_MONOTONIC, &t3);
    if (phase_seconds) {
        auto d = [](const timespec& a, const timespec& b) { return (b.tv_sec - a.tv_sec) + 1e-9 * (b.tv_nsec - a.tv_nsec); };
        phase_seconds[0] = d(t0, t1);
        phase_seconds[1] = d(t1, t2);
        phase_seconds[2] = d(t2, t3);
    }
    return 0;
}

// ---- split phases, for the multi-rank tests: a rank runs phase A on its shard, the caller sums the bins of all
// ranks, decides the valid set from the sums, and each rank then filters its own reads against that global set.
int orc_phase_a(void* h, uint32_t n_refs, const char* ref_name_blob, const uint32_t* ref_len, uint32_t avg_read_len,
                uint64_t n_records, const uint64_t* read_key, const uint16_t* flag, const int32_t* ref_id,
                const int32_t* begin_pos) {
    Oracle* o = static_cast<Oracle*>(h);
    o->init_refs(split_blob(ref_name_blob, n_refs), std::vector<uint32_t>(ref_len, ref_len + n_refs), avg_read_len);
    for (uint64_t i = 0; i < n_records; ++i) o->feed(std::to_string(read_key[i]), flag[i], ref_id[i], begin_pos[i]);
    o->finish_analyze();
    return o->hits == 0 ? 1 : 0;
}
// valid[R]: the globally decided valid_ref_ids; runs slimm.hpp:380-391 and :536-557 on this shard's reads only.
void orc_phase_b_with_valid(void* h, const uint8_t* valid) {
    Oracle* o = static_cast<Oracle*>(h);
    o->valid_refs.clear();
    for (uint32_t i = 0; i < o->refs.size(); ++i)
        if (valid[i]) o->valid_refs.insert(i);
    o->apply_valid();
    o->lca_direct();
}

// scalars: [hits, matches, uniq_matches, uniq_hits, uniq_matches2, reference_count, matched_ref_length,
//           failed_by_cov, failed_by_ucov, failed_by_min_read, n_valid, bin_width, min_reads, profile_count, profile_failed]
void orc_get_scalars(void* h, uint32_t* out15) {
    Oracle* o = static_cast<Oracle*>(h);
    uint32_t v[15] = {o->hits, o->matches, o->uniq_matches, o->uniq_hits, o->uniq_matches2, o->reference_count,
                      o->matched_ref_length, o->failed_by_cov, o->failed_by_ucov, o->failed_by_min_read,
                      static_cast<uint32_t>(o->valid_refs.size()), o->bin_width, o->min_reads, o->profile_count,
                      o->profile_failed};
    std::memcpy(out15, v, sizeof(v));
}
void orc_get_cutoffs(void* h, float* out3) {
    Oracle* o = static_cast<Oracle*>(h);
    out3[0] = o->coverage_cut_off();
    out3[1] = o->uniq_coverage_cut_off();
    out3[2] = o->matched_ref_length ? o->expected_coverage() : 0.0f;
}

// per-ref: 8 u32 columns [reads, uniq, uniq2, nbins, nz_cov, nz_ucov, nz_ucov2, valid], then 2 float columns
void orc_get_ref_stats(void* h, uint32_t* u /*[R*8]*/, float* f /*[R*2]*/) {
    Oracle* o = static_cast<Oracle*>(h);
    for (size_t i = 0; i < o->refs.size(); ++i) {
        Ref& r = o->refs[i];
        u[i * 8 + 0] = r.reads;
        u[i * 8 + 1] = r.uniq;
        u[i * 8 + 2] = r.uniq2;
        u[i * 8 + 3] = r.cov.n;
        u[i * 8 + 4] = static_cast<uint32_t>(r.cov.n - std::count(r.cov.h.begin(), r.cov.h.end(), 0u));
        u[i * 8 + 5] = static_cast<uint32_t>(r.ucov.n - std::count(r.ucov.h.begin(), r.ucov.h.end(), 0u));
        u[i * 8 + 6] = static_cast<uint32_t>(r.ucov2.n - std::count(r.ucov2.h.begin(), r.ucov2.h.end(), 0u));
        u[i * 8 + 7] = o->valid_refs.count(static_cast<uint32_t>(i)) ? 1u : 0u;
        f[i * 2 + 0] = r.abundance;
        f[i * 2 + 1] = r.uabundance;
    }
}

uint64_t orc_total_bins(void* h) {
    Oracle* o = static_cast<Oracle*>(h);
    uint64_t s = 0;
    for (auto& r : o->refs) s += r.cov.n;
    return s;
}
// which: 0 cov, 1 uniq_cov, 2 uniq_cov2; bins of ref 0, then ref 1, ... (no padding)
void orc_get_bins(void* h, int which, uint32_t* out) {
    Oracle* o = static_cast<Oracle*>(h);
    uint64_t k = 0;
    for (auto& r : o->refs) {
        const std::vector<uint32_t>& v = which == 0 ? r.cov.h : which == 1 ? r.ucov.h : r.ucov2.h;
        std::memcpy(out + k, v.data(), v.size() * sizeof(uint32_t));
        k += v.size();
    }
}

// taxon counts. stage 0 = after direct LCA hits only (slimm.hpp:536-557), stage 1 = final (after :560-610)
uint32_t orc_taxon_count_size(void* h, int stage) {
    Oracle* o = static_cast<Oracle*>(h);
    return static_cast<uint32_t>((stage ? o->taxon_count : o->lca_direct_count).size());
}
void orc_get_taxon_counts(void* h, int stage, uint32_t* taxid, uint32_t* count) {
    Oracle* o = static_cast<Oracle*>(h);
    uint32_t k = 0;
    for (auto& kv : (stage ? o->taxon_count : o->lca_direct_count)) {
        taxid[k] = kv.first;
        count[k] = kv.second;
        ++k;
    }
}
uint64_t orc_children_pairs_size(void* h, int stage) {
    Oracle* o = static_cast<Oracle*>(h);
    uint64_t s = 0;
    for (auto& kv : (stage ? o->taxon_children : o->lca_direct_children)) s += kv.second.size();
    return s;
}
void orc_get_children_pairs(void* h, int stage, uint32_t* taxid, uint32_t* ref) {
    Oracle* o = static_cast<Oracle*>(h);
    uint64_t k = 0;
    for (auto& kv : (stage ? o->taxon_children : o->lca_direct_children))
        for (uint32_t r : kv.second) {
            taxid[k] = kv.first;
            ref[k] = r;
            ++k;
        }
}

// text outputs: 0 profile, 1 raw, 2 coverage, 3 uniq_coverage, 4 uniq_coverage2
uint64_t orc_text_size(void* h, int which) {
    Oracle* o = static_cast<Oracle*>(h);
    const std::string* s[] = {&o->profile_tsv, &o->raw_tsv, &o->cov_csv, &o->ucov_csv, &o->ucov2_csv};
    return s[which]->size();
}
void orc_get_text(void* h, int which, char* out) {
    Oracle* o = static_cast<Oracle*>(h);
    const std::string* s[] = {&o->profile_tsv, &o->raw_tsv, &o->cov_csv, &o->ucov_csv, &o->ucov2_csv};
    std::memcpy(out, s[which]->data(), s[which]->size());
}

}  // extern "C"
