// Host-side scalar glue of the alignment-to-profile path: everything between the integer arrays the
// HIP kernels produce and the final profile text.  No GPU dependency (plain C++17), so it is unit-testable
// on a CPU-only machine.
//
// What lives here and why:
//   a6  abundance floats            reference src/slimm.hpp:259-302   (float32, sequential over refs)
//   a8  quantile cut-offs           src/misc.hpp:197-216, src/slimm.hpp:328-344, 672-688 (float32, order-sensitive)
//   a9  valid set + failure stats   src/slimm.hpp:353-378
//   a12 steps 2 and 3 (propagation) src/slimm.hpp:560-610
//   a13 write_abundance             src/slimm.hpp:690-843
// These are O(refs) / O(taxa) scalar loops whose float32 operation order decides integer results, so they stay on
// the host in the reference's order; the kernels only feed them per-reference integers.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace slimm {

constexpr uint32_t kLineageLen = 8;

struct HostConfig {
    uint32_t n_refs = 0;
    std::vector<uint32_t> ref_len;   // [R]
    std::vector<uint32_t> lineage;   // [R*8] taxids
    uint32_t bin_width = 0, avg_read_len = 0, min_reads = 0;
    float cov_cut_off = 0.95f, abundance_cut_off = 0.01f;
    std::string rank = "species";
    std::vector<uint32_t> tax_id, tax_rank;
    std::vector<std::string> tax_name;
};

// Sorted set of reference ids kept as an append buffer: unions append, readers materialise (sort + unique).
struct RefSet {
    std::vector<uint32_t> items;
    uint32_t mn = 0xffffffffu, mx = 0;
    bool dirty = false;
    bool present = false;  // an entry exists in taxon_id__children
    void add(uint32_t r) {
        items.push_back(r);
        if (r < mn) mn = r;
        if (r > mx) mx = r;
        dirty = true;
        present = true;
    }
    void add_all(const std::vector<uint32_t>& v) {
        present = true;
        if (v.empty()) return;
        items.insert(items.end(), v.begin(), v.end());
        // v is materialised (sorted) whenever it comes from another RefSet
        for (uint32_t r : v) {
            if (r < mn) mn = r;
            if (r > mx) mx = r;
        }
        dirty = true;
    }
    void materialise();
};

class HostProfile {
public:
    explicit HostProfile(const HostConfig& cfg);

    // ---- static tables derived from the configuration ----
    const HostConfig& config() const { return cfg_; }
    uint32_t n_refs() const { return cfg_.n_refs; }
    uint32_t n_taxa_dense() const { return static_cast<uint32_t>(dense_taxid_.size()); }
    const std::vector<uint32_t>& dense_taxid() const { return dense_taxid_; }  // ascending, unique
    const std::vector<uint32_t>& lineage_dense() const { return lin_dense_; }  // [R*8] indices into dense_taxid
    const std::vector<uint32_t>& nbins() const { return nbins_; }              // len/W + 1 (reference_contig.hpp:80)
    // Compact lineage rows for the device LCA: per level a dense 16-bit index (level 7: 15 bits, the top bit is the
    // per-run valid flag).  rows16_ok() is false when a level has too many distinct taxids for that.
    bool rows16_ok() const { return rows16_ok_; }
    const std::vector<uint16_t>& level_index() const { return lvl_idx_; }      // [R*8]
    const std::vector<uint32_t>& level_taxon() const { return lvl_taxon_; }    // concatenated per level: dense taxon
    const uint32_t* level_offset() const { return lvl_off_; }                  // [9]
    uint32_t bin_width() const { return cfg_.bin_width; }
    uint64_t total_bins() const { return total_bins_; }

    // ---- per-file state ----
    void reset();          // slimm::reset(): keeps the cut-off caches (Q8)
    void reset_cutoffs();  // a fresh object
    void get_cutoff_cache(float& cc, float& ucc) const { cc = cc_cache_; ucc = ucc_cache_; }
    void set_cutoff_cache(float cc, float ucc) { cc_cache_ = cc; ucc_cache_ = ucc; }

    // phase A results (per reference) -> a6 statistics
    void set_coverage(const uint32_t* reads_count, const uint32_t* uniq_reads_count, const uint32_t* nz_cov,
                      const uint32_t* nz_uniq_cov, uint32_t hits, uint32_t matches);
    void set_coverage_strided(const uint32_t* reads_count, const uint32_t* uniq_reads_count, const uint32_t* nz_cov,
                              const uint32_t* nz_uniq_cov, size_t stride, uint32_t hits, uint32_t matches);
    // a8 + a9
    void compute_valid();
    // phase B/C(1) results
    void set_partials(const uint32_t* uniq_reads_count2, const uint32_t* lca_count, const uint32_t* level_marks,
                      const uint64_t* pairs, uint32_t n_pairs);
    void set_partials_rows(const uint32_t* u2_rows, size_t stride, const uint32_t* lca_count, const uint32_t* level_marks,
                           const uint64_t* pairs, uint32_t n_pairs);
    void set_nz_uniq_cov2(const uint32_t* nz) { nz_ucov2_.assign(nz, nz + cfg_.n_refs); }
    const std::vector<uint32_t>& lca_count() const { return lca_count_; }
    const std::vector<uint32_t>& level_marks() const { return marks_; }
    const std::vector<uint64_t>& pairs() const { return pairs_; }
    // a12 steps 2,3
    void propagate();
    // a13
    const std::string& write_abundance();

    float coverage_cut_off();
    float uniq_coverage_cut_off();
    float expected_coverage() const;

    // results
    uint32_t hits = 0, matches = 0, uniq_matches = 0, uniq_hits = 0, uniq_matches2 = 0;
    uint32_t reference_count = 0, matched_ref_length = 0;
    uint32_t failed_by_cov = 0, failed_by_ucov = 0, failed_by_min_read = 0, n_valid = 0;
    uint32_t min_reads = 0;
    uint32_t profile_count = 0, profile_failed = 0;
    std::vector<uint32_t> reads_count, uniq_reads_count, uniq_reads_count2, nz_cov, nz_ucov;
    std::vector<uint8_t> valid;
    const std::vector<uint32_t>& valid_list() const { return valid_list_; }  // the set bits of `valid`, ascending
    std::vector<float> abundance, uniq_abundance;  // valid after abundances()
    void abundances();
    const std::vector<uint32_t>& nz_uniq_cov2() const { return nz_ucov2_; }

    // taxon counts / children.  stage 0 = direct LCA hits, stage 1 = after propagation
    void taxon_counts(int stage, std::vector<uint32_t>& taxid, std::vector<uint32_t>& count);
    void children_pairs(int stage, std::vector<uint32_t>& taxid, std::vector<uint32_t>& ref);

    bool have_coverage = false, have_valid = false, have_partials = false, have_counts = false;

private:
    uint32_t rank_of_dense(uint32_t d) const { return rank_d_[d]; }
    const std::string& name_of_dense(uint32_t d) const;
    void append_lineage(std::string& out, uint32_t rnk, const uint32_t* lin_dense_row, bool all_zero);
    void append_lineage_of_ref(std::string& out, uint32_t rnk, uint32_t ref);
    std::vector<std::string> lineage_text_[9];  // [rank][reference]: lineage text, filled on first use

    HostConfig cfg_;
    std::vector<uint32_t> dense_taxid_, lin_dense_, nbins_, rank_d_;
    std::vector<uint16_t> lvl_idx_;
    std::vector<uint32_t> lvl_taxon_;
    uint32_t lvl_off_[9] = {0};
    bool rows16_ok_ = false;
    std::vector<int32_t> name_idx_d_;  // index into cfg_.tax_name or -1
    uint64_t total_bins_ = 0;
    uint32_t zero_dense_ = 0xffffffffu;  // dense index of taxid 0 if present
    std::vector<uint32_t> considered_;   // slimm.hpp:498-514

    float cc_cache_ = 0.0f, ucc_cache_ = 0.0f;

    std::vector<uint32_t> nz_ucov2_;
    std::vector<uint32_t> active_;      // references with a non-zero statistic, ascending (set_coverage)
    std::vector<uint32_t> valid_list_;
    std::vector<float> cov_frac_, ucov_frac_;  // active_fractions(): per entry of active_
    bool frac_ready_ = false;
    void active_fractions();
    // partials
    std::vector<uint32_t> lca_count_, marks_;
    std::vector<uint64_t> pairs_;
    // counts after each stage
    std::vector<uint32_t> count_;         // [T]
    std::vector<uint8_t> has_count_;      // [T] membership in taxon_id__read_count
    std::vector<RefSet> kids_;            // [T]
    std::vector<float> pa_ab_, sum_ab_;   // write_abundance scratch, dense taxon index, kept zeroed between calls
    std::vector<uint32_t> pa_rd_, sum_rd_, order_scratch_, parents_scratch_;
    std::vector<uint8_t> pa_seen_;
    std::vector<uint32_t> touched_;       // taxa whose count / children entry exists (cleared cheaply on the next file)
    std::string profile_;
    bool profile_ready_ = false;
    bool abundance_ready_ = false;
    std::string empty_, zero_name_;
};

float quantile_cut_off(std::vector<float> v, float q);
uint32_t rank_from_string(const std::string& s);
std::string rank_long(uint32_t r);
std::string rank_short(uint32_t r);

}  // namespace slimm
