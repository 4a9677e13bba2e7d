// See host_profile.hpp.  Every block cites the reference lines whose arithmetic it reproduces; the
// data structures are this library's own (dense taxon indices, append-buffer sets), not the reference's.
#include "host_profile.hpp"

#include <algorithm>
#include <charconv>
#include <numeric>
#include <cstdio>
#include <cstring>
#include <unordered_map>

namespace slimm {

static const char* kRankNames[] = {"strain", "species", "genus", "family", "order", "class", "phylum", "superkingdom"};
static const char* kRankShort[] = {"r", "s", "g", "f", "o", "c", "p", "k"};

uint32_t rank_from_string(const std::string& s) {  // misc.hpp:37-48
    for (uint32_t i = 0; i < 8; ++i)
        if (s == kRankNames[i]) return i;
    return 8;
}
std::string rank_long(uint32_t r) { return r < 8 ? kRankNames[r] : "intermidiate"; }  // misc.hpp:51-62
std::string rank_short(uint32_t r) { return r < 8 ? kRankShort[r] : "i"; }            // misc.hpp:64-75

// f(i) for every non-zero a[i], ascending.  The per-file arrays are a few thousand words with a few dozen non-zeros; a
// plain `if (a[i])` loop over them cost ~1 ns per word, which was most of the host time between and after the two
// device phases.
template <typename F>
static inline void for_each_nonzero(const uint32_t* a, uint32_t n, F f) {
    uint32_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t x[4];
        memcpy(x, a + i, 32);
        if ((x[0] | x[1] | x[2] | x[3]) == 0) continue;
        for (uint32_t k = 0; k < 8; ++k)
            if (a[i + k]) f(i + k);
    }
    for (; i < n; ++i)
        if (a[i]) f(i);
}

// misc.hpp:197-216: float32 throughout, sum in input order, ascending sort, partial sums from the top,
// the `i > 0` guard compared as float like the template does for Type = float.
namespace {
// Ascending order of NON-NEGATIVE floats (their bit patterns order like their values): three counting passes of 11 bits.
// The cut-offs' inputs are quotients of counts; a few thousand to tens of thousands of them per file are 5 - 10 x the time
// of this in std::sort, and the device waits for the cut-offs between its two phases.  Returns false (v untouched) when a
// value has its sign bit set or is not a number.
bool sort_non_negative(std::vector<float>& v) {
    const size_t n = v.size();
    static thread_local std::vector<uint32_t> a, b;
    a.resize(n);
    b.resize(n);
    memcpy(a.data(), v.data(), n * 4);
    uint32_t h[3][2048];
    memset(h, 0, sizeof(h));
    uint32_t any = 0;
    for (size_t i = 0; i < n; ++i) {
        const uint32_t x = a[i];
        any |= x | (x > 0x7f800000u ? 0x80000000u : 0u);  // sign bit, or a NaN
        ++h[0][x & 2047u];
        ++h[1][(x >> 11) & 2047u];
        ++h[2][x >> 22];
    }
    if (any & 0x80000000u) return false;
    for (int p = 0; p < 3; ++p) {
        uint32_t s = 0;
        for (int d = 0; d < 2048; ++d) {
            const uint32_t c = h[p][d];
            h[p][d] = s;
            s += c;
        }
    }
    for (size_t i = 0; i < n; ++i) b[h[0][a[i] & 2047u]++] = a[i];
    for (size_t i = 0; i < n; ++i) a[h[1][(b[i] >> 11) & 2047u]++] = b[i];
    for (size_t i = 0; i < n; ++i) b[h[2][a[i] >> 22]++] = a[i];
    memcpy(v.data(), b.data(), n * 4);
    return true;
}
}  // namespace

float quantile_cut_off(std::vector<float> v, float q) {
    if (v.empty()) return 0;
    float total = std::accumulate(v.begin(), v.end(), 0.0f);
    float sub = 0.0f;
    if (v.size() < 256 || !sort_non_negative(v)) std::sort(v.begin(), v.end());
    uint32_t i = static_cast<uint32_t>(v.size() - 1);
    while ((float(sub) / total) < q && i > 0.0f) {
        sub += v[i];
        --i;
    }
    return v[i];
}

void RefSet::materialise() {
    if (!dirty) return;
    std::sort(items.begin(), items.end());
    items.erase(std::unique(items.begin(), items.end()), items.end());
    dirty = false;
}

HostProfile::HostProfile(const HostConfig& cfg) : cfg_(cfg) {
    const uint32_t R = cfg_.n_refs;
    if (cfg_.bin_width == 0) cfg_.bin_width = cfg_.avg_read_len;  // slimm.hpp:412-413
    nbins_.resize(R);
    total_bins_ = 0;
    for (uint32_t i = 0; i < R; ++i) {
        nbins_[i] = cfg_.bin_width ? cfg_.ref_len[i] / cfg_.bin_width + 1 : 0;  // reference_contig.hpp:80
        total_bins_ += nbins_[i];
    }
    // dense taxon index over every taxid the lineage table can produce
    dense_taxid_ = cfg_.lineage;
    std::sort(dense_taxid_.begin(), dense_taxid_.end());
    dense_taxid_.erase(std::unique(dense_taxid_.begin(), dense_taxid_.end()), dense_taxid_.end());
    lin_dense_.resize(cfg_.lineage.size());
    for (size_t k = 0; k < cfg_.lineage.size(); ++k)
        lin_dense_[k] = static_cast<uint32_t>(
            std::lower_bound(dense_taxid_.begin(), dense_taxid_.end(), cfg_.lineage[k]) - dense_taxid_.begin());
    if (!dense_taxid_.empty() && dense_taxid_[0] == 0) zero_dense_ = 0;
    for (uint32_t k = 0; k < cfg_.tax_id.size(); ++k)
        if (cfg_.tax_id[k] == 0) zero_name_ = cfg_.tax_name[k];

    // per-level dense indices for the 16-byte device rows
    rows16_ok_ = true;
    lvl_idx_.assign(static_cast<size_t>(R) * 8, 0);
    lvl_taxon_.clear();
    for (uint32_t lv = 0; lv < kLineageLen; ++lv) {
        std::vector<uint32_t> vals(R);
        for (uint32_t r = 0; r < R; ++r) vals[r] = lin_dense_[static_cast<size_t>(r) * 8 + lv];
        std::sort(vals.begin(), vals.end());
        vals.erase(std::unique(vals.begin(), vals.end()), vals.end());
        lvl_off_[lv] = static_cast<uint32_t>(lvl_taxon_.size());
        if (vals.size() > (lv == 7 ? 32767u : 65535u)) rows16_ok_ = false;
        if (rows16_ok_)
            for (uint32_t r = 0; r < R; ++r)
                lvl_idx_[static_cast<size_t>(r) * 8 + lv] = static_cast<uint16_t>(
                    std::lower_bound(vals.begin(), vals.end(), lin_dense_[static_cast<size_t>(r) * 8 + lv]) - vals.begin());
        lvl_taxon_.insert(lvl_taxon_.end(), vals.begin(), vals.end());
    }
    lvl_off_[8] = static_cast<uint32_t>(lvl_taxon_.size());

    // db.taxid__name: absent taxid reads as (strain_lv, "") -- the reference's operator[] default (Q6)
    std::unordered_map<uint32_t, uint32_t> pos;
    pos.reserve(cfg_.tax_id.size() * 2);
    for (uint32_t k = 0; k < cfg_.tax_id.size(); ++k) pos[cfg_.tax_id[k]] = k;  // later entries win, like map assignment
    const uint32_t T = n_taxa_dense();
    rank_d_.assign(T, 0);
    name_idx_d_.assign(T, -1);
    for (uint32_t d = 0; d < T; ++d) {
        auto it = pos.find(dense_taxid_[d]);
        if (it != pos.end()) {
            rank_d_[d] = cfg_.tax_rank[it->second];
            name_idx_d_[d] = static_cast<int32_t>(it->second);
        }
    }
    // slimm.hpp:498-514 get_considered_ranks
    if (cfg_.rank == "all") {
        for (uint32_t i = 8; i > 0; --i) considered_.push_back(i - 1);
    } else if (cfg_.rank == "superkingdom") {
        considered_.push_back(rank_from_string(cfg_.rank));
    } else {
        considered_.push_back(rank_from_string(cfg_.rank) + 1);
        considered_.push_back(rank_from_string(cfg_.rank));
    }
    reset();
}

const std::string& HostProfile::name_of_dense(uint32_t d) const {
    int32_t k = name_idx_d_[d];
    return k < 0 ? empty_ : cfg_.tax_name[k];
}

void HostProfile::reset() {  // slimm.hpp:167-188
    hits = matches = uniq_matches = uniq_hits = uniq_matches2 = 0;
    reference_count = matched_ref_length = 0;
    failed_by_cov = failed_by_ucov = failed_by_min_read = n_valid = 0;
    profile_count = profile_failed = 0;
    // (min_reads stays: the reference derives it INTO options.min_reads, src/slimm.hpp:458-459, which reset() does not touch -- Q8)
    have_coverage = have_valid = have_partials = have_counts = false;
    profile_.clear();
    profile_ready_ = false;
}

void HostProfile::reset_cutoffs() {  // a fresh `slimm` object: the cached cut-offs and the derived options.min_reads
    cc_cache_ = ucc_cache_ = 0.0f;
    min_reads = cfg_.min_reads;
}

void HostProfile::set_coverage(const uint32_t* rc, const uint32_t* urc, const uint32_t* nzc, const uint32_t* nzu,
                               uint32_t hits_, uint32_t matches_) {
    set_coverage_strided(rc, urc, nzc, nzu, 1, hits_, matches_);
}

// The four per-reference columns with a common stride (4: the device's packed {reads, non-zero cov, unique reads,
// non-zero uniq_cov} rows, read in place).
void HostProfile::set_coverage_strided(const uint32_t* rc, const uint32_t* urc, const uint32_t* nzc, const uint32_t* nzu,
                                       size_t stride, uint32_t hits_, uint32_t matches_) {
    const uint32_t R = cfg_.n_refs;
    reads_count.assign(R, 0u);
    uniq_reads_count.assign(R, 0u);
    nz_cov.assign(R, 0u);
    nz_ucov.assign(R, 0u);
    hits = hits_;
    matches = matches_;
    // references with anything to say (typically a few per cent): every later pass over the references walks this list
    active_.clear();
    if (stride == 4 && nzc == rc + 1 && urc == rc + 2 && nzu == rc + 3) {  // the device's 16-byte rows
        for (uint32_t i = 0; i < R; ++i) {
            uint64_t w[2];
            memcpy(w, rc + 4ull * i, 16);
            if (w[0] | w[1]) active_.push_back(i);
        }
    } else {
        for (uint32_t i = 0; i < R; ++i)
            if (rc[i * stride] | urc[i * stride] | nzc[i * stride] | nzu[i * stride]) active_.push_back(i);
    }
    uint32_t u = 0;
    for (uint32_t i : active_) {
        reads_count[i] = rc[i * stride];
        uniq_reads_count[i] = urc[i * stride];
        nz_cov[i] = nzc[i * stride];
        nz_ucov[i] = nzu[i * stride];
        u += uniq_reads_count[i];
    }
    uniq_matches = u;  // slimm.hpp:225, one per unique read
    uniq_hits = u;     // slimm.hpp:236
    // slimm.hpp:259-302: reference count and matched length here; the abundance floats (read by nobody on the way to the
    // profile, only by the raw-output columns) are computed by abundances() when first asked for
    reference_count = 0;
    matched_ref_length = 0;
    for (uint32_t i : active_) {
        if (reads_count[i] > 0) {
            ++reference_count;
            matched_ref_length += cfg_.ref_len[i];
        }
    }
    abundance_ready_ = false;
    frac_ready_ = false;
    // slimm.hpp:458-459
    if (min_reads == 0 && matches > 0) min_reads = 1 + ((matches - 1) / 10000);
    have_coverage = true;
}

// slimm.hpp:259-302 (two passes per abundance: the float sums run over the references in index order)
void HostProfile::abundances() {
    if (abundance_ready_) return;
    const uint32_t R = cfg_.n_refs;
    abundance.assign(R, 0.0f);
    uniq_abundance.assign(R, 0.0f);
    float total = 0.0f, utotal = 0.0f;
    for (uint32_t i : active_) {  // ascending reference index, like the reference's loops
        if (reads_count[i] > 0) {
            abundance[i] = float(reads_count[i] * 100) / hits;
            total += abundance[i] / cfg_.ref_len[i];
        }
        if (uniq_reads_count[i] > 0) {
            uniq_abundance[i] = float(uniq_reads_count[i] * 100) / uniq_hits;
            utotal += uniq_abundance[i] / cfg_.ref_len[i];
        }
    }
    for (uint32_t i : active_) {
        if (reads_count[i] > 0) abundance[i] = (abundance[i] * 100) / (total * cfg_.ref_len[i]);
        if (uniq_reads_count[i] > 0) uniq_abundance[i] = (uniq_abundance[i] * 100) / (utotal * cfg_.ref_len[i]);
    }
    abundance_ready_ = true;
}

// The two quotients the cut-offs and the validity test are made of -- non-zero bins / bins of cov and of uniq_cov
// (slimm.hpp:331-336, :357-361, :675-680; float(count) / uint32 bins = float / float) -- for the references with any
// non-zero statistic, in the order of active_: one pass, shared by the three of them.
void HostProfile::active_fractions() {
    if (frac_ready_) return;
    const size_t A = active_.size();
    cov_frac_.resize(A);
    ucov_frac_.resize(A);
    const uint32_t* act = active_.data();
    for (size_t k = 0; k < A; ++k) {
        const uint32_t i = act[k];
        const float b = float(nbins_[i]);
        cov_frac_[k] = float(nz_cov[i]) / b;
        ucov_frac_[k] = float(nz_ucov[i]) / b;
    }
    frac_ready_ = true;
}

// the quotients of the references with unique reads (slimm.hpp:333, :677), in reference order
static void fractions_with_unique_reads(const std::vector<uint32_t>& active, const std::vector<uint32_t>& uniq_reads,
                                        const std::vector<float>& frac, std::vector<float>& v) {
    const size_t A = active.size();
    v.resize(A);
    size_t n = 0;
    for (size_t k = 0; k < A; ++k) {  // (no branch: the store is kept when the reference counts)
        v[n] = frac[k];
        n += uniq_reads[active[k]] > 0 ? 1u : 0u;
    }
    v.resize(n);
}

float HostProfile::coverage_cut_off() {  // slimm.hpp:328-344
    if (cc_cache_ == 0.0 && cfg_.cov_cut_off < 1.0) {
        active_fractions();
        std::vector<float> v;
        fractions_with_unique_reads(active_, uniq_reads_count, cov_frac_, v);
        cc_cache_ = quantile_cut_off(std::move(v), cfg_.cov_cut_off);
    }
    return cc_cache_;
}

float HostProfile::uniq_coverage_cut_off() {  // slimm.hpp:672-688
    if (ucc_cache_ == 0.0 && cfg_.cov_cut_off < 1.0) {
        active_fractions();
        std::vector<float> v;
        fractions_with_unique_reads(active_, uniq_reads_count, ucov_frac_, v);
        ucc_cache_ = quantile_cut_off(std::move(v), cfg_.cov_cut_off);
    }
    return ucc_cache_;
}

float HostProfile::expected_coverage() const {  // slimm.hpp:346-349
    return float(cfg_.avg_read_len * matches) / matched_ref_length;
}

void HostProfile::compute_valid() {  // slimm.hpp:353-378
    const uint32_t R = cfg_.n_refs;
    valid.assign(R, 0);
    // The reference re-evaluates the (cached) cut-off getters inside the loop; their value cannot change while it
    // runs, so they are read once here.
    const float cc = coverage_cut_off();
    const float ucc = uniq_coverage_cut_off();
    active_fractions();
    // One pass over the references with any statistic, no data-dependent branch (the device waits for this between its two
    // phases; with 20 k references the branchy form took 76 us of the 190 the host spends there): a reference with reads
    // is valid when both quotients reach their cut-offs (:357-361), otherwise it counts for every test it fails (:364-375).
    const size_t A = active_.size();
    valid_list_.resize(A);
    uint32_t nv = 0, f_cov = 0, f_ucov = 0, f_min = 0;
    const uint32_t* act = active_.data();
    const float* cf = cov_frac_.data();
    const float* uf = ucov_frac_.data();
    uint32_t* vl = valid_list_.data();
    for (size_t k = 0; k < A; ++k) {
        const uint32_t i = act[k];
        const uint32_t rc = reads_count[i];
        const uint32_t has = rc != 0u, cov_ok = cf[k] >= cc, ucov_ok = uf[k] >= ucc;
        const uint32_t ok = has & cov_ok & ucov_ok, failed = has & (ok ^ 1u);
        valid[i] = static_cast<uint8_t>(ok);
        vl[nv] = i;
        nv += ok;
        f_ucov += failed & (ucov_ok ^ 1u);
        f_min += failed & (rc < min_reads ? 1u : 0u);
        f_cov += failed & (cov_ok ^ 1u);
    }
    valid_list_.resize(nv);
    n_valid = nv;
    failed_by_cov = f_cov;
    failed_by_ucov = f_ucov;
    failed_by_min_read = f_min;
    have_valid = true;
}

void HostProfile::set_partials(const uint32_t* u2, const uint32_t* lca, const uint32_t* marks, const uint64_t* pairs,
                               uint32_t n_pairs) {
    set_partials_rows(u2, 1, lca, marks, pairs, n_pairs);
}

// the same with uniq_reads_count2[r] = u2[r * stride] (the device's packed rows, read in place) and, for stride > 1,
// the non-zero uniq_cov2 counts in the next column
void HostProfile::set_partials_rows(const uint32_t* u2, size_t stride, const uint32_t* lca, const uint32_t* marks,
                                    const uint64_t* pairs, uint32_t n_pairs) {
    const uint32_t R = cfg_.n_refs, T = n_taxa_dense();
    if (stride == 1) {
        uniq_reads_count2.assign(u2, u2 + R);
    } else {
        uniq_reads_count2.resize(R);
        nz_ucov2_.resize(R);
        for (uint32_t i = 0; i < R; ++i) {
            uniq_reads_count2[i] = u2[i * stride];
            nz_ucov2_[i] = u2[i * stride + 1];
        }
    }
    lca_count_.assign(lca, lca + T);
    marks_.assign(marks, marks + R);
    pairs_.assign(pairs, pairs + n_pairs);
    uint32_t s = 0;
    for (uint32_t i = 0; i < R; ++i) s += uniq_reads_count2[i];
    uniq_matches2 = s;  // slimm.hpp:387
    have_partials = true;
    have_counts = false;
}

// slimm.hpp:533-611.  Step 1 (per-read LCA, :536-557) arrives from the device as
//   lca_count_[t]        number of multi-target reads whose LCA is dense taxon t
//   marks_[ref] bit lv   ref belongs to a read whose refs first agree at level lv  -> children[lineage[ref][lv]] has ref
//   pairs_               (t << 32 | ref) for reads that agree at no level (Q4)     -> children[t] has ref
void HostProfile::propagate() {
    const uint32_t R = cfg_.n_refs, T = n_taxa_dense();
    if (count_.size() != T) {
        count_.assign(T, 0);
        has_count_.assign(T, 0);
        kids_.assign(T, RefSet());
    } else {  // reuse the allocations of the previous file: only what was touched is cleared
        for (uint32_t t : touched_) {
            count_[t] = 0;
            has_count_[t] = 0;
            RefSet& k = kids_[t];
            k.items.clear();
            k.mn = 0xffffffffu;
            k.mx = 0;
            k.dirty = false;
            k.present = false;
        }
    }
    touched_.clear();
    auto touch = [&](uint32_t t) {
        if (!has_count_[t] && !kids_[t].present) touched_.push_back(t);
    };
    std::vector<uint32_t>& order = order_scratch_;  // taxa with direct hits, ascending
    order.clear();
    for_each_nonzero(lca_count_.data(), T, [&](uint32_t t) {
        touch(t);
        count_[t] = lca_count_[t];
        has_count_[t] = 1;
        order.push_back(t);
    });
    for_each_nonzero(marks_.data(), R, [&](uint32_t r) {
        uint32_t m = marks_[r] & 0xffu;
        while (m) {
            uint32_t lv = static_cast<uint32_t>(__builtin_ctz(m));
            m &= m - 1;
            const uint32_t t = lin_dense_[r * 8 + lv];
            touch(t);
            kids_[t].add(r);
        }
    });
    for (uint64_t p : pairs_) {
        const uint32_t t = static_cast<uint32_t>(p >> 32);
        touch(t);
        kids_[t].add(static_cast<uint32_t>(p));
    }

    // Step 2 (:560-586): every taxon with direct hits hands its count (snapshot value) and its children (live) to the
    // ranks above its own along the lineage of its smallest child.  The reference walks an unordered_map; any order
    // gives the same result in a consistent tree.  Here: lower ranks first, then ascending taxid.
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return rank_d_[a] < rank_d_[b]; });
    std::vector<uint32_t> ids;
    for (uint32_t t : order) {
        uint32_t rnk = rank_d_[t];
        RefSet& src = kids_[t];
        if (!src.present || src.items.empty()) continue;  // cannot happen: a counted LCA taxon has children
        src.materialise();
        ids = src.items;
        const uint32_t* lin = &lin_dense_[static_cast<size_t>(src.mn) * 8];
        uint32_t c = lca_count_[t];
        for (uint32_t j = rnk + 1; j < kLineageLen; ++j) {
            uint32_t u = lin[j];
            touch(u);
            count_[u] += c;
            has_count_[u] = 1;
            if (u != t) kids_[u].add_all(ids);
        }
    }
    // Step 3 (:589-610): unique reads of each reference go to lineage slots 1..7 (slot 0 is skipped, Q9), together with
    // the reference itself and whatever children its slot-0 taxon currently has.
    for_each_nonzero(uniq_reads_count2.data(), R, [&](uint32_t i) {
        const uint32_t u2 = uniq_reads_count2[i];
        const uint32_t* lin = &lin_dense_[static_cast<size_t>(i) * 8];
        touch(lin[0]);
        RefSet& s0 = kids_[lin[0]];
        s0.present = true;  // operator[] creates the entry
        s0.materialise();
        ids = s0.items;
        for (uint32_t j = 1; j < kLineageLen; ++j) {
            uint32_t u = lin[j];
            touch(u);
            count_[u] += u2;
            has_count_[u] = 1;
            kids_[u].add(i);
            kids_[u].add_all(ids);
        }
    });
    have_counts = true;
    profile_ready_ = false;
}

void HostProfile::taxon_counts(int stage, std::vector<uint32_t>& taxid, std::vector<uint32_t>& count) {
    taxid.clear();
    count.clear();
    const uint32_t T = n_taxa_dense();
    for (uint32_t t = 0; t < T; ++t) {
        if (stage == 0 ? (lca_count_[t] != 0) : (has_count_[t] != 0)) {
            taxid.push_back(dense_taxid_[t]);
            count.push_back(stage == 0 ? lca_count_[t] : count_[t]);
        }
    }
}

void HostProfile::children_pairs(int stage, std::vector<uint32_t>& taxid, std::vector<uint32_t>& ref) {
    taxid.clear();
    ref.clear();
    if (stage == 0) {  // rebuilt from what the device reported: (ref, level) marks and no-agreement pairs
        std::vector<uint64_t> pr(pairs_);
        for (uint32_t r = 0; r < cfg_.n_refs; ++r) {
            uint32_t m = marks_[r] & 0xffu;
            while (m) {
                uint32_t lv = static_cast<uint32_t>(__builtin_ctz(m));
                m &= m - 1;
                pr.push_back((static_cast<uint64_t>(lin_dense_[r * 8 + lv]) << 32) | r);
            }
        }
        std::sort(pr.begin(), pr.end());
        pr.erase(std::unique(pr.begin(), pr.end()), pr.end());
        for (uint64_t p : pr) {
            taxid.push_back(dense_taxid_[static_cast<uint32_t>(p >> 32)]);
            ref.push_back(static_cast<uint32_t>(p));
        }
        return;
    }
    for (uint32_t t = 0; t < n_taxa_dense(); ++t) {
        if (!kids_[t].present) continue;
        kids_[t].materialise();
        for (uint32_t r : kids_[t].items) {
            taxid.push_back(dense_taxid_[t]);
            ref.push_back(r);
        }
    }
}

// slimm.hpp:690-710: k__..|p__..|...|<rank>__name, missing names -> unknown_<rank>
// The lineage text of a reference's row down to rank rnk depends on the database only: built once per (reference, rank)
// and kept for the following files (40 profile rows x 7 name lookups and appends were a third of the profile text's time).
void HostProfile::append_lineage_of_ref(std::string& s, uint32_t rnk, uint32_t ref) {
    std::vector<std::string>& cache = lineage_text_[rnk < 9 ? rnk : 8];
    if (cache.empty()) cache.resize(cfg_.n_refs);
    std::string& t = cache[ref];
    if (t.empty()) append_lineage(t, rnk, &lin_dense_[static_cast<size_t>(ref) * 8], false);
    s += t;
}

void HostProfile::append_lineage(std::string& s, uint32_t rnk, const uint32_t* lin, bool all_zero) {
    for (uint32_t i = kLineageLen; i-- > rnk;) {  // the reference prepends rank by rank; same text, built front to back
        s += kRankShort[i < 8 ? i : 0];
        s += "__";
        const std::string& n = all_zero ? zero_name_ : name_of_dense(lin[i]);  // db.taxid__name[0] for the zero lineage
        if (n.empty()) {
            s += "unknown_";
            s += kRankNames[i < 8 ? i : 0];
        } else {
            s += n;
        }
        if (i != rnk) s += '|';
    }
}

// Number formatting: the reference streams floats / doubles into an ofstream with default flags, i.e. "%.6g".
// std::to_chars(general, 6) is defined as what printf("%.6g") prints in the C locale (and is 3x faster than snprintf,
// which was most of the 25 us the profile text took).
static void put_g(std::string& out, double v) {
    char buf[48];
    const auto r = std::to_chars(buf, buf + sizeof(buf), v, std::chars_format::general, 6);
    out.append(buf, static_cast<size_t>(r.ptr - buf));
}
static void put_u(std::string& out, uint32_t v) {
    char buf[16];
    const auto r = std::to_chars(buf, buf + sizeof(buf), v);
    out.append(buf, static_cast<size_t>(r.ptr - buf));
}

// slimm.hpp:733-843
const std::string& HostProfile::write_abundance() {
    if (profile_ready_) return profile_;
    std::string out;
    out.reserve(16384);
    out += "taxa_level\ttaxa_id\tlinage\tabundance\tread_count\n";
    const uint32_t T = n_taxa_dense();
    const uint32_t rnk = considered_.size() > 1 ? considered_[1] : considered_[0];
    const uint32_t parent_rnk = considered_[0];
    const char* rank_name = rnk < 8 ? kRankNames[rnk] : "intermidiate";

    // per-parent accumulators on the dense taxon index (the reference uses four unordered_maps keyed by taxid);
    // only taxa with an entry in taxon_id__read_count can matter, so the walks go over `touched_`
    if (pa_ab_.size() != T) {
        pa_ab_.assign(T, 0.0f);
        sum_ab_.assign(T, 0.0f);
        pa_rd_.assign(T, 0u);
        sum_rd_.assign(T, 0u);
        pa_seen_.assign(T, 0);
    }
    std::vector<uint32_t>& order = order_scratch_;
    order.clear();
    for (uint32_t t : touched_)
        if (has_count_[t]) order.push_back(t);
    std::sort(order.begin(), order.end());  // rows in ascending taxid order
    parents_scratch_.clear();

    for (uint32_t t : order) {  // :747-765 statistics of the upper level
        if (rank_d_[t] != parent_rnk) continue;
        pa_ab_[t] = float(count_[t]) / (matches)*100;
        pa_rd_[t] = count_[t];
    }

    uint32_t count = 0, failed = 0, sum_reads = 0;
    float sum_ab = 0.0f;
    const float cc = coverage_cut_off();

    for (uint32_t t : order) {  // :776-813
        if (rank_d_[t] != rnk) continue;
        RefSet& k = kids_[t];
        k.materialise();
        uint32_t glen = 0, nchild = 0;
        for (uint32_t child : k.items) {
            glen += cfg_.ref_len[child];  // u32 sum, wraps (Q11)
            ++nchild;
        }
        glen = nchild ? glen / nchild : 0;
        const uint32_t* lin_last = &lin_dense_[static_cast<size_t>(k.mx) * 8];  // lineage of the LAST child (Q12)
        float cov = float(count_[t] * cfg_.avg_read_len) / glen;                // u32 product, wraps (Q11)
        float ab = float(count_[t]) / (matches)*100;
        const std::string& name = name_of_dense(t);
        if (parent_rnk < kLineageLen) {
            const uint32_t parent = lin_last[parent_rnk];
            if (!pa_seen_[parent]) {
                pa_seen_[parent] = 1;
                parents_scratch_.push_back(parent);
            }
            sum_ab_[parent] += ab;
            sum_rd_[parent] += count_[t];
        }
        if (ab < cfg_.abundance_cut_off || cov < cc || name.empty()) {  // depth vs fraction (Q10)
            ++failed;
            continue;
        }
        // get_lineage_string(rank, taxid): lineage of the FIRST child, all zeros for taxid 0 (:712-730)
        const bool zero = dense_taxid_[t] == 0;
        const uint32_t* lin_first = &lin_dense_[static_cast<size_t>(k.mn) * 8];
        out += rank_name;
        out += '\t';
        put_u(out, dense_taxid_[t]);
        out += '\t';
        if (zero)
            append_lineage(out, rnk, lin_first, true);
        else
            append_lineage_of_ref(out, rnk, k.mn);
        out += '\t';
        put_g(out, ab);
        out += '\t';
        put_u(out, count_[t]);
        out += '\n';
        sum_ab += ab;
        sum_reads += count_[t];
        ++count;
    }

    std::sort(parents_scratch_.begin(), parents_scratch_.end());
    for (uint32_t parent : parents_scratch_) {  // :816-831 unclassified rows per parent
        // a parent that is not a counted taxon of the parent rank reads as 0 from the reference's maps
        const float uncl_ab = pa_ab_[parent] - sum_ab_[parent];
        const uint32_t uncl_reads = pa_rd_[parent] - sum_rd_[parent];
        const std::string& pname = name_of_dense(parent);
        if (uncl_ab > cfg_.abundance_cut_off && !pname.empty()) {  // name + "_unclassified" != "_unclassified"
            bool zero = dense_taxid_[parent] == 0;
            RefSet& k = kids_[parent];
            if (k.items.empty()) zero = true;  // the reference would throw from .at(); unreachable with a counted parent
            const uint32_t* lin_first = zero ? nullptr : &lin_dense_[static_cast<size_t>(k.mn) * 8];
            out += rank_name;
            out += '\t';
            put_u(out, dense_taxid_[parent]);
            out += "*\t";
            if (zero)
                append_lineage(out, parent_rnk, lin_first, true);
            else
                append_lineage_of_ref(out, parent_rnk, k.mn);
            out += '|';
            out += kRankShort[rnk < 8 ? rnk : 0];
            out += "__";
            out += pname;
            out += "_unclassified\t";
            put_g(out, uncl_ab);
            out += '\t';
            put_u(out, uncl_reads);
            out += '\n';
            sum_reads += uncl_reads;
            sum_ab += uncl_ab;
        }
    }
    // leave the accumulators clean for the next file
    for (uint32_t t : order) {
        pa_ab_[t] = 0.0f;
        pa_rd_[t] = 0u;
    }
    for (uint32_t parent : parents_scratch_) {
        sum_ab_[parent] = 0.0f;
        sum_rd_[parent] = 0u;
        pa_seen_[parent] = 0;
    }

    out += rank_name;  // :833-835
    out += "\t0*\t";
    append_lineage(out, rnk, nullptr, true);
    out += '\t';
    put_g(out, 100.0 - sum_ab);
    out += '\t';
    put_u(out, matches - sum_reads);
    out += '\n';
    profile_count = count;
    profile_failed = failed;
    profile_.swap(out);
    profile_ready_ = true;
    return profile_;
}

}  // namespace slimm
