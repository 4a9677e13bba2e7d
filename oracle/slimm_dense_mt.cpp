// ============================================================================
// slimm_dense_mt.cpp -- TEST INFRASTRUCTURE / CPU BASELINE ONLY.  NOT PART OF THE PRODUCT.
//
// What a careful all-core CPU implementation of SLIMM's alignment-to-profile hot path looks like: the reference's
// algorithm (src/slimm.hpp:194-303 analyze_alignments, :328-392 cut-offs + filter_alignments, :516-557 LCA of the reads
// that keep several references; src/read_stat.hpp:98-135) restated over DENSE integer arrays and run on every host
// core, where slimm_oracle.cpp follows the reference container by container on one thread.  It is the fair "what
// would the host do" number of bench.py's cpu_baseline_mt leg: the reference itself is single-threaded, so this is
// faster than anything the reference does, by design.  Checked against slimm_oracle.cpp in tests/test_dense_mt.py.
// Only tests/ and bench.py's cpu_baseline legs may load it; the product (slimm_amd/, libslimm_hip.so) never does.
//
// PARITY PIN STATUS: as slimm_oracle.cpp (PARITY UNPINNED beyond the two known-answer micro-cases of SURVEY.md
// Appendix C; the reference cannot be built in this image).
//
// Input: records grouped by read name (all records of a qName adjacent -- mapper output), as the product's
// SLIMM_ORDER_GROUPED.  Threads take contiguous slices of the stream cut at qName-run boundaries; histogram bins are
// shared arrays updated with relaxed atomic adds; per-reference counters and per-taxon LCA counts are private to a
// thread and merged at the end (the few abundant references of a sample would be one contended cache line otherwise).
// Scope of dmt_run*: phases A, B and C(1) -- everything that touches records or reads.  The scalar tail of the path
// (propagation of the counts up the lineages, src/slimm.hpp:560-610, and the rows of the profile, :733-843) is
// dmt_profile below: single-threaded, written from the reference with the reference's containers, and given the direct
// LCA counts + the (taxon, reference) pairs dmt_run3 collects it makes the full-size tests' comparison of the propagated
// counts and the profile an independent one (checked against slimm_oracle.cpp in tests/test_dense_mt.py).
// ============================================================================
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <set>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace {

constexpr uint64_t kKeyMask = (1ull << 62) - 1;

struct Target {
    uint32_t ref, gbin;
};

// what phase A leaves behind per thread for phase B: the reads' target lists, back to back
struct alignas(128) Shard {  // (a cache line pair of its own: the threads' counters never share one)
    std::vector<Target> targets;
    std::vector<uint32_t> read_end;  // end of every read's targets in `targets`
    // per-reference counters are private to the thread and summed afterwards: the few abundant references of a sample
    // would otherwise be one contended cache line for all threads
    std::vector<uint32_t> reads_count, uniq_count, uniq_count2;
    uint64_t hits = 0, reads = 0, uniq_reads = 0, uniq_reads2 = 0;
    std::unordered_map<uint32_t, uint32_t> lca;
    std::unordered_set<uint64_t> pairs;   // (LCA taxon << 32 | reference) of the reads that keep several references (dmt_run3)
};

inline uint32_t mate_of(uint16_t flag) { return (flag & 0x40) ? 1u : ((flag & 0x80) ? 2u : 0u); }  // slimm.hpp:205-208

// misc.hpp:197-216 get_quantile_cut_off<float>
float quantile_cut_off(std::vector<float> v, float q) {
    if (v.empty()) return 0;
    float total = std::accumulate(v.begin(), v.end(), 0.0f);
    float sub = 0.0f;
    std::sort(v.begin(), v.end());
    uint32_t i = static_cast<uint32_t>(v.size() - 1);
    while ((float(sub) / total) < q && i > 0.0f) {
        sub += v[i];
        --i;
    }
    return v[i];
}

inline void add32(uint32_t* p, uint32_t v) { __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }

template <typename F>
void parallel(unsigned n_threads, F f) {
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < n_threads; ++t) pool.emplace_back(f, t);
    f(0u);
    for (auto& th : pool) th.join();
}

// sum_i a[i] * ((i + 1) * 0x9E3779B97F4A7C15) mod 2^64: every bin with a weight of its own, so two arrays that differ in
// one bin, or hold the same counts at other positions, differ in the sum
uint64_t bin_checksum(const uint32_t* a, uint64_t n, unsigned n_threads) {
    std::vector<uint64_t> part(n_threads, 0);
    parallel(n_threads, [&](unsigned t) {
        uint64_t s = 0;
        for (uint64_t i = n * t / n_threads, e = n * (t + 1) / n_threads; i < e; ++i)
            s += static_cast<uint64_t>(a[i]) * ((i + 1) * 0x9E3779B97F4A7C15ull);
        part[t] = s;
    });
    uint64_t s = 0;
    for (uint64_t p : part) s += p;
    return s;
}

double seconds_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace

extern "C" {

// ref_cols [R * 5]: reads_count, uniq_reads_count, nz_cov, nz_uniq_cov, uniq_reads_count2
// scalars  [8]    : hits, matches, uniq_matches, uniq_matches2, n_valid, total bins, 0, 0
// lca      up to lca_cap (taxid, count) pairs of the reads that keep several references (src/slimm.hpp:536-557)
// seconds  [3]    : phase A (records -> histograms + statistics), cut-offs + phase B, LCA merge
// returns 0, 1 when no record is mapped (the reference's early return), -1 on bad arguments
// dmt_run2 also hands out what the full-size parity tests compare with slimm_get_bins:
//   bins_out  [3]: optional destinations (each total-bins words, any may be null) for cov, uniq_cov, uniq_cov2
//                  (reference_contig.hpp:110-112), references back to back without padding
//   checksums [3]: position-weighted 64-bit sums of the same three arrays, sum_i a[i] * ((i + 1) * 0x9E3779B97F4A7C15)
//                  mod 2^64 (bin_checksum below; tests/helpers.py computes the same over slimm_get_bins)
// dmt_run3 also collects (not in any timed phase's favour: only when pairs_out is given) the contributing children of step 1
// (src/slimm.hpp:555): the distinct (LCA taxon << 32 | reference) pairs, up to pair_cap of them, *n_pairs = how many exist;
// cutoffs_out [2] = the two float cut-offs
int dmt_run3(const uint64_t* key, const uint16_t* flag, const int32_t* ref, const int32_t* pos, uint64_t n, uint32_t n_refs,
             const uint32_t* ref_len, const uint32_t* lineage /*[n_refs * 8]*/, uint32_t avg_read_len, uint32_t bin_width,
             float cov_cut_off, uint32_t n_threads, uint32_t* ref_cols, uint64_t* scalars, uint32_t* lca_taxid,
             uint32_t* lca_count, uint32_t lca_cap, uint32_t* n_lca, double* seconds, uint32_t* const* bins_out,
             uint64_t* checksums, uint64_t* pairs_out, uint64_t pair_cap, uint64_t* n_pairs, float* cutoffs_out) {
    if (!n_refs || !ref_len || !lineage || !ref_cols || !scalars || !seconds) return -1;
    if (bin_width == 0) bin_width = avg_read_len;  // slimm.hpp:412-413
    if (bin_width == 0) return -1;
    n_threads = std::max(1u, n_threads);
    const uint32_t R = n_refs;
    const uint32_t half = avg_read_len / 2;
    // bins of every reference back to back (reference_contig.hpp:80: len / width + 1 bins)
    std::vector<uint64_t> bin_off(R + 1, 0);
    for (uint32_t r = 0; r < R; ++r) bin_off[r + 1] = bin_off[r] + ref_len[r] / bin_width + 1;
    const uint64_t B = bin_off[R];
    std::vector<uint32_t> cov(B, 0), ucov(B, 0), ucov2(B, 0);
    std::vector<uint32_t> reads_count(R, 0), uniq_count(R, 0), uniq_count2(R, 0), nz_cov(R, 0), nz_ucov(R, 0);
    std::vector<Shard> shard(n_threads);
    // contiguous cuts at qName-run boundaries
    std::vector<uint64_t> cut(n_threads + 1, n);
    cut[0] = 0;
    for (unsigned t = 1; t < n_threads; ++t) {
        uint64_t c = n * t / n_threads;
        while (c > 0 && c < n && ((key[c] ^ key[c - 1]) & kKeyMask) == 0) ++c;
        cut[t] = std::max(c, cut[t - 1]);
    }

    // ---------------------------------------------------------------- phase A (slimm.hpp:194-257)
    auto t0 = std::chrono::steady_clock::now();
    parallel(n_threads, [&](unsigned t) {
        Shard& s = shard[t];
        const uint64_t lo = cut[t], hi = cut[t + 1];
        s.targets.reserve((hi - lo) + 16);
        s.read_end.reserve((hi - lo) / 2 + 16);
        s.reads_count.assign(R, 0);
        s.uniq_count.assign(R, 0);
        s.uniq_count2.assign(R, 0);
        uint64_t n_hits = 0, n_reads = 0, n_uniq = 0;
        std::vector<Target> tmp[3];  // the targets of the run's up to three reads (mate 0 / 1 / 2), in file order
        uint64_t i = lo;
        while (i < hi) {
            uint64_t e = i + 1;
            while (e < hi && ((key[e] ^ key[i]) & kKeyMask) == 0) ++e;
            for (auto& v : tmp) v.clear();
            for (uint64_t k = i; k < e; ++k) {
                if ((flag[k] & 0x4) || ref[k] == -1) continue;  // slimm.hpp:197
                ++n_hits;
                const uint32_t r = static_cast<uint32_t>(ref[k]);
                if (r >= R) continue;  // (the product reports SLIMM_E_REF_RANGE; the streams of the bench hold none)
                const uint32_t center = std::min(static_cast<uint32_t>(pos[k]) + half, ref_len[r]);  // slimm.hpp:200 (Q3)
                std::vector<Target>& tg = tmp[mate_of(flag[k])];
                bool seen = false;  // read_stat.hpp:116-135: the first record of a (read, reference) pair decides (Q1)
                for (const Target& x : tg)
                    if (x.ref == r) {
                        seen = true;
                        break;
                    }
                if (!seen) tg.push_back(Target{r, static_cast<uint32_t>(bin_off[r] + center / bin_width)});
            }
            for (auto& tg : tmp) {
                if (tg.empty()) continue;
                ++n_reads;
                const bool uniq = tg.size() == 1;  // slimm.hpp:224
                if (uniq) {
                    ++n_uniq;
                    ++s.uniq_count[tg[0].ref];
                    add32(&ucov[tg[0].gbin], 1u);
                }
                for (const Target& x : tg) {
                    ++s.reads_count[x.ref];
                    add32(&cov[x.gbin], 1u);
                    s.targets.push_back(x);
                }
                s.read_end.push_back(static_cast<uint32_t>(s.targets.size()));
            }
            i = e;
        }
        s.hits = n_hits;
        s.reads = n_reads;
        s.uniq_reads = n_uniq;
    });
    uint64_t hits = 0, matches = 0, uniq_matches = 0;
    for (const Shard& s : shard) {
        hits += s.hits;
        matches += s.reads;
        uniq_matches += s.uniq_reads;
    }
    // non-zero bins per reference (reference_contig.hpp:84-91)
    std::atomic<uint32_t> next_ref{0};
    parallel(n_threads, [&](unsigned) {
        for (uint32_t r0; (r0 = next_ref.fetch_add(64)) < R;)
            for (uint32_t r = r0; r < std::min(R, r0 + 64); ++r) {
                uint32_t a = 0, b = 0;
                for (uint64_t g = bin_off[r]; g < bin_off[r + 1]; ++g) {
                    a += cov[g] != 0;
                    b += ucov[g] != 0;
                }
                nz_cov[r] = a;
                nz_ucov[r] = b;
                uint32_t rc = 0, uc = 0;
                for (const Shard& s : shard) {
                    rc += s.reads_count[r];
                    uc += s.uniq_count[r];
                }
                reads_count[r] = rc;
                uniq_count[r] = uc;
            }
    });
    seconds[0] = seconds_since(t0);
    scalars[0] = static_cast<uint32_t>(hits);
    scalars[1] = static_cast<uint32_t>(matches);
    scalars[2] = static_cast<uint32_t>(uniq_matches);
    scalars[5] = B;
    for (uint32_t r = 0; r < R; ++r) {
        ref_cols[r * 5 + 0] = reads_count[r];
        ref_cols[r * 5 + 1] = uniq_count[r];
        ref_cols[r * 5 + 2] = nz_cov[r];
        ref_cols[r * 5 + 3] = nz_ucov[r];
    }
    // the arrays themselves / their checksums, for the callers that compare them (not part of any timed phase)
    auto hand_out = [&]() {
        const std::vector<uint32_t>* arr[3] = {&cov, &ucov, &ucov2};
        for (int a = 0; a < 3; ++a) {
            if (bins_out && bins_out[a] && B) std::memcpy(bins_out[a], arr[a]->data(), B * sizeof(uint32_t));
            if (checksums) checksums[a] = bin_checksum(arr[a]->data(), B, n_threads);
        }
    };
    if (hits == 0) {
        hand_out();
        return 1;
    }

    // ---------------------------------------------------------------- cut-offs + phase B (slimm.hpp:328-392, 672-688)
    t0 = std::chrono::steady_clock::now();
    float cc = 0.0f, ucc = 0.0f;
    if (cov_cut_off < 1.0f) {
        std::vector<float> a, b;
        for (uint32_t r = 0; r < R; ++r)
            if (uniq_count[r] > 0) {
                const uint32_t nb = static_cast<uint32_t>(bin_off[r + 1] - bin_off[r]);
                a.push_back(float(nz_cov[r]) / nb);
                b.push_back(float(nz_ucov[r]) / nb);
            }
        cc = quantile_cut_off(a, cov_cut_off);
        ucc = quantile_cut_off(b, cov_cut_off);
    }
    std::vector<uint8_t> valid(R, 0);
    uint32_t n_valid = 0;
    for (uint32_t r = 0; r < R; ++r) {
        if (reads_count[r] == 0) continue;
        const uint32_t nb = static_cast<uint32_t>(bin_off[r + 1] - bin_off[r]);
        if (float(nz_cov[r]) / nb >= cc && float(nz_ucov[r]) / nb >= ucc) {
            valid[r] = 1;
            ++n_valid;
        }
    }
    parallel(n_threads, [&](unsigned t) {
        Shard& s = shard[t];
        uint32_t b = 0;
        uint64_t n_uniq2 = 0;
        std::vector<uint32_t> ids;
        for (uint32_t e : s.read_end) {
            ids.clear();
            uint32_t first_g = 0;
            for (uint32_t k = b; k < e; ++k)
                if (valid[s.targets[k].ref]) {
                    if (ids.empty()) first_g = s.targets[k].gbin;
                    ids.push_back(s.targets[k].ref);
                }
            b = e;
            if (ids.size() == 1) {  // slimm.hpp:384-389
                ++n_uniq2;
                ++s.uniq_count2[ids[0]];
                add32(&ucov2[first_g], 1u);
            } else if (ids.size() > 1) {  // slimm.hpp:516-531, 536-557: level scan; no agreeing level -> the last value read (Q4)
                std::sort(ids.begin(), ids.end());
                uint32_t taxon = 1;
                for (uint32_t lv = 0; lv < 8; ++lv) {
                    bool same = true;
                    const uint32_t first = lineage[static_cast<size_t>(ids[0]) * 8 + lv];
                    for (uint32_t id : ids) {
                        taxon = lineage[static_cast<size_t>(id) * 8 + lv];
                        same = same && taxon == first;
                    }
                    if (same) break;
                }
                ++s.lca[taxon];
                if (pairs_out) {   // slimm.hpp:555: taxon_id__children[lca].insert(ref_ids)
                    for (uint32_t id : ids) s.pairs.insert((static_cast<uint64_t>(taxon) << 32) | id);
                }
            }
        }
        s.uniq_reads2 = n_uniq2;
    });
    seconds[1] = seconds_since(t0);

    // ---------------------------------------------------------------- merge of the per-thread LCA counts
    t0 = std::chrono::steady_clock::now();
    std::unordered_map<uint32_t, uint32_t> lca;
    uint64_t uniq_matches2 = 0;
    for (const Shard& s : shard) {
        uniq_matches2 += s.uniq_reads2;
        for (const auto& kv : s.lca) lca[kv.first] += kv.second;
        for (uint32_t r = 0; r < R; ++r) uniq_count2[r] += s.uniq_count2[r];
    }
    seconds[2] = seconds_since(t0);
    scalars[3] = static_cast<uint32_t>(uniq_matches2);
    scalars[4] = n_valid;
    for (uint32_t r = 0; r < R; ++r) ref_cols[r * 5 + 4] = uniq_count2[r];
    uint32_t k = 0;
    for (const auto& kv : lca) {
        if (k < lca_cap && lca_taxid && lca_count) {
            lca_taxid[k] = kv.first;
            lca_count[k] = kv.second;
        }
        ++k;
    }
    if (n_lca) *n_lca = k;
    if (pairs_out && n_pairs) {
        std::unordered_set<uint64_t> all;
        for (const Shard& s : shard) all.insert(s.pairs.begin(), s.pairs.end());
        uint64_t m = 0;
        for (uint64_t pr : all) {
            if (m < pair_cap) pairs_out[m] = pr;
            ++m;
        }
        *n_pairs = m;
    }
    if (cutoffs_out) {
        cutoffs_out[0] = cc;
        cutoffs_out[1] = ucc;
    }
    hand_out();
    return 0;
}

int dmt_run2(const uint64_t* key, const uint16_t* flag, const int32_t* ref, const int32_t* pos, uint64_t n, uint32_t n_refs,
             const uint32_t* ref_len, const uint32_t* lineage, uint32_t avg_read_len, uint32_t bin_width, float cov_cut_off,
             uint32_t n_threads, uint32_t* ref_cols, uint64_t* scalars, uint32_t* lca_taxid, uint32_t* lca_count,
             uint32_t lca_cap, uint32_t* n_lca, double* seconds, uint32_t* const* bins_out, uint64_t* checksums) {
    return dmt_run3(key, flag, ref, pos, n, n_refs, ref_len, lineage, avg_read_len, bin_width, cov_cut_off, n_threads,
                    ref_cols, scalars, lca_taxid, lca_count, lca_cap, n_lca, seconds, bins_out, checksums, nullptr, 0, nullptr,
                    nullptr);
}

// ---------------------------------------------------------------------------------------------------------------------
// The scalar tail of the path, from the reference: get_reads_lca_count steps 2 and 3 (src/slimm.hpp:560-610) and the rows
// write_abundance prints (:733-843), with the reference's containers (std::unordered_map / std::set).  In: the header's
// lineages, what dmt_run3 counted (direct LCA counts, their contributing references, uniq_reads_count2), the database's
// taxid -> (rank, has a name) and the options the rows depend on.  Out: the propagated counts, the children sets as pairs,
// the rows {taxid, 1 = an "<id>*" row, abundance, read count} in the order the reference would print them for this
// container implementation (compare as sets: Q16).
// ---------------------------------------------------------------------------------------------------------------------
int dmt_profile(uint32_t n_refs, const uint32_t* lineage, const uint32_t* ref_len, const uint32_t* uniq_reads_count2,
                const uint32_t* lca_taxid, const uint32_t* lca_count, uint32_t n_lca, const uint64_t* pairs, uint64_t n_pairs,
                const uint32_t* tax_id, const uint32_t* tax_rank, const uint8_t* tax_named, uint32_t n_tax, uint32_t matches_count,
                uint32_t avg_read_length, uint32_t rank, uint32_t parent_rank, float abundance_cut_off, float coverage_cut_off,
                uint32_t* out_taxid, uint32_t* out_count, uint32_t out_cap, uint32_t* n_out, uint64_t* out_pairs, uint64_t out_pair_cap,
                uint64_t* n_out_pairs, uint32_t* row_taxid, uint8_t* row_star, float* row_abundance, uint32_t* row_reads,
                uint32_t row_cap, uint32_t* n_rows) {
    constexpr uint32_t LINAGE_LENGTH = 8;
    struct Tax {
        uint32_t rank = 0;   // (an absent taxid default-constructs to strain_lv and an empty name: Q6)
        bool named = false;
    };
    std::unordered_map<uint32_t, Tax> taxid__name;
    for (uint32_t i = 0; i < n_tax; ++i) taxid__name[tax_id[i]] = Tax{tax_rank[i], tax_named[i] != 0};
    std::unordered_map<uint32_t, uint32_t> taxon_id__read_count;
    std::unordered_map<uint32_t, std::set<uint32_t>> taxon_id__children;
    for (uint32_t i = 0; i < n_lca; ++i) taxon_id__read_count[lca_taxid[i]] += lca_count[i];                      // :551
    for (uint64_t i = 0; i < n_pairs; ++i) taxon_id__children[static_cast<uint32_t>(pairs[i] >> 32)].insert(static_cast<uint32_t>(pairs[i]));  // :555
    auto lin = [&](uint32_t ref_id) { return lineage + static_cast<size_t>(ref_id) * LINAGE_LENGTH; };
    // :560-586
    std::unordered_map<uint32_t, uint32_t> taxon_id__read_count_cp = taxon_id__read_count;
    for (auto t_id : taxon_id__read_count_cp) {
        const uint32_t rnk = taxid__name[t_id.first].rank;
        const uint32_t first_child = *taxon_id__children.at(t_id.first).begin();
        const uint32_t* linage = lin(first_child);
        std::set<uint32_t> ref_ids = taxon_id__children[t_id.first];
        for (uint32_t j = rnk + 1; j < LINAGE_LENGTH; ++j) {
            const uint32_t reciever_taxa_id = linage[j];
            taxon_id__read_count[reciever_taxa_id] += t_id.second;
            taxon_id__children[reciever_taxa_id].insert(ref_ids.begin(), ref_ids.end());
        }
    }
    // :589-610
    for (uint32_t i = 0; i < n_refs; ++i) {
        if (uniq_reads_count2[i] > 0) {
            const uint32_t* linage = lin(i);
            std::set<uint32_t> ref_ids = taxon_id__children[linage[0]];
            for (uint32_t j = 1; j < LINAGE_LENGTH; ++j) {
                const uint32_t reciever_taxa_id = linage[j];
                taxon_id__read_count[reciever_taxa_id] += uniq_reads_count2[i];
                taxon_id__children[reciever_taxa_id].insert(i);
                taxon_id__children[reciever_taxa_id].insert(ref_ids.begin(), ref_ids.end());
            }
        }
    }
    uint32_t k = 0;
    for (const auto& kv : taxon_id__read_count) {
        if (k < out_cap && out_taxid && out_count) {
            out_taxid[k] = kv.first;
            out_count[k] = kv.second;
        }
        ++k;
    }
    if (n_out) *n_out = k;
    uint64_t m = 0;
    for (const auto& kv : taxon_id__children)
        for (uint32_t r : kv.second) {
            if (m < out_pair_cap && out_pairs) out_pairs[m] = (static_cast<uint64_t>(kv.first) << 32) | r;
            ++m;
        }
    if (n_out_pairs) *n_out_pairs = m;

    // :733-843
    uint32_t n = 0;
    auto row = [&](uint32_t id, bool star, float ab, uint32_t reads) {
        if (n < row_cap && row_taxid) {
            row_taxid[n] = id;
            row_star[n] = star ? 1 : 0;
            row_abundance[n] = ab;
            row_reads[n] = reads;
        }
        ++n;
    };
    std::unordered_map<uint32_t, float> parent_abundance;
    std::unordered_map<uint32_t, uint32_t> parent_reads_count;
    for (auto t_id : taxon_id__read_count) {
        if (taxid__name[t_id.first].rank == parent_rank) {
            parent_abundance[t_id.first] = float(t_id.second) / (matches_count) * 100;
            parent_reads_count[t_id.first] = t_id.second;
        }
    }
    uint32_t sum_reads_count = 0;
    float sum_abundunce = 0.0;
    std::unordered_map<uint32_t, float> sum_abundunce_by_parent;
    std::unordered_map<uint32_t, uint32_t> sum_reads_count_by_parent;
    for (auto t_id : taxon_id__read_count) {
        if (taxid__name[t_id.first].rank == rank) {
            uint32_t genome_Length = 0, children_count = 0, last_child = 0;
            for (auto child : taxon_id__children.at(t_id.first)) {
                genome_Length += ref_len[child];
                last_child = child;
                ++children_count;
            }
            genome_Length = genome_Length / children_count;
            const uint32_t* linage = lin(last_child);
            const float cov = float(t_id.second * avg_read_length) / genome_Length;
            const float abundance = float(t_id.second) / (matches_count) * 100;
            const uint32_t parent_tax_id = linage[parent_rank];
            sum_abundunce_by_parent[parent_tax_id] += abundance;
            sum_reads_count_by_parent[parent_tax_id] += t_id.second;
            if (abundance < abundance_cut_off || cov < coverage_cut_off || !taxid__name[t_id.first].named) continue;
            row(t_id.first, false, abundance, t_id.second);
            sum_abundunce += abundance;
            sum_reads_count += t_id.second;
        }
    }
    for (auto ab_by_parent : sum_abundunce_by_parent) {
        const uint32_t parent_taxid = ab_by_parent.first;
        const float uncl_abundance = parent_abundance[parent_taxid] - sum_abundunce_by_parent[parent_taxid];
        const uint32_t unc_read_count = parent_reads_count[parent_taxid] - sum_reads_count_by_parent[parent_taxid];
        if (uncl_abundance > abundance_cut_off && taxid__name[parent_taxid].named) {
            row(parent_taxid, true, uncl_abundance, unc_read_count);
            sum_reads_count += unc_read_count;
            sum_abundunce += uncl_abundance;
        }
    }
    row(0, true, static_cast<float>(100.0 - sum_abundunce), matches_count - sum_reads_count);
    if (n_rows) *n_rows = n;
    return 0;
}

int dmt_run(const uint64_t* key, const uint16_t* flag, const int32_t* ref, const int32_t* pos, uint64_t n, uint32_t n_refs,
            const uint32_t* ref_len, const uint32_t* lineage, uint32_t avg_read_len, uint32_t bin_width, float cov_cut_off,
            uint32_t n_threads, uint32_t* ref_cols, uint64_t* scalars, uint32_t* lca_taxid, uint32_t* lca_count,
            uint32_t lca_cap, uint32_t* n_lca, double* seconds) {
    return dmt_run2(key, flag, ref, pos, n, n_refs, ref_len, lineage, avg_read_len, bin_width, cov_cut_off, n_threads,
                    ref_cols, scalars, lca_taxid, lca_count, lca_cap, n_lca, seconds, nullptr, nullptr);
}

}  // extern "C"
