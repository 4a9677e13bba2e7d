"""The all-core dense CPU restatement (oracle/slimm_dense_mt.cpp: bench.py's cpu_baseline_mt leg and the comparator of
the full-size GPU parity tests) against the oracle -- every per-reference column, the scalars, the direct LCA counts and
every bin of the three coverage arrays, at sizes where the big-table paths of the configurations live (all 20 k / 50 k
references of configs[2] / [4])."""
import numpy as np
import pytest

from oracle.binding import bin_checksum, dense_mt_run, run_workload
from slimm_amd.synth import CONFIGS, SynthConfig, make_workload


def assert_dense_equals_oracle(d, o, bins=True):
    assert not d["no_hits"]
    assert (d["hits"], d["matches"], d["uniq_matches"], d["uniq_matches2"], d["n_valid"]) == (
        o.scalars["hits"], o.scalars["matches"], o.scalars["uniq_matches"], o.scalars["uniq_matches2"], o.scalars["n_valid"])
    for k in ("reads_count", "uniq_reads_count", "uniq_reads_count2", "nz_cov", "nz_uniq_cov"):
        assert np.array_equal(d[k], getattr(o, k)), k
    assert d["lca_direct"] == o.lca_direct
    if "profile" in d:   # the scalar tail from the reference (dmt_profile): propagated counts, children sets, profile rows
        assert d["taxon_count"] == o.taxon_count
        assert d["taxon_children"] == o.taxon_children
        rows = o.profile_rows()
        assert set(d["profile"]) == set(rows)
        for key, (ab, reads) in d["profile"].items():
            assert reads == rows[key][1], key
            assert ab == pytest.approx(rows[key][0], rel=2e-5, abs=1e-6), key
    if bins:
        assert d["total_bins"] == o.cov.shape[0]
        for i, k in enumerate(("cov", "uniq_cov", "uniq_cov2")):
            assert np.array_equal(d[k], getattr(o, k)), k
            assert d["checksums"][i] == bin_checksum(getattr(o, k)), k


@pytest.mark.parametrize("threads", [1, 3, 8])
@pytest.mark.parametrize("case", ["config1", "config2-200k", "deep"])
def test_dense_mt_equals_the_oracle(case, threads):
    if case == "config1":
        w = make_workload(CONFIGS["config1"], seed=41)
    elif case == "config2-200k":
        w = make_workload(CONFIGS["config2"], seed=42, n_records=200_000)
    else:  # many hits per read, strain-level database: most reads keep several references
        w = make_workload(SynthConfig("deep", 150_000, 2_000, 12.0, present_frac=0.2, strain_level=True), seed=43)
    o = run_workload(w, use_qnames=False)
    assert_dense_equals_oracle(dense_mt_run(w, threads=threads, want_bins=True, want_profile=True), o)


@pytest.mark.parametrize("name,n", [("config2", 2_000_000), ("config3", 2_000_000), ("config4", 2_000_000), ("config5", 2_500_000)])
def test_dense_mt_equals_the_oracle_on_the_full_reference_sets(name, n):
    """>= 2 M records of each GPU configuration with ALL its references (5 k / 20 k / 20 k / 50 k; 20 - 200 M bins): the
    comparator of the full-size GPU tests is itself pinned to the oracle where the big tables are."""
    w = make_workload(CONFIGS[name], seed=7, n_records=n)
    o = run_workload(w, use_qnames=False)
    assert_dense_equals_oracle(dense_mt_run(w, threads=8, want_bins=True, want_profile=True), o)


@pytest.mark.parametrize("mk", ["tiny", "holes", "genus"])
def test_dense_mt_profile_tail_on_the_micro_cases(mk):
    """dmt_profile (the propagation and the profile rows, written from src/slimm.hpp:560-610, 733-843) on the two
    reference-observed micro-cases -- lineage holes, an accession absent from the database, taxid 0 propagating (Q5, Q6, Q13) --
    and at another rank."""
    from tests.cases import holes_case, tiny_case
    w = tiny_case() if mk != "holes" else holes_case()
    if mk == "genus":
        w.options.rank = "genus"
    o = run_workload(w, use_qnames=False)
    assert_dense_equals_oracle(dense_mt_run(w, threads=2, want_bins=True, want_profile=True), o)


def test_bin_checksum_tells_positions_apart():
    a = np.zeros(1000, dtype=np.uint32)
    a[10] = 3
    b = np.zeros(1000, dtype=np.uint32)
    b[11] = 3
    c = a.copy()
    c[10] = 2
    c[500] = 1
    assert len({bin_checksum(a), bin_checksum(b), bin_checksum(c)}) == 3
    big = np.full(40_000_000, 0xffffffff, dtype=np.uint32)   # several pieces, wrap-around in the sum
    want = sum(0xffffffff * (((i + 1) * 0x9E3779B97F4A7C15) % 2**64) for i in (0, 1, 39_999_999)) % 2**64
    got = (bin_checksum(big[:2]) + (bin_checksum(big) - bin_checksum(big[:39_999_999]))) % 2**64
    assert got == want


def test_dense_mt_no_mapped_record():
    w = make_workload(CONFIGS["config1"], seed=44, n_records=2000)
    w.records.flag[:] |= 4
    assert dense_mt_run(w, threads=2)["no_hits"]
