"""Shared checks: compare everything the HIP path (slimm_amd.profiler.Slimm) produced with the CPU oracle."""
from __future__ import annotations

import numpy as np
import pytest

from oracle.binding import OracleResult, parse_profile

ABUNDANCE_RTOL = 1e-6  # BASELINE.json north_star: relative-abundance floats within 1e-6


def partials_from_oracle(o: OracleResult, lineage: np.ndarray, dense_taxid: np.ndarray):
    """Re-express the oracle's direct-LCA result (stage 0) in the device's partial-result encoding."""
    T = dense_taxid.shape[0]
    R = lineage.shape[0]
    dense_of = {int(t): i for i, t in enumerate(dense_taxid.tolist())}
    lca = np.zeros(T, dtype=np.uint32)
    for t, c in o.lca_direct.items():
        lca[dense_of[t]] = c
    marks = np.zeros(R, dtype=np.uint32)
    pairs = []
    for t, r in o.lca_direct_children:
        lv = [k for k in range(8) if int(lineage[r, k]) == t]
        if lv:
            marks[r] |= 1 << lv[0]
        else:
            pairs.append((dense_of[t] << 32) | r)
    return o.uniq_reads_count2.copy(), lca, marks, np.array(sorted(pairs), dtype=np.uint64)


def assert_profiles_match(got_text: str, want_text: str, check_lineage: bool = True):
    got, want = parse_profile(got_text), parse_profile(want_text)
    # Q16: the reference sums a parent's child abundances in float32 in the iteration order of an unordered_map,
    # and prints an "unclassified" row when parent - sum > abundance_cut_off.  With a cut-off of 0 and every read of the parent
    # in its children, that difference is 0 or a float32 rounding step of the PARENT's abundance (4.8e-7 at a few per cent,
    # 3.8e-6 at 50 %: scripts/stress_random_cases.py seed 4291, family rank) by the ORDER of the sum alone: a row `<parent>*`
    # with read_count 0 is there or not there by that order, on both sides (found by scripts/stress_bgzf.py, seed 6061).  Such
    # rows may be on one side only.  The parent's abundance is not in the profile; abundances are per cent (<= 100), so the
    # bound is four float32 steps at 100 (3.05e-5) -- a row that small only prints with an abundance cut-off below it.
    q16_residue = 4.0 * float(np.spacing(np.float32(100.0)))

    def residue(rows, k):
        return k.endswith("*") and k != "0*" and rows[k][1] == 0 and abs(rows[k][0]) <= q16_residue
    only_got = {k for k in set(got) - set(want) if not residue(got, k)}
    only_want = {k for k in set(want) - set(got) if not residue(want, k)}
    assert not only_got and not only_want, f"profile rows differ: only got {only_got}, only want {only_want}"
    want = {k: v for k, v in want.items() if k in got}
    for k in want:
        assert got[k][1] == want[k][1], f"read_count of row {k}: {got[k][1]} != {want[k][1]}"
        # both sides print 6 significant digits; allow one unit in the last printed digit for a different float sum order
        assert got[k][0] == pytest.approx(want[k][0], rel=2e-5, abs=2e-5), f"abundance of row {k}"
        if check_lineage:
            assert got[k][2] == want[k][2], f"lineage of row {k}"


def assert_matches_oracle(s, o: OracleResult, bins: bool = True, check_lineage: bool = True):
    """Bit-exact on every integer; floats within the stated tolerance."""
    st = s.stats()
    for a, b in (("hits_count", "hits"), ("matches_count", "matches"), ("uniq_matches_count", "uniq_matches"),
                 ("uniq_hits_count", "uniq_hits"), ("uniq_matches_count2", "uniq_matches2"),
                 ("reference_count", "reference_count"), ("matched_ref_length", "matched_ref_length"),
                 ("failed_by_cov", "failed_by_cov"), ("failed_by_uniq_cov", "failed_by_uniq_cov"),
                 ("failed_by_min_read", "failed_by_min_read"), ("n_valid", "n_valid"), ("bin_width", "bin_width"),
                 ("min_reads", "min_reads")):
        assert st[a] == o.scalars[b], f"{a}: {st[a]} != {o.scalars[b]}"
    assert st["coverage_cut_off"] == pytest.approx(o.cutoffs[0], rel=ABUNDANCE_RTOL)
    assert st["uniq_coverage_cut_off"] == pytest.approx(o.cutoffs[1], rel=ABUNDANCE_RTOL)
    rc = s.ref_columns()
    for a, b in (("reads_count", o.reads_count), ("uniq_reads_count", o.uniq_reads_count),
                 ("uniq_reads_count2", o.uniq_reads_count2), ("nbins", o.nbins), ("nz_cov", o.nz_cov),
                 ("nz_uniq_cov", o.nz_uniq_cov), ("valid", o.valid)):
        assert np.array_equal(rc[a].astype(np.uint32), b.astype(np.uint32)), f"per-reference column {a} differs"
    np.testing.assert_allclose(rc["abundance"], o.abundance, rtol=ABUNDANCE_RTOL)
    np.testing.assert_allclose(rc["uniq_abundance"], o.uniq_abundance, rtol=ABUNDANCE_RTOL)
    if bins:
        assert np.array_equal(rc["nz_uniq_cov2"], o.nz_uniq_cov2), "nz_uniq_cov2 differs"
        for w, want in ((0, o.cov), (1, o.uniq_cov), (2, o.uniq_cov2)):
            got = s.bins(w)
            assert got.shape == want.shape
            if not np.array_equal(got, want):
                bad = np.nonzero(got != want)[0]
                raise AssertionError(f"coverage array {w}: {bad.size} bins differ, first at {bad[0]}: "
                                     f"{got[bad[0]]} != {want[bad[0]]}")
    assert s.taxon_counts(0) == o.lca_direct, "direct LCA counts differ"
    assert s.children_pairs(0) == o.lca_direct_children, "direct LCA children differ"
    assert s.taxon_counts(1) == o.taxon_count, "taxon_id__read_count differs"
    assert s.children_pairs(1) == o.taxon_children, "taxon_id__children differs"
    assert_profiles_match(s.write_abundance(), o.profile_tsv, check_lineage)
    st = s.stats()
    assert st["profile_count"] == o.scalars["profile_count"] and st["profile_failed"] == o.scalars["profile_failed"]


def force(monkeypatch, **knobs):
    """SLIMM_FORCE (slimm_amd/csrc/force.h): the ONE environment variable the library's test knobs are read from --
    force(monkeypatch, two_level=1, tile_shift=14) adds `two_level=1,tile_shift=14` to it for the rest of the test."""
    import os
    cur = dict(p.split("=", 1) if "=" in p else (p, "1") for p in os.environ.get("SLIMM_FORCE", "").split(",") if p)
    cur.update({k: str(v) for k, v in knobs.items()})
    monkeypatch.setenv("SLIMM_FORCE", ",".join(f"{k}={v}" for k, v in cur.items()))
