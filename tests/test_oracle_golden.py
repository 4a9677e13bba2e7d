"""Pins the CPU oracle against the only reference-produced vectors that exist for this path:
the two known-answer micro-cases of SURVEY.md Appendix C (tests/golden/tiny.json, holes.json)."""
import numpy as np
import pytest

from oracle.binding import run_workload
from tests.cases import load_golden
from slimm_amd.workload import accession_of


def _check_profile(rows, expected):
    assert set(rows) == set(expected)
    for k, (ab, cnt) in expected.items():
        assert rows[k][1] == cnt, k
        # the reference prints 6 significant digits; the golden carries those digits
        assert rows[k][0] == pytest.approx(ab, rel=1e-5), k


def test_tiny_matches_reference_observation():
    w, exp, _ = load_golden("tiny")
    r = run_workload(w)
    for k in ("hits", "matches", "uniq_matches", "uniq_matches2", "n_valid"):
        assert r.scalars[k] == exp[k], k
    assert r.cutoffs[0] == pytest.approx(exp["cutoffs"][0], rel=1e-5)
    assert r.cutoffs[1] == pytest.approx(exp["cutoffs"][1], rel=1e-5)
    for i, n in enumerate(w.ref_names):
        got = [r.reads_count[i], r.uniq_reads_count[i], r.uniq_reads_count2[i], r.nbins[i], r.nz_cov[i],
               r.nz_uniq_cov[i], r.nz_uniq_cov2[i]]
        assert [int(x) for x in got] == exp["refs"][accession_of(n)], n
        assert bool(r.valid[i]) == (accession_of(n) not in exp["invalid"])
    _check_profile(r.profile_rows(), exp["profile"])


def test_holes_matches_reference_observation():
    w, exp, _ = load_golden("holes")
    r = run_workload(w)
    for k in ("hits", "matches", "uniq_matches", "uniq_matches2", "n_valid"):
        assert r.scalars[k] == exp[k], k
    assert r.cutoffs[0] == pytest.approx(exp["cutoffs"][0], rel=1e-5)
    assert r.cutoffs[1] == pytest.approx(exp["cutoffs"][1], rel=1e-5)
    for i, n in enumerate(w.ref_names):
        got = [int(r.reads_count[i]), int(r.uniq_reads_count[i]), int(r.uniq_reads_count2[i])]
        assert got == exp["refs3"][accession_of(n)], n
    assert r.cov[: r.nbins[0]].tolist() == exp["H1_cov"]  # `edge` clamps into the last bin (Q3)
    _check_profile(r.profile_rows(), exp["profile"])


def test_keys_instead_of_names_give_the_same_result():
    """The oracle keyed by the decimal text of read_key must equal the oracle keyed by qName."""
    w, _, _ = load_golden("tiny")
    a = run_workload(w, use_qnames=True)
    b = run_workload(w, use_qnames=False)
    assert a.scalars == b.scalars
    assert np.array_equal(a.cov, b.cov) and np.array_equal(a.uniq_cov2, b.uniq_cov2)
    assert a.taxon_count == b.taxon_count and a.taxon_children == b.taxon_children
