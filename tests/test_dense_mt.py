"""The all-core dense CPU restatement (oracle/slimm_dense_mt.cpp, bench.py's cpu_baseline_mt leg) against the oracle."""
import numpy as np
import pytest

from oracle.binding import dense_mt_run, run_workload
from slimm_amd.synth import CONFIGS, SynthConfig, make_workload


@pytest.mark.parametrize("threads", [1, 3, 8])
@pytest.mark.parametrize("case", ["config1", "config2-200k", "deep"])
def test_dense_mt_equals_the_oracle(case, threads):
    if case == "config1":
        w = make_workload(CONFIGS["config1"], seed=41)
    elif case == "config2-200k":
        w = make_workload(CONFIGS["config2"], seed=42, n_records=200_000)
    else:  # many hits per read, strain-level database: most reads keep several references
        w = make_workload(SynthConfig("deep", 150_000, 2_000, 12.0, present_frac=0.2, strain_level=True), seed=43)
    o = run_workload(w, use_qnames=False, collect_bins=False)
    d = dense_mt_run(w, threads=threads)
    assert not d["no_hits"]
    assert (d["hits"], d["matches"], d["uniq_matches"], d["uniq_matches2"], d["n_valid"]) == (
        o.scalars["hits"], o.scalars["matches"], o.scalars["uniq_matches"], o.scalars["uniq_matches2"], o.scalars["n_valid"])
    for k in ("reads_count", "uniq_reads_count", "uniq_reads_count2", "nz_cov", "nz_uniq_cov"):
        assert np.array_equal(d[k], getattr(o, k)), k
    assert d["lca_direct"] == o.lca_direct


def test_dense_mt_no_mapped_record():
    w = make_workload(CONFIGS["config1"], seed=44, n_records=2000)
    w.records.flag[:] |= 4
    assert dense_mt_run(w, threads=2)["no_hits"]
