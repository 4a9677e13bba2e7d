"""Bin-layout corner cases of the histogram kernels against the oracle (GPU): thousands of tiny references inside one
8192-bin tile (more reference offsets than k_tile_hist stages in LDS), a single tile receiving far more than one work
item's worth of targets (cut into pieces: sums from the pieces, non-zero counts finished by k_pack), both at once,
and the direct-atomics / two-level bucketing fallbacks on the same inputs."""
import numpy as np
import pytest

from oracle.binding import run_workload
from slimm_amd.profiler import Slimm
from slimm_amd.synth import SynthConfig, make_workload
from tests.helpers import assert_matches_oracle, force

pytestmark = pytest.mark.gpu

TINY_REFS = SynthConfig("tiny_refs", 150_000, 6_000, 2.5, bin_width=1000, len_lo=1_000, len_hi=2_900, present_frac=0.5)
HOT_TILE = SynthConfig("hot_tile", 300_000, 12, 1.3, bin_width=1000, len_lo=300_000, len_hi=600_000, present_frac=0.5)
MIXED = SynthConfig("mixed", 250_000, 3_000, 1.8, bin_width=500, len_lo=600, len_hi=40_000, present_frac=0.05)


def _check(cfg, seed, grouped=True):
    w = make_workload(cfg, seed=seed)
    o = run_workload(w, use_qnames=False)
    s = Slimm.for_workload(w, device=0, grouped=grouped)
    s.push_records(w.records)
    prof = s.get_profiles()
    assert not o.no_hits and prof is not None
    assert_matches_oracle(s, o)
    return s


@pytest.mark.parametrize("cfg", [TINY_REFS, HOT_TILE, MIXED], ids=lambda c: c.name)
def test_layout_default_path(cfg):
    s = _check(cfg, seed=5)
    st = s.stats()
    if cfg is TINY_REFS:
        assert st["total_bins"] < 3 * 8192 and st["reference_count"] > 1000   # ~2000 references per tile
    if cfg is HOT_TILE:
        assert st["n_targets"] > 20 * st["total_bins"] / 8192                 # far more than 16 K targets in a tile


@pytest.mark.parametrize("cfg", [TINY_REFS, HOT_TILE], ids=lambda c: c.name)
@pytest.mark.parametrize("knob", [{"direct_atomics": 1}, {"two_level": 1}, {"wide_rows": 1}, {"fused_scan": 0}],
                         ids=lambda e: next(iter(e)))
def test_layout_fallback_paths(monkeypatch, cfg, knob):
    """SLIMM_FORCE direct_atomics / two_level=1 / wide_rows / fused_scan=0 (slimm_amd/csrc/force.h) on these layouts."""
    force(monkeypatch, **knob)
    _check(cfg, seed=6)


def test_layout_any_order():
    _check(MIXED, seed=7, grouped=False)


@pytest.mark.parametrize("cfg", [HOT_TILE, TINY_REFS], ids=lambda c: c.name)
@pytest.mark.parametrize("prepared", [True, False])
def test_layout_through_the_summary_exchange(cfg, prepared):
    """Coverage summary (per-reference sums + 'bin != 0' bitmaps) of these layouts, with the bitmaps written by the
    histogram kernels (split tiles finished by k_pack) and by the separate bitmap kernels."""
    import torch
    w = make_workload(cfg, seed=9)
    o = run_workload(w, use_qnames=False)
    s = Slimm.for_workload(w, device=0)
    s.prepare_summary(prepared)
    s.push_records(w.records)
    s.analyze_alignments()
    mine = s.coverage_summary_tensor()
    R = len(w.ref_len)
    Bp = (mine.numel() - 4 * R - 16) * 32 // 2
    words = mine.cpu().numpy().view(np.uint32)
    for which, arr in ((0, o.cov), (1, o.uniq_cov)):
        bits = np.unpackbits(words[4 * R + 16 + which * (Bp // 32):4 * R + 16 + (which + 1) * (Bp // 32)].view(np.uint8),
                             bitorder="little")
        assert int(bits.sum()) == int((arr != 0).sum())
    assert s.finish_coverage_merged(mine.clone(), 1)
    s.filter_alignments()
    s.get_reads_lca_count()
    assert_matches_oracle(s, o)
