"""record_order = SLIMM_ORDER_ANY: the device-side grouping by read identity (slimm_amd/csrc/group_by_ident.hip) against
the oracle, which groups through a hash map keyed by the read name like the reference (src/slimm.hpp:204-211) and
therefore takes the same shuffled stream.  The plan's knobs (hash bits per bucket, digit width, persistent workgroups)
are forced through their whole range: buckets of one identity (nothing for the finish to do), buckets of a few (lane
shifts), buckets of hundreds and of the whole stream (selection sweeps), one- to four-pass partitions, stretches of one
record and of several rounds."""
import numpy as np
import pytest

from oracle.binding import run_workload
from slimm_amd.profiler import Slimm
from slimm_amd.synth import CONFIGS, SynthConfig, make_workload
from slimm_amd.workload import Records, Workload
from tests.helpers import assert_matches_oracle, force
from tests.test_gpu_parity import _interleave_mates, _order_preserving_interleave, one_long_read_workload

pytestmark = pytest.mark.gpu

PLANS = [  # SLIMM_FORCE (group_bits, group_width, group_grid); None = the library's own choice
    (None, None, None),
    (1, 1, 2),       # two buckets: the finish's selection sweeps over half the stream each
    (3, 3, 5),
    (6, 2, 1),       # three passes of two bits in ONE stretch: many rounds per workgroup
    (11, 11, 7),     # the widest digit
    (20, 10, 512),   # 10-bit digits: the widest with lane-mask tables and an ordered round (156 KB of LDS)
    (27, 9, 64),     # three passes of nine bits (the plan of 100 M records)
    (12, 4, 512),
    (22, 11, 64),
    (24, 8, 3),
    (32, 8, 512),    # four passes, all 32 hash bits: buckets of one identity
]


def _plan(monkeypatch, plan):
    pairs = [f"{name}={v}" for name, v in zip(("group_bits", "group_width", "group_grid"), plan) if v is not None]
    if pairs:
        monkeypatch.setenv("SLIMM_FORCE", ",".join(pairs))
    else:
        monkeypatch.delenv("SLIMM_FORCE", raising=False)


def _shuffled(w: Workload, seed: int, keep_read_order: bool = False) -> Workload:
    rec = _order_preserving_interleave(w.records, seed) if keep_read_order else w.records.take(
        np.random.default_rng(seed).permutation(len(w.records)))
    return Workload(w.ref_names, w.ref_len, w.taxonomy, rec, w.avg_read_len, w.options, w.name + "-shuffled", grouped=False)


def _check(w: Workload, form: str = "four", batch: int = 0):
    o = run_workload(w, use_qnames=False)
    s = Slimm.for_workload(w, device=0, grouped=False)
    if form == "packed":
        s.push_records_packed(w.records, batch=batch)
    else:
        s.push_records(w.records, batch=batch)
    prof = s.get_profiles()
    if o.no_hits:
        assert prof is None
    else:
        assert_matches_oracle(s, o)
    s.close()


@pytest.mark.parametrize("plan", PLANS)
def test_every_plan_on_a_shuffled_small_file(monkeypatch, plan):
    _plan(monkeypatch, plan)
    w = _shuffled(make_workload(CONFIGS["config1"], seed=5), 11)
    w.records.flag[::13] |= 4            # unmapped records anywhere
    w.records.ref_id[3::17] = -1
    _check(w)
    _check(w, batch=777)


@pytest.mark.parametrize("plan", [PLANS[0], PLANS[1], PLANS[3], PLANS[4], PLANS[7]])
def test_plans_on_deeper_streams(monkeypatch, plan):
    """40 hits per read (buckets of 40 - 80 records at the default plan: both finish paths), interleaved mates, and
    packed records; 60 K records so that a stretch holds several rounds at small grids."""
    _plan(monkeypatch, plan)
    w = _shuffled(make_workload(SynthConfig("deep", 60_000, 1_500, 40.0, strain_level=True), seed=6), 12)
    _check(w)
    m = make_workload(SynthConfig("pairs", 30_000, 800, 6.0), seed=7, paired_frac=0.9)
    _check(_shuffled(_interleave_mates(m), 13))
    k = w.records.read_key & np.uint64((1 << 61) - 1)
    _check(Workload(w.ref_names, w.ref_len, w.taxonomy, Records(k, w.records.flag, w.records.ref_id, w.records.begin_pos),
                    w.avg_read_len, w.options, "deep61", grouped=False), form="packed")


@pytest.mark.parametrize("plan", [PLANS[0], PLANS[2], PLANS[8]])
def test_a_read_of_thousands_of_records_scattered_over_the_file(monkeypatch, plan):
    """One read with 9 000 records among thousands of ordinary ones, every record anywhere in the file: its bucket is
    longer than any window (the finish's long path), shared with other identities at the small plans."""
    _plan(monkeypatch, plan)
    _check(_shuffled(one_long_read_workload(9_000), 14))
    _check(_shuffled(one_long_read_workload(3_000, "last"), 15, keep_read_order=True))


def test_low_entropy_keys(monkeypatch):
    """Keys that are small consecutive integers, and keys that differ in their top bits only: the buckets come from a
    hash of the key, not from its bits."""
    _plan(monkeypatch, PLANS[0])
    w = _shuffled(make_workload(CONFIGS["config1"], seed=8), 16)
    ids = np.unique(w.records.read_key, return_inverse=True)[1].astype(np.uint64)
    for key in (ids, ids << np.uint64(40), ids * np.uint64(4096) + np.uint64(5)):
        _check(Workload(w.ref_names, w.ref_len, w.taxonomy, Records(key, w.records.flag, w.records.ref_id,
                                                                     w.records.begin_pos),
                        w.avg_read_len, w.options, "lowent", grouped=False))


def test_ragged_sizes_around_rounds_and_stretches(monkeypatch):
    """Stream lengths around the 4096-record round, the 512-record wave piece and the finish's 1024-record stretch, with
    a grid of 2 (stretches of several rounds) and the default one (stretches of a few records)."""
    base = _shuffled(make_workload(SynthConfig("rag", 20_000, 40, 3.0, bin_width=100, len_lo=5_000, len_hi=50_000,
                                               present_frac=0.5), seed=9), 17)
    for grid in (2, None):
        _plan(monkeypatch, (None, None, grid))
        for n in (1, 63, 64, 65, 511, 512, 513, 1023, 1024, 1025, 4095, 4096, 4097, 8191, 8193, 12_289):
            w = Workload(base.ref_names, base.ref_len, base.taxonomy, base.records.take(np.arange(n)), base.avg_read_len,
                         base.options, f"rag{n}", grouped=False)
            _check(w)
