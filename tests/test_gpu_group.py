"""slimm_group_*: several contexts in one process used like one (slimm_amd/csrc/group.hip).  On the single GPU of the
test box the members share device 0, so the two collectives run in their copy form; the dealing of records by read, the
phase order, the exchanges and the retry of an overflowing pair set are the code every group runs.  With distinct
devices the only difference is which primitive moves the buffers (ncclAllGather / ncclAllReduce)."""
import os

import numpy as np
import pytest

from oracle.binding import run_workload
from slimm_amd.profiler import SlimmGroup
from slimm_amd.synth import CONFIGS, SynthConfig, make_workload
from tests.helpers import assert_matches_oracle, assert_profiles_match, force

pytestmark = pytest.mark.gpu


def _run(w, members, batch, grouped=None, tmp_path=None):
    g = SlimmGroup(w, [0] * members, grouped=grouped)
    g.push_records(w.records, batch=batch)
    path = str(tmp_path / "p.tsv") if tmp_path is not None else None
    assert g.get_profiles(path)
    return g, path


@pytest.mark.parametrize("members,batch", [(1, 0), (2, 5000), (3, 1777), (4, 100_000)])
def test_group_on_one_device_equals_the_oracle(members, batch, tmp_path):
    w = make_workload(CONFIGS["config1"], seed=51)
    o = run_workload(w, use_qnames=False, collect_bins=False)
    g, path = _run(w, members, batch, tmp_path=tmp_path)
    assert_matches_oracle(g.member(0), o, bins=False)
    assert_profiles_match(open(path).read(), o.profile_tsv)
    if members > 1:   # every member holds the merged per-reference columns
        a, b = g.member(0).ref_columns(), g.member(members - 1).ref_columns()
        for k in ("reads_count", "uniq_reads_count", "nz_cov", "nz_uniq_cov", "uniq_reads_count2"):
            assert np.array_equal(a[k], b[k]), k
        # ... and got a share of the records: the stream was dealt in stretches of `batch`
        shares = [g.member(i).stats()["n_records"] for i in range(members)]
        assert sum(shares) == len(w.records) and (batch >= len(w.records) or min(shares) > 0)
    g.close()


def test_group_any_order_goes_by_key(tmp_path):
    w = make_workload(CONFIGS["config1"], seed=52, shuffled=True)
    o = run_workload(w, use_qnames=False, collect_bins=False)
    g, _ = _run(w, 3, 4000, grouped=False)
    assert_matches_oracle(g.member(0), o, bins=False)
    g.close()


def test_group_pair_set_overflow_goes_round_again(monkeypatch):
    """Reads whose references agree at no level (Q4) with a 64-entry pair set: slimm_install_merged_partials answers
    SLIMM_E_RETRY on every member, every member grows its table, and the phase is launched again."""
    force(monkeypatch, pair_cap="64")
    cfg = SynthConfig("q4group", 120_000, 10_000, 6.0, present_frac=0.5, len_lo=200_000, len_hi=600_000)
    w = make_workload(cfg, seed=53)
    rng = np.random.default_rng(2)
    m = w.records.ref_id >= 0
    jump = rng.random(len(w.records)) < 0.2
    w.records.ref_id[m & jump] = rng.integers(0, cfg.n_refs, size=int((m & jump).sum()), dtype=np.int32)
    w.records.begin_pos[m & jump] = 1000
    o = run_workload(w, use_qnames=False, collect_bins=False)
    assert len(o.lca_direct_children) > 64
    g, _ = _run(w, 2, 20_000)
    assert_matches_oracle(g.member(0), o, bins=False)
    g.close()


def test_group_reset_and_a_second_file_and_no_hits():
    w = make_workload(CONFIGS["config1"], seed=54)
    o = run_workload(w, use_qnames=False, collect_bins=False)
    g = SlimmGroup(w, [0, 0])
    for _ in range(2):
        g.reset()
        g.push_records(w.records, batch=3000)
        assert g.get_profiles()
        assert_matches_oracle(g.member(0), o, bins=False)
    g.reset()
    r = w.records.take(np.arange(2000))
    r.flag[:] |= 4
    g.push_records(r, batch=500)
    assert not g.get_profiles()
    g.close()


def test_group_one_run_longer_than_a_batch_stays_on_one_member():
    """A qName run that spans several pushes must not be cut: batches without a run boundary stay with the member that
    holds the run's beginning."""
    w = make_workload(CONFIGS["config1"], seed=55, n_records=6000)
    r = w.records
    r.read_key[1000:4000] = r.read_key[1000]      # one read name with 3000 records
    r.flag[1000:4000] &= np.uint16(~(0x40 | 0x80) & 0xffff)
    o = run_workload(w, use_qnames=False, collect_bins=False)
    g, _ = _run(w, 2, 700)
    assert_matches_oracle(g.member(0), o, bins=False)
    g.close()


@pytest.mark.skipif(os.environ.get("SLIMM_EMU") == "1", reason="needs librccl and a GPU")
def test_group_of_one_through_rccl(monkeypatch):
    """The RCCL form with a communicator of one: the calls (ncclCommInitAll, grouped ncclAllGather / ncclAllReduce on the
    member's stream) are the ones a group of eight makes."""
    force(monkeypatch, group_collectives="rccl")
    w = make_workload(CONFIGS["config1"], seed=56)
    o = run_workload(w, use_qnames=False, collect_bins=False)
    g = SlimmGroup(w, [0])
    assert g.uses_rccl
    g.push_records(w.records)
    assert g.get_profiles()
    assert_matches_oracle(g.member(0), o, bins=False)
    g.close()


# ---------------------------------------------------------------- the three forms of exchange 1, from C++
@pytest.mark.parametrize("members", [3, 4])
@pytest.mark.parametrize("form", ["summary", "sliced", "bins"])
def test_group_exchange_forms_equal_the_oracle(members, form):
    """All-gather of summaries, all-to-all of bitmap slices + small all-reduce, all-reduce of the integer bins (north_star's
    literal collective): groups of three and four on the one device through each form, every result the oracle's.  After
    the bins form member 0 holds the GLOBAL coverage arrays, uniq_cov2 included."""
    w = make_workload(CONFIGS["config1"], seed=57)
    o = run_workload(w, use_qnames=False)
    g = SlimmGroup(w, [0] * members)
    assert g.exchange == ("sliced" if members > 2 else "summary")      # what auto means
    g.set_exchange(form)
    assert g.exchange == form
    g.push_records(w.records, batch=2500)
    assert g.get_profiles()
    assert_matches_oracle(g.member(0), o, bins=False)
    if form == "bins":
        for k, want in enumerate((o.cov, o.uniq_cov, o.uniq_cov2)):
            for i in (0, members - 1):
                assert np.array_equal(g.member(i).bins(k), want), (k, i)
    else:   # the members' arrays are partial sums of the whole
        for k, want in enumerate((o.cov, o.uniq_cov, o.uniq_cov2)):
            assert np.array_equal(sum(g.member(i).bins(k) for i in range(members)), want), k
    g.close()


@pytest.mark.parametrize("shift", ["13", "14"])
@pytest.mark.parametrize("form", ["summary", "sliced"])
def test_group_bitmaps_with_both_tile_sizes(monkeypatch, shift, form):
    """The 'bin != 0' bitmaps k_tile_hist / k_pack write for the exchange are laid out in slices of tiles: both tile
    sizes (SLIMM_FORCE tile_shift) through the all-gather and the all-to-all form, split tiles included."""
    force(monkeypatch, tile_shift=shift)
    w = make_workload(SynthConfig("hot", 200_000, 12, 6.0, bin_width=50, len_lo=400_000, len_hi=900_000, present_frac=0.3),
                      seed=62)
    o = run_workload(w, use_qnames=False, collect_bins=False)
    g = SlimmGroup(w, [0, 0, 0])
    g.set_exchange(form)
    g.push_records(w.records, batch=30_000)
    assert g.get_profiles()
    assert_matches_oracle(g.member(0), o, bins=False)
    g.close()


@pytest.mark.parametrize("form", ["sliced", "bins"])
def test_group_exchange_forms_on_a_larger_stream_and_a_second_file(form):
    w = make_workload(CONFIGS["config2"], seed=58, n_records=300_000)
    o = run_workload(w, use_qnames=False, collect_bins=False)
    g = SlimmGroup(w, [0, 0, 0])
    g.set_exchange(form)
    for _ in range(2):
        g.reset()
        g.push_records(w.records, batch=40_000)
        assert g.get_profiles()
        assert_matches_oracle(g.member(0), o, bins=False)
    g.close()


@pytest.mark.skipif(os.environ.get("SLIMM_EMU") == "1", reason="needs librccl and a GPU")
@pytest.mark.parametrize("form", ["summary", "sliced", "bins"])
def test_group_of_one_through_rccl_in_every_form(monkeypatch, form):
    """Each form through the RCCL entry points with a communicator of one: ncclAllGather; ncclSend / ncclRecv in a group +
    ncclAllReduce; ncclAllReduce over the bins (and over uniq_cov2)."""
    force(monkeypatch, group_collectives="rccl")
    w = make_workload(CONFIGS["config1"], seed=59)
    o = run_workload(w, use_qnames=False)
    g = SlimmGroup(w, [0])
    assert g.uses_rccl
    g.set_exchange(form)
    g.push_records(w.records)
    assert g.get_profiles()
    assert_matches_oracle(g.member(0), o, bins=False)
    if form == "bins":   # (nz_uniq_cov2 is not part of the merged columns: the arrays themselves are compared)
        for k, want in enumerate((o.cov, o.uniq_cov, o.uniq_cov2)):
            assert np.array_equal(g.member(0).bins(k), want), k
    g.close()


def test_group_checked_push_reports_two_names_under_one_key():
    """Unordered input dealt by key: the records of two names that collide in the key meet on one member, whose device
    sort brings them together -- SLIMM_E_KEY_COLLISION, like one context on the same file."""
    from slimm_amd import capi
    w = make_workload(CONFIGS["config1"], seed=60, shuffled=True)
    r = w.records
    chk = ((r.read_key * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(32)).astype(np.uint32)
    o = run_workload(w, use_qnames=False, collect_bins=False)
    g = SlimmGroup(w, [0, 0, 0], grouped=False)
    g.push_records_checked(r, chk, batch=3000)
    assert g.get_profiles()
    assert_matches_oracle(g.member(0), o, bins=False)
    g.reset()
    chk2 = chk.copy()
    victim = 4321
    other = 777
    r.read_key[other] = r.read_key[victim]
    chk2[other] = chk2[victim] ^ np.uint32(0x5a5a5a5a)       # another "name" with the victim's key
    g.push_records_checked(r, chk2, batch=3000)
    with pytest.raises(capi.SlimmError) as e:
        g.get_profiles()
    assert e.value.code == capi.E_KEY_COLLISION
    with pytest.raises(capi.SlimmError):
        g.push_records(r)                                     # checked and unchecked pushes do not mix
    g.close()


@pytest.mark.parametrize("members,grouped", [(2, None), (3, False)])
def test_group_packed_push_deals_by_the_61_bit_identity(members, grouped):
    """slimm_group_push_records_packed: the flag bits ride in the key, so the dealing -- whole qName runs, or key mod
    members -- must look at the low 61 bits only; mates of one pair (bit 61 differs) land on one member."""
    from slimm_amd import capi
    w = make_workload(CONFIGS["config1"], seed=61, shuffled=(grouped is False))
    w.records.read_key &= np.uint64((1 << 61) - 1)            # what a producer of packed records hashes names to
    o = run_workload(w, use_qnames=False, collect_bins=False)
    g = SlimmGroup(w, [0] * members, grouped=grouped)
    g.push_records_packed(w.records, batch=3100)
    assert g.get_profiles()
    assert_matches_oracle(g.member(0), o, bins=False)
    with pytest.raises(capi.SlimmError):
        g.push_records(w.records)                             # forms do not mix within a file
    g.reset()
    g.push_records(w.records, batch=3100)
    assert g.get_profiles()
    assert_matches_oracle(g.member(0), o, bins=False)
    g.close()


@pytest.mark.parametrize("members,batch", [(1, 4000), (2, 3100), (3, 777)])
def test_group_marked_push_deals_whole_runs(members, batch):
    """slimm_group_push_records_marked: 8-byte records that say where a qName run starts; a batch's last run waits for
    the next batch, every member's stretch begins at a run start."""
    from slimm_amd import capi
    w = make_workload(CONFIGS["config1"], seed=63)
    o = run_workload(w, use_qnames=False, collect_bins=False)
    g = SlimmGroup(w, [0] * members)
    g.push_records_marked(w.records, batch=batch)
    assert g.get_profiles()
    assert_matches_oracle(g.member(0), o, bins=False)
    if members > 1:
        with pytest.raises(capi.SlimmError):
            g.push_records(w.records)                         # forms do not mix within a file
        shares = [g.member(i).stats()["n_records"] for i in range(members)]
        assert sum(shares) == len(w.records) and min(shares) > 0
    g.reset()
    g.push_records(w.records, batch=batch)
    assert g.get_profiles()
    assert_matches_oracle(g.member(0), o, bins=False)
    g.close()
