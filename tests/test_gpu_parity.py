"""Parity of the HIP path (through the C ABI) with the CPU oracle on a real MI355X.

Bit-exact for every integer (scalars, per-reference columns, every cov / uniq_cov / uniq_cov2 bin, per-taxon counts,
children sets, profile read counts); relative-abundance floats within 1e-6 (BASELINE.json north_star).
"""
import os

import numpy as np
import pytest

from oracle.binding import run_workload
from slimm_amd import capi
from slimm_amd.profiler import Slimm
from slimm_amd.synth import CONFIGS, SynthConfig, make_workload
from slimm_amd.workload import Options, Records, Workload
from tests.cases import Q18_EXPECTED, holes_case, load_golden, q18_case, records_from_sam, taxonomy_from_lineages, tiny_case
from tests.helpers import assert_matches_oracle, assert_profiles_match, force

pytestmark = pytest.mark.gpu


def run_gpu(w: Workload, grouped=None, batch=0, expect_hits=True, keep_bins=True) -> Slimm:
    s = Slimm.for_workload(w, device=0, grouped=grouped)
    if not keep_bins:
        s.keep_bins(False)
    s.push_records(w.records, batch=batch)
    prof = s.get_profiles()
    assert (prof is not None) == expect_hits
    return s


def check(w: Workload, grouped=None, batch=0, keep_bins=True, **kw):
    o = run_workload(w, use_qnames=w.records.qname is not None)
    s = run_gpu(w, grouped, batch, expect_hits=not o.no_hits, keep_bins=keep_bins)
    if not keep_bins:
        kw["bins"] = False
    if o.no_hits:
        assert s.stats()["hits_count"] == 0
    else:
        assert_matches_oracle(s, o, **kw)
    return s, o


# ---------------------------------------------------------------- golden micro-cases (reference-observed)
@pytest.mark.parametrize("name", ["tiny", "holes"])
def test_golden_micro_cases(name):
    w, exp, _ = load_golden(name)
    s, o = check(w)
    rows = {k: (v[0], v[1]) for k, v in __import__("oracle.binding", fromlist=["parse_profile"]).parse_profile(
        s.write_abundance()).items()}
    for k, (ab, cnt) in exp["profile"].items():
        assert rows[k][1] == cnt and rows[k][0] == pytest.approx(ab, rel=1e-5)
    st = s.stats()
    assert (st["hits_count"], st["matches_count"], st["uniq_matches_count"], st["uniq_matches_count2"]) == (
        exp["hits"], exp["matches"], exp["uniq_matches"], exp["uniq_matches2"])


@pytest.mark.parametrize("name", ["tiny", "holes"])
def test_golden_micro_cases_unordered_path(name):
    w, _, _ = load_golden(name)
    check(w, grouped=False)  # same file order, but through the sort path


# ---------------------------------------------------------------- quirks
def _two_ref_case(rows, lens=(1000, 1000), A=50, W=100, lin=None):
    lin = lin or {"X": [101, 11, 21, 31, 41, 51, 61, 2], "Y": [102, 12, 21, 31, 41, 51, 61, 2]}
    names = [a + ".1" for a in lin]
    return Workload(names, np.array(lens, dtype=np.uint32), taxonomy_from_lineages(lin), records_from_sam(rows, names),
                    avg_read_len=A, options=Options(bin_width=W, cov_cut_off=0.99), name="two")


def test_q1_first_position_of_a_pair_wins():
    rows = [("a", 0, "X.1", 10), ("a", 256, "X.1", 910), ("b", 0, "X.1", 500), ("b", 256, "Y.1", 100),
            ("b", 256, "X.1", 20), ("b", 256, "Y.1", 800)]
    s, o = check(_two_ref_case(rows))
    cov = s.bins(0)
    assert cov[0] == 1 and cov[9] == 0      # read a: only its first record on X counts
    assert cov[5] == 1                      # read b on X: first record (POS 500), not the later POS 20
    assert s.stats()["hits_count"] == 6 and s.stats()["matches_count"] == 2


def test_q2_mates_are_different_reads():
    rows = [("p", 0x41, "X.1", 10), ("p", 0x81, "X.1", 300), ("p", 0x41 | 0x100, "Y.1", 10), ("q", 0, "X.1", 700)]
    s, o = check(_two_ref_case(rows))
    assert s.stats()["matches_count"] == 3 and s.stats()["uniq_matches_count"] == 2


def test_q18_the_key_is_the_string_name_plus_mate_suffix():
    """src/slimm.hpp:204-208: `N`/0x40 and an unflagged read named `N.1` are ONE read (one key string); `N.1`/0x40 stays
    apart ("N.1.1").  Four-array records keyed by the canonical base (slimm_host_canonical_read_name), grouped and ANY."""
    w = q18_case()
    s, o = check(w)
    st = s.stats()
    assert (st["hits_count"], st["matches_count"], st["uniq_matches_count"]) == (
        Q18_EXPECTED["hits"], Q18_EXPECTED["matches"], Q18_EXPECTED["uniq_matches"])
    assert s.bins(0)[3] == 1 and s.bins(0)[5] == 0       # read "M.2": X once, at its first record's bin (Q1)
    for seed in range(4):
        wa = q18_case(list(np.random.default_rng(seed).permutation(18)))
        s, o = check(wa, grouped=False)
        assert s.stats()["matches_count"] == Q18_EXPECTED["matches"]


def test_q3_bin_clamp_and_wrap():
    rows = [("a", 0, "X.1", 1000), ("b", 0, "X.1", 0), ("c", 0, "X.1", 976), ("d", 0, "Y.1", 1)]
    s, o = check(_two_ref_case(rows))
    cov = s.bins(0)
    assert cov[10] == 2   # POS 1000 and 976: centre >= len clamps to len -> bin len/W
    assert cov[0] == 1    # POS 0 -> beginPos -1 -> uint32 wrap to A/2 - 1 = 24 -> bin 0


def test_q4_q5_lca_fallthrough_and_holes():
    lin = {"X": [101, 11, 21, 31, 41, 51, 61, 2], "Y": [102, 12, 22, 32, 42, 52, 62, 2157],
           "H1": [201, 0, 23, 33, 43, 53, 63, 2], "H2": [202, 0, 23, 33, 43, 53, 63, 2]}
    names = [a + ".1" for a in lin]
    rows = []
    for i in range(12):
        rows += [(f"u{i}", 0, names[i % 4], 1 + 70 * i)]
    rows += [("x1", 0, "X.1", 100), ("x1", 256, "Y.1", 100), ("x2", 0, "Y.1", 300), ("x2", 256, "X.1", 300),
             ("h", 0, "H1.1", 200), ("h", 256, "H2.1", 200)]
    w = Workload(names, np.full(4, 1000, dtype=np.uint32), taxonomy_from_lineages(lin), records_from_sam(rows, names),
                 avg_read_len=50, options=Options(bin_width=100, cov_cut_off=0.99), name="q4")
    s, o = check(w)
    d = s.taxon_counts(0)
    assert d[2157] == 2   # no level agrees -> lineage[largest ref id][7] (Y is ref 1 > X ref 0)
    assert d[0] == 1      # shared species hole agrees at level 1 -> taxid 0


def test_unknown_accession_gets_zero_lineage():
    w, _, _ = load_golden("holes")   # NODB.1 is not in the database (Q13)
    s, o = check(w)
    assert s.ref_columns()["reads_count"][4] == 11


# ---------------------------------------------------------------- seeded synthetic inputs vs the oracle
@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_config1_scale(seed):
    check(make_workload(CONFIGS["config1"], seed=seed))


@pytest.mark.parametrize("seed", [5, 6])
def test_config1_scale_shuffled_records(seed):
    check(make_workload(CONFIGS["config1"], seed=seed, shuffled=True))


def test_config1_with_holes():
    w = make_workload(CONFIGS["config1"], seed=7, hole_every=3)
    o = run_workload(w)
    s = run_gpu(w)
    # taxid 0's propagated count depends on unordered_map iteration order in the reference (Q17): pin everything up
    # to and including the direct LCA hits, and the per-reference integers
    assert s.taxon_counts(0) == o.lca_direct and s.children_pairs(0) == o.lca_direct_children
    rc = s.ref_columns()
    assert np.array_equal(rc["uniq_reads_count2"], o.uniq_reads_count2)
    assert np.array_equal(s.bins(2), o.uniq_cov2)


@pytest.mark.parametrize("n,seed", [(300_000, 1), (1_000_000, 2)])
def test_config2_shape(n, seed):
    check(make_workload(CONFIGS["config2"], seed=seed, n_records=n))


def test_config2_shape_shuffled():
    check(make_workload(CONFIGS["config2"], seed=3, n_records=400_000, shuffled=True))


def test_config3_shape_more_hits():
    cfg = SynthConfig("c3small", 400_000, 2_000, 8.0)
    check(make_workload(cfg, seed=4))


def test_config5_shape_strain_level_many_hits():
    cfg = SynthConfig("c5small", 400_000, 3_000, 40.0, strain_level=True)
    s, o = check(make_workload(cfg, seed=5))
    assert len(o.lca_direct) > 10  # deep-LCA stress really happened


def test_many_cross_superkingdom_reads_use_the_pair_set():
    # every 4th phylum is archaeal in the synthetic tree; a huge neighbour spread makes reads straddle it
    cfg = SynthConfig("q4many", 200_000, 10_000, 6.0, present_frac=0.5, len_lo=200_000, len_hi=600_000)
    w = make_workload(cfg, seed=8)
    rng = np.random.default_rng(0)
    m = w.records.ref_id >= 0
    jump = rng.random(len(w.records)) < 0.2
    w.records.ref_id[m & jump] = rng.integers(0, cfg.n_refs, size=int((m & jump).sum()), dtype=np.int32)
    w.records.begin_pos[m & jump] = 1000
    s, o = check(w)
    assert s.get_partials()["pairs"].shape[0] > 100


def test_direct_atomic_fallback_path(monkeypatch):
    """The global-atomic histogram (used when there are too many bin tiles for LDS) must agree too."""
    force(monkeypatch, direct_atomics="1")
    check(make_workload(CONFIGS["config2"], seed=15, n_records=200_000))
    check(make_workload(CONFIGS["config1"], seed=16))


def test_shallow_and_deep_multi_mapping():
    """The front end's three window paths against the oracle: runs of a few records (fast windows), and runs of 40 records
    on average, a third of which exceed 64 records (the chunked long-run path)."""
    check(make_workload(CONFIGS["config2"], seed=22, n_records=300_000))
    check(make_workload(SynthConfig("c5w", 200_000, 3_000, 40.0, strain_level=True), seed=23))
    w, _, _ = load_golden("tiny")
    check(w)
    check(w, grouped=False)


def _interleave_mates(w: Workload) -> Workload:
    """Reorder every qName run so the records of its two mates alternate (bowtie2 -k style output)."""
    rec = w.records
    n = len(rec)
    run_id = np.concatenate([[0], np.cumsum(rec.read_key[1:] != rec.read_key[:-1])])
    mate = np.where(rec.flag & 0x40, 1, np.where(rec.flag & 0x80, 2, 0))
    # rank of a record among the records of its (run, mate), then order by (run, rank, mate)
    order0 = np.lexsort((np.arange(n), mate, run_id))
    rk = np.empty(n, dtype=np.int64)
    grp = run_id[order0] * 4 + mate[order0]
    start = np.concatenate([[True], grp[1:] != grp[:-1]])
    first_idx = np.maximum.accumulate(np.where(start, np.arange(n), 0))
    rk[order0] = np.arange(n) - first_idx
    order = np.lexsort((mate, rk, run_id))
    return Workload(w.ref_names, w.ref_len, w.taxonomy, rec.take(order), w.avg_read_len, w.options, w.name + "-interleaved")


@pytest.mark.parametrize("hits", [6.0, 45.0])
def test_interleaved_mates(hits):
    """Mates alternating inside a qName run: the general window path (targets of a run reordered by mate), and at 45 hits
    per read the long-run path with one pass per mate number."""
    w = _interleave_mates(make_workload(SynthConfig("pairs", 200_000, 2_000, hits), seed=25, paired_frac=0.9))
    mate = np.where(w.records.flag & 0x40, 1, np.where(w.records.flag & 0x80, 2, 0))
    same = w.records.read_key[1:] == w.records.read_key[:-1]
    assert int((same & (mate[1:] < mate[:-1])).sum()) > 1000   # mates really interleave
    check(w)
    check(w, grouped=False)


def test_runs_of_hundreds_of_records():
    """Reads with hundreds of records: runs cross many slots; every chunk of a run is compared with all chunks before it."""
    cfg = SynthConfig("long", 400_000, 4_000, 300.0, strain_level=True, present_frac=0.2)
    s, o = check(make_workload(cfg, seed=24))
    assert o.scalars["hits"] / o.scalars["matches"] > 100


@pytest.mark.parametrize("two", ["0", "1"])
def test_both_bucketing_variants(monkeypatch, two):
    force(monkeypatch, two_level=two)
    check(make_workload(CONFIGS["config2"], seed=20, n_records=300_000))
    check(make_workload(CONFIGS["config1"], seed=21))


def one_long_read_workload(n_hit: int, where: str = "first") -> Workload:
    """A stream over 40 000 small references with one read that maps to n_hit DIFFERENT references (as many targets: its
    slot's values run on over the following slots' positions), first or last in the file."""
    base = make_workload(SynthConfig("many-refs", 30_000, 40_000, 2.0, bin_width=1000, len_lo=1_500, len_hi=3_000,
                                     present_frac=0.5), seed=31)
    r = base.records
    long_key = np.full(n_hit, int(r.read_key.max()) + 1, dtype=np.uint64)
    parts = [Records(long_key, np.zeros(n_hit, dtype=np.uint16), np.arange(n_hit, dtype=np.int32),
                     np.full(n_hit, 10, dtype=np.int32)), r]
    if where == "last":
        parts.reverse()
    rec = Records(*(np.concatenate([getattr(x, f) for x in parts]) for f in ("read_key", "flag", "ref_id", "begin_pos")))
    return Workload(base.ref_names, base.ref_len, base.taxonomy, rec, base.avg_read_len, base.options, f"long-read-{n_hit}")


@pytest.mark.parametrize("n_hit,where", [(3_000, "first"), (9_000, "first"), (35_000, "first"), (9_000, "last")])
def test_a_read_with_thousands_of_targets(n_hit, where):
    """The values of one slot then cover the positions of the next 2 / 8 / 34 slots: pieces of one slot that fill several
    rounds of the bucket scatter, a tile that receives thousands of entries from one slot."""
    s, o = check(one_long_read_workload(n_hit, where))
    assert s.stats()["n_targets"] > n_hit


@pytest.mark.parametrize("big", ["1", "0"])
def test_one_level_bucketing_with_a_separate_scan(monkeypatch, big):
    """SLIMM_FORCE fused_scan=0: k_tile_scan + the scatter of layouts of more than 4064 tiles on small layouts, reads of
    thousands of records included: the rounds ordered by tile in LDS (k_tile_scatter_big, the default) and the direct
    rounds (SLIMM_FORCE scatter_big=0)."""
    force(monkeypatch, fused_scan="0")
    force(monkeypatch, scatter_big=big)
    check(make_workload(CONFIGS["config2"], seed=23, n_records=300_000))
    check(make_workload(CONFIGS["config1"], seed=24))
    check(one_long_read_workload(9_000))


@pytest.mark.parametrize("matrix", ["2", "0"])
def test_matrix_bucketing_and_the_direct_rounds(monkeypatch, matrix):
    """Phase B of layouts beyond the fused kernel's 4064 tiles buckets its selectors through a count matrix (one row per
    counting workgroup, no global atomics in the scatter) when a tile gets few of them; SLIMM_FORCE matrix=0 keeps the direct
    rounds, 2 = always.  Forced onto small layouts here (SLIMM_FORCE fused_scan=0), a tile cut into several work items
    included; the big layouts run it at their real sizes (the prefix and full-size tests of configs 3 and 5)."""
    force(monkeypatch, fused_scan="0")
    force(monkeypatch, matrix=matrix)
    check(make_workload(CONFIGS["config2"], seed=26, n_records=300_000))
    check(make_workload(CONFIGS["config1"], seed=27))
    check(make_workload(CONFIGS["config1"], seed=28), grouped=False)
    check(one_long_read_workload(9_000))
    check(make_workload(SynthConfig("hot", 200_000, 12, 6.0, bin_width=50, len_lo=400_000, len_hi=900_000, present_frac=0.3),
                        seed=29))


@pytest.mark.parametrize("fused", ["1", "0"])
def test_wide_tile_work_items(monkeypatch, fused):
    """SLIMM_FORCE wide_tiles=1: the form k_tile_hist takes by itself when a file brings far more entries per tile than a
    packed work item holds (1 B records on 20 k references) -- 32-bit counts, work items of up to 262 144 entries --
    on small layouts: tiles that were cut into pieces become one item, a tile of 300 000 entries still is cut."""
    force(monkeypatch, wide_tiles="1")
    force(monkeypatch, fused_scan=fused)
    check(make_workload(CONFIGS["config1"], seed=30))
    check(make_workload(CONFIGS["config2"], seed=31, n_records=300_000))
    check(make_workload(SynthConfig("hot", 200_000, 12, 6.0, bin_width=50, len_lo=400_000, len_hi=900_000, present_frac=0.3),
                        seed=32))
    check(make_workload(SynthConfig("hotter", 600_000, 6, 1.5, bin_width=200, len_lo=300_000, len_hi=400_000, present_frac=1.0),
                        seed=33), keep_bins=False)


@pytest.mark.parametrize("shift", ["13", "14"])
@pytest.mark.parametrize("fused", ["1", "0"])
def test_both_tile_sizes(monkeypatch, shift, fused):
    """tile_hist.hip is built once per tile size (8192 / 16384 bins: namespaces tiles13 / tiles14) and a context picks one
    by its layout -- the small tiles up to the fused kernel's 4064, the large ones beyond.  SLIMM_FORCE tile_shift forces either
    onto any layout: small layouts through the large tiles (and, with their full reference sets, big ones through the
    small: test_prefix_of_the_big_configs_on_small_tiles), ordered and direct rounds, packed, wide and cut work items,
    without materialised arrays."""
    force(monkeypatch, tile_shift=shift)
    force(monkeypatch, fused_scan=fused)
    check(make_workload(CONFIGS["config1"], seed=34))
    check(make_workload(CONFIGS["config1"], seed=35), grouped=False)
    check(make_workload(CONFIGS["config2"], seed=36, n_records=300_000))
    check(one_long_read_workload(9_000))
    hot = make_workload(SynthConfig("hot", 200_000, 12, 6.0, bin_width=50, len_lo=400_000, len_hi=900_000, present_frac=0.3),
                        seed=37)
    check(hot)
    check(hot, keep_bins=False)
    force(monkeypatch, wide_tiles="1")
    check(hot)
    check(make_workload(SynthConfig("hotter", 600_000, 6, 1.5, bin_width=200, len_lo=300_000, len_hi=400_000, present_frac=1.0),
                        seed=38))


@pytest.mark.parametrize("name,n", [("config3", 2_000_000), ("config5", 1_000_000)])
def test_prefix_of_the_big_configs_on_small_tiles(monkeypatch, name, n):
    """The layouts that take the 16384-bin tiles by themselves through the 8192-bin ones (9 776 / 45 000 tiles: the scan in
    stretches of 16 K tiles, two-level bucketing)."""
    force(monkeypatch, tile_shift="13")
    check(make_workload(CONFIGS[name], seed=3, n_records=n))


def test_wide_lineage_rows_fallback_path(monkeypatch):
    """32-byte lineage rows (used when a level has more than 65535 distinct taxids) must agree with the oracle too."""
    force(monkeypatch, wide_rows="1")
    check(make_workload(CONFIGS["config2"], seed=17, n_records=200_000))
    w, _, _ = load_golden("holes")
    check(w)


def test_batched_push_equals_single_push():
    w = make_workload(CONFIGS["config1"], seed=11)
    a = run_gpu(w)
    b = run_gpu(w, batch=777)
    assert a.write_abundance() == b.write_abundance()
    assert np.array_equal(a.bins(0), b.bins(0)) and np.array_equal(a.bins(2), b.bins(2))


def test_device_resident_records_and_reset_reuse():
    import torch
    w = make_workload(CONFIGS["config2"], seed=12, n_records=200_000)
    o = run_workload(w, use_qnames=False)
    s = Slimm.for_workload(w, device=0)
    dev = torch.device("cuda:0")
    key = torch.from_numpy(w.records.read_key.view(np.int64)).to(dev)
    ref = torch.from_numpy(w.records.ref_id).to(dev)
    pos = torch.from_numpy(w.records.begin_pos).to(dev)
    flag = torch.from_numpy(w.records.flag.view(np.int16)).to(dev)
    torch.cuda.synchronize()
    for _ in range(3):  # the same context, reset between runs (slimm::reset)
        s.reset()
        s.reset_cutoffs()
        s.set_records_device(key, ref, pos, flag)
        assert s.get_profiles() is not None
        assert_matches_oracle(s, o)


def test_two_shards_on_one_gpu_through_the_summary_exchange():
    """Two contexts stand in for two ranks: their coverage summaries are concatenated like an all-gather would."""
    import torch
    w = make_workload(CONFIGS["config2"], seed=18, n_records=300_000)
    o = run_workload(w, use_qnames=False)
    owner = (w.records.read_key % np.uint64(2)).astype(np.int64)
    engines = []
    for r in range(2):
        s = Slimm.for_workload(w, device=0)
        s.push_records(w.records.take(np.nonzero(owner == r)[0]))
        s.analyze_alignments()
        engines.append(s)
    gathered = torch.cat([e.coverage_summary_tensor().clone() for e in engines])
    torch.cuda.synchronize()
    parts = []
    for e in engines:
        assert e.finish_coverage_merged(gathered, 2)
        e.filter_alignments()
        parts.append(e.get_partials())
    marks = parts[0]["level_marks"] | parts[1]["level_marks"]
    pairs = np.unique(np.concatenate([parts[0]["pairs"], parts[1]["pairs"]]))
    engines[0].set_partials(parts[0]["uniq_reads_count2"] + parts[1]["uniq_reads_count2"],
                            parts[0]["lca_count"] + parts[1]["lca_count"], marks, pairs)
    engines[0].get_reads_lca_count()
    assert_matches_oracle(engines[0], o, bins=False)
    # per-rank coverage arrays are partial sums of the whole
    assert np.array_equal(engines[0].bins(0) + engines[1].bins(0), o.cov)
    assert np.array_equal(engines[0].bins(2) + engines[1].bins(2), o.uniq_cov2)


@pytest.mark.parametrize("launched", [False, True])
@pytest.mark.parametrize("with_pairs", [False, True])
def test_two_shards_on_one_gpu_device_side_partials_merge(with_pairs, launched, monkeypatch):
    """The second exchange on the device: the two contexts' partials buffers are summed like an all-reduce would.
    launched: phase B through slimm_filter_alignments_launch (no host synchronisation before the merged results are
    installed); with pairs the pair set starts tiny, so that install asks every rank to go round again."""
    import torch
    if launched and with_pairs:
        force(monkeypatch, pair_cap="64")
    if with_pairs:   # reads straddling superkingdoms agree at no level (Q4): (taxon, ref) pairs
        cfg = SynthConfig("q4shards", 150_000, 10_000, 6.0, present_frac=0.5, len_lo=200_000, len_hi=600_000)
        w = make_workload(cfg, seed=23)
        rng = np.random.default_rng(1)
        m = w.records.ref_id >= 0
        jump = rng.random(len(w.records)) < 0.2
        w.records.ref_id[m & jump] = rng.integers(0, cfg.n_refs, size=int((m & jump).sum()), dtype=np.int32)
        w.records.begin_pos[m & jump] = 1000
    else:
        w = make_workload(CONFIGS["config2"], seed=22, n_records=300_000)
    o = run_workload(w, use_qnames=False)
    owner = (w.records.read_key % np.uint64(2)).astype(np.int64)
    engines = []
    for r in range(2):
        s = Slimm.for_workload(w, device=0)
        s.prepare_summary(True)   # bitmaps written by the histogram kernels (the other test leaves this off)
        s.push_records(w.records.take(np.nonzero(owner == r)[0]))
        s.analyze_alignments()
        engines.append(s)
    gathered = torch.cat([e.coverage_summary_tensor().clone() for e in engines])
    torch.cuda.synchronize()
    for e in engines:
        assert e.finish_coverage_merged(gathered, 2)
        e.filter_alignments_launch() if launched else e.filter_alignments()
    rounds = 0
    while True:
        tensors = [e.partials_tensor() for e in engines]
        torch.cuda.synchronize()
        total = tensors[0] + tensors[1]
        for t in tensors:
            t.copy_(total)
        torch.cuda.synchronize()
        totals = [e.install_merged_partials() for e in engines]
        assert totals[0] == totals[1]
        if totals[0] is not None:
            break
        assert launched
        rounds += 1
        for e in engines:
            e.filter_alignments_launch()
    assert (rounds > 0) == (launched and with_pairs)
    local = [e.get_partials() for e in engines]
    assert totals[0] == sum(p["pairs"].shape[0] for p in local)
    assert (totals[0] > 0) == with_pairs
    pairs = np.unique(np.concatenate([p["pairs"] for p in local]))
    for e, p in zip(engines, local):
        if totals[0]:
            e.set_partials(p["uniq_reads_count2"], p["lca_count"], p["level_marks"], pairs)
        e.get_reads_lca_count()
        assert_matches_oracle(e, o, bins=False)


@pytest.mark.parametrize("shift", [None, "14"])
@pytest.mark.parametrize("world", [2, 3])
def test_shards_on_one_gpu_through_the_sliced_exchange(monkeypatch, world, shift):
    """`world` contexts stand in for ranks: the bitmap chunks are routed like an all-to-all would, the additive vectors
    summed like an all-reduce (the bin tiles do not divide evenly by 3: the last slice is partly padding).  With the tile
    size the layout picks (8192 bins here) and with the 16384-bin tiles the 20 k-reference layouts pick."""
    import torch
    if shift:
        force(monkeypatch, tile_shift=shift)
    w = make_workload(CONFIGS["config2"], seed=27, n_records=300_000)
    o = run_workload(w, use_qnames=False)
    owner = (w.records.read_key % np.uint64(world)).astype(np.int64)
    engines, summaries = [], []
    for r in range(world):
        s = Slimm.for_workload(w, device=0)
        s.prepare_summary(world)
        s.push_records(w.records.take(np.nonzero(owner == r)[0]))
        s.analyze_alignments()
        engines.append(s)
        summaries.append(s.coverage_summary_tensor().clone())
    head = engines[0].summary_head_words()
    chunk = (summaries[0].numel() - head) // world
    vecs = []
    for j, e in enumerate(engines):
        received = torch.cat([summaries[i][head + j * chunk:head + (j + 1) * chunk] for i in range(world)])
        torch.cuda.synchronize()   # torch's stream wrote `received`; the library reads it on its own stream
        vecs.append(e.merge_summary_slices(received, world, j))
    total = torch.stack([v.clone() for v in vecs]).sum(dim=0).to(torch.int32)
    for v in vecs:
        v.copy_(total)
    torch.cuda.synchronize()
    parts = []
    for e in engines:
        assert e.finish_coverage_reduced()
        e.filter_alignments()
        parts.append(e.get_partials())
    rc = engines[0].ref_columns()
    assert np.array_equal(rc["nz_cov"], o.nz_cov) and np.array_equal(rc["reads_count"], o.reads_count)
    assert np.array_equal(rc["nz_uniq_cov"], o.nz_uniq_cov) and np.array_equal(rc["uniq_reads_count"], o.uniq_reads_count)
    marks = np.bitwise_or.reduce([p["level_marks"] for p in parts])
    pairs = np.unique(np.concatenate([p["pairs"] for p in parts]))
    engines[0].set_partials(sum(p["uniq_reads_count2"] for p in parts), sum(p["lca_count"] for p in parts), marks, pairs)
    engines[0].get_reads_lca_count()
    assert_matches_oracle(engines[0], o, bins=False)


@pytest.mark.parametrize("shift", [None, "14"])
def test_single_rank_through_the_exchange_code_path(monkeypatch, shift):
    from slimm_amd.distributed import sharded_profile
    if shift:
        force(monkeypatch, tile_shift=shift)
    w = make_workload(CONFIGS["config1"], seed=19)
    o = run_workload(w)
    for mode in ("summary", "sliced", "bins"):
        s = Slimm.for_workload(w, device=0)
        s.force_exchange = True
        s.push_records(w.records)
        assert sharded_profile(s, None, None, exchange=mode) is not None
        assert_matches_oracle(s, o)


def test_sliced_exchange_on_the_direct_atomics_fallback(monkeypatch):
    """The direct-atomics fallback has no tile kernels to lay the bitmaps out in slices.  One rank: the unsliced summary
    (bitmaps from the separate kernels) is its own single slice.  More ranks: slimm_prepare_summary(n > 1) refuses and
    the driver takes the all-gather form (the same decision on every rank: it depends on the configuration only)."""
    from slimm_amd.distributed import sharded_profile
    force(monkeypatch, direct_atomics="1")
    w = make_workload(CONFIGS["config1"], seed=29)
    o = run_workload(w)
    s = Slimm.for_workload(w, device=0)
    s.force_exchange = True
    s.push_records(w.records)
    assert sharded_profile(s, None, None, exchange="sliced") is not None
    assert_matches_oracle(s, o)
    with pytest.raises(Exception):
        s.prepare_summary(4)


def test_contexts_release_their_device_memory():
    """Create / run / destroy in a loop: free device memory must not drift (every DevBuf and stream is released)."""
    import gc
    import torch
    w = make_workload(CONFIGS["config2"], seed=33, n_records=200_000)

    def one_pass():
        s = Slimm.for_workload(w, device=0)
        s.prepare_summary(2)
        s.push_records(w.records)
        assert s.get_profiles() is not None
        s.close()

    one_pass()
    gc.collect()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(20):
        one_pass()
    gc.collect()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert abs(free0 - free1) < 64 << 20, (free0, free1)   # (allocator granularity, not a per-context leak)


@pytest.mark.parametrize("n_records", [768 * 64, 768 * 64 + 1, 768 * 129 - 5, 1024 * 64, 1024 * 64 + 1, 1_500_000])
def test_slot_boundaries(n_records):
    """Record counts at, one past and short of whole slots (768 records; 1024 in earlier builds), and many slots: runs
    straddle slot boundaries, the last slot is ragged."""
    w = make_workload(CONFIGS["config2"], seed=36, n_records=n_records)
    check(w)


def test_kernel_timing_reports_every_kernel():
    w = make_workload(CONFIGS["config1"], seed=13)
    s = Slimm.for_workload(w, device=0)
    s.enable_kernel_timing(True)
    s.push_records(w.records)
    s.get_profiles()
    t = s.kernel_times()
    for k in ("k_front", "k_tile_hist", "k_pack", "k_filter", "k_tile_hist2", "k_pack2"):
        assert t[k][1] >= 1 and t[k][0] > 0.0, k


# ---------------------------------------------------------------- edge cases
def test_empty_and_all_unmapped_inputs():
    w = tiny_case()
    s = Slimm.for_workload(w, device=0)
    assert s.get_profiles() is None                     # no records at all
    s.reset()
    n = 1000
    rec = Records(np.arange(n, dtype=np.uint64), np.full(n, 4, dtype=np.uint16), np.full(n, -1, dtype=np.int32),
                  np.full(n, -1, dtype=np.int32))
    s.push_records(rec)
    assert s.get_profiles() is None                     # reference: "[WARNING] No mapped reads found"
    assert s.stats()["hits_count"] == 0


def test_mapped_flag_with_invalid_ref_is_skipped_and_unmapped_flag_with_ref_too():
    rows = [("a", 0, "*", 10), ("b", 4, "X.1", 10), ("c", 0, "X.1", 10), ("d", 0, "Y.1", 10)]
    s, o = check(_two_ref_case(rows))
    assert s.stats()["hits_count"] == 2


def test_single_record():
    s, o = check(_two_ref_case([("only", 0, "Y.1", 500)]))
    assert s.stats()["uniq_matches_count2"] == 1


def test_reference_id_out_of_range_is_an_error():
    w = tiny_case()
    w.records.ref_id[3] = 99
    s = Slimm.for_workload(w, device=0)
    s.push_records(w.records)
    s.analyze_alignments()
    with pytest.raises(capi.SlimmError) as e:
        s.finish_coverage()
    assert e.value.code == capi.E_REF_RANGE


def test_a_read_with_tens_of_thousands_of_records_is_no_error():
    """The reference has no limit on the alignments of one read (src/read_stat.hpp:116-135).  20000 records of ONE read:
    over 5 references (every record but the first five is a repeat), and with the very last record on a new reference --
    it has to be compared with all 19999 records before it."""
    n = 20000
    base = tiny_case()
    for refs in ((np.arange(n) % 5).astype(np.int32), np.concatenate([np.zeros(n - 1), [1]]).astype(np.int32)):
        rec = Records(np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint16), refs, np.full(n, 10, dtype=np.int32))
        w = Workload(base.ref_names, base.ref_len, base.taxonomy, rec, base.avg_read_len, base.options, "one-read")
        s, o = check(w)
        assert s.stats()["matches_count"] == 1 and s.stats()["hits_count"] == n
    # ... and the same read with its two mates alternating (one pass per mate number over the run)
    flag = np.where(np.arange(n) % 2 == 0, 0x41, 0x81).astype(np.uint16)
    rec = Records(np.zeros(n, dtype=np.uint64), flag, (np.arange(n) % 5).astype(np.int32), np.full(n, 10, dtype=np.int32))
    w = Workload(base.ref_names, base.ref_len, base.taxonomy, rec, base.avg_read_len, base.options, "one-pair")
    s, o = check(w)
    assert s.stats()["matches_count"] == 2


def test_ragged_tile_boundaries():
    """Record counts around the 2048-record tile size, reads straddling tiles."""
    base = make_workload(SynthConfig("rag", 9000, 40, 5.0, bin_width=100, len_lo=5_000, len_hi=50_000,
                                     present_frac=0.5), seed=14)
    for n in (1, 2, 63, 64, 65, 255, 256, 257, 2047, 2048, 2049, 4096, 4097, 8191):
        w = Workload(base.ref_names, base.ref_len, base.taxonomy, base.records.take(np.arange(n)), base.avg_read_len,
                     base.options, f"rag{n}")
        check(w)
        check(w, grouped=False)


# ---------------------------------------------------------------- BASELINE.json full size: size-independent properties
def _order_preserving_interleave(rec: Records, seed: int) -> Records:
    """Permute records so reads interleave while the relative order inside each read name is kept."""
    n = len(rec)
    rng = np.random.default_rng(seed)
    pos = rng.permutation(n)
    a = np.lexsort((np.arange(n), rec.read_key))
    b = np.lexsort((pos, rec.read_key))
    newpos = np.empty(n, dtype=np.int64)
    newpos[a] = pos[b]
    inv = np.empty(n, dtype=np.int64)
    inv[newpos] = np.arange(n)
    return rec.take(inv)


def _device_interleave(key_t, seed: int):
    """_order_preserving_interleave on the device (torch): for int64 keys on the GPU, the gather index `inv` such that
    record j of the interleaved stream is record inv[j] of the input -- reads interleave at random, the relative order
    of the records of one read name is kept (what Q1 depends on: src/read_stat.hpp:116-135)."""
    import torch
    n = key_t.numel()
    g = torch.Generator(device=key_t.device)
    g.manual_seed(seed)
    pos = torch.randperm(n, device=key_t.device, generator=g)
    a = torch.sort(key_t, stable=True).indices                 # by (key, index)
    p1 = torch.argsort(pos)
    b = p1[torch.sort(key_t[p1], stable=True).indices]         # by (key, pos)
    newpos = torch.empty(n, dtype=torch.int64, device=key_t.device)
    newpos[a] = pos[b]
    del a, b, p1, pos
    inv = torch.empty(n, dtype=torch.int64, device=key_t.device)
    inv[newpos] = torch.arange(n, device=key_t.device)
    return inv


def _pack_keys_device(key_t, flag_t):
    """slimm_pack_key on tensors: (key & (2^61 - 1)) | mate << 61 | unmapped << 63."""
    import torch
    f = flag_t.to(torch.int64) & 0xffff
    mate = torch.where((f & 0x40) != 0, 1, torch.where((f & 0x80) != 0, 2, 0)).to(torch.int64)
    return (key_t & ((1 << 61) - 1)) | (mate << 61) | (((f & 0x4) != 0).to(torch.int64) << 63)


def _full_size_invariants(w: Workload, s: Slimm, permutation: bool):
    """Size-independent properties of one finished run (nothing here needs the oracle, which would take minutes)."""
    st = s.stats()
    rc = s.ref_columns()
    cov, ucov, ucov2 = s.bins(0), s.bins(1), s.bins(2)
    mapped = int((((w.records.flag & 4) == 0) & (w.records.ref_id >= 0)).sum())
    assert st["hits_count"] == mapped % 2**32                      # uint32 like the reference's counter (Q11)
    assert int(cov.sum(dtype=np.uint64)) == st["n_targets"] == int(rc["reads_count"].sum(dtype=np.uint64))
    assert int(ucov.sum(dtype=np.uint64)) == st["uniq_matches_count"] == int(rc["uniq_reads_count"].sum(dtype=np.uint64))
    assert int(ucov2.sum(dtype=np.uint64)) == st["uniq_matches_count2"] == int(rc["uniq_reads_count2"].sum(dtype=np.uint64))
    assert np.all(ucov <= cov)
    off = np.concatenate([[0], np.cumsum(rc["nbins"].astype(np.int64))])
    assert np.array_equal(np.add.reduceat((cov != 0).astype(np.int64), off[:-1]), rc["nz_cov"])
    assert np.array_equal(np.add.reduceat((ucov != 0).astype(np.int64), off[:-1]), rc["nz_uniq_cov"])
    assert np.array_equal(np.add.reduceat(cov.astype(np.int64), off[:-1]), rc["reads_count"])
    assert np.array_equal(np.add.reduceat(ucov.astype(np.int64), off[:-1]), rc["uniq_reads_count"])
    assert np.array_equal(np.add.reduceat(ucov2.astype(np.int64), off[:-1]), rc["uniq_reads_count2"])
    assert np.array_equal(np.add.reduceat((ucov2 != 0).astype(np.int64), off[:-1]), rc["nz_uniq_cov2"])
    # every target of an invalid reference is dropped by the filter: nothing of it may reach uniq_cov2
    assert not np.any(rc["uniq_reads_count2"][rc["valid"] == 0])
    # an independent count of the reads: distinct (qName, mate) among the mapped records (src/slimm.hpp:204-211)
    m =((w.records.flag & 4) == 0) & (w.records.ref_id >= 0)
    mate = np.where(w.records.flag[m] & 0x40, 1, np.where(w.records.flag[m] & 0x80, 2, 0)).astype(np.uint64)
    ident = (w.records.read_key[m] << np.uint64(2)) | mate
    n_reads = int(np.unique(ident).shape[0])
    assert st["matches_count"] == n_reads % 2**32
    # an independent count of the targets: distinct (read, reference) pairs
    pair_hash = ident * np.uint64(0x9E3779B97F4A7C15) + w.records.ref_id[m].astype(np.uint64)
    assert abs(int(np.unique(pair_hash).shape[0]) - st["n_targets"]) <= 2   # (64-bit mixing: collisions are ~1e-3 events)
    # reads whose targets all fail the filter vanish; the rest are unique-after-filter or counted at their LCA
    direct = s.taxon_counts(0)
    assert st["uniq_matches_count2"] + sum(direct.values()) <= st["matches_count"]
    assert st["uniq_matches_count2"] >= st["uniq_matches_count"] - int(rc["uniq_reads_count"][rc["valid"] == 0].sum())
    # profile read counts add up to the matched reads (the `0*` row is the remainder)
    from oracle.binding import parse_profile
    rows = parse_profile(s.write_abundance())
    assert sum(v[1] for v in rows.values()) == st["matches_count"]
    assert sum(v[0] for v in rows.values()) == pytest.approx(100.0, abs=1e-3)
    if permutation:
        # permutation invariance: interleaving reads (order inside a read kept) + the sort path gives identical results
        w2 = Workload(w.ref_names, w.ref_len, w.taxonomy, _order_preserving_interleave(w.records, 5), w.avg_read_len,
                      w.options, "interleaved", grouped=False)
        s2 = run_gpu(w2)
        assert np.array_equal(s2.bins(0), cov) and np.array_equal(s2.bins(1), ucov) and np.array_equal(s2.bins(2), ucov2)
        assert s2.taxon_counts(1) == s.taxon_counts(1) and s2.children_pairs(1) == s.children_pairs(1)
        assert s2.write_abundance() == s.write_abundance()
    return st


def assert_equals_dense_mt(s: Slimm, d: dict, bins=None):
    """Bit-exact at ANY size: the finished run of the HIP path against the all-core dense restatement
    (oracle/slimm_dense_mt.cpp, itself compared with the oracle on >= 2 M records of every configuration with its full
    reference set in tests/test_dense_mt.py): the scalars, every per-reference column, the number of valid references,
    the direct LCA count of every taxon, and all three coverage arrays -- bin by bin, and through the position-weighted
    64-bit checksum both sides compute on their own.  Semantics: src/slimm.hpp:219-257, 380-391, 536-557.
    bins: the three arrays of the GPU side when they are not s.bins(k) (a group: the members' partial arrays summed)."""
    from oracle.binding import bin_checksum
    st = s.stats()
    assert (st["hits_count"], st["matches_count"], st["uniq_matches_count"], st["uniq_matches_count2"], st["n_valid"]) == (
        d["hits"] % 2**32, d["matches"] % 2**32, d["uniq_matches"] % 2**32, d["uniq_matches2"] % 2**32, d["n_valid"])
    assert st["total_bins"] == d["total_bins"]
    rc = s.ref_columns()
    for k in ("reads_count", "uniq_reads_count", "uniq_reads_count2", "nz_cov", "nz_uniq_cov"):
        assert np.array_equal(rc[k], d[k]), f"per-reference column {k} differs from the dense restatement"
    assert int(rc["valid"].sum()) == d["n_valid"]
    assert s.taxon_counts(0) == d["lca_direct"], "direct LCA counts differ from the dense restatement"
    if "profile" in d:
        # the scalar tail, independently (dmt_profile: written from src/slimm.hpp:560-610, 733-843 with the reference's
        # containers, pinned to the oracle in tests/test_dense_mt.py): propagated counts, children sets, profile rows
        from oracle.binding import parse_profile
        assert s.taxon_counts(1) == d["taxon_count"], "propagated per-taxon counts differ from the dense restatement"
        assert s.children_pairs(1) == d["taxon_children"], "children sets differ from the dense restatement"
        rows = parse_profile(s.write_abundance())
        assert set(rows) == set(d["profile"])
        for key, (ab, reads) in d["profile"].items():
            assert rows[key][1] == reads, key
            assert rows[key][0] == pytest.approx(ab, rel=2e-5, abs=1e-6), key
    for i, k in enumerate(("cov", "uniq_cov", "uniq_cov2")):
        got = s.bins(i) if bins is None else bins[i]
        assert bin_checksum(got) == d["checksums"][i], f"checksum of {k} differs"
        if k in d:
            if not np.array_equal(got, d[k]):
                bad = np.nonzero(got != d[k])[0]
                raise AssertionError(f"{k}: {bad.size} bins differ, first at {bad[0]}: {got[bad[0]]} != {d[k][bad[0]]}")
        del got


def test_full_size_config2_bit_exact():
    """BASELINE.json configs[1] at full size (10 M records, 5 k refs): invariants, permutation invariance through the sort
    path, and every integer against the dense restatement."""
    from oracle.binding import dense_mt_run
    w = make_workload(CONFIGS["config2"], seed=1)
    s = run_gpu(w)
    _full_size_invariants(w, s, permutation=True)
    assert_equals_dense_mt(s, dense_mt_run(w, want_bins=True))
    # ... and EVERYTHING against the oracle itself at this size (one thread, the reference's containers: a few seconds for
    # 10 M records): the propagated per-taxon counts and children of src/slimm.hpp:560-610, the cut-offs, the profile rows
    assert_matches_oracle(s, run_workload(w, use_qnames=False))


def test_full_size_config2_and_5_as_run_marked_records():
    """The 8-byte run-marked form (slimm_push_records_marked) at full size: configs[1] (10 M records) and configs[4]
    (100 M records, 40 hits per read: hash-table and long-run paths) against the dense restatement, every integer."""
    from oracle.binding import dense_mt_run
    for name in ("config2", "config5"):
        w = make_workload(CONFIGS[name], seed=1)
        s = Slimm.for_workload(w, device=0)
        s.push_records_marked(w.records, batch=25_000_000)
        assert s.get_profiles() is not None
        assert_equals_dense_mt(s, dense_mt_run(w, want_bins=True))
        s.close()


def test_full_size_config3_bit_exact():
    """BASELINE.json configs[2] at full size: 100 M records, 20 k references, mean 8 hits per read -- as pushed four-array
    records, as PACKED 16-byte records (slimm_push_records_packed: the form bench.py measures; its identity is the low 61
    bits of the key, which the dense restatement gets too), and the same records in ANY order (reads interleaved at
    random, file order kept inside a read) through the device-side grouping (record_order = SLIMM_ORDER_ANY,
    src/slimm.hpp:204-211): every integer against the dense restatement each time."""
    import torch
    from oracle.binding import dense_mt_run
    w = make_workload(CONFIGS["config3"], seed=1)
    s = run_gpu(w)
    st = _full_size_invariants(w, s, permutation=False)
    assert st["n_records"] == 100_000_000 and st["total_bins"] > 60_000_000
    d = dense_mt_run(w, want_bins=True)
    assert_equals_dense_mt(s, d)
    profile = s.write_abundance()
    # the oracle itself on all 100 M records (one thread, ~40 s): what the dense restatement does not have -- the propagated
    # per-taxon counts and children (src/slimm.hpp:560-610), the cut-offs, the abundances, the profile rows
    assert_matches_oracle(s, run_workload(w, use_qnames=False, collect_bins=False), bins=False)
    s.close()
    # packed, pushed from host memory in batches
    w61 = _mask61(w)
    d61 = d if np.array_equal(w61.records.read_key, w.records.read_key) else dense_mt_run(w61, want_bins=True)
    s = Slimm.for_workload(w61, device=0)
    s.push_records_packed(w61.records, batch=25_000_000)
    assert s.get_profiles() is not None
    assert_equals_dense_mt(s, d61)
    s.close()
    # any order: interleaved on the device, resident, through the grouping
    dev = torch.device("cuda:0")
    r = w.records
    key = torch.from_numpy(r.read_key.view(np.int64)).to(dev)
    inv = _device_interleave(key, 5)
    key = key[inv]
    ref = torch.from_numpy(r.ref_id).to(dev)[inv]
    pos = torch.from_numpy(r.begin_pos).to(dev)[inv]
    flag = torch.from_numpy(r.flag.view(np.int16)).to(dev)[inv]
    assert not torch.equal(key[:1000], torch.from_numpy(r.read_key[:1000].view(np.int64)).to(dev))
    del inv
    torch.cuda.synchronize()
    s = Slimm.for_workload(w, device=0, grouped=False)
    s.set_records_device(key, ref, pos, flag)
    assert s.get_profiles() is not None
    assert s.stats()["n_records"] == 100_000_000
    assert_equals_dense_mt(s, d)
    assert s.write_abundance() == profile
    # ... and packed in any order
    s.reset(); s.reset_cutoffs()
    pk = _pack_keys_device(key, flag)
    torch.cuda.synchronize()            # (torch's stream made pk; the library reads it on a stream of its own)
    s.set_records_device_packed(pk, ref, pos)
    assert s.get_profiles() is not None
    assert_equals_dense_mt(s, d61)
    s.close()


def test_full_size_config5_bit_exact():
    """BASELINE.json configs[4] at full size: 100 M records, 50 k strain-level references, mean 40 hits per read
    (two-level bucketing, the long-run and deep-LCA paths at their real sizes)."""
    from oracle.binding import dense_mt_run
    w = make_workload(CONFIGS["config5"], seed=1)
    s = run_gpu(w)
    st = _full_size_invariants(w, s, permutation=False)
    assert st["n_records"] == 100_000_000 and st["total_bins"] > 150_000_000
    assert len(s.taxon_counts(0)) > 100   # strain-level database: LCAs at levels 0 / 1 are frequent
    assert_equals_dense_mt(s, dense_mt_run(w, want_bins=True))
    # the oracle itself on all 100 M records (one thread): the propagated per-taxon counts and children
    # (src/slimm.hpp:560-610), the cut-offs, the abundances and the profile rows at this size too
    assert_matches_oracle(s, run_workload(w, use_qnames=False, collect_bins=False), bins=False)
    s.close()


def test_full_size_config4_one_context_and_a_group_of_four():
    """BASELINE.json configs[3]: THE 1 B-record stream of `bench.py --config config4` (20 k refs, mean 8 hits per read;
    100 chunks of 10 M records, each a grouped file of whole reads), (a) through one context, the chunks copied straight
    into HBM, and (b) dealt by read over a group of four contexts (slimm_group_*: what `slimm --devices` and an 8-GPU
    node run; on the one device of the test box the two collectives are copies) -- both bit-exact against the dense
    restatement run on the same 1 B records on the host cores."""
    import torch
    from oracle.binding import dense_mt_run
    from slimm_amd.profiler import SlimmGroup
    from slimm_amd.synth import stream_chunks
    cfg = CONFIGS["config4"]
    n_stream = int(os.environ.get("SLIMM_TEST_CONFIG4_RECORDS", cfg.n_records))
    chunk = 10_000_000
    dev = torch.device("cuda:0")
    host = Records(np.empty(n_stream, dtype=np.uint64), np.empty(n_stream, dtype=np.uint16),
                   np.empty(n_stream, dtype=np.int32), np.empty(n_stream, dtype=np.int32))
    key = torch.empty(n_stream, dtype=torch.int64, device=dev)
    ref = torch.empty(n_stream, dtype=torch.int32, device=dev)
    pos = torch.empty(n_stream, dtype=torch.int32, device=dev)
    flag = torch.empty(n_stream, dtype=torch.int16, device=dev)
    w0 = None
    for c, wc in stream_chunks(cfg, 1, n_stream, chunk, threads=min(16, os.cpu_count() or 1)):
        lo, hi = c * chunk, c * chunk + len(wc.records)
        r = wc.records
        host.read_key[lo:hi], host.flag[lo:hi], host.ref_id[lo:hi], host.begin_pos[lo:hi] = r.read_key, r.flag, r.ref_id, r.begin_pos
        key[lo:hi] = torch.from_numpy(r.read_key.view(np.int64))
        ref[lo:hi] = torch.from_numpy(r.ref_id)
        pos[lo:hi] = torch.from_numpy(r.begin_pos)
        flag[lo:hi] = torch.from_numpy(r.flag.view(np.int16))
        if w0 is None:
            w0 = wc
    assert hi == n_stream
    w = Workload(w0.ref_names, w0.ref_len, w0.taxonomy, host, w0.avg_read_len, w0.options, "config4-stream")
    d = dense_mt_run(w, want_bins=True, want_profile=True)   # (with the propagation and the profile rows: the 1 B-record case has
    assert d["hits"] > 0.97 * n_stream                        # no oracle run to compare those with)
    assert len(d["profile"]) > 3 and len(d["taxon_count"]) > 1000
    # (a) one context
    s = Slimm.for_workload(w, device=0)
    torch.cuda.synchronize()
    s.set_records_device(key, ref, pos, flag)
    assert s.get_profiles() is not None
    assert s.stats()["n_records"] == n_stream
    assert_equals_dense_mt(s, d)
    profile = s.write_abundance()
    # (a2) the same resident records PACKED (16 bytes each: what bench.py's headline runs on)
    k61 = host.read_key & np.uint64((1 << 61) - 1)
    same61 = bool(np.array_equal(k61, host.read_key))
    w61 = w if same61 else Workload(w.ref_names, w.ref_len, w.taxonomy, Records(k61, host.flag, host.ref_id, host.begin_pos),
                                    w.avg_read_len, w.options, "config4-stream-61")
    d61 = d if same61 else dense_mt_run(w61, want_bins=True, want_profile=True)
    del k61
    s.reset(); s.reset_cutoffs()
    pk = _pack_keys_device(key, flag)
    torch.cuda.synchronize()            # (torch's stream made pk; the library reads it on a stream of its own)
    s.set_records_device_packed(pk, ref, pos)
    assert s.get_profiles() is not None
    assert_equals_dense_mt(s, d61)
    s.close()
    del pk
    # (a3) ANY order: every chunk's reads interleaved at random on the device (file order kept inside a read; reads never
    # span chunks), through the device-side grouping of record_order = SLIMM_ORDER_ANY -- the same integers again
    for lo in range(0, n_stream, chunk):
        hi = min(n_stream, lo + chunk)
        inv = _device_interleave(key[lo:hi], 1000 + lo // chunk)
        key[lo:hi] = key[lo:hi][inv]
        ref[lo:hi] = ref[lo:hi][inv]
        pos[lo:hi] = pos[lo:hi][inv]
        flag[lo:hi] = flag[lo:hi][inv]
    del inv
    torch.cuda.synchronize()
    s = Slimm.for_workload(w, device=0, grouped=False)
    s.set_records_device(key, ref, pos, flag)
    assert s.get_profiles() is not None
    assert s.stats()["n_records"] == n_stream
    assert_equals_dense_mt(s, d)
    assert s.write_abundance() == profile
    s.close()
    del key, ref, pos, flag
    torch.cuda.empty_cache()
    # (b) a group of four on the one device, the stream pushed chunk by chunk from host memory
    g = SlimmGroup(w, [0, 0, 0, 0])
    for lo in range(0, n_stream, chunk):
        g.push_records(host.take(slice(lo, min(n_stream, lo + chunk))))
    assert g.get_profiles()
    shares = [g.member(i).stats()["n_records"] for i in range(4)]
    assert sum(shares) == n_stream and min(shares) >= n_stream // 4 - 2 * chunk
    summed = [sum(g.member(i).bins(k).astype(np.uint64) for i in range(4)).astype(np.uint32) for k in range(3)]
    assert_equals_dense_mt(g.member(0), d, bins=summed)
    assert g.member(0).write_abundance() == profile
    g.close()


@pytest.mark.parametrize("name,n", [("config3", 3_000_000), ("config5", 2_000_000), ("config4", 2_000_000)])
def test_prefix_of_the_big_configs_with_their_full_reference_sets(name, n):
    """The first n records of configs[2] / [4] / [3] with ALL 20 k / 50 k references of the configuration (the big-table
    paths: > 4096 bin tiles -> separate tile scan, > 16 K tiles -> two-level bucketing, wide per-level indices)
    compared with the oracle bit for bit."""
    check(make_workload(CONFIGS[name], seed=2, n_records=n))


@pytest.mark.parametrize("hits", [6.0, 9.0, 14.0, 25.0])
def test_medium_depth_streams(hits):
    """6 - 25 records per qName run: windows of a few runs each, a growing share of runs of 64 records or more."""
    w = make_workload(SynthConfig("mid", 300_000, 300, hits, bin_width=200, len_lo=20_000, len_hi=200_000,
                                  strain_level=True), seed=51)
    check(w)
    check(w, grouped=False)


@pytest.mark.parametrize("mk", [tiny_case, holes_case, lambda: make_workload(CONFIGS["config1"], seed=61),
                                lambda: make_workload(CONFIGS["config2"], seed=62, n_records=400_000),
                                lambda: make_workload(SynthConfig("hot", 200_000, 12, 6.0, bin_width=50, len_lo=400_000,
                                                                  len_hi=900_000, present_frac=0.3), seed=63)])
def test_without_materialised_coverage_arrays(mk):
    """slimm_keep_bins(0): the tile kernels take sums, non-zero counts and per-taxon counts from the finished tiles in LDS
    and do not write them to HBM (tiles cut into pieces still are) -- every result but the arrays themselves is the
    same, and asking for the arrays is an error."""
    w = mk()
    s, o = check(w, keep_bins=False)
    rc = s.ref_columns()
    assert np.array_equal(rc["nz_uniq_cov2"], o.nz_uniq_cov2)
    with pytest.raises(capi.SlimmError):
        s.bins(0)
    with pytest.raises(capi.SlimmError):
        s.bins(2)
    with pytest.raises(capi.SlimmError):
        s.coverage_tensor()
    check(w, grouped=False, keep_bins=False)


# ---------------------------------------------------------------- streamed ingest (slimm_push_records_async)
@pytest.mark.parametrize("batch", [1000, 4096, 1 << 20])
def test_streamed_ingest_through_the_staging_sets(batch):
    """Records handed over through the two page-locked staging sets (ragged last batch, sets reused many times), phase A
    ordered behind the copies on the device: the same results as the synchronous push."""
    w = make_workload(CONFIGS["config1"], seed=5)
    o = run_workload(w)
    s = Slimm.for_workload(w, device=0)
    s.push_records_streamed(w.records, batch=batch)
    assert s.get_profiles() is not None
    assert_matches_oracle(s, o)
    # the context is reusable: a second file through the asynchronous form on caller-owned arrays
    s.reset()
    r = w.records
    s.push_records_async(r.read_key, r.ref_id, r.begin_pos, r.flag)
    s.push_wait()
    assert s.get_profiles() is not None
    assert_matches_oracle(s, o)


def test_reset_with_copies_on_their_way():
    w = make_workload(CONFIGS["config1"], seed=6)
    o = run_workload(w)
    s = Slimm.for_workload(w, device=0)
    r = w.records
    half = len(r) // 2
    s.push_records_async(r.read_key[:half], r.ref_id[:half], r.begin_pos[:half], r.flag[:half])
    s.reset()                       # waits for the copies; nothing of them is left
    s.push_records_streamed(r, batch=3000)
    assert s.get_profiles() is not None
    assert_matches_oracle(s, o)


# ---------------------------------------------------------------- check words (slimm_push_records_checked)
def _check_words(w):
    """A second hash of every record's name: here simply a mix of the key (equal names <=> equal keys in the synth)."""
    k = w.records.read_key.astype(np.uint64)
    return ((k * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(32)).astype(np.uint32)


@pytest.mark.parametrize("grouped", [True, False])
def test_check_words_change_nothing_when_names_and_keys_agree(grouped):
    w = make_workload(CONFIGS["config1"], seed=61, shuffled=not grouped)
    o = run_workload(w)
    s = Slimm.for_workload(w, device=0, grouped=grouped)
    s.push_records_checked(w.records, _check_words(w), batch=7000)
    assert s.get_profiles() is not None
    assert_matches_oracle(s, o)


@pytest.mark.parametrize("grouped", [True, False])
@pytest.mark.parametrize("where", ["inside a run", "run of 200 records", "first record of the stream", "last record"])
def test_two_names_under_one_key_are_reported(grouped, where):
    """Two different read names that were given one key: their records carry different check words.  Grouped input sees
    them when they are adjacent, any other order after the device sort, however far apart they were in the file."""
    w = make_workload(CONFIGS["config1"], seed=62, shuffled=not grouped)
    r = w.records
    chk = _check_words(w)
    starts = np.nonzero(np.concatenate([[True], r.read_key[1:] != r.read_key[:-1]]))[0]
    if where == "run of 200 records":
        i = 3000
        r.read_key[i:i + 200] = r.read_key[i]
        r.flag[i:i + 200] &= np.uint16(0xffff & ~(0x4 | 0x40 | 0x80))
        r.ref_id[i:i + 200] = np.abs(r.ref_id[i:i + 200])
        chk[i:i + 200] = chk[i]
        victim = i + 150
    elif where == "first record of the stream":
        r.read_key[1] = r.read_key[0]
        chk[1] = chk[0]
        victim = 1 if grouped else 0
    elif where == "last record":
        r.read_key[-1] = r.read_key[-2]
        chk[-1] = chk[-2]
        victim = len(r) - 1
    elif grouped:
        j = next(k for k in range(len(starts) - 1) if starts[k + 1] - starts[k] >= 3 and starts[k] > 5000)
        victim = starts[j] + 1
    else:
        victim = 6000
    if not grouped:   # far apart in the file: give the victim's key to a record elsewhere too (same key, other "name")
        other = (victim + len(r) // 2) % len(r)
        r.read_key[other] = r.read_key[victim]
        chk[other] = chk[victim]
    chk[victim] ^= np.uint32(0x5a5a5a5a)   # the record now belongs to "another name" with the same key
    s = Slimm.for_workload(w, device=0, grouped=grouped)
    s.push_records_checked(r, chk, batch=9000)
    with pytest.raises(capi.SlimmError) as e:
        s.get_profiles()
    assert e.value.code == capi.E_KEY_COLLISION
    # the context is usable again after a reset
    s.reset()
    w2 = make_workload(CONFIGS["config1"], seed=61, shuffled=not grouped)
    s.push_records_checked(w2.records, _check_words(w2))
    assert s.get_profiles() is not None


def test_checked_and_unchecked_pushes_do_not_mix():
    w = make_workload(CONFIGS["config1"], seed=63, n_records=4000)
    s = Slimm.for_workload(w, device=0)
    half = w.records.take(np.arange(2000))
    s.push_records(half)
    with pytest.raises(capi.SlimmError):
        s.push_records_checked(half, _check_words(w)[:2000])
    s.reset()
    s.push_records_checked(half, _check_words(w)[:2000])
    with pytest.raises(capi.SlimmError):
        s.push_records(half)


# ---------------------------------------------------------------- packed records (slimm_push_records_packed: 16 B/record)
def _mask61(w: Workload) -> Workload:
    """The same workload with 61-bit read keys (what a producer of packed records hashes names to)."""
    r = w.records
    rec = Records(r.read_key & np.uint64((1 << 61) - 1), r.flag, r.ref_id, r.begin_pos, r.qname)
    return Workload(w.ref_names, w.ref_len, w.taxonomy, rec, w.avg_read_len, w.options, w.name + "-61", grouped=w.grouped)


def test_pack_key_layout():
    k = np.array([0, 1, (1 << 61) - 1, (1 << 62) - 1, 0x123456789abcdef0], dtype=np.uint64)
    for flag, top in ((0, 0), (0x4, 4), (0x40, 1), (0x80, 2), (0xc0, 1), (0x44, 5), (0x900, 0), (0x84 | 0x100, 6)):
        got = Slimm.pack_keys(k, np.full(k.shape, flag, dtype=np.uint16))
        assert np.array_equal(got & np.uint64((1 << 61) - 1), k & np.uint64((1 << 61) - 1))
        assert np.all((got >> np.uint64(61)) == np.uint64(top)), flag


@pytest.mark.parametrize("grouped", [True, False])
@pytest.mark.parametrize("how", ["sync", "batches", "streamed"])
def test_packed_records_equal_the_four_array_form(grouped, how):
    """The three flag bits the record loop reads (src/slimm.hpp:197, 205-208) folded into the key's top bits: every
    result equals the oracle's on the four-array records -- through the single-pass front end and through the sort
    path, pushed in one piece, in ragged batches and through the staging sets."""
    w = _mask61(make_workload(CONFIGS["config1"], seed=71, shuffled=not grouped))
    o = run_workload(w)
    s = Slimm.for_workload(w, device=0, grouped=grouped)
    r = w.records
    if how == "sync":
        s.push_records_packed(r)
    elif how == "batches":
        s.push_records_packed(r, batch=1777)
    else:
        s.push_records_packed_streamed(r, batch=3000)
    assert s.get_profiles() is not None
    assert_matches_oracle(s, o)
    # the context takes the other form for its next file (and the forms do not mix within one)
    s.reset()
    s.push_records(r)
    with pytest.raises(capi.SlimmError):
        s.push_records_packed(r)
    assert s.get_profiles() is not None
    assert_matches_oracle(s, o)
    s.reset()
    s.push_records_packed(r, batch=5000)
    with pytest.raises(capi.SlimmError):
        s.push_records(r)
    assert s.get_profiles() is not None
    assert_matches_oracle(s, o)


@pytest.mark.parametrize("mk", [lambda: make_workload(CONFIGS["config2"], seed=72, n_records=400_000),
                                lambda: make_workload(SynthConfig("c5p", 300_000, 3_000, 40.0, strain_level=True), seed=73),
                                lambda: _interleave_mates(make_workload(SynthConfig("pairs", 150_000, 2_000, 6.0), seed=74,
                                                                        paired_frac=0.9))])
def test_packed_records_on_larger_streams(mk):
    """Short runs, runs of 64 records and more (the hash-table and long-run paths), interleaved mates -- packed."""
    w = _mask61(mk())
    o = run_workload(w, use_qnames=False)
    s = Slimm.for_workload(w, device=0)
    s.push_records_packed(w.records, batch=100_000)
    assert s.get_profiles() is not None
    assert_matches_oracle(s, o)


@pytest.mark.parametrize("grouped", [True, False])
def test_packed_records_resident_on_the_device(grouped):
    import torch
    w = _mask61(make_workload(CONFIGS["config2"], seed=75, n_records=250_000, shuffled=not grouped))
    o = run_workload(w, use_qnames=False)
    r = w.records
    dev = torch.device("cuda:0")
    t = [torch.from_numpy(a).to(dev) for a in (Slimm.pack_keys(r.read_key, r.flag).view(np.int64), r.ref_id, r.begin_pos)]
    torch.cuda.synchronize()
    s = Slimm.for_workload(w, device=0, grouped=grouped)
    for _ in range(2):
        s.reset()
        s.reset_cutoffs()
        s.set_records_device_packed(*t)
        assert s.get_profiles() is not None
        assert_matches_oracle(s, o)


# ---------------------------------------------------------------- run-marked records: 8 bytes each, no names on the device
@pytest.mark.parametrize("how", ["sync", "batches", "async", "streamed"])
def test_marked_records_equal_the_four_array_form(how):
    """For input grouped by name the read identity is the qName run a record lies in (src/slimm.hpp:204-211 with the
    records of a name adjacent): 8-byte records that only say where a run starts give every result of the oracle on the
    four-array records -- pushed in one piece, in ragged batches (runs straddle them), asynchronously and through the
    staging sets."""
    w = make_workload(CONFIGS["config1"], seed=91)
    o = run_workload(w)
    s = Slimm.for_workload(w, device=0)
    r = w.records
    words = Slimm.mark_words(r.read_key, r.flag, r.ref_id)
    if how == "sync":
        s.push_records_marked(r)
    elif how == "batches":
        s.push_records_marked(r, batch=1777)
    elif how == "async":
        for a in range(0, len(r), 2500):
            s.push_records_marked_async(words[a:a + 2500].copy(), r.begin_pos[a:a + 2500].copy())
            s.push_wait()
    else:
        s.push_records_marked_streamed(r, batch=3000)
    assert s.get_profiles() is not None
    assert_matches_oracle(s, o)
    with pytest.raises(capi.SlimmError):
        s.check_grouping()                       # no names to check
    # the context takes the other forms for its next files (and the forms do not mix within one)
    s.reset()
    s.push_records(r)
    with pytest.raises(capi.SlimmError):
        s.push_records_marked(r)
    assert s.get_profiles() is not None
    assert_matches_oracle(s, o)
    s.reset()
    s.push_records_marked(r, batch=5000)
    with pytest.raises(capi.SlimmError):
        s.push_records(r)
    with pytest.raises(capi.SlimmError):
        s.push_records_packed(r)
    assert s.get_profiles() is not None
    assert_matches_oracle(s, o)


@pytest.mark.parametrize("mk", [lambda: make_workload(CONFIGS["config2"], seed=92, n_records=400_000),
                                lambda: make_workload(SynthConfig("c5p", 300_000, 3_000, 40.0, strain_level=True), seed=93),
                                lambda: _interleave_mates(make_workload(SynthConfig("pairs", 150_000, 2_000, 6.0), seed=94,
                                                                        paired_frac=0.9)),
                                lambda: one_long_read_workload(9_000),
                                lambda: one_long_read_workload(35_000, "last"),
                                lambda: make_workload(SynthConfig("long", 400_000, 4_000, 300.0, strain_level=True,
                                                                  present_frac=0.2), seed=95),
                                lambda: make_workload(CONFIGS["config2"], seed=96, n_records=768 * 129 - 5)])
def test_marked_records_on_larger_streams(mk):
    """Short runs, runs of 64 records and more (hash table, staged and global long-run paths), interleaved mates, one
    read with tens of thousands of records, a ragged last slot -- run-marked."""
    w = mk()
    o = run_workload(w, use_qnames=False)
    s = Slimm.for_workload(w, device=0)
    s.push_records_marked(w.records, batch=100_000)
    assert s.get_profiles() is not None
    assert_matches_oracle(s, o)


def test_marked_records_edge_cases():
    """Unmapped records inside and between runs, a reference id out of range, an all-unmapped file, a context created
    for any record order."""
    w = make_workload(CONFIGS["config1"], seed=97)
    r = w.records
    r.flag[::7] |= 4                              # unmapped records keep their place in their runs
    r.ref_id[5::11] = -1
    o = run_workload(w)
    s = Slimm.for_workload(w, device=0)
    s.push_records_marked(r, batch=999)
    assert s.get_profiles() is not None
    assert_matches_oracle(s, o)
    s.reset()
    words = Slimm.mark_words(r.read_key, r.flag, r.ref_id)
    words[100] = (words[100] & np.uint32(0xe0000000)) | np.uint32(len(w.ref_names) + 5)      # no such reference
    s.push_records_marked(r, words=words)
    s.analyze_alignments()
    with pytest.raises(capi.SlimmError) as e:
        s.finish_coverage()
    assert e.value.code == capi.E_REF_RANGE
    s.reset()
    none = Records(r.read_key[:500], np.full(500, 4, dtype=np.uint16), r.ref_id[:500], r.begin_pos[:500])
    s.push_records_marked(none)
    assert s.get_profiles() is None               # "[WARNING] No mapped reads found"
    s.close()
    a = Slimm.for_workload(w, device=0, grouped=False)
    with pytest.raises(capi.SlimmError):
        a.push_records_marked(r)                  # no read identity to sort by
    a.close()


def test_marked_records_resident_on_the_device():
    import torch
    w = make_workload(CONFIGS["config2"], seed=98, n_records=250_000)
    o = run_workload(w, use_qnames=False)
    r = w.records
    dev = torch.device("cuda:0")
    t = [torch.from_numpy(a).to(dev) for a in (Slimm.mark_words(r.read_key, r.flag, r.ref_id).view(np.int32), r.begin_pos)]
    torch.cuda.synchronize()
    s = Slimm.for_workload(w, device=0)
    for _ in range(2):
        s.reset()
        s.reset_cutoffs()
        s.set_records_device_marked(*t)
        assert s.get_profiles() is not None
        assert_matches_oracle(s, o)


# ---------------------------------------------------------------- a stream declared grouped that is not (slimm_check_grouping)
def test_check_grouping_counts_names_that_come_back():
    """GROUPED compares adjacent records only (include/slimm_hip.h): a name that re-appears after other names is two
    reads to the single-pass front end, one to the reference (src/slimm.hpp:204-211).  The diagnostic counts such runs;
    declared SLIMM_ORDER_ANY the same stream gives the reference's results."""
    w = make_workload(CONFIGS["config1"], seed=81)
    s = Slimm.for_workload(w, device=0)
    s.push_records(w.records)
    assert s.check_grouping() == 0
    assert s.get_profiles() is not None
    # the last record of three multi-record runs moves 50 runs down the file, to a place between two runs
    r = w.records
    starts = np.nonzero(np.concatenate([[True], r.read_key[1:] != r.read_key[:-1]]))[0]
    ends = np.concatenate([starts[1:], [len(r)]])
    runs = [list(range(a, b)) for a, b in zip(starts, ends)]
    moved = 0
    j = 10
    while moved < 3:
        if len(runs[j]) >= 2:
            runs.insert(j + 50, [runs[j].pop()])
            moved += 1
            j += 100
        j += 1
    order = [i for run in runs for i in run]
    broken = Workload(w.ref_names, w.ref_len, w.taxonomy, r.take(np.array(order)), w.avg_read_len, w.options, "broken",
                      grouped=False)
    o = run_workload(broken, use_qnames=False)
    s.reset()
    s.push_records(broken.records)
    assert s.check_grouping() == 3
    # (the packed form is checked on its 61 identity bits)
    s.reset()
    s.push_records_packed(_mask61(broken).records)
    assert s.check_grouping() == 3
    assert s.get_profiles() is not None
    assert s.stats()["matches_count"] == o.scalars["matches"] + 3          # what the false promise costs: 3 reads too many
    check(broken, grouped=False)                                            # declared honestly: the reference's numbers


# ---------------------------------------------------------------- one context, file after file in different record forms
def test_record_forms_follow_each_other_on_one_context():
    """A context takes one form per file and any form for the next (include/slimm_hip.h: slimm_reserve).  A packed file
    leaves no flag array behind; a LARGER run-marked file after it must grow only the arrays its form has (it once
    copied the flag array that was never there), and so on through the forms, each result equal to the oracle's."""
    big = make_workload(CONFIGS["config2"], seed=102, n_records=60_000)
    small = Workload(big.ref_names, big.ref_len, big.taxonomy, big.records.take(np.arange(9_000)), big.avg_read_len,
                     big.options, "first9000")
    o_small, o_big = run_workload(small, use_qnames=False), run_workload(big, use_qnames=False)
    s = Slimm.for_workload(big, device=0)
    s.push_records_packed(_mask61(small).records)
    assert s.get_profiles() is not None
    assert_matches_oracle(s, run_workload(_mask61(small), use_qnames=False))
    s.reset(); s.reset_cutoffs()
    s.push_records_marked(big.records, batch=7_000)      # grows past the packed file's capacity in several steps
    assert s.get_profiles() is not None
    assert_matches_oracle(s, o_big)
    s.reset(); s.reset_cutoffs()
    s.push_records(small.records, batch=4_000)             # four arrays again: the flag array appears at the first push
    assert s.get_profiles() is not None
    assert_matches_oracle(s, o_small)
    s.reset(); s.reset_cutoffs()
    chk = np.full(len(big.records), 7, dtype=np.uint32)
    s.push_records_checked(big.records, chk, batch=11_000)  # check words join in; every array grows with its contents
    assert s.get_profiles() is not None
    assert_matches_oracle(s, o_big)
    s.reset(); s.reset_cutoffs()
    s.push_records_packed(_mask61(big).records, batch=13_000)
    assert s.get_profiles() is not None
    assert_matches_oracle(s, run_workload(_mask61(big), use_qnames=False))


def test_set_records_device_replaces_an_earlier_form():
    """slimm_set_records_device (four arrays) after packed or run-marked records were set on the same context WITHOUT a
    reset in between: the call replaces what was there, form included (it once kept `packed`, and the front end read
    flag bits out of the key)."""
    import torch
    w = make_workload(CONFIGS["config2"], seed=103, n_records=50_000)
    o = run_workload(w, use_qnames=False)
    r = w.records
    dev = torch.device("cuda:0")
    four = [torch.from_numpy(a).to(dev) for a in (r.read_key.view(np.int64), r.ref_id, r.begin_pos, r.flag.view(np.int16))]
    pk = [torch.from_numpy(a).to(dev) for a in (Slimm.pack_keys(r.read_key, r.flag).view(np.int64), r.ref_id, r.begin_pos)]
    mk = [torch.from_numpy(a).to(dev) for a in (Slimm.mark_words(r.read_key, r.flag, r.ref_id).view(np.int32), r.begin_pos)]
    torch.cuda.synchronize()
    s = Slimm.for_workload(w, device=0)
    s.set_records_device_packed(*pk)
    s.set_records_device(*four)
    assert s.get_profiles() is not None
    assert_matches_oracle(s, o)
    s.reset(); s.reset_cutoffs()
    s.set_records_device_marked(*mk)
    s.set_records_device(*four)
    assert s.get_profiles() is not None
    assert_matches_oracle(s, o)
    s.reset(); s.reset_cutoffs()
    s.set_records_device(*four)
    s.set_records_device_packed(*pk)
    assert s.get_profiles() is not None
    assert_matches_oracle(s, run_workload(_mask61(w), use_qnames=False))


def test_files_back_to_back_through_two_contexts(tmp_path):
    """INTEGRATION.md "files one after the other" / slimm_amd.distributed.FilesBackToBack (bench.py's timed loop): phase A
    of file k + 1 is launched on one context BEFORE the host finishes file k on the other (get_reads_lca_count,
    write_abundance).  Five files of one database -- different lengths, one of them without a mapped record, one in another
    record form -- each equal to the oracle's result for THAT file, whichever context it went through (the reference's unit
    of work: one file through one freshly reset object, src/slimm.hpp:950-956)."""
    from slimm_amd.distributed import FilesBackToBack
    files = [make_workload(CONFIGS["config1"], seed=61, n_records=n) for n in (9_000, 30_000, 4_000, 30_000, 17_000)]
    r = files[2].records
    r.flag[:] |= 0x4                     # file 2: every record unmapped -> "[WARNING] No mapped reads found", no profile
    want = [run_workload(w, use_qnames=False) for w in files]
    assert want[2].no_hits and not want[1].no_hits
    at = [0]

    def give(e):
        w = files[at[0]]
        if at[0] == 3:
            e.push_records_packed(_mask61(w).records)
        else:
            e.push_records(w.records, batch=7_000)
        at[0] += 1

    path = str(tmp_path / "profile.tsv")
    engines = [Slimm.for_workload(files[0], device=0), Slimm.for_workload(files[0], device=0)]
    fb = FilesBackToBack(engines, give, None, path)
    got = []
    for k in range(len(files)):
        before = fb.step()               # launches file k, finishes file k - 1 beside it
        if k:
            got.append(before)
    got.append(fb.flush())
    assert [g is None for g in got] == [False, False, True, False, False]
    for k, (text, o) in enumerate(zip(got, want)):
        if text is not None:
            assert_profiles_match(text, o.profile_tsv)
    assert open(path).read() == got[-1]
    # the contexts still hold the last two files (3 on engine 1, 4 on engine 0): every integer of each
    assert_matches_oracle(engines[0], want[4])
    o3 = run_workload(_mask61(files[3]), use_qnames=False)
    assert_matches_oracle(engines[1], o3)
    for e in engines:
        e.close()


def test_files_back_to_back_in_directory_mode_keep_the_first_files_cutoffs(tmp_path):
    """FilesBackToBack(directory_mode=True) = the reference's `-d` loop: ONE `slimm` object serves every file, so the
    cut-offs cached by file 1 are reused by the files behind it (src/slimm.hpp:155-156, 330, 674; Q8), although the files
    alternate between two contexts here.  The oracle object is kept across the files the same way."""
    from oracle.binding import Oracle
    from slimm_amd.distributed import FilesBackToBack
    base = make_workload(CONFIGS["config1"], seed=71, n_records=50_000)
    files = [Workload(base.ref_names, base.ref_len, base.taxonomy, base.records.take(np.arange(lo, hi)), base.avg_read_len,
                      base.options, base.name) for lo, hi in ((0, 30_000), (30_000, 36_000), (36_000, 50_000))]
    orc = Oracle(files[0].taxonomy, files[0].options)
    want = [orc.run(w.ref_names, w.ref_len, w.records, w.avg_read_len, use_qnames=False) for w in files]
    fresh = [run_workload(w, use_qnames=False) for w in files]
    assert any(a.cutoffs[:2] != b.cutoffs[:2] for a, b in zip(want[1:], fresh[1:]))     # the leak is visible in this input
    at = [0]

    def give(e):
        e.push_records(files[at[0]].records)
        at[0] += 1

    engines = [Slimm.for_workload(files[0], device=0), Slimm.for_workload(files[0], device=0)]
    fb = FilesBackToBack(engines, give, None, str(tmp_path / "p.tsv"), directory_mode=True)
    got = []
    for k in range(len(files)):
        before = fb.step()
        if k:
            got.append(before)
    got.append(fb.flush())
    for text, o in zip(got, want):
        assert_profiles_match(text, o.profile_tsv)
    assert_matches_oracle(engines[0], want[2])
    assert_matches_oracle(engines[1], want[1])
    for e in engines:
        e.close()
