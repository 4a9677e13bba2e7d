"""CPU tests of the host-side glue inside libslimm_hip.so (cut-offs, valid set, LCA propagation, profile writer).

A host-only context (device = -1) is fed the per-reference / per-taxon integers the kernels would produce -- here taken
from the oracle -- and everything downstream must equal the oracle.  No GPU call is made.
"""
import os

import numpy as np
import pytest

from oracle.binding import run_workload
from slimm_amd.profiler import Slimm, host_avg_read_length, host_bin_of, host_quantile_cut_off
from slimm_amd.synth import CONFIGS, make_workload
from tests.cases import holes_case, tiny_case
from tests.helpers import assert_matches_oracle, partials_from_oracle
import oracle.binding as ob


def _host_only_run(w):
    o = run_workload(w, use_qnames=w.records.qname is not None, collect_bins=False)
    s = Slimm.for_workload(w, device=-1)
    assert s.set_coverage_columns(o.reads_count, o.uniq_reads_count, o.nz_cov, o.nz_uniq_cov, o.scalars["hits"],
                                  o.scalars["matches"])
    s.filter_alignments()
    s.set_partials(*partials_from_oracle(o, w.lineage(), s.dense_taxid))
    s.get_reads_lca_count()
    return s, o


@pytest.mark.parametrize("mk", [tiny_case, holes_case])
def test_micro_cases(mk):
    s, o = _host_only_run(mk())
    assert_matches_oracle(s, o, bins=False)


@pytest.mark.parametrize("cfg,n,seed,kw", [
    ("config1", None, 3, {}),
    ("config1", None, 4, {"hole_every": 3}),
    ("config2", 200_000, 1, {}),
    ("config5", 100_000, 2, {}),
])
def test_synthetic(cfg, n, seed, kw):
    w = make_workload(CONFIGS[cfg], seed=seed, n_records=n, **kw)
    s, o = _host_only_run(w)
    # with holes the reference's result depends on unordered_map iteration order: the final counts
    # of taxid 0 are then not pinned, everything else is
    if kw.get("hole_every"):
        st = s.stats()
        assert st["n_valid"] == o.scalars["n_valid"]
        assert s.taxon_counts(0) == o.lca_direct and s.children_pairs(0) == o.lca_direct_children
    else:
        assert_matches_oracle(s, o, bins=False)


@pytest.mark.parametrize("rank", ["genus", "family", "phylum"])
def test_other_ranks(rank):
    w = make_workload(CONFIGS["config1"], seed=5)
    w.options.rank = rank
    s, o = _host_only_run(w)
    assert_matches_oracle(s, o, bins=False)


def test_quantile_cut_off_matches_oracle_order():
    rng = np.random.default_rng(0)
    for n in (1, 2, 5, 100, 5000):
        v = rng.random(n).astype(np.float32)
        for q in (0.0, 0.5, 0.95, 0.999):
            # restated in numpy float32 with the same operation order (misc.hpp:197-216)
            total = np.float32(0)
            for x in v:
                total = np.float32(total + x)
            sv = np.sort(v)
            i = n - 1
            sub = np.float32(0)
            while np.float32(sub / total) < np.float32(q) and i > 0:
                sub = np.float32(sub + sv[i])
                i -= 1
            assert host_quantile_cut_off(v, q) == float(sv[i])
    assert host_quantile_cut_off(np.zeros(0, dtype=np.float32), 0.95) == 0.0
    # from 256 values on the library sorts non-negative floats by their bit patterns (counting passes), anything else with
    # std::sort: many equal values, zeros, denormals, huge values, infinity -- and a negative value among them
    for n in (256, 1024, 20_000):
        v = (rng.integers(0, 300, n) / np.float32(301)).astype(np.float32)
        v[::7] = 0.0
        v[5] = np.float32(1e-42)
        v[6] = np.float32(3e38)
        for neg in (False, True):
            if neg:
                v[9] = np.float32(-0.25)
            sv = np.sort(v)
            total = np.float32(0)
            for x in v:
                total = np.float32(total + x)
            for q in (0.0, 0.3, 0.95):
                i = n - 1
                sub = np.float32(0)
                while np.float32(sub / total) < np.float32(q) and i > 0:
                    sub = np.float32(sub + sv[i])
                    i -= 1
                assert host_quantile_cut_off(v, q) == float(sv[i])
    v = np.full(2000, 0.5, dtype=np.float32)
    v[3] = np.inf
    # (0 / inf = 0 < q: the infinity is taken, inf / inf is no number, the loop ends one value further down)
    assert host_quantile_cut_off(v, 0.5) == 0.5 and host_quantile_cut_off(v, 0.0) == float("inf")


def test_bin_of_wraps_like_uint32():
    # src/slimm.hpp:200: int32 + uint32 -> uint32 (wraps), then min with the contig length
    assert host_bin_of(0, 100, 5000, 1000) == 0
    assert host_bin_of(4990, 100, 5000, 1000) == 5          # centre 5040 clamps to len 5000 -> bin len/W
    assert host_bin_of(-1, 100, 5000, 1000) == 0            # 0xFFFFFFFF + 50 wraps to 49
    assert host_bin_of(-1, 1, 5000, 1000) == 5              # A/2 == 0: 0xFFFFFFFF clamps to len
    assert host_bin_of(2**31 - 1, 100, 2**32 - 1, 1000) == (2**31 - 1 + 50) // 1000


def test_canonical_read_name_is_a_bijection_with_the_reference_key_string():
    """src/slimm.hpp:204-208 keys a read by qName + ".1" / ".2" / "" (Q18).  (base, mate) from slimm_host_canonical_read_name
    is equal for two records iff that string is; the Python mirror (workload.canonical_identity) agrees."""
    from slimm_amd.profiler import host_canonical_read_name
    from slimm_amd.workload import canonical_identity
    assert host_canonical_read_name("N", 0x41) == ("N", 0x41)
    assert host_canonical_read_name("N.1", 0) == ("N", 0x40)
    assert host_canonical_read_name("N.2", 0x100) == ("N", 0x180)
    assert host_canonical_read_name("N.1", 0x41) == ("N.1", 0x41)        # "N.1.1"
    assert host_canonical_read_name("N.1", 0x80) == ("N.1", 0x80)        # "N.1.2"
    assert host_canonical_read_name("N.3", 0) == ("N.3", 0)
    assert host_canonical_read_name("N.12", 0) == ("N.12", 0)
    assert host_canonical_read_name(".1", 0) == ("", 0x40)
    assert host_canonical_read_name("1", 0) == ("1", 0) and host_canonical_read_name("", 0) == ("", 0)
    assert host_canonical_read_name("N.1", 0x4) == ("N", 0x44)
    names, flags = [], []
    for base in ("a", "a.1", "a.2", "a.1.1", "a.1.2", "a.", "a..1", ".1", ".2", "1", "a.3"):
        for f in (0, 0x40, 0x80, 0xc0, 0x100):
            names.append(base)
            flags.append(f)
    seen = {}
    pb, pf = canonical_identity(names, flags)
    for i, (n, f) in enumerate(zip(names, flags)):
        key_string = n + (".1" if f & 0x40 else ".2" if f & 0x80 else "")
        b, cf = host_canonical_read_name(n, f)
        assert (b, cf) == (pb[i], int(pf[i]))
        ident = (b, 1 if cf & 0x40 else 2 if cf & 0x80 else 0)
        assert seen.setdefault(key_string, ident) == ident
    assert len(set(seen.values())) == len(seen)


def test_mark_word_layout():
    """word = reference + 1 (0: not mapped) | mate << 29 | starts a qName run << 31 (include/slimm_hip.h)."""
    from slimm_amd import capi
    L = capi.lib()
    assert L.slimm_mark_word(0, 0, 0) == 1
    assert L.slimm_mark_word(41, 0, 1) == (42 | 1 << 31)
    assert L.slimm_mark_word(7, 0x40, 0) == (8 | 1 << 29)
    assert L.slimm_mark_word(7, 0x80, 1) == (8 | 2 << 29 | 1 << 31)
    assert L.slimm_mark_word(7, 0xc0, 0) == (8 | 1 << 29)              # first-in-pair wins (src/slimm.hpp:205-208)
    assert L.slimm_mark_word(7, 0x4, 1) == 1 << 31                     # the unmapped flag (src/slimm.hpp:197)
    assert L.slimm_mark_word(-1, 0x80, 0) == 2 << 29                   # no reference
    key = np.array([5, 5, 9, 9, 9, 5], dtype=np.uint64)
    flag = np.array([0, 0x40, 0, 4, 0x80, 0], dtype=np.uint16)
    ref = np.array([1, 2, 3, 4, -1, 6], dtype=np.int32)
    w = Slimm.mark_words(key, flag, ref)
    assert list(w >> 31) == [1, 0, 1, 0, 0, 1]
    assert list(w & 0x1fffffff) == [2, 3, 4, 0, 0, 7]
    assert list(Slimm.mark_words(key, flag, ref, prev_key=5) >> 31) == [0, 0, 1, 0, 0, 1]   # the batch continues a run
    assert list(Slimm.mark_words(key, flag, ref, prev_key=6) >> 31) == [1, 0, 1, 0, 0, 1]


def test_pack_key_layout_on_the_host():
    """slimm_pack_key / slimm_pack_keys are host functions: (key & (2^61 - 1)) | mate << 61 | unmapped << 63."""
    k = np.array([0, 1, (1 << 61) - 1, (1 << 62) - 1, 0x123456789abcdef0], dtype=np.uint64)
    for flag, top in ((0, 0), (0x4, 4), (0x40, 1), (0x80, 2), (0xc0, 1), (0x44, 5), (0x900, 0), (0x84 | 0x100, 6)):
        got = Slimm.pack_keys(k, np.full(k.shape, flag, dtype=np.uint16))
        assert np.array_equal(got & np.uint64((1 << 61) - 1), k & np.uint64((1 << 61) - 1))
        assert np.all((got >> np.uint64(61)) == np.uint64(top)), flag


def test_avg_read_length():
    l = np.array([0, 100, 0, 101, 99, 150], dtype=np.uint32)
    assert host_avg_read_length(l, 100000) == ob.avg_read_length(l, 100000) == (100 + 101 + 99 + 150) // 4
    assert host_avg_read_length(l, 2) == ob.avg_read_length(l, 2) == 100
    assert host_avg_read_length(np.zeros(3, dtype=np.uint32)) == 0


def test_cutoff_cache_survives_reset_like_the_reference():
    """Q8: the cached cut-offs are not cleared by slimm::reset() (src/slimm.hpp:155-156, 167-188)."""
    w1 = make_workload(CONFIGS["config1"], seed=3)
    w2 = make_workload(CONFIGS["config1"], seed=3)  # same header and database, a different record stream
    w2.records = w2.records.take(np.arange(0, len(w2.records) // 3))
    o1, o2 = run_workload(w1), run_workload(w2)
    s = Slimm.for_workload(w1, device=-1)
    s.set_coverage_columns(o1.reads_count, o1.uniq_reads_count, o1.nz_cov, o1.nz_uniq_cov, o1.scalars["hits"],
                           o1.scalars["matches"])
    s.filter_alignments()
    c1 = s.stats()["coverage_cut_off"]
    s.reset()
    s.set_coverage_columns(o2.reads_count, o2.uniq_reads_count, o2.nz_cov, o2.nz_uniq_cov, o2.scalars["hits"],
                           o2.scalars["matches"])
    s.filter_alignments()
    assert s.stats()["coverage_cut_off"] == c1 != pytest.approx(o2.cutoffs[0])
    s.reset()
    s.reset_cutoffs()
    s.set_coverage_columns(o2.reads_count, o2.uniq_reads_count, o2.nz_cov, o2.nz_uniq_cov, o2.scalars["hits"],
                           o2.scalars["matches"])
    s.filter_alignments()
    assert s.stats()["coverage_cut_off"] == pytest.approx(o2.cutoffs[0], rel=1e-6)


def test_no_hits_and_bad_config():
    from slimm_amd import capi
    w = tiny_case()
    s = Slimm.for_workload(w, device=-1)
    z = np.zeros(w.n_refs, dtype=np.uint32)
    assert s.set_coverage_columns(z, z, z, z, 0, 0) is False
    w.options.rank = "superkingdom"  # broken in the reference (Q14): rejected
    with pytest.raises(capi.SlimmError):
        Slimm.for_workload(w, device=-1)


def test_collect_profiles_merges_samples(tmp_path):
    """scripts/collect_profiles.py: one row per taxon, one column pair per sample, zeros for
    taxa a sample does not have."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "collect_profiles", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts",
                                         "collect_profiles.py"))
    cp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cp)
    head = "taxa_level\ttaxa_id\tlinage\tabundance\tread_count\n"
    a = tmp_path / "a_profile.tsv"
    b = tmp_path / "b_profile.tsv"
    a.write_text(head + "species\t10\tk__K|s__x\t60.5\t605\nspecies\t0*\tk__|s__\t39.5\t395\n")
    b.write_text(head + "species\t10\tk__K|s__x\t10\t1\nspecies\t11\tk__K|s__y\t80\t8\nspecies\t0*\tk__|s__\t10\t1\n")
    out = tmp_path / "merged.tsv"
    assert cp.main(["-o", str(out), str(a), str(b)]) == 0
    lines = out.read_text().rstrip("\n").split("\n")
    assert lines[0].split("\t") == ["taxa_level", "taxa_id", "linage", "a_abundance", "b_abundance", "a_read_count",
                                     "b_read_count"]
    rows = {tuple(l.split("\t")[:2]): l.split("\t")[3:] for l in lines[1:]}
    assert rows[("species", "10")] == ["60.5", "10", "605", "1"]
    assert rows[("species", "11")] == ["0", "80", "0", "8"]
    assert lines[1].split("\t")[1] == "10"          # sorted by the first sample's abundance, descending


def test_q18_run_counts_for_producers_of_grouped_records():
    """slimm_host_q18_note / _regroup_needed (include/slimm_hip.h, "Q18 ON A GROUPED STREAM"): fed every record's (starts a
    run of one canonical base, name was shortened), they tell whether some run holds shortened names only -- the rule the
    device decoders and the command's reader apply.  Driven here the way a C-ABI producer would, from names and flags."""
    import ctypes as C

    from slimm_amd import capi
    from tests.cases import q18_apart_case, q18_case, tiny_case

    class Runs(C.Structure):
        _fields_ = [("short_starts", C.c_uint64), ("short_to_plain", C.c_uint64), ("last_short", C.c_int)]

    L = capi.lib()

    def needed(w):
        q = Runs()
        prev = None
        for name, flag in zip(w.records.qname, w.records.flags_in_file().tolist()):
            b = name.encode()
            out = C.c_uint16(0)
            n = L.slimm_host_canonical_read_name(b, len(b), int(flag), C.byref(out))
            base = b[:n]
            L.slimm_host_q18_note(C.byref(q), int(base != prev), int(n != len(b)))
            prev = base
        return bool(L.slimm_host_q18_regroup_needed(C.byref(q))), (q.short_starts, q.short_to_plain)

    assert needed(q18_apart_case()) == (True, (1, 0))
    assert needed(q18_apart_case(tail=("r.1",))) == (True, (1, 0))
    assert needed(q18_case()) == (False, (3, 3))
    assert needed(tiny_case()) == (False, (0, 0))
