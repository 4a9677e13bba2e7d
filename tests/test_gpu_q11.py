"""Quirk Q11 with numbers that really wrap: the reference does these in uint32 (src/slimm.hpp:265-266, 287, 755, 785, 792)
and the path reproduces the wrap-around -- genome-length sums past 2^32 (three 2 Gbp contigs under one species; a
phylum with more than 4.29 Gbp of children), `matched_ref_length`, `reads_count * 100` and `count * avg_read_length`
past 2^32 (45 M reads on one reference).  Everything is compared with the oracle, which does the same arithmetic in the
reference's types; the tests also assert that the wrap happened, so that they cannot pass vacuously."""
import numpy as np
import pytest

from slimm_amd.synth import SynthConfig, make_workload
from slimm_amd.workload import Options, Records, Workload
from tests.cases import records_from_sam, taxonomy_from_lineages
from tests.test_gpu_parity import check

pytestmark = pytest.mark.gpu


def test_three_2gbp_contigs_under_one_species():
    """genome_Length = (sum of the children's lengths) / n in uint32 (src/slimm.hpp:755, 785): 6e9 wraps to 1 705 032 704;
    matched_ref_length (src/slimm.hpp:265) wraps the same way."""
    lin = {"X1": [101, 11, 21, 31, 41, 51, 61, 2], "X2": [102, 11, 21, 31, 41, 51, 61, 2],
           "X3": [103, 11, 21, 31, 41, 51, 61, 2], "Y": [104, 12, 21, 31, 41, 51, 61, 2]}
    names = [a + ".1" for a in lin]
    lens = np.array([2_000_000_000, 2_000_000_000, 2_000_000_000, 3_000_000], dtype=np.uint32)
    rng = np.random.default_rng(5)
    rows = []
    for i in range(900):
        r = int(rng.integers(0, 4))
        rows.append((f"u{i}", 0, names[r], int(rng.integers(1, int(lens[r]) - 200))))
    for i in range(60):   # reads on two or three of the big contigs: LCA = species 11, children = the contigs
        a, b = rng.choice(3, size=2, replace=False)
        rows += [(f"m{i}", 0, names[a], int(rng.integers(1, 1_900_000_000))),
                 (f"m{i}", 256, names[b], int(rng.integers(1, 1_900_000_000)))]
    w = Workload(names, lens, taxonomy_from_lineages(lin), records_from_sam(rows, names), avg_read_len=100,
                 options=Options(bin_width=1_000_000, cov_cut_off=0.999, abundance_cut_off=0.0), name="2gbp")
    s, o = check(w)
    assert int(lens[:3].astype(np.uint64).sum()) > 2**32
    assert s.stats()["matched_ref_length"] == int(lens.astype(np.uint64).sum()) % 2**32   # all four have reads
    kids = {r for t, r in s.children_pairs(1) if t == 11}
    assert kids == {0, 1, 2}                                      # the species row divides a wrapped sum by 3


def test_phylum_rank_with_more_than_4gbp_of_children():
    """-r phylum on a database of 2 500 references: the lengths of a phylum's children add up past 2^32 (src/slimm.hpp:755)."""
    cfg = SynthConfig("phy", 400_000, 2_500, 2.0, bin_width=20_000, len_lo=4_000_000, len_hi=8_000_000, present_frac=0.95,
                      cov_cut_off=0.9999)
    w = make_workload(cfg, seed=91)
    w.options.rank = "phylum"
    w.options.abundance_cut_off = 0.0
    s, o = check(w)
    lin = w.lineage()
    by_phylum = {}
    for t, r in s.children_pairs(1):
        if t in set(lin[:, 6].tolist()):
            by_phylum[t] = by_phylum.get(t, 0) + int(w.ref_len[r])
    assert max(by_phylum.values()) > 2**32, by_phylum              # the sum the profile divides has wrapped


def test_45_million_reads_on_one_reference():
    """reads_count * 100 (src/slimm.hpp:266, 287) and count * avg_read_length (src/slimm.hpp:792) past 2^32."""
    n = 45_000_000
    lin = {"A": [101, 11, 21, 31, 41, 51, 61, 2], "B": [102, 12, 22, 31, 41, 51, 61, 2]}
    names = [a + ".1" for a in lin]
    lens = np.array([5_000_000, 4_000_000], dtype=np.uint32)
    i = np.arange(n, dtype=np.int64)
    ref = np.where(i % 1000 == 999, 1, 0).astype(np.int32)
    pos = ((i * 7919) % (int(lens.min()) - 200)).astype(np.int32)
    rec = Records(i.astype(np.uint64), np.zeros(n, dtype=np.uint16), ref, pos)
    w = Workload(names, lens, taxonomy_from_lineages(lin), rec, avg_read_len=100,
                 options=Options(bin_width=1000, cov_cut_off=0.99, abundance_cut_off=0.0), name="45M")
    s, o = check(w)
    rc = s.ref_columns()
    assert int(rc["reads_count"][0]) * 100 > 2**32 and int(rc["uniq_reads_count"][0]) * 100 > 2**32
    assert s.taxon_counts(1)[11] * 100 > 2**32
