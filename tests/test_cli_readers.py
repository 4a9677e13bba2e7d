"""CPU tests of the host-side readers of the `slimm` command line (SAM / BAM / header parsing), through
`slimm --dump-records`: files written by the independent Python writers in tests/bam_io.py must decode to the
records they were written from.  No GPU is touched."""
import os
import subprocess

import numpy as np
import pytest

from slimm_amd.synth import CONFIGS, make_workload
from tests.bam_io import qnames_of, write_bam, write_sam
from tests.cases import tiny_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "slimm_amd", "slimm")


def dump(path):
    out = subprocess.run([CLI, "--dump-records", path], capture_output=True, text=True, check=True).stdout.split("\n")
    head = out[0].split("\t")
    refs = [ln.split("\t")[1:] for ln in out if ln.startswith("@\t")]
    recs = [ln.split("\t") for ln in out[1:] if ln and not ln.startswith("@\t")]
    return head, refs, recs


@pytest.mark.parametrize("writer,fmt", [(write_sam, "SAM"), (write_bam, "BAM")])
@pytest.mark.parametrize("mk", [tiny_case, lambda: make_workload(CONFIGS["config1"], seed=31)])
def test_reader_round_trip(tmp_path, writer, fmt, mk):
    w = mk()
    p = str(tmp_path / ("x." + fmt.lower()))
    writer(p, w.ref_names, w.ref_len, w.records, read_len=w.avg_read_len)
    head, refs, recs = dump(p)
    assert head[1] == fmt
    assert [r[0] for r in refs] == w.ref_names and [int(r[1]) for r in refs] == w.ref_len.tolist()
    q = qnames_of(w.records)
    assert len(recs) == len(w.records)
    assert [r[0] for r in recs] == q
    assert [int(r[1]) for r in recs] == w.records.flag.tolist()
    assert [int(r[2]) for r in recs] == w.records.ref_id.tolist()
    assert [int(r[3]) for r in recs] == w.records.begin_pos.tolist()
    assert all(int(r[4]) == w.avg_read_len for r in recs)
    # equal names <=> equal keys
    by_name = {}
    for r in recs:
        assert by_name.setdefault(r[0], r[5]) == r[5]
    assert len(set(by_name.values())) == len(by_name)


def test_bam_spanning_many_bgzf_blocks(tmp_path):
    w = make_workload(CONFIGS["config2"], seed=32, n_records=40_000)   # ~6 MB of BAM records -> ~100 BGZF blocks
    p = str(tmp_path / "big.bam")
    write_bam(p, w.ref_names, w.ref_len, w.records)
    _, refs, recs = dump(p)
    assert len(refs) == 5000 and len(recs) == 40_000
    assert [int(r[3]) for r in recs] == w.records.begin_pos.tolist()


def test_sort_order_tag_and_errors(tmp_path):
    w = tiny_case()
    p = str(tmp_path / "q.sam")
    write_sam(p, w.ref_names, w.ref_len, w.records, hd="@HD\tVN:1.6\tSO:coordinate")
    assert dump(p)[0][3] == "3"       # SortOrder::Coordinate
    write_sam(p, w.ref_names, w.ref_len, w.records, hd="@HD\tVN:1.6\tSO:queryname")
    assert dump(p)[0][3] == "2"
    write_sam(p, w.ref_names, w.ref_len, w.records, hd="")
    assert dump(p)[0][3] == "0"
    bad = str(tmp_path / "trunc.bam")
    write_bam(bad, w.ref_names, w.ref_len, w.records)
    data = open(bad, "rb").read()
    open(bad, "wb").write(data[: len(data) // 2])
    r = subprocess.run([CLI, "--dump-records", bad], capture_output=True, text=True)
    assert r.returncode != 0
    r = subprocess.run([CLI, "--dump-records", str(tmp_path / "missing.bam")], capture_output=True, text=True)
    assert r.returncode != 0 and "Could not open" in r.stderr
