"""SAM text decoded on the device (slimm_push_sam_bytes, slimm_amd/csrc/sam_decode.hip) against the oracle, which takes the
same records as decoded arrays and groups them by their NAMES like the reference does (seqan::readRecord + the string-keyed
map of src/slimm.hpp:194-211).  The text comes from the independent Python writer (tests/bam_io.py); windows cut lines at
every offset."""
import numpy as np
import pytest

from oracle.binding import run_workload
from slimm_amd import capi
from slimm_amd.profiler import Slimm
from slimm_amd.synth import CONFIGS, SynthConfig, make_workload
from slimm_amd.workload import Records, Workload
from tests.bam_io import write_sam
from tests.cases import Q18_APART_EXPECTED, Q18_EXPECTED, q18_apart_case, q18_case
from tests.helpers import assert_matches_oracle
from tests.test_gpu_bam_decode import _named
from tests.test_gpu_parity import _interleave_mates

pytestmark = pytest.mark.gpu


def sam_body(tmp_path, w, read_len=None, tail_newline=True) -> bytes:
    p = str(tmp_path / "x.sam")
    write_sam(p, w.ref_names, w.ref_len, w.records, read_len=read_len or w.avg_read_len)
    lines = open(p, "rb").read().split(b"\n")
    body = b"\n".join(ln for ln in lines if ln and not ln.startswith(b"@"))
    return body + (b"\n" if tail_newline else b"")


def _check(tmp_path, w: Workload, grouped: bool, window: int, tail_newline=True):
    o = run_workload(w, use_qnames=True)
    data = sam_body(tmp_path, w, tail_newline=tail_newline)
    s = Slimm.for_workload(w, device=0, grouped=grouped)
    s.set_reference_names(w.ref_names)
    n = s.push_sam_bytes(data, window=window)
    assert n == len(w.records)
    prof = s.get_profiles()
    if o.no_hits:
        assert prof is None
    else:
        assert_matches_oracle(s, o)
    s.close()


@pytest.mark.parametrize("window", [0, 1 << 20, 100_003, 8_209, 977])
def test_grouped_text_decoded_on_the_device(tmp_path, window):
    """Name-grouped lines -> run-marked records by comparing adjacent QNAMEs on the device; windows of less than a piece, of a
    piece and a bit, of a megabyte, the whole text at once; unmapped flags, `*` references."""
    w = _named(make_workload(CONFIGS["config1"], seed=31))
    w.records.flag[::11] |= 4
    w.records.ref_id[5::19] = -1
    _check(tmp_path, w, True, window)


@pytest.mark.parametrize("window", [0, 250_007])
def test_unordered_text_decoded_on_the_device(tmp_path, window):
    w = _named(make_workload(CONFIGS["config1"], seed=32, shuffled=True))
    _check(tmp_path, w, False, window, tail_newline=False)      # (and a last line without its newline)


def test_long_names_interleaved_mates_and_q18(tmp_path):
    w = _named(_interleave_mates(make_workload(SynthConfig("pairs", 20_000, 300, 6.0), seed=33, paired_frac=0.9)), long_names=True)
    _check(tmp_path, w, True, 300_001)
    wq = q18_case()
    _check(tmp_path, wq, True, 97, tail_newline=False)
    for seed in (1, 2):
        wa = q18_case(list(np.random.default_rng(seed).permutation(18)))
        _check(tmp_path, wa, False, 211)


@pytest.mark.parametrize("window", [0, 61, 1_003])
def test_q18_shortened_names_apart_from_their_namesakes_ask_for_the_any_order_path(tmp_path, window):
    """tests/test_gpu_bam_decode.py, the test of the same name, through SAM text."""
    w = q18_apart_case()
    o = run_workload(w, use_qnames=True)
    assert o.scalars["matches"] == Q18_APART_EXPECTED["matches"] and o.scalars["uniq_matches"] == Q18_APART_EXPECTED["uniq_matches"]
    s = Slimm.for_workload(w, device=0, grouped=True)
    s.set_reference_names(w.ref_names)
    assert s.push_sam_bytes(sam_body(tmp_path, w), window=window) == len(w.records)
    assert s.q18_runs() == (1, 0)      # `r.1` and `r.2` share the base `r`: one run of shortened names only
    with pytest.raises(capi.SlimmError) as e:
        s.get_profiles()
    assert e.value.code == capi.E_REGROUP
    s.close()
    _check(tmp_path, w, False, window)
    wq = q18_case()
    s = Slimm.for_workload(wq, device=0, grouped=True)
    s.set_reference_names(wq.ref_names)
    s.push_sam_bytes(sam_body(tmp_path, wq), window=window)
    assert s.q18_runs() == (3, 3)
    s.close()


def test_unknown_reference_names_and_bad_lines(tmp_path):
    w = _named(make_workload(CONFIGS["config1"], seed=35, n_records=3_000))
    data = sam_body(tmp_path, w)
    # a reference name the header does not have reads as "no reference" (the host reader's map lookup fails the same way)
    lines = data.split(b"\n")
    f = lines[100].split(b"\t")
    f[2] = b"NOT_IN_THE_HEADER"
    lines[100] = b"\t".join(f)
    r = w.records
    ref = r.ref_id.copy()
    ref[100] = -1
    w2 = Workload(w.ref_names, w.ref_len, w.taxonomy, Records(r.read_key, r.flag, ref, r.begin_pos, r.qname), w.avg_read_len, w.options, "x")
    o = run_workload(w2, use_qnames=True)
    s = Slimm.for_workload(w, device=0)
    s.set_reference_names(w.ref_names)
    assert s.push_sam_bytes(b"\n".join(lines), window=50_000) == len(r)
    assert s.get_profiles() is not None
    assert_matches_oracle(s, o)
    # fewer than ten fields: the host reader's error
    s.reset(); s.reset_cutoffs()
    bad = list(data.split(b"\n"))
    bad[2000] = b"\t".join(bad[2000].split(b"\t")[:8])
    with pytest.raises(capi.SlimmError) as e:
        s.push_sam_bytes(b"\n".join(bad), window=64_000)
    assert "fewer than 10 fields" in str(e.value)
    # a header line among the alignments: refused (the host decoder skips it)
    s.reset(); s.reset_cutoffs()
    hdr = list(data.split(b"\n"))
    hdr.insert(1500, b"@CO\tcomment")
    with pytest.raises(capi.SlimmError) as e:
        s.push_sam_bytes(b"\n".join(hdr))
    assert "decode this file on the host" in str(e.value)
    # without the names: an error, not a guess
    s2 = Slimm.for_workload(w, device=0)
    with pytest.raises(capi.SlimmError):
        s2.push_sam_bytes(data)
    s.close(); s2.close()


def test_lines_longer_than_a_piece_and_windows_shorter_than_a_line(tmp_path):
    """Sequences of tens of kilobases: a line runs over several 8 KB pieces (no line starts in them), and windows smaller than
    a line (several pushes without a complete one); CR LF line ends (the carriage return stays in the last field, which nobody
    reads)."""
    w = _named(make_workload(SynthConfig("few", 300, 12, 2.0, bin_width=100, len_lo=5_000, len_hi=50_000, present_frac=0.8), seed=34))
    o = run_workload(w, use_qnames=True)
    p = str(tmp_path / "long.sam")
    write_sam(p, w.ref_names, w.ref_len, w.records, read_len=20_000)
    body = b"\n".join(ln for ln in open(p, "rb").read().split(b"\n") if ln and not ln.startswith(b"@")) + b"\n"
    for data, window in ((body, 50_021), (body, 9_973), (body.replace(b"\n", b"\r\n"), 33_333)):
        s = Slimm.for_workload(w, device=0, grouped=True)
        s.set_reference_names(w.ref_names)
        assert s.push_sam_bytes(data, window=window) == len(w.records)
        assert s.get_profiles() is not None
        assert_matches_oracle(s, o)
        s.close()
