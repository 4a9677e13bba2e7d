"""BAM alignment records decoded on the device (slimm_push_bam_bytes, slimm_amd/csrc/bam_decode.hip) against the oracle,
which takes the same records as decoded arrays and groups them by their NAMES like the reference does (seqan::readRecord
+ the string-keyed map of src/slimm.hpp:194-211).  The record bytes come from an independent Python writer
(tests/bam_io.py), regular and with record sizes that put boundaries anywhere; windows cut records at every offset."""
import numpy as np
import pytest

from oracle.binding import run_workload
from slimm_amd import capi
from slimm_amd.profiler import Slimm
from slimm_amd.synth import CONFIGS, SynthConfig, make_workload
from slimm_amd.workload import Records, Workload
from tests.bam_io import bam_record_bytes
from tests.cases import Q18_APART_EXPECTED, Q18_EXPECTED, q18_apart_case, q18_case
from tests.helpers import assert_matches_oracle
from tests.test_gpu_parity import _interleave_mates, one_long_read_workload

pytestmark = pytest.mark.gpu


def _named(w: Workload, long_names: bool = False) -> Workload:
    """The workload with read names (the oracle then groups by the names themselves)."""
    r = w.records
    ids = np.unique(r.read_key, return_inverse=True)[1]
    names = [("read/%d/" % i) + ("x" * (i % 180) if long_names else "") for i in ids.tolist()]
    return Workload(w.ref_names, w.ref_len, w.taxonomy, Records(r.read_key, r.flag, r.ref_id, r.begin_pos, names),
                    w.avg_read_len, w.options, w.name + "-named", grouped=w.grouped)


def _check(w: Workload, grouped: bool, window: int, irregular=None, read_len=100):
    o = run_workload(w, use_qnames=True)
    data = bam_record_bytes(w.records, read_len=read_len, irregular_seed=irregular)
    s = Slimm.for_workload(w, device=0, grouped=grouped)
    n = s.push_bam_bytes(data, window=window)
    assert n == len(w.records)
    prof = s.get_profiles()
    if o.no_hits:
        assert prof is None
    else:
        assert_matches_oracle(s, o)
    s.close()


@pytest.mark.parametrize("window", [0, 1 << 20, 100_003, 16_411])
@pytest.mark.parametrize("irregular", [None, 7])
def test_grouped_file_decoded_on_the_device(window, irregular):
    """Name-grouped records -> run-marked records by comparing adjacent NAMES on the device; windows of one piece and a
    bit, of a few pieces, of a megabyte, and the whole file at once."""
    w = _named(make_workload(CONFIGS["config1"], seed=31))
    w.records.flag[::11] |= 4
    w.records.ref_id[5::19] = -1
    _check(w, True, window, irregular)


@pytest.mark.parametrize("window", [0, 97, 211])
def test_q18_name_suffix_and_mate_flag_make_one_key(window):
    """Q18 (src/slimm.hpp:204-208: the key is the string qName + ".1" / ".2"): `N`/0x40 and an unflagged `N.1` are one read.
    Grouped: the adjacent-name compare works on the canonical base (also across windows: the carried name); any order: the
    hash and the check word do."""
    w = q18_case()
    _check(w, True, window, read_len=50)
    for seed in (1, 2, 3):
        wa = q18_case(list(np.random.default_rng(seed).permutation(18)))
        o = run_workload(wa, use_qnames=True)
        assert o.scalars["matches"] == Q18_EXPECTED["matches"] and o.scalars["uniq_matches"] == Q18_EXPECTED["uniq_matches"]
        _check(wa, False, window, read_len=50)


@pytest.mark.parametrize("window", [0, 97, 4_001])
def test_q18_shortened_names_apart_from_their_namesakes_ask_for_the_any_order_path(window):
    """A file grouped by QNAME whose key strings are NOT adjacent (`r`/0x40 ... fifty reads ... the unflagged `r.1`): the
    reference joins them through its hash map (src/slimm.hpp:204-211).  The grouped context counts the runs of shortened names
    that stand apart from an un-shortened namesake and refuses to analyse (SLIMM_E_REGROUP) instead of profiling 54 reads; the
    same bytes through an any-order context equal the oracle (52 reads).  q18_case (every shortened name NEXT to its namesake)
    stays on the grouped path: test_q18_name_suffix_and_mate_flag_make_one_key above."""
    w = q18_apart_case()
    o = run_workload(w, use_qnames=True)
    assert (o.scalars["hits"], o.scalars["matches"], o.scalars["uniq_matches"]) == (
        Q18_APART_EXPECTED["hits"], Q18_APART_EXPECTED["matches"], Q18_APART_EXPECTED["uniq_matches"])
    data = bam_record_bytes(w.records, read_len=50)
    s = Slimm.for_workload(w, device=0, grouped=True)
    assert s.push_bam_bytes(data, window=window) == len(w.records)
    assert s.q18_runs() == (1, 0)      # `r.1` and `r.2` share the base `r`: one run of shortened names only
    with pytest.raises(capi.SlimmError) as e:
        s.get_profiles()
    assert e.value.code == capi.E_REGROUP and "SLIMM_ORDER_ANY" in str(e.value)
    s.close()
    _check(w, False, window, read_len=50)
    # only `r.2` apart, `r.1` next to `r`: still one run of shortened names only
    w1 = q18_apart_case(tail=("r.2",))
    s = Slimm.for_workload(w1, device=0, grouped=True)
    s.push_bam_bytes(bam_record_bytes(w1.records, read_len=50), window=window)
    with pytest.raises(capi.SlimmError) as e:
        s.get_profiles()
    assert e.value.code == capi.E_REGROUP
    # ... and the next file of the same context starts from zero
    s.reset(); s.reset_cutoffs()
    wq = q18_case()
    s2 = Slimm.for_workload(wq, device=0, grouped=True)
    s2.push_bam_bytes(bam_record_bytes(wq.records, read_len=50), window=window)
    assert s2.q18_runs() == (3, 3)      # K.1 | K, W.2 | W, U.1 | U: each run holds its un-shortened namesake
    assert s2.get_profiles() is not None
    s.close(); s2.close()


def test_analysing_before_the_last_window_is_an_error():
    """A caller that forgets last != 0 would get the profile of a truncated record stream (windows are gathered and in
    flight inside the library): slimm_analyze_alignments refuses."""
    w = _named(make_workload(CONFIGS["config1"], seed=36, n_records=3_000))
    data = bam_record_bytes(w.records)
    s = Slimm.for_workload(w, device=0)
    got = capi.C.c_uint64(0)
    buf = (capi.C.c_uint8 * len(data)).from_buffer_copy(data)
    s._check(s.L.slimm_push_bam_bytes(s.ctx, buf, len(data), 0, capi.C.byref(got)))
    with pytest.raises(capi.SlimmError) as e:
        s.get_profiles()
    assert "last window" in str(e.value)
    s._check(s.L.slimm_push_bam_bytes(s.ctx, None, 0, 1, capi.C.byref(got)))
    assert s.get_profiles() is not None
    assert_matches_oracle(s, run_workload(w, use_qnames=True))
    s.close()


@pytest.mark.parametrize("window", [0, 250_007])
def test_unordered_file_decoded_on_the_device(window):
    """Any other order: name hash + check word computed on the device (the host reader's functions), then the grouping."""
    w = _named(make_workload(CONFIGS["config1"], seed=32, shuffled=True))
    _check(w, False, window, irregular=9)


def test_long_names_long_runs_and_interleaved_mates():
    w = _named(_interleave_mates(make_workload(SynthConfig("pairs", 20_000, 300, 6.0), seed=33, paired_frac=0.9)), long_names=True)
    _check(w, True, 300_001, irregular=3)
    _check(_named(one_long_read_workload(3_000)), True, 77_777)


def test_records_longer_than_a_piece_and_tiny_windows():
    """Sequences of tens of kilobases: records run over whole 16 KB pieces (no record starts in them; header look-alikes in
    random sequence bytes are verified away), and windows smaller than a record (several pushes without a complete one)."""
    w = _named(make_workload(SynthConfig("few", 400, 12, 2.0, bin_width=100, len_lo=5_000, len_hi=50_000, present_frac=0.8), seed=34))
    _check(w, True, 50_021, irregular=5, read_len=20_000)
    _check(w, True, 9_973, irregular=6, read_len=9_000)


def test_malformed_and_truncated_streams_are_errors():
    w = _named(make_workload(CONFIGS["config1"], seed=35, n_records=2_000))
    data = bytearray(bam_record_bytes(w.records))
    s = Slimm.for_workload(w, device=0)
    with pytest.raises(capi.SlimmError) as e:
        s.push_bam_bytes(bytes(data[:-10]))               # the last record is cut off
    assert "truncated" in str(e.value)
    s.reset()
    bad = bytearray(data)
    bad[200_000:200_004] = (5).to_bytes(4, "little") if False else bad[200_000:200_004]
    # a block_size below the fixed fields in the middle of the file
    import struct
    off, k = 0, 0
    while k < 1000:
        off += 4 + struct.unpack_from("<i", data, off)[0]
        k += 1
    bad[off:off + 4] = struct.pack("<i", 7)
    with pytest.raises(capi.SlimmError) as e:
        s.push_bam_bytes(bytes(bad), window=64_000)
    assert "bad BAM record" in str(e.value)
    # the forms do not mix within a file; the next file is fine again
    s.reset()
    s.push_records(w.records)
    with pytest.raises(capi.SlimmError):
        s.push_bam_bytes(bytes(data))
    s.reset(); s.reset_cutoffs()
    assert s.push_bam_bytes(bytes(data), window=123_457) == len(w.records)
    assert s.get_profiles() is not None
    assert_matches_oracle(s, run_workload(w, use_qnames=True))


def test_a_header_look_alike_at_a_piece_boundary_is_verified_away():
    """The piece's first record is GUESSED from the bytes (a plausible header whose two successors are plausible too).  A
    record whose quality string carries three well-formed little records right where a 16 KB piece begins is
    such a guess (the third piece's) -- a wrong one: the chain from the piece before does not end there (k_bam_verify walks the piece again
    from the true start)."""
    import struct
    w = _named(make_workload(CONFIGS["config1"], seed=36, n_records=600))
    r = w.records
    head = bam_record_bytes(Records(r.read_key[:20], r.flag[:20], r.ref_id[:20], r.begin_pos[:20], r.qname[:20]))
    rest = bam_record_bytes(Records(r.read_key[21:], r.flag[21:], r.ref_id[21:], r.begin_pos[21:], r.qname[21:]))
    fake = bam_record_bytes(Records(r.read_key[:3], r.flag[:3], r.ref_id[:3], r.begin_pos[:3], ["fake0", "fake1", "fake2"]))
    # record 20, by hand: a sequence of 30 000 bases whose qualities hold the look-alikes at stream offset 32 768
    name = r.qname[20].encode() + b"\x00"
    l_seq = 30_000
    fixed = struct.pack("<iiBBHHHIiii", int(r.ref_id[20]), int(r.begin_pos[20]), len(name), 255, 4680, 1, int(r.flag[20]), l_seq, -1, -1, 0)
    cigar = struct.pack("<I", (l_seq << 4) | 0)
    seq = bytes([0x11] * ((l_seq + 1) // 2))
    qual_at = len(head) + 4 + len(fixed) + len(name) + len(cigar) + len(seq)
    assert qual_at < 32_768 < qual_at + l_seq - len(fake)
    qual = bytearray([0x20] * l_seq)
    qual[32_768 - qual_at:32_768 - qual_at + len(fake)] = fake
    body = fixed + name + cigar + seq + bytes(qual)
    data = head + struct.pack("<i", len(body)) + body + rest
    o = run_workload(w, use_qnames=True)
    for window in (0, 40_000):
        s = Slimm.for_workload(w, device=0)
        assert s.push_bam_bytes(data, window=window) == len(r)
        assert s.get_profiles() is not None
        assert_matches_oracle(s, o)
        s.close()


def test_extreme_names_and_a_tiny_reference_set():
    """Names of 1 and of 254 characters, an empty name (l_read_name = 1), four reference sequences, records without one."""
    w = make_workload(SynthConfig("few", 3_000, 4, 1.5, bin_width=100, len_lo=5_000, len_hi=6_000, present_frac=1.0), seed=37)
    r = w.records
    ids = np.unique(r.read_key, return_inverse=True)[1]
    names = ["" if i == 5 else ("y" if i == 6 else ("n%d_" % i) + "z" * (254 - len("n%d_" % i) if i % 7 == 0 else i % 40)) for i in ids.tolist()]
    w = Workload(w.ref_names, w.ref_len, w.taxonomy, Records(r.read_key, r.flag, r.ref_id, r.begin_pos, names), w.avg_read_len,
                 w.options, "names", grouped=True)
    _check(w, True, 33_333, irregular=11)
    _check(Workload(w.ref_names, w.ref_len, w.taxonomy, w.records, w.avg_read_len, w.options, "names-any", grouped=False), False, 0)
