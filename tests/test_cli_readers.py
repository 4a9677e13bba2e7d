"""CPU tests of the host-side readers of the `slimm` command line (SAM / BAM / header parsing), through
`slimm --dump-records`: files written by the independent Python writers in tests/bam_io.py must decode to the
records they were written from.  No GPU is touched."""
import os
import subprocess

import numpy as np
import pytest

from slimm_amd.synth import CONFIGS, make_workload
from tests.bam_io import qnames_of, write_bam, write_sam
from tests.cases import q18_apart_case, q18_case, tiny_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "slimm_amd", "slimm")


def dump(path, extra=()):
    out = subprocess.run([CLI, "--dump-records", *extra, path], capture_output=True, text=True, check=True).stdout.split("\n")
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


@pytest.mark.parametrize("writer", [write_sam, write_bam])
def test_reader_hands_out_the_canonical_identity_of_q18(tmp_path, writer):
    """The reference's key is qName + ".1" / ".2" / "" (src/slimm.hpp:204-208): records get one (key, mate) iff they get one
    key string there.  The reader keys by the canonical base and sets the base's mate bit in the flag it hands out."""
    w = q18_case()
    p = str(tmp_path / "q18")
    writer(p, w.ref_names, w.ref_len, w.records, read_len=w.avg_read_len)
    _, _, recs = dump(p)
    q, raw = qnames_of(w.records), w.records.flags_in_file().tolist()
    assert [r[0] for r in recs] == q
    assert [int(r[1]) for r in recs] == w.records.flag.tolist()          # canonical: mate bits added
    ident = {}
    for r, name, f in zip(recs, q, raw):
        key_string = name + (".1" if f & 0x40 else ".2" if f & 0x80 else "")
        mate = 1 if int(r[1]) & 0x40 else 2 if int(r[1]) & 0x80 else 0
        assert ident.setdefault(key_string, (r[5], mate)) == (r[5], mate)
    assert len(set(ident.values())) == len(ident) == 12                   # (the unmapped record shares "U.1" with a mapped one)


@pytest.mark.parametrize("writer", [write_sam, write_bam])
def test_reader_tells_shortened_names_that_stand_apart_from_their_namesakes(tmp_path, writer):
    """Q18 on a file grouped by QNAME (include/slimm_hip.h): a run of shortened names only -- the unflagged `r.1`, fifty reads
    behind `r`/0x40 -- means the file must go through the any-order path; q18_case, where every shortened name stands next to
    its un-shortened namesake, must not."""
    for mk, want in ((q18_apart_case, "1"), (lambda: q18_apart_case(tail=("r.2",)), "1"), (q18_case, "0"), (tiny_case, "0")):
        w = mk()
        p = str(tmp_path / "x")
        writer(p, w.ref_names, w.ref_len, w.records, read_len=w.avg_read_len)
        err = subprocess.run([CLI, "--dump-records", p], capture_output=True, text=True, check=True).stderr
        assert f"#q18_regroup_needed\t{want}" in err, (w.name, err)


def test_bam_spanning_many_bgzf_blocks(tmp_path):
    w = make_workload(CONFIGS["config2"], seed=32, n_records=40_000)   # ~6 MB of BAM records -> ~100 BGZF blocks
    p = str(tmp_path / "big.bam")
    write_bam(p, w.ref_names, w.ref_len, w.records)
    _, refs, recs = dump(p)
    assert len(refs) == 5000 and len(recs) == 40_000
    assert [int(r[3]) for r in recs] == w.records.begin_pos.tolist()


def test_bam_with_irregular_records_is_split_correctly_by_the_parallel_reader(tmp_path, monkeypatch):
    """Record starts are found chunk-wise in parallel from guessed first records (verified against the chunk before).
    Irregular records -- names of 1..60 bytes, 0..3 CIGAR operations, reads of 0..400 bases, auxiliary bytes that
    contain whole fake record headers -- must come out exactly as a sequential walk would give them."""
    import struct
    from tests.bam_io import _bgzf_block, sam_header
    rng = np.random.default_rng(5)
    n, nref = 60_000, 7
    ref_names = [f"ref{i}" for i in range(nref)]
    ref_len = [100_000 + i for i in range(nref)]
    text = sam_header(ref_names, np.array(ref_len), "@HD\tVN:1.6\tSO:unsorted").encode()
    out = bytearray(b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", nref))
    for nm, l in zip(ref_names, ref_len):
        b = nm.encode() + b"\0"
        out += struct.pack("<i", len(b)) + b + struct.pack("<i", l)
    expect = []
    fake = struct.pack("<iiiBBHHHIiii", 200, 1, 5, 4, 0, 0, 1, 0, 50, -1, -1, 0) + b"abc\0"   # looks like a record start
    for i in range(n):
        name = ("r%d_" % i + "x" * int(rng.integers(0, 50)))[:int(rng.integers(1, 60))].encode() + b"\0"
        ncig = int(rng.integers(0, 4))
        lseq = int(rng.choice([0, 1, 35, 100, 151, 400]))
        ref = int(rng.integers(-1, nref))
        pos = int(rng.integers(-1, 90_000))
        flag = int(rng.choice([0, 4, 16, 65, 129, 256, 2048]))
        aux = (fake * int(rng.integers(0, 3))) + bytes(rng.integers(0, 256, size=int(rng.integers(0, 40)), dtype=np.uint8))
        body = (struct.pack("<iiBBHHHIiii", ref, pos, len(name), 30, 4680, ncig, flag, lseq, -1, -1, 0) + name
                + struct.pack("<%dI" % ncig, *([(10 << 4)] * ncig)) + bytes((lseq + 1) // 2) + bytes(lseq) + aux)
        out += struct.pack("<i", len(body)) + body
        expect.append((name[:-1].decode(), flag, ref, pos, lseq))
    p = str(tmp_path / "irregular.bam")
    with open(p, "wb") as f:
        for s0 in range(0, len(out), 0xff00):
            f.write(_bgzf_block(bytes(out[s0:s0 + 0xff00])))
        f.write(_bgzf_block(b""))
    for threads in ("1", "8"):
        _, refs, recs = dump(p, ("--decode-threads", threads))
        assert len(recs) == n
        got = [(r[0], int(r[1]), int(r[2]), int(r[3]), int(r[4])) for r in recs]
        assert got == expect


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


# ---------------------------------------------------------------- name-hash collisions
_M64 = (1 << 64) - 1


def _hash_state(words, n):
    """hash_read_name of alignment_file.cpp up to (not including) its tail step, for whole 8-byte words."""
    h = 0x9E3779B97F4A7C15 ^ ((n * 0xff51afd7ed558ccd) & _M64)
    for w in words:
        h ^= w
        h = (h * 0xff51afd7ed558ccd) & _M64
        h ^= h >> 32
    return h


def colliding_names():
    """Two different printable 16-byte names with the same hash: every step of the hash is a bijection of its state, so
    after two words the states are equal iff (state_1 ^ word_2) are -- the second word of one name is chosen to cancel
    the difference the first words made."""
    import itertools
    import struct

    a = b"readAAAAcollideX"
    wa = struct.unpack("<2Q", a)
    sa = _hash_state(wa[:1], 16)
    for tag in itertools.product(b"BCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789", repeat=4):
        first = b"read" + bytes(tag)
        (wb0,) = struct.unpack("<Q", first)
        sb = _hash_state([wb0], 16)
        wb1 = wa[1] ^ sa ^ sb
        second = struct.pack("<Q", wb1)
        if all(33 <= c < 127 for c in second):
            b = first + second
            assert b != a
            return a.decode(), b.decode()
    raise AssertionError("no printable partner found")


@pytest.mark.parametrize("writer", [write_sam, write_bam])
def test_adjacent_names_with_equal_hashes_stay_two_reads(tmp_path, writer):
    """Name-grouped input: the reader compares the NAMES of adjacent records whenever their keys are equal, so two
    different names can never be taken for one read (the key of the second one is moved to the next free value)."""
    from slimm_amd.workload import Records

    a, b = colliding_names()
    w = tiny_case()
    names = ["before", a, a, b, b, b, a, "after"]    # ... and the same name coming back later gets its own hash again
    n = len(names)
    rec = Records(np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint16), np.zeros(n, dtype=np.int32),
                  np.arange(n, dtype=np.int32) * 10, names)
    p = str(tmp_path / ("c." + ("sam" if writer is write_sam else "bam")))
    writer(p, w.ref_names, w.ref_len, rec, read_len=50)
    _, _, recs = dump(p)
    assert [r[0] for r in recs] == names
    keys = [r[5] for r in recs]
    assert keys[1] == keys[2] and keys[3] == keys[4] == keys[5]
    assert keys[2] != keys[3] and keys[5] != keys[6]
    assert len({keys[0], keys[1], keys[3], keys[7]}) == 4
    # without the name comparison the two names WOULD share a key: the construction is a real collision
    assert int(keys[3]) == (int(keys[1]) + 1) % (1 << 62) and keys[6] == keys[1]


@pytest.mark.parametrize("window_mb", [1, 3, 64])
@pytest.mark.parametrize("mmap", [True, False])
def test_raw_windows_are_the_inflated_record_bytes(tmp_path, window_mb, mmap):
    """AlignmentFile::read_raw (what `slimm` feeds the device decoder, slimm_push_bam_bytes): the windows, concatenated,
    are exactly the alignment-record bytes behind the BAM header -- whatever the window size, with the compressed bytes
    read in place from a mapping of the file or through buffered reads; the last window announces itself or an empty
    read follows it."""
    from tests.bam_io import bam_record_bytes
    w = make_workload(CONFIGS["config2"], seed=51, n_records=40_000)
    p = str(tmp_path / "x.bam")
    write_bam(p, w.ref_names, w.ref_len, w.records, read_len=w.avg_read_len, irregular_seed=4)
    want = bam_record_bytes(w.records, read_len=w.avg_read_len, irregular_seed=4)
    flags = ["--window-mb", str(window_mb)] + ([] if mmap else ["--no-mmap"])
    r = subprocess.run([CLI, "--dump-raw"] + flags + [p], capture_output=True)
    assert r.returncode == 0, r.stderr[-500:]
    assert r.stdout == want
    sizes = [int(ln.split("\t")[1]) for ln in r.stderr.decode().splitlines() if ln.startswith("window")]
    assert sum(sizes) == len(want) and max(sizes) <= window_mb << 20
    assert len(sizes) >= (len(want) + (window_mb << 20) - 1) // (window_mb << 20)
    # a truncated file is an error, not a short stream
    blob = open(p, "rb").read()
    bad = str(tmp_path / "cut.bam")
    open(bad, "wb").write(blob[:len(blob) * 2 // 3])
    r = subprocess.run([CLI, "--dump-raw"] + flags + [bad], capture_output=True)
    assert r.returncode != 0 and b"truncated" in r.stderr
