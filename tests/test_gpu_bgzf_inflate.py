"""BGZF blocks inflated on the device (slimm_amd/csrc/bgzf_inflate.hip; include/slimm_hip.h: slimm_bgzf_inflate) against zlib:
what seqan::BamFileIn does for the reference before a record is seen (call sites src/misc.hpp:498-522, src/slimm.hpp:194-208).
Every DEFLATE block type, every compression level's code shapes, empty and full-size blocks, and corrupt input."""
import ctypes as C
import struct
import zlib

import numpy as np
import pytest

from slimm_amd import capi

pytestmark = pytest.mark.gpu


def bgzf_block(data: bytes, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, raw=None) -> bytes:
    assert len(data) <= 65536
    if raw is None:
        c = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
        raw = c.compress(data) + c.flush()
    bsize = 12 + 6 + len(raw) + 8
    assert bsize <= 65536, bsize    # (BSIZE - 1 is a 16-bit field)
    head = b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, (bsize - 1) & 0xffff)
    return head + raw + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data))


def device_inflate(blob: bytes, cap=None):
    L = capi.lib()
    L.slimm_bgzf_inflate.restype = C.c_int
    src = np.frombuffer(blob, dtype=np.uint8) if blob else np.zeros(0, dtype=np.uint8)
    cap = (1 << 26) if cap is None else cap
    out = np.zeros(max(cap, 1), dtype=np.uint8)
    n = C.c_uint64()
    ms = C.c_double()
    err = C.create_string_buffer(256)
    rc = L.slimm_bgzf_inflate(0, src.ctypes.data_as(C.c_void_p), C.c_uint64(len(blob)), out.ctypes.data_as(C.c_void_p), C.c_uint64(cap),
                              C.byref(n), C.byref(ms), err, C.c_uint64(256))
    return rc, bytes(out[:n.value]) if rc == 0 else b"", err.value.decode(), ms.value


def payloads(rng):
    bam_like = b"".join(struct.pack("<iiiBBHHHIiii", 200 + k % 7, k % 50, 1000 * k, 9, 30, 4680, 1, 0, 100, -1, -1, 0) + b"read%05d\0" % (k // 3)
                        + bytes(50) + b"\x28" * 100 for k in range(300))
    return [b"", b"a", b"abc" * 7, bytes(1000), bytes(rng.integers(0, 256, 3000, dtype=np.uint8)),      # tiny, runs, incompressible
            bytes(rng.integers(0, 4, 65280, dtype=np.uint8)), bam_like[:65280], b"ACGT" * 16320,
            bytes(rng.integers(0, 256, 64000, dtype=np.uint8)),                                            # stored by zlib
            (b"x" * 300 + bytes(rng.integers(0, 256, 40, dtype=np.uint8))) * 150,
            # matches that overlap themselves at every distance below 8 (the pattern-repeating copy), of all lengths
            b"".join((b"abcdefg"[:d] * (3 + k % 60))[:3 + (k * 7) % 300] + bytes([k % 251]) for k in range(400) for d in range(1, 8))[:65000]]


def test_every_block_type_and_level_equals_zlib():
    rng = np.random.default_rng(5)
    blocks, want = [], []
    for data in payloads(rng):
        for level, strat in ((0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY),
                             (9, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE)):
            blocks.append(bgzf_block(data, level, strat))
            want.append(data)
    # several DEFLATE blocks in one BGZF block (Z_FULL_FLUSH in the middle), and the empty end-of-file block
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    a, b = b"hello world " * 500, bytes(rng.integers(0, 9, 20000, dtype=np.uint8))
    raw = c.compress(a) + c.flush(zlib.Z_FULL_FLUSH) + c.compress(b) + c.flush()
    blocks.append(bgzf_block(a + b, raw=raw))
    want.append(a + b)
    blocks.append(bgzf_block(b""))
    want.append(b"")
    rc, got, err, _ = device_inflate(b"".join(blocks))
    assert rc == 0, err
    assert got == b"".join(want)
    # one at a time too (a lane's tables must not depend on what another block left behind)
    for blk, w in zip(blocks[::5], want[::5]):
        rc, got, err, _ = device_inflate(blk)
        assert rc == 0 and got == w, err


def test_many_blocks_more_than_resident_lanes():
    """More blocks than the 512 x 64 lanes of a launch: every lane takes several, its scratch and tables are reused."""
    rng = np.random.default_rng(6)
    pool = [bytes(rng.integers(0, 1 + k % 200, 900 + 37 * (k % 23), dtype=np.uint8)) for k in range(97)]
    n = 40_000
    blob = b"".join(bgzf_block(pool[k % 97], 1 + k % 9) for k in range(97))   # (97 distinct blocks, repeated)
    blocks = [bgzf_block(pool[k], 1 + k % 9) for k in range(97)]
    blob = b"".join(blocks[k % 97] for k in range(n))
    rc, got, err, _ = device_inflate(blob)
    assert rc == 0, err
    assert got == b"".join(pool[k % 97] for k in range(n))


def test_corrupt_blocks_are_errors_not_output():
    rng = np.random.default_rng(7)
    data = bytes(rng.integers(0, 20, 30000, dtype=np.uint8))
    good = bgzf_block(data)
    rc, got, err, _ = device_inflate(good)
    assert rc == 0 and got == data
    # ISIZE that does not fit the stream (shorter and longer), a flipped byte in the middle of the codes, a reserved block type
    bad = [good[:-4] + struct.pack("<I", len(data) - 1), good[:-4] + struct.pack("<I", len(data) + 1)]
    flip = bytearray(good)
    flip[18 + 3] ^= 0x55
    bad.append(bytes(flip))
    bad.append(bgzf_block(b"zzz", raw=b"\x07" + bytes(8)))            # BFINAL, BTYPE = 3
    bad.append(bgzf_block(b"abc", raw=b"\x01\x03\x00\xfc\xfe" + b"abc"))   # stored: LEN / NLEN do not complement
    for blk in bad:
        rc, got, err, _ = device_inflate(good + blk + good)
        assert rc != 0 and "corrupt BGZF block" in err and "block 1" in err, (rc, err)
    # not BGZF at all / cut in the middle of a block / an output buffer too small
    assert device_inflate(b"\x1f\x8b\x08\x00" + bytes(40))[0] != 0
    rc, _, err, _ = device_inflate(good[:len(good) // 2])
    assert rc != 0 and "truncated" in err
    rc, _, err, _ = device_inflate(good, cap=100)
    assert rc != 0 and "too small" in err


def _bam_file(tmp_path, w, names, seed, level=6):
    """A BAM file of this repository's writer + where its alignment records start: (compressed bytes of the blocks from the
    first record-bearing one on, inflated bytes to skip there, the records' bytes)."""
    import gzip
    from tests.bam_io import bam_record_bytes, write_bam
    from slimm_amd.workload import Records
    r = w.records
    rec = Records(r.read_key, r.flag, r.ref_id, r.begin_pos, names)
    p = str(tmp_path / f"f{seed}.bam")
    write_bam(p, w.ref_names, w.ref_len, rec, read_len=w.avg_read_len, irregular_seed=seed)
    want = bam_record_bytes(rec, read_len=w.avg_read_len, irregular_seed=seed)
    blob = open(p, "rb").read()
    # walk the blocks: inflated offset of every block; the records are the file's last len(want) inflated bytes
    offs, p0, total = [], 0, 0
    while p0 < len(blob):
        bsize = blob[p0 + 16] + (blob[p0 + 17] << 8) + 1
        isize = struct.unpack("<I", blob[p0 + bsize - 4:p0 + bsize])[0]
        offs.append((p0, total))
        total += isize
        p0 += bsize
    start = total - len(want)
    k = max(i for i, (_, t) in enumerate(offs) if t <= start)
    return blob[offs[k][0]:], start - offs[k][1], want, rec


@pytest.mark.parametrize("grouped", [True, False])
def test_records_from_compressed_blocks_equal_the_oracle(tmp_path, grouped):
    """slimm_push_bgzf_blocks: the file's BGZF blocks go to the device as they are -- inflate, record boundaries, fields, names
    all happen there -- in one window, in small windows, and alternating with windows the host inflated
    (slimm_push_bam_bytes): the same records, the oracle's profile."""
    from oracle.binding import run_workload
    from slimm_amd.profiler import Slimm
    from slimm_amd.synth import CONFIGS, make_workload
    from slimm_amd.workload import Records, Workload
    from tests.helpers import assert_matches_oracle
    w = make_workload(CONFIGS["config2"], seed=71, n_records=60_000)
    r = w.records
    names = ["q%x" % k + "n" * int(k % 19) for k in r.read_key.tolist()]
    if not grouped:
        order = np.random.default_rng(2).permutation(len(r))
        r = Records(r.read_key[order], r.flag[order], r.ref_id[order], r.begin_pos[order])
        names = [names[i] for i in order]
        w = Workload(w.ref_names, w.ref_len, w.taxonomy, r, w.avg_read_len, w.options, "any", grouped=False)
    blocks, skip, want, rec = _bam_file(tmp_path, w, names, seed=9)
    assert skip > 0 and len(blocks) < len(want)
    wq = Workload(w.ref_names, w.ref_len, w.taxonomy, rec, w.avg_read_len, w.options, "bam", grouped=grouped)
    o = run_workload(wq, use_qnames=True)
    for window, host_every in ((0, 0), (200_000, 0), (70_000, 2), (70_000, 3), (1, 0)):
        s = Slimm.for_workload(wq, device=0, grouped=grouped)
        assert s.push_bgzf_blocks(blocks, skip=skip, window=window, host_every=host_every) == len(rec)
        assert s.get_profiles() is not None
        assert_matches_oracle(s, o)
        s.close()
    # a corrupt block in the middle of the file: an error of the push, nothing decoded from it
    bad = bytearray(blocks)
    bad[len(bad) // 2] ^= 0x40
    s = Slimm.for_workload(wq, device=0, grouped=grouped)
    with pytest.raises(capi.SlimmError) as e:
        s.push_bgzf_blocks(bytes(bad), skip=skip, window=150_000)
    assert "BGZF" in str(e.value) or "BAM record" in str(e.value)
    s.close()


def test_window_buffers_are_sized_by_the_file(tmp_path):
    """the first push of a large file used to reserve every buffer for the largest possible window (17 GB per
    context).  With slimm_set_input_size_hint the library reserves what the file will use; without it nothing ahead.  Same
    records either way; slimm_window_memory says what is held."""
    from oracle.binding import run_workload
    from slimm_amd.profiler import Slimm
    from slimm_amd.synth import CONFIGS, make_workload
    from slimm_amd.workload import Workload
    from tests.helpers import assert_matches_oracle
    w = make_workload(CONFIGS["config2"], seed=73, n_records=40_000)
    names = ["s%x" % k for k in w.records.read_key.tolist()]
    blocks, skip, want, rec = _bam_file(tmp_path, w, names, seed=13)
    wq = Workload(w.ref_names, w.ref_len, w.taxonomy, rec, w.avg_read_len, w.options, "bam", grouped=True)
    o = run_workload(wq, use_qnames=True)
    held = []
    for hint in (0, len(blocks), 40 * len(blocks)):       # not told / the truth / a caller that overstates
        s = Slimm.for_workload(wq, device=0, grouped=True)
        if hint:
            s._check(s.L.slimm_set_input_size_hint(s.ctx, hint))
        assert s.push_bgzf_blocks(blocks, skip=skip, window=300_000) == len(rec)
        m = capi.C.c_uint64(0)
        s._check(s.L.slimm_window_memory(s.ctx, capi.C.byref(m)))
        held.append(m.value)
        assert s.get_profiles() is not None
        assert_matches_oracle(s, o)
        with pytest.raises(capi.SlimmError):
            s._check(s.L.slimm_set_input_size_hint(s.ctx, 1))     # (only before a file's first window)
        s.reset(); s.reset_cutoffs()
        s._check(s.L.slimm_set_input_size_hint(s.ctx, len(blocks)))
        assert s.push_bgzf_blocks(blocks, skip=skip, window=0) == len(rec)
        assert_matches_oracle(s, o) if s.get_profiles() is not None else None
        s.close()
    # a file of a few megabytes holds far less than a gigabyte, told or not; a caller that overstates gets what it asked for
    assert held[0] < (1 << 29) and held[1] < (1 << 29) and held[2] >= held[1], held


def test_a_skip_larger_than_the_window_slack(tmp_path):
    """A caller may hand over a file's blocks from its very first one and name the whole BAM header as `skip` -- 17 MiB here
    (hundreds of thousands of contigs), more than the 16 MiB of slack in front of a window buffer: the blocks that lie wholly
    inside the header are dropped on the host, what is left to skip is less than one block.  A skip that ends past the
    window's bytes is an error."""
    from oracle.binding import run_workload
    from slimm_amd.profiler import Slimm
    from slimm_amd.synth import CONFIGS, make_workload
    from slimm_amd.workload import Workload
    from tests.helpers import assert_matches_oracle
    w = make_workload(CONFIGS["config1"], seed=72, n_records=5_000)
    names = ["h%x" % k for k in w.records.read_key.tolist()]
    blocks, skip, want, rec = _bam_file(tmp_path, w, names, seed=11)
    rng = np.random.default_rng(5)
    header = bytes(rng.integers(0, 64, size=17 * (1 << 20) + 12_345, dtype=np.uint8))    # stands for header text + contig table
    front = b"".join(bgzf_block(header[i:i + 65_280], level=1) for i in range(0, len(header), 65_280))
    wq = Workload(w.ref_names, w.ref_len, w.taxonomy, rec, w.avg_read_len, w.options, "bam", grouped=True)
    o = run_workload(wq, use_qnames=True)
    for window in (0, len(front) + 20_000):      # (the skip lies in the first window: the header's blocks and a few more)
        s = Slimm.for_workload(wq, device=0, grouped=True)
        assert s.push_bgzf_blocks(front + blocks, skip=len(header) + skip, window=window) == len(rec)
        assert s.get_profiles() is not None
        assert_matches_oracle(s, o)
        s.close()
    s = Slimm.for_workload(wq, device=0, grouped=True)
    with pytest.raises(capi.SlimmError) as e:
        s.push_bgzf_blocks(blocks, skip=len(want) + skip + 70_000)
    assert "skip" in str(e.value)
    s.close()


def _inflate_with(blob: bytes, how: int):
    L = capi.lib()
    src = np.frombuffer(blob, dtype=np.uint8)
    out = np.zeros(1 << 26, dtype=np.uint8)
    n, ms, err, lanes = C.c_uint64(), C.c_double(), C.create_string_buffer(256), C.c_uint32()
    rc = L.slimm_bgzf_inflate_with(0, src.ctypes.data_as(C.c_void_p), C.c_uint64(len(blob)), out.ctypes.data_as(C.c_void_p), C.c_uint64(out.size),
                                   C.byref(n), C.byref(ms), err, C.c_uint64(256), how, C.byref(lanes))
    return rc, bytes(out[:n.value]) if rc == 0 else b"", lanes.value


def test_the_two_phase_kernels_take_every_huffman_block_themselves():
    """slimm_bgzf_inflate_with: the two-phase kernels (bgzf_tokens.hip: Huffman decode into literals + match tokens, then
    the matches filled by pointer jumping in LDS) must inflate every block made of fixed / dynamic Huffman blocks THEMSELVES --
    their hand-over to the lane-per-block kernel is for stored blocks and irregular streams, and would hide a fault of theirs
    behind a correct result.  Both paths give zlib's bytes; the count of handed-over blocks is what is asserted here."""
    rng = np.random.default_rng(6)
    records = b"".join(struct.pack("<iiiBBHHHIiii", 230, k % 500, 977 * k, 44, 255, 4680, 1, 0, 100, -1, -1, 0) + b"A00123:45:HXYZABCDX:1:%04d:%07d:%08d\0" % (k % 9000, k * 13, k * 7919)
                       + bytes(rng.choice(np.frombuffer(bytes([0x11, 0x12, 0x14, 0x18, 0x21, 0x22, 0x24, 0x28, 0x41, 0x42, 0x44, 0x48, 0x81, 0x82, 0x84, 0x88]), dtype=np.uint8), 50))
                       + bytes(rng.choice(np.frombuffer(bytes([2, 6, 15, 22, 27, 33, 37, 40]), dtype=np.uint8), 100)) for k in range(290))
    huffman = [b"a", b"abc" * 7, bytes(1000), bytes(65280), records[:65280], b"ACGT" * 16320, bytes(rng.integers(0, 4, 65280, dtype=np.uint8)),
               (b"the quick brown fox jumps over the lazy dog " * 2000)[:65000], (b"x" * 300 + bytes(rng.integers(0, 256, 40, dtype=np.uint8))) * 150,
               bytes(rng.integers(0, 8, 30000, dtype=np.uint8)) + bytes(35000)]
    blocks, want = [], []
    for data in huffman:
        for level, strat in ((1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED),
                             (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE), (6, zlib.Z_FILTERED)):
            blocks.append(bgzf_block(data, level, strat))
            want.append(data)
    # several DEFLATE blocks in one BGZF block (zlib starts a new one every 16 K symbols; a full flush in the middle)
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    raw = c.compress(records[:30000]) + c.flush(zlib.Z_FULL_FLUSH) + c.compress(records[30000:60000]) + c.flush()
    stored_inside = b"\x00\x00\xff\xff" in raw     # (a full flush ends in an empty STORED block: that one block is handed over)
    assert stored_inside
    blocks.append(bgzf_block(records[:60000], raw=raw))
    want.append(records[:60000])
    blob = b"".join(blocks) * 3       # (192 blocks and more: three waves of lanes)
    rc, got, lanes = _inflate_with(blob, 0)
    assert rc == 0 and got == b"".join(want) * 3
    assert lanes == (3 if stored_inside else 0)
    rc, got1, lanes1 = _inflate_with(blob, 1)
    assert rc == 0 and got1 == got and lanes1 == 3 * len(blocks)
    # stored blocks go the other way, and mix with the rest
    mixed = bgzf_block(records[:60000], 0) + bgzf_block(records[:60000], 6) + bgzf_block(bytes(rng.integers(0, 256, 60000, dtype=np.uint8)), 6)
    rc, got, lanes = _inflate_with(mixed, 0)
    assert rc == 0 and lanes == 2 and got[:120000] == records[:60000] * 2


def test_a_block_that_claims_no_bytes_is_inflated_all_the_same():
    """ISIZE = 0 is the end-of-file block (a fixed-code block with its end-of-block code only), which nothing inflates; any other
    payload under an ISIZE of 0 -- a flipped length field -- must still give no byte and the CRC of none, through either path
    (found by scripts/stress_inflate.py: the parser used to drop every block of ISIZE 0)."""
    good = bgzf_block(b"hello world")
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    raw = c.compress(b"a") + c.flush()
    lying = bgzf_block(b"a", raw=raw)[:-4] + struct.pack("<I", 0)
    eof = bgzf_block(b"", raw=b"\x03\x00")
    empty_stored = bgzf_block(b"", raw=b"\x01\x00\x00\xff\xff")
    for how in (0, 1):
        rc, got, _ = _inflate_with(good + eof + good + empty_stored + good + eof, how)
        assert rc == 0 and got == b"hello world" * 3
        rc, got, _ = _inflate_with(good + lying + good, how)
        assert rc != 0


def _hand_made_fixed_block(symbols) -> bytes:
    """A raw DEFLATE stream of ONE fixed-Huffman block made by hand: symbols = [("lit", byte) | ("match", length, distance)].
    (zlib never emits a distance beyond the output; the writer wave has to refuse one.)"""
    bits = []
    def put(v, n):           # n bits, least significant first (extra bits, header)
        for k in range(n):
            bits.append((v >> k) & 1)
    def code(v, n):          # a Huffman code: most significant bit first
        for k in range(n - 1, -1, -1):
            bits.append((v >> k) & 1)
    def litlen(s):
        if s < 144:
            code(0x30 + s, 8)
        elif s < 256:
            code(0x190 + (s - 144), 9)
        elif s < 280:
            code(s - 256, 7)
        else:
            code(0xc0 + (s - 280), 8)
    lbase = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
    lext = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]
    dbase = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577]
    dext = [0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13]
    put(1, 1)
    put(1, 2)                # BFINAL, fixed codes
    for s in symbols:
        if s[0] == "lit":
            litlen(s[1])
        else:
            _, length, dist = s
            li = max(i for i in range(29) if lbase[i] <= length)
            litlen(257 + li)
            put(length - lbase[li], lext[li])
            di = max(i for i in range(30) if dbase[i] <= dist)
            code(di, 5)
            put(dist - dbase[di], dext[di])
    litlen(256)
    while len(bits) % 8:
        bits.append(0)
    return bytes(sum(bits[i + k] << k for k in range(8)) for i in range(0, len(bits), 8))


def test_decoder_and_writer_waves_hand_over():
    """k_inflate_decode is two waves per 64 blocks (bgzf_tokens.hip): a decoder wave and a writer wave with a ring of one word
    per lane and step between them.  What the hand-over has to survive: headers in the middle of a burst (a BGZF block of a
    hundred DEFLATE blocks), lanes that end at very different steps, launches whose last workgroup is partly empty, and a
    stream only the WRITER can refuse -- a distance that reaches in front of the output -- next to streams that go on."""
    rng = np.random.default_rng(11)
    text = (b"the quick brown fox jumps over the lazy dog " * 1500)[:60000]
    # a DEFLATE block every 300 - 900 bytes: 100 headers inside one BGZF block, fixed and dynamic codes in turn
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    raw, p = b"", 0
    while p < len(text):
        n = int(rng.integers(300, 900))
        raw += c.compress(text[p:p + n]) + c.flush(zlib.Z_BLOCK)
        p += n
    raw += c.flush()
    many_headers = bgzf_block(text, raw=raw)
    hand = [("lit", b) for b in b"abcdef"] + [("match", 20, 3), ("lit", 0x7a), ("match", 258, 1), ("match", 4, 280)]
    want_hand = bytearray(b"abcdef")
    for s in hand[6:]:
        if s[0] == "lit":
            want_hand.append(s[1])
        else:
            for _ in range(s[1]):
                want_hand.append(want_hand[-s[2]])
    hand_block = bgzf_block(bytes(want_hand), raw=_hand_made_fixed_block(hand))
    sizes = [1, 2, 3, 5, 63, 64, 65, 255, 256, 257, 4095, 4096, 4097, 12287, 12288, 12289, 65279, 65280]
    small = [bytes(rng.integers(0, 7, n, dtype=np.uint8)) for n in sizes]
    for n_copies in (1, 3):          # 1: one partly empty workgroup; 3: 66 blocks = one full pair of waves and two lanes
        blocks = [many_headers, hand_block] + [bgzf_block(d, 1 + k % 9) for k, d in enumerate(small)]
        blocks = blocks * n_copies + [many_headers] * (n_copies * 2)
        want = (text + bytes(want_hand) + b"".join(small)) * n_copies + text * (n_copies * 2)
        rc, got, lanes = _inflate_with(b"".join(blocks), 0)
        assert rc == 0 and got == want and lanes == 0, (rc, lanes)
    # a distance in front of the output: behind 6 literals nothing lies 7 back.  zlib refuses it; so must the device, whichever
    # kernel ends up saying so -- and the blocks around it are what they were
    far = _hand_made_fixed_block([("lit", b) for b in b"abcdef"] + [("match", 5, 7)])
    with pytest.raises(zlib.error):
        zlib.decompressobj(-15).decompress(far)
    bad_block = bgzf_block(b"abcdef" + b"?" * 5, raw=far)
    rc, got, err, _ = device_inflate(many_headers + bad_block + hand_block)
    assert rc != 0 and "corrupt BGZF block" in err and "block 1" in err, (rc, err)
    rc, got, err, _ = device_inflate(many_headers + hand_block)
    assert rc == 0 and got == text + bytes(want_hand)
