"""A synthetic BAM of a Workload, written fast enough for 100 M records: fixed-size records laid out with numpy,
compressed block by block (BGZF) on a thread pool, ten million records at a time.  Two kinds of content:

  realistic = False   every sequence byte 0x11, every quality 0x28, the name = the key in 16 hex digits: a file that
                      compresses 17-fold (rounds 1 - 4; kept as the easy case beside the other)
  realistic = True    what a BAM holds: random bases (4-bit codes of A/C/G/T), qualities from eight bins with run structure
                      (about 2.3 bits each), instrument-style names `A00123:45:HXYZABCDX:1:TTTT:XXXXXXX:YYYYYYYY` whose
                      three numeric fields are the key's 19 decimal digits; DEFLATE by libdeflate level 6 when the box has
                      it (what samtools links), zlib level 6 otherwise.  Compresses 3 - 4 fold, literal-heavy streams.

Either way equal keys <=> equal names.  For end-to-end timings of the `slimm` command
(scripts/cli_e2e.py, bench.py's cli_end_to_end leg); the readers' correctness tests use the independent, general writers
of tests/bam_io.py.  Written from the SAM/BAM specification."""
from __future__ import annotations

import os
import struct
import time
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

_HEX = np.frombuffer(b"0123456789abcdef", dtype="u1")


def _wrap(c: bytes, comp: bytes) -> bytes:
    return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, len(comp) + 25) + comp
            + struct.pack("<II", zlib.crc32(c) & 0xffffffff, len(c)))


def _bgzf(c: bytes, level: int = 1) -> bytes:
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    return _wrap(c, co.compress(c) + co.flush())


_libdeflate = None


def libdeflate():
    """libdeflate.so.0 through ctypes (the box has the library, no headers), or None."""
    global _libdeflate
    if _libdeflate is None:
        import ctypes as C
        try:
            L = C.CDLL("libdeflate.so.0")
            L.libdeflate_alloc_compressor.restype = C.c_void_p
            L.libdeflate_alloc_compressor.argtypes = [C.c_int]
            L.libdeflate_free_compressor.argtypes = [C.c_void_p]
            L.libdeflate_deflate_compress.restype = C.c_size_t
            L.libdeflate_deflate_compress.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
            _libdeflate = L
        except OSError:
            _libdeflate = False
    return _libdeflate or None


def _bgzf_many_libdeflate(chunks, level: int = 6):
    """One thread's share of blocks through one libdeflate compressor (ctypes calls release the GIL)."""
    import ctypes as C
    L = libdeflate()
    co = L.libdeflate_alloc_compressor(level)
    buf = C.create_string_buffer(0x10000 + 1024)
    out = []
    for c in chunks:
        n = L.libdeflate_deflate_compress(co, c, len(c), buf, len(buf))
        assert n, "libdeflate: block did not fit"
        out.append(_wrap(c, buf.raw[:n]))
    L.libdeflate_free_compressor(co)
    return b"".join(out)


_BASE_LUT = np.array([(1 << (b & 3)) << 4 | (1 << ((b >> 2) & 3)) for b in range(256)], dtype="u1")
_QUAL_BINS = np.array([2, 6, 15, 22, 27, 33, 37, 40], dtype="u1")
_NAME_PREFIX = b"A00123:45:HXYZABCDX:1:"


_pool_cache = {}


def _pools(read_len: int, rows: int = 1 << 18, seed: int = 7):
    """Pools of random sequences and quality strings the records draw from: the pool (40 MB) is far beyond DEFLATE's 32 KB
    window, so for the compressor every record's bases and qualities are fresh."""
    if (read_len, rows, seed) in _pool_cache:
        return _pool_cache[(read_len, rows, seed)]
    rng = np.random.Generator(np.random.PCG64(seed))
    seq = _BASE_LUT[rng.integers(0, 256, size=(rows, (read_len + 1) // 2), dtype="u1")]
    # qualities: a bin is kept with probability 0.6, else a new one drawn with weights that favour the high bins (about 2 bits
    # per quality value)
    r = rng.integers(0, 256, size=(rows, read_len), dtype="u1")
    change = rng.integers(0, 256, size=(rows, read_len), dtype="u1") < 102
    change[:, 0] = True
    cum = np.cumsum([0.03, 0.03, 0.05, 0.07, 0.10, 0.17, 0.35, 0.20])
    fresh = np.searchsorted(cum, (np.arange(256) + 0.5) / 256.0).astype("u1")[r]
    at = np.maximum.accumulate(np.where(change, np.arange(read_len, dtype=np.int32)[None, :], 0), axis=1)
    qual = _QUAL_BINS[np.take_along_axis(fresh, at, axis=1)]
    out = (np.ascontiguousarray(seq).view(f"S{seq.shape[1]}").reshape(rows),
           np.ascontiguousarray(qual).view(f"S{read_len}").reshape(rows))
    _pool_cache[(read_len, rows, seed)] = out
    return out


def _decimal_names(key: np.ndarray) -> np.ndarray:
    """[m, 44] bytes: PREFIX + 4 digits ':' 7 digits ':' 8 digits + NUL -- the 19 decimal digits of the (62-bit) key."""
    m = key.shape[0]
    out = np.empty((m, 44), dtype="u1")
    out[:, :22] = np.frombuffer(_NAME_PREFIX, dtype="u1")
    hi, lo = np.divmod(key.astype(np.uint64), np.uint64(10 ** 8))       # 11 digits | 8 digits
    hi2, mid = np.divmod(hi, np.uint64(10 ** 7))                         # 4 digits | 7 digits
    def put(v, ndig, col):
        v = v.astype(np.uint32)
        for d in range(ndig - 1, -1, -1):
            v, r = np.divmod(v, np.uint32(10))
            out[:, col + d] = r.astype("u1") + 48
    put(hi2, 4, 22)
    out[:, 26] = 58
    put(mid, 7, 27)
    out[:, 34] = 58
    put(lo, 8, 35)
    out[:, 43] = 0
    return out


def _record_dtype(read_len: int, l_name: int = 17):
    dt = np.dtype([("bs", "<i4"), ("ref", "<i4"), ("pos", "<i4"), ("lname", "u1"), ("mapq", "u1"), ("bin", "<u2"),
                   ("ncig", "<u2"), ("flag", "<u2"), ("lseq", "<i4"), ("nref", "<i4"), ("npos", "<i4"), ("tlen", "<i4"),
                   ("name", f"S{l_name}"), ("cigar", "<u4"), ("seq", f"S{(read_len + 1) // 2}"), ("qual", f"S{read_len}")])
    assert dt.itemsize == 36 + l_name + 4 + (read_len + 1) // 2 + read_len
    return dt


def write_synthetic_bam(path: str, ref_names, ref_len, records, read_len: int = 100,
                        hd: str = "@HD\tVN:1.6\tSO:unsorted\tGO:query", threads: int = 32, piece: int = 10_000_000,
                        realistic: bool = False, level: int = None) -> dict:
    """Returns {"records", "raw_bytes", "compressed_bytes", "seconds", "deflate"}."""
    t0 = time.time()
    n = len(records)
    dt = _record_dtype(read_len, 44 if realistic else 17)
    pool_seq = pool_qual = None
    if realistic:
        pool_seq, pool_qual = _pools(read_len)
    use_ld = realistic and libdeflate() is not None
    if level is None:
        level = 6 if realistic else 1
    text = (("" if not hd else hd + "\n") + "".join(f"@SQ\tSN:{nm}\tLN:{int(l)}\n" for nm, l in zip(ref_names, ref_len))).encode()
    head = bytearray(b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(ref_names)))
    for nm, l in zip(ref_names, ref_len):
        b = nm.encode() + b"\0"
        head += struct.pack("<i", len(b)) + b + struct.pack("<i", int(l))
    raw_bytes = len(head)

    l_name = 44 if realistic else 17
    n_seq = (read_len + 1) // 2
    # the same layout with opaque byte fields: numpy copies those with one memcpy per element (its S fields go byte by byte)
    vdt = np.dtype([(nm, dt.fields[nm][0] if dt.fields[nm][0].kind != "S" else np.dtype(f"V{dt.fields[nm][0].itemsize}")) for nm in dt.names])
    import threading
    local = threading.local()   # a builder thread reuses its piece buffer (first-touch page faults cost more than the fill)

    def piece_bytes(lo):
        hi = min(n, lo + piece)
        m = hi - lo
        if getattr(local, "buf", None) is None or local.buf.shape[0] < m:
            local.buf = np.zeros(min(piece, n), dtype=vdt)
            b = local.buf
            b["bs"] = dt.itemsize - 4
            b["lname"] = l_name
            b["mapq"] = 255
            b["bin"] = 4680
            b["ncig"] = 1
            b["lseq"] = read_len
            b["nref"] = -1
            b["npos"] = -1
            b["cigar"] = read_len << 4
            if not realistic:
                b["seq"] = np.frombuffer(b"\x11" * n_seq, dtype=f"V{n_seq}")
                b["qual"] = np.frombuffer(b"\x28" * read_len, dtype=f"V{read_len}")
        body = local.buf[:m]
        body["ref"] = records.ref_id[lo:hi]
        body["pos"] = records.begin_pos[lo:hi]
        body["flag"] = records.flag[lo:hi]
        if realistic:
            body["name"] = _decimal_names(records.read_key[lo:hi]).view("V44").reshape(m)
            rng = np.random.Generator(np.random.PCG64([lo, 0xba5e]))
            body["seq"] = pool_seq[rng.integers(0, pool_seq.shape[0], size=m)].view(f"V{n_seq}")
            body["qual"] = pool_qual[rng.integers(0, pool_qual.shape[0], size=m)].view(f"V{read_len}")
            return body.tobytes()
        kb = records.read_key[lo:hi].astype(">u8").view("u1").reshape(m, 8)
        names = np.empty((m, 17), dtype="u1")
        names[:, 0:16:2] = _HEX[kb >> 4]
        names[:, 1:16:2] = _HEX[kb & 15]
        names[:, 16] = 0
        body["name"] = names.view("V17").reshape(m)
        return body.tobytes()

    carry = bytes(head)   # bytes not yet written as whole 0xff00-byte blocks
    los = list(range(0, max(n, 1), piece))
    with open(path, "wb") as f, ThreadPoolExecutor(threads) as ex, ThreadPoolExecutor(3) as builders:
        ahead = [builders.submit(piece_bytes, lo) for lo in los[:3]]   # (numpy releases the GIL: three pieces in the making)
        for k, lo in enumerate(los):
            body = ahead.pop(0).result()
            if k + 3 < len(los):
                ahead.append(builders.submit(piece_bytes, los[k + 3]))
            raw = carry + body
            raw_bytes += len(body)
            del body
            last = k + 1 == len(los)
            whole = len(raw) if last else len(raw) // 0xff00 * 0xff00
            chunks = [raw[s:s + 0xff00] for s in range(0, whole, 0xff00)]
            if use_ld:
                per = max(1, (len(chunks) + 4 * threads - 1) // (4 * threads))
                for blob in ex.map(lambda part: _bgzf_many_libdeflate(part, level), [chunks[s:s + per] for s in range(0, len(chunks), per)]):
                    f.write(blob)
            else:
                for blk in ex.map(lambda c: _bgzf(c, level), chunks, chunksize=64):
                    f.write(blk)
            carry = raw[whole:]
            del raw, chunks
        f.write(_bgzf(b"", 6))   # the end-of-file marker block
    return {"records": n, "raw_bytes": raw_bytes, "compressed_bytes": os.path.getsize(path), "seconds": time.time() - t0,
            "deflate": (f"libdeflate level {level}" if use_ld else f"zlib level {level}"), "realistic": realistic}


def write_synthetic_sam(path: str, ref_names, ref_len, records, read_len: int = 100,
                        hd: str = "@HD\tVN:1.6\tSO:unsorted\tGO:query", piece: int = 2_000_000, threads: int = 12) -> dict:
    """The same records as SAM TEXT, fast enough for 100 M lines: fixed-width lines laid out with numpy (QNAME = the
    instrument-style name of the realistic BAM, FLAG in 4 and POS in 9 digits with leading zeros -- [0-9]+ like the
    specification says, and what strtoul makes of them is the number --, RNAME the reference's name; an unmapped record has
    RNAME `*` and an optional field that keeps its line as long as the others).  Needs reference names of one width.
    Returns {"records", "bytes", "seconds"}."""
    t0 = time.time()
    n = len(records)
    wn = len(ref_names[0])
    assert all(len(x) == wn for x in ref_names), "reference names of one width"
    names = np.frombuffer("".join(ref_names).encode(), dtype="u1").reshape(len(ref_names), wn)
    seq = np.frombuffer((b"ACGT" * ((read_len + 3) // 4))[:read_len], dtype="u1")
    qual = np.frombuffer(b"I" * read_len, dtype="u1")
    tail_a = b"\t255\t" + str(read_len).encode() + b"M\t*\t0\t0\t"
    L = 43 + 1 + 4 + 1 + wn + 1 + 9 + len(tail_a) + read_len + 1 + read_len + 1
    pad = wn - 1 - 6        # an unmapped line's optional field: TAB XP:Z: + this many characters
    assert pad >= 0

    def digits(v, nd, out, col):
        v = v.astype(np.uint32)
        for d in range(nd - 1, -1, -1):
            v, r = np.divmod(v, np.uint32(10))
            out[:, col + d] = r.astype("u1") + 48

    def piece_bytes(lo):
            hi = min(n, lo + piece)
            m = hi - lo
            out = np.empty((m, L), dtype="u1")
            out[:, :43] = _decimal_names(records.read_key[lo:hi])[:, :43]
            c = 43
            out[:, c] = 9
            digits(records.flag[lo:hi], 4, out, c + 1)
            c += 5
            out[:, c] = 9
            ref = records.ref_id[lo:hi]
            mapped = ref >= 0
            out[:, c + 1:c + 1 + wn] = names[np.where(mapped, ref, 0)]
            c += 1 + wn
            out[:, c] = 9
            digits((records.begin_pos[lo:hi].astype(np.int64) + 1).clip(0), 9, out, c + 1)
            c += 10
            out[:, c:c + len(tail_a)] = np.frombuffer(tail_a, dtype="u1")
            c += len(tail_a)
            out[:, c:c + read_len] = seq
            out[:, c + read_len] = 9
            out[:, c + read_len + 1:c + 2 * read_len + 1] = qual
            out[:, L - 1] = 10
            um = np.nonzero(~mapped)[0]
            if um.size:   # RNAME `*`: the rest of the line moves up, an optional field fills the end
                rows = out[um]
                r0 = 43 + 1 + 4 + 1
                fixed = rows[:, r0 + wn:L - 1].copy()
                rows[:, r0] = ord("*")
                rows[:, r0 + 1:r0 + 1 + fixed.shape[1]] = fixed
                tag = np.frombuffer(b"\tXP:Z:" + b"x" * pad, dtype="u1")
                rows[:, L - 1 - tag.size:L - 1] = tag
                out[um] = rows
            return out.tobytes()

    los = list(range(0, n, piece))
    with open(path, "wb") as f, ThreadPoolExecutor(threads) as ex:
        f.write(((hd + "\n") if hd else "").encode() + "".join(f"@SQ\tSN:{nm}\tLN:{int(l)}\n" for nm, l in zip(ref_names, ref_len)).encode())
        ahead = [ex.submit(piece_bytes, lo) for lo in los[:threads]]
        for k in range(len(los)):
            blob = ahead.pop(0).result()
            if k + threads < len(los):
                ahead.append(ex.submit(piece_bytes, los[k + threads]))
            f.write(blob)
            del blob
    return {"records": n, "bytes": os.path.getsize(path), "seconds": time.time() - t0}
