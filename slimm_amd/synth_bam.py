"""A synthetic BAM of a Workload, written fast enough for 100 M records: fixed-size records laid out with numpy,
compressed block by block (BGZF, zlib level 1) on a thread pool, ten million records at a time.  The read name of a
record is its key in 16 hex digits, so equal keys <=> equal names.  For end-to-end timings of the `slimm` command
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


def _bgzf(c: bytes, level: int = 1) -> bytes:
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    comp = co.compress(c) + co.flush()
    return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, len(comp) + 25) + comp
            + struct.pack("<II", zlib.crc32(c) & 0xffffffff, len(c)))


def _record_dtype(read_len: int):
    dt = np.dtype([("bs", "<i4"), ("ref", "<i4"), ("pos", "<i4"), ("lname", "u1"), ("mapq", "u1"), ("bin", "<u2"),
                   ("ncig", "<u2"), ("flag", "<u2"), ("lseq", "<i4"), ("nref", "<i4"), ("npos", "<i4"), ("tlen", "<i4"),
                   ("name", "S17"), ("cigar", "<u4"), ("seq", f"S{(read_len + 1) // 2}"), ("qual", f"S{read_len}")])
    assert dt.itemsize == 36 + 17 + 4 + (read_len + 1) // 2 + read_len
    return dt


def write_synthetic_bam(path: str, ref_names, ref_len, records, read_len: int = 100,
                        hd: str = "@HD\tVN:1.6\tSO:unsorted\tGO:query", threads: int = 32, piece: int = 10_000_000) -> dict:
    """Returns {"records", "raw_bytes", "compressed_bytes", "seconds"}."""
    t0 = time.time()
    n = len(records)
    dt = _record_dtype(read_len)
    text = (("" if not hd else hd + "\n") + "".join(f"@SQ\tSN:{nm}\tLN:{int(l)}\n" for nm, l in zip(ref_names, ref_len))).encode()
    head = bytearray(b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(ref_names)))
    for nm, l in zip(ref_names, ref_len):
        b = nm.encode() + b"\0"
        head += struct.pack("<i", len(b)) + b + struct.pack("<i", int(l))
    raw_bytes = len(head)

    def piece_bytes(lo):
        hi = min(n, lo + piece)
        m = hi - lo
        body = np.zeros(m, dtype=dt)
        body["bs"] = dt.itemsize - 4
        body["ref"] = records.ref_id[lo:hi]
        body["pos"] = records.begin_pos[lo:hi]
        body["lname"] = 17
        body["mapq"] = 255
        body["bin"] = 4680
        body["ncig"] = 1
        body["flag"] = records.flag[lo:hi]
        body["lseq"] = read_len
        body["nref"] = -1
        body["npos"] = -1
        kb = records.read_key[lo:hi].astype(">u8").view("u1").reshape(m, 8)
        names = np.empty((m, 17), dtype="u1")
        names[:, 0:16:2] = _HEX[kb >> 4]
        names[:, 1:16:2] = _HEX[kb & 15]
        names[:, 16] = 0
        body["name"] = names.view("S17").reshape(m)
        body["cigar"] = read_len << 4
        body["seq"] = b"\x11" * ((read_len + 1) // 2)
        body["qual"] = b"\x28" * read_len
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
            for blk in ex.map(_bgzf, chunks, chunksize=64):
                f.write(blk)
            carry = raw[whole:]
            del raw, chunks
        f.write(_bgzf(b"", 6))   # the end-of-file marker block
    return {"records": n, "raw_bytes": raw_bytes, "compressed_bytes": os.path.getsize(path), "seconds": time.time() - t0}
