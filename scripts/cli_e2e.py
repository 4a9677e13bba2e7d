"""End-to-end timing of the `slimm` command line on a synthetic BAM (decode on the host cores + GPU path + profile).
    python scripts/cli_e2e.py [n_records] [bam|sam]
The BAM is built with numpy (fixed-size records) and compressed block by block with zlib.
"""
import os, struct, subprocess, sys, tempfile, time, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concurrent.futures import ThreadPoolExecutor
from slimm_amd.synth import CONFIGS, make_workload
from tests.bam_io import sam_header, write_sldb, _bgzf_block

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
w = make_workload(CONFIGS["config2"], seed=1, n_records=n)
rec = w.records
L = 100
dt = np.dtype([("bs", "<i4"), ("ref", "<i4"), ("pos", "<i4"), ("lname", "u1"), ("mapq", "u1"), ("bin", "<u2"), ("ncig", "<u2"),
               ("flag", "<u2"), ("lseq", "<i4"), ("nref", "<i4"), ("npos", "<i4"), ("tlen", "<i4"), ("name", "S17"),
               ("cigar", "<u4"), ("seq", f"S{L // 2}"), ("qual", f"S{L}")])
assert dt.itemsize == 36 + 17 + 4 + 50 + 100
body = np.zeros(n, dtype=dt)
body["bs"] = dt.itemsize - 4
body["ref"] = rec.ref_id; body["pos"] = rec.begin_pos; body["lname"] = 17; body["mapq"] = 255; body["bin"] = 4680
body["ncig"] = 1; body["flag"] = rec.flag; body["lseq"] = L; body["nref"] = -1; body["npos"] = -1
hexd = np.frombuffer(b"0123456789abcdef", dtype="u1")
kb = rec.read_key.astype(">u8").view("u1").reshape(n, 8)
names = np.empty((n, 17), dtype="u1"); names[:, 0:16:2] = hexd[kb >> 4]; names[:, 1:16:2] = hexd[kb & 15]; names[:, 16] = 0
body["name"] = names.view("S17").reshape(n)
body["cigar"] = (L << 4)
body["seq"] = b"\x11" * (L // 2); body["qual"] = b"\x28" * L
tmp = tempfile.mkdtemp(prefix="slimm_e2e_")
text = sam_header(w.ref_names, w.ref_len, "@HD\tVN:1.6\tSO:unsorted\tGO:query").encode()
head = bytearray(b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(w.ref_names)))
for nm, l in zip(w.ref_names, w.ref_len):
    b = nm.encode() + b"\0"; head += struct.pack("<i", len(b)) + b + struct.pack("<i", int(l))
raw = bytes(head) + body.tobytes()
t0 = time.time()
chunks = [raw[s:s + 0xff00] for s in range(0, len(raw), 0xff00)]
def blk(c):
    co = zlib.compressobj(1, zlib.DEFLATED, -15); comp = co.compress(c) + co.flush()
    return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, len(comp) + 25) + comp
            + struct.pack("<II", zlib.crc32(c) & 0xffffffff, len(c)))
with ThreadPoolExecutor(32) as ex:
    out = list(ex.map(blk, chunks, chunksize=64))
bam = os.path.join(tmp, "sample.bam")
with open(bam, "wb") as f:
    for b in out: f.write(b)
    f.write(_bgzf_block(b""))
print(f"BAM: {n} records, {len(raw)/1e6:.0f} MB raw, {os.path.getsize(bam)/1e6:.0f} MB compressed, built in {time.time()-t0:.1f}s", flush=True)
db = os.path.join(tmp, "db.sldb"); write_sldb(db, w.taxonomy)
cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "slimm_amd", "slimm")
os.makedirs(os.path.join(tmp, "out"))
for threads in (0, 16, 32, 64):   # 0 = the command's own choice (logical CPUs or twice the cgroup quota, at most 64)
    env = dict(os.environ, SLIMM_CLI_TRACE="1")
    if threads:
        env["SLIMM_DECODE_THREADS"] = str(threads)
    t0 = time.time()
    r = subprocess.run([cli, "-w", "1000", "-o", os.path.join(tmp, "out") + "/", db, bam], capture_output=True, text=True, env=env)
    dt_ = time.time() - t0
    assert r.returncode == 0, r.stderr[-1000:]
    print("\n".join(l[l.index("[trace]"):] for l in r.stderr.splitlines() if "[trace]" in l))
    print(f"slimm DB BAM with {threads:2d} decode threads: {dt_:.2f} s wall -> {n/dt_/1e6:.2f} M records/s (process start, two passes over the "
          f"file, GPU path, profile)", flush=True)
print(open(os.path.join(tmp, "out", "sample_profile.tsv")).read()[:300])
