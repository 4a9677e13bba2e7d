"""The `slimm` command and the device inflate on a BAM that compresses like a BAM (slimm_amd/synth_bam.py, realistic=True:
random bases, binned qualities with runs, instrument-style names; libdeflate level 6) beside the easy 17-fold file.
    python scripts/realistic_cli.py [records] [easy|realistic|both] [inflate_blocks]
Prints: the file's compression ratio, `slimm DB IN.bam` process start to profile written with its stage trace (default
settings, and with every window inflated on the host: --device-inflate 0), the device inflate's rate on the file's
first `inflate_blocks` BGZF blocks (slimm_bgzf_inflate: kernel time from HIP events), and the host's inflate rate on the same
blocks with the cores this process may use (zlib in threads)."""
import ctypes as C, os, subprocess, sys, tempfile, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from slimm_amd import capi
from slimm_amd.synth import CONFIGS, make_workload
from slimm_amd.synth_bam import write_synthetic_bam
from tests.bam_io import write_sldb

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
which = sys.argv[2] if len(sys.argv) > 2 else "both"
n_blocks = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
w = make_workload(CONFIGS["config3"], seed=1, n_records=n)
tmp = tempfile.mkdtemp(prefix="slimm_real_")
db = os.path.join(tmp, "db.sldb"); write_sldb(db, w.taxonomy)
cli = os.path.join(ROOT, "slimm_amd", "slimm")
os.makedirs(os.path.join(tmp, "out"))
L = capi.lib()
L.slimm_bgzf_inflate.restype = C.c_int
for kind in (("easy", "realistic") if which == "both" else (which,)):
    bam = os.path.join(tmp, kind + ".bam")
    info = write_synthetic_bam(bam, w.ref_names, w.ref_len, w.records, read_len=100, realistic=(kind == "realistic"))
    print(f"== {kind}: {n} records, {info['raw_bytes'] / 1e9:.2f} GB of BAM in {info['compressed_bytes'] / 1e9:.2f} GB = "
          f"{info['raw_bytes'] / info['compressed_bytes']:.2f} x ({info['deflate']}), built in {info['seconds']:.0f} s", flush=True)
    for label, flags in (("default", []), ("host inflate only", ["--device-inflate", "0"])):
        best, tr = None, ""
        for _ in range(2):
            t0 = time.time()
            r = subprocess.run([cli] + flags + ["-w", "1000", "-o", os.path.join(tmp, "out") + "/", db, bam], capture_output=True, text=True,
                               env=dict(os.environ, SLIMM_TRACE="cli"))
            dt_ = time.time() - t0
            if r.returncode != 0:
                print(f"   {label}: FAILED {r.stderr[-300:]}")
                break
            if best is None or dt_ < best:
                best, tr = dt_, "\n".join("      " + l[l.index("[trace]"):] for l in r.stderr.splitlines() if "[trace]" in l)
        if best is not None:
            print(f"   slimm DB {kind}.bam [{label}]: {best:.3f} s = {n / best / 1e6:.1f} M records/s\n{tr}", flush=True)
    # the inflate by itself on the first n_blocks blocks
    blob = np.fromfile(bam, dtype=np.uint8, count=min(os.path.getsize(bam), n_blocks * 66000))
    cuts, p = [], 0
    while p + 18 <= blob.size and len(cuts) < n_blocks:
        bs = int(blob[p + 16]) + (int(blob[p + 17]) << 8) + 1
        if p + bs > blob.size:
            break
        cuts.append((p, bs))
        p += bs
    part = blob[:p]
    isz = sum(int(np.frombuffer(part[a + b - 4:a + b].tobytes(), dtype="<u4")[0]) for a, b in cuts)
    out = np.zeros(isz + 64, dtype=np.uint8)
    ms_best = None
    for it in range(3):
        nb, ms, err = C.c_uint64(), C.c_double(), C.create_string_buffer(256)
        rc = L.slimm_bgzf_inflate(0, part.ctypes.data_as(C.c_void_p), C.c_uint64(part.size), out.ctypes.data_as(C.c_void_p),
                                  C.c_uint64(out.size), C.byref(nb), C.byref(ms), err, C.c_uint64(256))
        assert rc == 0, err.value
        ms_best = ms.value if ms_best is None else min(ms_best, ms.value)
    def host(lo_hi):
        tot = 0
        for a, b in cuts[lo_hi[0]:lo_hi[1]]:
            tot += len(zlib.decompress(part[a + 18:a + b - 8].tobytes(), -15))
        return tot
    th = max(1, len(os.sched_getaffinity(0)))
    per = (len(cuts) + th - 1) // th
    t0 = time.time()
    with ThreadPoolExecutor(th) as ex:
        tot = sum(ex.map(host, [(i, min(len(cuts), i + per)) for i in range(0, len(cuts), per)]))
    t_host = time.time() - t0
    assert tot == isz == nb.value
    chk = zlib.decompress(part[cuts[-1][0] + 18:cuts[-1][0] + cuts[-1][1] - 8].tobytes(), -15)
    assert bytes(out[isz - len(chk):isz]) == chk
    print(f"   device inflate of {len(cuts)} blocks: {part.size / 1e6:.0f} MB -> {isz / 1e6:.0f} MB, kernel {ms_best:.2f} ms = {isz / ms_best / 1e6:.1f} GB/s "
          f"inflated ({part.size / ms_best / 1e6:.1f} GB/s compressed); zlib on {th} threads {t_host * 1e3:.0f} ms = {isz / t_host / 1e9:.2f} GB/s", flush=True)
    os.unlink(bam)
