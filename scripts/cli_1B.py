"""`slimm DB IN.bam` on the 1 B-record BAM north_star names (BASELINE.json configs[3]: 1 B records, 20 k references, mean 8
hits per read) -- VERDICT round 5, item 3; replaces seqan::BamFileIn + the record loop of the reference
(src/slimm.hpp:946-968, src/misc.hpp:498-522).
    python scripts/cli_1B.py [records] [dir | -] [one]          (`one`: the single-device runs only -- bench.py's leg)
Builds the file once with realistic content (slimm_amd/synth_bam.py, realistic=True: ~78 bytes per record compressed, 3.0 x),
the stream of bench.py's headline (chunk c = stream_chunk(config4, seed 1, c)), generated chunk by chunk so that the records
never stand in memory together; runs the command on one device and with `--devices 0,0` (the dealing path), and prints M
records/s, the stage trace, the library's peak device memory and the process's peak resident set.  The file goes where there
is room (the largest of /tmp, /dev/shm, $TMPDIR, or `dir`); when there is not enough for all records, the record count is cut
to what fits and the reason printed."""
import os, shutil, subprocess, sys, tempfile, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from slimm_amd.synth import CONFIGS, stream_chunk
from slimm_amd.synth_bam import write_synthetic_bam
from tests.bam_io import write_sldb

want = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
CHUNK = 10_000_000
BYTES_PER_RECORD = 79.5        # measured on the 100 M-record file: 7.80 GB

only_one = len(sys.argv) > 3 and sys.argv[3] == "one"
dirs = [sys.argv[2]] if len(sys.argv) > 2 and sys.argv[2] != "-" else [d for d in ("/tmp", "/dev/shm", os.environ.get("TMPDIR", "")) if d and os.path.isdir(d)]
free = {d: shutil.disk_usage(d).free for d in dirs}
print("free space:", {d: f"{v / 1e9:.1f} GB" for d, v in free.items()}, flush=True)
where = max(free, key=free.get)
fits = int(free[where] * 0.97 / BYTES_PER_RECORD) // CHUNK * CHUNK
n = min(want, fits)
if n < want:
    print(f"NOT ENOUGH ROOM for {want} records ({want * BYTES_PER_RECORD / 1e9:.0f} GB): {where} has {free[where] / 1e9:.1f} GB free -> {n} records", flush=True)
if n < CHUNK:
    print("skipped: no scratch space for even one chunk")
    sys.exit(0)
cfg = CONFIGS["config4"]


class Lazy:
    """records.<field>[lo:hi] of the stream, chunk by chunk (a builder thread asks for the four fields of one piece)."""
    def __init__(self):
        self.lock = threading.Lock()
        self.cache = {}     # chunk -> Records (the last few)

    def chunk(self, c):
        with self.lock:
            r = self.cache.get(c)
        if r is None:
            r = stream_chunk(cfg, 1, c, n, CHUNK).records
            with self.lock:
                self.cache[c] = r
                for old in sorted(self.cache)[:-6]:
                    del self.cache[old]
        return r

    class Field:
        def __init__(self, lazy, name):
            self.lazy, self.name = lazy, name

        def __getitem__(self, s):
            assert s.start % CHUNK == 0 and s.stop - s.start <= CHUNK
            a = getattr(self.lazy.chunk(s.start // CHUNK), self.name)
            return a[: s.stop - s.start]

    def __len__(self):
        return n

    def __getattr__(self, name):
        if name in ("ref_id", "begin_pos", "flag", "read_key"):
            return Lazy.Field(self, name)
        raise AttributeError(name)


w0 = stream_chunk(cfg, 1, 0, n, CHUNK)
tmp = tempfile.mkdtemp(prefix="slimm_1B_", dir=where)
try:
    db = os.path.join(tmp, "db.sldb"); write_sldb(db, w0.taxonomy)
    bam = os.path.join(tmp, "stream.bam")
    info = write_synthetic_bam(bam, w0.ref_names, w0.ref_len, Lazy(), read_len=100, realistic=True, piece=CHUNK)
    print(f"== {n} records, {info['raw_bytes'] / 1e9:.1f} GB of BAM in {info['compressed_bytes'] / 1e9:.2f} GB = "
          f"{info['raw_bytes'] / info['compressed_bytes']:.2f} x ({info['deflate']}), built in {info['seconds']:.0f} s in {where}", flush=True)
    cli = os.path.join(ROOT, "slimm_amd", "slimm")
    out = os.path.join(tmp, "out"); os.makedirs(out)
    profiles = []
    for label, extra in ((("one device", []),) if only_one else (("one device", []), ("--devices 0,0", ["--devices", "0,0"]))):
        for rep in range(2):
            t0 = time.time()
            r = subprocess.run([cli] + extra + ["-w", "1000", "-o", out + "/", db, bam], capture_output=True, text=True,
                               env=dict(os.environ, SLIMM_TRACE="cli"))
            dt_ = time.time() - t0
            if r.returncode != 0:
                print(f"   {label}: FAILED rc {r.returncode}: {r.stderr[-600:]}", flush=True)
                break
            tr = "\n".join("      " + l[l.index("[trace]"):] for l in r.stderr.splitlines() if "[trace]" in l)
            print(f"   slimm DB stream.bam [{label}, run {rep + 1}]: {dt_:.3f} s = {n / dt_ / 1e6:.1f} M records/s\n{tr}", flush=True)
            profiles.append(open(os.path.join(out, "stream_profile.tsv")).read())
    print("   same profile from every run:", len(set(profiles)) == 1 and len(profiles) > 0, flush=True)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
