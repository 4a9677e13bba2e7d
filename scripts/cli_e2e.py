"""End-to-end timing of the `slimm` command line on a synthetic BAM (decode on the host cores + GPU path + profile).
    python scripts/cli_e2e.py [n_records] [config]        (default: 4 000 000 records of config2)
The BAM comes from slimm_amd/synth_bam.py (fixed-size records, BGZF blocks compressed with zlib level 1 on a thread pool).
"""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slimm_amd.synth import CONFIGS, make_workload
from slimm_amd.synth_bam import write_synthetic_bam
from tests.bam_io import write_sldb

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
cfg = CONFIGS[sys.argv[2] if len(sys.argv) > 2 else "config2"]
w = make_workload(cfg, seed=1, n_records=n)
tmp = tempfile.mkdtemp(prefix="slimm_e2e_")
bam = os.path.join(tmp, "sample.bam")
info = write_synthetic_bam(bam, w.ref_names, w.ref_len, w.records, read_len=cfg.read_len)
print(f"BAM: {n} records, {info['raw_bytes']/1e6:.0f} MB raw, {info['compressed_bytes']/1e6:.0f} MB compressed, built in "
      f"{info['seconds']:.1f}s", flush=True)
db = os.path.join(tmp, "db.sldb"); write_sldb(db, w.taxonomy)
cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "slimm_amd", "slimm")
os.makedirs(os.path.join(tmp, "out"))
for threads in (0, 16, 32, 64):   # 0 = the command's own choice (logical CPUs or twice the cgroup quota, at most 64)
    env = dict(os.environ, SLIMM_TRACE="cli")
    flags = ["--decode-threads", str(threads)] if threads else []
    t0 = time.time()
    r = subprocess.run([cli] + flags + ["-w", "1000", "-o", os.path.join(tmp, "out") + "/", db, bam], capture_output=True, text=True, env=env)
    dt_ = time.time() - t0
    assert r.returncode == 0, r.stderr[-1000:]
    print("\n".join(l[l.index("[trace]"):] for l in r.stderr.splitlines() if "[trace]" in l))
    print(f"slimm DB BAM with {threads:2d} decode threads: {dt_:.2f} s wall -> {n/dt_/1e6:.2f} M records/s (process start, two passes over the "
          f"file, GPU path, profile)", flush=True)
print(open(os.path.join(tmp, "out", "sample_profile.tsv")).read()[:300])
os.unlink(bam)
