import os, subprocess, sys, tempfile, time
ROOT = os.getcwd()
sys.path.insert(0, ROOT)
from slimm_amd.synth import CONFIGS, make_workload
from slimm_amd.synth_bam import write_synthetic_bam
from tests.bam_io import write_sldb
n = 100_000_000
w = make_workload(CONFIGS["config3"], seed=1, n_records=n)
tmp = tempfile.mkdtemp(prefix="slimm_exit_", dir="/dev/shm")
db = os.path.join(tmp, "db.sldb"); write_sldb(db, w.taxonomy)
bam = os.path.join(tmp, "r.bam")
info = write_synthetic_bam(bam, w.ref_names, w.ref_len, w.records, read_len=100, realistic=True)
print("built", info["compressed_bytes"] / 1e9, "GB in", info["seconds"], flush=True)
os.makedirs(os.path.join(tmp, "out"))
for rep in range(4):
    for exe in ("slimm", "slimm_quick_exit"):
        t0 = time.time()
        r = subprocess.run([os.path.join(ROOT, "slimm_amd", exe), "-w", "1000", "-o", os.path.join(tmp, "out") + "/", db, bam], capture_output=True, text=True,
                           env=dict(os.environ, SLIMM_TRACE="cli"))
        dt = time.time() - t0
        end = [l for l in r.stderr.splitlines() if "reached its end" in l]
        cr = [l for l in r.stderr.splitlines() if "slimm_create" in l]
        print(f"{exe:18s} rc {r.returncode} wall {dt:.3f} s = {n / dt / 1e6:.1f} M records/s; {end[0][8:] if end else ''}; {cr[0][8:60] if cr else ''}", flush=True)
import shutil; shutil.rmtree(tmp)
