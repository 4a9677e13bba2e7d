"""`slimm DB IN.sam` on a synthetic SAM file (slimm_amd/synth_bam.py: write_synthetic_sam), the lines found and parsed on the
device (slimm_push_sam_bytes) against the host decoder (--host-decode).  python scripts/sam_cli.py [records]"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slimm_amd.synth import CONFIGS, make_workload
from slimm_amd.synth_bam import write_synthetic_sam
from tests.bam_io import write_sldb

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
w = make_workload(CONFIGS["config3"], seed=1, n_records=n)
tmp = tempfile.mkdtemp(prefix="slimm_sam_")
db = os.path.join(tmp, "db.sldb"); write_sldb(db, w.taxonomy)
sam = os.path.join(tmp, "sample.sam")
info = write_synthetic_sam(sam, w.ref_names, w.ref_len, w.records)
print(f"SAM: {n} records, {info['bytes'] / 1e9:.2f} GB, built in {info['seconds']:.0f} s", flush=True)
cli = os.path.join(ROOT, "slimm_amd", "slimm")
outs = {}
for label, flags in (("device decode", []), ("host decode", ["--host-decode"])):
    os.makedirs(os.path.join(tmp, label.split()[0]), exist_ok=True)
    best, tr = None, ""
    for _ in range(2 if label.startswith("device") else 1):
        t0 = time.time()
        r = subprocess.run([cli] + flags + ["-w", "1000", "-o", os.path.join(tmp, label.split()[0]) + "/", db, sam], capture_output=True, text=True,
                           env=dict(os.environ, SLIMM_TRACE="cli"))
        dt_ = time.time() - t0
        assert r.returncode == 0, r.stderr[-1500:]
        if best is None or dt_ < best:
            best, tr = dt_, "\n".join("      " + l[l.index("[trace]"):][:260] for l in r.stderr.splitlines() if "[trace]" in l)
    outs[label] = open(os.path.join(tmp, label.split()[0], "sample_profile.tsv")).read()
    print(f"   slimm DB sample.sam [{label}]: {best:.3f} s = {n / best / 1e6:.1f} M records/s ({info['bytes'] / best / 1e9:.1f} GB/s of text)\n{tr}", flush=True)
print("same profile:", outs["device decode"] == outs["host decode"])
os.unlink(sam)
