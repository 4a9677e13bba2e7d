"""How the `slimm` command should leave: wall time of the same run under the exit modes (GPU box)."""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slimm_amd.synth import CONFIGS, make_workload
from slimm_amd.synth_bam import write_synthetic_bam
from tests.bam_io import write_sldb
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
cfg = CONFIGS["config3"]
w = make_workload(cfg, seed=1, n_records=n)
tmp = tempfile.mkdtemp(prefix="slimm_exit_")
bam = os.path.join(tmp, "sample.bam")
write_synthetic_bam(bam, w.ref_names, w.ref_len, w.records, read_len=cfg.read_len)
db = os.path.join(tmp, "db.sldb"); write_sldb(db, w.taxonomy)
cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "slimm_amd", "slimm")
os.makedirs(os.path.join(tmp, "out"))
for rep in range(2):
    for name, extra in (("orderly", {}), ("fast-exit", {"SLIMM_FAST_EXIT": "1"}),
                        ("host-decode", {"SLIMM_CLI_HOST_DECODE": "1"})):
        env = dict(os.environ, SLIMM_CLI_TRACE="1", **extra)
        t0 = time.time()
        r = subprocess.run([cli, "-w", "1000", "-o", os.path.join(tmp, "out") + "/", db, bam], capture_output=True, text=True, env=env)
        t1 = time.time()
        dt = t1 - t0
        end = [l for l in r.stderr.splitlines() if "reached its end" in l]
        ent = [float(l.split(" at ")[1].split()[0]) for l in r.stderr.splitlines() if "main() entered at" in l]
        lv = [float(l.split(" at ")[1].split()[0]) for l in r.stderr.splitlines() if "leaving at" in l]
        extra_s = f"spawn -> main {1e3 * (ent[0] - t0):.0f} ms, main {1e3 * (lv[0] - ent[0]):.0f} ms, exit -> parent {1e3 * (t1 - lv[0]):.0f} ms" if ent and lv else ""
        print(f"{name:12s} {dt:.3f} s wall -> {n / dt / 1e6:.1f} M records/s; {end[-1][end[-1].index('main'):] if end else ''}; {extra_s}", flush=True)
os.unlink(bam)
