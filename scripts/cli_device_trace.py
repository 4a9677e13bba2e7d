"""Full stage trace of the `slimm` command on the 100 M-record BAM for a few (period, device window MB) settings (GPU box)."""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slimm_amd.synth import CONFIGS, make_workload
from slimm_amd.synth_bam import write_synthetic_bam
from tests.bam_io import write_sldb
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
settings = [tuple(int(v) for v in x.split(":")) for x in (sys.argv[2] if len(sys.argv) > 2 else "0:1900,6:1900,4:950,3:950").split(",")]
cfg = CONFIGS["config3"]
w = make_workload(cfg, seed=1, n_records=n)
tmp = tempfile.mkdtemp(prefix="slimm_dt_")
bam = os.path.join(tmp, "sample.bam")
write_synthetic_bam(bam, w.ref_names, w.ref_len, w.records, read_len=cfg.read_len)
db = os.path.join(tmp, "db.sldb"); write_sldb(db, w.taxonomy)
cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "slimm_amd", "slimm")
os.makedirs(os.path.join(tmp, "out"))
for rep in range(2):
    for period, mb in settings:
        env = dict(os.environ, SLIMM_CLI_TRACE="1", SLIMM_CLI_DEVICE_INFLATE=str(period), SLIMM_CLI_DEVICE_WINDOW_MB=str(mb))
        t0 = time.time()
        r = subprocess.run([cli, "-w", "1000", "-o", os.path.join(tmp, "out") + "/", db, bam], capture_output=True, text=True, env=env)
        t1 = time.time()
        ent = [float(l.split(" at ")[1].split()[0]) for l in r.stderr.splitlines() if "main() entered at" in l]
        lv = [float(l.split(" at ")[1].split()[0]) for l in r.stderr.splitlines() if "leaving at" in l]
        print(f"== period {period}, device window {mb} MB: {t1 - t0:.3f} s wall -> {n / (t1 - t0) / 1e6:.1f} M records/s; main {1e3 * (lv[0] - ent[0]):.0f} ms, exit -> parent {1e3 * (t1 - lv[0]):.0f} ms")
        if rep == 1:
            for l in r.stderr.splitlines():
                if l.startswith("[trace]") and "entered" not in l and "leaving" not in l:
                    print("   ", l[8:200])
os.unlink(bam)
