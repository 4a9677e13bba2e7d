"""The `slimm` command on a 100 M-record BAM with one window in P of those read in place inflated on the DEVICE
(SLIMM_CLI_DEVICE_INFLATE = P; 0 = none; a device window is ten host windows large), for the name-grouped file and for the
same records in no particular order (GPU box).
    python scripts/cli_inflate_share.py [records] [periods, e.g. 0,16,12,9,6]"""
import hashlib, os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slimm_amd.synth import CONFIGS, make_workload
from slimm_amd.synth_bam import write_synthetic_bam
from tests.bam_io import write_sldb
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
shares = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,16,12,9,6").split(",")]
cfg = CONFIGS["config3"]
w = make_workload(cfg, seed=1, n_records=n)
tmp = tempfile.mkdtemp(prefix="slimm_share_")
db = os.path.join(tmp, "db.sldb"); write_sldb(db, w.taxonomy)
cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "slimm_amd", "slimm")
os.makedirs(os.path.join(tmp, "out"))
for what in ("grouped", "unsorted"):
    bam = os.path.join(tmp, what + ".bam")
    if what == "grouped":
        write_synthetic_bam(bam, w.ref_names, w.ref_len, w.records, read_len=cfg.read_len)
    else:
        import torch
        dev = torch.device("cuda:0")
        k = torch.from_numpy(w.records.read_key.view(np.int64)).to(dev)
        m = k.numel()
        g = torch.Generator(device=dev); g.manual_seed(7200)
        pos = torch.randperm(m, device=dev, generator=g)
        a = torch.sort(k, stable=True).indices
        p1 = torch.argsort(pos)
        b = p1[torch.sort(k[p1], stable=True).indices]
        newpos = torch.empty(m, dtype=torch.int64, device=dev); newpos[a] = pos[b]
        inv = torch.empty(m, dtype=torch.int64, device=dev); inv[newpos] = torch.arange(m, device=dev)
        recu = w.records.take(inv.cpu().numpy())
        del k, pos, a, p1, b, newpos, inv
        torch.cuda.empty_cache()
        write_synthetic_bam(bam, w.ref_names, w.ref_len, recu, read_len=cfg.read_len, hd="@HD\tVN:1.6\tSO:unsorted")
        del recu
    print(f"== {what}: {os.path.getsize(bam) / 1e6:.0f} MB compressed", flush=True)
    sha = set()
    for rep in range(2):
        for t in shares:
            env = dict(os.environ, SLIMM_CLI_TRACE="1", SLIMM_CLI_DEVICE_INFLATE=str(t))
            t0 = time.time()
            r = subprocess.run([cli, "-w", "1000", "-o", os.path.join(tmp, "out") + "/", db, bam], capture_output=True, text=True, env=env)
            dt = time.time() - t0
            assert r.returncode == 0, r.stderr[-1500:]
            dd = [l for l in r.stderr.splitlines() if "device decode" in l]
            prof = [f for f in os.listdir(os.path.join(tmp, "out")) if f.endswith("_profile.tsv")]
            sha.add(hashlib.sha1(open(os.path.join(tmp, "out", prof[0]), "rb").read()).hexdigest()[:10])
            print(f"one window in {t:2d} on the device: {dt:.3f} s wall -> {n / dt / 1e6:.1f} M records/s; {dd[-1][dd[-1].index('inflate'):] if dd else ''}", flush=True)
    print("profiles:", sha, flush=True)
    os.unlink(bam)
