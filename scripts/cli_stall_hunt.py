"""Back-to-back runs of `slimm DB IN.bam` with the pipeline's event trace: where an occasional run of 2 s instead of 0.7 s loses its time.
    python scripts/cli_stall_hunt.py [runs] [flags of the command ...]        (SLIMM_STALL_FULL=file: run 0's whole stderr goes there)"""
import os, subprocess, sys, tempfile, time, shutil
ROOT = os.getcwd()
sys.path.insert(0, ROOT)
from slimm_amd.synth import CONFIGS, make_workload
from slimm_amd.synth_bam import write_synthetic_bam
from tests.bam_io import write_sldb
n = 100_000_000
w = make_workload(CONFIGS["config3"], seed=1, n_records=n)
tmp = tempfile.mkdtemp(prefix="slimm_stall_", dir="/dev/shm")
db = os.path.join(tmp, "db.sldb"); write_sldb(db, w.taxonomy)
bam = os.path.join(tmp, "r.bam")
write_synthetic_bam(bam, w.ref_names, w.ref_len, w.records, read_len=100, realistic=True)
os.makedirs(os.path.join(tmp, "out"))
first_prof = None
# SLIMM_STALL_LIBS=a.so,b.so: the runs alternate between these builds of the library (one file, one box: a fair comparison);
# SLIMM_STALL_PARENT_GPU=1: this process holds a GPU context of its own meanwhile (as bench.py does around its command legs)
libs = [l for l in os.environ.get("SLIMM_STALL_LIBS", "").split(",") if l]
# SLIMM_STALL_FLAGS="--read-buffers 4|--read-buffers 8": the runs alternate between these sets of flags of the command
flag_sets = [f.split() for f in os.environ.get("SLIMM_STALL_FLAGS", "").split("|") if f.strip()]
if os.environ.get("SLIMM_STALL_PARENT_GPU") == "1":
    import torch
    hold = torch.zeros(1 << 28, device="cuda:0")
    torch.cuda.synchronize()
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 14):
    env = dict(os.environ, SLIMM_TRACE="cli,push,host")
    if libs:
        env["SLIMM_HIP_LIB"] = os.path.abspath(libs[rep % len(libs)])
    t0 = time.time()
    extra = flag_sets[rep % len(flag_sets)] if flag_sets else []
    r = subprocess.run([os.path.join(ROOT, "slimm_amd", "slimm")] + sys.argv[2:] + extra + ["-w", "1000", "-o", os.path.join(tmp, "out") + "/", db, bam], capture_output=True, text=True,
                       env=env)
    dt = time.time() - t0
    prof = open(os.path.join(tmp, "out", "r_profile.tsv")).read() if r.returncode == 0 else None
    first_prof = prof if rep == 0 else first_prof
    push = [l.split("slimm_push_bam_bytes")[1].split("ms")[0].strip() for l in r.stderr.splitlines() if "slimm_push_bam_bytes" in l]
    # how long the even and the odd windows made the pusher wait when it finished them
    waits, t_fin = [0.0, 0.0], None
    for ln in r.stderr.splitlines():
        if "] finishing window" in ln:
            t_fin = float(ln.split("[push")[1].split("]")[0])
        elif "] finished window" in ln and t_fin is not None:
            w = int(ln.split("finished window")[1].split(":")[0])
            waits[w & 1] += float(ln.split("[push")[1].split("]")[0]) - t_fin
    print(f"finishing even/odd windows {waits[0]:.0f}/{waits[1]:.0f} ms ", end="")
    print((os.path.basename(os.path.dirname(env["SLIMM_HIP_LIB"])) + " " if libs else "") + (" ".join(extra) + " " if extra else "") + f"push {push[0] if push else '?'} ms ", end="")
    print(f"run {rep}: wall {dt:.3f} s = {n / dt / 1e6:.0f} M records/s, rc {r.returncode}, profile {'the same' if prof == first_prof and prof else 'DIFFERS / MISSING'}"
          + ("" if r.returncode == 0 else " -- " + r.stderr[-300:].replace("\n", " | ")), flush=True)
    if rep == 0 and os.environ.get("SLIMM_STALL_FULL"):
        open(os.environ["SLIMM_STALL_FULL"], "w").write(r.stderr)
    if rep == 0:
        print("\n".join("    " + l.strip()[:200] for l in r.stderr.splitlines() if "page-locked" in l or "reserved" in l or "planned" in l or "grows" in l))
    if dt > 1.2 or rep == 0:
        # the events with the gaps in front of them
        last = None
        for ln in r.stderr.splitlines():
            if "[push" in ln:
                try:
                    t = float(ln.split("[push")[1].split("]")[0])
                except ValueError:
                    continue
                if last is not None and t - last > 40:
                    print(f"    GAP {t - last:.0f} ms before: {ln.strip()[:160]}")
                last = t
            elif "[trace]" in ln or "[host]" in ln:
                print("   ", ln.strip()[:200])
shutil.rmtree(tmp)
