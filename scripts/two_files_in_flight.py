"""Would two files IN FLIGHT on one GPU (two host threads, each driving a context of its own on its own stream) finish more
files per second than files back to back?  The front end waits for the scalar unit, the bucketing for the LDS, the
histograms for memory -- kernels of two files might share a CU better than one file's do.  Measurement only:
    python scripts/two_files_in_flight.py [config3|config4] [files]
"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from slimm_amd.profiler import Slimm
from slimm_amd.synth import CONFIGS, make_workload
from slimm_amd.distributed import FilesBackToBack, sharded_profile_begin, sharded_profile_end

name = sys.argv[1] if len(sys.argv) > 1 else "config3"
files = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cfg = CONFIGS[name]
w = make_workload(cfg, seed=1, n_records=int(os.environ.get("RECORDS", "0")) or None)
dev = torch.device("cuda:0")
key = torch.from_numpy(w.records.read_key.view(np.int64)).to(dev); ref = torch.from_numpy(w.records.ref_id).to(dev)
pos = torch.from_numpy(w.records.begin_pos).to(dev); flag = torch.from_numpy(w.records.flag.view(np.int16)).to(dev)
n = len(w.records)

def engines(k):
    return [Slimm.for_workload(w, device=0) for _ in range(k)]

def give(e):
    e.set_records_device(key, ref, pos, flag)

def run(fb, k, path):
    if isinstance(fb, FilesBackToBack):
        for _ in range(k):
            fb.step()
        fb.flush()
    else:   # one engine: a file after the other, nothing beside anything
        for _ in range(k):
            fb.reset(); fb.reset_cutoffs(); give(fb)
            if sharded_profile_begin(fb, dev):
                sharded_profile_end(fb, path)

def timed(nthreads, per_thread_engines):
    fbs = [FilesBackToBack(engines(2), give, device=dev, path=f"/tmp/tfif_{i}.tsv") if per_thread_engines == 2 else engines(1)[0]
           for i in range(nthreads)]
    for fb in fbs: run(fb, 3, None)      # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ts = [threading.Thread(target=run, args=(fb, files // nthreads, f"/tmp/tfif_t{i}.tsv")) for i, fb in enumerate(fbs)]
    for t in ts: t.start()
    for t in ts: t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dt / files * 1e3

for nt, pe in ((1, 1), (1, 2), (2, 1), (2, 2)):
    ms = timed(nt, pe)
    print(f"{name}: {nt} thread(s) x {pe} engine(s): {ms:.3f} ms per file = {n / ms / 1e3:.0f} M records/s", flush=True)
