"""Cycle split of the bucket scatter (EXP 8: per wave) or of k_tile_hist (EXP 9: thread 0 of every workgroup), from a
build with the counters compiled in:
    scripts/build_variant.sh tprof '1s/^/#define EXP 8\\n/'
    SLIMM_HIP_LIB=build/var/tprof/libslimm_hip.so TPROF_EXP=8 python scripts/tprof_tiles.py [config]
Prints the mean s_memtime ticks per phase of the kernel, for the launches of phase A and phase B separately.
"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slimm_amd.profiler import Slimm
from slimm_amd.synth import CONFIGS, make_workload

cfg = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "config2"]
w = make_workload(cfg, seed=1)
dev = torch.device("cuda:0")
key = torch.from_numpy(w.records.read_key.view(np.int64)).to(dev); ref = torch.from_numpy(w.records.ref_id).to(dev)
pos = torch.from_numpy(w.records.begin_pos).to(dev); flag = torch.from_numpy(w.records.flag.view(np.int16)).to(dev)
s = Slimm.for_workload(w, device=0)
lib = ctypes.CDLL(os.environ["SLIMM_HIP_LIB"])
n = 8 * 4096
buf = (ctypes.c_ulonglong * n)()
names = {"8": ["prologue", "clear", "lds count", "cursor atomics", "scan", "stage", "store", "kernel"],
         "9": ["-", "item+clear", "count", "barrier", "stats", "store", "-", "kernel"]}[os.environ.get("TPROF_EXP", "8")]
for it in range(4):
    s.reset(); s.reset_cutoffs(); s.set_records_device(key, ref, pos, flag)
    for phase in ("A", "B"):
        lib.slimm_debug_prof_tiles(buf, n, 1)
        if phase == "A":
            s.analyze_alignments()
        else:
            s.finish_coverage(); s.filter_alignments()
        torch.cuda.synchronize()
        lib.slimm_debug_prof_tiles(buf, n, 0)
        a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8).astype(np.float64)
        a = a[a[:, 7] > 0]
        if len(a):
            print(f"run {it} phase {phase}: {len(a)} rows;", ", ".join(f"{nm} {a[:, i].mean():.0f}" for i, nm in enumerate(names)),
                  f"; max kernel {a[:, 7].max():.0f}", flush=True)
