"""Cycle split of the any-order grouping's scatter passes (k_gb_scatter, staged form), from a build with the counters compiled in:
    scripts/build_variant.sh gprof '1s/^/#define EXP 10\\n/'
    SLIMM_HIP_LIB=build/var/gprof/libslimm_hip.so [SLIMM_FORCE=group_width=10] python scripts/tprof_group.py [config]
Prints the mean cycles thread 0 of a workgroup spent per phase of a round, summed over the rounds of the three passes."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slimm_amd.profiler import Slimm
from slimm_amd.synth import CONFIGS, make_workload

cfg = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "config3"]
w = make_workload(cfg, seed=1)
dev = torch.device("cuda:0")
key = torch.from_numpy(w.records.read_key.view(np.int64)).to(dev); ref = torch.from_numpy(w.records.ref_id).to(dev)
pos = torch.from_numpy(w.records.begin_pos).to(dev); flag = torch.from_numpy(w.records.flag.view(np.int16)).to(dev)
s = Slimm.for_workload(w, device=0, grouped=False)
lib = ctypes.CDLL(os.environ["SLIMM_HIP_LIB"])
n = 8 * 1024
buf = (ctypes.c_ulonglong * n)()
names = ["loads + first-pass arithmetic", "peers + ranks (per item)", "barrier", "wave prefix + scan", "placement",
         "write-out", "clear + barrier", "round"]
for it in range(3):
    s.reset(); s.reset_cutoffs(); s.set_records_device(key, ref, pos, flag)
    lib.slimm_debug_prof_group(buf, n, 1)
    s.analyze_alignments()
    torch.cuda.synchronize()
    lib.slimm_debug_prof_group(buf, n, 0)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8).astype(np.float64)
    a = a[a[:, 7] > 0]
    print(f"run {it}: {len(a)} workgroups;", ", ".join(f"{nm} {a[:, i].mean():.0f}" for i, nm in enumerate(names)), flush=True)
