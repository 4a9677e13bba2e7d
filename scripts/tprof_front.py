"""Cycle split of k_front (staging / the whole slot), from a build with the counters compiled in:
    scripts/build_variant.sh fprof '1s/^/#define EXP 11\\n/'
    SLIMM_HIP_LIB=build/var/fprof/libslimm_hip.so python scripts/tprof_front.py [config]
Prints the mean cycles a wave spends from a slot's start to the end of its staging, and to the slot's end (four-array
records resident on the device)."""
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
s = Slimm.for_workload(w, device=0)
lib = ctypes.CDLL(os.environ["SLIMM_HIP_LIB"])
n = 4 * 8192
buf = (ctypes.c_ulonglong * n)()
for it in range(3):
    s.reset(); s.reset_cutoffs(); s.set_records_device(key, ref, pos, flag)
    lib.slimm_debug_prof_front(buf, n, 1)
    s.analyze_alignments()
    torch.cuda.synchronize()
    lib.slimm_debug_prof_front(buf, n, 0)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).astype(np.float64)
    a = a[a[:, 2] > 0]
    slots = a[:, 2].sum()
    print(f"run {it}: {int(slots)} slots; staging {a[:, 0].sum() / slots:.0f} cycles per slot, whole slot {a[:, 1].sum() / slots:.0f}, of it runs of 64 records or more {a[:, 3].sum() / slots:.0f}", flush=True)
