# scratch: per-phase cycle totals of k_filter_lca16 (build with -DEXP=7; every 32nd workgroup reports)
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, ".")
from slimm_amd.profiler import Slimm
from slimm_amd.synth import CONFIGS, make_workload
from slimm_amd import capi
w = make_workload(CONFIGS["config2"], seed=1, sample_seed=1)
eng = Slimm.for_workload(w, device=0, grouped=True)
dev = torch.device("cuda", 0)
key = torch.from_numpy(w.records.read_key.view(np.int64)).to(dev); ref = torch.from_numpy(w.records.ref_id).to(dev)
pos = torch.from_numpy(w.records.begin_pos).to(dev); flag = torch.from_numpy(w.records.flag.view(np.int16)).to(dev)
L = C.CDLL(capi.LIB_PATH)
out = (C.c_ulonglong * 8)()
for it in range(3):
    eng.reset(); eng.reset_cutoffs(); eng.set_records_device(key, ref, pos, flag)
    L.slimm_debug_prof_filter(out, 1)
    eng.get_profiles(); torch.cuda.synchronize()
L.slimm_debug_prof_filter(out, 0)
v = list(out); n = max(1, v[3])
for i, nm in enumerate(["offsets + targets + rows + LCA", "taxon + marks / pairs", "selector store"]):
    print(f"{nm:32s} {v[i]/n:10.0f} ticks per wave")
