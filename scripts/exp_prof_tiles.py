# scratch: per-phase cycle totals of k_tile_scatter (build with -DEXP=8)
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
NW = 4096
out = np.zeros(NW * 8, dtype=np.uint64)
for it in range(3):
    eng.reset(); eng.set_records_device(key, ref, pos, flag)
    L.slimm_debug_prof_tiles(out.ctypes.data_as(C.c_void_p), NW * 8, 1)
    eng.analyze_alignments(); torch.cuda.synchronize()
L.slimm_debug_prof_tiles(out.ctypes.data_as(C.c_void_p), NW * 8, 0)
v = out.reshape(NW, 8).astype(np.float64)
names = ["loads", "lds count", "reserve atomics", "scatter", "final barrier"]
for i, nm in enumerate(names): print(f"{nm:20s} mean {v[:, i].mean():10.0f}  p50 {np.median(v[:, i]):10.0f} p99 {np.percentile(v[:, i], 99):10.0f} ticks")
print("sum", v[:, :5].sum(axis=1).mean())
