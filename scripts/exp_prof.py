# scratch: per-phase cycle totals of k_runs (build with -DEXP=9)
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
NW = 19532
out = np.zeros(NW * 8, dtype=np.uint64)
for it in range(3):
    eng.reset(); eng.set_records_device(key, ref, pos, flag); eng.analyze_alignments(); torch.cuda.synchronize()
L.slimm_debug_prof(out.ctypes.data_as(C.c_void_p), NW * 8)
v = out.reshape(NW, 8).astype(np.float64)
names = ["stage1 loads+meta", "barrier1", "carry", "pre-walk", "walk", "finalize"]
for i, nm in enumerate(names): print(f"{nm:20s} mean {v[:, i].mean():10.0f}  p50 {np.median(v[:, i]):10.0f} p99 {np.percentile(v[:, i], 99):10.0f} ticks")
print("sum of phase means", v[:, :6].sum(axis=1).mean())
# start times per XCD-ish: tiles are dispatched round-robin; show spread of start ticks for tiles 0,8,16.. (same XCD)
t0 = v[0::32, 7]
print("start tick spread (same-XCD tiles):", t0.min(), t0.max(), (t0.max() - t0.min()))
