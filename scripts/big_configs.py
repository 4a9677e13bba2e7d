"""Runs the larger BASELINE.json configs once on the GPU: size-independent invariants + per-kernel times.
    python scripts/big_configs.py config3 [n_records]
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slimm_amd.profiler import Slimm
from slimm_amd.synth import CONFIGS, make_workload
from oracle.binding import parse_profile

name = sys.argv[1]
cfg = CONFIGS[name]
n = int(sys.argv[2]) if len(sys.argv) > 2 else cfg.n_records
t0 = time.time(); w = make_workload(cfg, seed=1, n_records=n); print(f"{name}: generated {n} records in {time.time()-t0:.1f}s", flush=True)
dev = torch.device("cuda:0")
key = torch.from_numpy(w.records.read_key.view(np.int64)).to(dev); ref = torch.from_numpy(w.records.ref_id).to(dev)
pos = torch.from_numpy(w.records.begin_pos).to(dev); flag = torch.from_numpy(w.records.flag.view(np.int16)).to(dev)
t0 = time.time(); s = Slimm.for_workload(w, device=0); print(f"context {time.time()-t0:.2f}s", flush=True)
s.enable_kernel_timing(True)
for it in range(3):
    s.reset(); s.reset_cutoffs(); s.set_records_device(key, ref, pos, flag)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    prof = s.get_profiles()
    dt = time.perf_counter() - t0
    print(f"run {it}: {dt*1e3:.2f} ms  -> {n/dt/1e6:.0f} M records/s", flush=True)
st = s.stats(); print({k: st[k] for k in ("hits_count", "matches_count", "n_targets", "uniq_matches_count", "uniq_matches_count2", "n_valid", "total_bins")})
kt = s.kernel_times()
for k, (ms, ln) in sorted(kt.items(), key=lambda kv: -kv[1][0]):
    if ln: print(f"  {k:18s} {ms/ln*1e3:9.1f} us x{ln}")
rc = s.ref_columns()
mapped = int((((w.records.flag & 4) == 0) & (w.records.ref_id >= 0)).sum())
assert st["hits_count"] == mapped % 2**32
assert int(rc["reads_count"].sum(dtype=np.uint64)) == st["n_targets"]
assert int(rc["uniq_reads_count"].sum(dtype=np.uint64)) == st["uniq_matches_count"]
assert int(rc["uniq_reads_count2"].sum(dtype=np.uint64)) == st["uniq_matches_count2"]
rows = parse_profile(prof)
assert sum(v[1] for v in rows.values()) == st["matches_count"], (sum(v[1] for v in rows.values()), st["matches_count"])
cov = s.bins(0); assert int(cov.sum(dtype=np.uint64)) == st["n_targets"]
off = np.concatenate([[0], np.cumsum(rc["nbins"].astype(np.int64))])
assert np.array_equal(np.add.reduceat((cov != 0).astype(np.int64), off[:-1]), rc["nz_cov"])
print("invariants ok;", len(rows), "profile rows")
