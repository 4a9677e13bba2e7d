# scratch: plain walk, tagged-word walk and hash classification (k_runs + k_runs_hash time) over the mean hits per read,
# 10 M records; python scripts/exp_walk_crossover.py [hits ...]
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slimm_amd.profiler import Slimm
from slimm_amd.synth import SynthConfig, make_workload
dev = torch.device("cuda:0")
for hits in [float(x) for x in (sys.argv[1:] or ("3", "4", "5", "6", "7", "8", "10", "12"))]:
    w = make_workload(SynthConfig("x", 10_000_000, 5000, hits), seed=1)
    key = torch.from_numpy(w.records.read_key.view(np.int64)).to(dev); ref = torch.from_numpy(w.records.ref_id).to(dev)
    pos = torch.from_numpy(w.records.begin_pos).to(dev); flag = torch.from_numpy(w.records.flag.view(np.int16)).to(dev)
    s = Slimm.for_workload(w, device=0)
    out = []
    for mode in ("walk", "tagged", "hash", "auto"):
        if mode == "auto": os.environ.pop("SLIMM_RUNS_KERNEL", None)
        else: os.environ["SLIMM_RUNS_KERNEL"] = mode
        kn = "k_runs_hash" if mode == "hash" else "k_runs"
        s.enable_kernel_timing(True); s.time_only_kernel(None)
        for it in range(6):
            if it == 2: s.kernel_times(reset=True)
            s.reset(); s.reset_cutoffs(); s.set_records_device(key, ref, pos, flag); s.get_profiles()
        kt = s.kernel_times(reset=True)
        ms = sum(kt[k][0] / max(1, kt[k][1]) for k in ("k_runs", "k_runs_hash") if k in kt)
        out.append(f"{mode} {ms * 1e3:7.1f} us")
    st = s.stats()
    print(f"hits {hits:4.1f}  records/read {st['hits_count'] / max(1, st['matches_count']):5.2f}  " + "  ".join(out), flush=True)
