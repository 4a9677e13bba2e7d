# scratch: plain vs tagged-word duplicate walk (both bodies of k_runs) over the mean hits per read, 10 M records
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slimm_amd.profiler import Slimm
from slimm_amd.synth import SynthConfig, make_workload
dev = torch.device("cuda:0")
for hits in (3.0, 4.0, 5.0, 6.0, 7.0, 8.0, 10.0, 12.0):
    w = make_workload(SynthConfig("x", 10_000_000, 5000, hits), seed=1)
    key = torch.from_numpy(w.records.read_key.view(np.int64)).to(dev); ref = torch.from_numpy(w.records.ref_id).to(dev)
    pos = torch.from_numpy(w.records.begin_pos).to(dev); flag = torch.from_numpy(w.records.flag.view(np.int16)).to(dev)
    s = Slimm.for_workload(w, device=0)
    out = []
    for mode in ("walk", "tagged", "auto"):
        if mode == "auto": os.environ.pop("SLIMM_RUNS_KERNEL", None)
        else: os.environ["SLIMM_RUNS_KERNEL"] = mode
        s.enable_kernel_timing(True); s.time_only_kernel("k_runs")
        for it in range(6):
            if it == 2: s.kernel_times(reset=True)
            s.reset(); s.reset_cutoffs(); s.set_records_device(key, ref, pos, flag); s.get_profiles()
        ms, n = s.kernel_times(reset=True)["k_runs"]
        out.append(f"{mode} {ms / n * 1e3:7.1f} us")
    st = s.stats()
    print(f"hits {hits:4.1f}  records/read {st['hits_count'] / max(1, st['matches_count']):5.2f}  " + "  ".join(out), flush=True)
