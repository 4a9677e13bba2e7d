"""Stress: tiny inputs around the 2048-record tile size, both record orders, in contexts created on device memory that
larger runs of other shapes have just dirtied (uninitialised-read / race hunting).  GPU only; prints the failures."""
import sys, numpy as np
sys.path.insert(0, '.')
import torch
torch.cuda.init()
from slimm_amd.synth import SynthConfig, CONFIGS, make_workload
from slimm_amd.workload import Workload
from slimm_amd.profiler import Slimm
from oracle.binding import run_workload
from tests.helpers import assert_matches_oracle
base = make_workload(SynthConfig("rag", 9000, 40, 5.0, bin_width=100, len_lo=5_000, len_hi=50_000, present_frac=0.5), seed=14)
dirty = [make_workload(CONFIGS["config2"], seed=3, n_records=300_000),
         make_workload(SynthConfig("hot_tile", 300_000, 12, 1.3, bin_width=1000, len_lo=300_000, len_hi=600_000, present_frac=0.5), seed=5),
         make_workload(SynthConfig("c5w", 200_000, 3_000, 40.0, strain_level=True), seed=23)]
cases = []
for n in (1, 2, 63, 64, 65, 255, 256, 257, 2047, 2048, 2049, 4096, 4097, 8191):
    w = Workload(base.ref_names, base.ref_len, base.taxonomy, base.records.take(np.arange(n)), base.avg_read_len, base.options, f"rag{n}")
    cases.append((n, w, run_workload(w, use_qnames=False)))
fails = 0
for rep in range(40):
    d = dirty[rep % 3]
    s = Slimm.for_workload(d, device=0, grouped=bool(rep & 1)); s.push_records(d.records); s.get_profiles(); s.close()
    for n, w, o in cases:
        for grouped in (True, False):
            s = Slimm.for_workload(w, device=0, grouped=grouped)
            s.push_records(w.records)
            prof = s.get_profiles()
            try:
                if o.no_hits: assert prof is None
                else: assert_matches_oracle(s, o)
            except AssertionError as e:
                fails += 1
                print("FAIL rep", rep, "n", n, "grouped", grouped, str(e)[:400].replace("\n", " | "))
            s.close()
print("fails", fails)
