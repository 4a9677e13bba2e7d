# scratch: the front end's targets from the GPU library next to those of the host emulation of the same sources
import os, sys, subprocess, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
which = sys.argv[1]
from slimm_amd import capi
if which == "dbg":
    capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), "..", "build", "dbg", "libslimm_hip.so")
if which == "emu":
    capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), "..", "tests", "native", "libslimm_emu.so")
from slimm_amd.profiler import Slimm
from slimm_amd.synth import CONFIGS, make_workload
n = int(sys.argv[2])
w = make_workload(CONFIGS["config2"], seed=3, n_records=n)
s = Slimm.for_workload(w, device=0)
s.push_records(w.records)
s.analyze_alignments()
ref, gbin = s.read_targets()
np.save(f"/tmp/dbg_{which}_ref.npy", ref); np.save(f"/tmp/dbg_{which}_gbin.npy", gbin)
print(which, len(ref), int((ref >> 31).sum()), int((gbin >> 31).sum()))
