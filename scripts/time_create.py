import sys, time
sys.path.insert(0, ".")
t0 = time.time()
from slimm_amd import capi
import ctypes as C
L = capi.lib()
t1 = time.time()
L.slimm_warm_up(0)
t2 = time.time()
from slimm_amd.profiler import Slimm
from slimm_amd.synth import CONFIGS, make_workload
w = make_workload(CONFIGS["config3"], seed=1, n_records=100_000)
import os
os.environ["SLIMM_TRACE"] = "host"
for i in range(3):
    t3 = time.time()
    s = Slimm.for_workload(w, device=0)
    t4 = time.time()
    s.reserve(100_000_000) if hasattr(s, "reserve") else None
    t5 = time.time()
    print(f"create {1e3*(t4-t3):.1f} ms, reserve(100 M) {1e3*(t5-t4):.1f} ms")
    s.close()
print(f"dlopen {1e3*(t1-t0):.1f} ms, warm_up {1e3*(t2-t1):.1f} ms")
