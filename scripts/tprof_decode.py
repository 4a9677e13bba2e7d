"""Cycle split of the device inflate's first phase (k_inflate_decode, bgzf_tokens.hip), from a build with the counters compiled in:
    scripts/build_variant.sh dprof '1s/^/#define EXP 12\\n/'
    SLIMM_HIP_LIB=build/var/dprof/libslimm_hip.so python scripts/tprof_decode.py [records] [easy|realistic] [blocks]
Prints the mean cycles lane 0 of a wave spent per phase of its steps (with the waits the probes force), the steps and the header time."""
import ctypes as C, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from slimm_amd import capi
from slimm_amd.synth import CONFIGS, make_workload
from slimm_amd.synth_bam import write_synthetic_bam

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30_000_000
kind = sys.argv[2] if len(sys.argv) > 2 else "realistic"
n_blocks = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
w = make_workload(CONFIGS["config3"], seed=1, n_records=n)
tmp = tempfile.mkdtemp(prefix="slimm_inf_")
bam = os.path.join(tmp, kind + ".bam")
info = write_synthetic_bam(bam, w.ref_names, w.ref_len, w.records, read_len=100, realistic=(kind == "realistic"))
L = capi.lib()
blob = np.fromfile(bam, dtype=np.uint8, count=min(os.path.getsize(bam), n_blocks * 66000))
os.unlink(bam)
p, k = 0, 0
while p + 18 <= blob.size and k < n_blocks:
    bs = int(blob[p + 16]) + (int(blob[p + 17]) << 8) + 1
    if p + bs > blob.size:
        break
    p += bs
    k += 1
part = blob[:p]
out = np.zeros(k * 65536 + 64, dtype=np.uint8)
names = ["wait for the input asked for a step ago", "first symbol (limits, two LDS round trips)", "second symbol", "state + stores queued",
         "steps", "headers", "wave", "lane 0: steps with a pair of literals"]
m = 8 * 1024
buf = (C.c_ulonglong * m)()
for it in range(3):
    L.slimm_debug_prof_decode(buf, m, 1)
    nb, ms, err, lb = C.c_uint64(), C.c_double(), C.create_string_buffer(256), C.c_uint32()
    rc = L.slimm_bgzf_inflate_with(0, part.ctypes.data_as(C.c_void_p), C.c_uint64(part.size), out.ctypes.data_as(C.c_void_p),
                                   C.c_uint64(out.size), C.byref(nb), C.byref(ms), err, C.c_uint64(256), 0, C.byref(lb))
    assert rc == 0, err.value
    L.slimm_debug_prof_decode(buf, m, 0)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8).astype(np.float64)
    per = max(1.0, (k / 64) / 1024.0)   # waves per counter row
    print(f"run {it}: {k} blocks, kernels {ms.value:.2f} ms = {nb.value / ms.value / 1e6:.1f} GB/s; per wave:",
          ", ".join(f"{nm} {a[:, i].mean() / per:.0f}" for i, nm in enumerate(names)), flush=True)
