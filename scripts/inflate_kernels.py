"""The device inflate by itself on a synthetic BAM (slimm_amd/synth_bam.py): both kernels' paths, the blocks the lane-per-block
kernel had to take, the rate.  For rocprofv3 --kernel-trace --stats (per-kernel times) and the PMC passes.
    python scripts/inflate_kernels.py [records] [easy|realistic] [blocks] [reps]"""
import ctypes as C, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from slimm_amd import capi
from slimm_amd.synth import CONFIGS, make_workload
from slimm_amd.synth_bam import write_synthetic_bam

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
kind = sys.argv[2] if len(sys.argv) > 2 else "realistic"
n_blocks = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
w = make_workload(CONFIGS["config3"], seed=1, n_records=n)
tmp = tempfile.mkdtemp(prefix="slimm_inf_")
bam = os.path.join(tmp, kind + ".bam")
info = write_synthetic_bam(bam, w.ref_names, w.ref_len, w.records, read_len=100, realistic=(kind == "realistic"))
print(f"{kind}: {info['raw_bytes'] / info['compressed_bytes']:.2f} x ({info['deflate']})", flush=True)
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
for how in (0, 1):
    best = None
    for it in range(reps):
        nb, ms, err, lb = C.c_uint64(), C.c_double(), C.create_string_buffer(256), C.c_uint32()
        rc = L.slimm_bgzf_inflate_with(0, part.ctypes.data_as(C.c_void_p), C.c_uint64(part.size), out.ctypes.data_as(C.c_void_p),
                                       C.c_uint64(out.size), C.byref(nb), C.byref(ms), err, C.c_uint64(256), how, C.byref(lb))
        assert rc == 0, err.value
        best = ms.value if best is None else min(best, ms.value)
    print(f"how={how} ({'two-phase' if how == 0 else 'lane per block'}): {k} blocks, {part.size / 1e6:.0f} MB -> {nb.value / 1e6:.0f} MB, kernels {best:.2f} ms = "
          f"{nb.value / best / 1e6:.1f} GB/s inflated; blocks through the lane-per-block kernel: {lb.value}", flush=True)
