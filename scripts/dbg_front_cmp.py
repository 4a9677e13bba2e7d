import numpy as np, sys
a_r, a_g = np.load("/tmp/dbg_gpu_ref.npy"), np.load("/tmp/dbg_gpu_gbin.npy")
b_r, b_g = np.load("/tmp/dbg_emu_ref.npy"), np.load("/tmp/dbg_emu_gbin.npy")
print("gpu", len(a_r), "emu", len(b_r))
n = min(len(a_r), len(b_r))
d = np.nonzero((a_r[:n] != b_r[:n]) | (a_g[:n] != b_g[:n]))[0]
print("first diffs at", d[:10])
if len(d):
    i = max(0, int(d[0]) - 6)
    for k in range(i, min(n, i + 24)):
        print(k, hex(a_r[k]), hex(a_g[k]), "|", hex(b_r[k]), hex(b_g[k]), "<--" if k in d[:50] else "")
