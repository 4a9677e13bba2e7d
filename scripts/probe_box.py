"""What the GPU box offers the host side: cores, memory, libraries; how fast the synthetic generator runs in a pool."""
import os, sys, time, subprocess, glob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
def sh(c):
    print("$", c, flush=True); print(subprocess.run(c, shell=True, capture_output=True, text=True).stdout, flush=True)
sh("nproc; cat /sys/fs/cgroup/cpu.max; cat /sys/fs/cgroup/memory.max; free -g; df -h /tmp /dev/shm | cat")
sh("ls -la /usr/lib/x86_64-linux-gnu/ | grep -i -E 'deflate|libz|isal|zstd'")
sh("lscpu | head -20")
from slimm_amd.synth import CONFIGS, make_workload
def gen(c):
    w = make_workload(CONFIGS["config4"], seed=1 + 1000 * c, n_records=10_000_000, sample_seed=1, shard=c)
    r = w.records
    return c, r.read_key, r.ref_id, r.begin_pos, r.flag
if __name__ == "__main__":
    t = time.time(); gen(0); print("one chunk, one process:", time.time() - t, flush=True)
    import multiprocessing as mp
    for workers in (8, 16, 32):
        ctx = mp.get_context("spawn")
        t = time.time()
        with ctx.Pool(workers) as p:
            n = 0
            for c, k, r, ps, f in p.imap_unordered(gen, range(32)):
                n += len(k)
        print(f"{workers} workers: 32 chunks ({n} records) in {time.time() - t:.1f}s", flush=True)
