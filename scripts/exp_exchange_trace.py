# scratch: where the forced-exchange step spends its host time (world size 1, NCCL), finer than bench.py --breakdown
import os, sys, time, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slimm_amd.profiler import Slimm
from slimm_amd.synth import CONFIGS, make_workload
from slimm_amd import distributed as D
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
w = make_workload(CONFIGS["config2"], seed=1, sample_seed=1)
eng = Slimm.for_workload(w, device=0, grouped=True); eng.force_exchange = True
key = torch.from_numpy(w.records.read_key.view(np.int64)).to(dev); ref = torch.from_numpy(w.records.ref_id).to(dev)
pos = torch.from_numpy(w.records.begin_pos).to(dev); flag = torch.from_numpy(w.records.flag.view(np.int16)).to(dev)
mode = sys.argv[1] if len(sys.argv) > 1 else "summary"
T = {}
def lap(name, t0):
    t = time.perf_counter(); T[name] = T.get(name, 0.0) + (t - t0); return t
N = 30
for it in range(N + 5):
    if it == 5: T.clear()
    eng.reset(); eng.reset_cutoffs(); eng.set_records_device(key, ref, pos, flag)
    t = time.perf_counter()
    eng.prepare_summary(1); t = lap("prepare_summary", t)
    eng.analyze_alignments(); t = lap("analyze(launch)", t)
    mine = eng.coverage_summary_tensor(); t = lap("coverage_summary_tensor (sync A)", t)
    g = torch.empty(mine.numel(), dtype=mine.dtype, device=mine.device); t = lap("torch.empty", t)
    dist.all_gather_into_tensor(g, mine); t = lap("all_gather call", t)
    torch.cuda.synchronize(dev); t = lap("sync after all_gather", t)
    eng.finish_coverage_merged(g, 1); t = lap("finish_coverage_merged", t)
    eng.filter_alignments(); t = lap("filter_alignments", t)
    pt = eng.partials_tensor(); t = lap("partials_tensor", t)
    dist.all_reduce(pt); t = lap("all_reduce call", t)
    torch.cuda.synchronize(dev); t = lap("sync after all_reduce", t)
    eng.install_merged_partials(); t = lap("install_merged_partials", t)
    eng.get_reads_lca_count(); t = lap("get_reads_lca_count", t)
    s = eng.write_abundance("/tmp/x_profile.tsv"); t = lap("write_abundance", t)
tot = 0
for k, v in T.items():
    print(f"{k:36s} {v / N * 1e6:8.1f} us"); tot += v
print(f"{'total':36s} {tot / N * 1e6:8.1f} us")
dist.destroy_process_group()
