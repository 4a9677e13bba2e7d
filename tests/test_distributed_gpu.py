"""The multi-process driver (slimm_amd/distributed.py, bench.py --gpus N) with the REAL HIP engine under more than one OS
process -- the rehearsal of what an 8-GPU node runs, on the one GPU of the test box: every rank is a process of its own
with its own `Slimm` context on cuda:0, the process group is gloo, and the collectives' device tensors are staged
through host memory (slimm_amd/distributed.py: _host_staged).  Everything else is the code path of an RCCL run: the
partitioner's cuts, the three exchange forms between phase A and the cut-offs, the launched phase B with one in-place
all-reduce of the partial results, rank 0 writing the profile.  Every rank must end with the single-process oracle's
result for the whole stream (reference unit of work: one file through one object, src/slimm.hpp:950-956)."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import os, sys
sys.path.insert(0, {root!r})
import numpy as np
import torch
import torch.distributed as dist

rank, world, port, case, tmp, exchange, cuts = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5], sys.argv[6], sys.argv[7]
os.environ["MASTER_ADDR"] = "127.0.0.1"
os.environ["MASTER_PORT"] = port
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
try:
    from oracle.binding import run_workload
    from slimm_amd.distributed import sharded_profile
    from slimm_amd.partition import shard_records
    from slimm_amd.profiler import Slimm
    from slimm_amd.synth import CONFIGS, SynthConfig, make_workload
    from slimm_amd.workload import Workload
    from tests.helpers import assert_matches_oracle

    if case == "config1":
        w = make_workload(CONFIGS["config1"], seed=21)
    elif case == "cross":   # reads straddling superkingdoms: (taxon, reference) pairs travel in the second exchange (Q4)
        cfg = SynthConfig("x", 60_000, 10_000, 5.0, present_frac=0.3, len_lo=20_000, len_hi=60_000)
        w = make_workload(cfg, seed=22)
        rng = np.random.default_rng(1)
        m = (w.records.ref_id >= 0) & (rng.random(len(w.records)) < 0.2)
        w.records.ref_id[m] = rng.integers(0, cfg.n_refs, size=int(m.sum()), dtype=np.int32)
        w.records.begin_pos[m] = 100
    else:
        w = make_workload(CONFIGS["config2"], seed=23, n_records=400_000)
    grouped = cuts == "contiguous"
    rec, still_grouped = shard_records(w.records, rank, world, grouped=grouped)
    assert still_grouped == grouped
    eng = Slimm(w.taxonomy, w.options, w.ref_names, w.ref_len, w.avg_read_len, device=0, grouped=still_grouped)
    eng.push_records(rec)
    text = sharded_profile(eng, torch.device("cuda:0"), os.path.join(tmp, "profile.tsv"), exchange=exchange)
    whole = run_workload(w, use_qnames=False, collect_bins=False)
    assert text is not None
    assert_matches_oracle(eng, whole, bins=False)
    st = eng.stats()
    assert st["n_records"] == len(rec)                      # its own share of the file ...
    tot = torch.tensor([len(rec)], dtype=torch.int64)
    dist.all_reduce(tot)
    assert int(tot[0]) == len(w.records)                     # ... of a partition of it
    dist.barrier()
    if rank == 0:
        assert open(os.path.join(tmp, "profile.tsv")).read() == text
    if case == "cross":
        assert len(eng.children_pairs(0)) > 0
    print("rank", rank, "ok", flush=True)
finally:
    dist.destroy_process_group()
'''


def _spawn(world, case, exchange, cuts):
    port = str(29500 + (os.getpid() % 1500) + 7 * world + {"config1": 0, "config2": 1, "cross": 2}[case]
               + {"summary": 0, "bins": 20, "sliced": 40, "auto": 60}[exchange] + (80 if cuts == "hash" else 0))
    with tempfile.TemporaryDirectory() as tmp:
        script = os.path.join(tmp, "worker.py")
        with open(script, "w") as f:
            f.write(_WORKER.format(root=ROOT))
        procs = [subprocess.Popen([sys.executable, script, str(r), str(world), port, case, tmp, exchange, cuts],
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
        outs = []
        for p in procs:
            try:
                outs.append(p.communicate(timeout=600)[0])
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                raise
        for r, (p, o) in enumerate(zip(procs, outs)):
            assert p.returncode == 0 and f"rank {r} ok" in o, f"rank {r} of {world} failed:\n{o[-3000:]}"


@pytest.mark.parametrize("case,exchange,cuts", [("config2", "summary", "contiguous"), ("config2", "sliced", "contiguous"),
                                                ("config1", "bins", "contiguous"), ("config2", "summary", "hash"),
                                                ("cross", "summary", "hash"), ("config2", "bins", "hash")])
def test_two_processes_with_real_engines_on_one_gpu(case, exchange, cuts):
    """World size 2: contiguous cuts of the grouped file (shards stay grouped: the single-pass front end) and `key mod n`
    (shards declared SLIMM_ORDER_ANY: the device-side grouping), through each exchange form."""
    _spawn(2, case, exchange, cuts)


@pytest.mark.parametrize("exchange,cuts", [("auto", "contiguous"), ("summary", "hash"), ("bins", "contiguous")])
def test_three_processes_with_real_engines_on_one_gpu(exchange, cuts):
    """World size 3: "auto" is the all-to-all form (slices of unequal fill)."""
    _spawn(3, "config2", exchange, cuts)


def _bench(args, nproc=1):
    env = dict(os.environ, PYTHONPATH=ROOT)
    if nproc == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    else:
        port = str(31000 + os.getpid() % 1000)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
               "--master-port", port, os.path.join(ROOT, "bench.py"), "--gpus", str(nproc)] + args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("nproc", [2, 3])
def test_bench_multi_process_control_flow_on_one_gpu(nproc):
    """`bench.py --gpus N` as the driver launches it (torch.distributed.run, one process per rank), rehearsed with
    --backend gloo and every rank on cuda:0: the chunk ownership, the per-rank resident fill, the exchange, the
    all-reduce of the elapsed time and of the record counts -- its line must describe the SAME job as the N = 1 line:
    the same total, the same profile."""
    common = ["--records", "20000000", "--chunk-records", "2500000", "--steps", "2", "--warmup", "1", "--quick"]
    one = _bench(common)
    many = _bench(common + ["--backend", "gloo"], nproc=nproc)
    assert one["n_gpus"] == 1 and many["n_gpus"] == nproc
    assert many["config"]["total_records"] == one["config"]["total_records"] == 20_000_000
    assert many["config"]["records_per_gpu"] < one["config"]["records_per_gpu"]
    assert many["config"]["process_group_ranks"] == nproc and many["config"]["backend"] == "gloo"
    assert many["scaling"] == "strong"
    for k in ("reads", "targets", "bins", "profile_rows", "profile_sha1"):
        assert many["config"][k] == one["config"][k], k
    assert many["config"]["exchange"] == ("summary" if nproc == 2 else "sliced")
    # (which kernel the shared GPU makes the longest is no property of the code: two processes time-slice one device)
    assert many["value"] > 0 and many["roofline"]["kernel"].startswith("k_") and many["roofline"]["frac"] > 0


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (how the driver's N = 1 record shows bench.py being started):
    the parent starts torch.distributed.run as a child before it touches a GPU, relays rank 0's one line and leaves with the
    child's exit code.  The N > 1 line carries the step's split and the same step with north_star's literal collective (one
    all-reduce over the integer bins)."""
    many = _bench(["--gpus", "2", "--backend", "gloo", "--quick", "--records", "4000000", "--chunk-records", "500000", "--steps", "2",
                   "--warmup", "1"])
    assert many["n_gpus"] == 2 and many["config"]["process_group_ranks"] == 2 and many["config"]["total_records"] == 4_000_000
    sp = many["step_split"]
    assert sp["kernels_ms"] > 0 and sp["collectives_ms"] > 0 and set(sp["collectives"]) == {"all_gather", "all_reduce"}
    xb = many["exchange_bins"]
    assert xb["same_profile"] is True and xb["value"] > 0 and xb["all_reduce_bytes"] > 0


def test_eight_processes_with_real_engines_on_one_gpu():
    """The world size of the driver's node: eight OS processes with real engines on cuda:0, "auto" = the all-to-all of eight
    bitmap slices + the small all-reduce, on contiguous cuts of the grouped file."""
    _spawn(8, "config2", "auto", "contiguous")


def test_bench_eight_ranks_rehearsal_on_one_gpu():
    """`python bench.py --gpus 8 --backend gloo --quick --records 40000000`: what the driver's
    8-GPU node runs, rehearsed with every rank on cuda:0 -- chunk ownership 16 chunks / 8 ranks, `sliced` with eight
    slices, `step_split`, `exchange_bins` -- and the same job as the N = 1 line."""
    common = ["--records", "40000000", "--chunk-records", "2500000", "--steps", "2", "--warmup", "1", "--quick"]
    one = _bench(common)
    many = _bench(["--gpus", "8", "--backend", "gloo"] + common)       # (bench.py starts its own ranks)
    assert many["n_gpus"] == 8 and many["config"]["process_group_ranks"] == 8 and many["config"]["exchange"] == "sliced"
    assert many["config"]["total_records"] == one["config"]["total_records"] == 40_000_000
    for k in ("reads", "targets", "bins", "profile_rows", "profile_sha1"):
        assert many["config"][k] == one["config"][k], k
    sp = many["step_split"]
    assert sp["kernels_ms"] > 0 and sp["collectives_ms"] > 0
    assert many["exchange_bins"]["same_profile"] is True
    print("N=8 rehearsal step_split:", json.dumps(sp))


def test_bench_with_a_rank_that_holds_no_records():
    """One chunk, two ranks: rank 0 owns nothing (partition.chunk_owner(1, 2) = [[], [0]]).  Its launchers have nothing to launch,
    the kernel timers' event pairs stay unrecorded -- and reading them left "invalid resource handle" as the thread's last HIP
    error, which the empty rank's SECOND step reported as its own (found with `bench.py --gpus 8 --records 40000000`: four chunks)."""
    common = ["--records", "10000000", "--chunk-records", "10000000", "--steps", "2", "--warmup", "1", "--quick"]
    one = _bench(common)
    many = _bench(["--gpus", "2", "--backend", "gloo"] + common)
    assert many["n_gpus"] == 2 and many["config"]["total_records"] == one["config"]["total_records"] == 10_000_000
    for k in ("reads", "targets", "bins", "profile_rows", "profile_sha1"):
        assert many["config"][k] == one["config"][k], k
