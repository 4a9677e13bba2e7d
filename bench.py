#!/usr/bin/env python3
"""Benchmark of the alignment-to-profile hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): M alignment-records/sec from the decoded-record stream to the final profile.

Workload, at EVERY N: BASELINE.json configs[3] -- ONE seeded 1 B-record stream (20 k refs, mean 8 hits/read, 1000 bp
bins, 100 bp reads) in 100 chunks of 10 M records, each chunk a grouped file of whole reads; rank r of N generates and
keeps chunks [r C / N, (r + 1) C / N) straight into HBM (strong scaling: the total is fixed, cuts fall on read
boundaries, slimm_amd/partition.py).  The stream fits one MI355X, so N = 1 runs all of it: the N = 1 line and the N > 1
lines are the same job.  A step = one pass of the whole hot path over the rank's records, which are resident in HBM as
packed 16-byte records (key with the three flag bits folded in, ref, pos: slimm_set_records_device_packed):
analyze_alignments -> (RCCL exchange when N > 1) -> finish_coverage -> filter_alignments -> get_reads_lca_count ->
write_abundance (profile TSV written to a file by rank 0), every step a fresh file for a fresh object.  `value` = total
records / max-over-ranks time.

ONE JSON line on rank 0.  Beside the contract's fields:
  roofline            the dominant kernel of the headline workload: algorithmic bytes (16 N + 8 P
                      for the front end) / its dispatch's own duration, measured live by HIP events handed to the
                      launch (hipExtLaunchKernelGGL) on the library's stream inside the timed steps
  roofline_config2 / _config3 / _config5   the same for BASELINE.json configs[1], [2] (the designated roofline run) and
                      [4] (100 M records, 50 k strain-level refs, 40 hits/read), resident, N = 1 only
  two_files_in_flight two host threads, each taking its files back to back through two contexts of its own: the kernels of
                      two files share the device (round 5: +5 % at 1 B records, +14 % at 100 M).  Reported beside `value`,
                      which keeps one file at a time
  value_with_push     the clock starts before the first record leaves page-locked HOST memory
                      (slimm_push_records_packed_async) and stops when the profile file is written -- SURVEY.md 8d (1)
  run_marked_records  the same stream handed over as run-marked 8-byte records (include/slimm_hip.h): resident rate, k_front
                      under 8 N + 8 P, push-inclusive rate -- beside the headline, which stays on SURVEY 8d's 16-byte records
  record_order_any    the headline stream with the reads of every chunk interleaved at random, and config 3 likewise,
                      through a context created for SLIMM_ORDER_ANY (the reference takes records in any order,
                      src/slimm.hpp:204-211): step rate, the grouping kernels (slimm_amd/csrc/group_by_ident.hip) under their
                      own byte model, and that the profile is the grouped stream's
  parity_in_run       the config-3 leg's scalars, five per-reference columns, direct LCA counts and three bin checksums
                      against oracle/slimm_dense_mt.cpp run on the same 100 M records in this very run
  cpu_baseline        the CPU oracle (a port of the reference algorithm, 1 thread) on a bounded prefix of the stream
  cpu_baseline_mt     the dense all-core restatement on a bounded prefix
  cli_end_to_end      `slimm DB IN.bam` on a 100 M-record synthetic BAM that compresses 3-fold (random bases, binned qualities,
                      instrument-style names), process start to profile written; the same with the host inflating, an unsorted
                      copy, and the 17.7-fold file of rounds 1 - 4 beside it
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


def group_plan(n_records):
    """slimm_group_plan: (passes, width, bits, grid) of the record_order = ANY grouping for a stream of n_records."""
    import ctypes as C

    from slimm_amd import capi
    v = [C.c_uint32() for _ in range(4)]
    capi.lib().slimm_group_plan(int(n_records), *[C.byref(x) for x in v])
    return tuple(x.value for x in v)


def algorithmic_bytes(st, n_records, Bp_words, rec_bytes=16):
    """Algorithmic HBM bytes per launch of each kernel (DESIGN.md section 'Kernels and their rooflines').

    N records, V mapped records, P targets = distinct (read, ref) pairs, M reads, U / U2 unique reads before / after
    the filter, B coverage bins.  The totals add up to SURVEY.md section 8d's figure (about 40 B/record + 8 B/bin).
    """
    N, V, P, M = n_records, st["hits_count"], st["n_targets"], st["matches_count"]
    U, U2, B = st["uniq_matches_count"], st["uniq_matches_count2"], Bp_words
    GP, GW, _, GG = group_plan(n_records)
    return {
        "memset_bins": 4 * (B // 8192 + 5200),       # counters, tail, tile counters, child marks (bins are written whole by the tile kernels)
        # record_order = ANY (group_by_ident.hip), per launch = the passes' average: the first pass reads the caller's
        # records (count: key + ref (+ flag); scatter: all of it), the others the 16-byte grouped form (count: identity)
        "k_group_count": ((rec_bytes - 4) * N + (GP - 1) * 8 * V) // GP,
        "k_group_scan": 2 * 4 * GG * (1 << GW),
        "k_group_scatter": (rec_bytes * N + 16 * V + (GP - 1) * 32 * V) // GP,
        "k_group_finish": 8 * V,                      # the identities once; only buckets holding several identities out of order are rewritten
        "k_front": rec_bytes * N + 8 * P + 16 * (N // 768 + 1),  # every record once (key 8 + ref 4 + pos 4 (+ flag 2) bytes);
                                                      # targets (ref word + bin word) and the slot descriptors out
        "k_hist": 8 * P + 8 * P + 8 * U,              # (fallback path) targets in; one 4-byte RMW per target / unique read
        "k_tile_count": 4 * P,                        # gbin in
        "k_tile_scan": 12 * (B // 8192 + 1),
        "k_tile_scatter": 4 * P + 2 * P,              # bin words in (the unique bit rides in bit 31), 16-bit bucket entries out
        "k_tile_hist": 2 * P + 8 * B,                 # bucket in, finished cov + uniq_cov tiles out (replaces the zero-fill)
        "k_tile_count2": 4 * M,                       # per-read bin (or marker) in
        "k_tile_scan2": 12 * (B // 8192 + 1),
        "k_tile_scatter2": 4 * M + 2 * U2,
        "k_tile_hist2": 2 * U2 + 4 * B,               # finished uniq_cov2 tiles out (replaces its zero-fill)
        "k_ref_stats": 8 * B,                         # (multi-GPU bins exchange / fallback) one streaming read of cov and uniq_cov
        "k_pack": 4 * 2 * 48,                         # counters + scalars copied behind the statistics k_tile_hist accumulated
        "k_pack2": 4 * 2 * (32 + 5000 + 9000),        # counters, child marks, per-taxon counts
        "k_filter": 8 * P + 4 * M,                    # targets in (lineage rows are cache resident); one selector per read out
        "k_ref_stats2": 4 * B,
    }


def cpu_quota_cores():
    """CPU time this process may use, in cores: the cgroup quota when there is one (a container on a 256-thread host
    may be held to 16 cores' worth), else the number of logical CPUs."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return round(int(q) / int(p), 2)
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return round(q / p, 2)
    except Exception:
        pass
    return float(os.cpu_count() or 1)


PMC_SUMMARY = os.path.join(ROOT, "profiles", "round6", "pmc_traffic_summary.json")


def pmc_traffic_bytes(kernel_name, workload):
    """HBM bytes per launch of `kernel_name` at `workload` ("config4", "config2" ...) from the committed rocprofv3 PMC
    passes (scripts/pmc_traffic.sh on MI355X; FETCH_SIZE and WRITE_SIZE collected in separate passes, unit KB).  gfx950
    correction per MI355X_MICROARCH.md section HBM: FETCH_SIZE tallies 128-byte requests at 64 bytes, i.e. half of a
    coalesced stream (calibrated in round 2 on k_ref_stats, whose 16 B/lane loads read exactly 8 B per bin: ratio
    2.03), so fetched bytes = 2 * FETCH_SIZE; WRITE_SIZE is exact.  None when no profile matches."""
    if not os.path.exists(PMC_SUMMARY):
        return None
    with open(PMC_SUMMARY) as f:
        d = json.load(f).get(workload, {})
    for k, v in d.items():
        if k.split("<")[0] == kernel_name and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            return int((2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024)
    return None


def cli_one_billion(records: int) -> dict:
    """`slimm DB IN.bam` on the 1 B-record realistic BAM (scripts/cli_1B.py, a process of its own): M records/s of the runs,
    the stage trace of the best one, device memory in use and the host's peak resident set."""
    import re
    need = records * 80
    room = max((shutil.disk_usage(d).free for d in ("/dev/shm", "/tmp") if os.path.isdir(d)), default=0)
    mem_free = None
    try:
        mx = open("/sys/fs/cgroup/memory.max").read().strip()
        cur = int(open("/sys/fs/cgroup/memory.current").read())
        mem_free = (int(mx) - cur) if mx != "max" else None
    except (OSError, ValueError):
        pass
    if room < need * 1.05:
        return {"skipped": f"no scratch room for the file: {need / 1e9:.0f} GB needed, {room / 1e9:.0f} GB free in /dev/shm and /tmp"}
    if mem_free is not None and mem_free < need + (40 << 30):
        return {"skipped": f"the memory cgroup has {mem_free / 1e9:.0f} GB left; the file in /dev/shm needs {need / 1e9:.0f} GB + 40"}
    t0 = time.time()
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "cli_1B.py"), str(records), "-", "one"], capture_output=True,
                           text=True, timeout=1200)
    except subprocess.TimeoutExpired:
        return {"error": "scripts/cli_1B.py did not finish in 1200 s"}
    out = r.stdout
    runs = [(float(a), float(b)) for a, b in re.findall(r"run \d+\]: ([0-9.]+) s = ([0-9.]+) M records/s", out)]
    if r.returncode != 0 or not runs:
        return {"error": (out[-300:] + r.stderr[-300:])}
    built = re.search(r"== (\d+) records, ([0-9.]+) GB of BAM in ([0-9.]+) GB = ([0-9.]+) x .* built in (\d+) s in (\S+)", out)
    mem = re.search(r"device memory in use ([0-9.]+) GB of (\d+) GB \(window pipeline ([0-9.]+) GB\); host peak resident set ([0-9.]+) GB", out)
    best = min(runs)
    trace_lines = [ln.strip() for ln in out.split("run %d]" % (runs.index(best) + 1))[-1].splitlines() if "[trace]" in ln][:16]
    o = {"value": best[1], "unit": "M records/s", "seconds": best[0], "runs_M_records_s": [b for _, b in runs],
         "what": "`slimm -w 1000 DB IN.bam`, process start to profile written, on the headline stream as ONE realistic BAM "
                 "(scripts/cli_1B.py); the first run reads a file that was just written, the second the same file again",
         "took_s": round(time.time() - t0, 1),
         "trace": [t for t in trace_lines if any(k in t for k in ("slimm_create", "device decode", "reader:", "rest of read", "slimm_destroy"))]}
    if built:
        o.update({"records": int(built.group(1)), "bam_GB": float(built.group(3)), "compression_ratio": float(built.group(4)),
                  "built_in_s": int(built.group(5)), "where": built.group(6)})
    if mem:
        o.update({"device_memory_in_use_GB": float(mem.group(1)), "window_pipeline_GB": float(mem.group(3)),
                  "host_peak_rss_GB": float(mem.group(4))})
    return o


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="config4",
                    help="the headline workload (default: BASELINE.json configs[3], the 1 B-record stream the metric's "
                         "1/2/4/8-GPU curve is quoted on; it fits one GPU)")
    ap.add_argument("--records", type=int, default=0, help="records of the stream over ALL ranks (default: the config's size)")
    ap.add_argument("--chunk-records", type=int, default=10_000_000, help="records per chunk of the stream")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--weak", action="store_true",
                    help="weak scaling instead: every rank its own --records-sized shard of the sample (round 1 / 2 default)")
    ap.add_argument("--no-marked", action="store_true", help="skip the leg that runs the stream as run-marked 8-byte records")
    ap.add_argument("--form", default="packed", choices=["packed", "four"],
                    help="record arrays handed to the library: 'packed' = 16 B/record (slimm_set_records_device_packed: key with "
                         "the three flag bits folded in, ref, pos), 'four' = 18 B/record (key, ref, pos, flag)")
    ap.add_argument("--no-bins", action="store_true",
                    help="do not materialise the coverage arrays in HBM (slimm_keep_bins(0): their statistics are taken from "
                         "the finished tiles in LDS either way; what `slimm` does without -co)")
    ap.add_argument("--record-order", default="grouped", choices=["grouped", "any"],
                    help="'any': the headline runs on the stream with the reads of every chunk INTERLEAVED at random (file order "
                         "kept inside a read) through a context created for record_order = SLIMM_ORDER_ANY (the device-side "
                         "grouping); --keep-file-order sends the grouped stream through that path instead (its scatter then "
                         "writes runs of a read's records to one place: 15 - 25 %% faster, and no input of interest)")
    ap.add_argument("--keep-file-order", action="store_true")
    ap.add_argument("--exchange", default="auto", choices=["auto", "summary", "sliced", "bins"],
                    help="multi-GPU exchange before the cut-offs: all-gather of sums + bin bitmaps (summary), all-to-all of "
                         "bitmap slices + small all-reduce (sliced), all-reduce of the bins; auto = summary up to 2 ranks")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process group of a multi-rank run: nccl = RCCL over xGMI, one GPU per rank (what the driver launches); "
                         "gloo = the rehearsal of the same control flow on ONE GPU -- every rank a process on cuda:(rank mod "
                         "devices), collectives staged through host memory (tests/test_distributed_gpu.py)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="run the multi-rank code path (process group, collectives) even with one rank")
    ap.add_argument("--kernel-timing", choices=("dominant", "all"), default="dominant",
                    help="HIP events in the timed steps: around the dominant kernel only (default) or around every launch")
    ap.add_argument("--breakdown", action="store_true", help="print the per-kernel tables to stderr")
    ap.add_argument("--engines", type=int, default=2, choices=[1, 2],
                    help="2 (default): the files of the timed loop go through two contexts in turn, and the host-only end of a "
                         "file -- propagation of the counts, profile text, the file written -- runs beside the device's front "
                         "end of the NEXT file on the other context (slimm_amd/distributed.py: FilesBackToBack); every file "
                         "through a freshly reset context, every profile written inside the timed region.  1: one context, "
                         "every call of a file behind the one before (rounds 1 - 3)")
    ap.add_argument("--gen-threads", type=int, default=0, help="threads generating chunks (default: the CPU quota / ranks)")
    # the legs beside the headline (N = 1 only)
    ap.add_argument("--quick", action="store_true", help="the headline measurement only (no other configs, push, CPU, CLI legs)")
    ap.add_argument("--roofline-configs", default="config2,config3,config5",
                    help="configurations measured beside the headline, resident, one context ('' = none)")
    ap.add_argument("--config-steps", type=int, default=5)
    ap.add_argument("--push-files", type=int, default=3, help="files of the pipelined push-inclusive measurement (0 = skip the leg)")
    ap.add_argument("--in-flight-files", type=int, default=20,
                    help="files of the two-files-in-flight leg (two host threads, each with two contexts of its own; 0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-any-order", action="store_true", help="skip the record_order = ANY legs (interleaved streams)")
    ap.add_argument("--cpu-sample", type=int, default=10_000_000, help="records the single-threaded CPU restatement is timed on")
    ap.add_argument("--cpu-mt-sample", type=int, default=100_000_000, help="records the all-core CPU restatement is timed on")
    ap.add_argument("--no-cli", action="store_true", help="skip the `slimm DB IN.bam` end-to-end leg")
    ap.add_argument("--cli-records", type=int, default=100_000_000)
    ap.add_argument("--cli-history", action="store_true",
                    help="also: the SAM file through --host-decode and the 17.7-fold file of rounds 1 - 4 (a minute more)")
    ap.add_argument("--no-cli-1b", action="store_true", help="skip the command on the 1 B-record BAM (cli_end_to_end.one_billion)")
    ap.add_argument("--cli-1b-records", type=int, default=1_000_000_000)
    args = ap.parse_args()

    # `python bench.py --gpus N` with N > 1 and no launcher around it: this process -- which has not touched a GPU and never
    # will -- starts `python -m torch.distributed.run` with N ranks as a CHILD (never an exec), lets rank 0's one JSON line
    # through on the inherited stdout and leaves with the child's exit code
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd).returncode)

    import torch
    import torch.distributed as dist

    from slimm_amd.distributed import FilesBackToBack, resolve_exchange, sharded_profile
    from slimm_amd.partition import chunk_owner
    from slimm_amd.profiler import Slimm
    from slimm_amd.synth import CONFIGS, make_workload, stream_chunks

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py needs a GPU (there is no CPU fallback for the hot path)", file=sys.stderr)
        sys.exit(2)
    if args.backend == "gloo":   # rehearsal: more ranks than devices share them
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    red_dev = dev if args.backend == "nccl" else torch.device("cpu")   # where this script's own small reductions live
    if world > 1 or args.force_exchange:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # (stdout carries ONE line: RCCL's version banner, printed at NCCL_DEBUG=VERSION / INFO, goes nowhere near it)
        os.environ["NCCL_DEBUG"] = os.environ.get("SLIMM_NCCL_DEBUG", "WARN")
        os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")   # (RCCL logs to stdout by default)
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    extras = rank == 0 and world == 1 and not args.quick and not args.force_exchange
    packed = args.form == "packed"
    rec_bytes = 16 if packed else 18
    out_path = os.path.join(tempfile.gettempdir(), f"slimm_bench_profile_{os.getpid()}.tsv")
    gen_threads = args.gen_threads or max(2, int(cpu_quota_cores()) // max(1, world))

    def to_packed(k, f):
        """slimm_pack_key on tensors: (key & (2^61 - 1)) | mate << 61 | unmapped << 63 (what a decoder writes)."""
        f = f.to(torch.int64) & 0xffff
        mate = torch.where((f & 0x40) != 0, 1, torch.where((f & 0x80) != 0, 2, 0)).to(torch.int64)
        return (k & ((1 << 61) - 1)) | (mate << 61) | (((f & 0x4) != 0).to(torch.int64) << 63)

    def interleave_index(k, seed):
        """For int64 name keys on the device: the gather index that interleaves the reads at random and keeps the relative
        order of the records of one name."""
        m = k.numel()
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        pos = torch.randperm(m, device=dev, generator=g)
        a = torch.sort(k, stable=True).indices              # by (name, index)
        p1 = torch.argsort(pos)
        b = p1[torch.sort(k[p1], stable=True).indices]      # by (name, pos)
        newpos = torch.empty(m, dtype=torch.int64, device=dev)
        newpos[a] = pos[b]
        del a, b, p1, pos
        inv = torch.empty(m, dtype=torch.int64, device=dev)
        inv[newpos] = torch.arange(m, device=dev)
        return inv

    class Resident:
        """A rank's records in HBM in the form handed to the library (+ optionally the same in page-locked host memory)."""

        def __init__(self, n, pinned=False):
            self.n = n
            self.key = torch.empty(n, dtype=torch.int64, device=dev)
            self.ref = torch.empty(n, dtype=torch.int32, device=dev)
            self.pos = torch.empty(n, dtype=torch.int32, device=dev)
            self.flag = None if packed else torch.empty(n, dtype=torch.int16, device=dev)
            self.host = None
            if pinned:
                self.host = [torch.empty(n, dtype=torch.int64).pin_memory(), torch.empty(n, dtype=torch.int32).pin_memory(),
                             torch.empty(n, dtype=torch.int32).pin_memory()]
                if not packed:
                    self.host.append(torch.empty(n, dtype=torch.int16).pin_memory())

        def fill(self, lo, rec):
            """The chunk goes to the device as it is; packing happens there (a decoder would write packed keys
            directly), and the page-locked host copy -- what value_with_push pushes -- is read back from the device."""
            hi = lo + len(rec)
            k = torch.from_numpy(rec.read_key.view(np.int64)).to(dev, non_blocking=True)
            f = torch.from_numpy(rec.flag.view(np.int16)).to(dev, non_blocking=True)
            self.key[lo:hi] = to_packed(k, f) if packed else k
            self.ref[lo:hi].copy_(torch.from_numpy(rec.ref_id), non_blocking=True)
            self.pos[lo:hi].copy_(torch.from_numpy(rec.begin_pos), non_blocking=True)
            if not packed:
                self.flag[lo:hi] = f
            if self.host is not None:
                for h, d in zip(self.host, [self.key, self.ref, self.pos] + ([] if packed else [self.flag])):
                    h[lo:hi].copy_(d[lo:hi], non_blocking=True)
            torch.cuda.synchronize()   # (the chunk's arrays may go away)
            return hi

        def interleave(self, chunk, seed):
            """In place, chunk by chunk (reads never span chunks): the reads interleaved at random, the file order of the
            records of one read name kept -- an input for record_order = SLIMM_ORDER_ANY that gives the grouped stream's
            results (src/read_stat.hpp:116-135 needs a read's records in file order, nothing else)."""
            ident_mask = (1 << 61) - 1 if packed else (1 << 62) - 1
            for c, lo in enumerate(range(0, self.n, chunk)):
                hi = min(self.n, lo + chunk)
                inv = interleave_index(self.key[lo:hi] & ident_mask, seed + c)
                for t in [self.key, self.ref, self.pos] + ([] if packed else [self.flag]):
                    t[lo:hi] = t[lo:hi][inv]
                del inv
            torch.cuda.synchronize()

        def give(self, eng):
            if packed:
                eng.set_records_device_packed(self.key, self.ref, self.pos)
            else:
                eng.set_records_device(self.key, self.ref, self.pos, self.flag)

        def push_async(self, eng):
            if packed:
                k, r, p = (h.numpy() for h in self.host)
                eng.push_records_packed_async(k.view(np.uint64), r, p)
            else:
                k, r, p, f = (h.numpy() for h in self.host)
                eng.push_records_async(k.view(np.uint64), r, p, f.view(np.uint16))

    # ------------------------------------------------------------------ the rank's share of the headline stream
    cfg = CONFIGS[args.config]
    t0 = time.time()
    n_stream = args.records or cfg.n_records
    want_push = extras and args.push_files > 0
    w_sample = None   # chunk 0's workload: header, database, options; rank 0 keeps its records for the CPU legs
    if args.weak:
        w = make_workload(cfg, seed=args.seed + 1000 * rank, n_records=n_stream, sample_seed=args.seed, shard=rank)
        res = Resident(len(w.records), pinned=want_push)
        res.fill(0, w.records)
        w_sample = w
        n_chunks = 1
    else:
        n_chunks = max(1, (n_stream + args.chunk_records - 1) // args.chunk_records)
        mine = list(chunk_owner(n_chunks, world)[rank])
        n_mine = sum(min(args.chunk_records, n_stream - c * args.chunk_records) for c in mine)
        res = Resident(n_mine, pinned=want_push)
        at = 0
        for c, wc in stream_chunks(cfg, args.seed, n_stream, args.chunk_records, chunks=mine, threads=gen_threads):
            at = res.fill(at, wc.records)
            if w_sample is None:
                w_sample = wc
        if w_sample is None:   # more ranks than chunks: this rank has no records, but takes part in the exchange
            w_sample = make_workload(cfg, seed=args.seed, n_records=1000, sample_seed=args.seed, shard=0)
        w = w_sample
    n_rec = res.n
    if args.record_order == "any" and not args.keep_file_order:
        res.interleave(args.chunk_records, 7000 + 100 * rank)
    gen_s = time.time() - t0

    def new_engine():
        e = Slimm.for_workload(w, device=local_rank, grouped=(args.record_order == "grouped"))
        e.force_exchange = args.force_exchange
        if args.no_bins:
            e.keep_bins(False)
        return e

    eng = new_engine()
    torch.cuda.synchronize()
    phase_times = {} if args.breakdown else None
    # the timed loop: files back to back, through two contexts in turn unless --engines 1
    eng2 = new_engine() if args.engines == 2 else None
    files = FilesBackToBack([eng] + ([eng2] if eng2 is not None else []), res.give, dev, out_path, phase_times=phase_times,
                            exchange=args.exchange)

    def step():
        if eng2 is not None:
            return files.step()
        eng.reset()
        eng.reset_cutoffs()            # every step is a fresh file for a fresh `slimm` object
        res.give(eng)
        return sharded_profile(eng, dev, out_path, phase_times=phase_times, exchange=args.exchange)

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    def measure(engine, one_step, steps, warmup, sync, flush=None):
        """`warmup` untimed steps with every launch bracketed by HIP events (that survey names the dominant kernel and
        gives the per-kernel table), then `steps` timed steps that bracket only the dominant kernel -- every event pair
        costs ~10 us of stream idle time, and 18 pairs per step would be charged to the value.  Returns (seconds,
        {kernel: (ms, launches)} scaled to `steps`, dominant kernel name, last profile)."""
        engine.enable_kernel_timing(True)
        engine.kernel_times(reset=True)
        survey_steps = warmup
        for i in range(warmup):
            one_step()
            if i == 0 and warmup > 1:   # the first step allocates and runs cold: keep it out of the survey
                if flush is not None:
                    flush()
                    if len(getattr(engine, "engines", ())) > 1:   # (every context's first file is a cold one)
                        one_step()
                        flush()
                engine.kernel_times(reset=True)
                survey_steps -= 1
        if flush is not None:
            flush()
        survey = engine.kernel_times(reset=True) if warmup > 0 else {}
        dom_name = None
        if survey and args.kernel_timing == "dominant":
            cand = {k: ms for k, (ms, n) in survey.items() if n and k not in ("memset_bins", "k_pick_runs")}
            if cand:
                dom_name = max(cand, key=cand.get)
                engine.time_only_kernel(dom_name)
        if phase_times is not None:
            phase_times.clear()
        sync()
        t1 = time.perf_counter()
        prof = None
        for _ in range(steps):
            prof = one_step()
        if flush is not None:   # (the last file's profile: written inside the timed region like every other)
            prof = flush()
        sync()
        el = time.perf_counter() - t1
        kt = engine.kernel_times(reset=True)
        engine.enable_kernel_timing(False)
        engine.time_only_kernel(None)
        if dom_name is not None:
            # the other kernels' figures (breakdown table, device_kernel_ms_per_step) come from the warm-up survey
            live = kt[dom_name]
            kt = {k: ((ms / survey_steps * steps), int(round(n / survey_steps * steps))) for k, (ms, n) in survey.items()}
            kt[dom_name] = live
        return el, kt, dom_name, prof

    if eng2 is not None:
        elapsed, ktimes, dom_name, profile = measure(files, step, args.steps, args.warmup, barrier, flush=files.flush)
        eng = files.last             # (both hold a whole file's results; the legs below go on with one context)
        (eng2 if eng is not eng2 else files.engines[0]).close()
    else:
        elapsed, ktimes, dom_name, profile = measure(eng, step, args.steps, args.warmup, barrier)
    if dist.is_initialized():
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    st = eng.stats()
    total_records = n_rec
    total_targets = int(st["n_targets"])   # (the one statistic that stays a rank's own: nothing downstream needs its sum)
    if dist.is_initialized():
        tot = torch.tensor([n_rec, total_targets], dtype=torch.int64, device=red_dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_records, total_targets = int(tot[0].item()), int(tot[1].item())
    ms_per_step = elapsed / args.steps * 1e3
    value = total_records / (elapsed / args.steps) / 1e6

    # ---- N > 1: where a step's time goes, and north_star's literal collective beside the default exchange
    step_split = exchange_bins = None
    if dist.is_initialized() and world > 1:
        import slimm_amd.distributed as sd

        def one(exchange):
            eng.reset()
            eng.reset_cutoffs()
            res.give(eng)
            return sharded_profile(eng, dev, out_path, exchange=exchange)

        # (a) the collectives of the default exchange bracketed by events on the engine's stream, a few untimed steps
        one(args.exchange)
        barrier()
        sd.COLLECTIVE_EVENTS = []
        n_split = 3
        for _ in range(n_split):
            one(args.exchange)
        barrier()
        coll = {}
        for name, nbytes, e0, e1 in sd.COLLECTIVE_EVENTS:
            a = coll.setdefault(name, [0.0, 0, 0])
            a[0] += e0.elapsed_time(e1) if e0 is not None else 0.0
            a[1] += 1
            a[2] = nbytes
        sd.COLLECTIVE_EVENTS = None
        coll_ms = sum(v[0] for v in coll.values()) / n_split
        cm = torch.tensor([coll_ms], dtype=torch.float64, device=red_dev)
        dist.all_reduce(cm, op=dist.ReduceOp.MAX)
        coll_ms = float(cm.item())
        step_split = {"collectives_ms": round(coll_ms, 4),
                      "collectives": {k: {"ms": round(v[0] / v[1], 4), "per_step": v[1] / n_split, "bytes": v[2]} for k, v in coll.items()},
                      "how": "events on the engine's stream around each collective, max over ranks, 3 untimed steps behind the timed region "
                             "(a rank that waits for a slower one inside a collective counts the wait as collective time)"}
        # (b) the same step with --exchange bins: ONE all-reduce over the integer bins before the uniqueness pass (north_star)
        if resolve_exchange(eng, args.exchange, world) != "bins" and not args.no_bins:
            one("bins")
            barrier()
            nb = max(3, args.steps // 4)
            t1 = time.perf_counter()
            for _ in range(nb):
                prof_b = one("bins")
            barrier()
            elb = time.perf_counter() - t1
            tb = torch.tensor([elb], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tb, op=dist.ReduceOp.MAX)
            elb = float(tb.item())
            exchange_bins = {"value": round(total_records / (elb / nb) / 1e6, 3), "unit": "M records/s", "ms_per_step": round(elb / nb * 1e3, 4),
                             "steps": nb, "engines": 1,
                             "collective": "ONE all_reduce(SUM, u32) over [cov | uniq_cov | 16 scalars] between phase A and the cut-offs "
                                           "(BASELINE.json north_star), then the small all-reduce of the partial results",
                             "all_reduce_bytes": int(eng.coverage_tensor().numel()) * 4,
                             "same_profile": bool(prof_b == profile)}

    def roofline_from(engine, stats, n_records, kt, steps, workload):
        """The roofline object of the dominant kernel + the per-kernel table, from HIP-event times `kt`
        ({kernel: (ms, launches)} over `steps` steps)."""
        if args.no_bins:
            Bp = int(stats["total_bins"])  # (the padded count is a property of the buffer, which is not exposed here)
        else:
            Bp = int(engine.coverage_buffer().__cuda_array_interface__["shape"][0] - 16) // 2
        # (with several ranks the merged statistics count the whole stream: this rank's kernels saw its own share)
        local = dict(stats)
        model = algorithmic_bytes(local, n_records, Bp, rec_bytes)
        per_kernel = {}
        for name, (ms, launches) in kt.items():
            if launches and name in model:
                steps_launches = launches / steps
                per_kernel[name] = {"ms_per_launch": ms / launches, "launches_per_step": steps_launches,
                                    "bytes_per_launch": model[name] / max(1.0, steps_launches if name == "memset_bins" else 1.0)}
        # the dominant kernel = most time per step (memsets are DMA fills, not kernels of this library)
        cand = {k: v for k, v in per_kernel.items() if k not in ("memset_bins", "k_pick_runs")}
        dom = max(cand, key=lambda k: cand[k]["ms_per_launch"] * cand[k]["launches_per_step"])
        d = cand[dom]
        achieved = d["bytes_per_launch"] / (d["ms_per_launch"] * 1e-3) / 1e9
        roof = {"kernel": dom, "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic_bytes(dom, workload),
                "traffic_source": "profiles/round6/pmc_traffic_summary.json (rocprofv3 --pmc, separate passes; 2 x FETCH_SIZE + "
                                  "WRITE_SIZE)",
                "bytes_per_launch": int(d["bytes_per_launch"]), "ms_per_launch": round(d["ms_per_launch"], 4)}
        if dom == "k_front":
            P = int(stats["n_targets"])
            roof["bytes_model"] = (f"{rec_bytes} B x {n_records} records + 8 B x {P} targets + slot descriptors "
                                   f"(SURVEY.md 8d: 16 N + 8 P = {16 * n_records + 8 * P}; the four-array form reads 18 N)")
        kernel_ms = sum(v["ms_per_launch"] * v["launches_per_step"] for v in per_kernel.values())
        return roof, per_kernel, kernel_ms

    def print_table(tag, per_kernel):
        for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1]["ms_per_launch"] * kv[1]["launches_per_step"]):
            gbs = v["bytes_per_launch"] / (v["ms_per_launch"] * 1e-3) / 1e9
            print(f"# {tag}{k:16s} {v['ms_per_launch']*1e3:9.1f} us/launch x{v['launches_per_step']:.0f}  "
                  f"{v['bytes_per_launch']/1e6:9.1f} MB  {gbs:8.1f} GB/s  {gbs/HBM_PEAK_GBS*100:5.1f}% of HBM peak", file=sys.stderr)

    def kernels_object(per_kernel):
        return {k: {"us": round(v["ms_per_launch"] * 1e3, 1),
                    "frac_of_hbm_peak": round(v["bytes_per_launch"] / (v["ms_per_launch"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 3)}
                for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1]["ms_per_launch"] * kv[1]["launches_per_step"])
                if v["ms_per_launch"] > 0}

    def any_order_leg(wk, resk, nk, steps, grouped_profile, workload_name, what):
        """The records of `resk` (already interleaved) through a context created for record_order = SLIMM_ORDER_ANY: the
        step, the grouping kernels (group_by_ident.hip) under their own byte model, and whether the profile is the grouped
        stream's."""
        torch.cuda.empty_cache()   # (the interleave's temporaries go back to the device before the engine takes its 32 B/record)
        enga = Slimm.for_workload(wk, device=local_rank, grouped=False)
        if args.no_bins:
            enga.keep_bins(False)

        def stepa():
            enga.reset()
            enga.reset_cutoffs()
            resk.give(enga)
            return enga.get_profiles(path=out_path)

        ela, kta, _, profa = measure(enga, stepa, steps, 2, torch.cuda.synchronize)
        sta = enga.stats()
        _, pka, kmsa = roofline_from(enga, sta, nk, kta, steps, workload_name)
        grp = {k: v for k, v in pka.items() if k.startswith("k_group")}
        g_ms = sum(v["ms_per_launch"] * v["launches_per_step"] for v in grp.values())
        g_bytes = sum(v["bytes_per_launch"] * v["launches_per_step"] for v in grp.values())
        passes, width, bits, grid = group_plan(nk)
        out = {"value": round(nk / (ela / steps) / 1e6, 3), "unit": "M records/s", "ms_per_step": round(ela / steps * 1e3, 4),
               "steps": steps, "records": what, "same_profile_as_the_grouped_stream": bool(profa == grouped_profile),
               "plan": {"passes": passes, "widest_digit_bits": width, "bucket_bits": bits, "workgroups": grid},
               "grouping": {"kernels": "k_group_count + k_group_scan + k_group_scatter (per pass) + k_group_finish",
                            "ms_per_step": round(g_ms, 4), "bytes_per_step": int(g_bytes),
                            "bytes_model": f"first pass {rec_bytes - 4} N (count) + {rec_bytes} N + 16 V (scatter), every other pass "
                                           "8 V + 32 V, finish 8 V",
                            "bound": "hbm", "achieved": round(g_bytes / (g_ms * 1e-3) / 1e9, 2) if g_ms else None, "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": round(g_bytes / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if g_ms else None},
               "device_kernel_ms_per_step": round(kmsa, 4), "kernels": kernels_object(pka)}
        if args.breakdown:
            print(f"# any order, {workload_name}: {nk} records, {ela / steps * 1e3:.3f} ms/step", file=sys.stderr)
            print_table("any ", pka)
        enga.close()
        return out

    if rank == 0:
        # (N > 1: the statistics are the merged ones; the byte model of this rank's kernels takes its share of them)
        st_local = dict(st)
        if world > 1:
            for k in ("hits_count", "matches_count", "uniq_matches_count", "uniq_matches_count2"):
                st_local[k] = int(st[k]) // world   # (n_targets is this rank's own already)
        roofline, per_kernel, kernel_ms = roofline_from(eng, st_local, n_rec, ktimes, args.steps, args.config)
        if step_split is not None:
            step_split = dict({"ms_per_step": round(ms_per_step, 4), "kernels_ms": round(kernel_ms, 4),
                               "host_and_idle_ms": round(ms_per_step - kernel_ms - step_split["collectives_ms"], 4)}, **step_split)
        if args.breakdown:
            print(f"# generate + copy {gen_s:.1f}s on {gen_threads} threads; records/rank {n_rec}; V={st['hits_count']} "
                  f"M={st['matches_count']} P={st['n_targets']} U={st['uniq_matches_count']} U2={st['uniq_matches_count2']} "
                  f"valid={st['n_valid']} B={st['total_bins']}", file=sys.stderr)
            print_table("", per_kernel)
            print(f"# device kernels {kernel_ms:.3f} ms of {ms_per_step:.3f} ms per step", file=sys.stderr)
            for k, v in phase_times.items():
                print(f"# host wall {k:28s} {v / args.steps * 1e6:9.1f} us/step", file=sys.stderr)

        # ---- two files in flight: two host threads, each working through its files back to back on two contexts of its own,
        # so that the kernels of two files share the device (the front end waits for the scalar unit, the bucketing for the
        # LDS, the histograms for memory).  Reported BESIDE `value`, which keeps one file at a time.
        in_flight = None
        if extras and world == 1 and args.in_flight_files >= 4 and args.record_order == "grouped" and args.engines == 2:
            import threading
            lanes_ = [FilesBackToBack([new_engine(), new_engine()], res.give, dev, out_path + f".t{i}") for i in range(2)]
            def work(fb, k):
                for _ in range(k):
                    fb.step()
                fb.flush()
            for fb in lanes_:
                work(fb, 3)          # (every context's first file allocates)
            per = args.in_flight_files // 2
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            ts = [threading.Thread(target=work, args=(fb, per)) for fb in lanes_]
            [t.start() for t in ts]
            [t.join() for t in ts]
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / (2 * per)
            for fb in lanes_:
                [e.close() for e in fb.engines]
            in_flight = {"value": round(n_rec / dt / 1e6, 3), "unit": "M records/s", "ms_per_file": round(dt * 1e3, 4),
                         "files": 2 * per,
                         "what": "TWO files in flight: two host threads, each taking its files back to back through two contexts "
                                 "of its own (four contexts, four streams); every file a freshly reset object, every profile "
                                 "written inside the timed region; not the headline -- `value` keeps one file at a time"}
            if args.breakdown:
                print(f"# two files in flight: {dt * 1e3:.3f} ms per file = {n_rec / dt / 1e6:.0f} M records/s", file=sys.stderr)

        # ---- push-inclusive rate: the clock starts before the first record leaves host
        # memory and stops when the profile file is written.  PCIe, not the path, bounds it.
        with_push = None
        if want_push:
            eng.reset()
            eng.reset_cutoffs()
            res.push_async(eng)              # (the first pass sizes the library's own record buffers)
            eng.get_profiles(path=out_path)
            torch.cuda.synchronize()
            single = []
            for _ in range(2):               # ONE file: push, then the path
                eng.reset()
                eng.reset_cutoffs()
                t1 = time.perf_counter()
                res.push_async(eng)
                eng.get_profiles(path=out_path)
                single.append(time.perf_counter() - t1)
            dt_single = min(single)
            # files back to back (the -d directory mode of the command): two contexts alternate, a file's copy runs on the
            # copy stream while the file before is profiled
            other = Slimm.for_workload(w, device=local_rank, grouped=True)
            if args.no_bins:
                other.keep_bins(False)
            other.reset()
            res.push_async(other)
            other.get_profiles(path=out_path)
            torch.cuda.synchronize()
            engs = [eng, other]
            n_files = 2 * args.push_files
            engs[0].reset()
            engs[0].reset_cutoffs()
            t1 = time.perf_counter()
            res.push_async(engs[0])
            for i in range(n_files):
                cur, nxt = engs[i & 1], engs[(i + 1) & 1]
                if i + 1 < n_files:
                    nxt.reset()
                    nxt.reset_cutoffs()
                    res.push_async(nxt)
                cur.get_profiles(path=out_path)
            dt_pipe = (time.perf_counter() - t1) / n_files
            other.close()
            with_push = {"value": round(n_rec / dt_single / 1e6, 3), "unit": "M records/s", "ms_per_step": round(dt_single * 1e3, 3),
                         "what": f"ONE file of {n_rec} records: clock from before the first slimm_push_records"
                                 f"{'_packed' if packed else ''}_async (records in page-locked host memory, {rec_bytes} B/record "
                                 "over PCIe) to the profile file written; best of 2",
                         "gb_per_s_over_pcie": round(rec_bytes * n_rec / dt_single / 1e9, 2),
                         "files_back_to_back": {"value": round(n_rec / dt_pipe / 1e6, 3), "unit": "M records/s",
                                                "ms_per_file": round(dt_pipe * 1e3, 3),
                                                "what": f"{n_files} files, two contexts alternating: a file's copy overlaps the "
                                                        "phases of the file before (the command's -d mode)"}}
        # ---- the same stream as RUN-MARKED records (8 B/record: reference + 1 | mate | "a qName run starts here", position):
        # what a producer of name-grouped input can hand over instead of hashed names (include/slimm_hip.h).  The words are
        # built on the device from the packed arrays; resident rate, k_front under 8 N + 8 P, and the push-inclusive rate.
        marked = None
        if extras and packed and args.record_order == "grouped" and not args.no_marked:
            word = torch.empty(n_rec, dtype=torch.int32, device=dev)
            piece = 50_000_000
            prev = None
            for a in range(0, n_rec, piece):
                b = min(n_rec, a + piece)
                k = res.key[a:b]
                ident = k & ((1 << 61) - 1)
                starts = torch.ones(b - a, dtype=torch.bool, device=dev)
                starts[1:] = ident[1:] != ident[:-1]
                if prev is not None:
                    starts[0] = bool(ident[0] != prev)
                prev = ident[-1].clone()
                r1 = torch.where((k < 0) | (res.ref[a:b] < 0), 0, res.ref[a:b].to(torch.int64) + 1)
                word[a:b] = (r1 | (((k >> 61) & 3) << 29) | (starts.to(torch.int64) << 31)).to(torch.int32)
                del k, ident, starts, r1
            torch.cuda.synchronize()

            def step_marked():
                eng.reset()
                eng.reset_cutoffs()
                eng.set_records_device_marked(word, res.pos)
                return sharded_profile(eng, dev, out_path, exchange=args.exchange)

            m_steps = max(3, args.steps // 2)
            el_m, kt_m, _, prof_m = measure(eng, step_marked, m_steps, 2, barrier)
            assert prof_m == profile, "the run-marked records must give the packed records' profile"
            kf_ms = kt_m["k_front"][0] / max(1, kt_m["k_front"][1])
            P = int(st["n_targets"])
            bytes_m = 8 * n_rec + 8 * P
            marked = {"value": round(n_rec / (el_m / m_steps) / 1e6, 3), "unit": "M records/s", "ms_per_step": round(el_m / m_steps * 1e3, 4),
                      "steps": m_steps, "records": "resident in HBM, run-marked form (8 B/record); same profile as the headline's",
                      "k_front": {"ms_per_launch": round(kf_ms, 4), "bytes_per_launch": bytes_m,
                                  "bytes_model": f"8 B x {n_rec} records + 8 B x {P} targets",
                                  "achieved": round(bytes_m / (kf_ms * 1e-3) / 1e9, 2), "unit": "GB/s",
                                  "frac": round(bytes_m / (kf_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}
            if want_push:
                hw = torch.empty(n_rec, dtype=torch.int32).pin_memory()
                hw.copy_(word)
                torch.cuda.synchronize()
                hwn, hpn = hw.numpy().view(np.uint32), res.host[2].numpy()
                single = []
                for i in range(3):               # (the first pass warms the library's buffers up)
                    eng.reset()
                    eng.reset_cutoffs()
                    t1 = time.perf_counter()
                    eng.push_records_marked_async(hwn, hpn)
                    eng.get_profiles(path=out_path)
                    if i:
                        single.append(time.perf_counter() - t1)
                dt = min(single)
                marked["with_push"] = {"value": round(n_rec / dt / 1e6, 3), "unit": "M records/s", "ms_per_step": round(dt * 1e3, 3),
                                       "gb_per_s_over_pcie": round(8 * n_rec / dt / 1e9, 2),
                                       "what": "ONE file: clock from before slimm_push_records_marked_async (page-locked host "
                                               "memory, 8 B/record over PCIe) to the profile file written; best of 2"}
                # ... and files back to back on two alternating contexts, as for the packed form
                other = Slimm.for_workload(w, device=local_rank, grouped=True)
                if args.no_bins:
                    other.keep_bins(False)
                other.push_records_marked_async(hwn, hpn)
                other.get_profiles(path=out_path)
                torch.cuda.synchronize()
                engs = [eng, other]
                n_files = 2 * args.push_files
                engs[0].reset()
                engs[0].reset_cutoffs()
                t1 = time.perf_counter()
                engs[0].push_records_marked_async(hwn, hpn)
                for i in range(n_files):
                    cur, nxt = engs[i & 1], engs[(i + 1) & 1]
                    if i + 1 < n_files:
                        nxt.reset()
                        nxt.reset_cutoffs()
                        nxt.push_records_marked_async(hwn, hpn)
                    cur.get_profiles(path=out_path)
                dt_pipe = (time.perf_counter() - t1) / n_files
                other.close()
                marked["with_push"]["files_back_to_back"] = {"value": round(n_rec / dt_pipe / 1e6, 3), "unit": "M records/s",
                                                             "ms_per_file": round(dt_pipe * 1e3, 3),
                                                             "what": f"{n_files} files, two contexts alternating"}
                del hw
            del word
        eng.close()
        # ---- record_order = ANY: the SAME stream with the reads of every chunk interleaved at random (the reference takes
        # records in any order: src/slimm.hpp:204-211), through the device-side grouping
        any_order = {}
        if extras and not args.no_any_order and args.record_order == "grouped":
            res.interleave(args.chunk_records, 7000)
            any_order[args.config] = any_order_leg(w, res, n_rec, max(3, args.steps // 4), profile, args.config,
                                                   f"the headline stream, every chunk of {args.chunk_records} records interleaved "
                                                   f"read by read at random (file order kept inside a read), resident, {args.form} form")
        del res
        torch.cuda.empty_cache()

        # ---- the other single-GPU configurations of BASELINE.json, resident, one context each
        legs = {}
        w_cli = None
        parity_in_run = None
        for name in ([c for c in args.roofline_configs.split(",") if c] if extras else []):
            if name == args.config or name not in CONFIGS:
                continue
            cfgk = CONFIGS[name]
            wk = make_workload(cfgk, seed=args.seed)
            nk = len(wk.records)
            def new_engk():
                e = Slimm.for_workload(wk, device=local_rank, grouped=True)
                if args.no_bins:
                    e.keep_bins(False)
                return e

            engk = new_engk()
            resk = Resident(nk)
            resk.fill(0, wk.records)

            def stepk():
                engk.reset()
                engk.reset_cutoffs()
                resk.give(engk)
                return engk.get_profiles(path=out_path)

            if args.engines == 2:
                filesk = FilesBackToBack([engk, new_engk()], resk.give, dev, out_path)
                elk, ktk, _, _ = measure(filesk, filesk.step, args.config_steps, 2, torch.cuda.synchronize, flush=filesk.flush)
                engk = filesk.last
                [e.close() for e in filesk.engines if e is not engk]
            else:
                elk, ktk, _, _ = measure(engk, stepk, args.config_steps, 2, torch.cuda.synchronize)
            stk = engk.stats()
            roofk, pkk, kmsk = roofline_from(engk, stk, nk, ktk, args.config_steps, name)
            roofk.pop("traffic_source", None)
            roofk.update({"workload": f"BASELINE.json configs[{list(CONFIGS).index(name)}] ({name}): {nk} records, {cfgk.n_refs} refs, "
                                      f"mean {cfgk.mean_hits} hits/read, {cfgk.bin_width} bp bins; records resident in HBM ({args.form})",
                          "steps": args.config_steps, "ms_per_step": round(elk / args.config_steps * 1e3, 4),
                          "value": round(nk / (elk / args.config_steps) / 1e6, 3), "unit_value": "M records/s",
                          "device_kernel_ms_per_step": round(kmsk, 4), "reads": stk["matches_count"],
                          "targets": stk["n_targets"], "bins": stk["total_bins"], "kernels": kernels_object(pkk)})
            legs[name] = roofk
            if args.breakdown:
                print(f"# {name}: {nk} records, {elk / args.config_steps * 1e3:.3f} ms/step", file=sys.stderr)
                print_table(name[-2:] + " ", pkk)
            profk = engk.write_abundance()
            if name == "config3" and not args.no_cpu_baseline:
                # the GPU result against the CPU restatement IN THE SAME RUN, on the same 100 M records
                # (the identity a packed record carries is the key's low 61 bits, so the CPU side gets those)
                from oracle.binding import bin_checksum, dense_mt_run
                from slimm_amd.workload import Records
                rk = wk.records
                rec61 = Records(rk.read_key & np.uint64((1 << 61) - 1), rk.flag, rk.ref_id, rk.begin_pos) if packed else rk
                t1 = time.perf_counter()
                dmt = dense_mt_run(wk, records=rec61)
                rc = engk.ref_columns()
                checks = {"scalars": (stk["hits_count"], stk["matches_count"], stk["uniq_matches_count"], stk["uniq_matches_count2"],
                                      stk["n_valid"]) == (dmt["hits"] % 2**32, dmt["matches"] % 2**32, dmt["uniq_matches"] % 2**32,
                                                          dmt["uniq_matches2"] % 2**32, dmt["n_valid"]),
                          "per_reference_columns": all(bool(np.array_equal(rc[c], dmt[c])) for c in
                                                       ("reads_count", "uniq_reads_count", "uniq_reads_count2", "nz_cov", "nz_uniq_cov")),
                          "lca_direct": engk.taxon_counts(0) == dmt["lca_direct"]}
                if not args.no_bins:
                    checks["bin_checksums"] = all(bin_checksum(engk.bins(i)) == dmt["checksums"][i] for i in range(3))
                parity_in_run = {"ok": all(checks.values()), "checks": checks, "workload": name,
                                 "against": f"oracle/slimm_dense_mt.cpp on the same {nk} records, {dmt['threads']} threads "
                                            f"({sum(dmt['seconds']):.2f} s; {time.perf_counter() - t1:.1f} s with the comparisons)"}
                del rec61
            if name == "config3" and not args.no_any_order:
                resk.interleave(nk, 7100)
                any_order[name] = any_order_leg(wk, resk, nk, args.config_steps, profk, name,
                                                f"{name}'s {nk} records interleaved read by read at random (file order kept inside "
                                                f"a read), resident, {args.form} form")
            engk.close()
            del resk
            torch.cuda.empty_cache()
            if name == "config3" and not args.no_cli:
                w_cli = wk

        cpu = cpu_mt = None
        if extras and not args.no_cpu_baseline:
            from oracle.binding import Oracle, dense_mt_run  # the CPU restatements (checker / baseline only)
            from slimm_amd.workload import Records

            rec0 = w_sample.records
            ns = min(args.cpu_sample, len(rec0))
            sample = rec0.take(slice(0, ns))
            orc = Oracle(w.taxonomy, w.options)
            t1 = time.perf_counter()
            o = orc.run(w.ref_names, w.ref_len, sample, w.avg_read_len, want_raw=False, want_cov=False, use_qnames=False,
                        collect_bins=False)
            wall = time.perf_counter() - t1
            cpu_s = sum(o.phase_seconds)  # the three phases + profile, excluding reference/bin allocation
            cpu = {"value": round(ns / cpu_s / 1e6, 4), "unit": "M records/s", "cores": 1, "kind": "port",
                   "sample": f"one pass over the first {ns} records of the stream (chunk 0; same refs/DB), phases A+B+C+profile "
                             f"{cpu_s:.1f}s of {wall:.1f}s wall",
                   "host": f"{os.cpu_count()} logical cores, cgroup quota {cpu_quota_cores()} cores"}
            # ... and all cores of the same host: the dense multi-threaded restatement of phases A, B and the per-read LCA
            # (oracle/slimm_dense_mt.cpp; equal to the oracle in tests/test_dense_mt.py and to the GPU at full size in the
            # -m gpu suite).  The reference itself is single-threaded: a reported baseline, never a code path of the product.
            n_mt = min(args.cpu_mt_sample, n_stream)
            if args.weak:
                parts = [rec0]
            else:
                parts = [wc.records for _, wc in stream_chunks(cfg, args.seed, n_stream, args.chunk_records,
                                                               chunks=range((n_mt + args.chunk_records - 1) // args.chunk_records),
                                                               threads=gen_threads)]
            recm = Records(*(np.concatenate([getattr(p, f) for p in parts]) for f in ("read_key", "flag", "ref_id", "begin_pos")))
            del parts
            ncpu = os.cpu_count() or 1
            best = threads = None
            t_all = time.perf_counter()
            for th in sorted({max(1, ncpu // 2), max(1, ncpu // 4), min(ncpu, 32)}):
                d = dense_mt_run(w, records=recm, threads=th)
                sec = sum(d["seconds"])
                if best is None or sec < best:
                    best, threads = sec, th
            cpu_mt = {"value": round(len(recm) / best / 1e6, 3), "unit": "M records/s", "cores": threads, "kind": "port",
                      "sample": f"best pass over the first {len(recm)} records of the stream, {threads} threads (of {ncpu // 2}, "
                                f"{ncpu // 4}, 32 on {ncpu} logical cores): phases A + B + per-read LCA {best * 1e3:.1f} ms (array "
                                f"allocation and the scalar profile tail excluded), {time.perf_counter() - t_all:.1f} s wall for all passes",
                      "cpu_quota_cores": cpu_quota_cores()}
            del recm

        # ---- the command line, process start to profile written, on a BAM large enough that HIP start-up amortises
        cli = None
        if extras and not args.no_cli:
            from slimm_amd.synth_bam import write_synthetic_bam
            from tests.bam_io import write_sldb

            if w_cli is None:
                w_cli = make_workload(CONFIGS["config3"], seed=args.seed)
            nb = min(args.cli_records, len(w_cli.records))
            tmp = tempfile.mkdtemp(prefix="slimm_bench_cli_")
            bam = os.path.join(tmp, "sample.bam")
            recb = w_cli.records if nb == len(w_cli.records) else w_cli.records.take(slice(0, nb))
            info = write_synthetic_bam(bam, w_cli.ref_names, w_cli.ref_len, recb, read_len=w_cli.avg_read_len, realistic=True)
            db = os.path.join(tmp, "db.sldb")
            write_sldb(db, w_cli.taxonomy)
            os.makedirs(os.path.join(tmp, "out"))
            def run_cli(path, stem, flags=()):
                runs, r = [], None
                for _ in range(2):
                    t1 = time.perf_counter()
                    r = subprocess.run([os.path.join(ROOT, "slimm_amd", "slimm"), *flags, "-w", "1000", "-o", os.path.join(tmp, "out") + "/",
                                        db, path], capture_output=True, text=True, env=dict(os.environ, SLIMM_TRACE="cli"))
                    runs.append(time.perf_counter() - t1)
                    if r.returncode != 0:
                        return None, r, None
                prof = open(os.path.join(tmp, "out", stem + "_profile.tsv")).read()
                return min(runs), r, prof

            def traces(r):
                tr = [ln[ln.index("[trace]") + 8:] for ln in r.stderr.splitlines() if "[trace] reader" in ln][-1:]
                dd = [ln[ln.index("[trace]") + 8:] for ln in r.stderr.splitlines() if "[trace] device decode" in ln][-1:]
                su = [ln[ln.index("[trace]") + 8:] for ln in r.stderr.splitlines() if "slimm_create" in ln][-1:]
                return (tr[0] if tr else None), (dd[0] if dd else None), (su[0] if su else None)

            best, r, prof_g = run_cli(bam, "sample")
            if best is not None:
                tr, dd, su = traces(r)
                ratio = info["raw_bytes"] / info["compressed_bytes"]
                cli = {"value": round(nb / best / 1e6, 3), "unit": "M records/s", "seconds": round(best, 3),
                       "what": f"`slimm -w 1000 DB IN.bam`, process start to profile written, on a BAM that compresses like a BAM: {nb} "
                               f"records of config3, name-grouped, random bases, binned qualities with runs, instrument-style names -- "
                               f"{info['raw_bytes'] / 1e9:.1f} GB of BAM records in {info['compressed_bytes'] / 1e9:.2f} GB = {ratio:.2f} x "
                               f"({info['deflate']}).  HIP start-up, read-length sample, the file's BGZF blocks read by pread and handed over "
                               f"COMPRESSED (slimm_push_bgzf_blocks), inflated on the device in two phases (bgzf_tokens.hip), record "
                               f"boundaries + fields + adjacent-name comparison on the device, GPU path; best of 2",
                       "compression_ratio": round(ratio, 3),
                       "bounds": {"pcie_s": round(info["compressed_bytes"] / 54e9, 3), "what": "compressed bytes / 54 GB/s; HIP start-up "
                                  "(0.1 - 0.3 s, in `start_up`) runs beside the first reads"},
                       "reader": tr, "device_decode": dd, "start_up": su, "bam_built_in_s": round(info["seconds"], 1)}
                # the same file with every window inflated by the host cores (rounds 1 - 4's path)
                best_h, rh, prof_h = run_cli(bam, "sample", flags=("--device-inflate", "0"))
                if best_h is not None:
                    _, ddh, _ = traces(rh)
                    cli["host_inflate"] = {"value": round(nb / best_h / 1e6, 3), "unit": "M records/s", "seconds": round(best_h, 3),
                                           "what": "--device-inflate 0: BGZF inflate by libdeflate on the host cores, inflated "
                                                   "windows over PCIe", "same_profile": bool(prof_h == prof_g), "device_decode": ddh}
                # the same records in NO particular order (header without GO:query): name hash + check word on the device,
                # then the device-side grouping of record_order = ANY
                try:
                    os.unlink(bam)
                    kk = torch.from_numpy(recb.read_key.view(np.int64)).to(dev)
                    inv = interleave_index(kk, 7200).cpu().numpy()
                    del kk
                    torch.cuda.empty_cache()
                    recu = recb.take(inv)
                    del inv
                    bam_u = os.path.join(tmp, "unsorted.bam")
                    info_u = write_synthetic_bam(bam_u, w_cli.ref_names, w_cli.ref_len, recu, read_len=w_cli.avg_read_len,
                                                 hd="@HD\tVN:1.6\tSO:unsorted", realistic=True)
                    del recu
                    best_u, ru, prof_u = run_cli(bam_u, "unsorted")
                    if best_u is not None:
                        _, ddu, _ = traces(ru)
                        cli["unsorted_file"] = {"value": round(nb / best_u / 1e6, 3), "unit": "M records/s", "seconds": round(best_u, 3),
                                                "what": "the same records with the reads interleaved at random, @HD SO:unsorted: key + "
                                                        "check word hashed from the names on the device, then the grouping of "
                                                        "record_order = ANY; best of 2",
                                                "compression_ratio": round(info_u["raw_bytes"] / info_u["compressed_bytes"], 3),
                                                "same_profile_as_the_grouped_file": bool(prof_u == prof_g),
                                                "device_decode": ddu,
                                                "bam_built_in_s": round(info_u["seconds"], 1)}
                    else:
                        cli["unsorted_file"] = {"error": ru.stderr[-400:]}
                    os.unlink(bam_u)
                except Exception as e:   # (the leg is a report, not a gate)
                    cli["unsorted_file"] = {"error": repr(e)[:300]}
                # the same records as SAM TEXT (the reference takes .sam and .bam alike, src/file_helper.hpp:73-75): lines found
                # and parsed on the device (slimm_push_sam_bytes), against the host decoder
                try:
                    from slimm_amd.synth_bam import write_synthetic_sam
                    sam = os.path.join(tmp, "sample_text.sam")
                    info_s = write_synthetic_sam(sam, w_cli.ref_names, w_cli.ref_len, recb, read_len=w_cli.avg_read_len)
                    best_s, rs, prof_s = run_cli(sam, "sample_text")
                    if best_s is not None:
                        _, dds, _ = traces(rs)
                        cli["sam_text"] = {"value": round(nb / best_s / 1e6, 3), "unit": "M records/s", "seconds": round(best_s, 3),
                                           "text_bytes": info_s["bytes"], "text_GB_s": round(info_s["bytes"] / best_s / 1e9, 2),
                                           "what": "`slimm DB IN.sam`: the file's text read by pread and handed over as it is "
                                                   "(slimm_push_sam_bytes), lines found and the four fields parsed on the device "
                                                   "(sam_decode.hip); bound by PCIe: text bytes / 54 GB/s = "
                                                   f"{info_s['bytes'] / 54e9:.2f} s; best of 2",
                                           "same_profile_as_the_bam": bool(prof_s == prof_g), "device_decode": dds,
                                           "sam_built_in_s": round(info_s["seconds"], 1)}
                        t1 = time.perf_counter()
                        rh2 = subprocess.run([os.path.join(ROOT, "slimm_amd", "slimm"), "--host-decode", "-w", "1000", "-o", os.path.join(tmp, "out") + "/", db, sam],
                                             capture_output=True, text=True) if args.cli_history else None
                        if rh2 is not None and rh2.returncode == 0:
                            sec_h = time.perf_counter() - t1
                            cli["sam_text"]["host_decoder"] = {"value": round(nb / sec_h / 1e6, 3), "seconds": round(sec_h, 3),
                                                               "what": "--host-decode (rounds 1 - 4's SAM path), one run"}
                    else:
                        cli["sam_text"] = {"error": rs.stderr[-400:]}
                    os.unlink(sam)
                except Exception as e:
                    cli["sam_text"] = {"error": repr(e)[:300]}
                # rounds 1 - 4's file beside it: every sequence byte 0x11, every quality 0x28 -- 17.7-fold
                try:
                    if not args.cli_history:
                        raise StopIteration
                    bam_e = os.path.join(tmp, "easy.bam")
                    info_e = write_synthetic_bam(bam_e, w_cli.ref_names, w_cli.ref_len, recb, read_len=w_cli.avg_read_len)
                    best_e, re_, prof_e = run_cli(bam_e, "easy")
                    if best_e is not None:
                        _, dde, _ = traces(re_)
                        cli["easy_file"] = {"value": round(nb / best_e / 1e6, 3), "unit": "M records/s", "seconds": round(best_e, 3),
                                            "compression_ratio": round(info_e["raw_bytes"] / info_e["compressed_bytes"], 3),
                                            "what": "the file of rounds 1 - 4 (constant sequences and qualities, hex names)",
                                            "same_profile": bool(prof_e == prof_g), "device_decode": dde,
                                            "bam_built_in_s": round(info_e["seconds"], 1)}
                    os.unlink(bam_e)
                except StopIteration:
                    pass
                except Exception as e:
                    cli["easy_file"] = {"error": repr(e)[:300]}
            else:
                cli = {"error": r.stderr[-400:]}
            try:
                os.unlink(bam)
            except OSError:
                pass
            shutil.rmtree(tmp, ignore_errors=True)
            # the command on the 1 B-record BAM north_star names (BASELINE.json configs[3]; ~78 GB, built once per run in
            # /dev/shm or /tmp: ~5 min): scripts/cli_1B.py in a process of its own, skipped -- with the reason -- when there is no
            # room for the file or the memory for it
            if isinstance(cli, dict) and "error" not in cli and not args.no_cli_1b:
                cli["one_billion"] = cli_one_billion(args.cli_1b_records)

        line = {
            "metric": "M alignment-records/sec -> final profile",
            "value": round(value, 3), "unit": "M records/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak" if args.weak else "strong",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"BASELINE.json configs[{list(CONFIGS).index(args.config)}] ({args.config}): one seeded stream of "
                                   f"{total_records} records over {world} GPU(s), {cfg.n_refs} refs, mean {cfg.mean_hits} hits/read, "
                                   f"{cfg.bin_width} bp bins, {cfg.read_len} bp reads",
                       "records_per_gpu": n_rec, "total_records": total_records, "refs": cfg.n_refs,
                       "reads": st["matches_count"], "targets": total_targets, "bins": st["total_bins"],
                       "record_order": args.record_order, "coverage_arrays": "not materialised" if args.no_bins else "in HBM",
                       "records": f"resident in HBM before the timed region, {args.form} form ({rec_bytes} B/record); "
                                  "value_with_push starts in host memory",
                       "seed": args.seed, "parallelism": f"reads sharded over {world} GPU(s)",
                       "stream": ("every rank its own shard of the sample" if args.weak else
                                  f"{n_chunks} chunks of {args.chunk_records} records, contiguous chunk ranges per rank"),
                       "exchange": (resolve_exchange(eng, args.exchange, world) if (world > 1 or args.force_exchange) else "none"),
                       "rccl_ranks": world if (dist.is_initialized() and args.backend == "nccl") else 0,
                       "process_group_ranks": dist.get_world_size() if dist.is_initialized() else 0,
                       "backend": args.backend if dist.is_initialized() else "none",
                       "files_in_flight": ("2 contexts in turn: the host-only end of file k (propagation, profile text, file written) "
                                           "runs beside the device's front end of file k + 1; every file through a freshly reset "
                                           "context, all profiles written inside the timed region") if args.engines == 2 else
                                          "1 context, every call of a file behind the one before",
                       "profile_sha1": __import__("hashlib").sha1((profile or "").encode()).hexdigest(),
                       "profile_rows": len(profile.strip().split("\n")) - 1 if profile else 0},
            "roofline": roofline,
            "step_split": step_split,
            "exchange_bins": exchange_bins,
            "scaling_note": (None if world == 1 else
                             "strong scaling of ONE 1 B-record stream: a rank's kernels take (11.7 ms / N) per step, while two collectives, "
                             "two host synchronisations and the host's cut-offs between the phases (0.13 ms + the collectives' latency) stay: "
                             "above N = 4 the step is mostly the latter (step_split says how much), so the curve flattens by "
                             "construction -- the work per step is 12 ms of one GPU"),
            "value_resident": round(value, 3),
            "two_files_in_flight": in_flight,
            "value_with_push": with_push,
            "run_marked_records": marked,
            "record_order_any": any_order or None,
            "parity_in_run": parity_in_run["ok"] if parity_in_run else None,
            "parity_in_run_detail": parity_in_run,
            "roofline_config2": legs.get("config2"), "roofline_config3": legs.get("config3"), "roofline_config5": legs.get("config5"),
            "cpu_baseline": cpu,
            "cpu_baseline_mt": cpu_mt,
            "cli_end_to_end": cli,
            "device_kernel_ms_per_step": round(kernel_ms, 4),
            # every kernel of the step from the warm-up survey (events around every launch), longest first
            "kernels": kernels_object(per_kernel),
            "kernel_timing": ("HIP events handed to the launch of " + (f"{dom_name} only in the timed steps (other kernels: warm-up "
                              "survey)" if dom_name else "every kernel in the timed steps")),
        }
        # the figures a reader looks for first, once more at the END of the line (logs that keep only a line's tail keep these)
        line["summary"] = {
            "value_M_records_s": line["value"], "ms_per_step": line["ms_per_step"], "roofline_frac": roofline["frac"],
            "roofline_kernel": roofline["kernel"], "n_gpus": world,
            "config2_ms": (legs.get("config2") or {}).get("ms_per_step"), "config2_frac": (legs.get("config2") or {}).get("frac"),
            "config3_ms": (legs.get("config3") or {}).get("ms_per_step"), "config3_frac": (legs.get("config3") or {}).get("frac"),
            "config5_ms": (legs.get("config5") or {}).get("ms_per_step"), "config5_frac": (legs.get("config5") or {}).get("frac"),
            "any_order_ms": {k: v["ms_per_step"] for k, v in (any_order or {}).items()},
            "any_order_grouping_frac": {k: v["grouping"]["frac"] for k, v in (any_order or {}).items()},
            "any_order_same_profile": all(v["same_profile_as_the_grouped_stream"] for v in (any_order or {}).values()) if any_order else None,
            "parity_in_run": parity_in_run["ok"] if parity_in_run else None,
            "two_files_in_flight": (in_flight or {}).get("value"),
            "value_with_push": (with_push or {}).get("value"), "run_marked_with_push": ((marked or {}).get("with_push") or {}).get("value"),
            "cli_M_records_s": (cli or {}).get("value"), "cli_compression_ratio": (cli or {}).get("compression_ratio"),
            "cli_host_inflate_M_records_s": ((cli or {}).get("host_inflate") or {}).get("value"),
            "cli_unsorted_M_records_s": ((cli or {}).get("unsorted_file") or {}).get("value"),
            "cli_easy_file_M_records_s": ((cli or {}).get("easy_file") or {}).get("value"),
            "cli_sam_text_M_records_s": ((cli or {}).get("sam_text") or {}).get("value"),
            "cpu_baseline": (cpu or {}).get("value"), "cpu_baseline_mt": (cpu_mt or {}).get("value"),
        }
        print(json.dumps(line))
        try:
            os.unlink(out_path)
        except OSError:
            pass
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
