#!/usr/bin/env python3
"""Benchmark of the alignment-to-profile hot path on MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): M alignment-records/sec from the decoded-record stream to the final profile.
A step = one pass of the whole hot path over one batch of synthetic records that are already resident in HBM:
analyze_alignments -> (all-reduce when N > 1) -> finish_coverage -> filter_alignments -> get_reads_lca_count ->
write_abundance (profile TSV written to a file by rank 0).  The workload at N = 1 is BASELINE.json configs[1]
(10 M synthetic 100 bp records, 5 k bacterial refs, mean 3 hits/read, 1000 bp bins); with N ranks every rank holds its
own 10 M-record shard of the same sample (weak scaling) and the value is total records / max-over-ranks time.

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel, from HIP events recorded on the library's
stream inside the timed region; `cpu_baseline` is the CPU oracle (a port of the reference algorithm, 1 thread) on a
bounded prefix of the same record stream.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


def algorithmic_bytes(st, n_records, Bp_words, rec_bytes=16):
    """Algorithmic HBM bytes per launch of each kernel (DESIGN.md section 'Kernels and their rooflines').

    N records, V mapped records, P targets = distinct (read, ref) pairs, M reads, U / U2 unique reads before / after
    the filter, B coverage bins.  The totals add up to SURVEY.md section 8d's figure (about 40 B/record + 8 B/bin).
    """
    N, V, P, M = n_records, st["hits_count"], st["n_targets"], st["matches_count"]
    U, U2, B = st["uniq_matches_count"], st["uniq_matches_count2"], Bp_words
    return {
        "memset_bins": 4 * (B // 8192 + 5200),       # counters, tail, tile counters, child marks (bins are written whole by the tile kernels)
        "k_scan_tiles": 2 * 8 * (N // 2048 + 1),      # per-tile counts in and out
        "sort_by_ident": 8 * (2 * 16 * V + 8 * V),   # (sort path only) 8 passes: keys+payload in and out, keys again for the histogram
        "k_valid_count": 6 * N,                       # (sort path only) flag u16 + ref i32
        "k_compact": 18 * N + 16 * V,                 # (sort path only) read every record once, write ident/ref/gbin
        "k_front": rec_bytes * N + 8 * P + 16 * (N // 1024 + 1),  # every record once (key 8 + ref 4 + pos 4 (+ flag 2) bytes);
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


def pmc_traffic_bytes(kernel_name, records_per_gpu):
    """HBM bytes per launch of `kernel_name` from the committed rocprofv3 PMC passes (scripts/pmc_traffic.sh run on
    MI355X with the default workload; FETCH_SIZE and WRITE_SIZE collected in separate passes, unit KB).  gfx950
    correction per MI355X_MICROARCH.md section HBM: FETCH_SIZE counts 128-byte requests as 64 bytes, i.e. half of a
    coalesced stream -- calibrated here on k_ref_stats, whose 16 B/lane loads read exactly 8 B per bin (ratio 2.03) --
    so fetched bytes = 2 * FETCH_SIZE; WRITE_SIZE is exact.  None when no profile matches."""
    path = next((q for q in (os.path.join(ROOT, "profiles", r, "pmc_traffic_summary.json") for r in ("round2", "round1"))
                 if os.path.exists(q)), None)
    if records_per_gpu != 10_000_000 or path is None:
        return None
    with open(path) as f:
        d = json.load(f)
    for k, v in d.items():
        if k.split("<")[0] == kernel_name and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            return int((2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024)
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="config2")
    ap.add_argument("--records", type=int, default=0, help="records per rank (default: the config's size)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=10_000_000,
                    help="records of the same stream the single-threaded CPU restatement is timed on (default: all)")
    ap.add_argument("--no-bins", action="store_true",
                    help="do not materialise the coverage arrays in HBM (slimm_keep_bins(0): their statistics are taken from "
                         "the finished tiles in LDS either way; what `slimm` does without -co)")
    ap.add_argument("--cpu-passes", type=int, default=2, help="passes of the CPU restatement over that sample (2 = ~14 s)")
    ap.add_argument("--breakdown", action="store_true", help="print the per-kernel table to stderr")
    ap.add_argument("--record-order", default="grouped", choices=["grouped", "any"],
                    help="'any' sends the same records through the device sort path (record_order = SLIMM_ORDER_ANY)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "summary", "sliced", "bins"],
                    help="multi-GPU exchange before the cut-offs: all-gather of sums + bin bitmaps (summary), all-to-all of "
                         "bitmap slices + small all-reduce (sliced), all-reduce of the bins; auto = summary up to 2 ranks")
    ap.add_argument("--kernel-timing", choices=("dominant", "all"), default="dominant",
                    help="HIP events in the timed steps: around the dominant kernel only (default) or around every launch")
    ap.add_argument("--no-config3", action="store_true",
                    help="skip the second measurement on BASELINE.json configs[2] (100 M records, 20 k refs: the "
                         "designated HBM-roofline run), which adds ~15 s at N = 1")
    ap.add_argument("--config3-steps", type=int, default=5)
    ap.add_argument("--config3-records", type=int, default=0, help="records of that run (default: all 100 M)")
    ap.add_argument("--push-batch", type=int, default=1 << 20, help="records per slimm_push_records call")
    ap.add_argument("--push-steps", type=int, default=3,
                    help="steps of the push-inclusive measurement (records start in host memory; 0 = skip)")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: ONE seeded stream of --records (default: the config's) records in 10 M-record "
                         "chunks, rank r generates and keeps chunks [r C / N, (r + 1) C / N) -- cuts at read boundaries "
                         "(slimm_amd/partition.py); implied by --config config4")
    ap.add_argument("--chunk-records", type=int, default=10_000_000, help="records per chunk of the strong-scaling stream")
    ap.add_argument("--form", default="packed", choices=["packed", "four"],
                    help="record arrays handed to the library: 'packed' = 16 B/record (slimm_set_records_device_packed: key with "
                         "the three flag bits folded in, ref, pos), 'four' = 18 B/record (key, ref, pos, flag)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="run the multi-rank code path (process group, collectives) even with one rank")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from slimm_amd.distributed import resolve_exchange, sharded_profile
    from slimm_amd.profiler import Slimm
    from slimm_amd.synth import CONFIGS, make_workload

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
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or args.force_exchange:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # (stdout carries ONE line: RCCL's version banner, printed at NCCL_DEBUG=VERSION / INFO, goes nowhere near it)
        os.environ["NCCL_DEBUG"] = os.environ.get("SLIMM_NCCL_DEBUG", "WARN")
        os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")   # (RCCL logs to stdout by default)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    cfg = CONFIGS[args.config]
    strong = args.strong or args.config == "config4"
    t0 = time.time()
    if strong:
        # ONE stream for every N: chunk c is make_workload(seed + 1000 c, shard = c) of the same sample, a grouped file
        # of whole reads; the stream is the chunks in order, and rank r owns a contiguous range of them.  A rank
        # generates only what it keeps, chunk by chunk straight into HBM.
        from slimm_amd.partition import chunk_owner

        n_stream = args.records or cfg.n_records
        n_chunks = max(1, (n_stream + args.chunk_records - 1) // args.chunk_records)
        mine = chunk_owner(n_chunks, world)[rank]
        parts = {"key": [], "ref": [], "pos": [], "flag": []}
        w = None
        for c in mine:
            wc = make_workload(cfg, seed=args.seed + 1000 * c, n_records=min(args.chunk_records, n_stream - c * args.chunk_records),
                               sample_seed=args.seed, shard=c)
            parts["key"].append(torch.from_numpy(wc.records.read_key.view(np.int64)).to(dev))
            parts["ref"].append(torch.from_numpy(wc.records.ref_id).to(dev))
            parts["pos"].append(torch.from_numpy(wc.records.begin_pos).to(dev))
            parts["flag"].append(torch.from_numpy(wc.records.flag.view(np.int16)).to(dev))
            if w is None:
                w = wc          # header, database, options (the same for every chunk) + the sample for the CPU baseline
        if w is None:           # more ranks than chunks: this rank has no records, but takes part in the exchange
            w = make_workload(cfg, seed=args.seed, n_records=1000, sample_seed=args.seed, shard=0)
            z = lambda dt: torch.zeros(0, dtype=dt, device=dev)
            key, ref, pos, flag = z(torch.int64), z(torch.int32), z(torch.int32), z(torch.int16)
        else:
            key, ref, pos, flag = (torch.cat(parts[k]) for k in ("key", "ref", "pos", "flag"))
        del parts
        n_rec = int(key.shape[0])
    else:
        n_rec = args.records or cfg.n_records
        w = make_workload(cfg, seed=args.seed + 1000 * rank, n_records=n_rec, sample_seed=args.seed, shard=rank)
        key = torch.from_numpy(w.records.read_key.view(np.int64)).to(dev)
        ref = torch.from_numpy(w.records.ref_id).to(dev)
        pos = torch.from_numpy(w.records.begin_pos).to(dev)
        flag = torch.from_numpy(w.records.flag.view(np.int16)).to(dev)
    gen_s = time.time() - t0

    eng = Slimm.for_workload(w, device=local_rank, grouped=(args.record_order == "grouped"))
    eng.force_exchange = args.force_exchange
    if args.no_bins:
        eng.keep_bins(False)
    torch.cuda.synchronize()
    out_path = os.path.join(tempfile.gettempdir(), f"slimm_bench_profile_{os.getpid()}.tsv")

    phase_times = {} if args.breakdown else None

    def to_packed(k, f):
        """slimm_pack_key on device tensors: (key & (2^61 - 1)) | mate << 61 | unmapped << 63 (what a decoder writes)."""
        f = f.to(torch.int64) & 0xffff
        mate = torch.where((f & 0x40) != 0, 1, torch.where((f & 0x80) != 0, 2, 0)).to(torch.int64)
        return (k & ((1 << 61) - 1)) | (mate << 61) | (((f & 0x4) != 0).to(torch.int64) << 63)

    pkey = None
    if args.form == "packed":
        pkey = to_packed(key, flag)
        del key, flag   # (the packed form needs neither)
        key = flag = None
        torch.cuda.synchronize()

    def step():
        eng.reset()
        eng.reset_cutoffs()            # every step is a fresh file for a fresh `slimm` object
        if pkey is not None:
            eng.set_records_device_packed(pkey, ref, pos)
        else:
            eng.set_records_device(key, ref, pos, flag)
        return sharded_profile(eng, dev, out_path, phase_times=phase_times, exchange=args.exchange)

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    # Warm-up steps run with every launch bracketed by HIP events: that survey names the dominant kernel and gives the
    # --breakdown table.  The timed steps bracket only the dominant kernel (its duration is what `roofline` reports),
    # because every event pair costs ~10 us of stream idle time and 18 pairs per step would be charged to `value`.
    eng.enable_kernel_timing(True)
    eng.kernel_times(reset=True)
    survey_steps = args.warmup
    for i in range(args.warmup):
        step()
        if i == 0 and args.warmup > 1:   # the first step allocates and runs cold: keep it out of the survey
            eng.kernel_times(reset=True)
            survey_steps -= 1
    survey = eng.kernel_times(reset=True) if args.warmup > 0 else {}
    dom_name = None
    if survey and args.kernel_timing == "dominant":
        cand = {k: ms for k, (ms, n) in survey.items() if n and k not in ("memset_bins", "k_pick_runs")}
        if cand:
            dom_name = max(cand, key=cand.get)
            eng.time_only_kernel(dom_name)
    if phase_times is not None:
        phase_times.clear()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        profile = step()
    barrier()
    elapsed = time.perf_counter() - t0
    ktimes = eng.kernel_times(reset=True)
    eng.enable_kernel_timing(False)
    eng.time_only_kernel(None)
    if dom_name is not None:
        # the other kernels' figures (breakdown table, device_kernel_ms_per_step) come from the warm-up survey
        live = ktimes[dom_name]
        ktimes = {k: ((ms / survey_steps * args.steps), int(round(n / survey_steps * args.steps)))
                  for k, (ms, n) in survey.items()}
        ktimes[dom_name] = live
    if dist.is_initialized():
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    st = eng.stats()
    total_records = n_rec * world
    if strong and dist.is_initialized():
        tot = torch.tensor([n_rec], dtype=torch.int64, device=dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_records = int(tot.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = total_records / (elapsed / args.steps) / 1e6

    def roofline_from(engine, stats, n_records, kt, steps, traffic_for=None):
        """The roofline object of the dominant kernel + the per-kernel table, from HIP-event times `kt`
        ({kernel: (ms, launches)} over `steps` steps)."""
        if args.no_bins:
            Bp = int(stats["total_bins"])  # (the padded count is a property of the buffer, which is not exposed here)
        else:
            Bp = int(engine.coverage_buffer().__cuda_array_interface__["shape"][0] - 16) // 2
        model = algorithmic_bytes(stats, n_records, Bp, 16 if args.form == "packed" else 18)
        per_kernel = {}
        for name, (ms, launches) in kt.items():
            if launches and name in model:
                steps_launches = launches / steps
                per_kernel[name] = {"ms_per_launch": ms / launches, "launches_per_step": steps_launches,
                                    "bytes_per_launch": model[name] / max(1.0, steps_launches if name == "memset_bins" else 1.0)}
        # of the two classification kernels the one the device did not pick returns at once (k_runs_hash then only sums
        # the per-tile counts by chunk): its bytes are those counts, not the records
        pair = [k for k in ("k_runs", "k_runs_hash") if k in per_kernel]
        if len(pair) == 2:
            idle = min(pair, key=lambda k: per_kernel[k]["ms_per_launch"])
            per_kernel[idle]["bytes_per_launch"] = 12 * (n_records // 2048 + 1)
        # the dominant kernel = most time per step (memsets are DMA fills, not kernels of this library)
        cand = {k: v for k, v in per_kernel.items() if k not in ("memset_bins", "k_pick_runs")}
        dom = max(cand, key=lambda k: cand[k]["ms_per_launch"] * cand[k]["launches_per_step"])
        d = cand[dom]
        achieved = d["bytes_per_launch"] / (d["ms_per_launch"] * 1e-3) / 1e9
        roof = {"kernel": dom, "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": pmc_traffic_bytes(dom, n_records) if traffic_for == "config2" else None,
                "traffic_source": "profiles/round2/pmc_traffic_summary.json (rocprofv3 --pmc, separate passes)",
                "bytes_per_launch": int(d["bytes_per_launch"]), "ms_per_launch": round(d["ms_per_launch"], 4)}
        kernel_ms = sum(v["ms_per_launch"] * v["launches_per_step"] for v in per_kernel.values())
        return roof, per_kernel, kernel_ms

    if rank == 0:
        roofline, per_kernel, kernel_ms = roofline_from(eng, st, n_rec, ktimes, args.steps, traffic_for=args.config)
        if args.breakdown:
            print(f"# generate {gen_s:.1f}s; records/rank {n_rec}; V={st['hits_count']} M={st['matches_count']} "
                  f"P={st['n_targets']} U={st['uniq_matches_count']} U2={st['uniq_matches_count2']} "
                  f"valid={st['n_valid']} B={st['total_bins']}", file=sys.stderr)
            for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1]["ms_per_launch"] * kv[1]["launches_per_step"]):
                gbs = v["bytes_per_launch"] / (v["ms_per_launch"] * 1e-3) / 1e9
                print(f"# {k:16s} {v['ms_per_launch']*1e3:9.1f} us/launch x{v['launches_per_step']:.0f}  "
                      f"{v['bytes_per_launch']/1e6:9.1f} MB  {gbs:8.1f} GB/s  {gbs/HBM_PEAK_GBS*100:5.1f}% of HBM peak",
                      file=sys.stderr)
            print(f"# device kernels {kernel_ms:.3f} ms of {ms_per_step:.3f} ms per step", file=sys.stderr)
            for k, v in phase_times.items():
                print(f"# host wall {k:28s} {v / args.steps * 1e6:9.1f} us/step", file=sys.stderr)

        # ---- push-inclusive rate: the clock starts before the first record leaves host
        # memory and stops when the profile file is written.  Never `value`: PCIe, not the path, bounds it.
        with_push = None
        if world == 1 and args.push_steps > 0 and not args.force_exchange and not strong:
            # Streamed ingest (slimm_push_records_async): the records sit in page-locked host memory (where a decoder
            # would have written them), the copies of file k + 1 run on the copy stream while file k is profiled on
            # another context -- the -d directory mode of the slimm command.  One "step" = one file: clock from the
            # moment its first record leaves host memory (pipelined: a file's copy overlaps its predecessor's phases)
            # to its profile file written.
            rec = w.records
            pinned = [torch.from_numpy(a).pin_memory().numpy() for a in
                      (rec.read_key, rec.ref_id, rec.begin_pos, rec.flag)]
            engs = [eng, Slimm.for_workload(w, device=local_rank, grouped=True)]
            if args.no_bins:
                engs[1].keep_bins(False)
            for e in engs:   # the first pass sizes the library's own record buffers
                e.reset()
                e.reset_cutoffs()
                e.push_records_async(*pinned)
                e.get_profiles(path=out_path)
            torch.cuda.synchronize()
            n_files = 2 * args.push_steps
            engs[0].reset()
            engs[0].reset_cutoffs()
            t1 = time.perf_counter()
            engs[0].push_records_async(*pinned)
            for i in range(n_files):
                cur, nxt = engs[i & 1], engs[(i + 1) & 1]
                if i + 1 < n_files:
                    nxt.reset()
                    nxt.reset_cutoffs()
                    nxt.push_records_async(*pinned)
                cur.get_profiles(path=out_path)
            dt = (time.perf_counter() - t1) / n_files
            engs[1].close()
            # ... and the unpipelined form: one file, synchronous slimm_push_records from pageable memory
            eng.reset()
            eng.reset_cutoffs()
            t1 = time.perf_counter()
            eng.push_records(rec, batch=args.push_batch)
            eng.get_profiles(path=out_path)
            dt_sync = time.perf_counter() - t1
            with_push = {"value": round(n_rec / dt / 1e6, 3), "unit": "M records/s", "ms_per_step": round(dt * 1e3, 4),
                         "what": f"{n_files} files of {n_rec} records from page-locked host memory through "
                                 "slimm_push_records_async, two contexts alternating (a file's copy overlaps the file "
                                 "before's phases), per file to profile written; 18 B/record over PCIe",
                         "gb_per_s_over_pcie": round(18.0 * n_rec / dt / 1e9, 2),
                         "single_file_sync_push": {"value": round(n_rec / dt_sync / 1e6, 3), "unit": "M records/s",
                                                   "what": f"one file, slimm_push_records (pageable arrays, batches of "
                                                           f"{args.push_batch}) to profile written"}}

        # ---- BASELINE.json configs[2]: the designated HBM-roofline run (100 M records, 20 k refs, mean 8 hits/read)
        roof3 = None
        if world == 1 and not args.no_config3 and args.config == "config2" and not args.force_exchange and not strong:
            cfg3 = CONFIGS["config3"]
            w3 = make_workload(cfg3, seed=args.seed, n_records=args.config3_records or cfg3.n_records)
            n3 = len(w3.records)
            eng3 = Slimm.for_workload(w3, device=local_rank, grouped=True)
            if args.no_bins:
                eng3.keep_bins(False)
            k3 = torch.from_numpy(w3.records.read_key.view(np.int64)).to(dev)
            r3 = torch.from_numpy(w3.records.ref_id).to(dev)
            p3 = torch.from_numpy(w3.records.begin_pos).to(dev)
            f3 = torch.from_numpy(w3.records.flag.view(np.int16)).to(dev)
            if args.form == "packed":
                k3 = to_packed(k3, f3)
            torch.cuda.synchronize()

            def step3():
                eng3.reset()
                eng3.reset_cutoffs()
                if args.form == "packed":
                    eng3.set_records_device_packed(k3, r3, p3)
                else:
                    eng3.set_records_device(k3, r3, p3, f3)
                return eng3.get_profiles(path=out_path)

            eng3.enable_kernel_timing(True)
            step3()                                   # cold step: allocations
            eng3.kernel_times(reset=True)
            step3()                                   # survey: every launch bracketed
            survey3 = eng3.kernel_times(reset=True)
            cand3 = {k: ms for k, (ms, n) in survey3.items() if n and k not in ("memset_bins", "k_pick_runs")}
            dom3 = max(cand3, key=cand3.get)
            eng3.time_only_kernel(dom3)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.config3_steps):
                step3()
            torch.cuda.synchronize()
            el3 = time.perf_counter() - t1
            live3 = eng3.kernel_times(reset=True)
            kt3 = {k: (ms * args.config3_steps, n * args.config3_steps) for k, (ms, n) in survey3.items()}
            kt3[dom3] = live3[dom3]
            st3 = eng3.stats()
            roof3, pk3, kms3 = roofline_from(eng3, st3, n3, kt3, args.config3_steps)
            roof3.pop("traffic_source", None)
            roof3.update({"workload": f"BASELINE.json configs[2] (config3): {n3} records, {cfg3.n_refs} refs, mean "
                                      f"{cfg3.mean_hits} hits/read, {cfg3.bin_width} bp bins; records resident in HBM "
                                      f"before the timed region",
                          "steps": args.config3_steps, "ms_per_step": round(el3 / args.config3_steps * 1e3, 4),
                          "value": round(n3 / (el3 / args.config3_steps) / 1e6, 3), "unit_value": "M records/s",
                          "device_kernel_ms_per_step": round(kms3, 4),
                          "reads": st3["matches_count"], "targets": st3["n_targets"], "bins": st3["total_bins"]})
            if args.breakdown:
                print(f"# config3: {n3} records, {el3 / args.config3_steps * 1e3:.3f} ms/step", file=sys.stderr)
                for k, v in sorted(pk3.items(), key=lambda kv: -kv[1]["ms_per_launch"] * kv[1]["launches_per_step"]):
                    gbs = v["bytes_per_launch"] / (v["ms_per_launch"] * 1e-3) / 1e9
                    print(f"# c3 {k:16s} {v['ms_per_launch']*1e3:9.1f} us/launch x{v['launches_per_step']:.0f}  "
                          f"{v['bytes_per_launch']/1e6:9.1f} MB  {gbs:8.1f} GB/s  {gbs/HBM_PEAK_GBS*100:5.1f}% of HBM peak",
                          file=sys.stderr)
            eng3.close()

        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            from oracle.binding import Oracle  # the CPU restatement of the reference algorithm (checker / baseline only)

            ns = min(args.cpu_sample, n_rec)
            sample = w.records.take(np.arange(ns))
            passes = max(1, args.cpu_passes)
            cpu_s = wall = 0.0
            for _ in range(passes):  # every pass a fresh object, like every GPU step
                orc = Oracle(w.taxonomy, w.options)
                t1 = time.perf_counter()
                o = orc.run(w.ref_names, w.ref_len, sample, w.avg_read_len, want_raw=False, want_cov=False,
                            use_qnames=False, collect_bins=False)
                wall += time.perf_counter() - t1
                cpu_s += sum(o.phase_seconds)  # the three phases + profile, excluding reference/bin allocation
            cpu = {"value": round(passes * ns / cpu_s / 1e6, 4), "unit": "M records/s", "cores": 1, "kind": "port",
                   "sample": f"{passes} pass(es) over the first {ns} records of the same stream (same refs/DB), phases "
                             f"A+B+C+profile {cpu_s:.1f}s of {wall:.1f}s wall",
                   "host": f"{os.cpu_count()} logical cores, cgroup quota {cpu_quota_cores()} cores"}

        # ---- the same host's cores, all of them: the dense multi-threaded restatement of phases A, B and the per-read
        # LCA (oracle/slimm_dense_mt.cpp, checked against the oracle in tests/test_dense_mt.py).  A reported baseline
        # like cpu_baseline -- the reference itself is single-threaded -- never a code path of the product.
        cpu_mt = None
        if world == 1 and not args.no_cpu_baseline and not strong:
            from oracle.binding import dense_mt_run

            ncpu = os.cpu_count() or 1
            best = threads = None
            t_all = time.perf_counter()
            # more threads than the memory system feeds only add contention on the histograms: the best of a few counts
            for th in sorted({ncpu, max(1, ncpu // 2), max(1, ncpu // 4), min(ncpu, 32)}):
                for _ in range(2):
                    d = dense_mt_run(w, threads=th)
                    sec = sum(d["seconds"])
                    if best is None or sec < best:
                        best, threads = sec, th
            agree = (d["hits"], d["matches"], d["uniq_matches"], d["uniq_matches2"]) == (
                st["hits_count"], st["matches_count"], st["uniq_matches_count"], st["uniq_matches_count2"])
            cpu_mt = {"value": round(len(w.records) / best / 1e6, 3), "unit": "M records/s", "cores": threads, "kind": "port",
                      "sample": f"best pass over all {len(w.records)} records of the same stream, {threads} threads (best of "
                                f"{ncpu}, {ncpu // 2}, {ncpu // 4}, 32 on {ncpu} logical cores): "
                                f"phases A + B + per-read LCA {best * 1e3:.1f} ms (array allocation and the scalar profile "
                                f"tail excluded), {time.perf_counter() - t_all:.1f} s wall for all passes",
                      "cpu_quota_cores": cpu_quota_cores(), "scalars_equal_gpu": bool(agree)}

        line = {
            "metric": "M alignment-records/sec -> final profile",
            "value": round(value, 3), "unit": "M records/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"BASELINE.json configs[{list(CONFIGS).index(args.config)}] ({args.config}): "
                                   f"{n_rec} records/GPU, {cfg.n_refs} refs, mean {cfg.mean_hits} hits/read, "
                                   f"{cfg.bin_width} bp bins, {cfg.read_len} bp reads",
                       "records_per_gpu": n_rec, "total_records": total_records, "refs": cfg.n_refs,
                       "reads": st["matches_count"], "targets": st["n_targets"], "bins": st["total_bins"],
                       "record_order": args.record_order, "coverage_arrays": "not materialised" if args.no_bins else "in HBM",
                       "records": "resident in HBM before the timed region (value_with_push starts in host memory)",
                       "seed": args.seed, "parallelism": f"reads sharded over {world} GPU(s)",
                       "stream": (f"one seeded stream of {total_records} records in chunks of {args.chunk_records}, "
                                  f"contiguous chunk ranges per rank") if strong else "one seeded shard per rank",
                       "exchange": (resolve_exchange(eng, args.exchange, world) if (world > 1 or args.force_exchange) else "none"),
                       "profile_rows": len(profile.strip().split("\n")) - 1 if profile else 0},
            "roofline": roofline,
            "roofline_config3": roof3,
            "value_with_push": with_push,
            "cpu_baseline": cpu,
            "cpu_baseline_mt": cpu_mt,
            "device_kernel_ms_per_step": round(kernel_ms, 4),
            # every kernel of the step from the warm-up survey (events around every launch), longest first: the dominant one
            # is whichever is longest on this box -- k_front and k_filter are within a few us of each other
            "kernels": {k: {"us": round(v["ms_per_launch"] * 1e3, 1),
                            "frac_of_hbm_peak": round(v["bytes_per_launch"] / (v["ms_per_launch"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 3)}
                        for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1]["ms_per_launch"] * kv[1]["launches_per_step"])
                        if v["ms_per_launch"] > 0},
            "kernel_timing": ("HIP events around " + (f"{dom_name} only in the timed steps (other kernels: warm-up survey)"
                                                      if dom_name else "every launch in the timed steps")),
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
