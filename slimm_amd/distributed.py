"""One-process-per-GPU sharding of the alignment-to-profile path.

The record stream is partitioned BY READ (all records of a read on one rank: splitting a read would break the
first-bin rule Q1 and the unique/multi classification).  Each rank runs phase A on its shard; ONE collective (RCCL
over xGMI when the backend is "nccl") then gives every rank what the cut-offs need from the others -- by default an
all-gather of per-reference sums plus one bit per bin, optionally the all-reduce of the integer bins themselves; every
rank derives identical non-zero-bin counts, cut-offs and valid set, runs phase B / C(1) on its own reads, and the small
additive partial results (uniq_reads_count2, per-taxon LCA counts, child marks, no-agreement pairs) are merged by a
second, tiny exchange.  The reference has no counterpart (single process, single thread).

The functions take an `engine` with the method names of `slimm_amd.profiler.Slimm`; the collectives are plain
`torch.distributed` calls, so the same code runs over gloo on CPU in the tests.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch
import torch.distributed as dist


def exchange_coverage(engine, group=None, mode: str = "summary") -> bool:
    """The exchange between phase A and the cut-offs; returns engine.finish_coverage*()'s answer.

    mode "summary" (default): ONE all-gather of each rank's [per-reference sums | scalars | 'bin != 0' bitmaps]
        (about 1/16 of the bins buffer); every rank then sums / ORs the gathered pieces on its own GPU.
    mode "bins": ONE all-reduce(SUM) over [cov | uniq_cov | scalars] in place -- needed only when the caller wants the
        global coverage arrays themselves (the reference's -co output); 16 x more bytes on the wire.
    """
    multi = dist.is_initialized() and dist.get_world_size(group) > 1
    if not multi and not getattr(engine, "force_exchange", False):
        return engine.finish_coverage()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if mode == "bins":
        buf = engine.coverage_tensor()
        if dist.is_initialized():
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
            if buf.is_cuda:
                torch.cuda.synchronize(buf.device)
        return engine.finish_coverage()
    mine = engine.coverage_summary_tensor()
    gathered = torch.empty(world * mine.numel(), dtype=mine.dtype, device=mine.device)
    if dist.is_initialized():
        dist.all_gather_into_tensor(gathered, mine, group=group)
    else:
        gathered.copy_(mine)
    if gathered.is_cuda:
        torch.cuda.synchronize(gathered.device)
    return engine.finish_coverage_merged(gathered, world)


def merge_partials(engine, device: Optional[torch.device] = None, group=None):
    """Second, small exchange: sums, ORs (as sums of 0/1 flags) and a set union of the per-rank partial results."""
    p = engine.get_partials()
    if not dist.is_initialized():
        return p  # nothing to merge with
    world = dist.get_world_size(group)
    dev = device or torch.device("cpu")
    R = p["uniq_reads_count2"].shape[0]
    T = p["lca_count"].shape[0]
    marks = p["level_marks"].astype(np.uint32)
    flags = ((marks[:, None] >> np.arange(8, dtype=np.uint32)[None, :]) & 1).astype(np.int32).reshape(-1)
    packed = np.concatenate([p["uniq_reads_count2"].view(np.int32), p["lca_count"].view(np.int32), flags,
                             np.array([p["pairs"].shape[0]], dtype=np.int32)])
    t = torch.from_numpy(packed).to(dev)
    npairs_local = int(p["pairs"].shape[0])
    mx = torch.tensor([npairs_local], dtype=torch.int64, device=dev)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    cap = int(mx.item())
    pairs = p["pairs"]
    if cap > 0:
        mine = torch.full((cap + 1,), -1, dtype=torch.int64, device=dev)
        mine[0] = npairs_local
        if npairs_local:
            mine[1:1 + npairs_local] = torch.from_numpy(pairs.view(np.int64)).to(dev)
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine, group=group)
        parts = []
        for g in gathered:
            g = g.cpu().numpy()
            parts.append(g[1:1 + int(g[0])])
        pairs = np.unique(np.concatenate(parts).view(np.uint64))
    out = t.cpu().numpy()
    u2 = out[:R].view(np.uint32).copy()
    lca = out[R:R + T].view(np.uint32).copy()
    fl = out[R + T:R + T + 8 * R].reshape(R, 8) > 0
    mk = (fl.astype(np.uint32) << np.arange(8, dtype=np.uint32)[None, :]).sum(axis=1).astype(np.uint32)
    return {"uniq_reads_count2": u2, "lca_count": lca, "level_marks": mk, "pairs": pairs}


def sharded_profile(engine, device: Optional[torch.device] = None, path: Optional[str] = None, group=None,
                    phase_times: Optional[dict] = None, exchange: str = "summary"):
    """slimm::get_profiles() (reference src/slimm.hpp:395-496) over a record stream sharded across ranks.

    `engine` already holds this rank's records.  Returns the profile text (identical on every rank) or None when no
    rank has a mapped record.
    """
    import time

    def lap(name, t0):
        if phase_times is not None:
            phase_times[name] = phase_times.get(name, 0.0) + (time.perf_counter() - t0)
        return time.perf_counter()

    multi = (dist.is_initialized() and dist.get_world_size(group) > 1) or getattr(engine, "force_exchange", False)
    t = time.perf_counter()
    engine.analyze_alignments()
    t = lap("analyze_alignments(launch)", t)
    have_hits = exchange_coverage(engine, group, exchange)
    t = lap("exchange + finish_coverage", t)
    if not have_hits:
        return None
    engine.filter_alignments()
    t = lap("filter_alignments", t)
    if multi or getattr(engine, "needs_set_partials", False):
        merged = merge_partials(engine, device, group)
        engine.set_partials(merged["uniq_reads_count2"], merged["lca_count"], merged["level_marks"], merged["pairs"])
        t = lap("merge_partials", t)
    engine.get_reads_lca_count()
    t = lap("get_reads_lca_count", t)
    write_here = path if (not dist.is_initialized() or dist.get_rank(group) == 0) else None
    out = engine.write_abundance(write_here)
    lap("write_abundance", t)
    return out
