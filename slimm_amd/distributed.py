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

import contextlib
from typing import Optional

import numpy as np
import torch
import torch.distributed as dist


def resolve_exchange(engine, mode: str, world: int) -> str:
    """What "auto" means for this engine and world size; other modes pass through."""
    if mode == "auto":
        return "sliced" if (world > 2 and hasattr(engine, "merge_summary_slices")) else "summary"
    return mode


def _engine_stream(engine):
    """(context manager, ordered): engines that expose their HIP stream (Slimm.torch_stream) get their collectives
    enqueued there -- ProcessGroupNCCL orders its own stream against the current one with events, so the library's
    kernels, the collective and the library's next kernels run back to back without the host waiting in between."""
    if hasattr(engine, "torch_stream") and torch.cuda.is_available():
        return torch.cuda.stream(engine.torch_stream()), True
    return contextlib.nullcontext(), False


def _fence(t, ordered: bool):
    if t.is_cuda and not ordered:
        torch.cuda.synchronize(t.device)


# The collectives of this module.  Over RCCL ("nccl") they take the device tensors as they are.  A backend without
# device collectives (gloo: the rehearsal of the multi-process driver with several processes on ONE GPU, tests/
# test_distributed_gpu.py and `bench.py --backend gloo`) gets them staged through host memory: device -> host on the
# current stream (the engine's, inside _engine_stream), the collective on the host copies, host -> device.
def _host_staged(t, group) -> bool:
    return t.is_cuda and dist.get_backend(group) != "nccl"


#
# COLLECTIVE_EVENTS: a measuring caller (bench.py's step_split) sets it to a list; every collective on a device tensor is
# then bracketed by two timing events on the current stream (the engine's) and noted as (name, bytes, start, end).
COLLECTIVE_EVENTS = None


@contextlib.contextmanager
def _timed(name, t):
    if COLLECTIVE_EVENTS is None or not t.is_cuda:
        yield
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    yield
    e1.record()
    COLLECTIVE_EVENTS.append((name, t.numel() * t.element_size(), e0, e1))


def _all_reduce_sum(t, group):
    with _timed("all_reduce", t):
        if _host_staged(t, group):
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


def _all_gather_into(out, mine, group):
    with _timed("all_gather", out):
        if _host_staged(mine, group):
            ho = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(ho, mine.cpu(), group=group)
            out.copy_(ho)
        else:
            dist.all_gather_into_tensor(out, mine, group=group)


def _all_to_all(out, mine, group):
    with _timed("all_to_all", mine):
        if _host_staged(mine, group):
            ho = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(ho, mine.cpu().contiguous(), group=group)
            out.copy_(ho)
        else:
            dist.all_to_all_single(out, mine, group=group)


def exchange_coverage(engine, group=None, mode: str = "auto") -> bool:
    """The exchange between phase A and the cut-offs; returns engine.finish_coverage*()'s answer.

    mode "summary" (default): ONE all-gather of each rank's [per-reference sums | scalars | 'bin != 0' bitmaps]
        (about 1/16 of the bins buffer); every rank then sums / ORs the gathered pieces on its own GPU.
    mode "sliced": the same information for many ranks: an all-to-all in which every rank receives only ITS slice
        (1/world) of every other rank's bitmaps, ORs them and counts non-zero bins per reference inside the slice, then
        ONE small all-reduce(SUM) of [per-reference sums, partial non-zero counts | scalars].  An all-gather delivers
        (world - 1) x 5 MB to every rank at config 2, the all-to-all (world - 1) / world x 5 MB.
    mode "auto": "summary" up to 2 ranks, "sliced" above (engines without sliced support: "summary").
    mode "bins": ONE all-reduce(SUM) over [cov | uniq_cov | scalars] in place -- needed only when the caller wants the
        global coverage arrays themselves (the reference's -co output); 16 x more bytes on the wire.
    """
    multi = dist.is_initialized() and dist.get_world_size(group) > 1
    if not multi and not getattr(engine, "force_exchange", False):
        return engine.finish_coverage()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    on_stream, ordered = _engine_stream(engine)
    with on_stream:
        if mode == "bins":
            buf = engine.coverage_tensor()
            if dist.is_initialized():
                _all_reduce_sum(buf, group)
                _fence(buf, ordered)
            return engine.finish_coverage()
        if resolve_exchange(engine, mode, world) == "sliced":
            mine = engine.coverage_summary_tensor()
            head = engine.summary_head_words()
            chunks = mine[head:]
            received = (engine.scratch("received", chunks.numel(), chunks) if hasattr(engine, "scratch")
                        else torch.empty_like(chunks))
            if dist.is_initialized():
                _all_to_all(received, chunks, group)
            else:
                received.copy_(chunks)
            _fence(received, ordered)
            vec = engine.merge_summary_slices(received, world, dist.get_rank(group) if dist.is_initialized() else 0)
            if dist.is_initialized():
                _all_reduce_sum(vec, group)
            _fence(vec, ordered)
            return engine.finish_coverage_reduced()
        mine = engine.coverage_summary_tensor()
        gathered = (engine.scratch("gathered", world * mine.numel(), mine) if hasattr(engine, "scratch")
                    else torch.empty(world * mine.numel(), dtype=mine.dtype, device=mine.device))
        if dist.is_initialized():
            _all_gather_into(gathered, mine, group)
        else:
            gathered.copy_(mine)
        _fence(gathered, ordered)
        return engine.finish_coverage_merged(gathered, world)


# level marks travel as one int64 per reference with one 8-bit field per level: a SUM over up to 255 ranks cannot carry
# from one field into the next, so "field != 0" after the all-reduce is the OR of the ranks' bits
_SPREAD = np.array([sum(((m >> lv) & 1) << (8 * lv) for lv in range(8)) for m in range(256)], dtype=np.int64)


def _gather_pairs(pairs: np.ndarray, total_pairs: int, dev, group) -> np.ndarray:
    """Union of every rank's (taxon << 32 | reference) pairs; total_pairs (the same on every rank) bounds the buffers."""
    world = dist.get_world_size(group)
    n_local = int(pairs.shape[0])
    mine = torch.full((total_pairs + 1,), -1, dtype=torch.int64, device=dev)
    mine[0] = n_local
    if n_local:
        mine[1:1 + n_local] = torch.from_numpy(pairs.view(np.int64)).to(dev)
    gathered = torch.empty(world * (total_pairs + 1), dtype=torch.int64, device=dev)
    _all_gather_into(gathered, mine, group)
    g = gathered.cpu().numpy().reshape(world, total_pairs + 1)
    return np.unique(np.concatenate([g[k, 1:1 + int(g[k, 0])] for k in range(world)]).view(np.uint64))


def merge_partials_on_device(engine, group=None, launched: bool = False) -> bool:
    """The second exchange without a host detour, for engines that expose their partial results as a device tensor
    (`partials_tensor` / `install_merged_partials`, i.e. the HIP library): ONE all-reduce(SUM) in place, one copy back.
    Returns False when the engine has no such tensor (the caller then merges through the host).
    launched: phase B was started with filter_alignments_launch() -- the install below is then the phase's only host
    synchronisation, and it may ask every rank to go round again (a pair set overflowed somewhere)."""
    if not hasattr(engine, "partials_tensor") or not dist.is_initialized():
        return False
    if dist.get_world_size(group) > 255:
        raise ValueError("level marks travel in 8-bit fields: at most 255 ranks")
    on_stream, ordered = _engine_stream(engine)
    with on_stream:
        while True:
            t = engine.partials_tensor()
            _all_reduce_sum(t, group)
            _fence(t, ordered)
            total_pairs = engine.install_merged_partials()
            if total_pairs is not None:
                break
            engine.filter_alignments_launch()   # (only ever after a launched phase)
    if total_pairs > 0:  # rare (Q4: reads whose references agree at no level); the same decision on every rank
        p = engine.get_partials()
        pairs = _gather_pairs(p["pairs"], total_pairs, t.device, group)
        engine.set_partials(p["uniq_reads_count2"], p["lca_count"], p["level_marks"], pairs)
    return True


def merge_partials(engine, device: Optional[torch.device] = None, group=None):
    """Second, small exchange through host arrays: ONE all-reduce(SUM) of [uniq_reads_count2 | per-taxon LCA counts |
    level marks | number of no-agreement pairs]; the pairs themselves (rare: reads whose references agree at no level,
    Q4) follow in an all-gather only when some rank has any."""
    p = engine.get_partials()
    if not dist.is_initialized():
        return p  # nothing to merge with
    world = dist.get_world_size(group)
    if world > 255:
        raise ValueError("merge_partials packs level marks into 8-bit fields: at most 255 ranks")
    dev = device or torch.device("cpu")
    R = p["uniq_reads_count2"].shape[0]
    T = p["lca_count"].shape[0]
    npairs_local = int(p["pairs"].shape[0])
    packed = np.empty(2 * R + T + 1, dtype=np.int64)
    packed[:R] = p["uniq_reads_count2"]
    packed[R:R + T] = p["lca_count"]
    packed[R + T:2 * R + T] = _SPREAD[p["level_marks"] & 0xff]
    packed[-1] = npairs_local
    t = torch.from_numpy(packed).to(dev)
    _all_reduce_sum(t, group)
    out = t.cpu().numpy()
    pairs = p["pairs"]
    total_pairs = int(out[-1])
    if total_pairs > 0:  # every rank sees the same total, so every rank takes this branch
        pairs = _gather_pairs(pairs, total_pairs, dev, group)
    u2 = out[:R].astype(np.uint32)
    lca = out[R:R + T].astype(np.uint32)
    sp = out[R + T:2 * R + T]
    mk = np.zeros(R, dtype=np.uint32)
    for lv in range(8):
        mk |= (((sp >> (8 * lv)) & 0xff) != 0).astype(np.uint32) << np.uint32(lv)
    return {"uniq_reads_count2": u2, "lca_count": lca, "level_marks": mk, "pairs": pairs}


def sharded_profile(engine, device: Optional[torch.device] = None, path: Optional[str] = None, group=None,
                    phase_times: Optional[dict] = None, exchange: str = "auto"):
    """slimm::get_profiles() (reference src/slimm.hpp:395-496) over a record stream sharded across ranks.

    `engine` already holds this rank's records.  Returns the profile text (identical on every rank) or None when no
    rank has a mapped record.
    """
    multi = (dist.is_initialized() and dist.get_world_size(group) > 1) or getattr(engine, "force_exchange", False)
    if not multi and phase_times is None and hasattr(engine, "get_profiles") and not getattr(engine, "needs_set_partials", False):
        if hasattr(engine, "prepare_summary"):
            engine.prepare_summary(0)
        return engine.get_profiles(path=path)  # one rank, nothing to exchange: the library's single call
    if not sharded_profile_begin(engine, device, group, phase_times, exchange):
        return None
    return sharded_profile_end(engine, path, group, phase_times)


def sharded_profile_begin(engine, device: Optional[torch.device] = None, group=None, phase_times: Optional[dict] = None,
                          exchange: str = "auto", after_launch=None) -> bool:
    """The device's part of `sharded_profile`: phase A, the exchange, the cut-offs, phase B and the merge of the partial
    results -- everything up to `slimm::get_reads_lca_count` (src/slimm.hpp:533).  `after_launch`, when given, is called
    once phase A has been launched and before the host waits for it: a caller that works through files back to back puts
    the host-only end of the file before (`sharded_profile_end` of ANOTHER engine) there, beside this file's front end.
    Returns False when no rank has a mapped record (`after_launch` has been called all the same)."""
    import time

    def lap(name, t0):
        if phase_times is not None:
            phase_times[name] = phase_times.get(name, 0.0) + (time.perf_counter() - t0)
        return time.perf_counter()

    multi = (dist.is_initialized() and dist.get_world_size(group) > 1) or getattr(engine, "force_exchange", False)
    t = time.perf_counter()
    if hasattr(engine, "prepare_summary"):
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        how = resolve_exchange(engine, exchange, world) if multi else "none"
        if how == "sliced":
            try:
                engine.prepare_summary(world)
            except Exception:  # the engine cannot slice (e.g. the direct-atomics fallback): same answer on every rank
                how = exchange = "summary"
        if how != "sliced":
            engine.prepare_summary(1 if how == "summary" else 0)
        exchange = how if multi else exchange
    engine.analyze_alignments()
    t = lap("analyze_alignments(launch)", t)
    if after_launch is not None:
        after_launch()
        t = lap("the file before: get_reads_lca_count + write_abundance", t)
    have_hits = exchange_coverage(engine, group, exchange)
    t = lap("exchange + finish_coverage", t)
    if not have_hits:
        return False
    launched = multi and dist.is_initialized() and hasattr(engine, "filter_alignments_launch") and hasattr(engine, "partials_tensor")
    if launched:
        engine.filter_alignments_launch()   # no host synchronisation until the merged results are installed
    else:
        engine.filter_alignments()
    t = lap("filter_alignments", t)
    if multi or getattr(engine, "needs_set_partials", False):
        if not merge_partials_on_device(engine, group, launched):
            merged = merge_partials(engine, device, group)
            engine.set_partials(merged["uniq_reads_count2"], merged["lca_count"], merged["level_marks"], merged["pairs"])
        t = lap("merge_partials", t)
    return True


def sharded_profile_end(engine, path: Optional[str] = None, group=None, phase_times: Optional[dict] = None):
    """The host's end of `sharded_profile`: the propagation of the LCA counts (src/slimm.hpp:560-610) and the profile
    (`write_abundance`, :733-843; rank 0 writes the file).  No device work, no collective."""
    import time
    t = time.perf_counter()
    engine.get_reads_lca_count()
    if phase_times is not None:
        phase_times["get_reads_lca_count"] = phase_times.get("get_reads_lca_count", 0.0) + (time.perf_counter() - t)
        t = time.perf_counter()
    write_here = path if (not dist.is_initialized() or dist.get_rank(group) == 0) else None
    out = engine.write_abundance(write_here)
    if phase_times is not None:
        phase_times["write_abundance"] = phase_times.get("write_abundance", 0.0) + (time.perf_counter() - t)
    return out


class FilesBackToBack:
    """Files one after the other through TWO engines in turn (the reference's unit of work is one file through one object,
    src/slimm.hpp:950-956; its `-d` mode loops over the files of a directory): while the device runs the front end of file
    k + 1 on one engine, the host finishes file k on the other -- propagation, profile text, the file written --, which
    otherwise is 0.25 ms of an idle GPU between two files.  Every file still goes through a freshly reset object, every
    profile is written; `flush()` finishes the last one.  `give(engine)` hands an engine the next file's records.

    By default every file is a FRESH `slimm` object (cut-off caches cleared: one `slimm DB IN.bam` run per file; what
    `bench.py` times).  `directory_mode=True` is the reference's `-d` loop instead: ONE object serves all files, so the
    cut-offs cached by the first file are reused by the files behind it (src/slimm.hpp:155-156, 330, 674; quirk Q8) --
    the cache is carried from the engine of file k to the engine of file k + 1 (known as soon as file k's phase A is
    finished, which is before file k + 1 starts)."""

    def __init__(self, engines, give, device=None, path=None, group=None, phase_times=None, exchange="auto",
                 directory_mode=False):
        self.directory_mode = directory_mode
        self.engines = list(engines)
        self.give, self.device, self.path, self.group = give, device, path, group
        self.phase_times, self.exchange = phase_times, exchange
        self.k = 0
        self.pending = None
        self.last = self.engines[0]
        self.profile = None

    def _finish_pending(self):
        if self.pending is not None:
            self.profile = sharded_profile_end(self.pending, self.path, self.group, self.phase_times)
            self.pending = None

    def step(self):
        """The next file: reset + records + phase A launched on the engine whose turn it is, the file before finished on
        the other one meanwhile, then this file up to its merged partial results.  Returns the profile of the file BEFORE
        (None: there was none, or it had no mapped record); `flush()` returns this file's."""
        e = self.engines[self.k % len(self.engines)]
        self.k += 1
        e.reset()
        if not self.directory_mode or self.k == 1:
            e.reset_cutoffs()        # every step is a fresh file for a fresh `slimm` object
        elif e is not self.last:
            e.set_cutoff_cache(*self.last.cutoff_cache())   # Q8: the one object of the -d loop keeps its cut-offs ...
            e.set_min_reads(int(self.last.stats()["min_reads"]))   # ... and the min_reads its first file derived
        self.give(e)
        self.last = e
        had_pending = self.pending is not None
        have_hits = sharded_profile_begin(e, self.device, self.group, self.phase_times, self.exchange,
                                          after_launch=self._finish_pending)
        before = self.profile if had_pending else None
        if have_hits:
            self.pending = e
        else:
            self.profile = None      # (src/slimm.hpp:451-455: no mapped reads, nothing written for this file)
        return before

    def flush(self):
        """Finishes the last file; returns its profile (None: no mapped record)."""
        self._finish_pending()
        return self.profile

    # the engines' kernel timers as one
    def enable_kernel_timing(self, on):
        for e in self.engines:
            e.enable_kernel_timing(on)

    def time_only_kernel(self, name):
        for e in self.engines:
            e.time_only_kernel(name)

    def kernel_times(self, reset=False):
        out = {}
        for e in self.engines:
            for k, (ms, n) in e.kernel_times(reset=reset).items():
                a = out.get(k, (0.0, 0))
                out[k] = (a[0] + ms, a[1] + n)
        return out
