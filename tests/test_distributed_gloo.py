"""world_size-2 tests of the sharded driver (slimm_amd/distributed.py) over gloo on CPU.

Each rank holds a shard of the reads; the result must equal the single-process oracle on the whole stream.
"""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _split_by_read(w, world):
    """Shards cut so that all records of a read name stay on one rank (hash of the key)."""
    from slimm_amd.workload import Workload

    owner = (w.records.read_key % np.uint64(world)).astype(np.int64)
    out = []
    for r in range(world):
        idx = np.nonzero(owner == r)[0]
        out.append(Workload(w.ref_names, w.ref_len, w.taxonomy, w.records.take(idx), w.avg_read_len, w.options,
                            f"{w.name}-shard{r}"))
    return out


def _worker(rank, world, port, case, tmp, exchange="summary", cuts="hash"):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.binding import run_workload
        from slimm_amd.distributed import sharded_profile
        from slimm_amd.synth import CONFIGS, SynthConfig, make_workload
        from tests.helpers import assert_matches_oracle
        from tests.shard_engine import OracleShardEngine

        if case == "config1":
            w = make_workload(CONFIGS["config1"], seed=21)
        elif case == "cross":
            cfg = SynthConfig("x", 60_000, 10_000, 5.0, present_frac=0.3, len_lo=20_000, len_hi=60_000)
            w = make_workload(cfg, seed=22)
            rng = np.random.default_rng(1)
            m = (w.records.ref_id >= 0) & (rng.random(len(w.records)) < 0.2)
            w.records.ref_id[m] = rng.integers(0, cfg.n_refs, size=int(m.sum()), dtype=np.int32)
            w.records.begin_pos[m] = 100
        else:
            w = make_workload(CONFIGS["config2"], seed=23, n_records=150_000)
        if cuts == "contiguous":   # the partitioner of the library: slices of the grouped file cut at qName runs
            from slimm_amd.partition import shard_records
            from slimm_amd.workload import Workload

            rec, still_grouped = shard_records(w.records, rank, world, grouped=True)
            assert still_grouped
            shard = Workload(w.ref_names, w.ref_len, w.taxonomy, rec, w.avg_read_len, w.options, f"{w.name}-cut{rank}")
        else:
            shard = _split_by_read(w, world)[rank]
        eng = OracleShardEngine(shard)
        text = sharded_profile(eng, None, os.path.join(tmp, "profile.tsv"), exchange=exchange)
        whole = run_workload(w, use_qnames=False, collect_bins=False)
        assert text is not None
        assert_matches_oracle(eng.host, whole, bins=False)
        if rank == 0:
            assert open(os.path.join(tmp, "profile.tsv")).read() == text
        if case == "cross":
            assert len(eng.get_partials()["pairs"]) > 0
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case,exchange", [("config1", "summary"), ("config2", "summary"), ("cross", "summary"),
                                           ("config1", "bins"), ("config2", "sliced")])
def test_two_ranks_equal_single_process(case, exchange):
    port = (29500 + (os.getpid() % 2000) + {"config1": 0, "config2": 1, "cross": 2}[case]
            + {"summary": 0, "bins": 3, "sliced": 4}[exchange])
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker, args=(2, port, case, tmp, exchange), nprocs=2, join=True)


@pytest.mark.parametrize("case", ["config1", "config2"])
def test_two_ranks_on_contiguous_cuts_of_the_file(case):
    """slimm_amd/partition.py: rank r keeps records [c_r, c_r+1) of the grouped stream, cut at qName-run boundaries."""
    port = 29500 + (os.getpid() % 2000) + 20 + {"config1": 0, "config2": 1}[case]
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker, args=(2, port, case, tmp, "summary", "contiguous"), nprocs=2, join=True)


def test_three_ranks_auto_exchange_is_sliced():
    """world_size 3: "auto" picks the all-to-all form; slices of unequal fill (the bins do not divide by 3)."""
    port = 29500 + (os.getpid() % 2000) + 9
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker, args=(3, port, "config2", tmp, "auto"), nprocs=3, join=True)


def _worker_back_to_back(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.binding import run_workload
        from slimm_amd.distributed import FilesBackToBack
        from slimm_amd.synth import CONFIGS, make_workload
        from tests.shard_engine import OracleShardEngine

        # three different files (one database: the taxonomy and the references of a seed do not depend on the record count)
        files = [make_workload(CONFIGS["config1"], seed=31, n_records=n) for n in (40_000, 25_000, 60_000)]
        shards = [_split_by_read(w, world)[rank] for w in files]

        class Engine(OracleShardEngine):     # a context that is reset and handed another file, like Slimm
            def __init__(self):
                self.ready = False

            def reset(self):
                self.ready = False

            def reset_cutoffs(self):
                pass

        at = [0]

        def give(e):
            OracleShardEngine.__init__(e, shards[at[0]])
            at[0] += 1

        path = os.path.join(tmp, "profile.tsv")
        fb = FilesBackToBack([Engine(), Engine()], give, None, path, exchange="summary")
        got = []
        for k in range(3):
            before = fb.step()               # returns the profile of the file BEFORE this one
            if k:
                got.append(before)
                if rank == 0:                # ... which has been written by now
                    assert open(path).read() == before
            dist.barrier()
        got.append(fb.flush())
        assert fb.flush() == got[-1]         # (nothing pending any more)
        from tests.helpers import assert_profiles_match
        assert len(set(got)) == 3
        for text, w in zip(got, files):
            assert_profiles_match(text, run_workload(w, use_qnames=False, collect_bins=False).profile_tsv)
        if rank == 0:
            assert open(path).read() == got[-1]
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_files_back_to_back_through_two_engines_per_rank():
    """slimm_amd/distributed.py FilesBackToBack (bench.py's timed loop): while a rank's second engine is in phase A of file
    k + 1, the host-only end of file k runs on the first -- the collectives of the two files must not interleave
    differently on different ranks, and every file's profile must be its own."""
    port = 29500 + (os.getpid() % 2000) + 33
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker_back_to_back, args=(2, port, tmp), nprocs=2, join=True)


def _unequal_split(w, world):
    """Shards of very unequal size with one EMPTY rank: reads go to rank (key mod 61) mod world', where world' skips rank
    world - 3 altogether and ranks get 1, 2, 3 ... shares -- the shapes an 8-GPU run meets when a file's reads are few."""
    from slimm_amd.workload import Workload

    empty = world - 3
    live = [r for r in range(world) if r != empty]
    shares = np.repeat(np.arange(len(live)), np.arange(1, len(live) + 1))      # rank k of `live` owns k + 1 shares
    owner = np.array(live, dtype=np.int64)[shares[(w.records.read_key % np.uint64(len(shares))).astype(np.int64)]]
    out = []
    for r in range(world):
        idx = np.nonzero(owner == r)[0]
        out.append(Workload(w.ref_names, w.ref_len, w.taxonomy, w.records.take(idx), w.avg_read_len, w.options, f"{w.name}-u{r}"))
    assert len(out[empty].records) == 0 and len({len(s.records) for s in out}) == world
    return out


def _worker_eight(rank, world, port, tmp, exchange, cuts):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.binding import run_workload
        from slimm_amd.distributed import sharded_profile
        from slimm_amd.partition import chunk_owner, shard_records
        from slimm_amd.synth import CONFIGS, make_workload
        from slimm_amd.workload import Workload
        from tests.helpers import assert_matches_oracle
        from tests.shard_engine import OracleShardEngine

        w = make_workload(CONFIGS["config2"], seed=29, n_records=120_000)
        if cuts == "contiguous":
            rec, still_grouped = shard_records(w.records, rank, world, grouped=True)
            assert still_grouped
            shard = Workload(w.ref_names, w.ref_len, w.taxonomy, rec, w.avg_read_len, w.options, f"{w.name}-cut{rank}")
        else:
            shard = _unequal_split(w, world)[rank]
        eng = OracleShardEngine(shard)
        text = sharded_profile(eng, None, os.path.join(tmp, "profile.tsv"), exchange=exchange)
        whole = run_workload(w, use_qnames=False, collect_bins=False)
        assert text is not None
        assert_matches_oracle(eng.host, whole, bins=False)
        if rank == 0:
            assert open(os.path.join(tmp, "profile.tsv")).read() == text
            # bench.py's chunk ownership at this world size: 100 chunks over 8 ranks = 12 or 13 each, all of them, in order
            own = chunk_owner(100, world)
            assert [len(o) for o in own] == [12, 13, 12, 13, 12, 13, 12, 13] and [c for o in own for c in o] == list(range(100))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exchange,cuts", [("auto", "unequal"), ("sliced", "contiguous"), ("summary", "unequal"), ("bins", "unequal")])
def test_eight_ranks_unequal_shards_and_an_empty_rank(exchange, cuts):
    """The world size the driver's node has: eight ranks through the real driver code over gloo -- the
    all-to-all of EIGHT bitmap slices ("auto" above two ranks), the all-gather and the bins all-reduce --, with shards of
    eight different sizes of which one is EMPTY (a rank without a single record still takes part in every collective), and
    on the partitioner's contiguous cuts."""
    port = 29500 + (os.getpid() % 2000) + 40 + {"auto": 0, "sliced": 1, "summary": 2, "bins": 3}[exchange]
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker_eight, args=(8, port, tmp, exchange, cuts), nprocs=8, join=True)
