"""The by-read partitioner (slimm_amd/partition.py): shards are a partition of the stream and split no read name."""
import numpy as np
import pytest

from slimm_amd.partition import KEY_MASK, chunk_owner, contiguous_cuts, owner_by_key, run_starts, shard_records
from slimm_amd.synth import CONFIGS, make_workload
from slimm_amd.workload import Records


def _names_per_shard(shards):
    return [set((s.read_key & KEY_MASK).tolist()) for s in shards]


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("grouped", [True, False])
def test_shards_partition_the_stream_and_split_no_read(world, grouped):
    w = make_workload(CONFIGS["config1"], seed=31, shuffled=not grouped)
    rec = w.records
    shards = [shard_records(rec, r, world, grouped)[0] for r in range(world)]
    assert sum(len(s) for s in shards) == len(rec)
    # union = stream (as multisets of records)
    def rows(r):
        return np.stack([r.read_key.astype(np.uint64), r.flag.astype(np.uint64), r.ref_id.astype(np.int64).astype(np.uint64),
                         r.begin_pos.astype(np.int64).astype(np.uint64)], axis=1)
    allrows = np.concatenate([rows(s) for s in shards])
    assert np.array_equal(np.sort(allrows.view("u8,u8,u8,u8"), axis=0), np.sort(rows(rec).view("u8,u8,u8,u8"), axis=0))
    # no read name on two ranks
    names = _names_per_shard(shards)
    for a in range(world):
        for b in range(a + 1, world):
            assert not (names[a] & names[b])
    if grouped:  # contiguous slices in file order, still grouped
        assert np.array_equal(np.concatenate([s.read_key for s in shards]), rec.read_key)
        for s in shards:
            k = s.read_key & KEY_MASK
            starts = np.nonzero(run_starts(s.read_key))[0]
            assert len(set(k[starts].tolist())) == len(starts)  # every name is one run
    # balance: within one read of the ideal for contiguous cuts, statistical for the hash
    sizes = np.array([len(s) for s in shards])
    if grouped:
        longest = int(np.diff(np.concatenate([np.nonzero(run_starts(rec.read_key))[0], [len(rec)]])).max())
        assert np.all(np.abs(sizes - len(rec) / world) <= longest + 1)
    else:
        assert sizes.min() > 0.8 * len(rec) / world


def test_cuts_edge_cases():
    assert contiguous_cuts(np.zeros(0, dtype=np.uint64), 4).tolist() == [0, 0, 0, 0, 0]
    one_read = np.full(10, 7, dtype=np.uint64)                      # a single run: everything on rank 0... or the last
    c = contiguous_cuts(one_read, 3)
    assert c[0] == 0 and c[-1] == 10 and all(x in (0, 10) for x in c)
    k = np.array([1, 1, 2, 2, 2, 3], dtype=np.uint64)
    assert contiguous_cuts(k, 2).tolist() == [0, 5, 6]               # 6 // 2 = 3 lies inside run "2": cut behind it
    # the top two bits of a key are not significant
    k2 = k | (np.uint64(3) << np.uint64(62))
    assert contiguous_cuts(k2, 2).tolist() == [0, 5, 6]
    assert np.array_equal(owner_by_key(k2, 5), owner_by_key(k, 5))


def test_chunk_owner_covers_every_chunk_once():
    for world in (1, 2, 3, 8):
        got = [c for r in chunk_owner(100, world) for c in r]
        assert got == list(range(100))
