"""By-read partitioning of a record stream over G ranks.

Every quantity of the path that is not additive over records is a property of a READ (all records of a qName + mate
number): the first-bin rule of (read, reference) pairs (src/read_stat.hpp:116-135, quirk Q1), the unique / multi
classification (src/slimm.hpp:224-237), the per-read filter and LCA (src/slimm.hpp:380-391, 516-557).  A partition is
therefore valid iff no read name is split; everything else (histograms, per-reference and per-taxon counts) is summed
or ORed across ranks by slimm_amd/distributed.py.

  grouped input (all records of a qName adjacent: mapper output, samtools sort -n / collate)
      contiguous cuts: boundary i is the first qName-run start at or behind record i * N / G.  A rank's shard is a
      slice of the file -- with a BGZF index it reads only its own byte range -- and stays grouped, so phase A takes
      the single-pass front end.
  any other order
      rank = key mod G (the key is a 62-bit hash of the name: uniform).  The shard keeps file order and is declared
      SLIMM_ORDER_ANY (device sort).

The reference has no counterpart (one process reads the whole file).
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

from .workload import Records

KEY_MASK = np.uint64((1 << 62) - 1)  # include/slimm_hip.h: only the low 62 bits of read_key are significant


def run_starts(read_key: np.ndarray) -> np.ndarray:
    """Boolean array: record i starts a qName run (its key differs from record i - 1's)."""
    k = read_key.astype(np.uint64, copy=False) & KEY_MASK
    s = np.ones(len(k), dtype=bool)
    if len(k) > 1:
        s[1:] = k[1:] != k[:-1]
    return s


def contiguous_cuts(read_key: np.ndarray, world: int) -> np.ndarray:
    """world + 1 record indices [0 = c_0 <= c_1 <= ... <= c_world = N]: rank r owns records [c_r, c_r+1).
    Cut i is the first run start at or behind i * N / world (N when there is none), so no qName run is split and the
    shards differ from N / world by less than one run."""
    n = len(read_key)
    cuts = np.zeros(world + 1, dtype=np.int64)
    cuts[world] = n
    if n == 0:
        return cuts
    k = read_key.astype(np.uint64, copy=False) & KEY_MASK
    for i in range(1, world):
        c = (i * n) // world
        # walk forward to the next run start; runs are short (mean hits per read), so this touches a few records
        while 0 < c < n and k[c] == k[c - 1]:
            c += 1
        cuts[i] = max(c, cuts[i - 1])
    return cuts


def owner_by_key(read_key: np.ndarray, world: int) -> np.ndarray:
    """Rank of every record for input in any order: key mod world."""
    return ((read_key.astype(np.uint64, copy=False) & KEY_MASK) % np.uint64(world)).astype(np.int64)


def shard_records(rec: Records, rank: int, world: int, grouped: bool) -> Tuple[Records, bool]:
    """This rank's records and whether they are still grouped by qName."""
    if world <= 1:
        return rec, grouped
    if grouped:
        c = contiguous_cuts(rec.read_key, world)
        return rec.take(np.arange(c[rank], c[rank + 1])), True
    idx = np.nonzero(owner_by_key(rec.read_key, world) == rank)[0]
    return rec.take(idx), False


def chunk_owner(n_chunks: int, world: int) -> List[range]:
    """Strong scaling over a stream given as n_chunks independent grouped chunks (bench.py --config config4: one seeded
    1 B-record stream in 10 M-record chunks): rank r generates and keeps chunks [r * C / G, (r + 1) * C / G)."""
    return [range((r * n_chunks) // world, ((r + 1) * n_chunks) // world) for r in range(world)]
