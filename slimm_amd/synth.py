"""Seeded synthetic SLIMM inputs shaped like BASELINE.json's configs.

There is no network and the reference ships no BAM, so every benchmark/parity input is generated:
a taxonomy with the lineage shape slimm_build produces (reference src/slimm_build.cpp:283-344: 8 slots,
own taxid first, superkingdom last), bacterial-size contigs, and a record stream in mapper order
(all records of a read contiguous) with multi-mapped reads hitting taxonomic neighbours, mates,
unmapped records and repeated (read, ref) pairs.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np

from .workload import Options, Records, Taxonomy, Workload, RANKS


@dataclass(frozen=True)
class SynthConfig:
    name: str
    n_records: int
    n_refs: int
    mean_hits: float
    bin_width: int = 1000
    read_len: int = 100
    strain_level: bool = False
    present_frac: float = 0.04
    len_lo: int = 2_000_000
    len_hi: int = 6_000_000
    cov_cut_off: float = 0.95


# BASELINE.json "configs" (index = position there)
CONFIGS = {
    "config1": SynthConfig("config1", 10_000, 16, 1.6, bin_width=100, len_lo=5_000, len_hi=50_000, present_frac=0.6),
    "config2": SynthConfig("config2", 10_000_000, 5_000, 3.0),
    "config3": SynthConfig("config3", 100_000_000, 20_000, 8.0),
    "config4": SynthConfig("config4", 1_000_000_000, 20_000, 8.0),
    "config5": SynthConfig("config5", 100_000_000, 50_000, 40.0, strain_level=True),
}


def synth_taxonomy(n_refs: int, strain_level: bool = False, hole_every: int = 0):
    """Accessions ACC000000..; lineage columns own, species, genus, family, order, class, phylum, superkingdom.

    A consistent tree: ~2 refs/species, 10/genus, 50/family, 200/order, 1000/class, 2000/phylum; every fourth phylum
    is archaeal.  strain_level: 2 contigs per strain and 10 refs per species (LCA at levels 0/1 becomes frequent).
    hole_every > 0 blanks the species slot of every hole_every-th species group (quirk Q5 material).
    """
    i = np.arange(n_refs, dtype=np.int64)
    sp_div = 10 if strain_level else 2
    own = 10_000_000 + (i // 2 if strain_level else i)
    lin = np.stack([
        own,
        1_000_000 + i // sp_div,
        500_000 + i // (sp_div * 5),
        200_000 + i // (sp_div * 25),
        100_000 + i // (sp_div * 100),
        50_000 + i // (sp_div * 500),
        10_000 + i // (sp_div * 1000),
        np.where((i // (sp_div * 1000)) % 4 == 3, 2157, 2),
    ], axis=1).astype(np.uint32)
    if hole_every:
        lin[(i // sp_div) % hole_every == hole_every - 1, 1] = 0
    accs = [f"ACC{k:06d}" for k in range(n_refs)]
    tid, rk = [], []
    for lv in range(7, -1, -1):  # lower columns last so they win where ids coincide (they do not, by construction)
        col = np.unique(lin[:, lv])
        col = col[col != 0]
        tid.append(col)
        rk.append(np.full(col.shape, lv, dtype=np.uint32))
    tid = np.concatenate(tid)
    rk = np.concatenate(rk)
    names = [f"{RANKS[r]}_{t}" for t, r in zip(tid.tolist(), rk.tolist())]
    return Taxonomy(accs, lin, tid, rk, names)


def make_workload(cfg: SynthConfig, seed: int = 1, n_records: Optional[int] = None, shuffled: bool = False,
                  unmapped_frac: float = 0.02, paired_frac: float = 0.30, repeat_frac: float = 0.05,
                  hole_every: int = 0, sample_seed: Optional[int] = None, shard: int = 0) -> Workload:
    """One input file.  `sample_seed` (default: seed) fixes the header, database and sample composition;
    `seed` fixes the record stream, so several shards of one sample share everything but their reads
    (`shard` keeps read identities disjoint between them)."""
    rng_s = np.random.Generator(np.random.PCG64(seed if sample_seed is None else sample_seed))
    rng = np.random.Generator(np.random.PCG64([seed, 0x5eed]))
    N = int(n_records if n_records is not None else cfg.n_records)
    R = cfg.n_refs
    tax = synth_taxonomy(R, cfg.strain_level, hole_every)
    ref_len = rng_s.integers(cfg.len_lo, cfg.len_hi, size=R, dtype=np.int64).astype(np.uint32)
    ref_names = [a + ".1" for a in tax.accessions]

    # organisms present in the sample, Zipf-like abundances
    n_present = max(4, int(R * cfg.present_frac))
    present = rng_s.choice(R, size=n_present, replace=False)
    wts = 1.0 / np.arange(1, n_present + 1) ** 0.8
    wts /= wts.sum()

    # reads: hit counts 1 + geometric, until N records are covered
    mean_records_per_read = cfg.mean_hits * (1.0 + repeat_frac)
    n_reads = int(N / mean_records_per_read * 1.05) + 16
    hits = rng.geometric(1.0 / cfg.mean_hits, size=n_reads).astype(np.int64)  # >= 1, mean = mean_hits
    unmapped = rng.random(n_reads) < unmapped_frac
    hits[unmapped] = 1
    home = present[rng.choice(n_present, size=n_reads, p=wts)].astype(np.int64)
    # read identity: a bijection of the read index into 62 bits, so distinct reads never share a key
    key = ((np.arange(n_reads, dtype=np.uint64) + np.uint64(shard << 40)) * np.uint64(0x9E3779B97F4A7C15)) \
        & np.uint64((1 << 62) - 1)
    # mates: a paired read name = two consecutive reads sharing the key, flags 0x41 / 0x81
    mate = np.zeros(n_reads, dtype=np.uint16)
    first_of_pair = (rng.random(n_reads) < paired_frac / 2)
    first_of_pair[-1] = False
    idx = np.nonzero(first_of_pair)[0]
    idx = idx[~first_of_pair[np.minimum(idx + 1, n_reads - 1)]]  # keep pairs disjoint
    mate[idx] = 0x41
    mate[idx + 1] = 0x81
    key[idx + 1] = key[idx]

    start = np.concatenate([[0], np.cumsum(hits)])
    total = int(start[-1])
    rd = np.repeat(np.arange(n_reads, dtype=np.int64), hits)
    k = np.arange(total, dtype=np.int64) - start[rd]
    spread = 6 if not cfg.strain_level else 12
    off = rng.integers(-spread, spread + 1, size=total)
    ref = np.where(k == 0, home[rd], np.clip(home[rd] + off, 0, R - 1))
    pos = (rng.random(total) * (ref_len[ref].astype(np.float64) - cfg.read_len)).astype(np.int64)
    flag = mate[rd] | np.where(k > 0, np.uint16(0x100), np.uint16(0)).astype(np.uint16)
    um = unmapped[rd]
    flag = np.where(um, mate[rd] | np.uint16(0x4), flag).astype(np.uint16)
    ref = np.where(um, -1, ref)
    pos = np.where(um, -1, pos)

    # repeated (read, ref) pairs: an extra record to the same reference right after the original, elsewhere on it
    rep = (rng.random(total) < repeat_frac) & ~um
    order = np.argsort(np.concatenate([np.arange(total), np.nonzero(rep)[0] + 0.5]), kind="stable")
    rd_all = np.concatenate([rd, rd[rep]])[order]
    ref_all = np.concatenate([ref, ref[rep]])[order]
    pos2 = (rng.random(int(rep.sum())) * (ref_len[ref[rep]].astype(np.float64) - cfg.read_len)).astype(np.int64)
    pos_all = np.concatenate([pos, pos2])[order]
    flag_all = np.concatenate([flag, flag[rep] | np.uint16(0x100)])[order]

    n_have = rd_all.shape[0]
    if n_have < N:
        raise RuntimeError(f"generator produced {n_have} < {N} records; raise the read estimate")
    sl = slice(0, N)
    rec = Records(key[rd_all[sl]], flag_all[sl].astype(np.uint16), ref_all[sl].astype(np.int32),
                  pos_all[sl].astype(np.int32))
    if shuffled:
        perm = rng.permutation(N)
        rec = rec.take(perm)
    opts = Options(bin_width=cfg.bin_width, cov_cut_off=cfg.cov_cut_off)
    return Workload(ref_names, ref_len, tax, rec, avg_read_len=cfg.read_len, options=opts,
                    name=f"{cfg.name}-seed{seed}" + ("-shuffled" if shuffled else ""), grouped=not shuffled)


def stream_chunk(cfg: SynthConfig, seed: int, c: int, n_stream: int, chunk_records: int) -> Workload:
    """Chunk c of THE strong-scaling stream of a configuration (bench.py --config config4, the full-size tests): a grouped
    file of whole reads of its own, make_workload(seed + 1000 c, shard = c) of the sample `seed`; the stream is the chunks
    in order."""
    return make_workload(cfg, seed=seed + 1000 * c, n_records=min(chunk_records, n_stream - c * chunk_records),
                         sample_seed=seed, shard=c)


def stream_chunks(cfg: SynthConfig, seed: int, n_stream: int, chunk_records: int = 10_000_000, chunks=None,
                  threads: int = 8):
    """Yields (c, Workload) for the given chunk numbers (default: all of the stream), in order, generated a few chunks
    ahead on `threads` threads (numpy releases the GIL in its sorts and generators; no child processes, so a caller that
    holds a GPU context need not care)."""
    from concurrent.futures import ThreadPoolExecutor

    n_chunks = max(1, (n_stream + chunk_records - 1) // chunk_records)
    todo = list(range(n_chunks) if chunks is None else chunks)
    if threads <= 1 or len(todo) <= 1:
        for c in todo:
            yield c, stream_chunk(cfg, seed, c, n_stream, chunk_records)
        return
    with ThreadPoolExecutor(threads) as ex:
        ahead = []
        it = iter(todo)
        for c in it:
            ahead.append((c, ex.submit(stream_chunk, cfg, seed, c, n_stream, chunk_records)))
            if len(ahead) >= 2 * threads:
                break
        while ahead:
            c, fut = ahead.pop(0)
            nxt = next(it, None)
            if nxt is not None:
                ahead.append((nxt, ex.submit(stream_chunk, cfg, seed, nxt, n_stream, chunk_records)))
            yield c, fut.result()
