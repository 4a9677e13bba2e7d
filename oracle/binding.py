"""ctypes binding of the CPU oracle (oracle/slimm_oracle.cpp).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline
leg of bench.py.  Nothing under slimm_amd/ imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libslimm_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "slimm_oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libslimm_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_create.restype = C.c_void_p
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_set_options.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_char_p]
        L.orc_db_add_accessions.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p, C.c_void_p]
        L.orc_db_add_taxa.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_char_p]
        L.orc_avg_read_length.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32]
        L.orc_avg_read_length.restype = C.c_uint32
        L.orc_reset.argtypes = [C.c_void_p]
        L.orc_run.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_char_p,
                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_run.restype = C.c_int
        L.orc_phase_a.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_phase_a.restype = C.c_int
        L.orc_phase_b_with_valid.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_get_scalars.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_get_cutoffs.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_get_ref_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_total_bins.argtypes = [C.c_void_p]
        L.orc_total_bins.restype = C.c_uint64
        L.orc_get_bins.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_taxon_count_size.argtypes = [C.c_void_p, C.c_int]
        L.orc_taxon_count_size.restype = C.c_uint32
        L.orc_get_taxon_counts.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_children_pairs_size.argtypes = [C.c_void_p, C.c_int]
        L.orc_children_pairs_size.restype = C.c_uint64
        L.orc_get_children_pairs.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_text_size.argtypes = [C.c_void_p, C.c_int]
        L.orc_text_size.restype = C.c_uint64
        L.orc_get_text.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        _lib = L
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _blob(strings) -> bytes:
    return ("\n".join(strings)).encode() + b"\0"


@dataclass
class OracleResult:
    no_hits: bool
    scalars: Dict[str, int]
    cutoffs: Tuple[float, float, float]  # coverage, uniq coverage, expected coverage
    reads_count: np.ndarray
    uniq_reads_count: np.ndarray
    uniq_reads_count2: np.ndarray
    nbins: np.ndarray
    nz_cov: np.ndarray
    nz_uniq_cov: np.ndarray
    nz_uniq_cov2: np.ndarray
    valid: np.ndarray
    abundance: np.ndarray
    uniq_abundance: np.ndarray
    cov: np.ndarray  # concatenated, unpadded
    uniq_cov: np.ndarray
    uniq_cov2: np.ndarray
    lca_direct: Dict[int, int]  # taxid -> count after direct LCA hits only
    lca_direct_children: set  # {(taxid, ref)}
    taxon_count: Dict[int, int]  # final
    taxon_children: set
    profile_tsv: str
    raw_tsv: str
    coverage_csv: Tuple[str, str, str]
    phase_seconds: Tuple[float, float, float]

    def profile_rows(self) -> Dict[str, Tuple[float, int, str]]:
        return parse_profile(self.profile_tsv)


def parse_profile(text: str) -> Dict[str, Tuple[float, int, str]]:
    """profile TSV -> {taxa_id column: (abundance, read_count, lineage)}; row order is not significant (Q16)."""
    rows = {}
    lines = text.strip("\n").split("\n")
    assert lines[0] == "taxa_level\ttaxa_id\tlinage\tabundance\tread_count", lines[0]
    for ln in lines[1:]:
        level, tid, lin, ab, cnt = ln.split("\t")
        assert tid not in rows, f"duplicate profile row {tid}"
        rows[tid] = (float(ab), int(cnt), lin)
    return rows


_SCALARS = ["hits", "matches", "uniq_matches", "uniq_hits", "uniq_matches2", "reference_count", "matched_ref_length",
            "failed_by_cov", "failed_by_uniq_cov", "failed_by_min_read", "n_valid", "bin_width", "min_reads",
            "profile_count", "profile_failed"]


class Oracle:
    """One `slimm` object of the reference (one database, options; run() = one input file)."""

    def __init__(self, taxonomy, options):
        L = lib()
        self.h = C.c_void_p(L.orc_create())
        L.orc_set_options(self.h, options.bin_width, options.min_reads, options.cov_cut_off,
                          options.abundance_cut_off, options.rank.encode())
        lin = np.ascontiguousarray(taxonomy.lineage, dtype=np.uint32)
        L.orc_db_add_accessions(self.h, len(taxonomy.accessions), _blob(taxonomy.accessions), _p(lin))
        L.orc_db_add_taxa(self.h, len(taxonomy.tax_name), _p(taxonomy.tax_id), _p(taxonomy.tax_rank),
                          _blob(taxonomy.tax_name))

    def __del__(self):
        try:
            lib().orc_destroy(self.h)
        except Exception:
            pass

    def run(self, ref_names, ref_len, records, avg_read_len, want_raw=True, want_cov=False, use_qnames=True,
            collect_bins=True) -> OracleResult:
        L = lib()
        L.orc_reset(self.h)
        ref_len = np.ascontiguousarray(ref_len, dtype=np.uint32)
        qblob = _blob(records.qname) if (use_qnames and records.qname is not None) else None
        # with names the oracle keys by the reference's string (name + ".1" / ".2"): it gets the flags as written (Q18)
        flag = records.flags_in_file() if qblob is not None else records.flag
        ph = np.zeros(3, dtype=np.float64)
        rc = L.orc_run(self.h, len(ref_names), _blob(ref_names), _p(ref_len), int(avg_read_len), len(records), qblob,
                       _p(records.read_key), _p(flag), _p(records.ref_id), _p(records.begin_pos),
                       int(want_raw), int(want_cov), _p(ph))
        return self._collect(rc, len(ref_names), collect_bins, ph)

    def phase_a(self, ref_names, ref_len, records, avg_read_len) -> OracleResult:
        """analyze_alignments only (for the multi-rank tests); keyed by read_key."""
        L = lib()
        L.orc_reset(self.h)
        ref_len = np.ascontiguousarray(ref_len, dtype=np.uint32)
        rc = L.orc_phase_a(self.h, len(ref_names), _blob(ref_names), _p(ref_len), int(avg_read_len), len(records),
                           _p(records.read_key), _p(records.flag), _p(records.ref_id), _p(records.begin_pos))
        self._n_refs = len(ref_names)
        return self._collect(rc, len(ref_names), True, np.zeros(3), cutoffs=False)

    def phase_b_with_valid(self, valid) -> OracleResult:
        """Per-read filter against an externally decided valid set + direct LCA hits, on this shard's reads."""
        valid = np.ascontiguousarray(valid, dtype=np.uint8)
        lib().orc_phase_b_with_valid(self.h, _p(valid))
        return self._collect(0, self._n_refs, True, np.zeros(3), cutoffs=False)

    def _collect(self, rc, R, collect_bins, ph, cutoffs=True) -> OracleResult:
        L = lib()
        sc = np.zeros(15, dtype=np.uint32)
        L.orc_get_scalars(self.h, _p(sc))
        cut = np.zeros(3, dtype=np.float32)
        if rc == 0 and cutoffs:
            L.orc_get_cutoffs(self.h, _p(cut))
        u = np.zeros((R, 8), dtype=np.uint32)
        f = np.zeros((R, 2), dtype=np.float32)
        L.orc_get_ref_stats(self.h, _p(u), _p(f))
        B = L.orc_total_bins(self.h)
        bins = []
        for w in range(3):
            b = np.zeros(B if collect_bins else 0, dtype=np.uint32)
            if collect_bins:
                L.orc_get_bins(self.h, w, _p(b))
            bins.append(b)

        def counts(stage):
            n = L.orc_taxon_count_size(self.h, stage)
            t = np.zeros(n, dtype=np.uint32)
            c = np.zeros(n, dtype=np.uint32)
            L.orc_get_taxon_counts(self.h, stage, _p(t), _p(c))
            return {int(a): int(b) for a, b in zip(t, c)}

        def children(stage):
            n = L.orc_children_pairs_size(self.h, stage)
            t = np.zeros(n, dtype=np.uint32)
            r = np.zeros(n, dtype=np.uint32)
            L.orc_get_children_pairs(self.h, stage, _p(t), _p(r))
            return set(zip(t.tolist(), r.tolist()))

        def text(which):
            n = L.orc_text_size(self.h, which)
            buf = C.create_string_buffer(n)
            L.orc_get_text(self.h, which, buf)
            return buf.raw.decode()

        return OracleResult(
            no_hits=(rc == 1), scalars={k: int(v) for k, v in zip(_SCALARS, sc)},
            cutoffs=(float(cut[0]), float(cut[1]), float(cut[2])),
            reads_count=u[:, 0].copy(), uniq_reads_count=u[:, 1].copy(), uniq_reads_count2=u[:, 2].copy(),
            nbins=u[:, 3].copy(), nz_cov=u[:, 4].copy(), nz_uniq_cov=u[:, 5].copy(), nz_uniq_cov2=u[:, 6].copy(),
            valid=u[:, 7].copy(), abundance=f[:, 0].copy(), uniq_abundance=f[:, 1].copy(),
            cov=bins[0], uniq_cov=bins[1], uniq_cov2=bins[2],
            lca_direct=counts(0), lca_direct_children=children(0), taxon_count=counts(1), taxon_children=children(1),
            profile_tsv=text(0), raw_tsv=text(1), coverage_csv=(text(2), text(3), text(4)),
            phase_seconds=(float(ph[0]), float(ph[1]), float(ph[2])))


def run_workload(w, **kw) -> OracleResult:
    """Run one Workload (slimm_amd.workload.Workload) through a fresh oracle."""
    return Oracle(w.taxonomy, w.options).run(w.ref_names, w.ref_len, w.records, w.avg_read_len, **kw)


def avg_read_length(l_seq: np.ndarray, sample: int = 100000) -> int:
    l_seq = np.ascontiguousarray(l_seq, dtype=np.uint32)
    return int(lib().orc_avg_read_length(_p(l_seq), l_seq.shape[0], sample))


# ---------------------------------------------------------------------------------------------------------------
# the all-core dense restatement (oracle/slimm_dense_mt.cpp): bench.py's cpu_baseline_mt leg, checked against the
# oracle above in tests/test_dense_mt.py
_mt_lib = None


def dense_mt_lib():
    global _mt_lib
    if _mt_lib is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libslimm_dense_mt.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `make -C oracle`")
        L = C.CDLL(path)
        L.dmt_run2.restype = C.c_int
        L.dmt_run2.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p,
                               C.c_uint32, C.c_uint32, C.c_float, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.dmt_run3.restype = C.c_int
        L.dmt_run3.argtypes = L.dmt_run2.argtypes + [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        L.dmt_profile.restype = C.c_int
        L.dmt_profile.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float,
                                  C.c_float, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        _mt_lib = L
    return _mt_lib


BIN_CHECKSUM_MUL = 0x9E3779B97F4A7C15


def bin_checksum(a: np.ndarray) -> int:
    """sum_i a[i] * ((i + 1) * 0x9E3779B97F4A7C15) mod 2^64 -- the position-weighted checksum dmt_run2 reports for its
    three coverage arrays, here over an array a caller holds (slimm_get_bins); in pieces, so that 200 M bins do not
    need 5 GB of temporaries."""
    a = np.ascontiguousarray(a, dtype=np.uint32)
    total = np.uint64(0)
    step = 1 << 24
    with np.errstate(over="ignore"):
        for s in range(0, a.shape[0], step):
            e = min(a.shape[0], s + step)
            wgt = np.arange(s + 1, e + 1, dtype=np.uint64) * np.uint64(BIN_CHECKSUM_MUL)
            total = total + (a[s:e].astype(np.uint64) * wgt).sum(dtype=np.uint64)
    return int(total)


def dense_mt_run(w, records=None, threads: int = 0, want_bins: bool = False, want_profile: bool = False) -> dict:
    """Phases A, B and the direct LCA counts of workload `w` (records grouped by read name) on `threads` host threads
    (0 = all).  Returns per-reference columns, scalars, {taxid: count}, the checksums of the three coverage arrays
    (bin_checksum) -- with want_bins the arrays themselves -- and the three phase times.  want_profile: also the scalar
    tail from the reference (dmt_profile: src/slimm.hpp:560-610, 733-843): "taxon_count", "taxon_children" (the propagated
    counts and children sets) and "profile" = {taxa_id column: (abundance, read count)}."""
    rec = w.records if records is None else records
    R = len(w.ref_names)
    lineage = np.ascontiguousarray(w.taxonomy.lineage_for_header(w.ref_names), dtype=np.uint32)
    ref_len = np.ascontiguousarray(w.ref_len, dtype=np.uint32)
    cols = np.zeros((R, 5), dtype=np.uint32)
    sc = np.zeros(8, dtype=np.uint64)
    cap = 1 << 20
    tx = np.zeros(cap, dtype=np.uint32)
    cn = np.zeros(cap, dtype=np.uint32)
    n_lca = C.c_uint32(0)
    sec = np.zeros(3, dtype=np.float64)
    threads = threads or (os.cpu_count() or 1)
    chk = np.zeros(3, dtype=np.uint64)
    bins = None
    out_ptrs = None
    if want_bins:
        bw = int(w.options.bin_width) or int(w.avg_read_len)
        B = int((ref_len.astype(np.int64) // bw + 1).sum())
        bins = [np.zeros(B, dtype=np.uint32) for _ in range(3)]
        out_ptrs = (C.c_void_p * 3)(*[b.ctypes.data for b in bins])
    pair_cap = 1 << 22
    pairs = np.zeros(pair_cap if want_profile else 1, dtype=np.uint64)
    n_pairs = C.c_uint64(0)
    cuts = np.zeros(2, dtype=np.float32)
    rc = dense_mt_lib().dmt_run3(_p(rec.read_key), _p(rec.flag), _p(rec.ref_id), _p(rec.begin_pos), len(rec), R, _p(ref_len),
                                 _p(lineage), int(w.avg_read_len), int(w.options.bin_width), float(w.options.cov_cut_off),
                                 int(threads), _p(cols), _p(sc), _p(tx), _p(cn), cap, C.byref(n_lca), _p(sec), out_ptrs,
                                 _p(chk), _p(pairs) if want_profile else None, pair_cap, C.byref(n_pairs), _p(cuts))
    if rc < 0:
        raise RuntimeError("dmt_run3: bad arguments")
    k = min(int(n_lca.value), cap)
    out = {"no_hits": rc == 1, "reads_count": cols[:, 0].copy(), "uniq_reads_count": cols[:, 1].copy(),
           "nz_cov": cols[:, 2].copy(), "nz_uniq_cov": cols[:, 3].copy(), "uniq_reads_count2": cols[:, 4].copy(),
           "hits": int(sc[0]), "matches": int(sc[1]), "uniq_matches": int(sc[2]), "uniq_matches2": int(sc[3]),
           "n_valid": int(sc[4]), "total_bins": int(sc[5]), "lca_direct": {int(a): int(b) for a, b in zip(tx[:k], cn[:k])},
           "checksums": tuple(int(x) for x in chk), "seconds": tuple(float(x) for x in sec), "threads": int(threads)}
    if bins is not None:
        out["cov"], out["uniq_cov"], out["uniq_cov2"] = bins
    out["cutoffs"] = (float(cuts[0]), float(cuts[1]))
    if want_profile and rc == 0:
        assert n_pairs.value <= pair_cap and n_lca.value <= cap
        t = w.taxonomy
        ranks = ["strain", "species", "genus", "family", "order", "class", "phylum", "superkingdom"]
        rank = ranks.index(w.options.rank)          # considered_ranks = {rank + 1, rank} (src/slimm.hpp:498-514)
        named = np.array([1 if nm else 0 for nm in t.tax_name], dtype=np.uint8)
        ocap = 1 << 20
        ot, oc = np.zeros(ocap, dtype=np.uint32), np.zeros(ocap, dtype=np.uint32)
        op = np.zeros(1 << 24, dtype=np.uint64)
        rt, rs = np.zeros(ocap, dtype=np.uint32), np.zeros(ocap, dtype=np.uint8)
        ra, rr = np.zeros(ocap, dtype=np.float32), np.zeros(ocap, dtype=np.uint32)
        n_out, n_op, n_rows = C.c_uint32(0), C.c_uint64(0), C.c_uint32(0)
        u2 = np.ascontiguousarray(out["uniq_reads_count2"], dtype=np.uint32)
        dense_mt_lib().dmt_profile(R, _p(lineage), _p(ref_len), _p(u2), _p(tx), _p(cn), k, _p(pairs), int(n_pairs.value),
                                   _p(np.ascontiguousarray(t.tax_id, dtype=np.uint32)), _p(np.ascontiguousarray(t.tax_rank, dtype=np.uint32)),
                                   _p(named), len(t.tax_name), out["matches"] & 0xffffffff, int(w.avg_read_len), rank, rank + 1,
                                   float(w.options.abundance_cut_off), float(cuts[0]), _p(ot), _p(oc), ocap, C.byref(n_out), _p(op),
                                   op.shape[0], C.byref(n_op), _p(rt), _p(rs), _p(ra), _p(rr), ocap, C.byref(n_rows))
        assert n_out.value <= ocap and n_op.value <= op.shape[0] and n_rows.value <= ocap
        out["taxon_count"] = {int(a): int(b) for a, b in zip(ot[:n_out.value], oc[:n_out.value])}
        pp = op[:n_op.value]
        out["taxon_children"] = set(zip((pp >> np.uint64(32)).astype(np.uint32).tolist(), (pp & np.uint64(0xffffffff)).astype(np.uint32).tolist()))
        out["profile"] = {f"{int(a)}{'*' if s_ else ''}": (float(ab), int(rd))
                          for a, s_, ab, rd in zip(rt[:n_rows.value], rs[:n_rows.value], ra[:n_rows.value], rr[:n_rows.value])}
    return out
