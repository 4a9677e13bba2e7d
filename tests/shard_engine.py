"""A stand-in engine for the multi-rank tests on CPU (no GPU in the test container).

Same method names as slimm_amd.profiler.Slimm.  The per-read work of this rank's shard (phase A, phase B / C(1)) runs
in the CPU ORACLE; everything the product does on the host (cut-offs, valid set, propagation, profile) runs in the
product's own host-only context.  This lets world_size-2 gloo tests drive slimm_amd.distributed exactly as bench.py
does, and check the sharding + the two exchanges.  Test infrastructure only.
"""
import numpy as np
import torch

from oracle.binding import Oracle
from slimm_amd.profiler import Slimm
from tests.helpers import partials_from_oracle


class OracleShardEngine:
    needs_set_partials = True  # the host-only context never saw this rank's per-read results

    def __init__(self, w):
        self.w = w
        self.orc = Oracle(w.taxonomy, w.options)
        self.host = Slimm.for_workload(w, device=-1)
        self.buf = None

    def analyze_alignments(self):
        w = self.w
        a = self.orc.phase_a(w.ref_names, w.ref_len, w.records, w.avg_read_len)
        self.nbins = a.nbins.astype(np.int64)
        tail = np.zeros(16, dtype=np.uint32)
        tail[0], tail[1] = a.scalars["hits"], a.scalars["matches"]
        self.B = int(a.cov.shape[0])
        self.buf = torch.from_numpy(np.concatenate([a.cov, a.uniq_cov, tail]).view(np.int32).copy())

    def coverage_tensor(self):
        return self.buf

    def coverage_summary_tensor(self):
        if getattr(self, "n_slices", 1) > 1:
            return self._sliced_summary()
        b = self.buf.numpy().view(np.uint32)
        cov, ucov, tail = b[:self.B], b[self.B:2 * self.B], b[2 * self.B:]
        off = np.concatenate([[0], np.cumsum(self.nbins)])[:-1]
        sums = np.stack([np.add.reduceat(cov.astype(np.uint64), off), np.add.reduceat(ucov.astype(np.uint64), off)],
                        axis=1).astype(np.uint32).reshape(-1)
        bits = np.concatenate([np.packbits(cov != 0, bitorder="little"), np.packbits(ucov != 0, bitorder="little")])
        pad = (-bits.shape[0]) % 4
        self._nbits_bytes = (self.B + 7) // 8
        words = np.concatenate([bits, np.zeros(pad, dtype=np.uint8)]).view(np.uint32)
        return torch.from_numpy(np.concatenate([sums, tail[:16], words]).view(np.int32).copy())

    # ---- all-to-all ("sliced") form: the bins are cut into n equal slices of whole 32-bit bitmap words ----
    def prepare_summary(self, n_slices=1):
        self.n_slices = int(n_slices)

    def summary_head_words(self):
        return 2 * self.nbins.shape[0] + 16

    def _slice_words(self):
        words = (self.B + 31) // 32
        return (words + self.n_slices - 1) // self.n_slices

    def _sliced_summary(self):
        b = self.buf.numpy().view(np.uint32)
        cov, ucov, tail = b[:self.B], b[self.B:2 * self.B], b[2 * self.B:]
        off = np.concatenate([[0], np.cumsum(self.nbins)])[:-1]
        sums = np.stack([np.add.reduceat(cov.astype(np.uint64), off), np.add.reduceat(ucov.astype(np.uint64), off)],
                        axis=1).astype(np.uint32).reshape(-1)
        sw = self._slice_words()
        total_bits = sw * self.n_slices * 32

        def words(x):
            bits = np.zeros(total_bits, dtype=np.uint8)
            bits[:self.B] = x != 0
            return np.packbits(bits, bitorder="little").view(np.uint32).reshape(self.n_slices, sw)

        chunks = np.concatenate([words(cov), words(ucov)], axis=1).reshape(-1)   # slice j: [cov words | ucov words]
        self._own_head = np.concatenate([sums, tail[:16]])
        return torch.from_numpy(np.concatenate([self._own_head, chunks]).view(np.int32).copy())

    def merge_summary_slices(self, received, n_ranks, rank):
        sw = self._slice_words()
        g = received.numpy().view(np.uint32).reshape(n_ranks, 2, sw)
        ored = np.bitwise_or.reduce(g, axis=0)
        R = self.nbins.shape[0]
        lo = rank * sw * 32
        off = np.concatenate([[0], np.cumsum(self.nbins)])
        vec = np.zeros(4 * R + 16, dtype=np.uint32)
        vec[0:4 * R:4] = self._own_head[0:2 * R:2]
        vec[2:4 * R:4] = self._own_head[1:2 * R:2]
        vec[4 * R:] = self._own_head[2 * R:]
        for a in range(2):
            bits = np.unpackbits(ored[a].view(np.uint8), bitorder="little")
            full = np.zeros(max(self.B, lo + bits.shape[0]), dtype=np.uint8)
            full[lo:lo + bits.shape[0]] = bits                     # this rank's slice only, at its place
            nz = np.add.reduceat(full[:self.B].astype(np.uint64), off[:-1]).astype(np.uint32)
            nz[self.nbins == 0] = 0
            vec[1 + 2 * a:4 * R:4] = nz
        self._vec = torch.from_numpy(vec.view(np.int32).copy())
        return self._vec

    def finish_coverage_reduced(self):
        v = self._vec.numpy().view(np.uint32)
        R = self.nbins.shape[0]
        return self.host.set_coverage_columns(v[0:4 * R:4], v[2:4 * R:4], v[1:4 * R:4], v[3:4 * R:4],
                                              int(v[4 * R]), int(v[4 * R + 1]))

    def finish_coverage_merged(self, gathered, n_ranks):
        g = gathered.numpy().view(np.uint32).reshape(n_ranks, -1)
        R = self.nbins.shape[0]
        sums = g[:, :2 * R].astype(np.uint64).sum(axis=0).astype(np.uint32).reshape(R, 2)
        tail = g[:, 2 * R:2 * R + 16].astype(np.uint64).sum(axis=0)
        bits = np.bitwise_or.reduce(g[:, 2 * R + 16:], axis=0).view(np.uint8)
        nb = self._nbits_bytes
        nzc = np.unpackbits(bits[:nb], bitorder="little")[:self.B]
        nzu = np.unpackbits(bits[nb:2 * nb], bitorder="little")[:self.B]
        off = np.concatenate([[0], np.cumsum(self.nbins)])[:-1]
        red = lambda x: np.add.reduceat(x.astype(np.uint64), off).astype(np.uint32)  # noqa: E731
        return self.host.set_coverage_columns(sums[:, 0], sums[:, 1], red(nzc), red(nzu), int(tail[0]), int(tail[1]))

    def finish_coverage(self):
        b = self.buf.numpy().view(np.uint32)
        cov, ucov, tail = b[:self.B], b[self.B:2 * self.B], b[2 * self.B:]
        off = np.concatenate([[0], np.cumsum(self.nbins)])[:-1]
        red = lambda x: np.add.reduceat(x.astype(np.uint64), off).astype(np.uint32)  # noqa: E731
        return self.host.set_coverage_columns(red(cov), red(ucov), red(cov != 0), red(ucov != 0), int(tail[0]), int(tail[1]))

    def filter_alignments(self):
        self.host.filter_alignments()
        valid = self.host.ref_columns()["valid"]
        self.b = self.orc.phase_b_with_valid(valid)

    def get_partials(self):
        u2, lca, marks, pairs = partials_from_oracle(self.b, self.w.lineage(), self.host.dense_taxid)
        return {"uniq_reads_count2": u2, "lca_count": lca, "level_marks": marks, "pairs": pairs}

    def set_partials(self, *a):
        self.host.set_partials(*a)

    def get_reads_lca_count(self):
        self.host.get_reads_lca_count()

    def write_abundance(self, path=None):
        return self.host.write_abundance(path)
