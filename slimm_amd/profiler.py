"""Python mirror of the reference's `class slimm` (reference src/slimm.hpp:92-165) on top of the C ABI.

Method names and call order are the reference's: `analyze_alignments` -> `filter_alignments` ->
`get_reads_lca_count` -> `write_abundance` (src/slimm.hpp:449-489); `get_profiles` runs them in that order.
All compute happens in libslimm_hip.so (HIP kernels + the C++ host glue); this file only moves pointers.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import capi
from .workload import Options, Records, Taxonomy, Workload


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class DeviceArray:
    """A device pointer + length exposed through __cuda_array_interface__ (lets torch alias library memory)."""

    def __init__(self, ptr: int, n: int, typestr: str = "<i4"):
        self.ptr, self.n = ptr, n
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


class Slimm:
    def _config(self, taxonomy, options, ref_names, ref_len, avg_read_len, device, grouped, lineage):
        self.ref_names = list(ref_names)
        self.n_refs = len(ref_names)
        self._ref_len = np.ascontiguousarray(ref_len, dtype=np.uint32)
        self._lineage = np.ascontiguousarray(
            lineage if lineage is not None else taxonomy.lineage_for_header(ref_names), dtype=np.uint32)
        assert self._lineage.shape == (self.n_refs, 8)
        self._names = (C.c_char_p * len(taxonomy.tax_name))(*[n.encode() for n in taxonomy.tax_name])
        return capi.Config(
            n_refs=self.n_refs, ref_len=_p(self._ref_len), lineage=_p(self._lineage),
            bin_width=options.bin_width, avg_read_len=int(avg_read_len), min_reads=options.min_reads,
            cov_cut_off=options.cov_cut_off, abundance_cut_off=options.abundance_cut_off, rank=options.rank.encode(),
            n_taxa=len(taxonomy.tax_name), tax_id=_p(taxonomy.tax_id), tax_rank=_p(taxonomy.tax_rank),
            tax_name=self._names, device=device, record_order=capi.ORDER_GROUPED if grouped else capi.ORDER_ANY)

    def __init__(self, taxonomy: Taxonomy, options: Options, ref_names: List[str], ref_len: np.ndarray,
                 avg_read_len: int, device: int = 0, grouped: bool = True, lineage: Optional[np.ndarray] = None,
                 adopt=None):
        """adopt: (ctx pointer, owner) -- a context that belongs to somebody else (a SlimmGroup member): used, never
        destroyed here."""
        self.L = capi.lib()
        self.ctx = C.c_void_p()
        cfg = self._config(taxonomy, options, ref_names, ref_len, avg_read_len, device, grouped, lineage)
        self._owner = None
        if adopt is not None:
            self.ctx, self._owner = C.c_void_p(adopt[0]), adopt[1]
        else:
            rc = self.L.slimm_create(C.byref(cfg), C.byref(self.ctx))
            if rc != capi.OK:
                msg = self.L.slimm_last_error(None)
                raise capi.SlimmError(rc, msg.decode() if msg else "")
        self.device = device
        self._keepalive = None
        n = C.c_uint32()
        tp = C.c_void_p()
        self._check(self.L.slimm_dense_taxa(self.ctx, C.byref(n), C.byref(tp)))
        self.n_taxa_dense = n.value
        self.dense_taxid = np.ctypeslib.as_array(C.cast(tp, C.POINTER(C.c_uint32)), shape=(n.value,)).copy()

    @classmethod
    def for_workload(cls, w: Workload, device: int = 0, grouped: Optional[bool] = None) -> "Slimm":
        return cls(w.taxonomy, w.options, w.ref_names, w.ref_len, w.avg_read_len, device=device,
                   grouped=w.grouped if grouped is None else grouped)

    def close(self):
        if self.ctx and self._owner is None:
            self.L.slimm_destroy(self.ctx)
        self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int) -> int:
        return capi.check(self.ctx, rc)

    # ---- record stream ----
    def reset(self):
        self._keepalive = None
        self._check(self.L.slimm_reset(self.ctx))

    def reset_cutoffs(self):
        self._check(self.L.slimm_reset_cutoffs(self.ctx))

    def cutoff_cache(self) -> Tuple[float, float]:
        """The cached cut-offs (src/slimm.hpp:155-156; 0.0 = not computed yet): they survive reset() (Q8)."""
        a, b = C.c_float(0), C.c_float(0)
        self._check(self.L.slimm_get_cutoff_cache(self.ctx, C.byref(a), C.byref(b)))
        return float(a.value), float(b.value)

    def set_cutoff_cache(self, coverage_cut_off: float, uniq_coverage_cut_off: float):
        self._check(self.L.slimm_set_cutoff_cache(self.ctx, C.c_float(coverage_cut_off), C.c_float(uniq_coverage_cut_off)))

    def set_min_reads(self, min_reads: int):
        """options.min_reads as the file before left it (src/slimm.hpp:458-459 derives it into the options: Q8)."""
        self._check(self.L.slimm_set_min_reads(self.ctx, int(min_reads)))

    def push_records(self, rec: Records, batch: int = 0):
        n = len(rec)
        step = batch or max(n, 1)
        for s in range(0, n, step):
            e = min(n, s + step)
            self._check(self.L.slimm_push_records(self.ctx, _p(rec.read_key[s:e]), _p(rec.ref_id[s:e]),
                                                  _p(rec.begin_pos[s:e]), _p(rec.flag[s:e]), e - s))

    def push_records_checked(self, rec: Records, check: np.ndarray, batch: int = 0):
        """slimm_push_records_checked: `check` = a second hash of every record's read name (uint32)."""
        n = len(rec)
        check = np.ascontiguousarray(check, dtype=np.uint32)
        step = batch or max(n, 1)
        for s in range(0, n, step):
            e = min(n, s + step)
            self._check(self.L.slimm_push_records_checked(self.ctx, _p(rec.read_key[s:e]), _p(rec.ref_id[s:e]),
                                                          _p(rec.begin_pos[s:e]), _p(rec.flag[s:e]), _p(check[s:e]), e - s))

    def staging(self, which: int, capacity: int):
        """The context's page-locked staging set `which` (0 / 1) as numpy arrays (key u64, ref i32, pos i32, flag u16)."""
        import numpy as np

        ptr = [C.c_void_p() for _ in range(4)]
        self._check(self.L.slimm_staging_buffers(self.ctx, which, capacity, *[C.byref(q) for q in ptr]))
        out = []
        for q, (ct, dt) in zip(ptr, ((C.c_uint64, np.uint64), (C.c_int32, np.int32), (C.c_int32, np.int32),
                                     (C.c_uint16, np.uint16))):
            out.append(np.ctypeslib.as_array(C.cast(q, C.POINTER(ct)), shape=(capacity,)).view(dt))
        return tuple(out)

    def push_records_streamed(self, rec: Records, batch: int = 1 << 20):
        """slimm_push_staged_async over the two staging sets: batch k is copied into one set (the producer's work: a
        BAM decoder writes there directly) while the DMA engine reads the other; returns without waiting for PCIe."""
        n = len(rec)
        sets = [self.staging(0, batch), self.staging(1, batch)]
        for i, s in enumerate(range(0, n, batch)):
            e = min(n, s + batch)
            which = i & 1
            self._check(self.L.slimm_staging_wait(self.ctx, which))
            k, r, p, f = sets[which]
            k[: e - s] = rec.read_key[s:e]
            r[: e - s] = rec.ref_id[s:e]
            p[: e - s] = rec.begin_pos[s:e]
            f[: e - s] = rec.flag[s:e]
            self._check(self.L.slimm_push_staged_async(self.ctx, which, e - s))

    def push_records_async(self, key, ref, pos, flag):
        """slimm_push_records_async on numpy arrays (page-locked ones are read by the DMA engine directly); keep them
        unchanged until push_wait()."""
        self._keepalive = (key, ref, pos, flag)
        self._check(self.L.slimm_push_records_async(self.ctx, _p(key), _p(ref), _p(pos), _p(flag), len(key)))

    # ---- packed records: 16 bytes each, the flag bits in the key's top three bits (slimm_pack_key) ----
    @staticmethod
    def pack_keys(read_key: np.ndarray, flag: np.ndarray) -> np.ndarray:
        """slimm_pack_keys: (read_key & (2^61 - 1)) | mate << 61 | unmapped << 63."""
        read_key = np.ascontiguousarray(read_key, dtype=np.uint64)
        flag = np.ascontiguousarray(flag, dtype=np.uint16)
        out = np.empty(read_key.shape[0], dtype=np.uint64)
        capi.lib().slimm_pack_keys(_p(read_key), _p(flag), read_key.shape[0], _p(out))
        return out

    def push_records_packed(self, rec: Records, batch: int = 0, packed_key: Optional[np.ndarray] = None):
        pk = self.pack_keys(rec.read_key, rec.flag) if packed_key is None else packed_key
        n = len(rec)
        step = batch or max(n, 1)
        for s in range(0, n, step):
            e = min(n, s + step)
            self._check(self.L.slimm_push_records_packed(self.ctx, _p(pk[s:e]), _p(rec.ref_id[s:e]), _p(rec.begin_pos[s:e]),
                                                         e - s))

    def push_records_packed_async(self, packed_key, ref, pos):
        self._keepalive = (packed_key, ref, pos)
        self._check(self.L.slimm_push_records_packed_async(self.ctx, _p(packed_key), _p(ref), _p(pos), len(packed_key)))

    def push_records_packed_streamed(self, rec: Records, batch: int = 1 << 20):
        """slimm_push_staged_packed_async over the two staging sets (their flag arrays stay unused)."""
        n = len(rec)
        sets = [self.staging(0, batch), self.staging(1, batch)]
        for i, s in enumerate(range(0, n, batch)):
            e = min(n, s + batch)
            which = i & 1
            self._check(self.L.slimm_staging_wait(self.ctx, which))
            k, r, p, _ = sets[which]
            k[: e - s] = self.pack_keys(rec.read_key[s:e], rec.flag[s:e])
            r[: e - s] = rec.ref_id[s:e]
            p[: e - s] = rec.begin_pos[s:e]
            self._check(self.L.slimm_push_staged_packed_async(self.ctx, which, e - s))

    def set_records_device_packed(self, packed_key, ref, pos):
        """torch tensors on this context's device: int64/uint64 packed key, int32 ref, int32 pos."""
        n = int(packed_key.shape[0])
        self._keepalive = (packed_key, ref, pos)
        self._check(self.L.slimm_set_records_device_packed(self.ctx, C.c_void_p(packed_key.data_ptr()),
                                                           C.c_void_p(ref.data_ptr()), C.c_void_p(pos.data_ptr()), n))

    # ---- run-marked records: 8 bytes each, grouped input only (include/slimm_hip.h)
    @staticmethod
    def mark_words(read_key: np.ndarray, flag: np.ndarray, ref_id: np.ndarray, prev_key: Optional[int] = None) -> np.ndarray:
        """slimm_mark_words: reference + 1 | mate << 29 | starts-a-qName-run << 31 per record."""
        read_key = np.ascontiguousarray(read_key, dtype=np.uint64)
        flag = np.ascontiguousarray(flag, dtype=np.uint16)
        ref_id = np.ascontiguousarray(ref_id, dtype=np.int32)
        out = np.empty(read_key.shape[0], dtype=np.uint32)
        pk = C.c_uint64(prev_key) if prev_key is not None else None
        capi.lib().slimm_mark_words(_p(read_key), _p(flag), _p(ref_id), read_key.shape[0], C.byref(pk) if pk is not None else None,
                                    _p(out))
        return out

    def push_records_marked(self, rec: Records, batch: int = 0, words: Optional[np.ndarray] = None):
        w = self.mark_words(rec.read_key, rec.flag, rec.ref_id) if words is None else words
        n = len(rec)
        step = batch or max(n, 1)
        for s in range(0, n, step):
            e = min(n, s + step)
            self._check(self.L.slimm_push_records_marked(self.ctx, _p(w[s:e]), _p(rec.begin_pos[s:e]), e - s))

    def push_records_marked_async(self, words, pos):
        self._keepalive = (words, pos)
        self._check(self.L.slimm_push_records_marked_async(self.ctx, _p(words), _p(pos), len(words)))

    def push_records_marked_streamed(self, rec: Records, batch: int = 1 << 20):
        """slimm_push_staged_marked_async over the two staging sets (the words go into the sets' ref_id arrays)."""
        n = len(rec)
        w = self.mark_words(rec.read_key, rec.flag, rec.ref_id)
        sets = [self.staging(0, batch), self.staging(1, batch)]
        for i, s in enumerate(range(0, n, batch)):
            e = min(n, s + batch)
            which = i & 1
            self._check(self.L.slimm_staging_wait(self.ctx, which))
            _, r, p, _ = sets[which]
            r[: e - s] = w[s:e].view(np.int32)
            p[: e - s] = rec.begin_pos[s:e]
            self._check(self.L.slimm_push_staged_marked_async(self.ctx, which, e - s))

    def set_records_device_marked(self, words, pos):
        """torch tensors on this context's device: int32/uint32 words, int32 pos."""
        n = int(words.shape[0])
        self._keepalive = (words, pos)
        self._check(self.L.slimm_set_records_device_marked(self.ctx, C.c_void_p(words.data_ptr()), C.c_void_p(pos.data_ptr()), n))

    def push_bam_bytes(self, data, window: int = 0) -> int:
        """slimm_push_bam_bytes: the alignment-record bytes of a BAM file (behind its header, BGZF-inflated), in windows of
        `window` bytes (0: one); the device finds the records and decodes them.  Returns the number of records."""
        buf = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
        n = buf.shape[0]
        step = window or max(n, 1)
        total = 0
        got = C.c_uint64()
        if n == 0:
            self._check(self.L.slimm_push_bam_bytes(self.ctx, None, 0, 1, C.byref(got)))
        for s in range(0, n, step):
            e = min(n, s + step)
            piece = np.ascontiguousarray(buf[s:e])
            self._check(self.L.slimm_push_bam_bytes(self.ctx, _p(piece), e - s, 1 if e == n else 0, C.byref(got)))
            total += got.value
        return total

    def set_reference_names(self, names):
        """The header's reference names for push_sam_bytes (slimm_set_reference_names)."""
        arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
        self._check(self.L.slimm_set_reference_names(self.ctx, arr))

    def push_sam_bytes(self, text, window: int = 0) -> int:
        """slimm_push_sam_bytes: the alignment lines of a SAM file (bytes behind the header) in windows of `window` bytes (0:
        one), cut anywhere.  Returns the number of records."""
        buf = np.frombuffer(text, dtype=np.uint8) if not isinstance(text, np.ndarray) else np.ascontiguousarray(text, dtype=np.uint8)
        n = buf.shape[0]
        step = window or max(n, 1)
        total, got, keep = 0, C.c_uint64(), []
        if n == 0:
            self._check(self.L.slimm_push_sam_bytes(self.ctx, None, 0, 1, C.byref(got)))
        for s in range(0, n, step):
            e = min(n, s + step)
            piece = np.ascontiguousarray(buf[s:e])
            self._check(self.L.slimm_push_sam_bytes(self.ctx, _p(piece), e - s, 1 if e == n else 0, C.byref(got)))
            keep = (keep + [piece])[-3:]
            total += got.value
        return total

    def push_bgzf_blocks(self, blob, skip: int = 0, window: int = 0, host_every: int = 0) -> int:
        """slimm_push_bgzf_blocks: whole BGZF blocks of a BAM file (compressed), the first of which holds the first alignment
        record `skip` inflated bytes in; windows of about `window` compressed bytes (0: one), cut at block boundaries.  The
        device inflates the blocks, finds the records and decodes them.  host_every = k > 0: every k-th window is inflated
        HERE (zlib) and handed over through slimm_push_bam_bytes instead -- the two forms may alternate within a file.
        Returns the number of records."""
        import zlib
        buf = np.frombuffer(blob, dtype=np.uint8) if not isinstance(blob, np.ndarray) else np.ascontiguousarray(blob, dtype=np.uint8)
        n = buf.shape[0]
        cuts, p = [0], 0
        while p < n:   # block boundaries: BSIZE - 1 is the BC subfield's value (the first extra subfield of the files handled here)
            assert buf[p] == 0x1f and buf[p + 1] == 0x8b and buf[p + 12] == ord("B") and buf[p + 13] == ord("C"), "not a BGZF block"
            p += int(buf[p + 16]) + (int(buf[p + 17]) << 8) + 1
            if window == 0 or p - cuts[-1] >= window or p >= n:
                cuts.append(p) if (window or p >= n) else None
        if cuts[-1] != n:
            cuts.append(n)
        total, got, keep = 0, C.c_uint64(), []
        if n == 0:
            self._check(self.L.slimm_push_bgzf_blocks(self.ctx, None, 0, 0, 1, C.byref(got)))
        for k in range(len(cuts) - 1):
            piece = np.ascontiguousarray(buf[cuts[k]:cuts[k + 1]])
            last = 1 if k == len(cuts) - 2 else 0
            if host_every and k % host_every == host_every - 1:
                raw, rest = bytearray(), piece.tobytes()
                while rest:
                    d = zlib.decompressobj(31)
                    raw += d.decompress(rest)
                    rest = d.unused_data
                piece = np.frombuffer(bytes(raw[(skip if k == 0 else 0):]), dtype=np.uint8)
                self._check(self.L.slimm_push_bam_bytes(self.ctx, _p(piece) if piece.size else None, piece.size, last, C.byref(got)))
            else:
                self._check(self.L.slimm_push_bgzf_blocks(self.ctx, _p(piece), piece.size, skip if k == 0 else 0, last, C.byref(got)))
            keep = (keep + [piece])[-3:]   # (a window's buffer stays until the next call has returned)
            total += got.value
        return total

    def push_wait(self):
        self._check(self.L.slimm_push_wait(self.ctx))

    def set_records_device(self, key, ref, pos, flag):
        """torch tensors on this context's device: int64/uint64 key, int32 ref, int32 pos, int16/uint16 flag."""
        n = int(key.shape[0])
        self._keepalive = (key, ref, pos, flag)
        self._check(self.L.slimm_set_records_device(self.ctx, C.c_void_p(key.data_ptr()), C.c_void_p(ref.data_ptr()),
                                                    C.c_void_p(pos.data_ptr()), C.c_void_p(flag.data_ptr()), n))

    def q18_runs(self) -> Tuple[int, int]:
        """slimm_get_q18_runs: (runs that start with a shortened name, shortened -> plain steps inside a run) of the file the
        device decoders have read; unequal = some run holds shortened names only (include/slimm_hip.h, Q18 ON A GROUPED STREAM)."""
        a, b = C.c_uint64(0), C.c_uint64(0)
        self._check(self.L.slimm_get_q18_runs(self.ctx, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def grouped_records(self):
        """slimm_grouped_records (record_order = ANY, after analyze_alignments): (ident, ref, gbin) of the mapped records
        as the device grouped them for the front end."""
        n = C.c_uint64()
        self._check(self.L.slimm_grouped_records(self.ctx, None, None, None, 0, C.byref(n)))
        ident = np.empty(n.value, dtype=np.uint64)
        ref = np.empty(n.value, dtype=np.uint32)
        gbin = np.empty(n.value, dtype=np.uint32)
        self._check(self.L.slimm_grouped_records(self.ctx, _p(ident), _p(ref), _p(gbin), n.value, C.byref(n)))
        return ident, ref, gbin

    def check_grouping(self) -> int:
        """slimm_check_grouping: qName runs whose identity started an earlier run too (0 = the stream is grouped)."""
        n = C.c_uint64()
        self._check(self.L.slimm_check_grouping(self.ctx, C.byref(n)))
        return int(n.value)

    # ---- the reference's phases ----
    def analyze_alignments(self):
        self._check(self.L.slimm_analyze_alignments(self.ctx))

    def coverage_buffer(self) -> DeviceArray:
        ptr = C.c_void_p()
        n = C.c_uint64()
        self._check(self.L.slimm_coverage_buffer(self.ctx, C.byref(ptr), C.byref(n)))
        return DeviceArray(ptr.value, n.value, "<i4")

    def _alias(self, name: str, ptr: int, n: int):
        """int32 torch tensor over n words of library memory at ptr; the wrapper is kept per buffer (building one from
        __cuda_array_interface__ costs tens of microseconds, which the per-file exchange would pay three times)."""
        import torch

        cache = self.__dict__.setdefault("_alias_cache", {})
        hit = cache.get(name)
        if hit is None or hit[0] != (ptr, n):
            hit = ((ptr, n), torch.as_tensor(DeviceArray(ptr, n, "<i4"), device=f"cuda:{self.device}"))
            cache[name] = hit
        return hit[1]

    def scratch(self, name: str, numel: int, like):
        """A reusable device tensor of the exchange (receive buffers), allocated once per size."""
        import torch

        cache = self.__dict__.setdefault("_scratch_cache", {})
        t = cache.get(name)
        if t is None or t.numel() != numel or t.dtype != like.dtype or t.device != like.device:
            t = torch.empty(numel, dtype=like.dtype, device=like.device)
            cache[name] = t
        return t

    def _after_torch(self):
        """Before a call that reads device memory the caller has written with torch (a gathered summary, a buffer reduced in
        place): torch's work on ITS current stream must be done -- the library's kernels run on the context's own stream,
        which waits for no other.  Nothing to do when the caller works under `torch.cuda.stream(engine.torch_stream())`."""
        import sys

        torch = sys.modules.get("torch")
        if self.device is None or self.device < 0 or torch is None or not torch.cuda.is_initialized():
            return  # (no torch in this process, or none that has touched the device: nothing of its is under way)
        cur = torch.cuda.current_stream(self.device)
        ext = getattr(self, "_ext_stream", None)
        if ext is None or cur.cuda_stream != ext.cuda_stream:
            cur.synchronize()

    def torch_stream(self):
        """The HIP stream the context's kernels run on, as a torch.cuda.ExternalStream.  Asking for it switches the
        context to stream-ordered buffers (slimm_set_stream_ordered): collectives issued under
        `torch.cuda.stream(engine.torch_stream())` are ordered with the library's kernels on the device and no
        torch.cuda.synchronize is needed between them (slimm_amd/distributed.py does exactly that)."""
        import torch

        if getattr(self, "_ext_stream", None) is None:
            p = C.c_void_p()
            self._check(self.L.slimm_get_stream(self.ctx, C.byref(p)))
            self._ext_stream = torch.cuda.ExternalStream(p.value, device=f"cuda:{self.device}")
            self._check(self.L.slimm_set_stream_ordered(self.ctx, 1))
        return self._ext_stream

    def coverage_tensor(self):
        """The coverage buffer as an int32 torch tensor aliasing the library's device memory (for the all-reduce)."""
        import torch

        b = self.coverage_buffer()
        return self._alias("coverage", b.ptr, b.n)

    def keep_bins(self, on: bool = True):
        """Whether the coverage arrays are materialised in HBM (needed by bins() and coverage_tensor(); default yes)."""
        self._check(self.L.slimm_keep_bins(self.ctx, 1 if on else 0))

    def prepare_summary(self, n_slices: int = 1):
        """Multi-GPU: have phase A write the coverage-summary bitmaps as a by-product (call before analyze_alignments).
        0 = off, 1 = all-gather layout, n > 1 = n slices for the all-to-all exchange."""
        self._check(self.L.slimm_prepare_summary(self.ctx, int(n_slices)))

    def summary_head_words(self) -> int:
        """Words of a coverage summary before its bitmap chunks."""
        return 4 * self.n_refs + 16

    def merge_summary_slices(self, received, n_ranks: int, rank: int):
        """`received`: int32 device tensor with every rank's bitmap chunk for this rank's slice (all_to_all output).
        Returns the additive [4R | 16] vector (aliasing library memory) to all_reduce(SUM) in place."""
        import torch

        self._slices_keepalive = received
        self._after_torch()
        ptr = C.c_void_p()
        n = C.c_uint64()
        self._check(self.L.slimm_merge_summary_slices(self.ctx, C.c_void_p(received.data_ptr()), int(n_ranks), int(rank),
                                                      C.byref(ptr), C.byref(n)))
        return self._alias("sum_vec", ptr.value, n.value)

    def finish_coverage_reduced(self) -> bool:
        self._after_torch()
        return self._check(self.L.slimm_finish_coverage_reduced(self.ctx)) != capi.E_NO_HITS

    def coverage_summary_tensor(self):
        """[per-ref sums | scalars | 'bin != 0' bitmaps] of this rank as an int32 tensor aliasing library memory."""
        import torch

        ptr = C.c_void_p()
        n = C.c_uint64()
        self._check(self.L.slimm_coverage_summary(self.ctx, C.byref(ptr), C.byref(n)))
        return self._alias("summary", ptr.value, n.value)

    def finish_coverage_merged(self, gathered, n_ranks: int) -> bool:
        """`gathered`: int32 device tensor holding the summaries of all ranks back to back (all_gather output)."""
        self._merged_keepalive = gathered
        self._after_torch()
        rc = self._check(self.L.slimm_finish_coverage_merged(self.ctx, C.c_void_p(gathered.data_ptr()), int(n_ranks)))
        return rc != capi.E_NO_HITS

    def finish_coverage(self) -> bool:
        """True when there are mapped records (False = the reference's 'No mapped reads' early return)."""
        self._after_torch()
        return self._check(self.L.slimm_finish_coverage(self.ctx)) != capi.E_NO_HITS

    def set_coverage_columns(self, reads_count, uniq_reads_count, nz_cov, nz_uniq_cov, hits, matches) -> bool:
        a = [np.ascontiguousarray(x, dtype=np.uint32) for x in (reads_count, uniq_reads_count, nz_cov, nz_uniq_cov)]
        return self._check(self.L.slimm_set_coverage_columns(self.ctx, _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]),
                                                             int(hits), int(matches))) != capi.E_NO_HITS

    def filter_alignments(self):
        self._check(self.L.slimm_filter_alignments(self.ctx))

    def filter_alignments_launch(self):
        """Multi-rank form: launches phase B and packs this rank's partial results, without waiting for the device;
        partials_tensor() -> all_reduce -> install_merged_partials() completes it (None from there = go round again)."""
        self._check(self.L.slimm_filter_alignments_launch(self.ctx))

    def get_partials(self) -> Dict[str, np.ndarray]:
        p = capi.Partials()
        self._check(self.L.slimm_get_partials(self.ctx, C.byref(p)))

        def arr(ptr, n, ct, dt):
            if n == 0:
                return np.zeros(0, dtype=dt)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(n,)).copy()

        return {"uniq_reads_count2": arr(p.uniq_reads_count2, p.n_refs, C.c_uint32, np.uint32),
                "lca_count": arr(p.lca_count, p.n_taxa_dense, C.c_uint32, np.uint32),
                "level_marks": arr(p.level_marks, p.n_refs, C.c_uint32, np.uint32),
                "pairs": arr(p.pairs, p.n_pairs, C.c_uint64, np.uint64)}

    def partials_tensor(self):
        """This rank's additive partial results ([uniq_reads_count2 | LCA counts | level marks | pair count]) as an int32
        tensor aliasing library memory: all_reduce(SUM) it in place, then call install_merged_partials()."""
        import torch

        ptr = C.c_void_p()
        n = C.c_uint64()
        self._check(self.L.slimm_partials_buffer(self.ctx, C.byref(ptr), C.byref(n)))
        return self._alias("partials", ptr.value, n.value)

    def install_merged_partials(self) -> int:
        """Installs the (summed) partials buffer; returns the number of (taxon, reference) pairs over all ranks -- or None
        after filter_alignments_launch when some rank's pair set overflowed (every rank then launches again)."""
        total = C.c_uint32()
        self._after_torch()
        if self._check(self.L.slimm_install_merged_partials(self.ctx, C.byref(total))) == capi.E_RETRY:
            return None
        return int(total.value)

    def set_partials(self, uniq_reads_count2, lca_count, level_marks, pairs):
        u2 = np.ascontiguousarray(uniq_reads_count2, dtype=np.uint32)
        lc = np.ascontiguousarray(lca_count, dtype=np.uint32)
        mk = np.ascontiguousarray(level_marks, dtype=np.uint32)
        pr = np.ascontiguousarray(pairs, dtype=np.uint64)
        p = capi.Partials(n_refs=self.n_refs, n_taxa_dense=self.n_taxa_dense, uniq_reads_count2=_p(u2), lca_count=_p(lc),
                          level_marks=_p(mk), pairs=_p(pr), n_pairs=int(pr.shape[0]))
        self._check(self.L.slimm_set_partials(self.ctx, C.byref(p)))

    def get_reads_lca_count(self):
        self._check(self.L.slimm_get_reads_lca_count(self.ctx))

    def write_abundance(self, path: Optional[str] = None) -> str:
        if path is not None:
            self._check(self.L.slimm_write_abundance_file(self.ctx, path.encode()))
        text = C.c_char_p()
        n = C.c_uint64()
        self._check(self.L.slimm_write_abundance(self.ctx, C.byref(text), C.byref(n)))
        return C.string_at(text, n.value).decode()

    def get_profiles(self, rec: Optional[Records] = None, path: Optional[str] = None) -> Optional[str]:
        """slimm::get_profiles() minus file I/O (src/slimm.hpp:395-496). None = no mapped reads."""
        if rec is not None:
            self.push_records(rec)
        if self._check(self.L.slimm_get_profiles(self.ctx, path.encode() if path else None)) == capi.E_NO_HITS:
            return None
        return self.write_abundance(None)

    # ---- results ----
    def stats(self) -> Dict[str, float]:
        s = capi.Stats()
        self._check(self.L.slimm_get_stats(self.ctx, C.byref(s)))
        return {n: getattr(s, n) for n, _ in capi.Stats._fields_}

    def ref_columns(self) -> Dict[str, np.ndarray]:
        R = self.n_refs
        out = {n: np.zeros(R, dtype=np.uint32) for n in ("reads_count", "uniq_reads_count", "uniq_reads_count2", "nbins",
                                                         "nz_cov", "nz_uniq_cov", "nz_uniq_cov2")}
        out["valid"] = np.zeros(R, dtype=np.uint8)
        out["abundance"] = np.zeros(R, dtype=np.float32)
        out["uniq_abundance"] = np.zeros(R, dtype=np.float32)
        cols = capi.RefColumns(**{k: _p(v) for k, v in out.items()})
        self._check(self.L.slimm_get_ref_columns(self.ctx, C.byref(cols)))
        return out

    def bins(self, which: int) -> np.ndarray:
        out = np.zeros(int(self.stats()["total_bins"]), dtype=np.uint32)
        self._check(self.L.slimm_get_bins(self.ctx, which, _p(out)))
        return out

    def read_targets(self):
        """The reads' target lists after analyze_alignments: (ref | head bit, global bin | unique bit), both uint32."""
        n = C.c_uint64()
        self._check(self.L.slimm_get_read_targets(self.ctx, None, None, 0, C.byref(n)))
        ref = np.zeros(n.value, dtype=np.uint32)
        gbin = np.zeros(n.value, dtype=np.uint32)
        self._check(self.L.slimm_get_read_targets(self.ctx, _p(ref), _p(gbin), n.value, C.byref(n)))
        return ref, gbin

    def taxon_counts(self, stage: int = 1) -> Dict[int, int]:
        n = C.c_uint32()
        self._check(self.L.slimm_taxon_count_size(self.ctx, stage, C.byref(n)))
        t = np.zeros(n.value, dtype=np.uint32)
        c = np.zeros(n.value, dtype=np.uint32)
        self._check(self.L.slimm_get_taxon_counts(self.ctx, stage, _p(t), _p(c)))
        return {int(a): int(b) for a, b in zip(t, c)}

    def children_pairs(self, stage: int = 1) -> set:
        n = C.c_uint64()
        self._check(self.L.slimm_children_pairs_size(self.ctx, stage, C.byref(n)))
        t = np.zeros(n.value, dtype=np.uint32)
        r = np.zeros(n.value, dtype=np.uint32)
        self._check(self.L.slimm_get_children_pairs(self.ctx, stage, _p(t), _p(r)))
        return set(zip(t.tolist(), r.tolist()))

    # ---- measurement ----
    def enable_kernel_timing(self, on: bool = True):
        self._check(self.L.slimm_enable_kernel_timing(self.ctx, int(on)))

    def time_only_kernel(self, name: Optional[str]):
        self._check(self.L.slimm_time_only_kernel(self.ctx, name.encode() if name else None))

    def kernel_times(self, reset: bool = True) -> Dict[str, Tuple[float, int]]:
        cap = 32
        names = (C.c_char_p * cap)()
        ms = (C.c_double * cap)()
        ln = (C.c_uint32 * cap)()
        n = C.c_uint32()
        self._check(self.L.slimm_kernel_times(self.ctx, names, ms, ln, cap, C.byref(n), int(reset)))
        return {names[i].decode(): (ms[i], ln[i]) for i in range(n.value)}


def host_quantile_cut_off(v: np.ndarray, q: float) -> float:
    v = np.ascontiguousarray(v, dtype=np.float32)
    return float(capi.lib().slimm_host_quantile_cut_off(_p(v), v.shape[0], C.c_float(q)))


def host_bin_of(begin_pos: int, avg_read_len: int, ref_len: int, bin_width: int) -> int:
    return int(capi.lib().slimm_host_bin_of(int(begin_pos), int(avg_read_len), int(ref_len), int(bin_width)))


def host_canonical_read_name(name: str, flag: int) -> Tuple[str, int]:
    """(base name, flag with the base's mate bit): the canonical identity of a record's read (src/slimm.hpp:204-208, Q18)."""
    b = name.encode()
    out = C.c_uint16(0)
    n = capi.lib().slimm_host_canonical_read_name(b, len(b), int(flag), C.byref(out))
    return b[:n].decode(), int(out.value)


def host_avg_read_length(l_seq: np.ndarray, sample: int = 100000) -> int:
    l_seq = np.ascontiguousarray(l_seq, dtype=np.uint32)
    return int(capi.lib().slimm_host_avg_read_length(_p(l_seq), l_seq.shape[0], sample))


class SlimmGroup:
    """slimm_group_*: several GPUs in one process, used like one context (include/slimm_hip.h).  `devices` may name one
    device several times (that is how the tests run a group on a single GPU: the collectives are then copies)."""

    def __init__(self, w: Workload, devices, grouped: Optional[bool] = None):
        self.L = capi.lib()
        self.g = C.c_void_p()
        self.w = w
        self.grouped = w.grouped if grouped is None else grouped
        self.devices = list(devices)
        # (the configuration arrays live in a Slimm-shaped holder; member 0 is then adopted into the same object)
        self._holder = Slimm.__new__(Slimm)
        cfg = Slimm._config(self._holder, w.taxonomy, w.options, w.ref_names, w.ref_len, w.avg_read_len, 0, self.grouped, None)
        dev = (C.c_int * len(self.devices))(*self.devices)
        rc = self.L.slimm_group_create(C.byref(cfg), dev, len(self.devices), C.byref(self.g))
        if rc != capi.OK:
            msg = self.L.slimm_group_last_error(None)
            raise capi.SlimmError(rc, msg.decode() if msg else "")

    def _check(self, rc: int) -> int:
        if rc < 0:
            msg = self.L.slimm_group_last_error(self.g)
            raise capi.SlimmError(rc, msg.decode() if msg else "")
        return rc

    def member(self, i: int = 0) -> Slimm:
        """Member i as a Slimm object (results: stats(), ref_columns(), taxon counts ... of the merged run on member 0)."""
        w = self.w
        return Slimm(w.taxonomy, w.options, w.ref_names, w.ref_len, w.avg_read_len, device=self.devices[i],
                     grouped=self.grouped, adopt=(self.L.slimm_group_context(self.g, i), self))

    @property
    def uses_rccl(self) -> bool:
        return bool(self.L.slimm_group_uses_rccl(self.g))

    def reset(self):
        self._check(self.L.slimm_group_reset(self.g))

    def push_records(self, rec: Records, batch: int = 0):
        n = len(rec)
        step = batch or max(n, 1)
        for s in range(0, n, step):
            e = min(n, s + step)
            self._check(self.L.slimm_group_push_records(self.g, _p(rec.read_key[s:e]), _p(rec.ref_id[s:e]),
                                                        _p(rec.begin_pos[s:e]), _p(rec.flag[s:e]), e - s))

    def push_records_checked(self, rec: Records, check: np.ndarray, batch: int = 0):
        n = len(rec)
        check = np.ascontiguousarray(check, dtype=np.uint32)
        step = batch or max(n, 1)
        for s in range(0, n, step):
            e = min(n, s + step)
            self._check(self.L.slimm_group_push_records_checked(self.g, _p(rec.read_key[s:e]), _p(rec.ref_id[s:e]),
                                                                _p(rec.begin_pos[s:e]), _p(rec.flag[s:e]), _p(check[s:e]), e - s))

    def push_records_packed(self, rec: Records, batch: int = 0):
        pk = Slimm.pack_keys(rec.read_key, rec.flag)
        n = len(rec)
        step = batch or max(n, 1)
        for s in range(0, n, step):
            e = min(n, s + step)
            self._check(self.L.slimm_group_push_records_packed(self.g, _p(pk[s:e]), _p(rec.ref_id[s:e]), _p(rec.begin_pos[s:e]),
                                                               e - s))

    def push_records_marked(self, rec: Records, batch: int = 0):
        w = Slimm.mark_words(rec.read_key, rec.flag, rec.ref_id)
        n = len(rec)
        step = batch or max(n, 1)
        for s in range(0, n, step):
            e = min(n, s + step)
            self._check(self.L.slimm_group_push_records_marked(self.g, _p(w[s:e]), _p(rec.begin_pos[s:e]), e - s))

    EXCHANGES = {"auto": 0, "summary": 1, "sliced": 2, "bins": 3}

    def set_exchange(self, mode: str):
        """slimm_group_set_exchange: "auto" | "summary" (all-gather) | "sliced" (all-to-all + small all-reduce) | "bins"
        (all-reduce of the coverage arrays themselves; member 0's bins() are then the global arrays)."""
        self._check(self.L.slimm_group_set_exchange(self.g, self.EXCHANGES[mode]))

    @property
    def exchange(self) -> str:
        k = self.L.slimm_group_exchange(self.g)
        return next(n for n, v in self.EXCHANGES.items() if v == k)

    def get_profiles(self, path: Optional[str] = None) -> bool:
        """False when no record is mapped (the reference's early return)."""
        return self._check(self.L.slimm_group_get_profiles(self.g, path.encode() if path else None)) != capi.E_NO_HITS

    def close(self):
        if self.g:
            self.L.slimm_group_destroy(self.g)
            self.g = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
