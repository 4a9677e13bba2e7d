"""Throughput of the device inflate (slimm_bgzf_inflate) on a BAM file of this repository's writer: a name-grouped file and the
same records with the reads interleaved (names in random order compress worse).  python scripts/inflate_rate.py [records]"""
import ctypes as C, os, sys, tempfile, time, zlib
sys.path.insert(0, ".")
import numpy as np
from slimm_amd import capi
from slimm_amd.synth import CONFIGS, make_workload
from slimm_amd.workload import Records
from tests.bam_io import write_bam

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
repeat = int(sys.argv[2]) if len(sys.argv) > 2 else 1   # the file's blocks this many times over (blocks are independent)
w = make_workload(CONFIGS["config3"], seed=1, n_records=n)
r = w.records
names = ["r%x" % k for k in r.read_key.tolist()]
L = capi.lib()
L.slimm_bgzf_inflate.restype = C.c_int
with tempfile.TemporaryDirectory() as d:
    for what in ("grouped", "interleaved"):
        if what == "interleaved":
            rng = np.random.default_rng(3)
            order = rng.permutation(n)
            rec = Records(r.read_key[order], r.flag[order], r.ref_id[order], r.begin_pos[order], [names[i] for i in order])
        else:
            rec = Records(r.read_key, r.flag, r.ref_id, r.begin_pos, names)
        p = os.path.join(d, what + ".bam")
        write_bam(p, w.ref_names, w.ref_len, rec, read_len=w.avg_read_len)
        blob = np.fromfile(p, dtype=np.uint8)
        t0 = time.time()
        want = bytearray()
        dec, rest = zlib.decompressobj(31), blob.tobytes()
        while rest:
            want += dec.decompress(rest)
            rest = dec.unused_data
            dec = zlib.decompressobj(31)
        t_cpu = time.time() - t0
        eof = 28   # (the empty end-of-file block stays at the end of the last copy only)
        big = np.concatenate([blob[:-eof]] * (repeat - 1) + [blob]) if repeat > 1 else blob
        out = np.zeros(len(want) * repeat + 64, dtype=np.uint8)
        for it in range(3):
            nb, ms, err = C.c_uint64(), C.c_double(), C.create_string_buffer(256)
            rc = L.slimm_bgzf_inflate(0, big.ctypes.data_as(C.c_void_p), C.c_uint64(big.size), out.ctypes.data_as(C.c_void_p),
                                      C.c_uint64(out.size), C.byref(nb), C.byref(ms), err, C.c_uint64(256))
            assert rc == 0, err.value
        assert nb.value == len(want) * repeat
        assert bytes(out[:len(want)]) == bytes(want) and bytes(out[nb.value - len(want):nb.value]) == bytes(want)
        print(f"{what} x {repeat}: {big.size / 1e6:.1f} MB compressed -> {nb.value / 1e6:.1f} MB ({nb.value / big.size:.1f} x), kernel {ms.value:.2f} ms = "
              f"{nb.value / ms.value / 1e6:.1f} GB/s inflated; one zlib thread {t_cpu * 1e3:.0f} ms per copy", flush=True)
