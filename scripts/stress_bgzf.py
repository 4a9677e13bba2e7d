"""Random small BAM files through slimm_push_bgzf_blocks (the device inflates, finds and decodes) with random window sizes,
some windows inflated on the host in between, grouped and in any order, against the oracle.
    python scripts/stress_bgzf.py [seeds] [first seed]        (GPU box, or SLIMM_EMU=1 for the host emulator)"""
import os, struct, sys, tempfile
sys.path.insert(0, ".")
import numpy as np
if os.environ.get("SLIMM_EMU") == "1":
    from slimm_amd import capi
    capi.LIB_PATH = os.path.join("tests", "native", "libslimm_emu.so"); capi._lib = None
from oracle.binding import run_workload
from slimm_amd.profiler import Slimm
from slimm_amd.workload import Records, Workload
from tests.bam_io import bam_record_bytes, write_bam
from tests.helpers import assert_matches_oracle
from tests.test_gpu_random import random_case

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
first = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
fails = 0
with tempfile.TemporaryDirectory() as d:
    for seed in range(first, first + n_seeds):
        rng = np.random.default_rng(seed)
        w = random_case(seed)
        r = w.records
        ids = np.unique(r.read_key, return_inverse=True)[1]
        names = ["q%d" % i + "n" * int(i % 23) for i in ids.tolist()]
        grouped = bool(seed % 2)
        if grouped:
            rec = Records(r.read_key, r.flag, r.ref_id, r.begin_pos, names)
        else:
            perm = rng.permutation(len(r))
            rec = Records(r.read_key[perm], r.flag[perm], r.ref_id[perm], r.begin_pos[perm], [names[i] for i in perm])
        wq = Workload(w.ref_names, w.ref_len, w.taxonomy, rec, w.avg_read_len, w.options, "bam", grouped=grouped)
        o = run_workload(wq, use_qnames=True)
        irr = int(rng.integers(1, 1 << 30)) if seed % 3 else None
        p = os.path.join(d, "f.bam")
        write_bam(p, w.ref_names, w.ref_len, rec, read_len=int(rng.integers(1, 300)), irregular_seed=irr)
        blob = open(p, "rb").read()
        # the blocks from the one holding the first record on, and the inflated bytes to skip there
        offs, at, total = [], 0, 0
        while at < len(blob):
            bsize = blob[at + 16] + (blob[at + 17] << 8) + 1
            offs.append((at, total))
            total += struct.unpack("<I", blob[at + bsize - 4:at + bsize])[0]
            at += bsize
        want_len = len(bam_record_bytes(rec, read_len=1, irregular_seed=None)) if False else None
        # (the header's inflated length: everything in front of the first record)
        import zlib
        raw, rest = bytearray(), blob
        while rest:
            dd = zlib.decompressobj(31); raw += dd.decompress(rest); rest = dd.unused_data
        l_text = struct.unpack("<i", raw[4:8])[0]; q = 8 + l_text; n_ref = struct.unpack("<i", raw[q:q + 4])[0]; q += 4
        for _ in range(n_ref):
            l_name = struct.unpack("<i", raw[q:q + 4])[0]; q += 4 + l_name + 4
        start = q
        k = max(i for i, (_, t) in enumerate(offs) if t <= start) if len(rec) else len(offs) - 1
        blocks, skip = blob[offs[k][0]:], (start - offs[k][1]) if len(rec) else (total - offs[k][1])
        window = int(rng.choice([0, 1, 700, 20_000, 300_000]))
        host_every = int(rng.choice([0, 0, 2, 3]))
        try:
            s = Slimm.for_workload(wq, device=0, grouped=grouped)
            got = s.push_bgzf_blocks(blocks, skip=skip, window=window, host_every=host_every)
            assert got == len(rec), (got, len(rec))
            prof = s.get_profiles()
            if o.no_hits: assert prof is None
            else: assert_matches_oracle(s, o)
            s.close()
        except Exception as e:
            fails += 1
            print("FAIL seed", seed, "grouped", grouped, "window", window, "host_every", host_every, "irregular", irr, str(e)[:300].replace("\n", " | "), flush=True)
print("fails", fails, "of", n_seeds)
