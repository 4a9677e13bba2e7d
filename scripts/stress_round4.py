"""Stress of round 4's two new device paths against the oracle, randomised: (a) record_order = ANY -- random small files
(tests/test_gpu_random.random_case) shuffled, under random plan knobs (hash bits 1 .. 32, digit width 1 .. 11, 1 .. 512
workgroups, staged or direct stores); (b) slimm_push_bam_bytes -- the same files as BAM record bytes with irregular record
sizes, pushed in windows of random sizes (a few hundred bytes to the whole file), grouped and in any order.
    python scripts/stress_round4.py [seeds] [first seed]         (GPU box, or SLIMM_EMU=1 for the host emulator)
Prints the failures and their count."""
import os, sys
sys.path.insert(0, ".")
import numpy as np
if os.environ.get("SLIMM_EMU") == "1":
    from slimm_amd import capi
    capi.LIB_PATH = os.path.join("tests", "native", "libslimm_emu.so"); capi._lib = None
from oracle.binding import run_workload
from slimm_amd.profiler import Slimm
from slimm_amd.workload import Records, Workload
from tests.bam_io import bam_record_bytes
from tests.helpers import assert_matches_oracle
from tests.test_gpu_random import random_case

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
fails = 0
for seed in range(first, first + n_seeds):
    rng = np.random.default_rng(seed)
    w = random_case(seed)
    r = w.records
    ids = np.unique(r.read_key, return_inverse=True)[1]
    names = ["q%d" % i + "n" * int(i % 23) for i in ids.tolist()]
    perm = rng.permutation(len(r))
    sh = Records(r.read_key[perm], r.flag[perm], r.ref_id[perm], r.begin_pos[perm], [names[i] for i in perm])
    gr = Records(r.read_key, r.flag, r.ref_id, r.begin_pos, names)
    # (a) the grouping under random knobs
    knobs = {"SLIMM_GROUP_BITS": int(rng.integers(1, 33)), "SLIMM_GROUP_WIDTH": int(rng.integers(1, 12)),
             "SLIMM_GROUP_GRID": int(rng.choice([1, 2, 3, 7, 64, 512])), "SLIMM_GROUP_STAGED": int(rng.integers(0, 2))}
    for k, v in knobs.items():
        os.environ[k] = str(v)
    wa = Workload(w.ref_names, w.ref_len, w.taxonomy, sh, w.avg_read_len, w.options, "any", grouped=False)
    oa = run_workload(wa, use_qnames=True)
    try:
        s = Slimm.for_workload(wa, device=0, grouped=False)
        s.push_records(sh, batch=int(rng.choice([0, 37, 1000])))
        prof = s.get_profiles()
        if oa.no_hits: assert prof is None
        else: assert_matches_oracle(s, oa)
        s.close()
    except Exception as e:
        fails += 1
        print("FAIL grouping seed", seed, knobs, str(e)[:300].replace("\n", " | "), flush=True)
    for k in knobs:
        del os.environ[k]
    # (b) the device BAM decoder: grouped (names compared) and any order (names hashed), random windows
    for recs, grouped, o in ((gr, True, None), (sh, False, oa)):
        wb = Workload(w.ref_names, w.ref_len, w.taxonomy, recs, w.avg_read_len, w.options, "bam", grouped=grouped)
        o = o or run_workload(wb, use_qnames=True)
        data = bam_record_bytes(recs, read_len=int(rng.integers(1, 400)), irregular_seed=int(rng.integers(1, 1 << 30)))
        window = int(rng.choice([0, 300, 5000, 16_384, 16_385, 70_001]))
        try:
            s = Slimm.for_workload(wb, device=0, grouped=grouped)
            assert s.push_bam_bytes(data, window=window) == len(recs)
            prof = s.get_profiles()
            if o.no_hits: assert prof is None
            else: assert_matches_oracle(s, o)
            s.close()
        except Exception as e:
            fails += 1
            print("FAIL bam seed", seed, "grouped", grouped, "window", window, str(e)[:300].replace("\n", " | "), flush=True)
print("fails", fails, "of", 3 * n_seeds)
