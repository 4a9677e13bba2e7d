"""Medium-size stress of the any-order grouping and the device BAM decoder: 0.3 - 3 M records of configs 2 / 3 / 5 (with their
full reference sets), the reads interleaved at random, random plan knobs / window sizes, against the oracle.
    python scripts/stress_medium.py [rounds] [first seed]"""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
from oracle.binding import run_workload
from slimm_amd.profiler import Slimm
from slimm_amd.synth import CONFIGS, make_workload
from slimm_amd.workload import Records, Workload
from tests.bam_io import bam_record_bytes
from tests.helpers import assert_matches_oracle

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
first = int(sys.argv[2]) if len(sys.argv) > 2 else 100
fails = 0
for seed in range(first, first + rounds):
    rng = np.random.default_rng(seed)
    cfg = ["config2", "config3", "config5"][seed % 3]
    n = int(rng.integers(300_000, 3_000_000))
    w = make_workload(CONFIGS[cfg], seed=seed, n_records=n)
    r = w.records
    o = run_workload(w, use_qnames=False, collect_bins=True)
    # interleave the reads at random, file order kept inside a read: every record gets a random place, then the places of a
    # read are handed to its records in file order
    ids = np.unique(r.read_key, return_inverse=True)[1]
    place = rng.permutation(len(r))
    by_place = np.lexsort((place, ids))                      # (read, place)
    by_file = np.lexsort((np.arange(len(r)), ids))           # (read, file order)
    newpos = np.empty(len(r), dtype=np.int64)
    newpos[by_file] = place[by_place]
    order = np.argsort(newpos)
    sh = Records(r.read_key[order], r.flag[order], r.ref_id[order], r.begin_pos[order])
    knobs = {"group_bits": int(rng.integers(14, 27)), "group_width": int(rng.integers(5, 12)),
             "group_grid": int(rng.choice([64, 509, 512])), "group_staged": int(rng.integers(0, 2))} if seed % 4 else {}
    os.environ["SLIMM_FORCE"] = ",".join(f"{k}={v}" for k, v in knobs.items())   # (slimm_amd/csrc/force.h)
    t0 = time.time()
    try:
        wa = Workload(w.ref_names, w.ref_len, w.taxonomy, sh, w.avg_read_len, w.options, "any", grouped=False)
        s = Slimm.for_workload(wa, device=0, grouped=False)
        s.push_records_packed(Records(sh.read_key & np.uint64((1 << 61) - 1), sh.flag, sh.ref_id, sh.begin_pos)) if seed % 3 == 0 else s.push_records(sh, batch=700_001)
        assert s.get_profiles() is not None
        if seed % 3 == 0:
            o61 = run_workload(Workload(w.ref_names, w.ref_len, w.taxonomy, Records(r.read_key & np.uint64((1 << 61) - 1), r.flag, r.ref_id, r.begin_pos), w.avg_read_len, w.options, "k61"), use_qnames=False, collect_bins=True)
            assert_matches_oracle(s, o61)
        else:
            assert_matches_oracle(s, o)
        s.close()
        print(f"seed {seed} {cfg} n={n} knobs={knobs} any-order ok ({time.time() - t0:.1f}s)", flush=True)
    except Exception as e:
        fails += 1
        print("FAIL any seed", seed, cfg, n, knobs, str(e)[:300].replace("\n", " | "), flush=True)
    os.environ.pop("SLIMM_FORCE", None)
    # the device BAM decoder on the grouped file (names compared on the device), windows of a random size
    try:
        names = ["r%x" % k for k in r.read_key.tolist()]
        gr = Records(r.read_key, r.flag, r.ref_id, r.begin_pos, names)
        wb = Workload(w.ref_names, w.ref_len, w.taxonomy, gr, w.avg_read_len, w.options, "bam", grouped=True)
        ob = run_workload(wb, use_qnames=True, collect_bins=False)
        data = bam_record_bytes(gr, read_len=int(rng.integers(30, 250)), irregular_seed=seed)
        s = Slimm.for_workload(wb, device=0, grouped=True)
        window = int(rng.choice([1 << 20, 3_000_001, 64 << 20]))
        assert s.push_bam_bytes(data, window=window) == len(gr)
        assert s.get_profiles() is not None
        assert_matches_oracle(s, ob, bins=False)
        s.close()
        print(f"seed {seed} bam window {window} ok", flush=True)
    except Exception as e:
        fails += 1
        print("FAIL bam seed", seed, cfg, n, str(e)[:300].replace("\n", " | "), flush=True)
print("fails", fails)
