"""Stress of round 5's bucket scatter (k_tile_scatter_big: rounds ordered by tile in LDS, one table word per tile) against the
all-core dense restatement, randomised: files of 3 - 40 M records over random layouts -- reference counts, lengths and bin
widths chosen so that the tile count falls on either side of the 4064 / 6144 limits --, random hits per read, grouped or in any
order, under random bucketing switches (SLIMM_FORCE: fused_scan, scatter_big, matrix, tile_shift, wide_tiles).
Files of this size give a workgroup several rounds, partial last rounds and slots of every fill.
    python scripts/stress_round5.py [seeds] [first seed]         (GPU box)
Prints the failures and their count."""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
from oracle.binding import dense_mt_run
from slimm_amd.profiler import Slimm
from slimm_amd.synth import SynthConfig, make_workload
from tests.test_gpu_parity import assert_equals_dense_mt

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
first = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
fails = 0
for seed in range(first, first + n_seeds):
    rng = np.random.default_rng(seed)
    n = int(rng.choice([3_000_000, 8_000_000, 20_000_000, 40_000_000]))
    refs = int(rng.choice([300, 3_000, 12_000, 30_000]))
    width = int(rng.choice([200, 1000, 3000]))
    lo = int(rng.choice([200_000, 1_000_000, 3_000_000])); hi = lo * int(rng.integers(2, 4))
    cfg = SynthConfig(f"s{seed}", n, refs, float(rng.choice([1.5, 4.0, 9.0, 25.0])), bin_width=width, len_lo=lo, len_hi=hi,
                      present_frac=float(rng.choice([0.02, 0.2, 0.8])), strain_level=bool(rng.integers(0, 2)))
    knobs = {}
    if rng.integers(0, 2): knobs["fused_scan"] = "0"
    if rng.integers(0, 3) == 0: knobs["scatter_big"] = "0"
    if rng.integers(0, 2): knobs["matrix"] = str(rng.choice(["0", "2"]))
    if rng.integers(0, 2): knobs["tile_shift"] = str(rng.choice(["13", "14"]))
    if rng.integers(0, 4) == 0: knobs["wide_tiles"] = "1"
    grouped = bool(rng.integers(0, 4))
    t0 = time.time()
    w = make_workload(cfg, seed=seed)
    os.environ["SLIMM_FORCE"] = ",".join(f"{k}={v}" for k, v in knobs.items())   # (slimm_amd/csrc/force.h)
    try:
        d = dense_mt_run(w, want_bins=True)
        s = Slimm.for_workload(w, device=0, grouped=grouped)
        s.push_records_packed(w.records, batch=25_000_000)
        assert s.get_profiles() is not None
        assert_equals_dense_mt(s, d)
        st = s.stats()
        print(f"ok seed {seed}: {n} records, {refs} refs, bins {st['total_bins']}, targets {st['n_targets']}, grouped {grouped}, "
              f"{knobs} ({time.time() - t0:.0f} s)", flush=True)
        s.close()
    except Exception as e:
        fails += 1
        print("FAIL seed", seed, cfg, knobs, "grouped", grouped, str(e)[:300].replace("\n", " | "), flush=True)
    os.environ.pop("SLIMM_FORCE", None)
    del w
print("fails", fails, "of", n_seeds)
