#!/bin/bash
# scratch: copies of the tile counters / cursors (contention of the bucket reservations vs the cost of summing the copies)
for r in 8 16 32 4; do
  touch slimm_amd/csrc/*.hip
  make -C slimm_amd/csrc -j4 CXXFLAGS="-O3 -std=c++17 -fPIC -DSLIMM_TILE_REPS=$r" 2>&1 | grep -E " error"
  echo "== reps $r"
  python bench.py --no-cpu-baseline --breakdown 2>&1 | grep -E "k_tile_scatter |k_tile_scatter2|k_tile_count |device kernels"
done
