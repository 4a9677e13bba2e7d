#!/bin/bash
for q in "$@"; do
  touch slimm_amd/csrc/runs.hip
  make -C slimm_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC -DSLIMM_Q_ITEMS=$q" 2>&1 | grep -E "error"
  echo "== Q=$q"
  python bench.py --no-cpu-baseline --steps 20 --warmup 4 --breakdown 2>&1 | grep -E "^# k_runs |device kernels"
done
