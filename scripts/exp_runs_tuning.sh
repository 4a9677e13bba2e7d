#!/bin/bash
# scratch: k_runs with other occupancy targets / pass sizes (rebuilds runs.o on the GPU box, restores nothing: run in a throw-away copy)
for flags in "-DSLIMM_RUNS_MINBLOCKS=3 -DSLIMM_Q_ITEMS=4" "-DSLIMM_RUNS_MINBLOCKS=4 -DSLIMM_Q_ITEMS=4" "-DSLIMM_RUNS_MINBLOCKS=2 -DSLIMM_Q_ITEMS=4" "-DSLIMM_RUNS_MINBLOCKS=2 -DSLIMM_Q_ITEMS=8" "-DSLIMM_RUNS_MINBLOCKS=5 -DSLIMM_Q_ITEMS=4"; do
  touch slimm_amd/csrc/runs.hip
  make -C slimm_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC $flags" 2>&1 | grep -E " error" 
  echo "== $flags"
  for i in 1 2; do python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['kernel'], d['roofline']['ms_per_launch'])"; done
done
