#!/bin/bash
# scratch: build runs.hip with extra -D flags on the GPU box; usage: exp_q.sh "-DA=1" "-DA=2" ...
for f in "$@"; do
  touch slimm_amd/csrc/runs.hip
  make -C slimm_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC $f" 2>&1 | grep -E "error"
  echo "== $f"
  python bench.py --no-cpu-baseline --steps 20 --warmup 4 --breakdown 2>&1 | grep -E "^# k_runs |device kernels"
done
