#!/bin/bash
# record_order = ANY at config 4 (1 B records) and config 3 for the variant libraries of build/var/ named on the command line
for v in base "$@"; do
  if [ $v = base ]; then unset SLIMM_HIP_LIB; else export SLIMM_HIP_LIB=$PWD/build/var/$v/libslimm_hip.so; fi
  echo "== $v"
  for c in config4 config3; do
    python3 bench.py --quick --config $c --record-order any --breakdown --steps 5 --warmup 2 2>&1 >/dev/null | grep -E "^# (k_group|k_front|device)"
  done
done
