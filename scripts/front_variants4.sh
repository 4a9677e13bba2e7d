#!/bin/bash
# kernels matching PAT at config 4 (1 B records) and at the configs named in CONFIGS (default: config2 config3 config5)
# for the variant libraries of build/var/:   scripts/front_variants4.sh "k_front|k_filter" prev ...
PAT=$1; shift
for v in base "$@"; do
  if [ $v = base ]; then unset SLIMM_HIP_LIB; else export SLIMM_HIP_LIB=$PWD/build/var/$v/libslimm_hip.so; fi
  echo "== $v"
  python bench.py --quick --breakdown --steps 5 --warmup 2 2>&1 >/dev/null | grep -E "^# ($PAT|device)"
  for c in ${CONFIGS:-config2 config3 config5}; do python bench.py --config $c --quick --breakdown --steps 10 2>&1 >/dev/null | grep -E "^# ($PAT|device)"; done
done
