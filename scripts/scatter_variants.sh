#!/bin/bash
# bucket scatter per-kernel lines at config 4 (1 B records) and config 3 for the variant libraries of build/var/ named on
# the command line (scripts/build_variant.sh), the library of the tree first
for v in base "$@"; do
  if [ $v = base ]; then unset SLIMM_HIP_LIB; else export SLIMM_HIP_LIB=$PWD/build/var/$v/libslimm_hip.so; fi
  echo "== $v"
  python bench.py --quick --breakdown --steps 5 --warmup 2 2>&1 >/dev/null | grep -E "^# (k_tile_scat|device)"
  python bench.py --config config3 --quick --breakdown --steps 10 2>&1 >/dev/null | grep -E "^# (k_tile_scat|device)"
done
unset SLIMM_HIP_LIB
echo "== config 3, SLIMM_FORCE=matrix=0 (phase B through the rounds)"
SLIMM_FORCE=matrix=0 python bench.py --config config3 --quick --breakdown --steps 10 2>&1 >/dev/null | grep -E "^# (k_tile_(scat|count2)|k_matrix|device)"
python bench.py --config config5 --quick --breakdown --steps 10 2>&1 >/dev/null | grep -E "^# (k_tile_scat|device)"
