#!/bin/bash
# per-kernel lines at configs 2, 3 and 5 (resident, packed) for the variant libraries of build/var/:
#   scripts/front_variants.sh "k_front|k_filter" hw16 hw32 ...
PAT=$1; shift
for v in base "$@"; do
  if [ $v = base ]; then unset SLIMM_HIP_LIB; else export SLIMM_HIP_LIB=$PWD/build/var/$v/libslimm_hip.so; fi
  python bench.py --config config2 --breakdown --steps 10 --warmup 3 --push-files 0 --no-cpu-baseline --no-cli \
      --roofline-configs config3,config5 > /dev/null 2> /tmp/v.err
  echo "== $v"; grep -E "$PAT|device kernels|records, " /tmp/v.err | grep -v "host wall"
done
