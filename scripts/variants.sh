#!/bin/bash
# bench --breakdown lines of the given kernels at config 2 for each variant library: scripts/variants.sh "k_front|k_filter" fb8 fb6 ...
PAT=$1; shift
for v in base "$@"; do
  if [ $v = base ]; then unset SLIMM_HIP_LIB; else export SLIMM_HIP_LIB=$PWD/build/var/$v/libslimm_hip.so; fi
  python bench.py --config config2 --quick --breakdown --steps 10 > /dev/null 2> /tmp/v.err
  echo "== $v"; grep -E "$PAT" /tmp/v.err | grep -v "host wall"
done
