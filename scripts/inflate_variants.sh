#!/bin/bash
# decode / resolve kernel times of library variants (scripts/build_variant.sh): scripts/inflate_variants.sh NAME ...
cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
  lib=slimm_amd/libslimm_hip.so; [ "$v" != base ] && lib=build/var/$v/libslimm_hip.so
  SLIMM_HIP_LIB=$PWD/$lib timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/var_$v -o t -- python3 scripts/inflate_kernels.py 30000000 realistic 65536 2 > gpurun_out/var_$v.txt 2>&1
  echo "== $v"; grep -E "k_inflate_(decode|resolve)" gpurun_out/var_$v/t_kernel_stats.csv | awk -F'","' '{print substr($1,1,40), $2, $4}' | sed 's/"//g'
done
