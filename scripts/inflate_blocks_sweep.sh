#!/bin/bash
# decode / resolve kernel times of the device inflate against the number of BGZF blocks of a launch (how many waves of
# k_inflate_decode the chip holds at once shows as the step in its time): scripts/inflate_blocks_sweep.sh N1 N2 ...
cd "$GRAFT_REPO_ROOT"
for n in "$@"; do
  cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
  timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sweep_$n -o t -- python3 scripts/inflate_kernels.py 30000000 realistic $n 2 > gpurun_out/sweep_$n.txt 2>&1
  echo "== $n blocks"; grep -E "k_inflate_(decode|resolve)" gpurun_out/sweep_$n/t_kernel_stats.csv | awk -F'","' '{print substr($1,2,30), "avg ns", $4, "min", $6}' | sed 's/"//g'
done
