#!/bin/bash
# the inflate part of scripts/round5_profiles.sh + the blocks sweep + the resolve probes
TAG=set06b
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/inf -o t -- python3 $R/scripts/inflate_kernels.py 30000000 realistic 65536 3 > $O/inflate_rate.txt 2>&1 )
cp $O/inf/t_kernel_stats.csv $O/inflate_kernel_stats.csv 2>/dev/null; rm -rf $O/inf
bash scripts/pmc_inflate.sh $TAG/inf_sq 30000000 realistic 65536 1 > $O/inflate_sq_counters.txt 2>&1; rm -rf $O/inf_sq
bash scripts/pmc_inflate_traffic.sh 30000000 realistic 65536 1 > $O/inflate_pmc_traffic.txt 2>&1
bash scripts/inflate_blocks_sweep.sh 8192 32768 65536 131072 > $O/inflate_blocks_sweep.txt 2>&1
python3 scripts/realistic_cli.py 100000000 realistic 65536 > $O/realistic_cli.txt 2>&1
python3 scripts/stress_inflate.py 60 7000 > $O/stress_inflate.txt 2>&1
python3 scripts/stress_bgzf.py 30 4000 > $O/stress_bgzf.txt 2>&1
SLIMM_HIP_LIB=build/var/rprof/libslimm_hip.so python3 scripts/tprof_resolve.py 30000000 realistic 65536 > $O/tprof_resolve.txt 2>&1
