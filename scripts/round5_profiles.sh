#!/bin/bash
# Round 5's judged figures, collected on the GPU box into gpurun_out/<tag>/ (copy what is to be kept into profiles/round5/):
#   scripts/round5_profiles.sh TAG
# = scripts/round_profiles.sh (headline bench, kernel trace, SQ counters, PMC traffic at configs 4 and 2) + the inflate kernels
# (kernel trace, SQ counters, PMC traffic) + the command on the realistic / easy BAM and on SAM text + the stress scripts.
TAG=${1:-round5}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
bash scripts/round_profiles.sh $TAG > $O/round_profiles.log 2>&1
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/inf -o t -- python3 $R/scripts/inflate_kernels.py 30000000 realistic 65536 3 > $O/inflate_rate.txt 2>&1 )
cp $O/inf/t_kernel_stats.csv $O/inflate_kernel_stats.csv 2>/dev/null; rm -rf $O/inf
bash scripts/pmc_inflate.sh $TAG/inf_sq 30000000 realistic 65536 1 > $O/inflate_sq_counters.txt 2>&1; rm -rf $O/inf_sq
bash scripts/pmc_inflate_traffic.sh 30000000 realistic 65536 1 > $O/inflate_pmc_traffic.txt 2>&1
python3 scripts/realistic_cli.py 100000000 both 32768 > $O/realistic_cli.txt 2>&1
python3 scripts/sam_cli.py 100000000 > $O/sam_cli.txt 2>&1
python3 scripts/stress_inflate.py 60 7000 > $O/stress_inflate.txt 2>&1
python3 scripts/stress_bgzf.py 30 4000 > $O/stress_bgzf.txt 2>&1
rm -rf $O/trace/*.csv.bak $O/sq/*.csv; du -sh $O
