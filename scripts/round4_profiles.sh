#!/bin/bash
# Round 4's judged figures, collected on the GPU box into gpurun_out/<tag>/ (copy what is to be kept into profiles/round4/):
#   scripts/round4_profiles.sh TAG
# = scripts/round_profiles.sh (headline bench, kernel trace, SQ counters, PMC traffic at configs 4 and 2) + the
# record_order = ANY path (kernel stats, PMC traffic and SQ counters of the grouping kernels at configs 2 and 3) + the
# `slimm` command's stage trace on a 100 M-record BAM + the GPU test suite.
TAG=${1:-round4}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
bash scripts/round_profiles.sh $TAG > $O/round_profiles.log 2>&1
export TMPDIR=/tmp
for c in config2 config3; do
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/any_$c -o t -- python3 $R/bench.py --steps 5 --warmup 2 --quick \
      --config $c --record-order any --breakdown > $O/any_${c}_bench.json 2> $O/any_${c}_breakdown.txt )
  cp $O/any_$c/t_kernel_stats.csv $O/any_${c}_kernel_stats.csv 2>/dev/null
  bash scripts/pmc_traffic.sh $c --record-order any > $O/any_${c}_pmc_traffic.txt 2>&1
  rm -rf $O/any_$c
done
bash scripts/pmc_sq.sh $TAG/any_sq --config config2 --record-order any > /dev/null 2>&1; cp $O/any_sq/summary.txt $O/any_config2_sq_counters.txt 2>/dev/null; rm -rf $O/any_sq
# files back to back (bench --engines 2, the default) against one context: config 2's step and the idle device between files
for e in 1 2; do
  python3 bench.py --quick --config config2 --engines $e --steps 100 --no-any-order 2> /dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config2 --engines $e: %.4f ms per file, %.0f M records/s' % (d['ms_per_step'], d['value']))" >> $O/config2_engines.txt
  ( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O/gaps$e -o t -- python3 $R/bench.py --quick --config config2 --engines $e --steps 6 --warmup 2 --no-any-order > /dev/null 2>&1 )
  echo "== bench.py --quick --config config2 --engines $e: the last three files" >> $O/config2_engines.txt
  python3 scripts/gaps.py $O/gaps$e/t_kernel_trace.csv 3 >> $O/config2_engines.txt; rm -rf $O/gaps$e
done
bash scripts/host_trace.sh $TAG/ht > $O/host_trace.txt 2>&1; rm -rf $O/ht
python3 scripts/inflate_rate.py 1000000 10 > $O/inflate_rate.txt 2>&1
python3 scripts/cli_e2e.py 100000000 config3 > $O/cli_100m_trace.txt 2>&1
python3 scripts/cli_exit_modes.py > $O/cli_100m_exit_modes.txt 2>&1
python3 -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1
rm -rf $O/trace/*.csv.bak $O/sq/*.csv; du -sh $O
