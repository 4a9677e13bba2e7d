# k_filter_compact: timing-only variants (wrong results) that take one part out each, and the SQ counters of the tree's kernel
O=gpurun_out/r6d; mkdir -p $O
for v in base fnogather fstage1; do
  if [ $v = base ]; then unset SLIMM_HIP_LIB; else export SLIMM_HIP_LIB=$PWD/build/var/$v/libslimm_hip.so; fi
  echo "== $v (1 B records)"; python3 bench.py --breakdown --quick --no-cli --no-cpu-baseline --no-any-order 2>&1 >/dev/null | grep -E "^# (k_filter|k_front|device)"
done > $O/filter_probe.txt 2>&1
unset SLIMM_HIP_LIB
bash scripts/pmc_sq.sh r6d/sq --no-cli --no-cpu-baseline --no-any-order > /dev/null 2>&1; grep -E "k_filter|k_front" $O/sq/summary.txt > $O/sq_counters.txt
cat $O/filter_probe.txt $O/sq_counters.txt
