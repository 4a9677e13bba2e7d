O=gpurun_out/r6c; mkdir -p $O
bash scripts/front_variants.sh "k_front|k_filter" fb2 fb8 > $O/filter_variants.txt 2>&1
for v in base fb2 fb8; do
  if [ $v = base ]; then unset SLIMM_HIP_LIB; else export SLIMM_HIP_LIB=$PWD/build/var/$v/libslimm_hip.so; fi
  echo "== $v (1 B records)"; python3 bench.py --breakdown --quick --no-cli --no-cpu-baseline 2>&1 >/dev/null | grep -E "^# (k_filter|k_front|device)"
done >> $O/filter_variants.txt 2>&1
unset SLIMM_HIP_LIB
cat $O/filter_variants.txt
python3 -m pytest tests -m gpu -q -x > $O/gpu_tests.txt 2>&1; tail -5 $O/gpu_tests.txt
