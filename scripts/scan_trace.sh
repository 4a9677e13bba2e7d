# scratch: rocprofv3 durations of the record-tile scan kernels at a given size (chunked vs one workgroup)
N=${1:-36000000}
export TMPDIR=/tmp; R=$PWD; mkdir -p $R/gpurun_out/scan; cd /tmp
for mode in chunked single; do
  if [ $mode = single ]; then export SLIMM_SCAN_CHUNK=100000000; else unset SLIMM_SCAN_CHUNK; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/scan -o $mode -- python3 $R/scripts/big_configs.py config2 $N > /dev/null 2>&1
  echo "== $mode"; grep -E "k_scan" $R/gpurun_out/scan/${mode}_kernel_stats.csv | cut -d, -f1-5
done
