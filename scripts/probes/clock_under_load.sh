#!/bin/bash
# the GPU's clocks and power while the headline workload runs: scripts/probes/clock_under_load.sh
python bench.py --quick --steps 4000 --warmup 2 > /dev/null 2>&1 &
B=$!
sleep 40
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Socket" | tr -s ' ' | head -6
  echo --
  sleep 0.4
done
wait $B
