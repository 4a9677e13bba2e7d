#!/bin/bash
# record_order = ANY: the grouping's plan knobs measured on the GPU box (bench.py --quick --record-order any).
#   scripts/group_variants.sh TAG "config2 config3" "8 9 10 11" "512 256"
TAG=${1:-group}; CONFIGS=${2:-config2}; WIDTHS=${3:-"8 11"}; GRIDS=${4:-"512"}
O=gpurun_out/$TAG; mkdir -p $O
for c in $CONFIGS; do for w in $WIDTHS; do for g in $GRIDS; do
  SLIMM_FORCE=group_width=$w,group_grid=$g python3 bench.py --quick --config $c --record-order any --breakdown --steps 10 --warmup 3 \
      > $O/${c}_w${w}_g${g}.json 2> $O/${c}_w${w}_g${g}.txt
  python3 - <<PY
import json
d = json.load(open("$O/${c}_w${w}_g${g}.json"))
k = d.get("kernels", {})
print("$c width $w grid $g: %.3f ms/step" % d["ms_per_step"], {n: round(v["us"], 1) for n, v in k.items() if n.startswith("k_group") or n == "k_front"})
PY
done; done; done 2>&1 | tee $O/summary.txt
