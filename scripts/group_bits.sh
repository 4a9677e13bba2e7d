#!/bin/bash
# record_order = ANY: FEWER hash bits than log2 n (fewer passes, a heavier finish), measured on the GPU box.
#   scripts/group_bits.sh TAG config2 "24:8 18:9 20:10 16:8 22:11"
TAG=${1:-bits}; C=${2:-config2}; PLANS=${3:-"24:8 18:9 20:10"}
O=gpurun_out/$TAG; mkdir -p $O
for p in $PLANS; do b=${p%%:*}; rest=${p#*:}; w=${rest%%:*}; P=0; case $rest in *:*) P=${rest##*:};; esac
  SLIMM_FORCE=group_bits=$b,group_width=$w,group_passes=$P python3 bench.py --quick --engines 1 --config $C --record-order any --breakdown --steps 10 --warmup 3 \
      > $O/${C}_b${b}_w${w}.json 2> $O/${C}_b${b}_w${w}.txt
  python3 - <<PY
import json
d = json.load(open("$O/${C}_b${b}_w${w}.json"))
k = d.get("kernels", {})
print("$C bits $b width $w: %.3f ms/step" % d["ms_per_step"], {n: (round(v["us"], 1)) for n, v in k.items() if n.startswith("k_group") or n == "k_front"}, d["config"]["profile_sha1"][:8])
PY
done 2>&1 | tee $O/summary.txt
