#!/bin/bash
# record_order = ANY at config 2 (or $CONFIG) for variant libraries built by scripts/build_variant.sh:
#   scripts/group_libs.sh TAG "base gi4:512 gb256:1024"      (name[:group_grid of SLIMM_FORCE])
TAG=${1:-glibs}; O=gpurun_out/$TAG; mkdir -p $O; C=${CONFIG:-config2}
for spec in $2; do
  v=${spec%%:*}; g=${spec#*:}; [ "$g" = "$spec" ] && g=
  if [ $v = base ]; then unset SLIMM_HIP_LIB; else export SLIMM_HIP_LIB=$PWD/build/var/$v/libslimm_hip.so; fi
  if [ -n "$g" ]; then export SLIMM_FORCE=group_grid=$g; else unset SLIMM_FORCE; fi
  python3 bench.py --quick --config $C --record-order any --breakdown --steps 10 --warmup 3 > $O/$v.json 2> $O/$v.txt
  python3 - <<PY
import json
d = json.load(open("$O/$v.json"))
k = d.get("kernels", {})
print("$spec: %.3f ms/step" % d["ms_per_step"], {n: round(v["us"], 1) for n, v in k.items() if n.startswith("k_group") or n == "k_front"})
PY
done 2>&1 | tee $O/summary.txt
