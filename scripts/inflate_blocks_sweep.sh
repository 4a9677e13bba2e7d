#!/bin/bash
# decode / resolve kernel times of the device inflate against the number of BGZF blocks of a launch (how many waves of
# k_inflate_decode the chip holds at once shows as the step in its time): scripts/inflate_blocks_sweep.sh N1 N2 ...
cd "$GRAFT_REPO_ROOT"
for n in "$@"; do
  cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
  timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sweep_$n -o t -- python3 scripts/inflate_kernels.py 30000000 realistic $n 2 > gpurun_out/sweep_$n.txt 2>&1
  python3 - "$n" <<'PY'
import csv, sys
n = sys.argv[1]
d = {}
for x in csv.DictReader(open(f"gpurun_out/sweep_{n}/t_kernel_trace.csv")):
    for key in ("k_inflate_decode", "k_inflate_resolve"):
        if key in x["Kernel_Name"]:
            d.setdefault(key, []).append((int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e6)
print(f"{n} blocks:", ", ".join(f"{k} {min(v):.2f} ms (of {len(v)} launches)" for k, v in d.items()))
PY
done
