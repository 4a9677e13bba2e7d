# rocprofv3 kernel trace + stats of the default bench (its own steps and warm-up: the first launch of a kernel in a process is
# 20-40 % slower than the rest, and an average over eight launches carries it); summary goes to gpurun_out/<tag>/
TAG=${1:-trace}
export TMPDIR=/tmp; R=$PWD; mkdir -p $R/gpurun_out/$TAG; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -o t -- python3 $R/bench.py --steps 20 --warmup 5 --quick > $R/gpurun_out/$TAG/bench.json 2> $R/gpurun_out/$TAG/bench.err
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/$TAG/t_kernel_trace.csv")))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_front' in r['Kernel_Name']]
s=idx[-1]; prev=None; busy=0; first=None; last=None
for r in rows[s-2:]:
    st=int(r['Start_Timestamp']); en=int(r['End_Timestamp'])
    gap=(st-prev) if prev else 0
    name=r['Kernel_Name'].split('(')[0].replace('void slimm::','').replace('slimm::','')[:34]
    print(f"{name:36s} dur {(en-st)/1000:8.1f} us  gap {gap/1000:8.1f} us")
    prev=en; busy+=en-st; first=first or st; last=en
print("busy", busy/1000, "span", (last-first)/1000)
PY
