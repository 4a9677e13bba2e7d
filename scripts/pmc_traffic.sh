# HBM traffic of every kernel from the TCC counters, in separate passes (FETCH_SIZE and WRITE_SIZE do not fit one pass):
#   bash scripts/pmc_traffic.sh [config]      (default: config4, the headline workload)  -> gpurun_out/pmc_traffic/summary_<config>.json
# profiles/roundN/pmc_traffic_summary.json = {"<config>": <that summary>, ...} is what bench.py reads for roofline.traffic.
CFG=${1:-config4}; shift; EXTRA="$@"   # further arguments go to bench.py (e.g. --record-order any)
export TMPDIR=/tmp; R=$PWD; OUT=$R/gpurun_out/pmc_traffic; mkdir -p $OUT; cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT -o fetch_$CFG -- python3 $R/bench.py --config $CFG --steps 2 --warmup 1 --quick $EXTRA > /dev/null 2>$OUT/err_f.txt
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT -o write_$CFG -- python3 $R/bench.py --config $CFG --steps 2 --warmup 1 --quick $EXTRA > /dev/null 2>$OUT/err_w.txt
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum TCC_HIT_sum --kernel-trace --output-format csv -d $OUT -o req_$CFG -- python3 $R/bench.py --config $CFG --steps 2 --warmup 1 --quick $EXTRA > /dev/null 2>$OUT/err_r.txt
python3 - <<PY
import csv, collections, glob, json
def kname(full):
    s=full.replace('void ','').replace('slimm::','').replace('(anonymous namespace)::','')
    depth=0
    for i,c in enumerate(s):
        if c=='<': depth+=1
        elif c=='>': depth-=1
        elif c=='(' and depth==0: return s[:i]
    return s
res=collections.defaultdict(dict)
for tag in ("fetch","write","req"):
    try: rows=list(csv.DictReader(open("$OUT/%s_${CFG}_counter_collection.csv"%tag)))
    except Exception as e: print("missing",tag,e); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
    for r in rows:
        k=kname(r["Kernel_Name"])
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k,v in agg.items():
        if k.startswith("__amd") or "at::" in k or "elementwise" in k: continue
        for a,b in v.items(): res[k][a]=b/len(n[k])
json.dump(res, open("$OUT/summary_${CFG}.json","w"), indent=1)
for k,v in res.items(): print(k, {a:round(b,1) for a,b in v.items()})
PY
