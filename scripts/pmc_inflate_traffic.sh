# HBM traffic of the inflate kernels from the TCC counters, in separate passes: bash scripts/pmc_inflate_traffic.sh [inflate_kernels.py args]
export TMPDIR=/tmp; R=$PWD; OUT=$R/gpurun_out/pmc_inflate; mkdir -p $OUT; cd /tmp
ARGS="${@:-30000000 realistic 65536 1}"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT -o fetch -- python3 $R/scripts/inflate_kernels.py $ARGS > /dev/null 2>$OUT/err_f.txt
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT -o write -- python3 $R/scripts/inflate_kernels.py $ARGS > /dev/null 2>$OUT/err_w.txt
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT -o req -- python3 $R/scripts/inflate_kernels.py $ARGS > /dev/null 2>$OUT/err_r.txt
python3 - <<PY
import csv, collections, json
def kname(full):
    s=full.replace('void ','').replace('slimm::','').replace('(anonymous namespace)::','')
    return s.split('(')[0]
res=collections.defaultdict(dict)
for tag in ("fetch","write","req"):
    try: rows=list(csv.DictReader(open("$OUT/%s_counter_collection.csv"%tag)))
    except Exception as e: print("missing",tag,e); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
    for r in rows:
        k=kname(r["Kernel_Name"])
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k,v in agg.items():
        if k.startswith("__amd"): continue
        for a,b in v.items(): res[k][a]=b/len(n[k])
json.dump(res, open("$OUT/summary.json","w"), indent=1)
for k,v in res.items():
    f=2.0*v.get("FETCH_SIZE",0)*1024; w=v.get("WRITE_SIZE",0)*1024
    print(f"{k}: fetched {f/1e9:.2f} GB (2 x FETCH_SIZE, the gfx950 correction), written {w/1e9:.2f} GB per launch; {v}")
PY
