# SQ instruction / wait counters of the inflate kernels (one rocprofv3 --pmc pass): scripts/pmc_inflate.sh TAG [inflate_kernels.py args]
TAG=${1:-sq_inf}; shift
export TMPDIR=/tmp; R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp
rocprofv3 --pmc ${PMC:-SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS} --kernel-trace --output-format csv -d $OUT -o a -- python3 $R/scripts/inflate_kernels.py "$@" > $OUT/out_a.txt 2>$OUT/err_a.txt
python3 - <<PY
import csv, collections, glob
def kname(full):
    s=full.replace('void ','').replace('slimm::','').replace('(anonymous namespace)::','')
    depth=0
    for i,c in enumerate(s):
        if c=='<': depth+=1
        elif c=='>': depth-=1
        elif c=='(' and depth==0: return s[:i]
    return s
for f in sorted(glob.glob("$OUT/*_counter_collection.csv")):
    rows=list(csv.DictReader(open(f)))
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
    for r in rows:
        k=kname(r["Kernel_Name"])
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    with open("$OUT/summary.txt","w") as o:
        for k,v in agg.items():
            if k.startswith("__amd"): continue
            line=f"{k} {len(n[k])} " + str({a:round(b/len(n[k])) for a,b in sorted(v.items())})
            print(line); o.write(line+"\n")
PY
