export TMPDIR=/tmp; R=$PWD; mkdir -p $R/gpurun_out/pmc2; cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmc2 -o a -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>$R/gpurun_out/pmc2/err_a.txt
rocprofv3 --pmc SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmc2 -o b -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>$R/gpurun_out/pmc2/err_b.txt
python3 - <<PY
import csv, collections, glob
for f in sorted(glob.glob("$R/gpurun_out/pmc2/*_counter_collection.csv")):
    rows=list(csv.DictReader(open(f)))
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
    for r in rows:
        k=r["Kernel_Name"].split("(")[0].replace("void slimm::","").replace("slimm::","")[:28]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k,v in agg.items():
        if k.startswith("__amd"): continue
        print(k, len(n[k]), {a:round(b/len(n[k])) for a,b in sorted(v.items())})
PY
