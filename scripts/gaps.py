"""Kernel-trace gaps of the last files of a rocprofv3 --kernel-trace CSV: python scripts/gaps.py t_kernel_trace.csv [n_files]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 2
idx = [i for i, r in enumerate(rows) if 'k_front' in r['Kernel_Name']]
s = idx[-nf]; prev = None; busy = 0; first = None; last = None
for r in rows[s - 1:]:
    st = int(r['Start_Timestamp']); en = int(r['End_Timestamp'])
    gap = (st - prev) if prev else 0
    name = r['Kernel_Name'].split('(')[0].replace('void slimm::', '').replace('slimm::', '')[:34]
    q = r.get('Queue_Id', '?')
    print(f"{name:36s} q{q:>3s} dur {(en - st) / 1000:8.1f} us  gap {gap / 1000:8.1f} us")
    prev = en; busy += en - st; first = first or st; last = en
print("busy", busy / 1000, "span", (last - first) / 1000)
