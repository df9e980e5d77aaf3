"""Per (kernel, grid) durations from a rocprofv3 kernel trace CSV.  Usage: python scripts/kernel_times.py <dir>"""
import csv, glob, os, sys
from collections import defaultdict
p = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
d = defaultdict(list)
for r in csv.DictReader(open(p)):
    d[(r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:44], r.get("Grid_Size_X") or r.get("Grid_Size"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:int(sys.argv[2]) if len(sys.argv) > 2 else 9]:
    print(f"{k[0]:44s} grid {k[1]:>8s} n {len(v):3d} avg {sum(v) / len(v):8.0f} min {min(v):8.0f} us")
