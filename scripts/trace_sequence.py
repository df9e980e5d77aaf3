"""Print the kernel sequence between the k-th and (k+2)-th occurrence of a marker kernel in a rocprofv3 kernel trace CSV:
python scripts/trace_sequence.py <run_kernel_trace.csv> [marker substring] [k]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "ray_embed"
k = int(sys.argv[3]) if len(sys.argv) > 3 else 2
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = []
for r in rows:
    n = r["Kernel_Name"]
    short = n[n.index("::") + 2:n.index("(", n.index("::"))] if "anonymous" in n else n[:40]
    seq.append((short, int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["Start_Timestamp"])))
idx = [i for i, s in enumerate(seq) if marker in s[0]]
for start in range(k, len(idx) - 2, 4):
    i0, i1 = idx[start], idx[start + 2]
    t0 = seq[i0][4]
    print("----")
    for s in seq[i0:i1]:
        print(f"{(s[4] - t0) / 1e3:9.1f} us  {s[0][:40]:40s} blocks {s[1]:6d} x {s[2]:2d}  {s[3]:8.1f} us")
    print(f"total {(seq[i1][4] - t0) / 1e3:.1f} us")
