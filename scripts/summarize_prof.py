"""Condense rocprofv3 CSV output (kernel trace + stats + PMC passes) into a text summary for profiles/.

Dispatches are grouped by (kernel, grid size) because the coarse (64 samples/ray) and fine
(192 samples/ray) networks launch the same kernel with different grids."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

out = sys.argv[1]


def find(sub, pat):
    hits = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return hits[0] if hits else None


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:64]


p = find("trace", "*kernel_stats.csv")
if p:
    print("== rocprofv3 --kernel-trace --stats :: bench_kernel_stats.csv (top rows) ==")
    for r in list(csv.DictReader(open(p)))[:14]:
        print(f"{short(r['Name']):64s} calls={r['Calls']:>4s} total_ns={r['TotalDurationNs']:>11s} avg_ns={float(r['AverageNs']):>10.0f} pct={float(r['Percentage']):.3f}")
p = find("trace", "*kernel_trace.csv")
if p:
    g = defaultdict(list)
    meta = {}
    for r in csv.DictReader(open(p)):
        k = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]))
        g[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        meta[k] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"])
    print("== kernel trace grouped by (kernel, grid threads) ==")
    for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1]))[:16]:
        m = meta[k]
        print(f"{k[0]:64s} grid={k[1]:>9d} n={len(v):>3d} avg_us={sum(v) / len(v) / 1e3:>9.1f} min_us={min(v) / 1e3:>9.1f} "
              f"vgpr={m[0]} agpr={m[1]} sgpr={m[2]} lds={m[3]} scratch={m[4]}")
else:
    print("no kernel trace found under", out)

for sub in ("pmc_mfma", "pmc_fetch", "pmc_write", "pmc_sq", "pmc_lds"):
    p = find(sub, "*counter_collection.csv")
    if not p:
        continue
    agg = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(p)):
        agg[(short(r["Kernel_Name"]), int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"== {sub} :: per-dispatch mean of each counter, grouped by (kernel, grid threads) ==")
    ranked = sorted(agg.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values()))
    # the four largest, plus the MLP kernels whatever their rank (the roofline kernel's traffic is small by design)
    rows = ranked[:4] + [kv for kv in ranked[4:] if kv[0][0].startswith(("mlp_", "wgrad_kernel"))]
    for k, cs in rows:
        desc = "  ".join(f"{c}={sum(v) / len(v):.5g}" for c, v in sorted(cs.items()))
        n = len(next(iter(cs.values())))
        print(f"{k[0]:48s} grid={k[1]:>9d} n={n:>3d}  {desc}")
