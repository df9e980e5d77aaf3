"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a small text summary for profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(sub, pat):
    hits = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return hits[0] if hits else None


def short(name):
    return name.split("(")[0][-70:]


p = find("trace", "*kernel_stats.csv")
if p:
    print("== rocprofv3 --kernel-trace --stats :: kernel_stats ==")
    rows = list(csv.DictReader(open(p)))
    for r in rows[:12]:
        print(f"{short(r['Name']):72s} calls={r['Calls']:>5s} total_ns={r['TotalDurationNs']:>12s} avg_ns={float(r['AverageNs']):>12.0f} pct={r['Percentage']}")
else:
    print("no kernel_stats.csv found under", out)

for sub in ("pmc_mfma", "pmc_fetch", "pmc_write", "pmc_sq"):
    p = find(sub, "*counter_collection.csv")
    if not p:
        print(f"== {sub}: no counter_collection.csv ==")
        continue
    agg = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(p)):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"== {sub} :: per-dispatch mean of counters (n dispatches) ==")
    for k, cs in sorted(agg.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values()))[:6]:
        desc = "  ".join(f"{c}={sum(v) / len(v):.4g} (n={len(v)})" for c, v in sorted(cs.items()))
        print(f"{k:72s} {desc}")
