#!/bin/bash
# Runs on the GPU box: rocprofv3 PMC passes (each counter group in its own run, --kernel-trace only) of an arbitrary repo command.
# Usage: scripts/gpu_pmc_cmd.sh <tag> <command ...>   -> gpurun_out/pmc_<tag>/<group>/..., per-kernel averages printed
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
run() { name=$1; shift; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o run -- "${CMD[@]}" > $OUT/$name.log 2>&1 < /dev/null); echo "$name rc=$?"; }
CMD=("$@")
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS
run wait SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY
python3 - <<PY
import csv, glob, collections
for grp in ("mfma", "lds", "wait"):
    fs = glob.glob("$OUT/%s/**/*counter_collection.csv" % grp, recursive=True)
    if not fs:
        print(grp, "no counter file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        n = r["Kernel_Name"]
        short = n[n.index("::") + 2:n.index("(", n.index("::"))] if "anonymous" in n else n[:30]
        agg[(short, int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items(), key=lambda kv: -sum(sum(x) for x in kv[1].values()))[:8]:
        print(grp, k[0][:28], "grid", k[1], " ".join(f"{c}={sum(x)/len(x):.4g}" for c, x in v.items()), "n=%d" % len(next(iter(v.values()))))
PY
