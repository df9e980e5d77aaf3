#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + PMC passes of the default bench command.
# Usage: scripts/gpu_profile.sh <tag>      -> gpurun_out/prof_<tag>/...
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline"
cd /tmp
# 1. per-kernel time
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $BENCH > $OUT/trace.log 2>&1
# 2. counters, each in its own pass (never combined with other trace domains)
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o bench -- $BENCH > $OUT/pmc_mfma.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- $BENCH > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- $BENCH > $OUT/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc_sq -o bench -- $BENCH > $OUT/pmc_sq.log 2>&1
cd $ROOT
python scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
