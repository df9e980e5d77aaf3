#!/bin/bash
# Runs on the GPU box (via gpurun): the -m gpu suite, the default bench, and a kernel-trace profile of the bench.
# Usage: scripts/gpu_round3.sh <tag> [tests|bench|prof ...]     -> gpurun_out/<tag>_*
set -u
TAG=${1:-r03a}; shift || true
WHAT=${*:-tests bench prof}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
for w in $WHAT; do
  case $w in
    tests)
      timeout 2400 python -m pytest tests -m gpu -q -rA --durations=15 -p no:cacheprovider > $OUT/${TAG}_pytest.log 2>&1
      echo "pytest rc=$?" >> $OUT/${TAG}_pytest.log
      tail -n 40 $OUT/${TAG}_pytest.log ;;
    bench)
      timeout 900 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
      echo "bench rc=$?"; tail -c 3000 $OUT/${TAG}_bench.json ;;
    prof)
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}/trace -o bench -- python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/prof_${TAG}_trace.log 2>&1)
      python scripts/summarize_prof.py $OUT/prof_${TAG} > $OUT/prof_${TAG}/summary_trace.txt 2>&1
      find $OUT/prof_${TAG} -name "*kernel_trace.csv" -size +2M -delete; find $OUT/prof_${TAG} -name "*.db" -delete
      head -n 60 $OUT/prof_${TAG}/summary_trace.txt ;;
    pmc)
      BENCH="python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-train"      # (extras on: the opt-in render legs incl. f16x2)
      (cd /tmp
       timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/prof_${TAG}/pmc_mfma -o bench -- $BENCH > $OUT/prof_${TAG}_pmc_mfma.log 2>&1
       timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_${TAG}/pmc_fetch -o bench -- $BENCH > $OUT/prof_${TAG}_pmc_fetch.log 2>&1
       timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_${TAG}/pmc_write -o bench -- $BENCH > $OUT/prof_${TAG}_pmc_write.log 2>&1)
      python scripts/summarize_prof.py $OUT/prof_${TAG} > $OUT/prof_${TAG}/summary_pmc.txt 2>&1
      find $OUT/prof_${TAG} -name "*.csv" -size +2M -delete; find $OUT/prof_${TAG} -name "*.db" -delete
      head -n 80 $OUT/prof_${TAG}/summary_pmc.txt ;;
  esac
done
