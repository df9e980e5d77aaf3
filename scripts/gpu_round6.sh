#!/bin/bash
# Runs on the GPU box (via gpurun): pieces of the round-6 evidence.   Usage: scripts/gpu_round6.sh <tag> [bench|tests|smoke|prof|pmc|extras ...]
set -u
TAG=${1:-r06a}; shift || true
WHAT=${*:-bench tests}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
for w in $WHAT; do
  case $w in
    bench)   # the driver's command, byte for byte; stdout and stderr kept apart, wall time taken outside
      t0=$(python3 -c "import time; print(time.time())")
      timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_bench_stdout.txt 2> $OUT/${TAG}_bench_stderr.txt
      rc=$?; t1=$(python3 -c "import time; print(time.time())")
      echo "bench rc=$rc wall_s=$(python3 -c "print(round($t1 - $t0, 1))") stdout_bytes=$(wc -c < $OUT/${TAG}_bench_stdout.txt) stdout_lines=$(wc -l < $OUT/${TAG}_bench_stdout.txt)" | tee $OUT/${TAG}_bench_meta.txt
      cp bench_full.json $OUT/${TAG}_bench_full.json 2>/dev/null
      cat $OUT/${TAG}_bench_stdout.txt ;;
    extras)
      timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --extras --cpu-seconds 80 --full-record $OUT/${TAG}_bench_extras_full.json > $OUT/${TAG}_bench_extras_stdout.txt 2> $OUT/${TAG}_bench_extras_stderr.txt
      echo "extras rc=$?"; cat $OUT/${TAG}_bench_extras_stdout.txt ;;
    tests)
      timeout 1500 python -m pytest tests -x -m gpu -q --durations=40 -p no:cacheprovider > $OUT/${TAG}_pytest.log 2>&1
      echo "pytest rc=$?" >> $OUT/${TAG}_pytest.log
      tail -n 60 $OUT/${TAG}_pytest.log ;;
    smoke)
      timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${TAG}_smoke.txt 2>&1; echo "smoke rc=$?"; tail -n 12 $OUT/${TAG}_smoke.txt ;;
    prof)
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}/trace -o bench -- python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --full-record $OUT/prof_${TAG}_bench_full.json > $OUT/prof_${TAG}_trace.log 2>&1)
      python scripts/summarize_prof.py $OUT/prof_${TAG} > $OUT/prof_${TAG}/summary.txt 2>&1
      find $OUT/prof_${TAG} -name "*kernel_trace.csv" -delete     # (tens of MiB; gpurun_out/ merges back only below 64 MiB)
      head -n 60 $OUT/prof_${TAG}/summary.txt ;;
    pmc)
      BENCH="python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --full-record /tmp/bench_full_pmc.json"
      (cd /tmp
       timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/prof_${TAG}/pmc_mfma -o bench -- $BENCH > $OUT/prof_${TAG}_pmc_mfma.log 2>&1
       timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_${TAG}/pmc_fetch -o bench -- $BENCH > $OUT/prof_${TAG}_pmc_fetch.log 2>&1
       timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_${TAG}/pmc_write -o bench -- $BENCH > $OUT/prof_${TAG}_pmc_write.log 2>&1)
      python scripts/summarize_prof.py $OUT/prof_${TAG} > $OUT/prof_${TAG}/summary_pmc.txt 2>&1
      python scripts/make_pmc_traffic.py $OUT/prof_${TAG} "scripts/gpu_round6.sh ${TAG} pmc (round-6 tree; separate --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of \`bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline\`): profiles/r06/rocprofv3_pmc_r06.txt" > $OUT/prof_${TAG}/make_pmc_traffic.log 2>&1
      cp profiles/pmc_traffic.json $OUT/${TAG}_pmc_traffic.json
      find $OUT/prof_${TAG} -name "*kernel_trace.csv" -delete -o -name "*counter_collection.csv" -delete
      head -n 80 $OUT/prof_${TAG}/summary_pmc.txt ;;
  esac
done
