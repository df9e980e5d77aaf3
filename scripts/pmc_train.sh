set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; TAG=${TAG:-r03q}
export TMPDIR=/tmp
cd /tmp
for mode in f16x2 ""; do
  CMD="python $ROOT/scripts/step_profile.py --rays 4096 --steps 6 --mode=$mode"
  sfx=${mode:-f32}
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_${TAG}_$sfx/pmc_fetch -o bench -- $CMD > $OUT/prof_${TAG}_${sfx}_fetch.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_${TAG}_$sfx/pmc_write -o bench -- $CMD > $OUT/prof_${TAG}_${sfx}_write.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/prof_${TAG}_$sfx/pmc_sq -o bench -- $CMD > $OUT/prof_${TAG}_${sfx}_tcc.log 2>&1
  cd $ROOT; python scripts/summarize_prof.py $OUT/prof_${TAG}_$sfx > $OUT/prof_${TAG}_$sfx/summary_pmc.txt 2>&1; cd /tmp
  find $OUT/prof_${TAG}_$sfx -name "*.csv" -size +2M -delete; find $OUT/prof_${TAG}_$sfx -name "*.db" -delete
  grep -E "mlp_|wgrad" $OUT/prof_${TAG}_$sfx/summary_pmc.txt | head -40
done
