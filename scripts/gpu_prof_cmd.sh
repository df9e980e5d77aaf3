#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel-trace stats of an arbitrary repo command.  Usage: scripts/gpu_prof_cmd.sh <tag> <command ...>
# -> gpurun_out/prof_<tag>/ + the head of its kernel_stats table (never reads stdin: a missing file is reported, not waited for)
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- "$@" > $OUT/run.log 2>&1 < /dev/null)
echo "rc=$?"
grep -v "rocprofv3\]" $OUT/run.log | tail -8
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then head -30 "$f" | cut -c1-220; else echo "no kernel_stats.csv under $OUT"; fi
