#!/bin/bash
# Compile every hand-scheduled kernel TU to ISA and run the asm-read hazard check on it.
# Usage: scripts/scan_kernels.sh [extra hipcc flags]   (e.g. -DDMN_TILE_RC='"a"')
set -u
cd "$(dirname "$0")/../dm_nerf_amd/csrc"
out=${SCAN_OUT:-/tmp/dmn_scan}; mkdir -p "$out"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function"
tus=${SCAN_TUS:-"mlp_fwd mlp_fwd_embedded mlp_fwd_train mlp_fwd_fused mlp_split mlp_bwd wgrad"}
for t in $tus; do
  hipcc $FLAGS "$@" -S --cuda-device-only $t.hip -o $out/$t.s 2>$out/$t.err &
done
wait
rc=0
for t in $tus; do
  printf "%-18s " $t
  python ../../scripts/check_asm_hazard.py $out/$t.s | tail -1
  [ ${PIPESTATUS[0]} -ne 0 ] && rc=1
  grep -E "vgpr_spill_count|private_segment_fixed_size" $out/$t.s | awk '{printf "%s ", $2} END {print ""}'
done
exit $rc
