#!/bin/bash
# Diagnostic builds of csrc/gemm_chain.hip (timing only -- their results are wrong on purpose): the chain kernel without its per-chunk
# barrier / weight requests / per-layer epilogue, linked against the shipped objects into diag_build/lib_chain_<name>.so.
# Run HERE (cross-compiles); then on the GPU box: DMNERF_DIAG_LIB=diag_build/lib_chain_<name>.so python scripts/generic_time.py 6x128 --render-only
set -e
cd "$(dirname "$0")/../dm_nerf_amd/csrc"
make -s > /dev/null
mkdir -p ../../diag_build build/var_chain
OBJS=$(ls build/*.o | grep -v -- "-hip-\|-host-\|build/gemm_chain.o")
for v in "$@"; do
  flags=""; for f in ${v//+/ }; do flags="$flags -DDMN_CH_$f"; done
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize $flags -c gemm_chain.hip -o build/var_chain/gemm_chain_$v.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../../diag_build/lib_chain_$v.so $OBJS build/var_chain/gemm_chain_$v.o
  echo "diag_build/lib_chain_$v.so"
done
