#!/bin/bash
# A diagnostic build of ONE translation unit of the library (timing experiments): scripts/diag_variant.sh <unit> <name> <flags ...>
# -> diag_build/lib_<unit>_<name>.so = the shipped objects with csrc/<unit>.hip rebuilt under the extra flags.  Run HERE (cross-compiles);
# on the GPU box: DMNERF_DIAG_LIB=diag_build/lib_<unit>_<name>.so python scripts/generic_time.py ...
set -e
UNIT=$1; NAME=$2; shift 2
cd "$(dirname "$0")/../dm_nerf_amd/csrc"
make -s > /dev/null
mkdir -p ../../diag_build build/var_diag
OBJS=$(ls build/*.o | grep -v -- "-hip-\|-host-\|build/$UNIT.o")
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize "$@" -c $UNIT.hip -o build/var_diag/${UNIT}_$NAME.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../diag_build/lib_${UNIT}_$NAME.so $OBJS build/var_diag/${UNIT}_$NAME.o
echo "diag_build/lib_${UNIT}_$NAME.so"
