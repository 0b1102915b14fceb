#!/bin/bash
# A library variant for A/B runs (tools/ab_lib.sh): tools/build_variant.sh NAME "-DFLAG ..." [source.hip]
# compiles ONE kernel source with extra flags and links it with the shipped objects into build/libcsadp_NAME.so
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; FLAGS=$2; SRC=${3:-csadp_bits.hip}
mkdir -p $ROOT/build/var
OBJ=$ROOT/build/var/${NAME}_${SRC%.hip}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -I$ROOT/include -I$ROOT/csa_amd/csrc $FLAGS -c $ROOT/csa_amd/csrc/$SRC -o $OBJ
OTHERS=$(ls $ROOT/build/obj/*.o | grep -v "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS $OBJ -o $ROOT/build/libcsadp_$NAME.so -lpthread
echo built build/libcsadp_$NAME.so
