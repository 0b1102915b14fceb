#!/bin/bash
# builds and runs tools/r05/refine_bench.cpp against the product's host sources (CPU only; device objects as built); PG=1 adds -pg and prints gprof's flat profile
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
SRC=$ROOT/csa_amd/csrc
HOST="csadp_api.cpp csadp_progressive.cpp csadp_hostutil.cpp csadp_rotations.cpp csadp_anchors.cpp csadp_msa.cpp csadp_engine.cpp"
DEV="$ROOT/build/obj/csadp_kernels.o $ROOT/build/obj/csadp_bits.o $ROOT/build/obj/csadp_pairio.o $ROOT/build/obj/csadp_cells.o $ROOT/build/obj/csadp_cells_tb.o"
FLAGS="-O2 -g -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I$ROOT/include -I$SRC -w"
[ -n "$PG" ] && FLAGS="$FLAGS -pg -fno-inline-functions"
mkdir -p $ROOT/build/var
gcc -O2 -c $ROOT/oracle/csa_dp_oracle.c -o $ROOT/build/var/oracle_bench.o
g++ $FLAGS $(for f in $HOST; do echo $SRC/$f; done) $ROOT/tools/r05/refine_bench.cpp $ROOT/build/var/oracle_bench.o $DEV -L/opt/rocm/lib -lamdhip64 -lpthread -Wl,-rpath,/opt/rocm/lib -o $ROOT/build/var/refine_bench || exit 1
cd $ROOT/build/var && CSADP_HOST_THREADS=1 ./refine_bench "$@"
[ -n "$PG" ] && gprof -b -p ./refine_bench gmon.out 2>/dev/null | head -25
