#!/bin/bash
# A/B of the granule prefetch distance of nw_fill_cells (2 blocks shipped vs 1), then the strip timers of the 1-block build
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
OUT=gpurun_out/${1:-r04x}; mkdir -p $OUT
bash tools/ab_cells.sh build/libcsadp_ahead1.so 2>&1 | tee $OUT/ab.txt
cp csa_amd/libcsadp.so /tmp/libcsadp_base.so
cp build/libcsadp_ahead1t.so csa_amd/libcsadp.so
timeout -k 10 200 python tools/r04/cells_times.py 2>&1 | tee $OUT/cells_times_ahead1.txt
cp /tmp/libcsadp_base.so csa_amd/libcsadp.so
