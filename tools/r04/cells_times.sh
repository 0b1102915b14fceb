#!/bin/bash
# strip timers of nw_fill_cells (probe build; the shipped library is put back afterwards)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
mkdir -p gpurun_out/${1:-r04w}
cp csa_amd/libcsadp.so /tmp/libcsadp_base.so
cp build/libcsadp_celltimers.so csa_amd/libcsadp.so
timeout -k 10 200 python tools/r04/cells_times.py 2>&1 | tee gpurun_out/${1:-r04w}/cells_times.txt
cp /tmp/libcsadp_base.so csa_amd/libcsadp.so
