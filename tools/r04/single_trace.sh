#!/bin/bash
# host stages of one large pair through csadp_align_batch, both routes
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/${1:-r05h}
( CSADP_TRACE_HOST=1 timeout -k 10 200 python tools/single_probe.py 16384 2>&1 | tail -30
  echo ---- CSADP_LONE_CELLS=0
  CSADP_LONE_CELLS=0 timeout -k 10 200 python tools/single_probe.py 16384 100000 2>&1 | tail -4 ) | tee gpurun_out/${1:-r05h}/single_trace.txt
