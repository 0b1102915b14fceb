#!/bin/bash
# A/B of granule_wait's vmcnt (plain layout of nw_fill_cells: CSADP_CELLS_FETCH=0) in one call: base = shipped (keeps 3 in flight), old = round 4's (2)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT
cp csa_amd/libcsadp.so /tmp/libcsadp_base.so
run() {
  CSADP_CELLS_FETCH=0 python tools/r04/fetch_threshold_probe.py child 2>&1 | awk -v t=$TAG '{print t" plain: "$0}'
  CSADP_CELLS_FETCH=0 python tools/cells_probe.py 2>&1 | tail -3 | awk -v t=$TAG '{print t" plain: "$0}' | cut -c1-150
}
for rep in 1 2; do
  TAG=keep3; cp /tmp/libcsadp_base.so csa_amd/libcsadp.so; run
  TAG=keep2; cp build/libcsadp_granule_old.so csa_amd/libcsadp.so; run
done
cp /tmp/libcsadp_base.so csa_amd/libcsadp.so
