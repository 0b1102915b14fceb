#!/bin/bash
# A/B of library builds on the profile-step kernels inside one call: tools/ab_cells.sh build/libcsadp_X.so ...
# (the shipped library is "base"): tools/cells_probe.py shapes, then csa_msa's three example sets
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
cp csa_amd/libcsadp.so /tmp/libcsadp_base.so
run() {
  python tools/cells_probe.py 2>&1 | awk -v t=$TAG '{print t": "$0}' | cut -c1-120
  python tools/msa_probe.py 2>&1 | grep "call 2" | awk -v t=$TAG '{print t": "$0}'
}
for rep in 1 2; do
  TAG=base; cp /tmp/libcsadp_base.so csa_amd/libcsadp.so; run
  for lib in "$@"; do TAG=$(basename $lib .so | sed s/libcsadp_//); cp $lib csa_amd/libcsadp.so; run; done
done
cp /tmp/libcsadp_base.so csa_amd/libcsadp.so
