#!/bin/bash
# A/B of library builds on csa_msa's three example sets inside one call: tools/ab_msa.sh build/libcsadp_X.so ...
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
cp csa_amd/libcsadp.so /tmp/libcsadp_base.so
for rep in 1 2 3; do
  for lib in /tmp/libcsadp_base.so "$@"; do
    cp $lib csa_amd/libcsadp.so 2>/dev/null
    python tools/msa_probe.py 2>&1 | grep "call [12]" | awk -v t=$(basename $lib .so | sed s/libcsadp_//) '{print t": "$0}' | cut -c1-110
  done
done
cp /tmp/libcsadp_base.so csa_amd/libcsadp.so
