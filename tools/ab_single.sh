#!/bin/bash
# A/B of library builds on single matrices inside one call: tools/ab_single.sh build/libcsadp_X.so ...
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
cp csa_amd/libcsadp.so /tmp/libcsadp_base.so
for rep in 1 2; do
  for lib in /tmp/libcsadp_base.so "$@"; do
    [ "$lib" != /tmp/libcsadp_base.so ] && cp $lib csa_amd/libcsadp.so
    python tools/single_probe.py 16384 200000 2>&1 | awk -v t=$(basename $lib .so) '{print t": "$0}' | cut -c1-130
    cp /tmp/libcsadp_base.so csa_amd/libcsadp.so
  done
done
