#!/bin/bash
# A/B of library builds on the one-matrix latency probe: tools/ab_single.sh build/libcsadp_X.so ...
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
cp csa_amd/libcsadp.so /tmp/libcsadp_base.so
for rep in 1 2; do
  for lib in /tmp/libcsadp_base.so "$@"; do cp $lib csa_amd/libcsadp.so; echo "$(basename $lib .so): $(python tools/single_probe.py 16384 200000 2>&1 | grep -o 'fill [0-9.]* ms  traceback+expand [0-9.]* ms' | tr '\n' ';')"; done
done
cp /tmp/libcsadp_base.so csa_amd/libcsadp.so
