#!/bin/bash
# the host's per-step scratch kept from step to step (base) against allocated per step (build/libcsadp_oldhost.so), same box, alternating
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT
cp csa_amd/libcsadp.so /tmp/libcsadp_base.so
for rep in 1 2 3; do
  for lib in base oldhost; do
    if [ $lib = base ]; then cp /tmp/libcsadp_base.so csa_amd/libcsadp.so; else cp build/libcsadp_$lib.so csa_amd/libcsadp.so; fi
    timeout -k 10 240 python tools/r05/profile_batch_probe.py 256x8x4000 512x8x4000 16x16x16000 2>&1 | grep "call 2" | cut -c1-235 | sed "s/^/$lib: /"
    timeout -k 10 120 python tools/msa_probe.py 2>&1 | grep "call 2" | sed "s/^/$lib: /"
  done
done
cp /tmp/libcsadp_base.so csa_amd/libcsadp.so
