#!/bin/bash
# nw_fill_cells at 150 registers (three waves per SIMD; the shipped library) against build/libcsadp_regs222.so (the same code at 222: two waves)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT
cp csa_amd/libcsadp.so /tmp/libcsadp_base.so
for rep in 1 2; do
  for lib in base ${VARIANT:-regs222}; do
    if [ $lib = base ]; then cp /tmp/libcsadp_base.so csa_amd/libcsadp.so; else cp build/libcsadp_$lib.so csa_amd/libcsadp.so; fi
    timeout -k 10 240 python tools/r05/profile_batch_probe.py 256x8x4000 512x8x4000 16x16x16000 64x16x16000 64x4x30000 2>&1 | grep "call 2" | cut -c1-150 | sed "s/^/$lib: /"
    timeout -k 10 120 python tools/msa_probe.py 2>&1 | grep "call 2" | sed "s/^/$lib: /"
    timeout -k 10 120 python tools/single_probe.py 2>&1 | tail -6 | sed "s/^/$lib: /"
  done
done
cp /tmp/libcsadp_base.so csa_amd/libcsadp.so
