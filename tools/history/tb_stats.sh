#!/bin/bash
# resolve statistics of the band-parallel traceback on one example set (needs build/libcsadp_tbstats.so: tools/build_variant.sh tbstats -DCSADP_TB_STATS csadp_cells_tb.hip)
cd $GRAFT_REPO_ROOT
cp csa_amd/libcsadp.so /tmp/base.so; cp build/libcsadp_tbstats.so csa_amd/libcsadp.so
python tools/msa_probe.py ${1:-Set3} 2>&1 | grep -E "resolve|scout" | tail -${2:-20}
cp /tmp/base.so csa_amd/libcsadp.so
