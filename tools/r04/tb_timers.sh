#!/bin/bash
# in-kernel clocks of the windowed pair traceback (build/libcsadp_tbtimers.so = tools/build_variant.sh tbtimers -DCSADP_TB_TIMERS): a synthetic
# 16 kbp pair, and the first pairs of the Mammals set at three words per lane
cd $GRAFT_REPO_ROOT
cp csa_amd/libcsadp.so /tmp/base.so; cp build/libcsadp_tbtimers.so csa_amd/libcsadp.so
python tools/single_probe.py 16384 2>&1 | grep -E "timers|fill" | sort | uniq -c | sort -rn | head -6
CSADP_BITS_WORDS=2 python tools/single_probe.py 16384 2>&1 | grep -E "timers|fill" | sort | uniq -c | sort -rn | head -6
python bench.py --no-cpu-baseline --no-extra-legs --mode strong --workload mammals --steps 2 --warmup 1 2>&1 | grep -E "timers" | sort | uniq -c | sort -rn | head -8
cp /tmp/base.so csa_amd/libcsadp.so
