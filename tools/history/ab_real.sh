#!/bin/bash
# A/B of library builds on real and synthetic pair batches inside one call: tools/ab_real.sh build/libcsadp_X.so ...
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
cp csa_amd/libcsadp.so /tmp/libcsadp_base.so
for rep in 1 2; do
  for lib in /tmp/libcsadp_base.so "$@"; do
    cp $lib csa_amd/libcsadp.so 2>/dev/null
    tag=$(basename $lib .so | sed s/libcsadp_//)
    for w in primates mammals; do
      python bench.py --mode strong --workload $w --steps 10 --warmup 2 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag $w: %.0f GCUPS  %.3f ms/step  verified %s' % (d['value'], d['ms_per_step'], d.get('verified')))"
    done
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag config4: %.0f GCUPS  %.3f ms/step  tb alone %.3f  verified %s' % (d['value'], d['ms_per_step'], d['kernel_ms']['traceback_and_expand_alone'], d.get('verified')))"
  done
done
cp /tmp/libcsadp_base.so csa_amd/libcsadp.so
