#!/bin/bash
# A/B in one call: nw_pack_planes / nw_expand_rows at wave priority 3 (build/libcsadp_ioprio.so) against the shipped library
cd ${GRAFT_REPO_ROOT:-.}
cp csa_amd/libcsadp.so /tmp/libcsadp_base.so
run() {
  for wl in mammals primates; do
  python bench.py --mode strong --workload $wl --steps 48 --warmup 8 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$TAG $wl: %.1f TCUPS  %.3f ms/step verified %s' % (d['value']/1e3, d['ms_per_step'], d['verified']))"
  done
  python bench.py --steps 48 --warmup 8 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$TAG config4: %.1f TCUPS  %.3f ms/step verified %s' % (d['value']/1e3, d['ms_per_step'], d['verified']))"
}
for rep in 1 2; do
  TAG=base; cp /tmp/libcsadp_base.so csa_amd/libcsadp.so; run
  TAG=ioprio; cp build/libcsadp_ioprio.so csa_amd/libcsadp.so; run
done
cp /tmp/libcsadp_base.so csa_amd/libcsadp.so
