#!/bin/bash
# A/B in one call: the traceback's walking wave with a SIMD to itself (build/libcsadp_walker.so: -DCSADP_TB_WALKER_ALONE) against the shipped library
cd ${GRAFT_REPO_ROOT:-.}
cp csa_amd/libcsadp.so /tmp/libcsadp_base.so
run() {
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
s=d['single_matrix']
print('$TAG: value %.1f  tb alone %.3f ms | one_shot %.3f ms (fill %.3f tb %.3f) = %.1f TCUPS | 200 kbp %.2f + %.2f ms | real %.1f %.1f | unrelated %.1f | config5 %.1f | c4all %.1f | verified %s %d' % (
  d['value']/1e3, d['kernel_ms']['traceback_and_expand_alone'], d['one_shot']['device_ms'], d['one_shot']['fill_ms'], d['one_shot']['traceback_expand_ms'], d['one_shot']['gcups_device']/1e3,
  s['200000']['fill_ms'], s['200000']['traceback_expand_ms'], d['real_sets']['Mammals']['gcups']/1e3, d['real_sets']['Primates']['gcups']/1e3, d['unrelated_16k']['gcups']/1e3,
  d['config5']['gcups']/1e3, d['config4_all']['gcups']/1e3, d['verified'], d['records']['checked_against_reference_digests_in_all_legs']))"
}
for rep in 1 2; do
  TAG=base; cp /tmp/libcsadp_base.so csa_amd/libcsadp.so; run
  TAG=walker; cp build/libcsadp_walker.so csa_amd/libcsadp.so; run
done
cp /tmp/libcsadp_base.so csa_amd/libcsadp.so
