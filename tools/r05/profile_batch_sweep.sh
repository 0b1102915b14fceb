#!/bin/bash
# N-sequence batches: round groups x host threads, helper-wave layout forced on / off
cd ${GRAFT_REPO_ROOT:-.}
for g in 2 4 8; do
  echo "== CSADP_ROUND_GROUPS=$g"
  CSADP_ROUND_GROUPS=$g python tools/r05/profile_batch_probe.py 512x8x4000 16x16x16000 2>&1 | grep "call [12]"
done
echo "== layouts (2 groups): CSADP_CELLS_FETCH=0 (plain: two workgroups of four waves per compute unit) / 100000 (helper waves: one of six)"
for f in 0 100000; do
  CSADP_CELLS_FETCH=$f python tools/r05/profile_batch_probe.py 512x8x4000 16x16x16000 2>&1 | grep "call 2" | sed "s/^/fetch=$f: /"
done
