#!/bin/bash
# N-sequence batches: round groups 1 / 2 / 3 / 4 / 6 with the level-by-level work list
cd ${GRAFT_REPO_ROOT:-.}
for g in 1 2 3 4 6; do
  CSADP_ROUND_GROUPS=$g timeout -k 10 200 python tools/r05/profile_batch_probe.py 256x8x4000 512x8x4000 16x16x16000 64x16x16000 64x4x30000 2>&1 | grep "call [12]" | cut -c1-190 | sed "s/^/groups=$g: /"
done
