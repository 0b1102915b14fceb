#!/bin/bash
# nw_fill_cells work list: job by job (0) against chunk level by chunk level (1), and the default rule (by launch size)
cd ${GRAFT_REPO_ROOT:-.}
for o in 0 1 default; do
  if [ $o = default ]; then unset CSADP_CELLS_ORDER; else export CSADP_CELLS_ORDER=$o; fi
  timeout -k 10 240 python tools/r05/profile_batch_probe.py 256x8x4000 512x8x4000 16x16x16000 64x16x16000 64x4x30000 2>&1 | grep "call [12]" | sed "s/^/order=$o: /"
done
unset CSADP_CELLS_ORDER
for o in 0 1; do
  CSADP_CELLS_ORDER=$o timeout -k 10 120 python tools/msa_probe.py 2>&1 | grep "call 2" | sed "s/^/order=$o: /"
done
