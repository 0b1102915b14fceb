#!/bin/bash
# the fetcher layout of nw_fill_cells: parity first, then the step probe with and without it, the example sets, the strip timers
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
TAG=${1:-r04y}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > "$OUT/pytest1.log" 2>&1
rc=$?
tail -4 "$OUT/pytest1.log"
[ $rc -lt 1 ] || exit $rc
for f in 0 256; do
  CSADP_CELLS_FETCH=$f timeout -k 10 200 python tools/cells_probe.py 2>&1 | awk -v t=fetch$f '{print t": "$0}' | cut -c1-130 | tee -a "$OUT/cells_probe.txt"
  for i in 1 2; do CSADP_CELLS_FETCH=$f timeout -k 10 300 python tools/msa_probe.py 2>&1 | grep "call 2" | awk -v t=fetch$f '{print t": "$0}'; done | tee -a "$OUT/msa_probe.log"
done
if [ -f build/libcsadp_celltimers.so ]; then
  cp csa_amd/libcsadp.so /tmp/libcsadp_base.so
  cp build/libcsadp_celltimers.so csa_amd/libcsadp.so
  timeout -k 10 200 python tools/r04/cells_times.py 2>&1 | tee $OUT/cells_times.txt
  cp /tmp/libcsadp_base.so csa_amd/libcsadp.so
fi
