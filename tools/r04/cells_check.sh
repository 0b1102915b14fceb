#!/bin/bash
# after a change to nw_fill_cells: its parity tests (cells mode of the pair tests, N-sequence families, the example sets' md5s), the step probe, mode N timings
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
TAG=${1:-r04u}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 240 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > "$OUT/pytest1.log" 2>&1
rc=$?
tail -4 "$OUT/pytest1.log"
[ $rc -lt 1 ] || exit $rc
timeout -k 10 400 python -m pytest tests/test_gpu_msa.py tests/test_gpu_dropin.py tests/test_gpu_bits.py -m gpu -q -x > "$OUT/pytest2.log" 2>&1
rc=$?
tail -4 "$OUT/pytest2.log"
[ $rc -lt 1 ] || exit $rc
timeout -k 10 200 python tools/cells_probe.py 2>&1 | tail -10 | tee "$OUT/cells_probe.txt"
for i in 1 2; do timeout -k 10 300 python tools/msa_probe.py 2>&1 | grep "call 2"; done | tee "$OUT/msa_probe.log"
bash tools/tb_cells_trace.sh Set3 > "$OUT/tbtrace_Set3.txt" 2>&1
grep "fill_cells" "$OUT/tbtrace_Set3.txt" | sort -k2 -n | tail -5
