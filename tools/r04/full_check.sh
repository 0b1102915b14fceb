#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
TAG=${1:-r04n}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 700 python -m pytest tests -m gpu -q -x > "$OUT/pytest.log" 2>&1
rc=$?
tail -6 "$OUT/pytest.log"
[ $rc -lt 1 ] || exit $rc
for i in 1 2; do timeout -k 10 300 python tools/msa_probe.py 2>&1 | grep "call 2"; done | tee "$OUT/msa_probe.log"
CSADP_TRACE_HOST=1 timeout -k 10 300 python tools/msa_probe.py Primates 2>&1 | grep "csadp round" | tail -16 | tee -a "$OUT/msa_probe.log"
