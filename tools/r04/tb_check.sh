#!/bin/bash
# after a change to the pair traceback: the GPU suite, then the rates of the default batch and the real sets at 20/5 and 48/8
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
TAG=${1:-r04d}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 700 python -m pytest tests -m gpu -q -x ${PYTEST_ARGS} > "$OUT/pytest.log" 2>&1
rc=$?
tail -15 "$OUT/pytest.log"
[ $rc -lt 1 ] || exit $rc
B="python3 $ROOT/bench.py --no-cpu-baseline --no-extra-legs"
fmt='import json,sys; d=json.loads(sys.stdin.read()); print("%s: %.0f GCUPS  %.3f ms/step  W %d  passes/launch %d streams %d  fill alone %.3f  tb alone %.3f  verified %s" % (sys.argv[1], d["value"], d["ms_per_step"], d["config"]["words_per_lane"], d["config"]["passes_per_launch"], d["config"]["launches_in_flight"], d["kernel_ms"]["fill_launch_alone"], d["kernel_ms"]["traceback_and_expand_alone"], d.get("verified")))'
$B --steps 20 --warmup 5 2>/dev/null | python3 -c "$fmt" "config4 20/5" | tee -a $OUT/summary.txt
$B --steps 48 --warmup 8 2>/dev/null | python3 -c "$fmt" "config4 48/8" | tee -a $OUT/summary.txt
for w in mammals primates; do
  $B --mode strong --workload $w --steps 20 --warmup 5 2>/dev/null | python3 -c "$fmt" "$w 20/5" | tee -a $OUT/summary.txt
  $B --mode strong --workload $w --steps 48 --warmup 8 2>/dev/null | python3 -c "$fmt" "$w 48/8" | tee -a $OUT/summary.txt
done
python3 tools/single_probe.py 2>&1 | tail -8 | tee -a $OUT/summary.txt
