#!/bin/bash
# quick loop for the pair traceback: its tests, its in-kernel clocks, the rates that depend on it
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
TAG=${1:-r04q}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 500 python -m pytest tests/test_gpu_bits.py tests/test_gpu_parity.py tests/test_gpu_tools.py -m gpu -q -x > "$OUT/pytest.log" 2>&1
rc=$?
tail -5 "$OUT/pytest.log"
[ $rc -lt 1 ] || exit $rc
bash tools/r04/tb_timers.sh 2>&1 | tee "$OUT/timers.txt"
B="python3 $ROOT/bench.py --no-cpu-baseline --no-extra-legs"
fmt='import json,sys; d=json.loads(sys.stdin.read()); print("%s: %.0f GCUPS  %.3f ms/step  W %d  passes/launch %d streams %d  fill alone %.3f  tb alone %.3f  verified %s" % (sys.argv[1], d["value"], d["ms_per_step"], d["config"]["words_per_lane"], d["config"]["passes_per_launch"], d["config"]["launches_in_flight"], d["kernel_ms"]["fill_launch_alone"], d["kernel_ms"]["traceback_and_expand_alone"], d.get("verified")))'
$B --steps 20 --warmup 5 2>/dev/null | python3 -c "$fmt" "config4 20/5" | tee -a $OUT/summary.txt
for w in mammals primates; do
  $B --mode strong --workload $w --steps 20 --warmup 5 2>/dev/null | python3 -c "$fmt" "$w 20/5" | tee -a $OUT/summary.txt
  $B --mode strong --workload $w --steps 48 --warmup 8 2>/dev/null | python3 -c "$fmt" "$w 48/8" | tee -a $OUT/summary.txt
done
python3 tools/single_probe.py 16384 200000 2>&1 | tail -2 | tee -a $OUT/summary.txt
