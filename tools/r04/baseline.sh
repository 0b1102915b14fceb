#!/bin/bash
# Round 4: where the tree stands -- GPU suite, the driver's bench line, the real sets at 20/5 and 48/8, kernel traces of the real sets.
#   tools/r04/baseline.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
TAG=${1:-r04a}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 600 python -m pytest tests -m gpu -q -x > "$OUT/pytest.log" 2>&1
rc=$?
tail -5 "$OUT/pytest.log"
[ $rc -lt 2 ] || exit $rc
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -20 "$OUT/bench.err"; exit 5; }
python3 - "$OUT/bench.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("value", d["value"], "ms/step", d["ms_per_step"], "verified", d.get("verified"))
for k in ("kernel_ms", "one_shot", "real_sets", "config5", "unrelated_16k", "single_matrix", "profile_path", "gcups_8d_h2d_d2h_inclusive"):
    print(k, json.dumps(d.get(k)))
print("cpu", json.dumps(d.get("cpu_baseline", {}).get("many_cores")), json.dumps(d.get("cpu_baseline", {}).get("many_cores_o3")))
PY
B="python3 $ROOT/bench.py --no-cpu-baseline --no-extra-legs"
fmt='import json,sys; d=json.loads(sys.stdin.read()); print("%s: %.0f GCUPS  %.3f ms/step  W %d  passes/launch %d streams %d  fill alone %.3f  tb alone %.3f  verified %s" % (sys.argv[1], d["value"], d["ms_per_step"], d["config"]["words_per_lane"], d["config"]["passes_per_launch"], d["config"]["launches_in_flight"], d["kernel_ms"]["fill_launch_alone"], d["kernel_ms"]["traceback_and_expand_alone"], d.get("verified")))'
for w in mammals primates; do
  $B --mode strong --workload $w --steps 20 --warmup 5 2>/dev/null | python3 -c "$fmt" "$w 20/5" | tee -a $OUT/summary.txt
  $B --mode strong --workload $w --steps 48 --warmup 8 2>/dev/null | python3 -c "$fmt" "$w 48/8" | tee -a $OUT/summary.txt
done
cd /tmp && export TMPDIR=/tmp
for w in mammals; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o ${w}_stats -- $B --mode strong --workload $w --steps 20 --warmup 5 > "$OUT/log_${w}_stats.txt" 2>&1
  CSADP_BITS_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o ${w}_solo -- $B --mode strong --workload $w --steps 12 --warmup 3 > "$OUT/log_${w}_solo.txt" 2>&1
done
find "$OUT" -name "*kernel_stats.csv" | xargs -r -n1 head -8
