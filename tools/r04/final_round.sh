#!/bin/bash
# the round's closing measurements: GPU suite, the driver's bench line (CPU baseline included), the 2-rank rehearsal of the self-launch
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
TAG=${1:-r04z}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 700 python -m pytest tests -m gpu -q > "$OUT/pytest.log" 2>&1
rc=$?
tail -6 "$OUT/pytest.log"
[ $rc -lt 1 ] || exit $rc
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -20 "$OUT/bench.err"; exit 5; }
python3 - "$OUT/bench.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("value", d["value"], "ms/step", d["ms_per_step"], "verified", d.get("verified"), "frac", d["roofline"]["frac"], "from idle", d["clocks"]["from_idle_clocks"], "alone", d["roofline"]["one_launch_alone"]["avg_launch_us"])
for k in ("records", "kernel_ms", "one_shot", "streaming", "real_sets", "config5", "unrelated_16k", "single_matrix", "profile_path"):
    print(k, json.dumps(d.get(k)))
cb = d.get("cpu_baseline", {})
print("cpu", cb.get("value"), cb.get("sample"), json.dumps(cb.get("many_cores")), json.dumps(cb.get("many_cores_o3")), json.dumps(cb.get("many_cores_config3")), json.dumps(cb.get("reference_faithful")))
PY
timeout -k 10 300 python bench.py --gpus 2 --share-device --backend gloo --steps 8 --warmup 2 --no-cpu-baseline --no-extra-legs > "$OUT/bench2.json" 2> "$OUT/bench2.err" || { tail -20 "$OUT/bench2.err"; exit 9; }
python3 -c "import json,sys; d=json.load(open('$OUT/bench2.json')); print('2 ranks (gloo, one shared GPU):', d['value'], d['n_gpus'], d['verified'], d['records'])"
timeout -k 10 300 python bench.py --gpus 2 --share-device --backend gloo --mode strong --workload mammals --steps 8 --warmup 2 --no-cpu-baseline --no-extra-legs > "$OUT/bench2m.json" 2> "$OUT/bench2m.err" || { tail -20 "$OUT/bench2m.err"; exit 9; }
python3 -c "import json,sys; d=json.load(open('$OUT/bench2m.json')); print('2 ranks mammals:', d['value'], d['verified'], d['records'])"
