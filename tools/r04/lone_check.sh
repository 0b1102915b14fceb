#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
TAG=${1:-r04m}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 500 python -m pytest tests/test_gpu_bits.py tests/test_gpu_tools.py -m gpu -q -x > "$OUT/pytest.log" 2>&1
rc=$?
tail -12 "$OUT/pytest.log"
[ $rc -lt 1 ] || exit $rc
timeout -k 10 500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -20 "$OUT/bench.err"; exit 5; }
python3 - "$OUT/bench.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("value", d["value"], "ms/step", d["ms_per_step"], "verified", d.get("verified"))
for k in ("kernel_ms", "one_shot", "streaming", "real_sets", "config5", "unrelated_16k", "single_matrix", "profile_path"):
    print(k, json.dumps(d.get(k)))
PY
