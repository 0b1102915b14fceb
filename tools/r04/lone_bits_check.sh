#!/bin/bash
# after a change to the one-wave-per-SIMD path of nw_fill_bits: its tests, the single matrices on the bit-parallel path, the one-shot leg
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
TAG=${1:-r05o}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 500 python -m pytest tests/test_gpu_bits.py tests/test_gpu_parity.py -m gpu -q -x > "$OUT/pytest.log" 2>&1
rc=$?
tail -4 "$OUT/pytest.log"
[ $rc -lt 1 ] || exit $rc
CSADP_LONE_CELLS=0 timeout -k 10 300 python tools/single_probe.py 2>&1 | tee "$OUT/single_bits.txt"
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -20 "$OUT/bench.err"; exit 5; }
python3 - "$OUT/bench.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("value", d["value"], "ms/step", d["ms_per_step"], "verified", d.get("verified"))
for k in ("kernel_ms", "one_shot", "single_matrix", "real_sets", "profile_path"):
    print(k, json.dumps(d.get(k))[:700])
PY
