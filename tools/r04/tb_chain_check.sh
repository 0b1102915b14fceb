#!/bin/bash
# after a change to the band-parallel walk: its parity tests, the example sets' md5s, per-launch durations of resolve / emit on the three sets
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
TAG=${1:-r05u}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_msa.py tests/test_gpu_dropin.py -m gpu -q -x > "$OUT/pytest.log" 2>&1
rc=$?
tail -4 "$OUT/pytest.log"
[ $rc -lt 1 ] || exit $rc
for i in 1 2; do timeout -k 10 300 python tools/msa_probe.py 2>&1 | grep "call 2"; done | tee "$OUT/msa_probe.log"
for s in Mammals Primates Set3; do
  bash tools/tb_cells_trace.sh $s > "$OUT/tbtrace_$s.txt" 2>&1
  python3 - "$ROOT/gpurun_out/tbtrace_$s/launches.txt" $s <<'PY'
import sys, collections
tot = collections.defaultdict(float); mx = collections.defaultdict(float); cnt = collections.Counter()
for line in open(sys.argv[1]):
    p = line.split()
    name = p[0].split("::")[-1].split("<")[0]
    us = float([x for x in p if x.replace(".", "", 1).isdigit()][0])
    tot[name] += us; mx[name] = max(mx[name], us); cnt[name] += 1
print(sys.argv[2], " ".join("%s %d x, %.0f us in all, max %.0f" % (k, cnt[k], tot[k], mx[k]) for k in sorted(tot) if k.startswith("nw_")))
PY
done | tee "$OUT/tb_kernels.txt"
