#!/usr/bin/env python3
"""Concurrency of a kernel trace (rocprofv3 --kernel-trace CSV): per kernel name the launches, mean duration, and over the busiest stretch how many
kernels of each kind were running at once.  usage: timeline.py <kernel_trace.csv>"""
import csv
import sys
import collections

rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
names = {}
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void csadp::", "").replace("csadp::", "")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    names.setdefault(n, []).append((s, e))
    ev.append((s, 1, n))
    ev.append((e, -1, n))
for n, v in sorted(names.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
    print("%-50s %5d launches, mean %8.1f us, total %9.1f ms" % (n[:50], len(v), sum(e - s for s, e in v) / len(v) / 1e3, sum(e - s for s, e in v) / 1e6))
# the timed region = the last 60 % of the trace's span (warm-up and set-up in front)
t0 = min(s for s, _, _ in ev)
t1 = max(s for s, _, _ in ev)
lo = t0 + int(0.4 * (t1 - t0))
ev.sort()
cur = collections.Counter()
acc = collections.Counter()
last = None
for t, d, n in ev:
    if last is not None and t > lo:
        key = tuple(sorted((k.split("<")[0], v) for k, v in cur.items() if v))
        acc[key] += t - max(last, lo)
    cur[n] += d
    last = t
tot = sum(acc.values())
print("concurrency over the last 60 %% of the trace (%.1f ms):" % (tot / 1e6))
for key, v in acc.most_common(12):
    print("  %5.1f %%  %s" % (100.0 * v / tot, ", ".join("%s x%d" % kv for kv in key) or "idle"))
