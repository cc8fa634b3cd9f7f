#!/usr/bin/env python3
"""Overlap accounting of a rocprofv3 --kernel-trace CSV: sum of kernel durations, busy time (union of the kernel
intervals), time with >= 2 kernels in flight, and the average duration of selected kernels.
  python tools/overlap_trace.py <dir with *_kernel_trace.csv> [label]"""
import csv
import glob
import json
import re
import sys

d = sys.argv[1]
label = sys.argv[2] if len(sys.argv) > 2 else d
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# drop everything before the last long gap (model build / warm-up leave gaps): keep the steady tail = last 60 % of launches
iv = iv[int(len(iv) * 0.4):]
tot = sum(e - s for s, e, _ in iv)
ev = []
for s, e, _ in iv:
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
busy = two = 0
depth = 0
last = ev[0][0]
for t, dlt in ev:
    if depth >= 1:
        busy += t - last
    if depth >= 2:
        two += t - last
    depth += dlt
    last = t
span = iv[-1][1] - iv[0][0]
names = {}
for s, e, n in iv:
    key = re.sub(r"^void ", "", n).replace("(anonymous namespace)::", "")
    key = re.sub(r"[<(].*", "", key)[:60]
    a = names.setdefault(key, [0, 0])
    a[0] += 1
    a[1] += e - s
out = {"label": label, "launches": len(iv), "span_ms": span / 1e6, "sum_kernel_ms": tot / 1e6, "busy_ms": busy / 1e6,
       "two_or_more_in_flight_ms": two / 1e6, "idle_ms": (span - busy) / 1e6,
       "avg_us": {k: v[1] / v[0] / 1e3 for k, v in sorted(names.items(), key=lambda kv: -kv[1][1])[:14]},
       "share_ms": {k: v[1] / 1e6 for k, v in sorted(names.items(), key=lambda kv: -kv[1][1])[:14]}}
print(json.dumps(out, indent=1))
