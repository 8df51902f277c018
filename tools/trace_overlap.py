#!/usr/bin/env python
"""How much do the kernels of a rocprofv3 --kernel-trace overlap?  Reads *_kernel_trace.csv, keeps the last `--tail-ms` of the
trace (the timed steps), prints per queue: kernels, summed duration; and overall: wall span, union of busy intervals, sum of
durations (sum / union = average concurrency)."""
import csv
import glob
import sys

paths = [p for a in sys.argv[1:] if not a.startswith("--") for p in glob.glob(a, recursive=True)]
tail_ms = float([a.split("=")[1] for a in sys.argv if a.startswith("--tail-ms=")][0]) if any(a.startswith("--tail-ms=") for a in sys.argv) else 400.0
rows = []
for p in paths:
    with open(p) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"][:40]))
rows.sort()
t_end = max(r[1] for r in rows)
rows = [r for r in rows if r[0] >= t_end - tail_ms * 1e6]
span = (max(r[1] for r in rows) - rows[0][0]) / 1e6
union, cur_s, cur_e = 0, None, None
for s, e, _, _ in rows:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
tot = sum(e - s for s, e, _, _ in rows)
print("kernels %d  span %.2f ms  busy(union) %.2f ms  sum of durations %.2f ms  avg concurrency %.2f" % (len(rows), span, union / 1e6, tot / 1e6, tot / union))
byq = {}
for s, e, q, _ in rows:
    a = byq.setdefault(q, [0, 0])
    a[0] += 1
    a[1] += e - s
for q, (n, d) in sorted(byq.items()):
    print("  queue %s: %d kernels, %.2f ms" % (q, n, d / 1e6))
