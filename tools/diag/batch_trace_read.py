#!/usr/bin/env python3
"""Concurrency of a kernel trace (rocprofv3 --kernel-trace csv): per queue busy time, and how long k queues were busy at once."""
import csv
import glob
import sys
from collections import defaultdict

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = []
per_q = defaultdict(float)
t_min = min(int(r["Start_Timestamp"]) for r in rows)
last = max(int(r["End_Timestamp"]) for r in rows)
cut = t_min + (last - t_min) * 0.45          # the second (timed) batch: skip the warm-up half
names = defaultdict(float)
for r in rows:
    s, e, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")
    if s < cut:
        continue
    ev.append((s, 1, q)); ev.append((e, -1, q))
    per_q[q] += (e - s) * 1e-3
    names[r["Kernel_Name"][:50]] += (e - s) * 1e-3
ev.sort()
active = defaultdict(int)
hist = defaultdict(float)
prev = ev[0][0]
for t, d, q in ev:
    k = sum(1 for v in active.values() if v > 0)
    hist[k] += (t - prev) * 1e-3
    prev = t
    active[q] += d
tot = sum(hist.values())
print("span %.0f us; queues busy at once -> share of the span: %s" % (tot, ", ".join("%d: %.0f%%" % (k, 100 * v / tot) for k, v in sorted(hist.items()))))
print("busy us per queue: %s" % ", ".join("%s: %.0f" % kv for kv in sorted(per_q.items())))
for k, v in sorted(names.items(), key=lambda kv: -kv[1])[:8]:
    print("   %-52s %.0f us" % (k, v))
