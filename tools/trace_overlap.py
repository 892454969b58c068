#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV of the headline loop (three contexts in flight) and reports how the kernels of
the different streams overlap: time with 0 / 1 / 2 / 3+ kernels running, per-kernel mean duration, and the gaps
between consecutive kernels of one stream (queue).

    python tools/trace_overlap.py <kernel_trace.csv>
"""
import csv
import json
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    name = r["Kernel_Name"].split("(")[0].split("::")[-1][:28]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "0"), r.get("Stream_Id", "0")))
ev = [e for e in ev if "orbx" in rows[0]["Kernel_Name"] or True]
ev.sort()
# keep the steady part: the last 60 % of the orbx kernels
orb = [e for e in ev if any(k in e[2] for k in ("pyr_", "fast_cells", "octree", "orient_desc", "sel_prefix"))]
orb = orb[int(len(orb) * 0.4):]
t0, t1 = orb[0][0], max(e[1] for e in orb)
points = []
for s, e, *_ in orb:
    points.append((s, 1))
    points.append((e, -1))
points.sort()
level, last, hist = 0, t0, defaultdict(int)
for t, d in points:
    hist[min(level, 4)] += t - last
    last = t
    level += d
total = t1 - t0
dur = defaultdict(list)
for s, e, n, *_ in orb:
    dur[n].append(e - s)
gaps = defaultdict(list)
byq = defaultdict(list)
for s, e, n, q, st in orb:
    byq[q].append((s, e, n))
for q, L in byq.items():
    L.sort()
    for a, b in zip(L, L[1:]):
        gaps[a[2] + " -> " + b[2]].append(b[0] - a[1])
out = {"span_ms": round(total / 1e6, 3), "kernels": len(orb),
       "time_share_by_kernels_running": {str(k): round(v / total, 4) for k, v in sorted(hist.items())},
       "mean_duration_us": {k: round(sum(v) / len(v) / 1e3, 2) for k, v in dur.items()},
       "count": {k: len(v) for k, v in dur.items()},
       "queues": {q: len(L) for q, L in byq.items()},
       "mean_gap_us_same_queue": {k: round(sum(v) / len(v) / 1e3, 2) for k, v in gaps.items()}}
print(json.dumps(out, indent=1))
