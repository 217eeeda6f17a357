"""Summarise a rocprofv3 --kernel-trace CSV as a timeline: per kernel name the average duration, and the gaps
between consecutive dispatches on the same queue (start[i+1] - end[i]); plus wall time covered vs sum of durations.
    python tools/trace_timeline.py <kernel_trace.csv> [t_skip_fraction]
"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
cut = t0 + (t1 - t0) * skip
rows = [r for r in rows if int(r["Start_Timestamp"]) >= cut]
byq = defaultdict(list)
for r in rows:
    byq[r.get("Queue_Id", "0")].append(r)
print("queues:", {q: len(v) for q, v in byq.items()})
dur = defaultdict(list)
gap_after = defaultdict(list)
for q, v in byq.items():
    for a, b in zip(v, v[1:]):
        gap_after[a["Kernel_Name"][:40]].append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
    for a in v:
        dur[a["Kernel_Name"][:40]].append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
print(f"{'kernel':42s} {'calls':>7s} {'avg_us':>8s} {'p50_us':>8s} {'gap_after_avg_us':>16s} {'gap_p50':>8s}")
for k in sorted(dur, key=lambda k: -sum(dur[k])):
    d = sorted(dur[k]); g = sorted(gap_after.get(k, [0]))
    print(f"{k:42s} {len(d):7d} {sum(d)/len(d)/1e3:8.2f} {d[len(d)//2]/1e3:8.2f} {sum(g)/len(g)/1e3:16.2f} {g[len(g)//2]/1e3:8.2f}")
wall = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
busy = sum(sum(v) for v in dur.values())
print(f"wall {wall/1e6:.2f} ms, sum of kernel durations {busy/1e6:.2f} ms, ratio {busy/wall:.2f}")
