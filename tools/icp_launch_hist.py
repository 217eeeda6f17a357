"""k_icp_nn launches of a rocprofv3 --kernel-trace CSV: duration and the gap to the next kernel of the queue, by grid size (live chunks).
    python tools/icp_launch_hist.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
bins = defaultdict(lambda: [0, 0.0, 0.0])
edges = (16, 64, 256, 1024, 4096, 1 << 30)
for i, r in enumerate(rows):
    if "k_icp_nn" not in r["Kernel_Name"]:
        continue
    wg = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1) if "Grid_Size_X" in r else int(r["Grid_Size"]) // 256
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    gap = (int(rows[i + 1]["Start_Timestamp"]) - int(r["End_Timestamp"])) / 1e3 if i + 1 < len(rows) else 0.0
    b = next(e for e in edges if wg <= e)
    bins[b][0] += 1; bins[b][1] += dur; bins[b][2] += max(gap, 0.0)
print("workgroups <=   launches   mean duration us   mean gap to the next kernel us   total ms")
for e in edges:
    n, d, g = bins[e]
    if n:
        print(f"{e if e < (1 << 30) else 'more':>13}   {n:8d}   {d / n:16.1f}   {g / n:30.1f}   {(d + g) / 1e3:8.2f}")
