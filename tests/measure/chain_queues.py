"""Which hardware queues do the chains of the train plan run on, and what does a chain's epoch cost there?

    (GPU box)  bash: for every mode, `rocprofv3 --kernel-trace --output-format csv -d DIR -o NAME -- python bench.py --steps 3 --warmup 1
               --graph-branches B ...`, then `python tests/measure/chain_queues.py DIR/.../NAME_kernel_trace.csv`

Reads a rocprofv3 kernel trace and prints, per Queue_Id: launches of the five epoch kernels, problems per launch (grid z), the
median duration of every kernel there, and the median gap between a kernel's end and the start of the next kernel ON THAT QUEUE
(the launch boundary of a dependent chain).  Two chains that share a queue show up as one queue with both launch sizes.
"""
import csv
import sys
from collections import defaultdict

import numpy as np

EPOCH = ("k_head", "k_nn_plan", "k_gradc", "k_bd", "k_l2")


def main(path):
    rows = list(csv.DictReader(open(path)))
    byq = defaultdict(list)
    for r in rows:
        name = r["Kernel_Name"]
        short = next((k for k in EPOCH if "creg::" + k in name), None)
        if short is None:
            continue
        byq[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, int(r.get("Grid_Size_Z", r.get("Grid_Size", 0)) or 0),
                                   int(r.get("Workgroup_Size_Z", 1) or 1), r.get("Stream_Id", "?")))
    t0 = min(v[0][0] for v in byq.values() if v)
    for q, ev in sorted(byq.items()):
        ev.sort()
        dur = defaultdict(list)
        gaps = []
        for a, b in zip(ev, ev[1:]):
            gaps.append(b[0] - a[1])
        for e in ev:
            dur[e[2]].append(e[1] - e[0])
        z = sorted({e[3] // max(e[4], 1) for e in ev})
        streams = sorted({e[5] for e in ev})
        span = (ev[-1][1] - ev[0][0]) / 1e3
        busy = sum(e[1] - e[0] for e in ev) / 1e3
        print(f"queue {q}: streams {streams}  {len(ev)} epoch launches, problems per launch {z}, first at {(ev[0][0] - t0) / 1e3:.0f} us, span {span:.0f} us, "
              f"kernels busy {busy / span:.2f} of it; median us: " + "  ".join(f"{k} {np.median(dur[k]) / 1e3:.2f}" for k in EPOCH if dur[k])
              + f";  gap to the next launch on the queue: median {np.median(gaps) / 1e3:.2f}, p90 {np.percentile(gaps, 90) / 1e3:.2f} us")
    # per epoch cost of a chain: k_head to the next k_head on the same queue with the same launch size
    for q, ev in sorted(byq.items()):
        for zz in sorted({e[3] // max(e[4], 1) for e in ev}):
            heads = [e[0] for e in ev if e[2] == "k_head" and e[3] // max(e[4], 1) == zz]
            d = np.diff(heads) / 1e3
            d = d[d < 500]
            if len(d):
                print(f"  queue {q}, {zz} problems per launch: epoch period median {np.median(d):.1f} us (p10 {np.percentile(d, 10):.1f}, p90 {np.percentile(d, 90):.1f})")


if __name__ == "__main__":
    main(sys.argv[1])
