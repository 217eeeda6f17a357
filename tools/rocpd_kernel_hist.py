"""Per-kernel duration histogram and launch-to-launch gaps from a rocprofv3 rocpd database (results.db).

    python tools/rocpd_kernel_hist.py gpurun_out/prof_km/km_results.db k_km_assign
"""
import sqlite3
import sys

import numpy as np


def main():
    db, pat = sys.argv[1], sys.argv[2]
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = list(c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
    names = sorted({r[0] for r in rows if pat in r[0]})
    for nm in names:
        sel = np.array([(r[1], r[2]) for r in rows if r[0] == nm], dtype=np.int64)
        dur = (sel[:, 1] - sel[:, 0]) / 1e3
        gap = (sel[1:, 0] - sel[:-1, 1]) / 1e3
        print(nm[:90])
        print(f"  calls {len(dur)}  duration us: mean {dur.mean():.1f}  p10 {np.percentile(dur, 10):.1f}  p50 {np.percentile(dur, 50):.1f}  "
              f"p90 {np.percentile(dur, 90):.1f}  p99 {np.percentile(dur, 99):.1f}  max {dur.max():.1f}")
        g = gap[gap < 1000] if len(gap) else gap
        if len(g):
            print(f"  gap to the next launch of the same kernel (us, gaps < 1 ms): mean {g.mean():.1f}  p50 {np.percentile(g, 50):.1f}  p90 {np.percentile(g, 90):.1f}")
        big = np.argsort(-dur)[:8]
        print("  longest:", ", ".join(f"#{i}: {dur[i]:.0f}" for i in sorted(big)))


if __name__ == "__main__":
    main()
