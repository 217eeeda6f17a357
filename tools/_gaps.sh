cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/gaps -o g -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-icp-variant > $GRAFT_REPO_ROOT/gpurun_out/gaps.log 2>&1
python - <<PY
import csv, glob, os
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/gaps"
kt = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(kt)))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]) for r in rows]
mc = glob.glob(root + "/**/*memory_copy_trace.csv", recursive=True)
if mc:
    for r in csv.DictReader(open(mc[0])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
ev.sort()
# take the last 40% of the run (timed region), find busy-union and idle gaps
t0 = ev[int(len(ev) * 0.6)][0]
ev = [e for e in ev if e[0] >= t0]
busy_end = ev[0][0]; idle = 0; gaps = []
for s, e, n in ev:
    if s > busy_end:
        idle += s - busy_end; gaps.append((s - busy_end, n))
    busy_end = max(busy_end, e)
span = busy_end - ev[0][0]
print("span ms %.2f idle ms %.2f (%.1f%%)" % (span / 1e6, idle / 1e6, 100.0 * idle / span))
gaps.sort(reverse=True)
print("largest gaps (us, next event):", [(round(g / 1e3, 1), n) for g, n in gaps[:12]])
from collections import Counter
c = Counter(); d = Counter()
for s, e, n in ev: c[n] += 1; d[n] += e - s
for n, t in d.most_common(14): print("  %-42s n=%6d total ms %8.2f avg us %7.2f" % (n, c[n], t / 1e6, t / c[n] / 1e3))
PY
