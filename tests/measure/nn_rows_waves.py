"""Where does a launch of the sixteen-queries-per-wave search (k_nn_rows) spend its time?  (the franka shape's dominant
kernel shows a VALU instruction active in 0.30 of the chip's SIMD-cycles.)  Measurement build:

    CREG_EXTRA_FLAGS=-DCREG_NN_WAVE_STAMPS python -m autourdf_amd.build --variant wstamp       # on the GPU box (hipcc is there; .gpurunignore keeps variant libraries out of the snapshot)
    CREG_LIB_VARIANT=wstamp python tests/measure/nn_rows_waves.py [franka|wx200_5]             # GPU box

Every wave records start / box bounds done / visits done / end on the 100 MHz wall clock, with its visit and candidate counts; the
launches of 100 mid-train epochs are taken apart: span of a launch, dispatch ramp (first to last wave start), wave lifetime and its
phases (mean, percentiles), how many waves are alive over the launch, and what the LAST waves of a launch look like."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import numpy as np                                                       # noqa: E402
import torch                                                             # noqa: E402
from autourdf_amd import _lib                                            # noqa: E402
from autourdf_amd.engine import BatchRegistrar                           # noqa: E402
from autourdf_amd.synthetic import initial_segmentation, make_sequence   # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "franka"
robot, N, K = {"franka": ("franka", 16384, 40), "wx200_5": ("wx200_5", 4096, 20)}[wl]
if wl == "wx200_5":
    os.environ["CREG_NN_ROWS"] = "1"
L = _lib.load()
fn = L.creg_debug_nn_waves
fn.restype = ctypes.c_longlong
fn.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int]
dev = torch.device("cuda:0")
seqs = [make_sequence(robot, s, 4, N) for s in range(5)]
mats0, clusters0, _ = initial_segmentation(seqs[0][0], K, seed=0)
reg = BatchRegistrar(mats0, clusters0, N, 5, "q", 512, 300, True, dev)
f64 = [torch.as_tensor(s[1], dtype=torch.float64, device=dev) for s in seqs]
reg.step(f64)                                                             # one registered frame: graphs captured, workspaces touched
torch.cuda.synchronize()
fn(None, 0, 1)
r = reg.seqs[0]
y = torch.as_tensor(seqs[0][2], dtype=torch.float32, device=dev)
n_ep = 60
prof = reg.plan.profile(r.m, y, r.pts_init, r.off_init, r.p_anchor, n_epochs=n_ep)
torch.cuda.synchronize()
cap = 1 << 19
buf = np.zeros((cap, 8), np.uint64)
n = fn(buf.ctypes.data, cap, 1)
rec = buf[:min(n, cap)].astype(np.int64)      # (per shard the first 2048 records: the first ~120 launches, complete)
rec = rec[np.argsort(rec[:, 0], kind="stable")]
t0, t1, t2, t3 = (rec[:, i] / 100.0 for i in range(4))                    # us
# launches: a wave that starts after every earlier wave has ended begins a new one
run_max = np.maximum.accumulate(t3)
new = np.ones(len(rec), bool)
new[1:] = t0[1:] > run_max[:-1]
lid = np.cumsum(new) - 1
n_launch = int(lid[-1]) + 1
sizes = np.bincount(lid)
cnts = np.bincount(sizes)
full = int(max(sz for sz in range(len(cnts)) if cnts[sz] >= 5))           # the wave count of a COMPLETE launch (once a shard's record slots are used up, launches are recorded in part)
keep = [l for l in range(n_launch) if sizes[l] == full][:n_ep]
print(f"{wl}: N={N} K={K}, plan launch carries {prof['nn_l1_problems_per_launch']} problems; {n} wave records, {n_launch} launches seen, {len(keep)} complete "
      f"launches of {full} waves analysed (the {n_ep} epochs' launches, then back-to-back ones); back-to-back launch time of this run {prof['nn_l1_back_to_back']:.2f} us")
span, ramp, life, ph_b, ph_v, ph_e, vis, cand, tail = [], [], [], [], [], [], [], [], []
alive_frac = []
late = {"visits": [], "life": [], "start": []}
for l in keep:
    m = lid == l
    a0, a1, a2, a3 = t0[m], t1[m], t2[m], t3[m]
    s0 = a0.min()
    span.append(a3.max() - s0); ramp.append(a0.max() - s0)
    life.append(a3 - a0); ph_b.append(a1 - a0); ph_v.append(a2 - a1); ph_e.append(a3 - a2)
    vis.append(rec[m, 4]); cand.append(rec[m, 5])
    # waves alive over the launch, as a fraction of all: integral of alive(t) / (waves x span)
    alive_frac.append(float((a3 - a0).sum() / (len(a0) * (a3.max() - s0))))
    order = np.argsort(a3)[-max(full // 100, 8):]                         # the last 1 % of the waves to finish
    late["visits"].append(rec[m, 4][order]); late["life"].append((a3 - a0)[order]); late["start"].append((a0 - s0)[order])
    tail.append(a3.max() - np.percentile(a3, 90))
cat = np.concatenate
life, ph_b, ph_v, ph_e, vis, cand = cat(life), cat(ph_b), cat(ph_v), cat(ph_e), cat(vis), cat(cand)
pc = lambda a: "  ".join(f"p{q} {np.percentile(a, q):.2f}" for q in (10, 50, 90, 99, 100))
print(f"launch span (first wave start -> last wave end): mean {np.mean(span):.2f} us (min {np.min(span):.2f}, max {np.max(span):.2f}); "
      f"dispatch ramp (first -> last wave START) mean {np.mean(ramp):.2f} us; the last 10 % of the waves end {np.mean(tail):.2f} us after the 90th percentile")
print(f"waves per SIMD if all were resident at once: {full / 1024:.2f}; mean fraction of a launch's span a wave is alive: {np.mean(alive_frac):.3f}")
print(f"wave lifetime us: mean {life.mean():.2f}  {pc(life)}")
print(f"  phase 1 (arguments, box table, queries, box-to-box bounds): mean {ph_b.mean():.2f}  {pc(ph_b)}")
print(f"  phase 2 (candidate selection + visits):                    mean {ph_v.mean():.2f}  {pc(ph_v)}")
print(f"  phase 3 (winner's coordinates, epilogue, stores drained):  mean {ph_e.mean():.2f}  {pc(ph_e)}")
print(f"visits per wave: mean {vis.mean():.2f}  {pc(vis)};  candidate blocks taken per wave: mean {cand.mean():.2f}  {pc(cand)}")
lv, ll, ls = cat(late["visits"]), cat(late["life"]), cat(late["start"])
print(f"the last 1 % of a launch's waves to finish: visits mean {lv.mean():.2f} (all waves {vis.mean():.2f}), lifetime mean {ll.mean():.2f} us (all {life.mean():.2f}), "
      f"started {ls.mean():.2f} us into the launch (ramp {np.mean(ramp):.2f})")
# lifetime by visit count
for v in sorted(set(int(x) for x in np.unique(vis)))[:12]:
    sel = vis == v
    print(f"    {v:2d} visits: {100 * sel.mean():5.1f} % of the waves, lifetime mean {life[sel].mean():.2f} us, phase 2 mean {ph_v[sel].mean():.2f} us")
dirs = rec[np.isin(lid, keep), 6]
for d in (0, 1):
    sel = dirs == d
    print(f"  direction {d} ({'predicted -> target' if d == 0 else 'target -> predicted'}): {100 * sel.mean():.1f} % of the waves, lifetime mean {life[sel].mean():.2f}, visits mean {vis[sel].mean():.2f}")
