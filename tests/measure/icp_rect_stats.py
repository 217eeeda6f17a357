"""configs[4]: how much of what k_icp_nn scans does it NEED?  Measurement build, on the GPU box:
    CREG_EXTRA_FLAGS=-DCREG_ICP_RECT_STATS python -m autourdf_amd.build --variant rect && CREG_LIB_VARIANT=rect python tests/measure/icp_rect_stats.py
A wave scans the rectangle of grid cells its 16 sources' (x - r, x + r) squares touch, r = the distance to the PREVIOUS match; this counts,
per wave, the entries of that rectangle and of the rectangle the FINAL nearest distances span -- the most a centre-out scan with a shrinking
radius could save."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np                                                       # noqa: E402
import torch                                                             # noqa: E402
from autourdf_amd import _lib, ops                                       # noqa: E402
from autourdf_amd.engine import IcpRegistrar                             # noqa: E402
from autourdf_amd.synthetic import initial_segmentation, make_sequence   # noqa: E402

L = _lib.load()
fn = L.creg_debug_icp_rect
fn.argtypes = [ctypes.POINTER(ctypes.c_double)]
dev = torch.device("cuda:0")
seq = make_sequence("chain32", 0, 9, 262144)
mats0, clusters0, _ = initial_segmentation(seq[0], 128, seed=0, iters=8)
reg = IcpRegistrar(mats0, clusters0, dev)
frames = [torch.as_tensor(f, dtype=torch.float64, device=dev) for f in seq[1:]]
reg.step(frames[0])
torch.cuda.synchronize()
ops.icp_nn_counters(reset=True)
for f in frames[1:]:
    reg.step(f)
torch.cuda.synchronize()
out = (ctypes.c_double * 4)()
assert fn(out) == 0
ct = ops.icp_nn_counters()
need, scanned, cells_s, cells_n = list(out)
print(f"{len(frames) - 1} frames: waves {ct['waves']:.0f}, source-iterations {ct['source_iterations']:.0f}")
print(f"entries per wave: scanned rectangle {scanned / ct['waves']:.1f} ({cells_s / ct['waves']:.1f} cells), rectangle of the final distances {need / ct['waves']:.1f} "
      f"({cells_n / ct['waves']:.1f} cells): ratio {need / scanned:.3f}")
