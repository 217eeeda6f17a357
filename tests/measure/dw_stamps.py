"""Debug build only (CREG_EXTRA_FLAGS=-DCREG_STAMPS): when k_dw's block kinds start and end inside a launch."""
import ctypes
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch                                                             # noqa: E402
from autourdf_amd import _lib                                           # noqa: E402

L = _lib.load()
fn = L.creg_debug_dw_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
out = (ctypes.c_ulonglong * 18)()
import bench                                                             # noqa: E402
sys.argv = ["bench.py", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-icp-variant", "--no-roofline"]
fn(None, 1)
bench.main()
torch.cuda.synchronize()
fn(out, 0)
v = np.array(list(out), dtype=np.float64)
st = v[:16].reshape(4, 4)
print(f"launches {v[16]:.0f}, mean launch span (first block start -> last block end) {v[17] / max(v[16], 1) / 100:.2f} us")
for kind, name in ((3, "encoder rows"), (0, "hidden rows"), (1, "output rows")):
    n = max(st[kind, 0], 1)
    print(f"  {name:13s}: {st[kind, 0] / max(v[16], 1):6.1f} blocks per launch, start at {st[kind, 1] / n / 100:5.2f} us, end at {st[kind, 2] / n / 100:5.2f} us, duration {st[kind, 3] / n / 100:5.2f} us (means)")
