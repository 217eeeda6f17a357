"""Debug build only (CREG_EXTRA_FLAGS=-DCREG_BD_STAMPS python autourdf_amd/build.py after touching train_engine.hip): where the two
roles of the train plan's backward launch k_bd spend a launch.  Runs bench.py's roofline leg (200 back-to-back launches of k_bd carrying
the problems of the larger graph branch) and prints, per role, when its workgroups start and end inside the launch and the mean duration
of each phase of a workgroup (100 MHz wall clock, thread 0 of every workgroup).  python tests/measure/bd_stamps.py [bench.py arguments]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch                                                             # noqa: E402
from autourdf_amd import _lib                                           # noqa: E402
import bench                                                             # noqa: E402

L = _lib.load()
fn = L.creg_debug_bd_stamps
fn.argtypes = [ctypes.c_void_p]
sys.argv = ["bench.py", "--steps", "5", "--warmup", "1", "--no-cpu-baseline", "--no-icp-variant"] + sys.argv[1:]
bench.main()
torch.cuda.synchronize()
out = (ctypes.c_double * 24)()
assert fn(out) == 0
v = list(out)
print(f"\nk_bd, {v[7]:.0f} launch slots: span first workgroup start -> last workgroup end {v[0]:.2f} us")
print(f"  B role (backward to the encoder): {v[5]:5.0f} workgroups per launch, first starts at {v[1]:5.2f} us, last ends at {v[2]:5.2f} us")
print(f"  D role (dW + Adam of the rows)  : {v[6]:5.0f} workgroups per launch, first starts at {v[3]:5.2f} us, last ends at {v[4]:5.2f} us")
names_b = ["B1 every load requested", "B2 operands landed, MFMAs, partial tiles to LDS, barrier", "B3 cross-wave reduction, activation gradient, g_x1 to LDS",
           "B4 dW1, Adam, encoder-row stores issued", "B5 next activation tile + all stores acknowledged"]
names_d = ["D1 activations staged", "D2 accumulation over the pose rows (incl. waiting for the row's parameters)", "D3 Adam + write-through stores acknowledged"]
print("  mean phase durations of a B workgroup (us): " + "; ".join(f"{n} {v[8 + i]:.2f}" for i, n in enumerate(names_b)) + f"; total {sum(v[8:13]):.2f}")
print("  mean phase durations of a D workgroup (us): " + "; ".join(f"{n} {v[16 + i]:.2f}" for i, n in enumerate(names_d)) + f"; total {sum(v[16:19]):.2f}")
