"""Stage 1 of the XCD-resident train, a MEASUREMENT build:

    CREG_EXTRA_FLAGS=-DCREG_XCD_PROBE python -m autourdf_amd.build --variant xcd        # on the GPU box (hipcc is there; .gpurunignore keeps variant libraries out of the snapshot)
    CREG_LIB_VARIANT=xcd python tests/measure/xcd_stage1.py                              # on the GPU box

(a) a barrier among the workgroups of ONE XCD (members found at run time through HW_REG_XCC_ID + a per-XCD ticket) with a 4 KB
    hand-off, in three publication forms;  (b) the plan's nearest-neighbour launch of configs[1] (N=4096, K=20) with every problem's
    blocks confined to one XCD's 32 CUs (problem = XCC_ID, blocks from a per-XCD queue), in both search forms (four / sixteen queries
    per wave), against the plan's own chip-wide launch.  Kill criteria of the brief: (a) > 2.5 us or (b) > 14 us."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import numpy as np                                                       # noqa: E402
import torch                                                             # noqa: E402
from autourdf_amd import _lib, ops                                       # noqa: E402
from autourdf_amd.engine import BatchRegistrar                           # noqa: E402
from autourdf_amd.synthetic import initial_segmentation, make_sequence   # noqa: E402

L = _lib.load()
fn = L.creg_debug_xcd_stage1
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.POINTER(_lib.TrainArgs), ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_double), ctypes.c_void_p]
dev = torch.device("cuda:0")
N, K, S = 4096, 20, 5
seq = make_sequence("wx200_5", 0, 3, N)
mats0, clusters0, _ = initial_segmentation(seq[0], K, seed=0)
y = torch.as_tensor(seq[1], dtype=torch.float32, device=dev)
first = True
for rows in ("0", "1"):
    os.environ["CREG_NN_ROWS"] = rows                                    # the plan reads it when it is created
    reg = BatchRegistrar(mats0, clusters0, N, S, "q", 512, 300, True, dev)
    r = reg.seqs[0]
    a = reg.plan._args(r.m, y, r.pts, r.off, r.p_step, 2e-4, 0.7, 5, 200, (None, None, None, None, None))
    for wg in ((4, 3, 2) if rows == "0" else (4, 2, 1)):
        for nz in (5, 1):
            out = (ctypes.c_double * 64)()
            _lib.check(fn(reg.plan.plan, ctypes.byref(a), nz, wg, out, None), "creg_debug_xcd_stage1")
            v = list(out)
            if first:
                first = False
                print("(a) barrier of one XCD's workgroups + 4 KB hand-off, 256 workgroups of 256 threads (8 groups at once), us per round, slowest member")
                print(f"    members per XCD: {v[9]:.0f} .. {v[10]:.0f}")
                for mode, name in enumerate(("XCD-local: plain stores, vmcnt(0), counter, sc1 polls + sc1 payload loads",
                                             "placement-independent: plain stores, agent release, counter, relaxed poll, agent acquire, plain loads",
                                             "write-through: sc1 stores, vmcnt(0), counter, sc1 polls + sc1 payload loads")):
                    print(f"    {name}: idle {v[mode]:.2f}, uneven arrival {v[3 + mode]:.2f}, wrong payload words {v[6 + mode]:.0f}")
            form = "sixteen queries per wave (nn_l1_rows)" if rows == "1" else "four queries per wave (nn_l1_block_pruned)"
            print(f"(b) {form}, {wg} workgroups of 512 per CU in the confined launch, {nz} problem(s):")
            print(f"    plan's launch: 1 problem chip-wide {v[16]:.2f} us, {nz} problems chip-wide {v[17]:.2f} us")
            print(f"    confined, problem = XCC_ID: {nz} problems {v[18]:.2f} us (agent-scope queue) / {v[19]:.2f} us (workgroup-scope queue); 1 problem {v[20]:.2f} us; "
                  f"outputs identical to the plan's launch: {bool(v[21])}; workgroups per XCD seen {[int(x) for x in v[24:32]]}")
    if rows == "0":
        print(f"(Stage 2 pre-check) k_l2 (the next hidden activation, H2 / 16 = 48 workgroups per problem), 200 back to back, ONE problem (two measurements): chip-wide {v[32]:.2f} / "
              f"{v[35]:.2f} us; its 48 workgroups confined to one XCD {v[33]:.2f} / {v[36]:.2f} us; h2 identical: {bool(v[34])}")
    del reg
    torch.cuda.synchronize()
