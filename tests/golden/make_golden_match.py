"""Mint tests/golden/match_reference.npz (BUILD CONTAINER only): the reference's own ``match()``
(/root/reference/PointCloud/mlp_reg.py:240-386) run on disk under ref_shims for BOTH branches (default MLP+MLP and
``--mlp_icp``) over 4 frames, with ``mlp_reg.train`` replaced by the deterministic stub of tests/_match_stub.py.

What this pins is the loop body of match(): the frame-0 state reloaded from the first output directory (:250-253),
which poses / clusters feed "Step", "Anchor" (:338-356) and the --mlp_icp train (:301-306), that ``masked_icp``
always receives the FRAME-0 local clusters as sources and the trained clouds of the CURRENT segmentation as boxes
(:248,325), what ``resample_cluster`` gets (:326,372), the dtypes and contents of matrix/NNNN.npy, cluster/NNNN.npz
and loss.txt (:262-263,331-332,377-378,384).  registration_icp inside masked_icp is the oracle's restatement
(open3d is absent), sklearn's k_means is live.

    python tests/golden/make_golden_match.py
"""
import glob
import os
import sys
import tempfile
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import ref_shims  # noqa: E402

ref_shims.install()
import mlp_reg as ref_reg  # noqa: E402  (reference)

from _match_stub import TrainStub  # noqa: E402
from _ply import write_sequence  # noqa: E402

sys.path.insert(0, "/root/repo")
from autourdf_amd.synthetic import initial_segmentation, make_sequence  # noqa: E402

K, N, T = 5, 512, 4


def run(mlp_icp, frames, mats0, clusters0):
    stub = TrainStub()
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            raw = "data/raw/toy/4_deg_20_cams/"
            write_sequence(raw + "V0000/", frames)           # never read: only its output directory (frame-0 state) is
            write_sequence(raw + "V0001/", frames)
            first = "data/part/toy_5_seg/4_deg_20_cams/V0000/"
            os.makedirs(first + "matrix"), os.makedirs(first + "cluster")
            np.save(first + "matrix/0000.npy", mats0)
            np.savez(first + "cluster/0000.npz", **{str(i): c for i, c in enumerate(clusters0)})
            g = ref_reg.__dict__
            g.update(ROBOT="toy", NUM_SEG=K, DOF=5, STEP_SZIE=4, NUM_CAMERAS=20, MLP_ICP=mlp_icp, VIS=False, ROT="q",
                     LOSS=True, NORMAL=False, DEVICE=torch.device("cpu"),
                     RAW_PATH_LIST=sorted(glob.glob(raw + "*/")), train=stub)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                ref_reg.match(raw + "V0001/", 1)
            out = "data/part/toy_5_seg/4_deg_20_cams/V0001/"
            res = {}
            for t in range(T):
                res[f"matrix{t}"] = np.load(out + f"matrix/{t:04}.npy")
                with np.load(out + f"cluster/{t:04}.npz") as z:
                    keys = list(z.keys())
                    assert keys == [str(i) for i in range(K)]
                    cl = [z[k] for k in keys]
                res[f"cluster{t}"] = np.concatenate(cl)
                res[f"cluster{t}_dtype_is_f64"] = np.array(all(c.dtype == np.float64 for c in cl))
                res[f"offsets{t}"] = np.cumsum([0] + [len(c) for c in cl]).astype(np.int32)
            res["loss"] = np.loadtxt(out + "loss.txt")
        finally:
            os.chdir(cwd)
    res.update({"log_" + k: v for k, v in stub.log_arrays().items()})
    return res


def main():
    frames = make_sequence("wx200_5", seq=6, n_frames=T, n_points=N)
    mats0, clusters0, _ = initial_segmentation(frames[0], K, seed=6)
    out = dict(frames=np.stack(frames), mats0=mats0, clusters0=np.concatenate(clusters0),
               offsets0=np.cumsum([0] + [len(c) for c in clusters0]).astype(np.int32))
    for tag, flag in (("mlp", False), ("icp", True)):
        out.update({f"{tag}_{k}": v for k, v in run(flag, frames, mats0, clusters0).items()})
    path = os.path.join(HERE, "match_reference.npz")
    np.savez_compressed(path, **out)
    print(f"match_reference.npz {os.path.getsize(path) / 1024:.1f} KB")
    for tag in ("mlp", "icp"):
        print(tag, "matrix dtypes", [out[f"{tag}_matrix{t}"].dtype for t in range(T)],
              "sizes", [out[f"{tag}_offsets{t}"].tolist() for t in range(T)], "loss", out[f"{tag}_loss"])


if __name__ == "__main__":
    main()
