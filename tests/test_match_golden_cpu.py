"""CPU: the oracle's restatement of match()'s loop body against the reference-minted match-level golden
(tests/golden/make_golden_match.py: the reference's match() run on disk with the deterministic train stub)."""
import numpy as np
import pytest

from oracle import registration
from tests._match_stub import TrainStub

T = 4


def _split(flat, off):
    return [flat[off[i]:off[i + 1]] for i in range(len(off) - 1)]


def check_against_golden(g, tag, mats, clusters, losses, log, pose_tol, cluster_tol):
    """Shared by the CPU (oracle) and GPU (product) replays."""
    np.testing.assert_allclose(np.asarray(losses), g[f"{tag}_loss"], rtol=0, atol=0)
    for t in range(T):
        ref_m = g[f"{tag}_matrix{t}"]
        assert np.asarray(mats[t]).dtype == ref_m.dtype, f"{tag} frame {t}: pose dtype {np.asarray(mats[t]).dtype} != {ref_m.dtype}"
        np.testing.assert_allclose(np.asarray(mats[t]), ref_m, rtol=0, atol=pose_tol, err_msg=f"{tag} poses, frame {t}")
        off = np.cumsum([0] + [len(c) for c in clusters[t]])
        np.testing.assert_array_equal(off, g[f"{tag}_offsets{t}"], err_msg=f"{tag} cluster sizes, frame {t}")
        assert all(np.asarray(c).dtype == np.float64 for c in clusters[t])
        np.testing.assert_allclose(np.concatenate(clusters[t]), g[f"{tag}_cluster{t}"], rtol=0, atol=cluster_tol,
                                   err_msg=f"{tag} clusters, frame {t}")
    # which train() got what: model slot (0 = model, 1 = model_rf), learning rate, cluster sizes, input checksums
    np.testing.assert_array_equal(log["model"], g[f"{tag}_log_model"])
    np.testing.assert_array_equal(log["lr"], g[f"{tag}_log_lr"])
    np.testing.assert_array_equal(log["sizes"], g[f"{tag}_log_sizes"])
    for k in ("sum_m", "sum_y", "sum_c"):
        np.testing.assert_allclose(log[k], g[f"{tag}_log_{k}"], rtol=0, atol=1e-3 if k != "sum_y" else 1e-4, err_msg=k)


@pytest.mark.parametrize("tag,mlp_icp", [("mlp", False), ("icp", True)])
def test_oracle_match_sequence_equals_reference_match(golden, tag, mlp_icp):
    g = golden("match_reference.npz")
    frames = list(g["frames"])
    clusters0 = _split(g["clusters0"], g["offsets0"])
    stub = TrainStub()
    mats, clusters, losses = registration.match_sequence(frames, g["mats0"], clusters0, stub, mlp_icp=mlp_icp)
    # the oracle runs the reference's own numpy / sklearn calls: poses 1e-12 (ICP through an SVD), clusters 1e-9
    check_against_golden(g, tag, mats, clusters, losses, stub.log_arrays(), 1e-9, 1e-9)


def test_golden_exercises_the_size_mismatch(golden):
    """From frame 2 on, the --mlp_icp branch hands masked_icp frame-0 sources and boxes of another segmentation."""
    g = golden("match_reference.npz")
    assert not np.array_equal(g["icp_offsets1"], g["offsets0"])
    # the train() of frame 2 ran on the re-sampled sizes, masked_icp's sources stayed frame 0's
    np.testing.assert_array_equal(g["icp_log_sizes"][1], np.diff(g["icp_offsets1"]))


def test_updating_the_icp_source_would_fail_the_golden(golden):
    """Round 1's composition (ICP source := re-sampled clusters) is visibly different: the golden detects it."""
    from oracle.icp import masked_icp
    g = golden("match_reference.npz")
    frames = list(g["frames"])
    src = _split(g["icp_cluster1"], g["icp_offsets1"])                 # what round 1 passed at frame 2
    stub = TrainStub()
    import torch
    stub(torch.tensor(g["mats0"], dtype=torch.float32), torch.tensor(frames[1], dtype=torch.float32), "model",
         [torch.tensor(c, dtype=torch.float32) for c in _split(g["clusters0"], g["offsets0"])])
    pred, _, m, _ = stub(torch.tensor(g["icp_matrix1"], dtype=torch.float32), torch.tensor(frames[2], dtype=torch.float32),
                         "model", [torch.tensor(c, dtype=torch.float32) for c in src])
    _, wrong = masked_icp(src, pred, frames[2], m.numpy())
    assert np.abs(wrong - g["icp_matrix2"]).max() > 1e-4
