"""N3: link refinement and the evaluation ICP filter on the K4 kernel's point-to-point mode."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _write_links(g, d, T):
    os.makedirs(d + "cluster")
    for t in range(T):
        np.savez(d + f"cluster/{t:04}.npz", **{str(i): g[f"in.{t}.{i}"] for i in range(4)})


def test_refine_links_clusters_vs_reference_golden(golden, tmp_path):
    from autourdf_amd.helper_functions import load_pc_npz
    from autourdf_amd.link import refine_links_clusters
    g = golden("link_refine_reference.npz")
    T, dof = int(g["T"]), int(g["dof"])
    d = str(tmp_path) + "/"
    _write_links(g, d, T)
    refine_links_clusters([d], 0, T, dof)
    for t in range(T):
        got = load_pc_npz(d + f"cluster_rf/{t:04}.npz")
        assert len(got) == dof + 1
        for i in range(dof + 1):
            assert got[i].dtype == np.float64
            np.testing.assert_allclose(got[i], g[f"out.{t}.{i}"], atol=1e-8)       # poses: 1e-5 demanded


def test_icp_p2p_single_equals_batch_and_oracle(golden):
    """creg_icp_p2p_f64 == the batched entry point bit for bit; both within 1e-8 of the oracle ICP;
    sources larger than the LDS budget (workspace path) and an empty pair."""
    from autourdf_amd import ops
    from oracle.icp import registration_icp
    from scipy.spatial.transform import Rotation
    dev = torch.device("cuda")
    rng = np.random.default_rng(2)
    tgt = [rng.uniform(-0.2, 0.2, size=(n, 3)) for n in (1500, 300, 64)]
    src = []
    for c in tgt:
        R = Rotation.from_rotvec(rng.normal(scale=0.02, size=3)).as_matrix()
        src.append(c[rng.permutation(len(c))[: len(c) - 7]] @ R.T + rng.normal(scale=0.003, size=3))
    src.append(np.zeros((0, 3))); tgt.append(rng.uniform(size=(10, 3)))                 # empty source: T stays I
    cat = lambda cs: torch.as_tensor(np.concatenate(cs), device=dev)
    off = lambda cs: torch.tensor(np.concatenate([[0], np.cumsum([len(c) for c in cs])]), dtype=torch.int32, device=dev)
    init = torch.eye(4, dtype=torch.float64, device=dev).repeat(len(src), 1, 1)
    T1, m1, it1 = ops.icp_p2p(cat(src), off(src), cat(tgt), off(tgt), init, th=1.0, max_iteration=1000)
    (T2, m2, it2), (T3, _, _) = ops.icp_p2p_batch([(cat(src), off(src), cat(tgt), off(tgt), init)] * 2, th=1.0, max_iteration=1000)
    assert torch.equal(T1, T2) and torch.equal(m1, m2) and torch.equal(it1, it2) and torch.equal(T2, T3)
    o = off(src).cpu().numpy()
    for i, (s, t) in enumerate(zip(src, tgt)):
        if len(s) == 0:
            np.testing.assert_array_equal(T1[i].cpu().numpy(), np.eye(4))
            continue
        T_ref, _, _, _ = registration_icp(s, t, 1.0, np.eye(4), 1000)
        np.testing.assert_allclose(T1[i].cpu().numpy(), T_ref, atol=1e-8)
        np.testing.assert_allclose(m1[o[i]:o[i + 1]].cpu().numpy(), s @ T_ref[:3, :3].T + T_ref[:3, 3], atol=1e-8)


def test_evaluation_icp_filter_and_chamfer_vs_oracle():
    """Sim/evaluation.py:69-81,358-362: ICP filter (th 0.01) then L1 Chamfer in float32."""
    from autourdf_amd.cluster_icp import PointCloud
    from autourdf_amd.evaluation import icp_filter, torch_chamfer_distance
    from oracle import chamfer as ochamfer, link as olink
    rng = np.random.default_rng(8)
    gt = rng.uniform(-0.1, 0.1, size=(900, 3))
    pred = gt[:850] + np.array([0.002, -0.001, 0.0015]) + rng.normal(scale=2e-4, size=(850, 3))
    T, moved = icp_filter(PointCloud(pred), PointCloud(gt))
    T_ref, moved_ref = olink.icp_filter(pred, gt)
    np.testing.assert_allclose(T, T_ref, atol=1e-8)
    np.testing.assert_allclose(moved.points, moved_ref, atol=1e-8)
    loss = torch_chamfer_distance(moved, PointCloud(gt))
    want = ochamfer.chamfer_distance(torch.tensor(moved_ref, dtype=torch.float32)[None], torch.tensor(gt, dtype=torch.float32)[None], norm=1)[0].item()
    assert abs(loss - want) <= 1e-6 * max(1.0, abs(want))


def test_icp_targets_beyond_the_lds_budget_use_the_workspace_path():
    """> 4096 targets of one pair do not fit the kernel's LDS table: the same loop then reads them through the
    workspace index list (both the point-to-point and the masked entry points), same poses as the oracle."""
    from autourdf_amd import ops
    from autourdf_amd.cluster_icp import masked_icp
    from oracle import icp as oicp
    from scipy.spatial.transform import Rotation
    dev = torch.device("cuda")
    rng = np.random.default_rng(12)
    tgt = rng.uniform(-0.3, 0.3, size=(6000, 3))
    R = Rotation.from_rotvec([0.01, -0.02, 0.015]).as_matrix()
    src = tgt[rng.permutation(6000)[:700]] @ R.T + np.array([0.002, -0.001, 0.003])
    off = lambda n: torch.tensor([0, n], dtype=torch.int32, device=dev)
    T, moved, it = ops.icp_p2p(torch.as_tensor(src, device=dev), off(700), torch.as_tensor(tgt, device=dev), off(6000),
                               torch.eye(4, dtype=torch.float64, device=dev)[None], th=1.0, max_iteration=200)
    T_ref, _, _, _ = oicp.registration_icp(src, tgt, 1.0, np.eye(4), 200)
    np.testing.assert_allclose(T[0].cpu().numpy(), T_ref, atol=1e-8)
    # masked entry point: one cluster whose box (x 1.2) keeps ~all 6000 frame points
    world = (src + 0.0).astype(np.float32)
    big = np.concatenate([world, np.array([[-0.35, -0.35, -0.35], [0.35, 0.35, 0.35]], np.float32)])   # stretch the box
    local = np.concatenate([src, [[-0.35, -0.35, -0.35], [0.35, 0.35, 0.35]]])
    w, M = masked_icp([local], [big], tgt, np.eye(4)[None])
    _, M_ref = oicp.masked_icp([local], [big], tgt, np.eye(4)[None])
    np.testing.assert_allclose(M, M_ref, atol=1e-8)


def test_icp_inlier_threshold_is_strict_like_open3d_search_hybrid():
    """open3d's KDTreeFlann::SearchHybrid keeps a neighbour with d^2 < r^2 (fidelity nit): a source whose nearest target
    lies EXACTLY on max_correspondence_distance is no correspondence.  Four sources sit on their targets, a fifth is exactly th = 0.5
    from its nearest one: with the strict rule the first fit is the identity (kernel and oracle agree), a hair more radius pulls the
    fifth pair in.  One iteration only: the identity fit carries 1e-15 of rounding, which would move the fifth source off the radius."""
    from autourdf_amd import ops
    from oracle.icp import registration_icp
    dev = torch.device("cuda")
    tgt = np.array([[2, 0, 0], [2, 1, 0], [2, 0, 1], [3, 0, 0], [0.5, 0, 0]], np.float64)
    src = np.array([[2, 0, 0], [2, 1, 0], [2, 0, 1], [3, 0, 0], [0, 0, 0]], np.float64)
    t = lambda a: torch.as_tensor(a, device=dev)
    off = torch.tensor([0, 5], dtype=torch.int32, device=dev)
    init = torch.eye(4, dtype=torch.float64, device=dev)[None]
    T, _, _ = ops.icp_p2p(t(src), off, t(tgt), off, init, th=0.5, max_iteration=1)
    T_ref, _, _, _ = registration_icp(src, tgt, 0.5, np.eye(4), 1)
    np.testing.assert_allclose(T_ref, np.eye(4), atol=1e-14)
    np.testing.assert_allclose(T[0].cpu().numpy(), np.eye(4), atol=1e-12)
    T2, _, _ = ops.icp_p2p(t(src), off, t(tgt), off, init, th=0.5 + 1e-9, max_iteration=1)
    T2_ref, _, _, _ = registration_icp(src, tgt, 0.5 + 1e-9, np.eye(4), 1)
    assert np.abs(T2[0].cpu().numpy() - np.eye(4)).max() > 1e-3
    np.testing.assert_allclose(T2[0].cpu().numpy(), T2_ref, atol=1e-8)


def test_icp_point_to_point_in_the_many_workgroup_regime_vs_oracle():
    """Round 5: point-to-point mode of K4's many-workgroup regime.  Sim/evaluation.py:358-362 registers WHOLE
    robot clouds; above 1024 sources per pair the search now runs 64 sources per workgroup over a cell grid of the pair's own target
    segment instead of one workgroup per pair.  (a) a 20000 x 20000 `icp_filter` against the oracle (1e-8: pose and moved cloud),
    (b) three pairs of different sizes in one call, each against the oracle and against the one-workgroup path's own result
    (CREG_ICP_P2P_ONE_WORKGROUP=1), iteration counts included."""
    import os
    from scipy.spatial.transform import Rotation
    from autourdf_amd import ops
    from autourdf_amd.cluster_icp import PointCloud
    from autourdf_amd.evaluation import icp_filter
    from oracle import icp as oicp, link as olink
    dev = torch.device("cuda", torch.cuda.current_device())
    rng = np.random.default_rng(21)
    # (a) a surface-like cloud (a noisy torus), prediction = a rotated, shifted, noisy subsample
    u, v = rng.uniform(0, 2 * np.pi, 20000), rng.uniform(0, 2 * np.pi, 20000)
    gt = np.stack([(0.3 + 0.08 * np.cos(v)) * np.cos(u), (0.3 + 0.08 * np.cos(v)) * np.sin(u), 0.08 * np.sin(v)], 1)
    R = Rotation.from_rotvec([0.004, -0.003, 0.005]).as_matrix()
    pred = gt[rng.permutation(20000)] @ R.T + np.array([0.0015, -0.001, 0.002]) + rng.normal(scale=2e-4, size=(20000, 3))
    T, moved = icp_filter(PointCloud(pred), PointCloud(gt))
    T_ref, moved_ref = olink.icp_filter(pred, gt)
    np.testing.assert_allclose(T, T_ref, atol=1e-8)
    np.testing.assert_allclose(moved.points, moved_ref, atol=1e-8)
    # (b) three pairs: 3000 / 1500 / 2600 sources against 4000 / 2500 / 3000 targets (1e-8 vs oracle; identical to the one-workgroup path)
    ns, nt = (3000, 1500, 2600), (4000, 2500, 3000)
    srcs, tgts = [], []
    for i, (a, b) in enumerate(zip(ns, nt)):
        t = rng.uniform(-0.2, 0.2, size=(b, 3)) * np.array([1.0, 0.6, 0.15]) + i
        Ri = Rotation.from_rotvec(rng.normal(scale=0.01, size=3)).as_matrix()
        srcs.append(t[rng.permutation(b)[:a]] @ Ri.T + rng.normal(scale=1e-3, size=3))
        tgts.append(t)
    so = torch.tensor(np.cumsum((0,) + ns), dtype=torch.int32, device=dev)
    to = torch.tensor(np.cumsum((0,) + nt), dtype=torch.int32, device=dev)
    args = (torch.as_tensor(np.concatenate(srcs), device=dev), so, torch.as_tensor(np.concatenate(tgts), device=dev), to,
            torch.eye(4, dtype=torch.float64, device=dev).repeat(3, 1, 1))
    T3, m3, it3 = ops.icp_p2p(*args, th=0.05, max_iteration=300)
    os.environ["CREG_ICP_P2P_ONE_WORKGROUP"] = "1"
    try:
        T1, m1, it1 = ops.icp_p2p(*args, th=0.05, max_iteration=300)
    finally:
        del os.environ["CREG_ICP_P2P_ONE_WORKGROUP"]
    for i in range(3):
        T_ref, _, _, n_ref = oicp.registration_icp(srcs[i], tgts[i], 0.05, np.eye(4), 300)
        np.testing.assert_allclose(T3[i].cpu().numpy(), T_ref, atol=1e-8)
        assert int(it3[i]) == n_ref == int(it1[i])
    np.testing.assert_allclose(T3.cpu().numpy(), T1.cpu().numpy(), atol=1e-12)
    np.testing.assert_allclose(m3.cpu().numpy(), m1.cpu().numpy(), atol=1e-12)


def test_icp_point_to_point_large_regime_empty_target_segment_and_far_coordinates():
    """ADVICE r5 (icp.hip, many-workgroup point-to-point mode): (a) a pair whose TARGET segment is empty used to give a NaN box centre and an
    infinite screen bound -- it must leave its pose at the initial one, 0 iterations' worth of change, and not disturb its neighbours;
    (b) targets far from the origin relative to their extent (|coordinate| / extent ~ 1e6: the float32 box is a rounding of the fp64
    targets, now widened by one ulp) still match the oracle."""
    from autourdf_amd import ops
    from oracle import icp as oicp
    dev = torch.device("cuda", torch.cuda.current_device())
    rng = np.random.default_rng(5)
    base = rng.uniform(-0.05, 0.05, size=(3000, 3)) * np.array([1.0, 0.7, 0.2])
    far = np.array([1.0e4, -2.0e4, 5.0e3])
    srcs = [base[:2600] + 1e-3, rng.uniform(size=(1500, 3)), (base + far)[:2800] + np.array([8e-4, -5e-4, 3e-4])]
    tgts = [base, np.zeros((0, 3)), base + far]
    so = torch.tensor(np.cumsum([0] + [len(a) for a in srcs]), dtype=torch.int32, device=dev)
    to = torch.tensor(np.cumsum([0] + [len(a) for a in tgts]), dtype=torch.int32, device=dev)
    T, moved, it = ops.icp_p2p(torch.as_tensor(np.concatenate(srcs), device=dev), so, torch.as_tensor(np.concatenate(tgts), device=dev), to,
                               torch.eye(4, dtype=torch.float64, device=dev).repeat(3, 1, 1), th=0.05, max_iteration=200)
    T = T.cpu().numpy()
    assert np.isfinite(T).all() and np.isfinite(moved.cpu().numpy()).all()
    np.testing.assert_allclose(T[1], np.eye(4), atol=0)                                   # nothing to register to: the pose stays
    for i in (0, 2):
        T_ref, _, _, n_ref = oicp.registration_icp(srcs[i], tgts[i], 0.05, np.eye(4), 200)
        np.testing.assert_allclose(T[i][:3, :3], T_ref[:3, :3], atol=1e-8)
        np.testing.assert_allclose(T[i][:3, 3], T_ref[:3, 3], atol=1e-8 if i == 0 else 2e-7)    # (translations of 1e4: fp64 ulp 2e-12 x the rotation's lever)
        assert int(it[i]) == n_ref
