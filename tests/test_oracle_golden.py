"""CPU: pin the oracle against the golden vectors minted from the reference (tests/golden/make_golden.py)."""
import numpy as np
import pytest
import torch

from oracle import chamfer, dq, icp, kmeans, models, registration
from oracle import transforms as T


def _split(flat, offsets):
    return [flat[offsets[i]:offsets[i + 1]] for i in range(len(offsets) - 1)]


@pytest.mark.parametrize("tag,tol", [("f32", 1e-6), ("f64", 1e-14)])
def test_dq_functions_match_reference(golden, tag, tol):
    g = golden("dq_reference.npz")
    t = lambda k: torch.from_numpy(g[f"{tag}_{k}"])
    M, d, db, noisy = t("M"), t("dq"), t("dq_b"), t("noisy")
    close = lambda a, k: np.testing.assert_allclose(a.numpy(), g[f"{tag}_{k}"], rtol=0, atol=tol)
    close(dq.transform_to_dualquat(M), "dq")
    close(dq.dualquat_to_transform(noisy), "to_transform")
    q, tr = dq.dualquat_to_quat_trans(noisy)
    close(q, "qt_q"); close(tr, "qt_t")
    R, tr = dq.dualquat_to_rot_trans(noisy)
    close(R, "rt_R"); close(tr, "rt_t")
    close(dq.dualquat_multiply(d, db), "mul")
    close(dq.dualquat_invert(noisy), "inv")
    close(dq.quaternion_conjugate(d[:, :4]), "conj")
    close(dq.quat_trans_to_dualquat(d[:, :4], M[:, :3, 3]), "from_qt")
    close(dq.rot_trans_to_dualquat(M[:, :3, :3], M[:, :3, 3]), "from_rt")
    close(dq.transform_from_rot_trans(M[:, :3, :3], M[:, :3, 3]), "assemble")
    close(dq.point_to_dualquat(M[:, :3, 3]), "point")


def test_dq_identities():
    g = torch.Generator().manual_seed(0)
    from scipy.spatial.transform import Rotation
    M = torch.eye(4, dtype=torch.float64).repeat(32, 1, 1)
    M[:, :3, :3] = torch.from_numpy(Rotation.random(32, random_state=3).as_matrix())
    M[:, :3, 3] = torch.randn(32, 3, generator=g, dtype=torch.float64)
    d = dq.transform_to_dualquat(M)
    assert (dq.dualquat_to_transform(d) - M).abs().max() < 1e-12
    ident = dq.dualquat_multiply(d, dq.dualquat_invert(d))
    assert (ident - torch.tensor([1., 0, 0, 0, 0, 0, 0, 0], dtype=torch.float64)).abs().max() < 1e-12


def test_transforms_against_scipy():
    from scipy.spatial.transform import Rotation
    rot = Rotation.random(200, random_state=5)
    R = torch.from_numpy(rot.as_matrix())
    q = T.matrix_to_quaternion(R)
    qs = rot.as_quat()[:, [3, 0, 1, 2]]
    qs = np.where(qs[:, :1] < 0, -qs, qs)
    np.testing.assert_allclose(q.numpy(), qs, atol=1e-14)
    np.testing.assert_allclose(T.quaternion_to_matrix(3.0 * q).numpy(), rot.as_matrix(), atol=1e-14)
    e = T.matrix_to_euler_angles(R, "XYZ")
    np.testing.assert_allclose(T.euler_angles_to_matrix(e, "XYZ").numpy(), rot.as_matrix(), atol=1e-13)
    np.testing.assert_allclose(T.rotation_6d_to_matrix(T.matrix_to_rotation_6d(R)).numpy(),
                               rot.as_matrix(), atol=1e-14)
    a, b = Rotation.random(50, random_state=6), Rotation.random(50, random_state=7)
    qa = torch.from_numpy(a.as_quat()[:, [3, 0, 1, 2]])
    qb = torch.from_numpy(b.as_quat()[:, [3, 0, 1, 2]])
    np.testing.assert_allclose(T.quaternion_to_matrix(T.quaternion_raw_multiply(qa, qb)).numpy(),
                               (a * b).as_matrix(), atol=1e-14)


def _load_sd(g, prefix):
    return {k[len(prefix):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix)}


def test_models_match_reference(golden):
    g = golden("models_reference.npz")
    q = models.QRegMLP(True, hidden_dim=32)
    q.load_state_dict(_load_sd(g, "q."))
    t, r = q(torch.from_numpy(g["q_in"]))
    np.testing.assert_allclose(t.detach().numpy(), g["q_out_t"], atol=1e-7)
    np.testing.assert_allclose(r.detach().numpy(), g["q_out_r"], atol=1e-7)
    d = models.DQRegMLP(hidden_dim=32)
    d.load_state_dict(_load_sd(g, "dq."))
    np.testing.assert_allclose(d(torch.from_numpy(g["dq_in"])).detach().numpy(), g["dq_out"], atol=1e-7)


def test_model_parameter_count():
    assert sum(p.numel() for p in models.QRegMLP(True, 512).parameters()) == 425991


def test_calculate_pc_matches_reference(golden):
    g = golden("calculate_pc.npz")
    out = registration.calculate_pc([torch.from_numpy(c) for c in _split(g["local"], g["offsets"])],
                                    torch.from_numpy(g["mats"]))
    np.testing.assert_array_equal(np.concatenate([o.numpy() for o in out]), g["world"])


@pytest.mark.parametrize("rot,ctor", [("q", lambda: models.QRegMLP(True, 32)), ("dq", lambda: models.DQRegMLP(32))])
def test_train_matches_reference_full_300_epochs(golden, rot, ctor):
    g = golden("train_reference.npz")
    model = ctor()
    model.load_state_dict(_load_sd(g, f"{rot}.sd."))
    clusters = [torch.from_numpy(c) for c in _split(g[f"{rot}_local"], g[f"{rot}_offsets"])]
    pred, best_m, min_loss, hist = registration.train(
        torch.from_numpy(g[f"{rot}_m"]), torch.from_numpy(g[f"{rot}_y"]), model, clusters, rot=rot)
    # same torch build, same op sequence: the trajectory is reproduced to rounding
    assert abs(min_loss - float(g[f"{rot}_min_loss"])) < 1e-7
    np.testing.assert_allclose(best_m.detach().numpy(), g[f"{rot}_best_m"], atol=1e-6)
    np.testing.assert_allclose(np.concatenate(pred), g[f"{rot}_best_pred"], atol=1e-6)
    sd_sum = sum(float(v.double().abs().sum()) for v in model.state_dict().values())
    assert abs(sd_sum - float(g[f"{rot}_final_sd_sum"])) < 1e-3 * sd_sum
    assert len(hist["loss"]) == 300


@pytest.mark.parametrize("rot,ctor", [("q", lambda: models.QRegMLP(True, 64)), ("dq", lambda: models.DQRegMLP(64))])
def test_train_hidden64_matches_reference_six_epochs_and_300(golden, rot, ctor):
    """train_reference_h64.npz (the reference's own train() at a width the HIP plan tiles): per-epoch losses of the first six
    epochs, the best pose, the parameters after them, and the full 300-epoch run."""
    g = golden("train_reference_h64.npz")
    clusters = [torch.from_numpy(c) for c in _split(g[f"{rot}_local"], g[f"{rot}_offsets"])]
    m, y = torch.from_numpy(g[f"{rot}_m"]), torch.from_numpy(g[f"{rot}_y"])
    model = ctor()
    model.load_state_dict(_load_sd(g, f"{rot}.sd."))
    _, best_m, min_loss, hist = registration.train(m, y, model, clusters, rot=rot, epochs=6)
    np.testing.assert_allclose(np.array(hist["loss"]), g[f"{rot}_e6_loss_hist"], rtol=1e-6)
    np.testing.assert_allclose(best_m.detach().numpy(), g[f"{rot}_e6_best_m"], atol=1e-6)
    for k, v in model.state_dict().items():
        np.testing.assert_allclose(v.numpy(), g[f"{rot}_e6.final.{k}"], atol=2e-6)
    model = ctor()
    model.load_state_dict(_load_sd(g, f"{rot}.sd."))
    _, best_m, min_loss, hist = registration.train(m, y, model, clusters, rot=rot)
    assert abs(min_loss - float(g[f"{rot}_e300_min_loss"])) < 1e-6
    np.testing.assert_allclose(best_m.detach().numpy(), g[f"{rot}_e300_best_m"], atol=2e-5)


@pytest.mark.parametrize("rot,ctor", [("6d", lambda: models.RRegMLP(64)), ("rpy", lambda: models.RegMLP(6, 3))])
def test_train_optional_representations_match_reference(golden, rot, ctor):
    """train_reference_rot.npz: the reference's own train() with --r 6d (RRegMLP, hidden 64) and --r rpy (RegMLP(6, 3), the
    reference's construction: hidden 3), mlp_reg.py:72-76,86-90,285-290: six epochs pinned per epoch, and the 300-epoch run."""
    g = golden("train_reference_rot.npz")
    clusters = [torch.from_numpy(c) for c in _split(g[f"{rot}_local"], g[f"{rot}_offsets"])]
    m, y = torch.from_numpy(g[f"{rot}_m"]), torch.from_numpy(g[f"{rot}_y"])
    model = ctor()
    model.load_state_dict(_load_sd(g, f"{rot}.sd."))
    _, best_m, min_loss, hist = registration.train(m, y, model, clusters, rot=rot, epochs=6)
    np.testing.assert_allclose(np.array(hist["loss"]), g[f"{rot}_e6_loss_hist"], rtol=1e-6)
    np.testing.assert_allclose(best_m.detach().numpy(), g[f"{rot}_e6_best_m"], atol=1e-6)
    for k, v in model.state_dict().items():
        np.testing.assert_allclose(v.numpy(), g[f"{rot}_e6.final.{k}"], atol=2e-6)
    model = ctor()
    model.load_state_dict(_load_sd(g, f"{rot}.sd."))
    _, best_m, min_loss, hist = registration.train(m, y, model, clusters, rot=rot)
    assert abs(min_loss - float(g[f"{rot}_e300_min_loss"])) < 1e-6
    np.testing.assert_allclose(best_m.detach().numpy(), g[f"{rot}_e300_best_m"], atol=2e-5)


def test_train_c1_headline_shape_oracle_reproduces_the_reference(golden):
    """train_reference_c1.npz: the reference's own train() with its default model (QRegMLP(True, hidden_dim=512), mlp_reg.py:281-282)
    at the BASELINE configs[1] shape (N = 4096, K = 20).  The oracle is the same op sequence on the same torch build: six epochs of
    it reproduce the reference's losses, the pose every epoch evaluated, the best pose / cloud and the trained parameters (a strided
    sample + per-tensor sums) to rounding."""
    g = golden("train_reference_c1.npz")
    sd = {k[5:]: torch.from_numpy(g[k].astype(np.float32)) for k in g.files if k.startswith("sd16.")}
    for k, v in sd.items():                                            # float16-representable: the fixture stores them exactly
        assert torch.equal(v, v.to(torch.float16).to(torch.float32)), k
    clusters = [torch.from_numpy(c) for c in _split(g["local"], g["offsets"])]
    assert len(clusters) == 20 and g["local"].shape == (4096, 3) and g["y"].shape == (4096, 3)
    model = models.QRegMLP(True, 512)
    model.load_state_dict(sd)
    pred, best_m, min_loss, hist = registration.train(torch.from_numpy(g["m"]), torch.from_numpy(g["y"]), model, clusters, rot="q", epochs=6)
    np.testing.assert_allclose(np.array(hist["loss"]), g["loss_hist"][:6], rtol=1e-6)
    np.testing.assert_allclose(np.array(hist["loss"]), g["e6_loss_hist"], rtol=1e-6)
    assert abs(min_loss - float(g["e6_min_loss"])) <= 1e-6 * min_loss
    np.testing.assert_allclose(best_m.detach().numpy(), g["e6_best_m"], atol=1e-6)
    np.testing.assert_allclose(np.concatenate(pred), g["e6_best_pred"], atol=1e-6)
    stride = int(g["sample_stride"])
    for k, v in model.state_dict().items():
        np.testing.assert_allclose(v.reshape(-1)[::stride].numpy(), g[f"e6.final_sample.{k}"], atol=2e-6, err_msg=k)
        assert abs(float(v.double().sum()) - float(g[f"e6.final_sum.{k}"])) < 1e-4, k
    # the best pose of a run is one of the poses it evaluated
    i = int(np.argmin(g["loss_hist"]))
    np.testing.assert_array_equal(g["pose_hist"][i], g["e300_best_m"])
    assert float(g["loss_hist"][i]) == float(g["e300_min_loss"])


def test_divergence_envelope_fixture_is_consistent(golden):
    """tests/golden/divergence_envelope_c1.npz (tests/measure/divergence_envelope.py cpu): the float32 oracle reproduces the reference
    trajectory exactly, permuted / float64 variants stay within the north star's 1e-5 up to N_e and then leave it by orders of magnitude."""
    e = golden("divergence_envelope_c1.npz")
    n_e = int(e["n_e"])
    assert float(e["oracle_vs_reference"].max()) == 0.0
    assert 3 <= n_e < 60
    assert float(np.maximum(e["envelope"], e["f64"])[: n_e + 1].max()) <= 1e-5
    assert float(np.maximum(e["envelope"], e["f64"])[n_e + 1]) > 1e-5
    assert float(e["envelope"][30:].max()) > 1e-3            # two correct float32 implementations are centimetres apart later on


def test_resample_matches_reference_and_live_sklearn(golden):
    g = golden("resample_reference.npz")
    local, labels = registration.resample_cluster(g["frame"], len(g["mats"]), g["mats"])
    np.testing.assert_array_equal(np.cumsum([0] + [len(c) for c in local]), g["offsets"])
    np.testing.assert_allclose(np.concatenate(local), g["local"], atol=1e-12)


def test_masked_icp_matches_reference(golden):
    g = golden("masked_icp_reference.npz")
    local = _split(g["local"], g["offsets"])
    world = _split(g["world_pred"], g["offsets"])
    w, m = icp.masked_icp(local, world, g["frame"], g["mats"])
    np.testing.assert_allclose(m, g["new_mats"], atol=1e-12)
    np.testing.assert_allclose(np.concatenate(w), g["new_world"], atol=1e-12)


def test_kabsch_known_motion_reflection_and_planar():
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(0)
    src = rng.normal(size=(100, 3))
    R, t = Rotation.random(random_state=1).as_matrix(), np.array([0.3, -0.2, 0.5])
    Tm = icp.kabsch(src, src @ R.T + t)
    np.testing.assert_allclose(Tm[:3, :3], R, atol=1e-12)
    np.testing.assert_allclose(Tm[:3, 3], t, atol=1e-12)
    refl = src * np.array([1, 1, -1])                      # det < 0 target: still a proper rotation
    assert abs(np.linalg.det(icp.kabsch(src, refl)[:3, :3]) - 1) < 1e-12
    planar = src * np.array([1, 1, 0])
    Tp = icp.kabsch(planar, planar @ R.T + t)
    np.testing.assert_allclose(planar @ Tp[:3, :3].T + Tp[:3, 3], planar @ R.T + t, atol=1e-12)
    np.testing.assert_array_equal(icp.kabsch(src[:0], src[:0]), np.eye(4))


@pytest.mark.parametrize("tag", ["small", "c1"])
def test_kmeans_matches_live_sklearn_golden(golden, tag):
    g = golden("kmeans_sklearn.npz")
    c, lab, inertia, n_iter = kmeans.k_means(g[f"{tag}_X"], g[f"{tag}_init"])
    np.testing.assert_array_equal(lab, g[f"{tag}_labels"])           # bit-exact assignments
    np.testing.assert_allclose(c, g[f"{tag}_centers"], atol=1e-13)
    assert abs(inertia - float(g[f"{tag}_inertia"])) < 1e-10 * max(1.0, inertia)


def test_kmeans_against_sklearn_live_random_and_duplicates():
    sk = pytest.importorskip("sklearn.cluster")
    rng = np.random.default_rng(2)
    for n, k in ((300, 5), (2000, 16), (64, 8)):
        X = rng.normal(size=(n, 3))
        if n == 64:
            X[32:] = X[:32]                                          # duplicates
        init = X[rng.choice(n, k, replace=False)] + 1e-3
        c, lab, inertia, _ = kmeans.k_means(X, init)
        c2, lab2, in2 = sk.k_means(X.copy(), init=init.copy(), n_clusters=k, n_init=1)
        np.testing.assert_array_equal(lab, lab2)
        np.testing.assert_allclose(c, c2, atol=1e-12)


def test_kmeans_empty_cluster_relocation_keeps_k_clusters():
    rng = np.random.default_rng(3)
    X = rng.normal(size=(200, 3))
    init = np.vstack([X[:3], [[50., 50, 50]]])                      # 4th seed owns nothing
    c, lab, _, _ = kmeans.k_means(X, init)
    assert len(np.unique(lab)) == 4


def test_chamfer_golden_and_dense_crosscheck(golden):
    g = golden("chamfer_l1.npz")
    x, y = g["x"], g["y"]
    dx, ix = chamfer.nn_l1(x, y)
    dy, iy = chamfer.nn_l1(y, x)
    np.testing.assert_array_equal(ix, g["ix"]); np.testing.assert_array_equal(iy, g["iy"])
    np.testing.assert_array_equal(dx, g["dx"]); np.testing.assert_array_equal(dy, g["dy"])
    xt = torch.from_numpy(x).requires_grad_(True)
    loss, _ = chamfer.chamfer_distance(xt[None], torch.from_numpy(y)[None], norm=1)
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 1e-7
    np.testing.assert_allclose(xt.grad.numpy(), g["grad"], atol=1e-9)
    dense, ix2, iy2 = chamfer.chamfer_l1_dense(torch.from_numpy(x), torch.from_numpy(y))
    np.testing.assert_array_equal(ix2.numpy(), ix); np.testing.assert_array_equal(iy2.numpy(), iy)
    # autograd through the dense form agrees except at exact zeros (pytorch3d's sign rule gives -1)
    xd = torch.from_numpy(x).requires_grad_(True)
    chamfer.chamfer_l1_dense(xd, torch.from_numpy(y))[0].backward()
    nz = np.abs(x[:, None, :] - y[None, :, :]).min(1).min(1) > 0
    np.testing.assert_allclose(xd.grad.numpy()[nz], xt.grad.numpy()[nz], atol=1e-7)


def test_chamfer_first_min_tie_break_and_ragged():
    x = np.zeros((3, 3), np.float32)
    y = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 0, 0]], np.float32)   # all at L1 distance 1
    d, i = chamfer.nn_l1(x, y)
    assert (i == 0).all() and (d == 1).all()
    d, i = chamfer.nn_l1(y[:1], x[:2])
    assert i[0] == 0


def test_fps_is_a_permutation_prefix_and_spreads():
    rng = np.random.default_rng(1)
    X = rng.normal(size=(500, 3))
    sel = kmeans.farthest_point_sample(X, 64)
    assert sel[0] == 0 and len(set(sel.tolist())) == 64
    d_sel = np.sqrt(((X[sel][:, None] - X[sel][None]) ** 2).sum(-1) + np.eye(64) * 1e9).min()
    d_rnd = np.sqrt(((X[:64][:, None] - X[:64][None]) ** 2).sum(-1) + np.eye(64) * 1e9).min()
    assert d_sel > d_rnd


# ------------------------------------------------------------------------------------------ N2: pose distance maps
@pytest.mark.parametrize("tag", ["a", "b"])
def test_coord_map_oracle_vs_reference_golden(golden, tag):
    """oracle.coord_map vs the reference CoordMap's own loops (coord_map.py:185-332 under shims)."""
    from oracle import coord_map as ocm
    g = golden("coord_map_reference.npz")
    M, bbox = g[f"{tag}.matrices"], float(g[f"{tag}.bounding_box"])
    coords = ocm.coords_from_matrices(M)
    np.testing.assert_allclose(coords, g[f"{tag}.coords"], atol=1e-12)
    assert abs(ocm.get_scale(coords) - float(g[f"{tag}.scale"])) < 1e-12
    for diff in (True, False):
        cmap, smap = ocm.coord_dist_map(M, bbox, diff)
        np.testing.assert_allclose(cmap, g[f"{tag}.diff{int(diff)}.map"], atol=1e-12)
        np.testing.assert_allclose(smap, g[f"{tag}.diff{int(diff)}.sum"], atol=1e-12)
    cmap, smap = ocm.coord_dist_map_legacy(coords)
    np.testing.assert_allclose(cmap, g[f"{tag}.legacy.map"], atol=1e-12)
    np.testing.assert_allclose(smap, g[f"{tag}.legacy.sum"], atol=1e-12)


def test_roma_restatement_vs_scipy_rotation():
    """The unpinned roma arithmetic against an independent implementation."""
    from scipy.spatial.transform import Rotation
    from oracle import coord_map as ocm
    rot = Rotation.random(200, random_state=3)
    R = rot.as_matrix()
    np.testing.assert_allclose(ocm.rotmat_to_rotvec(R), rot.as_rotvec(), atol=1e-12)
    q = ocm.rotmat_to_unitquat(R)
    qs = rot.as_quat()
    np.testing.assert_allclose(q * np.sign(q[:, 3:4]), qs * np.sign(qs[:, 3:4]), atol=1e-12)
    a, b = rot[:100], rot[100:]
    want = (a.inv() * b).magnitude()
    np.testing.assert_allclose(ocm.rotvec_geodesic_distance(a.as_rotvec(), b.as_rotvec()), want, atol=1e-9)
    np.testing.assert_allclose(ocm.rotmat_geodesic_distance(a.as_matrix(), b.as_matrix()), want, atol=1e-7)
    tiny = Rotation.from_rotvec(np.array([[1e-5, -2e-5, 3e-6], [0, 0, 0]]))
    np.testing.assert_allclose(ocm.rotmat_to_rotvec(tiny.as_matrix()), tiny.as_rotvec(), atol=1e-15)


# ------------------------------------------------------------------------------------------ N3: link refinement
def test_link_refine_oracle_vs_reference_golden(golden):
    """oracle.link.refine_links vs the reference refine_links_clusters run on disk (link.py:85-127)."""
    from oracle import link as olink
    g = golden("link_refine_reference.npz")
    T, dof = int(g["T"]), int(g["dof"])
    clusters = [[g[f"in.{t}.{i}"] for i in range(4)] for t in range(T)]
    moved = olink.refine_links(clusters, clusters[0], dof)
    for t in range(T):
        assert len(moved[t]) == dof + 1
        for i in range(dof + 1):
            np.testing.assert_allclose(moved[t][i], g[f"out.{t}.{i}"], atol=1e-12)


@pytest.mark.parametrize("tag,dtype", [("rot", np.float32), ("rot64", np.float64)])
def test_resample_rotated_poses_match_reference(golden, tag, dtype):
    """resample_cluster with ROTATED poses: the reference inverts each pose in its own dtype (float32 after train,
    mlp_reg.py:211,371; float64 after masked_icp, :326).  The oracle makes the same numpy call; across machines the
    float32 LAPACK inverse may differ in the last bits (BLAS build), hence 2e-7 on coordinates of ~0.1 m."""
    g = golden("resample_reference.npz")
    local, _ = registration.resample_cluster(g["frame"], 8, g["rot_mats"].astype(dtype))
    np.testing.assert_array_equal(np.cumsum([0] + [len(c) for c in local]), g[f"{tag}_offsets"])
    np.testing.assert_allclose(np.concatenate(local), g[f"{tag}_local"], rtol=0, atol=2e-7 if dtype == np.float32 else 1e-12)
    # the float32 inverse is visible: the two goldens differ by more than the float64 tolerance
    assert np.abs(g["rot_local"] - g["rot64_local"]).max() > 1e-9


def test_aabb_mask_indices_match_reference(golden):
    """G9: the exact mask membership of every cluster (cluster_icp.py:133-146) as the reference's masked_icp handed it
    to registration_icp."""
    g = golden("masked_icp_reference.npz")
    world = _split(g["world_pred"], g["offsets"])
    for c, w in enumerate(world):
        idx = np.nonzero(icp.aabb_mask(w, g["frame"], 1.2))[0]
        np.testing.assert_array_equal(idx, g["mask_idx"][g["mask_offsets"][c]:g["mask_offsets"][c + 1]])


def test_oracle_icp_search_dense_and_c_forms_agree():
    """oracle/icp.py:_correspond switches from the dense numpy form to creg_oracle.c's loop above 4e6 pairs (whole-cloud ICP,
    Sim/evaluation.py:358-362): identical nearest indices (first minimum, ties included: lattice coordinates) and distances."""
    from oracle import icp
    rng = np.random.default_rng(3)
    a = np.round(rng.uniform(-1, 1, (600, 3)) * 8) / 8            # a coarse lattice: many exactly equal distances
    b = np.round(rng.uniform(-1, 1, (800, 3)) * 8) / 8
    b[10] = b[500]
    d2 = ((a[:, None, :] - b[None, :, :]) ** 2).sum(-1)
    j = d2.argmin(1)
    dc, jc = icp._nn_l2_c(a, b)
    assert (j == jc).all() and (d2[np.arange(len(a)), j] == dc).all()
    old = icp._DENSE_PAIRS
    try:
        src = b[:300] + 0.01
        T0, f0, r0, n0 = icp.registration_icp(src, b, 1.0, np.eye(4), 50)
        icp._DENSE_PAIRS = 0
        T1, f1, r1, n1 = icp.registration_icp(src, b, 1.0, np.eye(4), 50)
    finally:
        icp._DENSE_PAIRS = old
    assert n0 == n1 and f0 == f1 and r0 == r1 and (T0 == T1).all()


@pytest.mark.parametrize("shape", ["allegro", "franka"])
def test_divergence_envelope_fixtures_of_the_other_shapes_are_consistent(golden, shape):
    """tests/golden/divergence_envelope_{allegro,franka}.npz against train_reference_{shape}.npz: the float32 oracle reproduced the
    reference's trajectory at every pinned epoch (bit for bit: same torch build, same op sequence), the permuted runs start inside
    1e-5, and the fixture carries what the GPU test reads."""
    e = golden(f"divergence_envelope_{shape}.npz")
    g = golden(f"train_reference_{shape}.npz")
    assert float(e["oracle_vs_reference"].max()) == 0.0
    assert e["envelope"].shape == (300,) and e["f64"].shape == (300,)
    assert float(e["envelope"][1]) < 1e-5 and int(e["n_e_perm"]) >= 1
    assert float(e["min_loss_rel_envelope"]) > 0 and float(e["best_pose_envelope"]) > 0
    assert len(g["loss_hist"]) == 300 and g["pose_hist_sel"].shape[0] == len(g["pose_epochs"])
