"""GPU parity: every HIP kernel behind the C ABI against the CPU oracle / golden fixtures.

Bar: bit-exact for indices, labels and the integer-order-independent quantities (L1 distances,
fp64 k-means distances); fp32 poses / losses within the tolerances written at each assert.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    from autourdf_amd import _lib
    _lib.load()                      # raises if libcreg.so is missing or the device is not gfx950
    return torch.device("cuda:0")


def _split(flat, offsets):
    return [flat[offsets[i]:offsets[i + 1]] for i in range(len(offsets) - 1)]


def _cuda(a, dev, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    return (t.to(dtype) if dtype else t).to(dev)


# ------------------------------------------------------------------------------------------ K1
@pytest.mark.parametrize("nx,ny", [(1, 1), (3, 70), (257, 64), (1000, 1333), (4096, 4096), (5000, 4096)])
def test_nn_l1_bit_exact_vs_oracle(dev, nx, ny):
    from autourdf_amd import ops
    from oracle import chamfer
    rng = np.random.default_rng(nx * 7 + ny)
    x = rng.normal(size=(nx, 3)).astype(np.float32)
    y = rng.normal(size=(ny, 3)).astype(np.float32)
    dx, ix, dy, iy = ops.nn_l1_bidir(_cuda(x, dev), _cuda(y, dev))
    odx, oix = chamfer.nn_l1(x, y)
    ody, oiy = chamfer.nn_l1(y, x)
    np.testing.assert_array_equal(ix.cpu().numpy(), oix)
    np.testing.assert_array_equal(iy.cpu().numpy(), oiy)
    np.testing.assert_array_equal(dx.cpu().numpy(), odx)          # same fp32 op order: bit-exact
    np.testing.assert_array_equal(dy.cpu().numpy(), ody)


def test_nn_l1_ties_and_duplicates_take_first_index(dev):
    from autourdf_amd import ops
    from oracle import chamfer
    rng = np.random.default_rng(0)
    base = rng.integers(-3, 4, size=(300, 3)).astype(np.float32)      # lattice -> many exact ties
    x, y = base[:200], np.concatenate([base[100:], base[100:150]])
    dx, ix, dy, iy = ops.nn_l1_bidir(_cuda(x, dev), _cuda(y, dev))
    np.testing.assert_array_equal(ix.cpu().numpy(), chamfer.nn_l1(x, y)[1])
    np.testing.assert_array_equal(iy.cpu().numpy(), chamfer.nn_l1(y, x)[1])


def test_nn_l1_randomised_sizes_and_ties_vs_oracle(dev):
    """40 seeded random shapes (1..9000 points per side, crossing the 4096-target LDS chunk and the 256 / 512 padding
    boundaries), half of them on a coarse lattice so that exact ties and duplicates are everywhere: indices and
    distances bit-exact against the C oracle."""
    from autourdf_amd import ops
    from oracle import chamfer
    rng = np.random.default_rng(2024)
    edge = [1, 2, 63, 64, 65, 255, 256, 257, 511, 512, 513, 4095, 4096, 4097, 8191, 8192, 8193]
    for trial in range(40):
        nx = int(rng.choice(edge)) if rng.random() < 0.4 else int(rng.integers(1, 9000))
        ny = int(rng.choice(edge)) if rng.random() < 0.4 else int(rng.integers(1, 9000))
        if trial % 2:
            x = rng.integers(-4, 5, size=(nx, 3)).astype(np.float32) / 4
            y = rng.integers(-4, 5, size=(ny, 3)).astype(np.float32) / 4
        else:
            x = rng.normal(size=(nx, 3)).astype(np.float32)
            y = (rng.normal(size=(ny, 3)) * 0.5 + 0.1).astype(np.float32)
        dx, ix, dy, iy = ops.nn_l1_bidir(_cuda(x, dev), _cuda(y, dev))
        odx, oix = chamfer.nn_l1(x, y)
        ody, oiy = chamfer.nn_l1(y, x)
        assert np.array_equal(ix.cpu().numpy(), oix) and np.array_equal(iy.cpu().numpy(), oiy), (trial, nx, ny)
        assert np.array_equal(dx.cpu().numpy(), odx) and np.array_equal(dy.cpu().numpy(), ody), (trial, nx, ny)


def test_chamfer_loss_and_grad_vs_golden(dev, golden):
    from autourdf_amd import ops
    g = golden("chamfer_l1.npz")
    x = _cuda(g["x"], dev).requires_grad_(True)
    loss, _ = ops.chamfer_distance(x[None], _cuda(g["y"], dev)[None], norm=1)
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) <= 1e-6 * abs(float(g["loss"]))      # rtol 1e-6
    np.testing.assert_allclose(x.grad.cpu().numpy(), g["grad"], rtol=1e-5, atol=1e-9)


def test_chamfer_full_size_properties(dev):
    """N = 16384 (franka config): symmetric role swap and permutation invariance of the indices."""
    from autourdf_amd import ops
    rng = np.random.default_rng(5)
    x = rng.normal(size=(16384, 3)).astype(np.float32)
    y = rng.normal(size=(16384, 3)).astype(np.float32)
    dx, ix, dy, iy = ops.nn_l1_bidir(_cuda(x, dev), _cuda(y, dev))
    dy2, iy2, dx2, ix2 = ops.nn_l1_bidir(_cuda(y, dev), _cuda(x, dev))
    assert torch.equal(ix, ix2) and torch.equal(iy, iy2) and torch.equal(dx, dx2) and torch.equal(dy, dy2)
    perm = rng.permutation(16384)
    dxp, ixp, _, _ = ops.nn_l1_bidir(_cuda(x, dev), _cuda(y[perm], dev))
    assert torch.equal(dxp, dx)                                   # distances do not depend on target order
    d_check = np.abs(x - y[perm][ixp.cpu().numpy()]).astype(np.float32)
    np.testing.assert_array_equal((d_check[:, 0] + d_check[:, 1]) + d_check[:, 2], dx.cpu().numpy())


# ------------------------------------------------------------------------------------------ K3
def test_cluster_transform_fwd_bwd(dev, golden):
    from autourdf_amd import ops
    g = golden("calculate_pc.npz")
    M = _cuda(g["mats"], dev).requires_grad_(True)
    pts, off = _cuda(g["local"], dev), _cuda(g["offsets"], dev, torch.int32)
    out = ops.cluster_transform(pts, off, M)
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["world"], atol=2e-7)
    w = torch.linspace(-1, 1, out.numel(), device=dev).reshape(out.shape)
    (out * w).sum().backward()
    Mc = torch.from_numpy(g["mats"]).requires_grad_(True)
    ref = torch.cat([c @ Mc[i, :3, :3].T + Mc[i, :3, 3] for i, c in
                     enumerate(_split(torch.from_numpy(g["local"]), g["offsets"]))])
    (ref * w.cpu()).sum().backward()
    np.testing.assert_allclose(M.grad.cpu().numpy()[:, :3], Mc.grad.numpy()[:, :3], rtol=1e-5, atol=1e-5)


def test_cluster_transform_empty_cluster(dev):
    from autourdf_amd import ops
    pts = torch.randn(10, 3, device=dev)
    off = torch.tensor([0, 4, 4, 10], dtype=torch.int32, device=dev)     # middle cluster empty
    M = torch.eye(4, device=dev).repeat(3, 1, 1).contiguous()
    M[:, :3, 3] = torch.tensor([[1., 0, 0], [0, 5, 0], [0, 0, 2]], device=dev)
    out = ops.cluster_transform(pts, off, M)
    exp = pts.clone(); exp[:4, 0] += 1; exp[4:, 2] += 2
    assert torch.allclose(out, exp)


# ------------------------------------------------------------------------------------------ K2
@pytest.mark.parametrize("tag", ["small", "c1"])
@pytest.mark.parametrize("mfma", [False, True])
def test_kmeans_labels_bit_exact_vs_sklearn_golden(dev, golden, tag, mfma):
    from autourdf_amd import ops
    g = golden("kmeans_sklearn.npz")
    c, lab, inertia, n_iter = ops.kmeans_lloyd(_cuda(g[f"{tag}_X"], dev), _cuda(g[f"{tag}_init"], dev), use_mfma=mfma)
    np.testing.assert_array_equal(lab.cpu().numpy(), g[f"{tag}_labels"])
    np.testing.assert_allclose(c.cpu().numpy(), g[f"{tag}_centers"], atol=1e-12)
    assert abs(inertia.item() - float(g[f"{tag}_inertia"])) < 1e-9 * max(1.0, float(g[f"{tag}_inertia"]))


@pytest.mark.parametrize("n,k", [(64, 8), (17, 1), (4096, 20), (16384, 40), (5000, 200), (262144, 128)])
def test_kmeans_assign_valu_mfma_oracle_identical(dev, n, k):
    from autourdf_amd import ops
    from oracle import kmeans
    rng = np.random.default_rng(n + k)
    X = rng.normal(size=(n, 3))
    C = X[rng.choice(n, k, replace=False)] + 1e-3
    if n == 64:
        C[3] = C[5]                                   # duplicate centre: first index must win
    a = ops.kmeans_assign(_cuda(X, dev), _cuda(C, dev), use_mfma=False).cpu().numpy()
    b = ops.kmeans_assign(_cuda(X, dev), _cuda(C, dev), use_mfma=True).cpu().numpy()
    np.testing.assert_array_equal(a, kmeans.assign(X, C))
    np.testing.assert_array_equal(b, a)


def test_kmeans_full_run_vs_oracle_random_and_empty_cluster(dev):
    from autourdf_amd import ops
    from oracle import kmeans
    rng = np.random.default_rng(3)
    X = rng.normal(size=(2000, 3))
    init = X[rng.choice(2000, 16, replace=False)] + 1e-3
    c, lab, inertia, n_iter = ops.kmeans_lloyd(_cuda(X, dev), _cuda(init, dev))
    oc, olab, oin, oit = kmeans.k_means(X, init)
    np.testing.assert_array_equal(lab.cpu().numpy(), olab)
    assert n_iter.item() == oit
    np.testing.assert_allclose(c.cpu().numpy(), oc, atol=1e-12)
    init2 = np.vstack([X[:3], [[50., 50, 50]]])       # 4th seed owns no point -> relocation path
    c, lab, _, _ = ops.kmeans_lloyd(_cuda(X[:200], dev), _cuda(init2, dev))
    oc, olab, _, _ = kmeans.k_means(X[:200], init2)
    np.testing.assert_array_equal(lab.cpu().numpy(), olab)
    assert len(np.unique(lab.cpu().numpy())) == 4


@pytest.mark.parametrize("n,k,kind", [(17, 3, "normal"), (300, 40, "normal"), (300, 1, "normal"), (5000, 20, "surface"),
                                      (20000, 128, "surface"), (20000, 300, "normal"), (70000, 64, "dupes"),
                                      (70000, 128, "lattice"), (150000, 33, "surface"), (300000, 128, "surface")])
def test_kmeans_pruned_persistent_form_identical_to_full_sweep(dev, n, k, kind):
    """creg_kmeans_lloyd_f64's VALU form (spatially sorted copy with dummy rows, centres pruned per workgroup and per wave, the
    persistent kernel with its in-launch hand-offs, relocation launches in between) against the matrix-core form (caller's
    order, every centre for every point, one launch per iteration): labels, iteration count, centres and inertia must be
    IDENTICAL -- the pruning may never change an argmin, the exact integer sums make the M-step order-independent, and the
    inertia is summed in the caller's order by both.  Shapes: n < 64, k = 1, k > 256 (several compaction passes), exact
    duplicates and a lattice (ties, points on cell borders), seeds far from the data (empty clusters -> relocation launches in
    the middle of a persistent run), sizes on both sides of the points-per-thread switches."""
    from autourdf_amd import ops
    rng = np.random.default_rng(n * 7 + k)
    if kind == "normal":
        X = rng.normal(size=(n, 3)) * np.array([1.0, 0.5, 0.25])
    elif kind == "surface":                          # a folded sheet: what a depth-camera frame looks like to the spatial sort
        u, v = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
        X = np.stack([u, v, 0.3 * np.sin(3 * u) * np.cos(2 * v) + 0.01 * rng.normal(size=n)], 1)
    elif kind == "dupes":
        base = rng.normal(size=(n // 7, 3))
        X = base[rng.integers(0, len(base), n)]
    else:                                            # lattice
        g = np.stack(np.meshgrid(*[np.arange(42)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float64)
        X = g[rng.choice(len(g), n, replace=False)] / 8.0
    init = X[rng.choice(n, k, replace=False)].copy()
    if k >= 8:
        init[: k // 4] += 40.0                       # a quarter of the seeds own nothing at first
    Xd, Id = _cuda(X, dev), _cuda(init, dev)
    a = ops.kmeans_lloyd(Xd, Id, max_iter=60)
    b = ops.kmeans_lloyd(Xd, Id, max_iter=60, use_mfma=True)
    assert a[3].item() == b[3].item()
    assert torch.equal(a[1], b[1])
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2])
    assert len(torch.unique(a[1])) == k or kind == "dupes"
    again = ops.kmeans_lloyd(Xd, Id, max_iter=60)
    assert torch.equal(a[1], again[1]) and torch.equal(a[0], again[0])


def test_kmeans_persistent_kernel_gives_up_and_the_call_starts_over(dev, tmp_path):
    """A workgroup of the persistent Lloyd kernel that waits too long sets `abort`; the host then discards the attempt and runs
    the call again with one launch per iteration.  Forced here with CREG_KM_SPIN_LIMIT=1 in a child process (the knob is read
    once per process): same labels, centres, inertia and iteration count as the unforced run in this process."""
    import subprocess
    import sys
    from autourdf_amd import ops
    rng = np.random.default_rng(11)
    u, v = rng.uniform(-1, 1, 40000), rng.uniform(-1, 1, 40000)
    X = np.stack([u, v, 0.2 * np.sin(4 * u) + 0.01 * rng.normal(size=40000)], 1)
    init = X[rng.choice(40000, 48, replace=False)]
    c, lab, inertia, n_iter = ops.kmeans_lloyd(_cuda(X, dev), _cuda(init, dev), max_iter=40)
    np.savez(tmp_path / "in.npz", X=X, init=init)
    code = ("import numpy as np, torch, sys; sys.path.insert(0, %r); from autourdf_amd import ops; g = np.load(%r); "
            "o = ops.kmeans_lloyd(torch.as_tensor(g['X'], device='cuda'), torch.as_tensor(g['init'], device='cuda'), max_iter=40); "
            "np.savez(%r, c=o[0].cpu().numpy(), lab=o[1].cpu().numpy(), inertia=o[2].cpu().numpy(), n_iter=o[3].cpu().numpy())"
            % (ROOT, str(tmp_path / "in.npz"), str(tmp_path / "out.npz")))
    env = dict(os.environ, CREG_KM_SPIN_LIMIT="1")
    subprocess.run([sys.executable, "-c", code], check=True, env=env, timeout=200)
    o = np.load(tmp_path / "out.npz")
    np.testing.assert_array_equal(o["lab"], lab.cpu().numpy())
    np.testing.assert_array_equal(o["c"], c.cpu().numpy())
    np.testing.assert_array_equal(o["inertia"], inertia.cpu().numpy())
    assert int(o["n_iter"][0]) == n_iter.item()


def test_resample_group_to_local_vs_reference_golden(dev, golden):
    from autourdf_amd import ops
    g = golden("resample_reference.npz")
    X, M = _cuda(g["frame"], dev), _cuda(g["mats"].astype(np.float64), dev)
    _, lab, _, _ = ops.kmeans_lloyd(X, M[:, :3, 3].contiguous())
    local, off = ops.group_to_local(X, lab, M)
    np.testing.assert_array_equal(off.cpu().numpy(), g["offsets"])
    np.testing.assert_allclose(local.cpu().numpy(), g["local"], atol=1e-12)


@pytest.mark.parametrize("tag,dtype,tol", [("rot", np.float32, 2e-7), ("rot64", np.float64, 1e-12)])
def test_resample_cluster_rotated_poses_vs_reference_golden(dev, golden, tag, dtype, tol):
    """The drop-in resample_cluster with rotated poses against the reference function's output: the pose inverse is
    the reference's own host call in the pose's dtype (float32 LAPACK after train: the last bits depend on the BLAS
    build of the host, hence 2e-7; float64 after masked_icp), the labels and the change of frame run on the GPU."""
    from autourdf_amd import mlp_reg
    from autourdf_amd.cluster_icp import Segments
    g = golden("resample_reference.npz")
    seg = Segments.from_arrays([g["frame"], g["frame"]])
    local = mlp_reg.resample_cluster(seg, 1, 8, g["rot_mats"].astype(dtype))
    np.testing.assert_array_equal(np.cumsum([0] + [len(c) for c in local]), g[f"{tag}_offsets"])
    assert all(c.dtype == np.float64 for c in local)
    np.testing.assert_allclose(np.concatenate(local), g[f"{tag}_local"], rtol=0, atol=tol)


def test_segments_k_means_cluster_vs_reference_golden(dev, golden):
    """Frame-0 segmentation: reference Segments.k_means_cluster (cluster_icp.py:47-107) under np.random.seed against
    the drop-in with the same seed -- sklearn's k-means++ draw sequence on the host, Lloyd + grouping on the GPU."""
    from autourdf_amd.cluster_icp import Segments
    g = golden("segments_reference.npz")
    seg = Segments.from_arrays([g["frame"]])
    np.random.seed(int(g["seed"]))
    seg.k_means_cluster(0, 8)
    np.testing.assert_array_equal(np.cumsum([0] + [len(c) for c in seg.init_segment_list]), g["offsets"])
    np.testing.assert_allclose(np.array(seg.init_matrix_list), g["matrices"], rtol=0, atol=1e-14)
    np.testing.assert_allclose(np.array(seg.init_coord_list), g["coords"], rtol=0, atol=1e-14)
    np.testing.assert_allclose(np.concatenate(seg.init_segment_list), g["segments"], rtol=0, atol=1e-14)
    seg2 = Segments.from_arrays([g["frame"]])                       # explicit seed = the same draw
    seg2.k_means_cluster(0, 8, seed=int(g["seed"]))
    np.testing.assert_array_equal(np.array(seg2.init_matrix_list), np.array(seg.init_matrix_list))


# ------------------------------------------------------------------------------------------ K5
def test_dq_rows_vs_reference_golden(dev, golden):
    from autourdf_amd import ops
    g = golden("dq_reference.npz")
    t = lambda k: _cuda(g[f"f32_{k}"], dev)
    close = lambda a, k, tol=2e-6: np.testing.assert_allclose(a.cpu().numpy(), g[f"f32_{k}"], rtol=0, atol=tol)
    close(ops.se3_to_dq(t("M")), "dq")
    close(ops.dq_to_se3(t("noisy")), "to_transform")
    q, tr = ops.dq_to_quat_trans(t("noisy"))
    close(q, "qt_q"); close(tr, "qt_t")
    close(ops.dq_multiply(t("dq"), t("dq_b")), "mul")
    close(ops.dq_invert(t("noisy")), "inv")
    close(ops.quat_trans_to_dq(t("dq")[:, :4].contiguous(), t("M")[:, :3, 3].contiguous()), "from_qt")
    close(ops.quat_to_matrix(t("noisy")[:, :4].contiguous()), "rt_R")
    R = t("M")[:, :3, :3].contiguous()
    close(ops.matrix_to_quat(R), "dq", 2e-6) if False else None
    np.testing.assert_allclose(ops.matrix_to_quat(R).cpu().numpy(), g["f32_dq"][:, :4], atol=2e-6)


def test_dq_func_dropin_all_eleven_functions_vs_reference_golden(dev, golden):
    """Every function of the drop-in module autourdf_amd.dq_func (same 11 names as PointCloud/dq_func.py:4-257) against
    EVERY key of dq_reference.npz (minted by the reference module itself): SURVEY 8 rows D1-D10 on the GPU, not only
    through the oracle."""
    from autourdf_amd import dq_func as D
    g = golden("dq_reference.npz")
    t = lambda k: _cuda(g[f"f32_{k}"], dev)
    close = lambda a, k, tol=2e-6: np.testing.assert_allclose(a.detach().cpu().numpy(), g[f"f32_{k}"], rtol=0, atol=tol)
    M, dq, dq_b, noisy = t("M"), t("dq"), t("dq_b"), t("noisy")
    R, tr = M[:, :3, :3], M[:, :3, 3]                                       # non-contiguous views, as callers hand them
    close(D.transform_from_rot_trans(R, tr), "assemble", 0)                  # D1: exact (assembly only)
    close(D.quaternion_conjugate(dq[:, :4]), "conj", 0)                      # D2: sign flips only
    close(D.quat_trans_to_dualquat(dq[:, :4], tr), "from_qt")                # D3
    close(D.rot_trans_to_dualquat(R, tr), "from_rt")                         # D4
    close(D.transform_to_dualquat(M), "dq")                                  # D5
    q, t2 = D.dualquat_to_quat_trans(noisy)                                  # D6 (incl. the reference's "returns the product" quirk)
    close(q, "qt_q"); close(t2, "qt_t")
    Rr, tt = D.dualquat_to_rot_trans(noisy)                                  # D7
    close(Rr, "rt_R"); close(tt, "rt_t")
    close(D.dualquat_to_transform(noisy), "to_transform")                    # D8
    close(D.dualquat_multiply(dq, dq_b), "mul")                              # D9
    close(D.dualquat_invert(noisy), "inv")
    close(D.point_to_dualquat(tr), "point", 0)                               # D10
    # leading batch dimensions are kept (the reference functions broadcast over them)
    assert D.transform_to_dualquat(M.reshape(8, 8, 4, 4)).shape == (8, 8, 8)
    assert D.dualquat_to_transform(noisy.reshape(4, 16, 8)).shape == (4, 16, 4, 4)


def test_dq_func_dropins_are_transparent_to_autograd(dev):
    """Like the reference's plain-torch functions (dq_func.py:47-257), every drop-in passes gradients: forward = HIP kernel,
    backward vs autograd through the oracle (fp64) on the same inputs."""
    from autourdf_amd import dq_func as D
    from oracle import dq as odq
    from scipy.spatial.transform import Rotation
    g = torch.Generator().manual_seed(8)
    M = torch.eye(4).repeat(32, 1, 1)
    M[:, :3, :3] = torch.from_numpy(Rotation.random(32, random_state=5).as_matrix()).float()
    M[:, :3, 3] = torch.randn(32, 3, generator=g)
    d1 = odq.transform_to_dualquat(M) + 0.05 * torch.randn(32, 8, generator=g)
    d2 = odq.transform_to_dualquat(M.flip(0)) + 0.05 * torch.randn(32, 8, generator=g)
    cases = [(D.transform_to_dualquat, odq.transform_to_dualquat, (M,)),
             (D.rot_trans_to_dualquat, odq.rot_trans_to_dualquat, (M[:, :3, :3].clone(), M[:, :3, 3].clone())),
             (D.quat_trans_to_dualquat, odq.quat_trans_to_dualquat, (d1[:, :4].clone(), M[:, :3, 3].clone())),
             (D.dualquat_to_quat_trans, odq.dualquat_to_quat_trans, (d1,)),
             (D.dualquat_to_rot_trans, odq.dualquat_to_rot_trans, (d1,)),
             (D.dualquat_multiply, odq.dualquat_multiply, (d1, d2)),
             (D.dualquat_invert, odq.dualquat_invert, (d1,))]
    for mine, ref, ins in cases:
        a = [x.clone().to(dev).requires_grad_(True) for x in ins]
        b = [x.clone().double().requires_grad_(True) for x in ins]
        oa, ob = mine(*a), ref(*b)
        oa, ob = (oa if isinstance(oa, tuple) else (oa,)), (ob if isinstance(ob, tuple) else (ob,))
        w = [torch.randn(y.shape, generator=g) for y in ob]
        sum((x * ww.to(dev)).sum() for x, ww in zip(oa, w)).backward()
        sum((y * ww.double()).sum() for y, ww in zip(ob, w)).backward()
        for x, y in zip(a, b):
            np.testing.assert_allclose(x.grad.cpu().numpy(), y.grad.numpy(), rtol=2e-4, atol=2e-5, err_msg=mine.__name__)


def test_dq_to_se3_backward_vs_autograd(dev):
    from autourdf_amd import ops
    from oracle import dq as odq
    gen = torch.Generator().manual_seed(0)
    d = torch.randn(40, 8, generator=gen)
    d[:, :4] = torch.nn.functional.normalize(d[:, :4], dim=1) * (1 + 0.1 * torch.randn(40, 1, generator=gen))
    gM = torch.randn(40, 4, 4, generator=gen)
    dd = d.double().requires_grad_(True)
    (odq.dualquat_to_transform(dd) * gM.double()).sum().backward()
    got = ops.dq_to_se3_bwd(d.to(dev), gM.to(dev))
    np.testing.assert_allclose(got.cpu().numpy(), dd.grad.numpy(), rtol=2e-5, atol=2e-5)


# ------------------------------------------------------------------------------------------ K4
def _deg_cases():
    """Correspondence sets on which a quaternion (Horn) fit and an SVD (Umeyama) fit could part ways:
    name -> (source (n,3), target (m,3), threshold, compare poses?)."""
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(5)
    R = Rotation.from_rotvec([0.2, -0.15, 0.25]).as_matrix()
    t = np.array([0.03, -0.02, 0.04])
    move = lambda p: p @ R.T + t
    plane = np.c_[rng.uniform(-1, 1, (60, 2)), np.zeros(60)]                  # rank-2 covariance
    line = np.outer(np.linspace(-1, 1, 40) + rng.normal(scale=0.01, size=40), [0.6, 0.64, 0.48])   # rank 1: the spin about the line is free
    two = np.array([[0.0, 0.0, 0.0], [0.5, 0.1, -0.2]])
    blob = rng.normal(scale=0.5, size=(80, 3))
    return {"planar": (plane, move(plane), 1.0, True),
            "collinear": (line, move(line), 1.0, False),
            "two_pairs": (two, move(two), 1.0, False),
            "no_pairs": (blob, blob + 10.0, 0.05, True),                      # nothing within the threshold: the pose stays
            "one_pair": (blob[:1], move(blob[:1]), 1.0, False),
            "mirrored": (blob, blob * np.array([1.0, 1.0, -1.0]) + t, 5.0, True)}   # det < 0 optimum: the best PROPER rotation


@pytest.mark.parametrize("name", ["planar", "collinear", "two_pairs", "no_pairs", "one_pair", "mirrored"])
def test_icp_fit_degenerate_correspondences_vs_oracle(dev, name):
    """K4's closed-form fit is Horn's quaternion (Newton on the characteristic polynomial + adjugate, Jacobi fallback),
    the oracle's is open3d's Umeyama / SVD with the reflection fix.  On degenerate correspondence sets both must land on a
    minimiser: the MOVED sources always agree; the pose itself where it is unique."""
    from autourdf_amd import ops
    from oracle import icp as oicp
    src, tgt, th, unique = _deg_cases()[name]
    init = np.eye(4)[None]
    T, moved, n_it = ops.icp_p2p(torch.tensor(src, device=dev), torch.tensor([0, len(src)], dtype=torch.int32, device=dev),
                                 torch.tensor(tgt, device=dev), torch.tensor([0, len(tgt)], dtype=torch.int32, device=dev),
                                 torch.tensor(init, device=dev), th=th, max_iteration=200)
    oT, fit, rmse, o_it = oicp.registration_icp(src, tgt, th, init[0], max_iteration=200)
    T, moved = T.cpu().numpy()[0], moved.cpu().numpy()
    assert np.isfinite(T).all() and abs(np.linalg.det(T[:3, :3]) - 1.0) < 1e-9
    np.testing.assert_allclose(T[:3, :3] @ T[:3, :3].T, np.eye(3), atol=1e-9)
    np.testing.assert_allclose(moved, src @ oT[:3, :3].T + oT[:3, 3], atol=1e-7)
    if unique:
        np.testing.assert_allclose(T, oT, atol=1e-7)
        assert int(n_it.cpu()[0]) == o_it


def test_kabsch_standalone_g5_cases_vs_oracle(dev):
    """creg_kabsch_f64 (golden class G5): known motion, weights, a mirrored target (reflection fix), a planar
    set, a segment without pairs, and the non-unique cases (collinear, two pairs) through the residual they must reach."""
    from scipy.spatial.transform import Rotation
    from autourdf_amd import ops
    from oracle import icp as oicp
    rng = np.random.default_rng(11)
    R, t = Rotation.from_rotvec([0.7, -0.4, 1.1]).as_matrix(), np.array([0.3, -0.2, 0.5])
    blob = rng.normal(size=(300, 3))
    plane = np.c_[rng.uniform(-1, 1, (80, 2)), np.zeros(80)]
    line = np.outer(np.linspace(-1, 1, 50), [0.6, 0.64, 0.48])
    segs = [(blob, blob @ R.T + t, None),                                             # exact motion
            (blob, blob @ R.T + t + 0.05 * rng.normal(size=blob.shape), rng.uniform(0, 2, 300)),   # noisy + weights
            (blob, blob * np.array([1, 1, -1.0]), None),                              # mirrored target: det < 0 optimum
            (plane, plane @ R.T + t, None),                                           # rank-2 covariance
            (np.zeros((0, 3)), np.zeros((0, 3)), None),                               # no pairs
            (line, line @ R.T + t, None),                                             # rank 1: a minimiser, not THE minimiser
            (blob[:2], blob[:2] @ R.T + t, None)]
    src = np.concatenate([a for a, _, _ in segs]); dst = np.concatenate([b for _, b, _ in segs])
    w = np.concatenate([np.ones(len(a)) if ww is None else ww for a, _, ww in segs])
    off = np.cumsum([0] + [len(a) for a, _, _ in segs]).astype(np.int32)
    T = ops.kabsch(torch.tensor(src, device=dev), torch.tensor(dst, device=dev), torch.tensor(off, device=dev),
                   torch.tensor(w, device=dev)).cpu().numpy()
    T1 = ops.kabsch(torch.tensor(src, device=dev), torch.tensor(dst, device=dev), torch.tensor(off, device=dev)).cpu().numpy()
    for i, (a, b, ww) in enumerate(segs):
        o = oicp.kabsch(a, b, ww)
        assert abs(np.linalg.det(T[i][:3, :3]) - 1) < 1e-12
        np.testing.assert_allclose(T[i][:3, :3] @ T[i][:3, :3].T, np.eye(3), atol=1e-12)
        res = lambda M: float((((a @ M[:3, :3].T + M[:3, 3]) - b) ** 2 * (1.0 if ww is None else ww[:, None])).sum())
        assert res(T[i]) <= res(o) * (1 + 1e-9) + 1e-20                               # never worse than the SVD fit
        if i < 5:
            np.testing.assert_allclose(T[i], o, atol=1e-10)
        if ww is None:
            np.testing.assert_array_equal(T1[i], T[i])                                 # weights of one == no weights
    np.testing.assert_allclose(T[0][:3, :3], R, atol=1e-12); np.testing.assert_allclose(T[0][:3, 3], t, atol=1e-12)
    np.testing.assert_array_equal(T[4], np.eye(4))


def test_masked_icp_ori_keeps_translation_vs_oracle(dev, golden):
    """ori=True (cluster_icp.py:161-163): the ICP rotation with the INPUT translation, on the inputs of the reference-minted
    masked_icp golden (every box there holds enough targets for a unique fit: a box with two targets makes the rotation
    about their line arbitrary, and Horn's and Umeyama's arbitrary choices differ -- test_icp_fit_degenerate_... covers those)."""
    from autourdf_amd.cluster_icp import masked_icp
    from oracle import icp as oicp
    g = golden("masked_icp_reference.npz")
    local, world = _split(g["local"], g["offsets"]), _split(g["world_pred"], g["offsets"])
    counts = [int(oicp.aabb_mask(wc, g["frame"], 1.2).sum()) for wc in world]
    assert all(c == 0 or c >= 10 for c in counts), counts          # (an empty box leaves the pose alone: unique too)
    w, m = masked_icp(local, world, g["frame"], g["mats"], ori=True)
    ow, om = oicp.masked_icp(local, world, g["frame"], g["mats"], ori=True)
    np.testing.assert_allclose(m[:, :3, 3], g["mats"][:, :3, 3].astype(np.float64), atol=0)     # translations untouched
    np.testing.assert_allclose(m, om, atol=1e-8)
    np.testing.assert_allclose(np.concatenate(w), np.concatenate(ow), atol=1e-8)
    np.testing.assert_allclose(m[:, :3, :3], g["new_mats"][:, :3, :3], atol=1e-8)               # the reference's rotations (ori=False golden)
    assert np.abs(m[:, :3, :3] - g["mats"][:, :3, :3]).max() > 1e-4                             # ... and they did move


def test_masked_icp_vs_reference_golden(dev, golden):
    from autourdf_amd.cluster_icp import masked_icp
    g = golden("masked_icp_reference.npz")
    local = _split(g["local"], g["offsets"])
    world = _split(g["world_pred"], g["offsets"])
    w, m = masked_icp(local, world, g["frame"], g["mats"])
    np.testing.assert_allclose(m, g["new_mats"], atol=1e-8)        # poses: 1e-5 demanded, 1e-8 reached
    np.testing.assert_allclose(np.concatenate(w), g["new_world"], atol=1e-8)


def test_aabb_mask_indices_vs_reference_golden(dev, golden):
    """G9: exact mask membership per cluster -- float32 box scaled about its centre, strict inequalities
    (cluster_icp.py:133-146) -- against what the reference's masked_icp handed to registration_icp."""
    from autourdf_amd import ops
    g = golden("masked_icp_reference.npz")
    off = torch.as_tensor(g["offsets"].astype(np.int32), device=dev)
    idx, cnt, boxes = ops.aabb_mask(_cuda(g["world_pred"], dev), off, _cuda(g["frame"], dev))
    cnt_h, idx_h = cnt.cpu().numpy(), idx.cpu().numpy()
    np.testing.assert_array_equal(cnt_h, np.diff(g["mask_offsets"]))
    for c in range(len(cnt_h)):
        np.testing.assert_array_equal(idx_h[c, :cnt_h[c]], g["mask_idx"][g["mask_offsets"][c]:g["mask_offsets"][c + 1]])
    w = _split(g["world_pred"], g["offsets"])
    lo = np.array([x.min(0) for x in w]); hi = np.array([x.max(0) for x in w])
    ctr, sz = (lo + hi) / np.float32(2), hi - lo
    np.testing.assert_array_equal(boxes.cpu().numpy(), np.concatenate([ctr - np.float32(0.6) * sz, ctr + np.float32(0.6) * sz], 1))


@pytest.mark.parametrize("from_pose", [False, True])
def test_masked_icp_large_cluster_regime_vs_oracle(dev, from_pose):
    """Clusters above the LDS source budget (the BASELINE configs[4] regime: 2048-point clusters) run the ICP iteration
    by iteration over many workgroups; same poses / iteration counts as the oracle's open3d-style loop, and the same as the
    single-launch kernel gives on the same problem cut into the small regime's limits."""
    from autourdf_amd import ops
    from autourdf_amd.synthetic import initial_segmentation, make_sequence
    from oracle import icp as oicp
    seq = make_sequence("wx200_5", 21, 2, 4600)
    mats, clusters, _ = initial_segmentation(seq[0], 4, seed=1)          # ~1150 points per cluster > 1024
    assert sum(len(c) for c in clusters) // 4 > 1024
    local, off = ops.pack_clusters(clusters, dev, torch.float64)
    M = _cuda(mats, dev)
    frame = _cuda(seq[1], dev)
    world32 = ops.cluster_transform(local.to(torch.float32), off, M.to(torch.float32))
    if from_pose:
        M_out, w_out, n_it = ops.masked_icp_batch([(local, None, off, frame, M)])[0]
    else:
        M_out, w_out, n_it = ops.masked_icp(local, world32, off, frame, M)
    world_h = _split(world32.cpu().numpy(), off.cpu().numpy())
    ow, om = oicp.masked_icp(clusters, world_h, seq[1], mats)
    np.testing.assert_allclose(M_out.cpu().numpy(), om, atol=1e-8)
    np.testing.assert_allclose(w_out.cpu().numpy(), np.concatenate(ow), atol=1e-8)
    assert (n_it.cpu().numpy() >= 1).all()


def test_icp_registrar_two_frames_vs_oracle(dev):
    """The ICP-style frame (K3 -> K4 -> K5 -> K2) device resident, against the oracle's composition
    of the same reference steps; labels bit-exact, poses far inside 1e-5."""
    from autourdf_amd.engine import IcpRegistrar
    from autourdf_amd.synthetic import initial_segmentation, make_sequence
    from oracle import dq as odq, icp as oicp, registration as oreg
    seq = make_sequence("wx200_5", 7, 3, 2048)
    mats, clusters, _ = initial_segmentation(seq[0], 12, seed=3)
    reg = IcpRegistrar(mats, clusters, dev)
    M = np.asarray(mats, np.float64)
    local = [np.asarray(c, np.float64) for c in clusters]
    for f in (1, 2):
        frame = np.asarray(seq[f], np.float64)
        M_gpu, dq_gpu, n_it = reg.step(torch.as_tensor(frame, device=dev))
        M32 = M.astype(np.float32)
        world32 = [c.astype(np.float32) @ m[:3, :3].T + m[:3, 3] for c, m in zip(local, M32)]
        _, M_new = oicp.masked_icp(local, world32, frame, M)
        np.testing.assert_allclose(M_gpu.cpu().numpy(), M_new, atol=1e-7)
        dq_ref = odq.transform_to_dualquat(torch.from_numpy(M_new.astype(np.float32))).numpy()
        np.testing.assert_allclose(dq_gpu.cpu().numpy(), dq_ref, atol=1e-5)
        # continue both sides from the SAME poses so the label comparison is exact by construction
        M = M_gpu.cpu().numpy()
        local, labels = oreg.resample_cluster(frame, len(clusters), M)
        off = reg.off.cpu().numpy()
        assert [len(c) for c in local] == list(np.diff(off))
        np.testing.assert_allclose(reg.local.cpu().numpy(), np.concatenate(local), atol=1e-9)
        assert (n_it.cpu().numpy() >= 1).all()


def test_batch_icp_registrar_equals_separate_registrars(dev):
    from autourdf_amd.engine import BatchIcpRegistrar, IcpRegistrar
    from autourdf_amd.synthetic import initial_segmentation, make_sequence
    seqs = [make_sequence("wx200_5", 40 + s, 3, 2048) for s in range(3)]
    mats, clusters, _ = initial_segmentation(seqs[0][0], 12, seed=3)
    breg = BatchIcpRegistrar(mats, clusters, 3, dev)
    singles = [IcpRegistrar(mats, clusters, dev) for _ in range(3)]
    for f in (1, 2):
        frames = [torch.as_tensor(s[f], dtype=torch.float64, device=dev) for s in seqs]
        outs = breg.step(frames)
        for r, fr, o in zip(singles, frames, outs):
            M, dq, it = r.step(fr)
            assert torch.equal(M, o[0]) and torch.equal(dq, o[1]) and torch.equal(it, o[2])
    for r, b in zip(singles, breg.regs):
        assert torch.equal(r.local, b.local) and torch.equal(r.off, b.off)


# ------------------------------------------------------------------------------------------ A1
def _train_case(golden, rot):
    g = golden("train_reference.npz")
    from oracle import models
    model = models.QRegMLP(True, 32) if rot == "q" else models.DQRegMLP(32)
    sd = {k[len(f"{rot}.sd."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f"{rot}.sd.")}
    return g, model, sd


def _oracle_model(rot, hidden):
    """The oracle's model of a --r choice (mlp_reg.py:276-291): QRegMLP / DQRegMLP / RRegMLP / RegMLP."""
    from oracle import models
    return {"q": lambda: models.QRegMLP(True, hidden), "dq": lambda: models.DQRegMLP(hidden), "6d": lambda: models.RRegMLP(hidden),
            "rpy": lambda: models.RegMLP(True, hidden)}[rot]()


def _order(rot):
    from autourdf_amd import ops
    return ops.DQ_PARAM_ORDER if rot == "dq" else ops.Q_PARAM_ORDER


@pytest.mark.parametrize("rot", ["q", "dq", "6d", "rpy"])
def test_train_probe_forward_and_pose_gradient_vs_oracle(dev, golden, rot):
    """One epoch's forward (pose, cloud, loss) and dL/d[R|t] against torch autograd on the oracle."""
    from autourdf_amd import ops
    from oracle import registration
    from oracle.chamfer import chamfer_distance
    data = rot if rot in ("q", "dq") else "q"                   # (the optional representations: on the 'q' fixture's problem)
    g, model, sd = _train_case(golden, data)
    hidden = 64                                                  # engine needs hidden % 64 == 0
    torch.manual_seed(3)
    model = _oracle_model(rot, hidden)
    for p in model.parameters():
        p.data.mul_(0.2)
    order = _order(rot)
    m, y = torch.from_numpy(g[f"{data}_m"]), torch.from_numpy(g[f"{data}_y"])
    clusters = [torch.from_numpy(c) for c in _split(g[f"{data}_local"], g[f"{data}_offsets"])]
    m2 = registration.pose_forward(m, model, rot)
    m2.retain_grad()
    pred = torch.cat(registration.calculate_pc(clusters, m2))
    loss, _ = chamfer_distance(pred[None], y[None], norm=1)
    loss.backward()
    plan = ops.TrainPlan(rot, len(clusters), hidden, pred.shape[0], y.shape[0], epochs=4, use_graph=False, device=dev)
    params = [model.state_dict()[k].clone().to(dev) for k in order]
    pts, off = ops.pack_clusters(clusters, dev)
    gm2, gpred, gloss, ggrad = plan.probe(m.to(dev), y.to(dev), pts, off, params)
    np.testing.assert_allclose(gm2.cpu().numpy(), m2.detach().numpy(), atol=2e-6)          # poses << 1e-5
    np.testing.assert_allclose(gpred.cpu().numpy(), pred.detach().numpy(), atol=2e-6)
    assert abs(gloss.item() - loss.item()) <= 2e-6 * abs(loss.item())
    np.testing.assert_allclose(ggrad.cpu().numpy()[:, :3, :], m2.grad.numpy()[:, :3, :], rtol=1e-4, atol=2e-6)


@pytest.mark.parametrize("rot", ["q", "dq"])
def test_train_vs_reference_train_directly_hidden64(dev, golden, rot):
    """A1 against the REFERENCE, no oracle in between: tests/golden/train_reference_h64.npz was minted by the reference's own
    train() (PointCloud/mlp_reg.py:17-152, run under ref_shims in the build container) at hidden 64 -- a width the plan tiles.
    Six epochs are pinned per step: every epoch's loss to 1e-5 relative, the best pose to 1e-5 (the north star's bound), the
    parameters after six Adam steps to 2e-5 (six updates of lr = 2e-4 each: a parameter whose gradient is rounding noise moves
    by +-lr whatever the implementation, see test_larger_configs_two_epochs_vs_oracle).  The 300-epoch run is compared loosely:
    argmin switches amplify 1 ulp, min_loss 1e-3 relative, poses 2e-3."""
    from autourdf_amd import ops
    g = golden("train_reference_h64.npz")
    order = ops.Q_PARAM_ORDER if rot == "q" else ops.DQ_PARAM_ORDER
    m, y = torch.from_numpy(g[f"{rot}_m"]).to(dev), torch.from_numpy(g[f"{rot}_y"]).to(dev)
    clusters = [torch.from_numpy(c) for c in _split(g[f"{rot}_local"], g[f"{rot}_offsets"])]
    pts, off = ops.pack_clusters(clusters, dev)
    for tag, epochs in (("e6", 6), ("e300", 300)):
        params = [torch.from_numpy(g[f"{rot}.sd.{k}"]).clone().to(dev) for k in order]
        plan = ops.TrainPlan(rot, len(clusters), 64, pts.shape[0], y.shape[0], epochs=epochs, use_graph=True, device=dev)
        bm, bp, res, lh, _ = plan.run(m, y, pts, off, params)
        res = res.cpu().numpy()
        if epochs == 6:
            np.testing.assert_allclose(lh.cpu().numpy().astype(np.float64), g[f"{rot}_e6_loss_hist"], rtol=1e-5)
            np.testing.assert_allclose(bm.cpu().numpy(), g[f"{rot}_e6_best_m"], atol=1e-5)
            np.testing.assert_allclose(bp.cpu().numpy(), g[f"{rot}_e6_best_pred"], atol=1e-5)
            for k, p in zip(order, params):
                np.testing.assert_allclose(p.cpu().numpy(), g[f"{rot}_e6.final.{k}"], atol=2e-5, err_msg=k)
        else:
            assert int(res[1]) == 300
            assert abs(res[0] - float(g[f"{rot}_e300_min_loss"])) <= 1e-3 * float(g[f"{rot}_e300_min_loss"])
            np.testing.assert_allclose(bm.cpu().numpy(), g[f"{rot}_e300_best_m"], atol=2e-3)


@pytest.mark.parametrize("rot,hidden", [("6d", 64), ("rpy", 3)])
def test_train_optional_representations_vs_reference_train_directly(dev, golden, rot, hidden):
    """--r 6d / --r rpy (mlp_reg.py:72-76, 86-90) on the fused plan against the REFERENCE's own train()
    (tests/golden/train_reference_rot.npz: RRegMLP at hidden 64; RegMLP(6, 3) as mlp_reg.py:285 constructs it -- hidden 3, which
    the plan runs zero-padded at width 64): every loss of six epochs 1e-5 relative, best pose and cloud 1e-5, parameters after six
    Adam steps 2e-5; the 300-epoch run to the envelope two float32 implementations keep."""
    from autourdf_amd import ops
    g = golden("train_reference_rot.npz")
    order = ops.Q_PARAM_ORDER
    m, y = torch.from_numpy(g[f"{rot}_m"]).to(dev), torch.from_numpy(g[f"{rot}_y"]).to(dev)
    clusters = [torch.from_numpy(c) for c in _split(g[f"{rot}_local"], g[f"{rot}_offsets"])]
    pts, off = ops.pack_clusters(clusters, dev)
    for tag, epochs in (("e6", 6), ("e300", 300)):
        params = [torch.from_numpy(g[f"{rot}.sd.{k}"]).clone().to(dev) for k in order]
        plan = ops.TrainPlan(rot, len(clusters), hidden, pts.shape[0], y.shape[0], epochs=epochs, use_graph=True, device=dev)
        bm, bp, res, lh, _ = plan.run(m, y, pts, off, params)
        res = res.cpu().numpy()
        if epochs == 6:
            np.testing.assert_allclose(lh.cpu().numpy().astype(np.float64), g[f"{rot}_e6_loss_hist"], rtol=1e-5)
            np.testing.assert_allclose(bm.cpu().numpy(), g[f"{rot}_e6_best_m"], atol=1e-5)
            np.testing.assert_allclose(bp.cpu().numpy(), g[f"{rot}_e6_best_pred"], atol=1e-5)
            for k, p in zip(order, params):
                assert tuple(p.shape) == tuple(g[f"{rot}_e6.final.{k}"].shape)
                np.testing.assert_allclose(p.cpu().numpy(), g[f"{rot}_e6.final.{k}"], atol=2e-5, err_msg=k)
        else:
            assert int(res[1]) == 300
            assert abs(res[0] - float(g[f"{rot}_e300_min_loss"])) <= 1e-3 * float(g[f"{rot}_e300_min_loss"])
            np.testing.assert_allclose(bm.cpu().numpy(), g[f"{rot}_e300_best_m"], atol=2e-3)


def _c1_case(golden, dev):
    from autourdf_amd import ops
    g = golden("train_reference_c1.npz")
    sd = {k[5:]: torch.from_numpy(g[k].astype(np.float32)) for k in g.files if k.startswith("sd16.")}
    m, y = torch.from_numpy(g["m"]).to(dev), torch.from_numpy(g["y"]).to(dev)
    pts, off = ops.pack_clusters([torch.from_numpy(c) for c in _split(g["local"], g["offsets"])], dev)
    return g, sd, m, y, pts, off


def test_train_c1_headline_shape_vs_reference_train_directly(dev, golden):
    """A1 at the shape the metric is quoted on (BASELINE configs[1]: N = 4096, K = 20, the reference's default QRegMLP(True, 512)),
    against the REFERENCE's own train() (tests/golden/train_reference_c1.npz, minted under ref_shims), no oracle in between:
    every loss of the first six epochs 1e-5 relative, the best pose and cloud of those epochs 1e-5 (the north star's bound),
    a strided sample of the parameters after six Adam steps (a parameter whose gradient is rounding noise moves by +-lr per step
    whatever the implementation: 6 x 2e-4 is the cap, the bulk agrees to 3e-6)."""
    from autourdf_amd import ops
    g, sd, m, y, pts, off = _c1_case(golden, dev)
    order = ops.Q_PARAM_ORDER
    params = [sd[k].clone().to(dev) for k in order]
    plan = ops.TrainPlan("q", 20, 512, pts.shape[0], y.shape[0], epochs=6, use_graph=True, device=dev)
    bm, bp, res, lh, _ = plan.run(m, y, pts, off, params)
    np.testing.assert_allclose(lh.cpu().numpy().astype(np.float64), g["loss_hist"][:6], rtol=1e-5)
    assert abs(float(res[0]) - float(g["e6_min_loss"])) <= 1e-5 * float(g["e6_min_loss"])
    np.testing.assert_allclose(bm.cpu().numpy(), g["e6_best_m"], atol=1e-5)
    np.testing.assert_allclose(bp.cpu().numpy(), g["e6_best_pred"], atol=1e-5)
    stride = int(g["sample_stride"])
    for k, p in zip(order, params):
        d = np.abs(p.cpu().numpy().reshape(-1)[::stride] - g[f"e6.final_sample.{k}"])
        assert np.median(d) < 3e-6 and (d < 3e-5).mean() >= 0.99 and d.max() < 1.25e-3, (k, np.median(d), d.max())
        assert abs(float(p.double().sum()) - float(g[f"e6.final_sum.{k}"])) < 2e-2, k


def test_train_c1_divergence_stays_inside_the_measured_oracle_envelope(dev, golden):
    """How tight can pose parity be after e epochs at the headline shape?  tests/golden/divergence_envelope_c1.npz
    (tests/measure/divergence_envelope.py cpu) MEASURES it: the float32 oracle -- which reproduces the reference's trajectory bit for
    bit -- against itself with the points of every cluster and the target rows permuted (six seeds) and against a float64 run.
    Up to N_e (8 on this problem) every variant stays within the north star's 1e-5; from then on two correct float32 implementations
    of the same mathematics are 1e-3..6e-2 apart (Adam turns rounding noise into +-lr steps, nearest-neighbour switches are kinks).
    The plan is held to exactly that: its pose after e Adam steps (a run of e epochs, then the forward of the trained parameters)
    against the REFERENCE's pose_hist[e]
      * e <= N_e: within 1e-5 (measured 3e-7..1e-6),
      * e = 30, 299: within twice the envelope of the permuted oracle runs at that epoch,
    and what train() returns after 300 epochs -- min_loss, best pose -- within twice the spread of the oracle variants."""
    from autourdf_amd import ops
    g, sd, m, y, pts, off = _c1_case(golden, dev)
    env = golden("divergence_envelope_c1.npz")
    n_e = int(env["n_e"])
    order = ops.Q_PARAM_ORDER
    probe_plan = ops.TrainPlan("q", 20, 512, pts.shape[0], y.shape[0], epochs=2, use_graph=False, device=dev)
    ref = g["pose_hist"]
    seen = {}
    for e in sorted({1, 2, 6, n_e, 30, 299}):
        params = [sd[k].clone().to(dev) for k in order]
        plan = ops.TrainPlan("q", 20, 512, pts.shape[0], y.shape[0], epochs=e, use_graph=e >= 2, device=dev)
        plan.run(m, y, pts, off, params, stop=10 ** 6)
        m2, _, _, _ = probe_plan.probe(m, y, pts, off, params)
        d = float(np.abs(m2.cpu().numpy()[:, :3, :] - ref[e][:, :3, :]).max())
        seen[e] = d
        # (round 6: the envelope also holds runs whose MLP matrix products sum in another order -- what a different BLAS does -- `gemm`)
        tol = 1e-5 if e <= n_e else 2.0 * float(max(env["envelope"][e], env["gemm"][e]))
        assert d <= tol, (e, d, tol, seen)
    params = [sd[k].clone().to(dev) for k in order]
    plan = ops.TrainPlan("q", 20, 512, pts.shape[0], y.shape[0], epochs=300, use_graph=True, device=dev)
    bm, _, res, _, _ = plan.run(m, y, pts, off, params)
    ml = float(g["e300_min_loss"])
    assert abs(float(res[0]) - ml) <= 2.0 * float(env["min_loss_rel_envelope"]) * ml, (float(res[0]), ml)
    assert float(np.abs(bm.cpu().numpy()[:, :3, :] - g["e300_best_m"][:, :3, :]).max()) <= 2.0 * float(env["best_pose_envelope"])


@pytest.mark.parametrize("shape,k,n", [("allegro", 30, 4096), ("franka", 40, 16384)])
def test_train_other_shapes_vs_reference_train_and_the_measured_envelope(dev, golden, shape, k, n):
    """VERDICT r4 ("what's weak" 1): the whole-frame parity claim rested on ONE problem.  tests/golden/train_reference_{allegro,franka}.npz
    (make_golden_shapes.py: the REFERENCE's own train() at BASELINE configs[3] / configs[2] shapes, from the pinned state of the configs[1]
    golden) and divergence_envelope_{shape}.npz (divergence_envelope.py cpu:<shape>: the float32 oracle against itself with permuted
    points and in float64) put the other two registration shapes under the same statements:
      * the forward is exact: the pose before any step within 2e-6 of the reference's, the first loss within 1e-5 relative;
      * ONE Adam step later the plan is within 1e-5 -- or, where the envelope says no second implementation is (the allegro shape:
        the float64 oracle is 4.3e-4 away after one step; a sum of +-1/N signs that is an output bias's gradient sits next to zero and
        one point within an ulp of its neighbour's coordinate flips its sign), within one full step of that bias in the other direction
        (2.5 lr);
      * what train() returns after 300 epochs -- min_loss, best pose -- and the pose at epoch 299 stay inside twice the spread of the
        oracle variants (allegro: the spread itself is 59 % / a flipped cluster; franka 0.4 % / 1.9e-2: DESIGN.md section 2)."""
    from autourdf_amd import ops
    g = golden(f"train_reference_{shape}.npz")
    env = golden(f"divergence_envelope_{shape}.npz")
    c1 = golden("train_reference_c1.npz")
    sd = {key[5:]: torch.from_numpy(c1[key].astype(np.float32)) for key in c1.files if key.startswith("sd16.")}
    order = ops.Q_PARAM_ORDER
    m, y = torch.from_numpy(g["m"]).to(dev), torch.from_numpy(g["y"]).to(dev)
    pts, off = ops.pack_clusters([torch.from_numpy(c) for c in _split(g["local"], g["offsets"])], dev)
    assert pts.shape[0] == n and len(g["offsets"]) == k + 1
    ref = {int(e): g["pose_hist_sel"][i] for i, e in enumerate(g["pose_epochs"])}
    spread = np.maximum(np.maximum(env["envelope"], env["f64"]), env["gemm"])      # permuted points, float64, another GEMM summation order
    probe_plan = ops.TrainPlan("q", k, 512, n, n, epochs=2, use_graph=False, device=dev)
    params = [sd[key].clone().to(dev) for key in order]
    m0, _, loss0, _ = probe_plan.probe(m, y, pts, off, params)
    assert float(np.abs(m0.cpu().numpy()[:, :3, :] - ref[0][:, :3, :]).max()) <= 2e-6
    assert abs(float(loss0) - float(g["loss_hist"][0])) <= 1e-5 * float(g["loss_hist"][0])
    seen = {}
    for e in (1, 299):
        params = [sd[key].clone().to(dev) for key in order]
        plan = ops.TrainPlan("q", k, 512, n, n, epochs=e, use_graph=e >= 2, device=dev)
        plan.run(m, y, pts, off, params, stop=10 ** 6)
        m2, _, _, _ = probe_plan.probe(m, y, pts, off, params)
        seen[e] = float(np.abs(m2.cpu().numpy()[:, :3, :] - ref[e][:, :3, :]).max())
    one_step = 1e-5 if float(spread[1]) <= 1e-5 else 2.5 * 2e-4
    assert seen[1] <= one_step, seen
    params = [sd[key].clone().to(dev) for key in order]
    plan = ops.TrainPlan("q", k, 512, n, n, epochs=300, use_graph=True, device=dev)
    bm, _, res, _, _ = plan.run(m, y, pts, off, params)
    ml = float(g["e300_min_loss"])
    assert np.isfinite(float(res[0])) and torch.isfinite(bm).all()
    if float(env["best_pose_envelope"]) < 0.5:
        # an envelope that can bite (franka: 0.43 % / 1.9e-2).  At the allegro shape the oracle variants themselves end a flipped cluster
        # apart (59 % / 2.0 on entries that cannot exceed ~2): bounds taken from that spread cannot fail and are not
        # asserted -- the late epochs of that shape are pinned by test_train_teacher_forced_late_epochs below instead
        assert seen[299] <= 2.0 * float(spread[299]), seen
        assert abs(float(res[0]) - ml) <= 2.0 * float(env["min_loss_rel_envelope"]) * ml, (float(res[0]), ml)
        assert float(np.abs(bm.cpu().numpy()[:, :3, :] - g["e300_best_m"][:, :3, :]).max()) <= 2.0 * float(env["best_pose_envelope"])


@pytest.mark.parametrize("shape,k,n", [("c1", 20, 4096), ("allegro", 30, 4096), ("franka", 40, 16384)])
def test_train_teacher_forced_late_epochs(dev, golden, shape, k, n):
    """exact parity beyond the divergence horizon.  Free-running trains of two float32 implementations part after
    ~13 epochs, so everything later -- Adam's bias corrections at t = 150..300, ReduceLROnPlateau after several cuts,
    best tracking late in a train -- was only ever compared against envelopes.  Here the errors cannot accumulate: the oracle (the
    reference's float32 trajectory: bit for bit in the build container, checked below against the reference-minted loss history) runs
    e epochs on the host, hands its state ENTERING epoch e to the plan (creg_train_plan_resume: parameters, Adam moments, step count,
    scheduler best / bad epochs / lr, min_loss, stop counter, best pose so far), the plan runs ONE epoch, at BASELINE configs[1], [3]
    and [2] shapes, hidden 512, e in {0, 13, 50, 150, 299}:
      * the epoch's loss 1e-6 relative, its pose 1e-5 (measured <= 3e-7 / 6e-7: profiles/r06_teacher_forced.log);
      * lr (used and next), step, epochs_run, scheduler bad-epoch count, stop counter, best epoch: EXACT; scheduler best / min_loss 1e-6;
      * the parameter update of every tensor within 3e-6 (1.5 % of a full 2e-4 step; measured <= 1.9e-6) outside a mask of |grad| < 1e-8.
        The FIRST Adam step (e = 0) normalises every gradient to +-1, so an element whose gradient is rounding noise around zero moves
        by a full +-lr whatever the implementation (allegro: 2 lr on 0.26 % of the elements, none at the other two shapes): there >= 99 %
        of the elements are within the bound and none moves by more than one flipped unit step;
      * Adam's second moments after the epoch 1e-2 of the tensor's largest (they see 0.001 g^2), the best pose / cloud 1e-5."""
    import _teacher as T
    from autourdf_amd import ops
    g = golden(f"train_reference_{shape}.npz")
    c1 = golden("train_reference_c1.npz")
    sd = {key[5:]: torch.from_numpy(c1[key].astype(np.float32)) for key in c1.files if key.startswith("sd16.")}
    order = ops.Q_PARAM_ORDER
    epochs = (0, 13, 50, 150, 299)
    snaps, hist = T.oracle_snapshots(g, sd, k, epochs)
    ref_loss = np.asarray(g["loss_hist"], np.float64)
    # the teacher IS the reference's computation: its losses agree with the reference-minted history for as long as ANY second float32
    # evaluation of the same code does -- the host's BLAS is another one than the build container's, where the two are bit-identical --
    # i.e. up to the shape's measured horizon N_e (divergence_envelope_*.npz: configs[1] 8 epochs, franka 1, allegro 0)
    n_same = min(int(golden(f"divergence_envelope_{shape}.npz")["n_e"]) + 1, 6)
    np.testing.assert_allclose(np.asarray(hist["loss"][:n_same], np.float64), ref_loss[:n_same], rtol=1e-5)
    m, y = torch.from_numpy(g["m"]).to(dev), torch.from_numpy(g["y"]).to(dev)
    pts, off = ops.pack_clusters([torch.from_numpy(c) for c in _split(g["local"], g["offsets"])], dev)
    plan = ops.TrainPlan("q", k, 512, n, y.shape[0], epochs=300, use_graph=False, device=dev)
    probe_plan = ops.TrainPlan("q", k, 512, n, y.shape[0], epochs=2, use_graph=False, device=dev)
    for e in epochs:
        s0, s1 = snaps[e], snaps[e + 1]
        m2, loss, params, after, extra = T.plan_epoch(plan, probe_plan, dev, order, m, y, pts, off, s0)
        r = T.compare(order, s0, s1, m2, loss, params, after, extra)
        assert r["loss_rel"] <= 1e-6 and r["pose"] <= 1e-5, (e, r)
        assert r["lr_used_exact"] and r["lr"][0] == r["lr"][1], (e, r["lr"])
        for key, (a, b) in r["exact"].items():
            assert a == b, (e, key, a, b)
        for key in ("sched_best", "min_loss"):
            assert abs(r[key][0] - r[key][1]) <= 1e-6 * abs(r[key][1]), (e, key, r[key])
        assert r["exp_avg_sq_rel"] <= 1e-2, (e, r)
        if "best_m" in r:
            assert r["best_m"] <= 1e-5 and r["best_pred"] <= 1e-5, (e, r)
        if e > 0:
            assert r["upd"] <= 3e-6, (e, r["upd"], r["upd_worst_tensor"])
        else:
            lr, bad, live_n = s0["lr"], 0, 0
            for i, key in enumerate(order):
                live = s0["epoch"]["grad"][key].reshape(-1).abs() >= 1e-8
                err = ((params[i].cpu() - s0["params"][key]) - (s1["params"][key] - s0["params"][key])).reshape(-1).abs()[live]
                bad += int((err > 3e-6).sum()); live_n += int(live.sum())
                assert float(err.max()) <= 2.0 * lr * 1.001, (key, float(err.max()))
            assert bad <= 1e-2 * live_n, (bad, live_n)             # (measured: 0 at configs[1] and franka, 0.26 % at the allegro shape)


def test_train_resume_continues_a_run(dev):
    """creg_train_plan_resume as checkpoint / resume: 12 epochs in one run against 5 epochs, the state read back, 7 more from it.
    The control state (lr, step, scheduler / stop counters, best epoch) is EXACT and the resumed run is bit-reproducible; losses, parameters
    and moments agree to rounding, not bit for bit: the first activation after a resume comes from k_l1 (a DPP wave sum over the inputs)
    where the running loop takes it from the registers of the rows k_bd has just updated (another association of the same sum) -- the
    trajectories then differ like any two float32 evaluations inside the divergence horizon (measured <= 2e-6 after 7 epochs here)."""
    from autourdf_amd import ops
    from autourdf_amd.synthetic import initial_segmentation, make_sequence
    from oracle import models
    seq = make_sequence("wx200_5", 7, 2, 1500)
    mats, cl, _ = initial_segmentation(seq[0], 7, seed=1)
    m = torch.tensor(mats, dtype=torch.float32, device=dev)
    pts, off = ops.pack_clusters([torch.tensor(c, dtype=torch.float32) for c in cl], dev)
    y = torch.tensor(seq[1], dtype=torch.float32, device=dev)
    lr0 = float(np.float32(2e-4))        # creg_train_args.lr is a float: the plain run's double lr is its widening; the state's lr is a double of its own
    for rot, ctor, order in (("q", lambda: models.QRegMLP(True, 64), ops.Q_PARAM_ORDER), ("dq", lambda: models.DQRegMLP(64), ops.DQ_PARAM_ORDER)):
        torch.manual_seed(3)
        sd = ctor().state_dict()
        fresh = lambda: [sd[kk].clone().to(dev) for kk in order]
        zeros = lambda ps: [torch.zeros_like(p) for p in ps]
        plan = ops.TrainPlan(rot, 7, 64, pts.shape[0], y.shape[0], epochs=12, use_graph=False, device=dev)
        pa = fresh()
        bm_a, bp_a, res_a, lh_a, lrh_a, st_a = plan.resume(m, y, pts, off, pa, {"exp_avg": zeros(pa), "exp_avg_sq": zeros(pa), "lr": lr0}, 12, patience=1)
        pb = fresh()
        bm_1, bp_1, _, lh_1, _, st_1 = plan.resume(m, y, pts, off, pb, {"exp_avg": zeros(pb), "exp_avg_sq": zeros(pb), "lr": lr0}, 5, patience=1)
        assert st_1["step"] == 5 and st_1["epochs_run"] == 5
        pb5 = [p.clone() for p in pb]
        bm_b, bp_b, res_b, lh_b, lrh_b, st_b = plan.resume(m, y, pts, off, pb, st_1, 7, patience=1, best=(bm_1, bp_1))
        # the resumed stretch is bit-reproducible
        pb2 = [p.clone() for p in pb5]
        bm_b2, _, _, lh_b2, _, st_b2 = plan.resume(m, y, pts, off, pb2, st_1, 7, patience=1, best=(bm_1, bp_1))
        assert all(torch.equal(a, b) for a, b in zip(pb, pb2)) and torch.equal(lh_b[5:12], lh_b2[5:12]) and torch.equal(bm_b, bm_b2)
        assert all(torch.equal(a, b) for a, b in zip(st_b["exp_avg"], st_b2["exp_avg"]))
        # control state exact against the one-run train; numbers to rounding
        for key in ("step", "epochs_run", "sched_bad", "count", "best_epoch", "stopped", "lr"):
            assert st_a[key] == st_b[key], (key, st_a[key], st_b[key])
        assert st_a["step"] == 12 and st_a["epochs_run"] == 12
        assert torch.equal(lrh_a[5:12], lrh_b[5:12]) and torch.equal(lh_a[:5], lh_1[:5]) and torch.isnan(lh_b[:5]).all()
        np.testing.assert_allclose(lh_b[5:12].cpu().numpy(), lh_a[5:12].cpu().numpy(), rtol=2e-5)
        assert abs(st_a["min_loss"] - st_b["min_loss"]) <= 2e-5 * abs(st_a["min_loss"])
        np.testing.assert_allclose(bm_b.cpu().numpy(), bm_a.cpu().numpy(), atol=1e-5)
        for a, b in zip(pa, pb):
            d = (a - b).abs()
            assert float(d.max()) <= 7 * 2.01 * 2e-4 and float((d > 5e-6).float().mean()) <= 1e-2, (float(d.max()), float((d > 5e-6).float().mean()))
        # ... and the resume entry point from a fresh state IS the plain run, bit for bit (same launches from the same staged inputs)
        pc = fresh()
        bm_c, bp_c, res_c, lh_c, _ = ops.TrainPlan(rot, 7, 64, pts.shape[0], y.shape[0], epochs=12, use_graph=True, device=dev).run(m, y, pts, off, pc, patience=1)
        assert torch.equal(bm_a, bm_c) and torch.equal(lh_a, lh_c) and all(torch.equal(a, c) for a, c in zip(pa, pc))


def test_train_same_target_keeps_the_frames_leaves_bit_identical(dev):
    """creg_train_args.y_unchanged (ops: same_target=True): "Anchor" after "Step" on the same frame keeps the target frame's k-d leaf
    blocks instead of sorting the frame again -- every output bit of the second train is what a rebuilding run gives, single and
    batched, also right after a probe (which overwrites the blocks and must invalidate them)."""
    from autourdf_amd import ops
    from autourdf_amd.synthetic import initial_segmentation, make_sequence
    from oracle import models
    seq = make_sequence("wx200_5", 31, 3, 1500)
    mats, cl, _ = initial_segmentation(seq[0], 7, seed=2)
    m = torch.tensor(mats, dtype=torch.float32, device=dev)
    pts, off = ops.pack_clusters([torch.tensor(c, dtype=torch.float32) for c in cl], dev)
    ys = [torch.tensor(seq[i], dtype=torch.float32, device=dev) for i in (1, 2)]
    torch.manual_seed(5)
    sds = [models.QRegMLP(True, 64).state_dict() for _ in range(4)]

    def params(i):
        return [sds[i][k].clone().to(dev) for k in ops.Q_PARAM_ORDER]

    for batch in (1, 2):
        outs = {}
        for keep in (False, True):
            plan = ops.TrainPlan("q", 7, 64, pts.shape[0], ys[0].shape[0], epochs=12, use_graph=True, device=dev, batch=batch)
            probs = lambda base, mm: [(mm[b], ys[b % 2], pts, off, params(base + b)) for b in range(batch)]
            step = plan.run_batch(probs(0, [m] * batch), lr=2e-4)
            anchor = plan.run_batch(probs(2, [o[0] for o in step]), lr=1e-4, same_target=keep)
            outs[keep] = [torch.cat([t.reshape(-1) for t in o[:4]]) for o in anchor]
            if keep:                               # a probe rebuilds the blocks for ITS frame: the flag must not trust them afterwards
                plan.probe(m, ys[1], pts, off, params(3))
                again = plan.run_batch(probs(2, [o[0] for o in step]), lr=1e-4, same_target=True)
                for a, b in zip(outs[True], again):
                    assert torch.equal(a, torch.cat([t.reshape(-1) for t in b[:4]]))
            torch.cuda.synchronize()
        for a, b in zip(outs[False], outs[True]):
            assert torch.equal(a, b)


@pytest.mark.parametrize("n_pts", [1500, 9000])
def test_train_same_target_claim_is_verified_on_the_device(dev, n_pts):
    """y_unchanged used to rest on the caller's word -- a buffer reused for the NEXT frame, or two registrars sharing a plan,
    and the search ran through the previous frame's k-d leaves: wrong neighbours, no error.  Round 5: the launch that would sort the
    frame compares a 64-bit position-dependent fingerprint of the staged frame with the one stored with the leaves and sorts again
    when they differ.  A false claim (another frame; the same points in another order; one coordinate changed in its last bit) must
    give exactly what an honest run gives, with both searches (1500 / 9000 points)."""
    from autourdf_amd import ops
    from autourdf_amd.synthetic import initial_segmentation, make_sequence
    from oracle import models
    seq = make_sequence("wx200_5", 32, 3, n_pts)
    mats, cl, _ = initial_segmentation(seq[0], 7, seed=2)
    m = torch.tensor(mats, dtype=torch.float32, device=dev)
    pts, off = ops.pack_clusters([torch.tensor(c, dtype=torch.float32) for c in cl], dev)
    y1 = torch.tensor(seq[1], dtype=torch.float32, device=dev)
    g = torch.Generator().manual_seed(1)
    bumped = y1.clone()
    bumped[n_pts // 2, 1] = torch.nextafter(bumped[n_pts // 2, 1], torch.tensor(10.0, device=dev))
    others = {"next frame": torch.tensor(seq[2], dtype=torch.float32, device=dev), "permuted": y1[torch.randperm(n_pts, generator=g).to(dev)],
              "one ulp": bumped}
    torch.manual_seed(5)
    sd = models.QRegMLP(True, 64).state_dict()
    params = lambda: [sd[k].clone().to(dev) for k in ops.Q_PARAM_ORDER]
    flat = lambda o: torch.cat([t.reshape(-1) for t in o[:4]])
    for name, y2 in others.items():
        honest = ops.TrainPlan("q", 7, 64, pts.shape[0], n_pts, epochs=8, use_graph=True, device=dev)
        honest.run(m, y1, pts, off, params())
        want = flat(honest.run(m, y2, pts, off, params()))
        lying = ops.TrainPlan("q", 7, 64, pts.shape[0], n_pts, epochs=8, use_graph=True, device=dev)
        lying.run(m, y1, pts, off, params())
        got = flat(lying.run(m, y2, pts, off, params(), same_target=True))
        assert torch.equal(got.nan_to_num(), want.nan_to_num()), name


@pytest.mark.parametrize("rot", ["q", "dq"])
def test_train_hidden32_reference_golden_runs_on_the_plan(dev, golden, rot):
    """tests/golden/train_reference.npz is the reference's own train() at hidden 32 (300 epochs) -- a width the kernels are not
    instantiated for: the plan runs it at 64 with the extra units' parameters zero (ops.TrainPlan docstring: their activations,
    gradients and Adam updates are exactly 0), so A1 is compared with this golden DIRECTLY too.  300 epochs: min_loss 1e-3
    relative and poses 2e-3, as for the hidden-64 golden (argmin switches amplify 1 ulp); the first six
    epochs against the oracle on the same inputs: losses 2e-5 relative, poses 1e-5, trained parameters (in the caller's own
    shapes) like test_train_three_steps_odd_shapes_vs_oracle."""
    from autourdf_amd import ops
    from oracle import registration
    g, model, sd = _train_case(golden, rot)
    order = ops.Q_PARAM_ORDER if rot == "q" else ops.DQ_PARAM_ORDER
    m, y = torch.from_numpy(g[f"{rot}_m"]), torch.from_numpy(g[f"{rot}_y"])
    clusters = [torch.from_numpy(c) for c in _split(g[f"{rot}_local"], g[f"{rot}_offsets"])]
    pts, off = ops.pack_clusters(clusters, dev)
    params = [sd[k].clone().to(dev) for k in order]
    plan = ops.TrainPlan(rot, len(clusters), 32, pts.shape[0], y.shape[0], epochs=300, use_graph=True, device=dev)
    assert plan.hidden == 64 and plan.hidden_model == 32
    bm, bp, res, lh, _ = plan.run(m.to(dev), y.to(dev), pts, off, params)
    res = res.cpu().numpy()
    assert int(res[1]) == 300
    assert abs(res[0] - float(g[f"{rot}_min_loss"])) <= 1e-3 * float(g[f"{rot}_min_loss"])
    np.testing.assert_allclose(bm.cpu().numpy(), g[f"{rot}_best_m"], atol=2e-3)
    assert all(p.shape == sd[k].shape for p, k in zip(params, order))
    # six epochs, tight, against the oracle from the same state
    model.load_state_dict(sd)
    _, best_m, min_loss, hist = registration.train(m, y, model, clusters, rot=rot, epochs=6)
    params = [sd[k].clone().to(dev) for k in order]
    plan6 = ops.TrainPlan(rot, len(clusters), 32, pts.shape[0], y.shape[0], epochs=6, use_graph=False, device=dev)
    bm, _, res, lh, _ = plan6.run(m.to(dev), y.to(dev), pts, off, params)
    np.testing.assert_allclose(lh.cpu().numpy(), np.array(hist["loss"], np.float32), rtol=2e-5)
    np.testing.assert_allclose(bm.cpu().numpy(), best_m.detach().numpy(), atol=1e-5)
    for name, p in zip(order, params):
        d = np.abs(p.cpu().numpy() - model.state_dict()[name].numpy())
        assert np.median(d) < 3e-6 and (d < 3e-5).mean() >= 0.99 and d.max() < 1.2e-3, (name, np.median(d), d.max())


@pytest.mark.parametrize("rot,hidden", [("q", 48), ("dq", 100), ("q", 200), ("dq", 300), ("6d", 100), ("rpy", 3), ("rpy", 48)])
def test_train_any_hidden_width_three_steps_vs_oracle(dev, rot, hidden):
    """Widths between the instantiated tiles (odd halves included: decoder_1 is hidden // 2 wide): three Adam steps against the
    oracle's model of the SAME width -- loss history 2e-5 relative, best pose 1e-5."""
    from autourdf_amd import ops
    from autourdf_amd.synthetic import initial_segmentation, make_sequence
    from oracle import models, registration
    seq = make_sequence("wx200_5", 23, 2, 1200)
    mats, cl, _ = initial_segmentation(seq[0], 6, seed=4)
    m, y = torch.tensor(mats, dtype=torch.float32), torch.tensor(seq[1], dtype=torch.float32)
    clusters = [torch.tensor(c, dtype=torch.float32) for c in cl]
    torch.manual_seed(13)
    model = _oracle_model(rot, hidden)
    order = _order(rot)
    params = [model.state_dict()[n].clone().to(dev) for n in order]
    pts, off = ops.pack_clusters(clusters, dev)
    plan = ops.TrainPlan(rot, len(clusters), hidden, pts.shape[0], y.shape[0], epochs=3, use_graph=True, device=dev)
    bm, _, _, lh, _ = plan.run(m.to(dev), y.to(dev), pts, off, params, lr=1e-3)
    _, best_m, _, hist = registration.train(m, y, model, clusters, rot=rot, epochs=3, learning_rate=1e-3)
    np.testing.assert_allclose(lh.cpu().numpy(), np.array(hist["loss"], np.float32), rtol=2e-5)
    np.testing.assert_allclose(bm.cpu().numpy(), best_m.detach().numpy(), atol=1e-5)
    assert all(tuple(p.shape) == tuple(model.state_dict()[n].shape) for p, n in zip(params, order))


@pytest.mark.parametrize("rot,hidden,k", [("q", 64, 7), ("q", 128, 5), ("dq", 64, 3), ("dq", 128, 9), ("q", 256, 21), ("dq", 512, 33),
                                          ("6d", 512, 20), ("6d", 64, 33), ("rpy", 128, 7), ("6d", 256, 142), ("6d", 256, 160), ("q", 256, 160), ("rpy", 256, 160), ("q", 512, 256), ("dq", 256, 256)])
def test_train_three_steps_odd_shapes_vs_oracle(dev, rot, hidden, k):
    """Hidden sizes and cluster counts whose staged activation blocks do NOT end on a 64 x 16-byte boundary (the
    LDS-DMA tail case): three full Adam steps against the oracle, loss history and poses."""
    from autourdf_amd import ops
    from autourdf_amd.synthetic import initial_segmentation, make_sequence
    from oracle import models, registration
    seq = make_sequence("wx200_5", 17, 2, 1500)
    mats, cl, _ = initial_segmentation(seq[0], k, seed=2)
    m = torch.tensor(mats, dtype=torch.float32)
    y = torch.tensor(seq[1], dtype=torch.float32)
    clusters = [torch.tensor(c, dtype=torch.float32) for c in cl]
    torch.manual_seed(9)
    model = _oracle_model(rot, hidden)
    order = _order(rot)
    params = [model.state_dict()[n].clone().to(dev) for n in order]
    pts, off = ops.pack_clusters(clusters, dev)
    plan = ops.TrainPlan(rot, k, hidden, pts.shape[0], y.shape[0], epochs=3, use_graph=True, device=dev)
    bm, bp, res, lh, _ = plan.run(m.to(dev), y.to(dev), pts, off, params, lr=1e-3)
    _, best_m, min_loss, hist = registration.train(m, y, model, clusters, rot=rot, epochs=3, learning_rate=1e-3)
    # (the largest cluster count: 160 clusters of 3-10 points each.  A dozen of the 10^5 weight gradients are exact cancellations there,
    #  Adam's FIRST step turns their rounding into +-lr -- 2 lr between two implementations -- and the second loss already differs by
    #  1e-3 whatever the representation: measured for 'q', '6d' and 'rpy' alike, and differently from one run of the CPU oracle to the
    #  next, while the plan returns the same bits every time.  Epoch 0 is held tightly, the rest to that.)
    wide = k >= 150
    np.testing.assert_allclose(lh.cpu().numpy()[:1], np.array(hist["loss"], np.float32)[:1], rtol=2e-5)
    np.testing.assert_allclose(lh.cpu().numpy(), np.array(hist["loss"], np.float32), rtol=3e-3 if wide else 2e-5)
    if not wide:
        np.testing.assert_allclose(bm.cpu().numpy(), best_m.detach().numpy(), atol=1e-5)
    # the trained parameters come back from the buffer an ODD number of optimizer steps leaves them in (k_params_home): Adam's
    # first steps move every weight by ~lr = 1e-3, so a stale buffer would be off by 1e-3, rounding by 1e-6
    # (a weight whose gradient is a cancelling sum has m/sqrt(v) decided by rounding -- Adam normalises it to a full step --
    # so a few elements in a thousand differ by a fraction of lr (measured: 0.3 % of W2 above 3e-5, none above 3e-4); a
    # stale buffer would put the MEDIAN at ~5e-4.  Median inside rounding, 99 % inside 3e-5, all inside half a step.)
    for name, p in zip(order, params):
        d = np.abs(p.cpu().numpy() - model.state_dict()[name].numpy())
        assert np.median(d) < (1e-4 if wide else 3e-6), (name, np.median(d))
        assert (d < 3e-5).mean() >= (0.5 if wide else 0.99), (name, (d < 3e-5).mean())
        assert d.max() < (7e-3 if wide else 5e-4), (name, d.max())         # (wide: three steps of 2 lr each at most)


def _plan_first_moments(plan, order, shapes):
    """The Adam first moments a plan holds after a run, per parameter tensor, read from the CALLER-OWNED workspace (DESIGN.md section 3:
    four flat float32 arrays of NPAR values each -- P, P1, m, v -- 256-byte aligned, in the order of ops.Q_PARAM_ORDER / DQ_PARAM_ORDER
    with decoder_1.0 / decoder_2.0 stacked into one W2).  After ONE optimizer step from zero moments m = (1 - beta1) g exactly, so
    this is the weight gradient of the first epoch, which nothing the plan returns exposes."""
    n = {k: int(np.prod(shapes[k])) for k in order}
    npar = sum(n.values())
    stride = (4 * npar + 255) // 256 * 256
    base = (plan.ws.data_ptr() + 255) // 256 * 256 - plan.ws.data_ptr()
    am = plan.ws[base + 2 * stride: base + 2 * stride + 4 * npar].view(torch.float32).cpu().numpy()
    if len(order) == 10:        # flat order: W1 b1 | W2 = [dec1.0 ; dec2.0] , b2 = [dec1.0.b ; dec2.0.b] | W3A b3A | W3B b3B
        flat = [order[0], order[1], order[2], order[6], order[3], order[7], order[4], order[5], order[8], order[9]]
    else:
        flat = list(order)
    out, o = {}, 0
    for k in flat:
        out[k] = am[o:o + n[k]].reshape(shapes[k])
        o += n[k]
    return out


@pytest.mark.parametrize("rot,k", [("6d", 142), ("6d", 144), ("6d", 160), ("q", 160), ("rpy", 160), ("q", 256), ("dq", 256), ("6d", 220)])
def test_train_weight_gradients_with_many_clusters_vs_oracle_autograd(dev, rot, k):
    """the sixth 16-byte feature piece of k_bd's B role exists only for '6d' with K >= 143 (K x 72 / 4 > 2560), and
    nothing held it to more than a smoke test: Adam's first step is +-lr whatever a gradient's size and its next ones divide by running
    moments with the same error, so parameters after three steps barely notice a gradient that lost the contribution of pose rows
    143..159 (measured here: even real-size clusters leave 0.01-0.04 % of the weights a full 2 lr apart after ONE step between two
    correct float32 implementations -- exact cancellations -- for 'q' and '6d' alike).  So the gradients themselves: after one optimizer
    step from zero moments the plan's first moments are (1 - beta1) g, and they sit in the caller's workspace.  On a 16384-point
    frame (56+ points per cluster) every weight gradient of every layer against torch autograd on the oracle's MLP + pose map, both
    pulled back from the SAME dL/d[R|t] (the plan's own, from `probe`, itself held to the oracle's with the usual tolerance: the Chamfer
    gradient is a sum of +-1/N signs, and one point whose coordinate sits within an ulp of its neighbour's flips a sign between any two
    float32 implementations -- at K = 160 on this frame exactly one does, 2 / 16384 in one translation gradient, which a comparison
    through the loss would have to tolerate in every layer).  An element's error is at most 5e-6 of the tensor's largest gradient
    (measured 1e-7 .. 3e-7: float32 sums over up to 160 rows / 768 units in another order), the encoder's -- what the sixth piece
    feeds: dW1 = g_x1^T . features -- included.  Round 5: up to 256 clusters (eight feature pieces; VERDICT r4 item 6 -- the
    reference's train() takes any K, parameters.json stops at 45)."""
    from autourdf_amd import ops
    from autourdf_amd.synthetic import initial_segmentation, make_sequence
    from oracle import registration
    from oracle.chamfer import chamfer_distance
    seq = make_sequence("franka", 23, 2, 16384)
    mats, cl, _ = initial_segmentation(seq[0], k, seed=4)
    assert min(len(c) for c in cl) >= 20
    m = torch.tensor(mats, dtype=torch.float32)
    y = torch.tensor(seq[1], dtype=torch.float32)
    clusters = [torch.tensor(c, dtype=torch.float32) for c in cl]
    hidden = 256
    torch.manual_seed(11)
    model = _oracle_model(rot, hidden)
    order = _order(rot)
    sd = model.state_dict()
    params = [sd[n].clone().to(dev) for n in order]
    pts, off = ops.pack_clusters(clusters, dev)
    plan = ops.TrainPlan(rot, k, hidden, pts.shape[0], y.shape[0], epochs=1, use_graph=False, device=dev)
    gm2, gpred, gloss, ggrad = plan.probe(m.to(dev), y.to(dev), pts, off, params)
    bm, bp, res, lh, _ = plan.run(m.to(dev), y.to(dev), pts, off, params, lr=1e-3)
    torch.cuda.synchronize()
    moments = _plan_first_moments(plan, order, {n: tuple(sd[n].shape) for n in order})
    m2 = registration.pose_forward(m, model, rot)
    m2.retain_grad()
    pred = torch.cat(registration.calculate_pc(clusters, m2))
    loss, _ = chamfer_distance(pred.unsqueeze(0), y.unsqueeze(0), norm=1)
    loss.backward(retain_graph=True)
    assert abs(float(lh[0]) - float(loss)) <= 2e-5 * float(loss)
    np.testing.assert_allclose(bm.cpu().numpy(), m2.detach().numpy(), atol=1e-5)
    # dL/d[R|t]: the plan's against the oracle's; a flipped sign of one point moves one entry by 2 / N (x its coordinate for dL/dR)
    gp, go = ggrad.cpu().numpy()[:, :3, :], m2.grad.numpy()[:, :3, :]
    assert int((np.abs(gp - go) > 1e-4 * np.abs(go) + 2e-6).sum()) <= 4, np.abs(gp - go).max()
    assert float(np.abs(gp - go).max()) <= 3.0 / y.shape[0]
    # the MLP + pose map's backward from the plan's dL/d[R|t]
    model.zero_grad()
    g_in = torch.zeros_like(m2)
    g_in[:, :3, :] = torch.from_numpy(gp)
    m2.backward(gradient=g_in)
    named = dict(model.named_parameters())
    for name in order:
        g_ref = named[name].grad.numpy()
        g_plan = moments[name] / np.float32(1.0 - 0.9)
        scale = float(np.abs(g_ref).max())
        assert scale > 0, name
        err = float(np.abs(g_plan - g_ref).max())
        assert err <= 5e-6 * scale, (name, err, scale)


@pytest.mark.parametrize("rot", ["q", "dq"])
@pytest.mark.parametrize("graph", [False, True])
def test_train_short_trajectory_vs_oracle(dev, golden, rot, graph):
    """E epochs with pinned weights: loss trajectory (rtol 1e-5), best pose (1e-5), updated weights."""
    from autourdf_amd import ops
    from oracle import models, registration
    g, _, _ = _train_case(golden, rot)
    hidden, E = 64, 6
    torch.manual_seed(4)
    model = models.QRegMLP(True, hidden) if rot == "q" else models.DQRegMLP(hidden)
    for p in model.parameters():
        p.data.mul_(0.2)
    order = ops.Q_PARAM_ORDER if rot == "q" else ops.DQ_PARAM_ORDER
    m, y = torch.from_numpy(g[f"{rot}_m"]), torch.from_numpy(g[f"{rot}_y"])
    clusters = [torch.from_numpy(c) for c in _split(g[f"{rot}_local"], g[f"{rot}_offsets"])]
    params = [model.state_dict()[k].clone().to(dev) for k in order]
    pts, off = ops.pack_clusters(clusters, dev)
    plan = ops.TrainPlan(rot, len(clusters), hidden, pts.shape[0], y.shape[0], epochs=E, use_graph=graph, device=dev)
    best_m, best_pred, result, lh, lrh = plan.run(m.to(dev), y.to(dev), pts, off, params)
    pred_np, o_best_m, o_min, hist = registration.train(m, y, model, clusters, rot=rot, epochs=E)
    np.testing.assert_allclose(lh.cpu().numpy(), np.array(hist["loss"], np.float32), rtol=1e-5)
    np.testing.assert_allclose(lrh.cpu().numpy(), np.array(hist["lr"], np.float32), rtol=1e-6)
    assert abs(result[0].item() - o_min) <= 1e-5 * abs(o_min)
    np.testing.assert_allclose(best_m.cpu().numpy(), o_best_m.detach().numpy(), atol=1e-5)
    np.testing.assert_allclose(best_pred.cpu().numpy(), np.concatenate(pred_np), atol=1e-5)
    for k, p in zip(order, params):                                # Adam-updated weights written back
        np.testing.assert_allclose(p.cpu().numpy(), model.state_dict()[k].numpy(), rtol=1e-3, atol=2e-6)


@pytest.mark.parametrize("rot,hidden", [("q", 64), ("dq", 64), ("6d", 128), ("rpy", 3)])
def test_train_is_bit_reproducible_and_graph_equals_eager(dev, golden, rot, hidden):
    """Eager launches, the captured graph, and the captured graph again: the same bits, for every pose representation
    (rpy at the reference's hidden 3: the zero-padded run)."""
    from autourdf_amd import ops
    g, _, _ = _train_case(golden, "q")
    torch.manual_seed(5)
    model = _oracle_model(rot, hidden)
    order = _order(rot)
    m, y = torch.from_numpy(g["q_m"]).to(dev), torch.from_numpy(g["q_y"]).to(dev)
    clusters = [torch.from_numpy(c) for c in _split(g["q_local"], g["q_offsets"])]
    pts, off = ops.pack_clusters(clusters, dev)
    outs = []
    for graph in (False, True, True):
        params = [model.state_dict()[k].clone().to(dev) for k in order]
        plan = ops.TrainPlan(rot, len(clusters), hidden, pts.shape[0], y.shape[0], epochs=40, use_graph=graph, device=dev)
        bm, bp, res, lh, _ = plan.run(m, y, pts, off, params)
        outs.append((bm.cpu(), lh.cpu(), torch.cat([p.flatten() for p in params]).cpu()))
    for o in outs[1:]:
        assert torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1]) and torch.equal(o[2], outs[0][2])


@pytest.mark.parametrize("rot,hidden,lr,stop", [("q", 512, 2e-4, 200), ("dq", 64, 2e-4, 200), ("6d", 128, 2e-4, 200), ("q", 64, 0.2, 3)])
def test_train_fused_backward_launch_is_bit_identical(dev, golden, rot, hidden, lr, stop, monkeypatch):
    """CREG_FUSED_GBD=1 (round 5 experiment, off by default because it measured slower: profiles/r05_fused_gbd_ab.log): the gradient
    reduction and the backward as ONE launch -- the consumers prefetch their parameter rows, wait for the gradient role's blocks inside
    the launch, then read the gradients with sc1 loads.  Same arithmetic in the same order: every output bit equals the two-launch
    plan's, eager and captured, also when the train stops early (the gradient blocks of a stopped train still count themselves in)."""
    from autourdf_amd import ops
    g, _, _ = _train_case(golden, "q")
    torch.manual_seed(5)
    model = _oracle_model(rot, hidden)
    if lr > 0.1:
        for p in model.parameters():
            p.data.mul_(0.2)
    order = _order(rot)
    m, y = torch.from_numpy(g["q_m"]).to(dev), torch.from_numpy(g["q_y"]).to(dev)
    clusters = [torch.from_numpy(c) for c in _split(g["q_local"], g["q_offsets"])]
    pts, off = ops.pack_clusters(clusters, dev)
    outs = []
    for fused, graph in (("0", True), ("1", False), ("1", True)):
        monkeypatch.setenv("CREG_FUSED_GBD", fused)
        params = [model.state_dict()[k].clone().to(dev) for k in order]
        plan = ops.TrainPlan(rot, len(clusters), hidden, pts.shape[0], y.shape[0], epochs=40, use_graph=graph, device=dev)
        bm, bp, res, lh, lrh = plan.run(m, y, pts, off, params, lr=lr, patience=1 if lr > 0.1 else 5, stop=stop)
        outs.append((bm.cpu(), lh.cpu(), torch.cat([p.flatten() for p in params]).cpu(), res.cpu(), lrh.cpu()))
    if lr > 0.1:
        assert int(outs[0][3][1]) < 40                      # it did stop early
    for o in outs[1:]:
        assert all(torch.equal(a, b, ) or (torch.isnan(a) == torch.isnan(b)).all() and torch.equal(torch.nan_to_num(a), torch.nan_to_num(b)) for a, b in zip(o, outs[0]))


@pytest.mark.parametrize("rot,k,lr,stop,batch", [("q", None, 2e-4, 200, 1), ("dq", None, 2e-4, 200, 3), ("q", 40, 2e-4, 200, 2), ("q", None, 0.2, 3, 1)])
def test_train_next_hidden_activation_inside_the_backward_launch_is_bit_identical(dev, golden, rot, k, lr, stop, batch, monkeypatch):
    """CREG_L2_IN_BD (round 6): the D role of k_bd goes on to the next hidden activation from the updated rows it still holds and the next
    encoder activation the B role of the same launch hands over (write-through stores, one arrival counter, sc1 loads) -- the k_l2 launch
    of every epoch disappears.  Same tiles, same k order, same cross-wave sums: every output bit equals the five-launch plan's, eager and
    captured, single and batched, more than 32 pose rows (two passes of the row loop), also when the train stops early and when a train is
    resumed from a state (the arrival count follows the step count)."""
    from autourdf_amd import ops
    g, _, _ = _train_case(golden, "q")
    torch.manual_seed(5)
    hidden = 512
    models = [_oracle_model(rot, hidden) for _ in range(batch)]
    if lr > 0.1:
        for mdl in models:
            for p in mdl.parameters():
                p.data.mul_(0.2)
    order = _order(rot)
    m, y = torch.from_numpy(g["q_m"]).to(dev), torch.from_numpy(g["q_y"]).to(dev)
    clusters = [torch.from_numpy(c) for c in _split(g["q_local"], g["q_offsets"])]
    if k is not None:                                       # more pose rows than one pass of the row loop (32): every cluster cut in pieces
        per = -(-k // len(clusters))
        pieces, rows = [], []
        for c, mrow in zip(clusters, m):
            for part in torch.chunk(c, per):
                pieces.append(part); rows.append(mrow)
        clusters, m = pieces, torch.stack(rows)
        assert len(clusters) > 32
    pts, off = ops.pack_clusters(clusters, dev)
    outs = []
    for l2in, graph in (("0", True), ("1", False), ("1", True)):
        monkeypatch.setenv("CREG_L2_IN_BD", l2in)
        params = [[mdl.state_dict()[kk].clone().to(dev) for kk in order] for mdl in models]
        plan = ops.TrainPlan(rot, len(clusters), hidden, pts.shape[0], y.shape[0], epochs=40, use_graph=graph, device=dev, batch=batch)
        res = plan.run_batch([(m, y, pts, off, pr) for pr in params], lr=lr, patience=1 if lr > 0.1 else 5, stop=stop)
        outs.append([t.cpu() for r in res for t in (r[0], r[3], r[2], r[4])] + [torch.cat([p.flatten() for pr in params for p in pr]).cpu()])
    if lr > 0.1:
        assert int(outs[0][2][1]) < 40                      # it did stop early
    same = lambda a, b: torch.equal(a, b) or ((torch.isnan(a) == torch.isnan(b)).all() and torch.equal(torch.nan_to_num(a), torch.nan_to_num(b)))
    for o in outs[1:]:
        assert all(same(a, b) for a, b in zip(o, outs[0]))
    if batch == 1 and lr < 0.1 and k is None:               # resumed: 40 epochs == 15 + 25 in this mode too (control exact, numbers to rounding -- see test_train_resume_continues_a_run)
        monkeypatch.setenv("CREG_L2_IN_BD", "1")
        lr0 = float(np.float32(lr))
        plan = ops.TrainPlan(rot, len(clusters), hidden, pts.shape[0], y.shape[0], epochs=40, use_graph=False, device=dev)
        pa = [models[0].state_dict()[kk].clone().to(dev) for kk in order]
        z = lambda ps: [torch.zeros_like(p) for p in ps]
        bm_a, _, _, lh_a, _, st_a = plan.resume(m, y, pts, off, pa, {"exp_avg": z(pa), "exp_avg_sq": z(pa), "lr": lr0}, 40)
        assert torch.equal(bm_a.cpu(), outs[0][0]) and same(lh_a.cpu(), outs[0][1])
        pb = [models[0].state_dict()[kk].clone().to(dev) for kk in order]
        bm_1, bp_1, _, _, _, st_1 = plan.resume(m, y, pts, off, pb, {"exp_avg": z(pb), "exp_avg_sq": z(pb), "lr": lr0}, 15)
        bm_b, _, _, lh_b, _, st_b = plan.resume(m, y, pts, off, pb, st_1, 25, best=(bm_1, bp_1))
        assert st_b["step"] == 40 and st_b["lr"] == st_a["lr"] and st_b["best_epoch"] == st_a["best_epoch"]
        np.testing.assert_allclose(lh_b[15:40].cpu().numpy(), lh_a[15:40].cpu().numpy(), rtol=1e-4)


@pytest.mark.parametrize("lr,stop,expect_stop", [(0.2, 3, True), (5e-2, 4, False)])
def test_train_early_stop_and_scheduler(dev, golden, lr, stop, expect_stop):
    """Large lr: the loss stops improving, ReduceLROnPlateau(patience=1) cuts lr, and with stop=3 the
    loop breaks early (before that epoch's backward); with lr 5e-2 it runs all epochs through
    several lr cuts.  Epoch count, lr trajectory and the NaN tail of the history must match."""
    from autourdf_amd import ops
    from oracle import models, registration
    g, _, _ = _train_case(golden, "q")
    torch.manual_seed(6)
    model = models.QRegMLP(True, 64)
    for p in model.parameters():
        p.data.mul_(0.2)
    m, y = torch.from_numpy(g["q_m"]), torch.from_numpy(g["q_y"])
    clusters = [torch.from_numpy(c) for c in _split(g["q_local"], g["q_offsets"])]
    pts, off = ops.pack_clusters(clusters, dev)
    params = [model.state_dict()[k].clone().to(dev) for k in ops.Q_PARAM_ORDER]
    plan = ops.TrainPlan("q", len(clusters), 64, pts.shape[0], y.shape[0], epochs=60, use_graph=True, device=dev)
    _, _, res, lh, lrh = plan.run(m.to(dev), y.to(dev), pts, off, params, lr=lr, patience=1, stop=stop)
    _, _, o_min, hist = registration.train(m, y, model, clusters, rot="q", epochs=60, learning_rate=lr,
                                           scheduler_patience=1, stop=stop)
    n = len(hist["loss"])
    assert (n < 60) == expect_stop
    assert int(res[1].item()) == n                                 # same (early-)stop epoch
    assert torch.isnan(lh[n:]).all() and not torch.isnan(lh[:n]).any()
    assert len(set(np.round(hist["lr"], 9))) >= 2                  # the scheduler did cut the lr
    np.testing.assert_allclose(lrh.cpu().numpy()[:n], np.array(hist["lr"], np.float32), rtol=1e-6)
    for k, p in zip(ops.Q_PARAM_ORDER, params):                    # no update after the break
        np.testing.assert_allclose(p.cpu().numpy(), model.state_dict()[k].numpy(), rtol=5e-3, atol=1e-5)


def test_train_300_epochs_golden_reference(dev, golden):
    """Full 300-epoch loop of the REFERENCE (hidden 32 fixture is re-run at hidden 64 by the oracle on
    the fly, since the engine tiles hidden in 64s): min_loss within 2 %, best pose within 2e-3 --
    300 Adam steps through argmin switches amplify 1-ulp differences, per-step parity is pinned above."""
    from autourdf_amd import ops
    from oracle import models, registration
    g, _, _ = _train_case(golden, "q")
    torch.manual_seed(7)
    model = models.QRegMLP(True, 64)
    for p in model.parameters():
        p.data.mul_(0.2)
    m, y = torch.from_numpy(g["q_m"]), torch.from_numpy(g["q_y"])
    clusters = [torch.from_numpy(c) for c in _split(g["q_local"], g["q_offsets"])]
    pts, off = ops.pack_clusters(clusters, dev)
    params = [model.state_dict()[k].clone().to(dev) for k in ops.Q_PARAM_ORDER]
    plan = ops.TrainPlan("q", len(clusters), 64, pts.shape[0], y.shape[0], epochs=300, use_graph=True, device=dev)
    best_m, _, res, lh, _ = plan.run(m.to(dev), y.to(dev), pts, off, params)
    _, o_best_m, o_min, hist = registration.train(m, y, model, clusters, rot="q")
    assert abs(res[0].item() - o_min) < 0.02 * o_min
    assert lh[0].item() == pytest.approx(hist["loss"][0], rel=1e-5)
    assert np.abs(best_m.cpu().numpy() - o_best_m.detach().numpy()).max() < 2e-3


# ------------------------------------------------------------------------------------------ N1
@pytest.mark.parametrize("n,m", [(500, 64), (5000, 4096), (20000, 512)])
def test_fps_indices_bit_exact_vs_oracle(dev, n, m):
    from autourdf_amd.fps import farthest_point_sample
    from oracle import kmeans
    X = np.random.default_rng(n).normal(size=(n, 3))
    if n == 500:
        X[100:120] = X[0:20]                              # duplicates: ties must resolve to the first index
    np.testing.assert_array_equal(farthest_point_sample(X, m), kmeans.farthest_point_sample(X, m))


def test_segments_sample_size_uses_fps(dev, tmp_path):
    from autourdf_amd.cluster_icp import PointCloud
    from oracle import kmeans
    X = np.random.default_rng(0).normal(size=(3000, 3))
    out = PointCloud(X).farthest_point_down_sample(256)
    np.testing.assert_array_equal(out.points, X[kmeans.farthest_point_sample(X, 256)])


@pytest.mark.parametrize("rot", ["q", "6d"])
def test_batched_plan_is_bit_identical_to_separate_runs(dev, golden, rot):
    """creg_train_plan_run_batch: 3 independent problems (different weights, targets and cluster sizes)
    advanced per launch give exactly the results of 3 separate plans."""
    from autourdf_amd import ops
    g, _, _ = _train_case(golden, "q")
    base = [torch.from_numpy(c) for c in _split(g["q_local"], g["q_offsets"])]
    y0, m0 = torch.from_numpy(g["q_y"]), torch.from_numpy(g["q_m"])
    n = sum(len(c) for c in base)
    problems, singles = [], []
    for b in range(3):
        torch.manual_seed(20 + b)
        model = _oracle_model(rot, 64)
        flat = torch.cat(base)
        cuts = sorted(torch.randperm(n - 1)[: len(base) - 1].add(1).tolist())          # different cluster sizes
        cl = [flat[a:z] for a, z in zip([0] + cuts, cuts + [n])]
        y = (y0 + 0.01 * b).to(dev)
        pts, off = ops.pack_clusters(cl, dev)
        mk = lambda: [model.state_dict()[k].clone().to(dev) for k in ops.Q_PARAM_ORDER]
        problems.append((m0.to(dev), y, pts, off, mk()))
        singles.append((m0.to(dev), y, pts, off, mk()))
    plan_b = ops.TrainPlan(rot, len(base), 64, n, y0.shape[0], epochs=30, use_graph=True, device=dev, batch=3)
    outs_b = plan_b.run_batch(problems)
    plan_1 = ops.TrainPlan(rot, len(base), 64, n, y0.shape[0], epochs=30, use_graph=True, device=dev)
    for b in range(3):
        o1 = plan_1.run(*singles[b])
        for tb, t1 in zip(outs_b[b], o1):
            assert torch.equal(tb, t1) or (torch.isnan(tb) == torch.isnan(t1)).all() and torch.equal(tb.nan_to_num(), t1.nan_to_num())
        for pb, p1 in zip(problems[b][4], singles[b][4]):
            assert torch.equal(pb, p1)
    assert not torch.equal(outs_b[0][0], outs_b[1][0])


@pytest.mark.parametrize("rot", ["q", "dq"])
def test_graph_branches_stress_bit_identical_to_single_runs(dev, rot):
    """Two (and three) concurrently running chains of the train kernels against single runs, many repetitions at
    the full hidden size: this is the test that caught the LDS-DMA tail overrun of k_bwd2 (the last DMA instruction
    wrote a full 1 KiB over the next LDS array; which write landed last was decided by memory contention from the
    other chain, so 1-10 % of runs differed).  Any timing-dependent hazard inside a kernel shows up here."""
    from autourdf_amd import ops
    from autourdf_amd.synthetic import initial_segmentation, make_sequence
    from oracle import models
    seq = make_sequence("wx200_5", 3, 3, 1024)
    mats, cl, _ = initial_segmentation(seq[0], 8, seed=1)
    m = torch.tensor(mats, dtype=torch.float32, device=dev)
    ys = [torch.tensor(seq[1] + 0.001 * b, dtype=torch.float32, device=dev) for b in range(5)]
    pts, off = ops.pack_clusters([torch.tensor(c, dtype=torch.float32) for c in cl], dev)
    torch.manual_seed(3)
    model, order = (models.QRegMLP(True, 512), ops.Q_PARAM_ORDER) if rot == "q" else (models.DQRegMLP(512), ops.DQ_PARAM_ORDER)
    mk = lambda: [model.state_dict()[k].clone().to(dev) for k in order]
    single = ops.TrainPlan(rot, 8, 512, pts.shape[0], ys[0].shape[0], epochs=40, use_graph=True, device=dev)
    ref = [[t.cpu() for t in single.run(m, ys[b], pts, off, mk())] for b in range(5)]
    for batch, branches, reps in ((5, 2, 12), (4, 2, 6), (5, 3, 4), (5, -2, 6), (5, -3, 4)):      # (negative: chain-stream mode, every chain its own graph on its own stream)
        for _ in range(reps):
            plan = ops.TrainPlan(rot, 8, 512, pts.shape[0], ys[0].shape[0], epochs=40, use_graph=True, device=dev,
                                 batch=batch, graph_branches=branches)
            outs = plan.run_batch([(m, ys[b], pts, off, mk()) for b in range(batch)])
            for b in range(batch):
                for got, want in zip(outs[b], ref[b]):
                    assert torch.equal(got.cpu().nan_to_num(), want.nan_to_num())


def test_graph_branches_bench_shape_bit_identical_to_single_runs(dev):
    """The exact shape bench.py runs (N=4096, K=20, hidden 512, 5 problems as 3 + 2): two chains vs single runs."""
    from autourdf_amd import ops
    from autourdf_amd.synthetic import initial_segmentation, make_sequence
    from oracle import models
    seq = make_sequence("wx200_5", 5, 2, 4096)
    mats, cl, _ = initial_segmentation(seq[0], 20, seed=0)
    m = torch.tensor(mats, dtype=torch.float32, device=dev)
    ys = [torch.tensor(seq[1] + 0.0005 * b, dtype=torch.float32, device=dev) for b in range(5)]
    pts, off = ops.pack_clusters([torch.tensor(c, dtype=torch.float32) for c in cl], dev)
    torch.manual_seed(1)
    model = models.QRegMLP(True, 512)
    mk = lambda: [model.state_dict()[k].clone().to(dev) for k in ops.Q_PARAM_ORDER]
    single = ops.TrainPlan("q", 20, 512, 4096, 4096, epochs=60, use_graph=True, device=dev)
    ref = [[t.cpu() for t in single.run(m, ys[b], pts, off, mk())] for b in range(5)]
    for _ in range(5):
        plan = ops.TrainPlan("q", 20, 512, 4096, 4096, epochs=60, use_graph=True, device=dev, batch=5)       # default: 2 chain streams
        outs = plan.run_batch([(m, ys[b], pts, off, mk()) for b in range(5)])
        for b in range(5):
            for got, want in zip(outs[b], ref[b]):
                assert torch.equal(got.cpu().nan_to_num(), want.nan_to_num())


def test_chain_streams_on_a_side_stream_and_plan_destroyed_while_busy(dev):
    """Chain streams fork from / join the CALLER's stream, whatever it is: a batch of 5 (two chains) and of 8 (three) enqueued on a
    torch side stream give the single runs' bits; the plan is dropped right after the enqueue (its destroy waits for its chains),
    and the results are read only after the side stream has been synchronised."""
    from autourdf_amd import ops
    from autourdf_amd.synthetic import initial_segmentation, make_sequence
    from oracle import models
    seq = make_sequence("wx200_5", 7, 2, 2048)
    mats, cl, _ = initial_segmentation(seq[0], 10, seed=3)
    m = torch.tensor(mats, dtype=torch.float32, device=dev)
    ys = [torch.tensor(seq[1] + 0.0007 * b, dtype=torch.float32, device=dev) for b in range(8)]
    pts, off = ops.pack_clusters([torch.tensor(c, dtype=torch.float32) for c in cl], dev)
    torch.manual_seed(4)
    model = models.QRegMLP(True, 128)
    mk = lambda: [model.state_dict()[k].clone().to(dev) for k in ops.Q_PARAM_ORDER]
    single = ops.TrainPlan("q", 10, 128, pts.shape[0], 2048, epochs=110, use_graph=True, device=dev)       # 2 graphs of 50 + 10 eager epochs
    ref = [[t.cpu() for t in single.run(m, ys[b], pts, off, mk())] for b in range(8)]
    side = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    for batch, chains in ((5, 2), (8, 3)):
        with torch.cuda.stream(side):
            plan = ops.TrainPlan("q", 10, 128, pts.shape[0], 2048, epochs=110, use_graph=True, device=dev, batch=batch)
            assert plan.info["graph_branches"] == chains
            params = [mk() for _ in range(batch)]
            outs = plan.run_batch([(m, ys[b], pts, off, params[b]) for b in range(batch)])
            ws = plan.ws                                  # (the workspace outlives the plan object: the enqueued work still uses it)
            del plan
        side.synchronize()
        for b in range(batch):
            for got, want in zip(outs[b], ref[b]):
                assert torch.equal(got.cpu().nan_to_num(), want.nan_to_num())
        del ws


def test_graph_branches_big_frame_default_is_three_chains_and_bit_identical_to_single_runs(dev):
    """Frames above 4096 points default to three chains (2 + 2 + 1 of 5 problems) from three problems on; the 4096-point shape runs
    one chain up to 4 problems, two for 5-7, three from 8 (chain streams: train_engine.hip, creg_train_plan_create)."""
    from autourdf_amd import ops
    from autourdf_amd.synthetic import initial_segmentation, make_sequence
    from oracle import models
    n = 6000
    seq = make_sequence("wx200_5", 5, 2, n)
    mats, cl, _ = initial_segmentation(seq[0], 12, seed=0)
    m = torch.tensor(mats, dtype=torch.float32, device=dev)
    ys = [torch.tensor(seq[1] + 0.0005 * b, dtype=torch.float32, device=dev) for b in range(5)]
    pts, off = ops.pack_clusters([torch.tensor(c, dtype=torch.float32) for c in cl], dev)
    torch.manual_seed(2)
    model = models.QRegMLP(True, 64)
    mk = lambda: [model.state_dict()[k].clone().to(dev) for k in ops.Q_PARAM_ORDER]
    single = ops.TrainPlan("q", 12, 64, n, n, epochs=40, use_graph=True, device=dev)
    assert single.info["graph_branches"] == 1
    assert ops.TrainPlan("q", 12, 64, 4096, 4096, epochs=40, use_graph=True, device=dev, batch=5).info["graph_branches"] == 2
    assert ops.TrainPlan("q", 12, 64, n, n, epochs=40, use_graph=True, device=dev, batch=2).info["graph_branches"] == 1
    assert ops.TrainPlan("q", 12, 64, 4096, 4096, epochs=40, use_graph=True, device=dev, batch=4).info["graph_branches"] == 1
    assert ops.TrainPlan("q", 12, 64, 4096, 4096, epochs=40, use_graph=True, device=dev, batch=8).info["graph_branches"] == 3
    ref = [[t.cpu() for t in single.run(m, ys[b], pts, off, mk())] for b in range(5)]
    for _ in range(3):
        plan = ops.TrainPlan("q", 12, 64, n, n, epochs=40, use_graph=True, device=dev, batch=5)
        assert plan.info["graph_branches"] == 3
        outs = plan.run_batch([(m, ys[b], pts, off, mk()) for b in range(5)])
        for b in range(5):
            for got, want in zip(outs[b], ref[b]):
                assert torch.equal(got.cpu().nan_to_num(), want.nan_to_num())


@pytest.mark.parametrize("rot,k,hidden,n_pred,n_tgt,lattice", [
    ("q", 20, 512, 4096, 4096, False),      # the bench shape
    ("dq", 8, 64, 1000, 777, False),        # ragged last block of targets
    ("q", 5, 64, 300, 50, False),           # fewer targets than one block
    ("q", 6, 64, 2048, 4096, True),         # lattice clouds: exact distance ties everywhere
    ("dq", 3, 128, 513, 4033, False),
    ("q", 4, 64, 700, 1, False),            # a single target
    ("q", 70, 64, 4096, 4096, False),       # 64 + 70 = 134 predicted blocks: three boxes per lane (round 2: exhaustive above 128)
    ("q", 150, 64, 9600, 4000, False),      # 150 + 150 = 300 predicted blocks: five boxes per lane
    ("q", 12, 64, 20000, 20000, False),     # > 16384 targets: two chunks of k-d leaves (the second one ragged), NBT = 2
    ("dq", 128, 64, 32768, 32768, False),   # the chain32 shape (BASELINE configs[4] scaled down): 2 chunks, 128 + 128 predicted blocks
    ("dq", 1, 64, 4096, 3000, False),       # one cluster of 64 blocks: six k-d levels
    ("q", 9, 64, 640, 900, False),          # cluster sizes forced to multiples of 64 below + an empty cluster
    ("q", 10, 64, 9000, 6000, False),       # > 4096 targets: 256-point blocks, four points per lane and visit
    ("dq", 40, 64, 16384, 16384, False),    # the franka-shaped config (BASELINE configs[2])
    ("q", 3, 64, 500, 5000, False),         # big frame, tiny clusters (one padded 256-block each)
    ("q", 2, 64, 16000, 4097, True),        # lattice ties with 256-point blocks; one cluster of 32 blocks
    ("q", 4, 64, 3000, 40000, True),        # lattice ties across three chunks of the frame
])
def test_train_pruned_search_bit_identical_to_exhaustive(dev, rot, k, hidden, n_pred, n_tgt, lattice):
    """nn_search 0 (predicted -> target direction over the Morton-sorted, boxed target frame) against nn_search 1
    (exhaustive both ways): every output of the plan -- poses, best prediction, loss and lr history, trained
    parameters -- must be identical bit for bit; the pruning is exact, ties on the original index included."""
    from autourdf_amd import ops
    from oracle import models
    g = torch.Generator().manual_seed(n_pred * 7 + n_tgt)
    if lattice:
        y = torch.randint(0, 12, (n_tgt, 3), generator=g).float() * 0.03125
        flat = torch.randint(0, 12, (n_pred, 3), generator=g).float() * 0.03125
    else:
        y = torch.rand(n_tgt, 3, generator=g) * 0.4
        flat = y[torch.randint(0, n_tgt, (n_pred,), generator=g)] + 0.004 * torch.randn(n_pred, 3, generator=g)
    cuts = sorted(torch.randperm(n_pred - 1, generator=g)[: k - 1].add(1).tolist())
    if n_pred == 640:
        cuts = [64, 128, 128, 320, 384, 448, 512, 576]          # whole blocks only, cluster 2 empty
    m = torch.eye(4).repeat(k, 1, 1)
    cl = []
    for a, z in zip([0] + cuts, cuts + [n_pred]):
        c = flat[a:z]
        ctr = c.mean(0) if z > a else torch.zeros(3)
        m[len(cl), :3, 3] = ctr
        cl.append(c if lattice else c - ctr)          # lattice: identity poses keep the coordinates exact
    if lattice:
        m[:, :3, 3] = 0
    pts, off = ops.pack_clusters(cl, dev)
    torch.manual_seed(5)
    model, order = (models.QRegMLP(True, hidden), ops.Q_PARAM_ORDER) if rot == "q" else (models.DQRegMLP(hidden), ops.DQ_PARAM_ORDER)
    outs, rows = [], []
    for mode in (0, 2, 1):
        params = [model.state_dict()[key].clone().to(dev) for key in order]
        plan = ops.TrainPlan(rot, k, hidden, n_pred, n_tgt, epochs=25, use_graph=True, device=dev, nn_search=mode)
        rows.append(plan.info["nn_queries_per_wave"] == 16)
        o = plan.run(m.to(dev), y.to(dev), pts, off, params)
        outs.append([t.cpu() for t in o] + [t.cpu() for t in params])
    assert not rows[1] and not rows[2]
    for a, b in zip(outs[1], outs[2]):                   # four queries per wave against exhaustive: every bit
        assert torch.equal(a.nan_to_num(), b.nan_to_num())
    # nn_search 0: the same unless the plan took the sixteen-queries-per-wave search (round 5: frames above 4096 points, or everywhere
    # under CREG_NN_ROWS=1), whose loss partials are summed per 16-slot group -- loss history and min_loss to 1e-6 then, everything
    # else (poses, best cloud, lr history, every trained parameter) still bit for bit
    # (the loss feeds comparisons -- best tracking, ReduceLROnPlateau, early stop -- so a last-bit difference AT one of them
    #  changes lr or the returned pose from there on.  Bit equality of everything else is therefore asserted when the two runs took the same
    #  decisions (same lr history, same best epoch), which is the rule; a run that did not is held to the loss tolerance on what it returns.)
    same_decisions = (not rows[0]) or (torch.equal(outs[0][4].nan_to_num(), outs[2][4].nan_to_num()) and float(outs[0][2][3]) == float(outs[2][2][3]))
    for i, (a, b) in enumerate(zip(outs[0], outs[2])):
        if rows[0] and i in (2, 3):
            assert torch.allclose(a.nan_to_num(), b.nan_to_num(), rtol=1e-6, atol=0.0) and bool((a.isnan() == b.isnan()).all())
        elif same_decisions:
            assert torch.equal(a.nan_to_num(), b.nan_to_num()), i
        else:
            assert torch.allclose(a.nan_to_num(), b.nan_to_num(), rtol=1e-3, atol=1e-3), i
    assert torch.isfinite(outs[0][0]).all()


def test_plans_of_different_sizes_coexist(dev):
    """The dynamic-LDS limit of a kernel is per kernel, not per plan: creating a smaller plan after a larger one must
    not lower it under the larger one's sort launches (16384-point frame, then 6000, then the first again)."""
    from autourdf_amd import ops
    from oracle import models
    g = torch.Generator().manual_seed(3)
    torch.manual_seed(3)
    model = models.QRegMLP(True, 64)

    def make(n_pred, n_tgt, k):
        y = torch.rand(n_tgt, 3, generator=g)
        flat = y[torch.randint(0, n_tgt, (n_pred,), generator=g)] + 0.002 * torch.randn(n_pred, 3, generator=g)
        cl = list(flat.chunk(k))
        m = torch.eye(4).repeat(k, 1, 1)
        for i, c in enumerate(cl):
            m[i, :3, 3] = c.mean(0)
        pts, off = ops.pack_clusters([c - c.mean(0) for c in cl], dev)
        plan = ops.TrainPlan("q", k, 64, n_pred, n_tgt, epochs=6, use_graph=True, device=dev)
        run = lambda: [t.cpu() for t in plan.run(m.to(dev), y.to(dev), pts, off,
                                                 [model.state_dict()[key].clone().to(dev) for key in ops.Q_PARAM_ORDER])]
        return run

    big = make(16384, 16384, 8)
    first = big()
    mid = make(6000, 6000, 5)
    mid()
    small = make(300, 300, 3)
    small()
    again = big()
    for a, b in zip(first, again):
        assert torch.equal(a.nan_to_num(), b.nan_to_num())


def test_train_pruned_search_random_shapes(dev):
    """20 random shapes of tests/measure/stress_pruned_search.py (sizes up to 16384, empty / whole-block clusters,
    duplicated points, lattice clouds): pruned and exhaustive plans agree bit for bit (600 shapes were run by hand,
    profiles/r01_pruned_search_stress.log)."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "measure", "stress_pruned_search.py")
    spec = importlib.util.spec_from_file_location("stress_pruned_search", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    g = torch.Generator().manual_seed(11)
    for _ in range(20):
        ok, shape, finite = mod.one(g, dev)
        assert ok and finite, shape


def test_group_to_local_batch_equals_separate_calls(dev):
    """creg_group_to_local_batch_f64 (all frames in one launch pair) against creg_group_to_local_f64 per frame."""
    from autourdf_amd import ops
    g = torch.Generator().manual_seed(9)
    n, k, B = 3001, 17, 5
    Xs = [torch.rand(n, 3, generator=g, dtype=torch.float64).to(dev) for _ in range(B)]
    labs = [torch.randint(0, k, (n,), generator=g, dtype=torch.int32).to(dev) for _ in range(B)]
    labs[2][labs[2] == 4] = 5                                   # an empty cluster
    Ms = []
    for _ in range(B):
        A = torch.linalg.qr(torch.randn(k, 3, 3, generator=g, dtype=torch.float64))[0]
        M = torch.eye(4, dtype=torch.float64).repeat(k, 1, 1)
        M[:, :3, :3] = A
        M[:, :3, 3] = torch.randn(k, 3, generator=g, dtype=torch.float64)
        Ms.append(M.to(dev))
    batch = ops.group_to_local_batch(Xs, labs, Ms)
    for X, l, M, (loc, off) in zip(Xs, labs, Ms, batch):
        loc1, off1 = ops.group_to_local(X, l, M)
        assert torch.equal(loc, loc1) and torch.equal(off, off1)


def test_kmeans_batch_single_launch_bit_identical_to_multi_launch(dev, golden):
    """creg_kmeans_lloyd_batch_f64 (one workgroup per frame, LDS resident, no host sync) against
    creg_kmeans_lloyd_f64 and the sklearn golden: labels, centres, inertia, n_iter all identical."""
    from autourdf_amd import ops
    from autourdf_amd.synthetic import initial_segmentation, make_sequence
    g = golden("kmeans_sklearn.npz")
    Xs, inits = [_cuda(g["c1_X"], dev)], [_cuda(g["c1_init"], dev)]
    for s in range(3):
        seq = make_sequence("wx200_5", 40 + s, 2, 4096)
        mats, _, _ = initial_segmentation(seq[0], 20, seed=s, iters=3)
        Xs.append(_cuda(seq[1], dev)); inits.append(_cuda(mats[:, :3, 3].copy(), dev))
    inits[3] = inits[3].clone(); inits[3][7] = torch.tensor([9.0, 9.0, 9.0], device=dev)        # forces an empty cluster
    outs = ops.kmeans_lloyd_batch(Xs, inits)
    np.testing.assert_array_equal(outs[0][1].cpu().numpy(), g["c1_labels"])
    for X, c0, o in zip(Xs, inits, outs):
        c, lab, inertia, n_iter = ops.kmeans_lloyd(X, c0)
        assert torch.equal(o[1], lab) and torch.equal(o[0], c) and torch.equal(o[2], inertia) and torch.equal(o[3], n_iter)
    small = [_cuda(g["small_X"], dev)] * 2
    o2 = ops.kmeans_lloyd_batch(small, [_cuda(g["small_init"], dev)] * 2)
    np.testing.assert_array_equal(o2[1][1].cpu().numpy(), g["small_labels"])


def test_kmeans_batch_franka_sized_frames_from_l2(dev):
    """Frames above the LDS budget (5120 < n <= 16384: BASELINE configs[2], N=16384, K=40) take the same one-launch
    path with the frame read from L2: identical to the multi-launch path (which syncs with the host) and to the
    oracle's labels, incl. a forced empty cluster."""
    from autourdf_amd import ops
    from autourdf_amd.synthetic import initial_segmentation, make_sequence
    from oracle import kmeans as okm
    Xs, inits, hosts = [], [], []
    for s in range(2):
        seq = make_sequence("franka", 50 + s, 2, 16384 if s == 0 else 9000)
        mats, _, _ = initial_segmentation(seq[0], 40, seed=s, iters=3)
        c0 = mats[:, :3, 3].copy()
        if s == 1:
            c0[5] = [7.0, 7.0, 7.0]                               # forces an empty cluster
        Xs.append(_cuda(seq[1], dev)); inits.append(_cuda(c0, dev)); hosts.append((seq[1], c0))
    for X, c0, (Xh, ch) in zip(Xs, inits, hosts):
        o = ops.kmeans_lloyd_batch([X, X], [c0, c0])
        c, lab, inertia, n_iter = ops.kmeans_lloyd(X, c0)
        for b in range(2):
            assert torch.equal(o[b][1], lab) and torch.equal(o[b][0], c) and torch.equal(o[b][2], inertia) and torch.equal(o[b][3], n_iter)
        _, olab, _, _ = okm.k_means(Xh, ch)
        np.testing.assert_array_equal(lab.cpu().numpy(), olab)


# ------------------------------------------------------------------------------------------ full-size properties
def test_full_size_c5_nn_and_kmeans_properties(dev):
    """BASELINE configs[4] sizes (N=262144, K=128): too big for the CPU oracle end to end, so
    size-independent properties + an exact spot check of 512 queries against the oracle."""
    from autourdf_amd import ops
    from autourdf_amd.synthetic import make_sequence
    from oracle import chamfer, kmeans
    n, k = 262144, 128
    fr = make_sequence("chain32", 0, 2, n)
    x, y = fr[0].astype(np.float32), fr[1].astype(np.float32)
    xd, yd = _cuda(x, dev), _cuda(y, dev)
    dx, ix, dy, iy = ops.nn_l1_bidir(xd, yd)
    # (1) role swap gives the same answer, (2) every reported distance is the distance to the reported index
    dy2, iy2, dx2, ix2 = ops.nn_l1_bidir(yd, xd)
    assert torch.equal(ix, ix2) and torch.equal(dx, dx2) and torch.equal(iy, iy2) and torch.equal(dy, dy2)
    d_chk = (xd - yd[ix]).abs()
    assert torch.equal((d_chk[:, 0] + d_chk[:, 1]) + d_chk[:, 2], dx)
    # (3) exact spot check: 512 random queries brute-forced by the oracle over all 262144 targets
    sel = np.random.default_rng(0).choice(n, 512, replace=False)
    od, oi = chamfer.nn_l1(x[sel], y)
    np.testing.assert_array_equal(ix.cpu().numpy()[sel], oi)
    np.testing.assert_array_equal(dx.cpu().numpy()[sel], od)
    # k-means: returned labels are the E-step of the returned centres (sklearn's final E-step), the MFMA
    # E-step agrees with the VALU one, and restarting from the solution stops at once (tol) without
    # increasing the inertia
    X = _cuda(fr[1], dev)
    init = X[torch.as_tensor(np.random.default_rng(1).choice(n, k, replace=False), device=dev)].clone()
    c, lab, inertia, n_iter = ops.kmeans_lloyd(X, init, max_iter=300)
    assert torch.equal(ops.kmeans_assign(X, c), lab) and torch.equal(ops.kmeans_assign(X, c, use_mfma=True), lab)
    assert len(torch.unique(lab)) == k
    sel = np.random.default_rng(2).choice(n, 4096, replace=False)
    np.testing.assert_array_equal(kmeans.assign(fr[1][sel], c.cpu().numpy()), lab.cpu().numpy()[sel])
    c2, lab2, inertia2, n_iter2 = ops.kmeans_lloyd(X, c, max_iter=300)
    assert n_iter2.item() <= 2 and inertia2.item() <= inertia.item() * (1 + 1e-12)
    assert torch.equal(ops.kmeans_assign(X, c2), lab2)
    local, off = ops.group_to_local(X, lab, torch.eye(4, dtype=torch.float64, device=dev).repeat(k, 1, 1).contiguous())
    assert off[-1].item() == n and torch.equal(torch.sort(local[:, 0])[0], torch.sort(X[:, 0])[0])     # a permutation


def test_full_size_c5_masked_icp_spot_check_vs_oracle(dev):
    """K4 at the BASELINE configs[4] shape (N=262144, K=128: ~2048-point clusters against a few thousand masked targets
    each -- the large-cluster regime, many workgroups per ICP iteration): the whole frame on the GPU, three clusters
    re-run by the oracle's open3d-style loop on their own masked targets; plus properties that hold for every cluster
    (rigid poses, world_out = pose . local, at least one iteration)."""
    from autourdf_amd import ops
    from autourdf_amd.synthetic import make_sequence
    from oracle import icp as oicp
    n, k = 262144, 128
    fr = make_sequence("chain32", 0, 2, n)
    X0, X1 = _cuda(fr[0], dev), _cuda(fr[1], dev)
    init = X0[torch.as_tensor(np.random.default_rng(1).choice(n, k, replace=False), device=dev)].clone()
    c, lab, _, _ = ops.kmeans_lloyd(X0, init, max_iter=20)
    M = torch.eye(4, dtype=torch.float64, device=dev).repeat(k, 1, 1)
    M[:, :3, 3] = c
    local, off = ops.group_to_local(X0, lab, M.contiguous())
    world32 = ops.cluster_transform(local.to(torch.float32), off, M.to(torch.float32))
    M_out, w_out, n_it = ops.masked_icp(local, world32, off, X1, M.contiguous())
    Mh, offh, localh = M_out.cpu().numpy(), off.cpu().numpy(), local.cpu().numpy()
    R = Mh[:, :3, :3]
    np.testing.assert_allclose(R @ R.transpose(0, 2, 1), np.tile(np.eye(3), (k, 1, 1)), atol=1e-12)
    np.testing.assert_allclose(np.linalg.det(R), 1.0, atol=1e-12)
    assert (n_it.cpu().numpy() >= 1).all() and np.isfinite(Mh).all()
    wh = w_out.cpu().numpy()
    for j in (0, 57, 127):
        src = localh[offh[j]:offh[j + 1]]
        np.testing.assert_allclose(wh[offh[j]:offh[j + 1]], src @ Mh[j, :3, :3].T + Mh[j, :3, 3], atol=1e-12)
        mask = oicp.aabb_mask(world32.cpu().numpy()[offh[j]:offh[j + 1]], fr[1], 1.2)
        T, _, _, it = oicp.registration_icp(src, fr[1][mask], 1.0, M[j].cpu().numpy())
        np.testing.assert_allclose(Mh[j], T, atol=1e-8)
        assert int(n_it[j]) == it


_ICP_SCREEN_CODE = '''import sys, numpy as np, torch
sys.path.insert(0, ROOT)
from autourdf_amd import ops
from autourdf_amd.synthetic import make_sequence
n, k = 65536, 24
fr = make_sequence("chain32", 3, 3, n)
dev = torch.device("cuda")
X = [torch.as_tensor(f, dtype=torch.float64, device=dev) for f in fr]
init = X[0][torch.as_tensor(np.random.default_rng(5).choice(n, k, replace=False), device=dev)].clone()
c, lab, _, _ = ops.kmeans_lloyd(X[0], init, max_iter=20)
M = torch.eye(4, dtype=torch.float64, device=dev).repeat(k, 1, 1)
M[:, :3, 3] = c
local, off = ops.group_to_local(X[0], lab, M.contiguous())
assert ops.masked_icp_regime(local.shape[0], n, k) == "many_workgroups"
out = {}
for t in (1, 2):
    world32 = ops.cluster_transform(local.to(torch.float32), off, M.to(torch.float32))
    M, w, it = ops.masked_icp(local, world32, off, X[t], M.contiguous())
    out[f"M{t}"], out[f"w{t}"], out[f"it{t}"] = M.cpu().numpy(), w.cpu().numpy(), it.cpu().numpy()
np.savez(OUT, **out)
'''


def test_masked_icp_float32_screen_changes_no_bit(dev, tmp_path):
    """k_icp_nn skips a trip of eight targets when a float32 evaluation proves that none of them can reach the running best
    (bound derived in csrc/icp.hip), and starts the running best a hair above the previous match: both only ever skip work.
    Two frames of a chain32-shaped sequence in the many-workgroup regime (N = 65536, K = 24: clusters of ~2700 points), the second
    from the first's poses so that previous matches exist -- poses, world clouds and iteration counts must be IDENTICAL to a run
    with CREG_ICP_SCREEN=0 in a child process (the knob is read once per process)."""
    import subprocess
    import sys
    outs = {}
    for scr in ("1", "0"):
        path = str(tmp_path / f"icp_{scr}.npz")
        env = dict(os.environ, CREG_ICP_SCREEN=scr)
        subprocess.run([sys.executable, "-c", f"ROOT = {ROOT!r}; OUT = {path!r}\n" + _ICP_SCREEN_CODE], check=True, env=env, timeout=400)
        outs[scr] = np.load(path)
    assert int(outs["1"]["it1"].max()) > 20 and int(outs["1"]["it2"].max()) > 20
    for key in outs["1"].files:
        np.testing.assert_array_equal(outs["1"][key], outs["0"][key], err_msg=key)


def _icp_vs_oracle(dev, clusters, mats, frame, scale=1.2, atol=1e-8):
    from autourdf_amd import ops
    from oracle import icp as oicp
    local, off = ops.pack_clusters(clusters, dev, torch.float64)
    M = _cuda(np.asarray(mats, np.float64), dev)
    world32 = ops.cluster_transform(local.to(torch.float32), off, M.to(torch.float32))
    M_out, w_out, n_it = ops.masked_icp(local, world32, off, _cuda(frame, dev), M, scale=scale)
    world_h = _split(world32.cpu().numpy(), off.cpu().numpy())
    ow, om = oicp.masked_icp(clusters, world_h, frame, mats, scale=scale)
    np.testing.assert_allclose(M_out.cpu().numpy(), om, atol=atol)
    np.testing.assert_allclose(w_out.cpu().numpy(), np.concatenate(ow), atol=atol)
    return n_it.cpu().numpy()


def test_masked_icp_fallback_paths_of_the_one_workgroup_kernel(dev):
    """The binned fast path needs <= 1024 source points and <= 4096 masked targets per cluster; a ragged segmentation (one
    cluster above the LDS source budget while the average stays below it) and a box holding more targets than fit take the
    unbinned fallback inside the same launch.  Both against the oracle."""
    from autourdf_amd.synthetic import make_sequence
    seq = make_sequence("wx200_5", 31, 2, 2600)
    X = seq[0]
    order = np.argsort(X[:, 2], kind="stable")                          # three slices along z: 1500 / 700 / 400 points
    parts = [X[order[:1500]], X[order[1500:2200]], X[order[2200:]]]
    mats = np.stack([np.eye(4) for _ in parts])
    for m, p in zip(mats, parts):
        m[:3, 3] = p.mean(0)
    clusters = [p - m[:3, 3] for p, m in zip(parts, mats)]
    assert len(X) // 3 <= 1024 < len(clusters[0])
    n_it = _icp_vs_oracle(dev, clusters, mats, seq[1])
    assert (n_it >= 1).all()
    # more masked targets than the LDS table holds: a 6000-point frame inside one generous box
    rng = np.random.default_rng(5)
    src = rng.normal(size=(600, 3)) * [0.05, 0.03, 0.02]
    frame = np.concatenate([src @ _rot_z(0.05).T + [0.004, -0.002, 0.001] + rng.normal(scale=2e-4, size=src.shape),
                            rng.normal(size=(5400, 3)) * [0.05, 0.03, 0.02]])
    assert len(frame) > 4096
    _icp_vs_oracle(dev, [src], np.eye(4)[None], frame, scale=3.0)


def _rot_z(a):
    return np.array([[np.cos(a), -np.sin(a), 0.0], [np.sin(a), np.cos(a), 0.0], [0.0, 0.0, 1.0]])


def test_masked_icp_large_regime_coordinate_pool_overflow(dev):
    """Large regime with boxes that all hold the whole frame: the clusters' masked targets exceed the coordinate pool
    (4 x frame points), so the later clusters stage their targets by gathering; same results as the oracle either way."""
    rng = np.random.default_rng(9)
    base = rng.normal(size=(1100, 3)) * [0.06, 0.04, 0.03]
    k = 6
    clusters, mats = [], []
    for c in range(k):
        m = np.eye(4)
        m[:3, :3] = _rot_z(0.01 * (c + 1))
        m[:3, 3] = [0.002 * c, -0.001 * c, 0.0005 * c]
        clusters.append(base + rng.normal(scale=1e-4, size=base.shape))
        mats.append(m)
    frame = base @ _rot_z(0.03).T + [0.003, 0.001, -0.002] + rng.normal(scale=3e-4, size=base.shape)
    frame = np.concatenate([frame, frame + rng.normal(scale=5e-4, size=frame.shape)])     # 2200 targets, all inside every box
    assert sum(len(c) for c in clusters) // k > 1024 and k * len(frame) > 4 * len(frame)
    n_it = _icp_vs_oracle(dev, clusters, np.stack(mats), frame, scale=1.5)
    assert (n_it >= 1).all()


@pytest.mark.parametrize("mfma", [False, True])
def test_kmeans_many_empty_clusters_deferred_relocation_large_frame(dev, mfma):
    """A frame large enough for the many-workgroup Lloyd path (two points per thread, hundreds of workgroups) seeded so that
    a third of the centres own no point at first: the M-step tail defers, the next launch relocates over all workgroups
    (segment maxima, rescans of the winners' segments).  Labels, iteration count and centres against the C oracle."""
    from autourdf_amd import ops
    from oracle import kmeans
    rng = np.random.default_rng(11)
    X = np.concatenate([rng.normal(size=(30000, 3)) * 0.3, rng.normal(size=(10000, 3)) * 0.1 + [2.0, 0.0, 0.0]])
    k = 24
    init = np.concatenate([X[rng.choice(len(X), k - 8, replace=False)] + 1e-4,
                           rng.normal(size=(8, 3)) * 0.01 + [40.0, 40.0, -40.0]])     # eight seeds far from every point
    c, lab, inertia, n_iter = ops.kmeans_lloyd(_cuda(X, dev), _cuda(init, dev), use_mfma=mfma)
    oc, olab, oin, oit = kmeans.k_means(X, init)
    np.testing.assert_array_equal(lab.cpu().numpy(), olab)
    assert n_iter.item() == oit and len(np.unique(olab)) == k
    np.testing.assert_allclose(c.cpu().numpy(), oc, atol=1e-11)
    np.testing.assert_allclose(inertia.item(), oin, rtol=1e-11)


def test_masked_icp_large_regime_with_an_empty_cluster(dev):
    """A cluster without points in the many-workgroup regime has no source chunk (hence no fit): it keeps its pose, reports
    one iteration, does not hold the loop open, and leaves the other clusters' results untouched."""
    from autourdf_amd import ops
    from autourdf_amd.synthetic import initial_segmentation, make_sequence
    seq = make_sequence("wx200_5", 21, 2, 4600)
    mats, clusters, _ = initial_segmentation(seq[0], 4, seed=1)
    frame = _cuda(seq[1], dev)

    def run(cl, mt):
        local, off = ops.pack_clusters(cl, dev, torch.float64)
        M = _cuda(np.asarray(mt, np.float64), dev)
        return ops.masked_icp_batch([(local, None, off, frame, M)])[0]

    cl3 = [np.concatenate([clusters[0], clusters[1]]), np.zeros((0, 3)), np.concatenate([clusters[2], clusters[3]])]
    assert sum(len(c) for c in cl3) // 3 > 1024                        # the average stays above the regime switch
    mt3 = np.stack([mats[0], np.eye(4), mats[2]])
    M3, w3, it3 = run(cl3, mt3)
    M3b, w3b, it3b = run([cl3[0], cl3[2]], np.stack([mats[0], mats[2]]))
    np.testing.assert_allclose(M3.cpu().numpy()[[0, 2]], M3b.cpu().numpy(), atol=1e-12)
    np.testing.assert_array_equal(M3.cpu().numpy()[1], np.eye(4))
    np.testing.assert_array_equal(it3.cpu().numpy()[[0, 2]], it3b.cpu().numpy())
    assert it3.cpu().numpy()[1] == 1
    np.testing.assert_allclose(w3.cpu().numpy(), w3b.cpu().numpy(), atol=1e-12)


def test_masked_icp_tiny_clusters_many_lanes_per_source(dev):
    """Clusters of 3 to 40 points: the one-workgroup kernel gives each source 64 / 32 / 16 / 8 lanes (all waves busy, the scanned
    range split that many ways, overshoot into the padding behind the target list).  Against the oracle."""
    rng = np.random.default_rng(17)
    sizes = [3, 7, 12, 20, 40, 90]
    clusters, mats, targets = [], [], []
    for c, ns in enumerate(sizes):
        ctr = np.array([0.3 * c, 0.1 * (c % 3), 0.05 * c])
        src = rng.normal(size=(ns, 3)) * [0.03, 0.02, 0.01]
        m = np.eye(4)
        m[:3, :3] = _rot_z(0.02 * (c + 1))
        m[:3, 3] = ctr
        clusters.append(src)
        mats.append(m)
        world = src @ _rot_z(0.02 * (c + 1) + 0.04).T + ctr + [0.002, -0.001, 0.001]
        targets.append(np.concatenate([world + rng.normal(scale=2e-4, size=world.shape), world[: ns // 2] + rng.normal(scale=5e-4, size=(ns // 2, 3))]))
    frame = np.concatenate(targets)[rng.permutation(sum(len(t) for t in targets))]
    n_it = _icp_vs_oracle(dev, clusters, np.stack(mats), frame, scale=1.6)
    assert (n_it >= 1).all()


def test_group_to_local_large_frame_stable_partition(dev):
    """The workgroup-per-cluster grouping of large frames (n > 16384): offsets, stable order inside a cluster and the change of
    frame against numpy."""
    from autourdf_amd import ops
    rng = np.random.default_rng(23)
    n, k = 50000, 37
    X = rng.normal(size=(n, 3))
    lab = rng.integers(0, k, size=n).astype(np.int32)
    lab[lab == 5] = 6                                                 # an empty cluster in the middle
    M = np.stack([np.eye(4) for _ in range(k)])
    for j in range(k):
        M[j, :3, :3] = _rot_z(0.1 * j)
        M[j, :3, 3] = rng.normal(size=3)
    local, off = ops.group_to_local(_cuda(X, dev), torch.as_tensor(lab, device=dev), _cuda(M, dev))
    counts = np.bincount(lab, minlength=k)
    np.testing.assert_array_equal(off.cpu().numpy(), np.concatenate([[0], np.cumsum(counts)]))
    ref = np.concatenate([(X[lab == j] - M[j, :3, 3]) @ M[j, :3, :3] for j in range(k)])      # inv(M) p for a rigid M
    np.testing.assert_allclose(local.cpu().numpy(), ref, atol=1e-12)


def test_train_plan_reports_its_search_form(dev):
    """creg_train_plan_info: a shape inside the block-pruned search's limits runs it in both directions -- since round 3 also
    70 clusters of one 64-point block each (70 + 70 = 140 predicted blocks: three boxes per lane; round 2 stopped at 128);
    a 20000-point frame (two chunks of k-d leaves; round 2 stopped at 16384); one beyond the limits (a frame above 65536
    points) says so -- and ops.TrainPlan warns -- instead of only being slower."""
    import warnings
    from autourdf_amd import ops
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        p = ops.TrainPlan("q", 20, 64, 4096, 4096, epochs=4, device=dev)
        q = ops.TrainPlan("q", 70, 64, 70 * 60, 4096, epochs=4, device=dev)
        c = ops.TrainPlan("q", 8, 64, 2048, 20000, epochs=4, device=dev)
    assert p.info["pruned_target_search"] and p.info["pruned_predicted_search"] and p.info["batch"] == 1
    assert q.info["pruned_target_search"] and q.info["pruned_predicted_search"]
    assert c.info["pruned_target_search"] and c.info["pruned_predicted_search"]
    with pytest.warns(RuntimeWarning, match="exhaustive nearest-neighbour search"):
        r = ops.TrainPlan("q", 8, 64, 2048, 70000, epochs=4, device=dev)
    assert not r.info["pruned_target_search"] and not r.info["pruned_predicted_search"]
