"""CPU: host logic and the C-ABI surface (no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_build_produces_library_and_every_declared_symbol_resolves():
    from autourdf_amd.build import build_lib
    lib_path = build_lib()
    assert os.path.exists(lib_path)
    header = open(os.path.join(ROOT, "include", "creg.h")).read()
    declared = set(re.findall(r"\b(creg_[a-z0-9_]+)\s*\(", header))
    declared -= {"creg_status"}
    from autourdf_amd import _lib
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    L = ctypes.CDLL(lib_path)
    for name in declared:
        assert getattr(L, name) is not None
    lib = _lib.load(check_device=False)
    assert lib.creg_version() >= 100


def test_product_path_has_no_cpu_fallback_and_never_imports_the_oracle():
    import torch
    from autourdf_amd import dq_func, ops
    with pytest.raises(RuntimeError):
        ops.nn_l1_bidir(torch.zeros(4, 3), torch.zeros(4, 3))
    with pytest.raises(RuntimeError):
        dq_func.transform_to_dualquat(torch.eye(4)[None])
    pkg = os.path.join(ROOT, "autourdf_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), fn
            assert "/root/reference" not in src, fn


def test_drop_in_signatures_match_the_reference_surface():
    import inspect
    from autourdf_amd import cluster_icp, dq_func, helper_functions, mlp_reg, model_utils
    sig = lambda f: list(inspect.signature(f).parameters)
    assert sig(mlp_reg.train) == ["m", "y", "model", "clusters", "stop", "learning_rate", "scheduler_patience",
                                  "scheduler_factor"]
    d = inspect.signature(mlp_reg.train).parameters
    assert (d["stop"].default, d["learning_rate"].default, d["scheduler_patience"].default,
            d["scheduler_factor"].default) == (200, 0.0002, 5, 0.7)
    assert sig(mlp_reg.calculate_pc) == ["local_clusters", "matrices"]
    assert sig(mlp_reg.resample_cluster) == ["segments", "idx", "n_clusters", "matrices", "normal", "visual"]
    assert sig(mlp_reg.match) == ["data_dir", "idx"]
    assert sig(cluster_icp.masked_icp)[:9] == ["clusters_local", "clusters_world", "step_pc_np", "matrices", "visual",
                                               "ori", "scale", "th", "colors"]
    assert sig(cluster_icp.Segments.__init__) == ["self", "data_path", "sample_size"]
    assert sig(cluster_icp.Segments.k_means_cluster)[:5] == ["self", "pc_id", "num", "normal", "colors"]
    for name in ("transform_from_rot_trans", "quaternion_conjugate", "quat_trans_to_dualquat", "rot_trans_to_dualquat",
                 "transform_to_dualquat", "dualquat_to_quat_trans", "dualquat_to_rot_trans", "dualquat_to_transform",
                 "dualquat_multiply", "dualquat_invert", "point_to_dualquat"):
        assert callable(getattr(dq_func, name))
    assert callable(helper_functions.save_pc_npz) and callable(helper_functions.load_pc_npz)
    q = model_utils.QRegMLP(True, hidden_dim=512)
    assert sum(p.numel() for p in q.parameters()) == 425991
    from oracle import models
    assert list(q.state_dict()) and set(q.state_dict()) == set(models.QRegMLP(True, 512).state_dict())
    assert set(model_utils.DQRegMLP(512).state_dict()) == set(models.DQRegMLP(512).state_dict())


def test_models_forward_match_reference_golden(golden):
    import torch
    from autourdf_amd import model_utils
    g = golden("models_reference.npz")
    q = model_utils.QRegMLP(True, hidden_dim=32)
    q.load_state_dict({k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("q.")})
    t, r = q(torch.from_numpy(g["q_in"]))
    np.testing.assert_allclose(t.detach().numpy(), g["q_out_t"], atol=1e-7)
    np.testing.assert_allclose(r.detach().numpy(), g["q_out_r"], atol=1e-7)
    d = model_utils.DQRegMLP(hidden_dim=32)
    d.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("dq.")})
    np.testing.assert_allclose(d(torch.from_numpy(g["dq_in"])).detach().numpy(), g["dq_out"], atol=1e-7)
    from oracle import models as omodels
    for mod in (model_utils, omodels):                      # product modules and oracle restatements alike
        r = mod.RRegMLP(hidden_dim=32)
        r.load_state_dict({k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("r6d.")})
        t, rr = r(torch.from_numpy(g["r6d_in"]))
        np.testing.assert_allclose(t.detach().numpy(), g["r6d_out_t"], atol=1e-7)
        np.testing.assert_allclose(rr.detach().numpy(), g["r6d_out_r"], atol=1e-7)
        e = mod.RegMLP(6, 3)
        e.load_state_dict({k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("rpy.")})
        t, rr = e(torch.from_numpy(g["rpy_in"]))
        np.testing.assert_allclose(t.detach().numpy(), g["rpy_out_t"], atol=1e-7)
        np.testing.assert_allclose(rr.detach().numpy(), g["rpy_out_r"], atol=1e-7)


def test_train_plan_parameter_layout_matches_every_reference_model():
    """The four --r choices (mlp_reg.py:276-291) on the fused plan: the tensor shapes ops.TrainPlan expects (in Q_PARAM_ORDER /
    DQ_PARAM_ORDER) are the models' own, at the reference's widths -- 512, and 3 for RegMLP(6, 3) -- and at the padded width."""
    import types
    from autourdf_amd import mlp_reg, model_utils, ops
    cases = [("q", model_utils.QRegMLP(True, 512)), ("dq", model_utils.DQRegMLP(512)), ("6d", model_utils.RRegMLP(512)),
             ("rpy", model_utils.RegMLP(6, 3)), ("6d", model_utils.RRegMLP(100)), ("rpy", model_utils.RegMLP(True, 64))]
    for rot, model in cases:
        got_rot, params, hidden = mlp_reg._model_params(model)
        assert got_rot == rot and hidden == model.encoder[0].out_features
        order = ops.DQ_PARAM_ORDER if rot == "dq" else ops.Q_PARAM_ORDER
        shapes = ops.TrainPlan._param_shapes(types.SimpleNamespace(rot=ops.TRAIN_ROT[rot]), hidden)
        assert len(params) == len(order) == len(shapes)
        for name, p, want in zip(order, params, shapes):
            assert tuple(p.shape) == tuple(want), (rot, name, tuple(p.shape), want)
        padded = ops.TrainPlan._param_shapes(types.SimpleNamespace(rot=ops.TRAIN_ROT[rot]), next(t for t in ops.TRAIN_HIDDEN_TILES if t >= hidden))
        assert all(all(a <= b for a, b in zip(sm, sp)) for sm, sp in zip(shapes, padded))
    with pytest.raises(NotImplementedError):
        import torch
        mlp_reg._model_params(torch.nn.Linear(3, 3))


def test_npz_roundtrip_keeps_order_and_dtype(tmp_path):
    from autourdf_amd.helper_functions import load_pc_npz, save_pc_npz
    rng = np.random.default_rng(0)
    segs = [rng.normal(size=(n, 3)) for n in (5, 0, 12, 3, 7, 1, 9, 2, 4, 6, 8, 10)]     # > 10: '10' sorts after '1'
    save_pc_npz(segs, str(tmp_path / "c.npz"))
    back = load_pc_npz(str(tmp_path / "c.npz"))
    assert len(back) == len(segs)
    for a, b in zip(segs, back):
        np.testing.assert_array_equal(a, b)


def test_ply_reader_ascii_and_binary(tmp_path):
    from autourdf_amd.cluster_icp import read_point_cloud
    pts = np.random.default_rng(1).normal(size=(17, 3))
    p1 = tmp_path / "a.ply"
    with open(p1, "w") as f:
        f.write("ply\nformat ascii 1.0\ncomment x\nelement vertex 17\nproperty double x\nproperty double y\n"
                "property double z\nend_header\n")
        for r in pts:
            f.write("%.17g %.17g %.17g\n" % tuple(r))
    np.testing.assert_array_equal(read_point_cloud(str(p1)).points, pts)
    p2 = tmp_path / "b.ply"
    rec = np.zeros(17, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    rec["x"], rec["y"], rec["z"] = pts[:, 0], pts[:, 1], pts[:, 2]
    with open(p2, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\nelement vertex 17\nproperty float x\nproperty float y\n"
                b"property float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n")
        f.write(rec.tobytes())
    np.testing.assert_array_equal(read_point_cloud(str(p2)).points, pts.astype(np.float32).astype(np.float64))


def test_synthetic_sequences_are_seeded_and_sized():
    from autourdf_amd.synthetic import initial_segmentation, make_sequence
    a = make_sequence("wx200_5", 2, 3, 4096)
    b = make_sequence("wx200_5", 2, 3, 4096)
    assert all(x.shape == (4096, 3) and x.dtype == np.float64 for x in a)
    assert all((x == y).all() for x, y in zip(a, b))
    assert not (a[0] == make_sequence("wx200_5", 3, 1, 4096)[0]).all()
    mats, clusters, lab = initial_segmentation(a[0], 20, seed=0)
    assert mats.shape == (20, 4, 4) and sum(len(c) for c in clusters) == 4096 and lab.dtype == np.int32
    for r in ("franka", "allegro", "chain32"):
        assert make_sequence(r, 0, 1, 512)[0].shape == (512, 3)


def test_kmeans_plusplus_follows_sklearns_draw_sequence(golden):
    """Segments.k_means_cluster seeds like scikit-learn does (global RandomState, k-means++ on centred data): with the
    same np.random.seed, live sklearn's k_means(init="k-means++") and our seeding + Lloyd give identical labels, and
    the reference-minted segmentation golden (reference Segments.k_means_cluster under np.random.seed) is reproduced."""
    import numpy as np
    from sklearn.cluster import k_means
    from autourdf_amd.cluster_icp import kmeans_plusplus_sklearn
    from oracle import kmeans as okm
    g = golden("segments_reference.npz")
    X = g["frame"]
    for seed in (int(g["seed"]), 0, 5):
        np.random.seed(seed)
        _, lab, _ = k_means(X.copy(), init="k-means++", n_clusters=8)
        np.random.seed(seed)
        idx = kmeans_plusplus_sklearn(X, 8, np.random.mtrand._rand)
        _, olab, _, _ = okm.k_means(X, X[idx])
        np.testing.assert_array_equal(lab, olab)
    np.random.seed(int(g["seed"]))
    idx = kmeans_plusplus_sklearn(X, 8, np.random.mtrand._rand)
    _, olab, _, _ = okm.k_means(X, X[idx])
    np.testing.assert_array_equal(np.cumsum([0] + list(np.bincount(olab, minlength=8))), g["offsets"])
    for i in range(8):
        np.testing.assert_allclose(X[olab == i].mean(0), g["matrices"][i][:3, 3], atol=1e-14)


def test_dq_autograd_formulas_match_the_oracle_values_and_gradients():
    """autourdf_amd._dq_autograd (the torch restatements the drop-in differentiates in its backward) against oracle.dq, which
    is pinned to the reference module's goldens: values and input gradients, fp64 on the CPU."""
    import torch
    from autourdf_amd import _dq_autograd as ag
    from oracle import dq as odq
    from scipy.spatial.transform import Rotation
    g = torch.Generator().manual_seed(4)
    M = torch.eye(4, dtype=torch.float64).repeat(24, 1, 1)
    M[:, :3, :3] = torch.from_numpy(Rotation.random(24, random_state=2).as_matrix())
    M[:, :3, 3] = torch.randn(24, 3, generator=g, dtype=torch.float64)
    d1 = odq.transform_to_dualquat(M) + 0.05 * torch.randn(24, 8, generator=g, dtype=torch.float64)
    d2 = odq.transform_to_dualquat(M.flip(0)) + 0.05 * torch.randn(24, 8, generator=g, dtype=torch.float64)
    cases = [(ag.transform_to_dualquat, odq.transform_to_dualquat, (M,)),
             (ag.quat_trans_to_dualquat, odq.quat_trans_to_dualquat, (d1[:, :4].clone(), M[:, :3, 3].clone())),
             (ag.dualquat_to_quat_trans, odq.dualquat_to_quat_trans, (d1,)),
             (ag.dualquat_multiply, odq.dualquat_multiply, (d1, d2)),
             (ag.dualquat_invert, odq.dualquat_invert, (d1,))]
    for mine, ref, ins in cases:
        a = [x.clone().requires_grad_(True) for x in ins]
        b = [x.clone().requires_grad_(True) for x in ins]
        oa, ob = mine(*a), ref(*b)
        oa, ob = (oa if isinstance(oa, tuple) else (oa,)), (ob if isinstance(ob, tuple) else (ob,))
        for x, y in zip(oa, ob):
            assert torch.allclose(x, y, atol=1e-13)
        w = [torch.randn(x.shape, generator=g, dtype=torch.float64) for x in oa]
        sum((x * ww).sum() for x, ww in zip(oa, w)).backward()
        sum((y * ww).sum() for y, ww in zip(ob, w)).backward()
        for x, y in zip(a, b):
            assert torch.allclose(x.grad, y.grad, atol=1e-12)


def test_file_writer_is_bounded_and_does_not_mask_the_frame_loops_error():
    """mlp_reg._FileWriter: at most MAX_PENDING frames wait behind the worker (a job pins a frame's device tensors), a
    submit blocks instead of queueing more, the worker's error surfaces in close() -- unless another exception is already propagating,
    which it must not replace."""
    import sys
    import threading
    import time
    from autourdf_amd import mlp_reg
    w = mlp_reg._FileWriter()
    gate, done = threading.Event(), []
    w.submit(lambda: gate.wait(10))                       # the worker is busy with this one
    for i in range(mlp_reg._FileWriter.MAX_PENDING):
        w.submit(done.append, i)                         # fills the queue
    t = threading.Thread(target=lambda: w.submit(done.append, "late"))
    t.start()
    time.sleep(0.3)
    assert t.is_alive()                                  # the submit beyond the bound waits
    gate.set()
    t.join(10)
    w.close()
    assert done == list(range(mlp_reg._FileWriter.MAX_PENDING)) + ["late"]

    def boom():
        raise ValueError("disk full")

    w = mlp_reg._FileWriter()
    w.submit(boom)
    time.sleep(0.2)
    with pytest.raises(KeyError):
        try:
            raise KeyError("the frame loop's own error")
        finally:
            w.close(propagating=sys.exc_info()[0] is not None)
    w = mlp_reg._FileWriter()
    w.submit(boom)
    time.sleep(0.2)
    with pytest.raises(ValueError):
        w.close()


def test_register_sequence_surfaces_a_writer_failure_inside_a_callers_except_block(monkeypatch, tmp_path):
    """`finally: writer.close(propagating=sys.exc_info()[0] is not None)` was also true when register_sequence was merely
    CALLED from inside somebody's `except` block -- a genuine file-writer failure was then swallowed and the frames were lost without a
    message.  The frame loop now tracks its own failure: the writer's error comes out, the loop's own error is still never replaced."""
    import time
    import torch
    from autourdf_amd import mlp_reg

    def boom():
        raise ValueError("disk full")

    def frames_ok(seg, K, m_t, cl_t, cl_init, icp_src, model, model_rf, mlp_icp, save_dir, writer, poses, best_losses):
        writer.submit(boom)
        time.sleep(0.2)

    def frames_fail(*a):
        a[10].submit(boom)
        time.sleep(0.2)
        raise KeyError("the frame loop's own error")

    monkeypatch.setattr(mlp_reg, "DEVICE", torch.device("cpu"))
    monkeypatch.setattr(mlp_reg, "_register_frames", frames_ok)
    args = (None, np.eye(4)[None].repeat(2, 0), [np.zeros((3, 3)), np.zeros((2, 3))])
    with pytest.raises(ValueError, match="disk full"):
        try:
            raise RuntimeError("the caller is handling something else")
        except RuntimeError:
            mlp_reg.register_sequence(*args, save_dir=str(tmp_path) + "/", models=(None, None))
    monkeypatch.setattr(mlp_reg, "_register_frames", frames_fail)
    with pytest.raises(KeyError):
        mlp_reg.register_sequence(*args, save_dir=str(tmp_path) + "/", models=(None, None))
