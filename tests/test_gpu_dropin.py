"""GPU: the drop-in module surface end to end (scripts/registration.sh path) and the larger configs."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _write_ply(path, pts):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty double x\nproperty double y\n"
                b"property double z\nend_header\n" % len(pts))
        f.write(np.ascontiguousarray(pts, "<f8").tobytes())


@pytest.fixture()
def workdir(tmp_path, monkeypatch):
    from autourdf_amd.synthetic import make_sequence
    for v in range(2):
        for t, fr in enumerate(make_sequence("wx200_5", v, 3, 1024)):
            _write_ply(str(tmp_path / f"data/raw/wx200_5/4_deg_20_cams/V{v:04}/{t:04}/robot.ply"), fr)
    json.dump({"wx200_5": {"num_seg": 8, "dof": 5}}, open(tmp_path / "parameters.json", "w"))
    monkeypatch.chdir(tmp_path)
    return tmp_path


@pytest.mark.parametrize("flags", [["--loss"], ["--mlp_icp"], ["--r", "dq"], ["--r", "6d"], ["--r", "rpy"], ["--normal"], ["--normal", "--mlp_icp"]])
def test_match_writes_the_reference_file_layout(workdir, flags, monkeypatch):
    from autourdf_amd import mlp_reg
    monkeypatch.setattr(mlp_reg, "EPOCHS", 12)
    mlp_reg._PLANS.clear()
    mlp_reg.main(["--robot", "wx200_5", "--num_video", "2"] + flags)
    base = workdir / "data/part/wx200_5_8_seg/4_deg_20_cams"
    for v in range(2):
        d = base / f"V{v:04}"
        for t in range(3):
            m = np.load(d / "matrix" / f"{t:04}.npy")
            assert m.shape == (8, 4, 4) and np.isfinite(m).all()
            np.testing.assert_allclose(m[:, 3], np.tile([0, 0, 0, 1.0], (8, 1)), atol=1e-6)
            R = m[:, :3, :3]
            np.testing.assert_allclose(R @ R.transpose(0, 2, 1), np.tile(np.eye(3), (8, 1, 1)), atol=5e-3 if "dq" in flags else 2e-5)
            with np.load(d / "cluster" / f"{t:04}.npz") as z:
                keys = list(z.keys())
                assert keys == [str(i) for i in range(8)]
                assert sum(len(z[k]) for k in keys) == 1024 and z["0"].dtype == np.float64
    # frame-0 state of the second sequence is the first one's (mlp_reg.py:242-253)
    np.testing.assert_array_equal(np.load(base / "V0000/matrix/0000.npy"), np.load(base / "V0001/matrix/0000.npy"))
    if "--loss" in flags:
        assert np.loadtxt(base / "V0000/loss.txt").shape == (2,)
    # local clusters really are inv(M) . world points of that frame
    from autourdf_amd.cluster_icp import read_point_cloud
    frame = read_point_cloud(str(workdir / "data/raw/wx200_5/4_deg_20_cams/V0000/0002/robot.ply")).points
    m = np.load(base / "V0000/matrix/0002.npy").astype(np.float64)
    with np.load(base / "V0000/cluster/0002.npz") as z:
        world = np.concatenate([z[str(i)] @ m[i, :3, :3].T + m[i, :3, 3] for i in range(8)])
    assert np.abs(np.sort(world, 0) - np.sort(frame, 0)).max() < 1e-6


def test_torchrun_world_size_one_of_the_dropin_module(workdir):
    """`python -m torch.distributed.run --nproc-per-node 1 -m autourdf_amd.mlp_reg ...`: the one-process-per-GPU launch of
    the drop-in (RCCL process group, frame-0 state by rank 0, barrier, this rank's shard) writes the reference's files."""
    import subprocess
    import sys
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), "-m", "autourdf_amd.mlp_reg", "--robot", "wx200_5", "--num_video", "2", "--loss"],
                       cwd=str(workdir), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    base = workdir / "data/part/wx200_5_8_seg/4_deg_20_cams"
    for v in range(2):
        for t in range(3):
            m = np.load(base / f"V{v:04}/matrix/{t:04}.npy")
            assert m.shape == (8, 4, 4) and np.isfinite(m).all()
            with np.load(base / f"V{v:04}/cluster/{t:04}.npz") as z:
                assert list(z.keys()) == [str(i) for i in range(8)] and sum(len(z[k]) for k in z.keys()) == 1024
        assert np.loadtxt(base / f"V{v:04}/loss.txt").shape == (2,)
    np.testing.assert_array_equal(np.load(base / "V0000/matrix/0000.npy"), np.load(base / "V0001/matrix/0000.npy"))


@pytest.mark.parametrize("rot,extra,nv", [("q", [], 2), ("dq", [], 2), ("q", ["--mlp_icp"], 2), ("6d", [], 2), ("q", [], 1), ("rpy", [], 1),
                                          ("q", ["--normal"], 2)])
def test_lock_step_run_equals_one_match_per_sequence(workdir, rot, extra, nv, monkeypatch):
    """main() registers all sequences in lock-step (match_all); --sequential is the reference's loop of match()
    calls.  Same frame-0 state + same model initialisation => identical files, bit for bit -- also for a single sequence
    (--num_video 1 takes the device-resident engine too: no host round trip per train) and for the optional pose representations."""
    import shutil
    from autourdf_amd import mlp_reg
    monkeypatch.setattr(mlp_reg, "EPOCHS", 12)
    base = workdir / "data/part/wx200_5_8_seg/4_deg_20_cams"
    mlp_reg._PLANS.clear()
    torch.manual_seed(0)
    mlp_reg.main(["--robot", "wx200_5", "--num_video", str(nv), "--loss", "--sequential", "--r", rot] + extra)
    shutil.copytree(base, workdir / "sequential")
    for v in range(nv):                                  # keep only the shared frame-0 state, as a finished first run leaves it
        for t in (1, 2):
            os.remove(base / f"V{v:04}/matrix/{t:04}.npy"); os.remove(base / f"V{v:04}/cluster/{t:04}.npz")
    torch.manual_seed(0)
    mlp_reg.main(["--robot", "wx200_5", "--num_video", str(nv), "--loss", "--r", rot] + extra)
    for v in range(nv):
        for t in range(3):
            np.testing.assert_array_equal(np.load(base / f"V{v:04}/matrix/{t:04}.npy"),
                                          np.load(workdir / f"sequential/V{v:04}/matrix/{t:04}.npy"))
            with np.load(base / f"V{v:04}/cluster/{t:04}.npz") as a, np.load(workdir / f"sequential/V{v:04}/cluster/{t:04}.npz") as b:
                assert list(a.keys()) == list(b.keys())
                for k in a.keys():
                    np.testing.assert_array_equal(a[k], b[k])
        np.testing.assert_array_equal(np.loadtxt(base / f"V{v:04}/loss.txt"), np.loadtxt(workdir / f"sequential/V{v:04}/loss.txt"))


def test_train_signature_drop_in_returns(workdir):
    """train(m, y, model, clusters, ...) -> (list of np (M_k,3) f32, list of .points objects, (K,4,4) tensor, float)."""
    from autourdf_amd import mlp_reg
    from autourdf_amd.model_utils import QRegMLP
    from autourdf_amd.synthetic import initial_segmentation, make_sequence
    dev = torch.device("cuda")
    seq = make_sequence("wx200_5", 0, 2, 1024)
    mats, clusters, _ = initial_segmentation(seq[0], 8, seed=0)
    mlp_reg.ROT, mlp_reg.DEVICE = "q", dev
    mlp_reg._PLANS.clear()
    old, mlp_reg.EPOCHS = mlp_reg.EPOCHS, 10
    try:
        model = QRegMLP(True, hidden_dim=512).to(dev)
        before = model.encoder[0].weight.detach().clone()
        out = mlp_reg.train(torch.tensor(mats, dtype=torch.float32, device=dev), torch.tensor(seq[1], dtype=torch.float32, device=dev),
                            model, [torch.tensor(c, dtype=torch.float32, device=dev) for c in clusters])
    finally:
        mlp_reg.EPOCHS = old
    pred_np, pred_pcd, best_m, min_loss = out
    assert len(pred_np) == 8 and all(p.dtype == np.float32 and p.shape[1] == 3 for p in pred_np)
    assert all(hasattr(p, "points") for p in pred_pcd) and best_m.shape == (8, 4, 4) and isinstance(min_loss, float)
    assert not torch.equal(before, model.encoder[0].weight)          # Adam updated the caller's module in place
    pcs = mlp_reg.calculate_pc([torch.tensor(c, dtype=torch.float32, device=dev) for c in clusters], best_m)
    np.testing.assert_allclose(torch.cat(pcs).cpu().numpy(), np.concatenate(pred_np), atol=1e-6)


@pytest.mark.parametrize("robot,n,k", [("wx200_5", 4096, 20), ("franka", 16384, 40), ("allegro", 4096, 30), ("chain32", 32768, 128)])
def test_larger_configs_two_epochs_vs_oracle(robot, n, k):
    """BASELINE configs[0] / [1] (the headline shape: wx200_5, N=4096, K=20) and configs 3-5 shapes (hidden 512), K up to 128,
    multi-chunk NN.  With identical weights the
    poses agree to 1e-5 (checked on the forward of epoch 0 and on the loss of both epochs).  After an Adam
    step they can differ more: Adam's first update is lr * g / (|g| + eps), so a weight whose gradient is
    rounding noise (|g| ~ 1e-10; a few dozen of 426k here) moves by +-lr depending on the summation
    order -- true of any second implementation, the reference on another BLAS included -- hence 2e-4
    on the pose after the update."""
    from autourdf_amd import ops
    from autourdf_amd.synthetic import initial_segmentation, make_sequence
    from oracle import models, registration
    dev = torch.device("cuda")
    seq = make_sequence(robot, 1, 2, n)
    mats, clusters, _ = initial_segmentation(seq[0], k, seed=1, iters=5)
    torch.manual_seed(1)
    model = models.QRegMLP(True, 512)
    m, y = torch.tensor(mats, dtype=torch.float32), torch.tensor(seq[1], dtype=torch.float32)
    cl = [torch.tensor(c, dtype=torch.float32) for c in clusters]
    params = [model.state_dict()[key].clone().to(dev) for key in ops.Q_PARAM_ORDER]
    pts, off = ops.pack_clusters(cl, dev)
    plan = ops.TrainPlan("q", k, 512, n, n, epochs=2, use_graph=True, device=dev)
    best_m, _, res, lh, _ = plan.run(m.to(dev), y.to(dev), pts, off, params)
    _, o_best, o_min, hist = registration.train(m, y, model, cl, rot="q", epochs=2)
    # epoch 0 (identical weights): 2e-6; epoch 1 follows the first Adam step, whose +-lr kicks on noise-gradient weights (above)
    # show in the loss at the 2e-5 level (measured 2.0e-5 on the headline shape, 0.157265 vs 0.157261)
    lh_h, o_l = lh.cpu().numpy(), np.array(hist["loss"], np.float32)
    assert abs(lh_h[0] - o_l[0]) <= 2e-6 * o_l[0]
    np.testing.assert_allclose(lh_h, o_l, rtol=5e-5)
    np.testing.assert_allclose(best_m.cpu().numpy(), o_best.detach().numpy(), atol=2e-4)
    torch.manual_seed(1)
    model0 = models.QRegMLP(True, 512)                                     # pristine weights: forward parity at 1e-5
    params0 = [model0.state_dict()[key].clone().to(dev) for key in ops.Q_PARAM_ORDER]
    m2, pred, loss0, _ = plan.probe(m.to(dev), y.to(dev), pts, off, params0)
    o_m2 = registration.pose_forward(m, model0, "q")
    np.testing.assert_allclose(m2.cpu().numpy(), o_m2.detach().numpy(), atol=1e-5)
    np.testing.assert_allclose(pred.cpu().numpy(), torch.cat(registration.calculate_pc(cl, o_m2)).detach().numpy(), atol=1e-5)
    assert abs(loss0.item() - hist["loss"][0]) <= 2e-5 * hist["loss"][0]


@pytest.mark.parametrize("rot", ["6d", "rpy"])
def test_optional_representations_short_trajectory_vs_oracle(rot):
    """--r 6d / --r rpy through the drop-in train(): the fused plan (RegMLP(6, 3) zero-padded to the kernels' width) against the
    all-CPU oracle, 5 epochs; the model's tensors come back trained, in their own shapes."""
    from autourdf_amd import mlp_reg, model_utils
    from autourdf_amd.synthetic import initial_segmentation, make_sequence
    from oracle import models, registration
    dev = torch.device("cuda")
    seq = make_sequence("wx200_5", 2, 2, 1024)
    mats, clusters, _ = initial_segmentation(seq[0], 8, seed=2)
    torch.manual_seed(9)
    o_model = models.RRegMLP(64) if rot == "6d" else models.RegMLP(6, 3)
    g_model = (model_utils.RRegMLP(64) if rot == "6d" else model_utils.RegMLP(6, 3))
    g_model.load_state_dict(o_model.state_dict())
    g_model = g_model.to(dev)
    m, y = torch.tensor(mats, dtype=torch.float32), torch.tensor(seq[1], dtype=torch.float32)
    cl = [torch.tensor(c, dtype=torch.float32) for c in clusters]
    _, o_best, o_min, hist = registration.train(m, y, o_model, cl, rot=rot, epochs=5)
    old = (mlp_reg.ROT, mlp_reg.EPOCHS)
    mlp_reg.ROT, mlp_reg.EPOCHS = rot, 5
    try:
        _, _, best_m, min_loss = mlp_reg.train(m.to(dev), y.to(dev), g_model, [c.to(dev) for c in cl])
    finally:
        mlp_reg.ROT, mlp_reg.EPOCHS = old
    assert abs(min_loss - o_min) <= 2e-5 * abs(o_min)
    np.testing.assert_allclose(best_m.detach().cpu().numpy(), o_best.detach().numpy(), atol=2e-5)
    for (name, p), q in zip(g_model.state_dict().items(), o_model.state_dict().values()):
        assert p.shape == q.shape
        np.testing.assert_allclose(p.cpu().numpy(), q.numpy(), atol=5e-5, err_msg=name)


@pytest.mark.parametrize("mode", ["sequences", "replay"])
def test_bench_under_torchrun_world_one_runs_the_rccl_gather(mode):
    """`python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1 ...` (the driver's launcher shape at N > 1, here with one
    rank): the process group is RCCL, the final `gather_poses` is a real all_gather_into_tensor on the GPU, and the line reports it."""
    import socket
    import subprocess
    import sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "5", "--mode", mode,
                        "--no-cpu-baseline", "--no-icp-variant", "--no-roofline", "--no-other-workloads", "--repeats", "2"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["value"] > 0
    g = d["rccl_gather"]
    assert g["backend"] == "nccl" and g["world"] == 1 and g["own_block_returned_intact"] is True and g["us"] > 0
    assert g["payload_bytes"] == 5 * 20 * 64
    if mode == "sequences":
        assert d["repeats"]["n"] == 2 and d["repeats"]["poses_identical_across_repeats"] is True


def test_bench_plain_single_process_creates_its_own_world_one_rccl_group():
    """`python bench.py --gpus 1` without a launcher (the driver's N = 1 command): bench.py creates the world-1 RCCL group itself so the
    job's one collective runs on hardware in every round's BENCH line (`rccl_world1_gather_us`)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "5",
                        "--no-cpu-baseline", "--no-icp-variant", "--no-roofline", "--no-other-workloads", "--repeats", "1"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["rccl_world1_gather_us"] > 0 and d["rccl_gather"]["backend"] == "nccl", d.get("rccl_gather")
    assert "created by bench.py" in d["rccl_gather"]["group"]


_TWO_PROC_CHILD = r"""
import json, os, sys, time
sys.path.insert(0, sys.argv[1])
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import numpy as np, torch
from autourdf_amd.engine import BatchRegistrar
from autourdf_amd.synthetic import initial_segmentation, make_sequence
dev = torch.device("cuda:0")
seq = [make_sequence("wx200_5", s, 4, 1024) for s in range(5)]
mats0, clusters0, _ = initial_segmentation(seq[0][0], 8, seed=0)
reg = BatchRegistrar(mats0, clusters0, 1024, 5, "q", 64, 40, True, dev)
torch.cuda.synchronize()
open(sys.argv[2] + ".ready", "w").close()
t0 = time.time()
while not os.path.exists(sys.argv[3] + ".ready") and time.time() - t0 < 120:      # both processes hold a context before either picks its chain streams
    time.sleep(0.01)
out = None
for f in range(1, 4):
    out = reg.step([torch.as_tensor(s[f], dtype=torch.float64, device=dev) for s in seq])
torch.cuda.synchronize()
print(json.dumps({"probe_us": reg.plan.chain_probe_us(), "chains": reg.plan.info["graph_branches"],
                  "checksum": float(sum(o[0].double().abs().sum() for o in out))}))
"""


def test_chain_stream_probe_with_two_processes_sharing_the_device(tmp_path):
    """the closest single-GPU stand-in for the runtime state of an 8-rank node -- two PROCESSES drive plans of five
    sequences (two chain streams each) on the one device at the same time.  The chain-stream pick (a 150 us spin kernel per candidate
    stream, accepted only if it overlaps the caller's: train_engine.hip pick_chain_streams) must still end with a verdict in each
    process -- concurrent streams found (0 < probe < 250 us) or, when the other process's queues make every candidate look shared, the
    documented fallback (-1) -- and whatever it picks, the poses are those of a process that has the device to itself."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    procs = [subprocess.Popen([sys.executable, "-c", _TWO_PROC_CHILD, root, me, other], cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for me, other in ((a, b), (b, a))]
    outs = [p.communicate(timeout=600) for p in procs]
    res = []
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so[-1000:] + se[-2000:]
        res.append(json.loads([l for l in so.splitlines() if l.startswith("{")][0]))
    open(a + ".ready", "w").close(); open(b + ".ready", "w").close()
    alone = subprocess.run([sys.executable, "-c", _TWO_PROC_CHILD, root, str(tmp_path / "c"), a], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert alone.returncode == 0, alone.stderr[-2000:]
    solo = json.loads([l for l in alone.stdout.splitlines() if l.startswith("{")][0])
    assert solo["chains"] == 2 and 0 < solo["probe_us"] < 250, solo            # a process alone on the device always finds a queue of its own
    for r in res:
        assert r["chains"] == 2
        assert r["probe_us"] == -1 or 0 < r["probe_us"] < 250, r
        assert r["checksum"] == solo["checksum"], (r, solo)                     # results never depend on the streams picked
    print("chain probe with two processes on the device:", [r["probe_us"] for r in res], "alone:", solo["probe_us"])


@pytest.mark.parametrize("robot", ["wx200_5", "franka"])
def test_real_robot_geometry_frames(robot, tmp_path, monkeypatch, golden):
    """VERDICT r5 item 6 / SURVEY N4: frames of the REAL robots -- the reference's URDF + meshes (Robot/interbotix_descriptions wx200,
    Robot/franka) posed along the reference's joint trajectories, seen through its camera ring, noise and farthest-point down-sampling as
    Sim/sim_data.py does it (tests/golden/make_golden_real_frames.py, minted in the build container: 2 sequences x 10 frames x 4096 points,
    num_seg 20) -- instead of the capsule chains every other test runs on:
      * k-means labels of a frame (Segments' seeding path: k-means++ + Lloyd on the GPU, then the frame-to-frame resample) bit-exact
        against the oracle; the nearest-neighbour search bit-exact;
      * train(): the forward of the entering parameters and teacher-forced single epochs (e = 0, 3, 7) within 1e-6 (loss) / 1e-5 (pose),
        lr and counters exact -- a free-running 8-epoch bound does not exist on these frames (see below);
      * match() on the PLY files writes the reference's file layout, and its clusters are inv(M) . the frame's points."""
    import os as _os
    if not _os.path.exists(_os.path.join(_os.path.dirname(__file__), "golden", f"frames_{robot}_real.npz")):
        pytest.skip(f"frames_{robot}_real.npz not minted")
    from autourdf_amd import mlp_reg, ops
    from autourdf_amd.synthetic import initial_segmentation
    from oracle import chamfer, kmeans as okm, models as omodels, registration as oreg
    dev = torch.device("cuda:0")
    g = golden(f"frames_{robot}_real.npz")
    frames = g["frames"].astype(np.float64)
    K = int(g["num_seg"])
    assert frames.shape == (2, 10, 4096, 3) and K == 20
    mats, clusters, _ = initial_segmentation(frames[0, 0], K, seed=0)
    # --- K2: the frame-to-frame re-segmentation seeded at the poses' translations, labels bit-exact; its iteration count
    f1 = torch.as_tensor(frames[0, 1], device=dev)
    seeds = torch.as_tensor(mats[:, :3, 3].copy(), device=dev)
    _, lab, _, n_it = ops.kmeans_lloyd(f1, seeds)
    _, olab, _, on_it = okm.k_means(frames[0, 1], mats[:, :3, 3])
    assert (lab.cpu().numpy() == olab).all() and int(n_it) == int(on_it)
    # --- K1: nearest neighbours between the predicted cloud (frame 0) and frame 1, indices and distances bit-exact
    x, y = frames[0, 0].astype(np.float32), frames[0, 1].astype(np.float32)
    dx, ix, dy, iy = ops.nn_l1_bidir(torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev))
    odx, oix = chamfer.nn_l1(x, y)
    ody, oiy = chamfer.nn_l1(y, x)
    assert (ix.cpu().numpy() == oix).all() and (dx.cpu().numpy() == odx).all() and (iy.cpu().numpy() == oiy).all() and (dy.cpu().numpy() == ody).all()
    # --- A1: train() at the reference's model (hidden 512) from the frame-0 state.  On these frames -- as on the allegro shape -- NO second
    #     float32 evaluation follows the reference for even one free-running step: the oracle with its GEMMs summed in another order is 1.1e-3
    #     away in pose after ONE epoch, 1.2e-2 after four (measured in the build container, DESIGN section 2: the first Adam step turns every
    #     gradient into +-lr, and sums of +-1/N signs sit next to zero), so the 8-epoch free-running bound of the capsule chains does not
    #     exist here.  What does: the forward of the entering parameters, and single epochs teacher-forced from the oracle's state.
    import _teacher as T
    torch.manual_seed(0)
    sd = {k: v.detach().clone() for k, v in omodels.QRegMLP(True, 512).state_dict().items()}
    off_np = np.cumsum([0] + [len(c) for c in clusters])
    case = {"m": mats.astype(np.float32), "y": y, "local": np.concatenate(clusters).astype(np.float32), "offsets": off_np}
    snaps, hist = T.oracle_snapshots(case, sd, K, (0, 3, 7))
    m_d, y_d = torch.from_numpy(case["m"]).to(dev), torch.from_numpy(y).to(dev)
    pts, off = ops.pack_clusters([torch.from_numpy(c) for c in T.split(case["local"], off_np)], dev)
    plan = ops.TrainPlan("q", K, 512, pts.shape[0], 4096, epochs=300, use_graph=False, device=dev)
    probe_plan = ops.TrainPlan("q", K, 512, pts.shape[0], 4096, epochs=2, use_graph=False, device=dev)
    for e in (0, 3, 7):
        r = T.compare(ops.Q_PARAM_ORDER, snaps[e], snaps[e + 1], *T.plan_epoch(plan, probe_plan, dev, ops.Q_PARAM_ORDER, m_d, y_d, pts, off, snaps[e]))
        assert r["loss_rel"] <= 1e-6 and r["pose"] <= 1e-5, (e, r)
        assert r["lr_used_exact"] and r["lr"][0] == r["lr"][1] and all(a == b for a, b in r["exact"].values()), (e, r)
        if e > 0:
            assert r["upd"] <= 3e-6, (e, r["upd"], r["upd_worst_tensor"])
    # --- F1: match() on the PLY files of two sequences x three frames
    for v in range(2):
        for t in range(3):
            _write_ply(str(tmp_path / f"data/raw/{robot}/4_deg_20_cams/V{v:04}/{t:04}/robot.ply"), frames[v, t])
    json.dump({robot: {"num_seg": K, "dof": 5}}, open(tmp_path / "parameters.json", "w"))
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(mlp_reg, "EPOCHS", 12)
    mlp_reg._PLANS.clear()
    mlp_reg.main(["--robot", robot, "--num_video", "2", "--loss"])
    base = tmp_path / f"data/part/{robot}_{K}_seg/4_deg_20_cams"
    for v in range(2):
        for t in range(3):
            mm = np.load(base / f"V{v:04}" / "matrix" / f"{t:04}.npy")
            assert mm.shape == (K, 4, 4) and np.isfinite(mm).all()
            with np.load(base / f"V{v:04}" / "cluster" / f"{t:04}.npz") as z:
                assert list(z.keys()) == [str(i) for i in range(K)] and sum(len(z[k]) for k in z.keys()) == 4096
    mm = np.load(base / "V0001/matrix/0002.npy").astype(np.float64)
    with np.load(base / "V0001/cluster/0002.npz") as z:
        world = np.concatenate([z[str(i)] @ mm[i, :3, :3].T + mm[i, :3, 3] for i in range(K)])
    assert np.abs(np.sort(world, 0) - np.sort(frames[1, 2], 0)).max() < 1e-6
    assert np.loadtxt(base / "V0000/loss.txt").shape == (2,)
