"""Mint tests/golden/frames_{wx200_5,franka}_real.npz (run in the BUILD CONTAINER only: it reads /root/reference/Robot).

    python tests/golden/make_golden_real_frames.py [wx200_5 franka]

VERDICT r5 item 6 / SURVEY N4: every GPU test and bench workload ran on the capsule chains of autourdf_amd/synthetic.py; these are
frames of the REAL robots -- the reference's URDFs and meshes (Robot/interbotix_descriptions wx200, Robot/franka), posed along the
reference's own joint trajectories (`angle_list`, restated call for call and pinned by sim_angle_list.npz), seen through the
reference's camera ring (20 cameras, numpy's global RandomState seeded per sequence like collect()), with its noise model, then
farthest-point down-sampled -- 2 sequences x 10 frames x 4096 points, float32, points only (< 1 MB).

How: autourdf_amd.sim_data's own host logic (URDF / mesh loading, forward kinematics, `data_collection`'s loop: draw until enough
visible points, noise, down-sampling) with its three GPU kernels replaced by their CPU restatements from oracle/ (sample_mesh,
visibility, farthest_point_sample: the same arithmetic in the same order, tests/test_gpu_sim.py pins kernel == restatement) -- the
build container has no GPU and the GPU box has no /root/reference.  A fixture is data: points (+ the joint angles that posed them)."""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from autourdf_amd import sim_data as S                                   # noqa: E402
from oracle import kmeans as okm, sim_data as osim                       # noqa: E402

REF = "/root/reference"
N_SEQ, N_FRAMES, N_POINTS = 2, 10, 4096


POOL = None


def _one_camera(job):
    return osim.visibility(*job)[0]


def cpu_env(env):
    """The three kernel calls of SimEnv / data_collection on the host (oracle restatements)."""
    r = env.robot

    def sample_surface(joint_positions, n, rng):
        T = r.fk(joint_positions, env.base)
        out, _ = osim.sample_mesh(r.tri, r.cum_area, r.tri_link, T, rng.random((n, 3)))
        return torch.as_tensor(out)

    def visible(joint_positions, pts, width=800, height=800, eps=0.004):
        # a point is visible when SOME camera sees it and every camera has its own depth buffer: one oracle call per camera, in a process
        # pool, OR-ed (the same result as one call over the ring; the per-triangle rasteriser is a Python loop: 126 586 triangles for franka)
        T = r.fk(joint_positions, env.base)
        c = env.cameras[0]
        jobs = [(r.tri, r.tri_link, T, env.cam_frames[i:i + 1], pts.numpy(), c["fov"], c["aspect"], c["near_val"], c["far_val"], width, height, eps)
                for i in range(len(env.cam_frames))]
        vis = np.zeros(len(pts), bool)
        for v in POOL.map(_one_camera, jobs):
            vis |= v
        return torch.as_tensor(vis)

    env.sample_surface, env.visible = sample_surface, visible
    return env


def mint(robot, width):
    params = json.load(open(os.path.join(REF, "parameters.json")))[robot]
    S.farthest_point_sample = lambda pts, m: okm.farthest_point_sample(pts.numpy(), m)
    frames, angles = [], []
    for seed in range(N_SEQ):
        t0 = time.perf_counter()
        np.random.seed(seed)                                             # (collect(): the ring of >= 20 cameras draws from the global state)
        env = cpu_env(S.SimEnv(os.path.join(REF, params["gt"]), base_orientation=params.get("sim_ori", [0, 0, 0]), dof=params["dof"],
                               radius=params.get("cam_dist", 1.5), num_cameras=20))
        a = S.angle_list(N_FRAMES, 4, params["dof"], env.joint_limits, np.array([0.9] * params["dof"]), seed)
        _, rec = S.data_collection(env, None, width=width, height=width, angle_list=a, noise_flag=True, num_points=N_POINTS, seed=seed)
        frames.append(np.stack([np.asarray(c.points) for c in rec]))
        angles.append(a)
        print(f"{robot} sequence {seed}: {N_FRAMES} frames in {time.perf_counter() - t0:.0f} s", flush=True)
    out = os.path.join(HERE, f"frames_{robot}_real.npz")
    np.savez_compressed(out, frames=np.stack(frames).astype(np.float32), angles=np.stack(angles), num_seg=np.int64(params["num_seg"]),
                        urdf=np.array(params["gt"]), depth_buffer=np.int64(width), triangles=np.int64(len(env.robot.tri)))
    print("wrote", out, f"{os.path.getsize(out) / 1024:.0f} KB")


if __name__ == "__main__":
    import multiprocessing
    POOL = multiprocessing.Pool(min(8, os.cpu_count() or 1))
    for rb in (sys.argv[1:] or ["wx200_5", "franka"]):
        # (franka: 126 586 triangles x 20 cameras through the oracle's per-triangle rasteriser -- a 400 x 400 depth buffer keeps it to minutes)
        mint(rb, 800 if rb != "franka" else 400)
    POOL.close()
    POOL.join()
