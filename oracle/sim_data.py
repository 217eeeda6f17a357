"""Oracle for the synthetic-frame generator.  TEST INFRASTRUCTURE ONLY.

* ``angle_list``: the reference's own function (Sim/sim_data.py:372-430) is imported under shims by
  tests/golden/make_golden_sim.py -> tests/golden/sim_angle_list.npz                     -- PINNED
* ``sample_mesh``: numpy restatement of creg_sample_mesh_f64 with the same operation order (bit-exact check)
* ``fk``: an independent forward-kinematics evaluation (scipy Rotation) for the URDF subset the generator
  reads.  The reference delegates both to PyBullet + OpenGL rendering (not restatable)   -- parity UNPINNED
"""
import numpy as np
from scipy.spatial.transform import Rotation


def sample_mesh(tri, cum_area, tri_link, link_T, u):
    tri, cum_area, link_T, u = (np.asarray(a, np.float64) for a in (tri, cum_area, link_T, u))
    target = u[:, 0] * cum_area[-1]
    f = np.minimum(np.searchsorted(cum_area, target, side="right"), len(cum_area) - 1)
    s = np.sqrt(u[:, 1])
    b0, b1, b2 = 1.0 - s, s * (1.0 - u[:, 2]), s * u[:, 2]
    t = tri[f]
    p = (b0[:, None] * t[:, 0] + b1[:, None] * t[:, 1]) + b2[:, None] * t[:, 2]
    T = link_T[np.asarray(tri_link)[f]]
    out = np.empty_like(p)
    for d in range(3):
        out[:, d] = ((T[:, d, 0] * p[:, 0] + T[:, d, 1] * p[:, 1]) + T[:, d, 2] * p[:, 2]) + T[:, d, 3]
    return out, np.asarray(tri_link)[f]


def fk(links, joints, q, root, base=None):
    """links: names; joints: dicts with name/type/parent/child/xyz/rpy/axis; q: {name: value}."""
    T = {root: np.eye(4) if base is None else np.asarray(base, np.float64)}
    todo = list(joints)
    while todo:
        j = next(j for j in todo if j["parent"] in T)
        todo.remove(j)
        O = np.eye(4)
        O[:3, :3] = Rotation.from_euler("xyz", j["rpy"]).as_matrix()          # extrinsic xyz = URDF fixed-axis rpy
        O[:3, 3] = j["xyz"]
        M = np.eye(4)
        a = np.asarray(j["axis"], np.float64)
        a = a / np.linalg.norm(a)
        v = float(q.get(j["name"], 0.0))
        if j["type"] in ("revolute", "continuous"):
            M[:3, :3] = Rotation.from_rotvec(a * v).as_matrix()
        elif j["type"] == "prismatic":
            M[:3, 3] = a * v
        T[j["child"]] = T[j["parent"]] @ O @ M
    return np.stack([T[l] for l in links])


def camera_ring(radius, num_cameras, rng=None, cam_angle=20):
    """Camera frames of the reference's ring (Sim/sim_data.py:88-116): fewer than 20 cameras sit evenly on a circle at
    `cam_angle` degrees elevation, 20 or more are drawn uniformly in azimuth and in elevation [0, pi/2); all look at the
    origin with +z up.  Rows: eye | forward | right | up (what creg_visibility_f64 takes)."""
    if num_cameras < 20:
        theta = np.linspace(0, 2 * np.pi, num_cameras, endpoint=False)
        phi = np.full(num_cameras, np.pi * cam_angle / 180)
    else:
        theta = rng.random(num_cameras) * 2 * np.pi
        phi = rng.random(num_cameras) * np.pi / 2
    eye = np.stack([radius * np.cos(theta) * np.cos(phi), radius * np.sin(theta) * np.cos(phi), radius * np.sin(phi)], 1)
    cams = []
    for e in eye:
        f = -e / np.linalg.norm(e)
        s = np.cross(f, [0.0, 0.0, 1.0])
        s = s / np.linalg.norm(s)
        u = np.cross(s, f)
        cams.append(np.concatenate([e, f, s, u]))
    return np.asarray(cams)


def _project(cam, p, tan_half, aspect, W, H):
    r = p - cam[0:3]
    d = (r[..., 0] * cam[3] + r[..., 1] * cam[4]) + r[..., 2] * cam[5]
    xc = (r[..., 0] * cam[6] + r[..., 1] * cam[7]) + r[..., 2] * cam[8]
    yc = (r[..., 0] * cam[9] + r[..., 1] * cam[10]) + r[..., 2] * cam[11]
    with np.errstate(divide="ignore", invalid="ignore"):
        nx, ny = xc / ((d * tan_half) * aspect), yc / (d * tan_half)
    return (nx * 0.5 + 0.5) * W, (1.0 - (ny * 0.5 + 0.5)) * H, d


def visibility(tri, tri_link, link_T, cams, pts, fov_deg=60.0, aspect=1.0, near=0.1, far=4.0, width=64, height=64, eps=0.004):
    """numpy restatement of creg_visibility_f64 (same operation order): returns (visible (n) bool, depth (C,H,W))."""
    tri, link_T, cams, pts = (np.asarray(a, np.float64) for a in (tri, link_T, cams, pts))
    tan_half = np.tan(fov_deg * 3.14159265358979323846 / 360.0)
    T = link_T[np.asarray(tri_link)]
    world = np.empty_like(tri)
    for a in range(3):
        world[:, :, a] = ((T[:, None, a, 0] * tri[:, :, 0] + T[:, None, a, 1] * tri[:, :, 1]) + T[:, None, a, 2] * tri[:, :, 2]) + T[:, None, a, 3]
    depth = np.full((len(cams), height, width), np.inf)
    for ci, cam in enumerate(cams):
        X, Y, D = _project(cam, world, tan_half, aspect, width, height)
        for f in range(len(tri)):
            if not (D[f] >= near).all():
                continue
            x, y, dd = X[f], Y[f], D[f]
            area = (x[1] - x[0]) * (y[2] - y[0]) - (x[2] - x[0]) * (y[1] - y[0])
            if area == 0.0:
                continue
            x0, x1 = max(0, int(np.floor(x.min() - 0.5))), min(width - 1, int(np.ceil(x.max() - 0.5)))
            y0, y1 = max(0, int(np.floor(y.min() - 0.5))), min(height - 1, int(np.ceil(y.max() - 0.5)))
            if x1 < x0 or y1 < y0:
                continue
            cx = np.arange(x0, x1 + 1)[None, :] + 0.5
            cy = np.arange(y0, y1 + 1)[:, None] + 0.5
            ia = 1.0 / area
            b0 = ((x[1] - cx) * (y[2] - cy) - (x[2] - cx) * (y[1] - cy)) * ia
            b1 = ((x[2] - cx) * (y[0] - cy) - (x[0] - cx) * (y[2] - cy)) * ia
            b2 = ((x[0] - cx) * (y[1] - cy) - (x[1] - cx) * (y[0] - cy)) * ia
            inside = (b0 >= 0) & (b1 >= 0) & (b2 >= 0)
            d = 1.0 / ((b0 * (1.0 / dd[0]) + b1 * (1.0 / dd[1])) + b2 * (1.0 / dd[2]))
            ok = inside & (d >= near) & (d <= far)
            sub = depth[ci, y0:y1 + 1, x0:x1 + 1]
            sub[ok] = np.minimum(sub[ok], d[ok])
    vis = np.zeros(len(pts), bool)
    for ci, cam in enumerate(cams):
        px, py, d = _project(cam, pts, tan_half, aspect, width, height)
        ok = (d >= near) & (d <= far)
        x, y = np.floor(np.where(ok, px, 0)).astype(int), np.floor(np.where(ok, py, 0)).astype(int)
        ok &= (x >= 0) & (x < width) & (y >= 0) & (y < height)
        zb = depth[ci, np.clip(y, 0, height - 1), np.clip(x, 0, width - 1)]
        vis |= ok & (d <= zb + eps)
    return vis, depth
