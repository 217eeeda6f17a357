"""Synthetic raw frames from a URDF + its triangle meshes, with the call surface of the
reference's ``Sim/sim_data.py`` where it makes sense: ``SimEnv(urdf_path, base_position, base_orientation,
dof=..., global_scale=...)`` with ``joint_params / joint_list / dof_list / joint_limits``,
``angle_list(num_step, step_size, dof, joint_limits, scale, seed_i)`` (:372-430, restated call for call on
numpy's legacy generator, so the joint trajectories are the reference's), ``data_collection(env, data_path,
angle_list=..., noise_flag=..., num_points=...)`` and ``save_step_data`` (:231-244: ``{step:04}/robot.ply`` +
``joint_cfg.txt``, ``noise.txt``).

What differs, by design: the reference steps a PyBullet simulation and fuses depth renders of 20 virtual
cameras (:166-198, :262-330); this build evaluates the URDF's forward kinematics itself and samples the visual
mesh surfaces by area on the GPU (``creg_sample_mesh_f64``), then applies the same noise model (:337-343) and the
same farthest-point down-sampling to ``num_points`` (:346,349; ``creg_fps_f64``).  Frames are therefore
geometry-faithful but include surfaces a camera ring would not see.  PyBullet, OpenGL and Open3D are not
needed.  Mesh formats: STL (binary / ASCII) and OBJ; COLLADA visuals raise.  The sampling and the
down-sampling have no CPU fallback.
"""
import os
import struct
import xml.etree.ElementTree as ET

import numpy as np
import torch

from . import _lib, ops
from .cluster_icp import PointCloud
from .fps import farthest_point_sample


# ------------------------------------------------------------------------------------------ joint trajectories
def angle_list(num_step, step_size, dof, joint_limits, scale, seed_i):
    """(num_step, dof) joint angles in radians: per joint, random targets inside the scaled limits at least
    20 % of the range away, approached in steps of step_size * (1 + U[0,1)) degrees (sim_data.py:372-430)."""
    start_rate, low_step_limit = 0.5, 0.2
    np.random.seed(seed_i)
    limits_deg = np.asarray(joint_limits, np.float64) * 180 / np.pi
    scaled = limits_deg * np.asarray(scale, np.float64).reshape(-1, 1)
    span = np.abs(scaled[:, 1] - scaled[:, 0])
    start = scaled[:, 0] + start_rate * (scaled[:, 1] - scaled[:, 0])
    columns = []
    for j in range(dof):
        col = []
        while len(col) < num_step:
            while True:
                target = np.random.rand() * (scaled[j][1] - scaled[j][0]) + scaled[j][0]
                if np.abs(target - start[j]) > low_step_limit * span[j]:
                    break
            step = step_size * (1 + np.random.rand())
            n = int(np.abs(target - start[j]) / step) + 1
            direction = 1 if target > start[j] else -1
            stop = start[j] + direction * step * n
            col += list(np.linspace(start[j], stop, n, endpoint=False))
            start[j] = stop
        columns.append(np.array(col)[:num_step])
    return np.vstack(columns).T * np.pi / 180


# ------------------------------------------------------------------------------------------ meshes
def _load_stl(path):
    with open(path, "rb") as f:
        raw = f.read()
    n = struct.unpack_from("<I", raw, 80)[0] if len(raw) >= 84 else -1
    if n >= 0 and len(raw) == 84 + 50 * n:                      # binary: 80-byte header, count, 50-byte records
        rec = np.frombuffer(raw, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", 9), ("a", "<u2")]), count=n, offset=84)
        return rec["v"].astype(np.float64).reshape(n, 3, 3)
    verts = [ln.split()[1:4] for ln in raw.decode("ascii", "replace").splitlines() if ln.strip().startswith("vertex")]
    v = np.asarray(verts, np.float64)
    if len(v) == 0 or len(v) % 3:
        raise IOError(f"{path}: neither a binary nor an ASCII STL")
    return v.reshape(-1, 3, 3)


def _load_obj(path):
    verts, tris = [], []
    with open(path, "r", errors="replace") as f:
        for ln in f:
            tok = ln.split()
            if not tok:
                continue
            if tok[0] == "v":
                verts.append([float(x) for x in tok[1:4]])
            elif tok[0] == "f":
                idx = [int(t.split("/")[0]) for t in tok[1:]]
                idx = [i - 1 if i > 0 else len(verts) + i for i in idx]
                for a in range(1, len(idx) - 1):                # fan triangulation of polygons
                    tris.append([idx[0], idx[a], idx[a + 1]])
    v = np.asarray(verts, np.float64)
    return v[np.asarray(tris, np.int64)] if tris else np.zeros((0, 3, 3))


def _dae_node_matrix(node, ns):
    """Local transform of a COLLADA <node>: its <matrix> / <translate> / <rotate> / <scale> children composed in
    document order (COLLADA 1.4.1, 5.5 'node': post-multiplied in the order listed)."""
    M = np.eye(4)
    for ch in node:
        tag = ch.tag[len(ns):] if ch.tag.startswith(ns) else ch.tag
        if tag not in ("matrix", "translate", "rotate", "scale") or ch.text is None:
            continue
        v = [float(x) for x in ch.text.split()]
        T = np.eye(4)
        if tag == "matrix":
            T = np.asarray(v, np.float64).reshape(4, 4)          # row-major in the document
        elif tag == "translate":
            T[:3, 3] = v[:3]
        elif tag == "scale":
            T[0, 0], T[1, 1], T[2, 2] = v[:3]
        else:                                                    # rotate: axis x y z, angle in degrees
            T[:3, :3] = _axis_angle(v[:3], np.deg2rad(v[3])) if np.linalg.norm(v[:3]) > 0 else np.eye(3)
        M = M @ T
    return M


def _load_dae(path):
    """Triangles of a COLLADA 1.4 document: every <instance_geometry> of the visual scene with its node transforms,
    <triangles> / <polylist> / <polygons> primitives (polygons fan-triangulated), the asset's unit (metres per unit) and
    up axis applied the way PyBullet's URDF importer does (Y_UP -> +90 degrees about x, X_UP -> -90 degrees about y;
    Bullet LoadMeshFromCollada.cpp).  Materials, normals and texture coordinates are not needed for surface sampling."""
    root = ET.parse(path).getroot()
    ns = root.tag[: root.tag.index("}") + 1] if root.tag.startswith("{") else ""
    f = lambda e, q: e.find(q.replace("c:", ns))
    fa = lambda e, q: e.findall(q.replace("c:", ns))
    unit, up = 1.0, "Y_UP"                                       # COLLADA's defaults
    asset = f(root, "c:asset")
    if asset is not None:
        u = f(asset, "c:unit")
        if u is not None and u.get("meter"):
            unit = float(u.get("meter"))
        a = f(asset, "c:up_axis")
        if a is not None and a.text:
            up = a.text.strip()
    geoms = {}
    for g in fa(root, "c:library_geometries/c:geometry"):
        mesh = f(g, "c:mesh")
        if mesh is None:
            continue
        sources = {}
        for src in fa(mesh, "c:source"):
            fa_ = f(src, "c:float_array")
            acc = f(src, "c:technique_common/c:accessor")
            if fa_ is None or fa_.text is None:
                continue
            stride = int(acc.get("stride", "3")) if acc is not None else 3
            sources["#" + src.get("id")] = np.asarray(fa_.text.split(), np.float64).reshape(-1, stride)
        verts = {}
        for v in fa(mesh, "c:vertices"):
            for inp in fa(v, "c:input"):
                if inp.get("semantic") == "POSITION":
                    verts["#" + v.get("id")] = sources[inp.get("source")]
        tris = []
        for prim in list(fa(mesh, "c:triangles")) + list(fa(mesh, "c:polylist")) + list(fa(mesh, "c:polygons")):
            inputs = fa(prim, "c:input")
            stride = max(int(i.get("offset", "0")) for i in inputs) + 1
            vin = next(i for i in inputs if i.get("semantic") == "VERTEX")
            pos, voff = verts[vin.get("source")][:, :3], int(vin.get("offset", "0"))
            kind = prim.tag[len(ns):]
            plists = [np.asarray(p_.text.split(), np.int64) for p_ in fa(prim, "c:p") if p_.text]
            if kind == "triangles":
                idx = np.concatenate(plists)[voff::stride] if plists else np.zeros(0, np.int64)
                tris.append(pos[idx].reshape(-1, 3, 3))
            else:
                if kind == "polylist":
                    counts = np.asarray(f(prim, "c:vcount").text.split(), np.int64)
                    idx = plists[0][voff::stride]
                    polys, at = [], 0
                    for c in counts:
                        polys.append(idx[at:at + c]); at += c
                else:
                    polys = [pl[voff::stride] for pl in plists]
                for poly in polys:
                    for a in range(1, len(poly) - 1):
                        tris.append(pos[[poly[0], poly[a], poly[a + 1]]][None])
        geoms["#" + g.get("id")] = np.concatenate(tris) if tris else np.zeros((0, 3, 3))
    out = []

    def walk(node, M):
        M = M @ _dae_node_matrix(node, ns)
        for ig in fa(node, "c:instance_geometry"):
            t = geoms.get(ig.get("url"))
            if t is not None and len(t):
                out.append(t @ M[:3, :3].T + M[:3, 3])
        for ch in fa(node, "c:node"):
            walk(ch, M)

    scenes = fa(root, "c:library_visual_scenes/c:visual_scene")
    for sc in scenes:
        for node in fa(sc, "c:node"):
            walk(node, np.eye(4))
    if not out:                                                  # no scene graph: every geometry as it stands
        out = [t for t in geoms.values() if len(t)]
    if not out:
        raise IOError(f"{path}: no triangle geometry found")
    tri = np.concatenate(out) * unit
    if up == "Y_UP":
        tri = tri @ np.array([[1.0, 0, 0], [0, 0, -1.0], [0, 1.0, 0]]).T      # (x, y, z) -> (x, -z, y)
    elif up == "X_UP":
        tri = tri @ np.array([[0, 0, -1.0], [0, 1.0, 0], [1.0, 0, 0]]).T     # (x, y, z) -> (-z, y, x)
    return tri


def load_mesh(path):
    """(F,3,3) float64 triangles of an STL, OBJ or COLLADA (.dae) file."""
    ext = os.path.splitext(path)[1].lower()
    if ext == ".stl":
        return _load_stl(path)
    if ext == ".obj":
        return _load_obj(path)
    if ext == ".dae":
        return _load_dae(path)
    raise NotImplementedError(f"{path}: only STL, OBJ and COLLADA visuals are supported")


def _primitive(geom):
    """Triangles of a URDF <box>/<cylinder>/<sphere> visual."""
    tag = geom.tag
    if tag == "box":
        sx, sy, sz = (float(v) / 2 for v in geom.get("size").split())
        c = np.array([[x, y, z] for x in (-sx, sx) for y in (-sy, sy) for z in (-sz, sz)])
        quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
        return np.array([[c[q[0]], c[q[a]], c[q[a + 1]]] for q in quads for a in (1, 2)])
    if tag in ("cylinder", "sphere"):
        r = float(geom.get("radius"))
        seg = 24
        th = np.linspace(0, 2 * np.pi, seg + 1)
        tris = []
        if tag == "cylinder":
            h = float(geom.get("length")) / 2
            for a, b in zip(th[:-1], th[1:]):
                pa, pb = np.array([r * np.cos(a), r * np.sin(a)]), np.array([r * np.cos(b), r * np.sin(b)])
                tris += [[[*pa, -h], [*pb, -h], [*pb, h]], [[*pa, -h], [*pb, h], [*pa, h]],
                         [[0, 0, h], [*pa, h], [*pb, h]], [[0, 0, -h], [*pb, -h], [*pa, -h]]]
        else:
            ph = np.linspace(0, np.pi, seg // 2 + 1)
            pt = lambda t, p: [r * np.sin(p) * np.cos(t), r * np.sin(p) * np.sin(t), r * np.cos(p)]
            for a, b in zip(th[:-1], th[1:]):
                for c0, c1 in zip(ph[:-1], ph[1:]):
                    tris += [[pt(a, c0), pt(a, c1), pt(b, c1)], [pt(a, c0), pt(b, c1), pt(b, c0)]]
        return np.asarray(tris, np.float64)
    raise NotImplementedError(f"URDF geometry <{tag}>")


# ------------------------------------------------------------------------------------------ URDF
def _rpy_matrix(rpy):
    """URDF fixed-axis roll-pitch-yaw: R = Rz(yaw) Ry(pitch) Rx(roll)."""
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def _origin(elem):
    T = np.eye(4)
    o = elem.find("origin") if elem is not None else None
    if o is not None:
        T[:3, :3] = _rpy_matrix([float(v) for v in o.get("rpy", "0 0 0").split()])
        T[:3, 3] = [float(v) for v in o.get("xyz", "0 0 0").split()]
    return T


def _axis_angle(axis, q):
    a = np.asarray(axis, np.float64)
    a = a / np.linalg.norm(a)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(q) * K + (1 - np.cos(q)) * (K @ K)


class UrdfRobot:
    """Kinematic tree + visual triangles of a URDF.  ``links`` in file order; ``joints`` in file order (PyBullet
    numbers joints the same way); ``tri`` (F,3,3) in link frames, ``tri_link`` (F,), ``cum_area`` (F,)."""

    def __init__(self, urdf_path, global_scale=1.0, package_dirs=()):
        self.path = os.path.abspath(urdf_path)
        root = ET.parse(self.path).getroot()
        self.links = [l.get("name") for l in root.findall("link")]
        self.link_index = {n: i for i, n in enumerate(self.links)}
        self.joints = []
        for j in root.findall("joint"):
            lim = j.find("limit")
            ax = j.find("axis")
            self.joints.append({
                "name": j.get("name"), "type": j.get("type"),
                "parent": j.find("parent").get("link"), "child": j.find("child").get("link"),
                "origin": _origin(j), "axis": [float(v) for v in (ax.get("xyz") if ax is not None else "1 0 0").split()],
                "limit": [float(lim.get("lower", 0)), float(lim.get("upper", 0))] if lim is not None else [0.0, 0.0]})
        for j in self.joints:
            j["origin"][:3, 3] *= global_scale
        children = {j["child"] for j in self.joints}
        roots = [l for l in self.links if l not in children]
        if len(roots) != 1:
            raise ValueError(f"{urdf_path}: expected one root link, found {roots}")
        self.root = roots[0]
        tris, owner = [], []
        for l in root.findall("link"):
            for vis in l.findall("visual"):
                geom = vis.find("geometry")
                if geom is None or len(geom) == 0:
                    continue
                g = geom[0]
                if g.tag == "mesh":
                    t = load_mesh(self._resolve(g.get("filename"), package_dirs))
                    t = t * np.array([float(v) for v in g.get("scale", "1 1 1").split()])
                else:
                    t = _primitive(g)
                T = _origin(vis)
                t = (t @ T[:3, :3].T + T[:3, 3]) * global_scale
                tris.append(t)
                owner.append(np.full(len(t), self.link_index[l.get("name")], np.int32))
        if not tris:
            raise ValueError(f"{urdf_path}: no visual geometry")
        self.tri = np.concatenate(tris)
        self.tri_link = np.concatenate(owner)
        e1, e2 = self.tri[:, 1] - self.tri[:, 0], self.tri[:, 2] - self.tri[:, 0]
        area = 0.5 * np.linalg.norm(np.cross(e1, e2), axis=1)
        keep = area > 0                                          # degenerate facets carry no surface
        self.tri, self.tri_link, area = self.tri[keep], self.tri_link[keep], area[keep]
        self.cum_area = np.cumsum(area)

    def _resolve(self, filename, package_dirs):
        if filename.startswith("package://"):
            rel = filename[len("package://"):]
            cands = [os.path.join(d, rel) for d in package_dirs] + [os.path.join(d, rel.split("/", 1)[-1]) for d in package_dirs]
            here = os.path.dirname(self.path)
            for up in range(4):                                  # packages usually sit a few levels above urdf/
                cands += [os.path.join(here, rel), os.path.join(here, rel.split("/", 1)[-1])]
                here = os.path.dirname(here)
        else:
            # relative names: next to the URDF, or relative to one of its ancestors / the working directory (the
            # reference's URDFs name meshes from the repository root, where its scripts are run)
            cands = [filename] if os.path.isabs(filename) else []
            here = os.path.dirname(self.path)
            for up in range(6):
                cands.append(os.path.join(here, filename))
                here = os.path.dirname(here)
            cands += [os.path.join(d, filename) for d in package_dirs] + [os.path.abspath(filename)]
        for c in cands:
            if os.path.exists(c):
                return c
        raise FileNotFoundError(f"mesh {filename!r} of {self.path} not found (tried {cands[:3]} ...)")

    def fk(self, q_by_joint, base=None):
        """Link poses (L,4,4) float64 for joint positions {name: value} (missing joints at 0)."""
        T = np.tile(np.eye(4), (len(self.links), 1, 1))
        T[self.link_index[self.root]] = np.eye(4) if base is None else base
        done = {self.root}
        pending = list(self.joints)
        while pending:
            rest = []
            for j in pending:
                if j["parent"] not in done:
                    rest.append(j)
                    continue
                M = np.eye(4)
                q = float(q_by_joint.get(j["name"], 0.0))
                if j["type"] in ("revolute", "continuous"):
                    M[:3, :3] = _axis_angle(j["axis"], q)
                elif j["type"] == "prismatic":
                    a = np.asarray(j["axis"], np.float64)
                    M[:3, 3] = a / np.linalg.norm(a) * q
                T[self.link_index[j["child"]]] = T[self.link_index[j["parent"]]] @ j["origin"] @ M
                done.add(j["child"])
            if len(rest) == len(pending):
                raise ValueError(f"{self.path}: joints {[j['name'] for j in rest]} hang off unknown links")
            pending = rest
        return T


class SimEnv:
    """The part of the reference's SimEnv (sim_data.py:15-64) that describes the robot: revolute joints in URDF
    order with their limits (:66-82), the first ``dof`` of them driven, the rest parked at mid range (:131-157)."""

    def __init__(self, urdf_path, base_position=[0, 0, 0], base_orientation=[0, 0, 0], gui=False, dof=5,
                 ground_flag=False, radius=1.5, num_cameras=3, global_scale=1.0, package_dirs=()):
        if gui:
            raise NotImplementedError("gui=True needs PyBullet's viewer (out of scope)")
        self.dof = dof
        self.robot = UrdfRobot(urdf_path, global_scale, package_dirs)
        self.base = np.eye(4)
        self.base[:3, :3] = _rpy_matrix(base_orientation)
        self.base[:3, 3] = base_position
        self.joint_params = {j["name"]: list(j["limit"]) for j in self.robot.joints if j["type"] == "revolute"}
        self.joint_list = list(self.joint_params.keys())
        self.dof_list = self.joint_list[:dof]
        self.joint_limits = np.array([self.joint_params[j] for j in self.dof_list])
        self._dev = None
        self._setup_cameras(radius, num_cameras)

    def _setup_cameras(self, radius, num_cameras=20, cam_angle=20):
        """The reference's camera ring (sim_data.py:88-116): fewer than 20 cameras evenly on a circle at `cam_angle` degrees
        elevation; 20 or more drawn from numpy's GLOBAL RandomState like the reference (uniform azimuth, elevation in
        [0, pi/2)); all on a sphere of `radius` looking at the origin, +z up, fov 60, aspect 1, near 0.1, far 4.
        `self.cameras` keeps the reference's dict list; `self.cam_frames` (C,12) = eye | forward | right | up."""
        if num_cameras < 20:
            theta = np.linspace(0, 2 * np.pi, num_cameras, endpoint=False)
            phi = np.pi * np.array([cam_angle] * num_cameras) / 180
        else:
            theta = np.random.rand(num_cameras) * 2 * np.pi
            phi = np.random.rand(num_cameras) * np.pi / 2
        xs, ys, zs = radius * np.cos(theta) * np.cos(phi), radius * np.sin(theta) * np.cos(phi), radius * np.sin(phi)
        self.cameras = [{'camera_pos': [x, y, z], 'target_pos': [0, 0, 0], 'up_vector': [0, 0, 1], 'fov': 60, 'aspect': 1.0,
                         'near_val': 0.1, 'far_val': 4} for x, y, z in zip(xs, ys, zs)]
        frames = []
        for c in self.cameras:
            e = np.asarray(c['camera_pos'], np.float64)
            f = (np.asarray(c['target_pos'], np.float64) - e)
            f = f / np.linalg.norm(f)
            s_ = np.cross(f, np.asarray(c['up_vector'], np.float64))
            s_ = s_ / np.linalg.norm(s_)
            frames.append(np.concatenate([e, f, s_, np.cross(s_, f)]))
        self.cam_frames = np.asarray(frames)

    def visible(self, joint_positions, pts, width=800, height=800, eps=0.004):
        """Which of `pts` (n,3 world points, device tensor) some camera of the ring sees (creg_visibility_f64)."""
        tri, _, own = self._device_mesh()
        T = torch.as_tensor(self.robot.fk(joint_positions, self.base), device=tri.device)
        cams = torch.as_tensor(self.cam_frames, device=tri.device)
        c = self.cameras[0]
        return ops.visibility(tri, own, T, cams, pts, c['fov'], c['aspect'], c['near_val'], c['far_val'], width, height, eps)

    def _device_mesh(self):
        if self._dev is None:
            d = _lib.device()
            r = self.robot
            self._dev = (torch.as_tensor(r.tri, device=d).contiguous(), torch.as_tensor(r.cum_area, device=d),
                         torch.as_tensor(r.tri_link, device=d))
        return self._dev

    def set_joint_positions(self, commands, manual_positions=0):
        """Joint name -> position: commanded for the driven joints, mid range (+ manual offset) for the others.
        (The reference reads the positions back from the physics step; here they are exact.)"""
        q = {}
        for j_id, name in enumerate(self.joint_list):
            lo, hi = sorted(self.joint_params[name])
            q[name] = float(commands[j_id]) if name in self.dof_list else (hi + lo) / 2 + manual_positions * (hi - lo) / 2
        return q

    def sample_surface(self, joint_positions, n, rng):
        """n area-weighted surface points of the posed robot, on the GPU (creg_sample_mesh_f64)."""
        tri, cum, own = self._device_mesh()
        T = torch.as_tensor(self.robot.fk(joint_positions, self.base), device=tri.device)
        u = torch.as_tensor(rng.random((n, 3)), device=tri.device)
        return ops.sample_mesh(tri, cum, own, T, u)

    def reset(self):
        self._dev = None


def save_step_data(step_id, combined_pcds, joint_positions, data_path, dof_list):
    """{step:04}/robot.ply (binary little-endian, double x/y/z as Open3D writes points) + joint_cfg.txt
    (sim_data.py:231-244)."""
    sub = data_path + f"{step_id:04}/"
    os.makedirs(sub, exist_ok=True)
    pts = np.asarray(combined_pcds.points, np.float64)
    with open(sub + "robot.ply", "wb") as f:
        f.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty double x\nproperty double y\n"
                 "property double z\nend_header\n" % len(pts)).encode("ascii"))
        f.write(np.ascontiguousarray(pts, "<f8").tobytes())
    with open(sub + "joint_cfg.txt", "w") as f:
        for name, pos in joint_positions.items():
            if name in dof_list:
                f.write(f"{name}:{pos:,.6f}\n")


def data_collection(env, data_path=None, width=800, height=800, visualize=False, angle_list=None, ground_flag=False,
                    noise_flag=False, num_points=5000, collision_flag=False, oversample=4, seed=0, occlusion=True):
    """One sequence: for every row of ``angle_list`` pose the robot, sample surface points, keep those that at least one
    camera of the ring sees (``occlusion``: depth buffers of ``width`` x ``height`` like the reference's rendered images,
    sim_data.py:286-306; more samples are drawn until ``oversample * num_points`` visible ones exist), add the
    reference's noise (translation N(0, 0.01) per frame and N(0, 0.0005) per point, not on the first frame;
    sim_data.py:333-343), farthest-point down-sample to ``num_points`` (:346,349) and save.
    Returns (collision=False, list of PointCloud) like the reference (self-collision checking is PyBullet's)."""
    if visualize:
        raise NotImplementedError("visualize=True needs Open3D's viewer (out of scope)")
    rng = np.random.default_rng(seed)
    noise, record = [], []
    for jp_id, cmd in enumerate(np.asarray(angle_list)):
        q = env.set_joint_positions(cmd)
        want = oversample * num_points
        pts = env.sample_surface(q, want, rng)
        if occlusion:
            kept = pts[env.visible(q, pts, width, height)]
            draws = 1
            while kept.shape[0] < want and draws < 16:            # interior / hidden surfaces: draw until enough are visible
                more = env.sample_surface(q, want, rng)
                kept = torch.cat([kept, more[env.visible(q, more, width, height)]])
                draws += 1
            if kept.shape[0] < num_points:
                raise RuntimeError(f"only {kept.shape[0]} of the sampled surface points are visible from the camera ring")
            pts = kept[:max(want, num_points)] if kept.shape[0] >= want else kept
        if noise_flag and jp_id != 0:
            pos_noise = rng.normal(0, 0.01, size=3)
            noise.append(pos_noise)
            pts = pts + torch.as_tensor(pos_noise, device=pts.device)
            pts = pts + torch.as_tensor(rng.normal(0, 0.0005, size=tuple(pts.shape)), device=pts.device)
        sel = farthest_point_sample(pts, num_points)
        cloud = PointCloud(pts[torch.as_tensor(sel, device=pts.device)].cpu().numpy())
        if data_path is not None:
            save_step_data(jp_id, cloud, q, data_path, env.dof_list)
        record.append(cloud)
    if noise_flag and data_path is not None:
        np.savetxt(data_path + "noise.txt", np.array(noise).reshape(-1, 3), fmt="%.6f")
    return False, record


def collect(robot, robot_params, num_step=10, step_size=4, epochs=5, scale=0.9, noise=True, num_points=5000,
            num_cameras=20, root="."):
    """`epochs` sequences of `num_step` frames under data/raw/{robot}/{step_size}_deg_{num_cameras}_cams/V{seed:04}/
    -- the directory layout of the reference's collect() (sim_data.py:465-531), which match() globs
    (mlp_reg.py:424).  robot_params needs the reference's keys 'gt' (URDF path), 'dof' and optionally 'sim_ori'.
    The self-collision rejection of seeds is PyBullet's and is not reproduced: seeds are 0..epochs-1."""
    paths = []
    for seed in range(epochs):
        data_path = os.path.join(root, f"data/raw/{robot}/{step_size}_deg_{num_cameras}_cams/V{seed:04}/")
        os.makedirs(data_path, exist_ok=True)
        np.random.seed(seed)                                       # the ring of >= 20 cameras draws from the global state
        env = SimEnv(os.path.join(root, robot_params["gt"]), base_orientation=robot_params.get("sim_ori", [0, 0, 0]),
                     dof=robot_params["dof"], radius=robot_params.get("cam_dist", 1.5), num_cameras=num_cameras)
        a_list = angle_list(num_step, step_size, robot_params["dof"], env.joint_limits, np.array([scale] * robot_params["dof"]), seed)
        data_collection(env, data_path=data_path, angle_list=a_list, noise_flag=noise, num_points=num_points, seed=seed)
        env.reset()
        paths.append(data_path)
    return paths


def main(argv=None):
    """python -m autourdf_amd.sim_data --robot wx200_5 [...]: the reference's flags (sim_data.py:537-551) minus the
    rendering ones; reads 'gt' / 'dof' / 'sim_ori' of the robot from ./parameters.json."""
    import argparse
    import json
    ap = argparse.ArgumentParser()
    ap.add_argument('--robot', type=str, default='franka')
    ap.add_argument('--scale', type=float, default=0.9)
    ap.add_argument('--step_size', type=int, default=4)
    ap.add_argument('--num_step', type=int, default=10)
    ap.add_argument('--epoch', type=int, default=5)
    ap.add_argument('--no_noise', action='store_true')
    ap.add_argument('--num_points', type=int, default=5000)
    ap.add_argument('--num_cameras', type=int, default=20, help="only names the output directory here")
    args = ap.parse_args(argv)
    with open('parameters.json') as f:
        params = json.load(f)[args.robot]
    if 'gt' not in params:
        raise SystemExit(f"parameters.json has no 'gt' URDF path for {args.robot!r} (use the reference's parameters.json)")
    for p in collect(args.robot, params, args.num_step, args.step_size, args.epoch, args.scale, not args.no_noise,
                     args.num_points, args.num_cameras):
        print(p)


if __name__ == "__main__":
    main()
