"""Tiny ascii PLY writer for tests: float64 coordinates printed with repr() so they read back bit-exact."""
import os


def write_ascii_ply(path, pts):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex %d\nproperty double x\nproperty double y\nproperty double z\nend_header\n" % len(pts))
        for p in pts:
            f.write("%r %r %r\n" % (float(p[0]), float(p[1]), float(p[2])))


def write_sequence(root, frames):
    """frames -> root/0000/robot.ply, root/0001/robot.ply, ... (the layout Segments._load_pc globs, cluster_icp.py:35-44)."""
    for t, fr in enumerate(frames):
        write_ascii_ply(os.path.join(root, f"{t:04}", "robot.ply"), fr)
