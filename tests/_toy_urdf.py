"""A small articulated robot written to disk by the tests: primitive, binary STL, ASCII STL and OBJ visuals;
revolute, fixed and prismatic joints with non-trivial origins.  (No reference asset is read on the GPU box.)"""
import os
import struct

import numpy as np


def _box_tris(sx, sy, sz, offset=(0, 0, 0)):
    c = np.array([[x, y, z] for x in (-sx, sx) for y in (-sy, sy) for z in (-sz, sz)], np.float64) / 2 + np.asarray(offset)
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    return np.array([[c[q[0]], c[q[a]], c[q[a + 1]]] for q in quads for a in (1, 2)]), c, quads


def write_toy_robot(d):
    os.makedirs(os.path.join(d, "meshes"), exist_ok=True)
    t1, _, _ = _box_tris(0.04, 0.04, 0.2, (0, 0, 0.1))
    with open(os.path.join(d, "meshes", "l1.stl"), "wb") as f:                      # binary STL
        f.write(b"toy".ljust(80, b" ") + struct.pack("<I", len(t1)))
        for t in t1.astype(np.float32):
            f.write(struct.pack("<12fH", 0, 0, 0, *t.reshape(-1), 0))
    t2, _, _ = _box_tris(0.03, 0.05, 0.15, (0, 0, 0.075))
    with open(os.path.join(d, "meshes", "l2.STL"), "w") as f:                       # ASCII STL, upper-case suffix
        f.write("solid l2\n")
        for t in t2:
            f.write(" facet normal 0 0 0\n  outer loop\n" + "".join(f"   vertex {v[0]:.9g} {v[1]:.9g} {v[2]:.9g}\n" for v in t)
                    + "  endloop\n endfacet\n")
        f.write("endsolid l2\n")
    _, c3, quads = _box_tris(0.02, 0.02, 0.1, (0, 0, 0.05))
    with open(os.path.join(d, "meshes", "l3.obj"), "w") as f:                       # OBJ with quad faces + normals syntax
        for v in c3:
            f.write(f"v {v[0]:.9g} {v[1]:.9g} {v[2]:.9g}\n")
        f.write("vn 0 0 1\n")
        for q in quads:
            f.write("f " + " ".join(f"{i + 1}//1" for i in q) + "\n")
    urdf = """<?xml version="1.0"?>
<robot name="toy">
  <link name="base"><visual><origin xyz="0 0 0.02" rpy="0 0 0"/><geometry><box size="0.2 0.2 0.04"/></geometry></visual></link>
  <link name="l1"><visual><geometry><mesh filename="meshes/l1.stl"/></geometry></visual></link>
  <link name="l2"><visual><origin xyz="0 0 0" rpy="0 0 0.3"/><geometry><mesh filename="meshes/l2.STL" scale="1 1 1.2"/></geometry></visual></link>
  <link name="l3"><visual><geometry><mesh filename="package://toy_description/meshes/l3.obj"/></geometry></visual></link>
  <link name="tip"><visual><geometry><sphere radius="0.015"/></geometry></visual></link>
  <joint name="waist" type="revolute"><parent link="base"/><child link="l1"/><origin xyz="0 0 0.04" rpy="0 0 0"/>
    <axis xyz="0 0 1"/><limit lower="-3.1" upper="3.1" effort="1" velocity="1"/></joint>
  <joint name="shoulder" type="revolute"><parent link="l1"/><child link="l2"/><origin xyz="0 0 0.2" rpy="0.2 -0.1 0.4"/>
    <axis xyz="0 1 0"/><limit lower="-1.5" upper="1.2" effort="1" velocity="1"/></joint>
  <joint name="slide" type="prismatic"><parent link="l2"/><child link="l3"/><origin xyz="0 0 0.18" rpy="0 0 0"/>
    <axis xyz="0 0 1"/><limit lower="0" upper="0.05" effort="1" velocity="1"/></joint>
  <joint name="wrist" type="revolute"><parent link="l3"/><child link="tip"/><origin xyz="0 0 0.1" rpy="0 0.5 0"/>
    <axis xyz="1 1 0"/><limit lower="-1" upper="2" effort="1" velocity="1"/></joint>
</robot>
"""
    path = os.path.join(d, "toy.urdf")
    with open(path, "w") as f:
        f.write(urdf)
    joints = [dict(name="waist", type="revolute", parent="base", child="l1", xyz=[0, 0, 0.04], rpy=[0, 0, 0], axis=[0, 0, 1]),
              dict(name="shoulder", type="revolute", parent="l1", child="l2", xyz=[0, 0, 0.2], rpy=[0.2, -0.1, 0.4], axis=[0, 1, 0]),
              dict(name="slide", type="prismatic", parent="l2", child="l3", xyz=[0, 0, 0.18], rpy=[0, 0, 0], axis=[0, 0, 1]),
              dict(name="wrist", type="revolute", parent="l3", child="tip", xyz=[0, 0, 0.1], rpy=[0, 0.5, 0], axis=[1, 1, 0])]
    return path, ["base", "l1", "l2", "l3", "tip"], joints
