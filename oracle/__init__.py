"""CPU oracle for the AutoURDF cluster-registration path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  Nothing under ``autourdf_amd/`` imports it; the product path has no CPU fallback.

What it restates (file:line into /root/reference, or the third-party wheel named):

* ``oracle.transforms``   pytorch3d==0.7.7 ``pytorch3d.transforms`` (not vendored; call sites
                          mlp_reg.py:13,65-90, dq_func.py:2)                 -- parity UNPINNED
* ``oracle.chamfer``      pytorch3d==0.7.7 ``loss.chamfer_distance(norm=1)`` (mlp_reg.py:96)
                                                                             -- parity UNPINNED
* ``oracle.kmeans``       scikit-learn ``cluster.k_means`` Lloyd path (mlp_reg.py:204)
                          -- PINNED against live sklearn 1.7.2 (tests/test_oracle_kmeans.py)
* ``oracle.dq``           reference PointCloud/dq_func.py:4-257
                          -- PINNED against the reference module imported under shims
                             (tests/golden/*.npz, made by tests/golden/make_golden.py)
* ``oracle.models``       reference PointCloud/model_utils.py:65-168 (QRegMLP, DQRegMLP) -- PINNED
* ``oracle.registration`` reference PointCloud/mlp_reg.py:17-237 (train, calculate_pc,
                          resample_cluster)                                  -- PINNED (composition
                          logic; third-party arithmetic inside it is the restatements above)
* ``oracle.icp``          open3d==0.18.0 ``registration_icp`` point-to-point (cluster_icp.py:157)
                          + reference cluster_icp.py:118-191 masking         -- parity UNPINNED

* ``oracle.coord_map``    reference PointCloud/coord_map.py:175-332 (CoordMap.load_matrix pose->quaternion,
                          get_scale, coord_dist_map both branches, coord_dist_map_legacy) -- PINNED
                          (reference loops run under shims -> tests/golden/coord_map_reference.npz);
                          the roma functions inside (rotmat_to_rotvec, rotvec_geodesic_distance,
                          rotmat_geodesic_distance; wheel absent)            -- parity UNPINNED

* ``oracle.link``         reference PointCloud/link.py:85-127 (refine_links_clusters) and the ICP filter of
                          Sim/evaluation.py:358-362                          -- PINNED (composition; reference
                          function run on disk under shims -> tests/golden/link_refine_reference.npz)

* ``oracle.sim_data``     numpy restatement of creg_sample_mesh_f64 (bit-exact check) and an independent
                          URDF forward kinematics (scipy Rotation); ``angle_list`` is pinned by the reference's
                          own function (Sim/sim_data.py:372-430 -> tests/golden/sim_angle_list.npz); the
                          PyBullet/OpenGL rendering it replaces is not restatable     -- parity UNPINNED

"UNPINNED" = the reference repository holds no test, golden vector or vendored source for that
third-party arithmetic; the restatement follows the published algorithm and
is cross-checked against independent implementations (scipy Rotation, torch.cdist, numpy SVD).
"""
