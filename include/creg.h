/*
 * creg.h -- C ABI of libcreg.so: MI355X (gfx950) kernels for AutoURDF's cluster-registration path.
 *
 * The reference (jl6017/AutoURDF) is pure Python and has no FFI layer of its own; the boundary
 * that `match()` / scripts/registration.sh exercise is the Python module surface of
 * PointCloud/{mlp_reg,cluster_icp,dq_func}.py.  Each entry point below names the
 * reference interface (file:line under /root/reference) whose arithmetic it replaces; the
 * same-named Python modules in autourdf_amd/ bind these through ctypes (autourdf_amd/_lib.py),
 * INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch / C++ types.
 *   - Every data pointer is a DEVICE pointer on the current HIP device unless marked HOST.
 *   - Row-major, xyz interleaved (N,3) exactly as the reference stores clouds (mlp_reg.py:296).
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream).
 *     All work is enqueued asynchronously on it; nothing synchronises unless documented.
 *   - The library never allocates caller-visible memory and keeps no pointer after return,
 *     except inside an explicit plan object (creg_train_plan_*), which owns only the workspace
 *     the caller handed it.
 *   - Return 0 on success, a negative creg_status otherwise; creg_last_error() gives the
 *     thread-local message.  No exception crosses the boundary.
 *   - Quaternions are real-first (w,x,y,z); dual quaternions are [real | dual] (8 floats).
 */
#ifndef CREG_H
#define CREG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* creg_stream_t;

enum creg_status {
    CREG_OK = 0,
    CREG_EINVAL = -1,       /* bad argument (null pointer, negative size, unsupported option) */
    CREG_EHIP = -2,         /* a HIP runtime call failed */
    CREG_ENOTCONVERGED = -3,
    CREG_EARCH = -4         /* current device is not gfx950 */
};

int creg_version(void);                 /* 10000*major + 100*minor + patch */
const char* creg_last_error(void);
/* 0 when the current device is a gfx950 part; CREG_EARCH / CREG_EHIP otherwise. */
int creg_device_check(void);

/* ------------------------------------------------------------------------------------------
 * K1  L1 nearest neighbour (K=1), both directions of the Chamfer term.
 * Replaces pytorch3d knn_points as reached by chamfer_distance(pred, y, norm=1) at
 * mlp_reg.py:96 (and Sim/evaluation.py:81).  d = (|dx|+|dy|)+|dz| in fp32, FIRST minimum wins.
 * x (nx,3), y (ny,3) fp32.  dx (nx) / ix (nx): distance / index of the nearest y for every x;
 * dy (ny) / iy (ny): the same for every y among x.  Either direction may be skipped by passing
 * NULL for both of its outputs.  nx, ny >= 1. */
int creg_nn_l1_bidir_f32(const float* x, int64_t nx, const float* y, int64_t ny,
                         float* dx, int64_t* ix, float* dy, int64_t* iy, creg_stream_t stream);

/* Backward of  gx_scale * sum_i |x_i - y[ix_i]|_1 + gy_scale * sum_j |y_j - x[iy_j]|_1  w.r.t. x
 * with pytorch3d's sign rule (p1 > p2 ? +1 : -1).  grad_x (nx,3) is overwritten.
 * `scratch` must hold creg_nn_l1_bwd_scratch_bytes(nx) bytes. */
size_t creg_nn_l1_bwd_scratch_bytes(int64_t nx);
int creg_nn_l1_bwd_f32(const float* x, int64_t nx, const float* y, int64_t ny,
                       const int64_t* ix, const int64_t* iy, float gx_scale, float gy_scale,
                       float* grad_x, void* scratch, creg_stream_t stream);

/* mean_i dx_i + mean_j dy_j  -> *loss (one float, device).  Replaces the reduction half of
 * chamfer_distance (point_reduction = batch_reduction = "mean", batch 1). */
int creg_chamfer_l1_reduce_f32(const float* dx, int64_t nx, const float* dy, int64_t ny,
                               float* loss, creg_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K3  per-cluster rigid transform.  Replaces calculate_pc (mlp_reg.py:155-170):
 * out[i] = R_k pts[i] + t_k for i in [seg_offsets[k], seg_offsets[k+1]), n = seg_offsets[k_last+1].
 * pts/out (n,3) fp32, M (k,4,4) fp32 row-major, seg_offsets (k+1) int32 DEVICE. */
int creg_cluster_transform_f32(const float* pts, int64_t n, const int32_t* seg_offsets, int32_t k,
                               const float* M, float* out, creg_stream_t stream);
/* grad_M (k,4,4): rows 0..2 get [sum g (x) p | sum g], row 3 is zero. */
int creg_cluster_transform_bwd_f32(const float* pts, const int32_t* seg_offsets, int32_t k,
                                   const float* grad_out, float* grad_M, creg_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K2  Lloyd k-means, fp64, 3-D.  Replaces sklearn.cluster.k_means(X, init=<array>, n_init=1)
 * as called by resample_cluster (mlp_reg.py:204) and Segments.k_means_cluster
 * (cluster_icp.py:67, after k-means++ seeding on the host).
 *   distance  d = fma(x2,b2, fma(x1,b1, fma(x0,b0, |c|^2))),  b = -2c   (first minimum wins)
 * X (n,3) fp64 is read only (centring happens on an internal copy), init (k,3) fp64.
 * Outputs: centers (k,3) fp64, labels (n) int32, inertia (1) fp64, n_iter (1) int32 -- device.
 * use_mfma != 0 selects the v_mfma_f64_16x16x4_f64 assignment kernel: every centre for every point, in the caller's order,
 * one launch per Lloyd iteration (E-step, exact incremental M-step sums, M-step tail; the convergence test runs on the
 * device and later launches return at once; the host enqueues 32 launches at a time and synchronises the stream in between
 * to read the `done` word).
 * use_mfma == 0 (the default of the drop-ins): the E-step runs over a spatially sorted copy of the frame and evaluates, per
 * workgroup and per wave, only the centres that can be nearest inside the bounding box of its points, and a persistent
 * kernel iterates inside one launch (the host synchronises once per run of iterations between two empty-cluster
 * relocations; when the device cannot hold the kernel's grid at once, one launch per iteration as above).  Both forms return
 * identical labels, centres, inertia and iteration counts. */
size_t creg_kmeans_workspace_bytes(int64_t n, int32_t k);
int creg_kmeans_lloyd_f64(const double* X, int64_t n, const double* init, int32_t k,
                          int32_t max_iter, double tol_rel, int32_t use_mfma,
                          double* centers, int32_t* labels, double* inertia, int32_t* n_iter,
                          void* workspace, size_t workspace_bytes, creg_stream_t stream);
/* The same k_means() for `batch` (<= 16) independent frames of identical size in ONE launch, one
 * workgroup per frame, centres and labels resident in LDS (n <= 16384, k <= 128; the centred frame too up to 5120 points, read from
 * L2 above that), fully asynchronous (no host
 * synchronisation at all).  Bit-identical to creg_kmeans_lloyd_f64.  X, init, centers, labels, inertia,
 * n_iter are HOST arrays of `batch` device pointers. */
size_t creg_kmeans_batch_workspace_bytes(int64_t n, int32_t k, int32_t batch);
int creg_kmeans_lloyd_batch_f64(const double* const* X, int64_t n, const double* const* init, int32_t k,
                                int32_t batch, int32_t max_iter, double tol_rel, double* const* centers,
                                int32_t* const* labels, double* const* inertia, int32_t* const* n_iter,
                                void* workspace, size_t workspace_bytes, creg_stream_t stream);
/* E-step only: labels[i] = argmin_k d(X_i, C_k) (no centring).  Asynchronous. */
int creg_kmeans_assign_f64(const double* X, int64_t n, const double* C, int32_t k,
                           int32_t use_mfma, int32_t* labels, creg_stream_t stream);

/* Stable grouping of points by label + change of frame, fp64.  Replaces the per-cluster mask /
 * inv(M_k) . [p;1] loop of resample_cluster (mlp_reg.py:208-217) and Segments.k_means_cluster
 * (cluster_icp.py:86-99).  M (k,4,4) fp64 local->world; out_local (n,3) holds the clusters
 * back to back in label order, each keeping the original point order; seg_offsets (k+1) int32.
 * m_is_inverse == 0: M is inverted in the kernel in fp64 (Gauss-Jordan, partial pivoting).
 * m_is_inverse != 0: M already holds inv(M_k) -- the drop-in resample_cluster inverts each pose on the host with
 * np.linalg.inv IN THE POSE'S OWN DTYPE exactly as mlp_reg.py:211 does (float32 on the default path: LAPACK sgesv,
 * whose rounding is BLAS-build specific and is not reproduced on the device), so cluster/NNNN.npz matches the
 * reference to the rounding of one 4-term fp64 dot product.  out = fma(I2,z, fma(I1,y, I0*x)) + I3 per row. */
int creg_group_to_local_f64(const double* X, int64_t n, const int32_t* labels, int32_t k,
                            const double* M, int32_t m_is_inverse, double* out_local, int32_t* seg_offsets,
                            creg_stream_t stream);
/* The same for `batch` (<= 16) frames of identical n and k in one pair of launches; X, labels, M, out_local and
 * seg_offsets are HOST arrays of `batch` device pointers.  Identical results to separate calls. */
int creg_group_to_local_batch_f64(const double* const* X, int64_t n, const int32_t* const* labels, int32_t k,
                                  const double* const* M, int32_t m_is_inverse, int32_t batch,
                                  double* const* out_local, int32_t* const* seg_offsets, creg_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * N1  farthest-point down-sampling, fp64.  Replaces open3d farthest_point_down_sample as used by
 * Segments._load_pc (cluster_icp.py:43): start at point 0, repeatedly select the point farthest
 * (squared L2) from the selected set, first maximum wins.  sel (m) int64 indices in selection order.
 * scratch: creg_fps_scratch_bytes(n) bytes. */
size_t creg_fps_scratch_bytes(int64_t n);
int creg_fps_f64(const double* X, int64_t n, int64_t m, int64_t* sel, void* scratch, creg_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K5  SE(3) <-> dual quaternion.  Replace transform_to_dualquat (dq_func.py:100-124) and
 * dualquat_to_transform (dq_func.py:170-186) incl. the pytorch3d matrix_to_quaternion /
 * quaternion_to_matrix they call.  M (k,4,4), dq (k,8) fp32.  The *_bwd forms give the vector-
 * Jacobian products used when these sit inside autograd (mlp_reg.py:78-84). */
int creg_se3_to_dq_f32(const float* M, int32_t k, float* dq, creg_stream_t stream);
int creg_dq_to_se3_f32(const float* dq, int32_t k, float* M, creg_stream_t stream);
int creg_dq_to_se3_bwd_f32(const float* dq, const float* grad_M, int32_t k, float* grad_dq,
                           creg_stream_t stream);
/* Remaining dq_func.py algebra on (k,8)/(k,4) rows (dq_func.py:29,47,126,148,188,213,238). */
int creg_dq_multiply_f32(const float* a, const float* b, int32_t k, float* out, creg_stream_t stream);
int creg_dq_invert_f32(const float* dq, int32_t k, float* out, creg_stream_t stream);
int creg_dq_to_quat_trans_f32(const float* dq, int32_t k, float* q, float* t, creg_stream_t stream);
int creg_quat_trans_to_dq_f32(const float* q, const float* t, int32_t k, float* dq, creg_stream_t stream);
/* pytorch3d.transforms rows used on the path (mlp_reg.py:65,68). */
int creg_matrix_to_quat_f32(const float* R, int32_t k, float* q, creg_stream_t stream);   /* R (k,3,3) */
int creg_quat_to_matrix_f32(const float* q, int32_t k, float* R, creg_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K4  masked point-to-point ICP per cluster, fp64.  Replaces masked_icp (cluster_icp.py:118-191)
 * including open3d registration_icp (TransformationEstimationPointToPoint, max_iteration,
 * relative_fitness = relative_rmse = 1e-6).
 * local (n,3) fp64 clusters back to back (the ICP sources, n = seg_offsets[k] points in total), world fp32
 * predicted clusters back to back (only their boxes are used: cluster_icp.py:133-135), frame (nf,3) fp64 target,
 * M (k,4,4) fp64 initial poses.  world_offsets (k+1, DEVICE) are the segment offsets of `world`; NULL = the same
 * as seg_offsets.  They differ in match()'s --mlp_icp branch: `clusters_local` is the frame-0 segmentation for the
 * whole sequence (mlp_reg.py:248, never reassigned), `clusters_world` the trained clouds of the current, re-sampled
 * one (mlp_reg.py:301-306,325), so cluster i has another point count in the two lists from frame 2 on.
 * M_out (k,4,4) fp64, world_out (n,3) fp64 = M_out applied to local.  keep_translation mirrors `ori`.
 * The nearest target of a source point is found exactly (the exhaustive scan's result, lowest frame index among equidistant
 * candidates) over binned lists pruned by the distance to the previous iteration's match.
 * Two regimes, same algorithm: clusters a CU can hold (<= 1024 points on average, frames <= 65536 points) run the whole
 * loop in ONE asynchronous launch; larger ones (BASELINE configs[4]: 2048-point clusters, 262144-point frames) run it
 * one launch per iteration over many workgroups: the host enqueues batches of 16 launches and reads the number of clusters
 * (and source chunks) still iterating through pinned memory ONE BATCH BEHIND the device (round 4: no stream synchronisation in
 * the loop; the call returns when the counters it has read say "converged", with at most one surplus batch of empty launches
 * and the final copy-out still in the queue -- outputs are ready when the stream is).  The results of the two regimes agree to
 * rounding (different summation trees), not bit for bit.
 * workspace: creg_icp_workspace_bytes(n, nf, k). */
size_t creg_icp_workspace_bytes(int64_t n, int64_t nf, int32_t k);
int creg_masked_icp_f64(const double* local, const float* world, const int32_t* world_offsets, int64_t n,
                        const int32_t* seg_offsets, int32_t k, const double* frame, int64_t nf, const double* M,
                        double scale, double th, int32_t max_iteration, int32_t keep_translation,
                        double* M_out, double* world_out, int32_t* n_iter_out,
                        void* workspace, size_t workspace_bytes, creg_stream_t stream);

/* The same for `batch` (<= 16) independent problems of identical n, nf and k (the frames of several
 * sequences) in ONE launch: grid (k, batch).  Results are those of `batch` separate calls, bit for bit.
 * workspace: creg_icp_batch_workspace_bytes(n, nf, k, batch). */
typedef struct creg_icp_problem {
    const double* local; const float* world; const int32_t* seg_offsets; const double* frame; const double* M;
    double* M_out; double* world_out; int32_t* n_iter_out;
    const int32_t* tgt_offsets;   /* NULL: masked mode above.  Non-NULL (k+1 offsets into `frame`): point-to-point
                                     mode, cluster i registers to frame[tgt_offsets[i] .. tgt_offsets[i+1]) unmasked,
                                     `world` is ignored and nf is the total number of target points */
    const int32_t* world_offsets; /* masked mode: (k+1) segment offsets of `world`; NULL = seg_offsets */
    /* masked mode with world == NULL: the mask boxes are those of the clusters in their CURRENT pose, i.e. of
       float32(M) applied to float32(local) exactly as creg_cluster_transform_f32 evaluates it -- the caller
       saves that launch and the two casts */
} creg_icp_problem;
size_t creg_icp_batch_workspace_bytes(int64_t n, int64_t nf, int32_t k, int32_t batch);
int creg_masked_icp_batch_f64(const creg_icp_problem* problems, int32_t batch, int64_t n, int32_t k, int64_t nf,
                              double scale, double th, int32_t max_iteration, int32_t keep_translation,
                              void* workspace, size_t workspace_bytes, creg_stream_t stream);

/* Step 1 of masked_icp on its own (cluster_icp.py:133-146): per cluster the float32 axis-aligned box of its predicted
 * world points scaled about its centre, and the indices of the frame points strictly inside it, ascending.
 * world fp32 clusters back to back with offsets world_offsets (k+1), frame (nf,3) fp64.  mask_idx (k, nf) int32: row c
 * holds mask_count[c] indices; boxes (k,6) fp32 [lo xyz | hi xyz] or NULL.  The same kernel as the large-cluster regime
 * of creg_masked_icp_f64 runs first. */
int creg_aabb_mask_f64(const float* world, const int32_t* world_offsets, int32_t k, const double* frame, int64_t nf,
                       double scale, int32_t* mask_idx, int32_t* mask_count, float* boxes, creg_stream_t stream);

/* The closed-form fit alone (SURVEY 8(b) / north star "per-cluster weighted SVD / least-squares SE(3) pose fits"): for every
 * segment [offsets[c], offsets[c+1]) of the PAIRED points src[i] <-> dst[i] ((n,3) fp64 each) the rigid T_out[c] (4x4 fp64)
 * minimising sum_i w_i |T src_i - dst_i|^2 (weights (n) fp64 >= 0, or NULL for 1) -- what open3d's
 * TransformationEstimationPointToPoint::ComputeTransformation (Eigen::umeyama without scaling, reflection fix) returns for
 * the correspondences of one ICP iteration (cluster_icp.py:157 through registration_icp).  A segment with no weight at all
 * gets the identity.  Horn's quaternion form: always a proper rotation; where the optimum is not unique (collinear or fewer
 * than three distinct pairs) it returns one of the minimisers. */
int creg_kabsch_f64(const double* src, const double* dst, const double* weights, int64_t n, const int32_t* offsets,
                    int32_t k, double* T_out, creg_stream_t stream);

/* N3  plain point-to-point ICP of k independent (source, target) cloud pairs in one launch: the
 * registration_icp(source, target, threshold, init, TransformationEstimationPointToPoint,
 * ICPConvergenceCriteria(max_iteration)) calls of link.refine_links_clusters (link.py:85-127, th = 1,
 * init = I, per link and time step) and Sim/evaluation.py:358-362 (th = 0.01).  Same kernel, iteration and
 * stopping rule as K4 without the box mask.  src (n_src,3) / tgt (n_tgt,3) fp64 segments back to back,
 * init and T_out (k,4,4) fp64, src_out = T_out applied to src.  Several time steps per launch:
 * creg_masked_icp_batch_f64 with tgt_offsets set.  workspace: creg_icp_workspace_bytes(n_src, n_tgt, k). */
int creg_icp_p2p_f64(const double* src, int64_t n_src, const int32_t* src_offsets, const double* tgt, int64_t n_tgt,
                     const int32_t* tgt_offsets, int32_t k, const double* init, double th, int32_t max_iteration,
                     double* T_out, double* src_out, int32_t* n_iter_out,
                     void* workspace, size_t workspace_bytes, creg_stream_t stream);

/* Measurement hook of the many-workgroup ICP regime (the frame of BASELINE configs[4]; bench.py's roofline of that leg): work counters
 * of the search kernel k_icp_nn, always on (wave-uniform tallies, one atomic per counter and wave), and -- while `timing` is set -- a
 * pair of HIP events around every k_icp_nn launch on the launch stream (the call then ends with a stream synchronisation).
 * out8 (HOST, 8 doubles, may be NULL): [0] waves that searched, [1] float32 screen trips (a trip = 8 staged targets x 64 lanes = 512
 * pair evaluations), [2] trips that went on to the fp64 evaluation, [3] source-iterations (live sources summed over launches),
 * [4] tie rescans, [5] summed k_icp_nn launch time in microseconds, [6] launches timed, [7] 0.  reset != 0: zero all of it afterwards.
 * timing: 1 / 0 switches the event bracketing on / off, < 0 leaves it.  Synchronises the device (hipMemcpyFromSymbol). */
int creg_icp_nn_counters(double* out8, int32_t reset, int32_t timing);

/* ------------------------------------------------------------------------------------------
 * N2  pose-sequence distance maps, fp64: the consumer of match()'s matrix/NNNN.npy files.  Replaces the
 * Python loops of CoordMap.coord_dist_map (coord_map.py:230-307; roma rotmat_to_rotvec,
 * utils.rotvec_geodesic_distance and rotmat_geodesic_distance inside) and load_matrix's
 * pose -> xyz + quaternion step (coord_map.py:204-219).
 * M (T,K,4,4) fp64 poses.  diff != 0: T-1 maps built from step-to-step motion differences and the
 * row-distance step (:250-281); diff == 0: T maps of pose distances (:283-301).
 * d_map (K,K,T') fp64 laid out like np.stack(..., axis=2); sum_map (K,K) = sum_t |d_map| (:304-305).
 * bounding_box = CoordMap.bounding_box (diagonal of the raw clouds' AABB, :153-173).
 * coords (n,7) = [x, y, z, qw, qx, qy, qz] with pytorch3d's matrix_to_quaternion in fp64. */
size_t creg_coord_dist_map_workspace_bytes(int32_t T, int32_t K);
int creg_coord_dist_map_f64(const double* M, int32_t T, int32_t K, double bounding_box, int32_t diff,
                            double* d_map, double* sum_map, void* workspace, size_t workspace_bytes,
                            creg_stream_t stream);
int creg_pose_coords_f64(const double* M, int64_t n, double* coords, creg_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * The `--normal` branch (mlp_reg.py:190-203, cluster_icp.py:49-62; CLI flag mlp_reg.py:399): Open3D normal estimation +
 * orientation, then sklearn k_means over [xyz | 0.5 * normal].
 *
 * creg_knn_normals_f64: for every point of X (n,3) fp64 its (up to) max_nn <= 32 nearest points, itself included, ascending
 * by (squared distance, index); radius > 0 keeps only squared distances < radius^2 (open3d KDTreeFlann::SearchHybrid), radius
 * <= 0 none (SearchKNN).  idx_out (n, max_nn) int32 (-1 past the count), cnt_out (n) int32, normals (n,3) fp64 -- any may be
 * NULL.  normals = PointCloud::EstimateNormals of those neighbourhoods: unit eigenvector of the smallest eigenvalue of their
 * covariance (raw-moment form), (0,0,1) for fewer than three neighbours; signs are NOT oriented (the host does that:
 * autourdf_amd/normals.py, orient_normals_consistent_tangent_plane). */
int creg_knn_normals_f64(const double* X, int64_t n, double radius, int32_t max_nn, int32_t* idx_out, int32_t* cnt_out,
                         double* normals, creg_stream_t stream);
/* sklearn.cluster.k_means(X, init=<array>, n_clusters=k, n_init=1) over DIM-dimensional features (dim = 6: the --normal
 * branch's [xyz | 0.5 n]; 3 also accepted), n <= 16384, k <= 128, one workgroup, no host sync.  X (n,dim), init (k,dim),
 * centers (k,dim) fp64; labels (n) int32; inertia (1) fp64; n_iter (1) int32; tol_rel = 1e-4 in the reference.
 * workspace: creg_kmeans_nd_workspace_bytes(n). */
size_t creg_kmeans_nd_workspace_bytes(int64_t n);
int creg_kmeans_lloyd_nd_f64(const double* X, int64_t n, int32_t dim, const double* init, int32_t k, int32_t max_iter,
                             double tol_rel, double* centers, int32_t* labels, double* inertia, int32_t* n_iter,
                             void* workspace, size_t workspace_bytes, creg_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * N4  synthetic frames from a URDF + triangle meshes (the data side: Sim/sim_data.py:246-370 renders and
 * fuses depth images of the PyBullet model; here the mesh surfaces are sampled directly).
 * tri (n_tri,3,3) fp64 triangles in their link's frame, cum_area (n_tri) inclusive prefix sum of their
 * areas, tri_link (n_tri) link index, link_T (n_links,4,4) link poses of the current joint state,
 * u (n,3) uniforms in [0,1).  out (n,3) world points, link_out (n) optional link index per point. */
int creg_sample_mesh_f64(const double* tri, const double* cum_area, const int32_t* tri_link, int32_t n_tri,
                         const double* link_T, int32_t n_links, const double* u, int64_t n, double* out,
                         int32_t* link_out, creg_stream_t stream);

/* Camera-ring visibility of sampled surface points: what the reference's rendered depth cameras impose on its frames
 * (Sim/sim_data.py:88-116 camera ring, :246-306 depth images -> point clouds).  The posed triangles are rasterised into one
 * fp64 depth buffer per camera (width x height, pinhole, vertical fov `fov_deg`, linear depth along the view axis, minimum at
 * the pixel centres); visible[i] = 1 when point i projects into some camera's image between near and far and lies no more
 * than `eps` behind that camera's buffer at its pixel.  cams (n_cams,12) fp64 = eye | forward | right | up.
 * workspace: creg_visibility_workspace_bytes(n_cams, width, height) (the depth buffers). */
size_t creg_visibility_workspace_bytes(int32_t n_cams, int32_t width, int32_t height);
int creg_visibility_f64(const double* tri, const int32_t* tri_link, int32_t n_tri, const double* link_T, int32_t n_links,
                        const double* cams, int32_t n_cams, double fov_deg, double aspect, double near_val, double far_val,
                        int32_t width, int32_t height, const double* pts, int64_t n, double eps, uint8_t* visible,
                        void* workspace, size_t workspace_bytes, creg_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * A1  the whole `train` loop (mlp_reg.py:17-152) as one device-resident plan: per epoch
 * pose -> sin/cos features -> MLP -> pose -> calculate_pc -> L1 Chamfer -> backward -> Adam ->
 * ReduceLROnPlateau, best-loss tracking and early stop, with no host round trip per epoch
 * (the reference syncs on loss.item() every epoch, mlp_reg.py:102).
 *
 * The reference's four --r choices (mlp_reg.py:64-90, models built at :276-291):
 * rot = 0: ROT == 'q'   with QRegMLP(True, hidden)  (model_utils.py:101-159)   pose row [t | quaternion], 7 -> 56 features
 * rot = 1: ROT == 'dq'  with DQRegMLP(hidden)       (model_utils.py:65-99)     dual quaternion, 8 -> 64 features, one decoder, ReLU
 * rot = 2: ROT == '6d'  with RRegMLP(hidden)        (model_utils.py:170-214)   [t | first two rows of R], 9 -> 72 features
 * rot = 3: ROT == 'rpy' with RegMLP(True, hidden)   (model_utils.py:216-281)   [t | XYZ Euler angles], 6 -> 48 features, Tanh after
 *                                                                              decoder_2 (the reference builds RegMLP(6, 3):
 *                                                                              hidden 3 -- run zero-padded at 64, see `hidden`)
 * Parameter order in `params` (torch nn.Linear layout, weight (out,in) row-major):
 *   rot 0, 2, 3: encoder.0.{weight,bias}, decoder_1.0.{w,b}, decoder_1.2.{w,b}, decoder_2.0.{w,b},
 *          decoder_2.2.{w,b}                                        (10 tensors)
 *   rot 1: encoder.0.{w,b}, decoder.0.{w,b}, decoder.2.{w,b}        (6 tensors)
 */
typedef struct creg_train_shape {
    int32_t rot;          /* 0 'q', 1 'dq', 2 '6d', 3 'rpy' */
    int32_t k;            /* clusters (poses), <= 256 (k_bd's LDS tiles: '6d' <= 227, <= 193 at hidden 512) */
    int32_t hidden;       /* hidden_dim in {64, 128, 256, 512} (512 in the reference); any other width <= 512: pass the next of these
                             and the parameters zero-padded to it -- exactly equivalent, see autourdf_amd/ops.py::TrainPlan */
    int32_t epochs;       /* 300 in the reference (mlp_reg.py:60) */
    int64_t n_pred;       /* sum of cluster sizes */
    int64_t n_tgt;        /* points in the target frame */
    int32_t use_graph;    /* 0: eager launches; 1: hipGraph of 50 epochs per launch; n > 1: n epochs per graph */
    int32_t batch;        /* independent problems (sequences) advanced per launch; 0 or 1 = one */
    int32_t graph_branches; /* how the problems of a batch share the GPU: the batch is cut into contiguous groups ("chains") whose
                               epochs run concurrently in different hardware queues, so that one group's latency-bound kernels
                               overlap another's.  0 = auto: chain streams (below), 1 chain up to 4 problems, 2 for 5-7, 3 from 8;
                               3 from 3 problems on when n_tgt > 4096.  -n: n CHAIN STREAMS -- every chain a linear graph on its own
                               stream (chain 0 on the caller's), forked / joined once per train by events; under 1.5 ms of host
                               time per train.  +n: n parallel branches inside ONE graph (rounds 2-3; the runtime enqueues every
                               node of such a graph from the host at each replay: 9-14 ms per train).  At most 3 chains are
                               useful (a fourth needs a fifth hardware queue).  Results do not depend on it. */
    int32_t nn_search;    /* 0 = auto: the nearest-neighbour searches run over k-d leaf blocks with boxes (exact
                             pruning) when n_tgt <= 65536 (four chunks of 16384 sorted per workgroup) and, for the
                             target -> predicted direction, when the predicted cloud fits 512 blocks (n_pred / 64 + k,
                             or n_pred / 256 + k above 4096 targets) with n_pred < 65535; 1 = exhaustive in both
                             directions; 2 = as 0, but always the four-queries-per-wave search of rounds 2-4.  Round 5: under 0,
                             frames of 4097..16384 target points (and a predicted cloud of at most 512 64-point blocks) take the
                             sixteen-queries-per-wave search (nn_l1.h: nn_l1_rows; franka shape 80.7 -> 104.2 frames/s).  Matches,
                             distances and the gradients of an epoch do not depend on it; the LOSS is summed per 16-slot group there
                             instead of per 32 original indices and can differ in its last bit -- and the loss feeds comparisons
                             (min_loss / best pose, ReduceLROnPlateau, early stop): a last-bit flip AT one of them changes lr or the
                             returned pose from there on, so trained parameters agree between searches only as far as no such
                             comparison is that close (none seen in the stress runs, not guaranteed).
                             creg_train_plan_info says what a plan chose (pruned_target_search is TRI-state: 0, 1, 2). */
} creg_train_shape;

typedef struct creg_train_args {
    const float* m;             /* (k,4,4) current poses */
    const float* y;             /* (n_tgt,3) target frame */
    const float* local_pts;     /* (n_pred,3) clusters in local frames, back to back */
    const int32_t* seg_offsets; /* (k+1) int32, DEVICE */
    float* const* params;       /* HOST array of 10 (rot 0, 2, 3) / 6 (rot 1) device pointers; updated in place */
    float lr;                   /* 2e-4 (Step) / 1e-4 (Anchor), mlp_reg.py:17,354 */
    float sched_factor;         /* 0.7 */
    int32_t sched_patience;     /* 5 */
    int32_t stop;               /* 200 */
    float* best_m;              /* (k,4,4) out */
    float* best_pred;           /* (n_pred,3) out: clusters at the best epoch, back to back */
    float* loss_hist;           /* (epochs) out, entries after an early stop are NaN; may be NULL */
    float* lr_hist;             /* (epochs) out (lr used by that epoch's Adam step); may be NULL */
    float* result;              /* (4) out: [min_loss, epochs_run, final_lr, best_epoch] */
    int32_t y_unchanged;        /* != 0: the caller guarantees that `y` holds the values it held in this plan's PREVIOUS run of this problem
                                   slot -- match() registers every frame twice, "Step" then "Anchor" (mlp_reg.py:338-356).  When every
                                   problem of a call says so the plan keeps the target frame's k-d leaf blocks (one bitonic sort per
                                   frame instead of one per train: 145 us at 4096 points, 673 us at 16384).  0 is always correct; since
                                   round 5 a non-zero value is VERIFIED on the device (a 64-bit position-dependent fingerprint of the staged
                                   frame against the one stored with the leaves): a frame that differs is sorted again, never searched
                                   through stale leaves. */
    int32_t reserved_;
} creg_train_args;

typedef struct creg_train_plan creg_train_plan;

/* What a plan decided from its shape (nothing here changes a result): which nearest-neighbour search each direction runs,
 * how many graph branches and problems per launch it uses.  A caller that sized its clouds beyond the pruned search's
 * limits sees it here instead of only in the timing. */
typedef struct creg_train_plan_info_t {
    int32_t pruned_target_search;     /* 1: predicted -> target search over k-d leaf blocks, four queries per wave; 2: sixteen queries per wave
                                         (both directions); 0: exhaustive (n_tgt > 65536 or nn_search = 1) */
    int32_t pruned_predicted_search;  /* 1: target -> predicted search over blocks; 0: exhaustive (more than 128 predicted blocks, ...) */
    int32_t graph_branches;           /* parallel chains in the captured graph */
    int32_t batch;                    /* problems the plan advances per run_batch call */
    int32_t epochs_per_graph;         /* 0: eager launches */
    int32_t nn_points_per_lane;       /* points per lane and block visit of the pruned search (1: 64-point blocks, 4: 256-point blocks) */
    int32_t nn_boxes_target;          /* boxes per lane of the search over the target frame (the k_nn_plan instance) */
    int32_t nn_boxes_predicted;       /* ... and over the predicted cloud */
    int32_t chain_probe_us;           /* chain streams (graph_branches <= 0, more than one chain), after the first run: the slowest joint run of two 150 us spin
                                         kernels on the caller's stream + the accepted chain streams, in us (< 250: they ran CONCURRENTLY, i.e. sit in different
                                         hardware queues); -1: no such stream was found and a chain shares a queue (its epochs then run after the other chain's);
                                         0: not probed (one chain, no run yet, or CREG_NO_QUEUE_PROBE=1) */
} creg_train_plan_info_t;

size_t creg_train_workspace_bytes(const creg_train_shape* shape);   /* covers shape.batch problems */
/* `workspace` (device, 256-byte aligned) must stay valid until creg_train_plan_destroy. */
int creg_train_plan_create(const creg_train_shape* shape, void* workspace, size_t workspace_bytes,
                           creg_train_plan** plan);
/* Enqueues the whole loop on `stream`; asynchronous (outputs are ready when the stream is). */
int creg_train_plan_run(creg_train_plan* plan, const creg_train_args* args, creg_stream_t stream);
/* Same for `n` == shape.batch independent problems of identical shape (K, N, hidden, epochs) advanced
 * together: every launch carries all of them in grid.z, so the latency-bound kernels of one sequence
 * overlap those of the others (sequences of a run are independent; frames of ONE sequence are not).
 * args[b] describes problem b; results are bit-identical to n separate creg_train_plan_run calls. */
int creg_train_plan_run_batch(creg_train_plan* plan, const creg_train_args* args, int32_t n, creg_stream_t stream);
int creg_train_plan_info(const creg_train_plan* plan, creg_train_plan_info_t* info);
int creg_train_plan_destroy(creg_train_plan* plan);

/* Round 6: a train continued from a caller-supplied state -- checkpoint / resume of `train`, and the hook the teacher-forced
 * late-epoch parity tests use (the reference's state entering epoch e, then ONE epoch of the plan).  What the reference keeps in Python
 * objects between epochs (mlp_reg.py:41-50 optimizer + scheduler, :96-119 the loop): torch.optim.Adam's exp_avg / exp_avg_sq per tensor
 * and its step count, ReduceLROnPlateau's best / num_bad_epochs / the current lr, train()'s own min_loss / count. */
typedef struct creg_train_state {
    float* const* exp_avg;      /* HOST array of device pointers: Adam's first moments, order and shapes of creg_train_args.params */
    float* const* exp_avg_sq;   /* ... second moments */
    double lr;                  /* the lr the NEXT optimizer step uses (optimizer.param_groups[0]['lr']) */
    double sched_best;          /* ReduceLROnPlateau.best (+inf before the first scheduler.step) */
    int32_t step;               /* optimizer steps taken so far */
    int32_t epochs_run;         /* epochs completed so far (= step unless stopped): the index the next loss is recorded at */
    int32_t sched_bad;          /* ReduceLROnPlateau.num_bad_epochs */
    int32_t count;              /* epochs since the loss last improved (mlp_reg.py:104-111) */
    int32_t best_epoch;         /* epoch min_loss was seen at; -1: none yet (then best_m / best_pred of the args are outputs only) */
    int32_t stopped;            /* != 0: the train had stopped early (the resumed epochs then change nothing) */
    float min_loss;             /* 1000 before the first epoch (mlp_reg.py:53) */
    int32_t reserved_;
} creg_train_state;
/* Runs `n_epochs` further epochs of problem `args` (slot 0 of the plan, eager launches) from `from`; from->step + n_epochs and
 * from->epochs_run + n_epochs must not exceed shape.epochs.  args->params: in (the parameters entering the resumed epochs) / out;
 * args->best_m / best_pred: in (when from->best_epoch >= 0) / out; loss_hist / lr_hist: entries [from->epochs_run, + n_epochs) are
 * written, the others are NaN.  exp_avg_out / exp_avg_sq_out (HOST arrays of device pointers, may be NULL): the moments after the
 * run; state_out (DEVICE, 12 doubles, may be NULL): [step, epochs_run, lr, sched_best, sched_bad, count, min_loss, best_epoch,
 * stopped, last_loss, 0, 0].  Asynchronous like creg_train_plan_run.  The next run_batch re-sorts its target frames. */
int creg_train_plan_resume(creg_train_plan* plan, const creg_train_args* args, const creg_train_state* from, int32_t n_epochs,
                           float* const* exp_avg_out, float* const* exp_avg_sq_out, double* state_out, creg_stream_t stream);

/* Test / profiling hook: run exactly one epoch's forward and return intermediates.
 * m2 (k,4,4), pred (n_pred,3), loss (1), grad_m2 (k,4,4: [dL/dR | dL/dt]); any may be NULL. */
int creg_train_plan_probe(creg_train_plan* plan, const creg_train_args* args, float* m2,
                          float* pred, float* loss, float* grad_m2, creg_stream_t stream);

/* Measurement hook: run n_epochs eagerly with a HIP event before / after each of the five kernels
 * of an epoch (order: head, nn_l1, gradc, bd, l2) on `stream`, synchronise, and write the
 * average microseconds per kernel to us_out (HOST, 16 floats): [0] 0, [1..5] event-bracketed per kernel in that order
 * (they include the dispatch latency of the launch, i.e. they are an upper bound of the kernel time),
 * [6] the nearest-neighbour kernel alone as 200 back-to-back launches between two events (kernel + launch gap),
 * [7] the problems that launch carries, [8] the backward launch k_bd the same way, [9] k_l2, [10] head, [11] gradc,
 * [12..15] reserved. */
int creg_train_plan_profile(creg_train_plan* plan, const creg_train_args* args, int32_t n_epochs,
                            float* us_out, creg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CREG_H */
