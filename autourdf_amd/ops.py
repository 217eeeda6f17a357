"""torch-facing wrappers over the C ABI: tensors in, tensors out, current CUDA(HIP) stream.

Only plumbing lives here (pointer extraction, output allocation, autograd glue); all arithmetic is
in libcreg.so.  Every function requires CUDA tensors and raises otherwise -- no CPU path exists.
"""
import ctypes

import numpy as np

import torch

from . import _lib


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need(t: torch.Tensor, dtype, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor on the MI355X (autourdf_amd has no CPU path)")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    return t.contiguous()


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


# ------------------------------------------------------------------------------ K1 nearest neighbour
def nn_l1_bidir(x: torch.Tensor, y: torch.Tensor):
    """x (nx,3), y (ny,3) fp32 -> (dx, ix, dy, iy): L1 nearest neighbour both ways, first min wins."""
    L = _lib.load()
    x, y = _need(x, torch.float32, "x"), _need(y, torch.float32, "y")
    nx, ny = x.shape[0], y.shape[0]
    if nx == 0 or ny == 0:
        raise ValueError("nn_l1_bidir: empty point cloud")      # pytorch3d rejects it as well
    dx = torch.empty(nx, dtype=torch.float32, device=x.device)
    dy = torch.empty(ny, dtype=torch.float32, device=x.device)
    ix = torch.empty(nx, dtype=torch.int64, device=x.device)
    iy = torch.empty(ny, dtype=torch.int64, device=x.device)
    _lib.check(L.creg_nn_l1_bidir_f32(_p(x), nx, _p(y), ny, _p(dx), _p(ix), _p(dy), _p(iy), _stream()),
               "creg_nn_l1_bidir_f32")
    return dx, ix, dy, iy


class _ChamferL1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y):
        L = _lib.load()
        dx, ix, dy, iy = nn_l1_bidir(x, y)
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        _lib.check(L.creg_chamfer_l1_reduce_f32(_p(dx), dx.numel(), _p(dy), dy.numel(), _p(loss), _stream()),
                   "creg_chamfer_l1_reduce_f32")
        ctx.save_for_backward(x.detach().contiguous(), y.detach().contiguous(), ix, iy)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        L = _lib.load()
        x, y, ix, iy = ctx.saved_tensors
        nx, ny = x.shape[0], y.shape[0]
        grad = torch.empty_like(x)
        scratch = torch.empty(L.creg_nn_l1_bwd_scratch_bytes(nx), dtype=torch.uint8, device=x.device)
        one = torch.tensor(1.0, dtype=torch.float32)
        gx, gy = float(one / nx), float(one / ny)
        _lib.check(L.creg_nn_l1_bwd_f32(_p(x), nx, _p(y), ny, _p(ix), _p(iy), gx, gy, _p(grad), _p(scratch),
                                        _stream()), "creg_nn_l1_bwd_f32")
        return grad * g, None


def chamfer_distance(x: torch.Tensor, y: torch.Tensor, norm: int = 1):
    """pytorch3d.loss.chamfer_distance(x, y, norm=1) for the reference's call (mlp_reg.py:96):
    x (1,P1,3) with grad, y (1,P2,3); returns (loss, None)."""
    if norm != 1:
        raise NotImplementedError("only norm=1 is on the registration path")
    if x.dim() != 3 or x.shape[0] != 1 or y.shape[0] != 1:
        raise NotImplementedError("batch of 1 only (as the reference calls it)")
    return _ChamferL1.apply(_need(x[0], torch.float32, "x"), _need(y[0], torch.float32, "y")), None


# ------------------------------------------------------------------------------ K3 cluster transform
class _ClusterTransform(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pts, offsets, M):
        L = _lib.load()
        out = torch.empty_like(pts)
        k = M.shape[0]
        _lib.check(L.creg_cluster_transform_f32(_p(pts), pts.shape[0], _p(offsets), k, _p(M), _p(out), _stream()),
                   "creg_cluster_transform_f32")
        ctx.save_for_backward(pts, offsets)
        ctx.k = k
        return out

    @staticmethod
    def backward(ctx, g):
        L = _lib.load()
        pts, offsets = ctx.saved_tensors
        gM = torch.empty(ctx.k, 4, 4, dtype=torch.float32, device=pts.device)
        _lib.check(L.creg_cluster_transform_bwd_f32(_p(pts), _p(offsets), ctx.k, _p(g.contiguous()), _p(gM), _stream()),
                   "creg_cluster_transform_bwd_f32")
        return None, None, gM


def cluster_transform(pts: torch.Tensor, offsets: torch.Tensor, M: torch.Tensor) -> torch.Tensor:
    """pts (n,3) fp32 clusters back to back, offsets (k+1) int32, M (k,4,4) fp32 -> world points."""
    return _ClusterTransform.apply(_need(pts, torch.float32, "pts"), _need(offsets, torch.int32, "offsets"),
                                   _need(M, torch.float32, "M"))


def pack_clusters(clusters, device, dtype=torch.float32):
    """list of (M_k,3) tensors/arrays -> (flat (n,3) tensor, offsets (k+1) int32 tensor), on `device`."""
    ts = [torch.as_tensor(c, dtype=dtype).reshape(-1, 3) for c in clusters]
    sizes = [t.shape[0] for t in ts]
    off = torch.zeros(len(ts) + 1, dtype=torch.int32)
    off[1:] = torch.cumsum(torch.tensor(sizes, dtype=torch.int64), 0).to(torch.int32)
    flat = torch.cat([t.to(device) for t in ts], 0) if ts else torch.zeros(0, 3, dtype=dtype, device=device)
    return flat.contiguous(), off.to(device)


# ------------------------------------------------------------------------------ K2 k-means
def kmeans_lloyd(X: torch.Tensor, init: torch.Tensor, max_iter: int = 300, tol: float = 1e-4,
                 use_mfma: bool = False):
    """sklearn.cluster.k_means(X, init=init, n_init=1): returns (centers (k,3) f64, labels (n) int32,
    inertia (1) f64, n_iter (1) int32), all on the device."""
    L = _lib.load()
    X, init = _need(X, torch.float64, "X"), _need(init, torch.float64, "init")
    n, k = X.shape[0], init.shape[0]
    ws_bytes = L.creg_kmeans_workspace_bytes(n, k)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=X.device)
    centers = torch.empty(k, 3, dtype=torch.float64, device=X.device)
    labels = torch.empty(n, dtype=torch.int32, device=X.device)
    inertia = torch.empty(1, dtype=torch.float64, device=X.device)
    n_iter = torch.empty(1, dtype=torch.int32, device=X.device)
    _lib.check(L.creg_kmeans_lloyd_f64(_p(X), n, _p(init), k, max_iter, tol, int(use_mfma), _p(centers), _p(labels),
                                       _p(inertia), _p(n_iter), _p(ws), ws_bytes, _stream()), "creg_kmeans_lloyd_f64")
    return centers, labels, inertia, n_iter


def kmeans_lloyd_nd(X: torch.Tensor, init: torch.Tensor, max_iter: int = 300, tol: float = 1e-4):
    """sklearn.cluster.k_means(X, init=init, n_init=1) over (n,6) features -- the `--normal` branch clusters [xyz | 0.5 normal]
    (mlp_reg.py:196-203) -- in one workgroup (n <= 16384, k <= 128): (centers (k,d) f64, labels (n) int32, inertia (1) f64,
    n_iter (1) int32), all on the device."""
    L = _lib.load()
    X, init = _need(X, torch.float64, "X"), _need(init, torch.float64, "init")
    n, d, k = X.shape[0], X.shape[1], init.shape[0]
    if init.shape[1] != d:
        raise ValueError("kmeans_lloyd_nd: X and init disagree on the feature count")
    if n > KMEANS_ND_MAX_N or k > KMEANS_ND_MAX_K:
        raise ValueError(f"the --normal re-segmentation (k-means over [xyz | 0.5 normal]) runs in one workgroup: frames of at most "
                         f"{KMEANS_ND_MAX_N} points and {KMEANS_ND_MAX_K} clusters, got {n} points / {k} clusters (the reference's sklearn "
                         "k_means has no such limit: down-sample the frames -- Segments(sample_size=...) -- or run without --normal)")
    ws_bytes = L.creg_kmeans_nd_workspace_bytes(n)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=X.device)
    centers = torch.empty(k, d, dtype=torch.float64, device=X.device)
    labels = torch.empty(n, dtype=torch.int32, device=X.device)
    inertia = torch.empty(1, dtype=torch.float64, device=X.device)
    n_iter = torch.empty(1, dtype=torch.int32, device=X.device)
    _lib.check(L.creg_kmeans_lloyd_nd_f64(_p(X), n, d, _p(init), k, max_iter, tol, _p(centers), _p(labels), _p(inertia), _p(n_iter),
                                          _p(ws), ws_bytes, _stream()), "creg_kmeans_lloyd_nd_f64")
    return centers, labels, inertia, n_iter


def knn_normals(X: torch.Tensor, radius: float, max_nn: int, want_normals: bool = True, want_idx: bool = False):
    """creg_knn_normals_f64: the (up to) max_nn nearest points of every point of X (n,3) f64 (itself included; radius > 0:
    only squared distances < radius^2, open3d's SearchHybrid; radius <= 0: plain SearchKNN) and, from them, open3d's
    estimate_normals (unoriented).  Returns (normals (n,3) f64 or None, idx (n,max_nn) int32 or None, counts (n) int32)."""
    L = _lib.load()
    X = _need(X, torch.float64, "X")
    n = X.shape[0]
    normals = torch.empty(n, 3, dtype=torch.float64, device=X.device) if want_normals else None
    idx = torch.empty(n, max_nn, dtype=torch.int32, device=X.device) if want_idx else None
    cnt = torch.empty(n, dtype=torch.int32, device=X.device)
    _lib.check(L.creg_knn_normals_f64(_p(X), n, float(radius), int(max_nn), _p(idx) if want_idx else None, _p(cnt),
                                      _p(normals) if want_normals else None, _stream()), "creg_knn_normals_f64")
    return normals, idx, cnt


KMEANS_ND_MAX_N, KMEANS_ND_MAX_K = (1 << 24) - 1, 128      # creg_kmeans_lloyd_nd_f64: one workgroup, centres in LDS (labels too up to 16384 points, in the workspace above)
KMEANS_BATCH_MAX_N = 16384         # labels + centres in one CU's LDS; the frame too up to 5120 points, from L2 above


def kmeans_lloyd_batch(Xs, inits, max_iter: int = 300, tol: float = 1e-4):
    """k_means() for a list of same-sized frames in ONE asynchronous launch (one workgroup per frame,
    n <= 16384).  Returns a list of (centers, labels, inertia, n_iter) like `kmeans_lloyd`."""
    L = _lib.load()
    Xs = [_need(x, torch.float64, "X") for x in Xs]
    inits = [_need(c, torch.float64, "init") for c in inits]
    B, n, k = len(Xs), Xs[0].shape[0], inits[0].shape[0]
    if any(x.shape[0] != n for x in Xs) or any(c.shape[0] != k for c in inits) or len(inits) != B:
        raise ValueError("kmeans_lloyd_batch: all problems must have the same n and k")
    dev = Xs[0].device
    ws_bytes = L.creg_kmeans_batch_workspace_bytes(n, k, B)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    outs = [(torch.empty(k, 3, dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.int32, device=dev),
             torch.empty(1, dtype=torch.float64, device=dev), torch.empty(1, dtype=torch.int32, device=dev)) for _ in range(B)]
    arr = lambda ts: (ctypes.c_void_p * B)(*[t.data_ptr() for t in ts])
    _lib.check(L.creg_kmeans_lloyd_batch_f64(arr(Xs), n, arr(inits), k, B, max_iter, tol, arr([o[0] for o in outs]),
                                             arr([o[1] for o in outs]), arr([o[2] for o in outs]), arr([o[3] for o in outs]),
                                             _p(ws), ws_bytes, _stream()), "creg_kmeans_lloyd_batch_f64")
    return outs


def kmeans_assign(X: torch.Tensor, C: torch.Tensor, use_mfma: bool = False) -> torch.Tensor:
    L = _lib.load()
    X, C = _need(X, torch.float64, "X"), _need(C, torch.float64, "C")
    labels = torch.empty(X.shape[0], dtype=torch.int32, device=X.device)
    _lib.check(L.creg_kmeans_assign_f64(_p(X), X.shape[0], _p(C), C.shape[0], int(use_mfma), _p(labels), _stream()),
               "creg_kmeans_assign_f64")
    return labels


def group_to_local(X: torch.Tensor, labels: torch.Tensor, M: torch.Tensor, m_is_inverse: bool = False):
    """Stable grouping by label and inv(M_k) change of frame: returns (local (n,3) f64, offsets (k+1) int32).
    m_is_inverse: M already holds the inverted poses (the drop-in inverts them on the host like mlp_reg.py:211)."""
    L = _lib.load()
    X, labels, M = _need(X, torch.float64, "X"), _need(labels, torch.int32, "labels"), _need(M, torch.float64, "M")
    k = M.shape[0]
    out = torch.empty_like(X)
    off = torch.empty(k + 1, dtype=torch.int32, device=X.device)
    _lib.check(L.creg_group_to_local_f64(_p(X), X.shape[0], _p(labels), k, _p(M), int(bool(m_is_inverse)), _p(out), _p(off), _stream()),
               "creg_group_to_local_f64")
    return out, off


GROUP_BATCH_MAX = 16


def group_to_local_batch(Xs, labels, Ms, m_is_inverse: bool = False):
    """`group_to_local` for a list (<= 16) of frames of identical n and k in one pair of launches.
    Returns a list of (local (n,3) f64, offsets (k+1) int32)."""
    L = _lib.load()
    B = len(Xs)
    if not 1 <= B <= GROUP_BATCH_MAX or len(labels) != B or len(Ms) != B:
        raise ValueError(f"group_to_local_batch: 1..{GROUP_BATCH_MAX} problems, one label / pose tensor each")
    Xs = [_need(x, torch.float64, "X") for x in Xs]
    labels = [_need(l, torch.int32, "labels") for l in labels]
    Ms = [_need(m, torch.float64, "M") for m in Ms]
    n, k = Xs[0].shape[0], Ms[0].shape[0]
    if any(x.shape[0] != n for x in Xs) or any(m.shape[0] != k for m in Ms) or any(l.shape[0] != n for l in labels):
        raise ValueError("group_to_local_batch: all problems must share n and k")
    outs = [(torch.empty_like(x), torch.empty(k + 1, dtype=torch.int32, device=x.device)) for x in Xs]
    arr = lambda ts: (ctypes.c_void_p * B)(*[t.data_ptr() for t in ts])
    _lib.check(L.creg_group_to_local_batch_f64(arr(Xs), n, arr(labels), k, arr(Ms), int(bool(m_is_inverse)), B, arr([o[0] for o in outs]),
                                               arr([o[1] for o in outs]), _stream()), "creg_group_to_local_batch_f64")
    return outs


def icp_p2p_batch(problems, th: float = 1.0, max_iteration: int = 100000):
    """N3: plain point-to-point ICP (open3d registration_icp semantics) of packed cloud pairs.
    problems: list (<= 16) of (src (n,3) f64, src_offsets (k+1) i32, tgt (m,3) f64, tgt_offsets (k+1) i32,
    init (k,4,4) f64) with identical n, m, k -- e.g. the time steps of link.refine_links_clusters
    (link.py:85-127), each holding k links.  One launch (grid links x problems).
    Returns a list of (T (k,4,4) f64, moved source (n,3) f64, iterations (k) i32)."""
    L = _lib.load()
    B = len(problems)
    if not 1 <= B <= ICP_BATCH_MAX:
        raise ValueError(f"icp_p2p_batch: 1..{ICP_BATCH_MAX} problems per launch, got {B}")
    arr = (_lib.IcpProblem * B)()
    keep, outs = [], []
    for b, (src, soff, tgt, toff, init) in enumerate(problems):
        src, tgt, init = _need(src, torch.float64, "src"), _need(tgt, torch.float64, "tgt"), _need(init, torch.float64, "init")
        soff, toff = _need(soff, torch.int32, "src_offsets"), _need(toff, torch.int32, "tgt_offsets")
        shape = (src.shape[0], tgt.shape[0], soff.shape[0] - 1)
        if b == 0:
            n, m, k = shape
        if shape != (n, m, k) or toff.shape[0] != k + 1 or init.shape[0] != k:
            raise ValueError("icp_p2p_batch: all problems must share n_src, n_tgt and k")
        dev = src.device
        T = torch.empty(k, 4, 4, dtype=torch.float64, device=dev)
        moved = torch.empty(n, 3, dtype=torch.float64, device=dev)
        n_it = torch.empty(k, dtype=torch.int32, device=dev)
        arr[b] = _lib.IcpProblem(_p(src), None, _p(soff), _p(tgt), _p(init), _p(T), _p(moved), _p(n_it), _p(toff), None)
        keep.append((src, soff, tgt, toff, init))
        outs.append((T, moved, n_it))
    ws_bytes = L.creg_icp_batch_workspace_bytes(n, m, k, B)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    _lib.check(L.creg_masked_icp_batch_f64(arr, B, n, k, m, 1.0, float(th), int(max_iteration), 0, _p(ws), ws_bytes,
                                           _stream()), "creg_masked_icp_batch_f64")
    return outs


def kabsch(src, dst, offsets, weights=None):
    """Closed-form (weighted) rigid fit of paired points per segment (creg_kabsch_f64): src, dst (n,3) f64, offsets (k+1)
    i32, weights (n) f64 or None -> T (k,4,4) f64 with T src ~ dst in the least-squares sense (open3d point-to-point
    estimation / Umeyama without scaling); identity for a segment without pairs."""
    L = _lib.load()
    src, dst = _need(src, torch.float64, "src"), _need(dst, torch.float64, "dst")
    offsets = _need(offsets, torch.int32, "offsets")
    if src.shape != dst.shape or src.dim() != 2 or src.shape[1] != 3:
        raise ValueError("kabsch: src and dst must both be (n,3)")
    if weights is not None:
        weights = _need(weights, torch.float64, "weights")
        if weights.shape[0] != src.shape[0]:
            raise ValueError("kabsch: one weight per pair")
    k = offsets.shape[0] - 1
    T = torch.empty(k, 4, 4, dtype=torch.float64, device=src.device)
    _lib.check(L.creg_kabsch_f64(_p(src), _p(dst), _p(weights) if weights is not None else None, src.shape[0], _p(offsets), k, _p(T),
                                 _stream()), "creg_kabsch_f64")
    return T


def icp_p2p(src, src_offsets, tgt, tgt_offsets, init, th: float = 1.0, max_iteration: int = 100000):
    """`icp_p2p_batch` for one problem, through creg_icp_p2p_f64."""
    L = _lib.load()
    src, tgt, init = _need(src, torch.float64, "src"), _need(tgt, torch.float64, "tgt"), _need(init, torch.float64, "init")
    soff, toff = _need(src_offsets, torch.int32, "src_offsets"), _need(tgt_offsets, torch.int32, "tgt_offsets")
    n, m, k = src.shape[0], tgt.shape[0], soff.shape[0] - 1
    if toff.shape[0] != k + 1 or init.shape[0] != k:
        raise ValueError("icp_p2p: offsets / init disagree on the number of pairs")
    T = torch.empty(k, 4, 4, dtype=torch.float64, device=src.device)
    moved = torch.empty(n, 3, dtype=torch.float64, device=src.device)
    n_it = torch.empty(k, dtype=torch.int32, device=src.device)
    ws_bytes = L.creg_icp_workspace_bytes(n, m, k)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=src.device)
    _lib.check(L.creg_icp_p2p_f64(_p(src), n, _p(soff), _p(tgt), m, _p(toff), k, _p(init), float(th), int(max_iteration),
                                  _p(T), _p(moved), _p(n_it), _p(ws), ws_bytes, _stream()), "creg_icp_p2p_f64")
    return T, moved, n_it


# ------------------------------------------------------------------------------ N4 mesh surface sampling
def sample_mesh(tri, cum_area, tri_link, link_T, u, with_links: bool = False):
    """Area-weighted points on an articulated triangle mesh (creg_sample_mesh_f64): tri (F,3,3) f64 in link
    frames, cum_area (F) f64, tri_link (F) i32, link_T (L,4,4) f64, u (n,3) f64 uniforms -> (n,3) f64 world points
    (and the link index of every point when with_links)."""
    L = _lib.load()
    tri, cum_area, link_T, u = (_need(t, torch.float64, nm) for t, nm in ((tri, "tri"), (cum_area, "cum_area"), (link_T, "link_T"), (u, "u")))
    tri_link = _need(tri_link, torch.int32, "tri_link")
    n, F = u.shape[0], tri.shape[0]
    if cum_area.shape[0] != F or tri_link.shape[0] != F:
        raise ValueError("sample_mesh: tri / cum_area / tri_link disagree on the number of triangles")
    out = torch.empty(n, 3, dtype=torch.float64, device=u.device)
    links = torch.empty(n, dtype=torch.int32, device=u.device) if with_links else None
    _lib.check(L.creg_sample_mesh_f64(_p(tri), _p(cum_area), _p(tri_link), F, _p(link_T), link_T.shape[0], _p(u), n, _p(out),
                                      _p(links) if with_links else None, _stream()), "creg_sample_mesh_f64")
    return (out, links) if with_links else out


def visibility(tri, tri_link, link_T, cams, pts, fov_deg=60.0, aspect=1.0, near=0.1, far=4.0, width=800, height=800,
               eps=0.004, return_depth=False):
    """Camera-ring visibility (creg_visibility_f64): tri (F,3,3) f64 link-frame triangles, tri_link (F) i32, link_T (L,4,4)
    f64, cams (C,12) f64 = eye | forward | right | up, pts (n,3) f64 world points -> (n) bool tensor (and the (C,H,W) f64
    depth buffers when return_depth)."""
    L = _lib.load()
    tri, link_T, cams, pts = (_need(t, torch.float64, nm) for t, nm in ((tri, "tri"), (link_T, "link_T"), (cams, "cams"), (pts, "pts")))
    tri_link = _need(tri_link, torch.int32, "tri_link")
    C, n = cams.shape[0], pts.shape[0]
    ws_bytes = L.creg_visibility_workspace_bytes(C, width, height)
    ws = torch.empty(ws_bytes // 8, dtype=torch.float64, device=pts.device)
    vis = torch.empty(n, dtype=torch.uint8, device=pts.device)
    _lib.check(L.creg_visibility_f64(_p(tri), _p(tri_link), tri.shape[0], _p(link_T), link_T.shape[0], _p(cams), C, float(fov_deg),
                                     float(aspect), float(near), float(far), int(width), int(height), _p(pts), n, float(eps), _p(vis),
                                     _p(ws), ws_bytes, _stream()), "creg_visibility_f64")
    return (vis.bool(), ws.reshape(C, height, width)) if return_depth else vis.bool()


# ------------------------------------------------------------------------------ N2 pose distance maps
def coord_dist_map(M: torch.Tensor, bounding_box: float, diff: bool = True):
    """CoordMap.coord_dist_map (coord_map.py:230-307) for poses M (T,K,4,4) f64 on the device:
    returns (coord_dist_map (K,K,T') f64, sum_map (K,K) f64), T' = T-1 if diff else T."""
    L = _lib.load()
    M = _need(M, torch.float64, "M")
    if M.dim() != 4 or M.shape[2:] != (4, 4):
        raise ValueError("coord_dist_map: M must be (T,K,4,4)")
    T, K = M.shape[:2]
    Tn = T - 1 if diff else T
    d_map = torch.empty(K, K, Tn, dtype=torch.float64, device=M.device)
    s_map = torch.empty(K, K, dtype=torch.float64, device=M.device)
    ws_bytes = L.creg_coord_dist_map_workspace_bytes(T, K)
    ws = torch.empty(max(ws_bytes, 256), dtype=torch.uint8, device=M.device)
    _lib.check(L.creg_coord_dist_map_f64(_p(M), T, K, float(bounding_box), int(bool(diff)), _p(d_map), _p(s_map), _p(ws),
                                         ws.numel(), _stream()), "creg_coord_dist_map_f64")
    return d_map, s_map


def pose_coords(M: torch.Tensor) -> torch.Tensor:
    """(...,4,4) f64 poses -> (...,7) [xyz, pytorch3d quaternion wxyz] (load_matrix, coord_map.py:204-219)."""
    L = _lib.load()
    M = _need(M, torch.float64, "M")
    n = M.numel() // 16
    out = torch.empty(M.shape[:-2] + (7,), dtype=torch.float64, device=M.device)
    _lib.check(L.creg_pose_coords_f64(_p(M), n, _p(out), _stream()), "creg_pose_coords_f64")
    return out


# ------------------------------------------------------------------------------ K5 row conversions
def masked_icp(local: torch.Tensor, world: torch.Tensor, offsets: torch.Tensor, frame: torch.Tensor, M: torch.Tensor,
               scale: float = 1.2, th: float = 1.0, max_iteration: int = 10000, ori: bool = False,
               world_offsets: torch.Tensor = None):
    """K4, device resident: AABB-masked point-to-point ICP of every cluster against `frame`, one launch
    (reference cluster_icp.py:118-191 + open3d registration_icp).  local (n,3) f64 cluster-frame
    points (the ICP sources), offsets (k+1) int32; world f32 predicted clusters back to back (the mask boxes) with
    segment offsets `world_offsets` (None: the same segmentation as `local`; match()'s --mlp_icp branch passes the
    frame-0 clusters as `local` and the trained clouds of the current segmentation as `world`, mlp_reg.py:325),
    frame (nf,3) f64, M (k,4,4) f64 initial poses.  Returns (M_out (k,4,4) f64, world_out (n,3) f64,
    iterations (k) int32).  Stream-ordered with no host sync in the one-launch regime (clusters of at most 1024 points on
    average and frames of at most 65536 points: `masked_icp_regime(n, nf, k) == "one_launch"`); in the many-workgroup regime
    (configs[4]-sized clusters) the call synchronises the stream every 16 ICP iterations to read two words (creg.h) -- it
    blocks the host and cannot be captured in a graph."""
    L = _lib.load()
    local, world = _need(local, torch.float64, "local"), _need(world, torch.float32, "world")
    frame, M = _need(frame, torch.float64, "frame"), _need(M, torch.float64, "M")
    offsets = _need(offsets, torch.int32, "offsets")
    n, nf, k = local.shape[0], frame.shape[0], offsets.shape[0] - 1
    if world_offsets is not None:
        world_offsets = _need(world_offsets, torch.int32, "world_offsets")
        if world_offsets.shape[0] != k + 1:
            raise ValueError("world_offsets must hold k+1 entries")
    elif world.shape[0] != n:
        raise ValueError("world must have local's size unless world_offsets is given")
    if M.shape[0] != k:
        raise ValueError("local/world/offsets/M disagree on sizes")
    ws_bytes = L.creg_icp_workspace_bytes(n, nf, k)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=local.device)
    M_out = torch.empty(k, 4, 4, dtype=torch.float64, device=local.device)
    w_out = torch.empty(n, 3, dtype=torch.float64, device=local.device)
    n_it = torch.empty(k, dtype=torch.int32, device=local.device)
    _lib.check(L.creg_masked_icp_f64(_p(local), _p(world), _p(world_offsets), n, _p(offsets), k, _p(frame), nf, _p(M), float(scale),
                                     float(th), int(max_iteration), int(bool(ori)), _p(M_out), _p(w_out), _p(n_it),
                                     _p(ws), ws_bytes, _stream()), "creg_masked_icp_f64")
    return M_out, w_out, n_it


ICP_BATCH_MAX = 16


def aabb_mask(world: torch.Tensor, world_offsets: torch.Tensor, frame: torch.Tensor, scale: float = 1.2):
    """Step 1 of masked_icp (cluster_icp.py:133-146): per cluster the scaled float32 box of its world points and the
    frame points strictly inside it.  Returns (mask_idx (k, nf) int32 -- row c holds count[c] ascending indices --,
    count (k) int32, boxes (k,6) fp32)."""
    L = _lib.load()
    world, frame = _need(world, torch.float32, "world"), _need(frame, torch.float64, "frame")
    world_offsets = _need(world_offsets, torch.int32, "world_offsets")
    k, nf = world_offsets.shape[0] - 1, frame.shape[0]
    idx = torch.empty(k, nf, dtype=torch.int32, device=frame.device)
    cnt = torch.empty(k, dtype=torch.int32, device=frame.device)
    boxes = torch.empty(k, 6, dtype=torch.float32, device=frame.device)
    _lib.check(L.creg_aabb_mask_f64(_p(world), _p(world_offsets), k, _p(frame), nf, float(scale), _p(idx), _p(cnt), _p(boxes),
                                    _stream()), "creg_aabb_mask_f64")
    return idx, cnt, boxes


def masked_icp_regime(n: int, nf: int, k: int) -> str:
    """Which form creg_masked_icp_f64 runs for n source points in k clusters against a frame of nf points: "one_launch"
    (everything in one launch, no host sync) or "many_workgroups" (one launch per ICP iteration, a stream synchronisation every
    16 iterations).  Mirrors the choice in csrc/icp.hip (ICP_SRC_LDS = 1024 source points per cluster on average, 65536 frame points)."""
    return "many_workgroups" if n // max(k, 1) > 1024 or nf > 65536 else "one_launch"


def masked_icp_batch(problems, scale: float = 1.2, th: float = 1.0, max_iteration: int = 10000, ori: bool = False):
    """`masked_icp` for a list of (local, world, offsets, frame, M[, world_offsets]) of identical sizes in ONE launch
    (grid clusters x problems); returns a list of (M_out, world_out, iterations), bit-identical to
    separate calls.  `world` None = the clusters in their current pose (`cluster_transform` of the float32 casts),
    evaluated inside the kernel; `world_offsets` as in `masked_icp`."""
    L = _lib.load()
    B = len(problems)
    if not 1 <= B <= ICP_BATCH_MAX:
        raise ValueError(f"masked_icp_batch: 1..{ICP_BATCH_MAX} problems per launch, got {B}")
    arr = (_lib.IcpProblem * B)()
    keep, outs = [], []
    n = nf = k = None
    for b, prob in enumerate(problems):
        local, world, offsets, frame, M = prob[:5]
        woff = prob[5] if len(prob) > 5 else None
        local = _need(local, torch.float64, "local")
        world = None if world is None else _need(world, torch.float32, "world")     # None: boxes of float32(M) . float32(local)
        frame, M = _need(frame, torch.float64, "frame"), _need(M, torch.float64, "M")
        offsets = _need(offsets, torch.int32, "offsets")
        shape = (local.shape[0], frame.shape[0], offsets.shape[0] - 1)
        if b == 0:
            n, nf, k = shape
        if woff is not None:
            woff = _need(woff, torch.int32, "world_offsets")
            if world is None or woff.shape[0] != k + 1:
                raise ValueError("masked_icp_batch: world_offsets needs world and k+1 entries")
        elif world is not None and world.shape[0] != n:
            raise ValueError("masked_icp_batch: world must have local's size unless world_offsets is given")
        if shape != (n, nf, k) or M.shape[0] != k:
            raise ValueError("masked_icp_batch: all problems must share n, nf and k")
        dev = local.device
        M_out = torch.empty(k, 4, 4, dtype=torch.float64, device=dev)
        w_out = torch.empty(n, 3, dtype=torch.float64, device=dev)
        n_it = torch.empty(k, dtype=torch.int32, device=dev)
        arr[b] = _lib.IcpProblem(_p(local), _p(world), _p(offsets), _p(frame), _p(M), _p(M_out), _p(w_out), _p(n_it), None,
                                     _p(woff))
        keep.append((local, world, offsets, frame, M, woff))
        outs.append((M_out, w_out, n_it))
    ws_bytes = L.creg_icp_batch_workspace_bytes(n, nf, k, B)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    _lib.check(L.creg_masked_icp_batch_f64(arr, B, n, k, nf, float(scale), float(th), int(max_iteration),
                                           int(bool(ori)), _p(ws), ws_bytes, _stream()), "creg_masked_icp_batch_f64")
    return outs


def _rows(fn_name, a, out_shape, b=None, out2_shape=None):
    L = _lib.load()
    a = _need(a, torch.float32, "input")
    k = a.shape[0]
    out = torch.empty((k,) + out_shape, dtype=torch.float32, device=a.device)
    fn = getattr(L, fn_name)
    if out2_shape is not None:
        out2 = torch.empty((k,) + out2_shape, dtype=torch.float32, device=a.device)
        _lib.check(fn(_p(a), k, _p(out), _p(out2), _stream()), fn_name)
        return out, out2
    if b is not None:
        b = _need(b, torch.float32, "input")
        _lib.check(fn(_p(a), _p(b), k, _p(out), _stream()), fn_name)
    else:
        _lib.check(fn(_p(a), k, _p(out), _stream()), fn_name)
    return out


def se3_to_dq(M): return _rows("creg_se3_to_dq_f32", M, (8,))
def dq_to_se3(dq): return _rows("creg_dq_to_se3_f32", dq, (4, 4))
def dq_to_se3_bwd(dq, gM): return _rows("creg_dq_to_se3_bwd_f32", dq, (8,), b=gM)
def dq_multiply(a, b): return _rows("creg_dq_multiply_f32", a, (8,), b=b)
def dq_invert(dq): return _rows("creg_dq_invert_f32", dq, (8,))
def dq_to_quat_trans(dq): return _rows("creg_dq_to_quat_trans_f32", dq, (4,), out2_shape=(3,))
def quat_trans_to_dq(q, t): return _rows("creg_quat_trans_to_dq_f32", q, (8,), b=t)
def matrix_to_quat(R): return _rows("creg_matrix_to_quat_f32", R, (4,))
def quat_to_matrix(q): return _rows("creg_quat_to_matrix_f32", q, (3, 3))


# ------------------------------------------------------------------------------ A1 train plan
Q_PARAM_ORDER = ["encoder.0.weight", "encoder.0.bias", "decoder_1.0.weight", "decoder_1.0.bias",
                 "decoder_1.2.weight", "decoder_1.2.bias", "decoder_2.0.weight", "decoder_2.0.bias",
                 "decoder_2.2.weight", "decoder_2.2.bias"]
DQ_PARAM_ORDER = ["encoder.0.weight", "encoder.0.bias", "decoder.0.weight", "decoder.0.bias",
                  "decoder.2.weight", "decoder.2.bias"]
#: creg_train_shape.rot: the reference's four --r choices (mlp_reg.py:64-90).  RRegMLP ('6d') and RegMLP ('rpy') name their
#: tensors like QRegMLP (model_utils.py:170-281): Q_PARAM_ORDER serves the three.
TRAIN_ROT = {"q": 0, "dq": 1, "6d": 2, "rpy": 3}
#: per rot: (input features of the encoder = 8 x the pose row's width, outputs of decoder_2)
_TWO_DECODER = {0: (56, 4), 2: (72, 6), 3: (48, 3)}


TRAIN_HIDDEN_TILES = (64, 128, 256, 512)      # widths the train kernels are instantiated for


def icp_nn_counters(reset: bool = False, timing=None) -> dict:
    """Work counters / launch timing of the many-workgroup ICP search (creg_icp_nn_counters; a measurement hook, synchronises)."""
    out = (ctypes.c_double * 8)()
    L = _lib.load()
    _lib.check(L.creg_icp_nn_counters(out, 1 if reset else 0, -1 if timing is None else int(bool(timing))), "creg_icp_nn_counters")
    keys = ("waves", "f32_trips", "f64_trips", "source_iterations", "tie_rescans", "nn_launch_us_total", "nn_launches_timed")
    return dict(zip(keys, [float(v) for v in out]))


class TrainPlan:
    """Device-resident `train` loop (mlp_reg.py:17-152) for one (rot, K, hidden, N) shape.

    `hidden` may be any width up to 512, like the reference's models (model_utils.py:65-168): a width the kernels are not
    instantiated for runs on the next one that is, with the extra units' weights and biases zero.  Such a unit's activation is
    exactly 0 (LeakyReLU / ReLU of 0), it adds exactly 0 to every sum it enters, and every gradient of its parameters is a
    product with that 0 or with the zero column that leads out of it -- so Adam leaves them at 0 and the trained model, losses
    and poses are those of the unpadded model; the caller's tensors keep their own shapes.

    `rot`: 'q' / 'dq' / '6d' / 'rpy' -- the reference's four --r choices (mlp_reg.py:64-90); `params` in Q_PARAM_ORDER (DQ_PARAM_ORDER
    for 'dq').  `graph_branches`: creg_train_shape.graph_branches (include/creg.h) -- 0 = chain streams, the library's count."""

    def __init__(self, rot: str, k: int, hidden: int, n_pred: int, n_tgt: int, epochs: int = 300,
                 use_graph: bool = True, device=None, batch: int = 1, graph_branches: int = 0,
                 nn_search: int = 0):
        self.L = _lib.load()
        self.rot = TRAIN_ROT[rot]
        self.device = _lib.device(torch.device(device) if device is not None else None)
        self.batch = int(batch)
        self.hidden_model = int(hidden)                                     # the caller's width
        if hidden < 2 or hidden > TRAIN_HIDDEN_TILES[-1]:
            raise ValueError(f"unsupported train shape: hidden {hidden} (2 .. {TRAIN_HIDDEN_TILES[-1]})")
        hidden = next(t for t in TRAIN_HIDDEN_TILES if t >= hidden)         # the width the kernels run at
        self.shape = _lib.TrainShape(self.rot, k, hidden, epochs, n_pred, n_tgt, int(use_graph), self.batch,
                                     int(graph_branches), int(nn_search))
        need = self.L.creg_train_workspace_bytes(ctypes.byref(self.shape))
        if need == 0:
            raise ValueError(f"unsupported train shape (rot {rot!r}, k = {k}, hidden {hidden}): the plan takes 1 <= k <= 256 clusters (fewer for --r 6d at hidden 512: the backward's LDS tiles) "
                             "and hidden widths up to 512 (include/creg.h, creg_train_shape)")
        self.ws = torch.empty(need + 256, dtype=torch.uint8, device=self.device)
        base = (self.ws.data_ptr() + 255) // 256 * 256
        self.plan = ctypes.c_void_p()
        _lib.check(self.L.creg_train_plan_create(ctypes.byref(self.shape), ctypes.c_void_p(base), need,
                                                 ctypes.byref(self.plan)), "creg_train_plan_create")
        self.k, self.n_pred, self.n_tgt, self.epochs, self.hidden = k, n_pred, n_tgt, epochs, hidden
        info = (ctypes.c_int32 * 9)()
        _lib.check(self.L.creg_train_plan_info(self.plan, info), "creg_train_plan_info")
        #: what the plan chose from its shape (never a result): searches, graph branches, problems per launch, epochs per graph
        self.info = {"pruned_target_search": bool(info[0]), "pruned_predicted_search": bool(info[1]), "graph_branches": int(info[2]),
                     "batch": int(info[3]), "epochs_per_graph": int(info[4]), "nn_points_per_lane": int(info[5]),
                     "nn_boxes_per_lane": (int(info[6]), int(info[7])), "nn_queries_per_wave": 16 if int(info[0]) == 2 else 4}
        if nn_search == 0 and not (self.info["pruned_target_search"] and self.info["pruned_predicted_search"]):
            import warnings
            which = [d for d, on in (("predicted -> target", self.info["pruned_target_search"]),
                                     ("target -> predicted", self.info["pruned_predicted_search"])) if not on]
            warnings.warn(f"creg train plan (k={k}, n_pred={n_pred}, n_tgt={n_tgt}): exhaustive nearest-neighbour search in the "
                          f"{' and '.join(which)} direction(s) -- the shape is beyond the block-pruned search's limits "
                          "(n_tgt <= 65536: four chunks of k-d leaves built in LDS; n_pred < 65535 and at most 512 blocks of the padded "
                          "predicted cloud); same results, several times the launch time", RuntimeWarning, stacklevel=2)

    def chain_probe_us(self) -> int:
        """creg_train_plan_info_t.chain_probe_us AFTER a run: < 250 = the chain streams ran concurrently with the caller's (own hardware queues),
        -1 = a chain shares a queue, 0 = not probed (one chain / no run yet)."""
        info = (ctypes.c_int32 * 9)()
        _lib.check(self.L.creg_train_plan_info(self.plan, info), "creg_train_plan_info")
        return int(info[8])

    def __del__(self):
        plan = getattr(self, "plan", None)
        if plan is not None and plan.value and ctypes is not None:      # (module globals may be gone at interpreter exit)
            self.L.creg_train_plan_destroy(plan)
            self.plan = None

    def _args(self, m, y, pts, offsets, params, lr, factor, patience, stop, outs):
        n = 6 if self.rot == 1 else 10
        if len(params) != n:
            raise ValueError(f"expected {n} parameter tensors, got {len(params)}")
        keep = [_need(m, torch.float32, "m"), _need(y, torch.float32, "y"), _need(pts, torch.float32, "pts"),
                _need(offsets, torch.int32, "offsets")]
        # the kernels index with the PLAN's sizes: a tensor of another shape would be read out of bounds, not rejected
        if tuple(keep[0].shape) != (self.k, 4, 4):
            raise ValueError(f"m must be ({self.k},4,4) for this plan, got {tuple(keep[0].shape)}")
        if tuple(keep[1].shape) != (self.n_tgt, 3):
            raise ValueError(f"y must be ({self.n_tgt},3) for this plan, got {tuple(keep[1].shape)}")
        if tuple(keep[2].shape) != (self.n_pred, 3):
            raise ValueError(f"pts must be ({self.n_pred},3) for this plan, got {tuple(keep[2].shape)}")
        if keep[3].numel() != self.k + 1:
            raise ValueError(f"offsets must hold {self.k + 1} entries, got {keep[3].numel()}")
        shapes_model = self._param_shapes(self.hidden_model)
        for p, want in zip(params, shapes_model):
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                raise TypeError("model parameters must be contiguous fp32 CUDA tensors")
            if p.numel() != int(np.prod(want)):
                raise ValueError(f"parameter tensor of {p.numel()} elements where the plan's model has {int(np.prod(want))}")
        self._padded = None
        if self.hidden_model != self.hidden:                                # zero-padded copies at the kernels' width (class docstring)
            padded = []
            for p, sm, sp in zip(params, shapes_model, self._param_shapes(self.hidden)):
                q = torch.zeros(sp, dtype=torch.float32, device=p.device)
                q[tuple(slice(0, d) for d in sm)] = p.view(sm)
                padded.append(q)
            self._padded = (padded, list(params), shapes_model)
            params = padded
        arr = (ctypes.c_void_p * n)(*[p.data_ptr() for p in params])
        self._keep = getattr(self, "_keep", [])[-64:] + [keep, arr, params]      # alive until the enqueued work has read them
        a = _lib.TrainArgs()
        a.m, a.y, a.local_pts, a.seg_offsets = [t.data_ptr() for t in keep]
        a.params = ctypes.cast(arr, ctypes.POINTER(ctypes.c_void_p))
        a.lr, a.sched_factor, a.sched_patience, a.stop = lr, factor, patience, stop
        a.best_m, a.best_pred, a.loss_hist, a.lr_hist, a.result = [o.data_ptr() if o is not None else None for o in outs]
        return a

    def _param_shapes(self, H):
        """Shapes of the model's tensors in Q_PARAM_ORDER / DQ_PARAM_ORDER at width H (model_utils.py:65-168)."""
        if self.rot != 1:      # QRegMLP: enc 56->H, dec1 H->H/2->3, dec2 H->H->4; RRegMLP: 72 features, 6 outputs; RegMLP: 48, 3 (+ Tanh)
            nin, nout = _TWO_DECODER[self.rot]
            return [(H, nin), (H,), (H // 2, H), (H // 2,), (3, H // 2), (3,), (H, H), (H,), (nout, H), (nout,)]
        return [(H, 64), (H,), (H, H), (H,), (8, H), (8,)]        # DQRegMLP: enc 64->H, H->H, H->8

    def _param_numels(self):
        """Element counts of the model's tensors at the kernels' width."""
        return [int(np.prod(s)) for s in self._param_shapes(self.hidden)]

    def _unpad(self, padded):
        """Trained parameters back into the caller's tensors (stream-ordered copies of the leading blocks)."""
        if padded is not None:
            for q, p, sm in zip(*padded):
                p.view(sm).copy_(q[tuple(slice(0, d) for d in sm)])

    def _outs(self):
        dev = self.device
        return (torch.empty(self.k, 4, 4, dtype=torch.float32, device=dev),
                torch.empty(self.n_pred, 3, dtype=torch.float32, device=dev),
                torch.empty(self.epochs, dtype=torch.float32, device=dev),
                torch.empty(self.epochs, dtype=torch.float32, device=dev),
                torch.empty(4, dtype=torch.float32, device=dev))

    def run(self, m, y, pts, offsets, params, lr=2e-4, factor=0.7, patience=5, stop=200, same_target=False):
        """Returns (best_m (k,4,4), best_pred (n_pred,3), result (4) = [min_loss, epochs_run, lr,
        best_epoch], loss_hist (epochs), lr_hist (epochs)); all device tensors, stream-ordered."""
        return self.run_batch([(m, y, pts, offsets, params)], lr, factor, patience, stop, same_target)[0]

    def run_batch(self, problems, lr=2e-4, factor=0.7, patience=5, stop=200, same_target=False):
        """problems: list of `batch` tuples (m, y, pts, offsets, params) of identical shape, advanced
        together (one launch carries all of them).  Returns one result tuple per problem, as `run`.
        same_target: the caller vouches that every problem's `y` holds the values of this plan's previous run of that slot
        ("Anchor" after "Step" on the same frame, mlp_reg.py:338-356): the frame's k-d leaf blocks are kept instead of rebuilt."""
        if len(problems) != self.batch:
            raise ValueError(f"this plan advances {self.batch} problems per launch, got {len(problems)}")
        arr = (_lib.TrainArgs * self.batch)()
        outs, padded = [], []
        for b, (m, y, pts, offsets, params) in enumerate(problems):
            best_m, best_pred, lh, lrh, result = o = self._outs()
            arr[b] = self._args(m, y, pts, offsets, params, lr, factor, patience, stop, o)
            arr[b].y_unchanged = 1 if same_target else 0
            padded.append(self._padded)
            outs.append((best_m, best_pred, result, lh, lrh))
        _lib.check(self.L.creg_train_plan_run_batch(self.plan, arr, self.batch, _stream()), "creg_train_plan_run_batch")
        for pd in padded:
            self._unpad(pd)
        return outs

    #: scalar fields of a train's control state (creg_train_state; `state_out` of creg_train_plan_resume holds them in this order + last_loss)
    STATE_FIELDS = ("step", "epochs_run", "lr", "sched_best", "sched_bad", "count", "min_loss", "best_epoch", "stopped")

    def resume(self, m, y, pts, offsets, params, state, n_epochs=1, factor=0.7, patience=5, stop=200, best=None):
        """`n_epochs` further epochs of one train from a caller-supplied optimizer / control state (creg_train_plan_resume): checkpoint /
        resume, and the teacher-forced parity hook.  `state`: dict with `exp_avg`, `exp_avg_sq` (lists of fp32 CUDA tensors shaped like
        `params`: torch.optim.Adam's moments) and the scalars of STATE_FIELDS (missing ones default to a fresh train's).  `best`:
        (best_m (k,4,4), best_pred (n_pred,3)) so far, required when state['best_epoch'] >= 0.  `params` are updated in place.
        Returns (best_m, best_pred, result, loss_hist, lr_hist, state_after) -- state_after like `state`, its scalars read back
        (synchronises the stream for them)."""
        if self.hidden_model != self.hidden:
            raise ValueError("resume needs the model at one of the kernels' widths (64, 128, 256, 512): no zero-padding of the moments")
        n = 6 if self.rot == 1 else 10
        ea, es = list(state["exp_avg"]), list(state["exp_avg_sq"])
        if len(ea) != n or len(es) != n:
            raise ValueError(f"expected {n} moment tensors of each kind")
        for q, p in zip(ea + es, list(params) * 2):
            if not (q.is_cuda and q.dtype == torch.float32 and q.is_contiguous() and q.numel() == p.numel()):
                raise TypeError("Adam moments must be contiguous fp32 CUDA tensors shaped like the parameters")
        o = self._outs()
        best_epoch = int(state.get("best_epoch", -1))
        if best_epoch >= 0:
            if best is None:
                raise ValueError("state['best_epoch'] >= 0 needs best=(best_m, best_pred)")
            o[0].copy_(best[0]); o[1].copy_(best[1])
        a = self._args(m, y, pts, offsets, params, float(state.get("lr", 2e-4)), factor, patience, stop, o)
        st = _lib.TrainState()
        arr_a = (ctypes.c_void_p * n)(*[q.data_ptr() for q in ea])
        arr_s = (ctypes.c_void_p * n)(*[q.data_ptr() for q in es])
        st.exp_avg, st.exp_avg_sq = ctypes.cast(arr_a, ctypes.POINTER(ctypes.c_void_p)), ctypes.cast(arr_s, ctypes.POINTER(ctypes.c_void_p))
        st.lr, st.sched_best = float(state.get("lr", 2e-4)), float(state.get("sched_best", float("inf")))
        st.step, st.epochs_run = int(state.get("step", 0)), int(state.get("epochs_run", state.get("step", 0)))
        st.sched_bad, st.count, st.best_epoch, st.stopped = int(state.get("sched_bad", 0)), int(state.get("count", 0)), best_epoch, int(state.get("stopped", 0))
        st.min_loss = float(state.get("min_loss", 1000.0))
        ea2, es2 = [torch.empty_like(q) for q in ea], [torch.empty_like(q) for q in es]
        out_a = (ctypes.c_void_p * n)(*[q.data_ptr() for q in ea2])
        out_s = (ctypes.c_void_p * n)(*[q.data_ptr() for q in es2])
        sc = torch.empty(12, dtype=torch.float64, device=self.device)
        _lib.check(self.L.creg_train_plan_resume(self.plan, ctypes.byref(a), ctypes.byref(st), int(n_epochs),
                                                 ctypes.cast(out_a, ctypes.POINTER(ctypes.c_void_p)), ctypes.cast(out_s, ctypes.POINTER(ctypes.c_void_p)),
                                                 _p(sc), _stream()), "creg_train_plan_resume")
        h = sc.cpu().tolist()
        after = {"exp_avg": ea2, "exp_avg_sq": es2, "last_loss": h[9]}
        for i, k in enumerate(self.STATE_FIELDS):
            after[k] = h[i] if k in ("lr", "sched_best", "min_loss") else int(h[i])
        return o[0], o[1], o[4], o[2], o[3], after

    def probe(self, m, y, pts, offsets, params):
        """One forward + pose-gradient evaluation (test hook): (m2, pred, loss, grad_m2)."""
        dev = self.device
        m2 = torch.empty(self.k, 4, 4, dtype=torch.float32, device=dev)
        pred = torch.empty(self.n_pred, 3, dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        gm = torch.empty(self.k, 4, 4, dtype=torch.float32, device=dev)
        a = self._args(m, y, pts, offsets, params, 2e-4, 0.7, 5, 200, (None, None, None, None, None))
        _lib.check(self.L.creg_train_plan_probe(self.plan, ctypes.byref(a), _p(m2), _p(pred), _p(loss), _p(gm),
                                                _stream()), "creg_train_plan_probe")
        return m2, pred, loss, gm

    KERNELS = ("_", "head", "nn_l1", "gradc", "bd", "l2")

    def profile(self, m, y, pts, offsets, params, n_epochs=50):
        """Average event-bracketed microseconds of each of the 5 epoch kernels, and the back-to-back launch time of each
        (synchronises; the plan's copy of the parameters moves on, the caller's tensors are not written)."""
        out = (ctypes.c_float * 16)()
        a = self._args(m, y, pts, offsets, params, 2e-4, 0.7, 5, 200, (None, None, None, None, None))
        _lib.check(self.L.creg_train_plan_profile(self.plan, ctypes.byref(a), n_epochs, out, _stream()),
                   "creg_train_plan_profile")
        d = dict(zip(self.KERNELS, [float(v) for v in out]))
        d.pop("_")
        d["nn_l1_back_to_back"] = float(out[6])
        d["nn_l1_problems_per_launch"] = int(out[7])
        d["bd_back_to_back"] = float(out[8])
        d["l2_back_to_back"] = float(out[9])
        d["head_back_to_back"] = float(out[10])
        d["gradc_back_to_back"] = float(out[11])
        return d
