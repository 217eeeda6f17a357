"""ctypes binding of libcreg.so (include/creg.h).  There is no CPU fallback: a missing library, a
missing GPU or a non-gfx950 device raises immediately."""
import ctypes
import os

import torch  # noqa: F401  -- FIRST: brings torch's bundled libamdhip64 (SONAME libamdhip64.so.7) into the
#                 process so libcreg.so binds to the same HIP runtime instead of a second copy from /opt/rocm

_HERE = os.path.dirname(os.path.abspath(__file__))
# CREG_LIB_VARIANT=asan: the sanitizer build of the same sources (python -m autourdf_amd.build --asan), for tools/run_sanitizer_suite.sh
_VARIANT = os.environ.get("CREG_LIB_VARIANT", "")
LIB_PATH = os.path.join(_HERE, "libcreg" + ("_" + _VARIANT if _VARIANT else "") + ".so")

vp, i64, i32, f32, f64, sz = (ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_float,
                              ctypes.c_double, ctypes.c_size_t)


class TrainShape(ctypes.Structure):
    _fields_ = [("rot", i32), ("k", i32), ("hidden", i32), ("epochs", i32), ("n_pred", i64),
                ("n_tgt", i64), ("use_graph", i32), ("batch", i32), ("graph_branches", i32), ("nn_search", i32)]


class TrainArgs(ctypes.Structure):
    _fields_ = [("m", vp), ("y", vp), ("local_pts", vp), ("seg_offsets", vp),
                ("params", ctypes.POINTER(vp)), ("lr", f32), ("sched_factor", f32),
                ("sched_patience", i32), ("stop", i32), ("best_m", vp), ("best_pred", vp),
                ("loss_hist", vp), ("lr_hist", vp), ("result", vp), ("y_unchanged", i32), ("reserved_", i32)]


class TrainState(ctypes.Structure):          # creg_train_state: the optimizer / control state creg_train_plan_resume continues from
    _fields_ = [("exp_avg", ctypes.POINTER(vp)), ("exp_avg_sq", ctypes.POINTER(vp)), ("lr", f64), ("sched_best", f64),
                ("step", i32), ("epochs_run", i32), ("sched_bad", i32), ("count", i32), ("best_epoch", i32), ("stopped", i32),
                ("min_loss", f32), ("reserved_", i32)]


class IcpProblem(ctypes.Structure):
    _fields_ = [("local", vp), ("world", vp), ("seg_offsets", vp), ("frame", vp), ("M", vp),
                ("M_out", vp), ("world_out", vp), ("n_iter_out", vp), ("tgt_offsets", vp), ("world_offsets", vp)]


# name -> (restype, argtypes); every symbol include/creg.h declares
SIGNATURES = {
    "creg_version": (ctypes.c_int, []),
    "creg_last_error": (ctypes.c_char_p, []),
    "creg_device_check": (ctypes.c_int, []),
    "creg_nn_l1_bidir_f32": (ctypes.c_int, [vp, i64, vp, i64, vp, vp, vp, vp, vp]),
    "creg_nn_l1_bwd_scratch_bytes": (sz, [i64]),
    "creg_nn_l1_bwd_f32": (ctypes.c_int, [vp, i64, vp, i64, vp, vp, f32, f32, vp, vp, vp]),
    "creg_chamfer_l1_reduce_f32": (ctypes.c_int, [vp, i64, vp, i64, vp, vp]),
    "creg_cluster_transform_f32": (ctypes.c_int, [vp, i64, vp, i32, vp, vp, vp]),
    "creg_cluster_transform_bwd_f32": (ctypes.c_int, [vp, vp, i32, vp, vp, vp]),
    "creg_kmeans_workspace_bytes": (sz, [i64, i32]),
    "creg_kmeans_lloyd_f64": (ctypes.c_int, [vp, i64, vp, i32, i32, f64, i32, vp, vp, vp, vp, vp, sz, vp]),
    "creg_kmeans_batch_workspace_bytes": (sz, [i64, i32, i32]),
    "creg_kmeans_lloyd_batch_f64": (ctypes.c_int, [vp, i64, vp, i32, i32, i32, f64, vp, vp, vp, vp, vp, sz, vp]),
    "creg_kmeans_assign_f64": (ctypes.c_int, [vp, i64, vp, i32, i32, vp, vp]),
    "creg_group_to_local_f64": (ctypes.c_int, [vp, i64, vp, i32, vp, i32, vp, vp, vp]),
    "creg_group_to_local_batch_f64": (ctypes.c_int, [vp, i64, vp, i32, vp, i32, i32, vp, vp, vp]),
    "creg_fps_scratch_bytes": (sz, [i64]),
    "creg_fps_f64": (ctypes.c_int, [vp, i64, i64, vp, vp, vp]),
    "creg_se3_to_dq_f32": (ctypes.c_int, [vp, i32, vp, vp]),
    "creg_dq_to_se3_f32": (ctypes.c_int, [vp, i32, vp, vp]),
    "creg_dq_to_se3_bwd_f32": (ctypes.c_int, [vp, vp, i32, vp, vp]),
    "creg_dq_multiply_f32": (ctypes.c_int, [vp, vp, i32, vp, vp]),
    "creg_dq_invert_f32": (ctypes.c_int, [vp, i32, vp, vp]),
    "creg_dq_to_quat_trans_f32": (ctypes.c_int, [vp, i32, vp, vp, vp]),
    "creg_quat_trans_to_dq_f32": (ctypes.c_int, [vp, vp, i32, vp, vp]),
    "creg_matrix_to_quat_f32": (ctypes.c_int, [vp, i32, vp, vp]),
    "creg_quat_to_matrix_f32": (ctypes.c_int, [vp, i32, vp, vp]),
    "creg_icp_workspace_bytes": (sz, [i64, i64, i32]),
    "creg_masked_icp_f64": (ctypes.c_int, [vp, vp, vp, i64, vp, i32, vp, i64, vp, f64, f64, i32, i32, vp, vp, vp, vp, sz, vp]),
    "creg_icp_batch_workspace_bytes": (sz, [i64, i64, i32, i32]),
    "creg_masked_icp_batch_f64": (ctypes.c_int, [ctypes.POINTER(IcpProblem), i32, i64, i32, i64, f64, f64, i32, i32, vp, sz, vp]),
    "creg_aabb_mask_f64": (ctypes.c_int, [vp, vp, i32, vp, i64, f64, vp, vp, vp, vp]),
    "creg_icp_p2p_f64": (ctypes.c_int, [vp, i64, vp, vp, i64, vp, i32, vp, f64, i32, vp, vp, vp, vp, sz, vp]),
    "creg_icp_nn_counters": (ctypes.c_int, [ctypes.POINTER(f64), i32, i32]),
    "creg_kabsch_f64": (ctypes.c_int, [vp, vp, vp, i64, vp, i32, vp, vp]),
    "creg_knn_normals_f64": (ctypes.c_int, [vp, i64, f64, i32, vp, vp, vp, vp]),
    "creg_kmeans_nd_workspace_bytes": (sz, [i64]),
    "creg_kmeans_lloyd_nd_f64": (ctypes.c_int, [vp, i64, i32, vp, i32, i32, f64, vp, vp, vp, vp, vp, sz, vp]),
    "creg_sample_mesh_f64": (ctypes.c_int, [vp, vp, vp, i32, vp, i32, vp, i64, vp, vp, vp]),
    "creg_visibility_workspace_bytes": (sz, [i32, i32, i32]),
    "creg_visibility_f64": (ctypes.c_int, [vp, vp, i32, vp, i32, vp, i32, f64, f64, f64, f64, i32, i32, vp, i64, f64, vp, vp, sz, vp]),
    "creg_coord_dist_map_workspace_bytes": (sz, [i32, i32]),
    "creg_coord_dist_map_f64": (ctypes.c_int, [vp, i32, i32, f64, i32, vp, vp, vp, sz, vp]),
    "creg_pose_coords_f64": (ctypes.c_int, [vp, i64, vp, vp]),
    "creg_train_workspace_bytes": (sz, [ctypes.POINTER(TrainShape)]),
    "creg_train_plan_create": (ctypes.c_int, [ctypes.POINTER(TrainShape), vp, sz, ctypes.POINTER(vp)]),
    "creg_train_plan_run": (ctypes.c_int, [vp, ctypes.POINTER(TrainArgs), vp]),
    "creg_train_plan_run_batch": (ctypes.c_int, [vp, ctypes.POINTER(TrainArgs), i32, vp]),
    "creg_train_plan_resume": (ctypes.c_int, [vp, ctypes.POINTER(TrainArgs), ctypes.POINTER(TrainState), i32, ctypes.POINTER(vp), ctypes.POINTER(vp), vp, vp]),
    "creg_train_plan_probe": (ctypes.c_int, [vp, ctypes.POINTER(TrainArgs), vp, vp, vp, vp, vp]),
    "creg_train_plan_profile": (ctypes.c_int, [vp, ctypes.POINTER(TrainArgs), i32, ctypes.POINTER(f32), vp]),
    "creg_train_plan_info": (ctypes.c_int, [vp, vp]),
    "creg_train_plan_destroy": (ctypes.c_int, [vp]),
}

_lib = None


def load(check_device: bool = True) -> ctypes.CDLL:
    """dlopen libcreg.so and bind every entry point.  Raises RuntimeError when it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m autourdf_amd.build` "
                "(hipcc --offload-arch=gfx950). autourdf_amd has no CPU fallback.")
        hip = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if os.path.exists(hip):
            ctypes.CDLL(hip, mode=ctypes.RTLD_GLOBAL)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)            # AttributeError if the .so lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = L
    if check_device:
        rc = _lib.creg_device_check()
        if rc != 0:
            raise RuntimeError("libcreg needs an MI355X (gfx950): " + _lib.creg_last_error().decode())
    return _lib


def check(rc: int, what: str = "libcreg call"):
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {_lib.creg_last_error().decode()}")


def device(*refs) -> torch.device:
    """The device a host-side entry point works on: the device of the first CUDA tensor among ``refs`` (or a
    ``torch.device`` handed in), otherwise the CURRENT device with its index spelled out -- never a bare "cuda"
    string, so a host that embeds the package on GPU != 0 keeps inputs, outputs and launches on one device."""
    for r in refs:
        if isinstance(r, torch.Tensor) and r.is_cuda:
            return r.device
        if isinstance(r, torch.device) and r.type == "cuda":
            return r if r.index is not None else torch.device("cuda", torch.cuda.current_device())
    return torch.device("cuda", torch.cuda.current_device())
