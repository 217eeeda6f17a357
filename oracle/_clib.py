"""ctypes loader for oracle/_build/libcreg_oracle.so (built by oracle/Makefile)."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libcreg_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "creg_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        vp, i64, i32, f32, f64 = (ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
                                  ctypes.c_float, ctypes.c_double)
        L.oracle_nn_l1_f32.argtypes = [vp, i64, vp, i64, vp, vp]
        L.oracle_nn_l1_f32.restype = None
        L.oracle_nn_l2_f64.argtypes = [vp, i64, vp, i64, vp, vp]
        L.oracle_nn_l2_f64.restype = None
        L.oracle_nn_l1_bwd_f32.argtypes = [vp, i64, vp, i64, vp, vp, f32, f32, vp, vp]
        L.oracle_nn_l1_bwd_f32.restype = None
        L.oracle_kmeans_assign_f64.argtypes = [vp, i64, vp, i32, vp]
        L.oracle_kmeans_assign_f64.restype = None
        L.oracle_kmeans_lloyd_f64.argtypes = [vp, i64, vp, i32, i32, f64, vp, vp, vp]
        L.oracle_kmeans_lloyd_f64.restype = ctypes.c_int
        L.oracle_fps_f64.argtypes = [vp, i64, i64, vp]
        L.oracle_fps_f64.restype = None
        L.oracle_set_threads.argtypes = [i32]
        L.oracle_set_threads.restype = None
        _lib = L
    return _lib
