"""MI355X-native drop-in for AutoURDF's cluster-registration path (PointCloud/mlp_reg.py, cluster_icp.py, dq_func.py).

Importing the package has NO process-wide side effects.  The command-line entry points (`python -m autourdf_amd.mlp_reg`, `bench.py`)
call `prefer_device_kernargs()` before the HIP runtime starts; a host program that embeds the package should do the same (or export
HIP_FORCE_DEV_KERNARG=1 itself): with kernel arguments in host memory every launch of the captured epoch graph starts ~1 us later
(headline 123 instead of ~150 frames/s in round 3's measurement)."""
import os as _os


def prefer_device_kernargs() -> bool:
    """`HIP_FORCE_DEV_KERNARG=1` unless the caller's environment already says something else.  The HIP runtime reads the variable when
    it initialises, so this only works BEFORE the first device call of the process: returns False (and warns) when it is too late."""
    import sys
    import warnings
    if "HIP_FORCE_DEV_KERNARG" in _os.environ:
        return True
    torch = sys.modules.get("torch")
    if torch is not None and torch.cuda.is_initialized():
        warnings.warn("autourdf_amd.prefer_device_kernargs(): the HIP runtime is already initialised, HIP_FORCE_DEV_KERNARG can no "
                      "longer be set for this process (export it before starting Python)", RuntimeWarning, stacklevel=2)
        return False
    _os.environ["HIP_FORCE_DEV_KERNARG"] = "1"
    return True
