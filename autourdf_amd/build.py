"""Build libcreg.so (hipcc, gfx950 only) in-tree so it travels with the repo snapshot.

    python -m autourdf_amd.build [--force] [--asan]

`--asan` / `--ubsan` build a SECOND library, libcreg_asan.so / libcreg_ubsan.so (objects under csrc/_obj_<variant>): the host side of
every entry point -- plans, chain streams, pinned rings, events, argument checks -- under AddressSanitizer or UBSan
(`-fno-gpu-sanitize`: device code is compiled as always).  ROCm's own ASan runtime (libclang_rt.asan) intercepts the HSA allocator and
aborts inside `hipInit` unless the whole ROCm stack is its -asan build ("out of memory ... hsa_amd_memory_pool_allocate", measured on
the GPU box), so the ASan objects are linked WITHOUT a runtime and take gcc's libasan (same interface, no HSA interceptors) from
LD_PRELOAD:
    CREG_LIB_VARIANT=asan LD_PRELOAD="/usr/lib/x86_64-linux-gnu/libasan.so.6 /usr/lib/x86_64-linux-gnu/libstdc++.so.6" ASAN_OPTIONS=detect_leaks=0 python -m pytest tests -m gpu
    CREG_LIB_VARIANT=ubsan python -m pytest tests -m gpu
(tools/run_sanitizer_suite.sh; SURVEY section 5 / VERDICT r4 item 8).
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libcreg.so")
SOURCES = ["core.hip", "nn_l1.hip", "transform.hip", "se3.hip", "kmeans.hip", "icp.hip", "fps.hip", "coord_map.hip", "sample.hip", "kmeans_nd.hip", "normals.hip", "train_engine.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall",
         "-Wno-unused-function",
         # MFMA accumulators in VGPRs: the default AGPR form costs 8 v_accvgpr_read per fp64 16x16x4 MFMA whose
         # result the VALU consumes (the K2 matrix-core E-step), which doubled that kernel's VALU work
         "-mllvm", "-amdgpu-mfma-vgpr-form"]


def kernel_source_sha256() -> str:
    """Fingerprint of the train plan's kernel sources: tools/summarize_profiles.py stores it beside the rocprofv3 counters it
    summarises, bench.py compares it with the tree it runs from and marks counters of another build `stale`."""
    import hashlib
    h = hashlib.sha256()
    for f in ("train_engine.hip", "nn_l1.h", "creg_dev.h", "creg_common.h"):
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libcreg.so cannot be built on this machine")
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


VARIANT_FLAGS = {"asan": ["-fsanitize=address", "-fno-gpu-sanitize", "-fno-omit-frame-pointer", "-g"],
                 "ubsan": ["-fsanitize=undefined", "-fno-sanitize-recover=undefined", "-fno-gpu-sanitize", "-fno-omit-frame-pointer", "-g"]}
VARIANT_LINK = {"asan": ["-Wl,--allow-shlib-undefined"],                  # the runtime comes from LD_PRELOAD (gcc's libasan)
                "ubsan": ["-fsanitize=undefined", "-fno-gpu-sanitize"]}


def build_lib(force: bool = False, verbose: bool = False, variant: str = "") -> str:
    OBJ = os.path.join(CSRC, "_obj" + ("_" + variant if variant else ""))
    LIB = os.path.join(HERE, "libcreg" + ("_" + variant if variant else "") + ".so")
    # any other variant name: an A/B build of the same sources with CREG_EXTRA_FLAGS, loaded with CREG_LIB_VARIANT=<name> (both builds
    # travel in one snapshot, so one gpurun call alternates them on one box)
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "creg.h"))
    hipcc = _hipcc()
    extra = os.environ.get("CREG_EXTRA_FLAGS", "").split()      # e.g. -DCREG_BACK_STAMPS for tools/back_stamps.py
    if variant in VARIANT_FLAGS:
        extra = extra + VARIANT_FLAGS[variant]
    objs, procs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [hipcc] + FLAGS + extra + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode()}")
        if verbose and out:
            print(out.decode())
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + VARIANT_LINK.get(variant, []) + objs
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True, variant="asan" if "--asan" in sys.argv else ("ubsan" if "--ubsan" in sys.argv else
                                                                   (sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else ""))))
