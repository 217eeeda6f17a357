#!/bin/bash
# The GPU suite (a) against the AddressSanitizer build and (b) against the UBSan build of libcreg's host side (python -m
# autourdf_amd.build --asan / --ubsan: built HERE, on demand -- hipcc is on the GPU box too; the two libraries are listed in
# .gpurunignore, so no snapshot carries them), and (c) with the
# background contention streams (CREG_TEST_CONTENTION=1).  SURVEY section 5 / VERDICT r4 item 8.
# ASan's runtime is gcc's libasan: ROCm's own intercepts the HSA allocator and aborts in hipInit without the -asan ROCm stack.
# Logs: gpurun_out/<tag>_{asan,ubsan,contention}_gpu_suite.log
#     tools/run_sanitizer_suite.sh r05
tag=${1:-r05}
mkdir -p gpurun_out
python -m autourdf_amd.build --asan > gpurun_out/${tag}_sanitizer_build.log 2>&1 && python -m autourdf_amd.build --ubsan >> gpurun_out/${tag}_sanitizer_build.log 2>&1 || { echo "sanitizer build failed"; tail -5 gpurun_out/${tag}_sanitizer_build.log; exit 1; }
asan_rt="/usr/lib/x86_64-linux-gnu/libasan.so.6 /usr/lib/x86_64-linux-gnu/libstdc++.so.6"     # (libstdc++ too: the runtime resolves __cxa_throw when it starts, before python has loaded any C++ library)
# (ASan's dlopen interceptor makes libasan the "caller" of every dlopen, so libtorch's RUNPATH no longer finds its own lazily loaded
#  libraries -- "libcaffe2_nvrtc.so: cannot open shared object file" in torch._C._cuda_init: name the directory)
torch_lib=$(python -c 'import torch, os; print(os.path.join(os.path.dirname(torch.__file__), "lib"))' 2>/dev/null)
ubsan_rt=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.ubsan_standalone-x86_64.so)
filter() { grep -v "Warning\|warnings.warn\|^$\|amdgpu.ids"; }
{
  echo "== pytest -m gpu, CREG_LIB_VARIANT=asan, LD_PRELOAD=$asan_rt"
  CREG_LIB_VARIANT=asan LD_PRELOAD="$asan_rt" LD_LIBRARY_PATH="$torch_lib:$LD_LIBRARY_PATH" ASAN_OPTIONS=detect_leaks=0 python -c 'from autourdf_amd import _lib; _lib.load(); print("loaded:", _lib.LIB_PATH)' 2>&1 | filter
  # (the three tests that start CHILD interpreters under torch.distributed.run / with an RCCL group are left out: the preloaded
  #  runtime would then sit under torchrun's agent and RCCL's own threads -- a rank exits 1 there without touching libcreg;
  #  they run in the two other legs)
  CREG_LIB_VARIANT=asan LD_PRELOAD="$asan_rt" LD_LIBRARY_PATH="$torch_lib:$LD_LIBRARY_PATH" ASAN_OPTIONS=detect_leaks=0 timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider -k "not torchrun and not rccl" 2>&1 | filter | tail -40
} > gpurun_out/${tag}_asan_gpu_suite.log 2>&1
{
  echo "== pytest -m gpu, CREG_LIB_VARIANT=ubsan, LD_PRELOAD=$ubsan_rt"
  CREG_LIB_VARIANT=ubsan LD_PRELOAD=$ubsan_rt UBSAN_OPTIONS=print_stacktrace=1 python -c 'from autourdf_amd import _lib; _lib.load(); print("loaded:", _lib.LIB_PATH)' 2>&1 | filter
  CREG_LIB_VARIANT=ubsan LD_PRELOAD=$ubsan_rt UBSAN_OPTIONS=print_stacktrace=1 timeout 1800 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | filter | tail -40
} > gpurun_out/${tag}_ubsan_gpu_suite.log 2>&1
{
  echo "== pytest -m gpu, CREG_TEST_CONTENTION=1"
  CREG_TEST_CONTENTION=1 timeout 1800 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | filter | tail -15
} > gpurun_out/${tag}_contention_gpu_suite.log 2>&1
for f in asan ubsan contention; do echo "--- $f"; tail -n 4 gpurun_out/${tag}_${f}_gpu_suite.log; done
