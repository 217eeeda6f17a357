import os as _os

# kernel arguments in device memory (this image's default; 20 % of the registered frames/s otherwise) unless the caller says otherwise;
# read by the HIP runtime when it initialises, i.e. effective when the package is imported before the first device call
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
