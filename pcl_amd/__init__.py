"""pcl_amd - MI355X-native kNN / radius search / ICP behind the PCL API.

The product is the C-ABI library `pcl_amd/lib/libpclmi.so` (include/pclmi.h) built from the
hand-written HIP sources in `pcl_amd/csrc/`; the C++ facade in include/pcl/ and the thin ctypes
layer in `pcl_amd.capi` are its callers.  There is no CPU fallback: every compute call fails
loudly when the library or a HIP device is missing.
"""
from . import capi  # noqa: F401
from .capi import PclmiError, load_library  # noqa: F401

__all__ = ["capi", "PclmiError", "load_library"]
