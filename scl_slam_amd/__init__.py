"""scl_slam_amd -- MI355X-native Scan Context loop-closure engine (host-side Python mirror).

The product is the C-ABI shared library ``scl_slam_amd/lib/libscl_engine.so`` (HIP kernels
for gfx950 + host engine, see ``include/scl_engine.h``).  This package is a thin ctypes
binding over it that mirrors the reference's plugin interface (``scan_descriptor``,
include/descriptor.h:21-36) so tests and benchmarks read like the reference's call sites.
There is no CPU fallback: importing works anywhere, creating an engine needs a HIP device.
"""
from ._native import load_library, LIB_PATH, NativeLibraryError  # noqa: F401
from .engine import (  # noqa: F401
    SclConfig, SclError, ScanContextEngine, ScanContextDescriptor, IcpParams, QUERY_STAGED,
)

__all__ = [
    "load_library", "LIB_PATH", "NativeLibraryError", "SclConfig", "SclError",
    "ScanContextEngine", "ScanContextDescriptor", "IcpParams", "QUERY_STAGED",
]
