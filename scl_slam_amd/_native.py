"""Loader of the C-ABI library.  Fails loudly: there is no fallback path."""
import ctypes
import os

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libscl_engine.so")
# SCL_ENGINE_LIB: another build of the same library (kernel experiments: scripts/build_variant.sh); said out loud when used
if os.environ.get("SCL_ENGINE_LIB"):
    LIB_PATH = os.path.abspath(os.environ["SCL_ENGINE_LIB"])
    import sys
    print(f"scl_slam_amd: loading the engine from SCL_ENGINE_LIB={LIB_PATH}", file=sys.stderr)


class NativeLibraryError(RuntimeError):
    pass


_lib = None


def load_library():
    """dlopen libscl_engine.so (built in-tree by ``make`` / ``__graft_entry__.build()``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} is missing: build it with `make` (or __graft_entry__.build()). "
            "scl_slam_amd has no CPU fallback.")
    try:
        _lib = ctypes.CDLL(LIB_PATH)
    except OSError as exc:  # pragma: no cover - environment specific
        raise NativeLibraryError(f"cannot load {LIB_PATH}: {exc}") from exc
    return _lib
