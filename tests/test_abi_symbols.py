"""The C-ABI library loads and exports every symbol include/scl_engine.h declares (no GPU needed)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADERS = [os.path.join(ROOT, "include", h) for h in ("scl_engine.h", "scl_messages.h", "scl_iris.h")]


def declared_symbols():
    names = set()
    for header in HEADERS:
        text = open(header).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(scl_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_header_declares_the_six_virtuals():
    names = declared_symbols()
    for n in ["scl_make_and_save", "scl_save_from_wire", "scl_detect_intra", "scl_detect_inter",
              "scl_get_index", "scl_get_size", "scl_icp_align", "scl_sc_distance_batch", "scl_ringkey_topk"]:
        assert n in names


def test_library_exports_every_declared_symbol():
    from scl_slam_amd import load_library, LIB_PATH
    assert os.path.exists(LIB_PATH), "build first: make (or __graft_entry__.build())"
    lib = load_library()
    missing = [n for n in declared_symbols() if not hasattr(lib, n)]
    assert not missing, f"declared in scl_engine.h but not exported: {missing}"


def test_no_device_is_a_loud_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from scl_slam_amd import ScanContextEngine, SclError
    with pytest.raises(SclError) as ei:
        ScanContextEngine()
    assert ei.value.status == -2        # SCL_ERR_NO_DEVICE


def test_status_strings_and_defaults():
    from scl_slam_amd import load_library
    from scl_slam_amd.engine import SclConfig, IcpParams, _bind
    lib = load_library(); _bind(lib)
    assert lib.scl_status_string(0) == b"ok"
    assert lib.scl_abi_version() >= 1
    c = SclConfig(); assert lib.scl_default_config(ctypes.byref(c)) == 0
    # scan_context_descriptor ctor defaults, descriptor.h:1308-1316
    assert (c.num_ring, c.num_sector, c.num_candidates) == (20, 60, 3)
    assert (c.dist_thres, c.lidar_height, c.max_radius) == (0.14, 1.65, 80.0)
    assert (c.num_exclude_recent, c.tree_making_period, c.search_ratio) == (100, 10, 0.1)
    p = IcpParams(); assert lib.scl_icp_default_params(ctypes.byref(p)) == 0
    # DM.h:1109-1112
    assert (p.max_iterations, p.max_correspondence_dist) == (50, 100.0)
    assert (p.transformation_epsilon, p.euclidean_fitness_epsilon) == (1e-6, 1e-6)
    assert (p.estimator, p.normal_radius) == (0, 1.0)


def test_header_is_plain_c99(tmp_path):
    """the boundary is a C ABI: the header must compile as C99 with no C++ or torch types in it"""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    src = tmp_path / "cabi.c"
    src.write_text('#include "scl_engine.h"\n#include "scl_messages.h"\n#include "scl_iris.h"\nint main(void) { scl_config c; scl_icp_params p; (void)p; return scl_default_config(&c) == SCL_OK ? 0 : 1; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                        "-fsyntax-only", str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
