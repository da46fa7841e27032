"""The C++ adapter (scan_descriptor plugin shape) driven like distributedMapping.h drives it."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "adapter_check")


@pytest.mark.gpu
def test_cpp_adapter_end_to_end_one_gpu_and_sharded():
    """the six virtuals through the C++ adapter: unsharded, devices = {0} and (emulated) {0, 0}: identical loops"""
    assert os.path.exists(BIN), "build first (make / __graft_entry__.build())"
    digests = []
    for shards in ("0", "1", "2"):
        out = subprocess.run([BIN, "260", shards], capture_output=True, text=True, timeout=300)
        print(out.stdout, out.stderr)
        assert out.returncode == 0 and "ADAPTER OK" in out.stdout
        digests.append([l for l in out.stdout.splitlines() if l.startswith(("detections digest", "loops found", "inter:"))])
    assert digests[0] == digests[1] == digests[2] and len(digests[0]) == 3


def test_adapter_binary_is_built_and_links_the_c_abi():
    assert os.path.exists(BIN)
    ldd = subprocess.run(["ldd", BIN], capture_output=True, text=True).stdout
    assert "libscl_engine.so" in ldd and "not found" not in ldd.split("libscl_engine.so")[1].split("\n")[0]
