"""The exchange of FullScanStream over RCCL (torch.distributed backend "nccl") on device tensors, with the
one rank a one-GPU box offers: same calls, shapes and asynchronous wait as the multi-GPU bench uses."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

from scl_slam_amd import ScanContextEngine
from scl_slam_amd.sharded import FullScanStream
from scl_slam_amd.synth import synth_descriptors

pytestmark = pytest.mark.gpu


def test_full_scan_stream_over_rccl_one_rank():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        R, S, n = 64, 120, 700
        descs = synth_descriptors(n, R, S, seed=77, revisit_frac=0.1)
        e = ScanContextEngine(num_ring=R, num_sector=S, num_exclude_recent=50, initial_capacity=1024)
        e.save_bulk(descs)
        queries = list(range(n - 1, n - 12, -1))
        single = [e.detect_full_range(q, 0, q - 50) for q in queries]
        st = FullScanStream(e, rank=0, world=1, device="cuda", depth=2, merge_every=4, scans_per_launch=2, always_exchange=True)
        for q in queries:
            st.submit(q, 0, q - 50)
        res = st.drain()
        assert len(res) == len(queries)
        for (nn, sh, d), (d2, g, sh2) in zip(single, res):
            assert (nn, sh) == (g, sh2) and d == d2
        # bench.py's form: arrays through the native pipeline, one exchange per chunk of scans
        st = FullScanStream(e, rank=0, world=1, device="cuda", depth=2, scans_per_launch=4, native_chunk=4, always_exchange=True)
        st.submit_many(np.array(queries, np.int32), 0, np.array(queries, np.int32) - 50)
        res = st.drain()
        assert len(res) == len(queries)
        for (nn, sh, d), (d2, g, sh2) in zip(single, res):
            assert (nn, sh) == (g, sh2) and d == d2
        e.close()
    finally:
        dist.destroy_process_group()
