"""Several database keyframes as queries in one launch (scl_detect_full_submit_many): every query's
result must be exactly what its own launch returns (bit-identical distance, same index and shift)."""
import numpy as np
import pytest

from scl_slam_amd import ScanContextEngine
from scl_slam_amd.sharded import FullScanStream
from scl_slam_amd.synth import synth_descriptors

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("R,S", [(64, 120), (20, 60)])
def test_batched_queries_equal_single_launches(R, S):
    n = 900
    descs = synth_descriptors(n, R, S, seed=31, revisit_frac=0.1)
    e = ScanContextEngine(num_ring=R, num_sector=S, num_candidates=3, num_exclude_recent=50, initial_capacity=1024)
    try:
        e.save_bulk(descs)
        queries = [n - 1, n - 7, 400, n - 2, 333, n - 3, 5]
        his = [q - 50 for q in queries]                      # each query has its own exclusion window (empty for q = 5)
        single = []
        for q, hi in zip(queries, his):
            single.append(e.detect_full_range(q, 0, hi))
        for group in (2, 3, 4, 7):
            got = []
            for i in range(0, len(queries), group):
                tickets = e.detect_full_submit_many(queries[i:i + group], 0, his[i:i + group])
                got += [e.detect_full_collect(t) for t in tickets]
            for a, b in zip(single, got):
                assert a[0] == b[0] and a[1] == b[1]
                assert np.float64(a[2]).view(np.uint64) == np.float64(b[2]).view(np.uint64)
        # the stream front end with two scans per launch delivers the same sequence
        st = FullScanStream(e, depth=2, merge_every=3, scans_per_launch=2)
        for q, hi in zip(queries, his):
            st.submit(q, 0, hi)
        res = st.drain()
        for a, (d, g, sh) in zip(single, res):
            assert (a[0], a[1]) == (g, sh) and a[2] == d
        # the native pipeline (scl_detect_full_stream) and the stream front end on top of it
        for spl, depth in ((1, 1), (2, 2), (3, 2), (4, 2), (2, 4)):
            nn, sh, d = e.detect_full_stream(queries, 0, his, spl, depth)
            for a, b in zip(single, zip(nn.tolist(), sh.tolist(), d.tolist())):
                assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2]
        st = FullScanStream(e, depth=2, merge_every=3, scans_per_launch=2, native_chunk=3)
        for q, hi in zip(queries, his):
            st.submit(q, 0, hi)
        for a, (d, g, sh) in zip(single, st.drain()):
            assert (a[0], a[1]) == (g, sh) and a[2] == d
    finally:
        e.close()
