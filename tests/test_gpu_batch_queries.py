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


def test_long_stream_crosses_chunks_with_empty_and_ragged_ranges():
    """scl_detect_full_stream on the screened grid works in chunks of up to 64 scans (a whole number of launches: 64, 60 at 12 or 10 scans per launch,
    63 at three) on alternating halves of its buffers; every
    launch carries the alignment of the launch behind it, the chunk's last one that of the next chunk's first.  Scans with
    an empty range, ranges that differ per scan, a last chunk of one scan and every scans-per-launch setting must give what
    each scan's own pass gives, bit for bit; the array form of the Python stream front end as well."""
    R, S, n = 64, 120, 1200
    descs = synth_descriptors(n, R, S, seed=77, revisit_frac=0.05)
    e = ScanContextEngine(num_ring=R, num_sector=S, num_candidates=3, num_exclude_recent=50, initial_capacity=n + 8)
    try:
        e.save_bulk(descs)
        rs = np.random.RandomState(5)
        for count in (150, 129, 65):
            qs = rs.randint(0, n, count).astype(np.int32)
            lo = rs.randint(0, 40, count).astype(np.int32)
            hi = np.maximum(qs - 50, 0).astype(np.int32)
            hi[::11] = lo[::11]                                        # empty ranges, also as the first or last scan of a chunk
            for k in (59, 60, 62, 63, 64): hi[k] = lo[k]
            single = [e.detect_full_range(int(q), int(a), int(b)) if b > a else (-1, 0, 1e7) for q, a, b in zip(qs, lo, hi)]
            for spl in (16, 12, 10, 4, 3, 1):
                nn, sh, d = e.detect_full_stream(qs, lo, hi, spl, 2)
                for i, a in enumerate(single):
                    assert (a[0], a[1]) == (nn[i], sh[i]) and np.float64(a[2]).view(np.uint64) == d[i].view(np.uint64), (count, spl, i)
            st = FullScanStream(e, depth=2, scans_per_launch=4, native_chunk=40)
            st.submit_many(qs, lo, hi)
            for i, (a, (dd, g, s_)) in enumerate(zip(single, st.drain())):
                assert (a[0], a[1]) == (g, s_) and a[2] == dd, (count, i)
    finally:
        e.close()


@pytest.mark.parametrize("R,S,n", [(64, 120, 600), (80, 180, 400), (20, 60, 500)])
def test_stream_calls_of_one_to_four_scans_equal_the_blocking_calls(R, S, n):
    """scl_detect_full_stream hands up to four scans to the blocking form's launch group (on the screened grids); whatever the path, the
    winners are those of scl_detect_full_range -- ragged and empty ranges included."""
    descs = synth_descriptors(n, R, S, seed=77, revisit_frac=0.05)
    eng = ScanContextEngine(num_ring=R, num_sector=S, initial_capacity=n)
    eng.save_bulk(descs)
    rs = np.random.RandomState(4)
    for m in (1, 2, 3, 4, 5):
        for rep in range(3):
            qs = rs.randint(0, n, size=m).astype(np.int32)
            los = rs.randint(0, n // 2, size=m).astype(np.int32)
            his = (los + rs.randint(0, n // 2, size=m)).astype(np.int32)
            if rep == 2: his[0] = los[0]                                   # an empty range among them
            nn, sh, dd = eng.detect_full_stream(qs, los, his, 16, 2)
            for i in range(m):
                one = eng.detect_full_range(int(qs[i]), int(los[i]), int(his[i]))
                assert (nn[i], sh[i]) == (one[0], one[1]) and dd[i].view(np.uint64) == np.float64(one[2]).view(np.uint64), (m, rep, i)
    eng.close()
