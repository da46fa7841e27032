"""The verification path's own sort and prefix sums (csrc/device_sort.hip; they replaced hipCUB's) against numpy's stable sort and cumsum,
through the C ABI's test hooks.  What is sorted there: (voxel, point) pairs of PCL's VoxelGrid (DM.h:1183-1185, 1200-1201), one cloud or the
26 submaps of a query as 26 segments of one buffer, and the Hilbert keys of an ICP batch's sources."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from scl_slam_amd import ScanContextEngine
    e = ScanContextEngine(num_ring=20, num_sector=60)
    yield e
    e.close()


def _expect(keys, bits, seg=None):
    mask = (1 << bits) - 1
    k = (keys.astype(np.uint64) & np.uint64(mask))
    if seg is None:
        return np.argsort(k, kind="stable")
    order = np.empty(keys.size, dtype=np.int64)
    for s in range(len(seg) - 1):
        a, b = seg[s], seg[s + 1]
        order[a:b] = a + np.argsort(k[a:b], kind="stable")
    return order


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 1000, 1024, 1025, 4097, 100_000, 1_300_000])
@pytest.mark.parametrize("bits", [32, 30, 12, 0])
def test_sort_u32_pairs_is_numpys_stable_sort(eng, n, bits):
    rs = np.random.RandomState(n * 37 + bits)
    keys = rs.randint(0, 2 ** 32, size=n, dtype=np.uint64).astype(np.uint32)
    if n > 100:
        keys[rs.randint(0, n, size=n // 3)] = keys[0]                    # many equal keys: the order among them is the input's
    vals = np.arange(n, dtype=np.uint32)
    ko, vo = eng.selftest_sort_pairs(keys, vals, bits)
    order = _expect(keys, bits)
    assert np.array_equal(vo, vals[order])
    assert np.array_equal(ko, keys[order])                                # the whole key travels, whatever bits were looked at


@pytest.mark.parametrize("few", [2, 7, 300])
def test_sort_with_a_handful_of_distinct_keys_is_stable(eng, few):
    n = 200_000
    rs = np.random.RandomState(few)
    keys = (rs.randint(0, few, size=n).astype(np.uint32) * np.uint32(0x01010101))
    vals = rs.randint(0, 2 ** 31, size=n).astype(np.uint32)
    ko, vo = eng.selftest_sort_pairs(keys, vals, 32)
    order = _expect(keys, 32)
    assert np.array_equal(vo, vals[order]) and np.array_equal(ko, keys[order])


@pytest.mark.parametrize("bits", [37, 40, 64, 33])
def test_sort_u64_pairs_single_segment(eng, bits):
    n = 250_000
    rs = np.random.RandomState(bits)
    keys = rs.randint(0, 2 ** 63, size=n, dtype=np.uint64) * np.uint64(2) + rs.randint(0, 2, size=n).astype(np.uint64)
    keys[::5] = keys[0]
    vals = np.arange(n, dtype=np.uint32)
    ko, vo = eng.selftest_sort_pairs(keys, vals, bits)
    order = _expect(keys, bits) if bits < 64 else np.argsort(keys, kind="stable")
    assert np.array_equal(vo, vals[order]) and np.array_equal(ko, keys[order])


@pytest.mark.parametrize("sizes", [
    [100_000] * 26,                                                        # configs[2]: the scan's submap and its 25 candidates'
    [0, 5, 0, 4096, 4097, 1, 0, 70_000, 1023, 1024, 1025, 0],              # empty and ragged segments, tile edges
    [3],
    [50_000] * 64,
])
def test_sort_u64_segments_each_on_its_own(eng, sizes):
    seg = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    n = int(seg[-1])
    rs = np.random.RandomState(len(sizes))
    vox = rs.randint(0, 2 ** 32, size=n, dtype=np.uint64)
    vox[rs.randint(0, n, size=n // 4)] = 0xFFFFFFFF                        # (the "no voxel" mark of voxel.hip)
    job = np.repeat(np.arange(len(sizes), dtype=np.uint64), sizes)
    keys = (job << np.uint64(32)) | vox
    vals = np.arange(n, dtype=np.uint32)
    ko, vo = eng.selftest_sort_pairs(keys, vals, 32, segment_offsets=seg)
    order = _expect(keys, 32, seg)
    assert np.array_equal(vo, vals[order]) and np.array_equal(ko, keys[order])
    assert np.array_equal(ko, np.sort(keys, kind="stable"))                # job-major input: the same as one sort on (job, voxel)


@pytest.mark.parametrize("n", [1, 3, 4, 5, 1023, 1024, 1025, 4096, 99_999, 912_673 + 1, 2_600_000])
@pytest.mark.parametrize("inclusive", [False, True])
def test_prefix_sum_is_numpys_cumsum(eng, n, inclusive):
    rs = np.random.RandomState(n)
    v = rs.randint(-3, 40, size=n).astype(np.int32)
    out = eng.selftest_prefix_sum(v, inclusive)
    c = np.cumsum(v.astype(np.int64))
    want = c if inclusive else np.concatenate([[0], c[:-1]])
    assert np.array_equal(out, want.astype(np.int32))


def test_sort_u64_one_segment_with_a_common_high_word_takes_the_record_passes(eng):
    """keys that share their high word within every segment go through 8-byte records between the first and the last pass (the hook says
    so to the sort when it sees it): one segment of 1.3 M pairs (big tiles) and of 5 000 (small tiles), bits 32 and 20"""
    for n, bits in ((1_300_000, 32), (5000, 32), (70_000, 20)):
        rs = np.random.RandomState(n + bits)
        keys = (np.uint64(7) << np.uint64(32)) | rs.randint(0, 2 ** 32, size=n, dtype=np.uint64)
        keys[::3] = keys[0]
        vals = rs.randint(0, 2 ** 32, size=n, dtype=np.uint64).astype(np.uint32)
        ko, vo = eng.selftest_sort_pairs(keys, vals, bits, segment_offsets=np.array([0, n], dtype=np.int32))
        order = _expect(keys, bits)
        assert np.array_equal(vo, vals[order]) and np.array_equal(ko, keys[order])
