"""K3 parity: GPU makeScancontext vs the CPU checker, bit for bit."""
import numpy as np
import pytest

import oracle_binding as ob
from scl_slam_amd import ScanContextEngine
from scl_slam_amd.synth import synth_scan

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("R,S,n", [(20, 60, 15000), (64, 120, 120000), (80, 180, 240000), (20, 60, 1), (20, 60, 0)])
def test_descriptor_matches_oracle(R, S, n):
    cloud = synth_scan(n, seed=R + n)
    eng = ScanContextEngine(num_ring=R, num_sector=S)
    cfg = ob.make_config(R=R, S=S)
    v_gpu = eng.make_descriptor(cloud)
    v_cpu = ob.make_scancontext(cfg, cloud)
    assert np.array_equal(v_gpu, v_cpu)
    eng.close()


def test_edge_points_and_strides():
    R, S = 20, 60
    pts = np.array([[0, 0, 1.0], [80.0, 0, 2.0], [80.00001, 0, 9.0], [10, 10, -5.0], [10, 10, -2000.0],
                    [0, 5, 1.0], [-5, 0, 1.0], [0, -5, 1.0], [np.nan, 1, 1], [1, 1, np.nan], [np.inf, 0, 3],
                    [1e-30, 1e-30, 0.5], [-1e-3, -1e-3, 0.25], [79.999, -0.0001, 4.0]], dtype=np.float32)
    cfg = ob.make_config(R=R, S=S)
    for stride in (3, 4, 8):
        cloud = np.zeros((len(pts), stride), np.float32); cloud[:, :3] = pts
        eng = ScanContextEngine(num_ring=R, num_sector=S)
        assert np.array_equal(eng.make_descriptor(cloud), ob.make_scancontext(cfg, cloud))
        eng.close()


def test_sector_edges_bitwise():
    # points placed on/around sector boundaries: any atan disagreement would flip a bin
    R, S = 64, 120
    rs = np.random.RandomState(1)
    k = np.arange(0, 360, 3.0)
    ang = np.deg2rad(np.concatenate([k, k + 1e-5, k - 1e-5, rs.uniform(0, 360, 20000)]))
    rad = rs.uniform(0.1, 79.9, size=ang.size)
    cloud = np.zeros((ang.size, 8), np.float32)
    cloud[:, 0] = rad * np.cos(ang); cloud[:, 1] = rad * np.sin(ang); cloud[:, 2] = rs.uniform(-1, 10, ang.size)
    eng = ScanContextEngine(num_ring=R, num_sector=S)
    assert np.array_equal(eng.make_descriptor(cloud), ob.make_scancontext(ob.make_config(R=R, S=S), cloud))
    eng.close()


def test_make_and_save_then_detect_keys():
    R, S = 20, 60
    eng = ScanContextEngine(num_ring=R, num_sector=S)
    db = ob.OracleDB(ob.make_config(R=R, S=S))
    for i in range(5):
        cloud = synth_scan(8000, seed=i)
        v_gpu = eng.make_and_save(cloud, robot=1, index=10 + i)
        v_cpu = db.make_and_save(cloud, robot=1, index=10 + i)
        assert np.array_equal(v_gpu, v_cpu)
        assert np.array_equal(eng.get_ringkey(i).view(np.uint32), db.ringkey(i).view(np.uint32))
    assert eng.get_size() == 5 and eng.get_index(3) == (1, 13) == db.get_index(3)
    eng.close()


def test_filtered_make_and_save_equals_two_calls():
    """makeDescriptors in one call (voxel filter + descriptor on the device) == scl_voxel_grid followed by
    scl_make_and_save: same wire values, same keys, same database entry; also the leaf-too-small case in which
    PCL returns the input unchanged."""
    from scl_slam_amd.synth import synth_scan
    R, S = 64, 120
    for n, leaf in ((60000, 0.4), (240000, 0.2), (5000, 0.001), (0, 0.4)):
        cloud = synth_scan(n, seed=7 + n) if n else np.zeros((0, 8), np.float32)
        e1 = ScanContextEngine(num_ring=R, num_sector=S); e2 = ScanContextEngine(num_ring=R, num_sector=S)
        v1 = e1.make_and_save(e1.voxel_grid(cloud, leaf) if n else cloud, 0, 0)
        v2, m = e2.make_and_save_filtered(cloud, leaf, 0, 0)
        assert np.array_equal(v1.view(np.uint32), v2.view(np.uint32))
        assert m == (e1.voxel_grid(cloud, leaf).shape[0] if n else 0)
        assert np.array_equal(e1.get_ringkey(0).view(np.uint32), e2.get_ringkey(0).view(np.uint32))
        assert np.array_equal(e1.get_sectorkey(0).view(np.uint64), e2.get_sectorkey(0).view(np.uint64))
        e1.close(); e2.close()
