"""performInterLoopClosure end to end (DM.h:1272-1385) across two engines, the way two robots run it:

  robot b's keyframe -> makeDescriptors (DM.h:988-1025) -> global_descriptor message on the wire -> robot a's
  globalDescriptorHandler (DM.h:556-629) and vice versa; robot b: detectInterLoopClosureID -> getIndex mapping
  (DM.h:1280-1284, with the "no loop" key -1) -> geometric_verification request with the transformed keyframe cloud
  (DM.h:1328-1333) on the wire -> robot a's geometricVerificationService (DM.h:1189-1268) from its keyframe store ->
  response with poseBetween (DM.h:1249-1259) on the wire -> loop_info on robot b (DM.h:1359-1382).

Every message crosses as ROS 1 wire bytes through the codec of include/scl_messages.h; the relative pose that comes
back must be the true one although robot b's own pose estimates carry drift."""
import ctypes
import math

import numpy as np
import pytest

from scl_slam_amd import ScanContextEngine, SclError
from scl_slam_amd import messages as M

pytestmark = pytest.mark.gpu


def _world(seed=5):
    rs = np.random.RandomState(seed)
    pts = [np.stack([rs.uniform(-60, 160, 260000), rs.uniform(-70, 70, 260000), np.full(260000, -1.65)], axis=1)]
    for _ in range(140):                                        # boxes: four walls each
        cx, cy = rs.uniform(-50, 150), rs.uniform(-60, 60)
        w, h = rs.uniform(2, 7), rs.uniform(1.5, 9)
        n = 700
        u = rs.uniform(-w, w, n); z = rs.uniform(-1.65, -1.65 + h, n)
        pts += [np.stack([cx + u, np.full(n, cy - w), z], 1), np.stack([cx + u, np.full(n, cy + w), z], 1),
                np.stack([np.full(n, cx - w), cy + u, z], 1), np.stack([np.full(n, cx + w), cy + u, z], 1)]
    return np.concatenate(pts).astype(np.float64)


def _scan(world, x, y, yaw, rs, n_max=14000):
    """what a LiDAR at (x, y, yaw) sees within 55 m, in its own frame, as pcl::PointXYZI records (32 bytes)"""
    d = world[:, :2] - np.array([x, y])
    near = world[(d ** 2).sum(1) < 55.0 ** 2]
    if len(near) > n_max:
        near = near[rs.choice(len(near), n_max, replace=False)]
    c, s = math.cos(-yaw), math.sin(-yaw)
    local = np.empty((len(near), 8), np.float32)
    local[:, 4:] = 0
    dx, dy = near[:, 0] - x, near[:, 1] - y
    local[:, 0] = c * dx - s * dy; local[:, 1] = s * dx + c * dy; local[:, 2] = near[:, 2]; local[:, 3] = 0
    local[:, 4] = 1.0
    return local


def _pose_T(eng, p):
    return eng.pose_to_matrix(*[float(v) for v in p])


class Robot:
    """what distributedMapping keeps per robot, reduced to the loop-closure path"""
    def __init__(self, rid, exclude):
        self.id = rid
        self.eng = ScanContextEngine(num_exclude_recent=exclude, knn_exclude_eps=float(np.finfo(np.float32).eps))
        self.poses = []                                         # cloudKeyPoses6D of this robot (its own ESTIMATES)
        self.seq = 0

    def make_descriptors(self, cloud, pose_est):
        """makeDescriptors, DM.h:988-1025: filter + descriptor + append, keep the keyframe, return the message bytes"""
        index = len(self.poses)
        values, _ = self.eng.make_and_save_filtered(cloud, 0.4, self.id, index)
        self.eng.keyframe_put(self.id, index, cloud)            # keyFrameArray.push_back, DM.h:674
        self.poses.append(np.asarray(pose_est, np.float64))
        m = M.GlobalDescriptor()
        self.seq += 1
        m.header = M.Header(self.seq, M.Time(1000 + index, 0), b"", 0)
        m.index = index
        M.lib().scl_msg_transform_from_pose(*[float(v) for v in pose_est], ctypes.byref(m.curPose))
        if index > 0:
            M.lib().scl_msg_transform_from_pose(*[float(v) for v in self.poses[index - 1]], ctypes.byref(m.prePose))
        m.values = values.ctypes.data_as(ctypes.POINTER(ctypes.c_float)); m.n_values = values.size
        return M.encode(m)

    def on_global_descriptor(self, wire, sender):
        """globalDescriptorHandler, DM.h:556-629 (descriptor part)"""
        d = M.decode(M.GlobalDescriptor, wire)
        self.eng.save_from_wire(M.values_of(d), sender, d.index)

    def serve_geometric_verification(self, wire, search_num=2):
        """geometricVerificationService, DM.h:1189-1268, on the robot that owns keyPre"""
        rq = M.decode(M.GeometricVerificationRequest, wire)
        stride, off = ctypes.c_int(), ctypes.c_int()
        assert M.lib().scl_msg_cloud_xyz_layout(ctypes.byref(rq.featureCloud), ctypes.byref(stride), ctypes.byref(off)) == 0
        n = rq.featureCloud.width * rq.featureCloud.height
        cloud = np.frombuffer(ctypes.string_at(rq.featureCloud.data, rq.featureCloud.n_data), np.float32).reshape(n, stride.value // 4)
        window = [_pose_T(self.eng, self.poses[k]) if 0 <= k < len(self.poses) else np.eye(4, dtype=np.float32)
                  for k in range(rq.keyPre - search_num, rq.keyPre + search_num + 1)]
        T, ok, ns, nt, nc, ni = self.eng.geometric_verification_from_store(cloud, 0.2, rq.robotPre, rq.keyPre, search_num, window, 0.3,
                                                                          1000, 0.25, 0.35, 7)
        rs = M.GeometricVerificationResponse()
        rs.success = 1 if ok else 0
        if ok:                                                  # DM.h:1249-1259: tCorrect = tfSVD * tWrong, poseFrom.between(poseTo)
            bet, _ = self.eng.loop_pose_between(T, self.remote_pose_cur, self.poses[rq.keyPre])
            rs.poseBetween = M.Transform(M.Vector3(*bet[:3]), M.Quaternion(*bet[3:]))
        return M.encode(rs), (ns, nt, nc, ni)


def test_inter_robot_loop_closure_chain():
    rs = np.random.RandomState(1)
    world = _world()
    a, b = Robot(0, exclude=8), Robot(1, exclude=8)
    n_kf = 40
    true_a = [(2.0 * j, 0.0, 0.0, 0.0, 0.0, 0.0) for j in range(n_kf)]
    true_b = [(2.0 * j + 0.6, 0.9, 0.0, 0.0, 0.0, 0.45) for j in range(n_kf)]
    drift = np.array([0.10, -0.08, 0.0, 0.0, 0.0, 0.008])       # robot b's odometry error (its own frame estimate)
    # robot a maps the street; robot b hears every descriptor
    for j in range(n_kf):
        wire = a.make_descriptors(_scan(world, *true_a[j][:2], true_a[j][5], rs), true_a[j])
        b.on_global_descriptor(wire, sender=0)
    assert a.eng.get_size() == n_kf and b.eng.get_size() == n_kf
    assert b.eng.get_index(7) == (0, 7) and b.eng.find_key(0, 7) == 7 and b.eng.find_key(1, 0) == -1
    loops, rejected = [], 0
    for j in range(n_kf):
        cloud = _scan(world, *true_b[j][:2], true_b[j][5], rs)
        est = np.array(true_b[j]) + drift
        wire = b.make_descriptors(cloud, est)
        a.on_global_descriptor(wire, sender=1)
        inter_ptr = b.eng.get_size() - 1                        # interLoopPtr walks the database, DM.h:1280
        loop_id, yaw, dist = b.eng.detect_inter(inter_ptr)
        robot_cur, key_cur = b.eng.get_index(inter_ptr)         # DM.h:1281, 1283
        assert (robot_cur, key_cur) == (1, j)
        if loop_id < 0:                                         # DM.h:1287: the reference has already indexed its map with -1 here;
            with pytest.raises(SclError):                       # the engine reports the key as out of range instead
                b.eng.get_index(loop_id)
            continue
        robot_pre, key_pre = b.eng.get_index(loop_id)           # DM.h:1282, 1284
        if robot_pre == robot_cur:
            continue                                            # an intra-robot match: not this path's business
        # stage 2, DM.h:1318-1336: the keyframe cloud moved by robot b's pose estimate, sent to the robot that owns keyPre
        rq = M.GeometricVerificationRequest()
        rq.keyPre, rq.keyCur, rq.robotPre, rq.robotCur = key_pre, key_cur, robot_pre, robot_cur
        moved = b.eng.transform_cloud(cloud, _pose_T(b.eng, est))
        fields = (M.PointField * 4)()
        M.lib().scl_msg_cloud_from_xyzi(moved.ctypes.data_as(ctypes.c_void_p), moved.shape[0], ctypes.byref(rq.featureCloud), fields)
        a.remote_pose_cur = est                                 # the service reads cloudKeyPoses6D of robotCur (DM.h:1249), known from curPose
        resp_wire, (ns, nt, nc, ni) = a.serve_geometric_verification(M.encode(rq))
        resp = M.decode(M.GeometricVerificationResponse, resp_wire)
        if not resp.success:
            rejected += 1
            continue
        # DM.h:1359-1382: the loop_info the factor graph gets
        li = M.LoopInfo()
        li.robot0, li.robot1, li.index0, li.index1, li.noise = robot_cur, robot_pre, key_cur, key_pre, 999.0
        li.betPose = resp.poseBetween
        li2 = M.decode(M.LoopInfo, M.encode(li))
        loops.append((li2.index0, li2.index1, li2.betPose))
    assert len(loops) >= 12 and rejected <= len(loops) // 2, (len(loops), rejected)
    for key_cur, key_pre, bet in loops:
        assert abs(key_pre - key_cur) <= 2                      # the same stretch of the street
        # truth: pose_b_true^-1 * pose_a_true -- independent of robot b's drift
        xb, yb, yawb = true_b[key_cur][0], true_b[key_cur][1], true_b[key_cur][5]
        xa, ya = true_a[key_pre][0], true_a[key_pre][1]
        c, s = math.cos(-yawb), math.sin(-yawb)
        tx, ty = c * (xa - xb) - s * (ya - yb), s * (xa - xb) + c * (ya - yb)
        out = [ctypes.c_double() for _ in range(6)]
        assert M.lib().scl_msg_transform_to_pose(ctypes.byref(bet), *[ctypes.byref(o) for o in out]) == 0
        x, y, z, roll, pitch, yaw = [o.value for o in out]
        assert abs(x - tx) < 0.08 and abs(y - ty) < 0.08 and abs(z) < 0.05, (key_cur, key_pre, x, y, tx, ty)
        assert abs(yaw + yawb) < 0.01 and abs(roll) < 0.01 and abs(pitch) < 0.01
    a.eng.close(); b.eng.close()


def test_database_dump_and_load_reproduce_detections(tmp_path):
    from scl_slam_amd.synth import synth_descriptors
    R, S, n = 64, 120, 900
    descs = synth_descriptors(n, R, S, seed=21, revisit_frac=0.05)
    robots = (np.arange(n) % 3).astype(np.int8); indexs = (np.arange(n) // 3).astype(np.int32)
    one = ScanContextEngine(num_ring=R, num_sector=S, initial_capacity=64)
    one.save_bulk(descs, robots, indexs)
    path = str(tmp_path / "db.scl")
    one.db_dump(path)
    assert one.find_key(2, 10) == 32 and one.find_key(1, 10 ** 6) == -1
    assert np.array_equal(one.get_descriptors(5, 40), descs[5:45])
    fresh = ScanContextEngine(num_ring=R, num_sector=S, initial_capacity=16)
    sharded = ScanContextEngine(num_ring=R, num_sector=S, initial_capacity=16, devices=[0, 0, 0], exchange=1)
    assert fresh.db_load(path) == n and sharded.db_load(path) == n
    for eng in (fresh, sharded):
        assert eng.get_size() == n
        for key in (0, 1, 457, n - 1):
            assert eng.get_index(key) == one.get_index(key)
            assert np.array_equal(eng.get_ringkey(key).view(np.uint32), one.get_ringkey(key).view(np.uint32))
        for cur in (n - 1, n - 40, 500):
            assert eng.detect_intra(cur) == one.detect_intra(cur)
            assert eng.detect_full(cur) == one.detect_full(cur)
    with pytest.raises(SclError):
        ScanContextEngine(num_ring=20, num_sector=60).db_load(path)        # another grid
    open(str(tmp_path / "bad.scl"), "wb").write(b"not a dump")
    with pytest.raises(SclError):
        fresh.db_load(str(tmp_path / "bad.scl"))
    # a header is not trusted.  A count the file cannot hold (here 2^31 - 1: tens of GB of index map if it were
    # believed) and a truncated file are refused before anything is sized by them, and nothing is appended
    raw = open(path, "rb").read()
    import struct
    hostile = raw[:8] + struct.pack("<iiii", 1, R, S, 2 ** 31 - 1) + raw[24:]
    open(str(tmp_path / "hostile.scl"), "wb").write(hostile)
    open(str(tmp_path / "short.scl"), "wb").write(raw[:len(raw) - 4096])
    for name in ("hostile.scl", "short.scl"):
        with pytest.raises(SclError):
            fresh.db_load(str(tmp_path / name))
    assert fresh.get_size() == n
    # the inter-robot path's tree state (D.h:1691-1703: rebuilt every TREE_MAKING_PERIOD_ calls) travels with the dump: an
    # engine that has answered some detectInterLoopClosureID calls and its reloaded copy give the same answers from there on
    src = ScanContextEngine(num_ring=R, num_sector=S, initial_capacity=64)
    src.save_bulk(descs[:700], robots[:700], indexs[:700])
    for cur in (650, 651, 652):
        src.detect_inter(cur)                                          # the tree now covers [0, 550), counter = 3
    src.save_bulk(descs[700:], robots[700:], indexs[700:])
    src.db_dump(str(tmp_path / "mid.scl"))
    twin = ScanContextEngine(num_ring=R, num_sector=S, initial_capacity=16)
    assert twin.db_load(str(tmp_path / "mid.scl")) == n
    for cur in range(n - 1, n - 13, -1):                               # crosses the next rebuild (period 10) on both
        assert twin.detect_inter(cur) == src.detect_inter(cur)
    src.close(); twin.close()
    one.close(); fresh.close(); sharded.close()


def test_icp_tail_pose_algebra():
    """DM.h:1130-1141 against the same algebra in numpy double (float stages where the reference uses Affine3f)"""
    e = ScanContextEngine()
    rs = np.random.RandomState(9)
    for _ in range(50):
        pc = np.concatenate([rs.uniform(-30, 30, 3), rs.uniform(-0.3, 0.3, 2), rs.uniform(-3, 3, 1)]).astype(np.float32)
        pp = np.concatenate([rs.uniform(-30, 30, 3), rs.uniform(-0.3, 0.3, 2), rs.uniform(-3, 3, 1)]).astype(np.float32)
        Ticp = e.pose_to_matrix(*[float(v) for v in np.concatenate([rs.uniform(-0.5, 0.5, 3), rs.uniform(-0.05, 0.05, 3)])])
        bet, rpy = e.loop_pose_between(Ticp, pc, pp)
        Tw = e.pose_to_matrix(*[float(v) for v in pc]).astype(np.float64)
        Tc = Ticp.astype(np.float64) @ Tw
        Tp = e.pose_to_matrix(*[float(v) for v in pp]).astype(np.float64)
        want = np.linalg.inv(Tc) @ Tp
        assert np.abs(bet[:3] - want[:3, 3]).max() < 2e-4
        qx, qy, qz, qw = bet[3:]
        Rq = np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qw * qz), 2 * (qx * qz + qw * qy)],
                       [2 * (qx * qy + qw * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qw * qx)],
                       [2 * (qx * qz - qw * qy), 2 * (qy * qz + qw * qx), 1 - 2 * (qx * qx + qy * qy)]])
        assert np.abs(Rq - want[:3, :3]).max() < 2e-6 and qw >= 0
        x, y, z, roll, pitch, yaw = e.matrix_to_pose(Tw.astype(np.float32))
        assert np.allclose([x, y, z, roll, pitch, yaw], pc, atol=2e-5)
    e.close()
