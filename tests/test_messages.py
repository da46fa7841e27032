"""ROS surface of the path kept intact: plain-struct mirrors of global_descriptor.msg:2-8,
loop_info.msg:2-9, geometric_verification.srv:1-8 and their ROS 1 wire encoding -- checked against buffers assembled
by hand with `struct` (little-endian, uint32 length prefixes), round trips, truncated input, and the interface files."""
import ctypes
import math
import os
import struct

import numpy as np
import pytest

from scl_slam_amd import messages as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tf(t, q):
    return M.Transform(M.Vector3(*t), M.Quaternion(*q))


def _tf_bytes(t, q):
    return struct.pack("<7d", *t, *q)


def _hdr_bytes(seq, sec, nsec, frame):
    return struct.pack("<III", seq, sec, nsec) + struct.pack("<I", len(frame)) + frame


def test_struct_layouts_follow_the_message_definitions():
    # field order = declaration order of the .msg / .srv files; geometry_msgs/Transform is 7 float64
    assert [f[0] for f in M.GlobalDescriptor._fields_] == ["header", "index", "prePose", "curPose", "values", "n_values"]
    assert [f[0] for f in M.LoopInfo._fields_] == ["header", "robot0", "robot1", "index0", "index1", "noise", "betPose"]
    assert [f[0] for f in M.GeometricVerificationRequest._fields_][:5] == ["keyPre", "keyCur", "robotPre", "robotCur", "featureCloud"]
    assert [f[0] for f in M.GeometricVerificationResponse._fields_] == ["success", "poseBetween"]
    assert ctypes.sizeof(M.Transform) == 56 and ctypes.sizeof(M.Time) == 8
    assert M.LoopInfo.noise.size == 4 and M.LoopInfo.robot0.size == 4
    # the interface files restate the reference's definitions field for field
    def fields(path):
        out = []
        for line in open(os.path.join(ROOT, "ros", path)):
            line = line.split("#")[0].strip()
            if line:
                out.append(tuple(line.split()))
        return out
    assert fields("msg/global_descriptor.msg") == [("Header", "header"), ("int32", "index"), ("geometry_msgs/Transform", "prePose"),
                                                   ("geometry_msgs/Transform", "curPose"), ("float32[]", "values")]
    assert fields("msg/loop_info.msg") == [("Header", "header"), ("int32", "robot0"), ("int32", "robot1"), ("int32", "index0"), ("int32", "index1"),
                                           ("float32", "noise"), ("geometry_msgs/Transform", "betPose")]
    assert fields("srv/geometric_verification.srv") == [("int32", "keyPre"), ("int32", "keyCur"), ("int32", "robotPre"), ("int32", "robotCur"),
                                                        ("sensor_msgs/PointCloud2", "featureCloud"), ("---",), ("bool", "success"),
                                                        ("geometry_msgs/Transform", "poseBetween")]


def test_global_descriptor_wire_bytes_and_round_trip():
    vals = np.arange(20 * 60, dtype=np.float32) * 0.25
    m = M.GlobalDescriptor()
    m.header = M.Header(7, M.Time(1700000000, 123456789), b"robot_a/odom", 12)
    m.index = 41
    m.prePose = _tf((1.0, 2.0, 3.0), (0.0, 0.0, 0.0, 1.0)); m.curPose = _tf((1.5, 2.5, 3.5), (0.1, 0.2, 0.3, 0.9))
    m.values = vals.ctypes.data_as(ctypes.POINTER(ctypes.c_float)); m.n_values = vals.size
    want = (_hdr_bytes(7, 1700000000, 123456789, b"robot_a/odom") + struct.pack("<i", 41) + _tf_bytes((1.0, 2.0, 3.0), (0.0, 0.0, 0.0, 1.0)) +
            _tf_bytes((1.5, 2.5, 3.5), (0.1, 0.2, 0.3, 0.9)) + struct.pack("<I", vals.size) + vals.tobytes())
    got = M.encode(m)
    assert got == want
    d = M.decode(M.GlobalDescriptor, got)
    assert (d.header.seq, d.header.stamp.sec, d.header.stamp.nsec, d.index) == (7, 1700000000, 123456789, 41)
    assert ctypes.string_at(d.header.frame_id, d.header.frame_id_len) == b"robot_a/odom"
    assert (d.curPose.rotation.w, d.prePose.translation.z) == (0.9, 3.0)
    assert np.array_equal(M.values_of(d), vals)
    for cut in (0, 5, len(got) - 1):                            # truncated input is an error, not a crash
        with pytest.raises(ValueError):
            M.decode(M.GlobalDescriptor, got[:cut])
    with pytest.raises(ValueError):
        M.decode(M.GlobalDescriptor, got + b"\0")                # trailing bytes too
    # a hostile length prefix must not read out of bounds
    bad = bytearray(got); off = len(_hdr_bytes(7, 0, 0, b"robot_a/odom")) + 4 + 112
    bad[off:off + 4] = struct.pack("<I", 0xFFFFFFF0)
    with pytest.raises(ValueError):
        M.decode(M.GlobalDescriptor, bytes(bad))


def test_loop_info_and_service_messages():
    li = M.LoopInfo()
    li.header = M.Header(1, M.Time(10, 20), None, 0)
    li.robot0, li.robot1, li.index0, li.index1, li.noise = 0, 2, 310, 17, 0.125
    li.betPose = _tf((0.5, -0.25, 0.0), (0.0, 0.0, math.sin(0.2), math.cos(0.2)))
    want = _hdr_bytes(1, 10, 20, b"") + struct.pack("<iiiif", 0, 2, 310, 17, 0.125) + _tf_bytes((0.5, -0.25, 0.0), (0.0, 0.0, math.sin(0.2), math.cos(0.2)))
    assert M.encode(li) == want
    d = M.decode(M.LoopInfo, want)
    assert (d.robot1, d.index0, d.index1, d.noise) == (2, 310, 17, 0.125) and d.betPose.rotation.z == math.sin(0.2)

    rs = M.GeometricVerificationResponse(1, _tf((1, 2, 3), (0, 0, 0, 1)))
    assert M.encode(rs) == b"\x01" + _tf_bytes((1, 2, 3), (0, 0, 0, 1))
    assert M.decode(M.GeometricVerificationResponse, M.encode(rs)).success == 1

    pts = np.zeros((5, 8), np.float32); pts[:, 0] = np.arange(5); pts[:, 1] = 2.0; pts[:, 2] = -1.0; pts[:, 4] = 9.0
    rq = M.GeometricVerificationRequest()
    rq.keyPre, rq.keyCur, rq.robotPre, rq.robotCur = 12, 340, 1, 0
    fields = (M.PointField * 4)()
    assert M.lib().scl_msg_cloud_from_xyzi(pts.ctypes.data_as(ctypes.c_void_p), 5, ctypes.byref(rq.featureCloud), fields) == 0
    pf = b"".join(struct.pack("<I", len(n)) + n + struct.pack("<IBI", o, 7, 1) for n, o in ((b"x", 0), (b"y", 4), (b"z", 8), (b"intensity", 16)))
    want = (struct.pack("<iiii", 12, 340, 1, 0) + _hdr_bytes(0, 0, 0, b"") + struct.pack("<II", 1, 5) + struct.pack("<I", 4) + pf +
            struct.pack("<BII", 0, 32, 160) + struct.pack("<I", 160) + pts.tobytes() + b"\x01")
    got = M.encode(rq)
    assert got == want
    d = M.decode(M.GeometricVerificationRequest, got)
    assert (d.keyPre, d.keyCur, d.robotPre, d.robotCur) == (12, 340, 1, 0) and d.featureCloud.width == 5 and d.featureCloud.n_fields == 4
    stride, off = ctypes.c_int(), ctypes.c_int()
    assert M.lib().scl_msg_cloud_xyz_layout(ctypes.byref(d.featureCloud), ctypes.byref(stride), ctypes.byref(off)) == 0
    assert (stride.value, off.value) == (32, 0)
    back = np.frombuffer(ctypes.string_at(d.featureCloud.data, d.featureCloud.n_data), np.float32).reshape(5, 8)
    assert np.array_equal(back, pts)


def test_cloud_layout_of_a_peer_message_is_checked_not_trusted():
    """a decoded PointCloud2 comes from another robot.  Layouts whose x / y / z reads would leave the record or the
    buffer -- or that are ambiguous -- are refused; the reference's own layout (pcl::toROSMsg of PointXYZI, DM.h:1329) passes."""
    L = M.lib()
    pts = np.zeros((5, 8), np.float32)
    buf = (ctypes.c_uint8 * pts.nbytes).from_buffer_copy(pts.tobytes())

    def layout(fields, point_step=32, row_step=None, n_data=None, width=5, height=1, bigendian=0):
        fs = (M.PointField * max(1, len(fields)))()
        for i, (name, off, dt, cnt) in enumerate(fields):
            fs[i].name = name; fs[i].name_len = len(name); fs[i].offset = off; fs[i].datatype = dt; fs[i].count = cnt
        c = M.Cloud()
        c.height, c.width, c.fields, c.n_fields = height, width, fs, len(fields)
        c.is_bigendian, c.point_step = bigendian, point_step
        c.row_step = point_step * width if row_step is None else row_step
        c.data = ctypes.cast(buf, ctypes.POINTER(ctypes.c_uint8)); c.n_data = pts.nbytes if n_data is None else n_data
        stride, off = ctypes.c_int(-1), ctypes.c_int(-1)
        return L.scl_msg_cloud_xyz_layout(ctypes.byref(c), ctypes.byref(stride), ctypes.byref(off)), stride.value, off.value

    xyz = [(b"x", 0, 7, 1), (b"y", 4, 7, 1), (b"z", 8, 7, 1)]
    assert layout(xyz + [(b"intensity", 16, 7, 1)]) == (0, 32, 0)
    assert layout(xyz, point_step=12, n_data=60) == (0, 12, 0)
    OK, INVALID, UNSUPPORTED = 0, -1, None
    rc_unsupported = layout([(b"x", 0, 7, 1), (b"y", 8, 7, 1), (b"z", 4, 7, 1)])[0]          # not consecutive
    assert rc_unsupported != 0
    bad = {
        "z runs past the record": layout([(b"x", 24, 7, 1), (b"y", 28, 7, 1), (b"z", 32, 7, 1)]),
        "x late in the record, no slack behind the last record": layout([(b"x", 16, 7, 1), (b"y", 20, 7, 1), (b"z", 24, 7, 1)]),
        "duplicate x": layout(xyz + [(b"x", 16, 7, 1)]),
        "count 2": layout([(b"x", 0, 7, 2), (b"y", 4, 7, 1), (b"z", 8, 7, 1)]),
        "float64 fields": layout([(b"x", 0, 8, 1), (b"y", 4, 8, 1), (b"z", 8, 8, 1)]),
        "big endian": layout(xyz, bigendian=1),
        "padded rows": layout(xyz, row_step=200),
        "step not a multiple of four": layout(xyz, point_step=14, n_data=70),
        "missing z": layout(xyz[:2]),
    }
    for why, (rc, _, _) in bad.items():
        assert rc == rc_unsupported, why
    assert layout(xyz, n_data=159)[0] not in (0, rc_unsupported)                              # data shorter than the records it declares
    assert layout(xyz, width=1 << 20, height=1 << 12)[0] != 0
    # x late in the record is fine when the buffer holds the bytes a whole-record read from data + offset touches
    assert layout([(b"x", 16, 7, 1), (b"y", 20, 7, 1), (b"z", 24, 7, 1)], width=4) == (0, 32, 16)


def test_transform_pose_conversions():
    rs = np.random.RandomState(3)
    for _ in range(200):
        x, y, z = rs.uniform(-50, 50, 3)
        roll, yaw = rs.uniform(-math.pi, math.pi, 2); pitch = rs.uniform(-1.5, 1.5)
        t = M.Transform()
        assert M.lib().scl_msg_transform_from_pose(x, y, z, roll, pitch, yaw, ctypes.byref(t)) == 0
        q = np.array([t.rotation.x, t.rotation.y, t.rotation.z, t.rotation.w])
        assert abs(np.linalg.norm(q) - 1) < 1e-14
        # Rz(yaw) Ry(pitch) Rx(roll) (pcl::getTransformation / tf::createQuaternionMsgFromRollPitchYaw)
        cr, sr, cp, sp, cy, sy = math.cos(roll), math.sin(roll), math.cos(pitch), math.sin(pitch), math.cos(yaw), math.sin(yaw)
        Rm = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]]) @ np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]]) @ np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
        qx, qy, qz, qw = q
        Rq = np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qw * qz), 2 * (qx * qz + qw * qy)],
                       [2 * (qx * qy + qw * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qw * qx)],
                       [2 * (qx * qz - qw * qy), 2 * (qy * qz + qw * qx), 1 - 2 * (qx * qx + qy * qy)]])
        assert np.abs(Rm - Rq).max() < 1e-14
        out = [ctypes.c_double() for _ in range(6)]
        assert M.lib().scl_msg_transform_to_pose(ctypes.byref(t), *[ctypes.byref(o) for o in out]) == 0
        got = [o.value for o in out]
        assert np.allclose(got, [x, y, z, roll, pitch, yaw], atol=1e-12)
