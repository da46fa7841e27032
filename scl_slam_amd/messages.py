"""ctypes mirrors of include/scl_messages.h: the ROS interface of the loop-closure path (global_descriptor, loop_info,
geometric_verification) as plain structs + the ROS 1 wire codec of the native library.  No ROS needed."""
import ctypes
from ctypes import POINTER, byref, c_char_p, c_double, c_float, c_int, c_int32, c_size_t, c_uint8, c_uint32, c_void_p

import numpy as np

from ._native import load_library


class Time(ctypes.Structure):
    _fields_ = [("sec", c_uint32), ("nsec", c_uint32)]


class Header(ctypes.Structure):
    _fields_ = [("seq", c_uint32), ("stamp", Time), ("frame_id", c_char_p), ("frame_id_len", c_uint32)]


class Vector3(ctypes.Structure):
    _fields_ = [("x", c_double), ("y", c_double), ("z", c_double)]


class Quaternion(ctypes.Structure):
    _fields_ = [("x", c_double), ("y", c_double), ("z", c_double), ("w", c_double)]


class Transform(ctypes.Structure):
    _fields_ = [("translation", Vector3), ("rotation", Quaternion)]


class GlobalDescriptor(ctypes.Structure):
    """dlc_slam/global_descriptor (msg/global_descriptor.msg:2-8)"""
    _fields_ = [("header", Header), ("index", c_int32), ("prePose", Transform), ("curPose", Transform),
                ("values", POINTER(c_float)), ("n_values", c_uint32)]


class LoopInfo(ctypes.Structure):
    """dlc_slam/loop_info (msg/loop_info.msg:2-9)"""
    _fields_ = [("header", Header), ("robot0", c_int32), ("robot1", c_int32), ("index0", c_int32), ("index1", c_int32),
                ("noise", c_float), ("betPose", Transform)]


class PointField(ctypes.Structure):
    _fields_ = [("name", c_char_p), ("name_len", c_uint32), ("offset", c_uint32), ("datatype", c_uint8), ("count", c_uint32)]


class Cloud(ctypes.Structure):
    _fields_ = [("header", Header), ("height", c_uint32), ("width", c_uint32), ("fields", POINTER(PointField)), ("n_fields", c_uint32),
                ("is_bigendian", c_uint8), ("point_step", c_uint32), ("row_step", c_uint32), ("data", POINTER(c_uint8)), ("n_data", c_uint32),
                ("is_dense", c_uint8)]


class GeometricVerificationRequest(ctypes.Structure):
    """dlc_slam/geometric_verification request (srv/geometric_verification.srv:1-5)"""
    _fields_ = [("keyPre", c_int32), ("keyCur", c_int32), ("robotPre", c_int32), ("robotCur", c_int32), ("featureCloud", Cloud),
                ("field_store", PointField * 16)]


class GeometricVerificationResponse(ctypes.Structure):
    """dlc_slam/geometric_verification response (srv/geometric_verification.srv:7-8)"""
    _fields_ = [("success", c_uint8), ("poseBetween", Transform)]


_bound = None


def lib():
    global _bound
    if _bound is not None:
        return _bound
    L = load_library()
    u8p, szp = POINTER(c_uint8), POINTER(c_size_t)
    for name, typ in (("global_descriptor", GlobalDescriptor), ("loop_info", LoopInfo),
                      ("geometric_verification_request", GeometricVerificationRequest),
                      ("geometric_verification_response", GeometricVerificationResponse)):
        enc = getattr(L, f"scl_msg_{name}_encode"); dec = getattr(L, f"scl_msg_{name}_decode")
        enc.restype = c_int; enc.argtypes = [POINTER(typ), u8p, c_size_t, szp]
        dec.restype = c_int; dec.argtypes = [u8p, c_size_t, POINTER(typ)]
    L.scl_msg_cloud_from_xyzi.restype = c_int
    L.scl_msg_cloud_from_xyzi.argtypes = [c_void_p, c_uint32, POINTER(Cloud), POINTER(PointField)]
    L.scl_msg_cloud_xyz_layout.restype = c_int
    L.scl_msg_cloud_xyz_layout.argtypes = [POINTER(Cloud), POINTER(c_int), POINTER(c_int)]
    L.scl_msg_transform_from_pose.restype = c_int
    L.scl_msg_transform_from_pose.argtypes = [c_double] * 6 + [POINTER(Transform)]
    L.scl_msg_transform_to_pose.restype = c_int
    L.scl_msg_transform_to_pose.argtypes = [POINTER(Transform)] + [POINTER(c_double)] * 6
    _bound = L
    return L


def encode(msg):
    """struct -> bytes (roscpp's wire format)"""
    L = lib()
    name = {GlobalDescriptor: "global_descriptor", LoopInfo: "loop_info", GeometricVerificationRequest: "geometric_verification_request",
            GeometricVerificationResponse: "geometric_verification_response"}[type(msg)]
    fn = getattr(L, f"scl_msg_{name}_encode")
    n = c_size_t()
    rc = fn(byref(msg), None, 0, byref(n))
    if rc != 0:
        raise ValueError(f"encode {name}: status {rc}")
    buf = (c_uint8 * max(1, n.value))()
    rc = fn(byref(msg), buf, n.value, byref(n))
    if rc != 0:
        raise ValueError(f"encode {name}: status {rc}")
    return bytes(buf[:n.value])


def decode(typ, data):
    """bytes -> struct; arrays and strings point into the returned buffer object (kept alive as `msg._buffer`)"""
    L = lib()
    name = {GlobalDescriptor: "global_descriptor", LoopInfo: "loop_info", GeometricVerificationRequest: "geometric_verification_request",
            GeometricVerificationResponse: "geometric_verification_response"}[typ]
    buf = (c_uint8 * max(1, len(data))).from_buffer_copy(data if len(data) else b"\0")
    msg = typ()
    rc = getattr(L, f"scl_msg_{name}_decode")(buf, len(data), byref(msg))
    if rc != 0:
        raise ValueError(f"decode {name}: status {rc}")
    msg._buffer = buf
    return msg


def values_of(msg):
    """the float32[] values of a decoded / filled global_descriptor as a numpy array (copy)"""
    return np.ctypeslib.as_array(msg.values, shape=(msg.n_values,)).copy() if msg.n_values else np.zeros(0, np.float32)
