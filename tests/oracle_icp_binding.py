"""ctypes binding of the ICP part of oracle/liboracle.so (test infrastructure only)."""
import ctypes
from ctypes import POINTER, byref, c_double, c_float, c_int, c_void_p

import numpy as np

import oracle_binding as ob


class IcpoParams(ctypes.Structure):
    _fields_ = [("max_iterations", c_int), ("max_correspondence_dist", c_double),
                ("transformation_epsilon", c_double), ("euclidean_fitness_epsilon", c_double),
                ("estimator", c_int), ("normal_radius", c_double)]


def _lib():
    L = ob.load()
    if not getattr(L, "_icp_bound", False):
        L.icpo_default_params.argtypes = [POINTER(IcpoParams)]
        L.icpo_nn.argtypes = [c_void_p, c_int, c_void_p, c_int, c_int, c_int, POINTER(c_int), POINTER(c_float)]
        L.icpo_rigid_svd.restype = c_int
        L.icpo_rigid_svd.argtypes = [c_void_p, c_void_p, c_int, POINTER(c_int), POINTER(c_int), c_int, POINTER(c_float)]
        L.icpo_transform.argtypes = [c_void_p, c_int, c_int, POINTER(c_float), c_void_p]
        L.icpo_icp_align.restype = c_int
        L.icpo_icp_align.argtypes = [c_void_p, c_int, c_void_p, c_int, c_int, POINTER(IcpoParams), POINTER(c_float),
                                     POINTER(c_float), POINTER(c_int), POINTER(c_int)]
        L.icpo_rotation_from_covariance.argtypes = [POINTER(c_double), POINTER(c_double)]
        L.icpo_normals.argtypes = [c_void_p, c_int, c_int, c_double, POINTER(c_float)]
        L.icpo_voxel_grid.restype = c_int
        L.icpo_voxel_grid.argtypes = [c_void_p, c_int, c_int, c_float, c_void_p]
        L.icpo_pose_to_matrix.argtypes = [c_float] * 6 + [POINTER(c_float)]
        L.icpo_ransac.restype = c_int
        L.icpo_ransac.argtypes = [c_void_p, c_void_p, c_int, POINTER(c_int), POINTER(c_int), c_int, c_int, c_double,
                                  ctypes.c_ulonglong, POINTER(c_int), POINTER(c_int), POINTER(c_double)]
        L.icpo_geometric_verification.restype = c_int
        L.icpo_geometric_verification.argtypes = [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_double, c_double,
                                                  ctypes.c_ulonglong, POINTER(c_float), POINTER(c_int), POINTER(c_int), POINTER(c_int)]
        L._icp_bound = True
    return L


def _c(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.shape[0], a.shape[1] * 4


def default_params(max_iterations=50, estimator=0, normal_radius=1.0):
    p = IcpoParams(); _lib().icpo_default_params(byref(p)); p.max_iterations = max_iterations
    p.estimator = estimator; p.normal_radius = normal_radius
    return p


def nn(src, tgt, use_grid=True):
    s, ns, st = _c(src); t, nt, _ = _c(tgt)
    idx = np.empty(ns, np.int32); d2 = np.empty(ns, np.float32)
    _lib().icpo_nn(s.ctypes.data_as(c_void_p), ns, t.ctypes.data_as(c_void_p), nt, st, 1 if use_grid else 0,
                   idx.ctypes.data_as(POINTER(c_int)), d2.ctypes.data_as(POINTER(c_float)))
    return idx, d2


def rigid_svd(src, tgt, si, ti):
    s, ns, st = _c(src); t, nt, _ = _c(tgt)
    si = np.ascontiguousarray(si, np.int32); ti = np.ascontiguousarray(ti, np.int32)
    T = np.empty(16, np.float32)
    rc = _lib().icpo_rigid_svd(s.ctypes.data_as(c_void_p), t.ctypes.data_as(c_void_p), st,
                               si.ctypes.data_as(POINTER(c_int)), ti.ctypes.data_as(POINTER(c_int)), si.size,
                               T.ctypes.data_as(POINTER(c_float)))
    assert rc == 0
    return T.reshape(4, 4)


def transform(cloud, T):
    a, n, st = _c(cloud)
    out = np.empty_like(a)
    Tm = np.ascontiguousarray(T, np.float32).reshape(16)
    _lib().icpo_transform(a.ctypes.data_as(c_void_p), n, st, Tm.ctypes.data_as(POINTER(c_float)), out.ctypes.data_as(c_void_p))
    return out


def icp_align(src, tgt, params=None):
    s, ns, st = _c(src); t, nt, _ = _c(tgt)
    p = params or default_params()
    T = np.empty(16, np.float32); fit = c_float(); conv = c_int(); it = c_int()
    _lib().icpo_icp_align(s.ctypes.data_as(c_void_p), ns, t.ctypes.data_as(c_void_p), nt, st, byref(p),
                          T.ctypes.data_as(POINTER(c_float)), byref(fit), byref(conv), byref(it))
    return T.reshape(4, 4), fit.value, bool(conv.value), it.value


def ransac(src, tgt, si, ti, max_iterations=1000, inlier_threshold=0.25, seed=1):
    s, ns, st = _c(src); t, nt, _ = _c(tgt)
    si = np.ascontiguousarray(si, np.int32); ti = np.ascontiguousarray(ti, np.int32)
    mask = np.empty(si.size, np.int32); best = c_int(); T = np.empty(12, np.float64)
    n = _lib().icpo_ransac(s.ctypes.data_as(c_void_p), t.ctypes.data_as(c_void_p), st, si.ctypes.data_as(POINTER(c_int)),
                           ti.ctypes.data_as(POINTER(c_int)), si.size, max_iterations, inlier_threshold, seed,
                           mask.ctypes.data_as(POINTER(c_int)), byref(best), T.ctypes.data_as(POINTER(c_double)))
    return mask, n, best.value, T.reshape(3, 4)


def geometric_verification(src, tgt, ransac_iterations=1000, inlier_threshold=0.25, inlier_ratio=0.45, seed=1):
    s, ns, st = _c(src); t, nt, _ = _c(tgt)
    T = np.empty(16, np.float32); ok = c_int(); nc = c_int(); ni = c_int()
    _lib().icpo_geometric_verification(s.ctypes.data_as(c_void_p), ns, t.ctypes.data_as(c_void_p), nt, st, ransac_iterations,
                                       inlier_threshold, inlier_ratio, seed, T.ctypes.data_as(POINTER(c_float)),
                                       byref(ok), byref(nc), byref(ni))
    return T.reshape(4, 4), bool(ok.value), nc.value, ni.value


def voxel_grid(cloud, leaf):
    a, n, st = _c(cloud)
    out = np.empty_like(a)
    m = _lib().icpo_voxel_grid(a.ctypes.data_as(c_void_p), n, st, leaf, out.ctypes.data_as(c_void_p))
    return None if m < 0 else out[:m].copy()


def pose_to_matrix(x, y, z, roll, pitch, yaw):
    T = np.empty(16, np.float32)
    _lib().icpo_pose_to_matrix(x, y, z, roll, pitch, yaw, T.ctypes.data_as(POINTER(c_float)))
    return T.reshape(4, 4)


def normals(tgt, radius=1.0):
    t, nt, st = _c(tgt)
    out = np.empty((nt, 3), np.float32)
    _lib().icpo_normals(t.ctypes.data_as(c_void_p), nt, st, radius, out.ctypes.data_as(POINTER(c_float)))
    return out
