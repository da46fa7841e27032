"""ctypes binding of oracle/liboracle.so -- the CPU checker (test infrastructure only)."""
import ctypes
import os
import subprocess
from ctypes import POINTER, byref, c_double, c_float, c_int, c_int8, c_void_p

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")
# SCL_ORACLE_LIB: another build of the checker -- `make sanitize` runs the CPU tests against oracle/liboracle_asan.so (ASan + UBSan)
if os.environ.get("SCL_ORACLE_LIB"):
    LIB = os.path.abspath(os.environ["SCL_ORACLE_LIB"])
REF_LIB = os.path.join(ORACLE_DIR, "_ref", "libnanoflann_ref.so")


class ScoConfig(ctypes.Structure):
    _fields_ = [
        ("num_ring", c_int), ("num_sector", c_int), ("num_candidates", c_int),
        ("dist_thres", c_double), ("lidar_height", c_double), ("max_radius", c_double),
        ("num_exclude_recent", c_int), ("tree_making_period", c_int),
        ("search_ratio", c_double), ("knn_exclude_eps", c_float),
    ]


def make_config(R=20, S=60, k=3, dist_thres=0.14, lidar_height=1.65, max_radius=80.0,
                exclude_recent=100, tree_period=10, search_ratio=0.1, knn_exclude_eps=0.0):
    return ScoConfig(R, S, k, dist_thres, lidar_height, max_radius, exclude_recent, tree_period,
                     search_ratio, knn_exclude_eps)


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"])
    L = ctypes.CDLL(LIB)
    dp, fp, ip = POINTER(c_double), POINTER(c_float), POINTER(c_int)
    CP = POINTER(ScoConfig)
    L.sco_atan_pos.restype = c_double; L.sco_atan_pos.argtypes = [c_double]
    L.sco_xy2theta.restype = c_float; L.sco_xy2theta.argtypes = [c_float, c_float]
    L.sco_make_scancontext.argtypes = [CP, c_void_p, c_int, c_int, dp, fp]
    L.sco_ringkey.argtypes = [c_int, c_int, dp, fp]
    L.sco_sectorkey.argtypes = [c_int, c_int, dp, dp]
    L.sco_fast_align.restype = c_int; L.sco_fast_align.argtypes = [c_int, dp, dp]
    L.sco_dist_direct.restype = c_double; L.sco_dist_direct.argtypes = [c_int, c_int, dp, dp]
    L.sco_distance.argtypes = [CP, dp, dp, dp, ip]
    L.sco_distance_fast.argtypes = [CP, dp, dp, dp, ip]
    L.sco_knn.restype = c_int; L.sco_knn.argtypes = [fp, c_int, c_int, fp, c_int, c_float, ip, fp]
    L.sco_db_create.restype = c_void_p; L.sco_db_create.argtypes = [CP]
    L.sco_db_destroy.argtypes = [c_void_p]
    L.sco_db_save_wire.argtypes = [c_void_p, fp, c_int8, c_int]
    L.sco_db_make_and_save.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int8, c_int, fp]
    L.sco_db_size.restype = c_int; L.sco_db_size.argtypes = [c_void_p]
    L.sco_db_get_index.argtypes = [c_void_p, c_int, POINTER(c_int8), ip]
    L.sco_db_desc.restype = dp; L.sco_db_desc.argtypes = [c_void_p, c_int]
    L.sco_db_ringkey.restype = fp; L.sco_db_ringkey.argtypes = [c_void_p, c_int]
    L.sco_db_detect_intra.argtypes = [c_void_p, c_int, ip, fp, dp, dp]
    L.sco_db_detect_inter.argtypes = [c_void_p, c_int, ip, fp, dp]
    L.sco_db_detect_full.argtypes = [c_void_p, c_int, ip, ip, ip, dp]
    L.sco_db_distance_batch.argtypes = [c_void_p, c_int, ip, c_int, dp, ip, c_int]
    L.sco_db_distance_batch_mt.argtypes = [c_void_p, c_int, ip, c_int, dp, ip, c_int, c_int]
    _lib = L
    return L


def _p(a, t):
    return a.ctypes.data_as(POINTER(t))


def wire_to_colmajor(values, R, S):
    """row-major float wire vector -> column-major fp64 matrix (the oracle's MatrixXd layout)."""
    v = np.asarray(values, dtype=np.float32).reshape(R, S)
    return np.ascontiguousarray(v.T.astype(np.float64)).reshape(-1)      # [c*R + r]


def distance(cfg, v1, v2, fast=False):
    L = load()
    R, S = cfg.num_ring, cfg.num_sector
    a = wire_to_colmajor(v1, R, S); b = wire_to_colmajor(v2, R, S)
    d = c_double(); s = c_int()
    (L.sco_distance_fast if fast else L.sco_distance)(byref(cfg), _p(a, c_double), _p(b, c_double), byref(d), byref(s))
    return d.value, s.value


def make_scancontext(cfg, cloud):
    L = load()
    a = np.ascontiguousarray(cloud, dtype=np.float32)
    R, S = cfg.num_ring, cfg.num_sector
    desc = np.empty(R * S, dtype=np.float64); vT = np.empty(R * S, dtype=np.float32)
    L.sco_make_scancontext(byref(cfg), a.ctypes.data_as(c_void_p), a.shape[0], a.shape[1] * 4,
                           _p(desc, c_double), _p(vT, c_float))
    return vT


def knn(keys, query, k, exclude_eps=0.0):
    L = load()
    keys = np.ascontiguousarray(keys, dtype=np.float32); query = np.ascontiguousarray(query, dtype=np.float32)
    idx = np.empty(k, dtype=np.int32); d2 = np.empty(k, dtype=np.float32)
    found = L.sco_knn(_p(keys, c_float), keys.shape[0], keys.shape[1], _p(query, c_float), k,
                      c_float(exclude_eps), _p(idx, c_int), _p(d2, c_float))
    return idx, d2, found


class OracleDB:
    """Mirror of scan_context_descriptor on the CPU checker."""

    def __init__(self, cfg):
        self.L = load(); self.cfg = cfg
        self.h = c_void_p(self.L.sco_db_create(byref(cfg)))
        self.R, self.S = cfg.num_ring, cfg.num_sector

    def close(self):
        if self.h:
            self.L.sco_db_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def save_wire(self, values, robot=0, index=0):
        v = np.ascontiguousarray(values, dtype=np.float32).reshape(-1)
        self.L.sco_db_save_wire(self.h, _p(v, c_float), robot, index)

    def save_bulk(self, values):
        v = np.ascontiguousarray(values, dtype=np.float32).reshape(-1, self.R * self.S)
        for i in range(v.shape[0]):
            self.L.sco_db_save_wire(self.h, _p(v[i], c_float), 0, self.size())

    def make_and_save(self, cloud, robot=0, index=0):
        a = np.ascontiguousarray(cloud, dtype=np.float32)
        vT = np.empty(self.R * self.S, dtype=np.float32)
        self.L.sco_db_make_and_save(self.h, a.ctypes.data_as(c_void_p), a.shape[0], a.shape[1] * 4,
                                    robot, index, _p(vT, c_float))
        return vT

    def size(self):
        return self.L.sco_db_size(self.h)

    def get_index(self, key):
        r = c_int8(); i = c_int()
        self.L.sco_db_get_index(self.h, key, byref(r), byref(i))
        return r.value, i.value

    def ringkey(self, key):
        return np.ctypeslib.as_array(self.L.sco_db_ringkey(self.h, key), shape=(self.R,)).copy()

    def ringkeys(self, n=None):
        n = self.size() if n is None else n
        return np.stack([self.ringkey(i) for i in range(n)]) if n else np.zeros((0, self.R), np.float32)

    def detect_intra(self, cur):
        lid = c_int(); sh = c_float(); d = c_double(); de = c_double()
        self.L.sco_db_detect_intra(self.h, cur, byref(lid), byref(sh), byref(d), byref(de))
        return lid.value, sh.value, d.value, de.value

    def detect_inter(self, cur):
        lid = c_int(); yaw = c_float(); d = c_double()
        self.L.sco_db_detect_inter(self.h, cur, byref(lid), byref(yaw), byref(d))
        return lid.value, yaw.value, d.value

    def detect_full(self, cur):
        lid = c_int(); nn = c_int(); sh = c_int(); d = c_double()
        self.L.sco_db_detect_full(self.h, cur, byref(lid), byref(nn), byref(sh), byref(d))
        return lid.value, nn.value, sh.value, d.value

    def distance_batch_mt(self, cur, cand, fast, threads):
        cand = np.ascontiguousarray(cand, dtype=np.int32); n = cand.size
        dist = np.empty(n, dtype=np.float64); shift = np.empty(n, dtype=np.int32)
        self.L.sco_db_distance_batch_mt(self.h, cur, _p(cand, c_int), n, _p(dist, c_double), _p(shift, c_int),
                                        1 if fast else 0, threads)
        return dist, shift

    def distance_batch(self, cur, cand=None, n=None, fast=True):
        if cand is not None:
            cand = np.ascontiguousarray(cand, dtype=np.int32); n = cand.size; cp = _p(cand, c_int)
        else:
            cp = None
        dist = np.empty(n, dtype=np.float64); shift = np.empty(n, dtype=np.int32)
        self.L.sco_db_distance_batch(self.h, cur, cp, n, _p(dist, c_double), _p(shift, c_int), 1 if fast else 0)
        return dist, shift


def load_ref_nanoflann():
    """The reference's own nanoflann (oracle/_ref), or None when not built (GPU box)."""
    if not os.path.exists(REF_LIB):
        return None
    L = ctypes.CDLL(REF_LIB)
    L.ref_nanoflann_knn.restype = c_int
    L.ref_nanoflann_knn.argtypes = [POINTER(c_float), c_int, c_int, POINTER(c_float), c_int,
                                    POINTER(ctypes.c_longlong), POINTER(c_float)]
    return L


def ref_knn(L, keys, query, k):
    keys = np.ascontiguousarray(keys, dtype=np.float32); query = np.ascontiguousarray(query, dtype=np.float32)
    idx = np.empty(k, dtype=np.int64); d2 = np.empty(k, dtype=np.float32)
    found = L.ref_nanoflann_knn(_p(keys, c_float), keys.shape[0], keys.shape[1], _p(query, c_float), k,
                                idx.ctypes.data_as(POINTER(ctypes.c_longlong)), _p(d2, c_float))
    return idx, d2, found
