"""ctypes binding of include/scl_iris.h: the LiDAR-Iris building blocks (image, templates, Hamming matching) on the GPU."""
import ctypes
from ctypes import POINTER, byref, c_char_p, c_double, c_float, c_int, c_int8, c_uint8, c_void_p

import numpy as np

from ._native import load_library


class IrisConfig(ctypes.Structure):
    """scl_iris_config; defaults = lidar_iris_descriptor's constructor defaults (descriptor.h:473-486)"""
    _fields_ = [("rows", c_int), ("cols", c_int), ("nscan", c_int), ("nscale", c_int), ("min_wavelength", c_int),
                ("mult", c_float), ("sigma_onf", c_float), ("device", c_int),
                ("dist_thres", c_double), ("num_exclude_recent", c_int), ("match_num", c_int), ("num_candidates", c_int),
                ("robot_num", c_int), ("this_id", c_int), ("knn_exclude_eps", c_float), ("wire_decode", c_int), ("shift_search", c_int)]


_bound = None


def _lib():
    global _bound
    if _bound is not None:
        return _bound
    L = load_library()
    P, u8, fp, ip = c_void_p, POINTER(c_uint8), POINTER(c_float), POINTER(c_int)
    sig = {
        "scl_iris_default_config": (c_int, [POINTER(IrisConfig)]),
        "scl_iris_create": (c_int, [POINTER(IrisConfig), POINTER(P)]),
        "scl_iris_destroy": (c_int, [P]),
        "scl_iris_last_error": (c_char_p, [P]),
        "scl_iris_make_image": (c_int, [P, P, c_int, c_int, u8, fp]),
        "scl_iris_make_and_save": (c_int, [P, P, c_int, c_int, c_int8, c_int, fp]),
        "scl_iris_save_image": (c_int, [P, u8, fp, c_int8, c_int]),
        "scl_iris_save_from_wire": (c_int, [P, fp, c_int8, c_int]),
        "scl_iris_get_size": (c_int, [P]),
        "scl_iris_get_size_of": (c_int, [P, c_int]),
        "scl_iris_local_to_global": (c_int, [P, c_int, c_int, ip]),
        "scl_iris_detect_intra": (c_int, [P, c_int, ip, fp, fp]),
        "scl_iris_detect_inter": (c_int, [P, c_int, ip, fp, fp]),
        "scl_iris_get_index": (c_int, [P, c_int, POINTER(c_int8), ip]),
        "scl_iris_get_image": (c_int, [P, c_int, u8, fp]),
        "scl_iris_get_feature": (c_int, [P, c_int, u8, u8]),
        "scl_iris_hamming": (c_int, [P, c_int, c_int, c_int, fp, ip]),
        "scl_iris_hamming_batch": (c_int, [P, c_int, ip, ip, c_int, fp, ip]),
        "scl_iris_hamming_all_shifts": (c_int, [P, c_int, ip, c_int, fp, ip]),
        "scl_iris_fft_match": (c_int, [P, c_int, c_int, c_int, fp, ip]),
        "scl_iris_compare": (c_int, [P, c_int, ip, c_int, fp, ip]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name); fn.restype = res; fn.argtypes = args
    _bound = L
    return L


class IrisEngine:
    """Mirror of lidar_iris_descriptor (descriptor.h:462-1302): same constructor arguments and defaults, the six plugin
    calls (make_and_save, save_from_wire, detect_intra, detect_inter, get_index, get_size) and the building blocks."""

    def __init__(self, rows=80, cols=360, nscan=64, dist_thres=0.32, num_exclude_recent=30, match_num=2, num_candidates=10,
                 nscale=4, min_wavelength=18, mult=1.6, sigma_onf=0.75, robot_num=1, this_id=0, device=0,
                 knn_exclude_eps=None, wire_decode=0, shift_search=0):
        self.L = _lib()
        cfg = IrisConfig()
        self._check_rc(self.L.scl_iris_default_config(byref(cfg)))
        cfg.rows, cfg.cols, cfg.nscan, cfg.nscale, cfg.min_wavelength = rows, cols, nscan, nscale, min_wavelength
        cfg.mult, cfg.sigma_onf, cfg.device = mult, sigma_onf, device
        cfg.dist_thres, cfg.num_exclude_recent, cfg.match_num, cfg.num_candidates = dist_thres, num_exclude_recent, match_num, num_candidates
        cfg.robot_num, cfg.this_id, cfg.wire_decode, cfg.shift_search = robot_num, this_id, wire_decode, shift_search
        if knn_exclude_eps is not None:
            cfg.knn_exclude_eps = knn_exclude_eps
        self.cfg = cfg
        self.h = c_void_p()
        rc = self.L.scl_iris_create(byref(cfg), byref(self.h))
        if rc != 0:
            self.h = c_void_p()
            raise RuntimeError(f"scl_iris_create: status {rc}")
        self.rows, self.cols, self.trows = rows, cols, 2 * nscale * rows

    @staticmethod
    def _check_rc(rc):
        if rc != 0:
            raise RuntimeError(f"scl_iris: status {rc}")

    def _check(self, rc, where):
        if rc != 0:
            raise RuntimeError(f"{where}: status {rc} ({self.L.scl_iris_last_error(self.h).decode()})")

    def close(self):
        if self.h and self.h.value:
            self.L.scl_iris_destroy(self.h); self.h = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _cloud(points):
        a = np.ascontiguousarray(points, dtype=np.float32)
        return a, a.shape[0], a.shape[1] * 4

    def make_image(self, points):
        a, n, st = self._cloud(points)
        img = np.empty((self.rows, self.cols), np.uint8); key = np.empty(self.rows, np.float32)
        self._check(self.L.scl_iris_make_image(self.h, a.ctypes.data_as(c_void_p), n, st, img.ctypes.data_as(POINTER(c_uint8)),
                                               key.ctypes.data_as(POINTER(c_float))), "scl_iris_make_image")
        return img, key

    def make_and_save(self, points, robot=0, index=0):
        a, n, st = self._cloud(points)
        out = np.empty(self.rows * self.cols + self.rows, np.float32)
        self._check(self.L.scl_iris_make_and_save(self.h, a.ctypes.data_as(c_void_p), n, st, robot, index, out.ctypes.data_as(POINTER(c_float))),
                    "scl_iris_make_and_save")
        return out

    def save_image(self, image, rowkey, robot=0, index=0):
        img = np.ascontiguousarray(image, np.uint8); key = np.ascontiguousarray(rowkey, np.float32)
        self._check(self.L.scl_iris_save_image(self.h, img.ctypes.data_as(POINTER(c_uint8)), key.ctypes.data_as(POINTER(c_float)), robot, index),
                    "scl_iris_save_image")

    def save_from_wire(self, values, robot=0, index=0):
        v = np.ascontiguousarray(values, np.float32)
        assert v.size == self.rows * self.cols + self.rows
        self._check(self.L.scl_iris_save_from_wire(self.h, v.ctypes.data_as(POINTER(c_float)), robot, index), "scl_iris_save_from_wire")

    def get_size(self, robot=-1):
        n = self.L.scl_iris_get_size_of(self.h, robot)
        if n < 0:
            self._check(n, "scl_iris_get_size_of")
        return n

    def local_to_global(self, robot, local):
        k = c_int()
        self._check(self.L.scl_iris_local_to_global(self.h, robot, local, byref(k)), "scl_iris_local_to_global")
        return k.value

    def _detect(self, fn, name, cur):
        loop, bias, dist = c_int(), c_float(), c_float()
        self._check(fn(self.h, cur, byref(loop), byref(bias), byref(dist)), name)
        return loop.value, bias.value, dist.value

    def detect_intra(self, cur):
        """(loop local index or -1, column shift, smallest distance seen) -- detectIntraLoopClosureID, D.h:1085-1151"""
        return self._detect(self.L.scl_iris_detect_intra, "scl_iris_detect_intra", cur)

    def detect_inter(self, cur):
        """(loop global key or -1, column shift, smallest distance seen) -- detectInterLoopClosureID, D.h:1153-1253"""
        return self._detect(self.L.scl_iris_detect_inter, "scl_iris_detect_inter", cur)

    def get_index(self, key):
        r, i = c_int8(), c_int()
        self._check(self.L.scl_iris_get_index(self.h, key, byref(r), byref(i)), "scl_iris_get_index")
        return r.value, i.value

    def get_image(self, key):
        img = np.empty((self.rows, self.cols), np.uint8); k = np.empty(self.rows, np.float32)
        self._check(self.L.scl_iris_get_image(self.h, key, img.ctypes.data_as(POINTER(c_uint8)), k.ctypes.data_as(POINTER(c_float))), "scl_iris_get_image")
        return img, k

    def get_feature(self, key):
        T = np.empty((self.trows, self.cols), np.uint8); M = np.empty((self.trows, self.cols), np.uint8)
        self._check(self.L.scl_iris_get_feature(self.h, key, T.ctypes.data_as(POINTER(c_uint8)), M.ctypes.data_as(POINTER(c_uint8))), "scl_iris_get_feature")
        return T, M

    def hamming(self, key1, key2, scale):
        d, b = c_float(), c_int()
        self._check(self.L.scl_iris_hamming(self.h, key1, key2, scale, byref(d), byref(b)), "scl_iris_hamming")
        return d.value, b.value

    def hamming_batch(self, key1, cand, scales):
        c = np.ascontiguousarray(cand, np.int32); s = np.ascontiguousarray(scales, np.int32)
        d = np.empty(c.size, np.float32); b = np.empty(c.size, np.int32)
        self._check(self.L.scl_iris_hamming_batch(self.h, key1, c.ctypes.data_as(POINTER(c_int)), s.ctypes.data_as(POINTER(c_int)), c.size,
                                                  d.ctypes.data_as(POINTER(c_float)), b.ctypes.data_as(POINTER(c_int))), "scl_iris_hamming_batch")
        return d, b

    def hamming_all_shifts(self, key1, cand):
        c = np.ascontiguousarray(cand, np.int32)
        d = np.empty(c.size, np.float32); b = np.empty(c.size, np.int32)
        self._check(self.L.scl_iris_hamming_all_shifts(self.h, key1, c.ctypes.data_as(POINTER(c_int)), c.size,
                                                       d.ctypes.data_as(POINTER(c_float)), b.ctypes.data_as(POINTER(c_int))), "scl_iris_hamming_all_shifts")
        return d, b

    def fft_match(self, key0, roll0, key1):
        """fftMatch(image of key0 turned by roll0 columns, image of key1), descriptor.h:793-932: (centre x as float32, compatible)"""
        cx, ok = c_float(), c_int()
        self._check(self.L.scl_iris_fft_match(self.h, key0, roll0, key1, byref(cx), byref(ok)), "scl_iris_fft_match")
        return np.float32(cx.value), bool(ok.value)

    def compare(self, key1, cand):
        """compare(key1, cand[i]) of descriptor.h:964-1024 (match_num as configured): distances and shifts"""
        c = np.ascontiguousarray(cand, np.int32)
        d = np.empty(c.size, np.float32); b = np.empty(c.size, np.int32)
        self._check(self.L.scl_iris_compare(self.h, key1, c.ctypes.data_as(POINTER(c_int)), c.size,
                                            d.ctypes.data_as(POINTER(c_float)), b.ctypes.data_as(POINTER(c_int))), "scl_iris_compare")
        return d, b
