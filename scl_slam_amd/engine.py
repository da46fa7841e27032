"""ctypes binding of include/scl_engine.h + a Python mirror of the reference's plugin class.

``ScanContextEngine`` exposes the C ABI one-to-one (numpy arrays in / out).
``ScanContextDescriptor`` mirrors ``scan_context_descriptor`` (reference
include/descriptor.h:1304-1801): same constructor arguments, same six method names,
same return conventions (``(-1, 0.0)`` = no loop).
"""
import ctypes
import os
from ctypes import POINTER, byref, c_char_p, c_double, c_float, c_int, c_int8, c_uint64, c_void_p

import numpy as np

from ._native import load_library

QUERY_STAGED = -1


class SclConfig(ctypes.Structure):
    """scl_config; defaults = scan_context_descriptor ctor defaults (D.h:1308-1316)."""
    _fields_ = [
        ("num_ring", c_int), ("num_sector", c_int), ("num_candidates", c_int),
        ("dist_thres", c_double), ("lidar_height", c_double), ("max_radius", c_double),
        ("num_exclude_recent", c_int), ("tree_making_period", c_int),
        ("search_ratio", c_double), ("knn_exclude_eps", c_float),
        ("device", c_int), ("initial_capacity", c_int),
    ]


class IcpParams(ctypes.Structure):
    """scl_icp_params; defaults = the settings at DM.h:1109-1112."""
    _fields_ = [
        ("max_iterations", c_int), ("max_correspondence_dist", c_double),
        ("transformation_epsilon", c_double), ("euclidean_fitness_epsilon", c_double),
        ("estimator", c_int), ("normal_radius", c_double),
    ]


class SclProfile(ctypes.Structure):
    _fields_ = [
        ("sc_distance_ms", c_double), ("sc_distance_launches", c_uint64), ("sc_distance_pairs", c_uint64),
        ("ringkey_topk_ms", c_double), ("ringkey_topk_launches", c_uint64),
        ("argmin_ms", c_double), ("argmin_launches", c_uint64),
        ("make_sc_ms", c_double), ("make_sc_launches", c_uint64), ("make_sc_points", c_uint64),
        ("ingest_ms", c_double), ("ingest_launches", c_uint64),
        ("icp_nn_ms", c_double), ("icp_nn_launches", c_uint64),
        ("icp_reduce_ms", c_double), ("icp_reduce_launches", c_uint64),
    ]


class SclError(RuntimeError):
    def __init__(self, status, where, detail=""):
        self.status = status
        super().__init__(f"{where}: status {status}" + (f" ({detail})" if detail else ""))


_bound = False


def _bind(lib):
    global _bound
    if _bound:
        return
    P = c_void_p
    fp, dp, ip = POINTER(c_float), POINTER(c_double), POINTER(c_int)
    sig = {
        "scl_status_string": (c_char_p, [c_int]),
        "scl_last_error": (c_char_p, [P]),
        "scl_abi_version": (c_int, []),
        "scl_default_config": (c_int, [POINTER(SclConfig)]),
        "scl_create": (c_int, [POINTER(SclConfig), POINTER(P)]),
        "scl_create_sharded": (c_int, [POINTER(SclConfig), ip, c_int, c_int, POINTER(P)]),
        "scl_shard_info": (c_int, [P, ip, ip]),
        "scl_destroy": (c_int, [P]),
        "scl_make_and_save": (c_int, [P, P, c_int, c_int, c_int8, c_int, fp]),
        "scl_save_from_wire": (c_int, [P, fp, c_int8, c_int]),
        "scl_make_and_save_filtered": (c_int, [P, P, c_int, c_int, c_float, c_int8, c_int, fp, ip]),
        "scl_detect_intra": (c_int, [P, c_int, ip, fp, dp]),
        "scl_detect_inter": (c_int, [P, c_int, ip, fp, dp]),
        "scl_get_index": (c_int, [P, c_int, POINTER(c_int8), ip]),
        "scl_get_size": (c_int, [P, c_int]),
        "scl_make_descriptor": (c_int, [P, P, c_int, c_int, fp]),
        "scl_save_bulk": (c_int, [P, fp, c_int, POINTER(c_int8), ip]),
        "scl_get_descriptor": (c_int, [P, c_int, fp]),
        "scl_get_ringkey": (c_int, [P, c_int, fp]),
        "scl_get_sectorkey": (c_int, [P, c_int, dp]),
        "scl_get_descriptors": (c_int, [P, c_int, c_int, fp]),
        "scl_find_key": (c_int, [P, c_int8, c_int, ip]),
        "scl_db_dump_file": (c_int, [P, c_char_p]),
        "scl_db_load_file": (c_int, [P, c_char_p, ip]),
        "scl_matrix_to_pose": (c_int, [fp, fp, fp, fp, fp, fp, fp]),
        "scl_loop_pose_between": (c_int, [fp, fp, fp, dp, dp]),
        "scl_stage_query": (c_int, [P, fp]),
        "scl_ringkey_topk": (c_int, [P, c_int, c_int, c_int, c_int, ip, fp, ip]),
        "scl_sc_distance_batch": (c_int, [P, c_int, ip, c_int, dp, ip]),
        "scl_sc_distance_matrix": (c_int, [P, ip, c_int, c_int, c_int, dp, ip]),
        "scl_detect_full": (c_int, [P, c_int, ip, ip, ip, dp]),
        "scl_detect_full_range": (c_int, [P, c_int, c_int, c_int, ip, ip, dp]),
        "scl_get_last_topk": (c_int, [P, c_int, ip, fp]),
        "scl_screen_distances": (c_int, [P, c_int, c_int, c_int, fp, ip, ip, fp]),
        "scl_screen_distances_many": (c_int, [P, ip, c_int, c_int, c_int, fp, fp]),
        "scl_detect_full_submit": (c_int, [P, c_int, c_int, c_int, ip]),
        "scl_detect_full_collect": (c_int, [P, c_int, ip, ip, dp]),
        "scl_detect_full_submit_many": (c_int, [P, ip, ip, ip, c_int, ip]),
        "scl_detect_full_stream": (c_int, [P, ip, ip, ip, c_int, c_int, c_int, ip, ip, dp]),
        "scl_topk_with_distance": (c_int, [P, c_int, c_int, c_int, c_int, ip, fp, dp, ip, ip]),
        "scl_icp_default_params": (c_int, [POINTER(IcpParams)]),
        "scl_icp_align": (c_int, [P, P, c_int, P, c_int, c_int, POINTER(IcpParams), fp, fp, ip, ip]),
        "scl_icp_align_batch": (c_int, [P, P, c_int, POINTER(c_void_p), ip, c_int, c_int, POINTER(IcpParams), fp, fp, ip, ip]),
        "scl_nn_correspondences": (c_int, [P, P, c_int, P, c_int, c_int, ip, fp]),
        "scl_nn_correspondences_moved": (c_int, [P, P, c_int, P, c_int, c_int, fp, ip, fp]),
        "scl_rigid_svd": (c_int, [P, P, c_int, P, c_int, c_int, ip, ip, c_int, fp]),
        "scl_transform_cloud": (c_int, [P, P, c_int, c_int, fp, P]),
        "scl_voxel_grid": (c_int, [P, P, c_int, c_int, c_float, P, c_int, ip]),
        "scl_pose_to_matrix": (c_int, [c_float, c_float, c_float, c_float, c_float, c_float, fp]),
        "scl_assemble_submap": (c_int, [P, POINTER(c_void_p), ip, fp, c_int, c_int, c_float, P, c_int, ip]),
        "scl_ransac_correspondences": (c_int, [P, P, c_int, P, c_int, c_int, ip, ip, c_int, c_int, c_double, c_uint64, ip, ip, ip, fp]),
        "scl_geometric_verification": (c_int, [P, P, c_int, P, c_int, c_int, c_int, c_double, c_double, c_uint64, fp, ip, ip, ip]),
        "scl_keyframe_put": (c_int, [P, c_int, c_int, P, c_int, c_int]),
        "scl_keyframe_count": (c_int, [P, c_int]),
        "scl_keyframe_get": (c_int, [P, c_int, c_int, P, c_int, ip]),
        "scl_submap_from_store": (c_int, [P, c_int, c_int, c_int, fp, c_float, P, c_int, ip]),
        "scl_loop_icp_from_store": (c_int, [P, c_int, c_int, fp, c_int, c_int, fp, c_float, POINTER(IcpParams), c_int, c_int,
                                            fp, fp, ip, ip, ip, ip]),
        "scl_loop_icp_batch_from_store": (c_int, [P, c_int, c_int, fp, c_int, ip, c_int, fp, c_float, POINTER(IcpParams), c_int, c_int,
                                                  fp, fp, ip, ip, ip, ip]),
        "scl_geometric_verification_from_store": (c_int, [P, P, c_int, c_int, c_float, c_int, c_int, c_int, fp, c_float, c_int, c_int,
                                                          c_int, c_double, c_double, c_uint64, fp, ip, ip, ip, ip, ip]),
        "scl_profile_enable": (c_int, [P, c_int]),
        "scl_profile_reset": (c_int, [P]),
        "scl_profile_get": (c_int, [P, POINTER(SclProfile)]),
        "scl_alignment_stats": (c_int, [P, POINTER(ctypes.c_uint64), POINTER(ctypes.c_uint64), c_int]),
        "scl_survivor_stats": (c_int, [P, POINTER(ctypes.c_uint64), POINTER(ctypes.c_uint64), POINTER(ctypes.c_uint64), c_int]),
        "scl_device_name": (c_int, [P, c_char_p, c_int]),
        "scl_make_and_save_many": (c_int, [P, POINTER(c_void_p), ip, c_int, c_int, POINTER(c_int8), ip, fp]),
        "scl_stream_from_points": (c_int, [P, POINTER(c_void_p), ip, c_int, c_int, POINTER(c_int8), ip, ip, ip, dp, fp]),
        "scl_stream_from_store": (c_int, [P, c_int, c_int, c_int, ip, ip, dp, fp]),
        "scl_host_alloc": (c_int, [P, ctypes.c_size_t, POINTER(c_void_p)]),
        "scl_host_free": (c_int, [P, c_void_p]),
        "scl_host_register": (c_int, [P, c_void_p, ctypes.c_size_t]),
        "scl_host_unregister": (c_int, [P, c_void_p]),
        "scl_selftest_atanf_blocks": (c_int, [P, c_int, c_int, POINTER(ctypes.c_uint64)]),
        "scl_host_copy_rate": (c_int, [P, ctypes.c_size_t, c_int, dp]),
        "scl_selftest_bin_paths": (c_int, [P, c_int, c_uint64, c_uint64, POINTER(ctypes.c_uint64), POINTER(ctypes.c_uint64)]),
        "scl_selftest_sort_pairs": (c_int, [P, c_int, c_void_p, POINTER(ctypes.c_uint32), c_int, c_int, POINTER(c_int), c_int, c_void_p, POINTER(ctypes.c_uint32)]),
        "scl_selftest_prefix_sum": (c_int, [P, POINTER(ctypes.c_int32), c_int, c_int, POINTER(ctypes.c_int32)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _bound = True


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a, ctype):
    return a.ctypes.data_as(POINTER(ctype))


def _cloud(points):
    """Accept (n,3|4|8) float32 arrays; returns (array, n, stride_bytes)."""
    a = np.ascontiguousarray(points, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] < 3:
        raise ValueError("point cloud must be (n, >=3) float32")
    return a, a.shape[0], a.shape[1] * 4


class ScanContextEngine:
    """One engine = one keyframe database resident in one GPU's HBM, or -- with ``devices=[...]`` -- ONE database
    sharded over several GPUs behind the same interface (``scl_create_sharded``: keyframe g on shard g % len(devices);
    ``exchange`` 0 auto, 1 host merge, 2 RCCL min all-reduce of the full-DB winners)."""

    def __init__(self, num_ring=20, num_sector=60, num_candidates=3, dist_thres=0.14,
                 lidar_height=1.65, max_radius=80.0, num_exclude_recent=100,
                 tree_making_period=10, search_ratio=0.1, knn_exclude_eps=0.0,
                 device=0, initial_capacity=4096, devices=None, exchange=0):
        self._lib = load_library()
        _bind(self._lib)
        cfg = SclConfig()
        self._lib.scl_default_config(byref(cfg))
        cfg.num_ring, cfg.num_sector, cfg.num_candidates = num_ring, num_sector, num_candidates
        cfg.dist_thres, cfg.lidar_height, cfg.max_radius = dist_thres, lidar_height, max_radius
        cfg.num_exclude_recent, cfg.tree_making_period = num_exclude_recent, tree_making_period
        cfg.search_ratio, cfg.knn_exclude_eps = search_ratio, knn_exclude_eps
        cfg.device, cfg.initial_capacity = device, initial_capacity
        self.cfg = cfg
        self.R, self.S = num_ring, num_sector
        self._pinned = {}
        self._h = c_void_p()
        if devices is not None:
            devs = np.ascontiguousarray(devices, dtype=np.int32)
            rc = self._lib.scl_create_sharded(byref(cfg), _ptr(devs, c_int), devs.size, exchange, byref(self._h))
        else:
            rc = self._lib.scl_create(byref(cfg), byref(self._h))
        if rc != 0:
            self._h = c_void_p()
            raise SclError(rc, "scl_create", self._lib.scl_status_string(rc).decode())

    def shard_info(self):
        """(number of shards, exchange in use: 0 none, 1 host merge, 2 RCCL)"""
        n, x = c_int(), c_int()
        self._check(self._lib.scl_shard_info(self._h, byref(n), byref(x)), "scl_shard_info")
        return n.value, x.value

    # -- plumbing -----------------------------------------------------------
    def _check(self, rc, where):
        if rc != 0:
            raise SclError(rc, where, self._lib.scl_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.scl_destroy(self._h)
            self._h = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- the six virtuals -----------------------------------------------------
    def make_and_save(self, points, robot=0, index=0):
        a, n, stride = _cloud(points)
        out = np.empty(self.R * self.S, dtype=np.float32)
        self._check(self._lib.scl_make_and_save(self._h, a.ctypes.data_as(c_void_p), n, stride,
                                                robot, index, _ptr(out, c_float)), "scl_make_and_save")
        return out

    def make_and_save_filtered(self, points, leaf, robot=0, index=0):
        """makeDescriptors (DM.h:996-1002): voxel filter + descriptor + append, the filtered cloud stays on the device"""
        a, n, stride = _cloud(points)
        out = np.empty(self.R * self.S, dtype=np.float32); m = c_int()
        self._check(self._lib.scl_make_and_save_filtered(self._h, a.ctypes.data_as(c_void_p), n, stride, leaf,
                                                         robot, index, _ptr(out, c_float), byref(m)), "scl_make_and_save_filtered")
        return out, m.value

    def _cloud_list(self, clouds):
        arrs, ptrs, counts, stride = [], [], [], None
        for c in clouds:
            a, n, st = _cloud(c)
            if stride is None:
                stride = st
            elif st != stride:
                raise ValueError("the clouds of a batch share one point stride")
            arrs.append(a); ptrs.append(a.ctypes.data); counts.append(n)
        m = len(arrs)
        return arrs, (c_void_p * max(1, m))(*ptrs), np.asarray(counts, dtype=np.int32), m, (stride or 16)

    def make_and_save_many(self, clouds, robots=None, indexs=None, want_values=True):
        """scl_make_and_save_many: the clouds' descriptors appended in order; returns the wire-format values [count, R*S] (or None)"""
        arrs, ptrs, counts, m, stride = self._cloud_list(clouds)
        out = np.empty((m, self.R * self.S), dtype=np.float32) if want_values else None
        rb = None if robots is None else np.ascontiguousarray(robots, dtype=np.int8)
        ib = None if indexs is None else np.ascontiguousarray(indexs, dtype=np.int32)
        self._check(self._lib.scl_make_and_save_many(self._h, ptrs, _ptr(counts, c_int), m, stride,
                                                     None if rb is None else _ptr(rb, c_int8), None if ib is None else _ptr(ib, c_int),
                                                     None if out is None else _ptr(out, c_float)), "scl_make_and_save_many")
        return out

    def stream_from_points(self, clouds, robots=None, indexs=None, want_values=False):
        """scl_stream_from_points: per scan descriptor + append + full-database detection; returns (nn_idx, shift, dist[, values])"""
        arrs, ptrs, counts, m, stride = self._cloud_list(clouds)
        nn = np.empty(m, dtype=np.int32); sh = np.empty(m, dtype=np.int32); dd = np.empty(m, dtype=np.float64)
        out = np.empty((m, self.R * self.S), dtype=np.float32) if want_values else None
        rb = None if robots is None else np.ascontiguousarray(robots, dtype=np.int8)
        ib = None if indexs is None else np.ascontiguousarray(indexs, dtype=np.int32)
        self._check(self._lib.scl_stream_from_points(self._h, ptrs, _ptr(counts, c_int), m, stride,
                                                     None if rb is None else _ptr(rb, c_int8), None if ib is None else _ptr(ib, c_int),
                                                     _ptr(nn, c_int), _ptr(sh, c_int), _ptr(dd, c_double),
                                                     None if out is None else _ptr(out, c_float)), "scl_stream_from_points")
        return (nn, sh, dd, out) if want_values else (nn, sh, dd)

    def stream_from_store(self, robot, first_index, count, want_values=False):
        """scl_stream_from_store: descriptors from the stored keyframes' clouds + append + full-database detection"""
        nn = np.empty(count, dtype=np.int32); sh = np.empty(count, dtype=np.int32); dd = np.empty(count, dtype=np.float64)
        out = np.empty((count, self.R * self.S), dtype=np.float32) if want_values else None
        self._check(self._lib.scl_stream_from_store(self._h, robot, first_index, count, _ptr(nn, c_int), _ptr(sh, c_int), _ptr(dd, c_double),
                                                    None if out is None else _ptr(out, c_float)), "scl_stream_from_store")
        return (nn, sh, dd, out) if want_values else (nn, sh, dd)

    def host_alloc(self, shape, dtype=np.float32):
        """a numpy array over pinned host memory (scl_host_alloc): clouds handed over from it travel by DMA.  Freed with the engine
        (or host_free(array)); the array must not be used after that."""
        shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = c_void_p()
        self._check(self._lib.scl_host_alloc(self._h, max(1, nbytes), byref(p)), "scl_host_alloc")
        buf = (ctypes.c_char * max(1, nbytes)).from_address(p.value)
        a = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
        self._pinned[a.ctypes.data] = p.value
        return a

    def host_free(self, array):
        p = self._pinned.pop(array.ctypes.data, None)
        if p is not None:
            self._check(self._lib.scl_host_free(self._h, c_void_p(p)), "scl_host_free")

    def host_copy_rate(self, nbytes=64 << 20, reps=8):
        g = c_double()
        self._check(self._lib.scl_host_copy_rate(self._h, nbytes, reps, byref(g)), "scl_host_copy_rate")
        return g.value

    def selftest_bin_paths(self, mode, seed, n_points):
        bad, sure = ctypes.c_uint64(), ctypes.c_uint64()
        self._check(self._lib.scl_selftest_bin_paths(self._h, mode, seed, n_points, byref(bad), byref(sure)), "scl_selftest_bin_paths")
        return bad.value, sure.value

    def selftest_sort_pairs(self, keys, values, bits, segment_offsets=None):
        """(keys, values) stably sorted on the key bits [0, bits) by csrc/device_sort.hip; keys uint32 or uint64"""
        keys = np.ascontiguousarray(keys); values = np.ascontiguousarray(values, dtype=np.uint32)
        assert keys.dtype in (np.uint32, np.uint64) and keys.shape == values.shape
        ko, vo = np.empty_like(keys), np.empty_like(values)
        seg = None if segment_offsets is None else np.ascontiguousarray(segment_offsets, dtype=np.int32)
        self._check(self._lib.scl_selftest_sort_pairs(self._h, keys.dtype.itemsize, c_void_p(keys.ctypes.data), _ptr(values, ctypes.c_uint32), keys.size, bits,
                                                      None if seg is None else _ptr(seg, c_int), 0 if seg is None else seg.size - 1,
                                                      c_void_p(ko.ctypes.data), _ptr(vo, ctypes.c_uint32)), "scl_selftest_sort_pairs")
        return ko, vo

    def selftest_prefix_sum(self, values, inclusive=False):
        v = np.ascontiguousarray(values, dtype=np.int32); out = np.empty_like(v)
        self._check(self._lib.scl_selftest_prefix_sum(self._h, _ptr(v, ctypes.c_int32), v.size, 1 if inclusive else 0, _ptr(out, ctypes.c_int32)), "scl_selftest_prefix_sum")
        return out

    def selftest_atanf_blocks(self, first_block, n_blocks):
        out = np.zeros(n_blocks, dtype=np.uint64)
        self._check(self._lib.scl_selftest_atanf_blocks(self._h, first_block, n_blocks, out.ctypes.data_as(POINTER(ctypes.c_uint64))),
                    "scl_selftest_atanf_blocks")
        return out

    def save_from_wire(self, values, robot=0, index=0):
        v = _f32(values).reshape(-1)
        if v.size != self.R * self.S:
            raise ValueError("values must hold R*S floats")
        self._check(self._lib.scl_save_from_wire(self._h, _ptr(v, c_float), robot, index), "scl_save_from_wire")

    def detect_intra(self, cur):
        lid, sh, d = c_int(), c_float(), c_double()
        self._check(self._lib.scl_detect_intra(self._h, cur, byref(lid), byref(sh), byref(d)), "scl_detect_intra")
        return lid.value, sh.value, d.value

    def detect_inter(self, cur):
        lid, yaw, d = c_int(), c_float(), c_double()
        self._check(self._lib.scl_detect_inter(self._h, cur, byref(lid), byref(yaw), byref(d)), "scl_detect_inter")
        return lid.value, yaw.value, d.value

    def get_index(self, key):
        r, i = c_int8(), c_int()
        self._check(self._lib.scl_get_index(self._h, key, byref(r), byref(i)), "scl_get_index")
        return r.value, i.value

    def get_size(self, id_in=-1):
        n = self._lib.scl_get_size(self._h, id_in)
        if n < 0:
            raise SclError(n, "scl_get_size")
        return n

    # -- building blocks -------------------------------------------------------
    def make_descriptor(self, points):
        a, n, stride = _cloud(points)
        out = np.empty(self.R * self.S, dtype=np.float32)
        self._check(self._lib.scl_make_descriptor(self._h, a.ctypes.data_as(c_void_p), n, stride,
                                                  _ptr(out, c_float)), "scl_make_descriptor")
        return out

    def save_bulk(self, values, robots=None, indexs=None):
        v = _f32(values).reshape(-1, self.R * self.S)
        count = v.shape[0]
        rp = ip = None
        if robots is not None:
            robots = np.ascontiguousarray(robots, dtype=np.int8); rp = _ptr(robots, c_int8)
        if indexs is not None:
            indexs = np.ascontiguousarray(indexs, dtype=np.int32); ip = _ptr(indexs, c_int)
        self._check(self._lib.scl_save_bulk(self._h, _ptr(v, c_float), count, rp, ip), "scl_save_bulk")

    def get_descriptor(self, key):
        out = np.empty(self.R * self.S, dtype=np.float32)
        self._check(self._lib.scl_get_descriptor(self._h, key, _ptr(out, c_float)), "scl_get_descriptor")
        return out.reshape(self.R, self.S)

    def get_ringkey(self, key):
        out = np.empty(self.R, dtype=np.float32)
        self._check(self._lib.scl_get_ringkey(self._h, key, _ptr(out, c_float)), "scl_get_ringkey")
        return out

    def get_sectorkey(self, key):
        out = np.empty(self.S, dtype=np.float64)
        self._check(self._lib.scl_get_sectorkey(self._h, key, _ptr(out, c_double)), "scl_get_sectorkey")
        return out

    def get_descriptors(self, first, count):
        out = np.empty((max(count, 1), self.R * self.S), dtype=np.float32)
        self._check(self._lib.scl_get_descriptors(self._h, first, count, _ptr(out, c_float)), "scl_get_descriptors")
        return out[:count].reshape(count, self.R, self.S)

    def find_key(self, robot, index):
        """reverse of get_index: database key of (robot, index) or -1 (DM.h:1281-1284)"""
        k = c_int()
        self._check(self._lib.scl_find_key(self._h, robot, index, byref(k)), "scl_find_key")
        return k.value

    def db_dump(self, path):
        self._check(self._lib.scl_db_dump_file(self._h, os.fsencode(path)), "scl_db_dump_file")

    def db_load(self, path):
        n = c_int()
        self._check(self._lib.scl_db_load_file(self._h, os.fsencode(path), byref(n)), "scl_db_load_file")
        return n.value

    def matrix_to_pose(self, T):
        Tm = _f32(T).reshape(16)
        v = [c_float() for _ in range(6)]
        self._check(self._lib.scl_matrix_to_pose(_ptr(Tm, c_float), *[byref(x) for x in v]), "scl_matrix_to_pose")
        return tuple(x.value for x in v)

    def loop_pose_between(self, T_icp, pose_cur, pose_pre):
        """DM.h:1130-1141: (translation xyz + quaternion xyzw, roll/pitch/yaw) of poseFrom.between(poseTo)"""
        Tm = _f32(T_icp).reshape(16); pc = _f32(pose_cur).reshape(6); pp = _f32(pose_pre).reshape(6)
        out = np.empty(7, np.float64); rpy = np.empty(3, np.float64)
        self._check(self._lib.scl_loop_pose_between(_ptr(Tm, c_float), _ptr(pc, c_float), _ptr(pp, c_float), _ptr(out, c_double), _ptr(rpy, c_double)),
                    "scl_loop_pose_between")
        return out, rpy

    def stage_query(self, values):
        v = _f32(values).reshape(-1)
        if v.size != self.R * self.S:
            raise ValueError("values must hold R*S floats")
        self._check(self._lib.scl_stage_query(self._h, _ptr(v, c_float)), "scl_stage_query")

    def ringkey_topk(self, query, lo, hi, k):
        idx = np.empty(k, dtype=np.int32); d2 = np.empty(k, dtype=np.float32); found = c_int()
        self._check(self._lib.scl_ringkey_topk(self._h, query, lo, hi, k, _ptr(idx, c_int), _ptr(d2, c_float),
                                               byref(found)), "scl_ringkey_topk")
        return idx, d2, found.value

    def sc_distance_batch(self, query, cand=None, n=None):
        if cand is not None:
            cand = np.ascontiguousarray(cand, dtype=np.int32); n = cand.size; cp = _ptr(cand, c_int)
        else:
            cp = None
            if n is None:
                raise ValueError("give cand or n")
        dist = np.empty(n, dtype=np.float64); shift = np.empty(n, dtype=np.int32)
        self._check(self._lib.scl_sc_distance_batch(self._h, query, cp, n, _ptr(dist, c_double),
                                                    _ptr(shift, c_int)), "scl_sc_distance_batch")
        return dist, shift

    def sc_distance_matrix(self, queries, lo, hi):
        """exact fp64 distance and shift of every (queries[r], keyframe lo..hi-1) pair: two arrays of shape (len(queries), hi - lo)"""
        q = np.ascontiguousarray(queries, dtype=np.int32)
        n = max(0, int(hi) - int(lo))
        dist = np.empty((q.size, n), dtype=np.float64); shift = np.empty((q.size, n), dtype=np.int32)
        self._check(self._lib.scl_sc_distance_matrix(self._h, _ptr(q, c_int), q.size, int(lo), int(hi), _ptr(dist, c_double), _ptr(shift, c_int)),
                    "scl_sc_distance_matrix")
        return dist, shift

    def detect_full(self, cur):
        lid, nn, sh, d = c_int(), c_int(), c_int(), c_double()
        self._check(self._lib.scl_detect_full(self._h, cur, byref(lid), byref(nn), byref(sh), byref(d)),
                    "scl_detect_full")
        return lid.value, nn.value, sh.value, d.value

    def detect_full_range(self, query, lo, hi):
        nn, sh, d = c_int(), c_int(), c_double()
        self._check(self._lib.scl_detect_full_range(self._h, query, lo, hi, byref(nn), byref(sh), byref(d)),
                    "scl_detect_full_range")
        return nn.value, sh.value, d.value

    def detect_full_submit(self, query, lo, hi):
        t = c_int()
        self._check(self._lib.scl_detect_full_submit(self._h, query, lo, hi, byref(t)), "scl_detect_full_submit")
        return t.value

    def detect_full_submit_many(self, queries, lo, hi):
        """several database keyframes as queries, up to four per launch; returns one ticket per query"""
        q = np.ascontiguousarray(queries, dtype=np.int32)
        m = q.shape[0]
        lo_ = np.ascontiguousarray(np.broadcast_to(np.asarray(lo, dtype=np.int32), (m,)))
        hi_ = np.ascontiguousarray(np.broadcast_to(np.asarray(hi, dtype=np.int32), (m,)))
        t = np.zeros(m, dtype=np.int32)
        self._check(self._lib.scl_detect_full_submit_many(self._h, _ptr(q, c_int), _ptr(lo_, c_int), _ptr(hi_, c_int), m, _ptr(t, c_int)),
                    "scl_detect_full_submit_many")
        return [int(x) for x in t]

    def detect_full_stream(self, queries, lo, hi, scans_per_launch=2, launches_in_flight=2):
        """a backlog of scans through the native submit / collect pipeline; returns (nn_idx, shift, dist) arrays"""
        q = np.ascontiguousarray(queries, dtype=np.int32)
        m = q.shape[0]
        lo_ = np.ascontiguousarray(np.broadcast_to(np.asarray(lo, dtype=np.int32), (m,)))
        hi_ = np.ascontiguousarray(np.broadcast_to(np.asarray(hi, dtype=np.int32), (m,)))
        nn = np.empty(max(m, 1), np.int32); sh = np.empty(max(m, 1), np.int32); d = np.empty(max(m, 1), np.float64)
        self._check(self._lib.scl_detect_full_stream(self._h, _ptr(q, c_int), _ptr(lo_, c_int), _ptr(hi_, c_int), m,
                                                     scans_per_launch, launches_in_flight, _ptr(nn, c_int), _ptr(sh, c_int),
                                                     _ptr(d, c_double)), "scl_detect_full_stream")
        return nn[:m], sh[:m], d[:m]

    def detect_full_collect(self, ticket):
        nn, sh, d = c_int(), c_int(), c_double()
        self._check(self._lib.scl_detect_full_collect(self._h, ticket, byref(nn), byref(sh), byref(d)),
                    "scl_detect_full_collect")
        return nn.value, sh.value, d.value

    def screen_distances_many(self, queries, lo, hi):
        """diagnostic: the screening pass of up to 16 scans in one launch over slots lo..hi-1 -> (approx [nq, n], eps)"""
        q = np.ascontiguousarray(queries, dtype=np.int32)
        n = max(0, min(hi, self.get_size()) - max(lo, 0))
        approx = np.empty((q.shape[0], max(n, 1)), np.float32); eps = c_float()
        self._check(self._lib.scl_screen_distances_many(self._h, _ptr(q, c_int), q.shape[0], lo, hi, _ptr(approx, c_float), byref(eps)),
                    "scl_screen_distances_many")
        return approx.reshape(-1)[:q.shape[0] * n].reshape(q.shape[0], n), eps.value

    def screen_distances(self, query, lo, hi):
        """diagnostic: (approximate distances of slots lo..hi-1, surviving slots, eps) of the screening pass"""
        n = max(0, min(hi, self.get_size()) - max(lo, 0))
        approx = np.empty(max(n, 1), np.float32); surv = np.empty(max(n, 1), np.int32); ns = c_int(); eps = c_float()
        self._check(self._lib.scl_screen_distances(self._h, query, lo, hi, _ptr(approx, c_float), _ptr(surv, c_int), byref(ns), byref(eps)),
                    "scl_screen_distances")
        return approx[:n], surv[:ns.value].copy(), eps.value

    def last_topk(self, k):
        idx = np.empty(k, dtype=np.int32); d2 = np.empty(k, dtype=np.float32)
        self._check(self._lib.scl_get_last_topk(self._h, k, _ptr(idx, c_int), _ptr(d2, c_float)), "scl_get_last_topk")
        return idx, d2

    def topk_with_distance(self, query, lo, hi, k):
        idx = np.empty(k, dtype=np.int32); d2 = np.empty(k, dtype=np.float32)
        dist = np.empty(k, dtype=np.float64); shift = np.empty(k, dtype=np.int32); found = c_int()
        self._check(self._lib.scl_topk_with_distance(self._h, query, lo, hi, k, _ptr(idx, c_int), _ptr(d2, c_float),
                                                     _ptr(dist, c_double), _ptr(shift, c_int), byref(found)),
                    "scl_topk_with_distance")
        return idx, d2, dist, shift, found.value

    # -- geometric verification ---------------------------------------------------
    def icp_default_params(self):
        p = IcpParams()
        self._lib.scl_icp_default_params(byref(p))
        return p

    def icp_align(self, src, tgt, params=None):
        s, ns, stride = _cloud(src)
        t, nt, stride_t = _cloud(tgt)
        if stride != stride_t:
            raise ValueError("source and target must share a record layout")
        p = params or self.icp_default_params()
        T = np.empty(16, dtype=np.float32); fit = c_float(); conv = c_int(); it = c_int()
        self._check(self._lib.scl_icp_align(self._h, s.ctypes.data_as(c_void_p), ns, t.ctypes.data_as(c_void_p), nt,
                                            stride, byref(p), _ptr(T, c_float), byref(fit), byref(conv), byref(it)),
                    "scl_icp_align")
        return T.reshape(4, 4), fit.value, bool(conv.value), it.value

    def icp_align_batch(self, src, tgts, params=None):
        """one source against several targets (the loop candidates of one scan), alignments overlapped on the device"""
        s_, ns, stride = _cloud(src)
        arrs = [_cloud(t) for t in tgts]
        for a, n, st in arrs:
            if st != stride:
                raise ValueError("source and targets must share a record layout")
        m = len(arrs)
        p = params or self.icp_default_params()
        ptrs = (c_void_p * max(1, m))(*[a.ctypes.data_as(c_void_p) for a, _, _ in arrs])
        counts = np.array([n for _, n, _ in arrs] or [0], dtype=np.int32)
        T = np.empty((max(1, m), 16), dtype=np.float32); fit = np.zeros(max(1, m), np.float32)
        conv = np.zeros(max(1, m), np.int32); it = np.zeros(max(1, m), np.int32)
        self._check(self._lib.scl_icp_align_batch(self._h, s_.ctypes.data_as(c_void_p), ns, ptrs, _ptr(counts, c_int), m, stride,
                                                  byref(p), _ptr(T, c_float), _ptr(fit, c_float), _ptr(conv, c_int), _ptr(it, c_int)),
                    "scl_icp_align_batch")
        return T[:m].reshape(m, 4, 4), fit[:m], conv[:m].astype(bool), it[:m]

    def nn_correspondences(self, src, tgt):
        s, ns, stride = _cloud(src)
        t, nt, stride_t = _cloud(tgt)
        if stride != stride_t:
            raise ValueError("source and target must share a record layout")
        idx = np.empty(ns, dtype=np.int32); d2 = np.empty(ns, dtype=np.float32)
        self._check(self._lib.scl_nn_correspondences(self._h, s.ctypes.data_as(c_void_p), ns,
                                                     t.ctypes.data_as(c_void_p), nt, stride,
                                                     _ptr(idx, c_int), _ptr(d2, c_float)), "scl_nn_correspondences")
        return idx, d2

    def nn_correspondences_moved(self, src, tgt, T):
        """Correspondences of `src` moved by the 4x4 `T`, searched warm from those of the unmoved cloud."""
        s, ns, stride = _cloud(src)
        t, nt, stride_t = _cloud(tgt)
        if stride != stride_t:
            raise ValueError("source and target must share a record layout")
        Tm = np.ascontiguousarray(T, dtype=np.float32).reshape(16)
        idx = np.empty(ns, dtype=np.int32); d2 = np.empty(ns, dtype=np.float32)
        self._check(self._lib.scl_nn_correspondences_moved(self._h, s.ctypes.data_as(c_void_p), ns,
                                                           t.ctypes.data_as(c_void_p), nt, stride, _ptr(Tm, c_float),
                                                           _ptr(idx, c_int), _ptr(d2, c_float)), "scl_nn_correspondences_moved")
        return idx, d2

    def rigid_svd(self, src, tgt, src_index, tgt_index):
        s, ns, stride = _cloud(src)
        t, nt, stride_t = _cloud(tgt)
        if stride != stride_t:
            raise ValueError("source and target must share a record layout")
        si = np.ascontiguousarray(src_index, dtype=np.int32); ti = np.ascontiguousarray(tgt_index, dtype=np.int32)
        T = np.empty(16, dtype=np.float32)
        self._check(self._lib.scl_rigid_svd(self._h, s.ctypes.data_as(c_void_p), ns, t.ctypes.data_as(c_void_p), nt,
                                            stride, _ptr(si, c_int), _ptr(ti, c_int), si.size, _ptr(T, c_float)),
                    "scl_rigid_svd")
        return T.reshape(4, 4)

    def ransac_correspondences(self, src, tgt, src_index, tgt_index, max_iterations=1000, inlier_threshold=0.25, seed=1):
        """CorrespondenceRejectorSampleConsensus (DM.h:1218-1225); defaults = ransacMaxIter / ransacOutlierTreshold (DM.h:187-188)"""
        s, ns, stride = _cloud(src)
        t, nt, stride_t = _cloud(tgt)
        if stride != stride_t:
            raise ValueError("source and target must share a record layout")
        si = np.ascontiguousarray(src_index, dtype=np.int32); ti = np.ascontiguousarray(tgt_index, dtype=np.int32)
        mask = np.empty(si.size, dtype=np.int32); ninl = c_int(); best = c_int(); T = np.empty(16, dtype=np.float32)
        self._check(self._lib.scl_ransac_correspondences(self._h, s.ctypes.data_as(c_void_p), ns, t.ctypes.data_as(c_void_p), nt,
                                                         stride, _ptr(si, c_int), _ptr(ti, c_int), si.size, max_iterations,
                                                         inlier_threshold, seed, _ptr(mask, c_int), byref(ninl), byref(best),
                                                         _ptr(T, c_float)), "scl_ransac_correspondences")
        return mask, ninl.value, best.value, T.reshape(4, 4)

    def geometric_verification(self, src, tgt, ransac_iterations=1000, inlier_threshold=0.25, inlier_ratio=0.45, seed=1):
        """compute core of geometricVerificationService (DM.h:1211-1243); defaults DM.h:187-189"""
        s, ns, stride = _cloud(src)
        t, nt, stride_t = _cloud(tgt)
        if stride != stride_t:
            raise ValueError("source and target must share a record layout")
        T = np.empty(16, dtype=np.float32); ok = c_int(); nc = c_int(); ni = c_int()
        self._check(self._lib.scl_geometric_verification(self._h, s.ctypes.data_as(c_void_p), ns, t.ctypes.data_as(c_void_p), nt,
                                                         stride, ransac_iterations, inlier_threshold, inlier_ratio, seed,
                                                         _ptr(T, c_float), byref(ok), byref(nc), byref(ni)),
                    "scl_geometric_verification")
        return T.reshape(4, 4), bool(ok.value), nc.value, ni.value

    def voxel_grid(self, cloud, leaf):
        """pcl::VoxelGrid with one leaf size (DM.h:501,503)"""
        a, n, stride = _cloud(cloud)
        out = np.empty_like(a); m = c_int()
        self._check(self._lib.scl_voxel_grid(self._h, a.ctypes.data_as(c_void_p), n, stride, leaf,
                                             out.ctypes.data_as(c_void_p), n, byref(m)), "scl_voxel_grid")
        return out[:m.value].copy()

    def pose_to_matrix(self, x, y, z, roll, pitch, yaw):
        T = np.empty(16, dtype=np.float32)
        self._check(self._lib.scl_pose_to_matrix(x, y, z, roll, pitch, yaw, _ptr(T, c_float)), "scl_pose_to_matrix")
        return T.reshape(4, 4)

    def assemble_submap(self, clouds, transforms, leaf):
        """loopFindNearKeyframes (DM.h:1163-1186): transformed keyframes concatenated, then voxel-filtered"""
        arrs = [np.ascontiguousarray(c, dtype=np.float32) for c in clouds]
        stride = arrs[0].shape[1] * 4 if arrs else 32
        counts = np.array([a.shape[0] for a in arrs], dtype=np.int32)
        ptrs = (c_void_p * max(1, len(arrs)))(*[a.ctypes.data_as(c_void_p) for a in arrs])
        Ts = _f32(np.stack(transforms)).reshape(-1, 16) if arrs else np.zeros((1, 16), np.float32)
        total = int(counts.sum())
        out = np.empty((max(total, 1), stride // 4), dtype=np.float32); m = c_int()
        self._check(self._lib.scl_assemble_submap(self._h, ptrs, _ptr(counts, c_int), _ptr(Ts, c_float), len(arrs), stride,
                                                  leaf, out.ctypes.data_as(c_void_p), total, byref(m)), "scl_assemble_submap")
        return out[:m.value].copy()

    def transform_cloud(self, cloud, T):
        a, n, stride = _cloud(cloud)
        out = np.empty_like(a)
        Tm = _f32(T).reshape(16)
        self._check(self._lib.scl_transform_cloud(self._h, a.ctypes.data_as(c_void_p), n, stride, _ptr(Tm, c_float),
                                                  out.ctypes.data_as(c_void_p)), "scl_transform_cloud")
        return out

    # -- on-device keyframe store (robots[id].keyFrameArray, DM.h:86) -------------
    def keyframe_put(self, robot, index, cloud):
        a, n, stride = _cloud(cloud)
        self._check(self._lib.scl_keyframe_put(self._h, robot, index, a.ctypes.data_as(c_void_p), n, stride), "scl_keyframe_put")

    def keyframe_count(self, robot):
        return self._lib.scl_keyframe_count(self._h, robot)

    def keyframe_get(self, robot, index, floats_per_point=8):
        n = c_int()
        self._check(self._lib.scl_keyframe_get(self._h, robot, index, None, 0, byref(n)), "scl_keyframe_get")
        out = np.empty((max(n.value, 1), floats_per_point), dtype=np.float32)
        self._check(self._lib.scl_keyframe_get(self._h, robot, index, out.ctypes.data_as(c_void_p), n.value, byref(n)), "scl_keyframe_get")
        return out[:n.value].copy()

    def submap_from_store(self, robot, key, search_num, poses, leaf, capacity, floats_per_point=8):
        """loopFindNearKeyframes (DM.h:1163-1186) on stored keyframes; poses[i] belongs to keyframe key-search_num+i"""
        Ts = _f32(np.asarray(poses)).reshape(-1, 16)
        assert Ts.shape[0] == 2 * search_num + 1
        out = np.empty((max(capacity, 1), floats_per_point), dtype=np.float32); m = c_int()
        self._check(self._lib.scl_submap_from_store(self._h, robot, key, search_num, _ptr(Ts, c_float), leaf,
                                                    out.ctypes.data_as(c_void_p), capacity, byref(m)), "scl_submap_from_store")
        return out[:m.value].copy()

    def loop_icp_from_store(self, robot, key_cur, pose_cur, key_pre, search_num, poses_pre, leaf, params=None,
                            min_src_points=300, min_tgt_points=1000):
        """performIntraLoopClosure stage 2 (DM.h:1104-1121) without moving clouds over PCIe"""
        p = params if params is not None else self.icp_default_params()
        Tc = _f32(np.asarray(pose_cur)).reshape(16)
        Tp = _f32(np.asarray(poses_pre)).reshape(-1, 16)
        assert Tp.shape[0] == 2 * search_num + 1
        T = np.empty(16, dtype=np.float32); fit = c_float(); conv = c_int(); it = c_int(); ns = c_int(); nt = c_int()
        self._check(self._lib.scl_loop_icp_from_store(self._h, robot, key_cur, _ptr(Tc, c_float), key_pre, search_num,
                                                      _ptr(Tp, c_float), leaf, byref(p), min_src_points, min_tgt_points,
                                                      _ptr(T, c_float), byref(fit), byref(conv), byref(it), byref(ns), byref(nt)),
                    "scl_loop_icp_from_store")
        return T.reshape(4, 4), fit.value, bool(conv.value), it.value, ns.value, nt.value

    def loop_icp_batch_from_store(self, robot, key_cur, pose_cur, keys_pre, search_num, poses_pre, leaf, params=None,
                                  min_src_points=300, min_tgt_points=1000):
        """stage 2 of performIntraLoopClosure for all loop candidates of one scan, ICP loops fused (BASELINE configs[2])"""
        p = params if params is not None else self.icp_default_params()
        keys = np.ascontiguousarray(keys_pre, dtype=np.int32); m = keys.size
        Tc = _f32(np.asarray(pose_cur)).reshape(16)
        Tp = _f32(np.asarray(poses_pre)).reshape(m, 2 * search_num + 1, 16)
        T = np.empty((max(m, 1), 16), np.float32); fit = np.zeros(max(m, 1), np.float32)
        conv = np.zeros(max(m, 1), np.int32); it = np.zeros(max(m, 1), np.int32); ns = c_int(); nt = np.zeros(max(m, 1), np.int32)
        self._check(self._lib.scl_loop_icp_batch_from_store(self._h, robot, key_cur, _ptr(Tc, c_float), m, _ptr(keys, c_int), search_num,
                                                            _ptr(Tp, c_float), leaf, byref(p), min_src_points, min_tgt_points,
                                                            _ptr(T, c_float), _ptr(fit, c_float), _ptr(conv, c_int), _ptr(it, c_int),
                                                            byref(ns), _ptr(nt, c_int)), "scl_loop_icp_batch_from_store")
        return T[:m].reshape(m, 4, 4), fit[:m], conv[:m].astype(bool), it[:m], ns.value, nt[:m]

    def geometric_verification_from_store(self, src, src_leaf, robot, key_pre, search_num, poses_pre, leaf,
                                          ransac_iterations=1000, inlier_threshold=0.25, inlier_ratio=0.45, seed=1,
                                          min_src_points=300, min_tgt_points=1000):
        """geometricVerificationService (DM.h:1189-1268) against a submap of stored keyframes"""
        a, n, stride = _cloud(src)
        Tp = _f32(np.asarray(poses_pre)).reshape(-1, 16)
        assert Tp.shape[0] == 2 * search_num + 1
        T = np.empty(16, dtype=np.float32); ok = c_int(); ns = c_int(); nt = c_int(); nc = c_int(); ni = c_int()
        self._check(self._lib.scl_geometric_verification_from_store(
            self._h, a.ctypes.data_as(c_void_p), n, stride, src_leaf, robot, key_pre, search_num, _ptr(Tp, c_float), leaf,
            min_src_points, min_tgt_points, ransac_iterations, inlier_threshold, inlier_ratio, seed,
            _ptr(T, c_float), byref(ok), byref(ns), byref(nt), byref(nc), byref(ni)), "scl_geometric_verification_from_store")
        return T.reshape(4, 4), bool(ok.value), ns.value, nt.value, nc.value, ni.value

    # -- measurement ------------------------------------------------------------
    def profile_enable(self, on=True):
        """0/False off, 1/True every kernel family, 2 the SC-distance kernel only (lightest)"""
        self._check(self._lib.scl_profile_enable(self._h, int(on)), "scl_profile_enable")

    def profile_reset(self):
        self._check(self._lib.scl_profile_reset(self._h), "scl_profile_reset")

    def profile(self):
        p = SclProfile()
        self._check(self._lib.scl_profile_get(self._h, byref(p)), "scl_profile_get")
        return {name: getattr(p, name) for name, _ in SclProfile._fields_}

    def alignment_stats(self, reset=False):
        """(pairs aligned by the full-database pass, pairs whose first shift needed the exact fp64 evaluation)"""
        a, b = c_uint64(0), c_uint64(0)
        self._check(self._lib.scl_alignment_stats(self._h, byref(a), byref(b), int(bool(reset))), "scl_alignment_stats")
        return int(a.value), int(b.value)

    def survivor_stats(self, reset=False):
        """(scans through an exact pass, keyframes scored exactly for them, largest count of one scan) since the last reset"""
        q, s, m = c_uint64(0), c_uint64(0), c_uint64(0)
        self._check(self._lib.scl_survivor_stats(self._h, byref(q), byref(s), byref(m), int(bool(reset))), "scl_survivor_stats")
        return int(q.value), int(s.value), int(m.value)

    def device_name(self):
        buf = ctypes.create_string_buffer(256)
        self._check(self._lib.scl_device_name(self._h, buf, 256), "scl_device_name")
        return buf.value.decode()


class ScanContextDescriptor:
    """Python mirror of ``scan_context_descriptor : scan_descriptor`` (D.h:1304-1801).

    Constructor arguments, method names and return conventions follow the reference;
    all work is done by the GPU engine behind the C ABI.
    """

    def __init__(self, numRing=20, numSector=60, numCandidates=3, distThres=0.14, lidarHeight=1.65,
                 maxRadius=80.0, numExcludeRecent=100, treeMakingPeriod=10, searchRatio=0.1, **engine_kw):
        self.engine = ScanContextEngine(numRing, numSector, numCandidates, distThres, lidarHeight, maxRadius,
                                        numExcludeRecent, treeMakingPeriod, searchRatio, **engine_kw)

    def makeAndSaveDescriptorAndKey(self, scan, robot, index):          # D.h:25
        return self.engine.make_and_save(scan, robot, index)

    def saveDescriptorAndKey(self, descriptorMat, robot, index):       # D.h:27
        self.engine.save_from_wire(descriptorMat, robot, index)

    def detectIntraLoopClosureID(self, currentPtr):                    # D.h:29
        lid, shift, _ = self.engine.detect_intra(currentPtr)
        return lid, shift

    def detectInterLoopClosureID(self, currentPtr):                    # D.h:31
        lid, yaw, _ = self.engine.detect_inter(currentPtr)
        return lid, yaw

    def getIndex(self, key):                                           # D.h:33
        return self.engine.get_index(key)

    def getSize(self, idIn=-1):                                        # D.h:35
        return self.engine.get_size(idIn)
