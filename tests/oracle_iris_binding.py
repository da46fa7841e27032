"""ctypes binding of oracle/iris_oracle.c -- the CPU checker of the LiDAR-Iris building blocks (test infrastructure only)."""
import ctypes
from ctypes import POINTER, byref, c_double, c_float, c_int, c_uint8, c_void_p

import numpy as np

import oracle_binding as ob


class IrisoConfig(ctypes.Structure):
    _fields_ = [("rows", c_int), ("cols", c_int), ("nscan", c_int), ("nscale", c_int), ("min_wavelength", c_int), ("mult", c_float), ("sigma_onf", c_float)]


def config(rows=80, cols=360, nscan=64, nscale=4, min_wavelength=18, mult=1.6, sigma_onf=0.75):
    return IrisoConfig(rows, cols, nscan, nscale, min_wavelength, mult, sigma_onf)


def _L():
    L = ob.load()
    u8 = POINTER(c_uint8)
    L.iriso_make_image.argtypes = [POINTER(IrisoConfig), c_void_p, c_int, c_int, u8, POINTER(c_float)]
    L.iriso_encode.argtypes = [POINTER(IrisoConfig), u8, u8, u8]
    L.iriso_responses.argtypes = [POINTER(IrisoConfig), u8, POINTER(c_double)]
    L.iriso_hamming.argtypes = [POINTER(IrisoConfig), u8, u8, u8, u8, c_int, POINTER(c_float), POINTER(c_int)]
    L.iriso_hamming_all.argtypes = [POINTER(IrisoConfig), u8, u8, u8, u8, POINTER(c_float), POINTER(c_int)]
    L.iriso_fft_match.restype = c_int
    L.iriso_fft_match.argtypes = [c_int, c_int, u8, u8, POINTER(c_float), POINTER(c_double)]
    L.iriso_compare.argtypes = [POINTER(IrisoConfig), c_int, u8, u8, u8, u8, u8, u8, POINTER(c_float), POINTER(c_int), POINTER(c_int)]
    return L


def _p(a):
    return a.ctypes.data_as(POINTER(c_uint8))


def make_image(cfg, cloud):
    a = np.ascontiguousarray(cloud, np.float32)
    img = np.empty((cfg.rows, cfg.cols), np.uint8); key = np.empty(cfg.rows, np.float32)
    _L().iriso_make_image(byref(cfg), a.ctypes.data_as(c_void_p), a.shape[0], a.shape[1] * 4, _p(img), key.ctypes.data_as(POINTER(c_float)))
    return img, key


def encode(cfg, image):
    tr = 2 * cfg.nscale * cfg.rows
    img = np.ascontiguousarray(image, np.uint8)
    T = np.empty((tr, cfg.cols), np.uint8); M = np.empty((tr, cfg.cols), np.uint8)
    _L().iriso_encode(byref(cfg), _p(img), _p(T), _p(M))
    return T, M


def responses(cfg, image):
    img = np.ascontiguousarray(image, np.uint8)
    out = np.empty((cfg.nscale, cfg.rows, cfg.cols, 2), np.float64)
    _L().iriso_responses(byref(cfg), _p(img), out.ctypes.data_as(POINTER(c_double)))
    return out


def hamming(cfg, T1, M1, T2, M2, scale):
    d, b = c_float(), c_int()
    _L().iriso_hamming(byref(cfg), _p(T1), _p(M1), _p(T2), _p(M2), scale, byref(d), byref(b))
    return d.value, b.value


def hamming_all(cfg, T1, M1, T2, M2):
    d, b = c_float(), c_int()
    _L().iriso_hamming_all(byref(cfg), _p(T1), _p(M1), _p(T2), _p(M2), byref(d), byref(b))
    return d.value, b.value


def fft_match(rows, cols, im0, im1):
    """fftMatch(im0, im1): (centre x as float32, compatible, [rot_scale.x, .y, angle, scale, tr.x, tr.y])"""
    a = np.ascontiguousarray(im0, np.uint8); b = np.ascontiguousarray(im1, np.uint8)
    cx = c_float(); dbg = (c_double * 6)()
    ok = _L().iriso_fft_match(rows, cols, _p(a), _p(b), byref(cx), dbg)
    return np.float32(cx.value), ok, list(dbg)


def compare(cfg, match_num, img1, T1, M1, img2, T2, M2):
    """compare(img1, img2): (distance, bias, [first estimate, second estimate])"""
    d, b = c_float(), c_int()
    sh = (c_int * 2)()
    arrs = [np.ascontiguousarray(x, np.uint8) for x in (img1, T1, M1, img2, T2, M2)]
    _L().iriso_compare(byref(cfg), match_num, *[_p(x) for x in arrs], byref(d), byref(b), sh)
    return d.value, b.value, [sh[0], sh[1]]
