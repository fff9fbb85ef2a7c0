"""ctypes front-end of the CPU oracle (``oracle/libgendr_oracle.so``).

TEST INFRASTRUCTURE ONLY.  Importable from ``tests/``, ``__graft_entry__.smoke``
and ``bench.py``'s ``cpu_baseline`` leg; the product package ``gendr_amd`` never
imports this module (a test enforces that).

PARITY PINNED to outputs of the reference's own kernels run on the GPU box
(``oracle/build_ref.py`` -> ``oracle/_ref``, ``oracle/ref_gpu.py``,
``tests/test_gpu_reference_pin.py``); see ``oracle/gendr_oracle.h``.

The functions mirror what the reference's Python layer does around the native
call (``gendr/functional/renderer.py:130-153`` forward allocs and background
pre-fill, ``:191-197`` backward allocs) so callers hand over plain arrays.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgendr_oracle.so")

DIST_FUNCS = {
    'hard': 0, 'heaviside': 0, 'uniform': 1, 'cubic_hermite': 2, 'wigner_semicircle': 3,
    'gaussian': 4, 'laplace': 5, 'logistic': 6, 'gudermannian': 7, 'hyperbolic_secant': 7,
    'cauchy': 8, 'reciprocal': 9, 'gumbel_max': 10, 'gumbel_min': 11, 'exponential': 12,
    'exponential_rev': 13, 'gamma': 14, 'gamma_rev': 15, 'levy': 16, 'levy_rev': 17,
}
ALPHA_FUNCS = {
    'hard': 0, 'max': 1, 'probabilistic': 2, 'einstein': 3, 'hamacher': 4, 'frank': 5,
    'yager': 6, 'aczel_alsina': 7, 'dombi': 8, 'schweizer_sklar': 9,
}
RGB_FUNCS = {'hard': 0, 'softmax': 1}
TEXTURE_TYPES = {'surface': 0, 'vertex': 1}


class _COpts(ctypes.Structure):
    _fields_ = [
        ("image_size", ctypes.c_int), ("dist_func", ctypes.c_int), ("dist_scale", ctypes.c_float),
        ("dist_squared", ctypes.c_int), ("dist_shape", ctypes.c_float), ("dist_shift", ctypes.c_float),
        ("dist_eps", ctypes.c_float), ("aggr_alpha_func", ctypes.c_int),
        ("aggr_alpha_t_conorm_p", ctypes.c_float), ("aggr_rgb_func", ctypes.c_int),
        ("aggr_rgb_eps", ctypes.c_float), ("aggr_rgb_gamma", ctypes.c_float),
        ("near_", ctypes.c_float), ("far_", ctypes.c_float), ("double_side", ctypes.c_int),
        ("texture_type", ctypes.c_int), ("texel_mode", ctypes.c_int), ("num_threads", ctypes.c_int),
        ("prob_threshold_scale", ctypes.c_float),
    ]


def build(force=False):
    """(Re)build the shared library with the committed Makefile."""
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
            for f in ("gendr_oracle.c", "gendr_oracle_body.inc", "gendr_oracle.h", "Makefile")):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        f, d, i = ctypes.c_float, ctypes.c_double, ctypes.c_int
        for name, ft in (("", f), ("_f64", d)):
            getattr(L, "gendr_oracle_sigmoid_forward" + name).restype = ft
            getattr(L, "gendr_oracle_sigmoid_forward" + name).argtypes = [i, ft, ft, ft, ft, ft]
            getattr(L, "gendr_oracle_sigmoid_backward" + name).restype = ft
            getattr(L, "gendr_oracle_sigmoid_backward" + name).argtypes = [i, ft, ft, ft, ft, ft]
            getattr(L, "gendr_oracle_t_conorm_forward" + name).restype = ft
            getattr(L, "gendr_oracle_t_conorm_forward" + name).argtypes = [i, ft, ft, i, ft]
            getattr(L, "gendr_oracle_t_conorm_backward" + name).restype = ft
            getattr(L, "gendr_oracle_t_conorm_backward" + name).argtypes = [i, ft, ft, i, ft]
        vp = ctypes.c_void_p
        po = ctypes.POINTER(_COpts)
        for sfx in ("_f32", "_f64"):
            getattr(L, "gendr_oracle_face_info" + sfx).argtypes = [vp, vp, i, i]
            getattr(L, "gendr_oracle_face_info" + sfx).restype = None
            getattr(L, "gendr_oracle_forward" + sfx).argtypes = [vp, vp, vp, vp, vp, i, i, i, po]
            getattr(L, "gendr_oracle_forward" + sfx).restype = None
            getattr(L, "gendr_oracle_backward" + sfx).argtypes = [vp] * 10 + [i, i, i, po]
            getattr(L, "gendr_oracle_backward" + sfx).restype = None
        L.gendr_oracle_count_pairs_f32.argtypes = [vp, vp, i, i, po]
        L.gendr_oracle_count_pairs_f32.restype = ctypes.c_longlong
        L.gendr_oracle_max_threads.restype = i
        L.gendr_oracle_set_libm_jitter.restype = None
        L.gendr_oracle_set_libm_jitter.argtypes = [i]
        _lib = L
    return _lib


def _opt(value, table):
    return value if isinstance(value, (int, np.integer)) else table[value]


def make_opts(image_size=256, dist_func='uniform', dist_scale=1e-2, dist_squared=False, dist_shape=None,
              dist_shift=None, dist_eps=1e4, aggr_alpha_func='probabilistic', aggr_alpha_t_conorm_p=None,
              aggr_rgb_func='softmax', aggr_rgb_eps=1e-3, aggr_rgb_gamma=1e-3, near=1, far=100,
              double_side=True, texture_type='surface', texel_mode=0, num_threads=0, prob_threshold_scale=1.0):
    """Same option names and defaults as ``gendr.functional.render``
    (``functional/renderer.py:239-264``); ``None`` parameters become 0.0."""
    o = _COpts()
    o.image_size = int(image_size)
    o.dist_func = int(_opt(dist_func, DIST_FUNCS))
    o.dist_scale = float(dist_scale)
    o.dist_squared = int(bool(dist_squared))
    o.dist_shape = 0.0 if dist_shape is None else float(dist_shape)
    o.dist_shift = 0.0 if dist_shift is None else float(dist_shift)
    o.dist_eps = float(dist_eps)
    o.aggr_alpha_func = int(_opt(aggr_alpha_func, ALPHA_FUNCS))
    o.aggr_alpha_t_conorm_p = 0.0 if aggr_alpha_t_conorm_p is None else float(aggr_alpha_t_conorm_p)
    o.aggr_rgb_func = int(_opt(aggr_rgb_func, RGB_FUNCS))
    o.aggr_rgb_eps = float(aggr_rgb_eps)
    o.aggr_rgb_gamma = float(aggr_rgb_gamma)
    o.near_ = float(near)
    o.far_ = float(far)
    o.double_side = int(bool(double_side))
    o.texture_type = int(_opt(texture_type, TEXTURE_TYPES))
    o.texel_mode = int(texel_mode)
    o.num_threads = int(num_threads)
    o.prob_threshold_scale = float(prob_threshold_scale)     # sensitivity analysis only, see gendr_oracle.h
    return o


def _sfx(dtype):
    return "_f32" if np.dtype(dtype) == np.float32 else "_f64"


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def face_info(faces, dtype=np.float32):
    faces = np.ascontiguousarray(faces, dtype=dtype).reshape(faces.shape[0], faces.shape[1], 9)
    B, nf = faces.shape[:2]
    info = np.zeros((B, nf, 27), dtype=dtype)
    getattr(lib(), "gendr_oracle_face_info" + _sfx(dtype))(_p(faces), _p(info), B, nf)
    return info


def forward(faces, textures, opts, background=(0., 0., 0.), dtype=np.float32):
    """Returns dict(rgba [B,4,is,is], aggrs_info [B,2,is,is], faces_info [B,nf,27])."""
    B, nf = faces.shape[:2]
    faces = np.ascontiguousarray(faces, dtype=dtype).reshape(B, nf, 9)
    textures = np.ascontiguousarray(textures, dtype=dtype)
    T = textures.shape[2]
    isz = opts.image_size
    info = face_info(faces, dtype)
    aggrs = np.zeros((B, 2, isz, isz), dtype=dtype)
    rgba = np.ones((B, 4, isz, isz), dtype=dtype)
    for k in range(3):
        rgba[:, k] *= background[k]
    getattr(lib(), "gendr_oracle_forward" + _sfx(dtype))(
        _p(faces), _p(textures), _p(info), _p(aggrs), _p(rgba), B, nf, T, ctypes.byref(opts))
    return dict(rgba=rgba, aggrs_info=aggrs, faces_info=info, faces=faces, textures=textures)


def backward(fwd, grad_rgba, opts, dtype=np.float32):
    """fwd = dict returned by :func:`forward`.  Returns (grad_faces [B,nf,9],
    grad_textures [B,nf,T,3], abs_faces, abs_textures)."""
    faces, textures = fwd["faces"], fwd["textures"]
    B, nf = faces.shape[:2]
    T = textures.shape[2]
    grad_rgba = np.ascontiguousarray(grad_rgba, dtype=dtype)
    gf = np.zeros((B, nf, 9), dtype=dtype)
    gt = np.zeros(textures.shape, dtype=dtype)
    af = np.zeros_like(gf)
    at = np.zeros_like(gt)
    getattr(lib(), "gendr_oracle_backward" + _sfx(dtype))(
        _p(faces), _p(textures), _p(fwd["rgba"]), _p(fwd["faces_info"]), _p(fwd["aggrs_info"]),
        _p(gf), _p(gt), _p(grad_rgba), _p(af), _p(at), B, nf, T, ctypes.byref(opts))
    return gf, gt, af, at


def count_pairs(faces, opts):
    B, nf = faces.shape[:2]
    faces = np.ascontiguousarray(faces, dtype=np.float32).reshape(B, nf, 9)
    info = face_info(faces, np.float32)
    return int(lib().gendr_oracle_count_pairs_f32(_p(faces), _p(info), B, nf, ctypes.byref(opts)))


class libm_jitter(object):
    """``with libm_jitter(+1): ...``: every single-precision libm result of the float oracle moves one ulp up (-1: down)
    inside the block -- the sensitivity runs of tests/criteria.py."""
    def __init__(self, j):
        self.j = int(j)

    def __enter__(self):
        lib().gendr_oracle_set_libm_jitter(self.j)

    def __exit__(self, *exc):
        lib().gendr_oracle_set_libm_jitter(0)
        return False


def max_threads():
    return int(lib().gendr_oracle_max_threads())


def sigmoid_forward(fid, sign, x, scale, shape=0.0, shift=0.0, f64=False):
    fn = lib().gendr_oracle_sigmoid_forward_f64 if f64 else lib().gendr_oracle_sigmoid_forward
    return fn(int(fid), sign, x, scale, shape, shift)


def sigmoid_backward(fid, sign, x, scale, shape=0.0, shift=0.0, f64=False):
    fn = lib().gendr_oracle_sigmoid_backward_f64 if f64 else lib().gendr_oracle_sigmoid_backward
    return fn(int(fid), sign, x, scale, shape, shift)


def t_conorm_forward(tid, a, b, face_id=0, p=0.0, f64=False):
    fn = lib().gendr_oracle_t_conorm_forward_f64 if f64 else lib().gendr_oracle_t_conorm_forward
    return fn(int(tid), a, b, int(face_id), p)


def t_conorm_backward(tid, a_all, b_cur, nf=0, p=0.0, f64=False):
    fn = lib().gendr_oracle_t_conorm_backward_f64 if f64 else lib().gendr_oracle_t_conorm_backward
    return fn(int(tid), a_all, b_cur, int(nf), p)
