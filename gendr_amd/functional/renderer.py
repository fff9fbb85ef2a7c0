"""Autograd surface of the generalized soft rasterizer on MI355X.

Drop-in for ``gendr/functional/renderer.py`` of the reference: the same
``GenDRFunction`` positional signature (``:13-39``), the same ``render()``
keyword surface and defaults (``:239-264``), string or integer ids for
``dist_func`` / ``aggr_alpha_func`` / ``aggr_rgb_func`` (``:91-94,106-109,116-119``).
The native work is done by ``libgendr_hip.so`` through ``gendr_amd._native``
(C ABI, ``include/gendr_hip.h``); there is no PyTorch or CPU fallback.

Host-side differences from the reference, all deliberate:
  * inputs are made contiguous instead of cloned (``:130-131``); outputs are
    ``torch.empty`` and fully written by the kernel instead of ones/zeros plus
    three background multiplies (``:136-151``);
  * ``None`` for ``dist_shape`` / ``dist_shift`` / ``aggr_alpha_t_conorm_p``
    means 0.0 (the reference's pybind signature rejects ``None`` although it is
    the documented default);
  * invalid option values raise ``ValueError`` on the host instead of printing
    from the device and producing NaN (``kernel.cu:296-299,491-494`` ...);
  * CPU tensors raise ``TypeError`` (the reference's check at ``:265`` compares a
    ``torch.device`` with a string and never fires).
"""
import ctypes
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _native

DIST_FUNC_IDS = {
    'hard': 0, 'heaviside': 0,
    'uniform': 1,
    'cubic_hermite': 2,
    'wigner_semicircle': 3,
    'gaussian': 4,
    'laplace': 5,
    'logistic': 6,
    'gudermannian': 7, 'hyperbolic_secant': 7,
    'cauchy': 8,
    'reciprocal': 9,
    'gumbel_max': 10,
    'gumbel_min': 11,
    'exponential': 12,
    'exponential_rev': 13,
    'gamma': 14,
    'gamma_rev': 15,
    'levy': 16,
    'levy_rev': 17,
}
AGGR_ALPHA_FUNC_IDS = {
    'hard': 0, 'max': 1, 'probabilistic': 2, 'einstein': 3, 'hamacher': 4,
    'frank': 5, 'yager': 6, 'aczel_alsina': 7, 'dombi': 8, 'schweizer_sklar': 9,
}
AGGR_RGB_FUNC_IDS = {'hard': 0, 'softmax': 1}
TEXTURE_TYPE_IDS = {'surface': 0, 'vertex': 1}

_TEXEL_MODES = {'reference': 0, 'clamp': 1}

# bench.py sets this to four torch.cuda.Event objects to time the native forward / backward calls with HIP
# events on the launch stream: [fwd start, fwd end, bwd start, bwd end].  None = no timing.
PROFILE_EVENTS = None


def _lookup(value, table, what):
    if isinstance(value, bool):
        raise ValueError('%s must be a name or an integer id, got %r' % (what, value))
    if isinstance(value, int):
        return value
    try:
        return table[value]
    except (KeyError, TypeError):
        raise KeyError('unknown %s %r; known: %s' % (what, value, sorted(table)))


def _flt(value):
    return 0.0 if value is None else float(value)


def make_params(image_size, background_color, dist_func, dist_scale, dist_squared, dist_shape, dist_shift,
                dist_eps, aggr_alpha_func, aggr_alpha_t_conorm_p, aggr_rgb_func, aggr_rgb_eps, aggr_rgb_gamma,
                near, far, double_side, texture_type, background_from_buffer=False):
    """Normalises the option set (names -> ids, None -> 0.0, numpy scalars -> float)
    and applies the reference's host-side asserts (``functional/renderer.py:96,101``)."""
    assert dist_scale is not None and dist_scale >= 0, dist_scale   # a negative scale is invalid
    assert dist_eps >= 1, dist_eps                                   # ignoring too close to the edge makes no sense
    p = _native.GendrParams()
    p.image_size = int(image_size)
    p.dist_func = _lookup(dist_func, DIST_FUNC_IDS, 'dist_func')
    p.dist_scale = float(dist_scale)
    p.dist_squared = 1 if dist_squared else 0
    p.dist_shape = _flt(dist_shape)
    p.dist_shift = _flt(dist_shift)
    p.dist_eps = float(dist_eps)
    p.aggr_alpha_func = _lookup(aggr_alpha_func, AGGR_ALPHA_FUNC_IDS, 'aggr_alpha_func')
    p.aggr_alpha_t_conorm_p = _flt(aggr_alpha_t_conorm_p)
    p.aggr_rgb_func = _lookup(aggr_rgb_func, AGGR_RGB_FUNC_IDS, 'aggr_rgb_func')
    p.aggr_rgb_eps = float(aggr_rgb_eps)
    p.aggr_rgb_gamma = float(aggr_rgb_gamma)
    p.near_ = float(near)
    p.far_ = float(far)
    p.double_side = 1 if double_side else 0
    p.texture_type = _lookup(texture_type, TEXTURE_TYPE_IDS, 'texture_type')
    bg = list(background_color)
    p.background[0], p.background[1], p.background[2] = float(bg[0]), float(bg[1]), float(bg[2])
    p.background_from_buffer = 1 if background_from_buffer else 0
    p.texel_mode = _TEXEL_MODES[os.environ.get('GENDR_TEXEL_MODE', 'reference')]
    p.cull = 0 if os.environ.get('GENDR_CULL', '1') == '0' else 1
    p.deterministic = 1 if os.environ.get('GENDR_DETERMINISTIC', '0') == '1' else 0
    p.skip_unlisted_aux = 0
    p.pool_entries_max = int(os.environ.get('GENDR_POOL_ENTRIES_MAX', '0'))
    p.pair_hints = int(os.environ.get('GENDR_PAIR_HINTS', '0'))     # 0 automatic, 1 on, -1 off (include/gendr_hip.h, ABI 6)
    p.loose_faces = int(os.environ.get('GENDR_LOOSE_FACES', '0'))   # likewise
    p.team = int(os.environ.get('GENDR_TEAM', '0'))                 # likewise (ABI 7: team kernels)
    return p


_CPP_AUTOGRAD = os.environ.get('GENDR_CPP_AUTOGRAD', '1') != '0'
_PARAMS_CACHE = {}
_ENV_KEYS = ('GENDR_TEXEL_MODE', 'GENDR_CULL', 'GENDR_DETERMINISTIC', 'GENDR_POOL_ENTRIES_MAX', 'GENDR_PAIR_HINTS', 'GENDR_LOOSE_FACES', 'GENDR_TEAM',
             'GENDR_SKIP_UNLISTED_AUX', 'GENDR_FUSED_CLEAR')


def _key_scalar(v):
    """An option value as the plain Python scalar make_params() would make of it.  Tensors (0-d, nn.Parameter) and numpy
    arrays hash by IDENTITY, so keyed on the object an in-place update (sigma.mul_(0.9), optimizer.step()) would keep
    hitting the entry of the old value; their value is read here, which costs one .item() -- no cache would cost
    the whole normalisation."""
    if isinstance(v, bool):
        return ('bool', v)                              # True == 1 as a key, but make_params() rejects a bool where an id is expected
    if v is None or isinstance(v, (str, int, float)):
        return v
    try:
        return float(v)                                 # numpy scalars, 0-d tensors / arrays: what the reference's __float__ does
    except (TypeError, ValueError):
        raise TypeError('unhashable option')


def _params_bytes(image_size, background_color, *options):
    """(gendr_params as bytes, fused gradient clear?) for the C++ node; cached per option set and environment -- an optimisation loop
    renders with the same options thousands of times, and normalising them costs the host more than the launch.  The key is built
    from the options' VALUES (see _key_scalar), never from the objects."""
    env = tuple(os.environ.get(k) for k in _ENV_KEYS)
    try:
        key = (_key_scalar(image_size), tuple(_key_scalar(c) for c in background_color), tuple(_key_scalar(o) for o in options), env)
        hit = _PARAMS_CACHE.get(key)
    except TypeError:                                   # an option no scalar can be made of: no caching, make_params() reports it
        key, hit = None, None
    if hit is None:
        p = make_params(image_size, background_color, *options)
        # aggrs_info stays inside the node (saved for backward, which never looks at tiles no face reaches)
        p.skip_unlisted_aux = 1 if os.environ.get('GENDR_SKIP_UNLISTED_AUX', '1') != '0' else 0
        hit = (bytes(p), os.environ.get('GENDR_FUSED_CLEAR', '1') != '0')
        if key is not None:
            if len(_PARAMS_CACHE) > 256:
                _PARAMS_CACHE.clear()
            _PARAMS_CACHE[key] = hit
    return hit


def check(code, what):
    if code != 0:
        msg = '%s: %s (code %d)' % (what, _native.error_string(code), code)
        if code in (-3, -4, -5, -6, -7, -8, -2):
            raise ValueError(msg)
        raise RuntimeError(msg)


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream_ptr(device=None):
    """The caller's current HIP stream on `device` (default: the current device) as a void*.  The raw-stream query is a
    plain C call; building a torch.cuda.Stream object for it costs ten times as much, and the eager step of the headline
    bench is only a few tens of microseconds of host work away from being host-bound."""
    if _raw_stream is not None:
        idx = torch.cuda.current_device() if device is None or device.index is None else device.index
        return ctypes.c_void_p(_raw_stream(idx))
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class _on_device:
    """`with torch.cuda.device(dev)` only when `dev` is not the current device already (the context manager costs ~10 us)."""
    __slots__ = ('ctx',)

    def __init__(self, dev):
        self.ctx = None if (dev.index is None or dev.index == torch.cuda.current_device()) else torch.cuda.device(dev)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            return self.ctx.__exit__(*exc)
        return False


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _require_device(t, name):
    if not t.is_cuda:
        raise TypeError('GenDR only supports CUDA Tensors (%s is on %s).' % (name, t.device))


def native_forward(faces, textures, params, rgba=None, aggrs_info=None):
    """faces [B,nf,9] / textures [B,nf,T,3], contiguous on one GPU, both float32 or both float64 (the float64
    instantiation, reference ``kernel.cu:1102``).  Returns (rgba [B,4,is,is], aggrs_info [B,2,is,is], workspace)."""
    L = _native.lib()
    B, nf = faces.shape[0], faces.shape[1]
    T = textures.shape[2]
    isz = params.image_size
    check(L.gendr_validate(ctypes.byref(params), B, nf, T), 'gendr.render')
    dev = faces.device
    dt = faces.dtype
    f64 = dt == torch.float64
    if rgba is None:
        rgba = torch.empty((B, 4, isz, isz), dtype=dt, device=dev)
    if aggrs_info is None:
        aggrs_info = torch.empty((B, 2, isz, isz), dtype=dt, device=dev)
    nbytes = (L.gendr_workspace_bytes_f64 if f64 else L.gendr_workspace_bytes)(B, nf, T, ctypes.byref(params))
    records = torch.empty((max(int(nbytes), 256),), dtype=torch.uint8, device=dev)
    ev = PROFILE_EVENTS
    with _on_device(dev):
        if ev is not None and ev[0] is not None:
            ev[0].record()
        check((L.gendr_forward_f64 if f64 else L.gendr_forward)(
            _ptr(faces), _ptr(textures), _ptr(rgba), _ptr(aggrs_info), _ptr(records),
            B, nf, T, ctypes.byref(params), _stream_ptr()), 'gendr_forward')
        if ev is not None and ev[1] is not None:
            ev[1].record()
    return rgba, aggrs_info, records


def gradient_buffers(faces, textures, fill=True):
    """(flat, grad_faces [B,nf,9], grad_textures like textures): one allocation (and at most one fill) for both."""
    B, nf = faces.shape[0], faces.shape[1]
    n_f, n_t = B * nf * 9, textures.numel()
    n_f_pad = (n_f + 63) // 64 * 64                        # keeps grad_textures 256-byte aligned
    n = (n_f_pad + n_t + 3) // 4 * 4                       # whole 16-byte stores for the fused clear
    flat = (torch.zeros if fill else torch.empty)(n, dtype=faces.dtype, device=faces.device)
    return flat, flat[:n_f].view(B, nf, 9), flat[n_f_pad:n_f_pad + n_t].view(textures.shape)


def native_backward(faces, textures, rgba, aggrs_info, records, grad_rgba, params, grad_faces=None, grad_textures=None):
    L = _native.lib()
    B, nf = faces.shape[0], faces.shape[1]
    T = textures.shape[2]
    dev = faces.device
    dt = faces.dtype
    f64 = dt == torch.float64
    if grad_faces is None and grad_textures is None:
        # one zero fill for both gradients (they are small: one launch instead of two)
        _, grad_faces, grad_textures = gradient_buffers(faces, textures)
    if grad_faces is None:
        grad_faces = torch.zeros((B, nf, 9), dtype=dt, device=dev)
    if grad_textures is None:
        grad_textures = torch.zeros(textures.shape, dtype=dt, device=dev)
    ev = PROFILE_EVENTS
    with _on_device(dev):
        if ev is not None:
            ev[2].record()
        check((L.gendr_backward_f64 if f64 else L.gendr_backward)(
            _ptr(faces), _ptr(textures), _ptr(rgba), _ptr(aggrs_info), _ptr(records),
            _ptr(grad_rgba), _ptr(grad_faces), _ptr(grad_textures),
            B, nf, T, ctypes.byref(params), _stream_ptr()), 'gendr_backward')
        if ev is not None:
            ev[3].record()
    return grad_faces, grad_textures


class GenDRFunction(Function):
    """``torch.autograd.Function`` with the reference's 19 positional inputs
    (``gendr/functional/renderer.py:13-39``) and output ``[B, 4, is, is]`` RGBA."""

    @staticmethod
    def forward(
            ctx,
            face_vertices,
            textures,
            image_size=256,
            background_color=[0, 0, 0],
            dist_func='uniform',
            dist_scale=1e-2,
            dist_squared=False,
            dist_shape=None,
            dist_shift=None,
            dist_eps=1e4,
            aggr_alpha_func='probabilistic',
            aggr_alpha_t_conorm_p=None,
            aggr_rgb_func='softmax',
            aggr_rgb_eps=1e-3,
            aggr_rgb_gamma=1e-3,
            near=1,
            far=100,
            double_side=True,
            texture_type='surface',
    ):
        _require_device(face_vertices, 'face_vertices')
        _require_device(textures, 'textures')
        params = make_params(image_size, background_color, dist_func, dist_scale, dist_squared, dist_shape,
                             dist_shift, dist_eps, aggr_alpha_func, aggr_alpha_t_conorm_p, aggr_rgb_func,
                             aggr_rgb_eps, aggr_rgb_gamma, near, far, double_side, texture_type)
        ctx.params = params
        ctx.fv_shape, ctx.fv_dtype = face_vertices.shape, face_vertices.dtype
        ctx.tex_shape, ctx.tex_dtype = textures.shape, textures.dtype

        B, nf = face_vertices.shape[:2]
        # float64 tensors are rendered in double, as the reference's AT_DISPATCH_FLOATING_TYPES does
        # (kernel.cu:1102,1117,1189); everything else in float32
        compute = torch.float64 if face_vertices.dtype == torch.float64 else torch.float32
        ctx.compute_dtype = compute
        # (the usual case -- float32, contiguous, one device -- takes the short way: every tensor op is 1-2 us of host time, and
        # the eager step of the headline bench is host-bound on a slow host)
        if face_vertices.dtype == compute and face_vertices.is_contiguous():
            faces = face_vertices.detach().view(B, nf, 9)
        else:
            faces = face_vertices.detach().reshape(B, nf, 9).to(compute).contiguous()
        if textures.dtype == compute and textures.is_contiguous() and textures.device == faces.device:
            tex = textures.detach()
        else:
            tex = textures.detach().to(device=faces.device, dtype=compute).contiguous()
        if tex.dim() != 4 or tex.shape[0] != B or tex.shape[1] != nf or tex.shape[3] != 3:
            raise ValueError('textures must be [B, nf, T, 3] matching face_vertices [B, nf, 3, 3]; got %s and %s'
                             % (tuple(textures.shape), tuple(face_vertices.shape)))

        # The gradients of the coming backward call are allocated now and zero-filled by the per-face setup kernel on its
        # way (gendr_params.clear_ptr): one launch less per step.  Used once; a second backward through the same graph
        # (retain_graph) allocates and fills its own.
        ctx.grad_buffers = None
        if compute == torch.float32 and B * nf > 0 and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]) \
                and os.environ.get('GENDR_FUSED_CLEAR', '1') != '0':
            flat, gf, gt = gradient_buffers(faces, tex, fill=False)
            params.clear_ptr = flat.data_ptr()
            params.clear_floats = flat.numel()
            ctx.grad_buffers = (flat, gf, gt)
        # aggrs_info stays inside this Function (saved for backward, which never looks at tiles no face reaches)
        params.skip_unlisted_aux = 1 if os.environ.get('GENDR_SKIP_UNLISTED_AUX', '1') != '0' else 0
        # pair hints (ABI 6) are for the backward call: a forward pass nobody differentiates does not pay for them
        if not (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]) and params.pair_hints == 0:
            params.pair_hints = -1
        try:
            soft_colors, aggrs_info, records = native_forward(faces, tex, params)
        finally:                               # a failed call must not leave a dangling clear_ptr on the params object
            params.clear_ptr = None
            params.clear_floats = 0
        ctx.save_for_backward(faces, tex, soft_colors, records, aggrs_info)
        return soft_colors

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_soft_colors):
        faces, tex, soft_colors, records, aggrs_info = ctx.saved_tensors
        grad = grad_soft_colors.to(ctx.compute_dtype).contiguous()
        bufs, ctx.grad_buffers = ctx.grad_buffers, None
        if bufs is not None and bufs[0].device == grad.device:
            grad_faces, grad_textures = native_backward(faces, tex, soft_colors, aggrs_info, records, grad, ctx.params,
                                                        grad_faces=bufs[1], grad_textures=bufs[2])
        else:
            grad_faces, grad_textures = native_backward(faces, tex, soft_colors, aggrs_info, records, grad, ctx.params)
        grad_faces = grad_faces.reshape(ctx.fv_shape).to(ctx.fv_dtype)
        grad_textures = grad_textures.reshape(ctx.tex_shape).to(ctx.tex_dtype)
        return (grad_faces, grad_textures) + (None,) * 17


def render(
    face_vertices,
    textures,
    image_size=256,
    background_color=[0, 0, 0],
    dist_func='uniform',
    dist_scale=1e-2,
    dist_squared=False,
    dist_shape=None,
    dist_shift=None,
    dist_eps=1e4,
    aggr_alpha_func='probabilistic',
    aggr_alpha_t_conorm_p=None,
    aggr_rgb_func='softmax',
    aggr_rgb_eps=1e-3,
    aggr_rgb_gamma=1e-3,
    near=1,
    far=100,
    double_side=True,
    texture_type='surface',
):
    """``gendr.functional.render`` (``functional/renderer.py:239-288``).

    float32 CUDA inputs go through the C++ autograd node (``csrc/gendr_torch.cpp``: the same host logic as ``GenDRFunction``
    without the Python frames -- one C++ call per pass, like the reference's pybind launcher ``generalized_renderer_cuda.cpp:74-192``);
    everything else (float64, ``GENDR_CPP_AUTOGRAD=0``, event-sampled steps of bench.py, an unbuilt extension) through
    ``GenDRFunction``.  Both end in the same two C-ABI calls of libgendr_hip.so."""
    if PROFILE_EVENTS is None and face_vertices.dtype == torch.float32 and textures.dtype == torch.float32 and face_vertices.is_cuda \
            and textures.is_cuda and _CPP_AUTOGRAD:
        ext = _native.torch_ext()
        if ext is not None:
            pb, fused = _params_bytes(image_size, background_color, dist_func, dist_scale, dist_squared, dist_shape, dist_shift, dist_eps,
                                      aggr_alpha_func, aggr_alpha_t_conorm_p, aggr_rgb_func, aggr_rgb_eps, aggr_rgb_gamma, near, far,
                                      double_side, texture_type)
            return ext.render(face_vertices, textures, pb, _native.torch_slot(), fused)
    return GenDRFunction.apply(
        face_vertices, textures, image_size, background_color,
        dist_func, dist_scale, dist_squared, dist_shape, dist_shift, dist_eps,
        aggr_alpha_func, aggr_alpha_t_conorm_p,
        aggr_rgb_func, aggr_rgb_eps, aggr_rgb_gamma,
        near, far, double_side, texture_type,
    )


# name used by BASELINE.json's north_star for the same operator
soft_rasterize = render
