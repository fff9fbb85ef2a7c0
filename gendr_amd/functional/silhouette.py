"""Alpha-only rendering with the silhouette loss fused into the kernel epilogue.

The reference's experiment scripts render RGBA and keep channel 3 (``experiments/opt_shape.py:257,296-303``,
``train_reconstruction.py:41-46``), then reduce it against target silhouettes with ``iou_loss``
(``opt_shape.py:20-24``, ``train_reconstruction.py:30-36``).  ``render_silhouette`` returns that channel alone --
bit-identical to ``render(...)[:, 3]`` -- from kernels that never touch colour, depth or the softmax state and write
one plane instead of six; ``silhouette_iou`` additionally accumulates the two sums the IoU needs inside the forward
kernel and forms the per-pixel gradient from two scalars per view inside the backward kernel, so that no gradient
image exists at all.  Same option names and defaults as ``render``; the texture options do not apply.
"""
import ctypes

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _native
from .renderer import make_params, check, _ptr, _stream_ptr, _require_device


def _params(image_size, dist_func, dist_scale, dist_squared, dist_shape, dist_shift, dist_eps,
            aggr_alpha_func, aggr_alpha_t_conorm_p, near, far):
    return make_params(image_size, [0, 0, 0], dist_func, dist_scale, dist_squared, dist_shape, dist_shift, dist_eps,
                       aggr_alpha_func, aggr_alpha_t_conorm_p, 'softmax', 1e-3, 1e-3, near, far, True, 'surface')


def _forward(face_vertices, params, target=None, want_grad=False):
    """Returns (faces, alpha, workspace, target, sums, grad_faces): grad_faces is the buffer of the coming backward call,
    zero-filled by the setup kernel of this one (gendr_params.clear_ptr), or None."""
    L = _native.lib()
    _require_device(face_vertices, 'face_vertices')
    B, nf = face_vertices.shape[:2]
    faces = face_vertices.detach().reshape(B, nf, 9).to(torch.float32).contiguous()
    dev = faces.device
    grad_faces = None
    if not want_grad and params.pair_hints == 0:
        params.pair_hints = -1               # pair hints (ABI 6) serve the backward call only
    if want_grad and B * nf > 0:
        n = (B * nf * 9 + 3) // 4 * 4
        flat = torch.empty(n, dtype=torch.float32, device=dev)
        params.clear_ptr, params.clear_floats = flat.data_ptr(), n
        grad_faces = flat[:B * nf * 9].view(B, nf, 9)
    try:
        isz = params.image_size
        alpha = torch.empty((B, isz, isz), dtype=torch.float32, device=dev)
        ws = torch.empty((max(int(L.gendr_silhouette_workspace_bytes(B, nf, ctypes.byref(params))), 256),), dtype=torch.uint8, device=dev)
        sums = None
        if target is not None:
            if tuple(target.shape) != (B, isz, isz):
                raise ValueError('target must be [B, image_size, image_size] = %s, got %s' % ((B, isz, isz), tuple(target.shape)))
            target = target.detach().to(device=dev, dtype=torch.float32).contiguous()
            sums = torch.empty((B, 2), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            check(L.gendr_silhouette_forward(_ptr(faces), _ptr(alpha), _ptr(ws), _ptr(target) if target is not None else None,
                                             _ptr(sums) if sums is not None else None, B, nf, ctypes.byref(params), _stream_ptr()),
                  'gendr_silhouette_forward')
    finally:
        # whatever happened above (a shape error, a failed launch): the caller's params object must not keep a pointer to a
        # buffer that is about to be freed
        params.clear_ptr, params.clear_floats = None, 0
    return faces, alpha, ws, target, sums, grad_faces


def _backward(faces, alpha, ws, params, grad_alpha=None, target=None, grad_iou=None, grad_faces=None):
    L = _native.lib()
    B, nf = faces.shape[:2]
    if grad_faces is None:
        grad_faces = torch.zeros((B, nf, 9), dtype=torch.float32, device=faces.device)
    with torch.cuda.device(faces.device):
        check(L.gendr_silhouette_backward(_ptr(alpha), _ptr(ws), _ptr(grad_alpha) if grad_alpha is not None else None,
                                          _ptr(target) if target is not None else None,
                                          _ptr(grad_iou) if grad_iou is not None else None,
                                          _ptr(grad_faces), B, nf, ctypes.byref(params), _stream_ptr()),
              'gendr_silhouette_backward')
    return grad_faces


class SilhouetteFunction(Function):
    """face_vertices [B,nf,3,3] -> alpha [B,is,is]  (== GenDRFunction(...)[:, 3])."""

    @staticmethod
    def forward(ctx, face_vertices, params):
        faces, alpha, ws, _, _, ctx.grad_faces = _forward(face_vertices, params, want_grad=ctx.needs_input_grad[0])
        ctx.params, ctx.shape, ctx.dtype = params, face_vertices.shape, face_vertices.dtype
        ctx.save_for_backward(faces, alpha, ws)
        return alpha

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_alpha):
        faces, alpha, ws = ctx.saved_tensors
        buf, ctx.grad_faces = ctx.grad_faces, None              # cleared by the forward call; good for one use
        g = _backward(faces, alpha, ws, ctx.params, grad_alpha=grad_alpha.to(torch.float32).contiguous(), grad_faces=buf)
        return g.reshape(ctx.shape).to(ctx.dtype), None


class SilhouetteIoUFunction(Function):
    """(face_vertices, target [B,is,is]) -> sums [B,2] = (sum(alpha t), sum(alpha (1 - t))) per view, alpha and the
    target as the kernel saw it (on the render device, float32) -- neither differentiable through this Function."""

    @staticmethod
    def forward(ctx, face_vertices, target, params):
        faces, alpha, ws, tgt, sums, ctx.grad_faces = _forward(face_vertices, params, target, want_grad=ctx.needs_input_grad[0])
        ctx.params, ctx.shape, ctx.dtype = params, face_vertices.shape, face_vertices.dtype
        ctx.save_for_backward(faces, alpha, ws, tgt)
        ctx.mark_non_differentiable(alpha, tgt)     # one call: a second call would replace the first one's set
        return sums, alpha, tgt

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_sums, _grad_alpha_unused, _grad_target_unused):
        faces, alpha, ws, tgt = ctx.saved_tensors
        buf, ctx.grad_faces = ctx.grad_faces, None
        g = _backward(faces, alpha, ws, ctx.params, target=tgt, grad_iou=grad_sums.to(torch.float32).contiguous(), grad_faces=buf)
        return g.reshape(ctx.shape).to(ctx.dtype), None, None


def render_silhouette(face_vertices, image_size=256, dist_func='uniform', dist_scale=1e-2, dist_squared=False,
                      dist_shape=None, dist_shift=None, dist_eps=1e4, aggr_alpha_func='probabilistic',
                      aggr_alpha_t_conorm_p=None, near=1, far=100):
    """Alpha channel of ``render`` alone, [B, image_size, image_size]."""
    p = _params(image_size, dist_func, dist_scale, dist_squared, dist_shape, dist_shift, dist_eps,
                aggr_alpha_func, aggr_alpha_t_conorm_p, near, far)
    return SilhouetteFunction.apply(face_vertices, p)


def silhouette_iou(face_vertices, target, image_size=256, dist_func='uniform', dist_scale=1e-2, dist_squared=False,
                   dist_shape=None, dist_shift=None, dist_eps=1e4, aggr_alpha_func='probabilistic',
                   aggr_alpha_t_conorm_p=None, near=1, far=100, return_alpha=False):
    """Per-view ``(intersect, union)`` of the rendered silhouettes with ``target`` [B,is,is], as ``iou_loss`` forms
    them (``opt_shape.py:21-23``): intersect = sum(a t), union = sum(a + t - a t) (without the 1e-6)."""
    p = _params(image_size, dist_func, dist_scale, dist_squared, dist_shape, dist_shift, dist_eps,
                aggr_alpha_func, aggr_alpha_t_conorm_p, near, far)
    sums, alpha, tgt = SilhouetteIoUFunction.apply(face_vertices, target, p)     # tgt: the target on the render device, float32
    intersect = sums[:, 0]
    union = tgt.sum((1, 2)) + sums[:, 1]
    return (intersect, union, alpha) if return_alpha else (intersect, union)


def silhouette_iou_loss(face_vertices, target, eps=1e-6, **options):
    """``iou_loss(render(...)[:, 3], target)`` of the reference's experiments (``opt_shape.py:20-24``;
    ``train_reconstruction.py:30-36`` is the same number), without materialising RGB planes or a gradient image."""
    intersect, union = silhouette_iou(face_vertices, target, **options)
    return (1. - intersect / (union + eps)).mean()
