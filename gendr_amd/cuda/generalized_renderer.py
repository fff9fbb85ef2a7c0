"""pybind-shaped module ``gendr.cuda.generalized_renderer``.

The reference exposes six functions here (``generalized_renderer_cuda.cpp:230-237``);
``animations/distributions_to_csv.py:19-57`` and ``animations/t_conorms.py:33-61`` call
the four scalar ones directly.  Same names, argument order and meaning; backed by
``libgendr_hip.so``.
"""
import ctypes

import torch

from .. import _native
from ..functional.renderer import make_params, native_forward, native_backward, check, _ptr, _stream_ptr


def _params(image_size, dist_func, dist_scale, dist_squared, dist_shape, dist_shift, dist_eps,
            aggr_alpha_func, aggr_alpha_t_conorm_p, aggr_rgb_func, aggr_rgb_eps, aggr_rgb_gamma,
            near, far, double_side, texture_type, from_buffer):
    return make_params(image_size, [0, 0, 0], int(dist_func), dist_scale, dist_squared, dist_shape, dist_shift,
                       dist_eps, int(aggr_alpha_func), aggr_alpha_t_conorm_p, int(aggr_rgb_func), aggr_rgb_eps,
                       aggr_rgb_gamma, near, far, double_side, int(texture_type), background_from_buffer=from_buffer)


def _check_input(t, name):
    if not t.is_cuda:
        raise RuntimeError('%s must be a CUDA tensor' % name)       # CHECK_CUDA, generalized_renderer_cuda.cpp:69
    if not t.is_contiguous():
        raise RuntimeError('%s must be contiguous' % name)          # CHECK_CONTIGUOUS, :70


def forward_render(faces, textures, faces_info, aggrs_info, soft_colors, image_size,
                   dist_func, dist_scale, dist_squared, dist_shape, dist_shift, dist_eps,
                   aggr_alpha_func, aggr_alpha_t_conorm_p, aggr_rgb_func, aggr_rgb_eps, aggr_rgb_gamma,
                   near, far, double_side, texture_type):
    """``soft_colors`` arrives pre-filled with the background (functional/renderer.py:144-151) and is
    updated in place; ``faces_info`` is filled in the reference's [B,nf,27] layout."""
    for t, n in ((faces, 'faces'), (textures, 'textures'), (faces_info, 'faces_info'),
                 (aggrs_info, 'aggrs_info'), (soft_colors, 'soft_colors')):
        _check_input(t, n)
    p = _params(image_size, dist_func, dist_scale, dist_squared, dist_shape, dist_shift, dist_eps,
                aggr_alpha_func, aggr_alpha_t_conorm_p, aggr_rgb_func, aggr_rgb_eps, aggr_rgb_gamma,
                near, far, double_side, texture_type, True)
    B, nf = faces.shape[:2]
    f9 = faces.reshape(B, nf, 9)
    with torch.cuda.device(faces.device):
        check(_native.lib().gendr_face_info(_ptr(f9), _ptr(faces_info), B, nf, _stream_ptr()), 'gendr_face_info')
    native_forward(f9, textures, p, rgba=soft_colors, aggrs_info=aggrs_info)
    return [faces_info, aggrs_info, soft_colors]


def backward_render(faces, textures, soft_colors, faces_info, aggrs_info, grad_faces, grad_textures,
                    grad_soft_colors, image_size,
                    dist_func, dist_scale, dist_squared, dist_shape, dist_shift, dist_eps,
                    aggr_alpha_func, aggr_alpha_t_conorm_p, aggr_rgb_func, aggr_rgb_eps, aggr_rgb_gamma,
                    near, far, double_side, texture_type):
    """Accumulates into ``grad_faces`` / ``grad_textures`` (zero-filled by the caller, functional/renderer.py:191-196)."""
    for t, n in ((faces, 'faces'), (textures, 'textures'), (soft_colors, 'soft_colors'), (faces_info, 'faces_info'),
                 (aggrs_info, 'aggrs_info'), (grad_faces, 'grad_faces'), (grad_textures, 'grad_textures'),
                 (grad_soft_colors, 'grad_soft_colors')):
        _check_input(t, n)
    p = _params(image_size, dist_func, dist_scale, dist_squared, dist_shape, dist_shift, dist_eps,
                aggr_alpha_func, aggr_alpha_t_conorm_p, aggr_rgb_func, aggr_rgb_eps, aggr_rgb_gamma,
                near, far, double_side, texture_type, False)
    L = _native.lib()
    B, nf = faces.shape[:2]
    T = textures.shape[2]
    f9 = faces.reshape(B, nf, 9)
    records = torch.empty((max(int(L.gendr_workspace_bytes(B, nf, T, ctypes.byref(p))), 256),),
                          dtype=torch.uint8, device=faces.device)
    with torch.cuda.device(faces.device):
        check(L.gendr_face_setup(_ptr(f9), _ptr(textures), _ptr(records), B, nf, T, ctypes.byref(p), _stream_ptr()),
              'gendr_face_setup')
    native_backward(f9, textures, soft_colors, aggrs_info, records, grad_soft_colors, p,
                    grad_faces=grad_faces.view(B, nf, 9), grad_textures=grad_textures)
    return [grad_faces, grad_textures]


def sigmoid_forward(function_id, sign, x, scale, dist_shape, dist_shift):
    return _native.lib().gendr_sigmoid_forward(int(function_id), sign, x, scale, dist_shape, dist_shift)


def sigmoid_backward(function_id, sign, x, scale, dist_shape, dist_shift):
    return _native.lib().gendr_sigmoid_backward(int(function_id), sign, x, scale, dist_shape, dist_shift)


def t_conorm_forward(t_conorm_id, a_existing, b_new, face_id, t_conorm_p):
    return _native.lib().gendr_t_conorm_forward(int(t_conorm_id), a_existing, b_new, int(face_id), t_conorm_p)


def t_conorm_backward(t_conorm_id, a_all, b_current, number_of_faces, t_conorm_p):
    return _native.lib().gendr_t_conorm_backward(int(t_conorm_id), a_all, b_current, int(number_of_faces), t_conorm_p)
