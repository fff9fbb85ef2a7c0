"""Fused lighting of surface textures (the lighting half of SURVEY.md row f-1), HIP kernels behind autograd.

``Lighting.forward`` of the reference (``gendr/lighting.py:48-71``) is about fifteen tensor kernels forward (zeros,
ambient add, surface normals = gather + 2 subtractions + cross + normalize, dot, relu, 3 multiplies, add, multiply
into the textures) and as many backward; here it is one kernel each way (``csrc/gendr_light.h``), same float
expressions in the same order.  Only shared (1-D) colours / directions and at most four directional lights go
through the kernel; anything else stays on the PyTorch composition in ``gendr_amd/lighting.py``.
"""
import torch

from .. import _native
from .renderer import check as _check


def light_params(ambient_intensity, ambient_color, directionals):
    """directionals: iterable of (intensity, colour[3], direction[3])."""
    lp = _native.GendrLightParams()
    lp.ambient_intensity = float(ambient_intensity)
    for k in range(3):
        lp.ambient_color[k] = float(ambient_color[k])
    directionals = list(directionals)
    if len(directionals) > _native.MAX_DIRECTIONAL:
        raise ValueError('at most %d directional lights go through the fused kernel' % _native.MAX_DIRECTIONAL)
    lp.n_directional = len(directionals)
    for i, (inten, col, direc) in enumerate(directionals):
        lp.intensity[i] = float(inten)
        for k in range(3):
            lp.color[i][k] = float(col[k])
            lp.direction[i][k] = float(direc[k])
    return lp


class LightFacesFunction(torch.autograd.Function):
    """(vertices [B,nv,3], faces [B|1,nf,3] int32, textures [B,nf,T,3], light params) -> lit textures [B,nf,T,3]."""

    @staticmethod
    def forward(ctx, vertices, faces, textures, lp):
        lib = _native.lib()
        if not (vertices.is_cuda and textures.is_cuda):
            raise RuntimeError('LightFacesFunction needs CUDA/HIP tensors (no CPU path; use gendr_amd.lighting on CPU)')
        vertices = vertices.to(torch.float32).contiguous()
        textures = textures.to(torch.float32).contiguous()
        faces = faces.to(torch.int32).contiguous()
        B, nv = vertices.shape[0], vertices.shape[1]
        nf, T = textures.shape[1], textures.shape[2]
        if faces.shape[0] not in (1, B) or faces.shape[1] != nf or textures.shape[0] != B:
            raise ValueError('LightFacesFunction: faces [B|1,nf,3], textures [B,nf,T,3], vertices [B,nv,3]')
        batched = int(faces.shape[0] == B and B > 1)
        out = torch.empty_like(textures)
        with torch.cuda.device(vertices.device):
            _check(lib.gendr_light_faces(vertices.data_ptr(), faces.data_ptr(), textures.data_ptr(), out.data_ptr(),
                                         B, nv, nf, T, batched, lp, torch.cuda.current_stream(vertices.device).cuda_stream),
                   'gendr_light_faces')
        ctx.save_for_backward(vertices, faces, textures)
        ctx.cfg = (B, nv, nf, T, batched, lp)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _native.lib()
        vertices, faces, textures = ctx.saved_tensors
        B, nv, nf, T, batched, lp = ctx.cfg
        grad_out = grad_out.to(torch.float32).contiguous()
        need_v, need_t = ctx.needs_input_grad[0], ctx.needs_input_grad[2]
        g_v = torch.zeros_like(vertices) if need_v else None
        g_t = torch.empty_like(textures) if need_t else None
        with torch.cuda.device(vertices.device):
            _check(lib.gendr_light_faces_backward(
                vertices.data_ptr(), faces.data_ptr(), textures.data_ptr(), grad_out.data_ptr(),
                g_t.data_ptr() if need_t else None, g_v.data_ptr() if need_v else None,
                B, nv, nf, T, batched, lp, torch.cuda.current_stream(vertices.device).cuda_stream), 'gendr_light_faces_backward')
        return g_v, None, g_t, None


def light_faces(vertices, faces, textures, ambient_intensity=0.5, ambient_color=(1, 1, 1),
                directionals=((0.5, (1, 1, 1), (0, 1, 0)),)):
    return LightFacesFunction.apply(vertices, faces, textures, light_params(ambient_intensity, ambient_color, directionals))
