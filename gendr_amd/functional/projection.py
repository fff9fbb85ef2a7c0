"""Fused camera transform + face gather, HIP kernels behind autograd.

``look_at`` / ``look`` (reference ``gendr/functional/look_at.py:59-67``, ``look.py``), ``perspective`` /
``orthogonal`` (``gendr/transform.py:14-47``) and the ``face_vertices`` gather (``functional/face_vertices.py:24-27``)
are one kernel each way here (``csrc/gendr_project.h``) instead of ~10 elementwise kernels and five passes over
the vertex data.  The 3x3 camera rotation is built from ``[B,3]`` tensors in plain PyTorch
(``geometry._camera_rotation``) so autograd still reaches ``eye`` / ``at`` / ``up``.

Results equal the unfused composition in ``geometry.py`` (itself pinned to the reference's own modules by
``tests/golden/glue``) to fp32 rounding: the matmul's summation order inside rocBLAS is not specified, the kernel
sums j = 0,1,2 left to right.
"""
import math

import numpy as np
import torch

from .. import _native
from .geometry import _as_vec3, _camera_rotation
from .renderer import check as _check


def _stream(device):
    return torch.cuda.current_stream(device).cuda_stream


_CHECKED = {}          # id(tensor) -> (weakref to the tensor, _version, nv it was checked against)


def _check_indices(faces, nv):
    """Range check of the face indices, for error reporting only: the kernels never read outside the vertex
    tensor (an out-of-range index yields NaN vertices and no gradient).  One fused device-side reduction and a
    single host read -- which blocks the host on the stream, so an index tensor is checked ONCE: the result is
    remembered per tensor OBJECT (a weak reference, so that a new tensor at a recycled address or id is never mistaken for
    it) and version counter (an in-place edit checks again).  A training loop that projects the same `faces` every step
    pays the synchronisation on the first step only.  Skipped while a HIP graph is being captured (a host read is illegal
    there) and with GENDR_CHECK_INDICES=0."""
    import os
    import weakref
    if faces.numel() == 0 or os.environ.get('GENDR_CHECK_INDICES', '1') == '0':
        return
    if faces.is_cuda and torch.cuda.is_current_stream_capturing():
        return
    key = id(faces)
    try:
        version = faces._version          # inference tensors have no version counter (RuntimeError): always checked
    except RuntimeError:
        version = None
    hit = _CHECKED.get(key) if version is not None else None
    if hit is not None and hit[0]() is faces and hit[1] == version and hit[2] <= nv:
        return
    lo, hi = torch.aminmax(faces)
    if bool((lo < 0) | (hi >= nv)):
        raise IndexError('face index out of range')
    if len(_CHECKED) > 64:
        for k in [k for k, v in _CHECKED.items() if v[0]() is None]:
            del _CHECKED[k]
        if len(_CHECKED) > 64:
            _CHECKED.clear()
    if version is not None:
        _CHECKED[key] = (weakref.ref(faces), version, nv)


class ProjectFacesFunction(torch.autograd.Function):
    """(vertices [B,nv,3], faces [B|1,nf,3] int32, camera [B,12]) -> face_vertices [B,nf,3,3]."""

    @staticmethod
    def forward(ctx, vertices, faces, camera, perspective, width_or_scale):
        lib = _native.lib()                                    # raises NativeLibraryError when the HIP library is missing
        if not vertices.is_cuda:
            raise RuntimeError('ProjectFacesFunction needs CUDA/HIP tensors (no CPU path; use geometry.look_at etc.)')
        if vertices.dtype != torch.float32:
            raise TypeError('ProjectFacesFunction: float32 vertices only')
        vertices = vertices.contiguous()
        camera = camera.to(torch.float32).contiguous()
        faces_given = faces                                    # the caller's tensor object: what the index check remembers
        faces = faces.to(torch.int32).contiguous()
        B, nv = vertices.shape[0], vertices.shape[1]
        nf = faces.shape[1]
        if faces.shape[0] not in (1, B) or camera.shape != (B, 12):
            raise ValueError('ProjectFacesFunction: faces must be [B|1,nf,3] and camera [B,12]')
        _check_indices(faces_given, nv)
        out = torch.empty(B, nf, 3, 3, dtype=torch.float32, device=vertices.device)
        batched = int(faces.shape[0] == B and B > 1)
        with torch.cuda.device(vertices.device):
            _check(lib.gendr_project_faces(vertices.data_ptr(), faces.data_ptr(), camera.data_ptr(), out.data_ptr(),
                                                  B, nv, nf, batched, int(bool(perspective)), float(width_or_scale),
                                                  _stream(vertices.device)), 'gendr_project_faces')
        ctx.save_for_backward(vertices, faces, camera)
        ctx.cfg = (B, nv, nf, batched, int(bool(perspective)), float(width_or_scale))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _native.lib()
        vertices, faces, camera = ctx.saved_tensors
        B, nv, nf, batched, persp, ws = ctx.cfg
        grad_out = grad_out.to(torch.float32).contiguous()
        grad_vertices = torch.zeros_like(vertices)
        need_cam = ctx.needs_input_grad[2]
        grad_camera = torch.zeros_like(camera) if need_cam else None
        with torch.cuda.device(vertices.device):
            _check(lib.gendr_project_faces_backward(
                vertices.data_ptr(), faces.data_ptr(), camera.data_ptr(), grad_out.data_ptr(), grad_vertices.data_ptr(),
                grad_camera.data_ptr() if need_cam else None, B, nv, nf, batched, persp, ws, _stream(vertices.device)),
                'gendr_project_faces_backward')
        return grad_vertices, None, grad_camera, None, None


class CameraFacesFunction(torch.autograd.Function):
    """(vertices [B,nv,3], faces [B|1,nf,3], eye, target, up [B,3]) -> face_vertices [B,nf,3,3]: camera rotation +
    translate + rotate + project + gather in two launches forward and two backward."""

    @staticmethod
    def forward(ctx, vertices, faces, eye, target, up, target_is_direction, perspective, width_or_scale):
        lib = _native.lib()
        if not vertices.is_cuda:
            raise RuntimeError('CameraFacesFunction needs CUDA/HIP tensors (no CPU path; use geometry.look_at etc.)')
        if vertices.dtype != torch.float32:
            raise TypeError('CameraFacesFunction: float32 vertices only')
        vertices = vertices.contiguous()
        faces_given = faces                                    # the caller's tensor object: what the index check remembers
        faces = faces.to(torch.int32).contiguous()
        eye, target, up = (t.to(torch.float32).contiguous() for t in (eye, target, up))
        B, nv = vertices.shape[0], vertices.shape[1]
        nf = faces.shape[1]
        if faces.shape[0] not in (1, B) or any(t.shape != (B, 3) for t in (eye, target, up)):
            raise ValueError('CameraFacesFunction: faces must be [B|1,nf,3]; eye, target, up [B,3]')
        _check_indices(faces_given, nv)
        dev = vertices.device
        camera = torch.empty(B, 12, dtype=torch.float32, device=dev)
        out = torch.empty(B, nf, 3, 3, dtype=torch.float32, device=dev)
        batched = int(faces.shape[0] == B and B > 1)
        is_dir, persp, ws = int(bool(target_is_direction)), int(bool(perspective)), float(width_or_scale)
        with torch.cuda.device(dev):
            st = _stream(dev)
            _check(lib.gendr_camera_rotation(eye.data_ptr(), target.data_ptr(), up.data_ptr(), camera.data_ptr(), B, is_dir, st),
                   'gendr_camera_rotation')
            _check(lib.gendr_project_faces(vertices.data_ptr(), faces.data_ptr(), camera.data_ptr(), out.data_ptr(),
                                           B, nv, nf, batched, persp, ws, st), 'gendr_project_faces')
        ctx.save_for_backward(vertices, faces, eye, target, up, camera)
        ctx.cfg = (B, nv, nf, batched, is_dir, persp, ws)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _native.lib()
        vertices, faces, eye, target, up, camera = ctx.saved_tensors
        B, nv, nf, batched, is_dir, persp, ws = ctx.cfg
        grad_out = grad_out.to(torch.float32).contiguous()
        need_v = ctx.needs_input_grad[0]
        need_e, need_t, need_u = ctx.needs_input_grad[2:5]
        need_cam = need_e or need_t or need_u
        grad_vertices = torch.zeros_like(vertices)
        grad_camera = torch.zeros_like(camera) if need_cam else None
        g_e = torch.empty_like(eye) if need_e else None
        g_t = torch.empty_like(target) if need_t else None
        g_u = torch.empty_like(up) if need_u else None
        ptr = lambda t: t.data_ptr() if t is not None else None
        with torch.cuda.device(vertices.device):
            st = _stream(vertices.device)
            _check(lib.gendr_project_faces_backward(
                vertices.data_ptr(), faces.data_ptr(), camera.data_ptr(), grad_out.data_ptr(), grad_vertices.data_ptr(),
                ptr(grad_camera), B, nv, nf, batched, persp, ws, st), 'gendr_project_faces_backward')
            if need_cam:
                _check(lib.gendr_camera_rotation_backward(eye.data_ptr(), target.data_ptr(), up.data_ptr(), grad_camera.data_ptr(),
                                                          ptr(g_e), ptr(g_t), ptr(g_u), B, is_dir, st),
                       'gendr_camera_rotation_backward')
        return (grad_vertices if need_v else None), None, g_e, g_t, g_u, None, None, None


def _width_or_scale(perspective, viewing_angle, viewing_scale):
    # transform.py:21-23: width = tan(angle in radians), computed there as a float32 tensor op
    if perspective:
        return float(np.tan(np.float32(viewing_angle / 180. * math.pi), dtype=np.float32))
    return float(viewing_scale)


def project_faces(vertices, faces, rotation, eye, perspective=True, viewing_angle=30., viewing_scale=1.):
    """``rotation`` [B,3,3] (rows = camera axes), ``eye`` [B,3] -> projected ``face_vertices`` [B,nf,3,3]."""
    camera = torch.cat([rotation.reshape(-1, 9), eye.to(rotation.dtype)], dim=1).to(torch.float32)
    return ProjectFacesFunction.apply(vertices, faces, camera, perspective,
                                      _width_or_scale(perspective, viewing_angle, viewing_scale))


def look_at_faces(vertices, faces, eye, at=[0, 0, 0], up=[0, 1, 0], perspective=True, viewing_angle=30., viewing_scale=1.):
    """== face_vertices(perspective(look_at(vertices, eye, at, up), viewing_angle), faces), fused."""
    if vertices.ndimension() != 3:
        raise ValueError('vertices Tensor should have 3 dimensions')
    B, dev = vertices.shape[0], vertices.device
    eye, at, up = _as_vec3(eye, dev, B), _as_vec3(at, dev, B), _as_vec3(up, dev, B)
    return CameraFacesFunction.apply(vertices, faces, eye, at, up, False, perspective,
                                     _width_or_scale(perspective, viewing_angle, viewing_scale))


def look_faces(vertices, faces, eye, direction=[0, 1, 0], up=None, perspective=True, viewing_angle=30., viewing_scale=1.):
    """== face_vertices(perspective(look(vertices, eye, direction, up), viewing_angle), faces), fused."""
    if vertices.ndimension() != 3:
        raise ValueError('vertices Tensor should have 3 dimensions')
    B, dev = vertices.shape[0], vertices.device
    if up is None:
        up = [0., 1., 0.]
    eye, direction, up = _as_vec3(eye, dev, B), _as_vec3(direction, dev, B), _as_vec3(up, dev, B)
    return CameraFacesFunction.apply(vertices, faces, eye, direction, up, True, perspective,
                                     _width_or_scale(perspective, viewing_angle, viewing_scale))
