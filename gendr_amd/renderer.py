"""``GenDR`` nn.Module: option holder in front of ``gendr_amd.functional.render``.

Same constructor arguments, defaults and ``ValueError``s as the reference's
``gendr/renderer.py:12-65``; options are plain attributes read at call time, so
scripts that mutate them between renders (``experiments/opt_shape.py:249,289``,
``animations/panda_tcn_p.py:104-109``) keep working.  Anti-aliasing renders at
twice the size and average-pools 2x2 (``gendr/renderer.py:68,92-93``).
"""
import torch.nn as nn
import torch.nn.functional as F

from .functional.renderer import render

_OPTION_DEFAULTS = (
    ('dist_func', 'uniform'), ('dist_scale', 1e-2), ('dist_squared', False), ('dist_shape', None),
    ('dist_shift', None), ('dist_eps', 1e4),
    ('aggr_alpha_func', 'probabilistic'), ('aggr_alpha_t_conorm_p', None),
    ('aggr_rgb_func', 'softmax'), ('aggr_rgb_eps', 1e-3), ('aggr_rgb_gamma', 1e-3),
    ('near', 1), ('far', 100), ('double_side', False), ('texture_type', 'surface'),
)


class GenDR(nn.Module):
    def __init__(self, image_size=256, background_color=[0, 0, 0], anti_aliasing=False,
                 dist_func='uniform', dist_scale=1e-2, dist_squared=False, dist_shape=None, dist_shift=None,
                 dist_eps=1e4,
                 aggr_alpha_func='probabilistic', aggr_alpha_t_conorm_p=None,
                 aggr_rgb_func='softmax', aggr_rgb_eps=1e-3, aggr_rgb_gamma=1e-3,
                 near=1, far=100, double_side=False, texture_type='surface'):
        super().__init__()
        if aggr_rgb_func not in ['hard', 'softmax']:
            raise ValueError('Aggregate function (RGB) currently only supports hard and softmax.')
        if texture_type not in ['surface', 'vertex']:
            raise ValueError('Texture type only support surface and vertex.')
        self.image_size = image_size
        self.background_color = background_color
        self.anti_aliasing = anti_aliasing
        given = locals()
        for name, _default in _OPTION_DEFAULTS:
            setattr(self, name, given[name])

    def _options(self):
        return {name: getattr(self, name) for name, _default in _OPTION_DEFAULTS}

    def forward_tensors(self, face_vertices, face_textures):
        scale = 2 if self.anti_aliasing else 1
        images = render(face_vertices, face_textures, image_size=self.image_size * scale,
                        background_color=self.background_color, **self._options())
        if self.anti_aliasing:
            images = F.avg_pool2d(images, kernel_size=2, stride=2)
        return images

    def forward(self, mesh):
        return self.forward_tensors(mesh.face_vertices, mesh.face_textures)

    # ---- alpha-only rendering (SURVEY f-4; not part of the reference's surface) -------------------------------------
    def _silhouette_options(self):
        o = self._options()
        for k in ('aggr_rgb_func', 'aggr_rgb_eps', 'aggr_rgb_gamma', 'double_side', 'texture_type'):
            o.pop(k)
        return o

    def silhouette(self, mesh):
        """``self(mesh)[:, 3]`` from the alpha-only kernels (no RGB / aggrs_info planes); anti-aliasing as in forward."""
        from .functional.silhouette import render_silhouette
        scale = 2 if self.anti_aliasing else 1
        alpha = render_silhouette(mesh.face_vertices, image_size=self.image_size * scale, **self._silhouette_options())
        if self.anti_aliasing:
            alpha = F.avg_pool2d(alpha[:, None], kernel_size=2, stride=2)[:, 0]
        return alpha

    def silhouette_iou_loss(self, mesh, target, eps=1e-6):
        """``iou_loss(self(mesh)[:, 3], target)`` (``experiments/opt_shape.py:20-24``) with the two IoU sums accumulated
        in the forward kernel and the gradient formed in the backward kernel.  Without anti-aliasing only."""
        from .functional.silhouette import silhouette_iou_loss
        if self.anti_aliasing:
            from .functional.silhouette import render_silhouette
            a = self.silhouette(mesh)
            dims = tuple(range(1, a.ndimension()))
            inter = (a * target).sum(dims)
            return (1. - inter / ((a + target - a * target).sum(dims) + eps)).mean()
        return silhouette_iou_loss(mesh.face_vertices, target, eps=eps, image_size=self.image_size, **self._silhouette_options())
