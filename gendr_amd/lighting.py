"""Lighting modules: textures are multiplied by ambient + directional light (reference ``gendr/lighting.py:11-71``)."""
import os

import torch
import torch.nn as nn

from . import functional as Fn
from .mesh import Mesh


class AmbientLighting(nn.Module):
    def __init__(self, light_intensity=0.5, light_color=(1, 1, 1)):
        super().__init__()
        self.light_intensity, self.light_color = light_intensity, light_color

    def forward(self, light):
        return Fn.ambient_lighting(light, self.light_intensity, self.light_color)


class DirectionalLighting(nn.Module):
    def __init__(self, light_intensity=0.5, light_color=(1, 1, 1), light_direction=(0, 1, 0)):
        super().__init__()
        self.light_intensity, self.light_color, self.light_direction = light_intensity, light_color, light_direction

    def forward(self, light, normals):
        return Fn.directional_lighting(light, normals, self.light_intensity, self.light_color, self.light_direction)


class Lighting(nn.Module):
    def __init__(self, intensity_ambient=0.5, color_ambient=[1, 1, 1], intensity_directionals=0.5,
                 color_directionals=[1, 1, 1], directions=[0, 1, 0]):
        super().__init__()
        self.ambient = AmbientLighting(intensity_ambient, color_ambient)
        self.directionals = nn.ModuleList([DirectionalLighting(intensity_directionals, color_directionals, directions)])

    def _fused_ok(self, mesh):
        """One HIP kernel instead of the tensor chain: CUDA float32 surface textures, shared colours / directions."""
        if os.environ.get('GENDR_FUSED_LIGHTING', '1') == '0' or mesh.texture_type != 'surface':
            return False
        if not (mesh.vertices.is_cuda and mesh.textures.is_cuda and mesh.textures.dtype == torch.float32):
            return False
        if len(self.directionals) > 4:
            return False

        def plain(v):
            return isinstance(v, (tuple, list)) and len(v) == 3 and all(isinstance(x, (int, float)) for x in v)
        mods = [self.ambient] + list(self.directionals)
        return all(plain(m.light_color) and isinstance(m.light_intensity, (int, float)) for m in mods) and \
            all(plain(d.light_direction) for d in self.directionals)

    def forward(self, mesh):
        if self._fused_ok(mesh):
            lit = Fn.light_faces(mesh.vertices, mesh.faces, mesh.textures, self.ambient.light_intensity, self.ambient.light_color,
                                 [(d.light_intensity, d.light_color, d.light_direction) for d in self.directionals])
            return Mesh(mesh.vertices, mesh.faces, lit, mesh.texture_res, mesh.texture_type)
        if mesh.texture_type == 'surface':
            shape, normals, expand = mesh.faces, 'surface_normals', True
        elif mesh.texture_type == 'vertex':
            shape, normals, expand = mesh.vertices, 'vertex_normals', False
        else:
            raise ValueError('texture type not applicable')
        light = self.ambient(torch.zeros(shape.shape, dtype=torch.float32, device=mesh.device))
        for directional in self.directionals:
            light = directional(light, getattr(mesh, normals))
        textures = mesh.textures * (light[:, :, None, :] if expand else light)
        return Mesh(mesh.vertices, mesh.faces, textures, mesh.texture_res, mesh.texture_type)
