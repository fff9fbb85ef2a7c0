"""``Mesh``: batched triangle-mesh container in front of the renderer (plain PyTorch glue).

Same constructor, class method and properties as the reference's ``gendr/mesh.py:12-126``; numpy inputs are
moved to the GPU when one is present (the reference calls ``.cuda()`` unconditionally)."""
import numpy as np
import torch
import torch.nn.functional as F

from . import functional as Fn


def _device():
    return torch.device('cuda') if torch.cuda.is_available() else torch.device('cpu')


class Mesh(object):
    def __init__(self, vertices, faces, textures=None, texture_res=1, texture_type='surface'):
        if isinstance(vertices, np.ndarray):
            vertices = torch.from_numpy(vertices).float().to(_device())
        if isinstance(faces, np.ndarray):
            faces = torch.from_numpy(faces).int().to(_device())
        if vertices.ndimension() == 2:
            vertices = vertices[None]
        if faces.ndimension() == 2:
            faces = faces[None]
        self._vertices, self._faces = vertices, faces
        self.device = vertices.device
        self.texture_type = texture_type
        self.batch_size, self.num_vertices = vertices.shape[:2]
        self.num_faces = faces.shape[1]

        if textures is None:
            if texture_type == 'surface':
                textures = torch.ones(self.batch_size, self.num_faces, texture_res ** 2, 3, dtype=torch.float32, device=self.device)
                self.texture_res = texture_res
            elif texture_type == 'vertex':
                textures = torch.ones(self.batch_size, self.num_vertices, 3, dtype=torch.float32, device=self.device)
                self.texture_res = 1
        else:
            if isinstance(textures, np.ndarray):
                textures = torch.from_numpy(textures).float().to(self.device)
            if textures.ndimension() == 3 and texture_type == 'surface':
                textures = textures[None]
            if textures.ndimension() == 2 and texture_type == 'vertex':
                textures = textures[None]
            self.texture_res = int(np.sqrt(textures.shape[2]))
        self._textures = textures

    @classmethod
    def from_obj(cls, filename_obj, normalization=False, load_texture=False, texture_res=1, texture_type='surface'):
        out = Fn.load_obj(filename_obj, normalization=normalization, texture_res=texture_res,
                          load_texture=load_texture, **({'texture_type': texture_type} if load_texture else {}))
        vertices, faces = out[0], out[1]
        textures = out[2] if load_texture else None
        return cls(vertices, faces, textures, texture_res, texture_type)

    def save_obj(self, filename_obj, save_texture=False, texture_res_out=16):
        if self.batch_size != 1:
            raise ValueError('Could not save when batch size > 1')
        Fn.save_obj(filename_obj, self.vertices[0], self.faces[0],
                    textures=self.textures[0] if save_texture else None,
                    texture_res=texture_res_out, texture_type=self.texture_type)

    faces = property(lambda self: self._faces)
    vertices = property(lambda self: self._vertices)
    textures = property(lambda self: self._textures)

    @property
    def face_vertices(self):
        return Fn.face_vertices(self._vertices, self._faces)

    @property
    def surface_normals(self):
        fv = self.face_vertices
        return F.normalize(torch.cross(fv[:, :, 2] - fv[:, :, 1], fv[:, :, 0] - fv[:, :, 1], dim=2), p=2, dim=2, eps=1e-6)

    @property
    def vertex_normals(self):
        return Fn.vertex_normals(self._vertices, self._faces)

    @property
    def face_textures(self):
        if self.texture_type == 'surface':
            return self._textures
        if self.texture_type == 'vertex':
            return Fn.face_vertices(self._textures, self._faces)
        raise ValueError('texture type not applicable')

    def voxelize(self, voxel_size=32):
        return Fn.voxelization(self.face_vertices * voxel_size / (voxel_size - 1) + 0.5, voxel_size, False)
