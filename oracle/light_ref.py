"""CPU oracle for the fused lighting (lighting half).  TEST INFRASTRUCTURE ONLY -- nothing under
gendr_amd/ may import this.

numpy restatement (fp32, same operation order) of
  ambient_lighting      gendr/functional/lighting.py:11-25   light += intensity * colour
  directional_lighting  gendr/functional/lighting.py:28-48   light += intensity * (colour * relu(<normal, direction>))
  surface normals       gendr/mesh.py:109-117                normalize(cross(v2 - v1, v0 - v1), eps 1e-6)
  Lighting.forward      gendr/lighting.py:48-71              textures * light[:, :, None, :]
Pinned by tests/golden/glue/glue.npz: `ambient` and `directional` there are outputs of the reference's own
functional/lighting.py (on vertex normals, which are inputs here).
"""
import numpy as np

F = np.float32


def ambient(light, intensity, color):
    return (light + F(intensity) * np.asarray(color, F)[None, None, :]).astype(F)


def directional(light, normals, intensity, color, direction):
    d = np.asarray(direction, F)
    n = np.asarray(normals, F)
    cosine = np.maximum((n[..., 0] * d[0] + n[..., 1] * d[1]) + n[..., 2] * d[2], F(0))
    return (light + F(intensity) * (np.asarray(color, F)[None, None, :] * cosine[..., None])).astype(F)


def surface_normals(vertices, faces):
    v = np.asarray(vertices, F)
    B = v.shape[0]
    f = np.broadcast_to(faces, (B,) + faces.shape[1:]).astype(np.int64)
    fv = np.stack([v[b][f[b]] for b in range(B)])                       # [B,nf,3,3]
    raw = np.cross(fv[:, :, 2] - fv[:, :, 1], fv[:, :, 0] - fv[:, :, 1]).astype(F)
    norm = np.sqrt((raw[..., 0] * raw[..., 0] + raw[..., 1] * raw[..., 1]) + raw[..., 2] * raw[..., 2])
    return (raw / np.maximum(norm, F(1e-6))[..., None]).astype(F)


def light_faces(vertices, faces, textures, ambient_intensity, ambient_color, directionals):
    n = surface_normals(vertices, faces)
    light = ambient(np.zeros(n.shape, F), ambient_intensity, ambient_color)
    for inten, col, direc in directionals:
        light = directional(light, n, inten, col, direc)
    return (np.asarray(textures, F) * light[:, :, None, :]).astype(F)
