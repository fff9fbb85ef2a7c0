"""Ambient and directional lighting terms (plain PyTorch).  Behavioural counterparts of the reference's
``gendr/functional/lighting.py:11-48``: both ADD into ``light`` in place and return it."""
import numpy as np
import torch
import torch.nn.functional as F


def _row(v, device):
    if isinstance(v, (tuple, list)):
        v = torch.tensor(v, dtype=torch.float32, device=device)
    elif isinstance(v, np.ndarray):
        v = torch.from_numpy(v).float().to(device)
    return v[None, :] if v.ndimension() == 1 else v


def ambient_lighting(light, light_intensity=0.5, light_color=(1, 1, 1)):
    """light [B, N, 3] += intensity * colour."""
    light += light_intensity * _row(light_color, light.device)[:, None, :]
    return light


def directional_lighting(light, normals, light_intensity=0.5, light_color=(1, 1, 1), light_direction=(0, 1, 0)):
    """light [B, N, 3] += intensity * colour * relu(<normal, direction>)."""
    color = _row(light_color, light.device)
    direction = _row(light_direction, light.device)
    cosine = F.relu((normals * direction).sum(dim=2))
    light += light_intensity * (color[:, None, :] * cosine[:, :, None])
    return light
