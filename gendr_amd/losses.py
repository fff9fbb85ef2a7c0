"""Mesh regularisers (plain PyTorch; reference ``gendr/losses.py:11-120``)."""
import numpy as np
import torch
import torch.nn as nn


class LaplacianLoss(nn.Module):
    """|L x|^2 with the row-normalised graph Laplacian of the mesh (dense nv x nv, nv <= ~1.4k in the scripts)."""

    def __init__(self, vertex, faces, average=False):
        super().__init__()
        self.nv, self.nf, self.average = vertex.size(0), faces.size(0), average
        f = faces.detach().cpu().numpy().astype(np.int64)
        adj = np.zeros((self.nv, self.nv), dtype=np.float32)
        for a, b in ((0, 1), (1, 2), (2, 0)):
            adj[f[:, a], f[:, b]] = -1
            adj[f[:, b], f[:, a]] = -1
        degree = -adj.sum(1)
        adj[np.arange(self.nv), np.arange(self.nv)] = degree
        adj = adj / degree[:, None]
        self.register_buffer('laplacian', torch.from_numpy(adj))

    def forward(self, x):
        y = torch.matmul(self.laplacian, x)
        per_item = y.pow(2).sum(tuple(range(1, y.ndimension())))
        return per_item.sum() / x.size(0) if self.average else per_item


class FlattenLoss(nn.Module):
    """(cos(dihedral) + 1)^2 summed over the reference's edge set (``gendr/losses.py:47-76``): the edges that
    appear as index columns (0, 1) or (1, 2) of some face -- an interior edge that is the (2, 0) edge of BOTH
    its faces is not in the set, exactly as in the reference.  For each edge, v2 / v3 are the opposite vertices of
    the first / second face (in face order) that contains it.  Edges with a single adjacent face are skipped (the
    reference's v2s / v3s lists fall out of step there)."""

    def __init__(self, faces, average=False):
        super().__init__()
        self.nf, self.average = faces.size(0), average
        f = faces.detach().cpu().numpy().astype(np.int64)
        wanted = set()
        for a, b in ((0, 1), (1, 2)):
            for tri in f:
                wanted.add((min(tri[a], tri[b]), max(tri[a], tri[b])))
        opposite = {}
        for tri in f:
            for a, b, c in ((0, 1, 2), (1, 2, 0), (2, 0, 1)):
                key = (min(tri[a], tri[b]), max(tri[a], tri[b]))
                if key in wanted:
                    opposite.setdefault(key, []).append(tri[c])
        rows = [(e[0], e[1], o[0], o[1]) for e, o in sorted(opposite.items()) if len(o) >= 2]
        idx = np.asarray(rows, dtype=np.int64).reshape(-1, 4)
        for k, name in enumerate(('v0s', 'v1s', 'v2s', 'v3s')):
            self.register_buffer(name, torch.from_numpy(idx[:, k].copy()))

    @staticmethod
    def _perp(a, b, eps):
        """Component of b orthogonal to a, and its length computed as |b| sin(angle) (with the eps guards of the reference)."""
        a2 = a.pow(2).sum(-1)
        b1 = (b.pow(2).sum(-1) + eps).sqrt()
        ab = (a * b).sum(-1)
        cos = ab / ((a2 + eps).sqrt() * b1 + eps)
        sin = (1 - cos.pow(2) + eps).sqrt()
        return b - a * (ab / (a2 + eps))[:, :, None], b1 * sin

    def forward(self, vertices, eps=1e-6):
        v0, v1 = vertices[:, self.v0s], vertices[:, self.v1s]
        edge = v1 - v0
        c1, l1 = self._perp(edge, vertices[:, self.v2s] - v0, eps)
        c2, l2 = self._perp(edge, vertices[:, self.v3s] - v0, eps)
        cos = (c1 * c2).sum(-1) / (l1 * l2 + eps)
        per_item = (cos + 1).pow(2).sum(tuple(range(1, cos.ndimension())))
        return per_item.sum() / vertices.size(0) if self.average else per_item
