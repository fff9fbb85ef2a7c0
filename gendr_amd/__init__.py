"""gendr_amd -- MI355X-native generalized soft rasterizer (the one hot path of Felix-Petersen/gendr).

``GenDR`` / ``functional.render`` / ``functional.GenDRFunction`` keep the reference's Python surface;
the per-pixel face loop and its backward are hand-written gfx950 HIP behind a C ABI
(``include/gendr_hip.h`` -> ``gendr_amd/libgendr_hip.so``).  Mesh / camera / lighting / loss classes are
plain-PyTorch glue so that the reference's experiment scripts find the names they import.
"""
from . import functional
from .mesh import Mesh
from .transform import Projection, LookAt, Look
from .lighting import AmbientLighting, DirectionalLighting, Lighting
from .renderer import GenDR
from .losses import LaplacianLoss, FlattenLoss

__all__ = ['functional', 'Mesh', 'Projection', 'LookAt', 'Look', 'AmbientLighting', 'DirectionalLighting',
           'Lighting', 'GenDR', 'LaplacianLoss', 'FlattenLoss']
