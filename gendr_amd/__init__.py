"""gendr_amd -- MI355X-native generalized soft rasterizer (the one hot path of Felix-Petersen/gendr).

``GenDR`` / ``functional.render`` / ``functional.GenDRFunction`` keep the reference's Python surface;
the per-pixel face loop and its backward are hand-written gfx950 HIP behind a C ABI
(``include/gendr_hip.h`` -> ``gendr_amd/libgendr_hip.so``).
"""
from . import functional
from .renderer import GenDR

__all__ = ['functional', 'GenDR']
