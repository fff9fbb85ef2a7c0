from .renderer import render, soft_rasterize, GenDRFunction
