from gendr_amd.functional import *           # noqa: F401,F403
from gendr_amd.functional import render, soft_rasterize, GenDRFunction   # noqa: F401
