"""Import alias: ``import gendr`` resolves to the MI355X build (``gendr_amd``), so that the reference's
``experiments/*.py`` and ``animations/*.py`` run unmodified against it."""
from gendr_amd import *                      # noqa: F401,F403
from gendr_amd import functional, mesh, transform, lighting, losses, renderer   # noqa: F401
