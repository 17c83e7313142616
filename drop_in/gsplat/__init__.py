"""Import shim: put ``drop_in/`` on ``PYTHONPATH`` and POGS's unmodified ``from gsplat.rendering import
rasterization`` (``pogs/pogs.py:11``) resolves to the MI355X rasterizer.  Only the names the reference imports
are provided."""
from pogs_amd.rendering import rasterization  # noqa: F401

__version__ = "1.4.0+pogs_amd"
