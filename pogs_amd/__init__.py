"""pogs_amd: MI355X-native Gaussian-splat rasterizer behind the gsplat ``rasterization()`` call that POGS
uses (``pogs/pogs.py:352,419,493,930``).  ``from pogs_amd.rendering import rasterization``."""
__version__ = "0.1.0"
