"""``gsplat.strategy`` as imported by ``pogs/pogs.py:12`` (``DefaultStrategy``) and nerfstudio's
``splatfacto.py:26`` (``DefaultStrategy, MCMCStrategy``)."""
from pogs_amd.strategy import DefaultStrategy  # noqa: F401


class MCMCStrategy:
    """Named by ``splatfacto.py:26`` and only constructed for ``strategy == "mcmc"`` (``splatfacto.py:283``), which
    POGS never selects (``POGSModelConfig`` keeps nerfstudio's default strategy): constructing it raises."""

    def __init__(self, *args, **kwargs):
        raise NotImplementedError("MCMCStrategy is outside the POGS path (POGS trains with DefaultStrategy)")
