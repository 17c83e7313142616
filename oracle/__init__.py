"""CPU oracle for the Gaussian-splat rasterizer path.  TEST INFRASTRUCTURE ONLY.

This package is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  ``pogs_amd`` never imports it and fails loudly without its HIP
library.

PARITY UNPINNED.  The arithmetic of this path lives in the third-party
package gsplat (pinned ``gsplat==1.4.0`` by
``src/pogs/dependencies/nerfstudio/pyproject.toml:67`` of the reference).
gsplat is not vendored in the reference, not installed here, and the
reference's own tests hold no golden vector for the rasterizer
(``src/pogs/dependencies/nerfstudio/tests/test_splatfacto_integration.py:37-38``
only gates PSNR/SSIM after a networked training run).  The functions below
restate gsplat 1.4.0's published algorithm (``gsplat/rendering.py`` and
``gsplat/cuda/_torch_impl.py``) and are anchored on the reference's call
sites ``pogs/pogs.py:352-371,419-437,493-510,930-948``.  They are pinned only
by closed-form known-answer tests and by mutual agreement between three
independent restatements (float64 brute force, float32 tile lists, C scalar
loops); see ``tests/test_oracle.py``.
"""
