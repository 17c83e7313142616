"""The C-ABI library loads on a machine without a GPU and exports exactly what include/pogs_rast.h declares.
No compute entry point is called here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "pogs_rast.h")


def _declared():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pr_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    names = _declared()
    for must in ("pr_project_fwd", "pr_project_bwd", "pr_sh_fwd", "pr_sh_bwd", "pr_isect_count",
                 "pr_isect_emit_sort", "pr_isect_offsets", "pr_raster_fwd", "pr_raster_bwd", "pr_version"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from pogs_amd import _lib

    assert os.path.exists(_lib.LIB_PATH), "build with `make -C pogs_amd/csrc` (or __graft_entry__.build())"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), f"{name} declared in pogs_rast.h but not exported"


def test_python_binding_covers_the_header():
    from pogs_amd import _lib

    assert sorted(_lib.SIGNATURES.keys()) == _declared()
    lib = _lib.load()
    assert lib.pr_version() >> 16 == 1
    assert lib.pr_error_string(0) == b"ok"
    assert b"invalid" in lib.pr_error_string(-1)
    assert lib.pr_max_channels() >= 97  # widest render in the reference: 96 hash features (+1 depth)


def test_argument_validation_needs_no_gpu():
    """Entry points reject malformed arguments before touching the device."""
    from pogs_amd import _lib

    lib = _lib.load()
    assert lib.pr_project_fwd(1, 4, None, None, None, None, None, 0, 10, 0.3, 0.01, 1e10, 0.0,
                              None, None, None, None, None, None) == -1          # width 0
    assert lib.pr_project_fwd(1, 4, None, None, None, None, None, 8, 8, 0.3, 0.01, 1e10, 0.0,
                              None, None, None, None, None, None) == -1          # null pointers
    assert lib.pr_sh_fwd(1, 4, 16, 4, None, None, None, None, 1, None, 3, None) == -2   # degree 4 unsupported
    tail = (None, None, None, None, 0, None)  # render_colors, render_alphas, last_ids, segments, n_segments, stream
    assert lib.pr_raster_fwd(1, 4, 3, None, None, None, None, None, 64, 64, 12, 6, 6, None, None, None, 0, -1,
                             *tail) == -2                        # tile size 12 unsupported
    assert lib.pr_raster_fwd(1, 4, 3, None, None, None, None, None, 64, 64, 16, 2, 2, None, None, None, 0, -1,
                             *tail) == -1                        # tile grid does not cover the image
    assert lib.pr_raster_fwd(1, 4, 3, None, None, None, None, None, 64, 64, 16, 4, 4, None, None, None, 0, 3,
                             *tail) == -1                        # depth channel index out of range
    assert lib.pr_raster_fwd(1, 4, 101, None, None, None, None, None, 64, 64, 16, 4, 4, None, None, None, 0, -1,
                             *tail) == -2                        # more channels than any kernel takes
    # channel segments: widths must add up to D, at most three
    keep, segs = _lib.segment_array([(None, 3, 3, 0), (None, 2, 2, 1)])
    out = ctypes.create_string_buffer(64)   # stands in for the output pointers: rejected before any use
    o = ctypes.cast(out, ctypes.c_void_p)
    assert lib.pr_raster_fwd(1, 4, 6, None, None, None, None, None, 64, 64, 16, 4, 4, o, None, None, 0, -1,
                             o, o, o, segs, 2, None) == -1       # 3 + 2 != 6
    assert lib.pr_raster_fwd(1, 4, 5, None, None, None, None, None, 64, 64, 16, 4, 4, o, None, None, 0, -1,
                             o, o, o, segs, 4, None) == -1       # more than three segments


def test_rasterization_fails_loudly_without_gpu_tensors():
    import torch

    from pogs_amd import rendering
    from tests._util import scene_args, small_scene

    a = scene_args(small_scene(n=8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        rendering.rasterization(**a, sh_degree=3)
    del torch


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "pogs_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "oracle/" not in text, f


def test_gsplat_import_shim_resolves_to_this_package():
    import subprocess
    import sys

    # exactly the names the reference imports: pogs/pogs.py:11-12, pogs/pogs_pipeline.py:29,
    # nerfstudio/models/splatfacto.py:26,29
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r);"
            "from gsplat.rendering import rasterization; import pogs_amd.rendering as r;"
            "from gsplat.strategy import DefaultStrategy; from gsplat.strategy import DefaultStrategy, MCMCStrategy;"
            "import pogs_amd.strategy as st; assert DefaultStrategy is st.DefaultStrategy;"
            "s = DefaultStrategy(prune_opa=0.1, grow_grad2d=0.0008, grow_scale3d=0.01, grow_scale2d=0.05,"
            " prune_scale3d=0.5, prune_scale2d=0.15, refine_scale2d_stop_iter=0, refine_start_iter=500,"
            " refine_stop_iter=15000, reset_every=3000, refine_every=100, pause_refine_after_reset=0, absgrad=True,"
            " revised_opacity=False, verbose=True);"
            "assert 'grad2d' in s.initialize_state(scene_scale=1.0);"
            "\ntry:\n    MCMCStrategy()\n    raise SystemExit('MCMCStrategy must raise')\nexcept NotImplementedError:\n    pass\n"
            "from gsplat.cuda._torch_impl import _quat_to_rotmat; import torch;"
            "assert rasterization is r.rasterization;"
            "assert torch.allclose(_quat_to_rotmat(torch.tensor([1.,0,0,0])), torch.eye(3))"
            % (os.path.join(ROOT, "drop_in"), ROOT))
    subprocess.run([sys.executable, "-c", code], check=True)
