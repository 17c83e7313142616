import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")


_TUNING_VARS = ("POGS_RAST_BWD_WINDOW", "POGS_RAST_FWD_BLOCKS", "POGS_RAST_FWD_PPL", "POGS_RAST_FWD_CHUNK", "POGS_RAST_FWD_MFMA", "POGS_RAST_BWD_PPL",
                "POGS_RAST_BWD_MFMA", "POGS_MLP_VARIANT", "POGS_RAST_DEPTH_SORT", "POGS_RAST_BWD_SPLIT")


@pytest.fixture
def tuning():
    """``tuning(POGS_RAST_BWD_PPL=4, ...)`` forces kernel variants for one test: the library reads these variables
    once, so the fixture sets them, has the library re-read them, and restores both afterwards."""
    from pogs_amd import _lib

    saved = {k: os.environ.get(k) for k in _TUNING_VARS}

    def apply(**overrides):
        assert set(overrides) <= set(_TUNING_VARS), overrides
        _lib.set_tuning(**overrides)

    yield apply
    _lib.set_tuning(**saved)


def pytest_sessionfinish(session, exitstatus):
    """Worst pure-relative gradient errors seen by assert_grad_close (elements above 1e-3 of the largest), per
    comparison: written next to the other run outputs so that the tolerance actually met can be reported."""
    try:
        from tests._util import REL_ERRORS
    except Exception:
        return
    if not REL_ERRORS:
        return
    import json

    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    worst = sorted(REL_ERRORS, key=lambda r: -r[1])
    rec = {"comparisons": len(REL_ERRORS), "worst_relative_error": worst[0][1],
           "median_of_worsts": sorted(r[1] for r in REL_ERRORS)[len(REL_ERRORS) // 2],
           "over_1e-3": sum(1 for r in REL_ERRORS if r[1] > 1e-3), "top": worst[:25]}
    try:
        from tests._util import MARGINS, PRECISION_PAIRS
        rec["closest_to_their_bound (error / allowed)"] = sorted(MARGINS, key=lambda r: -r[1])[:15]
        if PRECISION_PAIRS:
            by_tensor = {}
            for name, h, c, n in PRECISION_PAIRS:
                t = by_tensor.setdefault(name.split()[0], {"comparisons": 0, "hip_vs_f64_worst": 0.0, "c32_vs_f64_worst": 0.0})
                t["comparisons"] += 1
                t["hip_vs_f64_worst"] = max(t["hip_vs_f64_worst"], h)
                t["c32_vs_f64_worst"] = max(t["c32_vs_f64_worst"], c)
            for t in by_tensor.values():
                t["hip_over_c32"] = t["hip_vs_f64_worst"] / t["c32_vs_f64_worst"] if t["c32_vs_f64_worst"] else None
            rec["float32_restatement_next_to_the_kernels"] = {
                "what": "worst |x - f64| / |f64| over elements above 1e-3 max|f64|, same inputs and cotangents: HIP "
                        "kernels and oracle/raster_c.c (float32, gsplat's operation order, scalar loops)",
                "per_tensor": by_tensor,
                "pairs (name, hip_vs_f64, c32_vs_f64, elements)": sorted(PRECISION_PAIRS, key=lambda r: -r[1])[:40]}
    except Exception:
        pass
    json.dump(rec, open(os.path.join(out_dir, "grad_relative_errors.json"), "w"), indent=1)
