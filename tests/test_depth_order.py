"""pr_depth_order: the order of the C*N Gaussians by (depth bits, flat index) -- what a stable sort of the depth bits
gives, and what gsplat's 64-bit (tile | depth) sort resolves ties with.  Default: the library's own chained-scan radix
sort; POGS_RAST_DEPTH_SORT=1 selects rocPRIM's radix sort, =2 a sample sort between 32 K and 1.25 M Gaussians
(splitters from 8192 samples, one LDS sort per bucket); all three against torch's stable sort, on distributions chosen
to unbalance buckets and digits."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _order(lib, depths):
    C, N = depths.shape
    out = torch.full((C * N,), -1, dtype=torch.int32, device=depths.device)
    nb = lib.pr_depth_order_scratch_bytes(C, N)
    scratch = torch.empty(nb, dtype=torch.uint8, device=depths.device)
    rc = lib.pr_depth_order(C, N, ctypes.c_void_p(depths.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                            ctypes.c_void_p(scratch.data_ptr()), nb, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    return out


def _cases(n, gen):
    yield "uniform", torch.rand(n, generator=gen) * 2 + 2
    yield "all equal", torch.full((n,), 3.25)
    d = torch.rand(n, generator=gen) * 1e-3 + 3.0          # a wall facing the camera ...
    d[::1000] = torch.rand((n + 999) // 1000, generator=gen) * 1e6   # ... and a few far outliers
    yield "wall and outliers", d
    yield "exponential", torch.empty(n).exponential_(1.0, generator=gen) + 0.01
    d = torch.rand(n, generator=gen) * 40 + 0.05
    d[torch.rand(n, generator=gen) < 0.5] = 0.0            # culled entries are zeros
    d[n // 3:n // 3 + 5000] = d[min(7, n - 1)]             # a run of ties
    yield "half culled with ties", d
    yield "few distinct values", torch.randint(0, 5, (n,), generator=gen).float() + 1.0
    yield "sorted descending", torch.arange(n, 0, -1).float()


_SORTS = {"chained": None, "rocprim": "1", "sample": "2"}


@pytest.mark.parametrize("which", list(_SORTS))
@pytest.mark.parametrize("n", [32_767, 32_768, 100_003, 1_000_000, 1_250_000, 1_250_001])
def test_depth_order_matches_a_stable_sort(gpu, n, which, tuning):
    from pogs_amd import _lib

    lib = _lib.load()
    tuning(POGS_RAST_DEPTH_SORT=_SORTS[which])
    gen = torch.Generator().manual_seed(n)
    for name, d in _cases(n, gen):
        d = d.to(gpu).view(1, n)
        want = torch.sort(d.flatten().view(torch.int32), stable=True).indices
        got = _order(lib, d)
        assert torch.equal(got.long(), want), (name, n)
    # two cameras: one order over all C*N entries
    d = (torch.rand(2, n // 2, generator=gen) * 5 + 0.5).to(gpu)
    assert torch.equal(_order(lib, d).long(), torch.sort(d.flatten().view(torch.int32), stable=True).indices)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 4095, 4096, 4097, 8192, 12_289])
def test_chained_sort_around_its_tile_size(gpu, n):
    """One tile of the chained-scan sort holds 4096 pairs: sizes around one and two tiles, and tiny ones."""
    from pogs_amd import _lib

    lib = _lib.load()
    gen = torch.Generator().manual_seed(n)
    for name, d in _cases(n, gen):
        d = d.to(gpu).view(1, n)
        want = torch.sort(d.flatten().view(torch.int32), stable=True).indices
        assert torch.equal(_order(lib, d).long(), want), (name, n)


def test_both_sorts_agree(gpu, tuning):
    from pogs_amd import _lib

    lib = _lib.load()
    d = (torch.rand(1, 400_000, generator=torch.Generator().manual_seed(3)) * 9 + 0.1).to(gpu)
    a = _order(lib, d)
    tuning(POGS_RAST_DEPTH_SORT="2")       # sample sort
    b = _order(lib, d)
    assert torch.equal(a, b)
    tuning(POGS_RAST_DEPTH_SORT="1")       # rocPRIM
    assert torch.equal(a, _order(lib, d))
