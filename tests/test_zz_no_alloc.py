"""The entry point that allocates nothing, ``pr_depth_order_static``: through the C ABI, against ``pr_depth_order``
(the same kernels on a caller-supplied scratch buffer), bit-exact, on two calls in a row, the second with a smaller
input (the static array still holds the first call's contents), and its size limit."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _both(lib, depths):
    C, N = depths.shape
    stream = torch.cuda.current_stream().cuda_stream
    want = torch.empty(C * N, dtype=torch.int32, device=depths.device)
    nb = lib.pr_depth_order_scratch_bytes(C, N)
    scratch = torch.empty(nb, dtype=torch.uint8, device=depths.device)
    assert lib.pr_depth_order(C, N, _ptr(depths), _ptr(want), _ptr(scratch), nb, stream) == 0
    got = torch.full((C * N,), -1, dtype=torch.int32, device=depths.device)
    # torch's allocator counter sees allocations made through torch only; the driver's free-memory figure also sees a
    # hipMalloc made inside the library.  (Compared on calls after a device's first one, which loads nothing new.)
    torch.cuda.synchronize()
    before, free_before = torch.cuda.memory_allocated(), torch.cuda.mem_get_info()[0]
    assert lib.pr_depth_order_static(C, N, _ptr(depths), _ptr(got), stream) == 0
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == before
    if _both.warm:
        assert torch.cuda.mem_get_info()[0] >= free_before, "device memory was allocated inside the call"
    _both.warm = True
    return got, want


_both.warm = False


def test_depth_order_without_allocation_matches_the_scratch_buffer_entry_point(gpu):
    from pogs_amd import _lib

    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    first = (torch.rand(2, 300_001, generator=g) * 40 + 0.05).to(gpu)
    first[0, 1000:1400] = first[0, 7]           # ties resolve by ascending index
    first[1, ::9] = 0.0                          # culled entries are written as zeros by the projection
    got, want = _both(lib, first)
    assert torch.equal(got, want)
    assert torch.equal(got.long(), torch.sort(first.flatten().view(torch.int32), stable=True).indices)
    second = (torch.rand(1, 4_097, generator=g) * 3 + 0.5).to(gpu)     # smaller: stale keys lie behind it in the array
    got2, want2 = _both(lib, second)
    assert torch.equal(got2, want2)
    assert torch.equal(got2.long(), torch.sort(second.flatten().view(torch.int32), stable=True).indices)
    got3, want3 = _both(lib, first)              # and the large input again
    assert torch.equal(got3, want3)


def test_depth_order_without_allocation_rejects_what_does_not_fit(gpu):
    from pogs_amd import _lib

    lib = _lib.load()
    limit = lib.pr_depth_order_static_max()
    assert 1_000_000 <= limit < 4_000_000
    d = torch.rand(1, 16, device=gpu)
    out = torch.empty(16, dtype=torch.int32, device=gpu)
    assert lib.pr_depth_order_static(1, limit + 1, _ptr(d), _ptr(out), None) == -2     # checked before any launch
    assert b"static scratch" in lib.pr_last_hip_error()
    with pytest.raises(RuntimeError, match="static scratch"):
        _lib.check(lib.pr_depth_order_static(2, limit, _ptr(d), _ptr(out), None), "pr_depth_order_static")
    assert lib.pr_depth_order_static(1, 0, None, None, None) == 0
    assert lib.pr_depth_order_static(1, 16, None, _ptr(out), None) == -1
    # at the limit itself it runs
    big = (torch.rand(1, limit, generator=torch.Generator().manual_seed(1)) + 0.1).to(gpu)
    got, want = _both(lib, big)
    assert torch.equal(got, want)
