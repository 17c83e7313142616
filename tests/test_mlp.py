"""Fused channel MLP (pr_mlp_fwd / pr_mlp_bwd_input) against the torch restatement of pogs/pogs.py:153-161,516-522.

Two arithmetics (POGS_MLP_VARIANT): exact float32 MFMA -- differs from a float64 reference by summation order
only, four chained 64-term dot products of O(1) values agree to ~1e-5 relative, bound 2e-5 * scale -- and the
default split-bf16 (three bf16 products per float32 product, 2^-16 per product), bound 1e-4 * scale, the
tolerance the rendered colours themselves are held to.  Gradients: 1e-3 relative in both.
"""
import ctypes

import pytest
import torch

from oracle.mlp_ref import feature_mlp_ref
from pogs_amd import _lib
from tests._util import relu_ambiguous as _relu_ambiguous


ARITH = {"f32": ("1", 2e-5), "bf16x3": ("4", 1e-4)}


@pytest.fixture(params=sorted(ARITH))
def arith(request, tuning):
    variant, tol = ARITH[request.param]
    tuning(POGS_MLP_VARIANT=variant)
    return tol


def _weights(gen, scale=1.0):
    # torch.nn.Linear's default init is U(-1/8, 1/8) for 64 inputs; `scale` widens it so that ReLUs see both signs
    return [((torch.rand(64, 64, generator=gen, dtype=torch.float64) * 2 - 1) / 8 * scale) for _ in range(4)]


def test_mask_words_and_validation():
    lib = _lib.load()
    assert lib.pr_mlp_mask_words(0) == 0
    assert lib.pr_mlp_mask_words(1) == 96          # one 32-pixel tile: 64 lanes x 3 layers x 32 bits
    assert lib.pr_mlp_mask_words(64) == 2 * 96
    assert lib.pr_mlp_mask_words(65) == 3 * 96
    # no GPU is touched by these: argument errors are found first
    assert lib.pr_mlp_fwd(-1, None, None, None, None, None, None, 0.0, None, None, None) == -1
    assert lib.pr_mlp_fwd(0, None, None, None, None, None, None, 0.0, None, None, None) == 0
    assert lib.pr_mlp_fwd(8, None, None, None, None, None, None, 0.0, None, None, None) == -1
    assert lib.pr_mlp_bwd_input(8, None, None, None, None, None, None, 0.0, None, None, None) == -1
    buf = (ctypes.c_float * 1024)()
    base = ctypes.addressof(buf)
    mis = ctypes.c_void_p(base + 4)
    ok = ctypes.c_void_p(base + (16 - base % 16) % 16)
    assert lib.pr_mlp_fwd(1, mis, None, ok, ok, ok, ok, 0.0, ok, None, None) == -1


def test_oracle_matches_sequential():
    """The restatement against the reference's own construction: a torch.nn.Sequential with the same weights."""
    gen = torch.Generator().manual_seed(3)
    ws = [w.float() for w in _weights(gen)]
    seq = torch.nn.Sequential(torch.nn.Linear(64, 64, bias=False), torch.nn.ReLU(), torch.nn.Linear(64, 64, bias=False),
                              torch.nn.ReLU(), torch.nn.Linear(64, 64, bias=False), torch.nn.ReLU(),
                              torch.nn.Linear(64, 64, bias=False))
    with torch.no_grad():
        for lin, w in zip([seq[0], seq[2], seq[4], seq[6]], ws):
            lin.weight.copy_(w)
    x = torch.randn(1, 5, 7, 64, generator=gen)
    torch.testing.assert_close(feature_mlp_ref(x, None, ws), seq(x.view(-1, 64)).view(1, 5, 7, 64))


def test_module_is_a_sequential_with_reference_keys():
    from pogs_amd.mlp import FeatureMLP
    m = FeatureMLP()
    assert list(m.state_dict().keys()) == ["0.weight", "2.weight", "4.weight", "6.weight"]
    with pytest.raises(RuntimeError, match="HIP device"):
        m(torch.zeros(4, 64), torch.ones(4, 1))


@pytest.mark.gpu
@pytest.mark.parametrize("P", [1, 63, 64, 65, 1000, 256 * 1024 + 17])
@pytest.mark.parametrize("with_alpha", [False, True])
def test_forward_parity(P, with_alpha, arith):
    from pogs_amd.mlp import feature_mlp
    gen = torch.Generator().manual_seed(P)
    ws = _weights(gen, scale=3.0)
    x = torch.randn(P, 64, generator=gen, dtype=torch.float64)
    al = None
    if with_alpha:
        al = torch.rand(P, 1, generator=gen, dtype=torch.float64)
        al[::7] = 0.0
        al[3::11] = 1e-8  # below the clamp
        x = x * al.clamp(min=1e-3)
    ref = feature_mlp_ref(x.float().double(), None if al is None else al.float().double(), [w.float().double() for w in ws])
    got = feature_mlp(x.float().cuda(), None if al is None else al.float().cuda(), [w.float().cuda() for w in ws]).cpu()
    scale = max(ref.abs().max().item(), 1.0)
    assert (got.double() - ref).abs().max().item() <= arith * scale


@pytest.mark.gpu
def test_eval_cut_and_image_shape(arith):
    from pogs_amd.mlp import FeatureMLP
    gen = torch.Generator().manual_seed(11)
    m = FeatureMLP(device="cuda").eval()
    x = torch.randn(1, 37, 53, 64, generator=gen)
    al = torch.rand(1, 37, 53, 1, generator=gen)
    with torch.no_grad():
        got = m(x.cuda(), al.cuda()).cpu()
    ws = [w.detach().cpu().double() for w in m.linear_weights()]
    ref = feature_mlp_ref(x.double(), al.double(), ws, zero_below=0.8)
    assert got.shape == (1, 37, 53, 64)
    assert (got[al.squeeze(-1) < 0.8] == 0).all()
    assert (got.double() - ref).abs().max().item() <= arith * max(ref.abs().max().item(), 1.0)
    m.train()
    with torch.no_grad():
        got = m(x.cuda(), al.cuda()).cpu()
    ref = feature_mlp_ref(x.double(), al.double(), ws)
    assert (got.double() - ref).abs().max().item() <= arith * max(ref.abs().max().item(), 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("P", [65, 5000])
@pytest.mark.parametrize("train_weights", [False, True])
def test_backward_parity(P, train_weights, arith):
    """Frozen weights: the fused input-gradient kernel.  Trainable weights: input and weight gradients."""
    from pogs_amd.mlp import feature_mlp
    gen = torch.Generator().manual_seed(100 + P)
    ws = [w.float() for w in _weights(gen, scale=3.0)]
    x = torch.randn(P, 64, generator=gen)
    al = torch.rand(P, 1, generator=gen)
    al[::5] = 0.0
    v = torch.randn(P, 64, generator=gen)
    v[_relu_ambiguous(x, al, ws)] = 0.0

    xr = x.double().requires_grad_(True)
    wr = [w.double().requires_grad_(train_weights) for w in ws]
    feature_mlp_ref(xr, al.double(), wr).backward(v.double())

    xg = x.cuda().requires_grad_(True)
    wg = [w.cuda().requires_grad_(train_weights) for w in ws]
    feature_mlp(xg, al.cuda(), wg).backward(v.cuda())

    def close(a, b):
        assert (a.cpu().double() - b).abs().max().item() <= 1e-3 * max(b.abs().max().item(), 1e-6)  # grads: 1e-3 relative

    close(xg.grad, xr.grad)
    if train_weights:
        for a, b in zip(wg, wr):
            close(a.grad, b.grad)
    else:
        assert all(w.grad is None for w in wg)


@pytest.mark.gpu
@pytest.mark.parametrize("P,with_alpha,cut,input_grad", [(33, True, 0.0, False), (70001, False, 0.0, True),
                                                         (4096, True, 0.8, True)])
def test_fused_weight_gradients(P, with_alpha, cut, input_grad):
    """pr_mlp_fwd_train + pr_mlp_bwd_full: weight gradients (and, optionally, the feature gradient) of the whole
    module from one backward kernel; partial last tile, no alpha, the eval-mode cut, features without gradient."""
    from pogs_amd.mlp import feature_mlp
    gen = torch.Generator().manual_seed(7 + P)
    ws = [w.float() for w in _weights(gen, scale=3.0)]
    x = torch.randn(P, 64, generator=gen)
    al = torch.rand(P, 1, generator=gen) if with_alpha else None
    if al is not None:
        al[::7] = 0.0
    v = torch.randn(P, 64, generator=gen)
    amb = _relu_ambiguous(x, al, ws)
    assert amb.float().mean().item() < 0.25
    v[amb] = 0.0
    xr = x.double().requires_grad_(input_grad)
    wr = [w.double().requires_grad_(True) for w in ws]
    ref = feature_mlp_ref(xr, None if al is None else al.double(), wr, cut)
    ref.backward(v.double())
    xg = x.cuda().requires_grad_(input_grad)
    wg = [w.cuda().requires_grad_(True) for w in ws]
    out = feature_mlp(xg, None if al is None else al.cuda(), wg, cut)
    assert (out.detach().cpu().double() - ref.detach()).abs().max().item() <= 1e-4 * max(ref.abs().max().item(), 1.0)
    out.backward(v.cuda())
    for a, b in zip(wg, wr):
        assert a.grad.shape == (64, 64)
        assert (a.grad.cpu().double() - b.grad).abs().max().item() <= 1e-3 * b.grad.abs().max().item()
    if input_grad:
        assert (xg.grad.cpu().double() - xr.grad).abs().max().item() <= 1e-3 * xr.grad.abs().max().item()
    else:
        assert xg.grad is None
