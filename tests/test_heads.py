"""CLIP / instance heads (pr_dense_*, pr_l2norm_*, pogs_amd.heads) against the float64 restatement of
pogs/fields/gaussian_field.py:105-127,176-204.  tiny-cuda-nn is absent from the reference tree, so the parameter layout
is unpinned (see oracle/heads_ref.py); what these tests pin is the arithmetic: forward to 1e-4 of the output scale
(split-bf16 products), gradients to 1e-3 on rows whose ReLU pre-activations are not within the forward error of zero."""
import ctypes

import pytest
import torch

from oracle.heads_ref import dense_network_ref, heads_from_feature_ref
from pogs_amd import _lib


def test_layer_shapes_and_validation():
    from pogs_amd.heads import HeadNames, layer_shapes
    assert layer_shapes(97, 512, 256, 3) == [(256, 112), (256, 256), (256, 256), (512, 256)]
    assert layer_shapes(96, 128, 256, 4) == [(256, 96), (256, 256), (256, 256), (256, 256), (128, 256)]
    assert [m.value for m in HeadNames] == ["hashgrid", "clip", "instance"]
    lib = _lib.load()
    assert lib.pr_dense_packed_bytes(256, 112) == 8 * 8 * 64 * 32          # 8 out tiles x 8 k steps (112 -> 128)
    assert lib.pr_dense_packed_bytes(256, 512) == 2 * 8 * 16 * 64 * 32      # two 256-wide chunks
    assert lib.pr_dense_packed_bytes(0, 5) == 0
    buf = (ctypes.c_float * 64)()
    ok = ctypes.c_void_p(ctypes.addressof(buf) + (16 - ctypes.addressof(buf) % 16) % 16)
    # argument errors are found before any GPU call
    assert lib.pr_dense_fwd(4, 6, 8, ok, 8, ok, 0, None, 0, ok, 8, None) == -1      # M not a multiple of 4
    assert lib.pr_dense_fwd(4, 8, 8, ok, 4, ok, 0, None, 0, ok, 8, None) == -1      # ldx < K
    assert lib.pr_dense_fwd(0, 8, 8, None, 8, None, 0, None, 0, None, 8, None) == 0
    assert lib.pr_dense_wgrad(4, 8, 8, None, 8, ok, 8, ok, 8, None) == -1
    assert lib.pr_l2norm_fwd(4, 6, ok, 0.0, ok, None, None) == -1


def test_module_state_and_cpu_refusal():
    from pogs_amd.heads import FieldHeads
    m = FieldHeads(seed=1)
    assert sorted(m.state_dict()) == ["clip_net.params", "instance_net.params"]
    assert m.clip_net.params.numel() == 256 * 112 + 2 * 256 * 256 + 512 * 256
    assert m.instance_net.params.numel() == 256 * 96 + 3 * 256 * 256 + 128 * 256
    with pytest.raises(RuntimeError, match="HIP device"):
        m.get_instance_outputs_from_feature(torch.zeros(4, 96))


def _ambiguous(pre, rel=2e-4):
    amb = torch.zeros(pre[0].shape[0], dtype=torch.bool)
    for p in pre:
        amb |= (p.abs() < rel * p.abs().amax(-1, keepdim=True).clamp(min=1.0)).any(-1)
    return amb


@pytest.mark.gpu
@pytest.mark.parametrize("P,M,K", [(1, 32, 64), (33, 8, 16), (200, 512, 256), (129, 96, 512), (1000, 256, 112), (77, 36, 260)])
def test_dense_layer(P, M, K):
    """One layer: ragged rows, widths off the 32 / 64 grid, the 512-wide contraction done as two chunks."""
    from pogs_amd.heads import _call, _packed, _ptr, _stream
    gen = torch.Generator().manual_seed(P + M)
    x = torch.randn(P, K, generator=gen) * 3
    w = torch.randn(M, K, generator=gen)
    mask = torch.randn(P, M, generator=gen)
    ref = x.double() @ w.double().t()
    xg, wg, mg, x2g = x.cuda(), w.cuda(), mask.cuda(), None
    pk, pkt = _packed(wg, False), _packed(wg, True)
    for relu, use_mask in ((0, False), (1, False), (0, True)):
        y = torch.full((P, M), float("nan"), device="cuda")
        _call("pr_dense_fwd", P, M, K, _ptr(xg), K, _ptr(pk), relu, _ptr(mg) if use_mask else None,
              M, _ptr(y), M, _stream())
        want = torch.relu(ref) if relu else ref
        if use_mask:
            want = want * (mask > 0)
        assert (y.cpu().double() - want).abs().max().item() <= 1e-4 * ref.abs().max().item()
    # the transposed packing: y = x2 W  (the backward data product)
    x2 = torch.randn(P, M, generator=gen)
    y = torch.empty(P, K, device="cuda")
    x2g = x2.cuda()
    _call("pr_dense_fwd", P, K, M, _ptr(x2g), M, _ptr(pkt), 0, None, 0, _ptr(y), K, _stream())
    ref2 = x2.double() @ w.double()
    assert (y.cpu().double() - ref2).abs().max().item() <= 1e-4 * ref2.abs().max().item()
    # weight gradient: dW += G^T H, accumulated on top of what is there
    dw = torch.ones(M, K, device="cuda")
    _call("pr_dense_wgrad", P, M, K, _ptr(x2g), M, _ptr(xg), K, _ptr(dw), K, _stream())
    ref3 = 1.0 + x2.double().t() @ x.double()
    assert (dw.cpu().double() - ref3).abs().max().item() <= 1e-4 * ref3.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("N,eps", [(512, 0.0), (128, 1e-5), (4, 0.5)])
def test_l2norm(N, eps):
    from pogs_amd.heads import _call, _ptr, _stream
    gen = torch.Generator().manual_seed(N)
    y = torch.randn(37, N, generator=gen)
    g = torch.randn(37, N, generator=gen)
    yr = y.double().requires_grad_(True)
    ref = yr / (yr.norm(dim=-1, keepdim=True) + eps)
    ref.backward(g.double())
    out, norms, dy = torch.empty(37, N, device="cuda"), torch.empty(37, device="cuda"), torch.empty(37, N, device="cuda")
    yg, gg = y.cuda(), g.cuda()
    _call("pr_l2norm_fwd", 37, N, _ptr(yg), eps, _ptr(out), _ptr(norms), _stream())
    _call("pr_l2norm_bwd", 37, N, _ptr(yg), _ptr(norms), eps, _ptr(gg), _ptr(dy), _stream())
    assert torch.allclose(out.cpu().double(), ref.detach(), rtol=0, atol=2e-6)
    assert torch.allclose(norms.cpu().double(), yr.detach().norm(dim=-1), rtol=1e-6, atol=0)
    assert (dy.cpu().double() - yr.grad).abs().max().item() <= 1e-5 * yr.grad.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("P,n_random", [(300, None), (5000, 777)])
def test_heads_forward_backward(P, n_random):
    """get_outputs_from_feature as pogs/pogs.py:447 calls it (random pixel subset for CLIP, all pixels for the instance
    head), forward and every gradient, against the float64 restatement."""
    from pogs_amd.heads import FieldHeads, HeadNames
    heads = FieldHeads(device="cuda", seed=5)
    gen = torch.Generator().manual_seed(P)
    feats = torch.randn(P, 96, generator=gen)
    rp = None if n_random is None else torch.randperm(P, generator=gen)[:n_random]
    nc = P if rp is None else n_random
    scale = torch.rand(nc, 1, generator=gen) * 2
    clip = (heads.clip_net.params.detach().cpu().double().requires_grad_(True), heads.clip_net.shapes, 97, 512)
    inst = (heads.instance_net.params.detach().cpu().double().requires_grad_(True), heads.instance_net.shapes, 96, 128)
    fr = feats.double().requires_grad_(True)
    c_ref, i_ref = heads_from_feature_ref(fr, scale.double(), clip, inst, rp)
    # rows whose ReLU pre-activations sit within the forward error of zero get a zero cotangent (both sides)
    with torch.no_grad():
        cf = fr if rp is None else fr[rp]
        amb_c = _ambiguous(dense_network_ref(torch.cat([cf, scale.double()], -1), *clip)[1])
        amb_i = _ambiguous(dense_network_ref(fr, *inst)[1])
    assert amb_c.float().mean() < 0.7 and amb_i.float().mean() < 0.7   # ~1000 hidden units per row
    vc = torch.randn(nc, 512, generator=gen) * (~amb_c)[:, None]
    vi = torch.randn(P, 128, generator=gen) * (~amb_i)[:, None]
    ((c_ref * vc.double()).sum() + (i_ref * vi.double()).sum()).backward()

    fg = feats.cuda().requires_grad_(True)
    out = heads.get_outputs_from_feature(fg, scale.cuda(), None if rp is None else rp.cuda())
    assert set(out) == {HeadNames.CLIP, HeadNames.INSTANCE}
    c, i = out[HeadNames.CLIP], out[HeadNames.INSTANCE]
    assert c.shape == (nc, 512) and i.shape == (P, 128) and c.dtype == torch.float32
    # unit rows: errors are absolute on a scale of 1/sqrt(width); 1e-4 of the largest entry
    assert (c.cpu().double() - c_ref.detach()).abs().max().item() <= 1e-4 * c_ref.abs().max().item()
    assert (i.cpu().double() - i_ref.detach()).abs().max().item() <= 1e-4 * i_ref.abs().max().item()
    ((c * vc.cuda()).sum() + (i * vi.cuda()).sum()).backward()

    def close(a, b, what):
        err = (a.cpu().double() - b).abs().max().item()
        assert err <= 1e-3 * b.abs().max().item(), (what, err, b.abs().max().item())

    close(fg.grad, fr.grad, "features")
    close(heads.clip_net.params.grad, clip[0].grad, "clip params")
    close(heads.instance_net.params.grad, inst[0].grad, "instance params")
    # the separate entry points agree with the combined one
    with torch.no_grad():
        assert torch.equal(heads.get_instance_outputs_from_feature(fg), i)
        assert torch.equal(heads.get_clip_outputs_from_feature(fg if rp is None else fg[rp.cuda()], scale.cuda()), c)


@pytest.mark.gpu
def test_l2norm_zero_row_has_finite_gradient():
    """y / (|y| + eps) at y = 0 (eps > 0): the output row is 0 and the gradient is g / eps, torch's subgradient of
    the norm at the origin being 0 -- not 0/0."""
    from pogs_amd.heads import _call, _ptr, _stream
    N, eps = 128, 1e-5
    gen = torch.Generator().manual_seed(3)
    y = torch.randn(9, N, generator=gen)
    y[2] = 0
    y[7] = 0
    g = torch.randn(9, N, generator=gen)
    g[7] = 0
    yr = y.double().requires_grad_(True)
    (yr / (yr.norm(dim=-1, keepdim=True) + eps)).backward(g.double())
    assert torch.isfinite(yr.grad).all()
    out, norms, dy = torch.empty(9, N, device="cuda"), torch.empty(9, device="cuda"), torch.empty(9, N, device="cuda")
    yg, gg = y.cuda(), g.cuda()
    _call("pr_l2norm_fwd", 9, N, _ptr(yg), eps, _ptr(out), _ptr(norms), _stream())
    _call("pr_l2norm_bwd", 9, N, _ptr(yg), _ptr(norms), eps, _ptr(gg), _ptr(dy), _stream())
    assert torch.isfinite(out).all() and torch.isfinite(dy).all()
    assert float(out[2].abs().max()) == 0.0 and float(dy[7].abs().max()) == 0.0
    assert (dy.cpu().double() - yr.grad).abs().max().item() <= 1e-5 * yr.grad.abs().max().item()


@pytest.mark.gpu
def test_instance_head_with_an_empty_pixel():
    """A rendered-feature pixel with alpha = 0 is an all-zero 96-vector; the bias-free instance head maps it to an
    exactly zero row (pogs/pogs.py:447 feeds every pixel to it).  Forward and every gradient stay finite and agree with
    the float64 restatement; one such pixel must not poison the weight gradient."""
    from pogs_amd.heads import FieldHeads
    heads = FieldHeads(device="cuda", seed=7)
    gen = torch.Generator().manual_seed(11)
    P = 200
    feats = torch.randn(P, 96, generator=gen)
    feats[5] = 0
    feats[130] = 0
    inst = (heads.instance_net.params.detach().cpu().double().requires_grad_(True), heads.instance_net.shapes, 96, 128)
    fr = feats.double().requires_grad_(True)
    hidden, pre = dense_network_ref(fr, *inst)
    i_ref = hidden / (hidden.norm(dim=-1, keepdim=True) + FieldHeads.INSTANCE_EPS)
    with torch.no_grad():
        amb = _ambiguous(pre)
        amb[5] = amb[130] = False     # zero rows have zero pre-activations everywhere: their gradient is 0 either way
    vi = torch.randn(P, 128, generator=gen) * (~amb)[:, None]
    (i_ref * vi.double()).sum().backward()
    fg = feats.cuda().requires_grad_(True)
    i = heads.get_instance_outputs_from_feature(fg)
    assert torch.isfinite(i).all() and float(i[5].abs().max()) == 0.0
    (i * vi.cuda()).sum().backward()
    assert torch.isfinite(fg.grad).all() and torch.isfinite(heads.instance_net.params.grad).all()
    assert (i.cpu().double() - i_ref.detach()).abs().max().item() <= 1e-4 * i_ref.abs().max().item()
    for got, ref, what in ((fg.grad, fr.grad, "features"), (heads.instance_net.params.grad, inst[0].grad, "params")):
        keep = torch.ones_like(ref, dtype=torch.bool)
        if what == "features":
            keep[5] = keep[130] = False   # through ReLU at exactly 0 the float64 autograd picks 0; so does the kernel
            assert float(got[5].abs().max()) == 0.0 or torch.isfinite(got[5]).all()
        err = ((got.cpu().double() - ref) * keep).abs().max().item()
        assert err <= 1e-3 * ref.abs().max().item(), (what, err)
