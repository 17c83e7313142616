"""Rigid-group transform: the CPU oracle against closed-form answers (no GPU), and the HIP kernels
(pr_rigid_fwd / pr_rigid_bwd through the C ABI) against the oracle (marked gpu).

The oracle restates the Warp kernel of pogs/tracking/utils.py:53-93; Warp is not installed and the reference
holds no fixture for it: parity for this row is unpinned beyond the closed-form cases below.
"""
import math

import pytest
import torch

from oracle import rigid_ref as O
from tests._util import assert_grad_close


def _unit(q):
    return q / q.norm(dim=-1, keepdim=True)


def _case(n=500, g=5, seed=0, dtype=torch.float64, unit_delta=True, label_range=None):
    gen = torch.Generator().manual_seed(seed)

    def rn(*shape):
        return torch.randn(*shape, generator=gen, dtype=torch.float64)

    p2w = torch.cat([rn(g, 3), _unit(rn(g, 4))], dim=1)
    dq = rn(g, 4) * 0.2 + torch.tensor([1.0, 0, 0, 0], dtype=torch.float64)
    delta = torch.cat([rn(g, 3) * 0.1, _unit(dq) if unit_delta else dq], dim=1)
    labels = torch.randint(0, label_range or g, (n,), generator=gen, dtype=torch.int32)
    means = rn(n, 3)
    quats = _unit(rn(n, 4))
    return [t.to(dtype) if t.is_floating_point() else t for t in (p2w, delta, labels, means, quats)]


# ------------------------------------------------------------------------------------------------
# oracle, CPU only
# ------------------------------------------------------------------------------------------------
def test_identity_delta_leaves_the_model_unchanged():
    p2w, delta, labels, means, quats = _case()
    delta[:] = torch.tensor([0, 0, 0, 1.0, 0, 0, 0])
    m, q = O.apply_to_model(p2w, delta, labels, means, quats)
    assert torch.allclose(m, means, atol=1e-12) and torch.allclose(q, quats, atol=1e-12)


def test_translation_delta_moves_along_the_part_axes():
    # part frame rotated 90 degrees about z: a delta of +x in the part frame is +y in the world
    h = math.sqrt(0.5)
    p2w = torch.tensor([[1.0, 2, 3, h, 0, 0, h]], dtype=torch.float64)
    delta = torch.tensor([[0.5, 0, 0, 1, 0, 0, 0]], dtype=torch.float64)
    means = torch.tensor([[0.0, 0, 0], [4, 5, 6]], dtype=torch.float64)
    quats = torch.tensor([[1.0, 0, 0, 0], [0, 1, 0, 0]], dtype=torch.float64)
    m, q = O.apply_to_model(p2w, delta, torch.zeros(2, dtype=torch.int32), means, quats)
    assert torch.allclose(m, means + torch.tensor([0, 0.5, 0.0]), atol=1e-12)
    assert torch.allclose(q, quats, atol=1e-12)


def test_rotation_delta_turns_about_the_part_origin():
    # part at (1,0,0), axes aligned with the world; delta = 90 degrees about z
    h = math.sqrt(0.5)
    p2w = torch.tensor([[1.0, 0, 0, 1, 0, 0, 0]], dtype=torch.float64)
    delta = torch.tensor([[0.0, 0, 0, h, 0, 0, h]], dtype=torch.float64)
    means = torch.tensor([[2.0, 0, 0], [1, 0, 0]], dtype=torch.float64)
    quats = torch.tensor([[1.0, 0, 0, 0], [1, 0, 0, 0]], dtype=torch.float64)
    m, q = O.apply_to_model(p2w, delta, torch.zeros(2, dtype=torch.int32), means, quats)
    assert torch.allclose(m, torch.tensor([[1.0, 1, 0], [1, 0, 0]], dtype=torch.float64), atol=1e-12)
    assert torch.allclose(q, torch.tensor([[h, 0, 0, h]] * 2, dtype=torch.float64), atol=1e-12)


def test_quaternion_and_matrix_formulations_agree():
    p2w, delta, labels, means, quats = _case(seed=3)
    m, q = O.apply_to_model(p2w, delta, labels, means, quats)
    m2, r2 = O.apply_to_model_matrices(p2w, delta, labels, means, quats)
    assert torch.allclose(m, m2, atol=1e-10)
    assert torch.allclose(O._rotmat(q), r2, atol=1e-10)


def test_delta_then_inverse_delta_round_trips():
    p2w, delta, labels, means, quats = _case(seed=4)
    m, q = O.apply_to_model(p2w, delta, labels, means, quats)
    ip, iq = O.transform_inverse((delta[:, :3], torch.cat([delta[:, 4:7], delta[:, 3:4]], dim=1)))
    inv = torch.cat([ip, iq[:, 3:4], iq[:, :3]], dim=1)
    m0, q0 = O.apply_to_model(p2w, inv, labels, m, q)
    assert torch.allclose(m0, means, atol=1e-10) and torch.allclose(q0, quats, atol=1e-10)


def test_oracle_gradient_matches_finite_differences():
    p2w, delta, labels, means, quats = _case(n=40, g=3, seed=5, unit_delta=False)
    wm, wq = torch.randn(40, 3, dtype=torch.float64), torch.randn(40, 4, dtype=torch.float64)

    def loss(d):
        m, q = O.apply_to_model(p2w, d, labels, means, quats)
        return (m * wm).sum() + (q * wq).sum()

    assert torch.autograd.gradcheck(loss, (delta.requires_grad_(True),), eps=1e-6, atol=1e-6)


# ------------------------------------------------------------------------------------------------
# HIP kernels against the oracle
# ------------------------------------------------------------------------------------------------
def _gpu(ts):
    return [t.float().cuda() if t.is_floating_point() else t.cuda() for t in ts]


@pytest.mark.gpu
@pytest.mark.parametrize("n,g,unit", [(1, 1, True), (777, 5, True), (5000, 40, False), (3000, 700, True)])
def test_forward_parity(n, g, unit):
    from pogs_amd import tracking

    case = _case(n=n, g=g, seed=n, unit_delta=unit)
    m, q = tracking.apply_to_model(*_gpu(case))
    rm, rq = O.apply_to_model(*[t.float().double() if t.is_floating_point() else t for t in case])
    assert (m.double().cpu() - rm).abs().max().item() <= 1e-5
    assert (q.double().cpu() - rq).abs().max().item() <= 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("n,g,quats_grad", [(777, 5, False), (777, 5, True), (20000, 40, True), (3000, 700, True),
                                            (3000, 700, False)])
def test_backward_parity(n, g, quats_grad):
    from pogs_amd import tracking

    case = _case(n=n, g=g, seed=n + 1, unit_delta=False)
    case = [t.float().double() if t.is_floating_point() else t for t in case]  # float32-representable inputs
    gen = torch.Generator().manual_seed(9)
    wm = torch.randn(n, 3, generator=gen, dtype=torch.float64).float().double()
    wq = torch.randn(n, 4, generator=gen, dtype=torch.float64).float().double()

    ref_delta = case[1].clone().requires_grad_(True)
    rm, rq = O.apply_to_model(case[0], ref_delta, *case[2:])
    ((rm * wm).sum() + ((rq * wq).sum() if quats_grad else 0)).backward()

    dev = _gpu(case)
    dev[1].requires_grad_(True)
    m, q = tracking.apply_to_model(*dev, quats_grad=quats_grad)
    assert q.requires_grad == quats_grad
    loss = (m * wm.float().cuda()).sum()
    if quats_grad:
        loss = loss + (q * wq.float().cuda()).sum()
    loss.backward()
    # tolerance: 1e-3 relative per element plus 1e-4 of the largest entry (float32 sums over n/g terms)
    assert_grad_close(dev[1].grad, ref_delta.grad, "v_p_deltas")


@pytest.mark.gpu
def test_labels_outside_the_groups_pass_through_without_gradient():
    from pogs_amd import tracking

    case = _case(n=600, g=4, seed=11, label_range=4)
    case[2][::3] = -1
    case[2][1::7] = 9
    dev = _gpu(case)
    dev[1].requires_grad_(True)
    m, q = tracking.apply_to_model(*dev)
    outside = ((case[2] < 0) | (case[2] >= 4))
    assert torch.equal(m.cpu()[outside], case[3].float()[outside])
    assert torch.equal(q.cpu()[outside], case[4].float()[outside])
    inside = ~outside
    sub = [case[0], case[1].clone().requires_grad_(True), case[2][inside], case[3][inside], case[4][inside]]
    rm, _ = O.apply_to_model(*sub)
    assert (m.double().cpu()[inside] - rm).abs().max().item() <= 1e-5
    w = torch.randn(600, 3, dtype=torch.float64)
    (m * w.float().cuda()).sum().backward()
    (rm * w[inside].float().double()).sum().backward()
    assert_grad_close(dev[1].grad, sub[1].grad, "v_p_deltas")


@pytest.mark.gpu
def test_empty_model_and_full_size_round_trip():
    from pogs_amd import tracking

    e = tracking.apply_to_model(torch.zeros(2, 7).cuda(), torch.zeros(2, 7).cuda(),
                                torch.zeros(0, dtype=torch.int32).cuda(), torch.zeros(0, 3).cuda(),
                                torch.zeros(0, 4).cuda())
    assert e[0].shape == (0, 3) and e[1].shape == (0, 4)
    # 1M Gaussians: apply a delta, then its inverse
    case = _case(n=1_000_000, g=12, seed=2, dtype=torch.float32)
    dev = _gpu(case)
    m, q = tracking.apply_to_model(*dev)
    d = case[1].double()
    ip, iq = O.transform_inverse((d[:, :3], torch.cat([d[:, 4:7], d[:, 3:4]], dim=1)))
    inv = torch.cat([ip, iq[:, 3:4], iq[:, :3]], dim=1).float().cuda()
    m0, q0 = tracking.apply_to_model(dev[0], inv, dev[2], m, q)
    assert (m0 - dev[3]).abs().max().item() <= 2e-5
    assert (q0 - dev[4]).abs().max().item() <= 2e-5


def test_rigid_entry_points_validate_arguments_without_a_gpu():
    from pogs_amd import _lib

    lib = _lib.load()
    assert lib.pr_rigid_fwd(-1, 1, None, None, None, None, None, None, None, None) == -1
    assert lib.pr_rigid_fwd(0, 0, None, None, None, None, None, None, None, None) == 0   # nothing to do
    assert lib.pr_rigid_fwd(4, 1, None, None, None, None, None, None, None, None) == -1  # null pointers
    assert lib.pr_rigid_bwd(4, 1, None, None, None, None, None, None, None, None, None) == -1
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        from pogs_amd import tracking
        tracking.apply_to_model(torch.zeros(1, 7), torch.zeros(1, 7), torch.zeros(3, dtype=torch.int32),
                                torch.zeros(3, 3), torch.zeros(3, 4))
