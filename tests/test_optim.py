"""Fused Adam (pr_adam_step through pogs_amd.optim.Optimizers) against torch.optim.Adam driven as the reference
trainer drives it.  torch.optim.Adam is the reference's own optimiser, so this row's parity is pinned.
Tolerance: 2e-6 relative to the largest entry per tensor after 25 iterations (float32, fused multiply-adds)."""
import pytest
import torch

from oracle import adam_ref

CONFIG = {  # pogs/pogs_config.py:38-85 (the groups on the Gaussian path) and :27
    "means": {"lr": 1.6e-4, "eps": 1e-15, "scheduler": dict(lr_final=1.6e-6, max_steps=30000)},
    "features_dc": {"lr": 0.0025, "eps": 1e-15, "scheduler": None},
    "features_rest": {"lr": 0.0025 / 20, "eps": 1e-15, "scheduler": None},
    "opacities": {"lr": 0.05, "eps": 1e-15, "scheduler": None},
    "scales": {"lr": 0.005, "eps": 1e-15, "scheduler": dict(lr_final=1e-3, max_steps=30000)},
    "quats": {"lr": 0.001, "eps": 1e-15, "scheduler": None},
    "dino_feats": {"lr": 1e-2, "eps": 1e-15, "scheduler": dict(lr_final=1e-3, max_steps=6000, warmup_steps=5)},
}
ACCUM = {"features_dc": 10, "features_rest": 10, "dino_feats": 3}
SHAPES = {"means": (1001, 3), "features_dc": (1001, 3), "features_rest": (1001, 15, 3), "opacities": (1001, 1),
          "scales": (1001, 3), "quats": (1001, 4), "dino_feats": (1001, 64)}


def _grads(step):
    g = torch.Generator().manual_seed(1000 + step)
    return {k: torch.randn(s, generator=g) * (0.1 + 0.01 * step) for k, s in SHAPES.items()}


def test_schedule_matches_the_reference_lambda():
    from pogs_amd.optim import ExponentialDecay

    for cfg in (dict(lr_final=1.6e-6, max_steps=30000), dict(lr_final=1e-3, max_steps=6000, warmup_steps=50),
                dict(lr_final=None, max_steps=100, warmup_steps=10, ramp="linear")):
        ref = adam_ref.exponential_decay_lambda(0.01, **cfg)
        mine = ExponentialDecay(**cfg)
        for step in (0, 1, 7, 49, 50, 51, 99, 100, 5999, 6000, 40000):
            assert mine.lr(0.01, step) == pytest.approx(ref(step) * 0.01, rel=1e-12)


def test_accumulation_periods_follow_the_trainer():
    from pogs_amd.optim import Optimizers

    names = list(CONFIG)
    assert Optimizers.groups_to_zero(0, names, ACCUM) == names
    assert Optimizers.groups_to_zero(3, names, ACCUM) == [n for n in names if n not in ("features_dc", "features_rest")]
    assert "features_dc" in Optimizers.groups_to_step(9, names, ACCUM)
    assert "features_dc" not in Optimizers.groups_to_step(8, names, ACCUM)
    assert "dino_feats" in Optimizers.groups_to_step(2, names, ACCUM)


def test_entry_point_validates_without_a_gpu():
    import ctypes

    from pogs_amd import _lib

    lib = _lib.load()
    assert lib.pr_adam_max_groups() >= 8
    assert lib.pr_adam_step(None, 0, None) == 0
    assert lib.pr_adam_step(None, 2, None) == -1
    assert lib.pr_adam_step(None, 1000, None) == -1
    bad = (_lib.AdamGroup * 1)(_lib.AdamGroup(16, 16, 16, 16, 8, 0, 1e-3, 0.9, 0.999, 1e-15))   # step 0
    assert lib.pr_adam_step(ctypes.cast(bad, ctypes.c_void_p), 1, None) == -1
    odd = (_lib.AdamGroup * 1)(_lib.AdamGroup(20, 16, 16, 16, 8, 1, 1e-3, 0.9, 0.999, 1e-15))   # misaligned
    assert lib.pr_adam_step(ctypes.cast(odd, ctypes.c_void_p), 1, None) == -1
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        from pogs_amd.optim import Optimizers
        Optimizers({"a": {"lr": 1e-3}}, {"a": [torch.zeros(4, requires_grad=True)]})


@pytest.mark.gpu
def test_fused_adam_matches_torch_adam_over_a_training_schedule():
    from pogs_amd.optim import ExponentialDecay, Optimizers

    init = {k: torch.randn(s, generator=torch.Generator().manual_seed(7)) for k, s in SHAPES.items()}
    ref_p = {k: [v.clone().requires_grad_(True)] for k, v in init.items()}
    ref = adam_ref.Optimizers(CONFIG, ref_p)
    dev_p = {k: [v.clone().cuda().requires_grad_(True)] for k, v in init.items()}
    cfg = {k: dict(v, scheduler=None if v["scheduler"] is None else ExponentialDecay(**v["scheduler"]))
           for k, v in CONFIG.items()}
    opt = Optimizers(cfg, dev_p)
    names = list(CONFIG)
    for step in range(25):
        g = _grads(step)
        skip = "quats" if step % 4 == 1 else None       # a group without a gradient is not stepped
        ref.train_iteration(step, ACCUM, lambda n, i: None if n == skip else g[n])
        opt.zero_grad_some(Optimizers.groups_to_zero(step, names, ACCUM))
        for n in names:
            if n == skip:
                continue
            p = dev_p[n][0]
            p.grad = g[n].cuda() if p.grad is None else p.grad + g[n].cuda()
        opt.optimizer_step_some(Optimizers.groups_to_step(step, names, ACCUM))
        opt.scheduler_step_all(step)
        if skip:
            dev_p[skip][0].grad = None
            ref_p[skip][0].grad = None
    for n in names:
        want, got = ref_p[n][0].detach(), dev_p[n][0].detach().cpu()
        assert (got - want).abs().max().item() <= 2e-6 * want.abs().max().item(), n
        assert opt.lr(n) == pytest.approx(ref.optimizers[n].param_groups[0]["lr"], rel=1e-9), n
    # state interchanges with torch.optim.Adam
    sd = opt.state_dict()
    for n in names:
        t = torch.optim.Adam([torch.zeros(SHAPES[n], device="cuda")], lr=1.0)
        t.load_state_dict(sd[n])
        st, rs = t.state_dict()["state"][0], ref.optimizers[n].state_dict()["state"][0]
        assert float(st["step"]) == float(rs["step"])
        assert (st["exp_avg"].cpu() - rs["exp_avg"]).abs().max() <= 2e-6 * rs["exp_avg"].abs().max()
        assert (st["exp_avg_sq"].cpu() - rs["exp_avg_sq"]).abs().max() <= 2e-6 * rs["exp_avg_sq"].abs().max()
    opt2 = Optimizers(cfg, dev_p)
    opt2.load_optimizers(sd)
    assert opt2._state["means"][0]["step"] == opt._state["means"][0]["step"]


@pytest.mark.gpu
def test_ragged_sizes_and_many_tensors():
    """Sizes that are not multiples of the vector width or the block, and more tensors than one launch takes."""
    from pogs_amd.optim import Optimizers

    sizes = [1, 3, 4, 5, 4095, 4096, 4097, 16384 + 7] + [33] * 20
    g = torch.Generator().manual_seed(3)
    init = [torch.randn(s, generator=g) for s in sizes]
    grads = [torch.randn(s, generator=g) for s in sizes]
    ref_p = [v.clone().requires_grad_(True) for v in init]
    ref = torch.optim.Adam(ref_p, lr=0.01, eps=1e-15)
    dev_p = [v.clone().cuda().requires_grad_(True) for v in init]
    opt = Optimizers({"all": {"lr": 0.01}}, {"all": dev_p})
    for _ in range(3):
        for p, q, gr in zip(ref_p, dev_p, grads):
            p.grad, q.grad = gr.clone(), gr.cuda()
        ref.step()
        opt.optimizer_step_all()
    for p, q in zip(ref_p, dev_p):
        assert (q.detach().cpu() - p.detach()).abs().max().item() <= 2e-6 * max(p.detach().abs().max().item(), 1e-3)
