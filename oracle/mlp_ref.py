"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the channel MLP around the rendered feature image.

Follows pogs/pogs.py:153-161 (the module: four bias-free Linear layers, 64 wide, ReLU between) and
pogs/pogs.py:516-522 (alpha normalisation before it, eval-mode zeroing below alpha 0.8 after it), with the
same torch operators the reference calls, in float64 or float32 on the CPU.  Since the reference's own
implementation IS these torch operators, this oracle is pinned by construction.

Only tests/, __graft_entry__.smoke() and bench scripts' CPU legs may import this module.
"""
import torch


def feature_mlp_ref(feats, alphas, weights, zero_below=0.0):
    """feats [..., 64], alphas [..., 1] or None, weights: four [64, 64] Linear.weight matrices."""
    shape = feats.shape
    x = feats
    if alphas is not None:
        clamped = torch.clamp(alphas, min=1e-6)  # pogs.py:516
        x = torch.where(alphas > 0, x / clamped.detach(), torch.zeros(shape[-1], dtype=x.dtype))  # pogs.py:518
    h = x.reshape(-1, shape[-1])  # pogs.py:519
    for i, w in enumerate(weights):  # pogs.py:153-161, 520
        h = torch.nn.functional.linear(h, w)
        if i < len(weights) - 1:
            h = torch.relu(h)
    out = h.view(*shape[:-1], -1)
    if zero_below > 0 and alphas is not None:  # pogs.py:521-522
        out = out.clone()
        out[alphas.squeeze(-1) < zero_below] = 0
    return out
