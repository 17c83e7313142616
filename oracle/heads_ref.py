"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the CLIP / instance heads that follow the rasterizer.

Follows pogs/fields/gaussian_field.py:105-127 (two bias-free ReLU networks: 97 -> 256 x 3 -> 512 and
96 -> 256 x 4 -> 128) and :176-204 (concatenate the clip scale, run the networks, divide by the row norm -- plus
1e-5 for the instance head), with torch CPU operators in float64.  The networks themselves are tiny-cuda-nn
CutlassMLPs, a third-party dependency that is not in the reference tree (tinycudann, unpinned `pip install` in the
reference's README): their published behaviour is restated here -- no biases, input padded with ones to a multiple of
16 columns, matrices [n_out, n_in] back to back in one `params` vector.  Nothing the reference holds pins that
layout: PARITY UNPINNED for the heads.

Only tests/, __graft_entry__.smoke() and bench scripts' CPU legs may import this module.
"""
import torch


def dense_network_ref(x, params, shapes, n_input_dims, n_output_dims, norm_eps=None):
    """x [P, n_input_dims]; params flat; shapes [(n_out, n_in)]; returns ([P, n_output_dims], hidden pre-activations)."""
    k0 = shapes[0][1]
    h = x
    if k0 > n_input_dims:
        h = torch.cat([x, x.new_ones(x.shape[0], k0 - n_input_dims)], dim=-1)
    off, pre = 0, []
    for i, (n, k) in enumerate(shapes):
        w = params[off:off + n * k].view(n, k)
        off += n * k
        h = h @ w.t()
        if i < len(shapes) - 1:
            pre.append(h)
            h = torch.relu(h)
    if norm_eps is not None:
        h = h / (h.norm(dim=-1, keepdim=True) + norm_eps)   # gaussian_field.py:186,190
    return h[:, :n_output_dims], pre


def heads_from_feature_ref(features, clip_scale, clip, instance, random_pixels=None):
    """gaussian_field.py:176-192; clip / instance: (params, shapes, n_in, n_out)."""
    cf = features if random_pixels is None else features[random_pixels]
    c, _ = dense_network_ref(torch.cat([cf, clip_scale.view(-1, 1)], dim=-1), *clip, norm_eps=0.0)
    i, _ = dense_network_ref(features, *instance, norm_eps=1e-5)
    return c, i
