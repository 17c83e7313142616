"""CLIP and instance heads on rendered (or per-Gaussian) 96-channel hash features, on the matrix cores.

Stands in for the part of ``GaussianField`` that runs AFTER the rasterizer (``pogs/fields/gaussian_field.py:105-127``
the two networks, ``:176-204`` the three ``*_from_feature`` methods; called at ``pogs/pogs.py:447,685,718,828,917,952``
and ``pogs/tracking/optim.py:444``).  The hash-grid encodings themselves (``:94-100,146-174``) stay with tiny-cuda-nn:
they sit upstream of the rasterizer and only their width (96) matters here.

``DenseNetwork`` mirrors ``tcnn.Network(otype="CutlassMLP", activation="ReLU", output_activation="None")``: no biases,
the input padded to a multiple of 16 columns with ones (tiny-cuda-nn's padding value), one flat ``params`` tensor that
holds the layer matrices back to back, each ``[n_out, n_in]`` row-major.  tiny-cuda-nn is not part of the reference
tree, so this layout is taken from its published description and nothing in the reference pins it.  Arithmetic:
float32 storage, split-bf16 products (2^-16 per product); the reference runs these layers in float16.

Forward and backward are HIP kernels only (``pr_dense_*``, ``pr_l2norm_*``); there is no CPU path and no torch GEMM.
"""
from __future__ import annotations

import ctypes
import math
from enum import Enum
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib


class HeadNames(Enum):
    """Same members and values as ``GaussianFieldHeadNames`` (``pogs/field_components/gaussian_fieldheadnames.py``)."""
    HASHGRID = "hashgrid"
    CLIP = "clip"
    INSTANCE = "instance"


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _call(name: str, *args):
    _lib.check(getattr(_lib.load(), name)(*args), name)


def _round_up(n: int, m: int) -> int:
    return (n + m - 1) // m * m


def layer_shapes(n_input_dims: int, n_output_dims: int, n_neurons: int, n_hidden_layers: int) -> List[Tuple[int, int]]:
    """``[(n_out, n_in), ...]`` of the matrices in ``params``: input and output widths padded to multiples of 16."""
    k0, out = _round_up(n_input_dims, 16), _round_up(n_output_dims, 16)
    shapes = [(n_neurons, k0)] + [(n_neurons, n_neurons)] * (n_hidden_layers - 1) + [(out, n_neurons)]
    return shapes


def _packed(w: torch.Tensor, transpose: bool) -> torch.Tensor:
    n, k = w.shape
    m, kk = (k, n) if transpose else (n, k)
    buf = torch.empty(_lib.load().pr_dense_packed_bytes(m, kk), dtype=torch.uint8, device=w.device)
    _call("pr_dense_pack", m, kk, _ptr(w), 1 if transpose else k, k if transpose else 1, _ptr(buf), _stream())
    return buf


class _DenseStack(torch.autograd.Function):
    """x [P, K0] -> ReLU(x W0^T) ... W_{L-1}^T, then optionally y / (|y| + eps)."""

    @staticmethod
    def forward(ctx, x, params, shapes, norm_eps):
        if not x.is_cuda:
            raise RuntimeError("pogs_amd.heads needs tensors on a HIP device: there is no CPU path")
        x = x.contiguous().float()
        params = params.contiguous().float()
        P = x.shape[0]
        ws, off = [], 0
        for n, k in shapes:
            ws.append(params[off:off + n * k].view(n, k))
            off += n * k
        assert off == params.numel() and x.shape[1] == shapes[0][1]
        acts = []
        with torch.cuda.device(x.device):
            h = x
            for i, w in enumerate(ws):
                n, k = w.shape
                y = torch.empty(P, n, dtype=torch.float32, device=x.device)
                pk = _packed(w, False)   # held until the launch below has been issued
                _call("pr_dense_fwd", P, n, k, _ptr(h), k, _ptr(pk), int(i < len(ws) - 1), None, 0, _ptr(y), n, _stream())
                acts.append(y)
                h = y
            norms = None
            out = h
            if norm_eps is not None:
                out = torch.empty_like(h)
                norms = torch.empty(P, dtype=torch.float32, device=x.device)
                _call("pr_l2norm_fwd", P, h.shape[1], _ptr(h), float(norm_eps), _ptr(out), _ptr(norms), _stream())
        ctx.shapes, ctx.norm_eps = shapes, norm_eps
        ctx.save_for_backward(x, params, norms, *acts)
        return out

    @staticmethod
    def backward(ctx, v_out):
        x, params, norms, *acts = ctx.saved_tensors
        shapes = ctx.shapes
        P = x.shape[0]
        g = v_out.contiguous().float()
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        v_params = torch.zeros_like(params) if need_w else None
        offs, off = [], 0
        for n, k in shapes:
            offs.append(off)
            off += n * k
        with torch.cuda.device(x.device):
            if ctx.norm_eps is not None:
                gy = torch.empty_like(g)
                _call("pr_l2norm_bwd", P, g.shape[1], _ptr(acts[-1]), _ptr(norms), float(ctx.norm_eps), _ptr(g), _ptr(gy),
                      _stream())
                g = gy
            for i in range(len(shapes) - 1, -1, -1):
                n, k = shapes[i]
                h = acts[i - 1] if i > 0 else x
                if need_w:
                    dw = v_params[offs[i]:offs[i] + n * k]
                    _call("pr_dense_wgrad", P, n, k, _ptr(g), n, _ptr(h), k, _ptr(dw), k, _stream())
                if i == 0 and not need_x:
                    break
                w = params[offs[i]:offs[i] + n * k].view(n, k)
                gp = torch.empty(P, k, dtype=torch.float32, device=x.device)
                # through the ReLU that produced h: zero where h <= 0 (the first layer's input has no ReLU)
                pk = _packed(w, True)
                _call("pr_dense_fwd", P, k, n, _ptr(g), n, _ptr(pk), 0, _ptr(h) if i > 0 else None, k, _ptr(gp), k,
                      _stream())
                g = gp
        return (g if need_x else None), v_params, None, None


class DenseNetwork(torch.nn.Module):
    """Bias-free ReLU MLP with tiny-cuda-nn's ``Network`` constructor arguments and ``params`` state-dict key."""

    def __init__(self, n_input_dims: int, n_output_dims: int, n_neurons: int = 256, n_hidden_layers: int = 3,
                 device=None, seed: Optional[int] = None):
        super().__init__()
        self.n_input_dims, self.n_output_dims = n_input_dims, n_output_dims
        self.shapes = tuple(layer_shapes(n_input_dims, n_output_dims, n_neurons, n_hidden_layers))
        gen = torch.Generator().manual_seed(seed) if seed is not None else None
        chunks = []
        for n, k in self.shapes:   # Xavier-uniform, tiny-cuda-nn's default initialisation
            bound = math.sqrt(6.0 / (n + k))
            chunks.append(((torch.rand(n * k, generator=gen) * 2 - 1) * bound))
        self.params = torch.nn.Parameter(torch.cat(chunks).to(device))

    def matrices(self) -> List[torch.Tensor]:
        out, off = [], 0
        for n, k in self.shapes:
            out.append(self.params[off:off + n * k].view(n, k))
            off += n * k
        return out

    def padded_input(self, x: torch.Tensor) -> torch.Tensor:
        k0 = self.shapes[0][1]
        x = x.reshape(-1, self.n_input_dims).float()
        if k0 == self.n_input_dims:
            return x
        return torch.cat([x, x.new_ones(x.shape[0], k0 - self.n_input_dims)], dim=-1)

    def forward(self, x: torch.Tensor, norm_eps: Optional[float] = None) -> torch.Tensor:
        """``x [..., n_input_dims]`` -> ``[P, n_output_dims]``; ``norm_eps`` not None: rows divided by ``|row| + eps``."""
        out = _DenseStack.apply(self.padded_input(x), self.params, self.shapes, norm_eps)
        return out if out.shape[1] == self.n_output_dims else out[:, :self.n_output_dims]


class FieldHeads(torch.nn.Module):
    """``clip_net`` and ``instance_net`` of ``GaussianField`` with its ``*_from_feature`` methods."""

    INSTANCE_EPS = 1e-5   # pogs/fields/gaussian_field.py:188,195

    def __init__(self, feature_dims: int = 96, clip_n_dims: int = 512, instance_n_dims: int = 128, device=None,
                 seed: Optional[int] = None):
        super().__init__()
        self.clip_net = DenseNetwork(feature_dims + 1, clip_n_dims, 256, 3, device=device, seed=seed)
        self.instance_net = DenseNetwork(feature_dims, instance_n_dims, 256, 4, device=device,
                                         seed=None if seed is None else seed + 1)

    def get_clip_outputs_from_feature(self, features: torch.Tensor, clip_scale: torch.Tensor) -> torch.Tensor:
        x = torch.cat([features, clip_scale.view(-1, 1).to(features.dtype)], dim=-1)   # gaussian_field.py:184,202
        return self.clip_net(x, norm_eps=0.0)

    def get_instance_outputs_from_feature(self, features: torch.Tensor) -> torch.Tensor:
        return self.instance_net(features, norm_eps=self.INSTANCE_EPS)

    def get_outputs_from_feature(self, features: torch.Tensor, clip_scale: torch.Tensor,
                                 random_pixels: Optional[torch.Tensor] = None) -> Dict[HeadNames, torch.Tensor]:
        clip_features = features if random_pixels is None else features[random_pixels]   # gaussian_field.py:180-183
        return {HeadNames.CLIP: self.get_clip_outputs_from_feature(clip_features, clip_scale),
                HeadNames.INSTANCE: self.get_instance_outputs_from_feature(features)}
