"""Camera-sharded multi-GPU step: one process per GPU, one view per rank, Gaussian parameters replicated,
parameter gradients mean-all-reduced over RCCL (xGMI).

This is the reference's only parallelism -- DistributedDataParallel around a model that renders one random
camera per rank (``src/pogs/dependencies/nerfstudio/nerfstudio/scripts/train.py:98,139-145``,
``.../pipelines/base_pipeline.py:279-282``, ``pogs/pogs.py:263``) -- restated without DDP's bucket
machinery: every parameter's ``.grad`` is a view into ONE flat float32 buffer, so the exchange step is a
single large collective (few, large messages suit point-to-point xGMI links better than DDP's 25 MB
buckets) and no gather/scatter copies are needed either side of it.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional

import torch
import torch.distributed as dist
from torch import Tensor


class FlatGrads:
    """Owns one flat gradient buffer; ``params[i].grad`` aliases a slice of it."""

    def __init__(self, params: Iterable[Tensor]):
        self.params: List[Tensor] = [p for p in params if p.requires_grad]
        assert self.params, "no parameter requires grad"
        dev, dt = self.params[0].device, self.params[0].dtype
        total = 0
        self.offsets = []
        for p in self.params:
            assert p.is_leaf and p.device == dev and p.dtype == dt
            # keep every slice 256-byte aligned so kernels may use 16-byte accesses on it
            total = (total + 63) // 64 * 64
            self.offsets.append(total)
            total += p.numel()
        self.flat = torch.zeros(total, dtype=dt, device=dev)
        for p, off in zip(self.params, self.offsets):
            p.grad = self.flat[off:off + p.numel()].view_as(p)

    def zero_(self) -> None:
        """Zero all gradients in one fill; re-attaches the views if something replaced ``.grad``."""
        self.flat.zero_()
        for p, off in zip(self.params, self.offsets):
            g = p.grad
            if g is None or g.data_ptr() != self.flat.data_ptr() + off * self.flat.element_size():
                p.grad = self.flat[off:off + p.numel()].view_as(p)

    def all_reduce_mean(self, group: Optional[dist.ProcessGroup] = None, async_op: bool = False):
        """DDP semantics: every rank ends with the mean over ranks of the per-view gradients."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return None
        if dist.get_backend(group) == "gloo":  # gloo has no AVG
            if self.flat.is_cuda:
                # rehearsal on GPUs without RCCL: stage through the host explicitly (gloo's own handling of large
                # device tensors was seen to hang when two ranks share one GPU)
                host = self.flat.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
                self.flat.copy_(host.div_(dist.get_world_size(group)))
                return None
            work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
            if async_op:
                return work
            self.flat.div_(dist.get_world_size(group))
            return None
        return dist.all_reduce(self.flat, op=dist.ReduceOp.AVG, group=group, async_op=async_op)

    def nbytes(self) -> int:
        return self.flat.numel() * self.flat.element_size()


class GradExchange:
    """Mean all-reduce of the gradients WHERE AUTOGRAD LEFT THEM: no flat copy, no zero-fill + accumulate.

    With ``FlatGrads`` every backward pass accumulates into pre-zeroed views (one fill and one read-modify-write
    of every gradient per step); here ``.grad`` may be ``None`` before the backward pass, so autograd simply
    adopts the tensors the kernels produced.  Large gradients (SH coefficients, features) are then reduced
    in place, one collective each; the small ones (means, quaternions, scales, opacities) are packed into
    one persistent buffer so that they cost one collective, not four latencies."""

    def __init__(self, params: Iterable[Tensor], big_bytes: int = 32 << 20):
        self.params: List[Tensor] = [p for p in params if p.requires_grad]
        assert self.params, "no parameter requires grad"
        self.big = [p for p in self.params if p.numel() * p.element_size() >= big_bytes]
        self.small = [p for p in self.params if p.numel() * p.element_size() < big_bytes]
        n_small = sum(p.numel() for p in self.small)
        ref = self.params[0]
        self.pack = torch.empty(n_small, dtype=ref.dtype, device=ref.device) if n_small else None

    def nbytes(self) -> int:
        return sum(p.numel() * p.element_size() for p in self.params)

    @staticmethod
    def _reduce(t: Tensor, group, world: int):
        backend = dist.get_backend(group)
        if backend == "gloo":  # no AVG; device tensors are staged through the host (rehearsals only)
            host = t.cpu() if t.is_cuda else t
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            host.div_(world)
            if t.is_cuda:
                t.copy_(host)
            return None
        return dist.all_reduce(t, op=dist.ReduceOp.AVG, group=group, async_op=True)

    def all_reduce_mean(self, group: Optional[dist.ProcessGroup] = None) -> None:
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return
        world = dist.get_world_size(group)
        works = []
        for p in self.big:
            if p.grad is None:          # this rank's view did not reach the parameter: it contributes zeros
                p.grad = torch.zeros_like(p)
            elif not p.grad.is_contiguous():
                p.grad = p.grad.contiguous()
            works.append(self._reduce(p.grad, group, world))
        if self.pack is not None:
            at = 0
            for p in self.small:
                seg = self.pack[at:at + p.numel()]
                if p.grad is None:
                    seg.zero_()
                else:
                    seg.copy_(p.grad.reshape(-1))
                at += p.numel()
            works.append(self._reduce(self.pack, group, world))
        for w in works:
            if w is not None:
                w.wait()
        if self.pack is not None:
            at = 0
            for p in self.small:
                seg = self.pack[at:at + p.numel()].view_as(p)
                if p.grad is None or not p.grad.is_contiguous():
                    p.grad = seg.clone()
                else:
                    p.grad.copy_(seg)
                at += p.numel()


def view_for_rank(n_views: int, rank: int, world: int) -> List[int]:
    """Views of a batch handled by this rank: view v -> rank v mod world."""
    return [v for v in range(n_views) if v % world == rank]


def sharded_step(render_view, n_views: int, grads: FlatGrads, rank: int, world: int,
                 group: Optional[dist.ProcessGroup] = None) -> Dict[str, float]:
    """One multi-view step.  ``render_view(v)`` must run forward AND backward of view ``v`` (accumulating
    into the parameters' ``.grad``).  Gradients of the views a rank owns are summed locally, divided by
    the number of views so that the all-reduced SUM over ranks equals the mean over the batch."""
    grads.zero_()
    mine = view_for_rank(n_views, rank, world)
    for v in mine:
        render_view(v)
    if world > 1:
        # mean over the whole batch = (sum over ranks of local sums) / n_views
        grads.flat.mul_(float(world) / float(n_views))
        grads.all_reduce_mean(group)
    elif n_views > 1:
        grads.flat.div_(float(n_views))
    return {"views": float(len(mine))}


class _AllReduceExchange:
    """``GradExchange`` behind the step interface ``bench.py`` drives: nothing before the backward pass, the mean
    all-reduce of every gradient after it."""

    def __init__(self, params: Dict[str, Tensor], group=None):
        self.ex = GradExchange(params.values())
        self.group = group

    def begin_step(self) -> None:
        pass

    def finish_step(self) -> None:
        self.ex.all_reduce_mean(self.group)

    def describe(self) -> str:
        return "in-place gradient mean all-reduce after the backward pass (%.0f MB: SH, features, packed rest)" % (
            self.ex.nbytes() / 1e6)


def make_exchange(kind: str, params: Dict[str, Tensor], group=None):
    if kind == "allreduce":
        return _AllReduceExchange(params, group)
    raise ValueError(kind)
