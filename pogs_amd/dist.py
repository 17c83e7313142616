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
        if not (dist.is_available() and dist.is_initialized()) or (
                dist.get_world_size(group) == 1 and not _single_rank_collectives()):
            return None
        if dist.get_backend(group) == "gloo":  # gloo has no AVG
            if self.flat.is_cuda:
                # rehearsal on GPUs without RCCL: stage through the host explicitly (gloo's own handling of large
                # device tensors was seen to hang when two ranks share one GPU)
                host = self.flat.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
                self.flat.copy_(host.div_(dist.get_world_size(group)))
                return None
            # divide first: the handle returned for async_op then completes to the MEAN like the RCCL path
            self.flat.div_(dist.get_world_size(group))
            return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op) if async_op \
                else dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
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
        if not (dist.is_available() and dist.is_initialized()) or (
                dist.get_world_size(group) == 1 and not _single_rank_collectives()):
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



# ---------------------------------------------------------------------------------------------------------------
# Sharded exchange: reduce-scatter -> optimiser step on the local shard -> all-gather of the parameters, per bucket,
# overlapped with the rest of the backward pass and with the start of the next forward pass
# ---------------------------------------------------------------------------------------------------------------
class _Ready:
    """When a parameter's exchange of the previous step is complete: a device event on HIP, outstanding collective
    handles on the CPU (gloo).  ``wait()`` orders the CURRENT stream behind it (no host wait on HIP)."""

    def __init__(self, event=None, works=()):
        self.event, self.works = event, list(works)

    def wait(self) -> None:
        if self.event is not None:
            torch.cuda.current_stream().wait_event(self.event)
        for w in self.works:
            w.wait()
        self.works = []


def wait_ready(t: Optional[Tensor]) -> None:
    """Called by consumers (``rasterization`` does, for every tensor it is handed) right before the first kernel
    that reads ``t``: if ``t`` is a parameter whose exchange is still in flight, the current stream waits for it."""
    r = getattr(t, "_pogs_ready", None) if t is not None else None
    if r is not None:
        r.wait()
        t._pogs_ready = None


def pending(t: Optional[Tensor]) -> bool:
    return t is not None and getattr(t, "_pogs_ready", None) is not None


class _Bucket:
    def __init__(self, names, params, world, rank, align=64):
        self.names = list(names)
        self.params = list(params)
        ref = self.params[0]
        self.offsets, at = [], 0
        for p in self.params:
            at = (at + align - 1) // align * align
            self.offsets.append(at)
            at += p.numel()
        unit = world * align
        self.total = (at + unit - 1) // unit * unit
        self.shard_len = self.total // world
        self.lo, self.hi = rank * self.shard_len, (rank + 1) * self.shard_len
        # the parameters become views of one flat buffer, so the all-gather writes them in place
        self.flat_param = torch.zeros(self.total, dtype=ref.dtype, device=ref.device)
        with torch.no_grad():
            for p, off in zip(self.params, self.offsets):
                self.flat_param[off:off + p.numel()].copy_(p.detach().reshape(-1))
                p.data = self.flat_param[off:off + p.numel()].view(p.shape)
        self.flat_grad = torch.zeros(self.total, dtype=ref.dtype, device=ref.device)
        self.todo = set()
        self.launched = False

    def segments(self):
        """[(name, lo in the parameter, hi in the parameter, lo in the shard)]: this rank's part of every parameter."""
        out = []
        for name, p, off in zip(self.names, self.params, self.offsets):
            a, b = max(self.lo, off), min(self.hi, off + p.numel())
            if b > a:
                out.append((name, a - off, b - off, a - self.lo))
        return out

    def nbytes(self) -> int:
        return self.total * self.flat_param.element_size()


class ShardedExchange:
    """Gradient exchange + optimiser step of a camera-sharded step, as the reference's DistributedDataParallel +
    per-group Adam produce them (``nerfstudio/pipelines/base_pipeline.py:279-282``, ``trainer.py:494-510``) --
    every rank ends the step with the same updated parameters, computed from the MEAN gradient over ranks --
    restated for point-to-point xGMI links and one optimiser shard per GPU:

      reduce-scatter (mean) of a bucket's gradients  ->  ``update`` on this rank's 1/G shard  ->  all-gather of the
      updated parameters, straight into the parameters' storage.

    Same bytes on the links as an all-reduce (2 (G-1)/G S), the optimiser's 28 bytes per parameter divided by G,
    and three places to overlap: a bucket starts the moment autograd has produced its last gradient (the feature
    gradients leave while the SH / projection backward still run), its update and all-gather run on a second
    stream, and the NEXT forward pass only waits, per tensor and on the device, where it first reads it
    (``wait_ready``: the compositor waits for the features ~0.6 ms into the step).

    ``params``: {name: leaf tensor}.  Tensors of at least ``big_bytes`` get a bucket each; the rest share one.
    ``update(segments)``: optimiser step on ``[(name, param_slice, grad_slice, lo)]`` -- 1-D views of this rank's
    shard (``lo`` = offset of the slice inside the parameter) holding the MEAN gradient; must update ``param_slice``
    in place on the current stream.  ``ShardAdam`` is the fused HIP one.
    Parameters that received no gradient in a step contribute zeros (``find_unused_parameters=True`` semantics).
    One difference from the reference remains: a parameter that NO rank produced a gradient for is still stepped,
    with a zero gradient (its moments decay and it keeps moving along them), where ``torch.optim.Adam`` skips a
    parameter whose ``.grad`` is None.  A rank sees only its own ``.grad``; learning that every rank had none would
    take one more collective per step.  POGS's groups all receive gradients once their loss terms are active.

    The library reduce-scatter / all-gather of the ``nccl`` backend (RCCL) has executed with ONE participant only
    (``POGS_EXCHANGE_SINGLE_RANK_COLLECTIVES=1``, ``tests/test_bench_rehearsal.py``: the calls, streams and buffer
    shapes, no data between GPUs); ``direct=False`` on ``gloo`` with two ranks exercises the same bucket offsets, shard
    views and mean through an all-reduce that stands in for the reduce-scatter.
    """

    def __init__(self, params: Dict[str, Tensor], update, group: Optional[dist.ProcessGroup] = None,
                 big_bytes: int = 32 << 20, direct: Optional[bool] = None):
        self.group, self.update = group, update
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.backend = dist.get_backend(group) if dist.is_initialized() else "none"
        # direct = each GPU sends shard j of its gradients straight to GPU j (all 7 xGMI links at once, one hop) and
        # sums what it receives; otherwise the library's reduce-scatter (gloo has none: there an all-reduce of the
        # whole buffer stands in for it, so that the non-direct bookkeeping can be run on CPUs).
        self.direct = (self.backend != "nccl") if direct is None else bool(direct)
        # one rank normally skips the collectives; POGS_EXCHANGE_SINGLE_RANK_COLLECTIVES=1 keeps them (a rehearsal of the
        # library calls, streams and buffer shapes of the N > 1 path on a one-GPU machine)
        self.collective = self.world > 1 or _single_rank_collectives()
        self.big_bytes = big_bytes
        self.side = None
        self.hooks = []
        self.rebuild(params)

    # ---- layout -------------------------------------------------------------------------------------------------
    def rebuild(self, params: Dict[str, Tensor]) -> None:
        """(Re)build the buckets: at construction, and after a refinement has changed the number of Gaussians
        (the parameters are new tensors then).  Collective: every rank must call it with the same shapes."""
        for h in self.hooks:
            h.remove()
        self.hooks = []
        self.params = {k: p for k, p in params.items() if p.requires_grad}
        assert self.params, "no parameter requires grad"
        for p in self.params.values():
            assert p.is_leaf, "parameters must be leaf tensors"
        big = [(k, p) for k, p in self.params.items() if p.numel() * p.element_size() >= self.big_bytes]
        small = [(k, p) for k, p in self.params.items() if p.numel() * p.element_size() < self.big_bytes]
        self.buckets: List[_Bucket] = [_Bucket([k], [p], self.world, self.rank) for k, p in big]
        if small:
            self.buckets.append(_Bucket([k for k, _ in small], [p for _, p in small], self.world, self.rank))
        self.bucket_of = {name: b for b in self.buckets for name in b.names}
        ref = next(iter(self.params.values()))
        if ref.is_cuda and self.side is None:
            self.side = torch.cuda.Stream(ref.device)
        for name, p in self.params.items():
            self.hooks.append(p.register_post_accumulate_grad_hook(self._hook(name)))

    def nbytes(self) -> int:
        return sum(b.nbytes() for b in self.buckets)

    def describe(self) -> str:
        return ("%d buckets (%s MB): reduce-scatter(%s) -> optimiser shard -> all-gather of parameters, launched from "
                "autograd hooks, next forward waits per tensor" % (
                    len(self.buckets), "+".join("%.0f" % (b.nbytes() / 1e6) for b in self.buckets),
                    "direct all-to-all + local sum" if self.direct else "library"))

    # ---- one step -----------------------------------------------------------------------------------------------
    def begin_step(self, wait: Optional[List[str]] = None) -> None:
        """Before the forward pass (gradients already set to None by the caller).  ``wait``: parameters the caller
        reads itself before handing anything to ``rasterization`` (activations applied outside, e.g. exp of the
        log-scales); the others are waited for where they are first read."""
        for b in self.buckets:
            b.todo = set(b.names)
            b.launched = False
        for name in (wait or []):
            wait_ready(self.params[name])

    def wait_all(self) -> None:
        for p in self.params.values():
            wait_ready(p)

    def _hook(self, name):
        def fire(_param):
            b = self.bucket_of[name]
            b.todo.discard(name)
            if not b.todo and not b.launched:
                self._launch(b)
        return fire

    def finish_step(self) -> None:
        """After ``backward()``: starts the buckets whose hooks did not all fire (a parameter no gradient reached)."""
        for b in self.buckets:
            if not b.launched:
                self._launch(b)

    @torch.no_grad()
    def _launch(self, b: _Bucket) -> None:
        b.launched = True
        W = self.world
        # 1. the bucket's gradients as one flat buffer (a single-tensor bucket with no padding is used where it is)
        single = len(b.params) == 1 and b.total == b.params[0].numel()
        g0 = b.params[0].grad
        if single and g0 is not None and g0.is_contiguous():
            flat = g0.reshape(-1)
        else:
            flat = b.flat_grad
            for p, off in zip(b.params, b.offsets):
                seg = flat[off:off + p.numel()]
                if p.grad is None:
                    seg.zero_()
                else:
                    seg.copy_(p.grad.reshape(-1))
        # 2. reduce-scatter (mean): this rank's shard of the summed gradients
        works = []
        if not self.collective:
            shard = flat[b.lo:b.hi]
        elif self.direct and flat.is_cuda and self.backend == "gloo":
            # rehearsal of the N > 1 path on a shared GPU without RCCL: gloo moves host memory only
            h_send, h_recv = flat.cpu(), torch.empty(flat.shape, dtype=flat.dtype)
            dist.all_to_all_single(h_recv, h_send, group=self.group)
            recv = h_recv.to(flat.device)
            shard = None
        elif self.direct:
            recv = torch.empty_like(flat)
            works.append(dist.all_to_all_single(recv, flat, group=self.group, async_op=True))
            shard = None
        elif self.backend == "nccl":
            # the mean of shard j lands in a buffer of rank j's own (RCCL accepts the in-place form, output = input's
            # own slice, as well; a separate output needs no such assumption and costs 1/G of the bucket)
            if getattr(b, "shard_grad", None) is None:
                b.shard_grad = torch.empty(b.shard_len, dtype=flat.dtype, device=flat.device)
            shard = b.shard_grad
            works.append(dist.reduce_scatter_tensor(shard, flat, op=dist.ReduceOp.AVG, group=self.group, async_op=True))
        else:
            # no reduce-scatter in this backend: all-reduce the buffer (through the host when it lives on a GPU) and
            # keep the same in-place shard view
            if flat.is_cuda:
                h = flat.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                flat.copy_(h)
            else:
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            flat.div_(W)
            shard = flat[b.lo:b.hi]
        # 3. optimiser shard + all-gather, on the second stream (HIP) so that the backward kernels queued behind this
        #    hook do not wait for the links
        cuda = flat.is_cuda
        ctx = torch.cuda.stream(self.side) if cuda else _NullContext()
        if cuda:
            self.side.wait_stream(torch.cuda.current_stream())
        with ctx:
            for w in works:
                w.wait()
            if self.collective and self.direct:
                shard = recv.view(W, b.shard_len).sum(dim=0).div_(W)
            segs = [(name, b.flat_param[b.lo + at:b.lo + at + (hi - lo)], shard[at:at + (hi - lo)], lo)
                    for name, lo, hi, at in b.segments()]
            self.update(segs)
            ag = []
            if self.collective and cuda and self.backend == "gloo":       # rehearsal: through the host, blocking
                h_full = torch.empty(b.total, dtype=b.flat_param.dtype)
                dist.all_gather_into_tensor(h_full, b.flat_param[b.lo:b.hi].cpu(), group=self.group)
                b.flat_param.copy_(h_full)
            elif self.collective:
                ag.append(dist.all_gather_into_tensor(b.flat_param, b.flat_param[b.lo:b.hi], group=self.group,
                                                      async_op=True))
            if cuda:
                for w in ag:
                    w.wait()
                ev = torch.cuda.Event()
                ev.record(self.side)
                ready = _Ready(event=ev)
                # everything the second stream read was allocated on the caller's stream: keep it from being handed
                # out again there before the second stream is done with it
                flat.record_stream(self.side)
                if self.collective and self.direct:
                    recv.record_stream(self.side)
                if shard is not None:
                    shard.record_stream(self.side)
            else:
                ready = _Ready(works=ag)
        for p in b.params:      # one waiter shared by the bucket's parameters (CPU handles can be waited on once)
            p._pogs_ready = ready


def _single_rank_collectives() -> bool:
    import os

    return dist.is_available() and dist.is_initialized() and os.environ.get("POGS_EXCHANGE_SINGLE_RANK_COLLECTIVES") == "1"


class _NullContext:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class ShardAdam:
    """``update`` callback of ``ShardedExchange``: Adam (``torch.optim.Adam`` arithmetic, no weight decay) on this
    rank's slices, all slices of a bucket in ONE ``pr_adam_step`` launch.  ``config``: {name: {"lr": float,
    "eps": 1e-15, "betas": (0.9, 0.999)}}, the per-group settings of ``pogs/pogs_config.py:38-85``; ``lr`` may be
    changed between steps (schedulers).  Moments are kept per (parameter, slice) and created on first use."""

    def __init__(self, config: Dict[str, dict]):
        self.config = config
        self.state: Dict[tuple, dict] = {}

    def __call__(self, segments) -> None:
        import ctypes

        from . import _lib
        from .rendering import _call, _stream

        entries, keep = [], []
        for name, p, g, lo in segments:
            cfg = self.config[name]
            st = self.state.get((name, lo, p.numel()))
            if st is None:
                st = self.state[(name, lo, p.numel())] = {
                    "step": 0, "exp_avg": torch.zeros_like(p), "exp_avg_sq": torch.zeros_like(p)}
            st["step"] += 1
            if g.data_ptr() % 16 or p.data_ptr() % 16:
                raise RuntimeError("shard slices must be 16-byte aligned")
            b1, b2 = cfg.get("betas", (0.9, 0.999))
            keep.append(g)
            entries.append(_lib.AdamGroup(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(),
                                          st["exp_avg_sq"].data_ptr(), p.numel(), st["step"], cfg["lr"], b1, b2,
                                          cfg.get("eps", 1e-15)))
        lib = _lib.load()
        chunk = lib.pr_adam_max_groups()
        for i in range(0, len(entries), chunk):
            part = entries[i:i + chunk]
            arr = (_lib.AdamGroup * len(part))(*part)
            _call("pr_adam_step", ctypes.cast(arr, ctypes.c_void_p), len(part), _stream())

    def full_state(self, exchange: ShardedExchange) -> Dict[str, dict]:
        """{name: {"step", "exp_avg", "exp_avg_sq"}} with whole-parameter moment tensors on every rank (all-gather
        of the shards): what a refinement (``DefaultStrategy``) and a checkpoint need."""
        out = {}
        for b in exchange.buckets:
            for key in ("exp_avg", "exp_avg_sq"):
                shard = torch.zeros(b.shard_len, dtype=b.flat_param.dtype, device=b.flat_param.device)
                for name, lo, hi, at in b.segments():
                    st = self.state.get((name, lo, hi - lo))
                    if st is not None:
                        shard[at:at + hi - lo] = st[key]
                full = torch.empty(b.total, dtype=shard.dtype, device=shard.device)
                if exchange.collective:
                    dist.all_gather_into_tensor(full, shard, group=exchange.group)
                else:
                    full.copy_(shard)
                for name, p, off in zip(b.names, b.params, b.offsets):
                    out.setdefault(name, {})[key] = full[off:off + p.numel()].view(p.shape).clone()
            for name in b.names:
                steps = [st["step"] for (n, _, _), st in self.state.items() if n == name]
                out[name]["step"] = max(steps) if steps else 0
        if exchange.collective:   # a rank that holds no slice of a parameter learns its step count from the others
            names = sorted(out)
            t = torch.tensor([out[n]["step"] for n in names], dtype=torch.int64,
                             device=next(iter(exchange.params.values())).device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=exchange.group)
            for n, v in zip(names, t.tolist()):
                out[n]["step"] = int(v)
        return out

    def load_full_state(self, exchange: ShardedExchange, full: Dict[str, dict]) -> None:
        """Inverse of ``full_state`` for the exchange's CURRENT layout (after ``rebuild``): keep this rank's slices."""
        self.state = {}
        for b in exchange.buckets:
            for name, lo, hi, _ in b.segments():
                if name in full and full[name].get("exp_avg") is not None:
                    self.state[(name, lo, hi - lo)] = {
                        "step": int(full[name]["step"]),
                        "exp_avg": full[name]["exp_avg"].reshape(-1)[lo:hi].clone(),
                        "exp_avg_sq": full[name]["exp_avg_sq"].reshape(-1)[lo:hi].clone()}


def sync_strategy_state(state: Dict[str, object], group: Optional[dist.ProcessGroup] = None) -> None:
    """Make the densification statistics of every rank those of the whole camera batch, right before a refinement:
    ``grad2d`` and ``count`` are sums over the views a Gaussian was seen in -> SUM over ranks; ``radii`` is a
    maximum -> MAX.  With identical statistics (and the same random generator state for the split samples) every
    rank refines to the same Gaussian set, so the replicated parameters stay replicated; the exchange is then
    rebuilt for the new shapes (``ShardedExchange.rebuild``)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    for key, op in (("grad2d", dist.ReduceOp.SUM), ("count", dist.ReduceOp.SUM), ("radii", dist.ReduceOp.MAX)):
        t = state.get(key)
        if isinstance(t, Tensor):
            dist.all_reduce(t, op=op, group=group)


class _AllReduceExchange:
    """``GradExchange`` behind the step interface ``bench.py`` drives: nothing before the backward pass, the mean
    all-reduce of every gradient after it."""

    def __init__(self, params: Dict[str, Tensor], group=None):
        self.ex = GradExchange(params.values())
        self.group = group

    def begin_step(self, wait=None) -> None:
        pass

    def finish_step(self) -> None:
        self.ex.all_reduce_mean(self.group)

    def wait_all(self) -> None:
        pass

    def describe(self) -> str:
        return "in-place gradient mean all-reduce after the backward pass (%.0f MB: SH, features, packed rest)" % (
            self.ex.nbytes() / 1e6)


# per-group Adam settings of the reference's method config (pogs/pogs_config.py:38-85), by this build's parameter names
REFERENCE_ADAM = {
    "means": {"lr": 1.6e-4, "eps": 1e-15}, "quats": {"lr": 1e-3, "eps": 1e-15},
    "log_scales": {"lr": 5e-3, "eps": 1e-15}, "opacity_logits": {"lr": 5e-2, "eps": 1e-15},
    "sh": {"lr": 2.5e-3, "eps": 1e-15}, "features": {"lr": 1e-2, "eps": 1e-15},
}


def make_exchange(kind: str, params: Dict[str, Tensor], group=None, lr_scale: float = 1.0):
    """``"sharded"``: ``ShardedExchange`` with the fused Adam on the shards; ``"allreduce"``: mean all-reduce only.
    ``lr_scale`` multiplies the reference's learning rates (0 keeps the parameters where they are while the
    optimiser still reads and writes everything: what a benchmark with a fixed workload wants)."""
    if kind == "allreduce":
        return _AllReduceExchange(params, group)
    if kind == "sharded":
        cfg = {k: dict(REFERENCE_ADAM.get(k, {"lr": 1e-3, "eps": 1e-15})) for k in params}
        for c in cfg.values():
            c["lr"] = c["lr"] * lr_scale
        import os

        direct = os.environ.get("POGS_EXCHANGE_DIRECT")       # "1" / "0" override the backend's default form
        return ShardedExchange(params, ShardAdam(cfg), group, direct=None if direct is None else direct == "1")
    raise ValueError(kind)
