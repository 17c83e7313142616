"""Drop-in for ``gsplat.rendering.rasterization`` on MI355X.

Mirrors the call the reference makes at ``pogs/pogs.py:352-371`` (RGB / RGB+ED with SH),
``pogs/pogs.py:419-437`` (96-channel features, absgrad), ``pogs/pogs.py:493-510`` (64-channel
features) and ``pogs/pogs.py:930-948`` (no-grad features): same argument names and meaning, same
``(render_colors, render_alphas, meta)`` return, same ``meta`` keys (gsplat 1.4.0), same error
behaviour for malformed shapes (``AssertionError``).  All device work goes through the C ABI of
``libpogs_rast.so``; there is no PyTorch fallback.

One extension beyond gsplat: ``features=[N,F]`` composites F extra per-Gaussian channels in the SAME
pass as the SH colours (output channels ``[rgb | features | depth]``), so an RGB + feature render
costs one projection, one binning and one walk over the tile lists instead of two of each.
"""
from __future__ import annotations

import math
import os
import threading
import time
from typing import Dict, List, Optional, Tuple

import torch
from torch import Tensor

from . import _lib
from .dist import pending, wait_ready


def _ptr(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


class StageTimer:
    """Optional per-entry-point device timing: HIP events recorded on the stream the kernels are launched
    on (torch's current stream), read back after a synchronise.  Used by bench.py for the roofline line."""

    def __init__(self, only=None):
        self.spans = {}
        self.only = None if only is None else set(only)   # time these entry points only (events are not free)

    def record(self, name):
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        self.spans.setdefault(name, []).append(ev)
        return ev

    def summary(self):
        """{name: (launches, total_ms)}; call after torch.cuda.synchronize()."""
        return {k: (len(v), sum(a.elapsed_time(b) for a, b in v)) for k, v in self.spans.items()}


_timer: Optional[StageTimer] = None


def set_stage_timer(timer: Optional[StageTimer]) -> None:
    global _timer
    _timer = timer


def _call(name: str, *args) -> None:
    """Invoke one C-ABI entry point, raising RuntimeError on a non-zero status."""
    fn = getattr(_lib.load(), name)
    if _timer is None or (_timer.only is not None and name not in _timer.only):
        _lib.check(fn(*args), name)
        return
    a, b = _timer.record(name)
    a.record()
    rc = fn(*args)
    b.record()
    _lib.check(rc, name)


def _rows(t: Tensor, width: int) -> Tuple[Tensor, int]:
    """Return (tensor, row stride in floats) for a [..., width] float32 tensor whose rows may be strided
    (e.g. a column slice of a packed buffer); copies only when the layout is not row-regular."""
    flat_ok = t.stride(-1) == 1 and t.dim() >= 2
    if flat_ok:
        ld = t.stride(-2)
        ok = ld >= width
        for d in range(t.dim() - 3, -1, -1):
            ok = ok and t.stride(d) == t.stride(d + 1) * t.shape[d + 1]
        if ok:
            return t, ld
    t = t.contiguous()
    return t, width


def _check_device(*ts: Optional[Tensor]) -> torch.device:
    dev = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("pogs_amd.rasterization needs tensors on a HIP device (no CPU fallback)")
        if t.dtype != torch.float32:
            raise RuntimeError("pogs_amd.rasterization computes in float32")
        dev = t.device if dev is None else dev
        if t.device != dev:
            raise RuntimeError("all tensors must live on the same device")
    return dev


class _Projection(torch.autograd.Function):
    """EWA projection; stands in for gsplat ``fully_fused_projection`` (non-packed)."""

    @staticmethod
    def forward(ctx, means, quats, scales, viewmats, Ks, width, height, eps2d, near_plane, far_plane,
                radius_clip, calc_compensations):
        lib = _lib.load()
        C, N = viewmats.shape[0], means.shape[0]
        means, quats, scales = means.contiguous(), quats.contiguous(), scales.contiguous()
        viewmats, Ks = viewmats.contiguous(), Ks.contiguous()
        dev = means.device
        radii = torch.empty(C, N, dtype=torch.int32, device=dev)
        means2d = torch.empty(C, N, 2, dtype=torch.float32, device=dev)
        depths = torch.empty(C, N, dtype=torch.float32, device=dev)
        conics = torch.empty(C, N, 3, dtype=torch.float32, device=dev)
        comps = torch.empty(C, N, dtype=torch.float32, device=dev) if calc_compensations else None
        with torch.cuda.device(dev):
            _call("pr_project_fwd", C, N, _ptr(means), _ptr(quats), _ptr(scales), _ptr(viewmats), _ptr(Ks),
                                          width, height, eps2d, near_plane, far_plane, radius_clip, _ptr(radii),
                                          _ptr(means2d), _ptr(depths), _ptr(conics), _ptr(comps), _stream())
        ctx.save_for_backward(means, quats, scales, viewmats, Ks, radii)
        ctx.args = (width, height, eps2d)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)   # unused outputs arrive as None instead of freshly filled zero tensors
        return radii, means2d, depths, conics, comps

    @staticmethod
    def backward(ctx, _v_radii, v_means2d, v_depths, v_conics, v_comps):
        lib = _lib.load()
        means, quats, scales, viewmats, Ks, radii = ctx.saved_tensors
        width, height, eps2d = ctx.args
        C, N = radii.shape
        dev = means.device
        if v_means2d is None:
            v_means2d = torch.zeros(C, N, 2, dtype=torch.float32, device=dev)
        if v_conics is None:
            v_conics = torch.zeros(C, N, 3, dtype=torch.float32, device=dev)
        v_means2d, ld_m = _rows(v_means2d, 2)
        v_conics, ld_c = _rows(v_conics, 3)
        v_depths = None if v_depths is None else v_depths.contiguous()
        v_comps = None if v_comps is None else v_comps.contiguous()
        v_means = torch.empty(N, 3, dtype=torch.float32, device=dev)
        v_quats = torch.empty(N, 4, dtype=torch.float32, device=dev)
        v_scales = torch.empty(N, 3, dtype=torch.float32, device=dev)
        v_viewmats = torch.zeros(C, 4, 4, dtype=torch.float32, device=dev) if ctx.needs_input_grad[3] else None
        with torch.cuda.device(dev):
            _call("pr_project_bwd", C, N, _ptr(means), _ptr(quats), _ptr(scales), _ptr(viewmats), _ptr(Ks),
                                          width, height, eps2d, _ptr(radii), _ptr(v_means2d), ld_m, _ptr(v_depths),
                                          _ptr(v_conics), ld_c, _ptr(v_comps), _ptr(v_means), _ptr(v_quats),
                                          _ptr(v_scales), _ptr(v_viewmats), _stream())
        return (v_means if ctx.needs_input_grad[0] else None, v_quats if ctx.needs_input_grad[1] else None,
                v_scales if ctx.needs_input_grad[2] else None, v_viewmats, None, None, None, None, None, None,
                None, None)


class _ProjectionRigid(torch.autograd.Function):
    """``_Projection`` of a frozen model whose Gaussians are moved, inside the kernel, by their rigid group's
    ``p2w * delta * inverse(p2w)`` (the tracker's ``apply_to_model``, ``pogs/tracking/utils.py:53-93``); the gradient
    goes to ``p_deltas`` (and the view matrices) only.  Also returns the moved means, which the view-dependent
    colours need."""

    @staticmethod
    def forward(ctx, means, quats, scales, labels, init_p2ws, p_deltas, viewmats, Ks, width, height, eps2d, near_plane,
                far_plane, radius_clip, calc_compensations):
        C, N, G = viewmats.shape[0], means.shape[0], p_deltas.shape[0]
        means, quats, scales = means.contiguous(), quats.contiguous(), scales.contiguous()
        labels, init_p2ws, p_deltas = labels.contiguous(), init_p2ws.contiguous(), p_deltas.contiguous()
        viewmats, Ks = viewmats.contiguous(), Ks.contiguous()
        dev = means.device
        radii = torch.empty(C, N, dtype=torch.int32, device=dev)
        means2d = torch.empty(C, N, 2, dtype=torch.float32, device=dev)
        depths = torch.empty(C, N, dtype=torch.float32, device=dev)
        conics = torch.empty(C, N, 3, dtype=torch.float32, device=dev)
        comps = torch.empty(C, N, dtype=torch.float32, device=dev) if calc_compensations else None
        means_world = torch.empty(N, 3, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _call("pr_project_fwd_rigid", C, N, G, _ptr(init_p2ws), _ptr(p_deltas), _ptr(labels), _ptr(means),
                  _ptr(quats), _ptr(scales), _ptr(viewmats), _ptr(Ks), width, height, eps2d, near_plane, far_plane,
                  radius_clip, _ptr(radii), _ptr(means2d), _ptr(depths), _ptr(conics), _ptr(comps), _ptr(means_world),
                  _stream())
        ctx.save_for_backward(means, quats, scales, labels, init_p2ws, p_deltas, viewmats, Ks, radii)
        ctx.args = (width, height, eps2d)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)
        return radii, means2d, depths, conics, comps, means_world

    @staticmethod
    def backward(ctx, _v_radii, v_means2d, v_depths, v_conics, v_comps, v_means_world):
        means, quats, scales, labels, init_p2ws, p_deltas, viewmats, Ks, radii = ctx.saved_tensors
        if any(ctx.needs_input_grad[:3]):
            raise NotImplementedError("rigid= renders a frozen model: means, quats and scales receive no gradient "
                                      "(use pogs_amd.tracking.apply_to_model + rasterization to train the model too)")
        width, height, eps2d = ctx.args
        C, N = radii.shape
        dev = means.device
        if v_means2d is None:
            v_means2d = torch.zeros(C, N, 2, dtype=torch.float32, device=dev)
        if v_conics is None:
            v_conics = torch.zeros(C, N, 3, dtype=torch.float32, device=dev)
        v_means2d, ld_m = _rows(v_means2d, 2)
        v_conics, ld_c = _rows(v_conics, 3)
        v_depths = None if v_depths is None else v_depths.contiguous()
        v_comps = None if v_comps is None else v_comps.contiguous()
        v_means_world = None if v_means_world is None else v_means_world.contiguous()
        v_p_deltas = torch.zeros_like(p_deltas)
        v_viewmats = torch.zeros(C, 4, 4, dtype=torch.float32, device=dev) if ctx.needs_input_grad[6] else None
        with torch.cuda.device(dev):
            _call("pr_project_bwd_rigid", C, N, p_deltas.shape[0], _ptr(init_p2ws), _ptr(p_deltas), _ptr(labels),
                  _ptr(means), _ptr(quats), _ptr(scales), _ptr(viewmats), _ptr(Ks), width, height, eps2d, _ptr(radii),
                  _ptr(v_means2d), ld_m, _ptr(v_depths), _ptr(v_conics), ld_c, _ptr(v_comps), _ptr(v_means_world),
                  _ptr(v_p_deltas), _ptr(v_viewmats), _stream())
        return (None, None, None, None, None, v_p_deltas if ctx.needs_input_grad[5] else None, v_viewmats,
                None, None, None, None, None, None, None, None)


class _SphericalHarmonics(torch.autograd.Function):
    """View-dependent colour; stands in for gsplat ``spherical_harmonics`` fused with the caller's
    ``clamp_min(. + 0.5, 0)``."""

    @staticmethod
    def forward(ctx, degree, means, campos, coeffs, radii, post):
        lib = _lib.load()
        C, N, K = campos.shape[0], means.shape[0], coeffs.shape[1]
        means, campos, coeffs = means.contiguous(), campos.contiguous(), coeffs.contiguous()
        colors = torch.empty(C, N, 3, dtype=torch.float32, device=means.device)
        with torch.cuda.device(means.device):
            _call("pr_sh_fwd", C, N, K, degree, _ptr(means), _ptr(campos), _ptr(coeffs), _ptr(radii),
                                     int(post), _ptr(colors), 3, _stream())
        ctx.save_for_backward(means, campos, coeffs, radii)
        ctx.args = (degree, int(post))
        ctx.set_materialize_grads(False)
        return colors

    @staticmethod
    def backward(ctx, v_colors):
        lib = _lib.load()
        if v_colors is None:
            return None, None, None, None, None, None
        means, campos, coeffs, radii = ctx.saved_tensors
        degree, post = ctx.args
        C, N, K = campos.shape[0], means.shape[0], coeffs.shape[1]
        dev = means.device
        v_colors, ld = _rows(v_colors, 3)
        v_coeffs = torch.empty(N, K, 3, dtype=torch.float32, device=dev)
        v_means = torch.empty(N, 3, dtype=torch.float32, device=dev) if ctx.needs_input_grad[1] else None
        v_campos = torch.zeros(C, 3, dtype=torch.float32, device=dev) if ctx.needs_input_grad[2] else None
        with torch.cuda.device(dev):
            _call("pr_sh_bwd", C, N, K, degree, _ptr(means), _ptr(campos), _ptr(coeffs), _ptr(radii), post,
                                     _ptr(v_colors), ld, _ptr(v_coeffs), _ptr(v_means), _ptr(v_campos), _stream())
        return None, v_means, v_campos, (v_coeffs if ctx.needs_input_grad[3] else None), None, None


_ORDER_MIN_TILES = int(os.environ.get("POGS_RAST_ORDER_MIN_TILES", "2048"))  # fewer tiles than wave slots: nothing to balance
_ASYNC_DEPTH_ORDER_MIN = 100_000
# POGS_RAST_BINNING=lists builds the per-tile lists with per-tile counters and one LDS sort per tile instead of the
# two-level device sort.  Measured at 1M Gaussians / 1080p it is SLOWER (0.66 ms against 0.50 ms): 6.4M scattered
# integer atomics cost 0.18 ms in the counting pass and again in the emission.  Kept for small scenes and as a
# cross-check of the sorted lists (tests compare the two).
_TILE_LISTS = os.environ.get("POGS_RAST_BINNING", "sort") == "lists"


class _CallState:
    """What survives from one ``rasterization`` call to the next, shared by every thread and stream that renders:
    recent intersection counts per binning configuration (to size the lists without waiting for the count), a
    pool of page-locked words for the asynchronous read-back of the count, and the second stream of each (device,
    caller stream) pair.  Every access goes through one lock; a buffer or stream handed out belongs to the call
    that took it until it is given back, so calls in flight on different threads or streams (a viewer thread next
    to the trainer, autograd's backward thread re-rendering) share nothing mutable."""

    HISTORY = 8           # counts remembered per configuration
    STABLE = 0.02         # spread below which the tight margin applies
    LOOSE = 0.25          # spread up to which a guess is still made (cycling cameras)

    def __init__(self):
        self.lock = threading.Lock()
        self.history: Dict[tuple, List[int]] = {}
        self.pinned: Dict[int, List[Tensor]] = {}
        self.graveyard: List[Tensor] = []
        self.side_streams: Dict[tuple, "torch.cuda.Stream"] = {}
        self.speculate = os.environ.get("POGS_RAST_SPECULATE", "1") != "0"
        # how a speculating call learns the true count: "publish" = a one-thread kernel stores it into page-locked
        # memory and the host polls (no copy command, no event on the stream); "event" = copy + event
        self.readback = os.environ.get("POGS_RAST_COUNT_READBACK", "publish")
        self.sequence = 0

    def guess(self, key) -> Optional[int]:
        """Capacity to size the lists with, or None when the host should wait for the exact count."""
        with self.lock:
            h = self.history.get(key)
            if not self.speculate or h is None or len(h) < 2 or min(h) <= 0:
                return None
            hi, lo = max(h), min(h)
            if hi - lo <= self.STABLE * hi:
                return int(hi * 1.03) + 1024
            if hi - lo <= self.LOOSE * hi:
                return int(hi * 1.10) + 1024
            return None

    def record(self, key, n_isects: int, missed: bool) -> None:
        with self.lock:
            if missed:                       # a guess was too small: no speculation until two more exact counts
                self.history[key] = [n_isects]
                return
            h = self.history.setdefault(key, [])
            h.append(n_isects)
            del h[:-self.HISTORY]

    def take_pinned(self, dev: torch.device) -> Tensor:
        """Two page-locked words: [0] the count, [1] the sequence number of the publication that wrote it."""
        with self.lock:
            pool = self.pinned.setdefault(dev.index, [])
            if pool:
                return pool.pop()
        return torch.zeros(2, dtype=torch.int64, pin_memory=True)

    def next_sequence(self) -> int:
        with self.lock:
            self.sequence += 1
            return self.sequence

    def give_pinned(self, dev: torch.device, t: Tensor) -> None:
        with self.lock:
            self.pinned.setdefault(dev.index, []).append(t)

    def retire_pinned(self, t: Tensor) -> None:
        """Keep words that a failed call may still have a queued write to alive for the rest of the process (never
        handed out again, never freed: a page-locked block returned to torch could be given to another user)."""
        with self.lock:
            self.graveyard.append(t)

    def side_stream(self, dev: torch.device, main: "torch.cuda.Stream") -> "torch.cuda.Stream":
        key = (dev.index, main.cuda_stream)
        with self.lock:
            s = self.side_streams.get(key)
        if s is None:
            s = torch.cuda.Stream(dev)
            with self.lock:
                s = self.side_streams.setdefault(key, s)
        return s


_state = _CallState()


def set_count_speculation(on: bool) -> None:
    """Switch the sizing of the intersection lists from earlier counts on or off (off: every call waits for its
    exact count on the host, as gsplat does)."""
    _state.speculate = bool(on)


def _depth_order_async(depths: Optional[Tensor], means: Optional[Tensor] = None, viewmats: Optional[Tensor] = None):
    """Start the depth sort of all C*N Gaussians on a second stream and return (order, event): it runs next to the
    colour evaluation and the tile counting instead of in front of the emission (four small radix passes that leave
    most of the GPU idle when run alone).  With ``means`` [N,3] and ``viewmats`` [C,4,4] the keys are the camera-space
    depths of ALL Gaussians (``pr_view_depths``: the projection's own bits for the ones it keeps, where it reports 0
    for the ones it culls), so the order does not depend on what this call's intrinsics and image size cull and can
    serve other renders of the same model through the same view matrices, and the sort can start before the projection;
    else the projection's ``depths`` [C,N]."""
    lib = _lib.load()
    if means is not None:
        means, viewmats = means.detach().contiguous(), viewmats.detach().contiguous()
        dev, C, N = means.device, viewmats.shape[0], means.shape[0]
    else:
        depths = depths.detach().contiguous()
        dev, (C, N) = depths.device, depths.shape
    main = torch.cuda.current_stream(dev)
    side = _state.side_stream(dev, main)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        order = torch.empty(C * N, dtype=torch.int32, device=dev)
        nb = lib.pr_depth_order_scratch_bytes(C, N)
        scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
        keys = depths
        if means is not None:
            keys = torch.empty(C, N, dtype=torch.float32, device=dev)
            _call("pr_view_depths", C, N, _ptr(means), _ptr(viewmats), _ptr(keys), side.cuda_stream)
        _call("pr_depth_order", C, N, _ptr(keys), _ptr(order), _ptr(scratch), nb, side.cuda_stream)
        done = torch.cuda.Event()
        done.record(side)
    if means is not None:
        means.record_stream(side)
        viewmats.record_stream(side)
    else:
        depths.record_stream(side)
    order.record_stream(main)
    return order, done


def _await_published(host: Tensor, seq: int, timeout_s: float = 60.0) -> int:
    """Wait until the kernel of ``pr_publish_count`` has stored sequence number ``seq`` into host[1]; -> host[0].
    The count is normally there within tens of microseconds: a short spin, then 20 us sleeps (which release the
    interpreter lock for a viewer thread), and a look at the stream's error state from time to time so that a GPU
    fault surfaces as an exception instead of a minute-long stall."""
    words = host.numpy()
    spins = 0
    deadline = None
    while int(words[1]) != seq:
        spins += 1
        if spins < 200:
            continue
        time.sleep(20e-6)
        if spins % 2000 == 0:
            now = time.monotonic()
            if deadline is None:
                deadline = now + timeout_s
            torch.cuda.current_stream().query()   # raises if the device has reported an error
            if now > deadline:
                raise RuntimeError("pogs_amd: the intersection count never reached the host (sequence %d): %s"
                                   % (seq, _lib.load().pr_last_hip_error().decode()))
    return int(words[0])


def _bin_tiles(means2d: Tensor, radii: Tensor, depths: Tensor, tile_size: int, tile_width: int, tile_height: int,
               want_ids: bool, want_offsets: bool, want_order: bool = False, conics: Optional[Tensor] = None,
               opacities: Optional[Tensor] = None, depth_order=None):
    """Tile binning: -> tiles_per_gauss [C,N] int32, isect_ids [I] int64 or None, flatten_ids [I] int32,
    isect_offsets [C,th,tw] int32 or None[, tile_order [C*th*tw] int32 or None when ``want_order``: tiles
    sorted by list length, longest first (None below ``_ORDER_MIN_TILES`` tiles)].

    With ``conics`` [C,N,3] and ``opacities`` [C,N] only the (Gaussian, tile) pairs that can reach
    alpha >= 1/255 somewhere in the tile are binned (the render path); without them, gsplat's rule (every tile
    under the 3-sigma bounding box)."""
    lib = _lib.load()
    C, N = radii.shape
    dev = means2d.device
    means2d, radii, depths = means2d.detach().contiguous(), radii.contiguous(), depths.detach().contiguous()
    if conics is not None:
        conics, opacities = conics.detach().contiguous(), opacities.detach().contiguous()
    tiles_per_gauss = torch.empty(C, N, dtype=torch.int32, device=dev)
    offsets = torch.empty(C, tile_height, tile_width, dtype=torch.int32, device=dev) if want_offsets else None
    n_tiles = C * tile_height * tile_width
    order = None
    if want_order and want_offsets and n_tiles >= _ORDER_MIN_TILES and C * N > 0:
        order = torch.empty(n_tiles, dtype=torch.int32, device=dev)
    if C * N == 0:
        if offsets is not None:
            offsets.zero_()
        out = (tiles_per_gauss, torch.empty(0, dtype=torch.int64, device=dev) if want_ids else None,
               torch.empty(0, dtype=torch.int32, device=dev), offsets)
        return out + (None,) if want_order else out
    cum = torch.empty(C * N, dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        nb = lib.pr_isect_count_scratch_bytes(C, N)
        scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
        _call("pr_isect_count", C, N, _ptr(means2d), _ptr(radii), _ptr(conics), _ptr(opacities), tile_size,
              tile_width, tile_height, _ptr(tiles_per_gauss), _ptr(cum), _ptr(scratch), nb, _stream())

        order_ptr = None
        if depth_order is not None:     # (order, event) of _depth_order_async: the emission waits for it
            torch.cuda.current_stream(dev).wait_event(depth_order[1])
            order_ptr = depth_order[0].data_ptr()

        def emit(capacity, count_on_device):
            ids = torch.empty(capacity, dtype=torch.int64, device=dev) if want_ids else None
            flat = torch.empty(capacity, dtype=torch.int32, device=dev)
            nb = lib.pr_isect_sort_scratch_bytes(C, N, capacity, n_tiles if order is not None else 0)
            scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
            _call("pr_isect_emit_sort", C, N, _ptr(means2d), _ptr(radii), _ptr(depths), _ptr(conics),
                  _ptr(opacities), _ptr(tiles_per_gauss), tile_size, tile_width, tile_height, capacity, _ptr(ids), _ptr(flat), _ptr(offsets), _ptr(order),
                  cum[-1:].data_ptr() if count_on_device else None, order_ptr, _ptr(scratch), nb, _stream())
            return ids, flat

        # The intersection count sizes the lists.  Waiting for it on the host leaves the GPU idle for the
        # round trip, so when the same configuration has produced similar counts on the previous calls
        # (a tracking loop, repeated or cycling views) the lists are sized from the largest of them plus a
        # margin, the kernels read the true count from device memory, and the host compares afterwards -- by
        # then the GPU has the sort queued.  A guess that turns out too small costs one repeated call and
        # switches the speculation off until the counts are settled again.
        key = (dev.index, C, N, tile_size, tile_width, tile_height, conics is not None)
        guess = _state.guess(key)
        if guess is None:
            n_isects = int(cum[-1].item())
            isect_ids, flatten_ids = emit(n_isects, False)
        else:
            host = _state.take_pinned(dev)
            try:
                if _state.readback == "publish":
                    seq = _state.next_sequence()
                    _call("pr_publish_count", cum[-1:].data_ptr(), host.data_ptr(), seq, _stream())
                    isect_ids, flatten_ids = emit(guess, True)
                    n_isects = _await_published(host, seq)
                else:
                    host[:1].copy_(cum[-1:], non_blocking=True)
                    done = torch.cuda.Event()
                    done.record()
                    isect_ids, flatten_ids = emit(guess, True)
                    done.synchronize()
                    n_isects = int(host[0].item())
            except BaseException:
                # a kernel (or copy) that writes into `host` may still be queued: the words must not go back to the
                # pool, or be freed, before the stream has drained
                try:
                    torch.cuda.current_stream(dev).synchronize()
                finally:
                    _state.retire_pinned(host)
                raise
            _state.give_pinned(dev, host)
            if n_isects > guess:
                isect_ids, flatten_ids = emit(n_isects, False)
            else:
                flatten_ids = flatten_ids[:n_isects]
                isect_ids = None if isect_ids is None else isect_ids[:n_isects]
        _state.record(key, n_isects, guess is not None and n_isects > guess)
    if want_order:
        return tiles_per_gauss, isect_ids, flatten_ids, offsets, order
    return tiles_per_gauss, isect_ids, flatten_ids, offsets


def _tile_lists(means2d: Tensor, radii: Tensor, depths: Tensor, conics: Tensor, opacities: Tensor, tile_size: int,
                tile_width: int, tile_height: int, want_order: bool = True):
    """The compositor's lists without a device-wide sort (``pr_tile_lists_count`` / ``pr_tile_lists_fill``): only the
    (Gaussian, tile) pairs that can reach alpha >= 1/255 in the tile, per tile in (depth bits, flat index) order.
    -> (flatten_ids [I] int32, isect_offsets [C,th,tw] int32, tile_order [C*th*tw] int32 or None)."""
    lib = _lib.load()
    C, N = radii.shape
    dev = means2d.device
    means2d, radii, depths = means2d.detach().contiguous(), radii.contiguous(), depths.detach().contiguous()
    conics, opacities = conics.detach().contiguous(), opacities.detach().contiguous()
    n_tiles = C * tile_height * tile_width
    offsets = torch.empty(C, tile_height, tile_width, dtype=torch.int32, device=dev)
    order = None
    if want_order and n_tiles >= _ORDER_MIN_TILES and C * N > 0:
        order = torch.empty(n_tiles, dtype=torch.int32, device=dev)
    if C * N == 0 or n_tiles == 0:
        offsets.zero_()
        return torch.empty(0, dtype=torch.int32, device=dev), offsets, None
    counters = torch.empty(n_tiles, dtype=torch.int32, device=dev)
    total = torch.empty(1, dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        _call("pr_tile_lists_count", C, N, _ptr(means2d), _ptr(radii), _ptr(conics), _ptr(opacities), tile_size,
              tile_width, tile_height, None, _ptr(counters), _ptr(offsets), _ptr(order), _ptr(total), _stream())

        def fill(capacity):
            flat = torch.empty(capacity, dtype=torch.int32, device=dev)
            keys = torch.empty(capacity, dtype=torch.int64, device=dev)
            _call("pr_tile_lists_fill", C, N, _ptr(means2d), _ptr(radii), _ptr(depths), _ptr(conics), _ptr(opacities),
                  tile_size, tile_width, tile_height, _ptr(offsets), _ptr(counters), capacity, _ptr(total), _ptr(keys),
                  _ptr(flat), _stream())
            return flat

        # list sizes: from the counts of earlier calls when they are settled (see _CallState), else wait for the total
        key = (dev.index, C, N, tile_size, tile_width, tile_height, "lists")
        guess = _state.guess(key)
        if guess is None:
            n_isects = int(total.item())
            flatten_ids = fill(n_isects)
        else:
            host = _state.take_pinned(dev)
            try:
                host[:1].copy_(total, non_blocking=True)
                done = torch.cuda.Event()
                done.record()
                flatten_ids = fill(guess)
                done.synchronize()
                n_isects = int(host[0].item())
            except BaseException:
                try:
                    torch.cuda.current_stream(dev).synchronize()
                finally:
                    _state.retire_pinned(host)
                raise
            _state.give_pinned(dev, host)
            if n_isects > guess:      # the counters were consumed by the undersized fill: count again
                _call("pr_tile_lists_count", C, N, _ptr(means2d), _ptr(radii), _ptr(conics), _ptr(opacities),
                      tile_size, tile_width, tile_height, None, _ptr(counters), _ptr(offsets), _ptr(order),
                      _ptr(total), _stream())
                flatten_ids = fill(n_isects)
            else:
                flatten_ids = flatten_ids[:n_isects]
        _state.record(key, n_isects, guess is not None and n_isects > guess)
    return flatten_ids, offsets, order


def isect_tiles(means2d: Tensor, radii: Tensor, depths: Tensor, tile_size: int, tile_width: int,
                tile_height: int) -> Tuple[Tensor, Tensor, Tensor]:
    """Stands in for gsplat ``isect_tiles`` (non-packed, sorted).  means2d [C,N,2], radii [C,N] int32,
    depths [C,N] -> tiles_per_gauss [C,N] int32, isect_ids [I] int64 (sorted), flatten_ids [I] int32."""
    tiles_per_gauss, isect_ids, flatten_ids, _ = _bin_tiles(means2d, radii, depths, tile_size, tile_width,
                                                            tile_height, True, False)
    return tiles_per_gauss, isect_ids, flatten_ids


def _isect_ids_from_lists(isect_offsets: Tensor, flatten_ids: Tensor, depths: Tensor) -> Tensor:
    """gsplat's 64-bit intersection keys, camera << (32 + tile_bits) | tile << 32 | float_bits(depth), rebuilt
    from the tile lists: intersection k belongs to the last tile whose range starts at or before k."""
    C, th, tw = isect_offsets.shape
    n_tiles = th * tw
    tile_bits = int(n_tiles).bit_length()
    k = torch.arange(flatten_ids.numel(), device=flatten_ids.device, dtype=torch.int32)
    tile_flat = torch.searchsorted(isect_offsets.flatten(), k, right=True).long() - 1
    hi = ((tile_flat // n_tiles) << tile_bits) | (tile_flat % n_tiles)
    d = depths.detach().contiguous().view(-1).view(torch.int32)[flatten_ids.long()].long() & 0xFFFFFFFF
    return (hi << 32) | d


class RenderMeta(dict):
    """The ``meta`` dict of ``rasterization``.  Every gsplat key is present.  The binning tables gsplat
    reports (``tiles_per_gauss``, ``isect_ids``, ``flatten_ids``, ``isect_offsets``: every tile under each
    Gaussian's 3-sigma bounding box) are not what the compositor here walks -- it bins only the pairs that can
    contribute -- and nothing on the render path reads them, so they are computed when first asked for."""

    def __init__(self, values, lazy):
        super().__init__(values)
        self._lazy = lazy      # one thunk producing all the lazy entries at once
        for k in lazy.keys:
            dict.__setitem__(self, k, None)

    def _resolve(self):
        if self._lazy is not None:
            thunk, self._lazy = self._lazy, None
            for k, v in thunk().items():
                dict.__setitem__(self, k, v)

    def __getitem__(self, key):
        if self._lazy is not None and key in self._lazy.keys:
            self._resolve()
        return dict.__getitem__(self, key)

    def get(self, key, default=None):
        return self[key] if key in self else default

    def items(self):
        self._resolve()
        return dict.items(self)

    def values(self):
        self._resolve()
        return dict.values(self)

    def copy(self):
        self._resolve()
        return dict(self)

    # ``dict(meta)``, ``{**meta}`` and ``other.update(meta)`` copy a dict subclass's storage directly -- placeholders
    # included -- unless the subclass overrides ``__iter__``; with these two they go through keys() / __getitem__.
    def __iter__(self):
        self._resolve()
        return dict.__iter__(self)

    def keys(self):
        self._resolve()
        return dict.keys(self)


class _ReferenceBinning:
    """Thunk for RenderMeta: gsplat's binning tables of one render."""

    keys = ("tiles_per_gauss", "isect_ids", "flatten_ids", "isect_offsets")

    def __init__(self, means2d, radii, depths, tile_size, tile_width, tile_height):
        self.args = (means2d.detach(), radii, depths.detach(), tile_size, tile_width, tile_height)

    def __call__(self):
        tpg, ids, flat, offs = _bin_tiles(*self.args, True, True)
        return {"tiles_per_gauss": tpg, "isect_ids": ids, "flatten_ids": flat, "isect_offsets": offs}


def isect_offset_encode(isect_ids: Tensor, n_cameras: int, tile_width: int, tile_height: int) -> Tensor:
    """Stands in for gsplat ``isect_offset_encode``: int32 [C, tile_height, tile_width]."""
    lib = _lib.load()
    dev = isect_ids.device
    offsets = torch.empty(n_cameras, tile_height, tile_width, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _call("pr_isect_offsets", isect_ids.numel(), _ptr(isect_ids.contiguous()), n_cameras, tile_width,
                                        tile_height, _ptr(offsets), _stream())
    return offsets


def _tile_order(isect_offsets: Tensor, n_isects: int) -> Optional[Tensor]:
    """Flat tile indices sorted by list length, longest first: wavefronts pick the heavy tiles up first so
    that the launch does not end on a few long ones (results do not depend on the order)."""
    o = isect_offsets.flatten()
    if o.numel() < _ORDER_MIN_TILES:
        return None
    ends = torch.cat([o[1:], o.new_tensor([n_isects])])
    return torch.argsort(ends - o, descending=True).to(torch.int32)


def _segments(parts, C, N):
    """parts: tensors [C,N,w] (per camera) or [N,w] (shared by all cameras), in channel order ->
    (contiguous tensors, [(ptr, width, ld, shared)])."""
    tensors, items = [], []
    for t in parts:
        shared = t.dim() == 2
        assert t.shape[:-1] == ((N,) if shared else (C, N)), t.shape
        t = t.contiguous()
        tensors.append(t)
        items.append((t.data_ptr(), t.shape[-1], t.shape[-1], shared))
    return tensors, items


_EARLY_FILL_MIN_BYTES = 32 << 20


def _split_backward(D: int, tile_size: int) -> bool:
    """Run the backward compositor as two concurrent kernels (geometry | colour gradients)?  Opt-in through
    POGS_RAST_BWD_SPLIT=1; the colour half runs on the matrix cores when POGS_RAST_BWD_MFMA=4 (16x16 tiles, up to
    48 channels)."""
    return tile_size == 16 and D <= 48 and os.environ.get("POGS_RAST_BWD_SPLIT", "0") == "1"


def _grad_targets(parts, want, geom, C, N, dev):
    """Zero-filled accumulation targets of the backward compositor: the packed geometry gradients [C*N,8] and the
    narrow colour parts (rgb, depth) share one allocation and one fill; wide parts (features) get their own.
    -> (v_geom or None, its row pitch, [gradient buffer or None per part], [row pitch per part]).

    When the per-camera narrow parts fit beside the eight geometry values in a row of 16 floats, the pool is that
    table of rows, [C*N,16] = (geometry 0-7 | narrow parts | padding): the memory system delivers float atomics one
    aligned 64-byte segment at a time (about 21 G segments/s on an MI355X whatever they carry), and all of a
    Gaussian's narrow gradients then travel as one request instead of one per tensor.  The views handed out are
    row-strided; pr_sh_bwd / pr_project_bwd take row pitches."""
    narrow = [w and p.shape[-1] <= 4 and p.dim() == 3 for w, p in zip(want, parts)]
    if geom and any(narrow) and 8 + sum(p.shape[-1] for p, k in zip(parts, narrow) if k) <= 16:
        rows = torch.zeros(C, N, 16, dtype=torch.float32, device=dev)
        at, v_parts, lds = 8, [], []
        for p, w, k in zip(parts, want, narrow):
            if k:
                v_parts.append(rows[..., at:at + p.shape[-1]])
                lds.append(16)
                at += p.shape[-1]
            else:
                v_parts.append(torch.zeros_like(p) if w else None)
                lds.append(p.shape[-1])
        return rows[..., :8], 16, v_parts, lds
    pooled = [w and p.shape[-1] <= 4 for w, p in zip(want, parts)]
    n_pool = (C * N * 8 if geom else 0) + sum((p.numel() + 7) // 8 * 8 for p, k in zip(parts, pooled) if k)
    pool = torch.zeros(n_pool, dtype=torch.float32, device=dev) if n_pool else None
    at = C * N * 8 if geom else 0
    v_geom = pool[:at].view(C, N, 8) if geom else None
    v_parts = []
    for p, w, k in zip(parts, want, pooled):
        if not w:
            v_parts.append(None)
        elif k:
            v_parts.append(pool[at:at + p.numel()].view(p.shape))
            at += (p.numel() + 7) // 8 * 8
        else:
            v_parts.append(torch.zeros_like(p))
    return v_geom, 8, v_parts, [p.shape[-1] for p in parts]


class _Rasterize(torch.autograd.Function):
    """Tile compositing; stands in for gsplat ``rasterize_to_pixels``.  The composited channels are the
    concatenation of up to three tensors (``part0 | part1 | part2``, each [C,N,w] or -- shared by all cameras --
    [N,w]); the kernels gather from them in place and scatter the gradients into one buffer per part."""

    @staticmethod
    def forward(ctx, means2d, conics, part0, part1, part2, opacities, backgrounds, width, height, tile_size,
                isect_offsets, flatten_ids, absgrad, ed_channel, tile_order):
        lib = _lib.load()
        C, N = opacities.shape
        dev = means2d.device
        parts, items = _segments([p for p in (part0, part1, part2) if p is not None], C, N)
        D = sum(w for _, w, _, _ in items)
        means2d_c, conics, opacities = means2d.contiguous(), conics.contiguous(), opacities.contiguous()
        backgrounds = None if backgrounds is None else backgrounds.contiguous()
        th, tw = isect_offsets.shape[1:]
        render_colors = torch.empty(C, height, width, D, dtype=torch.float32, device=dev)
        render_alphas = torch.empty(C, height, width, 1, dtype=torch.float32, device=dev)
        last_ids = torch.empty(C, height, width, dtype=torch.int32, device=dev)
        if tile_order is None:
            tile_order = _tile_order(isect_offsets, flatten_ids.numel())
        single = len(parts) == 1 and not items[0][3]
        keep, segs = (None, None) if single else _lib.segment_array(items)
        with torch.cuda.device(dev):
            _call("pr_raster_fwd", C, N, D, _ptr(means2d_c), _ptr(conics), _ptr(parts[0]) if single else None,
                  _ptr(opacities), _ptr(backgrounds), width, height, tile_size, tw, th, _ptr(isect_offsets),
                  _ptr(tile_order), _ptr(flatten_ids), flatten_ids.numel(), ed_channel, _ptr(render_colors),
                  _ptr(render_alphas), _ptr(last_ids), segs, 0 if single else len(items), _stream())
        del keep
        ctx.save_for_backward(means2d_c, conics, opacities, backgrounds, isect_offsets, flatten_ids,
                              render_alphas, last_ids, render_colors if ed_channel >= 0 else None, tile_order,
                              *parts)
        ctx.args = (width, height, tile_size, absgrad, ed_channel, items,
                    [i for i, p in enumerate((part0, part1, part2)) if p is not None])
        ctx.means2d_obj = means2d  # the object callers hold in meta["means2d"]; receives .absgrad
        ctx.set_materialize_grads(False)
        return render_colors, render_alphas

    @staticmethod
    def backward(ctx, v_render_colors, v_render_alphas):
        lib = _lib.load()
        (means2d, conics, opacities, backgrounds, isect_offsets, flatten_ids, render_alphas, last_ids,
         render_colors, tile_order, *parts) = ctx.saved_tensors
        width, height, tile_size, absgrad, ed_channel, items, slots = ctx.args
        C, N = opacities.shape
        D = sum(w for _, w, _, _ in items)
        dev = means2d.device
        th, tw = isect_offsets.shape[1:]
        if v_render_colors is None:
            v_render_colors = torch.zeros(C, height, width, D, dtype=torch.float32, device=dev)
        if v_render_alphas is None:
            v_render_alphas = torch.zeros(C, height, width, 1, dtype=torch.float32, device=dev)
        v_render_colors, v_render_alphas = v_render_colors.contiguous(), v_render_alphas.contiguous()
        geom = ctx.needs_input_grad[0] or ctx.needs_input_grad[1] or ctx.needs_input_grad[5]
        want = [bool(ctx.needs_input_grad[2 + slot]) for slot in slots]
        # The kernel accumulates into zero-filled buffers.  For a large model filling them is 20-40 us of pure memory
        # traffic in front of the backward kernel.  The host usually reaches this point while the GPU is still in the
        # forward compositor (bound by vector instructions): issued on the second stream the fills run under it.
        n_bytes = 4 * ((C * N * 8 if geom else 0) + sum(p.numel() for p, w in zip(parts, want) if w))
        if n_bytes >= _EARLY_FILL_MIN_BYTES:
            main = torch.cuda.current_stream(dev)
            side = _state.side_stream(dev, main)
            with torch.cuda.stream(side):
                v_geom, geom_ld, v_parts, v_lds = _grad_targets(parts, want, geom, C, N, dev)
                filled = torch.cuda.Event()
                filled.record(side)
            main.wait_event(filled)
            for t in [v_geom] + v_parts:
                if t is not None:
                    t.record_stream(main)
        else:
            v_geom, geom_ld, v_parts, v_lds = _grad_targets(parts, want, geom, C, N, dev)
        if geom or any(v is not None for v in v_parts):
            single = len(parts) == 1 and not items[0][3] and v_lds[0] == items[0][1]
            keep_c, segs = (None, None) if single else _lib.segment_array(items)

            def launch(geom_flag, vg, vps):
                keep_g, vsegs = (None, None) if single else _lib.segment_array(
                    [(None if v is None else v.data_ptr(), w, vld, sh)
                     for v, vld, (_, w, _ld, sh) in zip(vps, v_lds, items)])
                with torch.cuda.device(dev):
                    _call("pr_raster_bwd", C, N, D, _ptr(means2d), _ptr(conics), _ptr(parts[0]) if single else None,
                          _ptr(opacities), _ptr(backgrounds), width, height, tile_size, tw, th, _ptr(isect_offsets),
                          _ptr(tile_order), _ptr(flatten_ids), flatten_ids.numel(), _ptr(render_alphas),
                          _ptr(last_ids), _ptr(render_colors), ed_channel, _ptr(v_render_colors),
                          _ptr(v_render_alphas), int(absgrad), int(geom_flag), _ptr(vg),
                          geom_ld, _ptr(vps[0]) if single else None, segs, vsegs,
                          0 if single else len(items), _stream())
                del keep_g

            if geom and any(v is not None for v in v_parts) and _split_backward(D, tile_size):
                # The geometry gradients (bound by vector instructions) and the colour gradients (deferred matrix
                # product, bound by the float atomics that deliver it) as two kernels that run side by side.
                main = torch.cuda.current_stream(dev)
                side = _state.side_stream(dev, main)
                ready = torch.cuda.Event()
                ready.record(main)
                side.wait_event(ready)
                with torch.cuda.stream(side):
                    launch(False, None, v_parts)
                    done = torch.cuda.Event()
                    done.record(side)
                launch(True, v_geom, [None] * len(v_parts))
                main.wait_event(done)
            else:
                launch(geom, v_geom, v_parts)
            del keep_c
        v_means2d = v_conics = v_opacities = None
        if geom:
            g = v_geom
            v_means2d, v_conics, v_opacities = g[..., 0:2], g[..., 2:5], g[..., 5]
            if absgrad:
                ctx.means2d_obj.absgrad = g[..., 6:8]
        v_backgrounds = None
        if backgrounds is not None and ctx.needs_input_grad[6]:
            v_backgrounds = (v_render_colors * (1.0 - render_alphas)).sum(dim=(1, 2))
            if ed_channel >= 0:
                v_backgrounds[:, ed_channel] = (v_render_colors[..., ed_channel] * (1.0 - render_alphas[..., 0])
                                                / render_alphas[..., 0].clamp(min=1e-10)).sum(dim=(1, 2))
        ctx.means2d_obj = None
        v_slots = [None, None, None]
        for v, slot in zip(v_parts, slots):
            v_slots[slot] = v
        return (v_means2d if ctx.needs_input_grad[0] else None, v_conics if ctx.needs_input_grad[1] else None,
                v_slots[0], v_slots[1], v_slots[2], v_opacities if ctx.needs_input_grad[5] else None, v_backgrounds,
                None, None, None, None, None, None, None, None)


def rasterize_to_pixels(means2d: Tensor, conics: Tensor, colors: Tensor, opacities: Tensor, image_width: int,
                        image_height: int, tile_size: int, isect_offsets: Tensor, flatten_ids: Tensor,
                        backgrounds: Optional[Tensor] = None, absgrad: bool = False,
                        ed_channel: int = -1, tile_order: Optional[Tensor] = None,
                        more_colors: Tuple[Tensor, ...] = ()) -> Tuple[Tensor, Tensor]:
    """Stands in for gsplat ``rasterize_to_pixels`` (non-packed): [C,N,*] inputs ->
    render_colors [C,H,W,D], render_alphas [C,H,W,1].  Extensions: ``ed_channel``, index of a depth
    channel to output as expected depth, accumulated depth / max(alpha, 1e-10), -1 = none; ``tile_order``,
    launch order of the tiles as ``_bin_tiles`` returns it, derived here when not given; ``more_colors``, up to
    two further tensors ([C,N,w], or [N,w] shared by all cameras) whose channels follow those of ``colors``
    -- composited in the same pass without being concatenated."""
    parts = (colors,) + tuple(more_colors) + (None, None)
    assert len(more_colors) <= 2
    return _Rasterize.apply(means2d, conics, parts[0], parts[1], parts[2], opacities, backgrounds, image_width,
                            image_height, tile_size, isect_offsets, flatten_ids, absgrad, int(ed_channel), tile_order)


def fully_fused_projection(means, quats, scales, viewmats, Ks, width, height, eps2d=0.3, near_plane=0.01,
                           far_plane=1e10, radius_clip=0.0, calc_compensations=False):
    """Stands in for gsplat ``fully_fused_projection`` (non-packed, quats+scales form)."""
    return _Projection.apply(means, quats, scales, viewmats, Ks, int(width), int(height), float(eps2d),
                             float(near_plane), float(far_plane), float(radius_clip), bool(calc_compensations))


class _CameraCenters(torch.autograd.Function):
    """c = inverse(V)[:, :3, 3] for rigid world->camera matrices V.  Forward uses the closed form -R^T t;
    backward is the derivative of the general matrix inverse, dM = -M dV M, so gradients equal those of
    ``torch.inverse(viewmats)[:, :3, 3]`` (what gsplat's ``rasterization`` differentiates through).  One small
    kernel each way (``pr_campos_fwd`` / ``pr_campos_bwd``)."""

    @staticmethod
    def forward(ctx, viewmats):
        viewmats = viewmats.contiguous()
        C = viewmats.shape[0]
        campos = torch.empty(C, 3, dtype=torch.float32, device=viewmats.device)
        with torch.cuda.device(viewmats.device):
            _call("pr_campos_fwd", C, _ptr(viewmats), _ptr(campos), _stream())
        ctx.save_for_backward(viewmats)
        return campos

    @staticmethod
    def backward(ctx, v_c):
        (viewmats,) = ctx.saved_tensors
        C = viewmats.shape[0]
        v_viewmats = torch.empty_like(viewmats)
        with torch.cuda.device(viewmats.device):
            _call("pr_campos_bwd", C, _ptr(viewmats), _ptr(v_c.contiguous()), _ptr(v_viewmats), _stream())
        return v_viewmats


def camera_centers(viewmats: Tensor) -> Tensor:
    """World-space camera centres of rigid world->camera matrices (the matrices ``get_viewmat`` builds,
    ``pogs/pogs.py:75-90``)."""
    return _CameraCenters.apply(viewmats)


def rasterization(
    means: Tensor, quats: Tensor, scales: Tensor, opacities: Tensor, colors: Tensor, viewmats: Tensor, Ks: Tensor,
    width: int, height: int, near_plane: float = 0.01, far_plane: float = 1e10, radius_clip: float = 0.0,
    eps2d: float = 0.3, sh_degree: Optional[int] = None, packed: bool = True, tile_size: int = 16,
    backgrounds: Optional[Tensor] = None, render_mode: str = "RGB", sparse_grad: bool = False,
    absgrad: bool = False, rasterize_mode: str = "classic", channel_chunk: int = 32, distributed: bool = False,
    camera_model: str = "pinhole", covars: Optional[Tensor] = None, features: Optional[Tensor] = None,
    rigid: Optional[Tuple[Tensor, Tensor, Tensor]] = None, depth_order=None, camera_groups: Optional[Tensor] = None,
    binning: Optional[Dict] = None,
) -> Tuple[Tensor, Tensor, Dict]:
    """Rasterize N 3-D Gaussians to C images; gsplat 1.4.0 ``rasterization`` signature.

    Supported configuration = the one POGS uses: dense (``packed`` is accepted and ignored: results and
    ``meta`` are always in the non-packed [C,N,...] layout the reference requests with ``packed=False``),
    ``sparse_grad=False``, ``distributed=False``, pinhole cameras, quats+scales (``covars=None``).
    ``channel_chunk`` is accepted for signature compatibility; channel tiling is chosen inside the library.

    Extensions beyond gsplat: ``features`` (see the module docstring) and ``rigid=(group_labels [N] int32,
    init_p2ws [G,7], p_deltas [G,7])``: the model is frozen and every Gaussian is moved by its group's
    ``p2w * delta * inverse(p2w)`` inside the projection kernels (the tracker's ``apply_to_model`` without its pass
    over the model, forward or backward); gradients reach ``p_deltas`` and the view matrices only.
    ``camera_groups`` ([C] integer, with ``rigid``): camera c sees only the Gaussians of rigid group
    ``camera_groups[c]`` -- the tracker's per-object renders (``obj_id`` gather, ``pogs/pogs.py:287-300``, one crop
    camera per object) as ONE call over C crop cameras of a common image size instead of C calls on C subsets.
    ``depth_order``: ``meta["depth_order"]`` of an earlier call on the SAME Gaussian positions and view matrices: the
    front-to-back order of the Gaussians depends on nothing else, so this call skips its depth sort.  An order produced
    WITHOUT ``rigid`` is built from the depths of all Gaussians, culled or not, and serves renders with any intrinsics,
    image size, tile size and channels (POGS renders three per training iteration: full-resolution colours, the
    half-resolution 96 channels, the 450x250 DINO channels); one produced WITH ``rigid`` comes from the projection's
    depths, where a culled Gaussian counts as depth 0, and serves calls with the same camera and image size only (the
    tracker's two renders per iteration).  The caller vouches for the sameness; nothing checks it.

    ``binning``: ``meta["binning"]`` of an earlier call that rendered the SAME Gaussian positions, shapes and opacities
    through the same cameras at the same image and tile size: this call composites over that call's per-tile lists and
    runs neither the depth sort nor the tile binning (nor waits for the intersection count).  Channels, appearance,
    render mode and which inputs carry gradients may differ (the tracker's feature render after its RGB+ED render).
    Image and tile size are checked, the rest is the caller's promise.
    """
    N = means.shape[0]
    C = viewmats.shape[0]
    assert means.shape == (N, 3), means.shape
    assert quats.shape == (N, 4), quats.shape
    assert scales.shape == (N, 3), scales.shape
    assert opacities.shape == (N,), opacities.shape
    assert viewmats.shape == (C, 4, 4), viewmats.shape
    assert Ks.shape == (C, 3, 3), Ks.shape
    assert render_mode in ["RGB", "D", "ED", "RGB+D", "RGB+ED"], render_mode
    assert rasterize_mode in ["classic", "antialiased"], rasterize_mode
    if covars is not None:
        raise NotImplementedError("covars input is outside the POGS path (quats+scales only)")
    if sparse_grad or distributed:
        raise NotImplementedError("sparse_grad / distributed modes are outside the POGS path")
    if camera_model != "pinhole":
        raise NotImplementedError("only pinhole cameras are on the POGS path")
    if sh_degree is None:
        assert (colors.dim() == 2 and colors.shape[0] == N) or (
            colors.dim() == 3 and colors.shape[:2] == (C, N)), colors.shape
    else:
        assert (colors.dim() == 3 and colors.shape[0] == N and colors.shape[2] == 3) or (
            colors.dim() == 4 and colors.shape[:2] == (C, N) and colors.shape[3] == 3), colors.shape
        assert (sh_degree + 1) ** 2 <= colors.shape[-2], colors.shape
    if backgrounds is not None:
        assert backgrounds.dim() == 2 and backgrounds.shape[0] == C, backgrounds.shape
    _check_device(means, quats, scales, opacities, colors, viewmats, Ks, backgrounds, features)
    width, height = int(width), int(height)

    # Parameters whose multi-GPU exchange of the previous step is still in flight (pogs_amd.dist.ShardedExchange)
    # are waited for on the device, each right before the first kernel that reads it.
    for t in (means, quats, scales, opacities, viewmats, Ks, backgrounds):
        wait_ready(t)
    if binning is not None:
        assert (binning["tile_size"], binning["width"], binning["height"], binning["n_cameras"]) == (
            tile_size, width, height, C), "binning= comes from a render of the same cameras, image size and tile size"
    sort_async = (binning is None and depth_order is None and not _TILE_LISTS and C * N >= _ASYNC_DEPTH_ORDER_MIN)
    if rigid is None:
        if sort_async:      # keys from the positions and the view matrices alone: the sort starts next to the projection
            depth_order = _depth_order_async(None, means, viewmats)
        radii, means2d, depths, conics, compensations = fully_fused_projection(
            means, quats, scales, viewmats, Ks, width, height, eps2d, near_plane, far_plane, radius_clip,
            calc_compensations=(rasterize_mode == "antialiased"))
    else:
        labels, init_p2ws, p_deltas = rigid
        G = p_deltas.shape[0]
        assert labels.shape == (N,) and labels.dtype == torch.int32, (labels.shape, labels.dtype)
        assert init_p2ws.shape == (G, 7) and p_deltas.shape == (G, 7), (init_p2ws.shape, p_deltas.shape)
        _check_device(init_p2ws, p_deltas)
        # `means` from here on: the moved means (what the view directions of the colours are taken from)
        radii, means2d, depths, conics, compensations, means = _ProjectionRigid.apply(
            means.detach(), quats.detach(), scales.detach(), labels, init_p2ws, p_deltas, viewmats, Ks, width, height,
            float(eps2d), float(near_plane), float(far_plane), float(radius_clip), rasterize_mode == "antialiased")
    # the two-level device sort runs its depth level on a second stream for large models
    if camera_groups is not None:
        assert rigid is not None and camera_groups.shape == (C,), "camera_groups goes with rigid=(labels, ...)"
        # a Gaussian of another group is treated as culled from here on (binning, colours, compositor); the projection's
        # own backward still visits it and finds zero cotangents
        radii = torch.where(rigid[0][None, :] == camera_groups.to(rigid[0].dtype)[:, None], radii,
                            torch.zeros_like(radii))
    if depth_order is not None:
        assert not _TILE_LISTS and depth_order[0].shape == (C * N,) and depth_order[0].device == depths.device
    elif sort_async:
        # (with rigid groups the positions are the moved ones, which only the projection knows: its depths then)
        depth_order = _depth_order_async(depths)
    opac = opacities[None, :].expand(C, N)
    if compensations is not None:
        opac = opac * compensations

    if features is not None:
        assert features.dim() == 2 and features.shape[0] == N, features.shape
    with_depth = render_mode in ("RGB+D", "RGB+ED", "D", "ED")
    depth_only = render_mode in ("D", "ED")
    use_sh = sh_degree is not None and not depth_only
    if use_sh and colors.dim() == 4:
        if C != 1:
            raise NotImplementedError("per-camera SH coefficients with C > 1")
        colors = colors[0]

    def evaluate_colours():
        wait_ready(colors)
        if use_sh:
            return _SphericalHarmonics.apply(int(sh_degree), means, camera_centers(viewmats), colors, radii, True)
        return colors                       # [C,N,D], or [N,D] shared by all cameras

    # The view-dependent colours normally run next to the depth sort; coefficients still on the links are
    # evaluated after the binning instead, which needs the geometry only.
    late_colours = pending(colors)
    rgb = None if depth_only or late_colours else evaluate_colours()
    n_channels = ((0 if depth_only else (3 if use_sh else colors.shape[-1]) + (features.shape[-1] if features is not None else 0))
                  + (1 if with_depth else 0))
    bgs = backgrounds
    if bgs is not None:
        if depth_only:
            bgs = torch.zeros(C, 1, device=bgs.device, dtype=bgs.dtype)
        elif with_depth:
            bgs = torch.cat([bgs, torch.zeros(C, 1, device=bgs.device, dtype=bgs.dtype)], dim=-1)
        assert bgs.shape == (C, n_channels), bgs.shape

    tile_width = math.ceil(width / float(tile_size))
    tile_height = math.ceil(height / float(tile_size))
    opac = opac.contiguous()
    # compositor lists: only the (Gaussian, tile) pairs that can contribute
    if binning is not None:
        flatten_ids, isect_offsets, tile_order = binning["flatten_ids"], binning["isect_offsets"], binning["tile_order"]
    elif depth_order is None and _TILE_LISTS:
        flatten_ids, isect_offsets, tile_order = _tile_lists(means2d, radii, depths, conics, opac, tile_size,
                                                             tile_width, tile_height)
    else:
        _, _, flatten_ids, isect_offsets, tile_order = _bin_tiles(
            means2d, radii, depths, tile_size, tile_width, tile_height, False, True, True, conics=conics,
            opacities=opac, depth_order=depth_order)

    # channels = [rgb or given colours | features | depth]: the compositor gathers from the tensors in place
    parts = []
    if not depth_only:
        parts.append(evaluate_colours() if late_colours else rgb)
        if features is not None:
            wait_ready(features)
            parts.append(features)          # [N,F], shared
    if with_depth:
        parts.append(depths[..., None])     # [C,N,1]
    assert n_channels == sum(p.shape[-1] for p in parts)

    # expected depth (ED): the depth channel is divided by max(alpha, 1e-10) inside the kernels
    ed_channel = n_channels - 1 if render_mode in ("ED", "RGB+ED") else -1
    render_colors, render_alphas = rasterize_to_pixels(
        means2d, conics, parts[0], opac, width, height, tile_size, isect_offsets, flatten_ids, backgrounds=bgs,
        absgrad=absgrad, ed_channel=ed_channel, tile_order=tile_order, more_colors=tuple(parts[1:]))

    meta = RenderMeta({
        "camera_ids": None, "gaussian_ids": None, "radii": radii, "means2d": means2d, "depths": depths,
        "conics": conics, "opacities": opac, "tile_width": tile_width, "tile_height": tile_height,
        "width": width, "height": height, "tile_size": tile_size, "n_cameras": C,
        "n_composited_isects": flatten_ids.numel(), "depth_order": depth_order,
        "binning": {"flatten_ids": flatten_ids, "isect_offsets": isect_offsets, "tile_order": tile_order,
                    "tile_size": tile_size, "width": width, "height": height, "n_cameras": C},
    }, _ReferenceBinning(means2d, radii, depths, tile_size, tile_width, tile_height))
    return render_colors, render_alphas, meta
