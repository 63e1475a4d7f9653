"""hipGraph capture of the shape-static, launch-bound chains of the step.

The frozen Inception-v3 encoder of the DAMSM loss (reference image_generation/model.py:182-287, used by
miscc/losses.py:421) is ~100 convolutions on 35 x 35 .. 8 x 8 maps plus pools and concatenations: ~330 launches forward and
~420 backward per training step, each 5 .. 40 us of device time -- the host needs longer to issue one than the device to run
it, and the generator's backward pass cannot start before the chain is through.  Its shapes never change (batch x 3 x 256 x
256 in, 17 x 17 region features and a sentence code out) and its weights are frozen, so the whole chain is captured ONCE
into two hipGraphs (forward; backward w.r.t. the image) and replayed with one launch each.

`GraphedCallable(fn)` wraps a function of ONE tensor -> tuple of tensors whose every kernel launch goes to torch's current
stream (the ctypes launches of objgan_hip.ops do: `_stream()` reads the raw current stream, which is the capture stream
while a capture is running) and whose host code has no device->host sync.  Per (input shape, gradient wanted, weight
version) it keeps the graphs, the static input / output / gradient buffers and the private memory pool the captured
kernels' intermediates live in.  Same kernels, same arguments, same order as the eager chain: results are bit-identical
(tests/test_modules_gpu.py::test_graphed_encoder_is_bit_identical_to_the_eager_chain).

Anything that cannot be captured falls back to the eager chain LOUDLY (one warning with the reason), never silently to a
different arithmetic: the fallback is the same kernels issued one by one.
"""
import warnings

import torch

from . import ops

# switch for A/B runs and for bench.py's per-kernel timing pass (hipEvents around every convolution launch cannot be
# recorded from inside a replayed graph): OBJGAN_GRAPHS=0 or graphs.enable(False)
import os as _os
_STATE = {"on": _os.environ.get("OBJGAN_GRAPHS", "1") != "0", "captures": 0, "replays": 0, "fallbacks": 0}


def enable(on):
    _STATE["on"] = bool(on)


def enabled():
    return _STATE["on"]


def stats():
    return dict(_STATE)


class _Captured(object):
    __slots__ = ("fwd", "bwd", "static_in", "static_outs", "static_gouts", "static_gin", "pool", "fn")


class GraphedCallable(object):
    """fn: tensor [fixed shape] -> tuple of tensors.  `version` (optional callable -> hashable) says when the captured
    graphs are stale (e.g. the `_version` counters of frozen weights that `load_state_dict` bumps)."""

    def __init__(self, fn, version=None, name="graph", warmup=2):
        self.fn, self.version, self.name, self.warmup = fn, version, name, int(warmup)
        self._graphs = {}
        self._broken = False

    # attribute access falls through to the wrapped callable (state_dict(), parameters(), eval(), .nef ...)
    def __getattr__(self, k):
        return getattr(self.__dict__["fn"], k)

    def __call__(self, x):
        want_grad = bool(torch.is_grad_enabled() and x.requires_grad)
        if (not _STATE["on"] or self._broken or not x.is_cuda or ops._H2_CENSUS is not None or ops._H2_VERIFY["on"]
                or torch.cuda.is_current_stream_capturing()):
            return self.fn(x)
        key = (tuple(x.shape), x.dtype, want_grad, ops.get_conv_math(), ops._REC["on"],
               self.version() if self.version is not None else None)
        cap = self._graphs.get(key)
        if cap is None:
            try:
                cap = self._capture(x, want_grad)
            except Exception as e:                       # noqa: BLE001  (capture is an optimisation: fall back, say why)
                self._broken = True
                _STATE["fallbacks"] += 1
                warnings.warn("objgan_hip.graphs: %s could not be captured (%s: %s); running it eagerly"
                              % (self.name, type(e).__name__, e))
                return self.fn(x)
            if len(self._graphs) >= 4:                   # (a new weight version / shape: drop the old pools)
                self._graphs.clear()
            self._graphs[key] = cap
        _STATE["replays"] += 1
        if not want_grad:
            cap.static_in.copy_(x)
            cap.fwd.replay()
            return tuple(o for o in cap.static_outs)
        return _Replay.apply(cap, x)

    def _capture(self, x, want_grad):
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream(device=x.device)
        side.wait_stream(cur)
        cap = _Captured()
        cap.fn = self.fn
        with torch.cuda.stream(side):
            static_in = x.detach().clone()
            if want_grad:
                static_in.requires_grad_(True)
            # eager warm-up on the side stream: filter banks get packed, launch plans and tap tables memoised, the
            # allocator sees the sizes -- nothing of that may happen for the first time inside the capture
            for _ in range(max(1, self.warmup)):
                with torch.set_grad_enabled(want_grad):
                    outs = self.fn(static_in)
                outs = outs if isinstance(outs, (tuple, list)) else (outs,)
                if want_grad:
                    torch.autograd.grad(outs, (static_in,), tuple(torch.ones_like(o) for o in outs))
                del outs
        cur.wait_stream(side)
        torch.cuda.synchronize(x.device)
        cap.pool = torch.cuda.graph_pool_handle()
        cap.static_in = static_in
        ops.capture_begin()
        try:
            cap.fwd = torch.cuda.CUDAGraph()
            with torch.cuda.graph(cap.fwd, pool=cap.pool):
                with torch.set_grad_enabled(want_grad):
                    outs = self.fn(static_in)
                outs = tuple(outs) if isinstance(outs, (tuple, list)) else (outs,)
            cap.static_outs = outs
            cap.bwd = cap.static_gouts = cap.static_gin = None
            if want_grad:
                cap.static_gouts = tuple(torch.zeros_like(o) for o in outs)
                cap.bwd = torch.cuda.CUDAGraph()
                with torch.cuda.graph(cap.bwd, pool=cap.pool):
                    (cap.static_gin,) = torch.autograd.grad(outs, (static_in,), cap.static_gouts)
                cap.static_outs = tuple(o.detach() for o in outs)
        finally:
            ops.capture_end()
        _STATE["captures"] += 1
        return cap


class _Replay(torch.autograd.Function):
    """autograd node of one captured forward / backward pair (the scheme of torch.cuda.make_graphed_callables: static
    buffers in, replay, static buffers out; the consumer is done with an output before the next replay overwrites it --
    one use per training step, on one stream)"""

    @staticmethod
    def forward(ctx, cap, x):
        cap.static_in.detach().copy_(x)
        cap.fwd.replay()
        ctx.cap = cap
        return tuple(o.detach() for o in cap.static_outs)

    @staticmethod
    def backward(ctx, *grads):
        cap = ctx.cap
        for buf, g in zip(cap.static_gouts, grads):
            if g is None:
                buf.zero_()
            else:
                buf.copy_(g)
        cap.bwd.replay()
        return None, cap.static_gin.detach()


def graphed_encoder(enc, name="CNN_ENCODER"):
    """the frozen image encoder behind a GraphedCallable; stale when any of its tensors is edited in place / reloaded"""
    params = list(enc.parameters()) + list(enc.buffers())

    def version():
        return (sum(p._version for p in params), enc.training)
    return GraphedCallable(enc, version=version, name=name)
