"""Static-shape forward + backward of one view captured in ONE hipGraph.

A training step launches ~45 kernels; at low resolution (the reference trains at 1/4 and 1/2
resolution for its first 6000 steps, freegaussian_model.py:626-633) the host cannot issue them as
fast as the GPU retires them.  ``GraphedRaster`` fixes everything that is data-dependent on the
host side -- the intersection lists get a fixed capacity, the count stays on the device
(``fg_bin_emit_sort_capacity``) -- captures rasterization + its backward once and then replays the
graph per step: one launch, no host wait.  Gradients land in the flat buffer of a
``viewdp.FlatGaussianParams`` exactly as in the eager path.

Caller's duty (a PyTorch rule for capturing a backward): no autograd graph built OUTSIDE the capture
may still reference the parameters when the capture starts (drop old ``render`` / ``info``
objects first) -- their AccumulateGrad nodes would run on the stream they were created on, not on
the capture stream, and the capture aborts.

If a replay finds more intersections than the capacity, its lists were truncated: ``step`` then
reports ``overflow=True``, re-runs the step eagerly (exact) and re-captures with a larger capacity."""
from __future__ import annotations

from typing import Optional

import torch

from . import ops
from .rasterization import rasterization
from .viewdp import FlatGaussianParams


class GraphedRaster:
    def __init__(self, params: FlatGaussianParams, width: int, height: int, sh_degree: int = 3,
                 render_mode: str = "RGB", capacity: Optional[int] = None, headroom: float = 1.5,
                 ctx: Optional[ops.RasterContext] = None):  # fmt: skip
        self.params, self.width, self.height = params, int(width), int(height)
        self.sh_degree, self.render_mode, self.headroom = sh_degree, render_mode, float(headroom)
        dev = params.flat.device
        ch = 3 + int(render_mode.endswith("D"))
        self.viewmat = torch.zeros(1, 4, 4, device=dev)
        self.K = torch.zeros(1, 3, 3, device=dev)
        self.v_render = torch.zeros(1, height, width, ch, device=dev)
        self.capacity = capacity
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.render = self.alpha = self.overflow = None
        self.ctx = ctx if ctx is not None else ops.current()  # launch policy, list capacities, hooks

    # one eager (exact) step; also measures the list length for the capacity
    def _eager(self):
        with ops.use(self.ctx), self.params.direct_grads():
            r, a, info = rasterization(*self.params.raster_inputs(), self.viewmat, self.K, self.width, self.height,
                                       sh_degree=self.sh_degree, render_mode=self.render_mode, packed=False,
                                       absgrad=True)  # fmt: skip
            r.backward(self.v_render)
        return r.detach(), a.detach(), int(info["raster_flatten_ids"].numel())

    def _capture(self):
        import gc

        gc.collect()  # drop unreachable autograd graphs (reference cycles) that still hold the parameters
        self.ctx.static_capacity = self.capacity
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):  # warm-up on a side stream, as torch's graph recipe asks
                for _ in range(2):
                    self._run_static()
            torch.cuda.current_stream().wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.render, self.alpha = self._run_static()
                self.overflow = self.ctx.last_overflow
        finally:
            self.ctx.static_capacity = None

    def _run_static(self):
        with ops.use(self.ctx), self.params.direct_grads():
            r, a, _ = rasterization(*self.params.raster_inputs(), self.viewmat, self.K, self.width, self.height,
                                    sh_degree=self.sh_degree, render_mode=self.render_mode, packed=False,
                                    absgrad=True)  # fmt: skip
            r.backward(self.v_render)
        return r.detach(), a.detach()

    def step(self, viewmat: torch.Tensor, K: torch.Tensor, v_render: torch.Tensor, check: bool = True):
        """One forward + backward for this view with the upstream gradient ``v_render``.
        -> (render [1,H,W,C], alpha [1,H,W,1], overflow: bool).  Outputs are static buffers
        (overwritten by the next step); gradients are in ``params.flat_grad``.  ``check=False``
        skips the 1-byte overflow readback (the caller then checks ``self.overflow`` itself)."""
        self.viewmat.copy_(viewmat.reshape(1, 4, 4), non_blocking=True)
        self.K.copy_(K.reshape(1, 3, 3), non_blocking=True)
        self.v_render.copy_(v_render, non_blocking=True)
        if self.graph is None:
            r, a, n = self._eager()
            if self.capacity is None or n > self.capacity:
                self.capacity = int(n * self.headroom) + 4096
            self._capture()
            return r, a, False
        self.graph.replay()
        if check and bool(self.overflow.item()):
            r, a, n = self._eager()  # exact redo of this step, then a bigger graph
            self.capacity = int(n * self.headroom) + 4096
            self._capture()
            return r, a, True
        return self.render, self.alpha, False
