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


class GraphedModelStep:
    """``FreeGaussianModel.get_outputs`` + loss + backward as ONE hipGraph, for the launch-bound sizes.

    The reference trains its first 6000 steps at 1/4 and 1/2 resolution (freegaussian_model.py:626-633,
    :807-815); there a step is ~60 launches of a few microseconds each and the eager host cannot keep the GPU
    busy (profiles/r02_low_resolutions.md).  Everything between the host side of the camera and the gradients
    -- deform MLP, activations, raster forward, background, loss, the whole backward -- is captured once per
    SHAPE and replayed.  What makes a shape: the Gaussian count and parameter storage (densification
    re-allocates both), the scheduled resolution, the SH degree, whether the deform net is active, the render
    mode.  A replay that finds more list entries than its fixed capacity is redone (the graph is re-captured
    with more room and replayed: exact).  The optimizers, ``after_train_iter`` and ``refinement_after`` stay
    eager and see ordinary ``.grad`` tensors (static buffers the replay refills; never ``set_to_none`` them
    while a graph is live -- ``harness.train_step`` skips its ``zero_grad`` on graphed steps).

    Not graphed (falls back to ``model.get_outputs``): eval, crop boxes, images above ``max_tiles`` (the eager
    path is GPU-bound there and its speculative list capacity needs no re-capture), the stage-2 control model,
    depth output during training, and every step with the deform net active (``step >= warm_up``).

    Why no MLP and why the returned loss is recomputed eagerly: on this stack (ROCm 7.2, PyTorch 2.10) a
    ``hipMemsetAsync`` issued under stream capture does not become a node of the graph (found in round 2 with this
    library's own fill).  PyTorch's multi-block reductions (``mean`` / ``sum`` / ``max`` over more than one
    workgroup) zero their semaphore words with exactly such a memset; on replay the words hold whatever the graph's
    pool last put there and, once that is non-zero, every later replay returns one workgroup's partial result
    (measured: the in-graph SSIM of a converging image reads 0.0 from the 7th replay on, the eager value of the same
    static image 0.996).  The gradient of a captured ``mean`` is a broadcast and does not depend on the reduction's
    value, so the raster + L1/SSIM step is safe as a graph -- its gradients equal the eager step's to 1e-5 --, but
    the loss VALUE is not, and an MLP backward (bias gradients are reductions) is not either.  The deform phase of
    the reference (from step 3000, at 1/2 resolution) therefore stays eager; the 1/4-resolution phase, the most
    launch-bound one, has no deform net (``warm_up`` = 3000 = ``resolution_schedule``)."""

    def __init__(self, model, loss_fn=None, max_tiles: int = 2200, headroom: float = 1.5,
                 ctx: Optional[ops.RasterContext] = None):  # fmt: skip
        # loss_fn None (or harness.main_loss itself): the model's own objective, (1 - l) L1 + l (1 - SSIM) with l =
        # model.config.ssim_lambda -- the same one the eager branch of harness.train_step gets from get_loss_dict, so
        # the objective does not change where the schedule leaves the graphed resolutions
        self.model, self.loss_fn = model, loss_fn
        self.max_tiles, self.headroom = int(max_tiles), float(headroom)
        self.ctx = ctx if ctx is not None else ops.current()
        self.key = None
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.capacity: Optional[int] = None
        self.captures = 0  # (re-)captures so far: shape changes + overflows
        self.replays = 0
        self.static = {}

    def applicable(self, camera) -> bool:
        from .model import FreeGaussianModel

        m = self.model
        if not m.training or m.crop_box is not None or m.device.type != "cuda" or m.config.use_bilateral_grid:
            return False  # (the grid of a training image is picked on the host, per camera)
        if m.step >= m.config.warm_up or m._render_mode() != "RGB":
            return False  # torch reductions (MLP bias gradients, depth max) are not replay-safe here: see the class docstring
        if type(m)._get_outputs_on_active_rows is not FreeGaussianModel._get_outputs_on_active_rows:
            return False  # (the stage-2 model assembles its inputs differently)
        s = m._get_downscale_factor()
        w, h = int(camera.width * (1 / s)), int(camera.height * (1 / s))
        return ((w + 15) // 16) * ((h + 15) // 16) <= self.max_tiles

    def _key(self, W, H):
        m = self.model
        deg = min(m.step // m.config.sh_degree_interval, m.config.sh_degree)
        # the storage of EVERY Gaussian parameter: densification re-allocates all of them, the opacity reset
        # (freegaussian_model.py:475-490, `.data = clamp(...)`) only one -- a graph must never read a stale one
        ptrs = tuple(p.data_ptr() for p in m.gauss_params.values())
        return (m.num_points, ptrs, W, H, deg, m.step >= m.config.warm_up, m._render_mode(), m.config.background_color,
                float(m.config.ssim_lambda))  # (the lambda is baked into the captured loss kernels' arguments)

    def _loss(self, rgb, gt):
        from . import harness

        if self.loss_fn is None or self.loss_fn is harness.main_loss:
            return harness.main_loss(rgb, gt, self.model.config.ssim_lambda)
        return self.loss_fn(rgb, gt)

    def _forward_backward(self):
        m, st = self.model, self.static
        out = m._outputs_from(st["viewmat"], st["K"], st["W"], st["H"], st["times"])
        loss = self._loss(out["rgb"], st["gt"])
        loss.backward()
        return out, loss

    def _capture(self):
        import gc

        m = self.model
        self.graph = None
        self.static.pop("out", None)
        self.static.pop("loss", None)
        self.ctx.static_capacity = self.capacity
        try:
            # Warm-up and capture on ONE stream.  An AccumulateGrad node remembers the stream of the forward that
            # created it and lives as long as any autograd graph that reaches the parameter: warm-up iterations on
            # a different stream than the capture (torch's recipe) left such nodes behind through `model.xys`, the
            # captured backward then accumulated on THAT stream, and the replays went wrong after a few steps
            # (first visible in the loss scalars).  So: same stream, and nothing that holds a graph survives into
            # the capture.
            stream = torch.cuda.Stream()
            stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(stream), ops.use(self.ctx):
                for _ in range(2):
                    for p in m.parameters():
                        p.grad = None  # the captured backward allocates the .grad tensors from the graph's pool
                    out, loss = self._forward_backward()
                    del out, loss
                    m.xys = None
            for p in m.parameters():
                p.grad = None
            gc.collect()
            torch.cuda.current_stream().wait_stream(stream)
            g = torch.cuda.CUDAGraph()
            with ops.use(self.ctx), torch.cuda.graph(g, stream=stream):
                out, loss = self._forward_backward()
                self.static["overflow"] = self.ctx.last_overflow
            self.graph, self.static["out"], self.static["loss"] = g, out, loss
            self.captures += 1
        finally:
            self.ctx.static_capacity = None

    def step(self, camera, gt_image):
        """-> (outputs dict, loss): static tensors, refilled by the next step.  Gradients are in ``.grad``."""
        m, st = self.model, self.static
        viewmat, K, W, H = m._camera_setup(camera)
        gt = m.get_gt_img(gt_image)
        times = camera.times.to(m.device)
        key = self._key(W, H)
        if key != self.key or self.graph is None:
            # a new shape: an eager forward measures the list length, then the capture
            self.key, self.graph = key, None
            st.clear()
            st.update(viewmat=viewmat.clone(), K=K.clone(), times=times.clone(), gt=gt.clone(), W=W, H=H)
            n = self._measure()
            if self.capacity is None or n > self.capacity or 3 * n * self.headroom < self.capacity:
                self.capacity = int(n * self.headroom) + 4096
            self._capture()
        else:
            st["viewmat"].copy_(viewmat, non_blocking=True)
            st["K"].copy_(K, non_blocking=True)
            st["times"].copy_(times, non_blocking=True)
            st["gt"].copy_(gt, non_blocking=True)
        self.graph.replay()
        self.replays += 1
        if bool(st["overflow"].item()):  # the list did not fit: more room, capture again, replay (exact)
            self.capacity = int(self._measure() * self.headroom) + 4096
            self._capture()
            self.graph.replay()
        with torch.no_grad():  # the value, outside the graph (in-graph reductions are not replay-safe: docstring)
            loss = self._loss(st["out"]["rgb"], st["gt"])
        return st["out"], loss

    def release(self) -> None:
        """Drop the graph and its static buffers (the next applicable step captures afresh)."""
        self.graph, self.key = None, None
        self.static.clear()

    def _measure(self) -> int:
        """Length of the raster list for the static inputs (an eager, exact forward)."""
        m, st = self.model, self.static
        with torch.no_grad(), ops.use(self.ctx):
            m._outputs_from(st["viewmat"], st["K"], st["W"], st["H"], st["times"])
        return int(m.last_list_length)
