"""View-sharded data parallelism for the raster path: one process per GPU, every rank holds all
Gaussians and renders its own camera views; ONE collective per step sums the flat
Gaussian-parameter gradient buffer over RCCL/xGMI (``torch.distributed`` backend "nccl").

This is new capability, not parity: the reference is single-GPU in practice (scripts/run.sh:58)
and torch DDP cannot wrap its model because densification replaces every parameter tensor
(freegaussian_model.py:433-436, :514-515).  Hence a flat buffer that survives
re-allocation, an explicit all-reduce, and explicit all-reduces of the densification statistics
(``xys_grad_norm``, ``vis_counts`` summed, ``max_2Dsize`` max-reduced, freegaussian_model.py:379-392).

Sizing for xGMI (7 links x ~153 GB/s per GPU): 59 floats = 236 B per Gaussian, 236 MB at 1M
Gaussians, sent as one all-reduce so RCCL can use every link of the full mesh."""
from __future__ import annotations

import os
from typing import Dict, Optional, Tuple

import torch
import torch.distributed as dist

# (name, trailing shape) of the raster-ready Gaussian parameters, 59 floats per Gaussian
LAYOUT: Tuple[Tuple[str, Tuple[int, ...]], ...] = (
    ("means", (3,)),
    ("quats", (4,)),
    ("scales", (3,)),
    ("opacities", ()),
    ("colors", (16, 3)),
)
FLOATS_PER_GAUSSIAN = 59


def _collective_world(group=None):
    """(world size, take the collective code paths?).  The paths are taken when there is more than one rank -- or, with
    ``FG_DP_FORCE_COLLECTIVES=1`` and an initialised process group, on ONE rank too: every collective call site of this file
    then goes through the backend (a 1-rank ``nccl`` communicator exercises ProcessGroupNCCL's stream, work-handle and
    tensor-lifetime semantics on the one GPU a box has -- not RCCL's ring kernels or the links)."""
    if not (dist.is_available() and dist.is_initialized()):
        return 1, False
    world = dist.get_world_size(group)
    return world, (world > 1 or os.environ.get("FG_DP_FORCE_COLLECTIVES") == "1")


def _numel(shape):
    n = 1
    for s in shape:
        n *= s
    return n


class FlatGaussianParams:
    """All Gaussian parameters as views into one flat fp32 buffer, with ``.grad`` of every view
    pre-bound to the matching view of one flat gradient buffer (autograd accumulates in place),
    so the per-step collective is a single all-reduce with no packing copy."""

    def __init__(self, n: int, device, dtype=torch.float32):
        self.n = n
        self.flat = torch.zeros(n * FLOATS_PER_GAUSSIAN, device=device, dtype=dtype)
        self.flat_grad = torch.zeros_like(self.flat)
        self.params: Dict[str, torch.Tensor] = {}
        off = 0
        self.grad_views: Dict[str, torch.Tensor] = {}
        for name, shape in LAYOUT:
            cnt = n * _numel(shape)
            p = self.flat[off : off + cnt].view((n,) + shape).requires_grad_(True)
            self.grad_views[name] = self.flat_grad[off : off + cnt].view((n,) + shape)
            p.grad = self.grad_views[name]
            self.params[name] = p
            off += cnt
        assert off == self.flat.numel()
        self._by_ptr = {p.data_ptr(): self.grad_views[k] for k, p in self.params.items()}

    @classmethod
    def from_scene(cls, scene, device) -> "FlatGaussianParams":
        self = cls(scene.means.shape[0], device)
        with torch.no_grad():
            for name in self.params:
                self.params[name].copy_(getattr(scene, name).to(device))
        return self

    def raster_inputs(self):
        p = self.params
        return p["means"], p["quats"], p["scales"], p["opacities"], p["colors"]

    def zero_grad(self) -> None:
        self.flat_grad.zero_()
        for k, p in self.params.items():
            p.grad = self.grad_views[k]

    def direct_grads(self):
        """Context manager: while active, the fused raster backward writes the gradients of these
        parameters STRAIGHT into their slices of the flat buffer (``RasterContext.grad_alloc`` of the current context) and
        ``.grad`` is unbound first, so autograd adopts those slices instead of adding into them:
        no AccumulateGrad pass and no zeroing (the kernels overwrite densely).  Valid when every
        parameter is used by exactly one ``rasterization`` call per backward (one view per step per
        rank -- the bench / view-DP case); use plain ``zero_grad()`` accumulation otherwise."""
        import contextlib

        from . import ops

        @contextlib.contextmanager
        def cm():
            for p in self.params.values():
                p.grad = None
            rctx = ops.current()  # the context the rasterization calls inside the block will capture
            prev = rctx.grad_alloc
            def alloc(t):
                # a FRESH view object each time: autograd only adopts (instead of cloning) a
                # gradient tensor nobody else holds a reference to
                base = self._by_ptr.get(t.data_ptr())
                return None if base is None else base.view(base.shape)

            rctx.grad_alloc = alloc
            try:
                yield self
            finally:
                rctx.grad_alloc = prev

        return cm()

    def factored_exchange(self, average: bool = True, group=None, per_view_means: bool = False):
        """Context manager for one view-sharded step with the FACTORED gradient exchange
        (use instead of ``direct_grads()`` + ``all_reduce_grads()``; one ``rasterization`` call
        with SH colours per step per rank, like ``direct_grads``).

        The SH-coefficient gradient is 48 of the 59 floats per Gaussian, but per view it is a
        rank-1 product ``basis_k(direction) * g`` of the clamp-masked colour gradient ``g[N,3]``.
        Instead of all-reducing 236 B per Gaussian, ranks (1) all-gather ``g`` + their camera
        position (12 B per Gaussian per rank), (2) all-reduce only the 11 non-colour floats
        (44 B), and (3) every rank rebuilds ``sum_views basis (x) g`` locally with
        ``fg_sh_grad_accumulate`` -- 192 B per Gaussian that never cross xGMI.  At 8 ranks and 1M
        Gaussians that is 84 MB received + a 44 MB all-reduce instead of a 236 MB all-reduce.
        The all-gather is issued from inside the backward as soon as ``g`` exists -- with shared means that is
        right behind the raster backward, under the per-Gaussian backward -- and the local rebuild overlaps the
        small all-reduce.  On exit every ``.grad`` holds the (averaged)
        sum over ranks, exactly as after ``all_reduce_grads`` up to fp32 summation order.

        ``per_view_means=True``: the ranks render different positions of the same Gaussians
        (per-view deformation, freegaussian_model.py:832-845), so a direction cannot be rebuilt
        from shared means + a camera position; the unit view direction then travels with the colour
        gradient (24 B instead of 12 B per Gaussian per rank)."""
        import contextlib

        from . import _lib, ops

        world, collective = _collective_world(group)
        n = self.n
        state = {}
        # The 44-byte head (means, quats, scales, opacities gradients) is final range by range as the per-Gaussian backward
        # runs: with ``head_slices`` = k > 1 that backward is launched as k grids over ranges of N and the all-reduce of a
        # range's four blocks is issued -- from a side stream that waits for that launch alone -- while the later ranges are
        # still being computed, instead of one all-reduce behind the whole backward (FG_DP_HEAD_SLICES, default 4 when
        # collectives run; shared means only: with per-view means the head is accumulated by autograd afterwards).
        head_slices = int(os.environ.get("FG_DP_HEAD_SLICES", "4")) if (collective and not per_view_means) else 1
        state["head_works"] = []

        pf = 6 if per_view_means else 3
        stride = n * 6 if per_view_means else (n + 1) * 3

        def gather(payload):
            if collective:
                gathered = torch.empty(world * stride, device=payload.device, dtype=torch.float32)
                if dist.get_backend(group) == "nccl":
                    state["work"] = dist.all_gather_into_tensor(gathered, payload, group=group, async_op=True)
                else:
                    outs = list(gathered.view(world, -1).unbind(0))
                    state["work"] = dist.all_gather(outs, payload, group=group, async_op=True)
                state["gathered"] = gathered
            else:
                state["gathered"] = payload

        def reduce_head_slice(n0, n1, event):
            """All-reduce rows [n0, n1) of the four head blocks, ordered behind ``event`` (the launch that wrote them)."""
            views = [self.grad_views[k][n0:n1] for k in ("means", "quats", "scales", "opacities")]
            side = state.get("side")
            if side is None:
                side = state["side"] = torch.cuda.Stream(device=self.flat.device)
            with torch.cuda.stream(side):
                side.wait_event(event)
                if dist.get_backend(group) == "nccl":
                    # one group call for the four blocks (a launch per block would be serial latency on the ring)
                    with dist._coalescing_manager(group=group, device=self.flat.device, async_ops=True) as cm:
                        for v in views:
                            dist.all_reduce(v, op=dist.ReduceOp.SUM, group=group)
                    state["head_works"].append(cm)
                else:
                    state["head_works"] += [dist.all_reduce(v, op=dist.ReduceOp.SUM, group=group, async_op=True) for v in views]

        def sink(what, *a):
            """Protocol with ops: "view" (per-Gaussian forward: the view is known), "records" (raster backward: the
            record gradients are final), "alloc" / "ready" (per-Gaussian backward, around fg_preprocess_bwd_factored)."""
            if what == "slices":  # how many ranges of N the per-Gaussian backward should be launched in
                return head_slices
            if what == "slice":  # rows [n0, n1) of the head are final once the launch just enqueued has run
                n0, n1 = a
                ev = torch.cuda.Event()
                ev.record()
                reduce_head_slice(n0, n1, ev)
                return None
            if what == "view":  # shared means: the camera position can be put in place during the forward
                if not per_view_means:
                    viewmat, dev = a
                    state["payload"] = torch.empty(stride, device=dev, dtype=torch.float32)
                    vm = viewmat.reshape(-1, 4)[:3]
                    state["payload"][3 * n :] = -(vm[:, :3].T @ vm[:, 3])  # camera position of this rank's view
                    state["early"] = True
                return None
            if what == "records":
                # g = clamp-masked colour gradient is FINAL as soon as the raster backward is: the record's colour is
                # max(sh + 0.5, 0), so the mask is (colour > 0), and slots 8..10 of the gradient record are dL/dcolour.
                # Its all-gather starts here, under the per-Gaussian backward, instead of behind it.
                if state.get("early") and not state.get("sent"):
                    splats, v_splats = a
                    g = state["payload"][: 3 * n].view(n, 3)
                    torch.where(splats[:, 6:9] > 0, v_splats[:, 8:11], torch.zeros((), device=g.device), out=g)
                    gather(state["payload"])
                    state["sent"] = True
                return None
            if what == "alloc":
                if state.get("sent"):  # the payload is in flight: the kernel's own copy of g goes to a scratch array
                    return torch.empty(n, pf, device=self.flat.device, dtype=torch.float32)
                state["payload"] = torch.empty(stride, device=self.flat.device, dtype=torch.float32)
                return state["payload"][: pf * n].view(n, pf)
            _v_rgb, means, viewmat, sh_degree, colors = a
            state.update(means=means, sh_degree=int(sh_degree), k_stored=int(colors.shape[1]))
            if state.get("sent"):
                return None
            payload = state["payload"]
            if not per_view_means:
                vm = viewmat.reshape(-1, 4)[:3]
                payload[3 * n :] = -(vm[:, :3].T @ vm[:, 3])  # camera position of this rank's view
            gather(payload)

        @contextlib.contextmanager
        def cm():
            rctx = ops.current()
            prev_sink = rctx.color_grad_sink
            rctx.color_grad_sink = sink
            try:
                # shared means: the raster inputs ARE the parameters, gradients land directly in the flat
                # buffer.  Per-view means are computed from the parameters (deformation): ordinary
                # accumulation into the pre-bound .grad views -- the caller zero_grad()s first.
                with (contextlib.nullcontext(self) if per_view_means else self.direct_grads()):
                    yield self
            finally:
                rctx.color_grad_sink = prev_sink
            if "payload" not in state:
                raise RuntimeError("factored_exchange: no SH-coloured rasterization backward ran inside the context")
            rest = self.flat_grad[: 11 * n]
            work = None
            if collective and not state["head_works"]:
                work = dist.all_reduce(rest, op=dist.ReduceOp.SUM, group=group, async_op=True)
            if "work" in state:
                state["work"].wait()
            scale = 1.0 / world if average else 1.0
            lib = _lib.load()
            vc = self.grad_views["colors"]
            _lib.check(lib.fg_sh_grad_accumulate(
                n, world, state["sh_degree"], state["k_stored"], state["means"].detach().contiguous().data_ptr(),
                state["gathered"].data_ptr(), stride, pf, scale, vc.data_ptr(),
                torch.cuda.current_stream().cuda_stream), "fg_sh_grad_accumulate")  # fmt: skip
            self.params["colors"].grad = vc
            if work is not None:
                work.wait()
            for w in state["head_works"]:  # (issued on the side stream: wait() orders the CURRENT stream behind them)
                w.wait()
            if collective and average and world > 1:
                rest.div_(world)

        return cm()

    def all_reduce_grads(self, average: bool = True, group=None) -> None:
        """Sum (then average) the flat gradient over all ranks: the one exchange step of a
        view-sharded iteration."""
        world, collective = _collective_world(group)
        if not collective:
            return
        dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=group)
        if average and world > 1:
            self.flat_grad.div_(world)


def all_reduce_densify_stats(xys_grad_norm: torch.Tensor, vis_counts: torch.Tensor, max_2dsize: torch.Tensor,
                             group=None) -> None:  # fmt: skip
    """Keep the densification statistics identical on every rank (reference accumulates them
    per step at freegaussian_model.py:379-392): sums for the gradient norm and visibility count,
    max for the largest screen-space radius.  In place."""
    if not _collective_world(group)[1]:
        return
    packed = torch.stack([xys_grad_norm.float(), vis_counts.float()])
    dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
    xys_grad_norm.copy_(packed[0])
    vis_counts.copy_(packed[1].to(vis_counts.dtype))
    dist.all_reduce(max_2dsize, op=dist.ReduceOp.MAX, group=group)


def shared_seed(seed: Optional[int] = None, group=None) -> int:
    """Broadcast rank 0's seed so that split_gaussians' ``torch.randn`` (freegaussian_model.py:530)
    and the random background (:651) draw the same numbers on every replica."""
    if seed is None:
        seed = int(torch.seed() % (2**31))
    if _collective_world(group)[1]:
        backend = dist.get_backend(group)
        dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
        t = torch.tensor([seed], dtype=torch.int64, device=dev)
        dist.broadcast(t, src=0, group=group)
        seed = int(t.item())
    torch.manual_seed(seed)
    return seed


def views_for_rank(n_views: int, rank: int, world: int):
    """Round-robin shard of camera views: rank r renders views r, r+world, ..."""
    return list(range(rank, n_views, world))


def all_reduce_model_grads(model: torch.nn.Module, average: bool = True, group=None) -> None:
    """View-DP gradient exchange for a model whose parameters are NOT views of one flat buffer
    (``FreeGaussianModel``: densification re-allocates its ParameterDict): every existing ``.grad``
    -- Gaussian parameters and the deform / control MLPs -- goes through ONE all-reduce of a
    flattened copy.  Parameters without a gradient on this rank (e.g. nothing visible) are
    treated as zeros so that all ranks reduce the same layout."""
    world, collective = _collective_world(group)
    if not collective:
        return
    params = [p for p in model.parameters() if p.requires_grad]
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average and world > 1:
        flat.div_(world)
    off = 0
    for p in params:
        n = p.grad.numel()
        p.grad.copy_(flat[off : off + n].view_as(p.grad))
        off += n


class ModelViewDP:
    """View-sharded data parallelism for ``FreeGaussianModel`` with the FACTORED gradient exchange (the training path's
    counterpart of ``FlatGaussianParams.factored_exchange``; the reference's hook is the DDP pass-through of
    freegaussian_pipeline.py:36-40, :62, which cannot wrap a model whose parameters densification replaces).

    Per step and rank, instead of one all-reduce of every gradient (236 B per Gaussian + the MLPs, after a ``torch.cat``
    and before a copy back):
      * the two SH-coefficient gradients (``features_dc`` 12 B + ``features_rest`` 180 B per Gaussian) never cross the
        links: the fused backward writes the clamp-masked colour gradient g (12 B) -- plus the unit view direction (12 B)
        once the deformation gives every rank its own means (freegaussian_model.py:832-845) -- into a payload that is
        ALL-GATHERED from inside the backward, and every rank rebuilds sum_views basis(direction) (x) g itself
        (``fg_sh_grad_accumulate_split``, straight into the two ``.grad`` tensors);
      * everything else -- means, scales, quats, opacities (44 B per Gaussian) and the MLP gradients -- is ONE all-reduce
        of a flat buffer the ``.grad`` tensors are views of (bound before the backward: autograd accumulates in place, no
        cat, no copy back), re-bound whenever densification has replaced the parameters.
    At 8 ranks and 1M Gaussians a rank receives 7 x 24 MB and takes part in a 46 MB all-reduce instead of a 238 MB one.
    Usage: ``harness.train_step(..., dp=ModelViewDP(model), stats_sync=sync_densify_stats)``."""

    COLOUR = ("features_dc", "features_rest")

    def __init__(self, model: torch.nn.Module, average: bool = True, group=None, sparse: Optional[str] = None):
        """``sparse`` (default: FG_DP_SPARSE, else "auto"): the gathered block of a rank holds only the Gaussians whose
        colour gradient is non-zero in its view -- [count, camera position | rows of (id, g (, direction))] at a capacity
        of 1.25 x the largest count any rank reported in the PREVIOUS step -- whenever that is at most 85 % of the dense
        block ("auto"), always, or never.  A count beyond the capacity is seen by every rank in the gathered headers and
        the step's gather is repeated densely; every rank expands the blocks locally (fg_payload_expand) and rebuilds as
        from dense ones: same gradients, bit for bit."""
        import os

        self.model, self.average, self.group = model, average, group
        self._flat: Optional[torch.Tensor] = None
        self._layout = None  # [(param, offset, numel)] of the flat buffer; the key it was built for
        self.bytes_last_step: Dict[str, int] = {}
        self.sparse = sparse if sparse is not None else os.environ.get("FG_DP_SPARSE", "auto")
        if self.sparse not in ("auto", "always", "never"):
            raise ValueError(f"sparse={self.sparse!r}: auto | always | never")
        self._sparse_plan = None  # (rows per rank, payload floats per row, N) of the next step's sparse blocks
        self.sparse_steps = self.dense_steps = self.sparse_overflows = 0
        self.force_overflow_next = False  # tests: treat the next sparse step's blocks as truncated (the dense repeat runs)

    def _others(self):
        gp = getattr(self.model, "gauss_params", {})
        colour = {id(gp[k]) for k in self.COLOUR if k in gp}
        return [p for p in self.model.parameters() if p.requires_grad and id(p) not in colour]

    def _bind(self) -> None:
        """``.grad`` of every non-colour parameter = its (zeroed) view of the flat buffer; the buffer is rebuilt when
        the parameter set changed (densification)."""
        params = self._others()
        key = tuple((p.data_ptr(), tuple(p.shape)) for p in params)
        if self._layout is None or self._layout[0] != key:
            off, spans = 0, []
            for p in params:
                off = (off + 3) & ~3  # 16-byte aligned views (fused optimizer steps take them as they are)
                spans.append((off, p.numel()))
                off += p.numel()
            self._flat = torch.zeros(off, device=params[0].device, dtype=torch.float32)
            self._layout = (key, spans)
        else:
            self._flat.zero_()
        for p, (off, n) in zip(params, self._layout[1]):
            p.grad = self._flat[off : off + n].view_as(p)

    def step(self):
        """Context manager around get_outputs + loss.backward of one view; the exchange runs on exit."""
        import contextlib

        from . import _lib, ops

        model, group = self.model, self.group
        world, collective = _collective_world(group)
        gp = model.gauss_params
        n = gp["means"].shape[0]
        state: Dict[str, object] = {}

        def gather(payload):
            if collective:
                gathered = torch.empty(world * payload.numel(), device=payload.device, dtype=torch.float32)
                if dist.get_backend(group) == "nccl":
                    work = dist.all_gather_into_tensor(gathered, payload, group=group, async_op=True)
                else:
                    work = dist.all_gather(list(gathered.view(world, -1).unbind(0)), payload, group=group, async_op=True)
                return gathered, work
            return payload, None

        def dense_payload():
            """[g (pf N) (| camera position (3))| count] -- the count rides behind the block fg_sh_grad_accumulate reads"""
            pf, payload = state["pf"], state["payload"]  # (g is a view of its head: the backward wrote it in place)
            if pf == 3:
                payload[3 * n : 3 * n + 3] = state["campos"]
            payload[-1:].view(torch.int32).copy_(state["count"])
            return payload

        def sink(what, *a):
            if what == "alloc":
                _n, dev, means = a
                # deformed means: every rank renders its own positions, so the direction travels with the gradient
                pf = 3 if means.data_ptr() == gp["means"].data_ptr() else 6
                stride = n * 6 if pf == 6 else (n + 1) * 3
                payload = torch.empty(stride + 1, device=dev, dtype=torch.float32)
                state.update(pf=pf, stride=stride, payload=payload, g=payload[: pf * n].view(n, pf))
                return state["g"]
            if what == "ready":
                _v_rgb, _means, viewmat, sh_degree, k_stored = a
                state.update(sh_degree=int(sh_degree), k_stored=int(k_stored))
                pf, g = state["pf"], state["g"]
                vm = viewmat.reshape(-1, 4)[:3]
                state["campos"] = -(vm[:, :3].T @ vm[:, 3])  # camera position of this rank's view (shared means)
                # which Gaussians took part in a pixel of this view (a non-zero colour gradient), and how many
                incl = torch.cumsum((g[:, :3] != 0).any(dim=1), 0, dtype=torch.int32)
                state["count"] = incl[-1:]
                plan = self._sparse_plan
                if plan is not None and plan[1:] == (pf, n):  # issued from inside the backward: the MLPs' backward runs under it
                    cap = plan[0]
                    block = torch.empty(4 + cap * (1 + pf), device=g.device, dtype=torch.float32)
                    _lib.check(_lib.load().fg_payload_compact(n, pf, g.data_ptr(), incl.data_ptr(), cap, block.data_ptr(),
                                                              torch.cuda.current_stream().cuda_stream), "fg_payload_compact")  # fmt: skip
                    block[1:4] = state["campos"]
                    state.update(form="sparse", cap=cap)
                    state["gathered"], state["work"] = gather(block)
                else:
                    state["form"] = "dense"
                    state["gathered"], state["work"] = gather(dense_payload())
            return None  # ("view" / "records": nothing to do here)

        @contextlib.contextmanager
        def cm():
            self._bind()
            for k in self.COLOUR:
                gp[k].grad = None
            rctx = ops.current()
            prev = rctx.color_grad_sink
            rctx.color_grad_sink = sink
            try:
                yield self
            finally:
                rctx.color_grad_sink = prev
            if "gathered" not in state:
                raise RuntimeError("ModelViewDP.step: no SH-coloured raster backward ran inside the context")
            flat = self._flat
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True) if collective else None
            if state["work"] is not None:
                state["work"].wait()
            pf, stride, lib = state["pf"], state["stride"], _lib.load()
            stream = torch.cuda.current_stream().cuda_stream
            blocks = state["gathered"].view(world, -1)
            # every rank's count, from the gathered headers / trailers: the same numbers on every rank.  The SPARSE form needs
            # them on the host before it expands (a truncated block means: repeat the gather densely); the dense form needs
            # them only to plan the next step, and reads them after its kernels are enqueued -- a blocking read here would
            # keep the host from queueing the rebuild under the all-reduce and the backward that are still running.
            counts_dev = (blocks[:, 0] if state["form"] == "sparse" else blocks[:, -1]).contiguous().view(torch.int32)
            counts = counts_dev.tolist() if state["form"] == "sparse" else None
            received = (world - 1) * blocks.shape[1] * 4
            if state["form"] == "sparse" and (max(counts) > state["cap"] or self.force_overflow_next):
                self.force_overflow_next = False
                # a block was truncated: all ranks see it, all repeat this step's gather densely
                self.sparse_overflows += 1
                state["form"] = "dense"
                state["gathered"], w2 = gather(dense_payload())
                if w2 is not None:
                    w2.wait()
                blocks = state["gathered"].view(world, -1)
                received += (world - 1) * blocks.shape[1] * 4
            if state["form"] == "sparse":
                self.sparse_steps += 1
                dense = torch.zeros(world * stride, device=blocks.device, dtype=torch.float32)
                _lib.check(lib.fg_payload_expand(n, pf, world, blocks.data_ptr(), blocks.shape[1], state["cap"], dense.data_ptr(),
                                                 stride, stream), "fg_payload_expand")  # fmt: skip
                src, src_stride = dense, stride
            else:
                self.dense_steps += 1
                src, src_stride = blocks, blocks.shape[1]
            scale = 1.0 / world if self.average else 1.0
            v_dc, v_rest = torch.empty_like(gp["features_dc"]), torch.empty_like(gp["features_rest"])
            _lib.check(lib.fg_sh_grad_accumulate_split(
                n, world, state["sh_degree"], state["k_stored"], gp["means"].detach().contiguous().data_ptr(),
                src.data_ptr(), src_stride, pf, scale, v_dc.data_ptr(), v_rest.data_ptr(), stream), "fg_sh_grad_accumulate_split")  # fmt: skip
            gp["features_dc"].grad, gp["features_rest"].grad = v_dc, v_rest
            if work is not None:
                work.wait()
                if self.average and world > 1:
                    flat.div_(world)
            if counts is None:
                counts = counts_dev.tolist()
            # the next step's form: sparse blocks at 1.25 x the largest count, if that is clearly smaller than dense ones
            cap = int(max(counts) * 1.25) + 1024
            sparse_floats, dense_floats = 4 + cap * (1 + pf), stride + 1
            use = self.sparse == "always" or (self.sparse == "auto" and sparse_floats < 0.85 * dense_floats)
            self._sparse_plan = (cap, pf, n) if use else None
            self.bytes_last_step = {"all_gather_received": received, "all_reduce": flat.numel() * 4,
                                    "plain_all_reduce_would_be": (flat.numel() + v_dc.numel() + v_rest.numel()) * 4,
                                    "payload_form": state["form"], "rows_with_colour_gradient": counts, "gaussians": n,
                                    "dense_block_bytes": dense_floats * 4, "sparse_block_bytes_at_these_counts": sparse_floats * 4}

        return cm()


def sync_densify_stats(model, group=None) -> None:
    """Before ``refinement_after`` on every rank: make the accumulated statistics identical
    (``all_reduce_densify_stats``) and re-seed the generator that draws the split samples."""
    if not _collective_world(group)[1]:
        return
    if model.xys_grad_norm is not None:
        all_reduce_densify_stats(model.xys_grad_norm, model.vis_counts, model.max_2Dsize, group=group)
        # every rank added its own "+1" start value to vis_counts (model.after_train_iter): keep one
        model.vis_counts.sub_(dist.get_world_size(group) - 1)
    shared_seed(group=group)
