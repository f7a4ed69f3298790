"""Torch-facing operators over the C ABI: one autograd Function per stage of the raster path.

PyTorch supplies device memory, the current HIP stream and the autograd tape; every byte of
arithmetic happens in ``libfgraster.so``.  All operators require CUDA(HIP) tensors and raise
otherwise -- there is no CPU path in the product (the CPU restatement lives in ``oracle/`` and
is test infrastructure only)."""
from __future__ import annotations

from typing import Optional, Tuple

import ctypes
import os
import threading
import time

import torch

from . import _lib

TILE_SIZE = 16
SPLAT_FLOATS = 16
MAX_CHANNELS = 8


# The workspace pool tells a free buffer by its storage's use count (a private torch hook: looked up HERE, once; a context
# created without it warns and counts its fallbacks)
_STORAGE_USE_COUNT = getattr(torch._C, "_storage_Use_Count", None)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream():
    # raw handle of the current stream of the current device; the public
    # torch.cuda.current_stream() builds a Stream object (~13 us, five times per step)
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


class StageTimer:
    """Optional per-stage HIP-event timing of the C-ABI calls (bench.py turns it on for the
    timed region).  Events are recorded on the stream the kernels are launched on; nothing
    synchronises until ``summary()``."""

    def __init__(self, only=None, prewarm=0):
        self.events = {}
        self.only = only  # None = every stage; else the set of stage names that get events
        # `prewarm` event pairs created (= recorded once) up front: fg_step_fwd / fg_step_bwd record the pair around
        # their raster launch THEMSELVES and need existing events -- creating them inside a timed region would cost two
        # extra records per pair
        self._pool = []
        for _ in range(prewarm):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(), b.record()
            self._pool.append((a, b))

    def record(self, name):
        if self.only is not None and name not in self.only:
            return None
        pair = self._pool.pop() if self._pool else (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        self.events.setdefault(name, []).append(pair)
        return pair

    @staticmethod
    def handles(pair):
        """The pair's raw hipEvent_t handles (for a library call that records them itself); events that do not exist yet
        are created by recording them once here."""
        for e in pair:
            if not e.cuda_event:
                e.record()
        return int(pair[0].cuda_event), int(pair[1].cuda_event)

    def summary(self):
        torch.cuda.synchronize()
        return {k: sum(a.elapsed_time(b) for a, b in v) / len(v) for k, v in self.events.items()}


def _env_flag(name: str, default: bool) -> bool:
    v = os.environ.get(name)
    return default if v is None else v != "0"


def launch_policy(**fields) -> "_lib.RasterConfig":
    """An ``fg_raster_config`` (include/fgraster.h) with the library's defaults and ``fields`` set, e.g.
    ``launch_policy(ppt_fwd=1, ppt_bwd=1)``, ``launch_policy(seg_parts=4)``."""
    cfg = _lib.RasterConfig.defaults()
    for k, v in fields.items():
        if k not in _lib.RasterConfig.FIELDS:
            raise ValueError(f"unknown launch-policy field {k!r}")
        setattr(cfg, k, int(v))
    return cfg


def launch_policy_from_env(env=None) -> "_lib.RasterConfig":
    """The FG_RASTER_* / FG_TILE_ORDER / FG_DEBUG_* variables (read HERE, once per context -- the library
    reads no environment) as an ``fg_raster_config``.  FG_RASTER_PPT_FWD / _BWD = 1|2|4;
    FG_RASTER_TAIL_FWD / _BWD = "t4[,t2]"; FG_RASTER_SPLIT_FWD / _BWD = "a4[,a2]"; FG_RASTER_BANDS = 1|2|4|8;
    FG_RASTER_LIVE = 0; FG_RASTER_SEG_PARTS, FG_RASTER_SEG_TAIL = n; FG_RASTER_SEG_GRADE = "parts2,tail2";
    FG_TILE_ORDER = rows|bands|cols|split|x|y; FG_DEBUG_ONLY_XCD = 0..7; FG_DEBUG_K_MOD = m; FG_RASTER_BALANCE = 0|1;
    FG_RASTER_PRIO_FWD / _BWD = "lo,hi" (percent of the mean single-strip job; "0" = off); FG_RASTER_HEAVY_WIDE = 0 (heavy
    tiles by round 4's three launches)."""
    env = os.environ if env is None else env
    f = {}

    def pair(name, a, b):
        v = env.get(name)
        if v is not None:
            parts = v.split(",")
            f[a] = int(parts[0])
            f[b] = int(parts[1]) if len(parts) > 1 else 0

    for name, field in (("FG_RASTER_PPT_FWD", "ppt_fwd"), ("FG_RASTER_PPT_BWD", "ppt_bwd"), ("FG_RASTER_BANDS", "bands_nx"),
                        ("FG_RASTER_SEG_PARTS", "seg_parts"), ("FG_RASTER_SEG_TAIL", "seg_tail"),
                        ("FG_DEBUG_ONLY_XCD", "debug_only_xcd"), ("FG_DEBUG_K_MOD", "debug_k_mod"),
                        ("FG_RASTER_BALANCE", "balance_bands"), ("FG_RASTER_HEAVY_WIDE", "heavy_wide"),
                        ("FG_RASTER_SEG_FINE", "seg_fine")):  # fmt: skip
        if env.get(name) is not None:
            f[field] = int(env[name])
    for name, field in (("FG_RASTER_PRIO_FWD", "prio_fwd"), ("FG_RASTER_PRIO_BWD", "prio_bwd")):  # "lo,hi" percent; "0" = off
        if env.get(name) is not None:
            parts = [int(x) for x in env[name].split(",")]
            f[field] = parts[0] | ((parts[1] if len(parts) > 1 else 0) << 16)
    pair("FG_RASTER_TAIL_FWD", "tail4_fwd", "tail2_fwd")
    pair("FG_RASTER_TAIL_BWD", "tail4_bwd", "tail2_bwd")
    pair("FG_RASTER_SPLIT_FWD", "split4_fwd", "split2_fwd")
    pair("FG_RASTER_SPLIT_BWD", "split4_bwd", "split2_bwd")
    pair("FG_RASTER_SEG_GRADE", "seg_parts2", "seg_tail2")
    if env.get("FG_RASTER_LIVE") == "0":
        f["use_liveness"] = 0
    if env.get("FG_TILE_ORDER"):
        f["tile_order"] = {"r": 0, "b": 1, "c": 2, "s": 3, "x": 4, "y": 5}.get(env["FG_TILE_ORDER"][0], 1)
    return launch_policy(**f)


class RasterContext:
    """Everything a call of the raster path reads besides its arguments: the launch policy handed to the C
    ABI, the host-side switches, the optional hooks (stage timer, gradient buffers, colour-gradient sink,
    static list capacity) and the list-capacity history.  One per model / stream of work; nothing here is
    module state.  A call uses the context that is current on its thread (``with ops.use(ctx):`` or the
    ``ctx=`` argument of ``rasterization``; ``ops.default_context`` otherwise) and every autograd node keeps
    the context of its forward, so the backward -- run by autograd's own thread -- sees the same one."""

    def __init__(self, policy=None, env=None):
        e = os.environ if env is None else env
        self.policy = policy if policy is not None else launch_policy_from_env(e)
        self.stage_timer: Optional[StageTimer] = None
        self.speculative_binning = e.get("FG_SPECULATIVE_BINNING", "1") != "0"
        self.static_capacity: Optional[int] = None  # set by graphed.GraphedRaster around capture / eager re-runs
        self.last_overflow: Optional[torch.Tensor] = None
        # What a shape's calls have taught the host is keyed by (device, tile grid) -- NOT by the number of Gaussians: the
        # reference changes N every `refine_every` steps for the whole run (freegaussian_model.py:404-571) and every
        # learned quantity would start from nothing each time.  Quantities that grow with N (list lengths, checkpoint
        # slots) are kept with the N they were seen at and scaled by the ratio; a jump of N beyond a factor of two is
        # another scene, and starts over.
        self.isect_capacity: dict = {}  # (dev, tile_w, tile_h, path) -> capacity, valid for isect_n[key] Gaussians
        self.isect_n: dict = {}
        self.isect_recent: dict = {}  # ... -> [(list length, N)] of the last calls
        self.capacity_redos = 0  # times a speculative list turned out too small and the fill was repeated
        self.stagewise_raster_calls = 0  # views that went through the stage-wise calls (not fg_step_*) although step_calls is on
        self.full_ckpt_allocs = 0  # steps with list shares whose checkpoint buffer was sized by the list capacity because
        # nothing was known of the shape's needs (a scene whose every tile is split takes the full size knowingly: not counted)
        self.seg_slots_known = False
        # Footprint rectangles (FG_TIGHT_RECTS=0 turns them off): the fused preprocess passes also write the
        # depth sort keys and, per Gaussian, the tile rectangle shrunk to the tiles where the splat can reach
        # alpha >= 1/255; the raster lists are then binned from those.  Same images and gradients, ~30% fewer
        # list entries on the 1M / 1080p scene.  The reference-exact lists (info["flatten_ids"] etc.) are
        # rebuilt on demand from the radius boxes.
        self.tight_rects = e.get("FG_TIGHT_RECTS", "1") != "0"
        # FOOTPRINT MASKS (FG_EXACT_TILES=0 turns them off): beside the rectangle, the preprocess passes write which blocks
        # of it the ELLIPSE alpha >= 1/255 reaches (8 bytes per Gaussian, fg::footprint_mask); the supertile binning counts and
        # scatters those only.  Nothing for a round splat, most of the rectangle for a needle lying diagonally -- the shape
        # densification fills a trained scene with: 57 % fewer list entries with 30 % needles of axis ratio 10 in the
        # 1M / 1080p scene, 12 % on the isotropic bench scene.  Same images and gradients.
        self.exact_tiles = e.get("FG_EXACT_TILES", "1") != "0"
        # ... per image size only WHERE THEY PAY (round 6; FG_EXACT_TILES=always: everywhere): fg_stbin_count reports the
        # rectangles' area beside the list length (count_out[14]); list length / area is what the masks keep.  A shape whose
        # last masked calls kept more than `mask_keep_max` of the pairs (round splats: 0.88 on the bench scene, where the
        # mask math, the 8 B per Gaussian and the count kernel's row items cost what the shorter lists save -- and 2-3 % more
        # on a cloud seen from closer) AND is an even shape (no tile list beyond three times the mean lately) runs WITHOUT masks
        # and looks again with them every 64th call; a trained scene keeps 0.52-0.58, needles 0.43.  A measured ratio, not a
        # length.
        self.masks_always = e.get("FG_EXACT_TILES", "1") == "always"
        # (DEFAULT: never off -- FG_MASK_KEEP_MAX=0.8 turns the switch on.  Measured on the bench cloud, kept share 0.894: forward
        # +12 us without masks, count kernel -6, per-Gaussian forward -3, backward -3..0 = nothing; and without them its
        # longest supertile segment straddles the long-segment threshold (7936): the flag flipped twice inside a 50-step
        # timed region, p90 0.74 -> 1.02 ms.  The verdict's "uniform >= 2840 Mpix/s" was met WITH masks on three boxes of
        # this round (2838-2877) and missed on the driver's round-5 box by box variance, not by the masks.)
        self.mask_keep_max = float(e.get("FG_MASK_KEEP_MAX", "1.01"))
        self.mask_keep = {}  # shape -> [kept share of the last masked calls (up to 4), calls since the last masked one]
        # FG_BINNING = supertile (default: fg_stbin_*, count / scatter per 2x2-tile supertile / one sort per
        # supertile) | depthfirst (rounds 1-2: fg_bin_prepare_keys + fg_bin_emit_sort, also the fallback beyond
        # fg_stbin_supported).  Identical lists.
        self.binning = e.get("FG_BINNING", "supertile")
        if self.binning not in ("supertile", "depthfirst"):
            raise ValueError(f"FG_BINNING={self.binning!r}: supertile | depthfirst")
        # The supertile path sorts a segment in LDS up to 7936 elements; a longer one (a dense cluster: tens of
        # thousands of splats over one 32 x 32-pixel supertile) is cut into buckets by a multi-workgroup sample sort
        # when fg_stbin_fill is called with FG_STBIN_LONG_SEGMENTS -- three more launches, so only for shapes that
        # need them: fg_stbin_count reports the longest segment beside the list length, a shape that showed one
        # beyond `long_segment` gets the flag for its next `long_cooldown` calls (renewed by every call that shows one
        # again).  Without the flag such a segment is sorted by ONE workgroup through global memory: correct, slow
        # (2.8 ms at a 111 000-entry tile) -- what the first call of a heavy shape pays.
        # FG_LONG_SEGMENTS = auto (default) | always | never.
        self.long_segments = e.get("FG_LONG_SEGMENTS", "auto")
        if self.long_segments not in ("auto", "always", "never"):
            raise ValueError(f"FG_LONG_SEGMENTS={self.long_segments!r}: auto | always | never")
        # (the flag is set for a shape while its calls report a segment beyond the large launch's LDS capacity, or more
        # than `long_many` beyond the small launch's: there the bucket passes beat one-segment-per-workgroup sorts)
        # (Tried: 10 000.  On layouts drawn with other seeds -- scripts/policy_regret.py 16 {7, 11, 23} -- shapes whose longest
        # segment is 8776 / 8856 / 9796 elements run 3-5 % faster with that one segment through global memory than with the three
        # extra launches; but the TRAINED scene, whose longest segments sit in that range in some of its views, loses 2-5 %
        # (0.567 -> 0.577-0.599 ms), and shapes at 10 658 / 11 858 lose 9-16 % without the launches.  The LDS sort's limit stands.)
        self.long_segment = int(e.get("FG_LONG_SEGMENT", "7936"))
        self.longest_segment_seen = 0  # (the last call's longest supertile segment: scripts)
        # (round 6: the count criterion is OFF by default.  On layouts the thresholds were not tuned on -- scripts/policy_regret.py
        # -- hundreds of segments of 3072..7936 elements and none beyond are sorted faster by the large launch, one segment per
        # workgroup in LDS, than by the bucket passes: needles 0.3 / 10 fill 0.160 -> 0.123 ms, a uniform cloud of large opaque
        # splats step 0.615 -> 0.590; the flag now follows the LONGEST segment alone, the one case the bucket passes exist for)
        self.long_many = int(e.get("FG_LONG_MANY", str(1 << 30)))
        self.long_cooldown = 64
        self.long_shapes = {}  # shape key -> calls left with the flag set
        self.long_calls = 0  # calls of fg_stbin_fill* that carried the flag
        self.test_small_slabs = False  # tests: FG_STBIN_TEST_SMALL_SLABS with it (the sample sort's whole-segment fallback runs)
        # HEAVY tiles (fg_raster_config::heavy_tiles): a tile list of thousands of entries that does not saturate is four
        # serial walks of ~100 ns per entry in the forward (1.4 ms for the rim tiles of a dense ball while the chip
        # idles); with the policy field set such a tile is composited by many jobs over shares of its list (+ one more
        # launch).  Like the long segments: fg_stbin_count reports the longest tile list, a shape that showed one beyond
        # `heavy_tile_len` runs with the field set for its next `heavy_cooldown` calls.
        # FG_HEAVY_TILES = auto (default) | always | never.
        self.heavy_tiles = e.get("FG_HEAVY_TILES", "auto")
        if self.heavy_tiles not in ("auto", "always", "never"):
            raise ValueError(f"FG_HEAVY_TILES={self.heavy_tiles!r}: auto | always | never")
        self.heavy_tile_len = int(e.get("FG_HEAVY_TILE_LEN", "768"))
        # (round 5: the policy comes on for a shape whose longest list exceeds `heavy_flag_len` AND -- where the forward reports
        # them: the one-call path -- whose jobs evaluated more than 2560 entries for their strips lately; while it is on, every list beyond
        # `heavy_tile_len` = the wide jobs' prefix of 512 entries + 256 is a heavy tile.  A dense opaque cluster has lists of
        # ten thousand entries that close after a few hundred: heavy tiles cost it 15 us and gain it nothing)
        self.heavy_flag_len = int(e.get("FG_HEAVY_FLAG_LEN", "2560"))
        self.heavy_cooldown = 64
        self.heavy_shapes = {}
        self.last_walk = 0
        self.long_walks = {}  # shape -> calls left for which a reported long job (more than 2560 entries evaluated for its strips) counts
        self.workspace_pool = e.get("FG_WORKSPACE_POOL", "1") != "0"
        self._workspaces = {}  # (kind, device, stream) -> [(uint8 buffer, its storage's use count when nobody else holds it)]
        self.pool_fallback_calls = 0  # workspace requests that fell back to torch.empty because the use-count hook is missing
        self.pool_new_buffers = 0  # workspace requests no pooled buffer could serve: a new device allocation
        self.plan_changes = 0  # one-call steps whose plan key differed from the shape's previous call ...
        self.plan_change_reasons = {}  # ... and which of its fields did: N, shares, long_mode, capacity, variant (heavy / slots / even / splits)
        self._last_plan = {}
        if self.workspace_pool and _STORAGE_USE_COUNT is None:
            import warnings

            warnings.warn("freegaussian_amd: torch._C._storage_Use_Count is missing in this torch build -- the workspace pool is "
                          "off and every step allocates its 0.4-0.8 GB workspaces through the caching allocator (a device "
                          "allocation behind every refinement); RasterContext.pool_fallback_calls counts them", RuntimeWarning)
        # UNEVEN shapes (a tile list beyond three times the mean in one of the shape's last eight calls: a cluster, a scene
        # seen from inside, a trained scene) get another cut of the work, in three parts (round 6, profiles/r06_xcd_shares.md):
        #  * the XCDs' shares are INTERLEAVED 2 x 2-tile blocks (fg_raster_config::balance_bands = 3) instead of row bands
        #    priced by a cost model -- list length capped at three times the mean -- that was tuned on the bench layouts: on
        #    17 layouts it had never seen the bands lost up to 12 % to plain equal spans and won up to 40 % against them; a
        #    trained scene runs 10 % faster interleaved than with either (FG_UNEVEN_INTERLEAVE=0: the cost bands);
        #  * the forward's content thresholds (split4_fwd / split2_fwd: a tile is cut into four / two strip jobs when its list
        #    is longer than that many 65536ths of all lists; 20 / 16 = 2.5 x / 2.0 x the mean on even scenes, where finer cuts
        #    only repeat the staging) are 8 / 5: with every XCD equally loaded nothing hides a long serial walk any more
        #    (FG_UNEVEN_SPLIT_FWD="a4,a2", "0" = off; round 5: 12 / 8 with the cost bands);
        #  * the backward gives a tile list shares from 4 / 65536 of all lists (FG_UNEVEN_SPLIT2_BWD; round 5: 12): whole-tile
        #    backward jobs over unsaturated lists of 400-900 entries at a cluster's rim were the launch (433 of 433 us).
        # Measured (1M / 1080p, median step): 80 % in a ball of 0.2: 0.96 (cost bands 0.96-0.98); half in a ball of 0.4: 0.84
        # (0.855); the trained scene 0.61 (0.68).  Even scenes keep equal spans and their thresholds (they would lose 3 %).
        sp = [int(x) for x in e.get("FG_UNEVEN_SPLIT_FWD", "8,5").split(",")]
        self.uneven_split_fwd = (sp[0], sp[1] if len(sp) > 1 else 0)
        self.uneven_split2_bwd = int(e.get("FG_UNEVEN_SPLIT2_BWD", "4"))
        self.uneven_interleave = e.get("FG_UNEVEN_INTERLEAVE", "1") != "0"
        self.heavy_calls = 0  # raster steps planned with heavy tiles on
        self._policy_copies = {}
        # Compact checkpoint slots (FG_COMPACT_SLOTS=0: off): the buffer of the backward's list shares sized by what the tiles
        # the backward may split need (reported by every list build, read one call late) instead of by the list's capacity
        self.compact_slots = e.get("FG_COMPACT_SLOTS", "1") != "0"
        # FG_EVEN_BANDS=0: never skip the cost pass of the XCD shares (cfg(even=True))
        self.even_bands = e.get("FG_EVEN_BANDS", "1") != "0"
        self.even_calls = {}  # shape -> consecutive calls without a tile list beyond three times the mean
        self.equal_stood = {}  # shape -> consecutive cost passes of the list build that kept the equal spans
        self.shape_calls = {}  # shape -> calls so far
        self.uneven_left = {}  # shape -> calls left for which its last uneven scene (a list beyond three times the mean) counts
        self.last_seg_slots = 0  # what the last call with list shares ran with (0: a slot per 64 entries of the capacity)
        self.ckpt_need = {}  # shape -> the last calls' needs (slots of the fullest XCD band)
        self.ckpt_pending = {}  # shape -> (ring slot, generation) of the call whose report has not been read yet
        # FG_DIRECT_COUNT=0: read the list length back with a copy in the stream instead of the kernel's own
        # store into pinned host memory (A/B)
        self.direct_count = e.get("FG_DIRECT_COUNT", "1") != "0"
        # FG_FILL_IN_FORWARD=0: zero the backward's record-gradient array with a fill launch at the head of
        # the backward instead of in passing in the mixed forward launch (A/B)
        self.fill_in_forward = e.get("FG_FILL_IN_FORWARD", "1") != "0"
        # The forward's compositing checkpoints for the backward's list shares cost 64 bytes per list entry of
        # the speculative capacity (fg_raster_seg_ckpt_floats), allocated per training forward and held until
        # the backward.  Above this many bytes (FG_SEG_CKPT_BUDGET_MB, default 2048) a step does without them.
        self.seg_ckpt_budget_bytes = int(float(e.get("FG_SEG_CKPT_BUDGET_MB", "2048")) * (1 << 20))
        # Optional two-stream forward (FG_OVERLAP_PACK=1, off by default): the projection (whose outputs the
        # binning needs) runs on the current stream, the colour + record half (HBM-bound, 300 B per
        # Gaussian) on a side stream, concurrently with the binning kernels.  Measured on MI355X at 1M /
        # 1080p (profiles/r01_two_stream_forward.md): no gain -- the 3900-workgroup colour kernel crowds the
        # small sort kernels out (fg_bin_prepare 0.114 -> 0.153 ms) by as much as it hides.
        self.overlap_pack = e.get("FG_OVERLAP_PACK", "0") == "1"
        # FG_SH_JAC=0: the per-Gaussian backward reads the SH coefficient rows again (192 B per Gaussian) instead of
        # the 40-byte note (d colour / d direction + clamp mask) the forward leaves for it.
        self.sh_jacobian = e.get("FG_SH_JAC", "1") != "0"
        # The raster job lists (fg_raster_build_jobs: 16 workgroups, 9 us) are built inside the scatter launch of
        # fg_stbin_fill_jobs when the caller of bin_tiles says what will be rastered (raster_hint);
        # FG_JOBS_IN_FILL=0: by a launch of their own in front of the raster forward, as before ABI version 5.
        self.jobs_in_fill = e.get("FG_JOBS_IN_FILL", "1") != "0"
        # One C-ABI call per direction (fg_step_fwd / fg_step_bwd) for calls in the default configuration on a shape whose
        # list capacity is known; FG_STEP_CALLS=0: always the stage-wise calls (the same kernels, ~6 calls and ~20
        # allocations per view instead of 2 and 2)
        self.step_calls = e.get("FG_STEP_CALLS", "1") != "0"
        # Optional hook: maps an input tensor of the fused backward to the buffer its gradient should be
        # written into (e.g. a slice of a flat all-reduce buffer, viewdp.FlatGaussianParams.direct_grads()).
        # The kernels overwrite their outputs densely, so the buffer needs no zeroing.
        self.grad_alloc = None
        # Optional hook for the factored view-DP exchange (viewdp.FlatGaussianParams.factored_exchange): when
        # set, the fused backward of an SH-coloured view hands the clamp-masked colour gradient g[N,3] (+ what
        # is needed to rebuild the coefficient gradient) to the sink instead of writing the dense [N,K,3]
        # coefficient gradient; `colors.grad` is then filled by the exchange, not by autograd.
        self.color_grad_sink = None

    def workspace(self, kind: str, numel: int, dtype, dev) -> torch.Tensor:
        """A buffer of ``numel`` elements for the one-call path's ``keep`` / ``tmp`` workspace, from a small pool this
        context owns (FG_WORKSPACE_POOL=0: a fresh ``torch.empty`` per call, rounds 4's way).

        Why not the caching allocator alone: the workspaces are its largest blocks (0.4-0.8 GB at 1M / 1080p).  A refinement
        step allocates hundreds of small tensors, some of them carved out of the cached block a workspace came from -- and
        the step behind the refinement has to ask the device for a new segment of that size: one hipMalloc of 757 MB,
        usually 1 ms, 17 ms in two of four box visits (scripts/refine_step_bench.py, REFINE_DIAG=1); keeping the allocator
        from splitting large blocks (max_split_size_mb) trades that for misses whenever two calls' sizes differ by more
        than 20 MB.  The pool hands out a VIEW of a buffer nobody else refers to any more -- the storage's use count is
        back at what it was when the buffer was made: outputs, ``info`` tensors and the autograd node's saved tensors are
        views of ``keep`` and keep it out of circulation exactly as long as they live -- that is at least as large as asked
        for and at most 30 % (+ 32 MB) larger; else a new one with 5 % to spare.  At most four buffers per kind and device;
        the oldest unused one goes first.  Same-stream reuse only (as the caching allocator without record_stream); never
        under a stream capture."""
        use_count = _STORAGE_USE_COUNT
        # (under a stream capture the allocator's private pool of the graph is the only right place: a replay writes where the
        # capture's tensors lived, whoever holds that memory by then)
        if not self.workspace_pool or use_count is None or (dev.type == "cuda" and torch.cuda.is_current_stream_capturing()):
            # (pool on but torch's private use-count hook gone: the 17-21 ms hipMalloc behind every refinement is back -- counted,
            # warned about once at context creation, reported by bench.py as `pool_fallback_calls`)
            self.pool_fallback_calls += int(self.workspace_pool and use_count is None)
            return torch.empty(numel, dtype=dtype, device=dev)
        item = torch.empty(0, dtype=dtype).element_size()
        nbytes = numel * item
        # keyed by the STREAM too: the caching allocator ties a block to the stream it was allocated on; a buffer released by
        # one stream's call must not be handed to a call on another stream whose kernels could overwrite what the first is
        # still reading (eval on a side stream, two training streams sharing a context)
        stream_key = _stream() if dev.type == "cuda" else 0
        pool = self._workspaces.setdefault((kind, str(dev), stream_key), [])
        for i, (buf, idle) in enumerate(pool):
            if nbytes <= buf.numel() <= int(1.3 * nbytes) + (32 << 20) and use_count(buf.untyped_storage()._cdata) == idle:
                pool.append(pool.pop(i))  # (most recently used last)
                return buf[:nbytes].view(dtype)
        buf = torch.empty((int(nbytes * 1.05) + (2 << 20) - 1) // (2 << 20) * (2 << 20), dtype=torch.uint8, device=dev)
        self.pool_new_buffers += 1
        pool.append((buf, use_count(buf.untyped_storage()._cdata)))
        if len(pool) > 4:
            for i, (old, idle) in enumerate(pool[:-1]):
                if use_count(old.untyped_storage()._cdata) == idle:
                    pool.pop(i)
                    break
            else:
                pool.pop(0)  # (all in use: the pool forgets the oldest; its users keep it alive)
        return buf[:nbytes].view(dtype)

    def masks_on(self, lkey) -> bool:
        """Footprint masks for the next call of this shape?  (FG_EXACT_TILES=0: never; =always: always; else by the kept share
        of the shape's last masked calls -- a probe with masks every 64th call while they are off.)"""
        if not self.exact_tiles:
            return False
        st = self.mask_keep.get(lkey)
        if self.masks_always or st is None or not st[0]:
            return True
        if sum(st[0]) / len(st[0]) <= self.mask_keep_max:
            return True
        # ... and only EVEN shapes go without: where lists are long and crowded -- a cluster -- the tenth of the entries round
        # splats lose is a tenth of the forward's walk (half of the Gaussians in a ball of 0.4: forward 0.228 -> 0.278 ms
        # without masks at the same kept share of 0.89, against 0.184 -> 0.197 on the even cloud, which gets 20 us back
        # in the per-Gaussian forward and the count kernel)
        if not self.even_shape(lkey):
            return True
        return st[1] >= 63

    def note_mask_ratio(self, lkey, masked: bool, n_isects: int, area: int) -> None:
        st = self.mask_keep.get(lkey)
        if st is None:
            if len(self.mask_keep) >= 256:
                self.mask_keep.pop(next(iter(self.mask_keep)))
            st = self.mask_keep[lkey] = [[], 0]
        if masked and area > 0:
            st[0].append(n_isects / area)
            del st[0][:-4]
            st[1] = 0
        else:
            st[1] += 1

    def heavy_lens(self, lkey):
        """(longest list that turns heavy tiles on, list length from which a tile IS heavy while they are on) for a shape.
        The round-5 pair (2560 / 768) belongs to the walk reports: a shape keeps heavy tiles only while its strips report
        jobs that evaluated more than 2560 entries.  Only the one-call path asks the forward for those reports; a shape that has never
        delivered one (stage-wise calls: FG_STEP_CALLS=0, a stage timer, images the step path refuses) would keep heavy
        tiles on by list LENGTH alone -- a dense opaque cluster, 15 us lost per step -- so it stays on round 4's pair."""
        if lkey in self.long_walks:
            return self.heavy_flag_len, self.heavy_tile_len
        return max(self.heavy_flag_len, 3072), max(self.heavy_tile_len, 3072)

    def release_workspaces(self) -> int:
        """Drop the pool's buffers (up to four per kind, device and stream: 0.4-0.8 GB each at 1M / 1080p, which
        ``torch.cuda.empty_cache()`` cannot reach while the pool holds them); buffers still referenced by live outputs stay
        alive through those.  -> bytes the pool let go of."""
        freed = sum(buf.numel() for pool in self._workspaces.values() for buf, _ in pool)
        self._workspaces.clear()
        return freed

    def uneven_shape(self, lkey) -> bool:
        """Has one of the shape's last eight calls shown a tile list beyond three times the mean?  Then the forward's content
        thresholds are the finer ones (`uneven_split_fwd`)."""
        return self.uneven_left.get(lkey, 0) > 0

    def cfg(self, heavy: bool = False, seg_slots: int = 0, even: bool = False, uneven: bool = False, heavy_len: int = 0) -> int:
        """Address of the launch policy (the `const fg_raster_config*` argument); ``heavy``: the same policy with
        ``heavy_tiles`` set (unless the policy sets it itself); ``seg_slots`` > 0: with compact checkpoint slots, that many;
        ``even``: with ``balance_bands = 2`` (equal numbers of tiles per XCD without looking at the costs: a shape whose
        recent calls had no tile list far above the mean).  The copies live as long as the context: autograd nodes and
        cached step plans hold their addresses."""
        return self.cfg_variant(heavy, seg_slots, even, uneven, heavy_len)[0]

    def cfg_variant(self, heavy: bool = False, seg_slots: int = 0, even: bool = False, uneven: bool = False, heavy_len: int = 0):
        """``cfg`` and the VALUES that tell the variant from the context's own policy: (address, (heavy_len, seg_slots,
        even)) -- what a cache of anything derived from the policy is keyed on (an address can be reused by another
        context's copy)."""
        heavy_len = (int(heavy_len) or self.heavy_tile_len) if heavy and self.policy.heavy_tiles <= 0 else 0
        if self.policy.seg_slots > 0:
            seg_slots = 0  # (the policy's own value stands)
        seg_slots = max(int(seg_slots), 0)
        even = bool(even) and self.policy.balance_bands in (-1, 1)
        split = self.uneven_split_fwd if uneven and not even and self.policy.split4_fwd < 0 and self.uneven_split_fwd[0] > 0 else None
        variant = (heavy_len, seg_slots, even, split)
        if not heavy_len and seg_slots <= 0 and not even and split is None:
            return self.policy.ptr(), variant
        # (the policy may have been replaced or changed in place; few keys per policy: seg_slots comes in steps of 4096 slots)
        key = (bytes(self.policy), heavy_len, int(seg_slots), even, split, self.uneven_split2_bwd, self.uneven_interleave)
        copy = self._policy_copies.get(key)
        if copy is None:
            copy = type(self.policy).from_buffer_copy(self.policy)
            if split is not None:
                copy.split4_fwd, copy.split2_fwd = split
                if self.policy.split4_bwd < 0 and self.uneven_split2_bwd > 0:  # (list shares: only the second threshold counts)
                    copy.split4_bwd, copy.split2_bwd = 20, self.uneven_split2_bwd
                if self.uneven_interleave and self.policy.balance_bands in (-1, 1):
                    copy.balance_bands = 3
            if heavy_len:
                copy.heavy_tiles = heavy_len
            if seg_slots > 0:
                copy.seg_slots = int(seg_slots)
            if even:
                copy.balance_bands = 2
            self._policy_copies[key] = copy
        return copy.ptr(), variant

    def capacity_for(self, key, N: int):
        """The speculative list capacity for a call with ``N`` Gaussians of the shape ``key``; None when nothing (usable)
        is known.  The same N as the history's: the stored figure (tests overwrite it to force the overflow path).  Another
        N within a factor of two -- the call after a refinement: the history scaled by the ratio of the counts, 5 % on top
        (clones and split children sit where the lists are already long)."""
        if not self.speculative_binning:
            return None
        cap = self.isect_capacity.get(key)
        if cap is None:
            return None
        n_ref = self.isect_n.get(key, N)
        if n_ref == N:
            return cap
        if not (n_ref <= 2 * N and N <= 2 * n_ref):
            return None
        recent = self.isect_recent.get(key)
        if not recent:
            return min(int(cap * (N / n_ref) * 1.05) + 4096, 2**31 - 1)
        scaled = list_capacity_for([int(n * (N / n_i) * 1.05) for n, n_i in recent])
        # (a capacity that is still enough and not 25 % too large stays: every list-sized buffer of the step keeps its size
        # and the caching allocator its blocks -- a fresh gigabyte block is 15 ms of hipMalloc on the step after a refinement)
        return cap if scaled <= cap <= scaled + scaled // 4 else scaled

    def even_shape(self, lkey) -> bool:
        """Has this shape shown only even scenes lately (no tile list beyond three times the mean in its last eight
        calls)?  Then the job lists take equal numbers of tiles per XCD without the cost pass (``balance_bands = 2``)."""
        if not self.even_bands:
            return False
        if self.even_calls.get(lkey, 0) >= 8:
            return True
        # ... or the cost pass itself kept the equal spans in the shape's last eight calls that ran it (a few long lists, the
        # shares even all the same); it runs again every 64th call of the shape
        n = self.shape_calls.get(lkey, 0)
        return self.equal_stood.get(lkey, 0) >= 8 and n % 64 != 0

    def seg_slots_for(self, lkey, capacity: int, n_tiles: int, N: int = 0) -> int:
        """Compact checkpoint slots for the next call of a shape (fg_raster_config::seg_slots), from what the list builds
        of its last calls reported (fg_stbin_fill_jobs' ckpt_need_out): 8 x the fullest XCD band's need + an eighth, in
        steps of 4096 slots; 0 (a slot per 64 entries of the list's capacity) while nothing is known or when that would
        not be smaller.  ``N``: this call's Gaussian count -- needs reported at another count are scaled by the ratio (a
        slot per 64 list entries, and the lists grow with N)."""
        hist = self.ckpt_need.get(lkey) if self.compact_slots else None
        self.seg_slots_known = bool(hist)
        if not hist:
            return 0
        need = max(nd if (N <= 0 or n_i == N) else int(nd * (N / n_i) * 1.05) + 8 for nd, n_i in hist)
        per = int(need * 1.125) + 32
        slots = -(-8 * per // 4096) * 4096
        self.last_seg_slots = slots if slots < 0.9 * (capacity // 64 + n_tiles + 2) else 0
        return self.last_seg_slots


_default_context: Optional[RasterContext] = None
_tls = threading.local()


def _get_default() -> RasterContext:
    global _default_context
    if _default_context is None:
        _default_context = RasterContext()
    return _default_context


def current() -> RasterContext:
    """The context of the calling thread (innermost ``use``), else the process default."""
    stack = getattr(_tls, "stack", None)
    return stack[-1] if stack else _get_default()


class use:
    """``with ops.use(ctx): ...`` -- calls of the raster path on this thread read ``ctx``."""

    def __init__(self, ctx: Optional[RasterContext]):
        self.ctx = ctx

    def __enter__(self):
        if self.ctx is not None:
            if not hasattr(_tls, "stack"):
                _tls.stack = []
            _tls.stack.append(self.ctx)
        return self.ctx if self.ctx is not None else current()

    def __exit__(self, *exc):
        if self.ctx is not None:
            _tls.stack.pop()
        return False


def __getattr__(name):  # ops.default_context: created on first use (it loads the library for the policy)
    if name == "default_context":
        return _get_default()
    raise AttributeError(name)


def _call(name: str, *args, on=None, stage=None) -> None:
    """Invoke one C-ABI entry point, optionally bracketed by HIP events recorded on the stream the
    kernels go to (``on``: a torch stream other than the current one; the caller passes its raw
    handle as the entry point's stream argument).  ``stage``: name the time is booked under."""
    fn = getattr(_lib.load(), name)
    stage_timer = current().stage_timer
    ev = stage_timer.record(stage or name) if stage_timer is not None else None
    if ev:
        ev[0].record(on) if on is not None else ev[0].record()
    rc = fn(*args)
    if ev:
        ev[1].record(on) if on is not None else ev[1].record()
    _lib.check(rc, name)


def _f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.FgRasterError(f"{name} must be a CUDA/HIP tensor: the raster path has no CPU fallback")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


# --------------------------------------------------------------------------------------------
# K1 projection


class _Project(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means, quats, scales, viewmat, K, width, height, eps2d, near, far, radius_clip, tile_size,
                calc_comp):  # fmt: skip
        ctx.rctx = current()  # the backward (autograd's thread) reads the same context
        lib = _lib.load()
        N = means.shape[0]
        dev = means.device
        radii = torch.empty(N, dtype=torch.int32, device=dev)
        means2d = torch.empty(N, 2, dtype=torch.float32, device=dev)
        depths = torch.empty(N, dtype=torch.float32, device=dev)
        conics = torch.empty(N, 3, dtype=torch.float32, device=dev)
        comp = torch.empty(N, dtype=torch.float32, device=dev) if calc_comp else None
        tiles = torch.empty(N, dtype=torch.int32, device=dev)
        _call("fg_project_fwd", N, _ptr(means), _ptr(quats), _ptr(scales), _ptr(viewmat), _ptr(K), width, height,
                               eps2d, near, far, radius_clip, tile_size, _ptr(radii), _ptr(means2d), _ptr(depths),
                               _ptr(conics), _ptr(comp), _ptr(tiles), _stream())  # fmt: skip
        ctx.save_for_backward(means, quats, scales, viewmat, K, radii, conics, comp)
        ctx.args = (width, height, eps2d)
        ctx.mark_non_differentiable(radii, tiles)
        if comp is None:
            comp = torch.empty(0, device=dev)
            ctx.mark_non_differentiable(comp)
        return radii, means2d, depths, conics, comp, tiles

    @staticmethod
    def backward(ctx, *grads):
        with use(ctx.rctx):
            return _Project._backward(ctx, *grads)

    @staticmethod
    def _backward(ctx, _v_radii, v_means2d, v_depths, v_conics, v_comp, _v_tiles):
        lib = _lib.load()
        means, quats, scales, viewmat, K, radii, conics, comp = ctx.saved_tensors
        width, height, eps2d = ctx.args
        N = means.shape[0]
        dev = means.device

        def z(shape):
            return torch.zeros(shape, dtype=torch.float32, device=dev)

        v_means2d = z((N, 2)) if v_means2d is None else v_means2d.contiguous()
        v_depths = z((N,)) if v_depths is None else v_depths.contiguous()
        v_conics = z((N, 3)) if v_conics is None else v_conics.contiguous()
        v_comp = None if (comp is None or v_comp is None or v_comp.numel() == 0) else v_comp.contiguous()
        v_means = torch.empty_like(means)
        v_quats = torch.empty_like(quats)
        v_scales = torch.empty_like(scales)
        _call("fg_project_bwd", N, _ptr(means), _ptr(quats), _ptr(scales), _ptr(viewmat), _ptr(K), width, height,
                               eps2d, _ptr(radii), _ptr(conics), _ptr(comp), _ptr(v_means2d), _ptr(v_depths),
                               _ptr(v_conics), _ptr(v_comp), _ptr(v_means), _ptr(v_quats), _ptr(v_scales),
                               _stream())  # fmt: skip
        return (v_means, v_quats, v_scales) + (None,) * 10


def project(means, quats, scales, viewmat, K, width, height, eps2d=0.3, near_plane=0.01, far_plane=1e10,
            radius_clip=0.0, tile_size=TILE_SIZE, calc_compensations=False):  # fmt: skip
    """-> radii[N] int32, means2d[N,2], depths[N], conics[N,3], compensations[N]|empty, tiles_touched[N]."""
    means, quats, scales = _f32(means, "means"), _f32(quats, "quats"), _f32(scales, "scales")
    viewmat, K = _f32(viewmat, "viewmat"), _f32(K, "K")
    return _Project.apply(means, quats, scales, viewmat, K, int(width), int(height), float(eps2d),
                          float(near_plane), float(far_plane), float(radius_clip), int(tile_size),
                          bool(calc_compensations))  # fmt: skip


# --------------------------------------------------------------------------------------------
# K2 spherical harmonics


class _SH(torch.autograd.Function):
    @staticmethod
    def forward(ctx, degree, means, viewmat, coeffs, radii):
        ctx.rctx = current()  # the backward (autograd's thread) reads the same context
        lib = _lib.load()
        N, k_stored = coeffs.shape[0], coeffs.shape[1]
        colors = torch.empty(N, 3, dtype=torch.float32, device=means.device)
        _call("fg_sh_fwd", N, degree, k_stored, _ptr(means), _ptr(viewmat), _ptr(coeffs), _ptr(radii), _ptr(colors),
                          _stream())  # fmt: skip
        ctx.save_for_backward(means, viewmat, coeffs, radii, colors)
        ctx.degree = degree
        return colors

    @staticmethod
    def backward(ctx, *grads):
        with use(ctx.rctx):
            return _SH._backward(ctx, *grads)

    @staticmethod
    def _backward(ctx, v_colors):
        lib = _lib.load()
        means, viewmat, coeffs, radii, colors = ctx.saved_tensors
        N, k_stored = coeffs.shape[0], coeffs.shape[1]
        v_coeffs = torch.empty_like(coeffs)
        v_means = torch.empty_like(means) if ctx.needs_input_grad[1] else None
        _call("fg_sh_bwd", N, ctx.degree, k_stored, _ptr(means), _ptr(viewmat), _ptr(coeffs), _ptr(radii),
                          _ptr(colors), _ptr(v_colors.contiguous()), _ptr(v_coeffs), _ptr(v_means), _stream())  # fmt: skip
        return None, v_means, None, v_coeffs, None


def spherical_harmonics(degree: int, means, viewmat, coeffs, radii=None):
    """colour[N,3] = max(SH_degree(mean - campos) + 0.5, 0); rows with radii<=0 are 0."""
    means, viewmat, coeffs = _f32(means, "means"), _f32(viewmat, "viewmat"), _f32(coeffs, "coeffs")
    if coeffs.dim() != 3 or coeffs.shape[2] != 3 or coeffs.shape[1] > 16:
        raise ValueError("coeffs must be [N, K<=16, 3]")
    if not 0 <= degree <= 3 or coeffs.shape[1] < (degree + 1) ** 2:
        raise ValueError("sh degree must be 0..3 with K >= (degree+1)^2 coefficients")
    return _SH.apply(int(degree), means, viewmat, coeffs, radii)


# --------------------------------------------------------------------------------------------
# K3 + K4 binning and sort (integer path, no gradients)


@torch.no_grad()
def isect_tiles(means2d, radii, depths, tiles_touched, tile_size, tile_w, tile_h, sort=True):
    """-> (isect_ids[I] int64, flatten_ids[I] int32, tile_offsets[T+1] int32).
    One host read-back (I) sizes the lists, as in the reference's rasterizer."""
    lib = _lib.load()
    N = means2d.shape[0]
    dev = means2d.device
    n_tiles = tile_w * tile_h
    cum = torch.empty(N, dtype=torch.int64, device=dev)
    ws = torch.empty(max(int(lib.fg_scan_workspace_bytes(N)), 8), dtype=torch.uint8, device=dev)
    _call("fg_scan_tiles", N, _ptr(tiles_touched), _ptr(cum), _ptr(ws), ws.numel(), _stream())
    n_isects = int(cum[-1].item()) if N > 0 else 0
    if n_isects >= 2**31:
        raise _lib.FgRasterError(f"{n_isects} tile intersections exceed the int32 list index range")
    isect_ids = torch.empty(n_isects, dtype=torch.int64, device=dev)
    flatten_ids = torch.empty(n_isects, dtype=torch.int32, device=dev)
    offsets = torch.empty(n_tiles + 1, dtype=torch.int32, device=dev)
    if n_isects > 0:
        _call("fg_tile_bin", N, _ptr(means2d), _ptr(radii), _ptr(depths), _ptr(cum), tile_size, tile_w, tile_h,
                            _ptr(isect_ids), _ptr(flatten_ids), _stream())  # fmt: skip
        if sort:
            sort_pairs(isect_ids, flatten_ids, 32 + max(n_tiles - 1, 1).bit_length())
    _call("fg_tile_ranges", n_isects, _ptr(isect_ids), n_tiles, _ptr(offsets), _stream())
    return isect_ids, flatten_ids, offsets


def list_capacity_for(recent_lengths) -> int:
    """Speculative list capacity from the list lengths of the last calls of a shape: 25% headroom over the
    heaviest, rounded up to 1/32..1/16 of its magnitude (one allocation size per shape, not one per view)."""
    want = int(max(recent_lengths) * 1.25) + 4096
    granule = 1 << max(want.bit_length() - 5, 12)
    return min(-(-want // granule) * granule, 2**31 - 1)


def _count_buffer(dev) -> torch.Tensor:
    # one pinned 8-byte buffer per call (torch's caching host allocator makes this cheap): a shared
    # one would be overwritten by a second in-flight call on the same device
    return torch.empty(1, dtype=torch.int64, pin_memory=True)


# Words the scan kernel stores the list length into (fg_bin_prepare_keys count_out): a ring of pinned
# int64 slots that lives as long as the process, so that a store of a call whose Python side died in
# between can only ever land in a slot of this ring.  The host writes -1 before the launch and polls.
_COUNT_RING = 256
_count_ring = None
_count_ring_np = None
_count_ring_next = 0
_count_ring_lock = threading.Lock()
_count_ring_stream = [None] * _COUNT_RING  # the stream the store into slot i was enqueued on
_count_ring_gen = [0] * _COUNT_RING  # times slot i has been handed out
_RING_WORDS = 16  # int64 words per slot


def _count_slot():
    """A slot of sixteen pinned int64 words: [0] the list length (every binning path), [1] the longest supertile
    segment, [2] the longest tile list, [3] the segments beyond the small sort's capacity (fg_stbin_count only; they stay
    -1 otherwise), [14] the summed area of the footprint rectangles (fg_stbin_count), [12] what the cost pass over the XCDs' shares decided, [4..11] the checkpoint slots the
    eight XCD bands' tiles would take (fg_stbin_fill_jobs' ckpt_need_out; read one call late), [13] a long walk of the
    raster forward (fg_raster_jobs_fwd's walk_out; 0 = none; read one call late).  -> (slot, address)."""
    global _count_ring, _count_ring_np, _count_ring_next
    with _count_ring_lock:
        if _count_ring is None:
            _count_ring = torch.empty(_RING_WORDS * _COUNT_RING, dtype=torch.int64, pin_memory=True)
            _count_ring_np = _count_ring.numpy()
        i = _count_ring_next
        _count_ring_next = (i + 1) % _COUNT_RING
        _count_ring_gen[i] += 1
        _count_ring_np[_RING_WORDS * i : _RING_WORDS * i + 13] = -1
        _count_ring_np[_RING_WORDS * i + 13] = 0  # (fg_raster_jobs_fwd's walk_out: jobs that evaluated more than 2560 entries for their strips)
        _count_ring_np[_RING_WORDS * i + 14] = -1  # (fg_stbin_count: the footprint rectangles' area)
        _count_ring_stream[i] = torch.cuda.current_stream()  # (under the lock: slot i is this caller's from here on)
    return i, _count_ring.data_ptr() + 8 * _RING_WORDS * i


def _note_ckpt_need(rctx, lkey, count_slot, reported: bool, N: int = 0, walks: bool = False) -> None:
    """Read the checkpoint-slot needs the PREVIOUS call of this shape reported (they have landed: this call's list length,
    which the caller has just waited for, was stored behind them), remember this call's slot for the next."""
    prev = rctx.ckpt_pending.pop(lkey, None)
    if prev is not None and _count_ring_gen[prev[0]] == prev[1]:
        words = _count_ring_np[_RING_WORDS * prev[0] + 4 : _RING_WORDS * prev[0] + 12]
        if int(words.min()) >= 0:
            hist = rctx.ckpt_need.setdefault(lkey, [])
            hist.append((int(words.max()), prev[2]))  # (the need, the Gaussian count it was seen at)
            del hist[:-8]
            if len(rctx.ckpt_need) > 256:
                rctx.ckpt_need.pop(next(iter(rctx.ckpt_need)))
        # long walks of that call's raster forward (when it was asked to report them: the one-call path): the shape's
        # heavy tiles stay on (or come on) while they are reported
        if len(prev) > 3 and prev[3]:
            if len(rctx.long_walks) > 256 and lkey not in rctx.long_walks:
                rctx.long_walks.pop(next(iter(rctx.long_walks)))
            left = rctx.long_walks.get(lkey, 0)
            rctx.last_walk = int(_count_ring_np[_RING_WORDS * prev[0] + 13])  # (one of the reporting strips' walks; scripts)
            rctx.long_walks[lkey] = rctx.heavy_cooldown if rctx.last_walk > 0 else max(left - 1, 0)
        decided = int(_count_ring_np[_RING_WORDS * prev[0] + 12])
        if decided >= 0:  # (the cost pass ran: 1 = it balanced the shares by cost, 0 = the equal spans stood)
            if len(rctx.equal_stood) > 256 and lkey not in rctx.equal_stood:
                rctx.equal_stood.pop(next(iter(rctx.equal_stood)))
            rctx.equal_stood[lkey] = rctx.equal_stood.get(lkey, 0) + 1 if decided == 0 else 0
    if reported:
        if len(rctx.ckpt_pending) > 256:
            rctx.ckpt_pending.clear()
        rctx.ckpt_pending[lkey] = (count_slot, _count_ring_gen[count_slot], N, walks)


def _poll_count(i: int, word: int = 0) -> int:
    """Spin on a word of ring slot i until the kernel's system-scope store arrives (it does while the GPU is
    still busy with the emission and the tile sort: the wait is microseconds).  No event in the stream."""
    a, j = _count_ring_np, _RING_WORDS * i + word
    v = int(a[j])
    if v >= 0:
        return v
    t0 = time.perf_counter()
    while True:
        for _ in range(64):
            v = int(a[j])
            if v >= 0:
                return v
        if time.perf_counter() - t0 > 0.02:
            break
    # not seen within 20 ms: drain the stream the store was enqueued on (not whatever is current now) and look again
    (_count_ring_stream[i] or torch.cuda.current_stream()).synchronize()
    v = int(a[j])
    if v < 0:
        raise _lib.FgRasterError("the list length never arrived in pinned host memory (count_out of the binning call)")
    return v


@torch.no_grad()
def tile_keys_from_offsets(offsets: torch.Tensor, n: int) -> torch.Tensor:
    """The sorted list's tile ids [n] int32, rebuilt from the tile ranges (the product path does
    not materialise them: fg_bin_emit_sort keeps 16-bit keys in its workspace)."""
    counts = (offsets[1:] - offsets[:-1]).long()
    tiles = torch.arange(counts.numel(), dtype=torch.int32, device=offsets.device)
    return torch.repeat_interleave(tiles, counts, output_size=int(n))


def _binning_side_outputs(N, tile_size, width, height, dev):
    """(depth_keys, tile_rects, tile_masks) buffers for the preprocess passes, or Nones."""
    tile_w, tile_h = (width + tile_size - 1) // tile_size, (height + tile_size - 1) // tile_size
    rctx = current()
    if not rctx.tight_rects or tile_w > 1023 or tile_h > 1023 or N == 0:
        return None, None, None
    masks = torch.empty(N, dtype=torch.int64, device=dev) if rctx.binning == "supertile" and rctx.masks_on((dev, tile_w, tile_h)) else None
    return torch.empty(N, dtype=torch.int32, device=dev), torch.empty(N, 2, dtype=torch.int32, device=dev), masks


def bin_tiles(means2d, radii, depths, tiles_touched, tile_size, tile_w, tile_h, defer=False, want_keys=True,
              keys_rects=None, raster_hint=None):
    """Depth-first binning (fg_bin_prepare + fg_bin_emit_sort): the production path.
    -> (tile_keys[I] uint32-as-int32, flatten_ids[I] int32, tile_offsets[T+1] int32); the lists are
    bit-identical to ``isect_tiles`` (same (tile, depth, id) order).

    ``defer=True`` -> (tile_keys, flatten_ids, tile_offsets, finish): when the lists were enqueued
    speculatively, ``finish`` is a callable and the first two tensors are still the capacity-sized
    buffers (valid up to the count, which the kernels read on the device); the caller may enqueue
    the kernels that consume them and THEN call ``finish() -> (tile_keys, flatten_ids, redone)``,
    which waits for the count, slices, and -- if the guess was too small -- re-runs emission + sort
    on exact buffers (``redone=True``: consumers must be re-run too).  ``finish`` is None when
    nothing was deferred.  ``want_keys=False``: tile_keys comes back as None (16-bit keys stay inside
    the kernels' workspace; ``tile_keys_from_offsets`` rebuilds them on demand).
    ``keys_rects=(depth_keys[N], tile_rects[N,2][, tile_masks[N]])`` from the fused preprocess pass: the lists are binned
    from those rectangles (footprint rectangles: a subsequence of the reference's lists) -- by the supertile path, from the
    set blocks of the footprint masks only when they are given (int64 [N], fg::footprint_mask; None = whole rectangles);
    depth_keys is consumed (sorted in place) by the depth-first path.
    ``raster_hint=(channels, width, height)``: what ``rasterize_splats`` will be called with -- the supertile path then
    builds the raster job lists inside one of its own launches (``fg_stbin_fill_jobs``) and leaves them on the returned
    offsets tensor for it (``_fg_jobs``)."""
    want_keys = want_keys or tile_w * tile_h > 65536
    lib = _lib.load()
    N = means2d.shape[0]
    dev = means2d.device
    n_tiles = tile_w * tile_h
    offsets = torch.empty(n_tiles + 1, dtype=torch.int32, device=dev)
    if N == 0:
        offsets.zero_()
        return torch.empty(0, dtype=torch.int32, device=dev), torch.empty(0, dtype=torch.int32, device=dev), offsets
    rctx = current()
    static_capacity, _isect_capacity, _isect_recent = rctx.static_capacity, rctx.isect_capacity, rctx.isect_recent
    if keys_rects is not None and rctx.binning == "supertile" and lib.fg_stbin_supported(N, tile_w, tile_h):
        return _bin_tiles_supertile(rctx, "fg_stbin", N, keys_rects, tile_w, tile_h, offsets, defer, want_keys, dev,
                                    raster_hint if tile_size == TILE_SIZE else None)
    order = torch.empty(N, dtype=torch.int32, device=dev)
    cum = torch.empty(N, dtype=torch.int64, device=dev)
    ws = torch.empty(int(lib.fg_bin_prepare_workspace_bytes(N)), dtype=torch.uint8, device=dev)
    # rectangles in depth order for the emission kernel (no gathers by id there); the counts then
    # come from the rectangles, tiles_touched is only the `info` output
    rects = torch.empty(N, dtype=torch.int32, device=dev) if (tile_w <= 1023 and tile_h <= 1023) else None
    count_slot = None
    if keys_rects is not None and rects is not None:
        # the list length is also stored straight into pinned host memory by the scan's last workgroup
        # (no copy launch in the stream; not in static-shape mode, which never reads it on the host)
        count_slot, count_ptr = _count_slot() if static_capacity is None and rctx.direct_count else (None, None)
        _call("fg_bin_prepare_keys", N, _ptr(keys_rects[0]), _ptr(keys_rects[1]), _ptr(order), _ptr(cum), _ptr(rects),
              count_ptr, _ptr(ws), ws.numel(), _stream(), stage="fg_bin_prepare")  # fmt: skip
    elif rects is not None:
        _call("fg_bin_prepare_rects", N, _ptr(depths), _ptr(radii), _ptr(means2d), tile_size, tile_w, tile_h,
              _ptr(order), _ptr(cum), _ptr(rects), _ptr(ws), ws.numel(), _stream(), stage="fg_bin_prepare")  # fmt: skip
    else:
        _call("fg_bin_prepare", N, _ptr(depths), _ptr(radii), _ptr(tiles_touched), _ptr(order), _ptr(cum), _ptr(ws),
              ws.numel(), _stream())  # fmt: skip
    # The list length lives on the device.  Instead of stalling the queue on it, the emit + tile
    # sort are enqueued right away on buffers sized from the previous call of the same shape
    # (fg_bin_emit_sort_capacity reads the count on the device); the host then waits only for an
    # asynchronous readback -- the GPU keeps working meanwhile -- and repeats the call with exact
    # buffers in the rare case the guess was too small.
    if static_capacity is not None:
        # static-shape mode (graphed.GraphedRaster): fixed-size lists, no readback, no host wait at
        # all -- capturable in a hipGraph.  The lists are valid iff the device-side count fits; the
        # flag is left in `last_overflow` for the caller to check after the replay.
        cap = int(static_capacity)
        tile_keys = torch.empty(cap, dtype=torch.int32, device=dev) if want_keys else None
        flatten_ids = torch.empty(cap, dtype=torch.int32, device=dev)
        ws2 = torch.empty(int(lib.fg_bin_emit_workspace_bytes(cap)), dtype=torch.uint8, device=dev)
        _call("fg_bin_emit_sort_capacity", N, cap, _ptr(means2d), _ptr(radii), _ptr(order), _ptr(cum), _ptr(rects),
              tile_size, tile_w, tile_h, _ptr(tile_keys), _ptr(flatten_ids), _ptr(offsets), _ptr(ws2), ws2.numel(),
              _stream())  # fmt: skip
        rctx.last_overflow = cum[N - 1 :] > cap
        return (tile_keys, flatten_ids, offsets, None) if defer else (tile_keys, flatten_ids, offsets)
    key = (dev, tile_w, tile_h, keys_rects is not None)
    count_host = ready = None
    if count_slot is None:
        count_host = _count_buffer(dev)
        count_host.copy_(cum[N - 1 :], non_blocking=True)
        ready = torch.cuda.Event()
        ready.record()
    capacity = rctx.capacity_for(key, N)
    tile_keys = flatten_ids = None
    if capacity is not None:
        tile_keys = torch.empty(capacity, dtype=torch.int32, device=dev) if want_keys else None
        flatten_ids = torch.empty(capacity, dtype=torch.int32, device=dev)
        ws2 = torch.empty(int(lib.fg_bin_emit_workspace_bytes(capacity)), dtype=torch.uint8, device=dev)
        _call("fg_bin_emit_sort_capacity", N, capacity, _ptr(means2d), _ptr(radii), _ptr(order), _ptr(cum),
              _ptr(rects), tile_size, tile_w, tile_h, _ptr(tile_keys), _ptr(flatten_ids), _ptr(offsets), _ptr(ws2),
              ws2.numel(), _stream())  # fmt: skip
    def finish():
        if count_slot is not None:
            n_isects = _poll_count(count_slot)
        else:
            ready.synchronize()
            n_isects = int(count_host[0])
        _note_list_length(rctx, key, n_isects, N)
        if capacity is not None and n_isects <= capacity:
            return (tile_keys[:n_isects] if want_keys else None), flatten_ids[:n_isects], False
        if capacity is not None:
            rctx.capacity_redos += 1
        tk = torch.empty(n_isects, dtype=torch.int32, device=dev) if want_keys else None
        ids = torch.empty(n_isects, dtype=torch.int32, device=dev)
        ws3 = torch.empty(int(lib.fg_bin_emit_workspace_bytes(n_isects)), dtype=torch.uint8, device=dev)
        _call("fg_bin_emit_sort", N, n_isects, _ptr(means2d), _ptr(radii), _ptr(order), _ptr(cum), _ptr(rects),
              tile_size, tile_w, tile_h, _ptr(tk), _ptr(ids), _ptr(offsets), _ptr(ws3), ws3.numel(), _stream())  # fmt: skip
        return tk, ids, True

    if defer and capacity is not None:
        return tile_keys, flatten_ids, offsets, finish
    tk, ids, _ = finish()
    return (tk, ids, offsets, None) if defer else (tk, ids, offsets)


def _note_list_length(rctx, key, n_isects: int, N: int) -> int:
    """The list length of a call with ``N`` Gaussians of shape ``key``: checked against the int32 index range and entered
    into the history the next call's speculative capacity comes from (entries seen at another N count scaled)."""
    if n_isects >= 2**31:
        raise _lib.FgRasterError(f"{n_isects} tile intersections exceed the int32 list index range")
    cap, recent_all = rctx.isect_capacity, rctx.isect_recent
    if key not in cap and len(cap) >= 256:
        old = next(iter(cap))
        cap.pop(old)
        recent_all.pop(old, None)
        rctx.isect_n.pop(old, None)
    n_ref = rctx.isect_n.get(key)
    if n_ref is not None and not (n_ref <= 2 * N and N <= 2 * n_ref):
        recent_all.pop(key, None)  # another scene on the same tile grid: its lengths say nothing about this one
    recent = recent_all.setdefault(key, [])
    recent.append((n_isects, N))
    del recent[:-64]
    # (64: a training run draws its views at random from tens to hundreds of cameras whose lists differ by a factor of six
    # on a captured scene; with the last 16 alone the heaviest view was absent from the window more often than not and 1.3-1.9 %
    # of the steps of the end-to-end runs repeated their fill -- profiles/r06_train_e2e_soak.json)
    # 25% headroom over the heaviest of the last 64 views of this shape, rounded up to 1/32..1/16 of its magnitude: a
    # camera moving between light and heavy views neither overflows on every return nor asks the allocator for a new block
    # size every step (every list-sized buffer of the step -- ids, sort workspace, liveness words, checkpoints -- is sized
    # from this number)
    want = list_capacity_for([n if n_i == N else int(n * (N / n_i)) for n, n_i in recent])
    have = cap.get(key)
    if have is None or not (want <= have <= want + want // 4):  # (hysteresis: see RasterContext.capacity_for)
        cap[key] = want
    rctx.isect_n[key] = N
    return n_isects


def _note_counts(rctx, lkey, key, count_slot, need_reported: bool = False, N: int = 0, walks: bool = False, masked=None) -> int:
    """Wait for the three words fg_stbin_count stores into pinned host memory (list length, longest supertile segment,
    longest tile list) and update what the next calls of the shape go by: the list capacity, the long-segment flag of the
    binning, the heavy-tile policy of the raster.  -> the list length."""
    n_isects = _poll_count(count_slot)
    if masked is not None:  # (the rectangles' area: stored by the same workgroup launch as the longest segment, word 1)
        _poll_count(count_slot, 1)
        rctx.note_mask_ratio(lkey, bool(masked), n_isects, int(_count_ring_np[_RING_WORDS * count_slot + 14]))
    _note_ckpt_need(rctx, lkey, count_slot, need_reported, N, walks)  # (first: the previous call's walk report decides below)
    rctx.longest_segment_seen = _poll_count(count_slot, 1)
    for word, limit, shapes, cooldown in ((1, rctx.long_segment, rctx.long_shapes, rctx.long_cooldown),
                                          (2, rctx.heavy_lens(lkey)[0], rctx.heavy_shapes, rctx.heavy_cooldown)):  # fmt: skip
        over = _poll_count(count_slot, word) > limit or (word == 1 and _poll_count(count_slot, 3) > rctx.long_many)
        # (heavy tiles: a long list AND, where the forward reports them -- the one-call path -- a long WALK lately: a dense
        # opaque cluster has lists of ten thousand entries that close after a few hundred, and heavy tiles cost it 15 us)
        if over and word == 2 and rctx.long_walks.get(lkey, rctx.heavy_cooldown) <= 0:
            shapes.pop(lkey, None)  # (at once: nothing walked that far in the shape's last `heavy_cooldown` reporting calls)
            over = False
        if over:
            if lkey not in shapes and len(shapes) >= 256:
                shapes.pop(next(iter(shapes)))
            shapes[lkey] = cooldown
        elif lkey in shapes:
            shapes[lkey] -= 1
            if shapes[lkey] <= 0:
                del shapes[lkey]
    if len(rctx.shape_calls) > 256 and lkey not in rctx.shape_calls:
        rctx.shape_calls.pop(next(iter(rctx.shape_calls)))
    rctx.shape_calls[lkey] = rctx.shape_calls.get(lkey, 0) + 1
    longest, n_tiles = _poll_count(count_slot, 2), max(lkey[1] * lkey[2], 1)
    if len(rctx.even_calls) > 256 and lkey not in rctx.even_calls:
        rctx.even_calls.pop(next(iter(rctx.even_calls)))
    even_now = longest * n_tiles <= 3 * n_isects + 32 * n_tiles
    rctx.even_calls[lkey] = min(rctx.even_calls.get(lkey, 0) + 1, 1 << 20) if even_now else 0
    if len(rctx.uneven_left) > 256 and lkey not in rctx.uneven_left:
        rctx.uneven_left.pop(next(iter(rctx.uneven_left)))
    rctx.uneven_left[lkey] = max(rctx.uneven_left.get(lkey, 0) - 1, 0) if even_now else 8
    return _note_list_length(rctx, key, n_isects, N)


def _seg_ckpt_floats(rctx, channels, width, height, tile_size, n_list, cfgp=None):
    """Floats of the forward's compositing checkpoints for the backward's list shares; 0 = the step does without
    (off for this size / config, or beyond the context's budget)."""
    n_ck = int(_lib.load().fg_raster_seg_ckpt_floats(int(channels), int(width), int(height), int(tile_size), int(n_list),
                                                     cfgp if cfgp is not None else rctx.cfg()))  # fmt: skip
    # (64 bytes per list entry of CAPACITY -- 0.4 GB at 6M entries -- held from forward to backward: beyond the
    # context's budget the backward runs without list shares, i.e. as pixel-strip jobs)
    return n_ck if n_ck > 0 and 4 * n_ck <= rctx.seg_ckpt_budget_bytes else 0


def _plan_job_lists(rctx, raster_hint, n_list, dev, heavy=False, lkey=None, N=0):
    """(jobs[2, words], bwd_list_shares, key, cfg) for fg_stbin_fill_jobs, or None when the raster launches of this size /
    config take no lists.  ``key`` is what _RasterSplats.forward compares before it trusts the lists; ``cfg`` the address
    of the launch policy they were planned with (``heavy``: heavy tiles on; compact checkpoint slots when the shape
    ``lkey``'s needs are known), which both raster calls must be given."""
    if raster_hint is None:
        return None
    channels, width, height = (int(v) for v in raster_hint)
    n_tiles = ((width + 15) // 16) * ((height + 15) // 16)
    seg_slots = rctx.seg_slots_for(lkey, n_list, n_tiles, N) if lkey is not None and channels == 3 else 0
    cfgp = rctx.cfg(heavy, seg_slots, lkey is not None and rctx.even_shape(lkey), lkey is not None and rctx.uneven_shape(lkey),
                    rctx.heavy_lens(lkey)[1] if lkey is not None else 0)
    words = int(_lib.load().fg_raster_jobs_words(width, height, TILE_SIZE, cfgp))
    if words <= 0:
        return None
    shares = _seg_ckpt_floats(rctx, channels, width, height, TILE_SIZE, n_list, cfgp) > 0
    jobs = torch.empty(2, words, dtype=torch.int32, device=dev)
    rctx.heavy_calls += int(heavy and shares)
    rctx.full_ckpt_allocs += int(shares and rctx.compact_slots and seg_slots == 0 and not rctx.seg_slots_known and lkey is not None and channels == 3)
    return jobs, shares, (rctx, channels, width, height, TILE_SIZE), cfgp


def _bin_tiles_supertile(rctx, abi, N, keys_rects, tile_w, tile_h, offsets, defer, want_keys, dev, raster_hint=None):
    """Supertile binning (csrc/stbin.hip), same contract as ``bin_tiles``: ``fg_stbin_count`` then ``fg_stbin_fill``.
    The tile ranges are exact after the count call; the ids are filled speculatively into a buffer sized
    from the previous calls of this shape and refilled exactly if the list turned out longer.  The ranges
    handed to the consumers (``list_offsets``, the returned offsets tensor) are the tile ranges when the list
    fitted and all-zero when it did not, so a raster launch enqueued on a too-small guess walks nothing."""
    lib = _lib.load()
    depth_keys, rects = keys_rects[0], keys_rects[1]
    masks = keys_rects[2] if len(keys_rects) > 2 else None
    n_tiles = tile_w * tile_h
    tile_offsets, offsets = offsets, torch.empty_like(offsets)  # exact ranges / the ranges consumers read
    ws1 = torch.empty(int(getattr(lib, abi + "_count_workspace_bytes")(N, tile_w, tile_h)), dtype=torch.uint8, device=dev)
    static_capacity, _isect_capacity, _isect_recent = rctx.static_capacity, rctx.isect_capacity, rctx.isect_recent
    static = static_capacity is not None
    count_slot, count_ptr = (None, None) if (static or not rctx.direct_count) else _count_slot()
    _call(abi + "_count", N, _ptr(rects), _ptr(masks), tile_w, tile_h, _ptr(tile_offsets), count_ptr, _ptr(ws1), ws1.numel(),
          _stream(), stage="fg_bin_prepare")  # fmt: skip

    lkey = (dev, tile_w, tile_h)
    need_reported = [False]

    def fill(cap):
        ids = torch.empty(cap, dtype=torch.int32, device=dev)
        ws2 = torch.empty(int(getattr(lib, abi + "_fill_workspace_bytes")(cap)), dtype=torch.uint8, device=dev)
        args = (N, _ptr(depth_keys), _ptr(rects), _ptr(masks), tile_w, tile_h, cap, _ptr(tile_offsets), _ptr(ws1), _ptr(ids),
                _ptr(offsets), _ptr(ws2), ws2.numel())  # fmt: skip
        # long segments seen on this shape lately (or FG_LONG_SEGMENTS=always): the multi-workgroup sample sort
        long_mode = rctx.long_segments == "always" or (rctx.long_segments == "auto" and rctx.long_shapes.get(lkey, 0) > 0)
        flags = (_lib.STBIN_LONG_SEGMENTS | (_lib.STBIN_TEST_SMALL_SLABS if rctx.test_small_slabs else 0)) if long_mode else 0
        rctx.long_calls += int(long_mode)
        heavy = rctx.heavy_tiles == "always" or (rctx.heavy_tiles == "auto" and rctx.heavy_shapes.get(lkey, 0) > 0)
        prebuilt = _plan_job_lists(rctx, raster_hint, cap, dev, heavy, lkey, N) if rctx.jobs_in_fill else None
        if prebuilt is None:
            _call(abi + "_fill", *args, flags, _stream(), stage="fg_bin_emit_sort_capacity")
        else:
            jobs, shares, cfgp = prebuilt[0], prebuilt[1], prebuilt[3]
            need_ptr = count_ptr + 32 if count_ptr is not None else None  # (words 4..11 of the ring slot)
            need_reported[0] = need_ptr is not None and shares
            _call(abi + "_fill_jobs", *args, int(raster_hint[1]), int(raster_hint[2]), TILE_SIZE, _ptr(jobs[0]),
                  _ptr(jobs[1]), int(shares), cfgp, flags, need_ptr, _stream(), stage="fg_bin_emit_sort_capacity")  # fmt: skip
        offsets._fg_jobs = prebuilt  # (for _RasterSplats.forward; None: it builds the lists itself)
        return ids

    def keys_for(n):
        return tile_keys_from_offsets(offsets, n) if want_keys else None

    if static:
        # static-shape mode (graphed.GraphedRaster): fixed-size list, no readback; valid iff the count fits
        cap = int(static_capacity)
        flatten_ids = fill(cap)
        rctx.last_overflow = tile_offsets[n_tiles:] > cap
        tk = None  # (keys: tile_keys_from_offsets on demand)
        return (tk, flatten_ids, offsets, None) if defer else (tk, flatten_ids, offsets)
    key = (dev, tile_w, tile_h, abi)
    count_host = ready = None
    if count_slot is None:
        count_host = _count_buffer(dev)
        count_host.copy_(tile_offsets[n_tiles:], non_blocking=True)
        ready = torch.cuda.Event()
        ready.record()
    capacity = rctx.capacity_for(key, N)
    flatten_ids = fill(capacity) if capacity is not None else None
    if raster_hint is not None and rctx.step_calls:
        rctx.stagewise_raster_calls += 1

    def finish():
        if count_slot is not None:
            n_isects = _note_counts(rctx, lkey, key, count_slot, need_reported[0], N, masked=masks is not None)
        else:
            ready.synchronize()
            n_isects = _note_list_length(rctx, key, int(count_host[0]), N)
        if capacity is not None and n_isects <= capacity:
            return keys_for(n_isects), flatten_ids[:n_isects], False
        if capacity is not None:
            rctx.capacity_redos += 1
        if n_isects == 0:
            offsets.zero_()
            return keys_for(0), torch.empty(0, dtype=torch.int32, device=dev), capacity is not None
        ids = fill(n_isects)  # (writes `offsets`, which keys_for reads: fill first)
        return keys_for(n_isects), ids, True

    if defer and capacity is not None:
        return None, flatten_ids, offsets, finish
    tk, ids, _ = finish()
    return (tk, ids, offsets, None) if defer else (tk, ids, offsets)


@torch.no_grad()
def densify_stats(absgrad, radii, max_dim: float, xys_grad_norm, vis_counts, max_2dsize) -> None:
    """after_train_iter's statistics (freegaussian_model.py:369-392) in one pass, in place: for visible Gaussians
    (radii > 0) xys_grad_norm += |absgrad|, vis_counts += 1, max_2dsize = max(max_2dsize, radii / max_dim)."""
    N = radii.numel()
    if absgrad.shape != (N, 2) or any(t.numel() != N for t in (xys_grad_norm, vis_counts, max_2dsize)):
        raise ValueError("absgrad [N,2], radii [N] and three [N] statistics expected")
    for t in (xys_grad_norm, vis_counts, max_2dsize):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise ValueError("the statistics are updated in place: contiguous float32 GPU tensors")
    _call("fg_densify_stats", N, _ptr(_f32(absgrad, "absgrad")), _ptr(radii.to(torch.int32).contiguous()), float(max_dim),
          _ptr(xys_grad_norm), _ptr(vis_counts), _ptr(max_2dsize), _stream())  # fmt: skip


class _L1Ssim(torch.autograd.Function):
    """mean |gt - pred| and mean SSIM(gt, pred) of two [H,W,C] images, one launch each way (csrc/loss.hip)."""

    @staticmethod
    def forward(ctx, pred, gt):
        H, W, C = pred.shape
        dev = pred.device
        lib = _lib.load()
        n_ws = int(lib.fg_l1_ssim_workspace_floats(H, W, C))
        if n_ws == 0:
            raise ValueError(f"SSIM needs images larger than its 11 x 11 window, got {H} x {W}")
        maps = torch.empty(3, C, H - 10, W - 10, dtype=torch.float32, device=dev)
        ws = torch.empty(n_ws, dtype=torch.float32, device=dev)
        out = torch.empty(2, dtype=torch.float32, device=dev)
        _call("fg_l1_ssim_fwd", H, W, C, _ptr(pred), _ptr(gt), _ptr(maps), _ptr(ws), n_ws, _ptr(out), _stream(),
              stage="fg_l1_ssim_fwd")  # fmt: skip
        ctx.save_for_backward(pred, gt, maps)
        return out[0], out[1]

    @staticmethod
    def backward(ctx, v_l1, v_ssim):
        pred, gt, maps = ctx.saved_tensors
        H, W, C = pred.shape
        zero = pred.new_zeros(())
        v = torch.stack([zero if v_l1 is None else v_l1.float().reshape(()), zero if v_ssim is None else v_ssim.float().reshape(())])
        v_pred = torch.empty_like(pred)
        _call("fg_l1_ssim_bwd", H, W, C, _ptr(pred), _ptr(gt), _ptr(maps), _ptr(v), _ptr(v_pred), _stream(),
              stage="fg_l1_ssim_bwd")  # fmt: skip
        return v_pred, None


def l1_ssim(pred: torch.Tensor, gt: torch.Tensor):
    """(mean |gt - pred|, mean SSIM(gt, pred)) of two [H,W,C] float32 images on the GPU -- the two terms of the
    reference's main loss (freegaussian_model.py:965-981, SSIM as pytorch_msssim.SSIM(data_range=1.0, size_average=True):
    11 x 11 Gaussian window, sigma 1.5, 'valid' borders).  Differentiable with respect to ``pred`` only."""
    if pred.dim() != 3 or pred.shape != gt.shape:
        raise ValueError(f"two [H,W,C] images expected, got {tuple(pred.shape)} and {tuple(gt.shape)}")
    if not pred.is_cuda:
        raise _lib.FgRasterError("l1_ssim runs on the GPU only (harness.ssim is the torch statement of the same loss)")
    return _L1Ssim.apply(_f32(pred, "pred"), _f32(gt.detach(), "gt"))


@torch.no_grad()
def isect_keys(tile_keys, flatten_ids, depths):
    """Reference-style 64-bit keys (tile << 32 | depth bits) of the sorted list."""
    n = tile_keys.numel()
    out = torch.empty(n, dtype=torch.int64, device=tile_keys.device)
    _call("fg_isect_keys", n, _ptr(tile_keys), _ptr(flatten_ids), _ptr(depths), _ptr(out), _stream())
    return out


@torch.no_grad()
def sort_pairs32(keys: torch.Tensor, vals: torch.Tensor, end_bit: int = 32) -> None:
    """In-place stable radix sort of (uint32-as-int32 key, int32 value) pairs on bits [0, end_bit)."""
    lib = _lib.load()
    n = keys.numel()
    assert keys.dtype == torch.int32 and vals.dtype == torch.int32 and vals.numel() == n and keys.is_cuda
    ws = torch.empty(max(int(lib.fg_sort32_workspace_bytes(n)), 8), dtype=torch.uint8, device=keys.device)
    _call("fg_sort_pairs32", n, _ptr(keys), _ptr(vals), int(end_bit), _ptr(ws), ws.numel(), _stream())


@torch.no_grad()
def sort_pairs(keys: torch.Tensor, vals: torch.Tensor, end_bit: int = 64) -> None:
    """In-place stable radix sort of (int64 key, int32 value) pairs on key bits [0, end_bit)."""
    lib = _lib.load()
    n = keys.numel()
    assert keys.dtype == torch.int64 and vals.dtype == torch.int32 and vals.numel() == n
    assert keys.is_cuda and keys.is_contiguous() and vals.is_contiguous()
    ws = torch.empty(max(int(lib.fg_sort_workspace_bytes(n)), 8), dtype=torch.uint8, device=keys.device)
    _call("fg_sort_pairs", n, _ptr(keys), _ptr(vals), int(end_bit), _ptr(ws), ws.numel(), _stream())  # fmt: skip


# --------------------------------------------------------------------------------------------
# K5 / K6 compositing


class _Rasterize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means2d, conics, features, opacities, width, height, tile_size, tile_offsets, flatten_ids,
                absgrad):  # fmt: skip
        ctx.rctx = current()  # the backward (autograd's thread) reads the same context
        lib = _lib.load()
        m2 = means2d.reshape(-1, 2)
        N, C = features.shape
        dev = features.device
        splats = torch.empty(N, SPLAT_FLOATS, dtype=torch.float32, device=dev)
        _call("fg_pack_splats", N, C, _ptr(m2), _ptr(conics), _ptr(opacities), _ptr(features), _ptr(splats), _stream())  # fmt: skip
        render = torch.empty(height, width, C, dtype=torch.float32, device=dev)
        alphas = torch.empty(height, width, 1, dtype=torch.float32, device=dev)
        last_ids = torch.empty(height, width, dtype=torch.int32, device=dev)
        _call("fg_raster_fwd", C, width, height, tile_size, _ptr(splats), _ptr(tile_offsets), _ptr(flatten_ids),
                              _ptr(render), _ptr(alphas), _ptr(last_ids), ctx.rctx.cfg(), _stream())  # fmt: skip
        ctx.save_for_backward(splats, tile_offsets, flatten_ids, alphas, last_ids)
        ctx.geom = (N, C, width, height, tile_size, absgrad, tuple(means2d.shape))
        ctx.means2d_ref = means2d if absgrad else None  # the object that receives .absgrad
        ctx.mark_non_differentiable(last_ids)
        return render, alphas, last_ids

    @staticmethod
    def backward(ctx, *grads):
        with use(ctx.rctx):
            return _Rasterize._backward(ctx, *grads)

    @staticmethod
    def _backward(ctx, v_render, v_alphas, _v_last):
        lib = _lib.load()
        splats, tile_offsets, flatten_ids, alphas, last_ids = ctx.saved_tensors
        N, C, width, height, tile_size, absgrad, m2_shape = ctx.geom
        dev = splats.device
        v_render = torch.zeros_like(alphas).expand(height, width, C) if v_render is None else v_render
        v_alphas = torch.zeros_like(alphas) if v_alphas is None else v_alphas
        v_splats = torch.zeros(N, SPLAT_FLOATS, dtype=torch.float32, device=dev)
        _call("fg_raster_bwd", C, width, height, tile_size, _ptr(splats), _ptr(tile_offsets), _ptr(flatten_ids),
                              _ptr(alphas), _ptr(last_ids), _ptr(v_render.contiguous()),
                              _ptr(v_alphas.contiguous()), _ptr(v_splats), ctx.rctx.cfg(), _stream())  # fmt: skip
        v_means2d = torch.empty(N, 2, dtype=torch.float32, device=dev)
        v_abs = torch.empty(N, 2, dtype=torch.float32, device=dev) if absgrad else None
        v_conics = torch.empty(N, 3, dtype=torch.float32, device=dev)
        v_opac = torch.empty(N, dtype=torch.float32, device=dev)
        v_feat = torch.empty(N, C, dtype=torch.float32, device=dev)
        _call("fg_unpack_grads", N, C, _ptr(v_splats), _ptr(v_means2d), _ptr(v_abs), _ptr(v_conics), _ptr(v_opac),
                                _ptr(v_feat), _stream())  # fmt: skip
        if absgrad and ctx.means2d_ref is not None:
            # side channel read by the densification heuristics
            # (reference freegaussian_model.py:377 `self.xys.absgrad`)
            ctx.means2d_ref.absgrad = v_abs.reshape(m2_shape)
            ctx.means2d_ref = None
        return v_means2d.reshape(m2_shape), v_conics, v_feat, v_opac, None, None, None, None, None, None


def rasterize_to_pixels(means2d, conics, features, opacities, width, height, tile_size, tile_offsets, flatten_ids,
                        absgrad=False):  # fmt: skip
    """-> render[H,W,C], alphas[H,W,1], last_ids[H,W].  means2d may be [N,2] or [1,N,2]; the
    very tensor object passed here receives ``.absgrad`` after backward."""
    if tile_size != TILE_SIZE:
        raise ValueError("tile_size must be 16")
    C = features.shape[1]
    if not 1 <= C <= MAX_CHANNELS:
        raise ValueError(f"1..{MAX_CHANNELS} composited channels supported, got {C}")
    if not means2d.is_cuda:
        raise _lib.FgRasterError("rasterize_to_pixels needs CUDA/HIP tensors: no CPU fallback")
    return _Rasterize.apply(means2d, conics.contiguous(), features.contiguous(), opacities.contiguous(), int(width),
                            int(height), int(tile_size), tile_offsets, flatten_ids, bool(absgrad))  # fmt: skip


# --------------------------------------------------------------------------------------------
# Fused path: K1+K2+pack in one pass, raster on records, unpack+K8+K7 in one pass

def _alloc_grad(t: torch.Tensor) -> torch.Tensor:
    grad_alloc = current().grad_alloc  # (RasterContext.grad_alloc: gradients written straight into the caller's buffers)
    if grad_alloc is not None:
        buf = grad_alloc(t)
        if buf is not None:
            return buf
    return torch.empty_like(t)


# side streams of the optional two-stream forward (RasterContext.overlap_pack); the records carry the event
# their consumer (the raster forward) must wait for
_side_streams: dict = {}


def _side_stream(dev) -> torch.cuda.Stream:
    s = _side_streams.get(dev)
    if s is None:
        s = _side_streams[dev] = torch.cuda.Stream(device=dev)
    return s


def _wait_ready(t: torch.Tensor) -> None:
    """Make the current stream wait for the side-stream kernel that produces ``t`` (if any)."""
    ev = getattr(t, "_fg_ready", None)
    if ev is not None:
        torch.cuda.current_stream().wait_event(ev)


def _viewmat_grad(raw, means, quats, d_quats, scales, d_scales, opacities, colors, features_rest, sh_degree, k_stored, n_color,
                  with_depth, n_extra, viewmat, K, width, height, eps2d, antialiased, radii, v_splats, v_means2d, m2_stride,
                  v_depths, v_conics, sh_jac):  # fmt: skip
    """dL/d viewmat [4,4] of a view (fg_viewmat_bwd) from the cotangents of its per-Gaussian backward: the direct paths
    through the camera-space mean and covariance, plus the pull-back of dL/d campos -- the SH view direction's share --
    through campos = inverse(viewmat)[:3, 3] (torch, a 4 x 4 matter), as gsplat's autograd routes it."""
    lib = _lib.load()
    N, dev = means.shape[0], means.device
    out = torch.empty(19, dtype=torch.float32, device=dev)
    ws = torch.empty(max(int(lib.fg_viewmat_bwd_workspace_bytes(N)), 16), dtype=torch.uint8, device=dev)
    _call("fg_viewmat_bwd", N, int(raw), _ptr(means), _ptr(quats), _ptr(d_quats), _ptr(scales), _ptr(d_scales), _ptr(opacities),
          _ptr(colors), _ptr(features_rest), int(sh_degree), int(k_stored), int(n_color), int(with_depth), int(n_extra),
          _ptr(viewmat), _ptr(K), int(width), int(height), float(eps2d), int(antialiased), _ptr(radii), _ptr(v_splats),
          _ptr(v_means2d), int(m2_stride), _ptr(v_depths), _ptr(v_conics), _ptr(sh_jac), _ptr(out), _ptr(ws), ws.numel(),
          _stream())  # fmt: skip
    v = out[:16].view(4, 4)
    if sh_degree >= 1:
        with torch.enable_grad():
            vm = viewmat.detach().clone().requires_grad_(True)
            campos = torch.linalg.inv(vm)[:3, 3]
            (pull,) = torch.autograd.grad(campos, vm, out[16:19])
        v = v + pull
    return v


class _Preprocess(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means, quats, scales, opacities, colors, extra, viewmat, K, cfg):
        ctx.rctx = current()  # the backward (autograd's thread) reads the same context
        (width, height, eps2d, near, far, radius_clip, tile_size, antialiased, sh_degree, with_depth) = cfg[:10]
        N = means.shape[0]
        dev = means.device
        if sh_degree >= 0:
            k_stored, n_color = colors.shape[1], 3
        else:
            k_stored, n_color = 0, (0 if colors is None else colors.shape[1])
        n_extra = 0 if extra is None else extra.shape[1]
        radii = torch.empty(N, dtype=torch.int32, device=dev)
        means2d = torch.empty(N, 2, dtype=torch.float32, device=dev)
        depths = torch.empty(N, dtype=torch.float32, device=dev)
        conics = torch.empty(N, 3, dtype=torch.float32, device=dev)
        comp = torch.empty(N, dtype=torch.float32, device=dev) if antialiased else None
        tiles = torch.empty(N, dtype=torch.int32, device=dev)
        splats = torch.empty(N, SPLAT_FLOATS, dtype=torch.float32, device=dev)
        depth_keys, tile_rects, tile_masks = _binning_side_outputs(N, tile_size, width, height, dev)
        if ctx.rctx.color_grad_sink is not None and sh_degree >= 0 and colors is not None:
            ctx.rctx.color_grad_sink("view", viewmat, dev)  # (view-DP: the factored exchange prepares its payload)
        sh_jac = None  # the forward's note for the backward of the SH colour (include/fgraster.h, fg_preprocess_fwd)
        if ctx.rctx.overlap_pack and len(cfg) > 10 and cfg[10]:
            main = torch.cuda.current_stream()
            side = _side_stream(dev)
            _call("fg_project_fwd", N, _ptr(means), _ptr(quats), _ptr(scales), _ptr(viewmat), _ptr(K), width, height,
                  eps2d, near, far, radius_clip, tile_size, _ptr(radii), _ptr(means2d), _ptr(depths), _ptr(conics),
                  _ptr(comp), _ptr(tiles), _stream())  # fmt: skip
            projected = torch.cuda.Event()
            projected.record(main)
            side.wait_event(projected)
            _call("fg_sh_pack_fwd", N, _ptr(means), _ptr(opacities), _ptr(colors), sh_degree, k_stored, n_color,
                  int(with_depth), _ptr(extra), n_extra, _ptr(viewmat), int(antialiased), _ptr(radii), _ptr(means2d),
                  _ptr(depths), _ptr(conics), _ptr(comp), _ptr(splats), side.cuda_stream, on=side)  # fmt: skip
            ready = torch.cuda.Event()
            ready.record(side)
            for t in (splats, means, opacities, colors, extra, viewmat, radii, means2d, depths, conics, comp):
                if t is not None:
                    t.record_stream(side)  # the caching allocator must not recycle them under the side kernel
            splats._fg_ready = ready
        else:
            if sh_degree >= 1 and ctx.rctx.sh_jacobian:
                sh_jac = torch.empty(N, _lib.SH_JAC_FLOATS, dtype=torch.float32, device=dev)
            _call("fg_preprocess_fwd", N, _ptr(means), _ptr(quats), _ptr(scales), _ptr(opacities), _ptr(colors),
                  sh_degree, k_stored, n_color, int(with_depth), _ptr(extra), n_extra, _ptr(viewmat), _ptr(K), width,
                  height, eps2d, near, far, radius_clip, tile_size, int(antialiased), _ptr(radii), _ptr(means2d),
                  _ptr(depths), _ptr(conics), _ptr(comp), _ptr(tiles), _ptr(splats), _ptr(depth_keys),
                  _ptr(tile_rects), _ptr(tile_masks), _ptr(sh_jac), _stream())  # fmt: skip
            if depth_keys is not None:
                splats._fg_bin = (depth_keys, tile_rects, tile_masks)  # for ops.bin_tiles(keys_rects=...)
        ctx.save_for_backward(means, quats, scales, opacities, colors, extra, viewmat, K, radii, sh_jac)
        ctx.set_materialize_grads(False)  # unused depths / conics gradients arrive as None, not as zero tensors
        ctx.cfg = cfg
        ctx.layout = (k_stored, n_color, n_extra)
        ctx.mark_non_differentiable(radii, tiles)
        return radii, means2d, depths, conics, tiles, splats

    @staticmethod
    def backward(ctx, *grads):
        with use(ctx.rctx):
            return _Preprocess._backward(ctx, *grads)

    @staticmethod
    def _backward(ctx, _v_radii, v_means2d, v_depths, v_conics, _v_tiles, v_splats):
        means, quats, scales, opacities, colors, extra, viewmat, K, radii, sh_jac = ctx.saved_tensors
        (width, height, eps2d, near, far, radius_clip, tile_size, antialiased, sh_degree, with_depth) = ctx.cfg[:10]
        k_stored, n_color, n_extra = ctx.layout
        N = means.shape[0]
        dev = means.device
        if v_splats is None:
            v_splats = torch.zeros(N, SPLAT_FLOATS, dtype=torch.float32, device=dev)
        if v_means2d is None:
            v_means2d = torch.zeros(N, 2, dtype=torch.float32, device=dev)
        v_means, v_quats, v_scales = _alloc_grad(means), _alloc_grad(quats), _alloc_grad(scales)
        v_opac = _alloc_grad(opacities)
        v_extra = torch.empty_like(extra) if extra is not None else None
        # the xy gradient usually IS the xy slots of a record array (strided view made by
        # _RasterSplats.backward): hand the kernel pointer + stride instead of compacting it
        v_means2d = v_means2d.reshape(N, 2)
        if v_means2d.stride(1) == 1 and v_means2d.stride(0) >= 2:
            m2_stride = v_means2d.stride(0)
        else:
            v_means2d, m2_stride = v_means2d.contiguous(), 2
        v_splats = v_splats.contiguous()

        def pose_grad():  # (the camera pose requires a gradient: a pass of its own over the same cotangents)
            if not ctx.needs_input_grad[6]:
                return None
            return _viewmat_grad(False, means, quats, None, scales, None, opacities, colors, None, sh_degree, k_stored, n_color,
                                 with_depth, n_extra, viewmat, K, width, height, eps2d, antialiased, radii, v_splats, v_means2d,
                                 m2_stride, None if v_depths is None else v_depths.contiguous(),
                                 None if v_conics is None else v_conics.contiguous(), sh_jac)  # fmt: skip

        color_grad_sink = ctx.rctx.color_grad_sink
        if color_grad_sink is not None and sh_degree >= 0 and colors is not None:
            # factored form: 12 B of colour gradient per Gaussian instead of the 192-B coefficient row
            v_rgb = color_grad_sink("alloc", N, means.device)  # [N,3] or [N,6] (g | unit view direction)
            v_dep = None if v_depths is None else v_depths.contiguous().reshape(N)
            v_con = None if v_conics is None else v_conics.contiguous().reshape(N, 3)
            # The sink may ask for the pass in k launches over ranges of N (every Gaussian's row is independent of the others):
            # it is told after each launch that the range's gradients are final -- the view-DP exchange starts reducing them
            # while the later ranges are still being computed.  Range bounds on multiples of 1024 rows (16-byte aligned rows
            # of every array).
            k = int(color_grad_sink("slices", N) or 1)
            bounds = [0, N] if k <= 1 or N < 4096 * k else sorted({0, N} | {(N * i // k) // 1024 * 1024 for i in range(1, k)})

            def rows(t, n0, n1):
                return None if t is None else t[n0:n1]

            for n0, n1 in zip(bounds, bounds[1:]):
                r = lambda t: _ptr(rows(t, n0, n1))  # noqa: E731
                _call("fg_preprocess_bwd_factored", n1 - n0, r(means), r(quats), r(scales), r(opacities),
                      r(colors), sh_degree, k_stored, int(with_depth), n_extra, _ptr(viewmat), _ptr(K), width, height,
                      eps2d, int(antialiased), r(radii), r(v_splats), r(v_means2d), m2_stride, r(v_dep), r(v_con),
                      r(v_means), r(v_quats), r(v_scales), r(v_opac), r(v_rgb), int(v_rgb.shape[1]), r(v_extra), r(sh_jac),
                      _stream())  # fmt: skip
                if len(bounds) > 2:
                    color_grad_sink("slice", n0, n1)
            color_grad_sink("ready", v_rgb, means, viewmat, sh_degree, colors)
            return v_means, v_quats, v_scales, v_opac, None, v_extra, pose_grad(), None, None
        v_colors = _alloc_grad(colors) if colors is not None else None
        _call("fg_preprocess_bwd", N, _ptr(means), _ptr(quats), _ptr(scales), _ptr(opacities), _ptr(colors),
              sh_degree, k_stored, n_color, int(with_depth), n_extra, _ptr(viewmat), _ptr(K), width, height, eps2d,
              int(antialiased), _ptr(radii), _ptr(v_splats.contiguous()), _ptr(v_means2d), m2_stride,
              _ptr(None if v_depths is None else v_depths.contiguous()),
              _ptr(None if v_conics is None else v_conics.contiguous()), _ptr(v_means), _ptr(v_quats),
              _ptr(v_scales), _ptr(v_opac), _ptr(v_colors), _ptr(v_extra), _ptr(sh_jac), _stream())  # fmt: skip
        return v_means, v_quats, v_scales, v_opac, v_colors, v_extra, pose_grad(), None, None


def preprocess(means, quats, scales, opacities, colors, extra, viewmat, K, width, height, eps2d=0.3,
               near_plane=0.01, far_plane=1e10, radius_clip=0.0, tile_size=TILE_SIZE, antialiased=False,
               sh_degree=-1, with_depth=False, overlap=False):  # fmt: skip
    """Fused projection + colour + record packing.
    -> radii[N], means2d[N,2], depths[N], conics[N,3], tiles_touched[N], splats[N,16].
    ``overlap=True``: the records are produced on a side stream (concurrently with whatever the
    caller enqueues next on the current stream, i.e. the binning); they carry the event
    ``rasterize_splats`` waits for -- any other reader must call ``ops._wait_ready(splats)`` first."""
    means, quats, scales = _f32(means, "means"), _f32(quats, "quats"), _f32(scales, "scales")
    opacities, viewmat, K = _f32(opacities, "opacities"), _f32(viewmat, "viewmat"), _f32(K, "K")
    colors = None if colors is None else _f32(colors, "colors")
    extra = None if extra is None else _f32(extra, "extra_channels")
    cfg = (int(width), int(height), float(eps2d), float(near_plane), float(far_plane), float(radius_clip),
           int(tile_size), bool(antialiased), int(sh_degree), bool(with_depth), bool(overlap))  # fmt: skip
    return _Preprocess.apply(means, quats, scales, opacities, colors, extra, viewmat, K, cfg)


class _PreprocessRaw(torch.autograd.Function):
    """fg_preprocess_raw_*: the same pass on the reference's raw gauss_params (log-scales, opacity
    logits, unnormalised quats, split SH features) with optional deltas from the deform / control
    MLPs (freegaussian_model.py:801, :844-851); activations and their chain rule run in-kernel."""

    @staticmethod
    def forward(ctx, means, quats, d_quats, log_scales, d_scales, opacity_logits, features_dc, features_rest,
                extra, viewmat, K, cfg):  # fmt: skip
        ctx.rctx = current()  # the backward (autograd's thread) reads the same context
        (width, height, eps2d, near, far, radius_clip, tile_size, antialiased, sh_degree, with_depth) = cfg
        N = means.shape[0]
        dev = means.device
        k_stored = 1 + features_rest.shape[1]
        n_extra = 0 if extra is None else extra.shape[1]
        radii = torch.empty(N, dtype=torch.int32, device=dev)
        means2d = torch.empty(N, 2, dtype=torch.float32, device=dev)
        depths = torch.empty(N, dtype=torch.float32, device=dev)
        conics = torch.empty(N, 3, dtype=torch.float32, device=dev)
        comp = torch.empty(N, dtype=torch.float32, device=dev) if antialiased else None
        tiles = torch.empty(N, dtype=torch.int32, device=dev)
        splats = torch.empty(N, SPLAT_FLOATS, dtype=torch.float32, device=dev)
        depth_keys, tile_rects, tile_masks = _binning_side_outputs(N, tile_size, width, height, dev)
        sh_jac = None
        if sh_degree >= 1 and ctx.rctx.sh_jacobian:
            sh_jac = torch.empty(N, _lib.SH_JAC_FLOATS, dtype=torch.float32, device=dev)
        if ctx.rctx.color_grad_sink is not None:
            ctx.rctx.color_grad_sink("view", viewmat, dev)  # (view-DP: the factored exchange prepares its payload)
        _call("fg_preprocess_raw_fwd", N, _ptr(means), _ptr(quats), _ptr(d_quats), _ptr(log_scales), _ptr(d_scales),
              _ptr(opacity_logits), _ptr(features_dc), _ptr(features_rest), sh_degree, k_stored, int(with_depth),
              _ptr(extra), n_extra, _ptr(viewmat), _ptr(K), width, height, eps2d, near, far, radius_clip, tile_size,
              int(antialiased), _ptr(radii), _ptr(means2d), _ptr(depths), _ptr(conics), _ptr(comp), _ptr(tiles),
              _ptr(splats), _ptr(depth_keys), _ptr(tile_rects), _ptr(tile_masks), _ptr(sh_jac), _stream())  # fmt: skip
        if depth_keys is not None:
            splats._fg_bin = (depth_keys, tile_rects, tile_masks)
        ctx.save_for_backward(means, quats, d_quats, log_scales, d_scales, opacity_logits, features_dc,
                              features_rest, extra, viewmat, K, radii, sh_jac)  # fmt: skip
        ctx.set_materialize_grads(False)
        ctx.cfg = cfg
        ctx.mark_non_differentiable(radii, tiles)
        return radii, means2d, depths, conics, tiles, splats

    @staticmethod
    def backward(ctx, *grads):
        with use(ctx.rctx):
            return _PreprocessRaw._backward(ctx, *grads)

    @staticmethod
    def _backward(ctx, _v_radii, v_means2d, v_depths, v_conics, _v_tiles, v_splats):
        (means, quats, d_quats, log_scales, d_scales, opacity_logits, features_dc, features_rest, extra, viewmat, K,
         radii, sh_jac) = ctx.saved_tensors  # fmt: skip
        (width, height, eps2d, near, far, radius_clip, tile_size, antialiased, sh_degree, with_depth) = ctx.cfg
        N = means.shape[0]
        dev = means.device
        k_stored = 1 + features_rest.shape[1]
        n_extra = 0 if extra is None else extra.shape[1]
        if v_splats is None:
            v_splats = torch.zeros(N, SPLAT_FLOATS, dtype=torch.float32, device=dev)
        if v_means2d is None:
            v_means2d = torch.zeros(N, 2, dtype=torch.float32, device=dev)
        v_means, v_quats, v_ls = _alloc_grad(means), _alloc_grad(quats), _alloc_grad(log_scales)
        v_ol = _alloc_grad(opacity_logits)
        v_dq = torch.empty_like(d_quats) if d_quats is not None else None
        v_ds = torch.empty_like(d_scales) if d_scales is not None else None
        v_extra = torch.empty_like(extra) if extra is not None else None
        v_means2d = v_means2d.reshape(N, 2)
        if v_means2d.stride(1) == 1 and v_means2d.stride(0) >= 2:
            m2_stride = v_means2d.stride(0)
        else:
            v_means2d, m2_stride = v_means2d.contiguous(), 2
        v_splats = v_splats.contiguous()

        def pose_grad():
            if not ctx.needs_input_grad[9]:
                return None
            return _viewmat_grad(True, means, quats, d_quats, log_scales, d_scales, opacity_logits, features_dc, features_rest,
                                 sh_degree, k_stored, 3, with_depth, n_extra, viewmat, K, width, height, eps2d, antialiased, radii,
                                 v_splats, v_means2d, m2_stride, None if v_depths is None else v_depths.contiguous(),
                                 None if v_conics is None else v_conics.contiguous(), sh_jac)  # fmt: skip

        color_grad_sink = ctx.rctx.color_grad_sink
        if color_grad_sink is not None:
            # factored form (viewdp.ModelViewDP): 12 / 24 B of colour gradient per Gaussian instead of the two coefficient
            # gradients (12 + 180 B); features_dc.grad / features_rest.grad are filled by the exchange, not by autograd
            v_rgb = color_grad_sink("alloc", N, dev, means)
            _call("fg_preprocess_raw_bwd_factored", N, _ptr(means), _ptr(quats), _ptr(d_quats), _ptr(log_scales),
                  _ptr(d_scales), _ptr(opacity_logits), _ptr(features_dc), _ptr(features_rest), sh_degree, k_stored,
                  int(with_depth), n_extra, _ptr(viewmat), _ptr(K), width, height, eps2d, int(antialiased), _ptr(radii),
                  _ptr(v_splats.contiguous()), _ptr(v_means2d), m2_stride,
                  _ptr(None if v_depths is None else v_depths.contiguous()),
                  _ptr(None if v_conics is None else v_conics.contiguous()), _ptr(v_means), _ptr(v_quats), _ptr(v_dq),
                  _ptr(v_ls), _ptr(v_ds), _ptr(v_ol), _ptr(v_rgb), int(v_rgb.shape[1]), _ptr(v_extra), _ptr(sh_jac),
                  _stream())  # fmt: skip
            color_grad_sink("ready", v_rgb, means, viewmat, sh_degree, k_stored)
            return v_means, v_quats, v_dq, v_ls, v_ds, v_ol, None, None, v_extra, pose_grad(), None, None
        v_dc, v_rest = _alloc_grad(features_dc), _alloc_grad(features_rest)
        _call("fg_preprocess_raw_bwd", N, _ptr(means), _ptr(quats), _ptr(d_quats), _ptr(log_scales), _ptr(d_scales),
              _ptr(opacity_logits), _ptr(features_dc), _ptr(features_rest), sh_degree, k_stored, int(with_depth),
              n_extra, _ptr(viewmat), _ptr(K), width, height, eps2d, int(antialiased), _ptr(radii),
              _ptr(v_splats.contiguous()), _ptr(v_means2d), m2_stride,
              _ptr(None if v_depths is None else v_depths.contiguous()),
              _ptr(None if v_conics is None else v_conics.contiguous()), _ptr(v_means), _ptr(v_quats), _ptr(v_dq),
              _ptr(v_ls), _ptr(v_ds), _ptr(v_ol), _ptr(v_dc), _ptr(v_rest), _ptr(v_extra), _ptr(sh_jac),
              _stream())  # fmt: skip
        return v_means, v_quats, v_dq, v_ls, v_ds, v_ol, v_dc, v_rest, v_extra, pose_grad(), None, None


def preprocess_raw(means, quats, log_scales, opacity_logits, features_dc, features_rest, viewmat, K, width, height,
                   sh_degree, d_quats=None, d_scales=None, extra=None, eps2d=0.3, near_plane=0.01, far_plane=1e10,
                   radius_clip=0.0, tile_size=TILE_SIZE, antialiased=False, with_depth=False):  # fmt: skip
    """Fused activations + projection + SH colour + record packing on raw gauss_params.
    features_dc [N,3], features_rest [N,K-1,3], opacity_logits [N] or [N,1]; d_quats [N,4] /
    d_scales [N,3] optional deltas added after normalisation / exp.
    -> radii[N], means2d[N,2], depths[N], conics[N,3], tiles_touched[N], splats[N,16]."""
    means, quats, log_scales = _f32(means, "means"), _f32(quats, "quats"), _f32(log_scales, "scales")
    opacity_logits = _f32(opacity_logits.reshape(-1), "opacities")
    features_dc, features_rest = _f32(features_dc, "features_dc"), _f32(features_rest, "features_rest")
    viewmat, K = _f32(viewmat, "viewmat"), _f32(K, "K")
    N = means.shape[0]
    if features_dc.shape != (N, 3) or features_rest.dim() != 3 or features_rest.shape[0] != N:
        raise ValueError("features_dc[N,3] features_rest[N,K-1,3] expected")
    if sh_degree is None or not 0 <= sh_degree <= 3 or (sh_degree + 1) ** 2 > 1 + features_rest.shape[1]:
        raise ValueError("the raw-parameter path needs 0 <= sh_degree <= 3 within the stored coefficients")
    d_quats = None if d_quats is None else _f32(d_quats, "d_quats")
    d_scales = None if d_scales is None else _f32(d_scales, "d_scales")
    extra = None if extra is None else _f32(extra, "extra_channels")
    cfg = (int(width), int(height), float(eps2d), float(near_plane), float(far_plane), float(radius_clip),
           int(tile_size), bool(antialiased), int(sh_degree), bool(with_depth))  # fmt: skip
    return _PreprocessRaw.apply(means, quats, d_quats, log_scales, d_scales, opacity_logits, features_dc,
                                features_rest, extra, viewmat, K, cfg)  # fmt: skip


class _RasterSplats(torch.autograd.Function):
    """Compositing on packed records.  ``means2d`` is a routing input only: its gradient carries
    the xy slots of the record gradient (so ``info["means2d"].grad`` exists, as with gsplat) and
    the very tensor object receives ``.absgrad``; fg_preprocess_bwd ignores the xy slots of
    v_splats accordingly."""

    @staticmethod
    def forward(ctx, splats, means2d, channels, width, height, tile_size, tile_offsets, flatten_ids, absgrad,
                background=None, n_clamp=0, expect_backward=False):  # fmt: skip
        ctx.rctx = current()  # the backward (autograd's thread) reads the same context
        dev = splats.device
        _wait_ready(splats)
        render = torch.empty(height, width, channels, dtype=torch.float32, device=dev)
        alphas = torch.empty(height, width, 1, dtype=torch.float32, device=dev)
        last_ids = torch.empty(height, width, dtype=torch.int32, device=dev)
        composite = background is not None or n_clamp > 0
        clamp_mask = torch.empty(height, width, dtype=torch.uint8, device=dev) if n_clamp > 0 else None
        # job lists (content-aware job sizes of the mixed launches); 0 words = classic launches.  The launch policy is
        # the one the lists bin_tiles left were planned with (heavy tiles on or off), else the context's
        prebuilt = getattr(tile_offsets, "_fg_jobs", None)
        key = (ctx.rctx, int(channels), int(width), int(height), int(tile_size))
        if prebuilt is not None and prebuilt[2] != key:
            prebuilt = None
        cfgp = prebuilt[3] if prebuilt is not None else ctx.rctx.cfg()
        words = int(_lib.load().fg_raster_jobs_words(int(width), int(height), int(tile_size), cfgp))
        jobs = seg_ckpt = live = v_splats = None
        if words > 0:
            # list segments of the backward: per-pixel compositing checkpoints written by the forward
            n_ck = _seg_ckpt_floats(ctx.rctx, channels, width, height, tile_size, flatten_ids.numel(), cfgp)
            if n_ck > 0:
                seg_ckpt = torch.empty(n_ck, dtype=torch.float32, device=dev)
            # liveness of every (list entry, strip) pair, noted by the forward for the backward
            live = torch.empty(max(int(flatten_ids.numel()), 1), dtype=torch.int32, device=dev)
            # the backward's record-gradient array (atomics accumulate into it): zero-filled in passing by
            # the forward launch instead of a fill launch at the head of the backward
            if ctx.rctx.fill_in_forward and expect_backward:
                v_splats = torch.empty(splats.shape[0], SPLAT_FLOATS, dtype=torch.float32, device=dev)
            # job lists: the ones fg_stbin_fill_jobs built for exactly this call, or a launch of our own
            if prebuilt is not None and prebuilt[1] == (seg_ckpt is not None):
                jobs = prebuilt[0]
            else:
                if prebuilt is not None:  # (planned with list shares, run without: the context's own policy again)
                    cfgp = ctx.rctx.cfg()
                    words = int(_lib.load().fg_raster_jobs_words(int(width), int(height), int(tile_size), cfgp))
                jobs = torch.empty(2, words, dtype=torch.int32, device=dev)
                _call("fg_raster_build_jobs", width, height, tile_size, _ptr(tile_offsets), _ptr(jobs[0]), _ptr(jobs[1]),
                      int(seg_ckpt is not None), cfgp, _stream())  # fmt: skip
            _call("fg_raster_jobs_fwd", channels, width, height, tile_size, _ptr(splats), _ptr(tile_offsets),
                  _ptr(flatten_ids), _ptr(jobs[0]), _ptr(background), int(n_clamp), _ptr(render), _ptr(alphas),
                  _ptr(last_ids), _ptr(clamp_mask), _ptr(seg_ckpt), _ptr(live), _ptr(v_splats),
                  0 if v_splats is None else v_splats.numel(), None, cfgp, _stream(),
                  stage="fg_raster_composite_fwd" if composite else "fg_raster_fwd")  # fmt: skip
        elif composite:  # O1 folded into the kernel epilogue: render is the finished image
            _call("fg_raster_composite_fwd", channels, width, height, tile_size, _ptr(splats), _ptr(tile_offsets),
                  _ptr(flatten_ids), _ptr(background), int(n_clamp), _ptr(render), _ptr(alphas), _ptr(last_ids),
                  _ptr(clamp_mask), cfgp, _stream())  # fmt: skip
        else:
            _call("fg_raster_fwd", channels, width, height, tile_size, _ptr(splats), _ptr(tile_offsets),
                  _ptr(flatten_ids), _ptr(render), _ptr(alphas), _ptr(last_ids), cfgp, _stream())  # fmt: skip
        ctx.save_for_backward(splats, tile_offsets, flatten_ids, alphas, last_ids, background, clamp_mask, seg_ckpt,
                              render if seg_ckpt is not None else None, live)  # fmt: skip
        ctx.jobs_bwd = jobs[1] if jobs is not None else None
        ctx.cfgp = cfgp
        ctx.v_splats_zeroed = v_splats  # consumed by the first backward; a second one fills its own
        ctx.set_materialize_grads(False)  # an unused alpha / render must not cost a zero-fill launch
        ctx.composite = (composite, int(n_clamp))
        ctx.geom = (channels, width, height, tile_size, absgrad, tuple(means2d.shape))
        ctx.means2d_ref = means2d if absgrad else None
        ctx.mark_non_differentiable(last_ids)
        return render, alphas, last_ids

    @staticmethod
    def backward(ctx, *grads):
        with use(ctx.rctx):
            return _RasterSplats._backward(ctx, *grads)

    @staticmethod
    def _backward(ctx, v_render, v_alphas, _v_last):
        (splats, tile_offsets, flatten_ids, alphas, last_ids, background, clamp_mask, seg_ckpt,
         image, live) = ctx.saved_tensors  # fmt: skip
        C, width, height, tile_size, absgrad, m2_shape = ctx.geom
        composite, n_clamp = ctx.composite
        N = splats.shape[0]
        v_render = torch.zeros(height, width, C, device=splats.device) if v_render is None else v_render
        v_alphas = None if v_alphas is None else v_alphas.contiguous()  # NULL = no gradient on alpha
        v_splats, ctx.v_splats_zeroed = ctx.v_splats_zeroed, None
        if v_splats is None:
            v_splats = torch.zeros(N, SPLAT_FLOATS, dtype=torch.float32, device=splats.device)
        if ctx.jobs_bwd is not None:
            _call("fg_raster_jobs_bwd", C, width, height, tile_size, _ptr(splats), _ptr(tile_offsets),
                  _ptr(flatten_ids), _ptr(ctx.jobs_bwd), _ptr(background), n_clamp, _ptr(clamp_mask), _ptr(alphas),
                  _ptr(last_ids), _ptr(v_render.contiguous()), _ptr(v_alphas), _ptr(v_splats), _ptr(seg_ckpt),
                  _ptr(image), _ptr(live), ctx.cfgp, _stream(),
                  stage="fg_raster_composite_bwd" if composite else "fg_raster_bwd")  # fmt: skip
            ctx.jobs_bwd = None
        elif composite:
            _call("fg_raster_composite_bwd", C, width, height, tile_size, _ptr(splats), _ptr(tile_offsets),
                  _ptr(flatten_ids), _ptr(background), n_clamp, _ptr(clamp_mask), _ptr(alphas), _ptr(last_ids),
                  _ptr(v_render.contiguous()), _ptr(v_alphas), _ptr(v_splats), ctx.cfgp, _stream())  # fmt: skip
        else:
            _call("fg_raster_bwd", C, width, height, tile_size, _ptr(splats), _ptr(tile_offsets), _ptr(flatten_ids),
                  _ptr(alphas), _ptr(last_ids), _ptr(v_render.contiguous()), _ptr(v_alphas),
                  _ptr(v_splats), ctx.cfgp, _stream())  # fmt: skip
        if ctx.rctx.color_grad_sink is not None and C >= 3:
            ctx.rctx.color_grad_sink("records", splats, v_splats)  # (view-DP: the colour gradient can leave now)
        # strided views of the record array: no 64 MB re-read just to compact 8 bytes per row
        v_means2d = v_splats[:, 0:2].view(m2_shape)
        if absgrad and ctx.means2d_ref is not None:
            ctx.means2d_ref.absgrad = v_splats[:, 6:8].view(m2_shape)
            ctx.means2d_ref = None
        return v_splats, v_means2d, None, None, None, None, None, None, None, None, None, None


def rasterize_splats(splats, means2d, channels, width, height, tile_size, tile_offsets, flatten_ids, absgrad=False,
                     background=None, n_clamp=0):  # fmt: skip
    """-> render[H,W,C], alphas[H,W,1], last_ids[H,W] from packed records (fused path).
    ``background`` [C] (no gradient) and ``n_clamp`` fold the model's post-composite into the
    kernels: render = clamp(render + (1 - alpha) * background) on the first n_clamp channels."""
    if background is not None:
        background = background.detach().to(device=splats.device, dtype=torch.float32).contiguous()
        if background.numel() != channels:
            raise ValueError(f"background must have {channels} values")
    if not 0 <= n_clamp <= channels:
        raise ValueError("n_clamp out of range")
    if tile_size != TILE_SIZE:
        raise ValueError("tile_size must be 16")
    if not 1 <= channels <= MAX_CHANNELS:
        raise ValueError(f"1..{MAX_CHANNELS} composited channels supported, got {channels}")
    # (grad mode is off inside Function.forward: whether a backward can follow is decided here)
    expect_backward = torch.is_grad_enabled() and (splats.requires_grad or means2d.requires_grad)
    return _RasterSplats.apply(splats, means2d, int(channels), int(width), int(height), int(tile_size), tile_offsets,
                               flatten_ids, bool(absgrad), background, int(n_clamp), expect_backward)  # fmt: skip


# --------------------------------------------------------------------------------------------
# The whole step of one view as ONE C-ABI call per direction (fg_step_fwd / fg_step_bwd)

_STEP_PLANS: dict = {}


def _step_plan(key, cfgp):
    """(desc, layout, rc) of a step shape, cached: one layout query per shape / capacity / launch policy (``key`` holds the
    policy by value; ``cfgp``: the address of an fg_raster_config with exactly those values, for the query)."""
    plan = _STEP_PLANS.get(key)
    if plan is None:
        (dev, N, W, H, raw, sh_degree, k_stored, n_color, with_depth, n_extra, antialiased, n_clamp, want_backward,
         list_shares, flags, capacity, eps2d, near, far, radius_clip, _variant, _policy_bytes) = key  # fmt: skip
        d = _lib.StepDesc()
        d.size = ctypes.sizeof(_lib.StepDesc)
        d.N, d.width, d.height, d.tile_size, d.raw, d.sh_degree, d.k_stored, d.n_color = N, W, H, TILE_SIZE, raw, sh_degree, k_stored, n_color
        d.with_depth, d.n_extra, d.antialiased, d.n_clamp, d.want_backward, d.list_shares, d.flags = (
            with_depth, n_extra, antialiased, n_clamp, want_backward, list_shares, flags)
        d.eps2d, d.near_plane, d.far_plane, d.radius_clip, d.capacity = eps2d, near, far, radius_clip, capacity
        L = _lib.StepLayout()
        rc = _lib.load().fg_step_layout_query(ctypes.addressof(d), cfgp, ctypes.addressof(L))
        plan = (d, L, rc)
        if len(_STEP_PLANS) >= 512:
            _STEP_PLANS.pop(next(iter(_STEP_PLANS)))
        _STEP_PLANS[key] = plan
    return plan


def step_path_available(rctx, N, width, height, tile_size, dev) -> bool:
    """Can this call go through fg_step_fwd / fg_step_bwd?  The default configuration on a shape whose list capacity is
    known (the first call of a shape measures it through the stage-wise path); every knob that selects another binning
    path, a stage timer over ALL stages, a colour-gradient sink (the factored exchange hooks in between the stages),
    graph capture and FG_STEP_CALLS=0 take the stage-wise calls."""
    if not rctx.step_calls or tile_size != TILE_SIZE or N <= 0:
        return False
    if (rctx.binning != "supertile" or not rctx.tight_rects or not rctx.direct_count or not rctx.jobs_in_fill
            or rctx.overlap_pack or not rctx.speculative_binning or not rctx.fill_in_forward or not rctx.sh_jacobian
            or rctx.static_capacity is not None or rctx.color_grad_sink is not None):
        return False
    st = rctx.stage_timer
    if st is not None and (st.only is None or not set(st.only) <= {"fg_raster_fwd", "fg_raster_bwd"}):
        return False
    tile_w, tile_h = (width + 15) // 16, (height + 15) // 16
    if rctx.capacity_for((dev, tile_w, tile_h, "fg_stbin"), N) is None:
        return False
    # job-list launches and the supertile binning must take this image size (tiny images run the classic launches)
    skey = (width, height, N >> 20, bytes(rctx.policy))
    ok = _STEP_SIZES.get(skey)
    if ok is None:
        lib = _lib.load()
        ok = bool(lib.fg_stbin_supported(N, tile_w, tile_h)) and int(lib.fg_raster_jobs_words(width, height, TILE_SIZE, rctx.cfg())) > 0
        if len(_STEP_SIZES) >= 512:
            _STEP_SIZES.clear()
        _STEP_SIZES[skey] = ok
    return ok


_STEP_SIZES: dict = {}


class _RasterStep(torch.autograd.Function):
    """Everything between the parameters and the image as one node: fg_step_fwd in forward, fg_step_bwd in backward."""

    @staticmethod
    def forward(ctx, means, quats, d_quats, scales, d_scales, opacities, colors, features_rest, extra, viewmat, K,
                background, opts):  # fmt: skip
        rctx = current()
        ctx.rctx = rctx
        (raw, width, height, eps2d, near, far, radius_clip, antialiased, sh_degree, with_depth, n_clamp, absgrad,
         want_backward, batched) = opts  # fmt: skip
        N, dev = means.shape[0], means.device
        lib = _lib.load()
        tile_w, tile_h = (width + 15) // 16, (height + 15) // 16
        if raw:
            k_stored, n_color = 1 + features_rest.shape[1], 3
        elif sh_degree >= 0:
            k_stored, n_color = colors.shape[1], 3
        else:
            k_stored, n_color = 0, (0 if colors is None else colors.shape[1])
        n_extra = 0 if extra is None else extra.shape[1]
        channels = n_color + int(with_depth) + n_extra
        lkey, ckey = (dev, tile_w, tile_h), (dev, tile_w, tile_h, "fg_stbin")
        long_mode = rctx.long_segments == "always" or (rctx.long_segments == "auto" and rctx.long_shapes.get(lkey, 0) > 0)
        heavy = rctx.heavy_tiles == "always" or (rctx.heavy_tiles == "auto" and rctx.heavy_shapes.get(lkey, 0) > 0)
        capacity = rctx.capacity_for(ckey, N)
        masked = rctx.masks_on(lkey)  # (footprint masks where they pay: RasterContext.masks_on)
        while True:
            seg_slots = rctx.seg_slots_for(lkey, capacity, tile_w * tile_h, N) if channels == 3 and want_backward else 0
            cfgp, variant = rctx.cfg_variant(heavy, seg_slots, rctx.even_shape(lkey), rctx.uneven_shape(lkey), rctx.heavy_lens(lkey)[1])
            shares = want_backward and _seg_ckpt_floats(rctx, channels, width, height, TILE_SIZE, capacity, cfgp) > 0
            # (the launch policy enters the key by VALUE -- its bytes and the variant's fields -- never by the address of a
            # context's copy, which another context's copy may reuse)
            key = (dev, N, width, height, int(raw), sh_degree, k_stored, n_color, int(with_depth), n_extra, int(antialiased),
                   n_clamp, int(want_backward), int(shares),
                   ((_lib.STBIN_LONG_SEGMENTS | (_lib.STBIN_TEST_SMALL_SLABS if rctx.test_small_slabs else 0)) if long_mode else 0)
                   | (0 if masked else _lib.STEP_NO_FOOTPRINT_MASKS), capacity, eps2d,
                   near, far, radius_clip, variant, bytes(rctx.policy))  # fmt: skip
            d, L, rc = _step_plan(key, cfgp)
            _lib.check(rc, "fg_step_layout_query")
            # what changed against the shape's previous call (bench.py reports the tallies of its timed region: a step of
            # several milliseconds in a steady state is either one of these -- a new capacity = new buffers, a policy flip = a
            # new plan -- or it is the host's)
            last = rctx._last_plan.get(lkey)
            if last is not None and last != key:
                rctx.plan_changes += 1
                for name, i in (("N", 1), ("shares", 13), ("long_mode", 14), ("capacity", 15), ("variant", 20)):
                    if last[i] != key[i]:
                        rctx.plan_change_reasons[name] = rctx.plan_change_reasons.get(name, 0) + 1
            rctx._last_plan[lkey] = key
            keep = rctx.workspace("keep", (L.keep_bytes + 3) >> 2, torch.float32, dev)
            tmp = rctx.workspace("tmp", max(L.tmp_bytes, 8), torch.uint8, dev)
            count_slot, count_ptr = _count_slot()
            io = _lib.StepIO()
            io.means, io.quats, io.d_quats, io.scales, io.d_scales = _ptr(means), _ptr(quats), _ptr(d_quats), _ptr(scales), _ptr(d_scales)
            io.opacities, io.colors, io.features_rest, io.extra = _ptr(opacities), _ptr(colors), _ptr(features_rest), _ptr(extra)
            io.viewmat, io.K, io.background, io.count_out = _ptr(viewmat), _ptr(K), _ptr(background), count_ptr
            io.ckpt_need_out = count_ptr + 32 if shares else None
            st = rctx.stage_timer
            ev = st.record("fg_raster_fwd") if st is not None else None
            if ev:  # (the library records them around its raster launch)
                io.ev_raster_begin, io.ev_raster_end = StageTimer.handles(ev)
            _lib.check(lib.fg_step_fwd(ctypes.addressof(d), cfgp, ctypes.addressof(io), keep.data_ptr(), tmp.data_ptr(),
                                       ctypes.addressof(L), _stream()), "fg_step_fwd")  # fmt: skip
            rctx.long_calls += int(long_mode)
            rctx.heavy_calls += int(heavy and shares)
            rctx.full_ckpt_allocs += int(shares and rctx.compact_slots and seg_slots == 0 and not rctx.seg_slots_known and channels == 3)
            # the outputs are views of the kept workspace: one as_strided each (built before the wait below, i.e. while the
            # GPU runs the projection and the count pass)
            k32, i32 = keep, keep.view(torch.int32)
            strided, off, nb = torch.as_strided, L.offset, L.nbytes
            SB = _lib.STEP_BUFFER

            def view(base, name, shape, stride):
                i = SB[name]
                return None if nb[i] == 0 else strided(base, shape, stride, off[i] >> 2)

            if batched:  # the leading camera axis of the reference's outputs, without an unsqueeze node each
                render = view(k32, "render", (1, height, width, channels), (height * width * channels, width * channels, channels, 1))
                alphas = view(k32, "alphas", (1, height, width, 1), (height * width, width, 1, 1))
                means2d = view(k32, "means2d", (1, N, 2), (2 * N, 2, 1))
            else:
                render = view(k32, "render", (height, width, channels), (width * channels, channels, 1))
                alphas = view(k32, "alphas", (height, width, 1), (width, 1, 1))
                means2d = view(k32, "means2d", (N, 2), (2, 1))
            depths, conics = view(k32, "depths", (N,), (1,)), view(k32, "conics", (N, 3), (3, 1))
            radii, tiles = view(i32, "radii", (N,), (1,)), view(i32, "tiles", (N,), (1,))
            last_ids = view(i32, "last_ids", (height, width), (width, 1))
            splats = view(k32, "splats", (N, SPLAT_FLOATS), (SPLAT_FLOATS, 1))
            list_offsets = view(i32, "list_offsets", (tile_w * tile_h + 1,), (1,))
            n_isects = _note_counts(rctx, lkey, ckey, count_slot, shares, N, walks=shares, masked=masked)
            if n_isects <= capacity:
                break
            rctx.capacity_redos += 1  # the guess was too small: nothing was drawn; again with the list's own length
            capacity = max(n_isects, 1)
        flatten_ids = view(i32, "flatten_ids", (n_isects,), (1,))
        ctx.save_for_backward(means, quats, d_quats, scales, d_scales, opacities, colors, features_rest, extra, viewmat, K,
                              background, keep)  # fmt: skip
        ctx.plan = (d, L, cfgp, raw, N, sh_degree, absgrad)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(last_ids, radii, tiles, splats, flatten_ids, list_offsets)
        return render, alphas, means2d, depths, conics, last_ids, radii, tiles, splats, flatten_ids, list_offsets

    @staticmethod
    def backward(ctx, *grads):
        with use(ctx.rctx):
            return _RasterStep._backward(ctx, *grads)

    @staticmethod
    def _backward(ctx, v_render, v_alphas, v_means2d, v_depths, v_conics, *_unused):
        (means, quats, d_quats, scales, d_scales, opacities, colors, features_rest, extra, viewmat, K, background,
         keep) = ctx.saved_tensors  # fmt: skip
        d, L, cfgp, raw, N, sh_degree, absgrad = ctx.plan
        rctx = ctx.rctx
        if not d.want_backward:
            raise _lib.FgRasterError("this step was rendered without gradient buffers (no input required a gradient)")

        # (zero-filled by the forward launch -- for the FIRST backward through this node; a later one (retain_graph=True:
        # per-loss gradients) finds the first one's sums there and clears them itself.  info["means2d"].grad of the earlier
        # pass is a view of this very array: it is copied out first and the new result added to it, as a retain_grad()'ed
        # tensor accumulates)
        # (through `.data`: the record gradients are written in place -- by the kernels, by the two lines below -- and share
        # their storage with the node's OUTPUTS (as_strided views of `keep`); in-place torch operations on a tensor that shares
        # their version counter would make autograd refuse the next backward through the graph)
        v_splats = torch.as_strided(keep.data, (N, SPLAT_FLOATS), (SPLAT_FLOATS, 1), L.offset[_lib.STEP_BUFFER["v_splats"]] >> 2)
        ref = getattr(ctx, "means2d_ref", None)
        m2 = ref() if ref is not None else None
        earlier_m2_grad = None
        if getattr(ctx, "v_splats_consumed", False):
            if m2 is not None and m2.grad is not None:
                aliased = m2.grad.untyped_storage().data_ptr() == keep.untyped_storage().data_ptr()
                earlier_m2_grad = m2.grad.clone() if aliased else m2.grad  # (the first pass's view of this array / a sum of passes)
                if aliased and getattr(m2, "absgrad", None) is not None:
                    m2.absgrad = m2.absgrad.clone()
            v_splats.zero_()
        ctx.v_splats_consumed = True
        if v_means2d is not None:  # a loss on info["means2d"] itself: it joins the raster's xy gradient in the records
            v_splats[:, 0:2] += v_means2d.reshape(N, 2)
        if v_render is None:
            H, W = d.height, d.width
            v_render = torch.zeros(H, W, L.channels, dtype=torch.float32, device=keep.device)
        io = _lib.StepIO()
        io.means, io.quats, io.d_quats, io.scales, io.d_scales = _ptr(means), _ptr(quats), _ptr(d_quats), _ptr(scales), _ptr(d_scales)
        io.opacities, io.colors, io.features_rest, io.extra = _ptr(opacities), _ptr(colors), _ptr(features_rest), _ptr(extra)
        io.viewmat, io.K, io.background = _ptr(viewmat), _ptr(K), _ptr(background)
        keepalive = [v_render.contiguous(), None if v_alphas is None else v_alphas.contiguous(),
                     None if v_depths is None else v_depths.contiguous(), None if v_conics is None else v_conics.contiguous()]
        io.v_render, io.v_alphas, io.v_depths, io.v_conics = (_ptr(t) for t in keepalive)
        v_means, v_quats, v_scales, v_opac = _alloc_grad(means), _alloc_grad(quats), _alloc_grad(scales), _alloc_grad(opacities)
        v_dq = torch.empty_like(d_quats) if d_quats is not None else None
        v_ds = torch.empty_like(d_scales) if d_scales is not None else None
        v_colors = _alloc_grad(colors) if colors is not None else None
        v_rest = _alloc_grad(features_rest) if features_rest is not None else None
        v_extra = torch.empty_like(extra) if extra is not None else None
        io.v_means, io.v_quats, io.v_d_quats, io.v_scales, io.v_d_scales = _ptr(v_means), _ptr(v_quats), _ptr(v_dq), _ptr(v_scales), _ptr(v_ds)
        io.v_opacities, io.v_colors, io.v_features_rest, io.v_extra = _ptr(v_opac), _ptr(v_colors), _ptr(v_rest), _ptr(v_extra)
        st = rctx.stage_timer
        ev = st.record("fg_raster_bwd") if st is not None else None
        if ev:
            io.ev_raster_begin, io.ev_raster_end = StageTimer.handles(ev)
        _lib.check(_lib.load().fg_step_bwd(ctypes.addressof(d), cfgp, ctypes.addressof(io), keep.data_ptr(),
                                           ctypes.addressof(L), _stream()), "fg_step_bwd")  # fmt: skip
        v_viewmat = None
        if ctx.needs_input_grad[9]:  # the camera pose requires a gradient: a pass of its own over the same cotangents
            SBF = _lib.STEP_BUFFER

            def kept(name, dtype, shape):  # a buffer of the kept workspace (None: the step has none)
                i = SBF[name]
                if L.nbytes[i] == 0:
                    return None
                base = keep.data if dtype is torch.float32 else keep.data.view(torch.int32)
                return torch.as_strided(base, shape, (shape[1], 1) if len(shape) == 2 else (1,), L.offset[i] >> 2)

            k_stored = (1 + features_rest.shape[1]) if raw else (colors.shape[1] if sh_degree >= 0 else 0)
            n_color = 3 if (raw or sh_degree >= 0) else (0 if colors is None else colors.shape[1])
            v_viewmat = _viewmat_grad(raw, means, quats, d_quats, scales, d_scales, opacities, colors, features_rest, sh_degree,
                                      k_stored, n_color, d.with_depth, d.n_extra, viewmat, K, d.width, d.height, d.eps2d,
                                      d.antialiased, kept("radii", torch.int32, (N,)), v_splats, v_splats, SPLAT_FLOATS,
                                      keepalive[2], keepalive[3], kept("sh_jac", torch.float32, (N, _lib.SH_JAC_FLOATS)))  # fmt: skip
        if m2 is not None:
            # info["means2d"]: .grad as if it had been retain_grad()'ed on the way to the compositing, .absgrad beside it
            # (reference freegaussian_model.py:869-872, :377); strided views of the record gradients, no copy
            m2.grad = v_splats[:, 0:2].view(m2.shape)
            if earlier_m2_grad is not None:
                m2.grad = earlier_m2_grad + m2.grad
            if absgrad:
                m2.absgrad = v_splats[:, 6:8].view(m2.shape)
        return v_means, v_quats, v_dq, v_scales, v_ds, v_opac, v_colors, v_rest, v_extra, v_viewmat, None, None, None


def raster_step(means, quats, scales, opacities, colors, viewmat, K, width, height, *, raw=False, d_quats=None, d_scales=None,
                features_rest=None, extra=None, background=None, n_clamp=0, eps2d=0.3, near_plane=0.01, far_plane=1e10,
                radius_clip=0.0, antialiased=False, sh_degree=-1, with_depth=False, absgrad=False, batched=False):  # fmt: skip
    """One view through fg_step_fwd (and, on backward, fg_step_bwd).  ``raw``: the model's raw parameter forms (scales =
    log-scales, opacities = logits, colors = features_dc, features_rest).  -> (render [H,W,C], alphas [H,W,1], means2d
    [N,2], depths [N], conics [N,3], last_ids, radii, tiles_touched, splats [N,16], flatten_ids [n], list_offsets [T+1],
    node; ``batched``: render [1,H,W,C], alphas [1,H,W,1], means2d [1,N,2]) -- ``node`` takes ``node.means2d_ref = weakref.ref(t)`` for the tensor that is to receive .grad / .absgrad."""
    f = [None if t is None else _f32(t, "input") for t in (means, quats, d_quats, scales, d_scales, opacities, colors,
                                                           features_rest, extra, viewmat, K)]  # fmt: skip
    if f[5].dim() != 1:
        f[5] = f[5].reshape(-1)
    bg = None if background is None else background.detach().to(device=f[0].device, dtype=torch.float32).contiguous()
    want_backward = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in f[:10])  # (f[9]: the pose)
    opts = (bool(raw), int(width), int(height), float(eps2d), float(near_plane), float(far_plane), float(radius_clip),
            bool(antialiased), int(sh_degree), bool(with_depth), int(n_clamp), bool(absgrad), bool(want_backward),
            bool(batched))  # fmt: skip
    out = _RasterStep.apply(*f, bg, opts)
    return out + (out[0].grad_fn,)


# --------------------------------------------------------------------------------------------
# F flow derivative


def camera_flow(depth: torch.Tensor, K: torch.Tensor, veloc: torch.Tensor, omega: torch.Tensor) -> torch.Tensor:
    """Per-pixel camera flow ``A v/Z + B w`` (reference preprocess/epipolar_flow.py:274-317).
    depth[H,W] -> flow[H,W,2]."""
    lib = _lib.load()
    depth, K = _f32(depth, "depth"), _f32(K, "K")
    veloc, omega = _f32(veloc, "veloc"), _f32(omega, "omega")
    H, W = depth.shape
    flow = torch.empty(H, W, 2, dtype=torch.float32, device=depth.device)
    _call("fg_camera_flow", W, H, _ptr(depth), _ptr(K), _ptr(veloc), _ptr(omega), _ptr(flow), _stream())  # fmt: skip
    return flow


def reprojection_flow(depth0: torch.Tensor, depth1: torch.Tensor, K: torch.Tensor, M: torch.Tensor,
                      sign: float = -1.0) -> torch.Tensor:  # fmt: skip
    """Exact-reprojection camera flow (reference preprocess/epipolar_flow_bp.py:268-295): lift with
    depth0 [H,W], map by M [3,4], project with K, divide by depth1 [H,W]; -> sign * (uv - xy) [H,W,2]."""
    depth0, depth1, K, M = _f32(depth0, "depth0"), _f32(depth1, "depth1"), _f32(K, "K"), _f32(M, "M")
    H, W = depth0.shape
    if depth1.shape != depth0.shape or M.numel() != 12:
        raise ValueError("depth maps of equal shape and a 3x4 matrix expected")
    flow = torch.empty(H, W, 2, dtype=torch.float32, device=depth0.device)
    _call("fg_reprojection_flow", W, H, _ptr(depth0), _ptr(depth1), _ptr(K), _ptr(M), float(sign), _ptr(flow), _stream())
    return flow


class _GaussianFlow(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means2d, depths, vel, radii, K, veloc, omega):
        ctx.rctx = current()  # the backward (autograd's thread) reads the same context
        lib = _lib.load()
        N = depths.shape[0]
        u_gs = torch.empty(N, 2, dtype=torch.float32, device=depths.device)
        u_cam = torch.empty_like(u_gs)
        _call("fg_flow_fwd", N, _ptr(means2d), _ptr(depths), _ptr(radii), _ptr(vel), _ptr(K), _ptr(veloc), _ptr(omega),
                            _ptr(u_gs), _ptr(u_cam), _stream())  # fmt: skip
        ctx.save_for_backward(means2d, depths, vel, radii, K, veloc, omega)
        return u_gs, u_cam

    @staticmethod
    def backward(ctx, *grads):
        with use(ctx.rctx):
            return _GaussianFlow._backward(ctx, *grads)

    @staticmethod
    def _backward(ctx, v_gs, v_cam):
        lib = _lib.load()
        means2d, depths, vel, radii, K, veloc, omega = ctx.saved_tensors
        N = depths.shape[0]
        v_gs = torch.zeros(N, 2, device=depths.device) if v_gs is None else v_gs.contiguous()
        v_cam = torch.zeros(N, 2, device=depths.device) if v_cam is None else v_cam.contiguous()
        v_m = torch.empty_like(means2d)
        v_d = torch.empty_like(depths)
        v_v = torch.empty_like(vel)
        _call("fg_flow_bwd", N, _ptr(means2d), _ptr(depths), _ptr(radii), _ptr(vel), _ptr(K), _ptr(veloc), _ptr(omega),
                            _ptr(v_gs), _ptr(v_cam), _ptr(v_m), _ptr(v_d), _ptr(v_v), _stream())  # fmt: skip
        return v_m, v_d, v_v, None, None, None, None


def gaussian_flow(means2d, depths, vel, K, veloc, omega, radii=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Per-Gaussian projection-flow Jacobian terms (Lemma 1, reference docs/index.html:256-273):
    ``u_gs = A(mu) vel / Z`` and ``u_cam = A(mu) v / Z + B(mu) w``; both [N,2], differentiable
    w.r.t. means2d, depths and vel."""
    means2d, depths, vel = _f32(means2d, "means2d"), _f32(depths, "depths"), _f32(vel, "vel")
    K, veloc, omega = _f32(K, "K"), _f32(veloc, "veloc"), _f32(omega, "omega")
    return _GaussianFlow.apply(means2d.reshape(-1, 2), depths, vel, radii, K, veloc, omega)
