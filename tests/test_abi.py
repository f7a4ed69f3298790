"""CPU checks: the C-ABI library loads and exports exactly what include/fgraster.h declares."""
import os
import re
import subprocess

import pytest

from freegaussian_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "fgraster.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fg_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_bound_and_exported():
    syms = _header_symbols()
    assert len(syms) >= 19
    lib = _lib.load()
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = set(re.findall(r" T (fg_\w+)", out))
    for s in syms:
        assert s in exported, f"{s} declared in fgraster.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature"
        assert getattr(lib, s) is not None
    assert set(_lib.SIGNATURES) == set(syms)


_POLICIES = []


def _policy_ptr(**fields):
    """Address of a launch policy that stays alive (RasterConfig.ptr() is the address of the Python object's buffer: a
    temporary would be freed before the call it is an argument of)."""
    from freegaussian_amd import ops

    _POLICIES.append(ops.launch_policy(**fields))
    return _POLICIES[-1].ptr()


def test_abi_version_and_error_strings():
    lib = _lib.load()
    assert lib.fg_abi_version() == _lib.ABI_VERSION
    assert lib.fg_error_string(0) == b"ok"
    assert b"invalid" in lib.fg_error_string(-1)
    with pytest.raises(_lib.FgRasterError):
        _lib.check(-3, "x")


def test_workspace_queries_need_no_gpu():
    lib = _lib.load()
    assert lib.fg_scan_workspace_bytes(1_000_000) >= 8 * (1_000_000 // 2048)
    n = 5_000_000
    assert lib.fg_sort_workspace_bytes(n) >= n * 12
    # job lists: none below 200 tiles (classic launches), 8 counters + 8 segments of 8 entries per
    # tile of the largest XCD band otherwise -- a band balanced by content holds up to 1.5 x the equal share of rows --
    # + a word per tile (the table of first checkpoint slots, fg_raster_config::seg_slots)
    assert lib.fg_raster_jobs_words(208, 144, 16, None) == 0  # 13 x 9 = 117 tiles
    assert lib.fg_raster_jobs_words(480, 270, 16, None) == 8 + 8 * 8 * 5 * 30 + 30 * 17  # bands of up to 3 (+ 2) rows
    assert lib.fg_raster_jobs_words(1920, 1080, 16, None) == 8 + 8 * 8 * 14 * 120 + 120 * 68
    from freegaussian_amd import ops
    assert lib.fg_raster_jobs_words(1920, 1080, 16, _policy_ptr(balance_bands=0)) == 8 + 8 * 8 * 9 * 120 + 120 * 68
    # checkpoint buffer of the list shares: a slot (4352 B) per 64 entries of the list's capacity and per tile, or -- compact
    # slots -- as many as the policy says; in front of them the strip table and the T_final plane
    full = lib.fg_raster_seg_ckpt_floats(3, 1920, 1080, 16, 6_400_000, None)
    policy = ops.launch_policy(seg_slots=40960)  # (kept alive across the call: ptr() is the object's own address)
    compact = lib.fg_raster_seg_ckpt_floats(3, 1920, 1080, 16, 6_400_000, policy.ptr())
    assert full - compact == (6_400_000 // 64 + 8160 + 2 - 40960) * 1088 and compact * 4 < 190e6 < 470e6 < full * 4
    assert lib.fg_raster_jobs_words(1920, 1080, 8, None) == 0  # unsupported tile size
    assert lib.fg_raster_build_jobs(1920, 1080, 16, None, None, None, 0, None, None) == -1


def test_argument_validation_without_gpu():
    """Bad arguments are rejected before any launch, so this is safe on a CPU-only box."""
    lib = _lib.load()
    assert lib.fg_project_fwd(-1, *([None] * 5), 1, 1, 0.3, 0.01, 1e10, 0.0, 16, *([None] * 7)) == -1
    assert lib.fg_raster_fwd(3, 64, 64, 8, *([None] * 8)) in (-1, -4)
    assert lib.fg_sh_fwd(10, 4, 16, *([None] * 6)) == -1
    assert lib.fg_sort_pairs(10, None, None, 70, None, 0, None) == -1


def test_argument_validation_of_the_newer_entry_points_without_gpu():
    lib = _lib.load()
    n = [None]
    # densify / gather / children / back-projection: negative sizes and missing buffers
    assert lib.fg_densify_flags(-1, 1, 100.0, 1e-3, 0.01, 0.05, 0.1, 0.5, 0.15, *(n * 7)) == -1
    assert lib.fg_densify_flags(8, 1, 100.0, 1e-3, 0.01, 0.05, 0.1, 0.5, 0.15, *(n * 7)) == -1
    assert lib.fg_densify_map(8, *(n * 5), 1, 1, 1, 2, None, None, None) == -1
    assert lib.fg_gather_rows(4, 0, None, None, 0, None, None) == -1
    assert lib.fg_gather_rows(0, 3, None, None, 0, None, None) == 0  # nothing to do
    assert lib.fg_split_children(0, 4, *(n * 6)) == -1
    assert lib.fg_mask_backproject(4, None, None, None, None, 8, 8, None, None, 2, 3, None, None) == -1  # M > stored
    # factored exchange: payload layout must be 3 or 6 floats
    assert lib.fg_sh_grad_accumulate(4, 2, 3, 16, None, None, 100, 4, 1.0, None, None) == -1
    assert lib.fg_sh_grad_accumulate(4, 2, 3, 16, None, None, 5, 3, 1.0, None, None) == -1  # stride < 3N+3
    # capacity emission needs a positive capacity and the count array
    assert lib.fg_bin_emit_sort_capacity(4, 0, *(n * 5), 16, 4, 4, *(n * 4), 0, None) == -1
    assert lib.fg_bin_prepare_rects(4, *(n * 3), 16, 4, 4, *(n * 4), 0, None) == -1
    # raw preprocess is SH-only
    assert lib.fg_preprocess_raw_fwd(4, *(n * 8), -1, 16, 0, None, 0, None, None, 32, 32, 0.3, 0.01, 1e10, 0.0, 16, 0,
                                     *(n * 12)) == -1  # fmt: skip
    assert lib.fg_bin_prepare_keys(4, *(n * 7), 0, None) == -1  # keys / rectangles / outputs missing
    assert lib.fg_bin_prepare_keys(0, *(n * 7), 0, None) == 0
    # composite raster: clamp count within the channels, mask required
    assert lib.fg_raster_composite_fwd(3, 32, 32, 16, *(n * 4), 4, *(n * 6)) == -1
    # fused loss: images no larger than the 11 x 11 window have no SSIM map; buffers required; workspace size is a query
    assert lib.fg_l1_ssim_workspace_floats(10, 64, 3) == 0 and lib.fg_l1_ssim_workspace_floats(64, 10, 3) == 0
    assert lib.fg_l1_ssim_workspace_floats(1080, 1920, 3) == 2 * 60 * 68 * 3  # a float2 per 32 x 16 tile and channel
    assert lib.fg_l1_ssim_fwd(10, 64, 3, *(n * 4), 0, None, None) == -1
    assert lib.fg_l1_ssim_fwd(64, 64, 3, *(n * 4), 0, None, None) == -1
    assert lib.fg_l1_ssim_bwd(64, 64, 3, *(n * 6)) == -1
    # Adam step: the update count starts at 1, betas in [0, 1), arrays required; nothing to do for n = 0
    assert lib.fg_adam_step(16, *(n * 4), 1e-3, 0.9, 0.999, 1e-15, 0, None) == -1
    assert lib.fg_adam_step(16, *(n * 4), 1e-3, 1.0, 0.999, 1e-15, 1, None) == -1
    assert lib.fg_adam_step(16, *(n * 4), 1e-3, 0.9, 0.999, 1e-15, 1, None) == -1
    assert lib.fg_adam_step(0, *(n * 4), 1e-3, 0.9, 0.999, 1e-15, 1, None) == 0
    # fill with job lists: the image must give the tile grid
    assert lib.fg_stbin_fill_jobs(4, *(n * 3), 4, 4, 100, *(n * 5), 0, 100, 64, 16, None, None, 0, None, 0, None, None) == -1
    # ... and (ABI 7) the flags word holds FG_STBIN_LONG_SEGMENTS or nothing; the workspace has room for the long
    # segments' bucket tables (40 bytes per bucket, a bucket per 1536 list entries + one per possible long segment)
    assert lib.fg_stbin_fill(4, *(n * 3), 4, 4, 100, *(n * 5), 0, 2, None) == -1
    # (ABI 8) footprint masks are relative to the footprint rectangles: no masks without rectangles
    assert lib.fg_preprocess_fwd(4, *(n * 5), -1, 0, 3, 0, None, 0, None, None, 32, 32, 0.3, 0.01, 1e10, 0.0, 16, 0,
                                 *(n * 8), None, 1, None, None) == -1  # fmt: skip
    assert lib.fg_stbin_count(4, None, None, 4, 4, None, None, None, 0, None) == -1
    assert _lib.STEP_NO_FOOTPRINT_MASKS == 2
    assert lib.fg_stbin_fill_workspace_bytes(1 << 23) >= 2 * 8 * (1 << 23) + 40 * ((1 << 23) // 1536 + (1 << 23) // 7936)
    assert _lib.STBIN_LONG_SEGMENTS == 1


def test_step_api_layout_and_argument_validation_without_gpu():
    """fg_step_* (ABI 7): one workspace query per step shape, every buffer 256-byte aligned inside the two
    caller-allocated workspaces; descriptors and pointers are checked before any launch."""
    import ctypes

    from freegaussian_amd import ops

    lib = _lib.load()
    d = _lib.StepDesc()
    d.size = ctypes.sizeof(_lib.StepDesc)
    d.N, d.width, d.height, d.tile_size, d.raw, d.sh_degree, d.k_stored, d.n_color = 1_000_000, 1920, 1080, 16, 1, 3, 16, 3
    d.n_clamp, d.want_backward, d.list_shares, d.capacity = 3, 1, 1, 6_400_000
    d.eps2d, d.near_plane, d.far_plane = 0.3, 0.01, 1e10
    L = _lib.StepLayout()
    assert lib.fg_step_layout_query(ctypes.addressof(d), None, ctypes.addressof(L)) == 0
    B = _lib.STEP_BUFFER
    assert L.channels == 3 and L.jobs_words == lib.fg_raster_jobs_words(1920, 1080, 16, None)
    assert L.seg_ckpt_floats == lib.fg_raster_seg_ckpt_floats(3, 1920, 1080, 16, 6_400_000, None)
    assert L.nbytes[B["render"]] == 1920 * 1080 * 3 * 4 and L.nbytes[B["flatten_ids"]] == 6_400_000 * 4
    assert L.nbytes[B["splats"]] == L.nbytes[B["v_splats"]] == 1_000_000 * 64 and L.nbytes[B["comp"]] == 0
    kept = [i for n, i in B.items() if n not in ("count_ws", "fill_ws")]
    assert all(L.offset[i] % 256 == 0 for i in range(len(B)))
    ends = sorted((L.offset[i], L.offset[i] + L.nbytes[i]) for i in kept if L.nbytes[i])
    assert all(a[1] <= b[0] for a, b in zip(ends, ends[1:])) and ends[-1][1] <= L.keep_bytes  # disjoint, inside
    assert L.nbytes[B["count_ws"]] == lib.fg_stbin_count_workspace_bytes(1_000_000, 120, 68)
    assert L.nbytes[B["fill_ws"]] == lib.fg_stbin_fill_workspace_bytes(6_400_000) and L.tmp_bytes >= L.nbytes[B["count_ws"]] + L.nbytes[B["fill_ws"]]
    # compact checkpoint slots: the same query, the policy's slots
    Lc = _lib.StepLayout()
    policy = ops.launch_policy(seg_slots=40960)
    assert lib.fg_step_layout_query(ctypes.addressof(d), policy.ptr(), ctypes.addressof(Lc)) == 0
    assert Lc.keep_bytes < L.keep_bytes - 250e6
    # no gradient buffers without a backward
    d.want_backward = 0
    assert lib.fg_step_layout_query(ctypes.addressof(d), None, ctypes.addressof(Lc)) == 0
    assert Lc.nbytes[B["v_splats"]] == Lc.nbytes[B["seg_ckpt"]] == Lc.nbytes[B["live"]] == 0
    d.want_backward = 1
    # rejected before any launch: a short descriptor, a raw step without SH, more clamped channels than channels, a tiny
    # image (classic launches: the stage-wise entry points), missing pointers
    io = _lib.StepIO()
    for field, value, rc in (("size", 8, -1), ("sh_degree", -1, -1), ("n_clamp", 4, -1), ("capacity", 0, -1), ("tile_size", 8, -1)):
        saved = getattr(d, field)
        setattr(d, field, value)
        assert lib.fg_step_layout_query(ctypes.addressof(d), None, ctypes.addressof(Lc)) == rc, field
        setattr(d, field, saved)
    d.width, d.height = 160, 96
    assert lib.fg_step_layout_query(ctypes.addressof(d), None, ctypes.addressof(Lc)) == -4
    d.width, d.height = 1920, 1080
    assert lib.fg_step_layout_query(None, None, ctypes.addressof(L)) == -1
    assert lib.fg_step_layout_query(ctypes.addressof(d), None, None) == -1
    assert lib.fg_step_fwd(ctypes.addressof(d), None, ctypes.addressof(io), None, None, ctypes.addressof(L), None) == -1
    assert lib.fg_step_bwd(ctypes.addressof(d), None, None, None, ctypes.addressof(L), None) == -1


def test_launch_policy_comes_through_the_abi_not_the_environment(monkeypatch):
    """ABI version 3 onwards: the library reads no environment variable; the launch policy is an fg_raster_config the
    host fills (ops.launch_policy_from_env maps the FG_RASTER_* variables of rounds 1-2 onto it)."""
    import ctypes

    from freegaussian_amd import ops

    for dirpath, _, files in os.walk(os.path.join(ROOT, "freegaussian_amd", "csrc")):
        for f in files:
            if f.endswith((".hip", ".h")):
                assert "getenv" not in open(os.path.join(dirpath, f)).read(), f
    lib = _lib.load()
    dflt = _lib.RasterConfig.defaults()
    assert dflt.size == ctypes.sizeof(_lib.RasterConfig) and dflt.seg_parts == -1 and dflt.use_liveness == 1
    # the fields round 4 appended: their defaults leave every new path to the library's / the host's own choice
    assert (dflt.balance_bands, dflt.heavy_tiles, dflt.seg_slots, dflt.prio_fwd, dflt.prio_bwd) == (-1, 0, 0, -1, -1)
    pr = ops.launch_policy_from_env({"FG_RASTER_PRIO_FWD": "250,350", "FG_RASTER_PRIO_BWD": "0", "FG_RASTER_BALANCE": "2"})
    assert (pr.prio_fwd, pr.prio_bwd, pr.balance_bands) == (250 | 350 << 16, 0, 2)
    # equal shares without the cost pass: lists sized for equal bands (no band beyond its share), like balance_bands = 0
    assert lib.fg_raster_jobs_words(1920, 1080, 16, _policy_ptr(balance_bands=2)) == lib.fg_raster_jobs_words(1920, 1080, 16, _policy_ptr(balance_bands=0))
    # the environment no longer reaches the library ...
    monkeypatch.setenv("FG_RASTER_PPT_FWD", "1")
    monkeypatch.setenv("FG_RASTER_PPT_BWD", "1")
    assert lib.fg_raster_jobs_words(1920, 1080, 16, None) > 0
    # ... the struct does: forced pixels per lane mean classic launches, i.e. no job lists
    forced = ops.launch_policy_from_env()
    assert forced.ppt_fwd == 1 and forced.ppt_bwd == 1
    assert lib.fg_raster_jobs_words(1920, 1080, 16, forced.ptr()) == 0
    assert lib.fg_raster_jobs_words(1920, 1080, 16, _policy_ptr(bands_nx=8)) == 8 + 8 * 8 * 15 * 68 + 120 * 68
    assert lib.fg_raster_seg_ckpt_floats(3, 1920, 1080, 16, 10**6, _policy_ptr(seg_parts=1)) == 0
    assert lib.fg_raster_seg_ckpt_floats(3, 1920, 1080, 16, 10**6, None) > 0
    env = {"FG_RASTER_TAIL_BWD": "7,9", "FG_RASTER_SPLIT_BWD": "2", "FG_RASTER_LIVE": "0", "FG_TILE_ORDER": "rows",
           "FG_RASTER_SEG_GRADE": "8,100", "FG_DEBUG_ONLY_XCD": "3"}  # fmt: skip
    p = ops.launch_policy_from_env(env)
    assert (p.tail4_bwd, p.tail2_bwd, p.split4_bwd, p.split2_bwd, p.use_liveness, p.tile_order, p.seg_parts2, p.seg_tail2,
            p.debug_only_xcd) == (7, 9, 2, 0, 0, 0, 8, 100, 3)  # fmt: skip
    with pytest.raises(ValueError):
        ops.launch_policy(no_such_field=1)
    # two contexts, two policies, one process: nothing is module state
    a, b = ops.RasterContext(policy=ops.launch_policy(ppt_fwd=2)), ops.RasterContext(env={})
    with ops.use(a):
        assert ops.current() is a
        with ops.use(b):
            assert ops.current() is b and ops.current().policy.ppt_fwd == 0
        assert ops.current().policy.ppt_fwd == 2
    assert ops.current() is ops.default_context


def test_product_does_not_import_oracle():
    """The product path must never route through the CPU oracle."""
    pkg = os.path.join(ROOT, "freegaussian_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f
                assert "raster_oracle" not in src, f


def test_ops_refuse_cpu_tensors():
    import torch

    from freegaussian_amd import rasterization

    N = 4
    with pytest.raises(_lib.FgRasterError):
        rasterization(torch.zeros(N, 3), torch.ones(N, 4), torch.ones(N, 3), torch.ones(N), torch.ones(N, 3),
                      torch.eye(4)[None], torch.eye(3)[None], 32, 32, packed=False)  # fmt: skip


def test_learned_launch_state_scales_with_the_gaussian_count_host_logic():
    """ops.RasterContext keeps what a shape's calls taught it per (device, tile grid); quantities that grow with N are kept
    with the N they were seen at and scaled by the ratio (the reference changes N every `refine_every` steps,
    freegaussian_model.py:404-571).  Pure host logic: no GPU call."""
    from freegaussian_amd import ops

    ctx = ops.RasterContext(env={})
    key = ("dev", 120, 68, "fg_stbin")
    assert ctx.capacity_for(key, 1_000_000) is None  # nothing known: the first call of a shape measures
    for n_isects in (4_400_000, 4_500_000, 4_300_000):
        ops._note_list_length(ctx, key, n_isects, 1_000_000)
    cap = ctx.capacity_for(key, 1_000_000)
    assert cap == ctx.isect_capacity[key] >= int(4_500_000 * 1.25)
    up, down, far_down = ctx.capacity_for(key, 1_030_000), ctx.capacity_for(key, 930_000), ctx.capacity_for(key, 600_000)
    assert up >= int(4_500_000 * 1.03 * 1.25) and down >= int(4_500_000 * 0.93 * 1.25)
    assert down == cap  # still enough and not 25 % too large: the buffers keep their sizes (hysteresis)
    assert int(4_500_000 * 0.6 * 1.25) <= far_down < cap
    assert ctx.capacity_for(key, 2_500_000) is None and ctx.capacity_for(key, 400_000) is None  # another scene altogether
    ctx.isect_capacity[key] = 1234  # (tests force the overflow path this way: the stored figure stands for the same N)
    assert ctx.capacity_for(key, 1_000_000) == 1234
    # a refinement later: the history is re-read at the new count
    ops._note_list_length(ctx, key, 4_200_000, 950_000)
    assert ctx.isect_n[key] == 950_000 and ctx.capacity_for(key, 950_000) >= int(4_500_000 * 0.95 * 1.25)
    # a jump beyond a factor of two forgets the history
    ops._note_list_length(ctx, key, 900_000, 200_000)
    assert ctx.isect_recent[key] == [(900_000, 200_000)]
    # checkpoint-slot needs: (need, N) pairs, scaled the same way
    lkey = ("dev", 120, 68)
    assert ctx.seg_slots_for(lkey, 6_000_000, 8160, 1_000_000) == 0 and not ctx.seg_slots_known
    ctx.ckpt_need[lkey] = [(4000, 1_000_000), (4100, 1_000_000)]
    same = ctx.seg_slots_for(lkey, 6_000_000, 8160, 1_000_000)
    more = ctx.seg_slots_for(lkey, 6_000_000, 8160, 1_100_000)
    assert ctx.seg_slots_known and same >= 8 * 4100 and same % 4096 == 0 and more > same
    assert ctx.seg_slots_for(lkey, 100_000, 8160, 1_000_000) == 0  # not smaller than a slot per 64 entries: by formula
    # the policy variant enters a cache key by VALUE (an address may be reused by another context's copy)
    p0, v0 = ctx.cfg_variant(False, 0, False)
    p1, v1 = ctx.cfg_variant(True, 8192, True)
    assert v0 == (0, 0, False, None) and v1 == (ctx.heavy_tile_len, 8192, True, None) and p1 != p0
    assert ctx.cfg_variant(True, 8192, True)[0] == p1 and ctx.cfg(True, 8192, True) == p1
    # a shape that showed an uneven scene lately: the forward's finer content thresholds (unless it is an even one by now)
    p2, v2 = ctx.cfg_variant(False, 0, False, True)
    assert v2 == (0, 0, False, ctx.uneven_split_fwd) and p2 not in (p0, p1) and ctx.cfg_variant(False, 0, True, True)[1][3] is None
    assert not ctx.uneven_shape(lkey)
    ctx.uneven_left[lkey] = 8
    assert ctx.uneven_shape(lkey)
    ctx.uneven_left[lkey] = 0
    assert not ctx.uneven_shape(lkey)


def test_heavy_tiles_follow_the_walks_the_forward_reports_host_logic(monkeypatch):
    """ops: a shape runs with heavy tiles when its longest tile list exceeds `heavy_flag_len` AND -- where the raster forward
    reports them (fg_raster_jobs_fwd walk_out, word 13 of the call's ring slot, read one call late) -- a strip walked more
    than 2560 entries in one of its last `heavy_cooldown` reporting calls; calls that cannot report (the stage-wise path)
    leave the verdict alone; nothing known = the list length decides.  Pure host logic: the ring is a numpy array here."""
    import numpy as np

    from freegaussian_amd import ops

    ring = np.zeros(ops._RING_WORDS * ops._COUNT_RING, dtype=np.int64)
    monkeypatch.setattr(ops, "_count_ring_np", ring)
    ctx = ops.RasterContext(env={})
    lkey, key = ("dev", 120, 68), ("dev", 120, 68, "fg_stbin")

    def call(slot, longest, walks=True):
        base = ops._RING_WORDS * slot
        ring[base : base + 4] = (4_000_000, 1000, longest, 0)  # list length, longest segment, longest tile list, segments > 3072
        ring[base + 4 : base + 13] = 0  # (checkpoint needs, the cost pass's decision)
        ops._count_ring_gen[slot] += 1
        ops._note_counts(ctx, lkey, key, slot, need_reported=True, N=1_000_000, walks=walks)
        return ctx.heavy_shapes.get(lkey, 0) > 0

    assert call(0, 12_000)  # nothing reported yet: a list of 12 000 entries turns the policy on
    assert not call(1, 12_000)  # call 0 could report and did not: a dense cluster that closes early
    assert not call(2, 12_000)
    ring[ops._RING_WORDS * 2 + 13] = 3100  # call 2's forward: a strip walked 3100 entries
    assert call(3, 12_000)
    assert call(4, 12_000, walks=False) and call(5, 12_000)  # (a stage-wise call in between: call 5 reads nothing from it)
    for i in range(ctx.heavy_cooldown + 2):  # no report for `heavy_cooldown` reporting calls: off again
        on = call(6 + i, 12_000)
    assert not on
    assert not call(100, 2000)  # below the flag length nothing turns it on
    ring[ops._RING_WORDS * 100 + 13] = 5000
    assert not call(101, 2000) and call(102, 12_000)


def test_workspace_pool_hands_out_only_buffers_nobody_refers_to_host_logic():
    """ops.RasterContext.workspace: the one-call path's `keep` / `tmp` buffers from a pool of the context's own.  A buffer is
    handed out again only when every view of its storage is gone (outputs, info tensors and saved tensors are views of
    `keep`), for requests it fits within 30 % (+ 32 MB); at most four per kind and device.  CPU tensors: same logic."""
    import torch

    from freegaussian_amd import ops

    if not hasattr(torch._C, "_storage_Use_Count"):
        pytest.skip("this torch has no torch._C._storage_Use_Count: the pool is bypassed")
    ctx, cpu = ops.RasterContext(env={}), torch.device("cpu")
    a = ctx.workspace("keep", 100_000, torch.float32, cpu)
    assert a.dtype == torch.float32 and a.numel() == 100_000
    first = a.data_ptr()
    view = a[100:200].view(torch.int32)  # (what an output or a saved tensor is)
    b = ctx.workspace("keep", 100_000, torch.float32, cpu)
    assert b.data_ptr() != first  # the first is still referred to
    del a
    c = ctx.workspace("keep", 100_000, torch.float32, cpu)
    assert c.data_ptr() not in (first, b.data_ptr())  # ... by `view` alone, still
    del view
    d = ctx.workspace("keep", 99_000, torch.float32, cpu)  # a slightly smaller request: the refinement's new N
    assert d.data_ptr() == first
    e = ctx.workspace("tmp", 400_000, torch.uint8, cpu)
    assert e.dtype == torch.uint8 and e.data_ptr() not in (first, b.data_ptr(), c.data_ptr())  # pools per kind
    del d
    big = ctx.workspace("keep", 100_000_000, torch.float32, cpu)  # far beyond any pooled buffer: a new one
    assert big.numel() == 100_000_000 and len(ctx._workspaces[("keep", "cpu", 0)]) == 4
    more = ctx.workspace("keep", 300_000_000, torch.float32, cpu)
    assert len(ctx._workspaces[("keep", "cpu", 0)]) == 4 and more.numel() == 300_000_000  # the oldest unused one made room
    off = ops.RasterContext(env={"FG_WORKSPACE_POOL": "0"})
    assert off.workspace("keep", 10, torch.float32, cpu).numel() == 10 and not off._workspaces
    # release_workspaces: the pool lets go of everything it holds (what torch.cuda.empty_cache() cannot reach)
    del big, more, b, c, e
    freed = ctx.release_workspaces()
    assert freed >= 4 * 300_000_000 and not ctx._workspaces


def test_workspace_pool_is_keyed_by_stream_and_counts_its_fallbacks(monkeypatch):
    """A buffer released by one stream's call is never handed to a call on another stream (the caching allocator's rule for
    the blocks the pool replaces); without torch's private use-count hook the pool is OFF, says so once, and counts."""
    import warnings

    import torch

    from freegaussian_amd import ops

    cpu = torch.device("cpu")
    if ops._STORAGE_USE_COUNT is not None:
        ctx = ops.RasterContext(env={})
        streams = iter([11, 11, 22, 11])
        monkeypatch.setattr(ops, "_stream", lambda: next(streams))
        cuda_like = type("D", (), {"type": "cuda", "__str__": lambda self: "cuda:0"})()
        monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: False)
        made = []
        monkeypatch.setattr(torch, "empty", (lambda real: (lambda *a, **k: made.append(a) or real(*a, **{**k, "device": cpu})))(torch.empty))
        a = ctx.workspace("keep", 1000, torch.float32, cuda_like)
        p = a.data_ptr()
        del a
        b = ctx.workspace("keep", 1000, torch.float32, cuda_like)  # same stream: the released buffer again
        assert b.data_ptr() == p
        del b
        c = ctx.workspace("keep", 1000, torch.float32, cuda_like)  # ANOTHER stream: never the first stream's buffer
        assert c.data_ptr() != p and ("keep", "cuda:0", 22) in ctx._workspaces
        del c
        d = ctx.workspace("keep", 1000, torch.float32, cuda_like)  # back on the first stream: its own buffer
        assert d.data_ptr() == p and ctx.pool_fallback_calls == 0
        monkeypatch.undo()
    monkeypatch.setattr(ops, "_STORAGE_USE_COUNT", None)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        ctx = ops.RasterContext(env={})
    assert any("_storage_Use_Count" in str(x.message) for x in w)
    assert ctx.workspace("keep", 10, torch.float32, cpu).numel() == 10 and ctx.pool_fallback_calls == 1 and not ctx._workspaces
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        ops.RasterContext(env={"FG_WORKSPACE_POOL": "0"})  # pool off by choice: nothing to warn about
    assert not w
