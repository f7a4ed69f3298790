"""GPU parity: every HIP stage, called through the C ABI, against the CPU oracle on the same
seeded inputs.  Integer results (radii, tile counts, sort keys, list order, tile ranges) must be
bit-exact; floating results within REL_TOL = 1e-4 (scale-relative), the bar BASELINE.json's
north_star states.  PARITY UNPINNED w.r.t. gsplat itself: see oracle/raster_oracle.py header."""
import math

import pytest
import torch

from freegaussian_amd import _lib, ops, rasterization
from freegaussian_amd.scenes import plumbing_scene, synthetic_scene
from helpers import REL_TOL, close_except_knife_edge, last_ids_agree, psnr, rel_err, rel_l2
from oracle import raster_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _smooth_loss(rgb, gt):
    """Mean squared error instead of the reference's L1 (freegaussian_model.py:951): L1's sign() is
    discontinuous, so a pixel within rounding of its target hands the two implementations OPPOSITE upstream
    gradients for that pixel (measured at 1M Gaussians / 1080p: a dozen such pixels move the parameter
    gradients by 3e-4) -- a property of the loss, not of the kernels under test.  The harness trains with L1."""
    return ((rgb - gt) ** 2).mean()



@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests selected but no GPU is visible (no CPU fallback exists)")
    _lib.load()


def _setenv_policy(monkeypatch, name, value):
    """The FG_RASTER_* knobs are read by the Python host when a RasterContext is created (the library reads no
    environment); tests switch them inside one process: the default context's launch policy is rebuilt from
    the patched environment (both are restored when the test ends)."""
    monkeypatch.setenv(name, value)
    monkeypatch.setattr(ops.default_context, "policy", ops.launch_policy_from_env())


def _scene(n=6000, w=200, h=120, seed=3, **kw):
    return synthetic_scene(n, w, h, n_views=2, seed=seed, **kw)


# ------------------------------------------------------------------------------------------
def test_wave_reduce16_transposed():
    lib = _lib.load()
    g = torch.Generator().manual_seed(0)
    x = torch.randint(-8, 9, (64, 16), generator=g).float()  # integers: exact sums
    out = torch.zeros(220, device=DEV)
    _lib.check(lib.fg_debug_wave_reduce16(x.to(DEV).data_ptr(), out.data_ptr(), None), "dbg")
    torch.cuda.synchronize()
    out = out.cpu()
    assert torch.equal(out[:16], x.sum(0))
    assert torch.equal(out[16:80], x.sum(0).repeat_interleave(4))
    assert torch.equal(out[80:144], x[:, 0].sum().expand(64))
    assert torch.equal(out[144:156], x.sum(0)[:12])  # 12-value butterfly
    assert torch.equal(out[156:200], x.sum(0)[:11].repeat_interleave(4))  # LDS-transposing reduction, 11 rows
    assert bool((out[200:220] == -1).all())


@pytest.mark.parametrize("n,end_bit", [(1, 64), (63, 64), (4096, 40), (4097, 45), (100_003, 45), (1_000_000, 45)])
def test_sort_pairs_bit_exact_and_stable(n, end_bit):
    g = torch.Generator().manual_seed(n)
    # few distinct high bits + many duplicate keys exercise stability
    keys = torch.randint(0, 2**13, (n,), generator=g) << 32 | torch.randint(0, 2**10, (n,), generator=g) << 20
    keys &= (1 << end_bit) - 1
    vals = torch.arange(n, dtype=torch.int32)
    ref_k, order = torch.sort(keys, stable=True)
    k, v = keys.to(DEV), vals.to(DEV)
    ops.sort_pairs(k, v, end_bit)
    assert torch.equal(k.cpu(), ref_k)
    assert torch.equal(v.cpu(), vals[order])


def test_sort_full_64bit_random():
    g = torch.Generator().manual_seed(7)
    n = 300_000
    keys = torch.randint(-(2**62), 2**62, (n,), generator=g).abs()
    vals = torch.arange(n, dtype=torch.int32)
    ref_k, order = torch.sort(keys, stable=True)
    k, v = keys.to(DEV), vals.to(DEV)
    ops.sort_pairs(k, v, 64)
    assert torch.equal(k.cpu(), ref_k) and torch.equal(v.cpu(), vals[order])


# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("view,off_centre", [(0, False), (1, False), (0, True)])
def test_project_bit_exact(view, off_centre):
    sc = _scene()
    # put some Gaussians behind / beside the camera and make a few huge
    sc.means[:50] *= 4.0
    sc.scales[50:60] *= 30.0
    vm, K = sc.viewmats[view], sc.Ks[view].clone()
    if off_centre:  # principal point at 30% / 62%: the FOV clamp stays symmetric about the optical axis
        K[0, 2], K[1, 2] = 0.3 * sc.width, 0.62 * sc.height  # (tests/test_oracle.py::test_fov_clamp_choice...)
    ref = O.project(sc.means, sc.quats, sc.scales, vm, K, sc.width, sc.height)
    radii, m2, d, con, comp, tiles = ops.project(
        sc.means.to(DEV), sc.quats.to(DEV), sc.scales.to(DEV), vm.to(DEV), K.to(DEV), sc.width, sc.height,
        calc_compensations=True,
    )  # fmt: skip
    assert 0 < (ref.radii > 0).sum() < sc.means.shape[0]
    assert torch.equal(radii.cpu(), ref.radii)
    # bit-exact floats feeding the integer path
    assert torch.equal(m2.cpu().view(torch.int32), ref.means2d.view(torch.int32))
    assert torch.equal(d.cpu().view(torch.int32), ref.depths.view(torch.int32))
    assert torch.equal(con.cpu().view(torch.int32), ref.conics.view(torch.int32))
    assert torch.equal(comp.cpu().view(torch.int32), ref.compensations.view(torch.int32))
    tw, th = (sc.width + 15) // 16, (sc.height + 15) // 16
    cnt, _, _ = O.isect_tiles(ref.means2d, ref.radii, ref.depths, 16, tw, th, sort=False)
    assert torch.equal(tiles.cpu(), cnt)


def test_isect_keys_order_and_ranges_bit_exact():
    sc = _scene(n=20000, w=333, h=207)  # ragged right/bottom tiles
    sc.scales[:30] *= 20.0
    vm, K = sc.viewmats[0], sc.Ks[0]
    ref = O.project(sc.means, sc.quats, sc.scales, vm, K, sc.width, sc.height)
    tw, th = (sc.width + 15) // 16, (sc.height + 15) // 16
    _, keys_u, vals_u = O.isect_tiles(ref.means2d, ref.radii, ref.depths, 16, tw, th, sort=False)
    _, keys_s, vals_s = O.isect_tiles(ref.means2d, ref.radii, ref.depths, 16, tw, th, sort=True)
    offs_ref = O.isect_offsets(keys_s, tw * th)
    radii, m2, d, con, comp, tiles = ops.project(
        sc.means.to(DEV), sc.quats.to(DEV), sc.scales.to(DEV), vm.to(DEV), K.to(DEV), sc.width, sc.height
    )
    ku, vu, _ = ops.isect_tiles(m2, radii, d, tiles, 16, tw, th, sort=False)
    assert torch.equal(ku.cpu(), keys_u) and torch.equal(vu.cpu(), vals_u)
    ks, vs, offs = ops.isect_tiles(m2, radii, d, tiles, 16, tw, th, sort=True)
    assert torch.equal(ks.cpu(), keys_s)
    assert torch.equal(vs.cpu(), vals_s)
    assert torch.equal(offs.cpu(), offs_ref)


def test_depth_first_binning_equals_full_key_sort():
    """bin_tiles (depth sort + 13-bit tile sort) must reproduce the 64-bit-key sort bit for bit."""
    sc = _scene(n=30000, w=333, h=207)
    sc.scales[:30] *= 20.0
    sc.means[100:140] = sc.means[100:101]  # identical depths: ties must stay in id order
    vm, K = sc.viewmats[1], sc.Ks[1]
    ref = O.project(sc.means, sc.quats, sc.scales, vm, K, sc.width, sc.height)
    tw, th = (sc.width + 15) // 16, (sc.height + 15) // 16
    _, keys_s, vals_s = O.isect_tiles(ref.means2d, ref.radii, ref.depths, 16, tw, th, sort=True)
    radii, m2, d, con, comp, tiles = ops.project(
        sc.means.to(DEV), sc.quats.to(DEV), sc.scales.to(DEV), vm.to(DEV), K.to(DEV), sc.width, sc.height
    )
    tk, ids, offs = ops.bin_tiles(m2, radii, d, tiles, 16, tw, th)
    assert torch.equal(ids.cpu(), vals_s)
    assert torch.equal(tk.cpu().long(), keys_s >> 32)
    assert torch.equal(offs.cpu(), O.isect_offsets(keys_s, tw * th))
    assert torch.equal(ops.isect_keys(tk, ids, d).cpu(), keys_s)


@pytest.mark.parametrize("mode,layout", [("classic", "isotropic"), ("antialiased", "isotropic"), ("classic", "needles"),
                                         ("antialiased", "needles")])
def test_footprint_rectangles_cull_only_dead_entries(mode, layout, monkeypatch):
    """The lists the compositing walks are the reference's lists minus (splat, tile) pairs in which the splat reaches
    alpha >= 1/255 at no pixel centre -- binned from the preprocess pass's footprint RECTANGLES (round 2) and, inside those,
    from its footprint MASKS (round 5: the blocks of the rectangle the ellipse itself reaches): each a subsequence, tile by
    tile, of the one before; every dropped entry is dead by the oracle's own alpha test; image bit-identical, gradients
    equal up to atomic order; and info["flatten_ids"] / ["isect_offsets"] / ["isect_ids"] still are the reference's full
    lists.  ``needles``: half of the Gaussians with one axis x 10 and one / 3, the shapes densification leaves
    (freegaussian_model.py:524-571) -- there the masks drop a quarter of what the rectangles keep (at this scene's 250-pixel
    focal length a needle is a few tiles long; 40 % at the bench's 1200)."""
    from freegaussian_amd.scenes import apply_layout

    if ops.default_context.overlap_pack:
        pytest.skip("FG_OVERLAP_PACK=1: the two-stream forward bins from the radius boxes")
    sc = _scene(n=30000, w=400, h=240, seed=13)
    if layout == "needles":
        apply_layout(sc, "needles:0.5:10")
    sc.opacities[::3] *= 0.05  # many faint splats: their 3-sigma boxes are mostly dead area
    sc.scales[:40] *= 12.0
    vm, K = sc.viewmats[:1].to(DEV), sc.Ks[:1].to(DEV)
    vr = torch.randn(1, sc.height, sc.width, 3, generator=torch.Generator().manual_seed(0)).to(DEV)
    outs = []
    for tight, exact in ((True, True), (True, False), (False, False)):
        monkeypatch.setattr(ops.default_context, "tight_rects", tight)
        monkeypatch.setattr(ops.default_context, "exact_tiles", exact)
        monkeypatch.setattr(ops.default_context, "masks_always", exact)  # (not "where they pay": this test wants them)
        t = [x.to(DEV).requires_grad_(True) for x in (sc.means, sc.quats, sc.scales, sc.opacities, sc.colors)]
        r, a, info = rasterization(*t, vm, K, sc.width, sc.height, sh_degree=3, packed=False, absgrad=True,
                                   rasterize_mode=mode)  # fmt: skip
        (r * vr).sum().backward()
        outs.append((r.detach(), a.detach(), [x.grad for x in t], info))
    (r2, a2, g2, i2), (r1, a1, g1, i1), (r0, a0, g0, i0) = outs
    assert torch.equal(r1, r0) and torch.equal(a1, a0) and torch.equal(r2, r0) and torch.equal(a2, a0)
    # (the same sums in another order: the dropped entries shift the 64-entry batches and with them which job adds a
    # splat's share first.  Round tiles: < 1e-5 pairwise; the needles' gradients are sums over hundreds of tiles per splat
    # with heavy cancellation -- two fp32 orders of such a sum differ by 6-10e-5 --: each run within the bar of the three
    # runs' fp64 mean)
    if layout == "isotropic":
        worst = max(max(rel_l2(x, y), rel_l2(z, y)) for x, y, z in zip(g1, g0, g2))
    else:
        worst = 0.0
        for x, y, z in zip(g1, g0, g2):
            mean = (x.double() + y.double() + z.double()) / 3.0
            worst = max(worst, rel_l2(x.double(), mean), rel_l2(y.double(), mean), rel_l2(z.double(), mean))
    print(f"{layout}: gradients of the three runs within {worst:.1e}")
    assert worst < (1e-5 if layout == "isotropic" else REL_TOL)
    # without footprint rectangles the raster lists ARE the reference lists
    assert i0["raster_flatten_ids"] is i0["flatten_ids"]
    full_ids, full_offs = i0["flatten_ids"].cpu(), i0["isect_offsets"].cpu()
    ids, offs = i1["raster_flatten_ids"].cpu(), i1["raster_isect_offsets"].cpu()
    ids_m, offs_m = i2["raster_flatten_ids"].cpu(), i2["raster_isect_offsets"].cpu()
    assert ids.numel() < 0.9 * full_ids.numel()  # the culling is not a no-op here
    print(f"{layout}: radius boxes {full_ids.numel()}, footprint rectangles {ids.numel()}, footprint masks {ids_m.numel()}")
    # (small round splats -- this scene's rectangles are mostly 1 x 1 and 2 x 2 tiles -- lose a corner now and then; needles
    # most of their rectangle)
    assert ids_m.numel() < (0.85 if layout == "needles" else 0.99) * ids.numel(), (ids_m.numel(), ids.numel())
    # the lazily rebuilt reference lists of the tight runs are the same lists
    for i in (i1, i2):
        assert torch.equal(i["flatten_ids"].cpu(), full_ids) and torch.equal(i["isect_offsets"].cpu(), full_offs)
    assert torch.equal(i1["isect_ids"].cpu(), i0["isect_ids"].cpu())
    ref = O.project(sc.means, sc.quats, sc.scales, sc.viewmats[0], sc.Ks[0], sc.width, sc.height)
    opac = sc.opacities * ref.compensations if mode == "antialiased" else sc.opacities
    tw = i1["tile_width"]
    dropped_total = [0, 0]
    for tile in range(0, tw * i1["tile_height"], 7):
        f = full_ids[int(full_offs[tile]) : int(full_offs[tile + 1])].tolist()
        c = ids[int(offs[tile]) : int(offs[tile + 1])].tolist()
        m = ids_m[int(offs_m[tile]) : int(offs_m[tile + 1])].tolist()
        for k, (outer, inner) in enumerate(((f, c), (c, m))):
            it = iter(outer)
            assert all(x in it for x in inner), (tile, k)  # subsequence, order kept
            dropped = sorted(set(outer) - set(inner))
            assert len(outer) - len(inner) == len(dropped)
            if dropped:
                dropped_total[k] += len(dropped)
                ty, tx = divmod(tile, tw)
                yy, xx = torch.meshgrid(torch.arange(16.0) + ty * 16 + 0.5, torch.arange(16.0) + tx * 16 + 0.5, indexing="ij")
                d = torch.tensor(dropped)
                _, _, _, alpha, valid = O._tile_terms(xx.reshape(-1), yy.reshape(-1), ref.means2d[d], ref.conics[d], opac[d])
                assert not bool(valid.any()), (tile, k, "a dropped entry reaches 1/255 somewhere")
    assert dropped_total[0] > 0 and dropped_total[1] > 0


def _enumerate_masked_lists(keys, rects, masks, tw, th):
    """The lists fg_stbin_* must produce from (depth key, rectangle, footprint mask) triples, by plain enumeration: the
    tiles of every set block, sorted by (tile, key, id)."""
    entries = []
    for i, ((rx, ry), m, k) in enumerate(zip(rects.tolist(), masks.tolist(), keys.tolist())):
        x0, y0, w, h = rx & 0xFFFF, rx >> 16, ry & 0xFFFF, ry >> 16
        if w == 0 or h == 0:
            continue
        bs = 1
        while 8 * bs < max(w, h):
            bs *= 2
        m &= (1 << 64) - 1
        for y in range(y0, y0 + h):
            for x in range(x0, x0 + w):
                if (m >> (8 * ((y - y0) // bs) + (x - x0) // bs)) & 1:
                    entries.append((y * tw + x, k & 0xFFFFFFFF, i))
    entries.sort()
    ids = torch.tensor([e[2] for e in entries], dtype=torch.int32)
    counts = torch.bincount(torch.tensor([e[0] for e in entries], dtype=torch.int64), minlength=tw * th)
    offs = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(counts, 0)]).to(torch.int32)
    return ids, offs


@pytest.mark.parametrize("seed", range(6))
def test_footprint_mask_binning_equals_plain_enumeration(seed, monkeypatch):
    """fg_stbin_count / fg_stbin_fill with footprint masks (ABI 8) against a plain enumeration of the set blocks' tiles:
    `torch.equal` lists and ranges.  ARBITRARY masks -- not only the convex ones fg_preprocess_fwd writes: holes, several
    runs per block row, empty masks, rectangles of every block size (1 ... 16 tiles per block), depth ties, capacity guesses
    that are too small -- so that the count's and the scatter's reading of a mask can never disagree."""
    g = torch.Generator().manual_seed(500 + seed)
    W, H = [(640, 400), (1920, 1080), (333, 777), (2560, 1440), (160, 96), (4000, 64)][seed]
    tw, th = (W + 15) // 16, (H + 15) // 16
    N = [3000, 20000, 5000, 8000, 300, 4000][seed]
    big = torch.rand(N, generator=g) < 0.1
    w = torch.where(big, torch.randint(1, 130, (N,), generator=g), torch.randint(1, 10, (N,), generator=g)).clamp(max=tw)
    h = torch.where(big, torch.randint(1, 130, (N,), generator=g), torch.randint(1, 10, (N,), generator=g)).clamp(max=th)
    x0 = (torch.rand(N, generator=g) * (tw - w + 1).float()).long().clamp(min=0)
    y0 = (torch.rand(N, generator=g) * (th - h + 1).float()).long().clamp(min=0)
    empty = torch.rand(N, generator=g) < 0.05
    w, h = torch.where(empty, torch.zeros_like(w), w), torch.where(empty, torch.zeros_like(h), h)
    rects = _pack_rects(x0, y0, w, h)
    dense = torch.randint(0, 2**31, (N, 2), generator=g)  # ~half of the blocks
    sparse = dense & torch.randint(0, 2**31, (N, 2), generator=g) & torch.randint(0, 2**31, (N, 2), generator=g)
    kind = torch.randint(0, 4, (N,), generator=g)
    bits = torch.where((kind == 0)[:, None], dense, torch.where((kind == 1)[:, None], sparse, torch.full_like(dense, 2**31 - 1)))
    masks = (bits[:, 0] | (bits[:, 1] << 32)) | (torch.randint(0, 2, (N,), generator=g) << 31) | (torch.randint(0, 2, (N,), generator=g) << 63)
    masks = torch.where(kind == 3, torch.zeros_like(masks), masks)
    # only bits of existing blocks may be set (what fg::footprint_mask guarantees)
    bs = torch.ones(N, dtype=torch.int64)
    for _ in range(8):
        bs = torch.where(8 * bs < torch.maximum(w, h), bs * 2, bs)
    nbx, nby = (w + bs - 1) // bs, (h + bs - 1) // bs
    valid = torch.zeros(N, dtype=torch.int64)
    for by in range(8):
        row = torch.where(by < nby, (1 << nbx) - 1, torch.zeros_like(nbx))
        valid |= row << (8 * by)
    masks = masks & valid
    keys = torch.randint(0, 50 if seed % 2 else 2**31 - 1, (N,), generator=g).to(torch.int32)  # (odd seeds: depth ties)
    want_ids, want_offs = _enumerate_masked_lists(keys, rects, masks, tw, th)
    z = torch.zeros(N, device=DEV)
    args = (torch.zeros(N, 2, device=DEV), z.int(), z, z.int(), 16, tw, th)
    monkeypatch.setattr(ops.default_context, "binning", "supertile")
    for long_segments in ("never", "always"):
        monkeypatch.setattr(ops.default_context, "long_segments", long_segments)
        ops.default_context.isect_capacity.clear()
        runs = [ops.bin_tiles(*args, keys_rects=(keys.to(DEV), rects.to(DEV), masks.to(DEV)), want_keys=False) for _ in range(2)]
        for k in list(ops.default_context.isect_capacity):
            ops.default_context.isect_capacity[k] = 1024  # far too small: the fill writes nothing, the host refills exactly
        runs.append(ops.bin_tiles(*args, keys_rects=(keys.to(DEV), rects.to(DEV), masks.to(DEV)), want_keys=False))
        for _, f, o in runs:
            assert torch.equal(o.cpu(), want_offs), long_segments
            assert torch.equal(f.cpu(), want_ids), long_segments


def test_more_than_65536_tiles_takes_the_32bit_tile_key_path():
    """257 x 257 = 66049 tiles: tile ids no longer fit the 16-bit in-workspace keys.  The lists must
    still equal the oracle's 64-bit-key sort, and the pixels of tiles with ids beyond 65535 the C
    oracle's."""
    W, H = 4112, 4100
    sc = _scene(n=8000, w=W, h=H, seed=5, cam_radius=3.0)
    vm, K = sc.viewmats[0], sc.Ks[0]
    ref = O.project(sc.means, sc.quats, sc.scales, vm, K, W, H)
    tw, th = (W + 15) // 16, (H + 15) // 16
    assert tw * th > 65536
    _, keys_s, vals_s = O.isect_tiles(ref.means2d, ref.radii, ref.depths, 16, tw, th, sort=True)
    offs_s = O.isect_offsets(keys_s, tw * th)
    gpu_in = [t.to(DEV) for t in (sc.means, sc.quats, sc.scales, sc.opacities, sc.colors)]
    r, a, info = rasterization(*gpu_in, vm[None].to(DEV), K[None].to(DEV), W, H, sh_degree=3, packed=False)
    assert torch.equal(info["flatten_ids"].cpu(), vals_s)
    assert torch.equal(info["isect_offsets"].cpu().reshape(-1), offs_s.reshape(-1))
    assert torch.equal(info["isect_ids"].cpu(), keys_s)
    # a crop whose tiles have ids beyond 65535, pixel by pixel against the scalar C oracle fed with
    # the (validated) lists of those tiles
    from oracle import c_oracle as CO

    cw, ch, x0, y0 = 640, 48, 16 * 100, 16 * 253
    offs_c, ids_c = info["isect_offsets"].cpu().reshape(-1), info["flatten_ids"].cpu()
    lists, coffs = [], [0]
    for ty in range(ch // 16):
        for tx in range(cw // 16):
            tt = (y0 // 16 + ty) * tw + (x0 // 16 + tx)
            lists.append(ids_c[int(offs_c[tt]) : int(offs_c[tt + 1])])
            coffs.append(coffs[-1] + lists[-1].numel())
    assert (y0 // 16 + 2) * tw + x0 // 16 > 65535 and coffs[-1] > 100
    cv, coffs = torch.cat(lists), torch.tensor(coffs, dtype=torch.int32)
    m2 = info["means2d"][0].cpu() - torch.tensor([float(x0), float(y0)])
    campos = torch.linalg.inv(vm)[:3, 3]
    rgb = torch.clamp_min(O.sh_eval(3, sc.means - campos, sc.colors) + 0.5, 0.0)
    rc, ac, _ = CO.raster_fwd(m2, info["conics"][0].cpu(), rgb, sc.opacities, cw, ch, 16, coffs, cv)
    assert close_except_knife_edge(r[0, y0 : y0 + ch, x0 : x0 + cw], rc, REL_TOL)
    assert close_except_knife_edge(a[0, y0 : y0 + ch, x0 : x0 + cw], ac, REL_TOL)


def test_mixed_launch_equals_classic_launch(monkeypatch):
    """Whole-tile, two-strip and single-strip jobs chosen by position and by list length
    (raster_*_mixed_kernel + fg_raster_build_jobs) against one launch shape for all tiles: the forward is bit-identical (a pixel's compositing does not depend on which
    wavefront owns it), the backward equal up to the order of its float atomics."""
    sc = _scene(n=20000, w=400, h=300, seed=23)
    vm, K = sc.viewmats[:1].to(DEV), sc.Ks[:1].to(DEV)
    outs = []
    for tail, split in (("0", "0"), ("7", "20,16"), ("3,5", "2,1"), ("2,2", "8,0"), ("100000", "0")):
        _setenv_policy(monkeypatch, "FG_RASTER_TAIL_FWD", tail)
        _setenv_policy(monkeypatch, "FG_RASTER_TAIL_BWD", tail)
        _setenv_policy(monkeypatch, "FG_RASTER_SPLIT_FWD", split)  # content-aware job sizes (job lists)
        _setenv_policy(monkeypatch, "FG_RASTER_SPLIT_BWD", split)
        t = [x.to(DEV).requires_grad_(True) for x in (sc.means, sc.quats, sc.scales, sc.opacities, sc.colors)]
        r, a, info = rasterization(*t, vm, K, sc.width, sc.height, sh_degree=3, render_mode="RGB+ED", absgrad=True, packed=False)
        g = torch.Generator().manual_seed(0)
        info["means2d"].retain_grad()
        (r * torch.randn(r.shape, generator=g).to(DEV)).sum().backward()
        outs.append((r.detach(), a.detach(), [x.grad for x in t], info["means2d"].absgrad))
    for r, a, grads, absg in outs[1:]:
        assert torch.equal(r, outs[0][0]) and torch.equal(a, outs[0][1])
        for x, y in zip(grads, outs[0][2]):
            assert rel_l2(x, y) < 1e-5
        assert rel_l2(absg, outs[0][3]) < 1e-5


@pytest.mark.parametrize("w,h", [(1600, 64), (3216, 48), (256, 832), (320, 192), (480, 270)])
def test_default_mixed_launch_on_odd_tile_grids_equals_classic_launch(w, h, monkeypatch):
    """From 200 tiles up the default is the mixed launch (strip jobs forward, list shares backward, liveness, job
    lists built inside the binning).  Grids with fewer tile rows than XCD bands (100 x 4, 201 x 3: some bands own no
    tile), a tall narrow one and the reference's quarter-resolution sizes against ONE classic launch shape: forward
    bit-identical, backward equal up to the order of its float atomics."""
    from freegaussian_amd import _lib

    sc = _scene(n=15000, w=w, h=h, seed=31)
    vm, K = sc.viewmats[:1].to(DEV), sc.Ks[:1].to(DEV)
    if int(_lib.load().fg_raster_jobs_words(w, h, 16, ops.default_context.cfg())) == 0:
        pytest.skip("classic launches forced by the environment")
    outs = []
    for tail in (None, "0"):
        if tail is not None:
            _setenv_policy(monkeypatch, "FG_RASTER_TAIL_FWD", tail)
            _setenv_policy(monkeypatch, "FG_RASTER_TAIL_BWD", tail)
            assert int(_lib.load().fg_raster_jobs_words(w, h, 16, ops.default_context.cfg())) == 0
        t = [x.to(DEV).requires_grad_(True) for x in (sc.means, sc.quats, sc.scales, sc.opacities, sc.colors)]
        r, a, info = rasterization(*t, vm, K, sc.width, sc.height, sh_degree=3, absgrad=True, packed=False)
        info["means2d"].retain_grad()
        (r * torch.randn(r.shape, generator=torch.Generator().manual_seed(0)).to(DEV)).sum().backward()
        outs.append((r.detach(), a.detach(), [x.grad for x in t], info["means2d"].absgrad))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert float(outs[0][1].max()) > 0.5  # something was rendered
    for x, y in zip(outs[0][2], outs[1][2]):
        assert rel_l2(x, y) < 1e-5
    assert rel_l2(outs[0][3], outs[1][3]) < 1e-5


@pytest.mark.parametrize("n,end_bit", [(5, 32), (4097, 13), (250_001, 32), (3_000_000, 13)])
def test_sort_pairs32_bit_exact_and_stable(n, end_bit):
    g = torch.Generator().manual_seed(n)
    keys = torch.randint(0, 2**31 - 1, (n,), generator=g) & ((1 << end_bit) - 1)
    if end_bit == 32:
        keys[::5] = keys[0]
    vals = torch.arange(n, dtype=torch.int32)
    ref_k, order = torch.sort(keys, stable=True)
    k, v = keys.to(torch.int32).to(DEV), vals.to(DEV)
    ops.sort_pairs32(k, v, end_bit)
    assert torch.equal(k.cpu().long(), ref_k) and torch.equal(v.cpu(), vals[order])


def test_speculative_binning_capacity_and_overflow():
    """bin_tiles enqueues emit + sort on guessed buffers (count read on the device) and only then
    waits for the count: same lists as the exact call; a guess that is too small is redone."""
    sc = _scene(n=30000, w=320, h=200, seed=11)
    p = O.project(sc.means, sc.quats, sc.scales, sc.viewmats[0], sc.Ks[0], sc.width, sc.height)
    tw, th = (sc.width + 15) // 16, (sc.height + 15) // 16
    cnt, _, _ = O.isect_tiles(p.means2d, p.radii, p.depths, 16, tw, th, sort=False)
    args = (p.means2d.to(DEV), p.radii.to(DEV), p.depths.to(DEV), cnt.to(DEV), 16, tw, th)
    ops.default_context.isect_capacity.clear()
    was = ops.default_context.speculative_binning
    ops.default_context.speculative_binning = False
    k0, f0, o0 = ops.bin_tiles(*args)
    ops.default_context.speculative_binning = True
    key = next(iter(ops.default_context.isect_capacity))
    assert ops.default_context.isect_capacity[key] >= f0.numel()
    k1, f1, o1 = ops.bin_tiles(*args)  # capacity path
    assert f1.numel() == f0.numel() and f1._base is not None  # a slice of the capacity buffer
    assert torch.equal(k0, k1) and torch.equal(f0, f1) and torch.equal(o0, o1)
    ops.default_context.isect_capacity[key] = max(f0.numel() // 3, 1)  # far too small: truncated, detected, redone
    k2, f2, o2 = ops.bin_tiles(*args)
    assert torch.equal(k0, k2) and torch.equal(f0, f2) and torch.equal(o0, o2)
    ops.default_context.isect_capacity[key] = f0.numel()  # exactly enough
    k3, f3, o3 = ops.bin_tiles(*args)
    assert torch.equal(k0, k3) and torch.equal(f0, f3) and torch.equal(o0, o3)
    ops.default_context.speculative_binning = was


def _supertile_vs_depth_first(N, W, H, rects, keys, monkeypatch, overflow=False):
    """The same keys / rectangles through both binning paths; returns the supertile path's lists."""
    tw, th = (W + 15) // 16, (H + 15) // 16
    z = torch.zeros(N, device=DEV)
    args = (torch.zeros(N, 2, device=DEV), z.int(), z, z.int(), 16, tw, th)
    outs = []
    # (the supertile path twice: long segments through one workgroup / through the multi-workgroup sample sort)
    # ... and (round 6) with the sample sort's per-bucket slabs cut to 1600 elements: buckets outgrow them and are
    # re-gathered between their splitters, or their whole segment goes through global memory
    for mode, long_segments, small in (("depthfirst", "auto", False), ("supertile", "never", False), ("supertile", "always", False),
                                       ("supertile", "always", True)):
        monkeypatch.setattr(ops.default_context, "binning", mode)
        monkeypatch.setattr(ops.default_context, "long_segments", long_segments)
        monkeypatch.setattr(ops.default_context, "test_small_slabs", small)
        ops.default_context.isect_capacity.clear()
        runs = [ops.bin_tiles(*args, keys_rects=(keys.clone(), rects), want_keys=False) for _ in range(2)]  # exact, speculative
        if overflow:
            for k in list(ops.default_context.isect_capacity):
                ops.default_context.isect_capacity[k] = 1024  # far too small: the fill writes nothing, the host refills exactly
            runs.append(ops.bin_tiles(*args, keys_rects=(keys.clone(), rects), want_keys=False))
        for _, f, o in runs[1:]:
            assert torch.equal(f, runs[0][1]) and torch.equal(o, runs[0][2])
        outs.append(runs[0])
    (_, f0, o0), (_, f1, o1), (_, f2, o2), (_, f3, o3) = outs
    assert torch.equal(o1, o0) and torch.equal(o2, o0) and torch.equal(o3, o0), "tile ranges differ"
    assert torch.equal(f1, f0), "lists differ"
    assert torch.equal(f2, f0), "lists differ (long-segment sample sort)"
    assert torch.equal(f3, f0), "lists differ (long-segment sample sort, buckets beyond their slabs)"
    return f2, o2


def _pack_rects(x0, y0, w, h):
    return torch.stack([x0 | (y0 << 16), w | (h << 16)], -1).to(torch.int32).contiguous()


@pytest.mark.parametrize("case", ["scene", "ties", "heavy_tile", "huge_tile", "long_cluster", "long_ties", "one", "all_culled",
                                  "few_rows", "2160p", "2880p"])
def test_supertile_binning_equals_depth_first_binning(case, monkeypatch):
    """csrc/stbin.hip (count -> scan -> scatter per supertile -> one sort per supertile, four tile lists read off
    it) against the depth-first binning on the same depth keys and footprint rectangles: `torch.equal` lists and
    ranges.
    Cases: a projected scene with a too-small capacity guess; thousands of EXACT depth ties (the id decides);
    tiles with more entries than fit the small LDS sort (the large one), than fit any (the pass through global
    memory, or -- FG_STBIN_LONG_SEGMENTS -- the sample sort: 20 000 entries = 8 buckets); a cluster of 300 000 + 70 000
    entries over two supertiles with depth ties (112 + 27 buckets) and one of 60 000 entries of ONE depth (the id alone
    decides, in the splitters as in the buckets); one Gaussian; nothing visible; an image of three tile rows; 32 400 tiles (the count kernel's grids in four passes); 57 600 tiles (the scatter's
    cursors in two passes, elements stored directly)."""
    g = torch.Generator().manual_seed(31)
    W, H = 640, 400
    if case == "scene":
        sc = _scene(n=50000, w=W, h=H, seed=19)
        t = [x.to(DEV) for x in (sc.means, sc.quats, sc.scales, sc.opacities, sc.colors)]
        _, _, _, _, _, splats = ops.preprocess(*t, None, sc.viewmats[0].to(DEV), sc.Ks[0].to(DEV), W, H, sh_degree=3)
        if not hasattr(splats, "_fg_bin"):
            pytest.skip("FG_TIGHT_RECTS=0: the preprocess pass writes no keys / rectangles")
        keys, rects = splats._fg_bin[:2]
        N = 50000
        f, o = _supertile_vs_depth_first(N, W, H, rects, keys, monkeypatch, overflow=True)
        assert f.numel() > 50_000
        return
    if case == "few_rows":
        W, H = 640, 40  # 3 tile rows, 2 supertile rows (the lower one half outside the image)
    if case == "2160p":
        W, H = 3840, 2160
    if case == "2880p":
        W, H = 5120, 2880  # 14 400 supertiles: the scatter's cursors in two passes, no staging; count grids in seven
    tw, th = (W + 15) // 16, (H + 15) // 16
    N = {"ties": 30000, "heavy_tile": 20000, "huge_tile": 40000, "long_cluster": 500_000, "long_ties": 100_000, "one": 1,
         "all_culled": 5000, "few_rows": 4000, "2160p": 200_000, "2880p": 150_000}[case]
    x0 = torch.randint(0, tw, (N,), generator=g)
    y0 = torch.randint(0, th, (N,), generator=g)
    w = torch.minimum(torch.randint(1, 5, (N,), generator=g), tw - x0)
    h = torch.minimum(torch.randint(1, 5, (N,), generator=g), th - y0)
    depth = torch.rand(N, generator=g) * 6 + 0.7
    if case == "ties":
        depth = depth[torch.randint(0, 40, (N,), generator=g)]  # 40 distinct depths: runs of hundreds of equal keys
        depth[:2000] = 3.0
    if case == "heavy_tile":
        x0[:6000], y0[:6000], w[:6000], h[:6000] = 7, 5, 2, 2  # 6000 entries in each of four tiles / supertiles (the large LDS sort)
        depth[1000:1500] = 2.5  # ... with ties inside
        w[6000:6100], h[6000:6100] = tw - x0[6000:6100], th - y0[6000:6100]  # and a few huge rectangles
    if case == "huge_tile":
        x0[:20000], y0[:20000], w[:20000], h[:20000] = 8, 4, 2, 2  # 20000 entries in one supertile: the pass through global memory
        depth[3000:3400] = 1.75
    if case == "long_cluster":
        x0[:300_000], y0[:300_000], w[:300_000], h[:300_000] = 8, 4, 2, 2  # one supertile, all four tiles
        x0[300_000:370_000], y0[300_000:370_000], w[300_000:370_000] = 11, 6, 1  # the right half of another, 1-2 tile rows
        depth[:300_000] = 3.9 + 0.2 * torch.rand(300_000, generator=g)  # a narrow band of depths ...
        depth[5000:9000] = 4.0  # ... with a run of ties
    if case == "long_ties":
        x0[:60_000], y0[:60_000], w[:60_000], h[:60_000] = 2, 2, 2, 1
        depth[:60_000] = 2.25
    keys = depth.float().view(torch.int32).clone()
    if case == "all_culled":
        w[:], h[:] = 0, 0
        keys[:] = -1
    cull = torch.rand(N, generator=g) < 0.15
    if case not in ("one",):
        w[cull], h[cull] = 0, 0
        keys[cull] = -1  # 0xFFFFFFFF
    rects = _pack_rects(x0, y0, w, h).to(DEV)
    f, o = _supertile_vs_depth_first(N, W, H, rects, keys.to(DEV), monkeypatch)
    # ... and against the rule itself: per tile, ids ordered by (depth bits, id)
    area = (w * h).long()
    assert int(o[-1]) == int(area.sum()) == f.numel()
    gid = torch.repeat_interleave(torch.arange(N), area)
    kk = torch.arange(int(area.sum())) - torch.repeat_interleave(torch.cumsum(area, 0) - area, area)
    wg = torch.clamp_min(w[gid], 1)
    tile = (y0[gid] + kk // wg) * tw + x0[gid] + kk % wg
    order = torch.argsort(((keys[gid].long() & 0xFFFFFFFF) << 20) | gid)  # (depth bits, id) ...
    order = order[torch.sort(tile[order], stable=True).indices]  # ... then stably by tile
    assert torch.equal(f.cpu().long(), gid[order])
    assert torch.equal(o.cpu().long(), torch.searchsorted(tile[order].contiguous(), torch.arange(tw * th + 1)))


def test_long_segments_switch_the_sample_sort_on_for_a_while():
    """fg_stbin_count reports the longest supertile segment beside the list length; a shape that showed one beyond
    the LDS sorts' capacity (a dense cluster) gets FG_STBIN_LONG_SEGMENTS for its next `long_cooldown` calls, renewed
    while such segments keep coming.  Same lists either way."""
    g = torch.Generator().manual_seed(77)
    W, H, N = 640, 400, 60_000
    tw, th = W // 16, H // 16
    x0 = torch.randint(0, tw, (N,), generator=g)
    y0 = torch.randint(0, th, (N,), generator=g)
    w = torch.minimum(torch.randint(1, 4, (N,), generator=g), tw - x0)
    h = torch.minimum(torch.randint(1, 4, (N,), generator=g), th - y0)
    x0[:30000], y0[:30000], w[:30000], h[:30000] = 10, 6, 2, 2  # 30 000 entries in one supertile
    keys = (torch.rand(N, generator=g) * 9 + 0.5).view(torch.int32).clone()
    rects = _pack_rects(x0, y0, w, h).to(DEV)
    z = torch.zeros(N, device=DEV)
    args = (torch.zeros(N, 2, device=DEV), z.int(), z, z.int(), 16, tw, th)
    ctx = ops.RasterContext()
    if ctx.binning != "supertile" or not ctx.direct_count or ctx.long_segments != "auto":
        pytest.skip("the environment selects another binning path / no count word / a fixed long-segment mode")
    ctx.long_cooldown = 3
    never = ops.RasterContext()
    never.long_segments = "never"
    with ops.use(never):
        ref = ops.bin_tiles(*args, keys_rects=(keys.to(DEV), rects), want_keys=False)
    assert never.long_calls == 0
    with ops.use(ctx):
        runs = [ops.bin_tiles(*args, keys_rects=(keys.to(DEV), rects), want_keys=False) for _ in range(4)]
    # call 0 sizes its list after the count arrived and already knows (one fill, flagged); calls 1-3 fill speculatively
    # with the flag the calls before left
    assert ctx.long_calls == 4 and len(ctx.long_shapes) == 1
    for _, f, o in runs:
        assert torch.equal(f, ref[1]) and torch.equal(o, ref[2])
    # the cluster dissolves: the flag stays for `long_cooldown` more calls, then goes
    x0[:30000] = torch.randint(0, tw - 1, (30000,), generator=g)
    y0[:30000] = torch.randint(0, th - 1, (30000,), generator=g)
    rects = _pack_rects(x0, y0, w, h).to(DEV)
    with ops.use(never):
        ref = ops.bin_tiles(*args, keys_rects=(keys.to(DEV), rects), want_keys=False)
    with ops.use(ctx):
        runs = [ops.bin_tiles(*args, keys_rects=(keys.to(DEV), rects), want_keys=False) for _ in range(5)]
    assert ctx.long_calls == 4 + 3 and not ctx.long_shapes
    for _, f, o in runs:
        assert torch.equal(f, ref[1]) and torch.equal(o, ref[2])
    # a light scene never sees the flag
    ctx2 = ops.RasterContext()
    with ops.use(ctx2):
        for _ in range(3):
            ops.bin_tiles(*args, keys_rects=(keys.to(DEV), rects), want_keys=False)
    assert ctx2.long_calls == 0 and not ctx2.long_shapes


@pytest.mark.parametrize("size", [(1920, 1080, 400_000), (960, 540, 100_000), (640, 368, 60_000), (2560, 1440, 2_500_000),
                                  (8192, 8192, 200_000)])  # (the last: no staged scatter -- the builders ride in the large-segment sort launch)
def test_job_lists_built_inside_the_fill_equal_those_of_the_separate_launch(size):
    """fg_stbin_fill_jobs (ABI 5): eight extra workgroups of the scatter launch build the raster job lists.
    Same int32 words as fg_raster_build_jobs on the same tile ranges, for both kinds of backward list; and
    rasterize_splats takes the lists bin_tiles left on the offsets tensor (no launch of its own)."""
    from freegaussian_amd import _lib
    W, H, N = size
    tw, th = (W + 15) // 16, (H + 15) // 16
    g = torch.Generator().manual_seed(5)
    x0 = torch.randint(0, tw, (N,), generator=g)
    y0 = torch.randint(0, th, (N,), generator=g)
    w = torch.minimum(torch.randint(1, 5, (N,), generator=g), tw - x0)
    h = torch.minimum(torch.randint(1, 5, (N,), generator=g), th - y0)
    x0[: N // 8], y0[: N // 8] = tw // 2, th // 2  # a heavy spot: tiles the content thresholds split
    w[: N // 8], h[: N // 8] = 1, 1
    keys = (torch.rand(N, generator=g) * 9 + 0.5).view(torch.int32).clone()
    rects = _pack_rects(x0, y0, w, h).to(DEV)
    z = torch.zeros(N, device=DEV)
    args = (torch.zeros(N, 2, device=DEV), z.int(), z, z.int(), 16, tw, th)
    for budget_mb in (2048, 0):  # list shares for the backward / pixel strips
        ctx = ops.RasterContext()
        if ctx.binning != "supertile" or not ctx.jobs_in_fill:
            pytest.skip("the environment selects another path")
        ctx.seg_ckpt_budget_bytes = budget_mb << 20
        ctx.heavy_tiles = "never"  # (the heavy tiles' two extra lists are filled in atomic order: compared as images, below)
        with ops.use(ctx):
            for _ in range(2):  # exact, then speculative
                _, ids, offs = ops.bin_tiles(*args, keys_rects=(keys.to(DEV), rects), want_keys=False,
                                             raster_hint=(3, W, H))  # fmt: skip
                words = int(_lib.load().fg_raster_jobs_words(W, H, 16, ctx.cfg()))
                if words == 0:
                    assert offs._fg_jobs is None
                    continue
                jobs, shares, key, _cfgp = offs._fg_jobs
                assert key == (ctx, 3, W, H, 16) and jobs.shape == (2, words)
                assert shares == (budget_mb > 0 and int(_lib.load().fg_raster_seg_ckpt_floats(3, W, H, 16, ids.numel(), ctx.cfg())) > 0)
                ref = torch.zeros_like(jobs)
                ops._call("fg_raster_build_jobs", W, H, 16, ops._ptr(offs), ops._ptr(ref[0]), ops._ptr(ref[1]), int(shares),
                          _cfgp, ops._stream())  # fmt: skip
                torch.cuda.synchronize()
                cap = (words - 8 - tw * th) // 8  # (behind the eight segments: the table of first checkpoint slots)
                for l in range(2):
                    assert torch.equal(jobs[l, :8], ref[l, :8]), "jobs per XCD differ"
                    for x in range(8):
                        n = int(ref[l, x])
                        assert 0 < n <= cap
                        assert torch.equal(jobs[l, 8 + x * cap : 8 + x * cap + n], ref[l, 8 + x * cap : 8 + x * cap + n])


def test_rasterization_takes_the_job_lists_of_the_fill(monkeypatch):
    """End to end: with the default context rasterization() issues fg_stbin_fill_jobs and NO fg_raster_build_jobs;
    with jobs_in_fill off the other way round; same image bit for bit, same gradients up to atomic order."""
    sc = _scene(n=40000, w=640, h=368, seed=21)  # 920 tiles: the mixed launches with job lists
    vm, K = sc.viewmats[:1].to(DEV), sc.Ks[:1].to(DEV)
    vr = torch.randn(1, sc.height, sc.width, 3, generator=torch.Generator().manual_seed(0)).to(DEV)
    outs, calls = [], []
    real = ops._call
    monkeypatch.setattr(ops, "_call", lambda name, *a, **k: (calls.append(name), real(name, *a, **k))[1])
    for in_fill in (True, False):
        ctx = ops.RasterContext()
        if ctx.binning != "supertile" or not ctx.jobs_in_fill or ctx.overlap_pack or not ctx.tight_rects:
            pytest.skip("the environment selects another path")
        if int(_lib.load().fg_raster_jobs_words(sc.width, sc.height, 16, ctx.cfg())) == 0:
            pytest.skip("classic launches forced by the environment: no job lists")
        ctx.jobs_in_fill = in_fill
        ctx.step_calls = False  # (the stage-wise calls are what this test looks at; fg_step_fwd makes them inside the library)
        for _ in range(2):  # exact lists, then speculative
            del calls[:]
            t = [x.to(DEV).requires_grad_(True) for x in (sc.means, sc.quats, sc.scales, sc.opacities, sc.colors)]
            with ops.use(ctx):
                r, a, info = rasterization(*t, vm, K, sc.width, sc.height, sh_degree=3, packed=False, absgrad=True)
                (r * vr).sum().backward()
            assert ("fg_stbin_fill_jobs" in calls) == in_fill and ("fg_raster_build_jobs" in calls) == (not in_fill), calls
            outs.append((r.detach(), a.detach(), [x.grad for x in t]))
    for r, a, g in outs[1:]:
        assert torch.equal(r, outs[0][0]) and torch.equal(a, outs[0][1])
        for x, y in zip(g, outs[0][2]):
            assert rel_l2(x, y) < 1e-5


@pytest.mark.parametrize("seed", range(12))
def test_supertile_binning_fuzz_against_depth_first(seed, monkeypatch):
    """Random image sizes (1 x 1 tiles ... 300 x 170), Gaussian counts (1 ... 200k), rectangle size mixes (points,
    a few tiles, a tenth of the image), depth-tie densities and culled fractions through both binning paths:
    `torch.equal` lists and ranges, with a speculative capacity, and with one that is too small."""
    g = torch.Generator().manual_seed(1000 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))  # noqa: E731
    tw, th = (ri(1, 12), ri(1, 12)) if seed % 4 == 0 else (ri(8, 300), ri(5, 170))
    W, H = 16 * tw - ri(0, 15), 16 * th - ri(0, 15)
    tw, th = (W + 15) // 16, (H + 15) // 16
    N = ri(1, 2000) if seed % 3 == 0 else ri(2000, 200_000)
    x0 = torch.randint(0, tw, (N,), generator=g)
    y0 = torch.randint(0, th, (N,), generator=g)
    kind = torch.rand(N, generator=g)
    big = max(2, tw // 3), max(2, th // 3)
    w = torch.where(kind < 0.6, torch.randint(1, 4, (N,), generator=g), torch.randint(1, 9, (N,), generator=g))
    h = torch.where(kind < 0.6, torch.randint(1, 4, (N,), generator=g), torch.randint(1, 9, (N,), generator=g))
    huge = kind > 0.995
    w = torch.where(huge, torch.randint(1, big[0] + 1, (N,), generator=g), w)
    h = torch.where(huge, torch.randint(1, big[1] + 1, (N,), generator=g), h)
    w, h = torch.minimum(w, tw - x0), torch.minimum(h, th - y0)
    depth = torch.rand(N, generator=g) * 50 + 0.01
    if seed % 2:
        depth = depth[torch.randint(0, max(1, N // 50), (N,), generator=g)]  # runs of ~50 equal keys
    keys = depth.float().view(torch.int32).clone()
    cull = torch.rand(N, generator=g) < float(torch.rand(1, generator=g)) * 0.5
    w[cull], h[cull] = 0, 0
    keys[cull] = -1
    rects = _pack_rects(x0, y0, w, h).to(DEV)
    f, o = _supertile_vs_depth_first(N, W, H, rects, keys.to(DEV), monkeypatch, overflow=True)
    assert int(o[-1]) == int((w * h).sum()) == f.numel()


@pytest.mark.parametrize("raw,mode", [(False, "RGB"), (False, "RGB+ED"), (True, "RGB"), (True, "RGB+ED")])
def test_one_call_per_direction_equals_the_stage_wise_calls(raw, mode, monkeypatch):
    """fg_step_fwd / fg_step_bwd (ABI 7: the whole view as one C-ABI call per direction, two workspaces) against the
    stage-wise calls on the same inputs: the same kernels in the same order -- image, alpha, last_ids, radii, lists bit
    for bit, every gradient, info["means2d"].grad and .absgrad up to the order of the backward's atomics.  Both front
    ends (rasterization, rasterize_gauss_params with background + clamp), with and without a depth channel; a capacity
    guess that is too small is redone inside the call."""
    from freegaussian_amd.rasterization import rasterize_gauss_params

    sc = _scene(n=30000, w=640, h=368, seed=23)  # 920 tiles: job-list launches
    vm, K = sc.viewmats[:1].to(DEV), sc.Ks[:1].to(DEV)
    C = 4 if mode.endswith("D") else 3
    vr = torch.randn(1, sc.height, sc.width, C, generator=torch.Generator().manual_seed(1)).to(DEV)
    ctx = ops.RasterContext()
    if not ctx.step_calls or ctx.binning != "supertile" or int(_lib.load().fg_raster_jobs_words(640, 368, 16, ctx.cfg())) == 0:
        pytest.skip("the environment selects the stage-wise path")
    calls = []
    real = ops._call
    monkeypatch.setattr(ops, "_call", lambda name, *a, **k: (calls.append(name), real(name, *a, **k))[1])

    def run():
        if raw:
            p = dict(means=sc.means, quats=sc.quats, log_scales=sc.scales.log(), opacity_logits=torch.logit(sc.opacities.clamp(1e-4, 1 - 1e-4)),
                     features_dc=sc.colors[:, 0, :].contiguous(), features_rest=sc.colors[:, 1:, :].contiguous())  # fmt: skip
            t = {k: v.to(DEV).requires_grad_(True) for k, v in p.items()}
            r, a, info = rasterize_gauss_params(t["means"], t["quats"], t["log_scales"], t["opacity_logits"], t["features_dc"],
                                                t["features_rest"], vm, K, sc.width, sc.height, 3, render_mode=mode, absgrad=True,
                                                background=torch.tensor([0.2, 0.7, 0.4], device=DEV), clamp=True, ctx=ctx)  # fmt: skip
        else:
            t = {k: getattr(sc, k).to(DEV).requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "colors")}
            r, a, info = rasterization(*t.values(), vm, K, sc.width, sc.height, sh_degree=3, packed=False, render_mode=mode,
                                       absgrad=True, ctx=ctx)  # fmt: skip
        info["means2d"].retain_grad()
        ((r * vr).sum() + 0.5 * a.sum()).backward()
        torch.cuda.synchronize()
        return (r.detach(), a.detach(), info["last_ids"], info["radii"], info["raster_flatten_ids"], info["raster_isect_offsets"],
                {k: v.grad for k, v in t.items()}, info["means2d"].grad.clone(), info["means2d"].absgrad.clone())  # fmt: skip

    ctx.step_calls = False
    ref = run()  # (also measures the shape's list capacity)
    ref = run()
    assert "fg_stbin_fill_jobs" in calls
    ctx.step_calls = True
    del calls[:]
    outs = [run()]
    assert not calls, calls  # no stage-wise entry point was called
    for k in list(ctx.isect_capacity):
        ctx.isect_capacity[k] = 2048  # far too small: an empty image, detected, the call repeated with the list's length
    redos = ctx.capacity_redos
    outs.append(run())
    assert ctx.capacity_redos == redos + 1 and not calls
    for o in outs:
        for i in range(6):
            assert torch.equal(o[i], ref[i]), i
        for k in ref[6]:
            assert rel_l2(o[6][k], ref[6][k]) < 1e-5, k
        assert rel_l2(o[7], ref[7]) < 1e-5 and rel_l2(o[8], ref[8]) < 1e-5


@pytest.mark.parametrize("raw", [False, True])
def test_one_call_per_direction_carries_losses_on_info_depths_and_conics(raw):
    """A loss on the differentiable per-Gaussian outputs (info["depths"], info["conics"]: the flow terms read them) must
    reach the parameters through fg_step_bwd exactly as through the stage-wise calls -- all four branches of fg_step_bwd
    forward v_depths / v_conics (round 4's raw branch passed null for both: the gradient vanished from the second call
    of a shape on)."""
    from freegaussian_amd.rasterization import rasterize_gauss_params

    sc = _scene(n=30000, w=640, h=368, seed=29)
    vm, K = sc.viewmats[:1].to(DEV), sc.Ks[:1].to(DEV)
    ctx = ops.RasterContext()
    if not ctx.step_calls or ctx.binning != "supertile" or int(_lib.load().fg_raster_jobs_words(640, 368, 16, ctx.cfg())) == 0:
        pytest.skip("the environment selects the stage-wise path")
    g = torch.Generator().manual_seed(5)
    wd, wc = torch.randn(1, sc.means.shape[0], generator=g).to(DEV), torch.randn(1, sc.means.shape[0], 3, generator=g).to(DEV)

    def run():
        if raw:
            p = dict(means=sc.means, quats=sc.quats, log_scales=sc.scales.log(), opacity_logits=torch.logit(sc.opacities.clamp(1e-4, 1 - 1e-4)),
                     features_dc=sc.colors[:, 0, :].contiguous(), features_rest=sc.colors[:, 1:, :].contiguous())  # fmt: skip
            t = {k: v.to(DEV).requires_grad_(True) for k, v in p.items()}
            r, a, info = rasterize_gauss_params(t["means"], t["quats"], t["log_scales"], t["opacity_logits"], t["features_dc"],
                                                t["features_rest"], vm, K, sc.width, sc.height, 3, ctx=ctx)  # fmt: skip
        else:
            t = {k: getattr(sc, k).to(DEV).requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "colors")}
            r, a, info = rasterization(*t.values(), vm, K, sc.width, sc.height, sh_degree=3, packed=False, ctx=ctx)
        # ONLY the per-Gaussian outputs carry the loss: nothing but v_depths / v_conics can move the parameters
        ((info["depths"] * wd).sum() + 1e-3 * (info["conics"] * wc).sum() + 0.0 * r.sum()).backward()
        torch.cuda.synchronize()
        return {k: v.grad for k, v in t.items()}

    ctx.step_calls = False
    run()
    ref = run()
    ctx.step_calls = True
    calls_before = ctx.capacity_redos
    got = run()
    assert ctx.capacity_redos == calls_before
    for k in ("means", "quats", "log_scales" if raw else "scales"):
        assert float(ref[k].abs().max()) > 0, k
        assert rel_l2(got[k], ref[k]) < 1e-6, k


def test_learned_launch_state_carries_over_a_change_of_the_gaussian_count():
    """The reference changes N every `refine_every` steps (split / duplicate / cull, freegaussian_model.py:404-571).  What the
    host has learned about a shape -- list capacity, checkpoint-slot needs, the even-scene / long-segment / heavy-tile flags --
    is keyed by the tile grid and scaled by the ratio of the counts: a call with N +- 3 % goes straight through
    fg_step_fwd / fg_step_bwd (no stage-wise call, no redo, compact checkpoint slots), with the results of a fresh context's
    stage-wise calls."""
    sc = synthetic_scene(1_000_000, 1920, 1080, n_views=2, sh_degree=3, seed=42)
    vm, K = sc.viewmats[:1].to(DEV), sc.Ks[:1].to(DEV)
    ctx = ops.RasterContext()
    ctx.masks_always = True  # (the lists of two contexts are compared below: both with footprint masks, whatever they keep)
    if not ctx.step_calls or ctx.binning != "supertile" or not ctx.compact_slots:
        pytest.skip("the environment selects the stage-wise path")
    vr = torch.randn(1, 1080, 1920, 3, generator=torch.Generator().manual_seed(1)).to(DEV)
    full = {k: getattr(sc, k).to(DEV) for k in ("means", "quats", "scales", "opacities", "colors")}

    def rows(n):  # the first n Gaussians, or all of them + copies of the first n - N (a densification's clones)
        N = sc.means.shape[0]
        return {k: (v[:n] if n <= N else torch.cat([v, v[: n - N]])).contiguous() for k, v in full.items()}

    def run(c, n):
        t = {k: v.clone().requires_grad_(True) for k, v in rows(n).items()}
        r, a, info = rasterization(*t.values(), vm, K, 1920, 1080, sh_degree=3, packed=False, absgrad=True, ctx=c)
        (r * vr).sum().backward()
        torch.cuda.synchronize()
        return r.detach(), info["radii"], info["raster_flatten_ids"], {k: v.grad for k, v in t.items()}

    for _ in range(4):  # the first call measures; the needs of the checkpoint slots are read one call late
        run(ctx, 1_000_000)
    assert ctx.last_seg_slots > 0
    for n in (970_000, 1_030_000, 1_000_000):
        assert ops.step_path_available(ctx, n, 1920, 1080, 16, torch.device(DEV, torch.cuda.current_device()))
        before = (ctx.stagewise_raster_calls, ctx.capacity_redos, ctx.full_ckpt_allocs)
        got = run(ctx, n)
        assert (ctx.stagewise_raster_calls, ctx.capacity_redos, ctx.full_ckpt_allocs) == before, (n, before)
        fresh = ops.RasterContext()
        fresh.step_calls = False
        fresh.masks_always = True
        ref = run(fresh, n)
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]) and torch.equal(got[2], ref[2])
        for k in ref[3]:
            assert rel_l2(got[3][k], ref[3][k]) < 1e-5, (n, k)
    # a different scene altogether on the same tile grid (N beyond a factor of two): measured afresh
    assert not ops.step_path_available(ctx, 300_000, 1920, 1080, 16, torch.device(DEV, torch.cuda.current_device()))


def test_second_backward_through_the_one_call_node_is_a_gradient_not_a_sum():
    """retain_graph=True and a second backward through the same fg_step_bwd node (per-loss gradients): the record-gradient
    array of the kept workspace is zero-filled by the forward launch ONCE -- the second pass must clear it itself, and
    info["means2d"].grad (a view of that array after the first pass) accumulates like any retain_grad()'ed tensor."""
    sc = _scene(n=30000, w=640, h=368, seed=31)
    vm, K = sc.viewmats[:1].to(DEV), sc.Ks[:1].to(DEV)
    ctx = ops.RasterContext()
    if not ctx.step_calls or ctx.binning != "supertile" or int(_lib.load().fg_raster_jobs_words(640, 368, 16, ctx.cfg())) == 0:
        pytest.skip("the environment selects the stage-wise path")
    g = torch.Generator().manual_seed(2)
    v1 = torch.randn(1, sc.height, sc.width, 3, generator=g).to(DEV)
    v2 = torch.randn(1, sc.height, sc.width, 3, generator=g).to(DEV)

    def leaves():
        return [getattr(sc, k).to(DEV).requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "colors")]

    def single(v):
        t = leaves()
        r, a, info = rasterization(*t, vm, K, sc.width, sc.height, sh_degree=3, packed=False, absgrad=True, ctx=ctx)
        info["means2d"].retain_grad()
        (r * v).sum().backward()
        return [x.grad.clone() for x in t], info["means2d"].grad.clone(), info["means2d"].absgrad.clone()

    single(v1)  # (measures the capacity: the next calls take fg_step_*)
    g1, m1, _ = single(v1)
    g2, m2, abs2 = single(v2)
    t = leaves()
    r, a, info = rasterization(*t, vm, K, sc.width, sc.height, sh_degree=3, packed=False, absgrad=True, ctx=ctx)
    assert type(r.grad_fn).__name__ != "_RasterSplatsBackward"
    info["means2d"].retain_grad()
    (r * v1).sum().backward(retain_graph=True)
    first = [x.grad.clone() for x in t]
    m_first = info["means2d"].grad.clone()
    for x in t:
        x.grad = None
    (r * v2).sum().backward(retain_graph=True)
    torch.cuda.synchronize()
    second = [x.grad.clone() for x in t]
    for x in t:
        x.grad = None
    # a third pass: the node's in-place clearing of the record gradients must not have touched the version counter its
    # outputs share (autograd refuses a view whose base "has been modified inplace")
    (r * v2).sum().backward()
    torch.cuda.synchronize()
    for a_, b_ in zip(first, g1):
        assert rel_l2(a_, b_) < 1e-5
    for a_, b_ in zip(second, g2):
        assert rel_l2(a_, b_) < 1e-5  # the second loss's gradient alone, not the sum of the two
    for x, b_ in zip(t, g2):
        assert rel_l2(x.grad, b_) < 1e-5
    assert rel_l2(m_first, m1) < 1e-5
    assert rel_l2(info["means2d"].grad, m1 + 2 * m2) < 1e-5  # accumulated over the three passes, as retain_grad() does
    assert rel_l2(info["means2d"].absgrad, abs2) < 1e-5


@pytest.mark.parametrize("w,h,n,mode,rmode", [(160, 96, 3000, "RGB+ED", "classic"), (640, 368, 30000, "RGB", "antialiased")])
def test_camera_pose_gradient_vs_oracle(w, h, n, mode, rmode):
    """``viewmats.requires_grad`` (the reference's CameraOptimizer, freegaussian_model.py:120, applied at :774): the fused
    path returns dL/d viewmat -- fg_viewmat_bwd over the cotangents of the per-Gaussian backward, the SH view direction's
    share pulled back through inverse(viewmat) -- within REL_TOL of the oracle's autograd; through the stage-wise calls
    (first call of a shape), the one-call-per-direction path (second call) and the raw-parameter front end; the parameter
    gradients beside it unchanged.  The intrinsics and the stage-by-stage operators refuse instead of returning zeros."""
    from freegaussian_amd.rasterization import rasterize_gauss_params

    sc = _scene(n=n, w=w, h=h, seed=41)
    g = torch.Generator().manual_seed(3)
    C = 4 if mode.endswith("D") else 3
    vr, va = torch.randn(1, h, w, C, generator=g), torch.randn(1, h, w, 1, generator=g)
    ins0 = [t.clone().requires_grad_(True) for t in (sc.means, sc.quats, sc.scales, sc.opacities, sc.colors)]
    vm0 = sc.viewmats[:1].clone().requires_grad_(True)
    r0, a0, _ = O.rasterization(*ins0, vm0, sc.Ks[:1], w, h, sh_degree=3, render_mode=mode, rasterize_mode=rmode)
    ((r0 * vr).sum() + (a0 * va).sum()).backward()
    assert float(vm0.grad[0, :3].abs().max()) > 0  # (row 3 carries what inverse() makes of it, on both sides)
    ctx = ops.RasterContext()
    for call in range(2):  # stage-wise, then (where the image takes job lists) fg_step_*
        ins1 = [t.detach().to(DEV).requires_grad_(True) for t in ins0]
        vm1 = sc.viewmats[:1].to(DEV).requires_grad_(True)
        r1, a1, _ = rasterization(*ins1, vm1, sc.Ks[:1].to(DEV), w, h, sh_degree=3, packed=False, render_mode=mode,
                                  rasterize_mode=rmode, ctx=ctx)  # fmt: skip
        ((r1 * vr.to(DEV)).sum() + (a1 * va.to(DEV)).sum()).backward()
        assert rel_l2(vm1.grad, vm0.grad) < REL_TOL, call
        for x, y in zip(ins1, ins0):
            assert rel_l2(x.grad, y.grad) < REL_TOL
    if mode == "RGB":  # the raw-parameter front end (the model's path): same pose gradient
        for call in range(2):
            p = dict(means=sc.means, quats=sc.quats, log_scales=sc.scales.log(), opacity_logits=torch.logit(sc.opacities.clamp(1e-6, 1 - 1e-6)),
                     features_dc=sc.colors[:, 0, :].contiguous(), features_rest=sc.colors[:, 1:, :].contiguous())  # fmt: skip
            t = {k: v.to(DEV).requires_grad_(True) for k, v in p.items()}
            vm2 = sc.viewmats[:1].to(DEV).requires_grad_(True)
            r2, a2, _ = rasterize_gauss_params(t["means"], t["quats"], t["log_scales"], t["opacity_logits"], t["features_dc"],
                                               t["features_rest"], vm2, sc.Ks[:1].to(DEV), w, h, 3, rasterize_mode=rmode, ctx=ctx)  # fmt: skip
            ((r2 * vr.to(DEV)).sum() + (a2 * va.to(DEV)).sum()).backward()
            assert rel_l2(vm2.grad, vm0.grad) < REL_TOL, call
    t = [x.to(DEV) for x in (sc.means, sc.quats, sc.scales, sc.opacities, sc.colors)]
    vm = sc.viewmats[:1].to(DEV).requires_grad_(True)
    with pytest.raises(NotImplementedError, match="fused"):
        rasterization(*t, vm, sc.Ks[:1].to(DEV), w, h, sh_degree=3, packed=False, fused=False)
    with pytest.raises(NotImplementedError, match="intrinsics"):
        rasterization(*t, vm.detach(), sc.Ks[:1].to(DEV).requires_grad_(True), w, h, sh_degree=3, packed=False)
    with torch.no_grad():  # (no tape, nothing to refuse)
        rasterization(*t, vm, sc.Ks[:1].to(DEV), w, h, sh_degree=3, packed=False, fused=False)


def test_rasterization_redoes_the_composite_when_the_list_guess_was_too_small():
    sc = _scene(n=20000, w=256, h=160, seed=17)
    t = [x.to(DEV) for x in (sc.means, sc.quats, sc.scales, sc.opacities, sc.colors)]
    vm, K = sc.viewmats[:1].to(DEV), sc.Ks[:1].to(DEV)
    ops.default_context.isect_capacity.clear()
    r0, a0, i0 = rasterization(*t, vm, K, sc.width, sc.height, sh_degree=3, packed=False)  # exact (no guess yet)
    r1, a1, i1 = rasterization(*t, vm, K, sc.width, sc.height, sh_degree=3, packed=False)  # speculative
    for key in list(ops.default_context.isect_capacity):
        ops.default_context.isect_capacity[key] = 1000  # force the overflow path
    r2, a2, i2 = rasterization(*t, vm, K, sc.width, sc.height, sh_degree=3, packed=False)
    for r, a, i in ((r1, a1, i1), (r2, a2, i2)):
        assert torch.equal(r, r0) and torch.equal(a, a0)
        assert torch.equal(i["flatten_ids"], i0["flatten_ids"]) and torch.equal(i["isect_offsets"], i0["isect_offsets"])
        assert torch.equal(i["tile_keys"], i0["tile_keys"])


def test_two_stream_forward_gives_identical_results():
    """ops.default_context.overlap_pack: projection on the current stream, colour + record packing (fg_sh_pack_fwd)
    on a side stream overlapping the binning; the raster forward waits for the records' event."""
    sc = _scene(n=15000, w=240, h=144, seed=23)
    vm, K = sc.viewmats[:1].to(DEV), sc.Ks[:1].to(DEV)
    outs = []
    was = ops.default_context.overlap_pack
    try:
        for flag in (False, True):
            ops.default_context.overlap_pack = flag
            t = [x.to(DEV).requires_grad_(True) for x in (sc.means, sc.quats, sc.scales, sc.opacities, sc.colors)]
            r, a, info = rasterization(*t, vm, K, sc.width, sc.height, sh_degree=3, render_mode="RGB+ED",
                                       packed=False, absgrad=True, rasterize_mode="antialiased")  # fmt: skip
            (r.sum() + a.sum()).backward()
            outs.append((r.detach(), a.detach(), info["radii"], [x.grad for x in t]))
    finally:
        ops.default_context.overlap_pack = was
    (r0, a0, rad0, g0), (r1, a1, rad1, g1) = outs
    assert torch.equal(r0, r1) and torch.equal(a0, a1) and torch.equal(rad0, rad1)
    for x, y in zip(g0, g1):
        assert rel_l2(y, x) < 1e-5


def test_isect_empty_scene():
    """All Gaussians behind the camera: I = 0, every range empty, render = 0."""
    sc = plumbing_scene()
    means = sc.means.clone()
    means[:, 2] -= 100.0
    r, a, info = rasterization(means.to(DEV), sc.quats.to(DEV), sc.scales.to(DEV), sc.opacities.to(DEV),
                               sc.colors.to(DEV), sc.viewmats.to(DEV), sc.Ks.to(DEV), 128, 128, sh_degree=0,
                               packed=False)  # fmt: skip
    assert info["flatten_ids"].numel() == 0
    assert int(info["isect_offsets"].abs().sum()) == 0
    assert float(r.abs().max()) == 0.0 and float(a.abs().max()) == 0.0
    assert int((info["radii"] > 0).sum()) == 0


# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("degree", [0, 1, 2, 3])
def test_sh_forward_backward(degree):
    sc = _scene(n=3000)
    vm = sc.viewmats[1]
    radii = torch.ones(3000, dtype=torch.int32)
    radii[::7] = 0
    coeffs = sc.colors.clone().requires_grad_(True)
    means = sc.means.clone().requires_grad_(True)
    campos = torch.linalg.inv(vm)[:3, 3]
    ref = torch.clamp_min(O.sh_eval(degree, means - campos, coeffs) + 0.5, 0.0) * (radii > 0)[:, None]
    vcol = torch.randn(3000, 3, generator=torch.Generator().manual_seed(1))
    (ref * vcol).sum().backward()
    c2 = sc.colors.to(DEV).requires_grad_(True)
    m2 = sc.means.to(DEV).requires_grad_(True)
    out = ops.spherical_harmonics(degree, m2, vm.to(DEV), c2, radii.to(DEV))
    (out * vcol.to(DEV)).sum().backward()
    assert rel_err(out, ref) < REL_TOL
    assert rel_err(c2.grad, coeffs.grad) < REL_TOL
    if degree > 0:
        assert rel_err(m2.grad, means.grad) < REL_TOL
    else:
        assert float(m2.grad.abs().max()) == 0.0


# ------------------------------------------------------------------------------------------
def _raster_inputs(sc, view=0, channels=3, seed=0):
    vm, K = sc.viewmats[view], sc.Ks[view]
    ref = O.project(sc.means, sc.quats, sc.scales, vm, K, sc.width, sc.height)
    tw, th = (sc.width + 15) // 16, (sc.height + 15) // 16
    _, keys, vals = O.isect_tiles(ref.means2d, ref.radii, ref.depths, 16, tw, th)
    offs = O.isect_offsets(keys, tw * th)
    g = torch.Generator().manual_seed(seed)
    feats = torch.rand(sc.means.shape[0], channels, generator=g)
    return ref, feats, offs, vals


@pytest.mark.parametrize("channels,ppt", [(1, 0), (3, 0), (4, 0), (6, 0), (8, 0), (3, 1), (3, 2), (3, 4), (5, 2), (4, 4)])
def test_raster_forward_backward_vs_oracle(channels, ppt, monkeypatch):
    """ppt = 0: the library's own choice for this tile count; 1 / 2 / 4: every pixels-per-lane body
    of the classic launch (4 / 2 / 1 wavefronts per tile) directly against the oracle."""
    if ppt:
        _setenv_policy(monkeypatch, "FG_RASTER_PPT_FWD", str(ppt))
        _setenv_policy(monkeypatch, "FG_RASTER_PPT_BWD", str(ppt))
    sc = _scene(n=8000, w=150, h=100)  # 150x100: partial tiles on both edges
    sc.opacities[:400] = 1.0  # exercise the 0.999 clamp branch
    ref, feats, offs, vals = _raster_inputs(sc, channels=channels)
    W, H = sc.width, sc.height
    r_ref, a_ref, last_ref = O.rasterize(ref.means2d, ref.conics, feats, sc.opacities, W, H, 16, offs, vals)
    g = torch.Generator().manual_seed(5)
    vr, va = torch.randn(H, W, channels, generator=g), torch.randn(H, W, 1, generator=g)
    gref = O.rasterize_backward(ref.means2d, ref.conics, feats, sc.opacities, W, H, 16, offs, vals, vr, va[..., 0],
                                alpha_out=a_ref)  # fmt: skip

    m2 = ref.means2d.to(DEV).requires_grad_(True)
    con = ref.conics.to(DEV).requires_grad_(True)
    ft = feats.to(DEV).requires_grad_(True)
    op = sc.opacities.to(DEV).requires_grad_(True)
    r, a, last = ops.rasterize_to_pixels(m2, con, ft, op, W, H, 16, offs.to(DEV), vals.to(DEV), absgrad=True)
    ((r * vr.to(DEV)).sum() + (a * va.to(DEV)).sum()).backward()
    assert rel_err(r, r_ref) < REL_TOL
    assert rel_err(a, a_ref) < REL_TOL
    # last contributing index: integer, only knife-edge pixels may differ (helpers.KNIFE_EDGE_*: the observed bound)
    assert last_ids_agree(last, last_ref)
    for name, x, y in [("means2d", m2.grad, gref[0]), ("absgrad", m2.absgrad, gref[1]), ("conics", con.grad, gref[2]),
                       ("features", ft.grad, gref[3]), ("opacities", op.grad, gref[4])]:  # fmt: skip
        assert rel_l2(x, y) < REL_TOL, name
        assert rel_err(x, y) < REL_TOL, name


def test_raster_dense_stack_hits_transmittance_stop():
    """Many opaque splats on top of each other: pixels must stop at T <= 1e-4 like the oracle."""
    g = torch.Generator().manual_seed(11)
    N, W, H = 600, 64, 48
    m2 = torch.rand(N, 2, generator=g) * torch.tensor([W, H])
    conics = torch.tensor([[0.02, 0.0, 0.02]]).repeat(N, 1)
    op = torch.full((N,), 0.9)
    depths = torch.rand(N, generator=g) + 1
    radii = torch.full((N,), 40, dtype=torch.int32)
    feats = torch.rand(N, 3, generator=g)
    tw, th = (W + 15) // 16, (H + 15) // 16
    _, keys, vals = O.isect_tiles(m2, radii, depths, 16, tw, th)
    offs = O.isect_offsets(keys, tw * th)
    r_ref, a_ref, last_ref = O.rasterize(m2, conics, feats, op, W, H, 16, offs, vals)
    assert (a_ref > 1 - 1.5e-4).float().mean() > 0.5  # the stop rule is what ends most pixels
    vr, va = torch.randn(H, W, 3, generator=g), torch.randn(H, W, 1, generator=g)
    gref = O.rasterize_backward(m2, conics, feats, op, W, H, 16, offs, vals, vr, va[..., 0], alpha_out=a_ref)
    t = [x.to(DEV).requires_grad_(True) for x in (m2, conics, feats, op)]
    r, a, last = ops.rasterize_to_pixels(*t, W, H, 16, offs.to(DEV), vals.to(DEV), absgrad=True)
    ((r * vr.to(DEV)).sum() + (a * va.to(DEV)).sum()).backward()
    assert rel_err(r, r_ref) < REL_TOL and rel_err(a, a_ref) < REL_TOL
    assert last_ids_agree(last, last_ref)
    for x, y in zip([t[0].grad, t[0].absgrad, t[1].grad, t[2].grad, t[3].grad], gref):
        assert rel_l2(x, y) < REL_TOL


# ------------------------------------------------------------------------------------------
def _run_both(sc, view, render_mode, sh_degree, rasterize_mode="classic", extra=None, seed=0):
    names = ["means", "quats", "scales", "opacities", "colors"]
    cpu = [sc.means, sc.quats, sc.scales, sc.opacities, sc.colors]
    ref_in = [t.clone().requires_grad_(True) for t in cpu]
    gpu_in = [t.to(DEV).requires_grad_(True) for t in cpu]
    kw = dict(width=sc.width, height=sc.height, sh_degree=sh_degree, render_mode=render_mode, packed=False,
              absgrad=True, rasterize_mode=rasterize_mode)  # fmt: skip
    vm, K = sc.viewmats[view : view + 1], sc.Ks[view : view + 1]
    r0, a0, i0 = O.rasterization(*ref_in, vm, K, extra_channels=extra, **kw)
    r1, a1, i1 = rasterization(*gpu_in, vm.to(DEV), K.to(DEV),
                               extra_channels=None if extra is None else extra.to(DEV), **kw)  # fmt: skip
    i1["means2d"].retain_grad()
    g = torch.Generator().manual_seed(seed)
    vr, va = torch.randn(r0.shape, generator=g), torch.randn(a0.shape, generator=g)
    ((r0 * vr).sum() + (a0 * va).sum()).backward()
    ((r1 * vr.to(DEV)).sum() + (a1 * va.to(DEV)).sum()).backward()
    return dict(zip(names, ref_in)), dict(zip(names, gpu_in)), (r0, a0, i0), (r1, a1, i1)


def test_cfg1_plumbing_end_to_end():
    """BASELINE configs[0]: 1k Gaussians, 128x128, one view."""
    sc = plumbing_scene()
    ref_in, gpu_in, (r0, a0, i0), (r1, a1, i1) = _run_both(sc, 0, "RGB", 0)
    assert torch.equal(i1["radii"].cpu(), i0["radii"])
    assert torch.equal(i1["isect_ids"].cpu(), i0["isect_ids"])
    assert torch.equal(i1["flatten_ids"].cpu(), i0["flatten_ids"])
    assert torch.equal(i1["isect_offsets"].cpu(), i0["isect_offsets"])
    assert rel_err(r1, r0) < REL_TOL and rel_err(a1, a0) < REL_TOL
    assert psnr(r1, r0) > 80
    for k in ref_in:
        assert rel_l2(gpu_in[k].grad, ref_in[k].grad) < REL_TOL, k
    assert i1["means2d"].grad is not None and i1["means2d"].absgrad.shape == (1, 1000, 2)
    assert i1["radii"].shape == (1, 1000) and i1["radii"].dtype == torch.int32


@pytest.mark.parametrize("render_mode,sh_degree,rmode", [("RGB", 3, "classic"), ("RGB+ED", 2, "classic"),
                                                         ("ED", 3, "classic"), ("RGB+ED", 3, "antialiased"),
                                                         ("RGB", None, "classic")])  # fmt: skip
def test_full_pipeline_modes(render_mode, sh_degree, rmode):
    sc = _scene(n=15000, w=256, h=144, seed=9)
    if sh_degree is None:
        sc.colors = torch.sigmoid(sc.colors[:, 0, :])
    ref_in, gpu_in, (r0, a0, i0), (r1, a1, i1) = _run_both(sc, 1, render_mode, sh_degree, rmode)
    assert torch.equal(i1["isect_ids"].cpu(), i0["isect_ids"])
    assert torch.equal(i1["flatten_ids"].cpu(), i0["flatten_ids"])
    assert r1.shape == r0.shape and a1.shape == a0.shape
    assert rel_err(r1, r0) < REL_TOL and rel_err(a1, a0) < REL_TOL
    for k in ref_in:
        if ref_in[k].grad is None:
            continue
        assert rel_l2(gpu_in[k].grad, ref_in[k].grad) < REL_TOL, k


def test_packed_mode_info():
    """packed=True + render_mode='ED' as preprocess/knn_gaussian.py:93-130 uses it."""
    sc = _scene(n=5000)
    sc.means[:100] *= 5
    with torch.no_grad():
        r0, a0, i0 = O.rasterization(sc.means, sc.quats, sc.scales, sc.opacities, sc.colors, sc.viewmats[:1],
                                     sc.Ks[:1], sc.width, sc.height, sh_degree=3, packed=True, render_mode="ED")
        r1, a1, i1 = rasterization(sc.means.to(DEV), sc.quats.to(DEV), sc.scales.to(DEV), sc.opacities.to(DEV),
                                   sc.colors.to(DEV), sc.viewmats[:1].to(DEV), sc.Ks[:1].to(DEV), sc.width,
                                   sc.height, sh_degree=3, packed=True, render_mode="ED")  # fmt: skip
    assert r1.shape == (1, sc.height, sc.width, 1)
    assert torch.equal(i1["gaussian_ids"].cpu(), i0["gaussian_ids"])
    assert i1["means2d"].shape == (i0["gaussian_ids"].numel(), 2)
    assert torch.equal(i1["means2d"].cpu(), i0["means2d"]) and torch.equal(i1["depths"].cpu(), i0["depths"])
    assert rel_err(r1, r0) < REL_TOL


def test_backward_is_repeatable_within_tolerance():
    """Atomic accumulation order may change the last bits, never more."""
    sc = _scene(n=8000)
    outs = []
    for _ in range(2):
        t = [x.to(DEV).requires_grad_(True) for x in (sc.means, sc.quats, sc.scales, sc.opacities, sc.colors)]
        r, a, _ = rasterization(*t, sc.viewmats[:1].to(DEV), sc.Ks[:1].to(DEV), sc.width, sc.height, sh_degree=3,
                                packed=False, absgrad=True)  # fmt: skip
        (r.square().sum() + a.sum()).backward()
        outs.append([x.grad.clone() for x in t] + [r.detach().clone()])
    assert torch.equal(outs[0][-1], outs[1][-1])  # forward is deterministic
    for x, y in zip(outs[0][:-1], outs[1][:-1]):
        assert rel_l2(x, y) < 1e-5


# ------------------------------------------------------------------------------------------
def test_flow_kernels_vs_oracle():
    g = torch.Generator().manual_seed(2)
    N, W, H = 5000, 96, 64
    K = torch.tensor([[80.0, 0, 47.5], [0, 90.0, 31.0], [0, 0, 1]])
    veloc, omega = torch.tensor([0.02, -0.01, 0.03]), torch.tensor([0.004, 0.01, -0.006])
    m2 = (torch.rand(N, 2, generator=g) * torch.tensor([W, H])).requires_grad_(True)
    dep = (torch.rand(N, generator=g) * 3 + 0.5).requires_grad_(True)
    vel = torch.randn(N, 3, generator=g).requires_grad_(True)
    ugs0, ucam0 = O.gaussian_flow(m2, dep, vel, K[0, 0], K[1, 1], K[0, 2], K[1, 2], veloc, omega)
    v1, v2 = torch.randn(N, 2, generator=g), torch.randn(N, 2, generator=g)
    ((ugs0 * v1).sum() + (ucam0 * v2).sum()).backward()
    t = [x.detach().to(DEV).requires_grad_(True) for x in (m2, dep, vel)]
    ugs1, ucam1 = ops.gaussian_flow(*t, K.to(DEV), veloc.to(DEV), omega.to(DEV))
    ((ugs1 * v1.to(DEV)).sum() + (ucam1 * v2.to(DEV)).sum()).backward()
    assert rel_err(ugs1, ugs0) < REL_TOL and rel_err(ucam1, ucam0) < REL_TOL
    for x, y in zip(t, (m2, dep, vel)):
        assert rel_l2(x.grad, y.grad) < REL_TOL
    Z = torch.rand(H, W, generator=g) * 4 + 0.3
    Z[3, 5] = float("inf")
    f0 = O.camera_flow(Z, K[0, 0], K[1, 1], K[0, 2], K[1, 2], veloc, omega)
    f1 = ops.camera_flow(Z.to(DEV), K.to(DEV), veloc.to(DEV), omega.to(DEV))
    assert rel_err(f1, f0) < REL_TOL and float(f1[3, 5].abs().max()) == 0.0


def test_reprojection_flow_kernel_vs_oracle_and_reference_golden():
    """F-spec' on the GPU (fg_reprojection_flow through flow.reprojection_flow_map) against the oracle
    restatement at 400x300 and against the reference's own output (g_flow_bp.npz)."""
    import os

    import numpy as np

    from freegaussian_amd import flow as FL

    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g_flow_bp.npz"))
    t = lambda k: torch.from_numpy(z[k])  # noqa: E731
    fx, fy, cx, cy = t("K").tolist()
    K = torch.tensor([[fx, 0.0, cx], [0.0, fy, cy], [0.0, 0.0, 1.0]])
    for tag in ("trans", "rot"):
        out = FL.reprojection_flow_map(t("Z").to(DEV), t("Z1").to(DEV), K, t(tag + ".c2w0"), t(tag + ".c2w1"),
                                       t("opticalflow"))  # fmt: skip
        assert torch.allclose(out["sceneflow"].cpu(), t(tag + ".sceneflow"), atol=1e-5)
        assert torch.allclose(out["interflow"].cpu(), t(tag + ".interflow"), atol=1e-5)
    g = torch.Generator().manual_seed(2)
    H, W = 300, 400
    Z = torch.rand(H, W, 1, generator=g) * 4 + 0.5
    Z[10:20, 30:50] = float("inf")
    Z1 = Z + torch.randn(H, W, 1, generator=g) * 0.01
    Z1[torch.isinf(Z1)] = 3.0
    K = torch.tensor([[310.0, 0.0, 201.5], [0.0, 305.0, 148.0], [0.0, 0.0, 1.0]])
    of = torch.randn(H, W, 2, generator=g)
    ref = O.camera_flow_reprojection(Z.double(), Z1.double(), t("rot.c2w0"), t("rot.c2w1"), K.double(), of.double())
    out = FL.reprojection_flow_map(Z.to(DEV), Z1.to(DEV), K, t("rot.c2w0"), t("rot.c2w1"), of)
    assert rel_err(out["sceneflow"], ref["sceneflow"].float()) < REL_TOL
    assert rel_err(out["interflow"], ref["interflow"].float()) < REL_TOL
    assert float(out["sceneflow"][10:20, 30:50].abs().max()) == 0.0


def test_composited_flow_channels_f1():
    """F1 (Corollary 1): sum_i T_i alpha_i (mu_t - mu_0) as two extra composited channels, with
    gradients reaching both sets of Gaussian positions."""
    sc = _scene(n=6000, w=128, h=96, seed=4)
    g = torch.Generator().manual_seed(8)
    disp = torch.randn(6000, 2, generator=g)
    ref_in, gpu_in, (r0, a0, i0), (r1, a1, i1) = _run_both(sc, 0, "RGB+ED", 3, extra=disp)
    assert r1.shape[-1] == 6
    assert rel_err(r1[..., 4:], r0[..., 4:]) < REL_TOL
    for k in ref_in:
        assert rel_l2(gpu_in[k].grad, ref_in[k].grad) < REL_TOL, k


# ------------------------------------------------------------------------------------------
# H1-H4, O1, S1: the host mirror of FreeGaussianModel.get_outputs around the HIP raster
def _model_and_camera(n=4000, W=160, H=96, step=4000, training=True, log_scale=-3.2):
    import copy

    from freegaussian_amd.model import Camera, FreeGaussianModel, FreeGaussianModelConfig
    from freegaussian_amd.scenes import look_at_viewmat

    torch.manual_seed(0)
    cfg = FreeGaussianModelConfig(background_color="white", num_downscales=0, warm_up=3000)
    model = FreeGaussianModel(cfg, seed_points=(torch.rand(n, 3) - 0.5) * 2.0, init_scales=log_scale)
    with torch.no_grad():
        model.gauss_params["scales"].fill_(log_scale)
        model.gauss_params["features_rest"].normal_(0, 0.1)
        for p in model.deform.parameters():
            p.mul_(0.3)
    model.step = step
    model.train(training)
    w2c = look_at_viewmat(torch.tensor([0.3, -0.2, -3.0]), torch.zeros(3))
    c2w_cv = torch.linalg.inv(w2c)
    c2w_gl = c2w_cv.clone()
    c2w_gl[:3, 1:3] *= -1  # OpenCV -> OpenGL camera axes (get_viewmat flips them back)
    f = W / 160.0  # same field of view at every size
    cam = Camera(c2w_gl[None, :3], 140.0 * f, 150.0 * f, W / 2, H / 2, W, H, times=torch.tensor([[0.4]]))
    ref = copy.deepcopy(model)
    return model.to(DEV), ref, cam


def _oracle_outputs(ref, cam, render_mode):
    """The same host math on CPU tensors, with the CPU oracle in place of the HIP raster."""
    from freegaussian_amd.utils import from_homogenous, get_viewmat, to_homogenous

    viewmat = get_viewmat(cam.camera_to_worlds)
    K = cam.get_intrinsics_matrices()
    colors, deg = ref._colors_and_degree()
    times = cam.times.expand(ref.num_points, -1)
    d_xyz, d_rot, d_scale = ref.deform(ref.means.detach(), times)
    means = from_homogenous(torch.bmm(d_xyz, to_homogenous(ref.means).unsqueeze(-1)).squeeze(-1))
    scales = torch.exp(ref.scales) + d_scale
    quats = ref.quats / ref.quats.norm(dim=-1, keepdim=True) + d_rot
    r, a, info = O.rasterization(means, quats, scales, torch.sigmoid(ref.opacities).squeeze(-1), colors, viewmat, K,
                                 cam.width, cam.height, sh_degree=deg, render_mode=render_mode, packed=False)  # fmt: skip
    rgb = torch.clamp(r[..., :3] + (1 - a) * torch.ones(3), 0.0, 1.0)
    return rgb.squeeze(0), a.squeeze(0), r, info


def test_model_get_outputs_training_step_matches_oracle():
    model, ref, cam = _model_and_camera()
    out = model.get_outputs(cam)
    rgb0, acc0, _, info0 = _oracle_outputs(ref, cam, "RGB")
    assert out["depth"] is None and out["rgb"].shape == (cam.height, cam.width, 3)
    assert rel_err(out["rgb"], rgb0) < REL_TOL and rel_err(out["accumulation"], acc0) < REL_TOL
    assert torch.equal(model.radii.cpu(), info0["radii"][0])
    gt = torch.rand(cam.height, cam.width, 3, generator=torch.Generator().manual_seed(3))
    _smooth_loss(out["rgb"], gt.to(DEV)).backward()
    _smooth_loss(rgb0, gt).backward()
    for k in ("means", "scales", "quats", "features_dc", "features_rest", "opacities"):
        assert rel_l2(model.gauss_params[k].grad, ref.gauss_params[k].grad) < REL_TOL, k
    gd = torch.cat([p.grad.flatten() for p in model.deform.parameters()])
    gd0 = torch.cat([p.grad.flatten() for p in ref.deform.parameters()])
    assert rel_l2(gd, gd0) < REL_TOL  # (GEMM chains in different orders on CPU / GPU: measured 1e-6)
    # S1: densification statistics consume xys.absgrad / radii exactly as the reference does
    model.after_train_iter(model.step)
    vis = model.radii > 0
    assert model.xys.absgrad.shape == (1, model.num_points, 2) and model.xys.grad is not None
    assert torch.equal(model.vis_counts, 1.0 + vis.float())
    assert torch.allclose(model.xys_grad_norm[vis], model.xys.absgrad[0][vis].norm(dim=-1))
    assert torch.allclose(model.max_2Dsize[vis], model.radii[vis].float() / max(cam.width, cam.height))


def test_model_get_outputs_eval_depth_and_background():
    model, ref, cam = _model_and_camera(training=False)
    model.background_color = torch.ones(3)
    out = model.get_outputs_for_camera(cam)
    with torch.no_grad():
        rgb0, acc0, r0, _ = _oracle_outputs(ref, cam, "RGB+ED")
    d0 = torch.where(acc0 > 0, r0[0, ..., 3:4], r0[0, ..., 3:4].max())
    assert out["depth"].shape == (cam.height, cam.width, 1) and out["background"].shape == (cam.height, cam.width, 3)
    assert rel_err(out["rgb"], rgb0) < REL_TOL and rel_err(out["depth"], d0) < REL_TOL


def test_crop_box_render_equals_rendering_the_subset():
    """Eval-only crop (reference :778-798): the render of the cropped model is the render of a model
    that only holds the rows inside the box."""
    import copy

    from freegaussian_amd.model import OrientedBox

    model, _, cam = _model_and_camera(n=3000, step=4000, training=False)
    box = OrientedBox(R=torch.eye(3), T=torch.tensor([0.2, 0.0, 0.0]), S=torch.tensor([1.0, 1.4, 1.2]))
    inside = box.within(model.gauss_params["means"]).reshape(-1)
    assert 100 < int(inside.sum()) < 2900
    sub = copy.deepcopy(model)
    for k in list(sub.gauss_params.keys()):
        sub.gauss_params[k] = torch.nn.Parameter(sub.gauss_params[k][inside].clone())
    model.set_crop(box)
    with torch.no_grad():
        a, b = model.get_outputs(copy.deepcopy(cam)), sub.get_outputs(copy.deepcopy(cam))
    assert torch.equal(a["rgb"], b["rgb"]) and torch.equal(a["depth"], b["depth"])
    assert model.radii.shape[0] == int(inside.sum())
    model.train()
    out = model.get_outputs(copy.deepcopy(cam))  # training ignores the box
    assert model.radii.shape[0] == 3000 and out["rgb"].shape == (cam.height, cam.width, 3)


def test_render_and_ground_truth_shapes_agree_at_every_scheduled_resolution():
    """1352x1014 at d = 4: 1014 / 4 = 253.5 -- render and resize_image target must both be 253 rows."""
    model, _, _ = _model_and_camera(n=500, W=1352, H=1014, step=100, training=True)
    from freegaussian_amd.model import Camera

    model.config.num_downscales = 2
    cam = Camera(torch.eye(4)[None, :3], 900.0, 900.0, 676.0, 507.0, 1352, 1014, times=torch.tensor([[0.1]]))
    img = torch.rand(1014, 1352, 3)
    for model.step in (100, 3100, 6100):
        out = model.get_outputs(cam)
        assert out["rgb"].shape == model.get_gt_img(img).shape, model.step
        assert (cam.width, cam.height) == (1352, 1014)


def test_control_model_stage2_scatter_and_render():
    from freegaussian_amd.model import FreeGaussianControlModel, FreeGaussianModelConfig

    model, ref, cam = _model_and_camera(n=3000, training=False)
    mask = torch.zeros(3000, 3, dtype=torch.bool)
    mask[:60, 0] = True
    mask[40:100, 1] = True
    mask[200:230, 2] = True
    init_cam = type(cam)(cam.camera_to_worlds, cam.fx, cam.fy, cam.cx, cam.cy, cam.width, cam.height,
                         times=torch.tensor([[0.0]]))  # fmt: skip
    cm = FreeGaussianControlModel(mask, init_cam, config=FreeGaussianModelConfig(background_color="black"),
                                  seed_points=ref.means.detach().clone())  # fmt: skip
    cm.load_state_dict(ref.state_dict(), strict=False)
    cm = cm.to(DEV).eval()
    cam.metadata["cameras0"] = init_cam
    with torch.no_grad():
        out = cm.get_outputs(cam)
    assert out["rgb"].shape == (cam.height, cam.width, 3) and bool(torch.isfinite(out["rgb"]).all())
    assert float(out["accumulation"].max()) > 0.1 and "deform" not in cm.get_param_groups()


# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("render_mode,sh_degree,rmode,extra", [("RGB", 3, "classic", False), ("RGB+ED", 1, "antialiased", True),
                                                               ("ED", 3, "classic", False), ("RGB", None, "classic", True),
                                                               ("RGB+ED", 0, "classic", False)])  # fmt: skip
def test_fused_path_equals_stagewise_path(render_mode, sh_degree, rmode, extra):
    """fg_preprocess_fwd/bwd (fused) vs project + SH + pack / unpack + SH bwd + project bwd."""
    sc = _scene(n=12000, w=208, h=130, seed=21)
    sc.means[:40] *= 4.0
    colors = sc.colors if sh_degree is not None else torch.sigmoid(sc.colors[:, 0, :])
    ex = torch.randn(12000, 2, generator=torch.Generator().manual_seed(2)) if extra else None
    outs = []
    for fused in (True, False):
        t = [x.to(DEV).requires_grad_(True) for x in (sc.means, sc.quats, sc.scales, sc.opacities, colors)]
        e = None if ex is None else ex.to(DEV).requires_grad_(True)
        r, a, info = rasterization(*t, sc.viewmats[:1].to(DEV), sc.Ks[:1].to(DEV), sc.width, sc.height,
                                   sh_degree=sh_degree, render_mode=render_mode, packed=False, absgrad=True,
                                   rasterize_mode=rmode, extra_channels=e, fused=fused)  # fmt: skip
        info["means2d"].retain_grad()
        g = torch.Generator().manual_seed(4)
        vr, va = torch.randn(r.shape, generator=g).to(DEV), torch.randn(a.shape, generator=g).to(DEV)
        ((r * vr).sum() + (a * va).sum()).backward()
        outs.append((r.detach(), a.detach(), info, [x.grad for x in t] + ([e.grad] if e is not None else []),
                     info["means2d"].grad, info["means2d"].absgrad))
    (r0, a0, i0, g0, m0, ab0), (r1, a1, i1, g1, m1, ab1) = outs
    # identical forward: same records, same lists, same kernels
    assert torch.equal(r0, r1) and torch.equal(a0, a1)
    for k in ("radii", "means2d", "depths", "conics", "flatten_ids", "isect_offsets", "tiles_per_gauss"):
        assert torch.equal(i0[k], i1[k]), k
    for x, y in zip(g0, g1):
        assert (x is None) == (y is None)  # e.g. colours are unused in "ED" mode
        if x is not None:
            assert rel_l2(x, y) < 1e-5
    assert rel_l2(m0, m1) < 1e-5 and rel_l2(ab0, ab1) < 1e-5


@pytest.mark.parametrize("sh_degree", [1, 2, 3])
@pytest.mark.parametrize("raw", [False, True])
def test_backward_from_the_forwards_sh_note_equals_the_backward_from_the_coefficient_rows(sh_degree, raw):
    """fg_preprocess_fwd leaves d colour / d direction and the clamp mask (40 B per Gaussian, `sh_jac`) so that
    fg_preprocess_bwd need not read the 192-byte coefficient rows again: same images, same gradients as the
    backward that recomputes both from the rows (FG_SH_JAC=0 / RasterContext.sh_jacobian = False)."""
    sc = _scene(n=12001, w=176, h=128, seed=23)  # (an odd count: the last workgroup's slab takes the scalar path)
    vm, K = sc.viewmats[:1].to(DEV), sc.Ks[:1].to(DEV)
    vr = torch.randn(1, sc.height, sc.width, 3, generator=torch.Generator().manual_seed(3)).to(DEV)
    outs = []
    for note in (True, False):
        ctx = ops.RasterContext()
        ctx.sh_jacobian = note
        if raw:
            from freegaussian_amd.rasterization import rasterize_gauss_params

            t = [x.to(DEV).requires_grad_(True) for x in (sc.means, sc.quats, sc.scales.log(), torch.logit(sc.opacities.clamp(1e-4, 1 - 1e-4)),
                                                          sc.colors[:, 0].contiguous(), sc.colors[:, 1:].contiguous())]  # fmt: skip
            r = rasterize_gauss_params(*t, vm, K, sc.width, sc.height, sh_degree=sh_degree, ctx=ctx)[0]
        else:
            t = [x.to(DEV).requires_grad_(True) for x in (sc.means, sc.quats, sc.scales, sc.opacities, sc.colors)]
            r = rasterization(*t, vm, K, sc.width, sc.height, sh_degree=sh_degree, packed=False, absgrad=True, ctx=ctx)[0]
        (r.reshape(vr.shape) * vr).sum().backward()
        outs.append((r.detach(), [x.grad for x in t]))
    (r0, g0), (r1, g1) = outs
    assert torch.equal(r0, r1)
    for x, y in zip(g0, g1):
        assert rel_l2(x, y) < 2e-6  # (the order of fp32 sums differs: sum over bases first / over channels first)
    assert float(g0[0].abs().max()) > 0 and float(g0[-1].abs().max()) > 0


def test_direct_grad_buffers_fill_the_flat_gradient():
    """viewdp.FlatGaussianParams.direct_grads(): the fused backward writes straight into the flat
    all-reduce buffer (no AccumulateGrad add, no zeroing) and matches ordinary autograd."""
    from freegaussian_amd.viewdp import FlatGaussianParams

    sc = _scene(n=9000, w=160, h=112, seed=6)
    vm, K = sc.viewmats[:1].to(DEV), sc.Ks[:1].to(DEV)
    vr = torch.randn(1, sc.height, sc.width, 3, generator=torch.Generator().manual_seed(1)).to(DEV)
    ref = [getattr(sc, n).to(DEV).requires_grad_(True) for n in ("means", "quats", "scales", "opacities", "colors")]
    r, _, _ = rasterization(*ref, vm, K, sc.width, sc.height, sh_degree=3, packed=False, absgrad=True)
    (r * vr).sum().backward()
    fp = FlatGaussianParams.from_scene(sc, DEV)
    fp.flat_grad.fill_(float("nan"))  # must be fully overwritten
    with fp.direct_grads():
        r2, _, _ = rasterization(*fp.raster_inputs(), vm, K, sc.width, sc.height, sh_degree=3, packed=False,
                                 absgrad=True)  # fmt: skip
        (r2 * vr).sum().backward()
    assert torch.equal(r, r2) and bool(torch.isfinite(fp.flat_grad).all())
    for p, q in zip(fp.raster_inputs(), ref):
        assert p.grad.data_ptr() >= fp.flat_grad.data_ptr()
        assert p.grad.data_ptr() < fp.flat_grad.data_ptr() + fp.flat_grad.numel() * 4
        assert rel_l2(p.grad, q.grad) < 1e-5


# ------------------------------------------------------------------------------------------
def test_full_size_cfg4_properties_and_oracle_crop():
    """BASELINE.json's full size (1M Gaussians, 1920x1080): size-independent properties of the
    whole path, plus a tile-aligned centre crop checked pixel by pixel against the C oracle."""
    from freegaussian_amd.scenes import north_star_scene
    from oracle import c_oracle as CO

    sc = north_star_scene(n_views=1)
    W, H = sc.width, sc.height
    t = [x.to(DEV).requires_grad_(True) for x in (sc.means, sc.quats, sc.scales, sc.opacities, sc.colors)]
    vm, K = sc.viewmats[:1].to(DEV), sc.Ks[:1].to(DEV)
    r, a, info = rasterization(*t, vm, K, W, H, sh_degree=3, packed=False, absgrad=True)
    keys, ids, offs = info["isect_ids"], info["flatten_ids"], info["isect_offsets"]
    I, T = ids.numel(), info["tile_width"] * info["tile_height"]
    # integer path: totals, sortedness, stability, ranges
    assert int(info["tiles_per_gauss"].sum()) == I and int(offs[-1]) == I and int(offs[0]) == 0
    assert bool((keys[1:] >= keys[:-1]).all())
    same = keys[1:] == keys[:-1]
    assert bool((ids[1:][same] > ids[:-1][same]).all())
    tile_of = (keys >> 32).to(torch.int32)
    assert torch.equal(offs, torch.searchsorted(tile_of, torch.arange(T + 1, device=DEV, dtype=torch.int32)).int())
    assert bool((info["radii"][0][ids.long()] > 0).all())
    # image: alpha in [0,1], finite, deterministic forward
    assert bool(torch.isfinite(r).all()) and float(a.detach().min()) >= 0.0 and float(a.detach().max()) <= 1.0
    r2, a2, _ = rasterization(*[x.detach() for x in t], vm, K, W, H, sh_degree=3, packed=False)
    assert torch.equal(r2, r.detach()) and torch.equal(a2, a.detach())
    # backward is linear in the upstream gradient
    g = torch.Generator().manual_seed(0)
    v1, v2 = torch.randn(r.shape, generator=g).to(DEV), torch.randn(r.shape, generator=g).to(DEV)
    grads = []
    for v in (v1, v2, v1 + v2):
        grads.append(torch.autograd.grad(r, t, v, retain_graph=True))
    for g1, g2, g12 in zip(*grads):
        assert rel_l2(g1 + g2, g12) < 1e-4
    assert all(bool(torch.isfinite(x).all()) for x in grads[2])
    # centre crop (tile aligned) against the scalar C oracle, fed with the GPU's own projection
    # (bit-exact vs the oracle, test_project_bit_exact) and the GPU's sorted lists of those tiles
    # (validated just above) -- so every pixel sees exactly the same splats in the same order
    cw, ch = 160, 96
    x0, y0 = (W - cw) // 2 // 16 * 16, (H - ch) // 2 // 16 * 16
    tw = info["tile_width"]
    offs_c, ids_c = offs.cpu(), ids.cpu()
    lists, coffs = [], [0]
    for ty in range(ch // 16):
        for tx in range(cw // 16):
            tt = (y0 // 16 + ty) * tw + (x0 // 16 + tx)
            lists.append(ids_c[int(offs_c[tt]) : int(offs_c[tt + 1])])
            coffs.append(coffs[-1] + lists[-1].numel())
    cv, coffs = torch.cat(lists), torch.tensor(coffs, dtype=torch.int32)
    m2 = info["means2d"][0].detach().cpu() - torch.tensor([float(x0), float(y0)])
    campos = torch.linalg.inv(sc.viewmats[0])[:3, 3]
    rgb = torch.clamp_min(O.sh_eval(3, sc.means - campos, sc.colors) + 0.5, 0.0)
    rc, ac, _ = CO.raster_fwd(m2, info["conics"][0].cpu(), rgb, sc.opacities, cw, ch, 16, coffs, cv)
    assert close_except_knife_edge(r[0, y0 : y0 + ch, x0 : x0 + cw], rc, REL_TOL)
    assert close_except_knife_edge(a[0, y0 : y0 + ch, x0 : x0 + cw], ac, REL_TOL)



def test_six_million_gaussians_at_1440p_properties():
    """Beyond BASELINE.json's sizes (a card with 288 GB holds scenes of several million Gaussians): 6M Gaussians at
    2560 x 1440 -- 14 400 tiles, a list of tens of millions of entries -- through the whole path, twice (the first call sizes
    the list exactly, the second runs on the speculative capacity and the learned launch policy), checked through the
    size-independent properties of the integer path and of both passes."""
    sc = synthetic_scene(6_000_000, 2560, 1440, n_views=1, sh_degree=3, seed=9, log_scale_mean=math.log(0.007))
    W, H = sc.width, sc.height
    t = [x.to(DEV).requires_grad_(True) for x in (sc.means, sc.quats, sc.scales, sc.opacities, sc.colors)]
    vm, K = sc.viewmats[:1].to(DEV), sc.Ks[:1].to(DEV)
    ctx = ops.RasterContext()
    with ops.use(ctx):
        outs = []
        for _ in range(2):
            r, a, info = rasterization(*t, vm, K, W, H, sh_degree=3, packed=False, absgrad=True)
            outs.append((r.detach().clone(), a.detach().clone()))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        keys, ids, offs = info["isect_ids"], info["flatten_ids"], info["isect_offsets"]
        I, T = ids.numel(), info["tile_width"] * info["tile_height"]
        assert T == 14_400 and I > 20_000_000
        assert int(info["tiles_per_gauss"].sum()) == I and int(offs[-1]) == I and int(offs[0]) == 0
        assert bool((keys[1:] >= keys[:-1]).all())
        same = keys[1:] == keys[:-1]
        assert bool((ids[1:][same] > ids[:-1][same]).all())
        tile_of = (keys >> 32).to(torch.int32)
        assert torch.equal(offs, torch.searchsorted(tile_of, torch.arange(T + 1, device=DEV, dtype=torch.int32)).int())
        del keys, same, tile_of
        assert bool((info["radii"][0][ids.long()] > 0).all())
        assert bool(torch.isfinite(r).all()) and float(a.detach().min()) >= 0.0 and float(a.detach().max()) <= 1.0
        assert float(a.detach().mean()) > 0.5
        g = torch.Generator().manual_seed(0)
        v1, v2 = torch.randn(r.shape, generator=g).to(DEV), torch.randn(r.shape, generator=g).to(DEV)
        grads = [torch.autograd.grad(r, t, v, retain_graph=True) for v in (v1, v2, v1 + v2)]
        for g1, g2, g12 in zip(*grads):
            assert rel_l2(g1 + g2, g12) < 1e-4
        assert all(bool(torch.isfinite(x).all()) for x in grads[2])
        # Gaussians outside every list get no gradient at all
        seen = torch.zeros(sc.means.shape[0], dtype=torch.bool, device=DEV)
        seen[ids.long()] = True
        assert bool((grads[2][0][~seen] == 0).all()) and bool((grads[2][4][~seen] == 0).all())
    ctx.release_workspaces()


def test_forty_eight_million_gaussians_pass_the_32_bit_element_counts():
    """48M Gaussians (x 48 SH floats = 2.3e9 elements: every flat index of the coefficient rows and their gradients is
    beyond 32 bits for the last fifteenth of the set), a list of 108M entries, 53 GiB at the peak: scripts/big_scene_check.py
    -- integer-path properties, gradients exactly zero outside the lists and present inside them for the highest ids, and the
    top sixteenth of the ids rendered alone equal, bit for bit, to the full call with everyone else's opacity at zero.
    (The same script at 100M Gaussians / 3840 x 2160 / 276M entries / 116 GiB: profiles/r06_big_scenes.txt.)"""
    if torch.cuda.get_device_properties(0).total_memory < 120 * 2**30:
        pytest.skip("needs ~60 GiB of device memory")
    from scripts.big_scene_check import check

    check(48.0, 1920, 1080, 0.003)
    torch.cuda.empty_cache()


# ------------------------------------------------------------------------------------------
def _oracle_full_res(sc, view, render_mode, sh_degree, seed=0, with_alpha_grad=True):
    """O.rasterization with the scalar C compositing (oracle/c_oracle.composite; it agrees with the
    torch compositing to 1e-6, tests/test_oracle.py) -- affordable at 8160 tiles -- and the HIP path
    on the same inputs and the same upstream gradients."""
    from oracle import c_oracle as CO

    names = ["means", "quats", "scales", "opacities", "colors"]
    cpu = [getattr(sc, n) for n in names]
    ref_in = [t.clone().requires_grad_(True) for t in cpu]
    gpu_in = [t.to(DEV).requires_grad_(True) for t in cpu]
    kw = dict(width=sc.width, height=sc.height, sh_degree=sh_degree, render_mode=render_mode, packed=False, absgrad=True)
    vm, K = sc.viewmats[view : view + 1], sc.Ks[view : view + 1]
    r0, a0, i0 = O.rasterization(*ref_in, vm, K, compositor=CO.composite, **kw)
    r1, a1, i1 = rasterization(*gpu_in, vm.to(DEV), K.to(DEV), **kw)
    i0["means2d"].retain_grad()
    i1["means2d"].retain_grad()
    g = torch.Generator().manual_seed(seed)
    vr, va = torch.randn(r0.shape, generator=g), torch.randn(a0.shape, generator=g)
    if not with_alpha_grad:
        va = torch.zeros_like(va)
    ((r0 * vr).sum() + (a0 * va).sum()).backward()
    ((r1 * vr.to(DEV)).sum() + (a1 * va.to(DEV)).sum()).backward()
    return dict(zip(names, ref_in)), dict(zip(names, gpu_in)), (r0, a0, i0), (r1, a1, i1)


def _assert_full_parity(ref_in, gpu_in, o0, o1, tol):
    (r0, a0, i0), (r1, a1, i1) = o0, o1
    assert torch.equal(i1["radii"].cpu(), i0["radii"])
    assert torch.equal(i1["flatten_ids"].cpu(), i0["flatten_ids"])
    assert torch.equal(i1["isect_offsets"].cpu().reshape(-1), i0["isect_offsets"].reshape(-1))
    assert close_except_knife_edge(r1[0], r0[0], REL_TOL) and close_except_knife_edge(a1[0], a0[0], REL_TOL)
    worst = {}
    for k in ref_in:
        worst[k] = rel_l2(gpu_in[k].grad, ref_in[k].grad)
    worst["means2d"] = rel_l2(i1["means2d"].grad, i0["means2d"].grad)
    worst["absgrad"] = rel_l2(i1["means2d"].absgrad, i0["means2d"].absgrad)
    assert all(v < tol for v in worst.values()), worst
    return worst


@pytest.mark.parametrize("layout,render_mode", [("uniform", "RGB"), ("clustered", "RGB"), ("uniform", "RGB+ED")])
def test_1080p_mixed_launch_forward_and_all_gradients_vs_oracle(layout, render_mode):
    """8160 tiles: the launch shape of the headline -- `raster_fwd/bwd_mixed_kernel` with job lists,
    whole-tile jobs (4 pixels per lane), the positional two-strip / single-strip tails and, on the
    clustered scene, content-split jobs -- put DIRECTLY against the oracle: image, every parameter
    gradient, the screen-space gradient and absgrad."""
    from freegaussian_amd import _lib

    if int(_lib.load().fg_raster_jobs_words(1920, 1080, 16, ops.default_context.cfg())) == 0:  # job-list launches are what runs here by default
        pytest.skip("classic launches forced by the environment (FG_RASTER_PPT_* / FG_TILE_ORDER)")
    sc = synthetic_scene(40_000, 1920, 1080, n_views=2, sh_degree=3, seed=5, log_scale_mean=math.log(0.03))
    if layout == "clustered":
        sc.means[:20_000] *= 0.15  # a ball at the centre: lists there are many times the mean
    ref_in, gpu_in, o0, o1 = _oracle_full_res(sc, 1, render_mode, 3)
    if layout == "clustered":
        offs = o0[2]["isect_offsets"].reshape(-1)
        lens = torch.diff(offs)
        assert int(lens.max()) > 20 * int(offs[-1]) // 65536  # some tiles are above the four-strip threshold
    _assert_full_parity(ref_in, gpu_in, o0, o1, REL_TOL)


@pytest.mark.parametrize("parts,layout,bg", [(2, "uniform", False), (4, "clustered", False), (3, "uniform", True), (16, "clustered", True)])
def test_segmented_backward_vs_oracle_and_vs_whole_list_walk(parts, layout, bg, monkeypatch):
    """List segments of the backward (forward checkpoints every 64 list entries; `parts` jobs per
    tile, each over its share of the list): every gradient against the oracle at 8160 tiles and against
    the unsegmented walk; with and without the composite epilogue (background + clamp), whose raw
    colours the backward has to rebuild from the finished image.  A share job resumes from the forward's
    exact state and applies the reference's per-pixel rounding factor of T_final
    (profiles/r02_backward_list_shares.md section 5): without it the two walks differ by ~8e-5."""
    from freegaussian_amd.rasterization import rasterize_gauss_params

    sc = synthetic_scene(40_000, 1920, 1080, n_views=2, sh_degree=3, seed=5, log_scale_mean=math.log(0.03))
    if layout == "clustered":
        sc.means[:20_000] *= 0.15  # lists of thousands of entries: many segments per tile
    if not bg:
        _setenv_policy(monkeypatch, "FG_RASTER_SEG_PARTS", str(parts))
        ref_in, gpu_in, o0, o1 = _oracle_full_res(sc, 1, "RGB", 3)
        worst = _assert_full_parity(ref_in, gpu_in, o0, o1, REL_TOL)
        seg_grads = {k: v.grad.clone() for k, v in gpu_in.items()}
        _setenv_policy(monkeypatch, "FG_RASTER_SEG_PARTS", "1")
        _, gpu_in1, _, o2 = _oracle_full_res(sc, 1, "RGB", 3)
        # the forward does not depend on it -- but for heavy tiles (lists beyond ops.RasterContext.heavy_tile_len: on the
        # clustered scene), which exist only with list shares and associate the same sums and products differently
        assert torch.equal(o2[0], o1[0]) or (layout == "clustered" and rel_err(o2[0], o1[0]) < 2e-6)
        diff = {k: rel_l2(seg_grads[k], gpu_in1[k].grad) for k in seg_grads}
        print("share jobs vs whole-list walk, rel-L2:", {k: f"{v:.1e}" for k, v in diff.items()})
        assert all(v < 1e-5 for v in diff.values()), diff  # observed 2-5e-7: atomic order only
        return
    # composite epilogue: the model's raw-parameter front end with a background and the clamp
    g = torch.Generator().manual_seed(3)
    raw = dict(means=sc.means, quats=sc.quats, log_scales=sc.scales.log(), opacity_logits=torch.logit(sc.opacities.clamp(1e-4, 1 - 1e-4)),
               features_dc=sc.colors[:, 0, :].contiguous(), features_rest=sc.colors[:, 1:, :].contiguous())  # fmt: skip
    vm, K = sc.viewmats[1:2].to(DEV), sc.Ks[1:2].to(DEV)
    vr = torch.randn(1, 1080, 1920, 3, generator=g).to(DEV)
    bgc = torch.tensor([0.3, 0.9, 0.1], device=DEV)
    outs = []
    for p_ in (parts, 1):
        _setenv_policy(monkeypatch, "FG_RASTER_SEG_PARTS", str(p_))
        t = {k: v.to(DEV).requires_grad_(True) for k, v in raw.items()}
        r, a, _ = rasterize_gauss_params(t["means"], t["quats"], t["log_scales"], t["opacity_logits"], t["features_dc"],
                                         t["features_rest"], vm, K, 1920, 1080, 3, background=bgc, clamp=True, absgrad=True)  # fmt: skip
        ((r * vr).sum() + a.sum()).backward()
        outs.append((r.detach(), {k: v.grad for k, v in t.items()}))
    assert torch.equal(outs[0][0], outs[1][0]) or (layout == "clustered" and rel_err(outs[0][0], outs[1][0]) < 2e-6)  # (heavy tiles: as above)
    diff = {k: rel_l2(outs[0][1][k], outs[1][1][k]) for k in raw}
    print("share jobs vs whole-list walk (composite epilogue), rel-L2:", {k: f"{v:.1e}" for k, v in diff.items()})
    assert all(v < 1e-5 for v in diff.values()), diff


@pytest.mark.parametrize("step_calls", [True, False])
def test_compact_checkpoint_slots_same_gradients_a_fraction_of_the_buffer(step_calls):
    """Compact checkpoint slots (fg_raster_config::seg_slots): the list build reports what the tiles the
    backward may split need, the host sizes the NEXT call's buffer by that -- slots for those tiles only, first slot per
    tile in the job lists' table -- instead of a slot per 64 entries of the list's capacity.  Same image, same gradients
    (float atomics' order apart) as with the full buffer, through the one-call-per-direction path and the stage-wise
    calls; with far too few slots (a policy of 64) the tiles without run unsplit: still the same gradients."""
    from freegaussian_amd.rasterization import rasterize_gauss_params

    sc = synthetic_scene(150_000, 1920, 1080, n_views=1, sh_degree=3, seed=11, log_scale_mean=math.log(0.02))
    raw = dict(means=sc.means, quats=sc.quats, log_scales=sc.scales.log(), opacity_logits=torch.logit(sc.opacities.clamp(1e-4, 1 - 1e-4)),
               features_dc=sc.colors[:, 0, :].contiguous(), features_rest=sc.colors[:, 1:, :].contiguous())  # fmt: skip
    vm, K = sc.viewmats[:1].to(DEV), sc.Ks[:1].to(DEV)
    vr = torch.randn(1, 1080, 1920, 3, generator=torch.Generator().manual_seed(3)).to(DEV)

    def run(ctx, calls=3):
        out = None
        for _ in range(calls):
            t = {k: v.to(DEV).requires_grad_(True) for k, v in raw.items()}
            with ops.use(ctx):
                r, a, info = rasterize_gauss_params(t["means"], t["quats"], t["log_scales"], t["opacity_logits"], t["features_dc"],
                                                    t["features_rest"], vm, K, 1920, 1080, 3, absgrad=True)  # fmt: skip
                ((r * vr).sum() + a.sum()).backward()
            out = (r.detach(), {k: v.grad for k, v in t.items()}, int(info["raster_flatten_ids"].numel()))
        return out

    full = ops.RasterContext(env={"FG_COMPACT_SLOTS": "0"})
    full.step_calls = step_calls
    r_full, g_full, n_list = run(full)
    assert full.last_seg_slots == 0  # (the needs are reported and noted all the same)
    compact = ops.RasterContext(env={})
    compact.step_calls = step_calls
    r_c, g_c, _ = run(compact, calls=4)
    lib = _lib.load()
    (need,) = [max(nd for nd, _n in v) for v in compact.ckpt_need.values()]
    slots = compact.last_seg_slots
    assert 0 < 8 * need <= slots < 0.9 * (n_list // 64 + 8160), (need, slots, n_list)
    b_full = 4 * lib.fg_raster_seg_ckpt_floats(3, 1920, 1080, 16, n_list, full.cfg())
    b_compact = 4 * lib.fg_raster_seg_ckpt_floats(3, 1920, 1080, 16, n_list, compact.cfg(False, slots))
    print(f"checkpoint buffer: {b_full / 1e6:.0f} MB -> {b_compact / 1e6:.0f} MB ({slots} slots, need {need} per XCD band)")
    assert b_compact < 0.7 * b_full
    assert torch.equal(r_c, r_full)
    diff = {k: rel_l2(g_c[k], g_full[k]) for k in g_full}
    assert all(v < 1e-5 for v in diff.values()), diff
    starved = ops.RasterContext(env={}, policy=ops.launch_policy(seg_slots=64))
    starved.step_calls = step_calls
    r_s, g_s, _ = run(starved)
    assert torch.equal(r_s, r_full)
    diff = {k: rel_l2(g_s[k], g_full[k]) for k in g_full}
    assert all(v < 1e-5 for v in diff.values()), diff


def test_launch_policies_the_host_picks_per_shape_change_no_result():
    """Policies that depend on what a shape's earlier calls reported: equal numbers of tiles per XCD without the
    cost pass (`balance_bands = 2`, after eight calls without a long tile list), issue priorities for the longest jobs
    (`prio_fwd` / `prio_bwd`), equal ROW bands (`balance_bands = 0`, rounds 1-3), the finer content thresholds of shapes that
    showed an uneven scene (round 5).  Same image bit for bit, same gradients up to the order of float atomics, whichever
    is on."""
    from freegaussian_amd.rasterization import rasterize_gauss_params

    sc = synthetic_scene(120_000, 1920, 1080, n_views=1, sh_degree=3, seed=13, log_scale_mean=math.log(0.02))
    raw = dict(means=sc.means, quats=sc.quats, log_scales=sc.scales.log(), opacity_logits=torch.logit(sc.opacities.clamp(1e-4, 1 - 1e-4)),
               features_dc=sc.colors[:, 0, :].contiguous(), features_rest=sc.colors[:, 1:, :].contiguous())  # fmt: skip
    vm, K = sc.viewmats[:1].to(DEV), sc.Ks[:1].to(DEV)
    vr = torch.randn(1, 1080, 1920, 3, generator=torch.Generator().manual_seed(3)).to(DEV)

    def run(ctx, calls):
        out = None
        for _ in range(calls):
            t = {k: v.to(DEV).requires_grad_(True) for k, v in raw.items()}
            with ops.use(ctx):
                r, a, _ = rasterize_gauss_params(t["means"], t["quats"], t["log_scales"], t["opacity_logits"], t["features_dc"],
                                                 t["features_rest"], vm, K, 1920, 1080, 3, absgrad=True)  # fmt: skip
                ((r * vr).sum() + a.sum()).backward()
            out = (r.detach(), {k: v.grad for k, v in t.items()})
        return out

    base = ops.RasterContext(env={"FG_EVEN_BANDS": "0"}, policy=ops.launch_policy(prio_fwd=0, prio_bwd=0))
    r0, g0 = run(base, 2)
    even = ops.RasterContext(env={})
    r1, g1 = run(even, 11)
    (lkey,) = list(even.even_calls)
    assert even.even_calls[lkey] >= 8 and even.even_shape(lkey) and not base.even_shape(lkey)
    rows = ops.RasterContext(env={}, policy=ops.launch_policy(balance_bands=0))
    r2, g2 = run(rows, 2)
    # round 5: a shape that showed an uneven scene lately gets finer content thresholds (the jobs change, no pixel's walk does)
    fine = ops.RasterContext(env={})
    run(fine, 1)
    fine.uneven_left[lkey] = 8
    r3, g3 = run(fine, 2)
    assert fine.uneven_shape(lkey) and not even.uneven_shape(lkey)
    for r, g in ((r1, g1), (r2, g2), (r3, g3)):
        assert torch.equal(r, r0)
        diff = {k: rel_l2(g[k], g0[k]) for k in g0}
        assert all(v < 1e-5 for v in diff.values()), diff


def test_clustered_1m_scene_lists_equal_the_oracles_through_the_long_segment_sort(monkeypatch):
    """80% of a million Gaussians in a ball of extent 0.2 at 1920 x 1080 (scripts/clustered_check.py: supertile
    segments of 16 000 ... 158 000 elements, the longest tile list 111 000 entries): csrc/stbin.hip with the long
    segments' sample sort, fed the REFERENCE's rectangles (the radius boxes of the oracle's projection), against the
    oracle's (tile | depth bits)-sorted lists: `torch.equal` ids and ranges."""
    sc = synthetic_scene(1_000_000, 1920, 1080, n_views=1, sh_degree=3, seed=42)
    sc.means[:800_000] *= 0.1
    W, H = sc.width, sc.height
    tw, th = (W + 15) // 16, (H + 15) // 16
    ref = O.project(sc.means, sc.quats, sc.scales, sc.viewmats[0], sc.Ks[0], W, H)
    _, keys_s, vals_s = O.isect_tiles(ref.means2d, ref.radii, ref.depths, 16, tw, th, sort=True)
    offs_ref = O.isect_offsets(keys_s, tw * th)
    x0, y0, x1, y1 = O.tile_rects(ref.means2d, ref.radii, 16, tw, th)
    rects = _pack_rects(x0, y0, x1 - x0, y1 - y0).to(DEV)
    keys = ref.depths.float().contiguous().view(torch.int32).clone()
    keys[ref.radii <= 0] = -1
    N = sc.means.shape[0]
    z = torch.zeros(N, device=DEV)
    args = (torch.zeros(N, 2, device=DEV), z.int(), z, z.int(), 16, tw, th)
    ctx = ops.RasterContext()
    if ctx.binning != "supertile":
        pytest.skip("the environment selects another binning path")
    ctx.long_segments = "always"
    with ops.use(ctx):
        for k in range(3):  # exact capacity, then speculative, then with buckets "outgrowing" their slabs from 1600 elements
            # (round 6: count + scatter in one pass into per-bucket slabs; a bucket beyond its slab sends its whole segment
            # through global memory in the bucket sort's last workgroup -- FG_STBIN_TEST_SMALL_SLABS makes two of five do so)
            ctx.test_small_slabs = k == 2
            _, ids, offs = ops.bin_tiles(*args, keys_rects=(keys.to(DEV), rects), want_keys=False)
            assert torch.equal(offs.cpu(), offs_ref)
            assert torch.equal(ids.cpu(), vals_s), k
    lens = torch.diff(offs_ref)
    assert int(lens.max()) > 100_000 and ctx.long_calls == 3


@pytest.mark.parametrize("form", ["wide", "three"])
@pytest.mark.parametrize("bg", [False, True])
def test_heavy_tiles_forward_over_list_shares_vs_oracle_and_vs_the_serial_walk(bg, form, monkeypatch):
    """fg_raster_config::heavy_tiles: a tile with a list beyond the threshold is walked serially for a prefix only; the
    strips still open there go on as WIDE jobs (``form`` "wide", fg_raster_config::heavy_wide: one 16-wavefront workgroup
    per strip, rounds of 16 batches composited by themselves, folded in LDS, the batches in which a pixel may stop walked
    again) or, round 4's form ("three"), as local jobs over shares of the list + one combine job per strip in two more
    launches; the backward gives such a tile up to 64 shares.  Two clusters -- an opaque one (pixels saturate within a
    few hundred entries: inside the prefix, or in walked batches) and a faint one (lists of thousands that never saturate:
    every batch taken whole, many rounds) -- against the oracle (lists, image, every gradient at the bar) and against the
    serial walk (1e-6; last_ids equal but for knife-edge pixels)."""
    from freegaussian_amd.rasterization import rasterize_gauss_params

    if form == "three":
        _setenv_policy(monkeypatch, "FG_RASTER_HEAVY_WIDE", "0")

    sc = synthetic_scene(60_000, 1920, 1080, n_views=2, sh_degree=3, seed=5, log_scale_mean=math.log(0.03))
    sc.means[:20_000] = sc.means[:20_000] * 0.1 + torch.tensor([-0.9, 0.3, 0.0])  # opaque cluster
    sc.means[20_000:40_000] = sc.means[20_000:40_000] * 0.1 + torch.tensor([0.8, -0.2, 0.0])  # faint cluster
    sc.opacities[20_000:40_000] *= 0.04
    ctx = ops.default_context
    if int(_lib.load().fg_raster_jobs_words(1920, 1080, 16, ctx.cfg())) == 0 or ctx.seg_ckpt_budget_bytes <= 0:
        pytest.skip("classic launches / no list shares in this environment")
    monkeypatch.setattr(ctx, "heavy_tile_len", 1024)
    # (what earlier tests learned about this shape -- a list capacity of tens of millions of entries puts the checkpoint
    # buffer beyond the context's budget, and without list shares there are no heavy tiles)
    for name in ("isect_capacity", "isect_recent", "ckpt_need", "ckpt_pending"):
        monkeypatch.setattr(ctx, name, {})
    outs = {}
    for mode in ("always", "never"):
        monkeypatch.setattr(ctx, "heavy_tiles", mode)
        calls = ctx.heavy_calls
        if not bg:
            ref_in, gpu_in, o0, o1 = _oracle_full_res(sc, 1, "RGB", 3)
            lens = torch.diff(o1[2]["raster_isect_offsets"].reshape(-1))
            assert int(lens.max()) > 4096 and int((lens > 1024).sum()) >= 8
            _assert_full_parity(ref_in, gpu_in, o0, o1, REL_TOL)
            outs[mode] = (o1[0].detach(), o1[1].detach(), o1[2]["last_ids"], {k: v.grad for k, v in gpu_in.items()})
        else:  # the composite epilogue in the combine jobs, its prologue in the shares of the backward
            raw = dict(means=sc.means, quats=sc.quats, log_scales=sc.scales.log(),
                       opacity_logits=torch.logit(sc.opacities.clamp(1e-4, 1 - 1e-4)),
                       features_dc=sc.colors[:, 0, :].contiguous(), features_rest=sc.colors[:, 1:, :].contiguous())  # fmt: skip
            t = {k: v.to(DEV).requires_grad_(True) for k, v in raw.items()}
            vr = torch.randn(1, 1080, 1920, 3, generator=torch.Generator().manual_seed(3)).to(DEV)
            r, a, info = rasterize_gauss_params(t["means"], t["quats"], t["log_scales"], t["opacity_logits"], t["features_dc"],
                                                t["features_rest"], sc.viewmats[1:2].to(DEV), sc.Ks[1:2].to(DEV), 1920, 1080, 3,
                                                background=torch.tensor([0.3, 0.9, 0.1], device=DEV), clamp=True, absgrad=True)  # fmt: skip
            ((r * vr).sum() + a.sum()).backward()
            outs[mode] = (r.detach(), a.detach(), info["last_ids"], {k: v.grad for k, v in t.items()})
        assert (ctx.heavy_calls > calls) == (mode == "always")
    (r1, a1, l1, g1), (r0, a0, l0, g0) = outs["always"], outs["never"]
    assert rel_err(r1, r0) < 2e-6 and rel_err(a1, a0) < 2e-6
    assert last_ids_agree(l1, l0)
    for k in g1:
        assert rel_l2(g1[k], g0[k]) < 1e-5, k


def test_wide_jobs_leave_their_list_of_open_strips_empty_for_a_second_forward(monkeypatch):
    """The prefix jobs of heavy tiles append the strips they leave open to a list inside the job-list buffer, the wide
    launch behind the main one reads it and its last workgroup resets it: a second forward over the SAME lists (the ones
    bin_tiles left on the offsets tensor) must find it empty and give the same image, bit for bit."""
    W, H = 1920, 1080
    sc = synthetic_scene(60_000, W, H, n_views=2, sh_degree=3, seed=5, log_scale_mean=math.log(0.03))
    sc.means[20_000:40_000] = sc.means[20_000:40_000] * 0.1 + torch.tensor([0.8, -0.2, 0.0])  # faint cluster: long open lists
    sc.opacities[20_000:40_000] *= 0.04
    ctx = ops.RasterContext()
    if int(_lib.load().fg_raster_jobs_words(W, H, 16, ctx.cfg())) == 0 or ctx.policy.heavy_wide == 0 or not ctx.jobs_in_fill:
        pytest.skip("classic launches / the three-launch form in this environment")
    ctx.heavy_tiles, ctx.heavy_tile_len = "always", 1792
    with ops.use(ctx), torch.no_grad():
        t = [x.to(DEV) for x in (sc.means, sc.quats, sc.scales, sc.opacities, sc.colors)]
        radii, m2, depths, conics, tiles, splats = ops.preprocess(*t, None, sc.viewmats[1].to(DEV), sc.Ks[1].to(DEV), W, H,
                                                                  0.3, 0.01, 1e10, 0.0, 16, False, 3, False)  # fmt: skip
        _, ids, offs = ops.bin_tiles(m2, radii, depths, tiles, 16, W // 16, (H + 15) // 16, want_keys=False,
                                     keys_rects=getattr(splats, "_fg_bin", None), raster_hint=(3, W, H))  # fmt: skip
        assert getattr(offs, "_fg_jobs", None) is not None and offs._fg_jobs[1], "list shares expected (checkpoint budget)"
        lens = torch.diff(offs)
        assert int(lens.max()) > 2560  # (the faint cluster's lists never close: open strips behind the prefix)
        jobs = offs._fg_jobs[0]
        # words of a list: 8 + 8 cap | local list 8 + 8 * 8192 (wide jobs: the open strips' list) | heavy list 8 + 8 * 2048 | slot table
        cap = (jobs.shape[1] - 8 - (8 + 8 * 8192) - (8 + 8 * 2048) - (W // 16) * ((H + 15) // 16)) // 8
        outs = []
        for _ in range(2):
            r, a, last = ops.rasterize_splats(splats, m2[None], 3, W, H, 16, offs, ids)
            torch.cuda.synchronize()
            outs.append((r.clone(), a.clone(), last.clone()))
            assert int(jobs[0, 8 + 8 * cap]) == 0 and int(jobs[0, 8 + 8 * cap + 1]) == 0, "open list / ticket not reset"
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])


@pytest.mark.parametrize("cluster", ["faint", "opaque"])
def test_heavy_tiles_stay_on_only_where_the_forward_reports_long_walks(cluster):
    """The one-call path hands the raster forward a pinned word (fg_raster_jobs_fwd walk_out; word 9 of io->ckpt_need_out); a
    strip that evaluated more than 2560 list entries for its strips (until late in round 6: walked) stores that number there, the host reads it one call late and keeps
    `heavy_tiles` for the shape only while such walks are reported.  A faint cluster (lists of thousands that never close)
    keeps them; an opaque one (longer lists still, closed after a few hundred entries) loses them after its first reporting
    call.  Same image either way (2e-6: heavy tiles associate differently)."""
    from freegaussian_amd.rasterization import rasterize_gauss_params

    W, H = 1920, 1080
    n_in = 60_000 if cluster == "faint" else 30_000  # (in the cluster; 30 000 more around it)
    sc = synthetic_scene(n_in + 30_000, W, H, n_views=2, sh_degree=3, seed=5, log_scale_mean=math.log(0.03))
    sc.means[:n_in] = sc.means[:n_in] * 0.1 + torch.tensor([0.8, -0.2, 0.0])
    if cluster == "faint":
        sc.opacities[:n_in] *= 0.04  # lists of up to 8700 entries, pixels open to the end
    else:
        sc.opacities[:n_in] = 0.95  # lists of up to 11 000 entries, no pixel takes more than ~2400
    ctx = ops.RasterContext(env={})
    if not ctx.step_calls or int(_lib.load().fg_raster_jobs_words(W, H, 16, ctx.cfg())) == 0:
        pytest.skip("the stage-wise calls / classic launches in this environment")
    raw = dict(means=sc.means, quats=sc.quats, log_scales=sc.scales.log(), opacity_logits=torch.logit(sc.opacities.clamp(1e-4, 1 - 1e-4)),
               features_dc=sc.colors[:, 0, :].contiguous(), features_rest=sc.colors[:, 1:, :].contiguous())  # fmt: skip
    t = {k: v.to(DEV).requires_grad_(True) for k, v in raw.items()}  # (a training forward: list shares, checkpoints, reports)
    vm, K = sc.viewmats[1:2].to(DEV), sc.Ks[1:2].to(DEV)
    images, heavy = [], []
    with ops.use(ctx):
        for _ in range(6):
            calls = ctx.heavy_calls
            r, _a, info = rasterize_gauss_params(t["means"], t["quats"], t["log_scales"], t["opacity_logits"], t["features_dc"],
                                                 t["features_rest"], vm, K, W, H, 3)  # fmt: skip
            torch.cuda.synchronize()
            images.append(r.detach().clone())
            heavy.append(ctx.heavy_calls > calls)
            del r, _a
    lens = torch.diff(info["raster_isect_offsets"].reshape(-1))
    assert int(lens.max()) > ctx.heavy_flag_len  # long lists in both scenes: round 4's rule turned heavy tiles on for either
    (lkey,) = list(ctx.long_walks)
    print(cluster, "heavy tiles per call:", heavy, "walk countdown:", ctx.long_walks[lkey], "longest list:", int(lens.max()))
    assert heavy[1] or heavy[2]  # (on by the list length until a forward has reported)
    if cluster == "faint":
        assert all(heavy[2:]) and ctx.long_walks[lkey] > 0
    else:
        assert not any(heavy[4:]) and ctx.long_walks[lkey] == 0
    for im in images[1:]:
        assert rel_err(im, images[0]) < 2e-6


def test_full_size_cfg4_whole_frame_and_all_gradients_vs_oracle():
    """BASELINE.json's headline input itself -- 1M Gaussians, 1920x1080, SH 3, 7.2M intersections --
    forward and backward against the oracle on the WHOLE frame (torch projection / SH / sort + C
    compositing, ~30 s of CPU): lists bit-exact, image within the bar, every gradient incl. absgrad."""
    from freegaussian_amd.scenes import north_star_scene

    sc = north_star_scene(n_views=1)
    ref_in, gpu_in, o0, o1 = _oracle_full_res(sc, 0, "RGB", 3, with_alpha_grad=False)  # the bench's upstream gradient
    assert o0[2]["flatten_ids"].numel() > 7_000_000
    worst = _assert_full_parity(ref_in, gpu_in, o0, o1, REL_TOL)  # the bar itself (share jobs without rho: up to 1.2e-4)
    assert psnr(o1[0], o0[0]) > 80
    print("cfg4 whole-frame rel-L2 of gradients vs oracle:", {k: f"{v:.2e}" for k, v in worst.items()})


def test_clustered_scene_content_split_jobs_match_classic_launch_and_oracle(monkeypatch):
    """A non-uniform scene at 1920x1080 (8160 tiles: job lists are active by default): half of the
    Gaussians sit in a small ball, so the centre tiles hold lists many times the mean and are split
    by CONTENT.  Forward must equal the classic launch bit for bit, backward up to atomic order, and
    a crop over the heavy tiles the scalar C oracle."""
    from oracle import c_oracle as CO

    sc = synthetic_scene(200_000, 1920, 1080, n_views=1, sh_degree=3, seed=9)
    sc.means[:100_000] *= 0.2
    W, H = sc.width, sc.height
    vm, K = sc.viewmats[:1].to(DEV), sc.Ks[:1].to(DEV)
    g = torch.Generator().manual_seed(0)
    vr = torch.randn(1, H, W, 3, generator=g).to(DEV)
    outs = []
    heavy_calls = ops.default_context.heavy_calls
    for classic in (False, True):
        if classic:
            _setenv_policy(monkeypatch, "FG_RASTER_TAIL_FWD", "0")
            _setenv_policy(monkeypatch, "FG_RASTER_TAIL_BWD", "0")
        t = [x.to(DEV).requires_grad_(True) for x in (sc.means, sc.quats, sc.scales, sc.opacities, sc.colors)]
        r, a, info = rasterization(*t, vm, K, W, H, sh_degree=3, packed=False, absgrad=True)
        (r * vr).sum().backward()
        outs.append((r.detach(), a.detach(), [x.grad for x in t], info))
    (r, a, grads, info), (r_c, a_c, grads_c, _) = outs
    offs, ids = info["isect_offsets"].reshape(-1), info["flatten_ids"]
    lens = torch.diff(torch.cat([offs, offs.new_tensor([ids.numel()])])) if offs.numel() == 8160 else torch.diff(offs)
    assert int(lens.max()) > 20 * ids.numel() // 65536  # some tiles really are above the quarter threshold
    if ops.default_context.heavy_calls == heavy_calls:
        assert torch.equal(r, r_c) and torch.equal(a, a_c)
    else:  # (the shape ran with heavy tiles -- earlier tests of the process flagged it, or FG_HEAVY_TILES=always --: their wide
        # jobs associate the same sums and products differently)
        assert rel_err(r, r_c) < 2e-6 and rel_err(a, a_c) < 2e-6
    for x, y in zip(grads, grads_c):
        assert rel_l2(x, y) < REL_TOL  # (list shares of the backward: suffix colours by subtraction, one more rounding)
    # centre crop (the heavy tiles) against the C oracle, as in the cfg4 test
    cw, ch = 160, 96
    x0, y0 = (W - cw) // 2 // 16 * 16, (H - ch) // 2 // 16 * 16
    tw = info["tile_width"]
    offs_c, ids_c = offs.cpu(), ids.cpu()
    ends = torch.cat([offs_c[1:], torch.tensor([ids_c.numel()], dtype=offs_c.dtype)]) if offs_c.numel() == 8160 else offs_c[1:]
    lists, coffs = [], [0]
    for ty in range(ch // 16):
        for tx in range(cw // 16):
            tt = (y0 // 16 + ty) * tw + (x0 // 16 + tx)
            lists.append(ids_c[int(offs_c[tt]) : int(ends[tt])])
            coffs.append(coffs[-1] + lists[-1].numel())
    cv, coffs = torch.cat(lists), torch.tensor(coffs, dtype=torch.int32)
    m2 = info["means2d"][0].detach().cpu() - torch.tensor([float(x0), float(y0)])
    campos = torch.linalg.inv(sc.viewmats[0])[:3, 3]
    rgb = torch.clamp_min(O.sh_eval(3, sc.means - campos, sc.colors) + 0.5, 0.0)
    rc, ac, _ = CO.raster_fwd(m2, info["conics"][0].cpu(), rgb, sc.opacities, cw, ch, 16, coffs, cv)
    assert close_except_knife_edge(r[0, y0 : y0 + ch, x0 : x0 + cw], rc, REL_TOL)
    assert close_except_knife_edge(a[0, y0 : y0 + ch, x0 : x0 + cw], ac, REL_TOL)


# ------------------------------------------------------------------------------------------
# BASELINE.json configs[1], [2], [4] at (or near) their stated sizes
def test_cfg2_conerf_like_300k_psnr_and_gradients():
    """configs[1] 'CoNeRF single scene (~300k Gaussians), fwd+bwd PSNR match': 300k, 960x540
    (dataset absent -> synthetic)."""
    sc = synthetic_scene(300_000, 960, 540, n_views=1, seed=42)
    ref_in, gpu_in, (r0, a0, i0), (r1, a1, i1) = _run_both(sc, 0, "RGB", 3)
    assert torch.equal(i1["flatten_ids"].cpu(), i0["flatten_ids"])
    assert torch.equal(i1["isect_offsets"].cpu(), i0["isect_offsets"])
    assert psnr(r1, r0) >= 60.0
    assert close_except_knife_edge(r1, r0, REL_TOL) and close_except_knife_edge(a1, a0, REL_TOL)
    for k in ref_in:
        assert rel_l2(gpu_in[k].grad, ref_in[k].grad) < REL_TOL, k


@pytest.mark.parametrize("N,W,H", [(60_000, 480, 270), (300_000, 960, 540)])
def test_cfg3_flow_derivative_scene(N, W, H):
    """configs[2] 'LiveScene-sim scene with flow-derivative loss enabled': second pose = first pose
    moved by dt=(0.02,0,0), dw=(0,0.01,0); per-Gaussian rigid screw motion; F1 composited flow via
    render_with_flow, F2 per-Gaussian Jacobian terms, camera flow map -- all against the oracle.
    The second case is SURVEY.md section 8d's stated size (300k Gaussians, 960x540), with the C
    compositing inside the oracle; the first keeps the pure-torch oracle in the loop."""
    from freegaussian_amd import flow as FL
    from freegaussian_amd.utils import exp_se3, from_homogenous, to_homogenous
    from oracle import c_oracle as CO

    compositor = CO.composite if N > 100_000 else None
    sc = synthetic_scene(N, W, H, n_views=1, seed=42)
    g = torch.Generator().manual_seed(5)
    screw = torch.cat([torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1),
                       torch.randn(N, 3, generator=g) * 0.3], -1)  # fmt: skip
    T = exp_se3(screw, torch.full((N, 1), 0.02))
    means_t = from_homogenous(torch.bmm(T, to_homogenous(sc.means).unsqueeze(-1)).squeeze(-1))
    vm_t = sc.viewmats[:1]
    dvm = torch.eye(4)
    ang = 0.01
    dvm[:3, :3] = torch.tensor([[math.cos(ang), 0, math.sin(ang)], [0, 1, 0], [-math.sin(ang), 0, math.cos(ang)]])
    dvm[0, 3] = 0.02
    vm_0 = (dvm @ vm_t[0])[None]
    K = sc.Ks[:1]

    # oracle: two projections, displacement channels, one raster
    ref = [x.clone().requires_grad_(True) for x in (means_t, sc.means, sc.quats, sc.scales, sc.opacities, sc.colors)]
    p0 = O.project(ref[1], ref[2], ref[3], vm_0[0], K[0], W, H)
    pt = O.project(ref[0], ref[2], ref[3], vm_t[0], K[0], W, H)
    both = ((p0.radii > 0) & (pt.radii > 0))[:, None]
    disp = torch.where(both, pt.means2d - p0.means2d, torch.zeros_like(pt.means2d))
    r0, a0, _ = O.rasterization(ref[0], ref[2], ref[3], ref[4], ref[5], vm_t, K, W, H, sh_degree=3,
                                render_mode="RGB+ED", extra_channels=disp, compositor=compositor)  # fmt: skip
    target = torch.randn(1, H, W, 2, generator=g) * 0.1
    loss0 = (r0[..., 4:] - target).abs().mean() + r0[..., :3].mean()
    loss0.backward()

    gpu = [x.detach().to(DEV).requires_grad_(True) for x in ref]
    out = FL.render_with_flow(gpu[0], gpu[1], gpu[2], gpu[3], gpu[4], gpu[5], vm_t.to(DEV), vm_0.to(DEV),
                              K.to(DEV), W, H, sh_degree=3, render_mode="RGB+ED")  # fmt: skip
    loss1 = FL.flow_loss(out["flow_gs"], target.to(DEV)) + out["render"][..., :3].mean()
    loss1.backward()
    assert out["flow_gs"].shape == (1, H, W, 2) and out["render"].shape == (1, H, W, 4)
    assert close_except_knife_edge(out["flow_gs"], r0[..., 4:], REL_TOL)
    assert close_except_knife_edge(out["render"], r0[..., :4], REL_TOL)
    assert abs(loss1.item() - loss0.item()) < 1e-5
    for a, b, name in zip(gpu, ref, ["means_t", "means_0", "quats", "scales", "opacities", "colors"]):
        assert rel_l2(a.grad, b.grad) < REL_TOL, name

    # F2 + camera flow map with the (v, w) of this camera pair
    c2w_t, c2w_0 = torch.linalg.inv(vm_t[0]), torch.linalg.inv(vm_0[0])
    for m in (c2w_t, c2w_0):
        m[:3, 1:3] *= -1  # back to the OpenGL convention relative_camera_motion expects
    v, w = FL.relative_camera_motion(c2w_0[:3], c2w_t[:3])
    depth0 = r0[0, ..., 3].detach()
    fx, fy, cx, cy = K[0, 0, 0], K[0, 1, 1], K[0, 0, 2], K[0, 1, 2]
    cam0 = O.camera_flow(depth0.double(), fx.double(), fy.double(), cx.double(), cy.double(), v, w)
    depth0 = depth0.clamp_min(0.05)  # the SAME depth map on both sides (empty pixels have depth 0)
    cam0 = O.camera_flow(depth0.double(), fx.double(), fy.double(), cx.double(), cy.double(), v, w)
    cam1 = FL.camera_flow_map(depth0.to(DEV), K[0].to(DEV), v.float().to(DEV), w.float().to(DEV))
    assert rel_err(cam1, cam0.float()) < REL_TOL
    vel = torch.randn(N, 3, generator=g)
    ug0, uc0 = O.gaussian_flow(pt.means2d.detach(), pt.depths.detach().clamp_min(1e-3), vel, fx, fy, cx, cy, v.float(), w.float())
    ug1, uc1 = ops.gaussian_flow(out["info"]["means2d"][0].detach(), out["info"]["depths"][0].detach().clamp_min(1e-3),
                                 vel.to(DEV), K[0].to(DEV), v.float().to(DEV), w.float().to(DEV))  # fmt: skip
    assert rel_err(ug1, ug0) < REL_TOL and rel_err(uc1, uc0) < REL_TOL


@pytest.mark.parametrize("n,W,H", [(6000, 160, 96), (50_000, 1920, 1080), (1_000_000, 1920, 1080)])
def test_cfg5_control_stage2_matches_oracle_host_path(n, W, H):
    """configs[4] 'freegaussian-control stage-2': frozen deform -> per-attribute mean displacement
    -> control MLP -> deltas scattered into the masked Gaussians -> the same raster call.
    Second case: 1920x1080 = 8160 tiles -- the mixed / job-list launches under the stage-2 front end, 5% of
    the rows masked; oracle with the C compositing.  Third case: SURVEY.md section 8d's stated size ("as
    cfg4": 1M Gaussians at 1920x1080)."""
    import copy

    from freegaussian_amd.model import FreeGaussianControlModel, FreeGaussianModelConfig
    from freegaussian_amd.utils import from_homogenous, get_viewmat, to_homogenous
    from oracle import c_oracle as CO

    compositor = CO.composite if W * H > 10**6 else None
    _, base, cam = _model_and_camera(n=n, W=W, H=H, training=True, log_scale=-3.2 if n < 10**6 else -5.0)
    N = base.num_points
    mask = torch.zeros(N, 3, dtype=torch.bool)
    mask[: N // 20, 0] = True  # 5% of the rows set (cfg5), overlapping attributes
    mask[N // 30 : N // 12, 1] = True
    mask[N // 6 : N // 6 + N // 60, 2] = True
    init_cam = type(cam)(cam.camera_to_worlds, cam.fx, cam.fy, cam.cx, cam.cy, cam.width, cam.height,
                         times=torch.tensor([[0.0]]))  # fmt: skip
    cfg = FreeGaussianModelConfig(background_color="black")
    cm_cpu = FreeGaussianControlModel(mask, init_cam, config=cfg, seed_points=base.means.detach().clone(), init_scales=-4.0)
    cm_cpu.load_state_dict(base.state_dict(), strict=False)
    with torch.no_grad():
        for p in cm_cpu.control.parameters():
            p.mul_(0.2)
    cm = copy.deepcopy(cm_cpu).to(DEV).train()
    cm_cpu.train()
    out = cm.get_outputs(cam)

    # the same host math on CPU + the oracle raster
    sel = mask.any(-1)
    pts, pmask = cm_cpu.means[sel], mask[sel]
    with torch.no_grad():
        def deformed(t):
            T, _, _ = cm_cpu.deform(pts, t.expand(pts.shape[0], -1))
            return from_homogenous(torch.bmm(T, to_homogenous(pts).unsqueeze(-1)).squeeze(-1))

        delta = deformed(cam.times) - deformed(init_cam.times)
        d_avg = torch.stack([delta[pmask[:, i]].mean(0) for i in range(3)])
    value = pmask.float() @ d_avg / pmask.sum(-1, keepdim=True)
    d_xyz, d_rot, d_scale = cm_cpu.control(pts, value)
    idx = sel.nonzero().squeeze(-1)
    means = cm_cpu.means + torch.zeros_like(cm_cpu.means).index_put((idx,), d_xyz)
    scales = torch.exp(cm_cpu.scales) + torch.zeros_like(cm_cpu.scales).index_put((idx,), d_scale)
    quats = cm_cpu.quats / cm_cpu.quats.norm(dim=-1, keepdim=True) + torch.zeros_like(cm_cpu.quats).index_put((idx,), d_rot)
    colors, deg = cm_cpu._colors_and_degree()
    r0, a0, _ = O.rasterization(means, quats, scales, torch.sigmoid(cm_cpu.opacities).squeeze(-1), colors,
                                get_viewmat(cam.camera_to_worlds), cam.get_intrinsics_matrices(), cam.width,
                                cam.height, sh_degree=deg, render_mode="RGB", packed=False, compositor=compositor)  # fmt: skip
    rgb0 = torch.clamp(r0[..., :3] + (1 - a0) * torch.zeros(3), 0.0, 1.0).squeeze(0)
    assert deg == 3 and cm.step == 30000
    assert close_except_knife_edge(out["rgb"], rgb0, REL_TOL)
    gt = torch.rand(cam.height, cam.width, 3, generator=torch.Generator().manual_seed(9))
    _smooth_loss(out["rgb"], gt.to(DEV)).backward()
    _smooth_loss(rgb0, gt).backward()
    gc = torch.cat([p.grad.flatten() for p in cm.control.parameters()])
    gc0 = torch.cat([p.grad.flatten() for p in cm_cpu.control.parameters()])
    assert rel_l2(gc, gc0) < REL_TOL  # (measured 2e-6 at 1M Gaussians)
    assert rel_l2(cm.gauss_params["means"].grad, cm_cpu.gauss_params["means"].grad) < REL_TOL
    assert all(p.grad is None for p in cm.deform.parameters())  # frozen (evaluated under no_grad)


@pytest.mark.parametrize("H,W,C", [(11, 11, 3), (12, 37, 1), (64, 48, 3), (270, 480, 3), (135, 241, 3), (1080, 1920, 3)])
def test_fused_l1_ssim_loss_equals_the_torch_statement(H, W, C):
    """csrc/loss.hip (one launch forward, one backward) against `harness.ssim` -- the torch statement of
    pytorch_msssim.SSIM(data_range=1, size_average=True): 11 x 11 Gaussian window, sigma 1.5, valid borders -- and
    torch's |gt - pred|.mean(), values and the gradient with respect to pred; tile-edge sizes, one channel, the
    smallest image SSIM is defined on; bit-identical on repetition (no float atomics)."""
    from freegaussian_amd.harness import main_loss, ssim

    g = torch.Generator().manual_seed(H * 1000 + W)
    gt = torch.rand(H, W, C, generator=g)
    pred = (gt + 0.15 * torch.randn(H, W, C, generator=g)).clamp(0, 1)  # (correlated, as a render is; clamped pixels tie with gt at 0 / 1)
    pred[0, 0] = gt[0, 0]  # |.| at zero: gradient 0, as torch
    ref_in = pred.double().requires_grad_(True)
    l1_ref = (gt.double() - ref_in).abs().mean()
    ss_ref = ssim(gt.double().permute(2, 0, 1)[None], ref_in.permute(2, 0, 1)[None])
    (0.8 * l1_ref + 0.2 * (1 - ss_ref) + 0.3 * ss_ref * l1_ref).backward()  # (both outputs' gradients in use, non-trivially)
    gpu_in = pred.to(DEV).requires_grad_(True)
    l1, ss = ops.l1_ssim(gpu_in, gt.to(DEV))
    (0.8 * l1 + 0.2 * (1 - ss) + 0.3 * ss * l1).backward()
    assert abs(float(l1) - float(l1_ref)) < 1e-6 * max(float(l1_ref), 1e-3)
    assert abs(float(ss) - float(ss_ref)) < 2e-6
    assert rel_l2(gpu_in.grad.cpu().double(), ref_in.grad) < 2e-5
    l1b, ssb = ops.l1_ssim(gpu_in, gt.to(DEV))
    assert torch.equal(l1b, l1) and torch.equal(ssb, ss)
    # the model's loss goes through it on the GPU
    if C == 3:
        m = main_loss(gpu_in, gt.to(DEV), 0.2)
        assert abs(float(m) - float(0.8 * l1_ref + 0.2 * (1 - ss_ref))) < 2e-6


def test_densification_statistics_kernel_equals_the_torch_lines():
    """fg_densify_stats (S1, after_train_iter :369-392) against the masked torch updates it replaces: visible rows
    accumulate |absgrad|, a visit count and the largest screen radius; invisible rows (radius 0, or negative) stay
    bit for bit as they were."""
    g = torch.Generator().manual_seed(3)
    N = 100_003
    absgrad = torch.rand(N, 2, generator=g) * 10.0 ** torch.randint(-6, 1, (N, 1), generator=g).float()
    radii = torch.randint(-2, 60, (N,), generator=g, dtype=torch.int32)
    stats = [torch.rand(N, generator=g), torch.randint(1, 9, (N,), generator=g).float(), torch.rand(N, generator=g) * 0.05]
    ref = [t.clone() for t in stats]
    vis = radii > 0
    ref[0] += torch.where(vis, absgrad.norm(dim=-1), torch.zeros(N))
    ref[1] += vis.float()
    ref[2] = torch.maximum(ref[2], torch.where(vis, radii.float() / 1920.0, torch.zeros(N)))
    dev = [t.to(DEV) for t in stats]
    ops.densify_stats(absgrad.to(DEV), radii.to(DEV), 1920.0, *dev)
    assert rel_l2(dev[0].cpu(), ref[0]) < 1e-7 and torch.equal(dev[1].cpu(), ref[1]) and torch.equal(dev[2].cpu(), ref[2])
    for a, b in zip(dev, stats):
        assert torch.equal(a.cpu()[~vis], b[~vis])


def test_tall_linear_gradients_equal_nn_linear():
    """deform._TallLinear: the MLP's linears over hundreds of thousands of rows with the weight gradient as a batched
    product over row chunks + a sum (the library's one-kernel [out, N] x [N, in] product runs at a fraction of its rate):
    same output bits, gradients equal to summation order; a row count that is not a multiple of the chunk."""
    from freegaussian_amd.deform import _TallLinear, _linear

    g = torch.Generator().manual_seed(9)
    N = 4 * _TallLinear.CHUNK + 1234
    lin = torch.nn.Linear(63, 256).to(DEV)
    x = torch.randn(N, 63, generator=g).to(DEV)
    go = torch.randn(N, 256, generator=g).to(DEV)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya = _linear(lin, xa)
    assert ya.grad_fn is not None and "TallLinear" in type(ya.grad_fn).__name__
    ya.backward(go)
    ga = [xa.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone()]
    lin.zero_grad()
    yb = lin(xb)
    yb.backward(go)
    assert torch.equal(ya.detach(), yb.detach())
    for a, b in zip(ga, [xb.grad, lin.weight.grad, lin.bias.grad]):
        assert rel_l2(a, b) < 1e-5  # (two fp32 summation orders over 34 000 rows)
    exact = go.double().t() @ x.double()  # the weight gradient in double: the chunked sum is no further from it
    assert rel_l2(ga[1].double(), exact) < max(2 * rel_l2(lin.weight.grad.double(), exact), 2e-6)
    assert "TallLinear" not in type(_linear(lin, x[:100].requires_grad_(True)).grad_fn).__name__  # small inputs: nn.Linear


def test_fused_adam_step_equals_torch_adam():
    """optim.FusedAdam (csrc/adam.hip: one launch per tensor) against torch.optim.Adam on the same parameters and
    gradients: 25 steps with a changing learning rate (the schedules write group["lr"]), the reference's eps = 1e-15,
    tensor sizes around the float4 boundary; parameters and both moments agree to float rounding; the state layout
    is torch's (the densification edits it in place)."""
    from freegaussian_amd.optim import FusedAdam

    g = torch.Generator().manual_seed(11)
    shapes = [(1,), (3,), (1000, 3), (777, 15, 3), (4097, 4), (100_003,)]
    a = [torch.randn(s, generator=g).to(DEV).requires_grad_(True) for s in shapes]
    b = [x.detach().clone().requires_grad_(True) for x in a]
    oa, ob = FusedAdam(a, lr=1.6e-4, eps=1e-15), torch.optim.Adam(b, lr=1.6e-4, eps=1e-15)
    for step in range(25):
        lr = 1.6e-4 * (0.9 ** step)
        for o in (oa, ob):
            o.param_groups[0]["lr"] = lr
        for x, y in zip(a, b):
            gr = (torch.randn(x.shape, generator=g) * (10.0 ** float(torch.randint(-6, 2, (1,), generator=g)))).to(DEV)
            if step % 7 == 3:
                gr[::2] = 0  # zero gradients (culled Gaussians): sqrt(v) -> tiny denominators with eps = 1e-15
            x.grad, y.grad = gr.clone(), gr.clone()
        oa.step()
        ob.step()
    for x, y in zip(a, b):
        assert rel_l2(x.detach(), y.detach()) < 1e-6
        sa, sb = oa.state[x], ob.state[y]
        assert set(sa) == {"step", "exp_avg", "exp_avg_sq"} and float(sa["step"]) == float(sb["step"]) == 25
        assert rel_l2(sa["exp_avg"], sb["exp_avg"]) < 1e-6 and rel_l2(sa["exp_avg_sq"], sb["exp_avg_sq"]) < 1e-6
    assert set(oa.state_dict()["state"][0]) == set(ob.state_dict()["state"][0])
    # step_all: the tensors of several FusedAdam optimizers (one launch for every 16) and a torch optimizer beside them;
    # per tensor the same bits as that optimizer's own step()
    from freegaussian_amd.optim import step_all

    ps = [torch.randn(s_, generator=g).to(DEV).requires_grad_(True) for s_ in [(5,), (64, 3), (1000,)] * 7]  # 21 tensors
    qs = [x.detach().clone().requires_grad_(True) for x in ps]
    for x, y in zip(ps, qs):
        x.grad = torch.randn(x.shape, generator=g).to(DEV)
        y.grad = x.grad.clone()
    o1 = [FusedAdam([x], lr=1e-3 * (i + 1), eps=1e-15) for i, x in enumerate(ps[:20])] + [torch.optim.Adam([ps[20]], lr=1e-3)]
    o2 = [FusedAdam([y], lr=1e-3 * (i + 1), eps=1e-15) for i, y in enumerate(qs[:20])] + [torch.optim.Adam([qs[20]], lr=1e-3)]
    for _ in range(2):
        step_all(o1)
        for o in o2:
            o.step()
    assert all(torch.equal(x.detach(), y.detach()) for x, y in zip(ps, qs))
    assert all(float(o.state[x]["step"]) == 2 for o, x in zip(o1, ps))
    # gradients that are views into one flat buffer at offsets that are not multiples of 16 bytes (viewdp's layout)
    flat = torch.randn(3 + 7 * 3 + 5, generator=g).to(DEV)
    p1, p2 = (torch.zeros(7, 3, device=DEV, requires_grad=True), torch.zeros(5, device=DEV, requires_grad=True))
    q1, q2 = (x.detach().clone().requires_grad_(True) for x in (p1, p2))
    p1.grad, p2.grad = flat[3:24].view(7, 3), flat[24:]
    q1.grad, q2.grad = p1.grad.clone(), p2.grad.clone()
    FusedAdam([p1, p2], lr=1e-2).step()
    torch.optim.Adam([q1, q2], lr=1e-2).step()
    assert rel_l2(p1.detach(), q1.detach()) < 1e-6 and rel_l2(p2.detach(), q2.detach()) < 1e-6


def test_harness_training_steps_reduce_loss():
    """End to end: a few optimisation steps of the host harness (reference loss + optimizer table)
    through the HIP raster must fit a target rendered from perturbed parameters."""
    import copy

    from freegaussian_amd import harness as Hn

    model, _, cam = _model_and_camera(n=3000, W=128, H=96, step=1500, training=True)  # before warm_up: static
    target = copy.deepcopy(model)
    with torch.no_grad():
        target.gauss_params["features_dc"].add_(0.3 * torch.randn_like(target.gauss_params["features_dc"]))
        target.gauss_params["means"].add_(0.01 * torch.randn_like(target.gauss_params["means"]))
    target.eval()
    with torch.no_grad():
        gt = target.get_outputs(copy.deepcopy(cam))["rgb"].clamp(0, 1)
    opts = Hn.build_optimizers(model)
    hist = [Hn.train_step(model, opts, copy.deepcopy(cam), gt, 1500 + i) for i in range(25)]
    assert hist[-1]["loss"] < 0.7 * hist[0]["loss"] and hist[-1]["psnr"] > hist[0]["psnr"] + 1.0
    assert model.xys_grad_norm is not None and float(model.vis_counts.max()) == 26.0
    # the Gaussian groups step through optim.FusedAdam, the MLP groups through torch's Adam
    from freegaussian_amd.optim import FusedAdam

    assert all(isinstance(opts[k], FusedAdam) for k in model.gauss_params) and not isinstance(opts["deform"], FusedAdam)
    # metrics_every = 5: loss / psnr are read back on steps divisible by 5 only; training continues the same way
    more = [Hn.train_step(model, opts, copy.deepcopy(cam), gt, 1525 + i, metrics_every=5) for i in range(10)]
    assert [("loss" in m) for m in more] == [True, False, False, False, False, True, False, False, False, False]
    assert more[5]["loss"] < hist[-1]["loss"] * 1.05 and all(m["gaussian_count"] == model.num_points for m in more)


def test_edge_cases_zero_gaussians_and_short_sh_tables():
    """N = 0 renders an empty image without launching on empty buffers; SH tables shorter than 16
    bases ([N,9,3] with degree 2, [N,1,3] with degree 0) go through the fused path."""
    z = torch.zeros
    r, a, info = rasterization(z(0, 3, device=DEV), z(0, 4, device=DEV), z(0, 3, device=DEV), z(0, device=DEV),
                               z(0, 16, 3, device=DEV), torch.eye(4, device=DEV)[None], torch.eye(3, device=DEV)[None],
                               40, 24, sh_degree=3, packed=False, render_mode="RGB+ED")  # fmt: skip
    assert r.shape == (1, 24, 40, 4) and float(r.abs().max()) == 0.0 and info["radii"].shape == (1, 0)
    assert info["isect_offsets"].numel() == 3 * 2 + 1 and info["flatten_ids"].numel() == 0
    sc = _scene(n=4000, w=96, h=64, seed=12)
    for deg, K in ((2, 9), (0, 1), (1, 16)):
        cpu = [sc.means, sc.quats, sc.scales, sc.opacities, sc.colors[:, :K].contiguous()]
        ref = [t.clone().requires_grad_(True) for t in cpu]
        gpu = [t.to(DEV).requires_grad_(True) for t in cpu]
        r0, a0, _ = O.rasterization(*ref, sc.viewmats[:1], sc.Ks[:1], 96, 64, sh_degree=deg, packed=False)
        r1, a1, _ = rasterization(*gpu, sc.viewmats[:1].to(DEV), sc.Ks[:1].to(DEV), 96, 64, sh_degree=deg, packed=False)
        g = torch.randn(r0.shape, generator=torch.Generator().manual_seed(deg))
        (r0 * g).sum().backward()
        (r1 * g.to(DEV)).sum().backward()
        assert close_except_knife_edge(r1, r0, REL_TOL)
        assert gpu[4].grad.shape == (4000, K, 3) and rel_l2(gpu[4].grad, ref[4].grad) < REL_TOL
        assert rel_l2(gpu[0].grad, ref[0].grad) < REL_TOL
    with pytest.raises(ValueError):
        rasterization(*[t.to(DEV) for t in cpu[:4]], torch.rand(4000, 9, device=DEV), sc.viewmats[:1].to(DEV),
                      sc.Ks[:1].to(DEV), 96, 64, sh_degree=None, packed=False)  # 9 channels > 8


def test_factored_view_dp_exchange_two_ranks_on_one_gpu():
    """viewdp.FlatGaussianParams.factored_exchange (all-gather of the colour gradient + small
    all-reduce + fg_sh_grad_accumulate) against the plain flat all-reduce, 2 ranks sharing this GPU
    over gloo, real kernels, run as child processes."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for per_view, port in (("0", "29533"), ("1", "29535")):  # shared means / per-view (deformed) means
        env = dict(os.environ, FG_BENCH_BACKEND="gloo", FG_PER_VIEW_MEANS=per_view)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", port, os.path.join(root, "scripts", "exchange_check.py")]  # fmt: skip
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0 and "exchange ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_world_size_8_exchange_and_bench_launcher_over_gloo_on_one_gpu():
    """The driver's 8-GPU RCCL run must not be the first time world = 8 executes: (1) the factored exchange
    against the plain all-reduce with EIGHT ranks (eight payload strides in the all-gather, eight views in the
    local SH rebuild), shared and per-view means, over gloo with all ranks on this GPU; (2) `bench.py --gpus 8`
    through its own launcher with FG_BENCH_BACKEND=gloo: n_gpus 8, the factored exchange kept, no fallback."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for per_view, port in (("0", "29541"), ("1", "29543")):
        env = dict(os.environ, FG_BENCH_BACKEND="gloo", FG_PER_VIEW_MEANS=per_view, OMP_NUM_THREADS="4")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr",
               "127.0.0.1", "--master-port", port, os.path.join(root, "scripts", "exchange_check.py")]  # fmt: skip
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and "exchange ok" in out.stdout and "world=8" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(FG_BENCH_BACKEND="gloo", OMP_NUM_THREADS="4")
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1",
                          "--n-gauss", "50000", "--width", "640", "--height", "360"], env=env, capture_output=True,
                         text=True, timeout=900)  # fmt: skip
    assert res.returncode == 0, res.stderr[-3000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 8 and len(line["per_rank_mpix_per_s"]) == 8
    assert line["exchange"]["kind"] == "factored" and line["exchange"]["fallback"] is None
    assert "gloo" in line["exchange"]["backend"] and "gloo gradient exchange" in line["config"]["workload"]


@pytest.mark.parametrize("world,exchange,warm_up,sparse", [(2, "factored", None, "never"), (2, "factored", 12, "never"),
                                                          (2, "plain", None, "never"), (8, "factored", 12, "never"),
                                                          (2, "factored", None, "always"), (2, "factored", 12, "always"),
                                                          (8, "factored", 12, "always"), (2, "factored", 12, "auto")])
def test_view_dp_training_keeps_the_ranks_in_lockstep(world, exchange, warm_up, sparse):
    """FreeGaussianModel trained view-sharded on 2 / 8 ranks sharing this GPU (gloo): the gradient exchange --
    viewdp.ModelViewDP's factored form (colour gradients all-gathered from inside the backward, the rest one all-reduce
    of a flat buffer; with the deformation MLP active the view directions travel along and the MLP gradients ride in
    the all-reduce) or the plain all-reduce --, the densification-statistics exchange and shared split samples keep the
    replicas bit-identical through refinements; the two exchanges give the same gradients on the same model state
    (scripts/dp_train_check.py, child processes).  ``sparse``: the gathered blocks hold only the Gaussians with a colour
    gradient in the rank's view (round 5: fg_payload_compact / fg_payload_expand) -- bit-identical gradients, an overflowing
    block detected by every rank and repeated densely."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FG_BENCH_BACKEND="gloo", FG_DP_EXCHANGE=exchange, FG_DP_SPARSE=sparse)
    if warm_up is not None:
        env["FG_DP_WARM_UP"] = str(warm_up)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(29534 + world + 16 * ("never", "always", "auto").index(sparse)),
           os.path.join(root, "scripts", "dp_train_check.py")]  # fmt: skip
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "dp lockstep ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    assert "factored vs plain exchange" in out.stdout
    if exchange == "factored":
        assert "all_gather_received" in out.stdout


def test_graphed_raster_replays_equal_eager_steps_and_recovers_from_overflow():
    """graphed.GraphedRaster: forward + backward captured in one hipGraph with fixed-capacity lists
    (count read on the device, no host wait); replays for other views equal the eager path; a view
    whose list does not fit is detected, redone eagerly and the graph re-captured."""
    from freegaussian_amd.graphed import GraphedRaster
    from freegaussian_amd.viewdp import FlatGaussianParams

    sc = synthetic_scene(30000, 320, 192, n_views=4, seed=31)
    fp = FlatGaussianParams.from_scene(sc, DEV)
    ref = FlatGaussianParams.from_scene(sc, DEV)
    g = GraphedRaster(fp, sc.width, sc.height, sh_degree=3)
    gen = torch.Generator().manual_seed(2)
    for it, v in enumerate([0, 1, 2, 1, 3]):
        vm, K = sc.viewmats[v : v + 1].to(DEV), sc.Ks[v : v + 1].to(DEV)
        vr = torch.randn(1, sc.height, sc.width, 3, generator=gen).to(DEV)
        if it == 3:
            g.capacity = 2000  # far too small for any view: next replay must overflow
            g._capture()
        r, a, overflow = g.step(vm, K, vr)
        assert overflow == (it == 3)
        with ref.direct_grads():
            r0, a0, _ = rasterization(*ref.raster_inputs(), vm, K, sc.width, sc.height, sh_degree=3, packed=False,
                                      absgrad=True)  # fmt: skip
            r0.backward(vr)
        assert torch.equal(r, r0.detach()) and torch.equal(a, a0.detach())
        assert rel_l2(fp.flat_grad, ref.flat_grad) < 1e-5  # float atomics: not bit-identical
    assert g.graph is not None and g.capacity > 2000


@pytest.mark.parametrize("live", ["0", "1"])
def test_graph_replays_with_classic_forward_and_listed_backward(live, monkeypatch):
    """A forced FG_RASTER_TAIL_BWD on a small image pairs the classic forward launch with the mixed,
    job-list backward: the forward then zero-fills the record-gradient array with a launch of its own
    (a hipMemsetAsync did not end up in torch's captured graph: replays accumulated onto the previous
    replay's gradients) and has written no liveness words for the backward to trust."""
    from freegaussian_amd.graphed import GraphedRaster
    from freegaussian_amd.viewdp import FlatGaussianParams

    _setenv_policy(monkeypatch, "FG_RASTER_TAIL_BWD", "7,9")
    _setenv_policy(monkeypatch, "FG_RASTER_SPLIT_BWD", "2,1")
    _setenv_policy(monkeypatch, "FG_RASTER_LIVE", live)
    sc = synthetic_scene(30000, 320, 192, n_views=4, seed=31)
    fp = FlatGaussianParams.from_scene(sc, DEV)
    ref = FlatGaussianParams.from_scene(sc, DEV)
    g = GraphedRaster(fp, sc.width, sc.height, sh_degree=3)
    gen = torch.Generator().manual_seed(2)
    for v in [0, 1, 2, 1]:
        vm, K = sc.viewmats[v : v + 1].to(DEV), sc.Ks[v : v + 1].to(DEV)
        vr = torch.randn(1, sc.height, sc.width, 3, generator=gen).to(DEV)
        r, a, overflow = g.step(vm, K, vr)
        snap = fp.flat_grad.clone()
        torch.cuda.synchronize()
        with ref.direct_grads():
            r0, a0, _ = rasterization(*ref.raster_inputs(), vm, K, sc.width, sc.height, sh_degree=3, packed=False,
                                      absgrad=True)  # fmt: skip
            r0.backward(vr)
        assert not overflow and torch.equal(r, r0.detach())
        assert rel_l2(snap, ref.flat_grad) < 1e-5
        del r0, a0


def test_graphed_step_optimises_the_models_own_objective():
    """config.ssim_lambda != 0.2: the graphed branch of harness.train_step and the eager one compute the
    same loss -- (1 - l) L1 + l (1 - SSIM) with the model's l -- so its gradients equal the eager gradients of
    get_loss_dict, and differ from those of the hard-coded 0.2."""
    import copy

    from freegaussian_amd import harness as Hn
    from freegaussian_amd.graphed import GraphedModelStep

    grads = {}
    for kind in ("eager", "graphed", "eager_0.2"):
        torch.manual_seed(5)
        model, _, cam = _model_and_camera(n=4000, W=320, H=192, step=1, training=True)
        model.config.ssim_lambda = 0.6
        model.config.background_color = "black"
        gt = torch.rand(192, 320, 3, generator=torch.Generator().manual_seed(9)).to(DEV)
        model.step_cb(1)
        if kind == "graphed":
            g = GraphedModelStep(model)
            assert g.applicable(cam)
            g.step(copy.deepcopy(cam), gt)
        else:
            out = model.get_outputs(copy.deepcopy(cam))
            if kind == "eager":
                model.get_loss_dict(out, {"image": gt})["main_loss"].backward()
            else:
                Hn.main_loss(out["rgb"], gt).backward()
        grads[kind] = {k: v.grad.detach().clone() for k, v in model.gauss_params.items()}
    for k in ("means", "features_dc", "opacities", "scales"):
        assert rel_l2(grads["graphed"][k], grads["eager"][k]) < 1e-5, k
    a, b = grads["eager_0.2"]["features_dc"], grads["eager"]["features_dc"]
    assert float((a - b).norm() / b.norm()) > 1e-2  # (not a parity comparison: the other lambda is another objective)


def test_graphed_model_step_trains_like_the_eager_step_through_refinements():
    """harness.train_step(graphed=GraphedModelStep(...)): get_outputs + loss + backward replayed as one hipGraph
    at a launch-bound size, against the eager step: 200 steps with a refinement at step 100 (the Gaussian set is
    re-allocated: re-capture), an SH degree change every 80 steps (another shape), a list-capacity overflow forced
    half way, and the deform net switching on at step 150 (from there the step is eager again: torch reductions,
    which an MLP backward contains, are not replay-safe on this stack -- graphed.GraphedModelStep).  One step from the same state: the replay's gradients equal the eager step's up
    to the order of float atomics (1e-5).  Over many steps that noise is amplified without bound -- Adam turns a
    gradient that is zero up to rounding (the radial component of a quaternion) into +-lr steps, the L1 loss's
    sign() flips on pixels that sit on their target -- so the run is held to the same trajectory instead (counts
    within 3%, losses within 10%): the comparison the eager-vs-torch densification test makes."""
    import copy

    from freegaussian_amd import harness as Hn
    from freegaussian_amd.graphed import GraphedModelStep

    runs = []
    for use_graph in (False, True):
        torch.manual_seed(123)
        model, _, cam = _model_and_camera(n=6000, W=320, H=192, step=1, training=True)
        c = model.config
        c.warm_up, c.refine_start, c.refine_every, c.reset_alpha_every = 150, 50, 100, 30
        c.densify_grad_thresh, c.stop_screen_size_at, c.sh_degree_interval = 2e-4, 0, 80
        with torch.no_grad():
            model.gauss_params["opacities"].copy_(torch.randn(6000, 1, generator=torch.Generator().manual_seed(3)).to(DEV) * 1.5)
        target = copy.deepcopy(model)
        with torch.no_grad():
            target.gauss_params["features_dc"].add_(0.3)
        target.eval()
        with torch.no_grad():
            gt = target.get_outputs(copy.deepcopy(cam))["rgb"].clamp(0, 1)
        opts = Hn.build_optimizers(model)
        g = GraphedModelStep(model, Hn.main_loss) if use_graph else None
        assert g is None or g.applicable(cam)
        # one step from the initial state, gradients only
        model.step_cb(1)
        if g is not None:
            g.step(copy.deepcopy(cam), gt)
        else:
            Hn.main_loss(model.get_outputs(copy.deepcopy(cam))["rgb"], model.get_gt_img(gt)).backward()
        first_grads = {k: v.grad.detach().clone() for k, v in model.gauss_params.items()}
        if g is None:
            for v in model.parameters():
                v.grad = None
        torch.manual_seed(7)
        hist, snap, snap3 = [], None, first_grads
        for i in range(1, 201):
            if g is not None and i == 60:
                g.capacity = 2000  # a graph whose list capacity is far too small: the replay's overflow flag must
                g._capture()  # trigger the redo (measure, capture with room, replay)
            hist.append(Hn.train_step(model, opts, copy.deepcopy(cam), gt, i, num_train_data=2, graphed=g))
            if i == 99:
                snap = {k: v.detach().clone() for k, v in model.gauss_params.items()}
        runs.append((hist, (snap3, snap), model, g))
    (h0, (t0, s0), m0, _), (h1, (t1, s1), m1, g) = runs
    # captures: the first shape, the forced small graph + its redo, SH degree 1 (+ the refinement at step 100 when it
    # changes the set);
    # replays: steps 1..149 (+ the first-step check, + the redo)
    assert 149 <= g.replays <= 153 and g.captures >= 4 and g.graph is None, (g.replays, g.captures)
    assert h0[98]["gaussian_count"] == h1[98]["gaussian_count"] == 6000
    for k in s0:
        # the first step's gradients: atomic order only.  (The scene's Gaussians are isotropic, so the gradient of
        # the rotations is zero up to rounding -- ~1e-13 of noise, which has no relative error to speak of.)
        assert float(t0[k].norm()) < 1e-9 or rel_l2(t1[k], t0[k]) < 1e-5, k
        assert s1[k].shape == s0[k].shape
    c0, c1 = [h["gaussian_count"] for h in h0], [h["gaussian_count"] for h in h1]
    assert c0[-1] != 6000 and len(set(c0)) >= 2  # the Gaussian set was rebuilt under the graph
    assert all(abs(a - b) <= 0.03 * b for a, b in zip(c1, c0)), (sorted(set(c0)), sorted(set(c1)))
    for i, (a, b) in enumerate(zip(h1, h0)):
        assert abs(a["loss"] - b["loss"]) <= 0.1 * abs(b["loss"]) + 1e-7, (i, a, b)
    # the deform net took part after the warm-up (step 150) in both runs -- eagerly
    assert all(p.grad is not None for p in m1.deform.parameters()) and not g.applicable(cam)


def test_partial_requires_grad_noncontiguous_and_half_inputs():
    """Boundary hygiene: inputs that are non-contiguous views, lower precision, or only partly
    differentiable behave like their contiguous fp32 counterparts."""
    sc = _scene(n=5000, w=160, h=96, seed=41)
    vm, K = sc.viewmats[:1].to(DEV), sc.Ks[:1].to(DEV)
    base = [x.to(DEV) for x in (sc.means, sc.quats, sc.scales, sc.opacities, sc.colors)]
    ref = [x.clone().requires_grad_(True) for x in base]
    r0, a0, _ = rasterization(*ref, vm, K, sc.width, sc.height, sh_degree=3, packed=False)
    (r0.sum() + a0.sum()).backward()
    # (1) non-contiguous: every input is a strided slice of a wider buffer
    wide = [torch.cat([x, x], dim=-1) if x.dim() > 1 else torch.stack([x, x], dim=-1) for x in base]
    nc = [w[..., : b.shape[-1]] if b.dim() > 1 else w[..., 0] for w, b in zip(wide, base)]
    assert not nc[0].is_contiguous() and not nc[3].is_contiguous()
    r1, a1, _ = rasterization(*nc, vm, K, sc.width, sc.height, sh_degree=3, packed=False)
    assert torch.equal(r1, r0.detach()) and torch.equal(a1, a0.detach())
    # (2) only the means are differentiable
    part = [base[0].clone().requires_grad_(True)] + [x.clone() for x in base[1:]]
    r2, a2, _ = rasterization(*part, vm, K, sc.width, sc.height, sh_degree=3, packed=False)
    (r2.sum() + a2.sum()).backward()
    assert rel_l2(part[0].grad, ref[0].grad) < 1e-5 and all(x.grad is None for x in part[1:])
    # (3) bf16 colours / fp16 opacities are promoted to fp32 at the boundary
    lo = [base[0], base[1], base[2], base[3].half(), base[4].bfloat16()]
    r3, a3, _ = rasterization(*lo, vm, K, sc.width, sc.height, sh_degree=3, packed=False)
    r4, a4, _ = rasterization(base[0], base[1], base[2], base[3].half().float(), base[4].bfloat16().float(), vm, K,
                              sc.width, sc.height, sh_degree=3, packed=False)  # fmt: skip
    assert torch.equal(r3, r4) and torch.equal(a3, a4)


# ------------------------------------------------------------------------------------------
# Randomised parity, in the gate (round 5): tests/fuzz_cases.py holds the case generator (the one of the committed sweeps,
# profiles/r0*_fuzz_parity.txt), the bar and the fp64 arbitration.  Sweep 11 is the one whose case 1 (`ED`, antialiased,
# packed, two Gaussians) the round-4 review singled out; its first 48 cases hold every small case that ever exceeded 1e-4
# on a single cotangent draw (1, 12, 17, 21, 23, 27, 29, 32, 33, 39, 46, 47).  The modes the reference's preprocess and
# eval calls use -- render_mode="ED", packed=True (preprocess/knn_gaussian.py:93-121, render_depth.py:99-113), "RGB+ED"
# (freegaussian_model.py:821-824, :884-888) -- are two thirds of the draws.
@pytest.mark.parametrize("case", range(48))
def test_randomised_parity_small(case):
    import fuzz_cases

    ok, msg = fuzz_cases.check(fuzz_cases.Case(11, case, big=False), device=DEV)
    print(msg)
    assert ok, msg


@pytest.mark.parametrize("case", range(8))
def test_randomised_parity_big(case):
    """640x360 ... 1920x1080, 3000 ... 40000 Gaussians: job lists, spans, strips, list shares, liveness, and on the second
    call of each shape the one-call-per-direction entry points."""
    import fuzz_cases

    ok, msg = fuzz_cases.check(fuzz_cases.Case(4, case, big=True), device=DEV)
    print(msg)
    assert ok, msg


@pytest.mark.gpu
def test_footprint_masks_only_where_they_pay():
    """fg_stbin_count reports the footprint rectangles' area beside the list length (count_out[14], ABI 9); a shape keeps
    its footprint masks only while list length / area -- what the masks keep -- stays below RasterContext.mask_keep_max.
    Round splats (the bench cloud: 0.88 kept) lose the masks after their first masked call and look again every 64th call;
    needles (0.43) keep them.  Same image in both states, bit for bit; the lists of the unmasked call are the rectangles'."""
    from freegaussian_amd.scenes import apply_layout

    def run(sc, ctx, n_calls):
        vm, K = sc.viewmats[:1].to(DEV), sc.Ks[:1].to(DEV)
        ins = [getattr(sc, k).to(DEV) for k in ("means", "quats", "scales", "opacities", "colors")]
        out = []
        for _ in range(n_calls):
            on = ctx.masks_on((torch.device(DEV, torch.cuda.current_device()), 60, 34))
            with torch.no_grad():
                r, a, info = rasterization(*ins, vm, K, 960, 540, sh_degree=3, packed=False, ctx=ctx)
            out.append((on, r.clone(), int(info["raster_flatten_ids"].numel())))
        return out

    round_ = synthetic_scene(200_000, 960, 540, n_views=1, sh_degree=3, seed=42)
    ctx = ops.RasterContext(env={"FG_MASK_KEEP_MAX": "0.8"})  # (the switch is off by default: RasterContext.mask_keep_max)
    got = run(round_, ctx, 80)
    lkey = next(iter(ctx.mask_keep))
    assert got[0][0] and 0.8 < ctx.mask_keep[lkey][0][0] < 0.97  # the first call is masked; round splats keep ~0.9
    # masked while the shape has not shown eight even scenes in a row, then off, with a look every 64th call
    flags = [on for on, _, _ in got]
    first_off = flags.index(False)
    assert 8 <= first_off <= 10 and not any(flags[first_off : first_off + 63]) and flags[first_off + 63] and not flags[first_off + 64]
    assert got[first_off][2] > got[0][2] and got[first_off + 63][2] == got[0][2]  # the unmasked lists are the rectangles'
    for _, r, _ in got[1:]:
        assert torch.equal(r, got[0][1])  # the masks drop only pairs no pixel takes
    always = run(round_, ops.RasterContext(env={"FG_EXACT_TILES": "always"}), 3)
    never = run(round_, ops.RasterContext(env={"FG_EXACT_TILES": "0"}), 3)
    assert all(on for on, _, _ in always) and not any(on for on, _, _ in never)
    assert always[2][2] == got[0][2] and never[2][2] == got[first_off][2]
    needles = apply_layout(synthetic_scene(200_000, 960, 540, n_views=1, sh_degree=3, seed=42), "needles:0.4:10")
    ctx2 = ops.RasterContext(env={"FG_MASK_KEEP_MAX": "0.8"})
    got2 = run(needles, ctx2, 6)
    dflt = run(round_, ops.RasterContext(env={}), 12)
    assert all(on for on, _, _ in dflt)  # default: masks stay on (the measured ratio is reported, not acted on)
    assert all(on for on, _, _ in got2) and max(next(iter(ctx2.mask_keep.values()))[0]) < 0.7


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(400, 240), (1000, 600), (1912, 1080)])
def test_interleaved_xcd_shares_on_odd_and_even_tile_grids(size):
    """fg_raster_config::balance_bands = 3: the image's 2 x 2-tile blocks dealt to the XCDs round-robin (what uneven shapes
    get since round 6).  Tile grids with an odd width and / or height (25 x 15, 63 x 38, 120 x 68): the edge blocks' missing
    tiles have no job, every real tile exactly one owner -- the image is the default policy's bit for bit, the gradients
    equal up to the order of float atomics; with list shares, finer thresholds and heavy tiles on top."""
    W, H = size
    sc = synthetic_scene(150_000, W, H, n_views=1, sh_degree=3, seed=21, log_scale_mean=math.log(0.02), focal=1200.0 * W / 1920.0)
    sc.means[:60_000] *= 0.15  # a cluster: long lists at the centre, uneven by any measure
    vm, K = sc.viewmats[:1].to(DEV), sc.Ks[:1].to(DEV)
    vr = torch.randn(1, H, W, 3, generator=torch.Generator().manual_seed(2)).to(DEV)
    ins = {k: getattr(sc, k).to(DEV) for k in ("means", "quats", "scales", "opacities", "colors")}

    def run(ctx, calls=3):
        for _ in range(calls):
            t = {k: v.clone().requires_grad_(True) for k, v in ins.items()}
            r, a, info = rasterization(*t.values(), vm, K, W, H, sh_degree=3, packed=False, absgrad=True, ctx=ctx)
            ((r * vr).sum() + a.sum()).backward()
        torch.cuda.synchronize()
        return r.detach(), a.detach(), {k: v.grad for k, v in t.items()}, info

    base = ops.RasterContext(env={"FG_UNEVEN_SPLIT_FWD": "0", "FG_HEAVY_TILES": "never"})
    if int(_lib.load().fg_raster_jobs_words(W, H, 16, base.cfg())) == 0:
        pytest.skip("classic launches at this size: no job lists")
    r0, a0, g0, _ = run(base)
    for env, policy in (({"FG_HEAVY_TILES": "never"}, ops.launch_policy(balance_bands=3)),
                        ({"FG_HEAVY_TILES": "never"}, ops.launch_policy(balance_bands=3, split4_fwd=8, split2_fwd=5, split4_bwd=20, split2_bwd=4)),
                        ({}, None)):  # (the host's own choice for this shape: uneven from its second call on)
        ctx = ops.RasterContext(env=env, policy=policy)
        r, a, g, info = run(ctx)
        if policy is None:
            assert ctx.uneven_shape(next(iter(ctx.uneven_left))) and ctx.cfg_variant(False, 0, False, True)[1][3] == (8, 5)
            assert rel_err(r, r0) < 2e-6 and rel_err(a, a0) < 2e-6  # (heavy tiles may be on: not the serial walk bit for bit)
        else:
            assert torch.equal(r, r0) and torch.equal(a, a0)
        diff = {k: rel_l2(g[k], g0[k]) for k in g0}
        assert all(v < 1e-5 for v in diff.values()), (env, diff)
