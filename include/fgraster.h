/*
 * fgraster.h -- C ABI of libfgraster.so, the MI355X (gfx950) Gaussian-raster hot path.
 *
 * This is the drop-in boundary for the ONE call the reference makes into native code:
 *     gsplat.rendering.rasterization(...)
 *         freegaussian/freegaussian_model.py:847-868          (stage 1 training / eval)
 *         freegaussian/freegaussian_control_model.py:158-179  (stage 2)
 *         preprocess/knn_gaussian.py:93-113, preprocess/render_depth.py:99,157,
 *         preprocess/render_color.py:93, preprocess/o3d_color_splat.py:188   (packed / "ED")
 * and for the flow-derivative math of preprocess/epipolar_flow.py:274-309 and
 * docs/index.html:256-300.  The Python binding that calls these symbols is
 * freegaussian_amd/_lib.py (ctypes); INTEGRATION.md shows the stub a maintainer adds.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the comment says "host";
 *   - the library allocates nothing, keeps no global state, reads no environment variable and never
 *     synchronises (launch policy comes in through fg_raster_config): the
 *     caller owns every buffer (scratch sizes from the *_workspace_bytes queries) and every
 *     call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *   - re-entrant across streams; return 0 on success, a negative FG_ERR_* code otherwise;
 *     nothing is thrown across the ABI;
 *   - all floating data is fp32, row-major, contiguous; "radii" is int32 with >0 <=> visible;
 *   - one camera per call (the reference asserts camera.shape[0]==1,
 *     freegaussian_model.py:773); views are sharded over processes, not batched.
 */
#ifndef FGRASTER_H
#define FGRASTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* fg_stream_t;

#define FG_OK 0
#define FG_ERR_INVALID_ARG (-1)
#define FG_ERR_LAUNCH (-2)
#define FG_ERR_WORKSPACE (-3)
#define FG_ERR_UNSUPPORTED (-4)

#define FG_MAX_CHANNELS 8    /* composited feature channels per splat (RGB, depth, flow, ...) */
#define FG_SPLAT_FLOATS 16   /* one 64-byte record per Gaussian, see fg_pack_splats */
#define FG_SH_JAC_FLOATS 10  /* per-Gaussian note of the SH colour for the backward, see fg_preprocess_fwd */
#define FG_ABI_VERSION 9

int fg_abi_version(void);
const char* fg_error_string(int code);

/* ---- K1: projection (replaces the fully-fused projection stage implied by the kwargs
 * viewmats/Ks/near_plane/far_plane/rasterize_mode at freegaussian_model.py:847-865) --------
 * means[N,3] quats[N,4](wxyz, any norm) scales[N,3]; viewmat[16] world->camera (OpenCV axes,
 * as built by freegaussian/utils.py:162-179); K[9].  Outputs (culled Gaussians: 0):
 * radii[N] means2d[N,2] depths[N] conics[N,3] compensations[N](nullable) and
 * tiles_touched[N] = number of tile_size x tile_size tiles in the splat's bounding rectangle. */
int fg_project_fwd(int N, const float* means, const float* quats, const float* scales,
                   const float* viewmat, const float* K, int width, int height, float eps2d,
                   float near_plane, float far_plane, float radius_clip, int tile_size,
                   int32_t* radii, float* means2d, float* depths, float* conics,
                   float* compensations, int32_t* tiles_touched, fg_stream_t stream);

/* Backward of K1 (autograd of the call at freegaussian_model.py:847; viewmat gradients are
 * not produced: camera_optimizer mode is "off", freegaussian_model.py:120).
 * v_compensations / compensations nullable together.  Writes v_means[N,3] v_quats[N,4]
 * v_scales[N,3] (zeros where radii==0). */
int fg_project_bwd(int N, const float* means, const float* quats, const float* scales,
                   const float* viewmat, const float* K, int width, int height, float eps2d,
                   const int32_t* radii, const float* conics, const float* compensations,
                   const float* v_means2d, const float* v_depths, const float* v_conics,
                   const float* v_compensations, float* v_means, float* v_quats,
                   float* v_scales, fg_stream_t stream);

/* ---- K2: spherical harmonics (sh_degree / colors[N,K,3] kwargs, freegaussian_model.py:801,
 * :826-830).  colour = max(SH(dir) + 0.5, 0), dir = mean - camera position (derived from
 * viewmat).  coeffs[N,k_stored,3]; degree 0..3; only radii>0 rows are shaded (others 0). */
int fg_sh_fwd(int N, int degree, int k_stored, const float* means, const float* viewmat,
              const float* coeffs, const int32_t* radii, float* colors, fg_stream_t stream);
/* v_coeffs[N,k_stored,3] is written densely (sparse_grad=False, freegaussian_model.py:863);
 * v_means[N,3] (nullable) receives the view-direction gradient. */
int fg_sh_bwd(int N, int degree, int k_stored, const float* means, const float* viewmat,
              const float* coeffs, const int32_t* radii, const float* colors,
              const float* v_colors, float* v_coeffs, float* v_means, fg_stream_t stream);

/* ---- K3: tile binning (tile_size=16, freegaussian_model.py:806) --------------------------
 * fg_scan_tiles: inclusive prefix sum, cum_tiles[N] int64; cum_tiles[N-1] = I, the number of
 * (Gaussian, tile) intersections, which the host reads to size the lists. */
size_t fg_scan_workspace_bytes(int N);
int fg_scan_tiles(int N, const int32_t* tiles_touched, int64_t* cum_tiles, void* workspace,
                  size_t workspace_bytes, fg_stream_t stream);
/* isect_ids[I]  = (tile_id << 32) | float32 bits of depth;  flatten_ids[I] = Gaussian id.
 * Emission order: Gaussian-major, row-major over the tile rectangle. */
int fg_tile_bin(int N, const float* means2d, const int32_t* radii, const float* depths,
                const int64_t* cum_tiles, int tile_size, int tile_w, int tile_h,
                int64_t* isect_ids, int32_t* flatten_ids, fg_stream_t stream);

/* ---- K4: stable LSD radix sort of (key, value) pairs on key bits [0, end_bit), in place
 * (workspace holds the ping-pong copies and histograms), then per-tile ranges:
 * tile_offsets[n_tiles+1], tile t owns sorted entries [tile_offsets[t], tile_offsets[t+1]). */
size_t fg_sort_workspace_bytes(int64_t n);
int fg_sort_pairs(int64_t n, int64_t* keys, int32_t* vals, int end_bit, void* workspace,
                  size_t workspace_bytes, fg_stream_t stream);
int fg_tile_ranges(int64_t n, const int64_t* sorted_keys, int n_tiles, int32_t* tile_offsets,
                   fg_stream_t stream);

/* 32-bit-key variant of fg_sort_pairs (same kernels, half the key traffic). */
size_t fg_sort32_workspace_bytes(int64_t n);
int fg_sort_pairs32(int64_t n, uint32_t* keys, int32_t* vals, int end_bit, void* workspace,
                    size_t workspace_bytes, fg_stream_t stream);

/* ---- K3+K4, depth-first form (what rasterization() uses; identical final lists) -------------
 * Instead of sorting I 45-bit (tile|depth) keys, sort the N Gaussians by depth once
 * (fg_bin_prepare: order[N] = ids by ascending depth bits, culled last, stable; cum_tiles[k] =
 * inclusive tile count of order[0..k]), emit (tile, id) pairs in that order, and stably sort
 * them on the ceil(log2 T) tile bits only (fg_bin_emit_sort).  A stable sort by tile of a
 * depth-ordered sequence IS the (tile, depth, id) order of the 64-bit-key sort, so
 * flatten_ids / tile_offsets are bit-identical to fg_tile_bin + fg_sort_pairs + fg_tile_ranges,
 * at ~1/7 of the sort traffic.  The host reads cum_tiles[N-1] (= I) between the two calls.
 * fg_isect_keys rebuilds the reference-style 64-bit keys (tile << 32 | depth bits) on demand. */
size_t fg_bin_prepare_workspace_bytes(int N);
int fg_bin_prepare(int N, const float* depths, const int32_t* radii, const int32_t* tiles_touched,
                   int32_t* order, int64_t* cum_tiles, void* workspace, size_t workspace_bytes,
                   fg_stream_t stream);
/* The same, but the tile counts come from the tile rectangles, computed here from means2d / radii
 * with the arithmetic of the projection pass (width x height == tiles_touched), and
 * rects_sorted[N] receives every Gaussian's rectangle (x0 | y0 << 10 | width << 20) IN DEPTH ORDER:
 * handed to fg_bin_emit_sort[_capacity], the emission reads it coalesced instead of gathering
 * radii / means2d by id (35 -> 20 us at 1M Gaussians).  Needs tile_w, tile_h <= 1023. */
int fg_bin_prepare_rects(int N, const float* depths, const int32_t* radii, const float* means2d,
                         int tile_size, int tile_w, int tile_h, int32_t* order, int64_t* cum_tiles,
                         int32_t* rects_sorted, void* workspace, size_t workspace_bytes,
                         fg_stream_t stream);
/* The same from what fg_preprocess_fwd / fg_preprocess_raw_fwd already wrote: depth_keys[N] (sorted
 * IN PLACE here -- a scratch array from the caller's point of view) and tile_rects[N,2] (read
 * only).  No key / rectangle kernel runs and the first radix pass numbers the values itself.  With
 * the preprocess pass's FOOTPRINT rectangles the lists hold only (splat, tile) pairs in which the
 * splat can reach alpha >= 1/255 -- the reference's lists minus dead entries, same order.
 * count_out (nullable): one more copy of the list length cum_tiles[N-1], stored with system scope --
 * pass the device-visible address of pinned host memory and the host has the count after an event
 * wait, with no copy launch in the stream (the reference reads it with a blocking .item()). */
int fg_bin_prepare_keys(int N, uint32_t* depth_keys, const int32_t* tile_rects, int32_t* order,
                        int64_t* cum_tiles, int32_t* rects_sorted, int64_t* count_out, void* workspace,
                        size_t workspace_bytes, fg_stream_t stream);
/* ---- K3+K4, supertile form (what rasterization() runs when the fused preprocess pass supplied keys and
 * rectangles; identical lists) -------------------------------------------------------------------------
 * Count per tile and per supertile (2 x 2 tiles) -> scans -> one 8-byte element (depth bits << 32 | id << 4 |
 * which of the four tiles) scattered per (Gaussian, supertile) pair into the supertile's segment -> every
 * segment sorted ONCE by (depth bits, Gaussian id) in LDS and the four tile lists read off the sorted run:
 * 6 launches instead of the 26 of fg_bin_prepare_keys + fg_bin_emit_sort, 2.5x fewer scattered and sorted
 * elements than (Gaussian, tile) pairs on the 1M / 1080p scene (csrc/stbin.hip).
 * tile_rects / depth_keys: the optional outputs of fg_preprocess_fwd.  fg_stbin_count writes
 * tile_offsets[T + 1] (exact, independent of any capacity) and, if count_out is not NULL, FIVE words of a block of
 * SIXTEEN int64 (ABI 9; a block of four before) with system scope (pinned host memory): count_out[0] the list length,
 * count_out[2] the longest tile list (a host turns fg_raster_config::heavy_tiles on from it), count_out[1] the longest
 * supertile segment, count_out[3] the number of segments of more than 3072 elements, and count_out[14] the summed AREA of
 * the footprint rectangles in tiles -- what the list length would be without tile_masks: a host keeps the masks for an
 * image size only while list length / area says they drop enough (ops.RasterContext.masks_on: below 0.8).  (Words 4..13
 * and 15 belong to fg_stbin_fill_jobs' ckpt_need_out and fg_raster_jobs_fwd's walk_out when the host hands them the same
 * block, as ops.py does.)
 * (segments beyond 7936 elements are sorted by one workgroup through global memory -- correct, slow -- unless
 * fg_stbin_fill is called with FG_STBIN_LONG_SEGMENTS, see below).  fg_stbin_fill writes flatten_ids[0 .. tile_offsets[T]) and
 * list_offsets[T + 1] = tile_offsets -- or, when the list is longer than `capacity`, no ids at all and
 * list_offsets = 0 (empty lists: consumers enqueued speculatively behind the call walk nothing; the host then
 * repeats the call with exact buffers, the count workspace stays valid).  Consumers of flatten_ids read
 * list_offsets, not tile_offsets.  N < 2^28; fg_stbin_supported = 0 for tile grids wider than ~2700 tiles
 * (callers then take fg_bin_prepare_keys + fg_bin_emit_sort).
 * Replaces the binning + sort implied by tile_size=16 at freegaussian_model.py:806,857. */
int fg_stbin_supported(int N, int tile_w, int tile_h);
size_t fg_stbin_count_workspace_bytes(int N, int tile_w, int tile_h);
/* tile_masks (ABI 8; nullable, the optional output of fg_preprocess_fwd): per Gaussian which blocks of its footprint
 * rectangle the ELLIPSE alpha >= 1/255 reaches -- the rectangle's w x h tiles in at most 8 x 8 blocks of b x b tiles
 * (b = 1 up to 8 x 8 tiles, then the next power of two with ceil(max(w, h) / b) <= 8), bit 8 by + bx.  Counted and
 * scattered are the set blocks' tiles only: a needle lying diagonally across a 12 x 12-tile rectangle enters 20 lists, not
 * 144 (30 % needles of axis ratio 10 in the 1M / 1080p scene: 57 % fewer list entries, profiles/r05_exact_tiles.md); the
 * lists stay an order-preserving subsequence of the reference's and every dropped entry is one no pixel of the tile would
 * have taken.  The SAME array must be given to fg_stbin_count and fg_stbin_fill (the scatter drops exactly the pairs the
 * count did not count).  NULL: whole rectangles, as before ABI 8. */
int fg_stbin_count(int N, const int32_t* tile_rects, const uint64_t* tile_masks, int tile_w, int tile_h,
                   int32_t* tile_offsets, int64_t* count_out, void* workspace, size_t workspace_bytes, fg_stream_t stream);
size_t fg_stbin_fill_workspace_bytes(int64_t capacity);
/* flags (ABI 7): FG_STBIN_LONG_SEGMENTS -- supertile segments beyond the small-segment launch's LDS capacity (3072
 * elements; a dense cluster: thousands to hundreds of thousands of splats over one 32 x 32-pixel supertile) are cut
 * into buckets of ~1536 elements by a multi-workgroup sample sort (splitters from a sorted sample inside the
 * large-segment launch, then a count, a scatter, a bucket-sort and an overflow launch) instead of being sorted one
 * segment per 1024-thread workgroup in LDS (up to 7936 elements) or by ONE workgroup through global memory (beyond).
 * Same lists bit for bit either way; a host sets the flag for shapes whose earlier calls reported a segment beyond
 * 7936 elements (count_out[1]) or more than a handful beyond 3072 (count_out[3]) -- ops.bin_tiles does -- on scenes
 * without such segments the flag only costs three empty launches (ABI 9: count + scatter of the long elements in one pass
 * into per-bucket slabs of the workspace; four launches before). */
#define FG_STBIN_LONG_SEGMENTS 1
#define FG_STEP_NO_FOOTPRINT_MASKS 2 /* (fg_step_desc::flags only) */
/* (ABI 9, with FG_STBIN_LONG_SEGMENTS; a TEST hook: a bucket of the sample sort counts as having outgrown its slab from 1600
 * elements instead of 3072 -- it is gathered again from its segment and sorted by a larger LDS sort -- and, in even
 * supertiles, a segment with a bucket beyond 1728 elements instead of 7936 goes through global memory whole, so that both
 * paths behind an outgrown slab -- a 1e-6 event per bucket otherwise -- run in tests; same lists) */
#define FG_STBIN_TEST_SMALL_SLABS 4
int fg_stbin_fill(int N, const uint32_t* depth_keys, const int32_t* tile_rects, const uint64_t* tile_masks, int tile_w,
                  int tile_h, int64_t capacity, const int32_t* tile_offsets, const void* count_workspace,
                  int32_t* flatten_ids, int32_t* list_offsets, void* workspace, size_t workspace_bytes,
                  int flags, fg_stream_t stream);
/* (fg_stbin_fill_jobs, with the job lists further down: this call and fg_raster_build_jobs in the same launches) */

/* tile_keys may be NULL in both emit entry points when the caller does not need the keys (up to
 * 65536 tiles): they are then kept as 16-bit values inside the workspace -- 34 instead of 48 bytes
 * of traffic per intersection over emission + the two sort passes. */
size_t fg_bin_emit_workspace_bytes(int64_t n_isects);
/* rects_sorted: from fg_bin_prepare_rects, or NULL (rectangles are then recomputed from means2d /
 * radii, which may be NULL otherwise). */
int fg_bin_emit_sort(int N, int64_t n_isects, const float* means2d, const int32_t* radii,
                     const int32_t* order, const int64_t* cum_tiles, const int32_t* rects_sorted,
                     int tile_size, int tile_w, int tile_h, uint32_t* tile_keys, int32_t* flatten_ids,
                     int32_t* tile_offsets, void* workspace, size_t workspace_bytes, fg_stream_t stream);
/* fg_bin_emit_sort without the host round trip for the count: the buffers hold `capacity`
 * entries (tile_keys[capacity], flatten_ids[capacity], workspace for `capacity`), the number of
 * intersections is read ON THE DEVICE from cum_tiles[N-1].  When that count exceeds the capacity
 * the lists are truncated and invalid: the caller compares its own (asynchronous) readback of
 * cum_tiles[N-1] with the capacity and repeats the call with exact buffers in that case. */
int fg_bin_emit_sort_capacity(int N, int64_t capacity, const float* means2d, const int32_t* radii,
                              const int32_t* order, const int64_t* cum_tiles,
                              const int32_t* rects_sorted, int tile_size, int tile_w, int tile_h,
                              uint32_t* tile_keys, int32_t* flatten_ids, int32_t* tile_offsets,
                              void* workspace, size_t workspace_bytes, fg_stream_t stream);
int fg_isect_keys(int64_t n_isects, const uint32_t* tile_keys, const int32_t* flatten_ids,
                  const float* depths, int64_t* isect_ids, fg_stream_t stream);

/* ---- K5/K6: front-to-back alpha compositing over 16x16 tiles -----------------------------
 * fg_pack_splats builds one 64-byte record per Gaussian:
 *   [x, y, opacity, conic_a, conic_b, conic_c, f0..f(C-1), 0...]      C <= FG_MAX_CHANNELS
 * features[N,C] are the composited channels (RGB | RGB+depth | depth | +flow ...). */
int fg_pack_splats(int N, int channels, const float* means2d, const float* conics,
                   const float* opacities, const float* features, float* splats,
                   fg_stream_t stream);
/* LAUNCH POLICY of the raster kernels (ABI version 3: rounds 1-2 read FG_RASTER_* environment variables
 * inside the library).  Every raster entry point takes a `const fg_raster_config*`; NULL = the defaults
 * measured on MI355X (csrc/raster.hip), and so is every field left at its fg_raster_config_init value.
 * Results do not depend on the policy (same images; gradients up to float summation order) -- it decides
 * how the work is cut into jobs.  The SAME config must go to all calls of one image (job-list words, build,
 * forward, checkpoint size, backward).  The Python host fills it from the FG_RASTER_* variables it reads
 * (freegaussian_amd/ops.py::LaunchPolicy); a non-Python host fills the struct directly. */
typedef struct fg_raster_config {
  int32_t size;            /* sizeof(fg_raster_config) of the caller's build (set by fg_raster_config_init) */
  int32_t ppt_fwd;         /* pixels per lane of the forward: 0 = by tile count; 1 | 2 | 4 = forced (classic launch) */
  int32_t ppt_bwd;         /* the same for the backward */
  int32_t tile_order;      /* classic launches, workgroup -> tile map: -1 = default (2: XCD row bands walked
                              column-major); 0 rows, 1 bands, 2 cols, 3 split, 4 / 5 rectangles */
  int32_t bands_nx;        /* mixed launches: XCD shares as nx x (8 / nx) rectangles; 0 / 1 = row bands (default) */
  int32_t tail4_fwd;       /* forward: the last tail4 tiles of every XCD's sequence as four single-strip jobs ... */
  int32_t tail2_fwd;       /* ... the tail2 tiles before them as two two-strip jobs; -1 = defaults; 0,0 = classic launch */
  int32_t tail4_bwd;
  int32_t tail2_bwd;
  int32_t split4_fwd;      /* content split: a tile longer than total * split4 / 65536 becomes four jobs ... */
  int32_t split2_fwd;      /* ... longer than total * split2 / 65536 two; -1 = defaults, 0 = off */
  int32_t split4_bwd;
  int32_t split2_bwd;
  int32_t use_liveness;    /* 0: the backward ignores the forward's liveness words (A/B); default 1 */
  int32_t seg_parts;       /* list shares per split tile of the backward (0 / 1 = off); -1 = default (5; 6 below 5000 tiles) */
  int32_t seg_tail;        /* tiles per XCD, at the end of its sequence, whose lists are split (0 = every tile); -1 = default */
  int32_t seg_parts2;      /* graded tail: the last seg_tail2 tiles get seg_parts2 shares; -1 = default (off) */
  int32_t seg_tail2;
  int32_t debug_only_xcd;  /* measurement hooks of the classic launches: only this XCD's workgroups work (-1 = off) */
  int32_t debug_k_mod;     /* ... only every m-th tile of each XCD (0 = off) */
  int32_t balance_bands;   /* job lists (ABI 7): the XCDs' shares of the tile grid are SPANS of the row-major tile sequence
                              with equal numbers of tiles (a span may begin and end inside a row; walked column-major) --
                              or, when the heaviest such share would be more than 15% above the mean, bands of whole tile
                              rows holding equal shares of the tiles' expected cost (list lengths, capped): a cluster of
                              splats under one band no longer sets the launch time; 0 = equal numbers of ROWS always
                              (rounds 1-3: 8 or 9 of a 1080p frame's 68); 2 = equal numbers of tiles always, the costs are
                              not looked at and lists and grids are sized for equal shares (for a host that knows the
                              scene is even: fg_stbin_count's count_out[2] against the mean list; 5-10 us less per step);
                              3 (ABI 8, round 6) = INTERLEAVED: the image's 2 x 2-tile blocks dealt to the XCDs round-robin in
                              raster order -- every XCD a uniform sample of the image, balanced on any content without a
                              cost model (what a host picks for uneven scenes; lists and grids sized as for the cost bands);
                              -1 / 1 = default; p >= 100: the threshold in percent of the mean (100 = always by cost) */
  int32_t heavy_tiles;     /* forward, job lists + list segments (ABI 7): a tile whose list is longer than this many entries
                              is walked serially for its first 2048 entries only; if pixels are still open there, the rest
                              of the list is composited by MANY jobs over shares of it -- every 64-entry batch by itself
                              -- and one combining pass per strip (two more launches) instead of four serial walks; the
                              backward gives such a tile up to 64 shares.  For scenes with unsaturated lists of thousands
                              of entries (a host turns it on when fg_stbin_count's count_out[2] says so); not bit-identical
                              to the serial walk (1e-7 relative).  <= 0 = off (default); values below 2560 mean 2560
                              (heavy_wide, below: the one-launch form that replaced this in round 5, from 256 entries) */
  int32_t seg_slots;       /* list segments, job lists (ABI 7): COMPACT checkpoint slots -- the buffer of
                              fg_raster_seg_ckpt_floats holds this many slots (4352 B each; rounded up to a multiple of 8)
                              instead of one per 64 entries of the list's capacity: only the tiles the backward may cut
                              into shares own slots, ceil(list length / 64) each, granted by the list build per XCD band
                              (an eighth of the slots each) from the end of the band's sequence while they last; a tile
                              without slots runs as one job, without checkpoints -- slower, never wrong.  What the bands
                              would take together is reported by fg_stbin_fill_jobs (ckpt_need_out): size the next call's
                              buffer as 8 x the largest word + a margin.  The SAME value must reach the list build, the size
                              query and both raster calls.  <= 0 = off (default): a slot per 64 entries of every tile */
  int32_t prio_fwd;        /* job lists / mixed launches (ABI 7): issue priority of a job's wavefront by its expected length,
                              lo | hi << 16 in percent of the launch's mean single-strip job (list length x the job kind's
                              cost per entry): above lo the wavefront runs at priority 2, above hi at 3 (s_setprio) -- the
                              launch's longest jobs, which start first and set its end, get through sooner; -1 = default
                              (forward 250 | 350 << 16, backward 120 | 160 << 16), 0 = off */
  int32_t prio_bwd;
  int32_t heavy_wide;      /* forward, heavy tiles (ABI 8): 1 / -1 (default) = the four strip jobs of a heavy tile walk its list's
                              first 512 entries; a strip with pixels still open there continues as ONE job of a 16-wavefront
                              workgroup in a launch behind the main one: the list in rounds of 16 64-entry batches, every
                              wavefront its batch by itself, the batches folded in LDS, a batch in which a pixel may stop
                              walked again from the true state; the job ends with the round in which its last pixel stops.
                              heavy_tiles may then be as low as 768.  0 = round 4's form (2048 entries serially, then local
                              jobs and combine jobs in two launches; heavy_tiles >= 2560) */
  int32_t seg_fine;        /* list segments (ABI 8): the checkpoint grid -- the forward leaves a checkpoint in front of every 64th
                              entry of a tile's list for the list's first seg_fine entries and in front of every 128th behind
                              them (a checkpoint slot each; the backward's shares are cut at those entries): a list of the
                              tail needs the fine grid for its five shares, a list of thousands has segments to spare.
                              -1 = default (640), 0 = every 64th throughout (rounds 2-4; also with heavy_wide = 0); rounded up
                              to a multiple of 64.  The SAME value must reach the list build, the size function and both raster
                              calls. */
} fg_raster_config;
void fg_raster_config_init(fg_raster_config* config);

/* render[H,W,C] alphas[H,W] last_ids[H,W] (index into the sorted list of the last splat that
 * contributed; tile start - 1 if none).  No background: the caller composites it
 * (freegaussian_model.py:875-877). */
int fg_raster_fwd(int channels, int width, int height, int tile_size, const float* splats,
                  const int32_t* tile_offsets, const int32_t* flatten_ids, float* render,
                  float* alphas, int32_t* last_ids, const fg_raster_config* config,
                    fg_stream_t stream);
/* v_alphas may be NULL (no gradient on alpha).
 * v_splats[N,16] must be ZEROED by the caller; per-Gaussian gradients are accumulated as
 *   [v_x, v_y, v_opacity, v_conic_a, v_conic_b, v_conic_c, |v_x|, |v_y|, v_f0..v_f(C-1)]
 * (|v_x|,|v_y| = absgrad, freegaussian_model.py:864 / :377). */
int fg_raster_bwd(int channels, int width, int height, int tile_size, const float* splats,
                  const int32_t* tile_offsets, const int32_t* flatten_ids,
                  const float* alphas, const int32_t* last_ids, const float* v_render,
                  const float* v_alphas, float* v_splats, const fg_raster_config* config,
                    fg_stream_t stream);
/* The same kernels with the model's post-composite O1 folded in (SURVEY.md section 8f row 3;
 * freegaussian_model.py:875-877 `rgb = clamp(render[..., :3] + (1 - alpha) * background, 0, 1)`):
 *   image[c] = render[c] + (1 - alpha) * background[c]   (background[C], nullable = none),
 * the first n_clamp channels clamped to [0,1].  clamp_mask[H,W] (uint8, required when
 * n_clamp > 0) receives per pixel bit c = "channel c was strictly outside [0,1]"; the backward
 * takes it back, zeroes those channels of v_image and folds -sum_c v_image[c]*background[c]
 * into the alpha gradient, so that v_image / v_alphas are the gradients of `rgb` / `alpha`. */
int fg_raster_composite_fwd(int channels, int width, int height, int tile_size, const float* splats,
                            const int32_t* tile_offsets, const int32_t* flatten_ids,
                            const float* background, int n_clamp, float* image, float* alphas,
                            int32_t* last_ids, uint8_t* clamp_mask, const fg_raster_config* config,
                    fg_stream_t stream);
int fg_raster_composite_bwd(int channels, int width, int height, int tile_size, const float* splats,
                            const int32_t* tile_offsets, const int32_t* flatten_ids,
                            const float* background, int n_clamp, const uint8_t* clamp_mask,
                            const float* alphas, const int32_t* last_ids, const float* v_image,
                            const float* v_alphas, float* v_splats, const fg_raster_config* config,
                    fg_stream_t stream);
/* ---- Job lists for the raster launches ---------------------------------------------------------
 * At 200 tiles and more the library runs the raster kernels as one wavefront per JOB: a whole tile
 * (4 pixels per lane), half a tile or a quarter.  Without a list the job sizes depend on the tile's
 * position only (the end of every XCD's tile sequence is split); with a list they also depend on
 * the tile's list length (tiles far longer than the mean are split wherever they are), which is
 * what non-uniform scenes need (scripts/clustered_check.py).  Caller-allocated like everything else:
 *   words = fg_raster_jobs_words(...)   int32 words of ONE list; 0 = the library would not use lists
 *                                        for this size / config (call the plain entry points)
 *   fg_raster_build_jobs(...)            one small launch, after tile_offsets exist; either list nullable
 *   fg_raster_jobs_fwd / _bwd            fg_raster_composite_fwd / _bwd with a list (jobs == NULL:
 *                                        identical to those)
 * LIST SEGMENTS of the backward (3 channels): floats = fg_raster_seg_ckpt_floats(...) (0 = off for
 * this size / config); seg_ckpt[floats], uninitialised, goes to BOTH calls and `image` (the
 * forward's output) to the backward: the forward leaves every pixel's compositing state at every
 * 64th entry of a tile's list (only for the tiles the backward's job list splits: hand both calls the
 * lists of ONE fg_raster_build_jobs call), the last list index each (tile, strip) used and every
 * pixel's exact final transmittance; the backward runs several jobs per tile, each over its share of
 * the list, instead of one serial walk.  Same gradients up to float summation order (a share job
 * reproduces the reference's T_final = 1 - alpha rounding per pixel).  The buffer is opaque.  NULL = off.
 * fg_raster_build_jobs(bwd_list_shares = 1) must then have built the backward's list (its entries
 * are (tile, part, parts) instead of (tile, strip)).
 * LIVENESS (any channel count): live_words[n_isects] uint32, uninitialised, to BOTH calls: the
 * forward notes per (list entry, 4-row strip) whether any pixel took the entry (byte s of word i =
 * strip s of entry i), and the backward evaluates exactly those pairs instead of re-testing every
 * strip of every entry -- same gradients, about a third fewer vector issue cycles.  NULL = off.
 * ZERO FILL IN PASSING: zero_buf[zero_floats] (nullable) is zero-filled by the forward call -- meant
 * for the v_splats array of the coming fg_raster_jobs_bwd, which accumulates with atomics: the forward
 * launch leaves the memory pipe mostly idle, a separate fill costs ~10 us and a launch boundary. */
int64_t fg_raster_jobs_words(int width, int height, int tile_size, const fg_raster_config* config);
int fg_raster_build_jobs(int width, int height, int tile_size, const int32_t* tile_offsets,
                         int32_t* jobs_fwd, int32_t* jobs_bwd, int bwd_list_shares, const fg_raster_config* config,
                    fg_stream_t stream);
/* fg_stbin_fill + fg_raster_build_jobs in the same launches: eight extra workgroups at the end of the
 * scatter launch build the forward's job list from tile_offsets beside the scatter, eight at the head of the
 * small-segment sort launch the backward's -- no launch
 * between the sorted lists and fg_raster_jobs_fwd (10 us of a 0.75 ms step on the 1M / 1080p scene).  width, height,
 * tile_size must give tile_w x tile_h; jobs_fwd / jobs_bwd / bwd_list_shares / config as for fg_raster_build_jobs
 * (both lists NULL: identical to fg_stbin_fill).  The lists depend on tile_offsets only: they stay valid when the
 * call is repeated with a larger capacity.
 * ckpt_need_out (nullable; int64[9], pinned host memory or device memory, written by the launch): with
 * bwd_list_shares and a forward list, word x < 8 = the checkpoint slots the tiles of XCD x's band that the backward may
 * split would take together (whether or not they got them; whatever fg_raster_config::seg_slots is): 8 x the largest
 * word, + a margin, is the seg_slots that leaves no tile without.  Word 8 (whenever the forward list is built): what the
 * cost pass over the XCDs' shares decided -- 1 = bands balanced by cost, 0 = the equal spans stood, -1 = it did not run
 * (balance_bands 0 / 2, small or huge grids): a host whose last calls of a shape all read 0 can set balance_bands = 2.  * walk_out (ABI 8; nullable; one int64 in pinned host or device memory, zeroed by the caller): a forward job whose strip
 * evaluated more than 2560 list entries for its strips (the entries its strip mask reaches; until late in round 6: walked) stores that number there (system scope; any such job's, not the largest) -- lists
 * that are long AND stay open, which is what fg_raster_config::heavy_tiles is for: a host turns the policy on by it
 * instead of by the longest LIST alone (a dense opaque cluster has lists of ten thousand entries and closes after a few
 * hundred).
 */
int fg_stbin_fill_jobs(int N, const uint32_t* depth_keys, const int32_t* tile_rects, const uint64_t* tile_masks,
                       int tile_w, int tile_h, int64_t capacity, const int32_t* tile_offsets, const void* count_workspace,
                       int32_t* flatten_ids, int32_t* list_offsets, void* workspace, size_t workspace_bytes,
                       int width, int height, int tile_size, int32_t* jobs_fwd, int32_t* jobs_bwd,
                       int bwd_list_shares, const fg_raster_config* config, int flags, int64_t* ckpt_need_out,
                       fg_stream_t stream);
int fg_raster_jobs_fwd(int channels, int width, int height, int tile_size, const float* splats,
                       const int32_t* tile_offsets, const int32_t* flatten_ids, const int32_t* jobs,
                       const float* background, int n_clamp, float* image, float* alphas,
                       int32_t* last_ids, uint8_t* clamp_mask, float* seg_ckpt, uint32_t* live_words,
                       float* zero_buf, int64_t zero_floats, int64_t* walk_out, const fg_raster_config* config,
                       fg_stream_t stream);
int64_t fg_raster_seg_ckpt_floats(int channels, int width, int height, int tile_size, int64_t n_isects, const fg_raster_config* config);
int fg_raster_jobs_bwd(int channels, int width, int height, int tile_size, const float* splats,
                       const int32_t* tile_offsets, const int32_t* flatten_ids, const int32_t* jobs,
                       const float* background, int n_clamp, const uint8_t* clamp_mask,
                       const float* alphas, const int32_t* last_ids, const float* v_image,
                       const float* v_alphas, float* v_splats, const float* seg_ckpt, const float* image,
                       const uint32_t* live_words, const fg_raster_config* config,
                    fg_stream_t stream);
/* Split v_splats back into per-tensor gradients (any output nullable). */
int fg_unpack_grads(int N, int channels, const float* v_splats, float* v_means2d,
                    float* v_means2d_abs, float* v_conics, float* v_opacities,
                    float* v_features, fg_stream_t stream);

/* ---- Fused per-Gaussian stages (what rasterization() uses) ----------------------------------
 * fg_preprocess_fwd = fg_project_fwd + fg_sh_fwd + fg_pack_splats in one pass; bit-identical
 * outputs.  Composited channels of the record, in order: colour (3 from SH when sh_degree >= 0
 * with colors[N,k_stored,3]; or n_color direct channels colors[N,n_color] when sh_degree = -1;
 * n_color may be 0), camera depth if with_depth, then n_extra channels from extra[N,n_extra];
 * at most FG_MAX_CHANNELS in total.  antialiased != 0 multiplies the record's opacity by the
 * compensation (rasterize_mode="antialiased", freegaussian_model.py:110-119).
 * Optional outputs for the depth-first binning (both nullable; fg_bin_prepare_keys consumes them):
 *   depth_keys[N] u32  = float bits of the camera depth, 0xFFFFFFFF when culled;
 *   tile_rects[N,2] i32 = the FOOTPRINT rectangle {x0 | y0 << 16, w | h << 16}: the reference's
 *   radius-box tile rectangle (floor / ceil of (mean +- radius) / tile) shrunk to the tiles in which
 *   the splat can reach alpha >= 1/255 at a pixel centre (opacity-aware extents of the ellipse
 *   sigma <= ln(255 o), inflated so that rounding only ever keeps more).  Lists binned from it are
 *   the reference's lists minus entries that contribute to no pixel, in the same order; images and
 *   gradients are unchanged.  (0, 0) when culled or when nothing is reached.
 *   tile_masks[N] u64 (ABI 8; nullable, needs tile_rects) = the FOOTPRINT MASK of the rectangle, see fg_stbin_count: which
 *   of its (at most 8 x 8) blocks the ellipse itself reaches with a pixel centre; 0 when the rectangle is empty.
 * Optional output for the backward (nullable; SH colours of degree >= 1 only, ignored otherwise):
 *   sh_jac[N,FG_SH_JAC_FLOATS] f32 = d colour_c / d direction_d before the clamp (9 floats, c-major) and the
 *   clamp mask (bit c of the 10th float's bits: channel c passed max(. + 0.5, 0)).  Handed to
 *   fg_preprocess_bwd it replaces the 192-byte coefficient row the backward would read again per
 *   Gaussian by 40 bytes (same gradients up to the order of fp32 sums).  Rows of culled Gaussians are 0. */
int fg_preprocess_fwd(int N, const float* means, const float* quats, const float* scales,
                      const float* opacities, const float* colors, int sh_degree, int k_stored,
                      int n_color, int with_depth, const float* extra, int n_extra,
                      const float* viewmat, const float* K, int width, int height, float eps2d,
                      float near_plane, float far_plane, float radius_clip, int tile_size,
                      int antialiased, int32_t* radii, float* means2d, float* depths, float* conics,
                      float* compensations, int32_t* tiles_touched, float* splats,
                      uint32_t* depth_keys, int32_t* tile_rects, uint64_t* tile_masks, float* sh_jac, fg_stream_t stream);
/* Camera-pose gradient (ABI 8): dL/d viewmat of the view from the SAME cotangents fg_preprocess_bwd /
 * fg_preprocess_raw_bwd consume -- v_splats (the raster backward's record gradients), v_means2d (+ stride), v_depths,
 * v_conics (nullable) -- for a host whose camera pose requires a gradient (the reference's CameraOptimizer,
 * freegaussian_model.py:120: "off" in every shipped config, applied at :774; gsplat returns v_viewmats from its projection
 * backward and lets autograd carry the SH colour's share through dirs = means - inverse(viewmats)[:3, 3]).  raw != 0: the
 * parameter forms of fg_preprocess_raw_fwd (quats + d_quats, log-scales + d_scales, opacity logits, colors = features_dc,
 * features_rest); raw == 0: those of fg_preprocess_fwd (d_quats, d_scales, features_rest ignored).  sh_jac: the forward's
 * note (nullable: the coefficient rows are read instead).
 * out[19] f32: out[0..15] = dL/d viewmat, row-major 4 x 4 (through the camera-space mean p = W m + t and covariance
 * W C W^T; last row 0), out[16..18] = dL/d campos, the gradient w.r.t. the camera POSITION -W^-1 t that the SH view
 * direction is taken from -- the host adds its pull-back through the matrix inverse (d(A^-1) = -A^-1 dA A^-1) to out[0..15].
 * Deterministic (per-workgroup partial sums in `workspace`, summed in a fixed order).  A pass of its own, ~the cost of the
 * projection backward: the per-Gaussian backward is unchanged when the pose needs no gradient. */
size_t fg_viewmat_bwd_workspace_bytes(int N);
int fg_viewmat_bwd(int N, int raw, const float* means, const float* quats, const float* d_quats, const float* scales,
                   const float* d_scales, const float* opacities, const float* colors, const float* features_rest,
                   int sh_degree, int k_stored, int n_color, int with_depth, int n_extra, const float* viewmat,
                   const float* K, int width, int height, float eps2d, int antialiased, const int32_t* radii,
                   const float* v_splats, const float* v_means2d, int v_means2d_stride, const float* v_depths,
                   const float* v_conics, const float* sh_jac, float* out, void* workspace, size_t workspace_bytes,
                   fg_stream_t stream);
/* The colour + record half of fg_preprocess_fwd on its own: inputs are the projection outputs of
 * fg_project_fwd (radii, means2d, depths, conics; compensations when antialiased).  Splitting the
 * forward this way lets a host run this HBM-bound half on a second stream while the
 * latency-bound binning (fg_bin_prepare / fg_bin_emit_sort) runs on the first; records are
 * bit-identical to fg_preprocess_fwd's. */
int fg_sh_pack_fwd(int N, const float* means, const float* opacities, const float* colors, int sh_degree,
                   int k_stored, int n_color, int with_depth, const float* extra, int n_extra,
                   const float* viewmat, int antialiased, const int32_t* radii, const float* means2d,
                   const float* depths, const float* conics, const float* compensations, float* splats,
                   fg_stream_t stream);
/* fg_preprocess_bwd = fg_unpack_grads + fg_sh_bwd + fg_project_bwd in one pass.  v_splats[N,16]
 * is the record fg_raster_bwd accumulated; its xy slots are IGNORED and v_means2d is used
 * instead (autograd routes that gradient through info["means2d"] so .grad exists there): row i
 * is at v_means2d + i * v_means2d_stride floats (2 = contiguous [N,2]; 16 = the xy slots of a
 * record array, i.e. v_means2d may simply point at v_splats);
 * v_depths[N] / v_conics[N,3] (nullable) are extra gradients on those outputs.  Every output is
 * overwritten densely (zeros for culled Gaussians): v_means[N,3] v_quats[N,4] v_scales[N,3]
 * v_opacities[N] v_colors (same shape as colors) v_extra[N,n_extra].
 * sh_jac (nullable): the note fg_preprocess_fwd wrote for the same inputs; NULL = the coefficient rows
 * are read and the clamp mask / direction gradient recomputed from them. */
int fg_preprocess_bwd(int N, const float* means, const float* quats, const float* scales,
                      const float* opacities, const float* colors, int sh_degree, int k_stored,
                      int n_color, int with_depth, int n_extra, const float* viewmat, const float* K,
                      int width, int height, float eps2d, int antialiased, const int32_t* radii,
                      const float* v_splats, const float* v_means2d, int v_means2d_stride,
                      const float* v_depths, const float* v_conics, float* v_means, float* v_quats,
                      float* v_scales, float* v_opacities, float* v_colors, float* v_extra,
                      const float* sh_jac, fg_stream_t stream);

/* The same two passes on the RAW parameter forms of the reference's gauss_params
 * (freegaussian_model.py:187-196), with the activations its get_outputs applies before the raster
 * call folded in, SH path only (sh_degree >= 0):
 *   quats          -> quats / |quats| + d_quats            (:845;  d_quats[N,4] nullable)
 *   log_scales     -> exp(log_scales) + d_scales           (:844;  d_scales[N,3] nullable)
 *   opacity_logits -> sigmoid(opacity_logits)              (:851;  [N])
 *   colours        -> cat(features_dc[N,1,3], features_rest[N,k_stored-1,3])   (:801)
 * The backward returns gradients of the raw forms (v_quats through the normalisation,
 * v_log_scales through exp, v_opacity_logits through sigmoid, v_features_dc / v_features_rest)
 * and of the deltas (v_d_quats / v_d_scales, required exactly when the delta was given); every
 * output is overwritten densely. */
int fg_preprocess_raw_fwd(int N, const float* means, const float* quats, const float* d_quats,
                          const float* log_scales, const float* d_scales,
                          const float* opacity_logits, const float* features_dc,
                          const float* features_rest, int sh_degree, int k_stored, int with_depth,
                          const float* extra, int n_extra, const float* viewmat, const float* K,
                          int width, int height, float eps2d, float near_plane, float far_plane,
                          float radius_clip, int tile_size, int antialiased, int32_t* radii,
                          float* means2d, float* depths, float* conics, float* compensations,
                          int32_t* tiles_touched, float* splats, uint32_t* depth_keys,
                          int32_t* tile_rects, uint64_t* tile_masks, float* sh_jac, fg_stream_t stream);
int fg_preprocess_raw_bwd(int N, const float* means, const float* quats, const float* d_quats,
                          const float* log_scales, const float* d_scales,
                          const float* opacity_logits, const float* features_dc,
                          const float* features_rest, int sh_degree, int k_stored, int with_depth,
                          int n_extra, const float* viewmat, const float* K, int width, int height,
                          float eps2d, int antialiased, const int32_t* radii, const float* v_splats,
                          const float* v_means2d, int v_means2d_stride, const float* v_depths,
                          const float* v_conics, float* v_means, float* v_quats, float* v_d_quats,
                          float* v_log_scales, float* v_d_scales, float* v_opacity_logits,
                          float* v_features_dc, float* v_features_rest, float* v_extra,
                          const float* sh_jac, fg_stream_t stream);

/* ---- X: factored SH-gradient exchange for view-sharded data parallelism (section 8e) ----------
 * The SH coefficient gradient of one view is rank-1 per Gaussian: v_coeffs[i,k,:] =
 * basis_k(dir_view(i)) * g[i,:] with g the clamp-masked colour gradient.  Instead of all-reducing
 * 192 B per Gaussian, ranks all-gather g (12 B) and rebuild the sum locally.
 * fg_preprocess_bwd_factored = fg_preprocess_bwd (SH colours) that writes v_rgb[N,v_rgb_floats]
 * instead of v_colors: g (3 floats), or g + the unit view direction the basis was evaluated at
 * (6 floats) when every rank renders its own deformed means.  fg_sh_grad_accumulate: payload =
 * n_views blocks of view_stride floats; payload_floats = 3: block v = [g_v (3N floats) | camera
 * position of view v (3 floats) | padding] and the direction is normalize(means - cam_v);
 * payload_floats = 6: block v = N rows [g_v | direction] (means unused, may be NULL).  Writes
 * v_coeffs[N,k_stored,3] = scale * sum_v basis(direction_v) (x) g_v densely. */
int fg_preprocess_bwd_factored(int N, const float* means, const float* quats, const float* scales,
                               const float* opacities, const float* colors, int sh_degree, int k_stored,
                               int with_depth, int n_extra, const float* viewmat, const float* K,
                               int width, int height, float eps2d, int antialiased, const int32_t* radii,
                               const float* v_splats, const float* v_means2d, int v_means2d_stride,
                               const float* v_depths, const float* v_conics, float* v_means,
                               float* v_quats, float* v_scales, float* v_opacities, float* v_rgb,
                               int v_rgb_floats, float* v_extra, const float* sh_jac, fg_stream_t stream);
int fg_sh_grad_accumulate(int N, int n_views, int sh_degree, int k_stored, const float* means,
                          const float* payload, int64_t view_stride, int payload_floats, float scale,
                          float* v_coeffs, fg_stream_t stream);
/* The same exchange for the MODEL's parameter layout (ABI 7; freegaussian_model.py:187-196: features_dc [N,3] and
 * features_rest [N,K-1,3] are two tensors, the other gradients come out of fg_preprocess_raw_bwd):
 * fg_preprocess_raw_bwd_factored = fg_preprocess_raw_bwd that writes v_rgb[N,v_rgb_floats] instead of the two
 * coefficient gradients; fg_sh_grad_accumulate_split rebuilds them from the gathered payloads into the two arrays. */
int fg_preprocess_raw_bwd_factored(
    int N, const float* means, const float* quats, const float* d_quats, const float* log_scales, const float* d_scales,
    const float* opacity_logits, const float* features_dc, const float* features_rest, int sh_degree, int k_stored,
    int with_depth, int n_extra, const float* viewmat, const float* K, int width, int height, float eps2d, int antialiased,
    const int32_t* radii, const float* v_splats, const float* v_means2d, int v_means2d_stride, const float* v_depths,
    const float* v_conics, float* v_means, float* v_quats, float* v_d_quats, float* v_log_scales, float* v_d_scales,
    float* v_opacity_logits, float* v_rgb, int v_rgb_floats, float* v_extra, const float* sh_jac, fg_stream_t stream);
int fg_sh_grad_accumulate_split(int N, int n_views, int sh_degree, int k_stored, const float* means,
                                const float* payload, int64_t view_stride, int payload_floats, float scale,
                                float* v_features_dc, float* v_features_rest, fg_stream_t stream);
/* Sparse payload of the factored view-DP exchange (ABI 8).  Only Gaussians that took part in a pixel of the view have a
 * colour gradient; a rank's all-gathered block can be [count, camera position(3) | capacity rows of (id, g[3] (, unit
 * direction[3]))] instead of N dense rows (the reference is single-GPU, scripts/run.sh:58: new capability).
 * fg_payload_compact: the rows of dense[N, payload_floats] with a non-zero g, in id order -- incl_scan[N] (int32) is the
 * inclusive scan of the caller's row flags, row i goes to slot incl_scan[i] - 1 --, into out[4 + capacity * (1 +
 * payload_floats)]; out[0] = the count as int bits (rows beyond capacity are dropped: count > capacity tells the receiver
 * not to use the block), out[1..3] are left to the caller.
 * fg_payload_expand: n_views such blocks (block_stride floats apart) back into dense blocks of dense_stride floats laid out
 * as fg_sh_grad_accumulate[_split] reads them; the dense blocks must be zero on entry. */
int fg_payload_compact(int N, int payload_floats, const float* dense, const int32_t* incl_scan, int64_t capacity,
                       float* out, fg_stream_t stream);
int fg_payload_expand(int N, int payload_floats, int n_views, const float* compact, int64_t block_stride,
                      int64_t capacity, float* dense, int64_t dense_stride, fg_stream_t stream);

/* ---- One call per direction (ABI 7): the whole eager step of one view ------------------------------------------
 * fg_step_fwd = fg_preprocess_fwd (raw = 0) or fg_preprocess_raw_fwd (raw = 1) -> fg_stbin_count -> fg_stbin_fill_jobs ->
 * fg_raster_jobs_fwd; fg_step_bwd = fg_raster_jobs_bwd -> the matching per-Gaussian backward (the factored one when
 * io->v_rgb is given).  Same kernels, same results as the stage-wise calls: what changes is the host's work per step
 * -- three small structs instead of ~150 marshalled arguments, two workspace allocations instead of ~20 buffers -- which
 * is the whole cost of a launch-bound step (the reference's quarter- and half-resolution phases,
 * freegaussian_model.py:626-633).  Replaces, like those calls, the rasterization(...) of freegaussian_model.py:847-868
 * and its autograd backward.
 *   fg_step_desc    what is rendered (sizes, SH degree, channels, composite epilogue, list capacity, flags)
 *   fg_step_io      device pointers: the inputs (plain or raw parameter forms, as the two preprocess entry points take
 *                   them), count_out (pinned host memory, the four words of fg_stbin_count), and for the backward the
 *                   upstream gradients and the output gradients
 *   fg_step_layout  from fg_step_layout_query: offsets / sizes of every buffer inside the two caller-allocated
 *                   workspaces -- `keep` (read by the backward and by the caller: FG_STEP_RADII .. FG_STEP_CLAMP_MASK;
 *                   uninitialised) and `tmp` (FG_STEP_COUNT_WS, FG_STEP_FILL_WS: free again once the call's launches
 *                   have run).  FG_ERR_UNSUPPORTED: shapes the supertile binning or the job-list launches do not take
 *                   (use the stage-wise entry points).
 * A list longer than desc->capacity leaves empty lists (see fg_stbin_fill) and an image of the background: the host
 * sees it in count_out[0] and repeats fg_step_fwd with a larger capacity and fresh workspaces. */
enum {
  FG_STEP_RADII, FG_STEP_MEANS2D, FG_STEP_DEPTHS, FG_STEP_CONICS, FG_STEP_COMP, FG_STEP_TILES, FG_STEP_SPLATS,
  FG_STEP_DEPTH_KEYS, FG_STEP_TILE_RECTS, FG_STEP_TILE_MASKS, FG_STEP_SH_JAC, FG_STEP_TILE_OFFSETS, FG_STEP_LIST_OFFSETS, FG_STEP_FLATTEN_IDS,
  FG_STEP_JOBS, FG_STEP_LIVE, FG_STEP_SEG_CKPT, FG_STEP_V_SPLATS, FG_STEP_RENDER, FG_STEP_ALPHAS, FG_STEP_LAST_IDS,
  FG_STEP_CLAMP_MASK, FG_STEP_COUNT_WS, FG_STEP_FILL_WS, FG_STEP_BUFFERS
};
typedef struct fg_step_desc {
  int32_t size;           /* sizeof(fg_step_desc) */
  int32_t N, width, height, tile_size;
  int32_t raw;            /* 1: the raw parameter forms of fg_preprocess_raw_* (SH colours only) */
  int32_t sh_degree;      /* -1: colors[N,n_color] direct channels */
  int32_t k_stored, n_color, with_depth, n_extra, antialiased;
  int32_t n_clamp;        /* composite epilogue: clamp the first n_clamp channels; io->background nullable */
  int32_t want_backward;  /* 0: no liveness words / checkpoints / record-gradient array / SH note */
  int32_t list_shares;    /* backward over shares of the tiles' lists (three channels; fg_raster_seg_ckpt_floats) */
  int32_t flags;          /* FG_STBIN_LONG_SEGMENTS | FG_STEP_NO_FOOTPRINT_MASKS (ABI 8: the step bins by footprint masks -- see
                             fg_stbin_count -- unless this is set: whole rectangles, as before) */
  float eps2d, near_plane, far_plane, radius_clip;
  int64_t capacity;       /* list entries the workspaces hold */
} fg_step_desc;
typedef struct fg_step_io {
  const float *means, *quats, *d_quats, *scales, *d_scales, *opacities, *colors, *features_rest, *extra, *viewmat, *K,
      *background;        /* raw = 1: scales = log-scales, opacities = logits, colors = features_dc */
  int64_t* count_out;     /* nullable */
  /* backward only */
  const float *v_render, *v_alphas, *v_depths, *v_conics;  /* upstream gradients; all but v_render nullable */
  float *v_means, *v_quats, *v_d_quats, *v_scales, *v_d_scales, *v_opacities, *v_colors, *v_features_rest, *v_extra;
  float* v_rgb;           /* non-NULL: the factored form (v_colors / v_features_rest unused) */
  int32_t v_rgb_floats;
  /* optional hipEvent_t pair recorded on `stream` around the call's raster launch (forward call: the raster forward,
     backward call: the raster backward): a host that times the dominant kernel of the step does not have to take the
     stage-wise entry points for it */
  void *ev_raster_begin, *ev_raster_end;
  int64_t* ckpt_need_out; /* nullable, forward only: fg_stbin_fill_jobs' ckpt_need_out (int64[9]); + one more word, [9]:
                             fg_raster_jobs_fwd's walk_out (the caller zeroes it) */
} fg_step_io;
typedef struct fg_step_layout {
  int64_t keep_bytes, tmp_bytes;
  int64_t offset[FG_STEP_BUFFERS], nbytes[FG_STEP_BUFFERS];  /* nbytes 0: the step has no such buffer */
  int64_t jobs_words, seg_ckpt_floats;
  int32_t channels;
} fg_step_layout;
int fg_step_layout_query(const fg_step_desc* desc, const fg_raster_config* config, fg_step_layout* out);
int fg_step_fwd(const fg_step_desc* desc, const fg_raster_config* config, const fg_step_io* io, void* keep, void* tmp,
                const fg_step_layout* layout, fg_stream_t stream);
int fg_step_bwd(const fg_step_desc* desc, const fg_raster_config* config, const fg_step_io* io, void* keep,
                const fg_step_layout* layout, fg_stream_t stream);

/* S1 in one pass: the densification statistics after_train_iter keeps (freegaussian_model.py:369-392), all in place:
 * for radii[i] > 0: xys_grad_norm[i] += |absgrad[i]| (absgrad [N,2]), vis_counts[i] += 1,
 * max_2dsize[i] = max(max_2dsize[i], radii[i] / max_dim); other rows untouched. */
int fg_densify_stats(int N, const float* absgrad, const int32_t* radii, float max_dim, float* xys_grad_norm,
                     float* vis_counts, float* max_2dsize, fg_stream_t stream);
/* ---- D: adaptive density control ---------------------------------
 * The reference's refinement_after / split_gaussians / dup_gaussians / cull_gaussians and the
 * Adam-state surgery around them (freegaussian_model.py:313-367, :404-571), as: one decision
 * pass, prefix sums (the caller's), one map pass, ONE coalesced row copy per tensor into its final
 * place, one fix-up pass over the new rows.
 *
 * fg_densify_flags: flags[N], bit 0 split, bit 1 duplicate, bit 2 old row survives, bit 3 its
 * split children survive the cull, bit 4 its duplicate survives.  Thresholds as in the config
 * (:68-88); pass split_screen_size / cull_scale_thresh / cull_screen_size < 0 to disable the
 * step-dependent tests (:425, :505-511); do_densify = 0 is the cull-only branch (:466-467).
 * max_dim = max(H, W) of the last render (:421).  max_2dsize nullable. */
int fg_densify_flags(int N, int do_densify, float max_dim, float densify_grad_thresh,
                     float densify_size_thresh, float split_screen_size, float cull_alpha_thresh,
                     float cull_scale_thresh, float cull_screen_size, const float* xys_grad_norm,
                     const float* vis_counts, const float* max_2dsize, const float* log_scales,
                     const float* opacity_logits, uint8_t* flags, fg_stream_t stream);
/* pos_old / pos_child / pos_dup [N]: EXCLUSIVE prefix sums of flag bits 2 / 3 / 4; rank_split[N]:
 * of bit 0.  n_old, n_child: totals of bits 2, 3; n_split_total: of bit 0.  Row order of the new set
 * (the reference's): surviving old rows, children sample-major, duplicates.  Outputs for every
 * new-set row j: src_index[j] = the old row it copies; sample_index[j] = row of the
 * randn((n_split_samples * n_split_total, 3)) draw (:530) for a child, -2 for the duplicate of a
 * split parent (it carries the shrunk scales: `dups` is evaluated after the in-place shrink,
 * :428-431), -1 otherwise. */
int fg_densify_map(int N, const uint8_t* flags, const int32_t* pos_old, const int32_t* pos_child,
                   const int32_t* pos_dup, const int32_t* rank_split, int n_old, int n_child,
                   int n_split_total, int n_split_samples, int32_t* src_index, int32_t* sample_index,
                   fg_stream_t stream);
/* dst[j,:] = src[src_index[j],:] for rows of row_floats floats; rows j >= zero_from are zeroed
 * instead (new rows of the Adam moments, :343-356). */
int fg_gather_rows(int64_t n_rows, int row_floats, const float* src, const int32_t* src_index,
                   int64_t zero_from, float* dst, fg_stream_t stream);
/* In place on rows [first_row, first_row + n_rows) of the NEW set: children (sample_index >= 0):
 * mean += R(q/|q|) (exp(s) * z), s = log(exp(s)/1.6) (:530-549); sample_index == -2: only the
 * scales shrink. */
int fg_split_children(int first_row, int n_rows, const int32_t* sample_index, const float* samples,
                      float* means, float* log_scales, const float* quats, fg_stream_t stream);

/* ---- M: attribute-mask back-projection -------------------------
 * One key frame of preprocess/knn_gaussian.py:116-132: every visible Gaussian (radii > 0) whose
 * centre, truncated toward zero, lies in the image and whose depth agrees with the rendered
 * expected depth there (-0.1 d < d - depth_i < d) ORs that pixel's labels into its row:
 *   gaussian_masks[i, j] |= atrb_masks[y, x, j] & mask_valids[j]      j < n_attributes
 * atrb_masks[H,W,n_labels_stored] and mask_valids[n_labels_stored] are bool (1 byte), the last
 * stored label (background) is dropped as the reference's `[..., :-1]` does: pass
 * n_attributes = n_labels_stored - 1.  gaussian_masks[N,n_attributes] bool, caller-zeroed, is
 * accumulated over frames and saved as gaussian_mask_NxM.npy (freegaussian_pipeline.py:45-47). */
int fg_mask_backproject(int N, const float* means2d, const float* depths, const int32_t* radii,
                        const float* depth_map, int width, int height, const uint8_t* atrb_masks,
                        const uint8_t* mask_valids, int n_labels_stored, int n_attributes,
                        uint8_t* gaussian_masks, fg_stream_t stream);

/* ---- F: flow derivative -------------------------------------------------------------------
 * Per-pixel camera flow A v / Z + B w (preprocess/epipolar_flow.py:274-309; pixel centres at
 * integer coordinates, infinite depth -> 0, :315-317).  depth[H,W], veloc[3], omega[3],
 * flow[H,W,2]. */
int fg_camera_flow(int width, int height, const float* depth, const float* K,
                   const float* veloc, const float* omega, float* flow, fg_stream_t stream);
/* Exact-reprojection camera flow (preprocess/epipolar_flow_bp.py:268-295): per pixel, lift with
 * depth0, map by M[3,4] (row-major; the host composes it from the two poses exactly as the
 * reference does, c2w1 . c2w0^-1 after its OpenGL->OpenCV column flip, :265-266, :275-276), project
 * with K, divide by depth1, subtract the pixel (pixel centres at integer coordinates, :268).
 * flow[H,W,2] = sign * (uv - xy); sign = -1 is the "sceneflow" the reference returns (:295).
 * Infinite depth0 -> 0 (:285-287). */
int fg_reprojection_flow(int width, int height, const float* depth0, const float* depth1,
                         const float* K, const float* M, float sign, float* flow, fg_stream_t stream);
/* Per-Gaussian projection-flow Jacobian (Lemma 1, docs/index.html:256-273, code sign
 * convention of epipolar_flow.py:277-298): for visible Gaussian i at means2d mu_i, depth Z_i,
 * camera-frame velocity vel[i]:   u_gs[i] = A(mu_i) vel[i] / Z_i
 *                                 u_cam[i] = A(mu_i) veloc / Z_i + B(mu_i) omega. */
int fg_flow_fwd(int N, const float* means2d, const float* depths, const int32_t* radii,
                const float* vel, const float* K, const float* veloc, const float* omega,
                float* u_gs, float* u_cam, fg_stream_t stream);
int fg_flow_bwd(int N, const float* means2d, const float* depths, const int32_t* radii,
                const float* vel, const float* K, const float* veloc, const float* omega,
                const float* v_u_gs, const float* v_u_cam, float* v_means2d, float* v_depths,
                float* v_vel, fg_stream_t stream);

/* ---- L: the image loss of the training step, fused (csrc/loss.hip) --------------------------------------
 * mean|gt - pred| and mean SSIM(gt, pred) of two [H, W, C] images -- the two terms of the reference's main loss
 * (freegaussian_model.py:965-981; SSIM = pytorch_msssim.SSIM(data_range=1.0, size_average=True, channel=3), :22, :211:
 * 11-tap Gaussian window, sigma 1.5, 'valid' borders) -- in one launch forward and one backward instead of ~200 torch
 * launches each way (10.4 ms per step at 1920 x 1080 on an MI355X, scripts/loss_time.py).  height, width > 10.
 *   fg_l1_ssim_fwd: out[0] = mean |gt - pred|, out[1] = mean SSIM (device floats; reduced in a fixed order: reproducible);
 *                   maps[3 * C * (H-10) * (W-10)]: the SSIM map's partial derivatives, for the backward (opaque);
 *                   workspace[fg_l1_ssim_workspace_floats(H, W, C)] floats.
 *   fg_l1_ssim_bwd: v_out[2] (device) = dL/d out[0], dL/d out[1]  ->  v_pred[H, W, C] (overwritten; gt gets no gradient). */
size_t fg_l1_ssim_workspace_floats(int height, int width, int channels);
int fg_l1_ssim_fwd(int height, int width, int channels, const float* pred, const float* gt, float* maps,
                   float* workspace, size_t workspace_floats, float* out, fg_stream_t stream);
int fg_l1_ssim_bwd(int height, int width, int channels, const float* pred, const float* gt, const float* maps,
                   const float* v_out, float* v_pred, fg_stream_t stream);

/* ---- A: one Adam update of a dense float32 tensor in one launch (csrc/adam.hip) ---------------------------
 * torch.optim.Adam's defaults (no weight decay, no amsgrad), the same fp32 operations in the same order as torch's
 * own step: param, exp_avg, exp_avg_sq updated in place from grad; step = the 1-based count of this update (bias
 * corrections 1 - beta^step computed in double, as torch does; the hyper-parameters are doubles for the same reason).  All four arrays 16-byte aligned, n elements.
 * Replaces optimizer.step() of the reference's six Gaussian parameter groups (freegaussian_config.py optimizers):
 * 1.42 ms -> 0.35 ms per iteration at 1M Gaussians (scripts/train_step_bench.py). */
int fg_adam_step(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, double lr, double beta1,
                 double beta2, double eps, int64_t step, fg_stream_t stream);
/* The same for several tensors in ONE launch (every 16 tensors one): each tensor with its own hyper-parameters and
 * update count, results identical to fg_adam_step per tensor.  At the reference's low resolutions an iteration is bound
 * by launches: six Gaussian parameter groups, six launches -> one. */
#define FG_ADAM_MAX_TENSORS 16
typedef struct fg_adam_tensor {
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  int64_t n;
  double lr, beta1, beta2, eps;
  int64_t step;
} fg_adam_tensor;
int fg_adam_step_multi(int count, const fg_adam_tensor* tensors, fg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FGRASTER_H */
