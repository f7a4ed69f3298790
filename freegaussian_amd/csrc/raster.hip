// K5 / K6: front-to-back alpha compositing over 16x16 pixel tiles, forward and backward.
//
// Replaces the rasterize stage of the rasterization(...) call at
// /root/reference freegaussian/freegaussian_model.py:847-868 (render_mode, packed=False,
// absgrad=True) and its autograd backward.
//
// Design for gfx950 (DESIGN.md "raster kernels"):
//  * every Gaussian is one 64-byte, line-aligned record [x y o a | b c f0 f1 | f2.. ] so the
//    per-tile gather is one cache line per list entry;
//  * a workgroup owns one tile; each lane owns PPT pixels of one column (PPT = 1, 2 or 4, i.e.
//    4, 2 or 1 wavefronts per tile).  With PPT = 4 a single 64-lane wavefront owns the whole
//    tile: no cross-wave reduction is needed in the backward and one LDS broadcast read of a
//    record feeds 4 pixel evaluations per lane;
//  * the tile's list is staged through LDS in batches of one entry per lane; records are read
//    back at a wave-uniform address (LDS broadcast, conflict-free ds_read_b128);
//  * backward: per-lane partial gradients of the 16 accumulators are summed over the wave with
//    a transposing butterfly (v_permlane32_swap / v_permlane16_swap / DPP, no LDS) that leaves
//    accumulator i in lanes 4i..4i+3, then ONE atomic instruction adds 16 consecutive floats
//    (one 64-byte line) of the Gaussian's gradient record;
//  * blockIdx -> tile mapping keeps each XCD (blockIdx % 8) on its own horizontal band of the
//    image so neighbouring tiles share records in one 4 MiB L2.
#include "fg_common.h"

#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "jobs_build.h"

namespace {
using namespace fgjobs;

constexpr int TILE = 16;

__host__ __device__ constexpr int rec_vec4(int C) { return (6 + C + 3) / 4; }

// XCD-aware workgroup -> tile maps.  Workgroups are dealt round-robin over the 8 XCDs in id
// order, so ids with equal (id % 8) share an L2 -- and an XCD that gets the heavy part of the
// image stalls the in-order dispatcher for everybody.
//  mode 0 "rows":  tile row r goes to XCD r % 8 (every XCD sees every part of the image: balanced;
//                  horizontal neighbours still share an L2).  Grid = 8 * ceil(tile_h/8) * tile_w,
//                  ids past an XCD's last row own no tile (-1).
//  mode 1 "bands": each XCD owns a contiguous band of tiles, walked row-major.
//  mode 3 "split": as "cols" but every XCD owns two half-height bands, one from the upper and one
//                  from the lower half of the image (better balance on centre-weighted scenes).
//  mode 2 "cols":  each XCD owns a band of whole tile rows, walked column-major, so a tile and
//                  its vertical neighbours (which share most of their splats) run back to back.
__device__ __forceinline__ int tile_of_block(int b, int n, int tile_w, int tile_h, int mode) {
  const int xcd = b & 7, k = b >> 3;
  // measurement hook (FG_DEBUG_ONLY_XCD=x): only the workgroups of XCD x do their tile, so that a
  // launch's duration is that XCD's share of the work
  const int only = ((mode >> 8) & 15) - 1, kmod = mode >> 12;
  mode &= 255;
  if (only >= 0 && xcd != only) return -1;
  if (kmod > 1 && k % kmod != 0) return -1;  // FG_DEBUG_K_MOD=m: every m-th tile of each XCD only
  if (mode == 0) {
    const int row = (k / tile_w) * 8 + xcd;
    return row < tile_h ? row * tile_w + (k % tile_w) : -1;
  }
  if (mode == 2) {  // row bands, walked column-major: vertical neighbours are adjacent in time
    const int q = tile_h >> 3, r = tile_h & 7;
    const int rows = q + (xcd < r ? 1 : 0);
    const int row0 = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    if (rows == 0) return -1;
    const int col = k / rows, row = row0 + k % rows;
    return col < tile_w ? row * tile_w + col : -1;
  }
  if (mode == 3) {  // 16 row groups; XCD x owns groups x and x+8 (a light and a heavy part of a
                    // centre-weighted image), each walked column-major
    int kk = k;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int g = xcd + 8 * half;
      const int r0 = (g * tile_h) >> 4, r1 = ((g + 1) * tile_h) >> 4;
      const int rows = r1 - r0, cnt = rows * tile_w;
      if (kk < cnt) return (r0 + kk % rows) * tile_w + kk / rows;
      kk -= cnt;
    }
    return -1;
  }
  if (mode == 4 || mode == 5) {  // 4x2 (mode 4) or 2x4 (mode 5) rectangles, one per XCD, column-major
    const int nx = mode == 4 ? 4 : 2, ny = 8 / nx;
    const int rx = xcd % nx, ry = xcd / nx;
    const int c0 = (rx * tile_w) / nx, c1 = ((rx + 1) * tile_w) / nx;
    const int r0 = (ry * tile_h) / ny, r1 = ((ry + 1) * tile_h) / ny;
    const int rows = r1 - r0, cnt = rows * (c1 - c0);
    if (k >= cnt) return -1;
    return (r0 + k % rows) * tile_w + c0 + k / rows;
  }
  if (b >= n) return -1;
  const int q = n >> 3, r = n & 7;
  const int first = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return first + k;
}

// Mixed launch: job k of XCD x.  The XCD's tiles are those of tile_of_block mode 2 (a band of whole
// tile rows, column-major); the first n - tail of them are whole-tile jobs (strip = -1), each of
// the last `tail` is four single-strip jobs.  Why: a tile's list is walked serially by its
// wavefront, and a launch ends with wavefronts that run alone on their SIMDs; measured
// (FG_DEBUG_K_MOD hook), 1/5 of the tiles of the 1M / 1080p scene take half the time of all of
// them.  Shorter jobs at the end of every XCD's sequence shorten that tail; splitting every tile
// would repeat the per-entry work everywhere (slower: profiles/r01_ppt_by_tiles.md).
__device__ __forceinline__ int job_of_block(int b, int tile_w, int tile_h, int nx, int tail_tiles, int& strip) {
  // tail_tiles = tail4 | tail2 << 16: the last tail4 tiles of the sequence are four single-strip
  // jobs (strip = 0..3), the tail2 tiles before them two two-strip jobs (strip = 4 + half)
  const int tail4_req = tail_tiles & 0xFFFF, tail2_req = tail_tiles >> 16;
  const int xcd = b & 7, k = b >> 3;
  const Band band = band_of_xcd(xcd, tile_w, tile_h, nx);
  const int n = band.nrows * band.ncols;
  const int tail4 = min(tail4_req, n), tail2 = min(tail2_req, n - tail4), n_main = n - tail4 - tail2;
  int idx;
  if (k < n_main) {
    idx = k;
    strip = -1;
  } else if (k < n_main + 2 * tail2) {
    idx = n_main + ((k - n_main) >> 1);
    strip = 4 + ((k - n_main) & 1);
  } else {
    const int t = k - n_main - 2 * tail2;
    idx = n_main + tail2 + (t >> 2);
    strip = t & 3;
  }
  if (idx >= n) return -1;
  return band_tile(band, idx, tile_w);
}

__global__ void __launch_bounds__(1024)
build_jobs_kernel(fgjobs::JobBuild jb, const int32_t* __restrict__ tile_offsets) {
  __shared__ uint32_t s_rows[fgjobs::FG_BAND_MAX_ROWS];
  fgjobs::build_jobs_block<1024>((int)blockIdx.x, jb, tile_offsets, s_rows);
}

// job k of XCD (b & 7) from a list; tile or -1.  *prefix: a strip job of a HEAVY tile (jobs_build.h: it walks the
// first FG_HEAVY_PREFIX entries only)
__device__ __forceinline__ int job_from_list(int b, const int32_t* __restrict__ jobs, int cap, int& strip,
                                             bool* no_ckpt = nullptr, bool* prefix = nullptr) {
  const int xcd = b & 7, k = b >> 3;
  if (k >= jobs[xcd]) return -1;
  const int e = jobs[8 + (size_t)xcd * cap + k];
  strip = (e & 7) - 1;
  if (no_ckpt) *no_ckpt = (e & FG_JOB_NO_CKPT) != 0;
  if (prefix) *prefix = (e & FG_JOB_PREFIX) != 0;
  return (e & ~(FG_JOB_NO_CKPT | FG_JOB_PREFIX)) >> 3;
}

// the same from a list of list-share jobs (JobParams::seg_parts): tile, part, parts
__device__ __forceinline__ int share_from_list(int b, const int32_t* __restrict__ jobs, int cap, int& part, int& parts) {
  const int xcd = b & 7, k = b >> 3;
  if (k >= jobs[xcd]) return -1;
  const int e = jobs[8 + (size_t)xcd * cap + k];
  parts = (e & 63) + 1;
  part = (e >> 6) & 63;
  return e >> 12;
}

struct Splat {
  float x, y, o, a, b, c;
};

// exp(-sigma) = exp2(e2) with  e2 = -log2(e) * sigma,  sigma = 0.5 (a dx^2 + c dy^2) + b dx dy,
// arranged per splat as e2 = (hc dy + bdx) dy + hadx2 (Horner in dy, constants folded): two FMAs per
// pixel slot.  Forward and backward share it, so they take identical skip decisions.
// Lane masks straight from one vector compare (v_cmp writing an SGPR pair), and selects that take
// such a mask: predicates stay in scalar registers and are combined by the scalar unit.
// Predicate codes are LLVM's FCmpInst values: OGE 3, OLE 5, ULE 13.
__device__ __forceinline__ uint64_t lanes_oge(float a, float b) { return __builtin_amdgcn_fcmpf(a, b, 3); }
__device__ __forceinline__ uint64_t lanes_ole(float a, float b) { return __builtin_amdgcn_fcmpf(a, b, 5); }
__device__ __forceinline__ uint64_t lanes_ule(float a, float b) { return __builtin_amdgcn_fcmpf(a, b, 13); }
__device__ __forceinline__ uint64_t lanes_uge(float a, float b) { return __builtin_amdgcn_fcmpf(a, b, 11); }
__device__ __forceinline__ uint64_t lanes_sle(int a, int b) { return __builtin_amdgcn_sicmp(a, b, 41); }  // ICMP_SLE
__device__ __forceinline__ float lane_select(uint64_t m, float if_set, float if_clear) {
  float r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(m));
  return r;
}
// if_set where the mask bit is set, else 0: the zero is the instruction's inline constant (as a
// "v" operand of the general form it occupied a VGPR for the whole kernel)
__device__ __forceinline__ float lane_select0(uint64_t m, float if_set) {
  float r;
  asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(if_set), "s"(m));
  return r;
}
__device__ __forceinline__ int lane_select(uint64_t m, int if_set, int if_clear) {
  int r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(m));
  return r;
}

// Optional post-composite folded into the raster kernels (reference O1, freegaussian_model.py:875-877):
//   out[c] = render[c] + (1 - alpha) * background[c],  the first n_clamp channels clamped to [0,1].
// clamp_mask[H,W] keeps, per pixel, bit c = "channel c was strictly outside [0,1]" (torch.clamp
// passes the gradient at the bounds themselves), written by the forward and read by the backward.
struct Composite {
  const float* background;  // [C] or nullptr
  int n_clamp;
  uint8_t* clamp_mask;      // [H,W] when n_clamp > 0
};

// List segmentation of the backward (mixed launches, 3 composited channels): a tile's list is walked
// serially by its wavefront, ~8000 tile jobs are fewer than two per wavefront slot of the chip, and the
// launch ends when the SIMD with the longest sum of walks ends.  The forward therefore leaves a
// CHECKPOINT of every pixel's compositing state -- (T, C0, C1, C2) before list entry g -- at every
// FG_SEG_ENTRIES-th entry of a tile's list (g = start + c * FG_SEG_ENTRIES, slot g / FG_SEG_ENTRIES:
// unique, tiles own disjoint index ranges), and a backward job walks only ITS share of a tile's
// segments, starting from the checkpoint at its upper end: more, shorter, independent jobs without
// repeating any per-entry work (splitting a tile by pixels repeats the record read, the reduction
// and the atomic of every entry in every part).
#ifndef FG_SEG_ENTRIES
#define FG_SEG_ENTRIES 64
#endif
static_assert(FG_SEG_ENTRIES == fgjobs::FG_SEG_ENTRIES_H, "a heavy tile's batch is a checkpoint segment");
#ifndef FG_HEAVY_AHEAD
#define FG_HEAVY_AHEAD 8
#endif
#ifndef FG_WALK_REPORT
// What a forward job reports (fg_raster_jobs_fwd walk_out): the list entries it EVALUATED for its strips -- those whose strip mask
// reaches them -- when more than this.  (Until late in round 6: the entries it WALKED.  On layouts the thresholds were never looked
// at with -- scripts/policy_regret.py 16 11, scripts/walk_values.py -- clouds of a million small half-transparent splats walk 2575-2862
// entries in every view, of which a one-strip job evaluates a fraction: heavy tiles then serve a remainder of a few hundred entries
// behind the 2560-entry prefix with a launch of their own and lose 13 % of the step, and the trained scene 1-2 %; the gate's faint
// cluster of LARGE splats walks 2700-4700 entries, evaluates every one of them for every strip, and gains 12 % from heavy tiles: what
// separates them is the work of the serial job, not the length of its walk.)
#define FG_WALK_REPORT 2560
#endif
#ifndef FG_HEAVY_SUB
#define FG_HEAVY_SUB 4  // workgroups per listed local job
#endif
// measured on MI355X (1M Gaussians, 1080p, profiles/r02_backward_list_shares.md): the last 400 tiles of
// every XCD's sequence as 3 shares each over 128-entry segments: 0.398 -> 0.374 ms against two-strip jobs
// for the last 300; 4 shares over 64-entry segments: 0.370 -> 0.349 (the forward writes checkpoints only
// for the tiles the backward splits -- FG_JOB_NO_CKPT -- so the finer grain costs it nothing)
#ifndef FG_SEG_PARTS_DEFAULT
#define FG_SEG_PARTS_DEFAULT 5
#endif
#ifndef FG_SEG_PARTS_SMALL
#define FG_SEG_PARTS_SMALL 6
#endif
#ifndef FG_SEG_TAIL_DEFAULT
#define FG_SEG_TAIL_DEFAULT 400
#endif
// The checkpoint buffer starts with a table int32[n_tiles][4]: per tile and 4-row strip, the last list
// index any pixel of the strip used (written by the forward's jobs, one strip each or all four; read by
// the backward's share jobs, which then know their share of the list before any pixel state arrives);
// the checkpoint slots follow, aligned to a slot (256 float4).
// Behind the table: a plane float[height * width] with every pixel's final transmittance exactly as the
// forward held it.  The reference's backward starts from T_final = 1 - alpha_out, i.e. from a value
// rounded at ulp(1) -- 6e-4 relative on a saturated pixel (T ~ 1e-4) -- and rebuilds every T_i from it, so
// ALL its T-dependent terms carry that pixel's factor rho = (1 - alpha_out) / T_exact.  A share job that
// resumes from the forward's exact checkpoint has to apply rho itself to agree with the reference's
// arithmetic (without it: 8e-5 relative L2 on the gradients of the 1M / 1080p frame against 1.5e-5).
__host__ __device__ __forceinline__ size_t seg_plane_offset4(int n_tiles) { return ((size_t)n_tiles + 255) / 256 * 256; }
__host__ __device__ __forceinline__ size_t seg_slots_offset4(int n_tiles, int width, int height) {
  return seg_plane_offset4(n_tiles) + ((size_t)width * height / 4 + 256) / 256 * 256;
}
// A checkpoint slot: 256 float4 (the tile's pixels, row-major) + 256 bytes (per pixel, HEAVY tiles only: the last entry
// of the slot's 64-entry batch the pixel took, + 1; 0 = none) = 272 float4.
constexpr int FG_SEG_SLOT4 = TILE * TILE + TILE * TILE / 16;
// The slot of the batch that starts at list entry g = start + 64 c of `tile` (start = the tile's first entry) is slot0 + c.
// slot0 by formula (no table): start / 64 + tile -- every batch of every tile its own, the first (c = 0: a heavy tile's
// prefix state; no checkpoint) and a short last one included: a tile of len entries has ceil(len / 64) <= floor(len / 64)
// + 1 batches, hence the "+ tile"; the buffer then has n_isects / 64 + n_tiles slots.  slot0 from the job lists' table
// (compact slots, jobs_build.h JobBuild::slot_budget): only the tiles the backward may cut into shares own slots,
// ceil(len / 64) each; -1 = none.
__device__ __forceinline__ int seg_slot_base(const int32_t* __restrict__ slot_tab, int start, int tile) {
  return slot_tab ? __builtin_amdgcn_readfirstlane(slot_tab[tile]) : start / FG_SEG_ENTRIES + tile;
}
__device__ __forceinline__ size_t seg_slot_index(int slot0, int start, int g, int fine = FG_SEG_FINE_NEVER) {
  return (size_t)slot0 + (size_t)seg_index(g - start, fine);  // (fine: the checkpoint grid, jobs_build.h)
}
struct Segments {
  float4* ckpt;             // [slots][FG_SEG_SLOT4]; nullptr = no segmentation
  const int32_t* slot_tab;  // compact slots: first slot per tile (-1: none), nullptr = by formula (seg_slot_base)
  const float* render_raw;  // backward only: the forward's accumulated colours [H,W,3] (C_final); with a
                            // composite epilogue they are rebuilt from the finished image instead
  int parts;                // backward: jobs per split tile (1 = whole list)
  int tail;                 // backward: the last `tail` tiles of every XCD's sequence are split (0 = all)
  int prio;                 // backward: issue priority thresholds of the jobs (job_priority), 0 = off
  int fine = FG_SEG_FINE_NEVER;  // the checkpoint grid (jobs_build.h seg_index): the forward's and the list build's
};

// Optional work counters (make stats -> libfgraster_stats.so; never in the product library).
#ifdef FG_RASTER_STATS
__device__ unsigned long long fg_raster_stats[16];
#define FG_STAT(i, n) do { const unsigned long long n_ = (unsigned long long)(n); \
    if (fg::lane_id() == 0) atomicAdd(&fg_raster_stats[i], n_); } while (0)
#else
#define FG_STAT(i, n) do { } while (0)
#endif
#ifdef FG_RASTER_TIMELINE
// Job timeline of the mixed launches (make timeline -> libfgraster_timeline.so): per job start / end on the 100 MHz wall clock, the
// hardware slot it ran on and what it was -- scripts/raster_timeline.py turns it into occupancy over time.
#define FG_TL_CAP (1 << 17)
__device__ unsigned long long fg_timeline[FG_TL_CAP * 6];
__device__ unsigned int fg_timeline_n;
__shared__ unsigned long long fg_tl_acc[7];  // [0] ticks spent staging batches, [1] clock at the first batch,
                                             // [2] job start, [3..6] clocks at FG_TL_MARK(0..3)
#define FG_TL_BEGIN() const unsigned long long tl_t0_ = wall_clock64(); \
  if (threadIdx.x == 0) { fg_tl_acc[0] = fg_tl_acc[1] = fg_tl_acc[3] = fg_tl_acc[4] = fg_tl_acc[5] = fg_tl_acc[6] = 0; fg_tl_acc[2] = tl_t0_; }
#define FG_TL_MARK(i) do { const unsigned long long m_ = wall_clock64(); if (threadIdx.x == 0) fg_tl_acc[3 + (i)] = m_ - fg_tl_acc[2]; } while (0)
#define FG_TL_STAGE_BEGIN() const unsigned long long tl_s0_ = wall_clock64(); \
  if (threadIdx.x == 0 && fg_tl_acc[1] == 0) fg_tl_acc[1] = tl_s0_
#define FG_TL_STAGE_END() do { if (threadIdx.x == 0) fg_tl_acc[0] += wall_clock64() - tl_s0_; } while (0)
#define FG_TL_END(kernel, tile, strip, part, parts) do { if (threadIdx.x == 0) {                                 \
    const unsigned slot_ = atomicAdd(&fg_timeline_n, 1u);                                                          \
    if (slot_ < FG_TL_CAP) {                                                                                       \
      fg_timeline[6 * slot_ + 0] = tl_t0_;                                                                         \
      fg_timeline[6 * slot_ + 1] = wall_clock64();                                                                 \
      fg_timeline[6 * slot_ + 2] = (unsigned long long)(unsigned)__builtin_amdgcn_s_getreg(63492) |                \
                                   ((unsigned long long)((unsigned)__builtin_amdgcn_s_getreg(63508) & 15u) << 32) | \
                                   (fg_tl_acc[0] << 36);                                                           \
      fg_timeline[6 * slot_ + 3] = (unsigned long long)(kernel) | ((unsigned long long)((strip) + 1) << 4) |       \
                                   ((unsigned long long)(part) << 8) | ((unsigned long long)(parts) << 12) |       \
                                   ((unsigned long long)(tile) << 16) |                                            \
                                   ((fg_tl_acc[1] ? fg_tl_acc[1] - tl_t0_ : 0ull) << 40);                          \
      fg_timeline[6 * slot_ + 4] = fg_tl_acc[3] | (fg_tl_acc[4] << 32);                                            \
      fg_timeline[6 * slot_ + 5] = fg_tl_acc[5] | (fg_tl_acc[6] << 32);                                            \
    } } } while (0)
#else
#define FG_TL_BEGIN() do { } while (0)
#define FG_TL_STAGE_BEGIN() do { } while (0)
#define FG_TL_MARK(i) do { } while (0)
#define FG_TL_STAGE_END() do { } while (0)
#define FG_TL_END(kernel, tile, strip, part, parts) do { } while (0)
#endif

// 1: the backward sums its per-splat accumulators through LDS; 0: register butterfly (A/B switch)
#ifndef FG_BWD_LDS_REDUCE
#define FG_BWD_LDS_REDUCE 1
#endif

// 1: the backward evaluates the alpha pre-test of all pixel slots of an entry as independent chains
// before any per-slot branch; 0: slot by slot (A/B switch)
#ifndef FG_BWD_BATCHED_PRETEST
#define FG_BWD_BATCHED_PRETEST 1
#endif

#ifndef FG_FWD_STRIP_TEST_MIN_PPT
#define FG_FWD_STRIP_TEST_MIN_PPT 2
#endif

// The lane id, recomputed where it is used (two instructions the compiler cannot hoist): the
// per-batch staging code of the raster kernels then does not pin its LDS addresses in registers
// across the per-entry loops (one staging pass per 64 entries; the registers are worth a wavefront
// of occupancy in the backward).
__device__ __forceinline__ int fresh_lane_id() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

struct SigmaTerms {
  float hc, bdx, hadx2;
};
__device__ __forceinline__ SigmaTerms sigma_terms(const Splat& s, float dx) {
  constexpr float NL = -1.4426950408889634f;  // -log2(e)
  SigmaTerms t;
  t.hc = (0.5f * NL) * s.c;
  t.bdx = (NL * s.b) * dx;
  t.hadx2 = ((0.5f * NL) * s.a) * dx * dx;
  return t;
}
// The same terms from conic coefficients the loader lane already scaled while staging the record
// in LDS (a' = -log2(e)/2 a, b' = -log2(e) b, c' = -log2(e)/2 c): identical products, three
// multiplies per (entry, wavefront) fewer.
constexpr float FG_NEG_LOG2E = -1.4426950408889634f;
__device__ __forceinline__ SigmaTerms sigma_terms_prescaled(float a_s, float b_s, float c_s, float dx) {
  SigmaTerms t;
  t.hc = c_s;
  t.bdx = b_s * dx;
  t.hadx2 = a_s * dx * dx;
  return t;
}
__device__ __forceinline__ float neg_sigma_log2e(const SigmaTerms& t, float dy) {
  return fmaf(fmaf(t.hc, dy, t.bdx), dy, t.hadx2);
}

// dy of pixel slot k of a lane: (s.y - pyb) - off with pyb = the lane's row of the TILE's first strip
// (+ 0.5) and off = 4 * (strip of the slot), a constant or a scalar.  Every launch shape -- whole
// tile, two strips, one strip, any pixels-per-lane -- evaluates the SAME two subtractions for a
// given pixel, so forward and backward take identical skip decisions whatever jobs they were cut
// into; and a lane keeps one row coordinate instead of one per slot.
template <int PPT>
__device__ __forceinline__ float slot_dy(float dy_base, int wave, int k) {
  // PPT 4: wave == 0, strips 0..3; PPT 2: strips wave, wave + 2; PPT 1: strip wave
  return dy_base - (float)(4 * (wave + k * (4 / PPT)));
}

template <int C>
__device__ __forceinline__ void read_record(const float4* rec, Splat& s, float (&f)[C]) {
  const float4 v0 = rec[0];
  const float4 v1 = rec[1];
  s.x = v0.x; s.y = v0.y; s.o = v0.z; s.a = v0.w; s.b = v1.x; s.c = v1.y;
  float tmp[4 * rec_vec4(C) - 6];
  tmp[0] = v1.z; tmp[1] = v1.w;
#pragma unroll
  for (int v = 2; v < rec_vec4(C); ++v) {
    const float4 q = rec[v];
    tmp[4 * v - 6] = q.x; tmp[4 * v - 5] = q.y; tmp[4 * v - 4] = q.z; tmp[4 * v - 3] = q.w;
  }
#pragma unroll
  for (int c = 0; c < C; ++c) f[c] = tmp[c];
}

// Pixel-centre bounds of a tile and of its four 4-row strips, wave-uniform: ten scalar registers
// (as hoisted vector values they cost ten VGPRs for the whole kernel -- two wavefronts per SIMD of
// occupancy in the backward).
struct StripBounds {
  float x_first, x_last, y_first[4], y_last[4];
};
__device__ __forceinline__ StripBounds strip_bounds(float tile_x0, float tile_y0) {
  StripBounds sb;
  sb.x_first = fg::uniform(tile_x0 + 0.5f);
  sb.x_last = fg::uniform(tile_x0 + ((float)TILE - 0.5f));
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) {
    sb.y_first[s4] = fg::uniform(tile_y0 + 4.f * s4 + 0.5f);
    sb.y_last[s4] = fg::uniform(tile_y0 + 4.f * s4 + 3.5f);
  }
  return sb;
}

// Which of the tile's four 4-row strips can this splat reach with alpha >= 1/255?  (fg::alpha_extent:
// result-preserving by construction; shared with the binning's tight tile rectangles.)
__device__ __forceinline__ unsigned strip_mask(float gx, float gy, float o, float a, float b, float c,
                                               const StripBounds& sb) {
  float ex, ey;
  const int kind = fg::alpha_extent(o, a, b, c, ex, ey);
  if (kind != 1) return kind == 0 ? 0u : 0xFu;
  if (!fg::extent_reaches_bounds(gx, ex, sb.x_first, sb.x_last)) return 0u;
  unsigned m = 0;
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4)
    if (fg::extent_reaches_bounds(gy, ey, sb.y_first[s4], sb.y_last[s4])) m |= 1u << s4;
  return m;
}

// strips (4-row bands of the tile) owned by wavefront `wave`; pixel slot k of a lane lies in
// strip  wave + k * (4 / PPT)
template <int PPT>
__device__ __forceinline__ unsigned wave_strips(int wave) {
  return PPT == 1 ? (1u << wave) : PPT == 2 ? ((1u << wave) | (1u << (wave + 2))) : 0xFu;
}

// Shared staging of one tile job: NT = 64 * (cooperating wavefronts).
template <int C, int NT>
struct FwdShared {
  float4 rec[NT][rec_vec4(C)];
  uint32_t mask[NT];
  uint16_t list[NT / 64][NT];  // per-wavefront compacted (entry | strips << 8)
};

// One tile job: NW cooperating wavefronts (threadIdx.x in [0, 64 NW)), PPT pixels per lane; the
// job's first wavefront owns strip group `wave_base` (0 for a whole tile; the strip index for a
// single-strip job of the mixed launch, PPT == 1 / NW == 1).
//
// HEAVY tiles (fg_raster_config::heavy_tiles; three channels, one wavefront per job): a list of thousands of entries
// that does not saturate its pixels is a serial walk of 100 ns per entry for a wavefront alone on its SIMD (80% of the
// Gaussians in a ball: four strip jobs of 1.5 ms each while the chip idles, profiles/r04_job_timeline.md).  Compositing
// is associative, so such a tile takes three launches:
//   MODE 3 (the main launch, four strip jobs as for any long list) walks only the list's first FG_HEAVY_PREFIX entries
//          and leaves every pixel's state there in the checkpoint slot of that entry (T < 0: the pixel is finished).
//          Most long lists -- a dense, opaque cluster -- saturate inside the prefix and nothing more happens to the tile.
//   MODE 1 ("local" jobs, many per tile, any order; they return at once when the prefix finished the tile) composite each
//          64-entry batch of their share of the rest of the list BY ITSELF -- from T = 1, C = 0 -- and leave (T_b, C_b)
//          per pixel in the batch's slot (T_b < 0: the pixel met the stop rule inside the batch), the pixel's last entry
//          in the slot's byte plane.
//   MODE 2 ("combine" jobs, one per strip) continue from the prefix's state and take a batch in ONE step, C += T C_b,
//          T *= T_b, unless some pixel could stop inside it (T T_b within 1e-5 of the stop threshold, or a local stop):
//          that batch is walked entry by entry from the running state like any other job.  The running state before every
//          batch goes into the slot: the backward's checkpoints.
// Not bit-identical to the serial walk (products and sums are associated differently: 1e-7 relative); the stop decisions
// are the serial walk's own.
template <int C, int PPT, int NW, int MODE = 0>
__device__ __forceinline__ void raster_fwd_body(FwdShared<C, 64 * NW>& sh, int tile, int wave_base, int width,
                                                int height, int tile_w, const float4* __restrict__ splats,
                                                const int32_t* __restrict__ tile_offsets,
                                                const int32_t* __restrict__ flatten_ids, float* __restrict__ render,
                                                float* __restrict__ alphas, int32_t* __restrict__ last_ids,
                                                const Composite& comp, float4* __restrict__ ckpt = nullptr,
                                                uint32_t* __restrict__ live_words = nullptr, int local_part = 0,
                                                const int32_t* __restrict__ slot_tab = nullptr,
                                                int32_t* __restrict__ open_list = nullptr,
                                                long long* __restrict__ walk_out = nullptr,
                                                int seg_fine = FG_SEG_FINE_NEVER) {
  constexpr int NT = 64 * NW;
  constexpr int RSTEP = TILE / PPT;
  constexpr int NV = rec_vec4(C);
  static_assert(MODE == 0 || (C == 3 && NW == 1), "heavy-tile jobs: three channels, one wavefront");
  auto& lds = sh.rec;
  auto& lds_mask = sh.mask;
  auto& lds_list = sh.list;
  const int tile_y = tile / tile_w, tile_x = tile - tile_y * tile_w;
  const int n_tiles = tile_w * ((height + TILE - 1) / TILE);
  const int start = tile_offsets[tile];
  int end = tile_offsets[tile + 1];
  int first = start;  // MODE 1: this job's share of the list behind the prefix, whole batches; MODE 2: all of it
  if constexpr (MODE == 1) {
    // local_part = part | sub << 8: a listed job's batches are taken by FG_HEAVY_SUB workgroups, a quarter each -- the
    // local pass lasts as long as its longest job (19 batches for a 79 000-entry list) while most of the chip idles
    const int per_b = heavy_batches_per_job(end - start), per = per_b * FG_SEG_ENTRIES, off = (local_part & 255) * per;
    const int sub_b = (per_b + FG_HEAVY_SUB - 1) / FG_HEAVY_SUB;
    first = start + FG_HEAVY_PREFIX + off + (local_part >> 8) * sub_b * FG_SEG_ENTRIES;
    end = min(min(end, start + FG_HEAVY_PREFIX + off + per), first + sub_b * FG_SEG_ENTRIES);
    if (first >= end) return;
  }
  if constexpr (MODE == 2) first = start + FG_HEAVY_PREFIX;
  // (MODE 3: local_part = the prefix length when the caller has its own -- the wide jobs' prefix -- else FG_HEAVY_PREFIX)
  const int prefix_len = MODE == 3 && local_part > 0 ? local_part : FG_HEAVY_PREFIX;
  if constexpr (MODE == 3) end = min(end, start + prefix_len);
  const int slot0 = (MODE != 0 || (C == 3 && NW == 1)) && ckpt ? seg_slot_base(slot_tab, start, tile) : -1;
  float4* const slots = slot0 >= 0 ? ckpt + seg_slots_offset4(n_tiles, width, height) : nullptr;
  if constexpr (MODE != 0) {
    if (!slots) return;  // (a heavy tile always owns slots: jobs_build.h)
  }
  // one wavefront per workgroup (NW == 1, the mixed launches): the wave index is the constant 0 and
  // every per-wave LDS address folds into an instruction offset instead of a register
  const int lane = fg::lane_id(), wl = NW == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6),
            wave = wave_base + wl;
  const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const int col = threadIdx.x & 15, row0 = 4 * wave_base + (threadIdx.x >> 4);
  const int ix = tile_x * TILE + col;
  const float px = (float)ix + 0.5f;
  const StripBounds sb = strip_bounds((float)(tile_x * TILE), (float)(tile_y * TILE));
  const unsigned my_strips = wave_strips<PPT>(wave);

  // Per-slot pixel state.  The "finished" flags live as 64-bit lane masks in scalar registers and
  // every predicate below is built from vector-compare results with scalar logic: as bool
  // variables the compiler kept them as 0/1 VGPRs and paid 5-6 vector instructions per slot to
  // convert back and forth (ISA of the bool version: v_and/v_cmp_eq/v_cndmask per flag per slot).
  float T[PPT], acc[PPT][C];
  int last[PPT];
  uint64_t done[PPT], outside[PPT];
  // the lane's row in the tile's FIRST strip (slot_dy adds the strip)
  const float pyb = (float)(tile_y * TILE + (int)(threadIdx.x >> 4) - 4 * wl) + 0.5f;
  const uint64_t full = __ballot(true);
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const int iy = tile_y * TILE + row0 + k * RSTEP;
    T[k] = 1.f;
    last[k] = start - 1;
    done[k] = outside[k] = __ballot(!(ix < width && iy < height));
#pragma unroll
    for (int c = 0; c < C; ++c) acc[k][c] = 0.f;
  }

  if constexpr (MODE == 1 || MODE == 2) {
    // the state the prefix jobs left at entry start + FG_HEAVY_PREFIX -- in the slot of the tile's FIRST batch, which
    // nothing else uses (no checkpoint in front of the first entry; the slot of entry start + FG_HEAVY_PREFIX takes that
    // batch's local composite, then its checkpoint)
    const float4* at = slots + seg_slot_index(slot0, start, start) * FG_SEG_SLOT4;
    bool open = false;
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      const float4 v = at[(row0 + k * RSTEP) * TILE + col];
      open = open || v.x > 0.f;
      if constexpr (MODE == 2) {
        const int iy = min(tile_y * TILE + row0 + k * RSTEP, height - 1);
        T[k] = fabsf(v.x);
        acc[k][0] = v.y; acc[k][1] = v.z; acc[k][2] = v.w;
        done[k] |= __ballot(!(v.x > 0.f));
        last[k] = last_ids[(size_t)iy * width + min(ix, width - 1)];
      }
    }
    if (!__any(open)) return;  // the prefix finished every pixel: its outputs stand
  }
  int batch = first;
  int evaluated = 0;  // (uniform) entries that reached this job's strips so far: what it reports as its work (FG_WALK_REPORT)
  // (Round 5, measured and dropped: the ids of batch b + 1 fetched while batch b is walked -- one register, 62 -> 64 VGPRs.  A
  // batch's staging is two dependent round trips, the id, then the record: a third of a strip job's life on a long list once
  // its SIMD's other wavefronts have gone.  Forward +-0 on the uniform scene and on half of the Gaussians in a ball, +8 us
  // with 30 % needles: the exposed part is the record gather and the barrier, not the id.)
  while (batch < end) {
    uint64_t all_done = full;
#pragma unroll
    for (int k = 0; k < PPT; ++k) all_done &= done[k];
    if constexpr (MODE == 1) {
      // a batch by itself: from T = 1, C = 0, nothing taken
      __syncthreads();
#pragma unroll
      for (int k = 0; k < PPT; ++k) {
        T[k] = 1.f;
        last[k] = batch - 1;
        done[k] = outside[k];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[k][c] = 0.f;
      }
    } else {
      // barrier (protects the LDS batch of the previous iteration) + tile-wide early exit
      if (__syncthreads_and(all_done == full)) break;
    }
    if constexpr (MODE == 2) {
      // batches nobody can stop in, taken whole: FG_HEAVY_AHEAD slots are fetched together (one round trip), then
      // applied in order up to the first that has to be walked
      constexpr int D = FG_HEAVY_AHEAD;
      static_assert(PPT == 1, "combine jobs: one strip, one pixel per lane");
      const int pix = row0 * TILE + col;
      float4 L[D];
      uint32_t lb[D];
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const int bd = min(batch + d * FG_SEG_ENTRIES, end - 1);
        const float4* slot = slots + seg_slot_index(slot0, start, bd) * FG_SEG_SLOT4;
        L[d] = slot[pix];
        lb[d] = reinterpret_cast<const uint8_t*>(slot + TILE * TILE)[pix];
      }
      bool walk = false;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        if (walk || batch >= end) break;  // (uniform)
        const bool may_stop = L[d].x < 0.f || T[0] * L[d].x <= FG_T_STOP * 1.00001f;
        const uint64_t stops = __ballot(may_stop) & ~done[0];
        if (stops != 0ull) {
          walk = true;
          break;
        }
        // the state BEFORE the batch: the backward's checkpoint (over the local result just read)
        if (batch > start) slots[seg_slot_index(slot0, start, batch) * FG_SEG_SLOT4 + pix] = make_float4(T[0], acc[0][0], acc[0][1], acc[0][2]);
        const float t_in = lane_select0(~done[0], T[0]);  // (finished pixels: nothing is added, nothing changes)
        acc[0][0] = fmaf(t_in, L[d].y, acc[0][0]);
        acc[0][1] = fmaf(t_in, L[d].z, acc[0][1]);
        acc[0][2] = fmaf(t_in, L[d].w, acc[0][2]);
        T[0] = lane_select(~done[0], T[0] * L[d].x, T[0]);
        last[0] = (lb[d] != 0u && !((done[0] >> lane) & 1ull)) ? batch + (int)lb[d] - 1 : last[0];
        batch += FG_SEG_ENTRIES;
      }
      if (!walk) continue;  // (all D taken, or the list's end reached: the loop condition decides)
    }
    if constexpr (C == 3 && NW == 1 && MODE != 1) {
      // checkpoint for the segmented backward (struct Segments): the state before entry `batch`
      if (slots && batch > start && ((batch - start) & (FG_SEG_ENTRIES - 1)) == 0 && seg_starts_at(batch - start, seg_fine)) {
        float4* slot = slots + seg_slot_index(slot0, start, batch, seg_fine) * FG_SEG_SLOT4;
#pragma unroll
        for (int k = 0; k < PPT; ++k)
          slot[(row0 + k * RSTEP) * TILE + col] = make_float4(T[k], acc[k][0], acc[k][1], acc[k][2]);
      }
    }
    FG_TL_STAGE_BEGIN();
    const int idx = batch + (int)threadIdx.x;
    unsigned mask = 0;
    if (idx < end) {
      const float4* rec = splats + (size_t)flatten_ids[idx] * (FG_SPLAT_FLOATS / 4);
      float4 v[NV];
#pragma unroll
      for (int q = 0; q < NV; ++q) v[q] = rec[q];
      mask = strip_mask(v[0].x, v[0].y, v[0].z, v[0].w, v[1].x, v[1].y, sb);
      // the forward only ever uses the conic inside the exponent: stage it pre-scaled
      v[0].w *= 0.5f * FG_NEG_LOG2E;
      v[1].x *= FG_NEG_LOG2E;
      v[1].y *= 0.5f * FG_NEG_LOG2E;
#pragma unroll
      for (int q = 0; q < NV; ++q) lds[threadIdx.x][q] = v[q];
    }
    lds_mask[threadIdx.x] = mask;
    __syncthreads();
    FG_TL_STAGE_END();
    // Each wavefront compacts the entries that can reach its own strips into a private index list
    // (list order preserved: ballot + popcount prefix), then walks it with a plain counted loop.
    // The scalar unit is shared by the CU's four SIMDs and was the busiest pipe of this kernel
    // (PMC: SALU ~0.6x VALU instructions): a bit-scan walk costs ~10 scalar ops per entry more.
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) {
      const unsigned m = lds_mask[64 * i + lane];
      const bool rel = (m & my_strips) != 0u;
      const uint64_t bal = __ballot(rel);
      if (rel) lds_list[wl][cnt + __popcll(bal & lt_mask)] = (uint16_t)((64 * i + lane) | (m << 8));
      cnt += __popcll(bal);
    }
    __builtin_amdgcn_wave_barrier();  // the list is private to this wavefront
    evaluated += cnt;
    // LIVENESS for the backward (one wavefront per job only): bit j of live[k] = "list entry batch + j
    // had a contributing pixel in slot k".  Written out per batch, one byte per (entry, strip); the
    // backward then evaluates exactly the (entry, strip) pairs that did something here instead of
    // re-testing every strip of every entry -- the compares and the exponential of those tests are
    // the most expensive instructions of its loop (vector issue costs).
    // (kept in lane j of a vector register, set with v_writelane: the scalar unit, shared by the CU's
    // four SIMDs, is this kernel's busiest pipe -- as 64-bit scalar masks the bookkeeping cost the
    // forward 0.211 -> 0.223 ms)
    int live[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) live[k] = 0;
    // "every pixel of this wavefront is finished" is re-checked every 8 entries only: the check is
    // six scalar instructions and the scalar unit, shared by the CU's four SIMDs, is this kernel's
    // busiest pipe (an entry walked after the last pixel finished finds no valid lane and does nothing)
    for (int n0 = 0; n0 < cnt; n0 += 8) {
      all_done = full;
#pragma unroll
      for (int k = 0; k < PPT; ++k) all_done &= done[k];
      if (all_done == full) break;  // this wavefront has nothing left to do
      const int n1 = min(n0 + 8, cnt);
      // The list word of iteration n+1 is read during iteration n: the per-entry chain list word ->
      // record -> arithmetic was two LDS round trips long and a SIMD's 8 wavefronts did not cover it
      // (masking 5% of the slot tests away changed nothing; this: 0.211 -> 0.201 ms).  Fetching the whole
      // record one entry ahead as well, two register sets taking turns, loses again: 72 VGPRs, 0.217 ms.
      unsigned next_v = lds_list[wl][n0];
      for (int n = n0; n < n1; ++n) {
      // the entry is wave-uniform: move it to a scalar register so that the record address and
      // the list index are scalar arithmetic (as a vector value the compiler spent a quarter-rate
      // v_mul_lo_u32 per entry on the address)
      FG_STAT(8, 1);
      const unsigned packed = __builtin_amdgcn_readfirstlane(next_v);
      next_v = lds_list[wl][min(n + 1, NT - 1)];
      const int j = packed & 255u;
      Splat s;
      float f[C];
      read_record<C>(lds[j], s, f);
      const float dx = s.x - px, dy_base = s.y - pyb;
      const SigmaTerms st = sigma_terms_prescaled(s.a, s.b, s.c, dx);
#pragma unroll
      for (int k = 0; k < PPT; ++k) {
        // PPT == 4: skip the strips the splat cannot reach; with 1-2 slots per lane the test
        // costs more scalar work than it saves
        if (PPT >= FG_FWD_STRIP_TEST_MIN_PPT && !((packed >> (8 + wave + k * (4 / PPT))) & 1u)) continue;
        // one wave-uniform branch, then select-predicated straight-line code (no nested
        // divergent ifs: each costs exec save/restore and merge copies)
        const float dy = slot_dy<PPT>(dy_base, wave, k);
        const float e2 = neg_sigma_log2e(st, dy);
        const float alpha = fminf(FG_ALPHA_MAX, s.o * __builtin_amdgcn_exp2f(e2));
        // valid = !done && !(sigma < 0 || alpha < 1/255)
        const uint64_t valid = lanes_ule(e2, 0.f) & lanes_oge(alpha, FG_ALPHA_SKIP) & ~done[k];
        FG_STAT(9, 1);
        if (valid == 0ull) continue;
        FG_STAT(10, 1);
        FG_STAT(11, __popcll(valid));
        // (j comes from a scalar ALU instruction and live[k] was last written dozens of instructions
        // ago: none of gfx950's lane-access wait states applies)
        if (NW == 1) asm("v_writelane_b32 %0, 1, %1" : "+v"(live[k]) : "s"(j));
        const float next_T = T[k] * (1.f - alpha);
        const uint64_t stop = valid & lanes_ole(next_T, FG_T_STOP);
        const uint64_t take = valid & ~stop;
        const float vis = lane_select0(take, alpha * T[k]);
#pragma unroll
        for (int c = 0; c < C; ++c) acc[k][c] += f[c] * vis;
        last[k] = lane_select(take, batch + j, last[k]);
        T[k] = lane_select(take, next_T, T[k]);
        done[k] |= stop;
      }
      }
    }
    if constexpr (NW == 1) {
      if (live_words && batch + lane < end) {
        // lane j flushes entry batch + j: byte s of the word = strip s (entries this wavefront never
        // listed, or never reached because all its pixels were done, get an explicit 0)
        if constexpr (PPT == 4) {
          live_words[batch + lane] = (uint32_t)live[0] | ((uint32_t)live[1] << 8) | ((uint32_t)live[2] << 16) |
                                     ((uint32_t)live[3] << 24);
        } else {
          uint8_t* bytes = reinterpret_cast<uint8_t*>(live_words + (batch + lane));
#pragma unroll
          for (int k = 0; k < PPT; ++k) bytes[wave + k * (4 / PPT)] = (uint8_t)live[k];
        }
      }
    }
    if constexpr (MODE == 1) {
      // the batch's own composite; T < 0: the pixel met the stop rule inside (the combine job walks the batch then)
      float4* slot = slots + seg_slot_index(slot0, start, batch) * FG_SEG_SLOT4;
      uint8_t* slot_last = reinterpret_cast<uint8_t*>(slot + TILE * TILE);
#pragma unroll
      for (int k = 0; k < PPT; ++k) {
        const bool stopped = ((done[k] & ~outside[k]) >> lane) & 1ull;
        const int pix = (row0 + k * RSTEP) * TILE + col;
        slot[pix] = make_float4(stopped ? -1.f : T[k], acc[k][0], acc[k][1], acc[k][2]);
        slot_last[pix] = (uint8_t)(last[k] >= batch ? last[k] - batch + 1 : 0);
      }
    }
    batch += NT;
  }
  if constexpr (MODE == 1) return;
  if constexpr (MODE == 3) {
    // the state at the end of the prefix, for the local and combine jobs (T < 0: finished, or outside the image)
    if (slots && tile_offsets[tile + 1] - start > prefix_len) {
      float4* at = slots + seg_slot_index(slot0, start, start) * FG_SEG_SLOT4;
      uint64_t all_done = full;
#pragma unroll
      for (int k = 0; k < PPT; ++k) {
        const bool fin = (done[k] >> lane) & 1ull;
        at[(row0 + k * RSTEP) * TILE + col] = make_float4(fin ? -T[k] : T[k], acc[k][0], acc[k][1], acc[k][2]);
        all_done &= done[k];
      }
      // wide jobs: the strips still open, as a list for the launch behind this one ([0] = how many, entries from [8])
      if (open_list && all_done != full && lane == 0) open_list[8 + atomicAdd(&open_list[0], 1)] = tile << 3 | (wave_base + 1);
    }
  }

  if constexpr (C == 3 && NW == 1) {
    if (ckpt) {  // the table in front of the checkpoint slots (seg_slots_offset4)
      int32_t* tl = reinterpret_cast<int32_t*>(ckpt) + 4 * tile;
      int walked = 0;
#pragma unroll
      for (int k = 0; k < PPT; ++k) {
        const int m = fg::wave_max_i32(last[k]);
        if (lane == 0) tl[wave + k * (4 / PPT)] = m;
        walked = max(walked, m - start + 1);
      }
      // LONG JOBS reported to the host (nullable; pinned memory, system scope; any of them, not the longest): a job that
      // evaluated more than FG_WALK_REPORT entries for its strips is what heavy tiles are for -- the host turns them on by it
      (void)walked;
      if (MODE == 0 && walk_out && evaluated > FG_WALK_REPORT && lane == 0)
        __hip_atomic_store(walk_out, (long long)evaluated, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const int iy = tile_y * TILE + row0 + k * RSTEP;
    if (ix < width && iy < height) {
      const size_t pix = (size_t)iy * width + ix;
      const float alpha_out = 1.f - T[k];
      if constexpr (C == 3 && NW == 1) {
        if (ckpt) reinterpret_cast<float*>(ckpt + seg_plane_offset4(n_tiles))[pix] = T[k];  // exact T_final
      }
      if (comp.background || comp.n_clamp > 0) {
        const float om = 1.f - alpha_out;  // as the host expression (1 - alpha) rounds
        unsigned blocked = 0;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          float v = acc[k][c];
          if (comp.background) v += om * comp.background[c];
          if (c < comp.n_clamp) {
            if (v < 0.f || v > 1.f) blocked |= 1u << c;
            v = fminf(fmaxf(v, 0.f), 1.f);
          }
          acc[k][c] = v;
        }
        if (comp.n_clamp > 0) comp.clamp_mask[pix] = (uint8_t)blocked;
      }
#pragma unroll
      for (int c = 0; c < C; ++c) render[pix * C + c] = acc[k][c];
      alphas[pix] = alpha_out;
      last_ids[pix] = last[k];
    }
  }
}


template <int C, int PPT>
__global__ void __launch_bounds__(256 / PPT)
raster_fwd_kernel(int width, int height, int tile_w, int tile_h, int order_mode, const float4* __restrict__ splats,
                  const int32_t* __restrict__ tile_offsets, const int32_t* __restrict__ flatten_ids,
                  float* __restrict__ render, float* __restrict__ alphas, int32_t* __restrict__ last_ids,
                  Composite comp) {
  constexpr int NW = 256 / PPT / 64;
  __shared__ FwdShared<C, 64 * NW> sh;
  const int tile = tile_of_block(blockIdx.x, tile_w * tile_h, tile_w, tile_h, order_mode);
  if (tile < 0) return;
  raster_fwd_body<C, PPT, NW>(sh, tile, 0, width, height, tile_w, splats, tile_offsets, flatten_ids, render, alphas,
                              last_ids, comp);
}

// Mixed launch (one wavefront per workgroup): every XCD walks its band column-major as above; the
// first tiles of its sequence are whole-tile jobs (4 pixels per lane), the last `tail_tiles` are
// split into four single-strip jobs (1 pixel per lane) -- see launch_fwd_mixed.
// ISSUE PRIORITY by expected job length (fg_raster_config::prio_fwd / prio_bwd: lo | hi << 16 in percent of the mean
// single-strip job; 0 = off).  The launches end when their longest jobs end; those start in the first microsecond and
// then share their SIMD's issue slots evenly with 5-7 shorter jobs (profiles/r04_job_timeline.md).  s_setprio raises
// a wavefront's priority at its SIMD's instruction arbiter: the long jobs get through sooner, the short ones -- which
// have slack -- a little later; the work is the same.  weight8: cost per list entry of the job's kind in eighths of a
// single strip's.  Measured (profiles/r04_job_priority.md): uniform scene +-0, clustered scenes -2.5 % / -5 % per step;
// with low thresholds in the forward (every whole-tile job raised) the uniform forward loses 9 %.
__device__ __forceinline__ void job_priority(const int32_t* __restrict__ tile_offsets, int n_tiles, int len, int weight8,
                                             int prio) {
  if (prio <= 0) return;
  const long long total = tile_offsets[n_tiles];
  const long long lhs = (long long)len * weight8 * n_tiles * 100, ref = total * 8;
  const int lo = prio & 0xFFFF, hi = prio >> 16;
  if (hi > 0 && lhs > ref * hi) __builtin_amdgcn_s_setprio(3);
  else if (lhs > ref * lo) __builtin_amdgcn_s_setprio(2);
}

// (three channels: eight wavefronts per SIMD asked for -- 64 registers, which the allocator meets without spilling; the
// heavy tiles' local jobs would otherwise cost the launch its eighth wavefront, 66 registers)
template <int C>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(C == 3 ? 8 : 4, 8)))
raster_fwd_mixed_kernel(int width, int height, int tile_w, int tile_h, int nx, int tail_tiles,
                        const int32_t* __restrict__ jobs, int cap,
                        const float4* __restrict__ splats, const int32_t* __restrict__ tile_offsets,
                        const int32_t* __restrict__ flatten_ids, float* __restrict__ render,
                        float* __restrict__ alphas, int32_t* __restrict__ last_ids, Composite comp,
                        float4* __restrict__ ckpt, uint32_t* __restrict__ live_words,
                        float4* __restrict__ zero4, long long zero_n4, const int32_t* __restrict__ slot_tab, int prio,
                        int prefix_len, int32_t* __restrict__ open_list, long long* __restrict__ walk_out, int seg_fine) {
  __shared__ FwdShared<C, 64> sh;
  // The record-gradient array of the coming backward is zero-filled here, a slice per workgroup: this
  // kernel leaves most of the memory pipe idle, a separate fill launch costs ~10 us plus its boundary.
  if (zero4) {
    const long long per = (zero_n4 + gridDim.x - 1) / gridDim.x;
    const long long z0 = per * blockIdx.x, z1 = min(zero_n4, z0 + per);
    for (long long z = z0 + threadIdx.x; z < z1; z += 64) zero4[z] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // With a list the grid covers the positional job count plus a margin; fg_raster_build_jobs
  // makes the list fit it (build_jobs_kernel).  (A grid of the list's full capacity -- 4 jobs per
  // tile -- left tens of thousands of empty workgroups to dispatch: 2160p forward 0.52 -> 0.64 ms;
  // workgroups walking on through a longer list cost 12-19 registers in both kernels.)
  int strip;
  bool no_ckpt = false, prefix = false;
  const int tile = jobs ? job_from_list(blockIdx.x, jobs, cap, strip, &no_ckpt, &prefix)
                        : job_of_block(blockIdx.x, tile_w, tile_h, nx, tail_tiles, strip);
  if (tile < 0) return;
  if (no_ckpt) ckpt = nullptr;
  FG_TL_BEGIN();
  if (prio > 0) {
    const int len = tile_offsets[tile + 1] - tile_offsets[tile];
    job_priority(tile_offsets, tile_w * tile_h, prefix ? min(len, prefix_len) : len, strip < 0 ? 13 : (strip >= 4 ? 10 : 8), prio);
  }
  if constexpr (C == 3) {
    if (prefix && ckpt) {  // a strip of a heavy tile: the list's first FG_HEAVY_PREFIX entries only
      raster_fwd_body<C, 1, 1, 3>(sh, tile, strip, width, height, tile_w, splats, tile_offsets, flatten_ids, render, alphas,
                                  last_ids, comp, ckpt, live_words, prefix_len, slot_tab, open_list, nullptr, seg_fine);
      FG_TL_END(1, tile, strip, 1, 1);
      return;
    }
  }
  if (strip < 0)
    raster_fwd_body<C, 4, 1>(sh, tile, 0, width, height, tile_w, splats, tile_offsets, flatten_ids, render, alphas,
                             last_ids, comp, ckpt, live_words, 0, slot_tab, nullptr, walk_out, seg_fine);
  else if (strip >= 4)
    raster_fwd_body<C, 2, 1>(sh, tile, strip - 4, width, height, tile_w, splats, tile_offsets, flatten_ids, render,
                             alphas, last_ids, comp, ckpt, live_words, 0, slot_tab, nullptr, walk_out, seg_fine);
  else
    raster_fwd_body<C, 1, 1>(sh, tile, strip, width, height, tile_w, splats, tile_offsets, flatten_ids, render,
                             alphas, last_ids, comp, ckpt, live_words, 0, slot_tab, nullptr, walk_out, seg_fine);
  FG_TL_END(1, tile, strip, 0, 1);
}

// entry v of the eight segments of a list taken as one (jobs[0..7] = entries per segment), or -1 behind the last
__device__ __forceinline__ int job_of_all_segments(const int32_t* __restrict__ jobs, int cap, int v) {
#pragma unroll
  for (int x = 0; x < 8; ++x) {
    const int n = jobs[x];
    if (v < n) return jobs[8 + x * cap + v];
    v -= n;
  }
  return -1;
}
// The heavy tiles' local jobs (raster_fwd_body MODE 1) and combine jobs (MODE 2): the two launches behind
// raster_fwd_mixed_kernel.  jobs = the launch's own list behind the main one (jobs_build.h); workgroup b takes jobs
// b, b + gridDim, ... of the eight segments taken as one list (the host cannot know how many there are).
__global__ void __launch_bounds__(64)
raster_fwd_local_kernel(int width, int height, int tile_w, const int32_t* __restrict__ jobs,
                        const float4* __restrict__ splats, const int32_t* __restrict__ tile_offsets,
                        const int32_t* __restrict__ flatten_ids, float* __restrict__ render, float* __restrict__ alphas,
                        int32_t* __restrict__ last_ids, Composite comp, float4* __restrict__ ckpt,
                        uint32_t* __restrict__ live_words, const int32_t* __restrict__ slot_tab) {
  __shared__ FwdShared<3, 64> sh;
  // (the eight XCD segments as ONE list: the heavy tiles sit under one or two XCDs' bands, their jobs are for the chip)
  for (int v = blockIdx.x;; v += gridDim.x) {
    const int e = job_of_all_segments(jobs, FG_LOCAL_CAP, v / FG_HEAVY_SUB);
    if (e < 0) break;
    FG_TL_BEGIN();
    raster_fwd_body<3, 4, 1, 1>(sh, e >> 8, 0, width, height, tile_w, splats, tile_offsets, flatten_ids, render, alphas,
                                last_ids, comp, ckpt, live_words, (e & 255) | (v % FG_HEAVY_SUB) << 8, slot_tab);
    FG_TL_END(1, e >> 8, 5, 0, 1);
  }
}
__global__ void __launch_bounds__(64)
raster_fwd_combine_kernel(int width, int height, int tile_w, const int32_t* __restrict__ jobs,
                          const float4* __restrict__ splats, const int32_t* __restrict__ tile_offsets,
                          const int32_t* __restrict__ flatten_ids, float* __restrict__ render, float* __restrict__ alphas,
                          int32_t* __restrict__ last_ids, Composite comp, float4* __restrict__ ckpt,
                          uint32_t* __restrict__ live_words, const int32_t* __restrict__ slot_tab) {
  __shared__ FwdShared<3, 64> sh;
  for (int v = blockIdx.x;; v += gridDim.x) {
    const int e = job_of_all_segments(jobs, FG_HEAVY_CAP, v);
    if (e < 0) break;
    FG_TL_BEGIN();
    raster_fwd_body<3, 1, 1, 2>(sh, e >> 3, (e & 7) - 1, width, height, tile_w, splats, tile_offsets, flatten_ids, render,
                                alphas, last_ids, comp, ckpt, live_words, 0, slot_tab);
    FG_TL_END(1, e >> 3, (e & 7) - 1, 0, 2);
  }
}

// WIDE jobs (fg_raster_config::heavy_wide, round 5): what a heavy tile's prefix jobs leave open, by ONE workgroup of
// FG_WIDE_WAVES wavefronts per strip in a launch behind the main one -- instead of round 4's local + combine launches.
//   main launch               four single-strip PREFIX jobs per heavy tile (MODE 3) walk the list's first FG_WIDE_PREFIX entries as
//                             any strip job would -- a tile that closes its pixels there (a dense, opaque cluster) costs what
//                             it costs without heavy tiles -- and leave every pixel's state in the tile's first checkpoint slot.
//   raster_fwd_wide_kernel    a strip with pixels still open continues in ROUNDS of FG_WIDE_WAVES 64-entry batches.  In a round
//                             every wavefront
//     1. stages ITS batch (its own LDS copy) and composites it by itself -- from T = 1, C = 0 -- for the pixels open at the
//        round's start, and leaves (T_b, C_b, last entry taken) per pixel in LDS (T_b < 0: the pixel met the stop rule inside
//        the batch);
//     2. folds the batches in front of its own onto the round's starting state ("base"), per pixel: the state BEFORE its batch
//        -- the backward's checkpoint -- unless an earlier batch of the round MAY STOP the pixel (a local stop, or the running
//        T within 1e-5 of the threshold: the combine jobs' rule);
//     3. the wavefront of the FIRST batch that may stop a pixel walks its batch again for those pixels, from the true state,
//        entry by entry -- every wavefront for its own pixels at the same time -- and publishes the outcome as the pixel's new
//        base (stopped, or the state behind that batch); 2-3 are repeated only if a walked pixel did NOT stop (normally one
//        walk: a pixel stops once);
//     4. writes its batch's liveness bytes: an entry is live iff it passed the alpha test of a pixel that had not stopped.
//   The round's end state is the next round's base; the job ends with the round in which its last pixel stops: at most one
//   round of batches is composited in vain and nothing goes through memory but the checkpoints.  A round costs about two batch
//   walks of latency whatever its length -- 1024 entries in 10-15 us against 110-170 us of a lone wavefront's serial walk --
//   but holds half a CU meanwhile (measured: every strip of every list beyond 2560 entries as a wide job from its first entry,
//   beside the main launch on a second stream, took 80 % of the Gaussians in a ball of 0.2 from 0.69 to 0.34 ms forward and
//   cost every other layout 20-120 %, profiles/r05_wide_jobs.md): a tool for the lists that are long AND stay open, which is
//   what the prefix finds out.  Stop decisions are the serial walk's own (step 3 is the serial walk); sums and products
//   associate as in the three-launch form (1e-7 relative to the serial walk).
#ifndef FG_WIDE_WAVES
#define FG_WIDE_WAVES 16
#endif
#ifndef FG_WIDE_GRID
#define FG_WIDE_GRID 256  // workgroups of the launch (an empty launch of 512 cost 11 us)
#endif
#ifndef FG_WIDE_PREFIX
#define FG_WIDE_PREFIX 512  // entries of a heavy tile's list the four serial strip jobs walk first (a multiple of 64)
// (1536 while the host turned heavy tiles on by the longest LIST: what a strip job walks while the main launch lasts, so that
// a scene whose long lists close early lost nothing.  Since it goes by reported long WALKS -- fg_raster_jobs_fwd walk_out --
// heavy tiles are on only where strips stay open for thousands of entries, and there a short prefix and a low threshold win:
// 80 % of the Gaussians in a ball of 0.2, prefix / threshold 1536 / 2560: forward 0.492 ms, 1024 / 1280: 0.446, 768 / 1024:
// 0.434, 512 / 768: 0.397, 256 / 512: 0.440; profiles/r05_uneven_splits.md section 4)
#endif
static_assert(FG_WIDE_PREFIX % FG_SEG_ENTRIES == 0 && FG_WIDE_PREFIX >= FG_SEG_ENTRIES, "whole batches");
template <int NWV>
struct WideShared {
  FwdShared<3, 64> stage[NWV];  // a wavefront's batch
  float4 part[NWV][64];         // (T_b | -1, C_b) of the batch by itself, per pixel of the strip
  int32_t part_last[NWV][64];   // the last entry the pixel took in the batch, -1: none
  float4 base[64];              // per pixel (T, C) behind batch base_w of the round (-1: at the round's start)
  int32_t base_last[64];
  int32_t base_w[64];
  int32_t base_fin[64];         // the pixel has stopped (or lies outside the image)
  uint32_t live_or[NWV][2];     // step 4's OR over the lanes
};

template <int NWV>
__device__ __forceinline__ void raster_fwd_wide_body(WideShared<NWV>& sh, int tile, int strip, int width, int height,
                                                     int tile_w, const float4* __restrict__ splats,
                                                     const int32_t* __restrict__ tile_offsets,
                                                     const int32_t* __restrict__ flatten_ids, float* __restrict__ render,
                                                     float* __restrict__ alphas, int32_t* __restrict__ last_ids,
                                                     const Composite& comp, float4* __restrict__ ckpt,
                                                     uint32_t* __restrict__ live_words,
                                                     const int32_t* __restrict__ slot_tab, long long* __restrict__ walk_out,
                                                     int seg_fine) {
  constexpr int C = 3, NV = rec_vec4(C);
  constexpr float MAY_STOP = FG_T_STOP * 1.00001f;
  const int tile_y = tile / tile_w, tile_x = tile - tile_y * tile_w;
  const int n_tiles = tile_w * ((height + TILE - 1) / TILE);
  const int start = tile_offsets[tile], end = tile_offsets[tile + 1];
  const int slot0 = ckpt ? seg_slot_base(slot_tab, start, tile) : -1;
  if (slot0 < 0 || end - start <= FG_WIDE_PREFIX) return;  // (a heavy tile owns slots and is longer than its prefix: jobs_build.h)
  float4* const slots = ckpt + seg_slots_offset4(n_tiles, width, height);
  const int lane = fg::lane_id(), w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  FwdShared<3, 64>& st = sh.stage[w];
  const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const int col = lane & 15, row = 4 * strip + (lane >> 4);
  const int ix = tile_x * TILE + col, iy = tile_y * TILE + row;
  const bool inside = ix < width && iy < height;
  const size_t pix = (size_t)min(iy, height - 1) * width + min(ix, width - 1);
  const float px = (float)ix + 0.5f;
  const StripBounds sb = strip_bounds((float)(tile_x * TILE), (float)(tile_y * TILE));
  // (the lane's row in the tile's FIRST strip: the pixel's dy by the same two subtractions as in every other job, slot_dy)
  const float pyb = (float)(tile_y * TILE + (lane >> 4)) + 0.5f;
  const float strip_off = (float)(4 * strip);
  const uint64_t full = __ballot(true);
  {
    // the state the prefix job of this strip left (T < 0: finished there, or outside the image)
    const float4 v = slots[seg_slot_index(slot0, start, start) * FG_SEG_SLOT4 + row * TILE + col];
    if (!__any(v.x > 0.f)) return;  // (every wavefront reads the same words) the prefix closed the strip: its outputs stand
    if (w == 0) {
      sh.base[lane] = make_float4(fabsf(v.x), v.y, v.z, v.w);
      sh.base_last[lane] = last_ids[pix];
      sh.base_w[lane] = -1;
      sh.base_fin[lane] = !(v.x > 0.f);
    }
  }
  int cnt = 0;  // entries of this wavefront's batch that reach the strip (its private list)
  int ev_mine = 0, sc_mine = 0;  // ... summed over this wavefront's batches, and the entries those batches held (the job's report)
  // one batch, entry by entry, from the state given (raster_fwd_body's loop, one pixel per lane).  BITS: bit j of
  // vlo | vhi << 32 = "entry batch + j passed this lane's alpha test" (step 4)
  uint32_t vlo = 0, vhi = 0;
  auto walk = [&](int batch, float& T, float (&acc)[C], int& last, uint64_t& done, auto bits) {
    for (int n0 = 0; n0 < cnt; n0 += 8) {
      if (done == full) break;
      const int n1 = min(n0 + 8, cnt);
      unsigned next_v = st.list[0][n0];
      for (int n = n0; n < n1; ++n) {
        const unsigned packed = __builtin_amdgcn_readfirstlane(next_v);
        next_v = st.list[0][min(n + 1, 63)];
        const int j = packed & 255u;
        Splat s;
        float f[C];
        read_record<C>(st.rec[j], s, f);
        const float dx = s.x - px, dy_base = s.y - pyb;
        const SigmaTerms sg = sigma_terms_prescaled(s.a, s.b, s.c, dx);
        const float dy = dy_base - strip_off;
        const float e2 = neg_sigma_log2e(sg, dy);
        const float alpha = fminf(FG_ALPHA_MAX, s.o * __builtin_amdgcn_exp2f(e2));
        const uint64_t valid = lanes_ule(e2, 0.f) & lanes_oge(alpha, FG_ALPHA_SKIP) & ~done;
        if (valid == 0ull) continue;
        if constexpr (decltype(bits)::value) {
          if (j < 32) vlo |= (uint32_t)lane_select(valid, 1 << j, 0);
          else vhi |= (uint32_t)lane_select(valid, 1 << (j - 32), 0);
        }
        const float next_T = T * (1.f - alpha);
        const uint64_t stop = valid & lanes_ole(next_T, FG_T_STOP);
        const uint64_t take = valid & ~stop;
        const float vis = lane_select0(take, alpha * T);
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] += f[c] * vis;
        last = lane_select(take, batch + j, last);
        T = lane_select(take, next_T, T);
        done |= stop;
      }
    }
  };

  // This wavefront's entry of the NEXT round is fetched a round ahead -- the id at the head of a round, the record behind the
  // walk, while the round's folding and voting go on: a round was two dependent round trips to memory before anything else
  // (a third of its 15 us).
  float4 v_next[NV];
  {
    const int idx = start + FG_WIDE_PREFIX + 64 * w + lane;
    if (idx < end) {
      const float4* rec = splats + (size_t)flatten_ids[idx] * (FG_SPLAT_FLOATS / 4);
#pragma unroll
      for (int q = 0; q < NV; ++q) v_next[q] = rec[q];
    }
  }
  for (int r0 = start + FG_WIDE_PREFIX; r0 < end; r0 += 64 * NWV) {
    __syncthreads();  // the base is settled (and nobody reads the last round's parts any more)
    const uint64_t done0 = __ballot(sh.base_fin[lane] != 0);
    if (done0 == full) break;  // (the same words for every wavefront: uniform over the workgroup)
    const int batch = r0 + 64 * w;
    const int idx_next = batch + 64 * NWV + lane;
    const int gid_next = idx_next < end ? flatten_ids[idx_next] : 0;
    // 1. the batch by itself
    float T = 1.f, acc[C] = {0.f, 0.f, 0.f};
    int last = -1;
    uint64_t done = done0;
    cnt = 0;
    vlo = vhi = 0;
    if (batch < end) {
      const int idx = batch + lane;
      unsigned mask = 0;
      if (idx < end) {
        float4 v[NV];
#pragma unroll
        for (int q = 0; q < NV; ++q) v[q] = v_next[q];
        mask = strip_mask(v[0].x, v[0].y, v[0].z, v[0].w, v[1].x, v[1].y, sb);
        v[0].w *= 0.5f * FG_NEG_LOG2E;
        v[1].x *= FG_NEG_LOG2E;
        v[1].y *= 0.5f * FG_NEG_LOG2E;
#pragma unroll
        for (int q = 0; q < NV; ++q) st.rec[lane][q] = v[q];
      }
      const bool rel = (mask >> strip) & 1u;
      const uint64_t bal = __ballot(rel);
      if (rel) st.list[0][__popcll(bal & lt_mask)] = (uint16_t)(lane | (mask << 8));
      cnt = __popcll(bal);
      ev_mine += cnt;
      sc_mine += min(64, end - batch);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();  // records and list are private to this wavefront
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      walk(batch, T, acc, last, done, std::true_type{});
    }
    if (idx_next < end) {
      const float4* rec = splats + (size_t)gid_next * (FG_SPLAT_FLOATS / 4);
#pragma unroll
      for (int q = 0; q < NV; ++q) v_next[q] = rec[q];
    }
    {
      const bool stopped = ((done & ~done0) >> lane) & 1ull;
      sh.part[w][lane] = make_float4(stopped ? -1.f : T, acc[0], acc[1], acc[2]);
      sh.part_last[w][lane] = last;
    }
    if (lane < 2) sh.live_or[w][lane] = 0u;
    // 2. + 3.
    float t_in = 1.f, c_in[C] = {0.f, 0.f, 0.f};
    int l_in = start - 1, bw = -1;
    bool fin = false;
    float4 ck = make_float4(1.f, 0.f, 0.f, 0.f);  // the state before this wavefront's batch
    float4 mp = make_float4(1.f, 0.f, 0.f, 0.f);
    __syncthreads();
    for (;;) {
      bw = sh.base_w[lane];
      fin = sh.base_fin[lane] != 0;
      l_in = sh.base_last[lane];
      {
        const float4 b = sh.base[lane];
        t_in = b.x; c_in[0] = b.y; c_in[1] = b.z; c_in[2] = b.w;
      }
      bool earlier = false;  // a batch in front of this one may stop the pixel: its wavefront's business
      // (five batches' loads at a time, in front of their arithmetic: they do not depend on the running state)
      for (int q0 = 0; q0 < w; q0 += 5) {
        float4 pq[5];
        int plq[5];
#pragma unroll
        for (int u = 0; u < 5; ++u) {
          const int qq = min(q0 + u, NWV - 1);
          pq[u] = sh.part[qq][lane];
          plq[u] = sh.part_last[qq][lane];
        }
#pragma unroll
        for (int u = 0; u < 5; ++u) {
          const float4 p = pq[u];
          const int q = q0 + u;
          const bool use = q < w && q > bw && !fin && !earlier;
          const bool ms = use && (p.x < 0.f || t_in * p.x <= MAY_STOP);
          earlier = earlier || ms;
          if (use && !ms) {
            c_in[0] = fmaf(t_in, p.y, c_in[0]);
            c_in[1] = fmaf(t_in, p.z, c_in[1]);
            c_in[2] = fmaf(t_in, p.w, c_in[2]);
            t_in *= p.x;
            l_in = plq[u] >= 0 ? plq[u] : l_in;
          }
        }
      }
      mp = sh.part[w][lane];
      const bool reached = w > bw && !fin && !earlier;
      if (reached) ck = make_float4(t_in, c_in[0], c_in[1], c_in[2]);
      const bool mine = reached && (mp.x < 0.f || t_in * mp.x <= MAY_STOP);
      const uint64_t active = __ballot(mine);
      if (!__syncthreads_or(active != 0ull)) break;
      bool went_on = false;  // a pixel walked on through the batch that might have stopped it: the fold behind it has to be redone
      if (active != 0ull) {
        uint64_t d2 = ~active;
        walk(batch, t_in, c_in, l_in, d2, std::false_type{});
        if (mine) {
          sh.base[lane] = make_float4(t_in, c_in[0], c_in[1], c_in[2]);
          sh.base_last[lane] = l_in;
          sh.base_w[lane] = w;
          sh.base_fin[lane] = (int)((d2 >> lane) & 1ull);
          went_on = !((d2 >> lane) & 1ull);
        }
      }
      if (!__syncthreads_or(went_on)) {
        // every walked pixel stopped: the states folded above stand for the others; the stopped ones' last entries for step 4
        fin = sh.base_fin[lane] != 0;
        if (fin) l_in = sh.base_last[lane];
        if (fin) bw = sh.base_w[lane];
        break;
      }
    }
    // (nobody reads another wavefront's words behind the last vote)
    if (batch > start && batch < end && seg_starts_at(batch - start, seg_fine))
      slots[seg_slot_index(slot0, start, batch, seg_fine) * FG_SEG_SLOT4 + row * TILE + col] = ck;
    // 4. liveness for the backward, exact: entry batch + j is live iff it passed the alpha test of a pixel that had not
    // stopped by then -- the bits of step 1 (taken from T = 1: a superset), cut at the pixel's last entry if it has stopped
    if (live_words && batch < end) {
      const int rel_last = fin ? l_in - batch : 63;  // (fin: l_in = the pixel's last entry, read from the base above)
      const uint32_t klo = rel_last < 0 ? 0u : (rel_last >= 31 ? ~0u : (2u << rel_last) - 1u);
      const uint32_t khi = rel_last < 32 ? 0u : (rel_last >= 63 ? ~0u : (2u << (rel_last - 32)) - 1u);
      if (vlo & klo) atomicOr(&sh.live_or[w][0], vlo & klo);
      if (vhi & khi) atomicOr(&sh.live_or[w][1], vhi & khi);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const uint32_t word = sh.live_or[w][lane >> 5];
      if (batch + lane < end) reinterpret_cast<uint8_t*>(live_words + (batch + lane))[strip] = (uint8_t)((word >> (lane & 31)) & 1u);
    }
    if (w == NWV - 1) {  // the round's end state: the next round's base
      if (!fin) {
        if (w > bw) {  // (no stop in this batch: that was settled above)
          c_in[0] = fmaf(t_in, mp.y, c_in[0]);
          c_in[1] = fmaf(t_in, mp.z, c_in[1]);
          c_in[2] = fmaf(t_in, mp.w, c_in[2]);
          t_in *= mp.x;
          const int pl = sh.part_last[w][lane];
          l_in = pl >= 0 ? pl : l_in;
        }
        sh.base[lane] = make_float4(t_in, c_in[0], c_in[1], c_in[2]);
        sh.base_last[lane] = l_in;
      }
      sh.base_w[lane] = -1;
    }
  }
  __syncthreads();
  if (w != 0) return;
  // the strip's outputs, as raster_fwd_body's epilogue
  const float4 b = sh.base[lane];
  const int last = sh.base_last[lane];
  float accf[C] = {b.y, b.z, b.w};
  {
    const int m = fg::wave_max_i32(last);
    if (lane == 0) reinterpret_cast<int32_t*>(ckpt)[4 * tile + strip] = m;
    // the job's report, in the measure of the serial jobs' (FG_WALK_REPORT): the entries of the walk that reach this strip --
    // the share seen in wavefront 0's batches (one in sixteen of the remainder) applied to the whole walk, prefix included
    const long long reach = (long long)(m - start + 1) * ev_mine / max(sc_mine, 1);
    if (walk_out && reach > FG_WALK_REPORT && lane == 0)
      __hip_atomic_store(walk_out, reach, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (inside) {
    const float alpha_out = 1.f - b.x;
    reinterpret_cast<float*>(ckpt + seg_plane_offset4(n_tiles))[pix] = b.x;  // exact T_final
    if (comp.background || comp.n_clamp > 0) {
      const float om = 1.f - alpha_out;
      unsigned blocked = 0;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        float v = accf[c];
        if (comp.background) v += om * comp.background[c];
        if (c < comp.n_clamp) {
          if (v < 0.f || v > 1.f) blocked |= 1u << c;
          v = fminf(fmaxf(v, 0.f), 1.f);
        }
        accf[c] = v;
      }
      if (comp.n_clamp > 0) comp.clamp_mask[pix] = (uint8_t)blocked;
    }
#pragma unroll
    for (int c = 0; c < C; ++c) render[pix * C + c] = accf[c];
    alphas[pix] = alpha_out;
    last_ids[pix] = last;
  }
}

// open_list: [0] = the strips the prefix jobs of the main launch left open, [8 ...] = tile << 3 | (strip + 1) each; [1] = a
// ticket: the last workgroup to leave zeroes both words (the list build zeroes them too; a second forward over the same
// lists must not find the first one's entries)
// (the 16 stages are ~75 KB of STATIC LDS: beyond the 64 KB of earlier CDNA parts -- this library is gfx950 code)
static_assert(sizeof(WideShared<FG_WIDE_WAVES>) <= 160 * 1024 / 2, "a wide job's stages: two workgroups per CU of gfx950's 160 KB LDS");
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libfgraster's kernels are written for gfx950 (MI355X): 160 KB of LDS per CU, wave64; build with --offload-arch=gfx950"
#endif
__global__ void __launch_bounds__(64 * FG_WIDE_WAVES)
raster_fwd_wide_kernel(int width, int height, int tile_w, int32_t* __restrict__ open_list,
                       const float4* __restrict__ splats, const int32_t* __restrict__ tile_offsets,
                       const int32_t* __restrict__ flatten_ids, float* __restrict__ render, float* __restrict__ alphas,
                       int32_t* __restrict__ last_ids, Composite comp, float4* __restrict__ ckpt,
                       uint32_t* __restrict__ live_words, const int32_t* __restrict__ slot_tab,
                       long long* __restrict__ walk_out, int seg_fine) {
  __shared__ WideShared<FG_WIDE_WAVES> sh;
  const int n_open = __builtin_amdgcn_readfirstlane(open_list[0]);
  for (int v = blockIdx.x; v < n_open; v += gridDim.x) {
    const int e = open_list[8 + v];
    FG_TL_BEGIN();
    raster_fwd_wide_body<FG_WIDE_WAVES>(sh, e >> 3, (e & 7) - 1, width, height, tile_w, splats, tile_offsets, flatten_ids,
                                        render, alphas, last_ids, comp, ckpt, live_words, slot_tab, walk_out, seg_fine);
    __syncthreads();  // (the next job's first words go where wavefront 0 has just read)
    FG_TL_END(1, e >> 3, (e & 7) - 1, 0, 3);
  }
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(&open_list[1], 1) == (int)gridDim.x - 1) {
      open_list[0] = 0;
      open_list[1] = 0;
    }
  }
}

template <int C, int NT>
struct BwdShared {
  float4 rec[NT][rec_vec4(C)];
  // (gid and mask right behind the records: with three channels and one wavefront they start at multiples of 256
  // bytes, which one ds_read2st64_b32 addresses from the entry's 4 * j without an add)
  int32_t gid[NT];
  uint32_t mask[NT];
  // per-wavefront reduction buffer; read with ds_read_b128: keep it 16-byte aligned (unaligned it
  // cost 0.46 -> 0.70 ms)
  alignas(16) float red[NT / 64][(8 + C) * fg::FG_RED_STRIDE];
  int32_t mx[NT / 64];
};

template <int C, int PPT, int NW, bool LIVE = false>
__device__ __forceinline__ void raster_bwd_body(BwdShared<C, 64 * NW>& sh, int tile, int wave_base, int width,
                                                int height, int tile_w, const float4* __restrict__ splats,
                                                const int32_t* __restrict__ tile_offsets,
                                                const int32_t* __restrict__ flatten_ids,
                                                const float* __restrict__ alphas,
                                                const int32_t* __restrict__ last_ids,
                                                const float* __restrict__ v_render,
                                                const float* __restrict__ v_alphas, float* __restrict__ v_splats,
                                                const Composite& comp, const Segments& seg = Segments{nullptr, nullptr, nullptr, 1, 0, 0},
                                                int part = 0, const uint32_t* __restrict__ live_words = nullptr) {
  constexpr int NT = 64 * NW;
  constexpr int RSTEP = TILE / PPT;
  constexpr int NV = rec_vec4(C);
  auto& lds = sh.rec;
  auto& lds_gid = sh.gid;
  auto& lds_mask = sh.mask;
  auto& lds_max = sh.mx;
  auto& lds_red = sh.red;
  const int tile_y = tile / tile_w, tile_x = tile - tile_y * tile_w;
  const int start = tile_offsets[tile], end = tile_offsets[tile + 1];
  if (end <= start) return;
  // one wavefront per workgroup (NW == 1, the mixed launches): the wave index is the constant 0 and
  // every per-wave LDS address folds into an instruction offset instead of a register
  const int lane = fg::lane_id(), wl = NW == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6),
            wave = wave_base + wl;
  const int col = threadIdx.x & 15, row0 = 4 * wave_base + (threadIdx.x >> 4);
  const int ix = tile_x * TILE + col;
  const float px = (float)ix + 0.5f;
  const StripBounds sb = strip_bounds((float)(tile_x * TILE), (float)(tile_y * TILE));
  const unsigned my_strips = wave_strips<PPT>(wave);

  float T[PPT], rest[PPT], vr[PPT][C];
  const float pyb = (float)(tile_y * TILE + (int)(threadIdx.x >> 4) - 4 * wl) + 0.5f;  // see slot_dy
  int last[PPT];
  int my_max = start - 1;
  // A share job (struct Segments) takes its bounds from the forward's per-strip table in front of the
  // checkpoints: the last list index each strip used is known before any pixel state has arrived, so
  // the checkpoint and final-colour loads go out together with the pixel loads instead of one
  // dependent round trip later (job timeline: 14 us of a 67 us share job were prologue).
  FG_TL_MARK(0);  // tile range known
  int lo = start, hi = end, bin_final = start - 1;
  int slot_last[PPT];
  bool from_ckpt = false;  // wave-uniform: pixels whose list continues beyond hi resume from a checkpoint
  const float4* ck_slot = nullptr;
  bool share_job = false;
  if constexpr (C == 3 && NW == 1) share_job = seg.ckpt && seg.parts > 1;
  if (share_job) {
    const int4 tl = reinterpret_cast<const int4*>(seg.ckpt)[tile];
    const int tls[4] = {tl.x, tl.y, tl.z, tl.w};
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      slot_last[k] = __builtin_amdgcn_readfirstlane(tls[wave + k * (4 / PPT)]);
      bin_final = max(bin_final, slot_last[k]);
    }
    const int n_used = bin_final - start + 1;
    if (n_used <= 0) return;
    const int nseg = seg_count(n_used, seg.fine);
    const int c0 = part * nseg / seg.parts, c1 = (part + 1) * nseg / seg.parts;
    if (c0 == c1) return;  // fewer segments than parts: this part is empty
    lo = start + seg_bound(c0, seg.fine);
    hi = bin_final + 1;
    if (c1 < nseg) {
      hi = start + seg_bound(c1, seg.fine);
      from_ckpt = true;
      ck_slot = seg.ckpt + seg_slots_offset4(tile_w * ((height + TILE - 1) / TILE), width, height) +
                seg_slot_index(seg_slot_base(seg.slot_tab, start, tile), start, hi, seg.fine) * FG_SEG_SLOT4;
    }
  }
  FG_TL_MARK(1);  // share bounds known
  // Pixel state in two phases: first EVERY load of every pixel slot, branch-free (lanes outside the image
  // read the nearest pixel inside and drop the value), then the arithmetic.  With the loads inside the
  // per-slot conditionals the compiler could not hoist slot k+1's loads above slot k's branches: four
  // serialised round trips, 11 us of a 68 us share job (job timeline, FG_TL_MARK).
  struct Px {
    float alpha, va;
    int last;
    unsigned blocked;
    float v[C];
  } raw[PPT];
  float4 ckv[PPT];
  float cfv[PPT][3], tfe[PPT];
  const float* t_exact = nullptr;
  if constexpr (C == 3 && NW == 1) {
    if (from_ckpt)
      t_exact = reinterpret_cast<const float*>(seg.ckpt + seg_plane_offset4(tile_w * ((height + TILE - 1) / TILE)));
  }
  {
    const int ixc = min(ix, width - 1);
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      const int iyc = min(tile_y * TILE + row0 + k * RSTEP, height - 1);
      const size_t pix = (size_t)iyc * width + ixc;
      raw[k].alpha = alphas[pix];
      raw[k].last = last_ids[pix];
#pragma unroll
      for (int c = 0; c < C; ++c) raw[k].v[c] = v_render[pix * C + c];
    }
    if (v_alphas) {
#pragma unroll
      for (int k = 0; k < PPT; ++k)
        raw[k].va = v_alphas[(size_t)min(tile_y * TILE + row0 + k * RSTEP, height - 1) * width + ixc];
    } else {
#pragma unroll
      for (int k = 0; k < PPT; ++k) raw[k].va = 0.f;  // v_alphas == nullptr: no gradient on alpha
    }
    if (comp.n_clamp > 0) {
#pragma unroll
      for (int k = 0; k < PPT; ++k)
        raw[k].blocked = comp.clamp_mask[(size_t)min(tile_y * TILE + row0 + k * RSTEP, height - 1) * width + ixc];
    } else {
#pragma unroll
      for (int k = 0; k < PPT; ++k) raw[k].blocked = 0u;
    }
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      ckv[k] = make_float4(1.f, 0.f, 0.f, 0.f);
      cfv[k][0] = cfv[k][1] = cfv[k][2] = 0.f;
      tfe[k] = 1.f;
    }
    if constexpr (C == 3 && NW == 1) {
      // share jobs: the forward's checkpoint at the job's upper end -- T before entry hi, and the colour
      // composited from entry hi on (C_final - C_before_hi) as the suffix sum the alpha gradient needs --
      // loaded for every pixel, used by those whose list continues beyond hi
      if (from_ckpt) {
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
          const int iyc = min(tile_y * TILE + row0 + k * RSTEP, height - 1);
          const size_t pix = (size_t)iyc * width + ixc;
          ckv[k] = ck_slot[(row0 + k * RSTEP) * TILE + col];
          tfe[k] = t_exact[pix];
#pragma unroll
          for (int c = 0; c < 3; ++c) cfv[k][c] = seg.render_raw[pix * 3 + c];
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const int iy = tile_y * TILE + row0 + k * RSTEP;
    const bool inside = ix < width && iy < height;
    const float alpha_f = inside ? raw[k].alpha : 0.f;
    T[k] = 1.f - alpha_f;  // final transmittance
    last[k] = inside ? raw[k].last : start - 1;
    float va = inside ? raw[k].va : 0.f;
    const unsigned blocked = inside ? raw[k].blocked : 0u;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      float v = inside ? raw[k].v[c] : 0.f;
      if ((blocked >> c) & 1u) v = 0.f;                  // clamped channel: no gradient
      if (comp.background) va -= v * comp.background[c];  // d/dalpha of (1 - alpha) * bg
      vr[k][c] = v;
    }
    rest[k] = T[k] * va;  // T_final * dL/dalpha, minus the composited suffix of a resumed pixel (below)
    if constexpr (C == 3 && NW == 1) {
      if (from_ckpt) {
        // (last >= hi implies the pixel is inside the image)
        const float prefix[3] = {ckv[k].y, ckv[k].z, ckv[k].w};
        float sfx = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          // raw final colour: the forward's accumulator, or (composite epilogue) the finished image
          // minus the background term -- exact where the gradient is not blocked by the clamp
          float f = cfv[k][c];
          if (comp.background) f -= (1.f - alpha_f) * comp.background[c];
          sfx += vr[k][c] * (f - prefix[c]);
        }
        const bool resume = last[k] >= hi;
        // rho: the reference's rounded T_final over the exact one (seg_plane_offset4)
        const float rho = T[k] * __builtin_amdgcn_rcpf(fmaxf(tfe[k], 1e-30f));
        T[k] = resume ? ckv[k].x * rho : T[k];
        rest[k] -= resume ? sfx * rho : 0.f;
      }
    }
    my_max = max(my_max, last[k]);
  }
  FG_TL_MARK(2);  // pixel state loaded
  if (!share_job) {
    // last list entry any pixel of the tile used
    bin_final = __builtin_amdgcn_readfirstlane(fg::wave_max_i32(my_max));  // (uniform: batch bounds in scalar registers)
    if (NW > 1) {
      if (lane == 0) lds_max[wl] = bin_final;
      __syncthreads();
#pragma unroll
      for (int w = 0; w < NW; ++w) bin_final = max(bin_final, lds_max[w]);
    }
    if (bin_final - start + 1 <= 0) return;
    hi = bin_final + 1;
    // last list index any pixel of each of the job's pixel slots (= strips) used (scalars)
    if constexpr (LIVE) {
#pragma unroll
      for (int k = 0; k < PPT; ++k) slot_last[k] = __builtin_amdgcn_readfirstlane(fg::wave_max_i32(last[k]));
    }
  }
  const int n_used = bin_final - start + 1;
  (void)n_used;
  const int n_batches = (hi - lo + NT - 1) / NT;
  if (threadIdx.x == 0) { FG_STAT(5, n_used); FG_STAT(6, end - start); }
  // store addresses of the per-entry reduction, one register per four rows, built once per job
  uint32_t red_wr[(8 + C + 3) / 4];
#pragma unroll
  for (int i = 0; i < (8 + C + 3) / 4; ++i) red_wr[i] = fg::lds_opaque(&lds_red[wl][4 * i * fg::FG_RED_STRIDE + lane]);
  float g[16];  // per-splat gradient accumulators of this lane (see the comment at their use)
#pragma unroll
  for (int q = 0; q < 16; ++q) asm volatile("v_mov_b32 %0, 0" : "=v"(g[q]));

  for (int b = n_batches - 1; b >= 0; --b) {
    const int batch = __builtin_amdgcn_readfirstlane(lo + b * NT);  // (uniform; the list index of an entry is then scalar arithmetic)
    __syncthreads();
    FG_TL_STAGE_BEGIN();
    const int tl = NW == 1 ? fresh_lane_id() : (int)threadIdx.x;  // this thread's staging slot
    const int idx = batch + tl;
    unsigned mask = 0;
    // (the id is loaded alongside the liveness word, not behind it: one round trip less per batch)
    const int gid_early = (LIVE && idx < hi) ? flatten_ids[idx] : 0;
    if constexpr (LIVE) {
      // the forward's liveness byte per strip, valid up to the last entry any pixel of that strip
      // used (beyond it the forward never wrote); dead entries' records are not gathered
      if (idx < hi) {
        const uint32_t lw = live_words[idx];
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
          const int s4 = wave + k * (4 / PPT);  // the slot's strip (uniform)
          if (((lw >> (8 * s4)) & 1u) && idx <= slot_last[k]) mask |= 1u << s4;
        }
      }
    }
    if (LIVE ? mask != 0u : idx < hi) {
      const int gid = LIVE ? gid_early : flatten_ids[idx];
      lds_gid[tl] = gid;
      const float4* rec = splats + (size_t)gid * (FG_SPLAT_FLOATS / 4);
      float4 v[NV];
#pragma unroll
      for (int q = 0; q < NV; ++q) v[q] = rec[q];
      if constexpr (!LIVE) mask = strip_mask(v[0].x, v[0].y, v[0].z, v[0].w, v[1].x, v[1].y, sb);
      if constexpr (C == 3) {  // three spare floats in the 48-byte LDS copy: the pre-scaled conic rides along
        v[2].y = v[0].w * (0.5f * FG_NEG_LOG2E);
        v[2].z = v[1].x * FG_NEG_LOG2E;
        v[2].w = v[1].y * (0.5f * FG_NEG_LOG2E);
      }
#pragma unroll
      for (int q = 0; q < NV; ++q) lds[tl][q] = v[q];
    }
    lds_mask[tl] = mask;
    __syncthreads();
    FG_TL_STAGE_END();
    // back to front over the entries that can reach this wavefront's strips
    for (int i = NT / 64 - 1; i >= 0; --i) {
      uint64_t todo = __ballot((lds_mask[64 * i + lane] & my_strips) != 0u);
      while (todo != 0ull) {
        const int bit = 63 - __builtin_clzll(todo);
        todo &= ~(1ull << bit);
        const int j = 64 * i + bit;
        const int idx_j = batch + j;
        if (NW > 1) {
          // wave-uniform skip: no pixel of THIS wavefront reaches the entry (with one wavefront per
          // tile every staged entry is <= bin_final = max(last), i.e. always reached by some pixel)
          bool reach = false;
#pragma unroll
          for (int k = 0; k < PPT; ++k) reach = reach || (idx_j <= last[k]);
          if (!__any(reach)) continue;
        }

        FG_STAT(0, 1);
        Splat s;
        float f[C];
        read_record<C>(lds[j], s, f);
        // (the Gaussian's id for the atomics at the end of the entry is read here, with the record: its LDS
        // round trip is then not on the chain between the reduction and the atomics: 0.344 -> 0.340 ms.
        // Deferring the read-back half of the reduction behind the next entry's record read as well: 80
        // VGPRs, 0.347 -- dropped.)
        const int gid_v = lds_gid[j];
        const float dx = s.x - px, dy_base = s.y - pyb;
        SigmaTerms st;
        if constexpr (C == 3) {
          const float4 tail = lds[j][2];  // (f2, a', b', c'): the load read_record already issued
          st = sigma_terms_prescaled(tail.y, tail.z, tail.w, dx);
        } else {
          st = sigma_terms(s, dx);
        }
        // Per pixel slot: one wave-uniform branch ("does any lane contribute?"), then straight-line
        // select-predicated arithmetic.  Nested divergent ifs made the compiler re-materialise
        // the 11 accumulators at every merge point (~30 v_mov per slot in the ISA).
        // The 16 accumulators are zeroed by 16 separate asm moves: a plain `g[q] = 0` loop becomes a
        // memset, SROA then promotes g to ONE <16 x float> value (a 512-bit register tuple) and
        // every conditional update turns into whole-tuple copies (8 v_mov_b64 per slot per path).
        // They are zeroed once per tile and again after each reduction that consumed them: an
        // entry to which no lane contributes leaves them untouched (all updates sit behind the
        // uniform any-valid branches).
        // one live pixel slot: `valid` lanes contribute (select-predicated straight-line code)
#define FG_BWD_SLOT_UPDATE(K, VALID, VIS, ALPHA)                                                        \
  do {                                                                                                  \
    FG_STAT(2, 1);                                                                                      \
    FG_STAT(3, __popcll(VALID));                                                                        \
    contributed = true;                                                                                 \
    const float dy = slot_dy<PPT>(dy_base, wave, K);                                                    \
    const float ov = s.o * (VIS);                                                                       \
    const float a_eff = lane_select0(VALID, ALPHA);                                                     \
    const float ra = __builtin_amdgcn_rcpf(1.f - a_eff); /* 1 ulp; 1 - alpha >= 1e-3; 1 if masked */    \
    T[K] *= ra;                                                                                         \
    const float fac = a_eff * T[K];                                                                     \
    float cdot = 0.f;                                                                                   \
    _Pragma("unroll") for (int c = 0; c < C; ++c) {                                                     \
      g[8 + c] += fac * vr[K][c];                                                                       \
      cdot += f[c] * vr[K][c];                                                                          \
    }                                                                                                   \
    /* rest[K] = T_final dL/dalpha - (what the entries behind this one composited, dotted with dL/dC) */ \
    const float v_alpha = fmaf(cdot, T[K], rest[K] * ra);                                               \
    rest[K] = fmaf(-cdot, fac, rest[K]);                                                                \
    const uint64_t open = (VALID) & lanes_ole(ov, FG_ALPHA_MAX); /* alpha not clamped: gradient flows */ \
    const float v_o = lane_select0(open, (VIS) * v_alpha);       /* d/d opacity */                      \
    const float v_sigma = -s.o * v_o;                                                                   \
    g[2] += v_o;                                                                                        \
    /* dx is the same for all pixel slots of a lane: the conic and the mean gradients need only the     \
       moments S0 = sum v_sigma (= -o g[2]), S1 = sum v_sigma dy, S2 = sum v_sigma dy^2 per entry       \
       (g[4], g[5], zero at the start of every contributing entry) -- finished below */                 \
    const float vsdy = v_sigma * dy;                                                                    \
    g[4] += vsdy;                                                                                       \
    g[5] = fmaf(vsdy, dy, g[5]);                                                                        \
    /* absgrad sums |.| per pixel: not a moment.  |v_sigma| |h| + g as ONE fma with abs source modifiers (as     \
       g += |v_sigma h| the compiler issued a multiply and an add: 2 of the slot's 37 vector instructions) */    \
    g[6] = fmaf(fabsf(v_sigma), fabsf(fmaf(s.b, dy, adx)), g[6]);                                       \
    g[7] = fmaf(fabsf(v_sigma), fabsf(fmaf(s.c, dy, bdx)), g[7]);                                       \
  } while (0)
        bool contributed = false;
        const float adx = s.a * dx, bdx = s.b * dx;
        if constexpr (LIVE) {
          // The forward recorded which (entry, strip) pairs had a contributing pixel (live_words): only
          // those are evaluated -- no exponential, no compares for the others.
          const unsigned smask = __builtin_amdgcn_readfirstlane(lds_mask[j]);
#pragma unroll
          for (int k = 0; k < PPT; ++k) {
            if (!((smask >> (wave + k * (4 / PPT))) & 1u)) continue;  // wave-uniform
            const float e2 = neg_sigma_log2e(st, slot_dy<PPT>(dy_base, wave, k));
            const float vis = __builtin_amdgcn_exp2f(e2);
            const float alpha = fminf(FG_ALPHA_MAX, s.o * vis);
            const uint64_t valid = lanes_sle(idx_j, last[k]) & lanes_ule(e2, 0.f) & lanes_uge(alpha, FG_ALPHA_SKIP);
            FG_STAT(1, 1);
            if (valid == 0ull) continue;  // (a one-ulp disagreement with the forward's test)
            FG_BWD_SLOT_UPDATE(k, valid, vis, alpha);
          }
        } else {
          // Pre-test of all pixel slots first, as PPT independent instruction chains with no branch
          // between them: one wavefront issues a DEPENDENT vector instruction only every ~8 clocks
          // (scripts/micro/valu_rate.hip) and a tile has few wavefronts, so the serial
          // sub-fma-fma-exp-mul-min-cmp-branch chain per slot was latency-bound.  Slots the strip
          // mask rules out are evaluated too (their masks come out empty: the culling is
          // result-preserving), which costs less than the branches did.
          float vis_k[PPT], alpha_k[PPT];
          uint64_t valid_k[PPT];
#pragma unroll
          for (int k = 0; k < PPT; ++k) {
            const float e2 = neg_sigma_log2e(st, slot_dy<PPT>(dy_base, wave, k));
            vis_k[k] = __builtin_amdgcn_exp2f(e2);
            alpha_k[k] = fminf(FG_ALPHA_MAX, s.o * vis_k[k]);
            valid_k[k] = lanes_sle(idx_j, last[k]) & lanes_ule(e2, 0.f) & lanes_uge(alpha_k[k], FG_ALPHA_SKIP);
            FG_STAT(1, 1);
          }
#pragma unroll
          for (int k = 0; k < PPT; ++k) {
            const uint64_t valid = valid_k[k];
            if (valid == 0ull) continue;  // scalar
            FG_BWD_SLOT_UPDATE(k, valid, vis_k[k], alpha_k[k]);
          }
        }
#undef FG_BWD_SLOT_UPDATE
        if (!contributed) continue;  // wave-uniform: only ever set under the uniform any-valid branches
        // v_mean2d = (a dx S0 + b S1, b dx S0 + c S1), v_conic = (1/2 dx^2 S0, dx S1, 1/2 S2)
        {
          const float s0 = -s.o * g[2], s1 = g[4];
          g[0] = fmaf(s.b, s1, adx * s0);
          g[1] = fmaf(s.c, s1, bdx * s0);
          g[3] = (0.5f * dx * dx) * s0;
          g[4] = dx * s1;
          g[5] *= 0.5f;
        }
        FG_STAT(4, 1);
#if FG_BWD_LDS_REDUCE
        (void)lds_red;
        if (true) {  // 8 + C live accumulators: summed through LDS (the butterflies' swaps and DPP
                     // operations cost 4-8 issue clocks each, fg_common.h)
          const float total = fg::wave_reduce_rows_lds<8 + C>(g, red_wr, lds_red[wl], lane);
          if ((lane & 3) == 0 && (lane >> 2) < 8 + C) {
            const int gid_s = __builtin_amdgcn_readfirstlane(gid_v);
            float* dst = v_splats + (size_t)gid_s * FG_SPLAT_FLOATS + (lane >> 2);
            __hip_atomic_fetch_add(dst, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        } else
#endif
        if (C <= 4) {  // only record slots 0..11 are in use: the cheaper 12-value butterfly
          const float total = fg::wave_reduce12_transposed(g);
          if (fg::wave_reduce12_owner(lane)) {
            // the Gaussian id is wave-uniform: scalar base address + per-lane slot offset
            const int gid_s = __builtin_amdgcn_readfirstlane(gid_v);
            float* dst = v_splats + (size_t)gid_s * FG_SPLAT_FLOATS + fg::wave_reduce12_index(lane);
            __hip_atomic_fetch_add(dst, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        } else {
          const float total = fg::wave_reduce16_transposed(g);
          if ((lane & 3) == 0) {
            const int gid_s = __builtin_amdgcn_readfirstlane(gid_v);
            float* dst = v_splats + (size_t)gid_s * FG_SPLAT_FLOATS + (lane >> 2);
            __hip_atomic_fetch_add(dst, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        // only the accumulators the slot updates add to need clearing (g[2], g[4 .. 8 + C); g[0], g[1], g[3]
        // are assigned per entry); two at a time with v_mov_b64 where they pair up
        asm volatile("v_mov_b32 %0, 0" : "=v"(g[2]));
#pragma unroll
        for (int q = 4; q + 1 < 8 + C; q += 2) {
          struct F2 { float a, b; };
          double z;
          asm volatile("v_mov_b64 %0, 0" : "=v"(z));
          const F2 h = __builtin_bit_cast(F2, z);
          g[q] = h.a;
          g[q + 1] = h.b;
        }
        if ((8 + C) & 1) asm volatile("v_mov_b32 %0, 0" : "=v"(g[8 + C - 1]));
      }
    }
  }
}

template <int C, int PPT>
__global__ void __launch_bounds__(256 / PPT)
raster_bwd_kernel(int width, int height, int tile_w, int tile_h, int order_mode, const float4* __restrict__ splats,
                  const int32_t* __restrict__ tile_offsets, const int32_t* __restrict__ flatten_ids,
                  const float* __restrict__ alphas, const int32_t* __restrict__ last_ids,
                  const float* __restrict__ v_render, const float* __restrict__ v_alphas,
                  float* __restrict__ v_splats, Composite comp) {
  constexpr int NW = 256 / PPT / 64;
  __shared__ BwdShared<C, 64 * NW> sh;
  const int tile = tile_of_block(blockIdx.x, tile_w * tile_h, tile_w, tile_h, order_mode);
  if (tile < 0) return;
  raster_bwd_body<C, PPT, NW>(sh, tile, 0, width, height, tile_w, splats, tile_offsets, flatten_ids, alphas, last_ids,
                              v_render, v_alphas, v_splats, comp);
}

template <int C, bool LIVE>
__global__ void __launch_bounds__(64)
raster_bwd_mixed_kernel(int width, int height, int tile_w, int tile_h, int nx, int tail_tiles,
                        const int32_t* __restrict__ jobs, int cap,
                        const float4* __restrict__ splats, const int32_t* __restrict__ tile_offsets,
                        const int32_t* __restrict__ flatten_ids, const float* __restrict__ alphas,
                        const int32_t* __restrict__ last_ids, const float* __restrict__ v_render,
                        const float* __restrict__ v_alphas, float* __restrict__ v_splats, Composite comp,
                        Segments seg, const uint32_t* __restrict__ live_words) {
  __shared__ BwdShared<C, 64> sh;
  int strip = -1, part = 0, tile;
  if (seg.parts > 1) {
    // list-share jobs (struct Segments): whole tiles, some of them cut into several jobs over shares
    // of their list -- from the job list (by position and by content, build_jobs_kernel) or, without
    // one, the last seg.tail tiles of every XCD's sequence
    if (jobs) {
      tile = share_from_list(blockIdx.x, jobs, cap, part, seg.parts);
    } else {
      // consecutive workgroups of the XCD are the shares of one tile (they share its records in the L2);
      // part numbers rotated by the tile's position: workgroups go to a CU's SIMDs round-robin, tiles
      // with fewer segments than parts leave the same part numbers empty, and unrotated those all
      // landed on the same SIMDs (4 parts: 0.77 ms against 0.43 for 3)
      const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
      const Band band = band_of_xcd(xcd, tile_w, tile_h, nx);
      const int n = band.nrows * band.ncols;
      const int tail = seg.tail > 0 ? min(seg.tail, n) : n, n_main = n - tail;
      int t = k;
      if (k >= n_main) {
        t = n_main + (k - n_main) / seg.parts;
        part = ((k - n_main) - (t - n_main) * seg.parts + t) % seg.parts;
      } else {
        seg.parts = 1;  // (a by-value copy: this workgroup walks its tile's whole list)
      }
      tile = job_of_block((t << 3) | xcd, tile_w, tile_h, nx, 0, strip);
    }
  } else {
    tile = jobs ? job_from_list(blockIdx.x, jobs, cap, strip)
                : job_of_block(blockIdx.x, tile_w, tile_h, nx, tail_tiles, strip);
  }
  if (tile < 0) return;
  FG_TL_BEGIN();
  if (seg.prio > 0) {
    const int len = tile_offsets[tile + 1] - tile_offsets[tile];
    job_priority(tile_offsets, tile_w * tile_h, seg.parts > 1 ? len / seg.parts : len, strip < 0 ? 13 : (strip >= 4 ? 10 : 8), seg.prio);
  }
  if (strip < 0)
    raster_bwd_body<C, 4, 1, LIVE>(sh, tile, 0, width, height, tile_w, splats, tile_offsets, flatten_ids, alphas,
                                   last_ids, v_render, v_alphas, v_splats, comp, seg, part, live_words);
  else if (strip >= 4)
    raster_bwd_body<C, 2, 1, LIVE>(sh, tile, strip - 4, width, height, tile_w, splats, tile_offsets, flatten_ids,
                                   alphas, last_ids, v_render, v_alphas, v_splats, comp, seg, part, live_words);
  else
    raster_bwd_body<C, 1, 1, LIVE>(sh, tile, strip, width, height, tile_w, splats, tile_offsets, flatten_ids, alphas,
                                   last_ids, v_render, v_alphas, v_splats, comp, seg, part, live_words);
  FG_TL_END(2, tile, strip, part, seg.parts);
}

__global__ void __launch_bounds__(256)
pack_splats_kernel(int N, int C, const float* __restrict__ means2d, const float* __restrict__ conics,
                   const float* __restrict__ opacities, const float* __restrict__ features,
                   float* __restrict__ splats) {
  // 16 lanes per Gaussian: lane l of the group writes float l of the record (coalesced 64 B)
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t i = t >> 4;
  const int l = (int)(t & 15);
  if (i >= N) return;
  float v = 0.f;
  if (l < 2) v = means2d[2 * i + l];
  else if (l == 2) v = opacities[i];
  else if (l < 6) v = conics[3 * i + (l - 3)];
  else if (l - 6 < C) v = features[(size_t)i * C + (l - 6)];
  splats[i * FG_SPLAT_FLOATS + l] = v;
}

__global__ void __launch_bounds__(256)
unpack_grads_kernel(int N, int C, const float* __restrict__ v_splats, float* __restrict__ v_means2d,
                    float* __restrict__ v_means2d_abs, float* __restrict__ v_conics,
                    float* __restrict__ v_opacities, float* __restrict__ v_features) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t i = t >> 4;
  const int l = (int)(t & 15);
  if (i >= N) return;
  const float v = v_splats[i * FG_SPLAT_FLOATS + l];
  if (l < 2) {
    if (v_means2d) v_means2d[2 * i + l] = v;
  } else if (l == 2) {
    if (v_opacities) v_opacities[i] = v;
  } else if (l < 6) {
    if (v_conics) v_conics[3 * i + (l - 3)] = v;
  } else if (l < 8) {
    if (v_means2d_abs) v_means2d_abs[2 * i + (l - 6)] = v;
  } else if (l - 8 < C) {
    if (v_features) v_features[(size_t)i * C + (l - 8)] = v;
  }
}

// Pixels per lane.  Measured on MI355X (1M Gaussians, 1080p, profiles/r01_ppt_sweep.md): with
// strip culling the forward is fastest with 2 pixels per lane (2 wavefronts per tile) and the
// backward with 4 (one wavefront per tile: the 16-value wave reduction + atomic is paid once
// per 256 pixels and no cross-wave step exists).  fg_raster_config::ppt_fwd / ppt_bwd override.
//
// LAUNCH POLICY.  Everything below that used to be read from FG_RASTER_* environment variables comes in
// through the caller's fg_raster_config (include/fgraster.h; NULL = the measured defaults): the library
// reads no environment and keeps no state.  `Cfg` is the caller's struct with every "-1 = default" left
// as it is; the functions below resolve a field where they use it.
using Cfg = fg_raster_config;
Cfg resolve(const fg_raster_config* c) {
  Cfg r;
  fg_raster_config_init(&r);
  if (c) {
    // a caller compiled against an older, shorter struct: take the fields it has
    const size_t n = (size_t)c->size < sizeof(Cfg) ? (size_t)c->size : sizeof(Cfg);
    if (n >= sizeof(int32_t)) memcpy(&r, c, n);
    r.size = (int32_t)sizeof(Cfg);
  }
  return r;
}
int forced_ppt(int v) { return (v == 1 || v == 2 || v == 4) ? v : 0; }
// tile order of the classic launches (profiles/r01_tile_order.md: column-major bands, 2, measured best)
// | the two measurement hooks (debug_only_xcd: only that XCD's workgroups work; debug_k_mod: every m-th tile)
int tile_order_mode(const Cfg& c) {
  const int order = (c.tile_order >= 0 && c.tile_order <= 5) ? c.tile_order : 2;
  const int only = (c.debug_only_xcd >= 0 && c.debug_only_xcd <= 7) ? ((c.debug_only_xcd + 1) << 8) : 0;
  return order | only | ((c.debug_k_mod > 0 ? c.debug_k_mod : 0) << 12);
}
int launch_grid(int mode, int tile_w, int tile_h) {
  switch (mode) {
    case 1: return tile_w * tile_h;
    case 3: return 8 * 2 * ((tile_h + 15) / 16) * tile_w;
    case 4: return 8 * ((tile_w + 3) / 4) * ((tile_h + 1) / 2);
    case 5: return 8 * ((tile_w + 1) / 2) * ((tile_h + 3) / 4);
    default: return 8 * ((tile_h + 7) / 8) * tile_w;
  }
}
// Pixels per lane by tile count (fg_raster_config::ppt_fwd / ppt_bwd override).  A tile's list is walked
// serially by each of its wavefronts, so with few tiles the launch is bound by the longest list,
// not by throughput: more, smaller wavefronts per tile then win although they repeat the
// per-entry work.  Measured (profiles/r01_ppt_by_tiles.md): 8160 tiles -> fwd 2 / bwd 4;
// 2040 tiles (960x540) -> 1 / 1: fwd 0.178 -> 0.117 ms, bwd 0.420 -> 0.317 ms;
// 510 tiles (480x270) -> 1 / 1: fwd 0.138 -> 0.071, bwd 0.357 -> 0.151.
int raster_ppt_fwd(const Cfg& c, int n_tiles) {
  const int forced = forced_ppt(c.ppt_fwd);
  if (forced) return forced;
  return n_tiles >= 6000 ? 2 : 1;
}
int raster_ppt_bwd(const Cfg& c, int n_tiles) {
  const int forced = forced_ppt(c.ppt_bwd);
  if (forced) return forced;
  return n_tiles >= 6000 ? 4 : (n_tiles >= 3000 ? 2 : 1);
}

template <int C, int PPT>
int launch_fwd(const Cfg& cfg, int width, int height, const float* splats, const int32_t* tile_offsets,
               const int32_t* flatten_ids, float* render, float* alphas, int32_t* last_ids, Composite comp,
               hipStream_t s) {
  const int tile_w = (width + TILE - 1) / TILE, tile_h = (height + TILE - 1) / TILE;
  const int mode = tile_order_mode(cfg);
  const int grid = launch_grid(mode & 255, tile_w, tile_h);
  hipLaunchKernelGGL((raster_fwd_kernel<C, PPT>), dim3(grid), dim3(256 / PPT), 0, s, width, height, tile_w,
                     tile_h, mode, reinterpret_cast<const float4*>(splats), tile_offsets, flatten_ids, render,
                     alphas, last_ids, comp);
  return hipGetLastError() == hipSuccess ? FG_OK : FG_ERR_LAUNCH;
}

template <int C, int PPT>
int launch_bwd(const Cfg& cfg, int width, int height, const float* splats, const int32_t* tile_offsets,
               const int32_t* flatten_ids, const float* alphas, const int32_t* last_ids, const float* v_render,
               const float* v_alphas, float* v_splats, Composite comp, hipStream_t s) {
  const int tile_w = (width + TILE - 1) / TILE, tile_h = (height + TILE - 1) / TILE;
  const int mode = tile_order_mode(cfg);
  const int grid = launch_grid(mode & 255, tile_w, tile_h);
  hipLaunchKernelGGL((raster_bwd_kernel<C, PPT>), dim3(grid), dim3(256 / PPT), 0, s, width, height, tile_w,
                     tile_h, mode, reinterpret_cast<const float4*>(splats), tile_offsets, flatten_ids, alphas,
                     last_ids, v_render, v_alphas, v_splats, comp);
  return hipGetLastError() == hipSuccess ? FG_OK : FG_ERR_LAUNCH;
}

// Tiles per XCD at the end of its sequence that the mixed launch splits into four single-strip
// jobs (TAIL4) / into two two-strip jobs placed before those (TAIL2).  Absolute counts: the tail
// of a launch is about one job per wavefront slot of the XCD (128 SIMDs x 4-8 slots), whatever the
// image size -- 510 / 300 were the best settings at 1080p, 1440p and 2160p alike
// (profiles/r01_tail_split.md); at most half / 30% of the XCD's tiles.  0 = classic launch.
// fg_raster_config::tail4_fwd / tail2_fwd / tail4_bwd / tail2_bwd override (-1 = these defaults).
// content thresholds in 1/65536 of the total list length (16 = total / 2^12): measured on the
// uniform and on the clustered scene (profiles/r01_tail_split.md); at 8160 tiles 16 / 20 are 2.0x /
// 2.5x the mean list length
#ifndef FG_SPLIT4_FWD
#define FG_SPLIT4_FWD 20
#endif
#ifndef FG_SPLIT2_FWD
#define FG_SPLIT2_FWD 16
#endif
#ifndef FG_SPLIT4_BWD
#define FG_SPLIT4_BWD 20
#endif
#ifndef FG_SPLIT2_BWD
#define FG_SPLIT2_BWD 16
#endif
#ifndef FG_TAIL4_TILES_FWD
#define FG_TAIL4_TILES_FWD 510
#endif
#ifndef FG_TAIL2_TILES_FWD
#define FG_TAIL2_TILES_FWD 0
#endif
#ifndef FG_TAIL4_TILES_BWD
#define FG_TAIL4_TILES_BWD 0
#endif
#ifndef FG_TAIL2_TILES_BWD
#define FG_TAIL2_TILES_BWD 300
#endif
// mixed launches (job lists, liveness, list shares) from this many tiles up; below: classic launches.
// (900 until late in round 3 -- set before the backward had list shares.  With them the reference's quarter-resolution
// phase, 480x270 = 510 tiles / 100k Gaussians, runs its backward 0.140 -> 0.067 ms and the step's kernels 0.293 -> 0.222 ms
// (forward 0.066 -> 0.072: four strip jobs per tile each stage the list); 320x192 = 240 tiles: 0.261 -> 0.231.)
#ifndef FG_MIXED_MIN_TILES
#define FG_MIXED_MIN_TILES 200
#endif
// Returns tail4 | tail2 << 16 (both clamped to 16 bits); req4 >= 0: the caller's numbers (req2 < 0: 0).
int raster_tail(const Cfg& cfg, int req4, int req2, int n_tiles, int dflt4, int dflt2) {
  int t4, t2 = 0;
  if (req4 >= 0) {
    t4 = req4;
    if (req2 > 0) t2 = req2;
  } else {
    if (n_tiles < FG_MIXED_MIN_TILES || (tile_order_mode(cfg) & 255) != 2 || (tile_order_mode(cfg) >> 8) != 0) return 0;
    const int per_xcd = n_tiles / 8;
    if (n_tiles < 5000) {
      // 900..5000 tiles (640x360 ... 1280x720; 960x540 is the reference's half-resolution phase): fewer tiles than
      // wavefront slots -- every tile of the forward as four single-strip jobs, every tile of the
      // backward in list shares (seg_tail), liveness and checkpoints as at full size.  960x540 / 300k
      // Gaussians: step 0.589 -> 0.536 ms (backward 0.274 -> 0.182, forward 0.100 -> 0.109);
      // 1280x720 / 500k: 0.725 -> 0.650.
      t4 = dflt4 > 0 ? 0xFFFF : 0;
      t2 = dflt4 > 0 ? 0 : (dflt2 < per_xcd * 3 / 10 ? dflt2 : per_xcd * 3 / 10);
    } else {
      t4 = dflt4 < per_xcd / 2 ? dflt4 : per_xcd / 2;
      t2 = dflt2 < per_xcd * 3 / 10 ? dflt2 : per_xcd * 3 / 10;
    }
  }
  t4 = t4 < 0 ? 0 : (t4 > 0xFFFF ? 0xFFFF : t4);
  t2 = t2 < 0 ? 0 : (t2 > 0x7FFF ? 0x7FFF : t2);
  return t4 | (t2 << 16);
}
// tail4 | tail2 << 16 of the mixed launch, or 0 = classic launch for this image size / environment
int mixed_tail_fwd(const Cfg& c, int n_tiles) {
  if (forced_ppt(c.ppt_fwd)) return 0;  // forced pixels per lane: the classic launch
  return raster_tail(c, c.tail4_fwd, c.tail2_fwd, n_tiles, FG_TAIL4_TILES_FWD, FG_TAIL2_TILES_FWD);
}
int mixed_tail_bwd(const Cfg& c, int n_tiles) {
  if (forced_ppt(c.ppt_bwd)) return 0;
  return raster_tail(c, c.tail4_bwd, c.tail2_bwd, n_tiles, FG_TAIL4_TILES_BWD, FG_TAIL2_TILES_BWD);
}
// tiles of the largest XCD rectangle for an nx x (8 / nx) arrangement
int band_tiles_for(int tile_w, int tile_h, int nx) {
  int m = 0;
  for (int x = 0; x < 8; ++x) {
    const Band b = band_of_xcd(x, tile_w, tile_h, nx);
    m = b.ncols * b.nrows > m ? b.ncols * b.nrows : m;
  }
  return m;
}
// Whole-row bands (nx = 1) unless fg_raster_config::bands_nx = 2|4|8 asks for another arrangement.  Measured at
// 1080p (profiles/r02_job_timeline.md): 8 x 1 column strips give every XCD exactly 1020 tiles instead of
// 1080 / 960, and all eight then finish when the 1080-tile XCDs did before (an XCD's time is set by its
// long jobs and the drain after them, not by its tile count); the forward is 5% slower (0.218 against
// 0.208 ms), 4 x 2 rectangles 13%.
int band_nx(const Cfg& c) {
  const int v = c.bands_nx;
  return (v == 1 || v == 2 || v == 4 || v == 8) ? v : 1;
}
// Tile rows an XCD's band may hold when the job builder balances the bands by content (fgjobs::balanced_row_bands): one
// and a half times the equal share.  Everything below sizes grids and list segments for the LARGEST band, so this is
// what lets a light XCD take rows off a heavy one; the workgroups it adds to a launch return at once (~1 ns each).
int band_rows_limit(const Cfg& c, int tile_h) {
  const int equal = (tile_h + 7) / 8;
  if (band_nx(c) != 1 || c.balance_bands == 0 || c.balance_bands == 2) return 0;  // (equal rows / equal spans / interleaved blocks: no band beyond its equal share)
  const int lim = equal + (equal + 1) / 2;
  return lim < tile_h ? lim : tile_h;
}
int band_tiles_max(const Cfg& c, int tile_w, int tile_h) {
  const int rows = band_rows_limit(c, tile_h);
  return rows > 0 ? rows * tile_w : band_tiles_for(tile_w, tile_h, band_nx(c));
}
int mixed_grid(const Cfg& c, int tile_w, int tile_h, int tail) {  // positional jobs only
  const int n_max = band_tiles_max(c, tile_w, tile_h);
  int t4 = tail & 0xFFFF, t2 = tail >> 16;
  t4 = t4 < n_max ? t4 : n_max;
  t2 = t2 < n_max - t4 ? t2 : n_max - t4;
  return 8 * (n_max + 3 * t4 + t2);
}
int jobs_cap(const Cfg& c, int tile_w, int tile_h) { return 8 * band_tiles_max(c, tile_w, tile_h); }
// grid of a launch that reads job lists: the positional job count + half a job per tile for the
// content splits (the builder fits the list into it)
int listed_grid(const Cfg& c, int tile_w, int tile_h, int tail) {
  const int g = mixed_grid(c, tile_w, tile_h, tail) / 8;
  const int cap = jobs_cap(c, tile_w, tile_h);
  const int want = g + band_tiles_max(c, tile_w, tile_h) / 2;
  return 8 * (want < cap ? want : cap);
}
// Content thresholds of the mixed launch: "a4,a2" = split a tile in four when its list is longer than
// total * a4 / 65536, in two when longer than total * a2 / 65536 (fg_raster_config::split4_* / split2_*:
// -1 = the defaults, 0 = off; a given split4 with split2 < 0 means split2 = 0).
int raster_split(int req4, int req2, int dflt4, int dflt2) {
  int s4 = dflt4, s2 = dflt2;
  if (req4 >= 0) {
    s4 = req4;
    s2 = req2 > 0 ? req2 : 0;
  }
  s4 = s4 < 0 ? 0 : (s4 > 0x7FFF ? 0x7FFF : s4);
  s2 = s2 < 0 ? 0 : (s2 > 0x7FFF ? 0x7FFF : s2);
  return s4 | (s2 << 16);
}

// workgroups of a launch of list-share jobs: the positional count (+ half a job per tile for content
// splits when a list is read: the builder fits the list into it)
int seg_parts2(const Cfg& c);
int seg_tail2(const Cfg& c);
int seg_grid(const Cfg& c, int tile_w, int tile_h, int parts, int tail, bool listed) {
  const int n_max = band_tiles_max(c, tile_w, tile_h);
  const int split = tail > 0 ? (tail < n_max ? tail : n_max) : n_max;
  int per_xcd = n_max - split + split * parts + (listed ? n_max / 2 : 0);
  if (listed && seg_parts2(c) > parts) per_xcd += (seg_tail2(c) < split ? seg_tail2(c) : split) * (seg_parts2(c) - parts);
  const int cap = jobs_cap(c, tile_w, tile_h);
  return 8 * (listed && per_xcd > cap ? cap : per_xcd);
}
// heavy_tiles: list length beyond which a tile of the forward is a heavy tile (raster_fwd_body MODE 1 / 2); <= 0: off
// heavy_wide: 1 (default, -1) = what a heavy tile's prefix jobs (FG_WIDE_PREFIX entries, in the main launch) leave open goes to
// WIDE jobs (raster_fwd_wide_kernel, one launch behind the main one); 0 = round 4's form (prefix of FG_HEAVY_PREFIX entries,
// local jobs, combine jobs: A/B)
bool heavy_wide(const Cfg& c) { return c.heavy_wide != 0; }
int heavy_len(const Cfg& c) {
  const int least = heavy_wide(c) ? FG_WIDE_PREFIX + 256 : FG_HEAVY_PREFIX + 512;
  return c.heavy_tiles > 0 ? (c.heavy_tiles < least ? least : c.heavy_tiles) : 0;
}
// seg_fine: the checkpoint grid (jobs_build.h seg_index) -- a checkpoint per 64 entries for a list's first seg_fine entries,
// per 128 behind; -1 = the default, 0 = per 64 throughout (rounds 2-4); the three-launch heavy tiles keep a slot per batch
#ifndef FG_SEG_FINE_DEFAULT
#define FG_SEG_FINE_DEFAULT 640
#endif
int seg_fine(const Cfg& c) {
  if (c.seg_fine == 0 || (heavy_len(c) > 0 && !heavy_wide(c))) return FG_SEG_FINE_NEVER;
  const int v = c.seg_fine < 0 ? FG_SEG_FINE_DEFAULT : c.seg_fine;
  return (v + 63) / 64 * 64;
}
// seg_slots: checkpoint slots of the buffer the raster calls are given, eight equal shares of them an XCD band's (compact
// slots: jobs_build.h JobBuild::slot_budget); 0 = one slot per 64 list entries of every tile, by formula
int seg_slots(const Cfg& c) { return c.seg_slots > 0 ? (c.seg_slots + 7) / 8 * 8 : 0; }
int jobs_cap(const Cfg& c, int tile_w, int tile_h);
// words of a list in front of its table of first slots (the main list, + the heavy tiles' two lists)
int64_t slot_table_offset(const Cfg& c, int tile_w, int tile_h) {
  return 8 + 8 * (int64_t)jobs_cap(c, tile_w, tile_h) + (heavy_len(c) > 0 ? FG_LOCAL_WORDS + FG_HEAVY_WORDS : 0);
}
const int32_t* slot_table(const Cfg& c, const int32_t* jobs, int tile_w, int tile_h) {
  return jobs && seg_slots(c) > 0 ? jobs + slot_table_offset(c, tile_w, tile_h) : nullptr;
}
// prio_fwd / prio_bwd: issue priority thresholds of the mixed launches' jobs (job_priority), lo | hi << 16 in percent of
// the mean single-strip job; -1 = the measured default, 0 = off
#ifndef FG_PRIO_FWD_DEFAULT
#define FG_PRIO_FWD_DEFAULT (250 | 350 << 16)
#endif
#ifndef FG_PRIO_BWD_DEFAULT
#define FG_PRIO_BWD_DEFAULT (120 | 160 << 16)
#endif
int job_prio(int v, int dflt) { return v < 0 ? dflt : v; }
// use_liveness = 0: the backward ignores the forward's liveness bytes (A/B)
const uint32_t* live_use(const Cfg& c, const uint32_t* live_words) { return c.use_liveness == 0 ? nullptr : live_words; }
// seg_tail = tiles per XCD, at the end of its sequence, whose lists are split (0 = every tile)
int seg_tail(const Cfg& c, int n_tiles) {
  return c.seg_tail >= 0 ? c.seg_tail : (n_tiles < 5000 ? 0 : FG_SEG_TAIL_DEFAULT);  // below 5000 tiles: every tile
}
// ... clamped so that the positional jobs of the largest XCD band (+ the margin for content splits)
// fit a job list segment
int seg_tail_fit(const Cfg& c, int tile_w, int tile_h, int parts, int tail) {
  const int n_max = band_tiles_max(c, tile_w, tile_h);
  if (tail <= 0 || tail > n_max) tail = n_max;
  const int room = jobs_cap(c, tile_w, tile_h) - n_max - n_max / 2;
  const int fit = parts > 1 ? room / (parts - 1) : n_max;
  return tail < fit ? tail : (fit > 1 ? fit : 1);
}
// seg_parts2 / seg_tail2: the last tail2 tiles of every XCD's sequence get parts2 jobs
#ifndef FG_SEG_PARTS2_DEFAULT
#define FG_SEG_PARTS2_DEFAULT 0
#endif
#ifndef FG_SEG_TAIL2_DEFAULT
#define FG_SEG_TAIL2_DEFAULT 0
#endif
int seg_parts2(const Cfg& c) {
  const int v = c.seg_parts2 >= 0 ? c.seg_parts2 : FG_SEG_PARTS2_DEFAULT;
  return v > 16 ? 16 : v;
}
int seg_tail2(const Cfg& c) { return c.seg_tail2 >= 0 ? c.seg_tail2 : FG_SEG_TAIL2_DEFAULT; }
// seg_parts = jobs per tile of the segmented backward (0 / 1 = off)
int seg_parts(const Cfg& c, int n_tiles) {
  // below 5000 tiles there are fewer tiles than wavefront slots: 6 shares per tile (960x540: backward
  // 0.181 -> 0.178, 1280x720: 0.246 -> 0.236; 8: 0.175 / 0.229)
  const int v = c.seg_parts >= 0 ? c.seg_parts : (n_tiles < 5000 ? FG_SEG_PARTS_SMALL : FG_SEG_PARTS_DEFAULT);
  return v < 1 ? 1 : (v > 16 ? 16 : v);
}

// Zero-fill as a kernel of our own: a hipMemsetAsync issued on a stream under torch's graph capture did
// not end up in the graph here (replays then accumulated onto the previous replay's gradients).
__global__ void __launch_bounds__(256) zero_fill_kernel(float* __restrict__ p, long long n) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) p[i] = 0.f;
}
int zero_fill(float* p, long long n, hipStream_t s) {
  if (n <= 0) return FG_OK;
  const long long blocks = (n + 1023) / 1024;
  hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, s, p, n);
  return hipGetLastError() == hipSuccess ? FG_OK : FG_ERR_LAUNCH;
}

template <int C>
int launch_fwd_mixed(const Cfg& cfg, int width, int height, int tail, const int32_t* jobs, const float* splats,
                     const int32_t* tile_offsets, const int32_t* flatten_ids, float* render, float* alphas,
                     int32_t* last_ids, Composite comp, hipStream_t s, float* ckpt = nullptr,
                     uint32_t* live_words = nullptr, float* zero_buf = nullptr, long long zero_floats = 0,
                     long long* walk_out = nullptr) {
  const int tile_w = (width + TILE - 1) / TILE, tile_h = (height + TILE - 1) / TILE;
  const int cap = jobs_cap(cfg, tile_w, tile_h);
  if (zero_buf && ((zero_floats & 3) || (reinterpret_cast<uintptr_t>(zero_buf) & 15))) {
    if (zero_fill(zero_buf, zero_floats, s) != FG_OK) return FG_ERR_LAUNCH;
    zero_buf = nullptr;
  }
  // compact checkpoint slots (seg_slots): the tiles' first slots are in the list's table; without a list there is no
  // table and no checkpoints
  const int32_t* slot_tab = slot_table(cfg, jobs, tile_w, tile_h);
  if (seg_slots(cfg) > 0 && !slot_tab) ckpt = nullptr;
  // wide jobs: the open strips' list lives where the local jobs' list would (unused in this form; its eight count words are
  // zeroed by every list build): written by this launch's prefix jobs, read and reset by the launch behind it
  const bool wide = C == 3 && jobs && ckpt && heavy_len(cfg) > 0 && heavy_wide(cfg);
  int32_t* const open_list = wide ? const_cast<int32_t*>(jobs) + 8 + 8 * (size_t)cap : nullptr;
  hipLaunchKernelGGL((raster_fwd_mixed_kernel<C>), dim3(jobs ? listed_grid(cfg, tile_w, tile_h, tail) : mixed_grid(cfg, tile_w, tile_h, tail)),
                     dim3(64), 0, s, width, height, tile_w, tile_h, band_nx(cfg), tail, jobs, cap,
                     reinterpret_cast<const float4*>(splats), tile_offsets, flatten_ids, render, alphas, last_ids, comp,
                     reinterpret_cast<float4*>(ckpt), live_words, reinterpret_cast<float4*>(zero_buf),
                     zero_buf ? zero_floats / 4 : 0ll, slot_tab, job_prio(cfg.prio_fwd, FG_PRIO_FWD_DEFAULT),
                     heavy_wide(cfg) ? FG_WIDE_PREFIX : FG_HEAVY_PREFIX, wide ? open_list : nullptr, walk_out, seg_fine(cfg));
  if constexpr (C == 3) {
    // heavy tiles: their combine jobs, once every local job has left its batches' composites
    if (wide) {
      // heavy tiles: the strips their prefix jobs left open, as wide jobs
      hipLaunchKernelGGL(raster_fwd_wide_kernel, dim3(FG_WIDE_GRID), dim3(64 * FG_WIDE_WAVES), 0, s, width, height, tile_w,
                         open_list, reinterpret_cast<const float4*>(splats), tile_offsets, flatten_ids, render,
                         alphas, last_ids, comp, reinterpret_cast<float4*>(ckpt), live_words, slot_tab, walk_out, seg_fine(cfg));
    } else if (jobs && ckpt && heavy_len(cfg) > 0) {
      const int32_t* local = jobs + 8 + 8 * (size_t)cap;
      hipLaunchKernelGGL(raster_fwd_local_kernel, dim3(8 * 1024), dim3(64), 0, s, width, height, tile_w, local,
                         reinterpret_cast<const float4*>(splats), tile_offsets, flatten_ids, render, alphas, last_ids, comp,
                         reinterpret_cast<float4*>(ckpt), live_words, slot_tab);
      hipLaunchKernelGGL(raster_fwd_combine_kernel, dim3(8 * 256), dim3(64), 0, s, width, height, tile_w,
                         local + FG_LOCAL_WORDS, reinterpret_cast<const float4*>(splats), tile_offsets, flatten_ids, render,
                         alphas, last_ids, comp, reinterpret_cast<float4*>(ckpt), live_words, slot_tab);
    }
  }
  return hipGetLastError() == hipSuccess ? FG_OK : FG_ERR_LAUNCH;
}

template <int C>
int launch_bwd_mixed(const Cfg& cfg, int width, int height, int tail, const int32_t* jobs, const float* splats,
                     const int32_t* tile_offsets, const int32_t* flatten_ids, const float* alphas,
                     const int32_t* last_ids, const float* v_render, const float* v_alphas, float* v_splats,
                     Composite comp, hipStream_t s, Segments seg = Segments{nullptr, nullptr, nullptr, 1, 0, 0},
                     const uint32_t* live_words = nullptr) {
  const int tile_w = (width + TILE - 1) / TILE, tile_h = (height + TILE - 1) / TILE;
  const int cap = jobs_cap(cfg, tile_w, tile_h);
  int grid = jobs ? listed_grid(cfg, tile_w, tile_h, tail) : mixed_grid(cfg, tile_w, tile_h, tail);
  if (seg.parts > 1) grid = seg_grid(cfg, tile_w, tile_h, seg.parts, seg.tail, jobs != nullptr);
  if (live_words)
    hipLaunchKernelGGL((raster_bwd_mixed_kernel<C, true>), dim3(grid), dim3(64), 0, s, width, height, tile_w, tile_h, band_nx(cfg), tail,
                       jobs, cap, reinterpret_cast<const float4*>(splats), tile_offsets, flatten_ids, alphas, last_ids,
                       v_render, v_alphas, v_splats, comp, seg, live_words);
  else
    hipLaunchKernelGGL((raster_bwd_mixed_kernel<C, false>), dim3(grid), dim3(64), 0, s, width, height, tile_w, tile_h, band_nx(cfg), tail,
                       jobs, cap, reinterpret_cast<const float4*>(splats), tile_offsets, flatten_ids, alphas, last_ids,
                       v_render, v_alphas, v_splats, comp, seg, live_words);
  return hipGetLastError() == hipSuccess ? FG_OK : FG_ERR_LAUNCH;
}



#define FG_DISPATCH_C(CALL)                  \
  switch (channels) {                        \
    case 1: CALL(1); break;                  \
    case 2: CALL(2); break;                  \
    case 3: CALL(3); break;                  \
    case 4: CALL(4); break;                  \
    case 5: CALL(5); break;                  \
    case 6: CALL(6); break;                  \
    case 7: CALL(7); break;                  \
    case 8: CALL(8); break;                  \
    default: return FG_ERR_UNSUPPORTED;      \
  }

}  // namespace

extern "C" int fg_pack_splats(int N, int channels, const float* means2d, const float* conics,
                              const float* opacities, const float* features, float* splats,
                              fg_stream_t stream) {
  if (N < 0 || channels < 1 || channels > FG_MAX_CHANNELS) return FG_ERR_INVALID_ARG;
  if (N == 0) return FG_OK;
  if (!means2d || !conics || !opacities || !features || !splats) return FG_ERR_INVALID_ARG;
  const int64_t threads = (int64_t)N * 16;
  hipLaunchKernelGGL(pack_splats_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                     fg_hip_stream(stream), N, channels, means2d, conics, opacities, features, splats);
  FG_RETURN_IF_LAUNCH_FAILED();
  return FG_OK;
}

extern "C" int fg_unpack_grads(int N, int channels, const float* v_splats, float* v_means2d,
                               float* v_means2d_abs, float* v_conics, float* v_opacities,
                               float* v_features, fg_stream_t stream) {
  if (N < 0 || channels < 1 || channels > FG_MAX_CHANNELS) return FG_ERR_INVALID_ARG;
  if (N == 0) return FG_OK;
  if (!v_splats) return FG_ERR_INVALID_ARG;
  const int64_t threads = (int64_t)N * 16;
  hipLaunchKernelGGL(unpack_grads_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                     fg_hip_stream(stream), N, channels, v_splats, v_means2d, v_means2d_abs, v_conics,
                     v_opacities, v_features);
  FG_RETURN_IF_LAUNCH_FAILED();
  return FG_OK;
}

namespace {

int raster_fwd_any(const fg_raster_config* config, int channels, int width, int height, int tile_size, const float* splats,
                   const int32_t* tile_offsets, const int32_t* flatten_ids, float* render, float* alphas,
                   int32_t* last_ids, Composite comp, fg_stream_t stream, const int32_t* jobs = nullptr,
                   float* seg_ckpt = nullptr, uint32_t* live_words = nullptr, float* zero_buf = nullptr,
                   long long zero_floats = 0, long long* walk_out = nullptr) {
  if (width <= 0 || height <= 0 || zero_floats < 0) return FG_ERR_INVALID_ARG;
  if (tile_size != TILE) return FG_ERR_UNSUPPORTED;
  if (!splats || !tile_offsets || !render || !alphas || !last_ids) return FG_ERR_INVALID_ARG;
  if (comp.n_clamp < 0 || comp.n_clamp > channels || (comp.n_clamp > 0 && !comp.clamp_mask)) return FG_ERR_INVALID_ARG;
  hipStream_t s = fg_hip_stream(stream);
  int rc = FG_OK;
  const int n_tiles = ((width + TILE - 1) / TILE) * ((height + TILE - 1) / TILE);
  const Cfg cfg = resolve(config);
  const int ppt = raster_ppt_fwd(cfg, n_tiles);
  int tail = mixed_tail_fwd(cfg, n_tiles);
  if (tail == 0) jobs = nullptr;  // classic launch (small image / forced pixels per lane)
  if (zero_buf && zero_floats > 0 && tail == 0) {  // only the mixed launch zero-fills in passing
    if (zero_fill(zero_buf, zero_floats, s) != FG_OK) return FG_ERR_LAUNCH;
    zero_buf = nullptr;
  }
  if (zero_floats == 0) zero_buf = nullptr;
#define CALL(CC)                                                                                                    \
  rc = (tail > 0)   ? launch_fwd_mixed<CC>(cfg, width, height, tail, jobs, splats, tile_offsets, flatten_ids, render,    \
                                         alphas, last_ids, comp, s, CC == 3 ? seg_ckpt : nullptr, live_words,       \
                                         zero_buf, zero_floats, walk_out)                                           \
       : (ppt == 4) ? launch_fwd<CC, 4>(cfg, width, height, splats, tile_offsets, flatten_ids, render, alphas, last_ids, \
                                      comp, s)                                                                      \
       : (ppt == 2) ? launch_fwd<CC, 2>(cfg, width, height, splats, tile_offsets, flatten_ids, render, alphas, last_ids, \
                                      comp, s)                                                                      \
                    : launch_fwd<CC, 1>(cfg, width, height, splats, tile_offsets, flatten_ids, render, alphas, last_ids, \
                                      comp, s)
  FG_DISPATCH_C(CALL)
#undef CALL
  return rc;
}

int raster_bwd_any(const fg_raster_config* config, int channels, int width, int height, int tile_size, const float* splats,
                   const int32_t* tile_offsets, const int32_t* flatten_ids, const float* alphas,
                   const int32_t* last_ids, const float* v_render, const float* v_alphas, float* v_splats,
                   Composite comp, fg_stream_t stream, const int32_t* jobs = nullptr,
                   const float* seg_ckpt = nullptr, const float* image = nullptr,
                   const uint32_t* live_words = nullptr) {
  if (width <= 0 || height <= 0) return FG_ERR_INVALID_ARG;
  if (tile_size != TILE) return FG_ERR_UNSUPPORTED;
  if (!splats || !tile_offsets || !alphas || !last_ids || !v_render || !v_splats) return FG_ERR_INVALID_ARG;
  if (comp.n_clamp < 0 || comp.n_clamp > channels || (comp.n_clamp > 0 && !comp.clamp_mask)) return FG_ERR_INVALID_ARG;
  hipStream_t s = fg_hip_stream(stream);
  int rc = FG_OK;
  const int n_tiles = ((width + TILE - 1) / TILE) * ((height + TILE - 1) / TILE);
  const Cfg cfg = resolve(config);
  const int ppt = raster_ppt_bwd(cfg, n_tiles);
  int tail = mixed_tail_bwd(cfg, n_tiles);
  if (tail == 0) jobs = nullptr;
  // liveness words and checkpoints exist only if the forward of this image size was a mixed launch (a
  // forced tail4_bwd on a small image pairs a classic forward with a mixed backward)
  if (mixed_tail_fwd(cfg, n_tiles) == 0) {
    live_words = nullptr;
    seg_ckpt = nullptr;
  }
  // list segmentation: 3 channels, checkpoints written by the forward of this very image
  Segments seg{nullptr, nullptr, nullptr, 1, 0, 0};
  // (compact checkpoint slots: the table of the tiles' first slots is in the list; no list, no shares)
  const int32_t* slot_tab = slot_table(cfg, jobs, (width + TILE - 1) / TILE, (height + TILE - 1) / TILE);
  if (seg_slots(cfg) > 0 && !slot_tab) seg_ckpt = nullptr;
  if (channels == 3 && seg_ckpt && image && tail > 0 && seg_parts(cfg, n_tiles) > 1)
    seg = Segments{reinterpret_cast<float4*>(const_cast<float*>(seg_ckpt)), slot_tab, image, seg_parts(cfg, n_tiles),
                   seg_tail_fit(cfg, (width + TILE - 1) / TILE, (height + TILE - 1) / TILE, seg_parts(cfg, n_tiles),
                                seg_tail(cfg, n_tiles)), 0};
  seg.prio = job_prio(cfg.prio_bwd, FG_PRIO_BWD_DEFAULT);
  seg.fine = seg_fine(cfg);
#define CALL(CC)                                                                                            \
  rc = (tail > 0)   ? launch_bwd_mixed<CC>(cfg, width, height, tail, jobs, splats, tile_offsets, flatten_ids,    \
                                         alphas, last_ids, v_render, v_alphas, v_splats, comp, s, seg,      \
                                         live_use(cfg, live_words))                                              \
       : (ppt == 4) ? launch_bwd<CC, 4>(cfg, width, height, splats, tile_offsets, flatten_ids, alphas, last_ids, \
                                      v_render, v_alphas, v_splats, comp, s)                                \
       : (ppt == 2) ? launch_bwd<CC, 2>(cfg, width, height, splats, tile_offsets, flatten_ids, alphas, last_ids, \
                                      v_render, v_alphas, v_splats, comp, s)                                \
                    : launch_bwd<CC, 1>(cfg, width, height, splats, tile_offsets, flatten_ids, alphas, last_ids, \
                                      v_render, v_alphas, v_splats, comp, s)
  FG_DISPATCH_C(CALL)
#undef CALL
  return rc;
}

}  // namespace

extern "C" int fg_raster_fwd(int channels, int width, int height, int tile_size, const float* splats,
                             const int32_t* tile_offsets, const int32_t* flatten_ids, float* render,
                             float* alphas, int32_t* last_ids, const fg_raster_config* config, fg_stream_t stream) {
  return raster_fwd_any(config, channels, width, height, tile_size, splats, tile_offsets, flatten_ids, render, alphas,
                        last_ids, Composite{nullptr, 0, nullptr}, stream);
}

extern "C" int fg_raster_bwd(int channels, int width, int height, int tile_size, const float* splats,
                             const int32_t* tile_offsets, const int32_t* flatten_ids, const float* alphas,
                             const int32_t* last_ids, const float* v_render, const float* v_alphas,
                             float* v_splats, const fg_raster_config* config, fg_stream_t stream) {
  return raster_bwd_any(config, channels, width, height, tile_size, splats, tile_offsets, flatten_ids, alphas, last_ids,
                        v_render, v_alphas, v_splats, Composite{nullptr, 0, nullptr}, stream);
}

extern "C" int fg_raster_composite_fwd(int channels, int width, int height, int tile_size, const float* splats,
                                       const int32_t* tile_offsets, const int32_t* flatten_ids,
                                       const float* background, int n_clamp, float* image, float* alphas,
                                       int32_t* last_ids, uint8_t* clamp_mask, const fg_raster_config* config,
                                       fg_stream_t stream) {
  return raster_fwd_any(config, channels, width, height, tile_size, splats, tile_offsets, flatten_ids, image, alphas,
                        last_ids, Composite{background, n_clamp, clamp_mask}, stream);
}

extern "C" int fg_raster_composite_bwd(int channels, int width, int height, int tile_size, const float* splats,
                                       const int32_t* tile_offsets, const int32_t* flatten_ids,
                                       const float* background, int n_clamp, const uint8_t* clamp_mask,
                                       const float* alphas, const int32_t* last_ids, const float* v_image,
                                       const float* v_alphas, float* v_splats, const fg_raster_config* config,
                                       fg_stream_t stream) {
  return raster_bwd_any(config, channels, width, height, tile_size, splats, tile_offsets, flatten_ids, alphas, last_ids,
                        v_image, v_alphas, v_splats, Composite{background, n_clamp, const_cast<uint8_t*>(clamp_mask)},
                        stream);
}

// ---- job lists ---------------------------------------------------------------------------------
extern "C" void fg_raster_config_init(fg_raster_config* c) {
  if (!c) return;
  c->size = (int32_t)sizeof(fg_raster_config);
  c->ppt_fwd = c->ppt_bwd = 0;
  c->tile_order = -1;
  c->bands_nx = 0;
  c->tail4_fwd = c->tail2_fwd = c->tail4_bwd = c->tail2_bwd = -1;
  c->split4_fwd = c->split2_fwd = c->split4_bwd = c->split2_bwd = -1;
  c->use_liveness = 1;
  c->seg_parts = c->seg_tail = c->seg_parts2 = c->seg_tail2 = -1;
  c->debug_only_xcd = -1;
  c->debug_k_mod = 0;
  c->balance_bands = -1;
  c->heavy_tiles = 0;
  c->seg_slots = 0;
  c->prio_fwd = c->prio_bwd = -1;
  c->heavy_wide = -1;
  c->seg_fine = -1;
}

extern "C" int64_t fg_raster_jobs_words(int width, int height, int tile_size, const fg_raster_config* config) {
  if (width <= 0 || height <= 0 || tile_size != TILE) return 0;
  const int tile_w = (width + TILE - 1) / TILE, tile_h = (height + TILE - 1) / TILE;
  const int n_tiles = tile_w * tile_h;
  const Cfg cfg = resolve(config);
  if ((tile_order_mode(cfg) & 255) != 2 || (tile_order_mode(cfg) >> 8) != 0) return 0;
  if (mixed_tail_fwd(cfg, n_tiles) == 0 && mixed_tail_bwd(cfg, n_tiles) == 0) return 0;  // classic launches: no lists
  // (+ the heavy tiles' two lists behind the main one, + the table of the tiles' first checkpoint slots)
  return slot_table_offset(cfg, tile_w, tile_h) + n_tiles;
}

int fgjobs::plan_jobs(int width, int height, int tile_size, int32_t* jobs_fwd, int32_t* jobs_bwd, int bwd_list_shares,
                      const fg_raster_config* config, fgjobs::JobBuild* out) {
  out->jobs_fwd = out->jobs_bwd = nullptr;
  if (width <= 0 || height <= 0) return FG_ERR_INVALID_ARG;
  if (tile_size != TILE) return FG_ERR_UNSUPPORTED;
  if (!jobs_fwd && !jobs_bwd) return FG_OK;
  const int tile_w = (width + TILE - 1) / TILE, tile_h = (height + TILE - 1) / TILE;
  const int n_tiles = tile_w * tile_h;
  const Cfg cfg = resolve(config);
  const int tf = mixed_tail_fwd(cfg, n_tiles), tb = mixed_tail_bwd(cfg, n_tiles);
  const int sf = raster_split(cfg.split4_fwd, cfg.split2_fwd, FG_SPLIT4_FWD, FG_SPLIT2_FWD);
  const int sb = raster_split(cfg.split4_bwd, cfg.split2_bwd, FG_SPLIT4_BWD, FG_SPLIT2_BWD);
  const int sp = seg_parts(cfg, n_tiles);
  const bool shares = bwd_list_shares && tb > 0 && sp > 1;
  // heavy tiles need the checkpoint buffer (three channels, list shares on) and the forward's list
  const int hl = shares && tf > 0 && jobs_fwd ? heavy_len(cfg) : 0;
  const int hw = hl > 0 && heavy_wide(cfg);
  const JobParams pf{tf & 0xFFFF, tf >> 16, sf & 0xFFFF, sf >> 16, listed_grid(cfg, tile_w, tile_h, tf) / 8, 0, 0, 0, 0, hl, hw};
  // the backward's list: pixel strips, or (bwd_list_shares: the caller will hand the checkpoint buffer
  // of fg_raster_seg_ckpt_floats to both raster calls) shares of the tiles' lists
  const int st = seg_tail_fit(cfg, tile_w, tile_h, sp, seg_tail(cfg, n_tiles));
  const JobParams pb = shares ? JobParams{0, 0, 0, sb >> 16, seg_grid(cfg, tile_w, tile_h, sp, st, true) / 8, sp, st,
                                          seg_parts2(cfg), seg_tail2(cfg) < st ? seg_tail2(cfg) : st, hl, hw}
                              : JobParams{tb & 0xFFFF, tb >> 16, sb & 0xFFFF, sb >> 16, listed_grid(cfg, tile_w, tile_h, tb) / 8, 0, 0, 0, 0, 0, 0};
  const int rows_limit = band_rows_limit(cfg, tile_h);  // (the grids and list segments are sized for it: band_tiles_max)
  *out = fgjobs::JobBuild{tile_w, tile_h, band_nx(cfg), jobs_cap(cfg, tile_w, tile_h), pf, pb, jobs_fwd, jobs_bwd,
                          8 + 8 * jobs_cap(cfg, tile_w, tile_h), rows_limit,
                          cfg.balance_bands == 0 ? 0 : (cfg.balance_bands == 2 ? -1 : (cfg.balance_bands == 3 ? -2 : (cfg.balance_bands > 3 ? cfg.balance_bands : 115))),
                          shares ? seg_slots(cfg) / 8 : 0, (int)slot_table_offset(cfg, tile_w, tile_h), nullptr, seg_fine(cfg)};
  return FG_OK;
}

extern "C" int fg_raster_build_jobs(int width, int height, int tile_size, const int32_t* tile_offsets,
                                    int32_t* jobs_fwd, int32_t* jobs_bwd, int bwd_list_shares,
                                    const fg_raster_config* config, fg_stream_t stream) {
  if (!tile_offsets) return FG_ERR_INVALID_ARG;
  fgjobs::JobBuild jb;
  const int rc = fgjobs::plan_jobs(width, height, tile_size, jobs_fwd, jobs_bwd, bwd_list_shares, config, &jb);
  if (rc != FG_OK) return rc;
  if (!jb.jobs_fwd && !jb.jobs_bwd) return FG_OK;
  hipLaunchKernelGGL(build_jobs_kernel, dim3(fgjobs::FG_JOB_BLOCKS), dim3(1024), 0, fg_hip_stream(stream), jb, tile_offsets);
  FG_RETURN_IF_LAUNCH_FAILED();
  return FG_OK;
}

extern "C" int fg_raster_jobs_fwd(int channels, int width, int height, int tile_size, const float* splats,
                                  const int32_t* tile_offsets, const int32_t* flatten_ids, const int32_t* jobs,
                                  const float* background, int n_clamp, float* image, float* alphas,
                                  int32_t* last_ids, uint8_t* clamp_mask, float* seg_ckpt, uint32_t* live_words,
                                  float* zero_buf, int64_t zero_floats, int64_t* walk_out,
                                  const fg_raster_config* config, fg_stream_t stream) {
  return raster_fwd_any(config, channels, width, height, tile_size, splats, tile_offsets, flatten_ids, image, alphas,
                        last_ids, Composite{background, n_clamp, clamp_mask}, stream, jobs, seg_ckpt, live_words,
                        zero_buf, (long long)zero_floats, reinterpret_cast<long long*>(walk_out));
}

extern "C" int64_t fg_raster_seg_ckpt_floats(int channels, int width, int height, int tile_size, int64_t n_isects,
                                             const fg_raster_config* config) {
  if (channels != 3 || width <= 0 || height <= 0 || tile_size != TILE || n_isects <= 0) return 0;
  const int n_tiles = ((width + TILE - 1) / TILE) * ((height + TILE - 1) / TILE);
  const Cfg cfg = resolve(config);
  if (seg_parts(cfg, n_tiles) <= 1 || mixed_tail_bwd(cfg, n_tiles) == 0 || mixed_tail_fwd(cfg, n_tiles) == 0) return 0;
  const int64_t slots = seg_slots(cfg) > 0 ? seg_slots(cfg) : n_isects / FG_SEG_ENTRIES + n_tiles + 2;
  return ((int64_t)seg_slots_offset4(n_tiles, width, height) + slots * (int64_t)FG_SEG_SLOT4) * 4;
}

extern "C" int fg_raster_jobs_bwd(int channels, int width, int height, int tile_size, const float* splats,
                                  const int32_t* tile_offsets, const int32_t* flatten_ids, const int32_t* jobs,
                                  const float* background, int n_clamp, const uint8_t* clamp_mask,
                                  const float* alphas, const int32_t* last_ids, const float* v_image,
                                  const float* v_alphas, float* v_splats, const float* seg_ckpt,
                                  const float* image, const uint32_t* live_words, const fg_raster_config* config,
                                  fg_stream_t stream) {
  return raster_bwd_any(config, channels, width, height, tile_size, splats, tile_offsets, flatten_ids, alphas, last_ids,
                        v_image, v_alphas, v_splats, Composite{background, n_clamp, const_cast<uint8_t*>(clamp_mask)},
                        stream, jobs, seg_ckpt, image, live_words);
}

#ifdef FG_RASTER_STATS
// out[16] on the host; reset != 0 clears the counters afterwards.  Only in libfgraster_stats.so.
extern "C" int fg_debug_raster_stats(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(fg_raster_stats), sizeof(unsigned long long) * 16) != hipSuccess)
    return FG_ERR_LAUNCH;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(fg_raster_stats), z, sizeof(z)) != hipSuccess) return FG_ERR_LAUNCH;
  }
  return FG_OK;
}
#endif

#ifdef FG_RASTER_TIMELINE
// Copies up to cap records (4 x u64 each) to the host and returns how many jobs were recorded.
extern "C" int fg_debug_raster_timeline(unsigned long long* out, int cap, int reset) {
  unsigned n = 0;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(fg_timeline_n), 4) != hipSuccess) return -1;
  const unsigned m = n < (unsigned)cap ? n : (unsigned)cap;
  if (m && hipMemcpyFromSymbol(out, HIP_SYMBOL(fg_timeline), (size_t)(m < FG_TL_CAP ? m : FG_TL_CAP) * 48) != hipSuccess)
    return -1;
  if (reset) {
    const unsigned z = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(fg_timeline_n), &z, 4) != hipSuccess) return -1;
  }
  return (int)n;
}
#endif
