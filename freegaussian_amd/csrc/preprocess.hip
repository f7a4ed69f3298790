// Fused per-Gaussian stages: K1 + K2 + record packing (forward) and record unpacking + K8 + K7
// (backward), one pass over the parameters each way.
//
// Forward  reads 44 B (means, quats, scales, opacity) + 12k B of SH coefficients per Gaussian and
// writes the info arrays (radii, means2d, depths, conics, tiles_touched: 32 B) plus the 64-byte
// splat record the raster kernels gather.  Backward reads the 64-byte gradient record the raster
// backward accumulated (+ v_means2d, which autograd routes separately so that
// info["means2d"].grad exists) and writes v_means / v_quats / v_scales / v_opacities and the dense
// 192-byte v_coeffs row.  One 64-lane wavefront per 64 Gaussians; the coefficient rows (32 at a time), the
// [64 x 16] record slab and the [64 x 10] note for the backward stream through LDS so that every global access
// is a fully coalesced 16-byte-per-lane stream; rows are read back per lane at padded strides (49 / 20 / 11
// floats) that are LDS-bank-conflict free.  The backward reads the forward's note (d colour / d direction +
// clamp mask, 40 B) instead of the coefficient row.
//
// Same arithmetic as project.hip / sh.hip (shared headers); results are bit-identical to the
// unfused entry points.
#include "project_math.h"
#include "sh_math.h"
#include "slab_io.h"

#include <stdlib.h>

namespace {
using namespace fgp;
using namespace fgsh;

constexpr int REC = FG_SPLAT_FLOATS;  // 16 floats per record
constexpr int RSTRIDE = 20;           // padded LDS stride of a record (floats, 16-B aligned)
// The forward's note for the backward of the SH colour (sh_degree >= 1): J[c][d] = d colour_c / d dir_d before the
// clamp (9 floats, row-major by channel) and the clamp mask (bit c: channel c passed max(. + 0.5, 0)) as the bits
// of a 10th.  The backward then needs 40 bytes per Gaussian instead of the 192-byte coefficient row.
constexpr int JAC = FG_SH_JAC_FLOATS;
constexpr int JSTRIDE = 11;           // odd LDS stride of a Jacobian row

struct FeatLayout {
  int sh_degree;  // >= 0: colours are SH coefficients [N,k_stored,3]; -1: direct colours [N,n_color]
  int k_stored;
  int n_color;    // colour channels composited (3 for SH; 0 if the render mode has no colour)
  int with_depth; // append the camera depth as a channel
  int n_extra;    // extra per-Gaussian channels [N,n_extra]
};

// Raw parameter forms of the reference's gauss_params (freegaussian_model.py:187-196) and the
// activations its get_outputs applies before the raster call (:801, :844-851), folded into this
// pass:  quats -> q/|q| (+ d_quats),  scales = log-scales -> exp(.)
// (+ d_scales),  opacities = logits -> sigmoid(.),  colours split into features_dc [N,3] (passed as
// `colors`) and features_rest [N,k_stored-1,3].
struct RawForm {
  int enabled;
  const float* d_quats;        // [N,4] nullable
  const float* d_scales;       // [N,3] nullable
  const float* features_rest;  // [N,k_stored-1,3]
};

struct Activated {
  float q[4], s[3], o;
  float qn[4], inv_norm;  // normalised raw quaternion and 1/|q| (raw form only)
  float es[3];            // exp(log-scale) (raw form only)
};

__device__ __forceinline__ Activated load_activated(const RawForm& raw, int i, const float* __restrict__ quats,
                                                    const float* __restrict__ scales,
                                                    const float* __restrict__ opacities) {
  Activated a;
  const float4 q = reinterpret_cast<const float4*>(quats)[i];
  const float s0 = scales[3 * i], s1 = scales[3 * i + 1], s2 = scales[3 * i + 2];
  const float o = opacities[i];
  if (!raw.enabled) {
    a.q[0] = q.x; a.q[1] = q.y; a.q[2] = q.z; a.q[3] = q.w;
    a.s[0] = s0; a.s[1] = s1; a.s[2] = s2;
    a.o = o;
    a.inv_norm = 1.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) a.qn[c] = a.q[c];
#pragma unroll
    for (int c = 0; c < 3; ++c) a.es[c] = a.s[c];
    return a;
  }
  a.inv_norm = 1.f / sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  a.qn[0] = q.x * a.inv_norm; a.qn[1] = q.y * a.inv_norm; a.qn[2] = q.z * a.inv_norm; a.qn[3] = q.w * a.inv_norm;
  a.es[0] = expf(s0); a.es[1] = expf(s1); a.es[2] = expf(s2);
#pragma unroll
  for (int c = 0; c < 4; ++c) a.q[c] = a.qn[c] + (raw.d_quats ? raw.d_quats[4 * i + c] : 0.f);
#pragma unroll
  for (int c = 0; c < 3; ++c) a.s[c] = a.es[c] + (raw.d_scales ? raw.d_scales[3 * i + c] : 0.f);
  a.o = 1.f / (1.f + expf(-o));
  return a;
}

// The coefficient rows of a 64-Gaussian workgroup come through LDS in two HALVES of 32 rows: 6.3 KB of LDS per
// workgroup instead of 12.5 and 24 instead of 48 registers of loads in flight per lane -- these passes run as
// fast as a CU holds wavefronts (forward 87 / 100 / 117 us at 12 / 9 / 7 workgroups per CU,
// profiles/r03_preprocess_occupancy.md).  Half h: rows 32 h .. 32 h + 31 of the workgroup at LDS rows 0 .. 31;
// lanes 32 h .. 32 h + 31 then evaluate their rows (the other half idles through that stretch).
// Raw form: the row is split in two arrays, features_dc [N,3] (`colors`, base 0) and features_rest
// [N,k_stored-1,3] (bases 1..).
constexpr int HROWS = BLOCK / 2;
static_assert(HROWS * ROW >= BLOCK * RSTRIDE && HROWS * ROW >= BLOCK * JSTRIDE, "record / note slabs reuse the LDS rows");
#ifndef FG_PRE_FWD_WAVES
#define FG_PRE_FWD_WAVES 6
#endif
#ifndef FG_PRE_BWD_WAVES
#define FG_PRE_BWD_WAVES 4
#endif
__device__ __forceinline__ void stage_coeffs_half(float* lds, const FeatLayout& fl, const RawForm& raw,
                                                  const float* __restrict__ colors, int row0, int nrows, int kk, int h,
                                                  const uint8_t* row_live) {
  const int r0 = row0 + h * HROWS, nr = min(HROWS, nrows - h * HROWS);
  if (nr <= 0) return;
  const uint8_t* live = row_live ? row_live + h * HROWS : nullptr;
  if (!raw.enabled) {
    slab_to_lds_at<ROW, HROWS>(lds, 0, colors + (size_t)r0 * 3 * fl.k_stored, nr, 3 * fl.k_stored, 3 * kk, live);
    return;
  }
  slab_to_lds_at<ROW, HROWS>(lds, 0, colors + (size_t)r0 * 3, nr, 3, 3);  // features_dc: 12 B rows, always fetched
  if (kk > 1)
    slab_to_lds_at<ROW, HROWS>(lds, 3, raw.features_rest + (size_t)r0 * 3 * (fl.k_stored - 1), nr, 3 * (fl.k_stored - 1),
                               3 * (kk - 1), live);
}

// What the SH colour and its backward need from a Gaussian's coefficient row: the colour before + 0.5 / clamp, and
// (want_jac) d colour_c / d direction_d.
struct ShSums {
  float col[3];
  float jac[9];
};
__device__ __forceinline__ void sh_row_sums(ShSums& a, const float* row, int degree, int kk, float dx, float dy, float dz,
                                            bool want_jac) {
  float basis[16];
  sh_basis(degree, dx, dy, dz, basis);
  sh_dot(basis, row, kk, a.col[0], a.col[1], a.col[2]);
  if (want_jac) {
    float bx[16], by[16], bz[16];
    sh_basis_grad(degree, dx, dy, dz, bx, by, bz);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float jx = 0.f, jy = 0.f, jz = 0.f;
#pragma unroll
      for (int k = 1; k < 16; ++k)
        if (k < kk) {
          jx += bx[k] * row[3 * k + c]; jy += by[k] * row[3 * k + c]; jz += bz[k] * row[3 * k + c];
        }
      a.jac[3 * c] = jx; a.jac[3 * c + 1] = jy; a.jac[3 * c + 2] = jz;
    }
  }
}

// Shrink the tile range [t0, t1) to the tiles whose pixel centres the extent [g - e, g + e] reaches
// (fg::extent_reaches, the comparison the raster kernels' strip masks use): a closed-form guess,
// then one fix-up step each way against that very comparison, so that the two can never disagree.
__device__ __forceinline__ void tighten_tile_range(float g, float e, float ts, int& t0, int& t1) {
  if (t1 <= t0) return;
  const float lo_f = fminf(fmaxf(ceilf((g - e - (ts - 0.5f)) / ts), (float)t0), (float)t1);
  const float hi_f = fminf(fmaxf(floorf((g + e - 0.5f) / ts), (float)t0 - 1.f), (float)(t1 - 1));
  int lo = (int)lo_f, hi = (int)hi_f;  // first tile, last tile (inclusive)
  if (lo > t0 && fg::extent_reaches(g, e, (float)(lo - 1) * ts, ts)) --lo;
  else if (lo <= hi && !fg::extent_reaches(g, e, (float)lo * ts, ts)) ++lo;
  if (hi < t1 - 1 && fg::extent_reaches(g, e, (float)(hi + 1) * ts, ts)) ++hi;
  else if (hi >= lo && !fg::extent_reaches(g, e, (float)hi * ts, ts)) --hi;
  t0 = lo;
  t1 = hi + 1 > lo ? hi + 1 : lo;
}

// PACK_ONLY: the projection outputs (radii, means2d, depths, conics, compensations) are INPUTS
// (written by fg_project_fwd); only the colour + record part runs -- the half of the forward that
// can overlap the latency-bound binning on a second stream (fg_sh_pack_fwd).
template <bool PACK_ONLY>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(FG_PRE_FWD_WAVES)))
preprocess_fwd_kernel(int N, FeatLayout fl, RawForm raw, const float* __restrict__ means, const float* __restrict__ quats,
                      const float* __restrict__ scales, const float* __restrict__ opacities,
                      const float* __restrict__ colors, const float* __restrict__ extra,
                      const float* __restrict__ viewmat, const float* __restrict__ K, int width, int height,
                      float eps2d, float near_plane, float far_plane, float radius_clip, int tile_size, int tile_w,
                      int tile_h, int antialiased, int32_t* __restrict__ radii, float* __restrict__ means2d,
                      float* __restrict__ depths, float* __restrict__ conics, float* __restrict__ compensations,
                      int32_t* __restrict__ tiles_touched, float* __restrict__ splats,
                      uint32_t* __restrict__ depth_keys, int2* __restrict__ tile_rects,
                      unsigned long long* __restrict__ tile_masks, float* __restrict__ sh_jac, int skip_culled) {
  __shared__ float lds[HROWS * ROW];  // half the coefficient slab at a time, then the record slab, then the Jacobian slab
  __shared__ uint8_t row_live[BLOCK];
  const int row0 = blockIdx.x * BLOCK;
  const int nrows = min(BLOCK, N - row0);
  const int i = row0 + threadIdx.x;
  const int kk = fl.sh_degree >= 0 ? (fl.sh_degree + 1) * (fl.sh_degree + 1) : 0;
  // skip_culled: the coefficient rows are fetched AFTER the projection, visible Gaussians only (a
  // culled Gaussian's row is 192 of its 236 bytes); otherwise up front, under the projection
  if (kk > 0 && !skip_culled) stage_coeffs_half(lds, fl, raw, colors, row0, nrows, kk, 0, nullptr);

  // ---- K1 -------------------------------------------------------------------------------------
  bool ok = false;
  Fwd f;
  float mx = 0.f, my = 0.f, mz = 0.f, opac = 0.f;
  if (PACK_ONLY) {
    if (i < N) {
      mx = means[3 * i]; my = means[3 * i + 1]; mz = means[3 * i + 2];
      opac = opacities[i];
      ok = radii[i] > 0;
      f.m2x = means2d[2 * i]; f.m2y = means2d[2 * i + 1];
      f.pz = depths[i];
      f.conic_a = conics[3 * i]; f.conic_b = conics[3 * i + 1]; f.conic_c = conics[3 * i + 2];
      f.comp = compensations ? compensations[i] : 1.f;
      f.radius_f = 0.f;
    }
  } else if (i < N) {
    const Cam cam = load_cam(viewmat, K);
    mx = means[3 * i]; my = means[3 * i + 1]; mz = means[3 * i + 2];
    const Activated a = load_activated(raw, i, quats, scales, opacities);
    opac = a.o;
    f = project_core(cam, mx, my, mz, a.q[0], a.q[1], a.q[2], a.q[3], a.s[0], a.s[1], a.s[2], width, height,
                     eps2d);
    ok = (f.pz >= near_plane) && (f.pz <= far_plane) && (f.det > 0.0f);
    ok = ok && isfinite(f.radius_f) && (f.radius_f > radius_clip);
    const float fw = (float)width, fh = (float)height;
    ok = ok && !((f.m2x + f.radius_f <= 0.0f) || (f.m2x - f.radius_f >= fw) || (f.m2y + f.radius_f <= 0.0f) ||
                 (f.m2y - f.radius_f >= fh));
  }
  float rec[REC];
#pragma unroll
  for (int c = 0; c < REC; ++c) rec[c] = 0.f;
  float jac[JAC];
#pragma unroll
  for (int c = 0; c < JAC; ++c) jac[c] = 0.f;
  int32_t radius = 0, touched = 0;
  float o_comp = 0.f;
  int x0 = 0, x1 = 0, y0 = 0, y1 = 0;  // the reference's tile rectangle (radius box)
  if (ok) {
    radius = (int32_t)f.radius_f;
    const float ts = (float)tile_size;
    const float r = (float)radius / ts;
    const float tx = f.m2x / ts, ty = f.m2y / ts;
    x0 = min(max((int)floorf(tx - r), 0), tile_w), x1 = min(max((int)ceilf(tx + r), 0), tile_w);
    y0 = min(max((int)floorf(ty - r), 0), tile_h), y1 = min(max((int)ceilf(ty + r), 0), tile_h);
    touched = (x1 - x0) * (y1 - y0);
    o_comp = f.comp;
    rec[0] = f.m2x; rec[1] = f.m2y;
    rec[2] = antialiased ? opac * f.comp : opac;
    rec[3] = f.conic_a; rec[4] = f.conic_b; rec[5] = f.conic_c;
  }
  if (!PACK_ONLY && i < N) {
    radii[i] = radius;
    means2d[2 * i] = rec[0];
    means2d[2 * i + 1] = rec[1];
    depths[i] = ok ? f.pz : 0.f;
    conics[3 * i] = rec[3]; conics[3 * i + 1] = rec[4]; conics[3 * i + 2] = rec[5];
    if (compensations) compensations[i] = o_comp;
    tiles_touched[i] = touched;
    // Inputs of the depth-first binning, produced here because everything is in registers: the
    // depth sort key and the FOOTPRINT rectangle -- the radius box shrunk to the tiles the splat can
    // reach with alpha >= 1/255 (fg::alpha_extent; 31% fewer (splat, tile) pairs on the 1M / 1080p
    // scene: a 3-sigma box ignores the opacity).  Lists built from it are the reference's lists
    // minus entries that contribute to no pixel, in the same order.
    if (depth_keys) depth_keys[i] = ok ? (uint32_t)__float_as_int(f.pz) : 0xFFFFFFFFu;
    if (tile_rects) {
      int2 rc = make_int2(0, 0);
      if (ok) {
        float ex, ey;
        const int kind = fg::alpha_extent(rec[2], rec[3], rec[4], rec[5], ex, ey);
        int a0 = x0, a1 = x1, b0 = y0, b1 = y1;
        if (kind == 0) a1 = a0;
        if (kind == 1) {
          tighten_tile_range(rec[0], ex, (float)tile_size, a0, a1);
          tighten_tile_range(rec[1], ey, (float)tile_size, b0, b1);
        }
        if (a1 > a0 && b1 > b0) rc = make_int2(a0 | (b0 << 16), (a1 - a0) | ((b1 - b0) << 16));
      }
      tile_rects[i] = rc;
      // ... and which blocks of that rectangle the ellipse itself reaches (fg::footprint_mask)
      if (tile_masks)
        tile_masks[i] = rc.y == 0 ? 0ull
                                  : fg::footprint_mask(rec[2], rec[3], rec[4], rec[5], rec[0], rec[1], rc.x & 0xFFFF, rc.x >> 16,
                                                       rc.y & 0xFFFF, rc.y >> 16, (float)tile_size);
    }
  }

  // ---- K2 + features ----------------------------------------------------------------------------
  if (kk > 0 && skip_culled) {
    row_live[threadIdx.x] = ok;
    __syncthreads();
    stage_coeffs_half(lds, fl, raw, colors, row0, nrows, kk, 0, row_live);
  }
  const bool want_jac = sh_jac && kk > 1;
  ShSums sums;
#pragma unroll
  for (int c = 0; c < 3; ++c) sums.col[c] = 0.f;
#pragma unroll
  for (int c = 0; c < 9; ++c) sums.jac[c] = 0.f;
  if (kk > 0) {
    float dx = 0.f, dy = 0.f, dz = 1.f;
    if (ok) {
      float inv;
      view_dir(viewmat, mx, my, mz, dx, dy, dz, inv);
    }
    const float* row = lds + (threadIdx.x & (HROWS - 1)) * ROW;
    __syncthreads();  // first half staged
    if (ok && threadIdx.x < HROWS) sh_row_sums(sums, row, fl.sh_degree, kk, dx, dy, dz, want_jac);
    __syncthreads();
    stage_coeffs_half(lds, fl, raw, colors, row0, nrows, kk, 1, skip_culled ? row_live : nullptr);
    __syncthreads();
    if (ok && threadIdx.x >= HROWS) sh_row_sums(sums, row, fl.sh_degree, kk, dx, dy, dz, want_jac);
  }
  if (ok) {
    int c0 = 6;
    if (kk > 0) {
      const float r = sums.col[0], g = sums.col[1], b = sums.col[2];
      rec[6] = fmaxf(r + 0.5f, 0.f); rec[7] = fmaxf(g + 0.5f, 0.f); rec[8] = fmaxf(b + 0.5f, 0.f);
      c0 = 9;
      if (want_jac) {
#pragma unroll
        for (int c = 0; c < 9; ++c) jac[c] = sums.jac[c];
        // (the comparison the backward made on its recomputed colours)
        jac[9] = __int_as_float((int)(r + 0.5f > 0.f) | ((int)(g + 0.5f > 0.f) << 1) | ((int)(b + 0.5f > 0.f) << 2));
      }
    } else {
#pragma unroll
      for (int c = 0; c < FG_MAX_CHANNELS; ++c)
        if (c < fl.n_color) rec[6 + c] = colors[(size_t)i * fl.n_color + c];
      c0 = 6 + fl.n_color;
    }
    // depth and extra channels land at a run-time offset: unrolled select, no dynamic indexing
    const float depth_v = f.pz;
#pragma unroll
    for (int c = 6; c < REC - 2; ++c) {
      if (fl.with_depth && c == c0) rec[c] = depth_v;
      const int e = c - c0 - fl.with_depth;
      if (e >= 0 && e < fl.n_extra && c >= c0 + fl.with_depth) rec[c] = extra[(size_t)i * fl.n_extra + e];
    }
  }
  // ---- record slab out through LDS (coalesced 16 B per lane) -----------------------------------
  __syncthreads();
  {
    float* row = lds + threadIdx.x * RSTRIDE;
#pragma unroll
    for (int c = 0; c < REC; ++c) row[c] = rec[c];
  }
  __syncthreads();
  lds_to_slab_at<RSTRIDE, false>(splats + (size_t)row0 * REC, lds, 0, nrows, REC, REC);
  if (sh_jac && kk > 1) {  // ---- Jacobian slab out the same way
    __syncthreads();
    float* jrow = lds + threadIdx.x * JSTRIDE;
#pragma unroll
    for (int c = 0; c < JAC; ++c) jrow[c] = jac[c];
    __syncthreads();
    lds_to_slab_at<JSTRIDE, true>(sh_jac + (size_t)row0 * JAC, lds, 0, nrows, JAC, JAC);
  }
}

// NOTE: the forward's sh_jac is given (sh_degree >= 1) -- the coefficient rows are not read; otherwise the same sums
// are recomputed from the rows, in two windows like the forward.
template <bool NOTE>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(NOTE ? FG_PRE_BWD_WAVES : 3)))
preprocess_bwd_kernel(int N, FeatLayout fl, RawForm raw, float* __restrict__ v_d_quats,
                      float* __restrict__ v_d_scales, float* __restrict__ v_features_rest,
                      float* __restrict__ v_rgb, int v_rgb_floats,
                      const float* __restrict__ means, const float* __restrict__ quats,
                      const float* __restrict__ scales, const float* __restrict__ opacities,
                      const float* __restrict__ colors, const float* __restrict__ viewmat,
                      const float* __restrict__ K, int width, int height, float eps2d, int antialiased,
                      const int32_t* __restrict__ radii, const float* __restrict__ v_splats,
                      const float* __restrict__ v_means2d, int m2_stride, const float* __restrict__ v_depths,
                      const float* __restrict__ v_conics, float* __restrict__ v_means, float* __restrict__ v_quats,
                      float* __restrict__ v_scales, float* __restrict__ v_opacities, float* __restrict__ v_colors,
                      float* __restrict__ v_extra, const float* __restrict__ sh_jac, int skip_culled) {
  __shared__ float lds[HROWS * ROW];   // half the coefficient slab in (no note) / half the v_coeffs slab out at a time
  __shared__ uint8_t row_live[BLOCK];
  const int row0 = blockIdx.x * BLOCK;
  const int nrows = min(BLOCK, N - row0);
  const int i = row0 + threadIdx.x;
  const int kk = fl.sh_degree >= 0 ? (fl.sh_degree + 1) * (fl.sh_degree + 1) : 0;
  const bool active = (i < N) && radii[i] > 0;
  // the unit view direction the SH basis is evaluated at (also sent along by the factored exchange); with the
  // forward's note it is not needed before the projection backward and is computed behind it (registers)
  float sh_dx = 0.f, sh_dy = 0.f, sh_dz = 1.f, sh_inv = 0.f;
  auto view_direction = [&]() { view_dir(viewmat, means[3 * i], means[3 * i + 1], means[3 * i + 2], sh_dx, sh_dy, sh_dz, sh_inv); };
  if (!NOTE && active && kk > 0) view_direction();
  ShSums sums;
#pragma unroll
  for (int c = 0; c < 3; ++c) sums.col[c] = 0.f;
#pragma unroll
  for (int c = 0; c < 9; ++c) sums.jac[c] = 0.f;
  if (!NOTE && kk > 1) {
    if (skip_culled) {  // only the visible Gaussians' coefficient rows are read back
      row_live[threadIdx.x] = active;
      __syncthreads();
    }
    const float* row = lds + (threadIdx.x & (HROWS - 1)) * ROW;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (h) __syncthreads();  // everyone is done with the first half
      stage_coeffs_half(lds, fl, raw, colors, row0, nrows, kk, h, skip_culled ? row_live : nullptr);
      __syncthreads();
      if (active && (int)(threadIdx.x / HROWS) == h) sh_row_sums(sums, row, fl.sh_degree, kk, sh_dx, sh_dy, sh_dz, true);
    }
  }

  float g_m[3] = {0.f, 0.f, 0.f}, g_q[4] = {0.f, 0.f, 0.f, 0.f}, g_s[3] = {0.f, 0.f, 0.f};
  float g_o = 0.f;
  float vr = 0.f, vg = 0.f, vb = 0.f;
  float rec[REC];
#pragma unroll
  for (int c = 0; c < REC; ++c) rec[c] = 0.f;
  if (active) {
    // the lane's own 64-byte gradient record, four 16-byte loads (the lines are shared by the four
    // loads and served from cache after the first; staging them through LDS as well would cost
    // 20 KB per workgroup, i.e. a third of the resident wavefronts of this bandwidth-bound pass)
    const float4* rp = reinterpret_cast<const float4*>(v_splats) + (size_t)i * (REC / 4);
#pragma unroll
    for (int q4 = 0; q4 < REC / 4; ++q4) {
      const float4 v = rp[q4];
      rec[4 * q4] = v.x; rec[4 * q4 + 1] = v.y; rec[4 * q4 + 2] = v.z; rec[4 * q4 + 3] = v.w;
    }
  }
  // the feature gradients that pass straight through go out first (zeros for culled Gaussians): the record's
  // upper half is then dead through the projection backward (registers: 136 -> 128 = a fourth wavefront per SIMD)
  if (i < N) {
    const int ncol = kk > 0 ? 3 : fl.n_color;
    if (kk == 0) {
#pragma unroll
      for (int c = 0; c < FG_MAX_CHANNELS; ++c)
        if (c < fl.n_color) v_colors[(size_t)i * fl.n_color + c] = rec[8 + c];
    }
    if (v_extra) {
#pragma unroll
      for (int c = 0; c < FG_MAX_CHANNELS; ++c) {
        const int e = c - ncol - fl.with_depth;
        if (e >= 0 && e < fl.n_extra) v_extra[(size_t)i * fl.n_extra + e] = rec[8 + c];
      }
    }
  }
  if (active) {
    {
      const Cam cam = load_cam(viewmat, K);
      const float mx = means[3 * i], my = means[3 * i + 1], mz = means[3 * i + 2];
      const Activated a = load_activated(raw, i, quats, scales, opacities);
      const float s[3] = {a.s[0], a.s[1], a.s[2]};
      const Fwd f = project_core(cam, mx, my, mz, a.q[0], a.q[1], a.q[2], a.q[3], s[0], s[1], s[2], width, height,
                                 eps2d);
      // feature gradients: [colour | depth | extra] start at record slot 8
      const int ncol = kk > 0 ? 3 : fl.n_color;
      float v_depth = v_depths ? v_depths[i] : 0.f;
#pragma unroll
      for (int c = 0; c < FG_MAX_CHANNELS; ++c)
        if (fl.with_depth && c == ncol) v_depth += rec[8 + c];
      // ---- opacity / compensation -----------------------------------------------------------------
      float vcomp = 0.f;
      if (antialiased) {
        g_o = rec[2] * f.comp;
        vcomp = rec[2] * a.o;
      } else {
        g_o = rec[2];
      }
      // ---- K7 -------------------------------------------------------------------------------------
      const float vca = rec[3] + (v_conics ? v_conics[3 * i] : 0.f);
      const float vcb = rec[4] + (v_conics ? v_conics[3 * i + 1] : 0.f);
      const float vcc = rec[5] + (v_conics ? v_conics[3 * i + 2] : 0.f);
      project_backward(cam, f, s, eps2d, v_means2d[(size_t)m2_stride * i], v_means2d[(size_t)m2_stride * i + 1],
                       v_depth, vca, vcb, vcc,
                       antialiased != 0, vcomp, g_m, g_q, g_s);
    }
    if (raw.enabled) {
      // chain rule through the activations; the deltas receive the activated-value gradients as is.  The
      // activations are evaluated AGAIN here (index made opaque so that the two evaluations are not merged): kept
      // from before the projection backward their nine values would be live through it.
      int i2 = i;
      asm volatile("" : "+v"(i2));
      const Activated a = load_activated(raw, i2, quats, scales, opacities);
      if (v_d_quats) reinterpret_cast<float4*>(v_d_quats)[i] = make_float4(g_q[0], g_q[1], g_q[2], g_q[3]);
      if (v_d_scales) { v_d_scales[3 * i] = g_s[0]; v_d_scales[3 * i + 1] = g_s[1]; v_d_scales[3 * i + 2] = g_s[2]; }
      const float dp = g_q[0] * a.qn[0] + g_q[1] * a.qn[1] + g_q[2] * a.qn[2] + g_q[3] * a.qn[3];
#pragma unroll
      for (int c = 0; c < 4; ++c) g_q[c] = (g_q[c] - dp * a.qn[c]) * a.inv_norm;  // through q/|q|
#pragma unroll
      for (int c = 0; c < 3; ++c) g_s[c] *= a.es[c];                              // through exp
      g_o *= a.o * (1.f - a.o);                                                   // through sigmoid
    }
    // ---- K8 (behind K7: the means' gradient gets the SH colour's share added) ------------------------
    if (NOTE && kk > 0) view_direction();
    if (kk > 1) {
      float j[JAC];
      if (NOTE) {
        // the lane's own 40-byte note: five 8-byte loads (the lines are shared by neighbouring lanes)
        const float2* jp = reinterpret_cast<const float2*>(sh_jac + (size_t)i * JAC);
#pragma unroll
        for (int q2 = 0; q2 < JAC / 2; ++q2) {
          const float2 v = jp[q2];
          j[2 * q2] = v.x; j[2 * q2 + 1] = v.y;
        }
      } else {
#pragma unroll
        for (int c = 0; c < 9; ++c) j[c] = sums.jac[c];
        // the clamp mask: colour = max(sh + 0.5, 0)
        j[9] = __int_as_float((int)(sums.col[0] + 0.5f > 0.f) | ((int)(sums.col[1] + 0.5f > 0.f) << 1) |
                              ((int)(sums.col[2] + 0.5f > 0.f) << 2));
      }
      const int m = __float_as_int(j[9]);
      vr = (m & 1) ? rec[8] : 0.f;
      vg = (m & 2) ? rec[9] : 0.f;
      vb = (m & 4) ? rec[10] : 0.f;
      const float vdx = vr * j[0] + vg * j[3] + vb * j[6];
      const float vdy = vr * j[1] + vg * j[4] + vb * j[7];
      const float vdz = vr * j[2] + vg * j[5] + vb * j[8];
      const float dp = vdx * sh_dx + vdy * sh_dy + vdz * sh_dz;
      g_m[0] += (vdx - dp * sh_dx) * sh_inv;
      g_m[1] += (vdy - dp * sh_dy) * sh_inv;
      g_m[2] += (vdz - dp * sh_dz) * sh_inv;
    } else if (kk == 1) {
      const float* c0p = colors + (size_t)i * 3 * (raw.enabled ? 1 : fl.k_stored);
      vr = (C0 * c0p[0] + 0.5f > 0.f) ? rec[8] : 0.f;
      vg = (C0 * c0p[1] + 0.5f > 0.f) ? rec[9] : 0.f;
      vb = (C0 * c0p[2] + 0.5f > 0.f) ? rec[10] : 0.f;
    }
  } else if (raw.enabled && i < N) {
    if (v_d_quats) reinterpret_cast<float4*>(v_d_quats)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (v_d_scales) { v_d_scales[3 * i] = 0.f; v_d_scales[3 * i + 1] = 0.f; v_d_scales[3 * i + 2] = 0.f; }
  }
  if (i < N) {
    v_means[3 * i] = g_m[0]; v_means[3 * i + 1] = g_m[1]; v_means[3 * i + 2] = g_m[2];
    reinterpret_cast<float4*>(v_quats)[i] = make_float4(g_q[0], g_q[1], g_q[2], g_q[3]);
    v_scales[3 * i] = g_s[0]; v_scales[3 * i + 1] = g_s[1]; v_scales[3 * i + 2] = g_s[2];
    v_opacities[i] = g_o;
  }
  // factored SH gradient (view-DP exchange): the masked colour gradient itself, 12 B instead of the
  // 192-B coefficient row it expands to (v_coeffs[k] = basis_k(dir) * v_rgb, rebuilt after the
  // exchange by fg_sh_grad_accumulate)
  if (v_rgb && i < N) {
    float* o = v_rgb + (size_t)v_rgb_floats * i;
    o[0] = vr; o[1] = vg; o[2] = vb;
    if (v_rgb_floats == 6) {  // per-view (deformed) means: the direction travels with the gradient
      o[3] = sh_dx; o[4] = sh_dy; o[5] = sh_dz;
    }
  }
  // ---- v_coeffs rows out through LDS, half the workgroup's rows at a time ------------------------------
  if (kk > 0 && v_colors) {
    float basis[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) basis[k] = 0.f;
    if (active) sh_basis(fl.sh_degree, sh_dx, sh_dy, sh_dz, basis);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r0 = row0 + h * HROWS, nr = min(HROWS, nrows - h * HROWS);
      if (nr <= 0) break;
      __syncthreads();  // everyone is done with the LDS rows (coefficient halves / the previous output half)
      if ((int)(threadIdx.x / HROWS) == h) {
        float* row = lds + (threadIdx.x & (HROWS - 1)) * ROW;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const float bk = (k < kk) ? basis[k] : 0.f;
          row[3 * k] = bk * vr; row[3 * k + 1] = bk * vg; row[3 * k + 2] = bk * vb;
        }
      }
      __syncthreads();
      if (raw.enabled) {
        lds_to_slab_at<ROW, true, HROWS>(v_colors + (size_t)r0 * 3, lds, 0, nr, 3, 3);  // v_features_dc
        if (fl.k_stored > 1)
          lds_to_slab_at<ROW, true, HROWS>(v_features_rest + (size_t)r0 * 3 * (fl.k_stored - 1), lds, 3, nr,
                                           3 * (fl.k_stored - 1), 45);
      } else {
        lds_to_slab_at<ROW, true, HROWS>(v_colors + (size_t)r0 * 3 * fl.k_stored, lds, 0, nr, 3 * fl.k_stored, 48);
      }
    }
  }
}

// ---- camera-pose gradient ---------------------------------------------------------------------------------------------
// dL/d viewmat of one view, from the cotangents the per-Gaussian backward consumes (the record gradients of the raster
// backward, + gradients on means2d / depths / conics).  The reference keeps a CameraOptimizer in the loop
// (freegaussian_model.py:120 -- "off" in every shipped config --, applied at :774); gsplat returns v_viewmats from its
// projection backward and lets autograd carry the SH colour's share through `dirs = means - inverse(viewmats)[:3, 3]`.
// Three paths from the pose W | t to the loss:  p = W m + t (camera-space mean),  CC = W C W^T (camera-space covariance),
// and the camera position -W^-1 t inside the SH view direction -- the last leaves here as dL/d campos (3 floats) and the
// host applies the inverse's derivative (a 4 x 4 matter).  A pass of its own, taken only when the pose requires a gradient:
// the hot per-Gaussian backward stays as it is.  Deterministic: per-workgroup partial sums, summed in a fixed order.
constexpr int VM_BLOCK = 256, VM_GRID_MAX = 1024, VM_VALUES = 15;  // dL/dW (9), dL/dt (3), dL/d campos (3)
__global__ void __launch_bounds__(VM_BLOCK)
viewmat_bwd_kernel(int N, FeatLayout fl, RawForm raw, const float* __restrict__ means, const float* __restrict__ quats,
                   const float* __restrict__ scales, const float* __restrict__ opacities, const float* __restrict__ colors,
                   const float* __restrict__ viewmat, const float* __restrict__ K, int width, int height, float eps2d,
                   int antialiased, const int32_t* __restrict__ radii, const float* __restrict__ v_splats,
                   const float* __restrict__ v_means2d, int m2_stride, const float* __restrict__ v_depths,
                   const float* __restrict__ v_conics, const float* __restrict__ sh_jac, float* __restrict__ partials) {
  __shared__ float red[VM_BLOCK / 64][16];
  float acc[VM_VALUES];
#pragma unroll
  for (int k = 0; k < VM_VALUES; ++k) acc[k] = 0.f;
  const Cam cam = load_cam(viewmat, K);
  const int kk = fl.sh_degree >= 0 ? (fl.sh_degree + 1) * (fl.sh_degree + 1) : 0;
  const int ncol = kk > 0 ? 3 : fl.n_color;
  for (int i = blockIdx.x * VM_BLOCK + threadIdx.x; i < N; i += gridDim.x * VM_BLOCK) {
    if (radii[i] <= 0) continue;
    const float m[3] = {means[3 * i], means[3 * i + 1], means[3 * i + 2]};
    const Activated a = load_activated(raw, i, quats, scales, opacities);
    const Fwd f = project_core(cam, m[0], m[1], m[2], a.q[0], a.q[1], a.q[2], a.q[3], a.s[0], a.s[1], a.s[2], width, height,
                               eps2d);
    float rec[REC];
    const float4* rp = reinterpret_cast<const float4*>(v_splats) + (size_t)i * (REC / 4);
#pragma unroll
    for (int q4 = 0; q4 < REC / 4; ++q4) {
      const float4 v = rp[q4];
      rec[4 * q4] = v.x; rec[4 * q4 + 1] = v.y; rec[4 * q4 + 2] = v.z; rec[4 * q4 + 3] = v.w;
    }
    float v_depth = v_depths ? v_depths[i] : 0.f;
#pragma unroll
    for (int c = 0; c < FG_MAX_CHANNELS; ++c)
      if (fl.with_depth && c == ncol) v_depth += rec[8 + c];
    const float vcomp = antialiased ? rec[2] * a.o : 0.f;
    const float vca = rec[3] + (v_conics ? v_conics[3 * i] : 0.f);
    const float vcb = rec[4] + (v_conics ? v_conics[3 * i + 1] : 0.f);
    const float vcc = rec[5] + (v_conics ? v_conics[3 * i + 2] : 0.f);
    float vp[3], vCC[3][3];
    project_backward_camera(cam, f, eps2d, v_means2d[(size_t)m2_stride * i], v_means2d[(size_t)m2_stride * i + 1], v_depth,
                            vca, vcb, vcc, antialiased != 0, vcomp, vp, vCC);
    // p = W m + t
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[3 * r + c] += vp[r] * m[c];
      acc[9 + r] += vp[r];
    }
    // CC = W C W^T with C = M M^T:  dL/dW = (vCC + vCC^T) W C = 2 vCC (W C)
    {
      float C[3][3], T[3][3];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) C[r][c] = f.M[r][0] * f.M[c][0] + f.M[r][1] * f.M[c][1] + f.M[r][2] * f.M[c][2];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) T[r][c] = cam.W[r][0] * C[0][c] + cam.W[r][1] * C[1][c] + cam.W[r][2] * C[2][c];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[3 * r + c] += 2.f * (vCC[r][0] * T[0][c] + vCC[r][1] * T[1][c] + vCC[r][2] * T[2][c]);
    }
    // SH colour: direction = normalize(mean - campos); d colour / d direction from the forward's note, or from the row
    if (kk > 1) {
      float dx, dy, dz, inv;
      view_dir(viewmat, m[0], m[1], m[2], dx, dy, dz, inv);
      float j[JAC];
      if (sh_jac) {
#pragma unroll
        for (int c = 0; c < JAC; ++c) j[c] = sh_jac[(size_t)i * JAC + c];
      } else {
        float row[48];
#pragma unroll
        for (int c = 0; c < 48; ++c) row[c] = 0.f;
        if (raw.enabled) {
#pragma unroll
          for (int c = 0; c < 3; ++c) row[c] = colors[(size_t)i * 3 + c];
#pragma unroll
          for (int c = 3; c < 48; ++c)
            if (c < 3 * kk) row[c] = raw.features_rest[(size_t)i * 3 * (fl.k_stored - 1) + (c - 3)];
        } else {
#pragma unroll
          for (int c = 0; c < 48; ++c)
            if (c < 3 * kk) row[c] = colors[(size_t)i * 3 * fl.k_stored + c];
        }
        ShSums sums;
        sh_row_sums(sums, row, fl.sh_degree, kk, dx, dy, dz, true);
#pragma unroll
        for (int c = 0; c < 9; ++c) j[c] = sums.jac[c];
        j[9] = __int_as_float((int)(sums.col[0] + 0.5f > 0.f) | ((int)(sums.col[1] + 0.5f > 0.f) << 1) |
                              ((int)(sums.col[2] + 0.5f > 0.f) << 2));
      }
      const int mk = __float_as_int(j[9]);
      const float vr = (mk & 1) ? rec[8] : 0.f, vg = (mk & 2) ? rec[9] : 0.f, vb = (mk & 4) ? rec[10] : 0.f;
      const float vdx = vr * j[0] + vg * j[3] + vb * j[6];
      const float vdy = vr * j[1] + vg * j[4] + vb * j[7];
      const float vdz = vr * j[2] + vg * j[5] + vb * j[8];
      const float dp = vdx * dx + vdy * dy + vdz * dz;
      // d/d(mean - campos), and campos enters with a minus sign
      acc[12] -= (vdx - dp * dx) * inv;
      acc[13] -= (vdy - dp * dy) * inv;
      acc[14] -= (vdz - dp * dz) * inv;
    }
  }
  const int lane = fg::lane_id(), wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < VM_VALUES; ++k) {
    float v = acc[k];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) v += __shfl_xor(v, d);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 16) {
    float v = 0.f;
    if (threadIdx.x < VM_VALUES)
      for (int w = 0; w < VM_BLOCK / 64; ++w) v += red[w][threadIdx.x];
    partials[(size_t)blockIdx.x * 16 + threadIdx.x] = v;
  }
}
// out[0 .. 15] = dL/d viewmat (row-major 4 x 4, last row 0), out[16 .. 18] = dL/d campos
__global__ void __launch_bounds__(256) viewmat_sum_kernel(const float* __restrict__ partials, int n_blocks, float* __restrict__ out) {
  __shared__ float part[16][16];
  const int k = threadIdx.x & 15, s = threadIdx.x >> 4;
  float v = 0.f;
  for (int b = s; b < n_blocks; b += 16) v += partials[(size_t)b * 16 + k];
  part[s][k] = v;
  __syncthreads();
  if (threadIdx.x < 16) {
    float t = 0.f;
    for (int q = 0; q < 16; ++q) t += part[q][threadIdx.x];
    part[0][threadIdx.x] = t;
  }
  __syncthreads();
  if (threadIdx.x < 19) {
    float o = 0.f;
    if (threadIdx.x < 12) {
      const int r = threadIdx.x >> 2, c = threadIdx.x & 3;
      o = c < 3 ? part[0][3 * r + c] : part[0][9 + r];
    } else if (threadIdx.x >= 16) {
      o = part[0][12 + (threadIdx.x - 16)];
    }
    out[threadIdx.x] = o;
  }
}

// -DFG_PREPROCESS_SKIP_CULLED=1 (a build-time A/B switch; the library reads no environment): fetch the
// coefficient rows of visible Gaussians only, after the projection / the radii are known.  OFF: measured on
// MI355X (1M / 1080p, 16% culled, profiles/r02_preprocess_skip_culled.md) it saves the 30 MB of dead rows and
// LOSES more than that by putting the slab fetch behind the cull instead of under it: forward 0.072 -> 0.099
// ms, backward 0.109 -> 0.134 ms.
#ifndef FG_PREPROCESS_SKIP_CULLED
#define FG_PREPROCESS_SKIP_CULLED 0
#endif
int skip_culled_rows() { return FG_PREPROCESS_SKIP_CULLED; }

bool layout_ok(const FeatLayout& fl) {
  if (fl.sh_degree > 3 || fl.n_extra < 0 || fl.n_color < 0) return false;
  if (fl.sh_degree >= 0 && (fl.k_stored < (fl.sh_degree + 1) * (fl.sh_degree + 1) || fl.k_stored > 16)) return false;
  const int ncol = fl.sh_degree >= 0 ? 3 : fl.n_color;
  const int total = ncol + (fl.with_depth ? 1 : 0) + fl.n_extra;
  return total >= 1 && total <= FG_MAX_CHANNELS;
}

}  // namespace

namespace {

int launch_preprocess_fwd(int N, RawForm raw, const float* means, const float* quats, const float* scales,
                          const float* opacities, const float* colors, int sh_degree, int k_stored, int n_color,
                          int with_depth, const float* extra, int n_extra, const float* viewmat, const float* K,
                          int width, int height, float eps2d, float near_plane, float far_plane, float radius_clip,
                          int tile_size, int antialiased, int32_t* radii, float* means2d, float* depths,
                          float* conics, float* compensations, int32_t* tiles_touched, float* splats,
                          uint32_t* depth_keys, int32_t* tile_rects, uint64_t* tile_masks, float* sh_jac,
                          fg_stream_t stream) {
  FeatLayout fl{sh_degree, k_stored, sh_degree >= 0 ? 3 : n_color, with_depth ? 1 : 0, n_extra};
  if (N < 0 || width <= 0 || height <= 0 || tile_size <= 0 || !layout_ok(fl)) return FG_ERR_INVALID_ARG;
  if (tile_masks && !tile_rects) return FG_ERR_INVALID_ARG;  // (the masks are relative to the rectangles)
  if (N == 0) return FG_OK;
  if (!means || !quats || !scales || !opacities || !viewmat || !K || !radii || !means2d || !depths || !conics ||
      !tiles_touched || !splats)
    return FG_ERR_INVALID_ARG;
  if ((fl.n_color > 0 && !colors) || (n_extra > 0 && !extra)) return FG_ERR_INVALID_ARG;
  if (raw.enabled && (sh_degree < 0 || (k_stored > 1 && !raw.features_rest))) return FG_ERR_INVALID_ARG;
  const int tile_w = (width + tile_size - 1) / tile_size, tile_h = (height + tile_size - 1) / tile_size;
  hipLaunchKernelGGL(preprocess_fwd_kernel<false>, dim3((N + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, fg_hip_stream(stream), N,
                     fl, raw, means, quats, scales, opacities, colors, extra, viewmat, K, width, height, eps2d,
                     near_plane, far_plane, radius_clip, tile_size, tile_w, tile_h, antialiased, radii, means2d,
                     depths, conics, compensations, tiles_touched, splats, depth_keys,
                     reinterpret_cast<int2*>(tile_rects), reinterpret_cast<unsigned long long*>(tile_masks), sh_jac,
                     skip_culled_rows());
  FG_RETURN_IF_LAUNCH_FAILED();
  return FG_OK;
}

int launch_preprocess_bwd(int N, RawForm raw, float* v_d_quats, float* v_d_scales, float* v_features_rest,
                          float* v_rgb, int v_rgb_floats,
                          const float* means, const float* quats, const float* scales, const float* opacities,
                          const float* colors, int sh_degree, int k_stored, int n_color, int with_depth, int n_extra,
                          const float* viewmat, const float* K, int width, int height, float eps2d, int antialiased,
                          const int32_t* radii, const float* v_splats, const float* v_means2d, int v_means2d_stride,
                          const float* v_depths, const float* v_conics, float* v_means, float* v_quats,
                          float* v_scales, float* v_opacities, float* v_colors, float* v_extra,
                          const float* sh_jac, fg_stream_t stream) {
  FeatLayout fl{sh_degree, k_stored, sh_degree >= 0 ? 3 : n_color, with_depth ? 1 : 0, n_extra};
  if (N < 0 || width <= 0 || height <= 0 || !layout_ok(fl) || v_means2d_stride < 2) return FG_ERR_INVALID_ARG;
  if (N == 0) return FG_OK;
  if (!means || !quats || !scales || !opacities || !viewmat || !K || !radii || !v_splats || !v_means2d ||
      !v_means || !v_quats || !v_scales || !v_opacities)
    return FG_ERR_INVALID_ARG;
  // v_colors may be null only in the factored form (SH colours, v_rgb given instead)
  if ((fl.n_color > 0 && (!colors || (!v_colors && !(v_rgb && sh_degree >= 0)))) || (n_extra > 0 && !v_extra))
    return FG_ERR_INVALID_ARG;
  if (v_rgb && sh_degree < 0) return FG_ERR_UNSUPPORTED;
  if (v_rgb && v_rgb_floats != 3 && v_rgb_floats != 6) return FG_ERR_INVALID_ARG;
  if (raw.enabled) {
    // (factored form: neither coefficient gradient is written -- v_colors and v_features_rest are both null)
    if (sh_degree < 0 || (k_stored > 1 && (!raw.features_rest || (!v_features_rest && !v_rgb)))) return FG_ERR_INVALID_ARG;
    if (v_rgb && (v_colors || v_features_rest)) return FG_ERR_INVALID_ARG;
    if ((raw.d_quats != nullptr) != (v_d_quats != nullptr) || (raw.d_scales != nullptr) != (v_d_scales != nullptr))
      return FG_ERR_INVALID_ARG;
  }
  const bool note = sh_jac != nullptr && sh_degree >= 1;
  hipLaunchKernelGGL(note ? preprocess_bwd_kernel<true> : preprocess_bwd_kernel<false>, dim3((N + BLOCK - 1) / BLOCK),
                     dim3(BLOCK), 0, fg_hip_stream(stream), N, fl, raw, v_d_quats, v_d_scales, v_features_rest, v_rgb, v_rgb_floats, means, quats, scales, opacities, colors,
                     viewmat, K, width, height, eps2d, antialiased, radii, v_splats, v_means2d, v_means2d_stride,
                     v_depths, v_conics, v_means, v_quats, v_scales, v_opacities, v_colors, v_extra, sh_jac,
                     skip_culled_rows());
  FG_RETURN_IF_LAUNCH_FAILED();
  return FG_OK;
}

}  // namespace

extern "C" int fg_preprocess_fwd(int N, const float* means, const float* quats, const float* scales,
                                 const float* opacities, const float* colors, int sh_degree, int k_stored,
                                 int n_color, int with_depth, const float* extra, int n_extra,
                                 const float* viewmat, const float* K, int width, int height, float eps2d,
                                 float near_plane, float far_plane, float radius_clip, int tile_size,
                                 int antialiased, int32_t* radii, float* means2d, float* depths, float* conics,
                                 float* compensations, int32_t* tiles_touched, float* splats,
                                 uint32_t* depth_keys, int32_t* tile_rects, uint64_t* tile_masks, float* sh_jac,
                                 fg_stream_t stream) {
  return launch_preprocess_fwd(N, RawForm{0, nullptr, nullptr, nullptr}, means, quats, scales, opacities, colors,
                               sh_degree, k_stored, n_color, with_depth, extra, n_extra, viewmat, K, width, height,
                               eps2d, near_plane, far_plane, radius_clip, tile_size, antialiased, radii, means2d,
                               depths, conics, compensations, tiles_touched, splats, depth_keys, tile_rects, tile_masks,
                               sh_jac, stream);
}

extern "C" int fg_preprocess_bwd(int N, const float* means, const float* quats, const float* scales,
                                 const float* opacities, const float* colors, int sh_degree, int k_stored,
                                 int n_color, int with_depth, int n_extra, const float* viewmat, const float* K,
                                 int width, int height, float eps2d, int antialiased, const int32_t* radii,
                                 const float* v_splats, const float* v_means2d, int v_means2d_stride,
                                 const float* v_depths, const float* v_conics, float* v_means, float* v_quats,
                                 float* v_scales, float* v_opacities, float* v_colors, float* v_extra,
                                 const float* sh_jac, fg_stream_t stream) {
  return launch_preprocess_bwd(N, RawForm{0, nullptr, nullptr, nullptr}, nullptr, nullptr, nullptr, nullptr, 3, means, quats,
                               scales, opacities, colors, sh_degree, k_stored, n_color, with_depth, n_extra, viewmat,
                               K, width, height, eps2d, antialiased, radii, v_splats, v_means2d, v_means2d_stride,
                               v_depths, v_conics, v_means, v_quats, v_scales, v_opacities, v_colors, v_extra,
                               sh_jac, stream);
}

extern "C" int fg_preprocess_raw_fwd(int N, const float* means, const float* quats, const float* d_quats,
                                     const float* log_scales, const float* d_scales,
                                     const float* opacity_logits, const float* features_dc,
                                     const float* features_rest, int sh_degree, int k_stored, int with_depth,
                                     const float* extra, int n_extra, const float* viewmat, const float* K,
                                     int width, int height, float eps2d, float near_plane, float far_plane,
                                     float radius_clip, int tile_size, int antialiased, int32_t* radii,
                                     float* means2d, float* depths, float* conics, float* compensations,
                                     int32_t* tiles_touched, float* splats, uint32_t* depth_keys,
                                     int32_t* tile_rects, uint64_t* tile_masks, float* sh_jac, fg_stream_t stream) {
  return launch_preprocess_fwd(N, RawForm{1, d_quats, d_scales, features_rest}, means, quats, log_scales,
                               opacity_logits, features_dc, sh_degree, k_stored, 3, with_depth, extra, n_extra,
                               viewmat, K, width, height, eps2d, near_plane, far_plane, radius_clip, tile_size,
                               antialiased, radii, means2d, depths, conics, compensations, tiles_touched, splats,
                               depth_keys, tile_rects, tile_masks, sh_jac, stream);
}

extern "C" int fg_preprocess_raw_bwd(int N, const float* means, const float* quats, const float* d_quats,
                                     const float* log_scales, const float* d_scales,
                                     const float* opacity_logits, const float* features_dc,
                                     const float* features_rest, int sh_degree, int k_stored, int with_depth,
                                     int n_extra, const float* viewmat, const float* K, int width, int height,
                                     float eps2d, int antialiased, const int32_t* radii, const float* v_splats,
                                     const float* v_means2d, int v_means2d_stride, const float* v_depths,
                                     const float* v_conics, float* v_means, float* v_quats, float* v_d_quats,
                                     float* v_log_scales, float* v_d_scales, float* v_opacity_logits,
                                     float* v_features_dc, float* v_features_rest, float* v_extra,
                                     const float* sh_jac, fg_stream_t stream) {
  return launch_preprocess_bwd(N, RawForm{1, d_quats, d_scales, features_rest}, v_d_quats, v_d_scales,
                               v_features_rest, nullptr, 3, means, quats, log_scales, opacity_logits, features_dc, sh_degree,
                               k_stored, 3, with_depth, n_extra, viewmat, K, width, height, eps2d, antialiased, radii,
                               v_splats, v_means2d, v_means2d_stride, v_depths, v_conics, v_means, v_quats,
                               v_log_scales, v_opacity_logits, v_features_dc, v_extra, sh_jac, stream);
}

extern "C" int fg_preprocess_raw_bwd_factored(
    int N, const float* means, const float* quats, const float* d_quats, const float* log_scales, const float* d_scales,
    const float* opacity_logits, const float* features_dc, const float* features_rest, int sh_degree, int k_stored,
    int with_depth, int n_extra, const float* viewmat, const float* K, int width, int height, float eps2d, int antialiased,
    const int32_t* radii, const float* v_splats, const float* v_means2d, int v_means2d_stride, const float* v_depths,
    const float* v_conics, float* v_means, float* v_quats, float* v_d_quats, float* v_log_scales, float* v_d_scales,
    float* v_opacity_logits, float* v_rgb, int v_rgb_floats, float* v_extra, const float* sh_jac, fg_stream_t stream) {
  if (!v_rgb) return FG_ERR_INVALID_ARG;
  return launch_preprocess_bwd(N, RawForm{1, d_quats, d_scales, features_rest}, v_d_quats, v_d_scales, nullptr, v_rgb,
                               v_rgb_floats, means, quats, log_scales, opacity_logits, features_dc, sh_degree, k_stored, 3,
                               with_depth, n_extra, viewmat, K, width, height, eps2d, antialiased, radii, v_splats,
                               v_means2d, v_means2d_stride, v_depths, v_conics, v_means, v_quats, v_log_scales,
                               v_opacity_logits, nullptr, v_extra, sh_jac, stream);
}

extern "C" int fg_preprocess_bwd_factored(int N, const float* means, const float* quats, const float* scales,
                                          const float* opacities, const float* colors, int sh_degree, int k_stored,
                                          int with_depth, int n_extra, const float* viewmat, const float* K,
                                          int width, int height, float eps2d, int antialiased, const int32_t* radii,
                                          const float* v_splats, const float* v_means2d, int v_means2d_stride,
                                          const float* v_depths, const float* v_conics, float* v_means,
                                          float* v_quats, float* v_scales, float* v_opacities, float* v_rgb,
                                          int v_rgb_floats, float* v_extra, const float* sh_jac, fg_stream_t stream) {
  if (!v_rgb) return FG_ERR_INVALID_ARG;
  return launch_preprocess_bwd(N, RawForm{0, nullptr, nullptr, nullptr}, nullptr, nullptr, nullptr, v_rgb, v_rgb_floats, means, quats,
                               scales, opacities, colors, sh_degree, k_stored, 3, with_depth, n_extra, viewmat, K,
                               width, height, eps2d, antialiased, radii, v_splats, v_means2d, v_means2d_stride,
                               v_depths, v_conics, v_means, v_quats, v_scales, v_opacities, nullptr, v_extra, sh_jac,
                               stream);
}

extern "C" int fg_sh_pack_fwd(int N, const float* means, const float* opacities, const float* colors, int sh_degree,
                              int k_stored, int n_color, int with_depth, const float* extra, int n_extra,
                              const float* viewmat, int antialiased, const int32_t* radii, const float* means2d,
                              const float* depths, const float* conics, const float* compensations, float* splats,
                              fg_stream_t stream) {
  FeatLayout fl{sh_degree, k_stored, sh_degree >= 0 ? 3 : n_color, with_depth ? 1 : 0, n_extra};
  if (N < 0 || !layout_ok(fl)) return FG_ERR_INVALID_ARG;
  if (N == 0) return FG_OK;
  if (!means || !opacities || !viewmat || !radii || !means2d || !depths || !conics || !splats) return FG_ERR_INVALID_ARG;
  if ((fl.n_color > 0 && !colors) || (n_extra > 0 && !extra) || (antialiased && !compensations)) return FG_ERR_INVALID_ARG;
  hipLaunchKernelGGL(preprocess_fwd_kernel<true>, dim3((N + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, fg_hip_stream(stream),
                     N, fl, RawForm{0, nullptr, nullptr, nullptr}, means, nullptr, nullptr, opacities, colors, extra,
                     viewmat, nullptr, 0, 0, 0.f, 0.f, 0.f, 0.f, 16, 0, 0, antialiased, const_cast<int32_t*>(radii),
                     const_cast<float*>(means2d), const_cast<float*>(depths), const_cast<float*>(conics),
                     const_cast<float*>(compensations), nullptr, splats, nullptr, nullptr, nullptr, nullptr, 0);
  FG_RETURN_IF_LAUNCH_FAILED();
  return FG_OK;
}

namespace {
int viewmat_grid(int N) {
  const int g = (N + VM_BLOCK - 1) / VM_BLOCK;
  return g < VM_GRID_MAX ? (g > 0 ? g : 1) : VM_GRID_MAX;
}
}  // namespace

extern "C" size_t fg_viewmat_bwd_workspace_bytes(int N) { return N > 0 ? (size_t)viewmat_grid(N) * 16 * sizeof(float) : 0; }

extern "C" int fg_viewmat_bwd(int N, int raw, const float* means, const float* quats, const float* d_quats,
                              const float* scales, const float* d_scales, const float* opacities, const float* colors,
                              const float* features_rest, int sh_degree, int k_stored, int n_color, int with_depth,
                              int n_extra, const float* viewmat, const float* K, int width, int height, float eps2d,
                              int antialiased, const int32_t* radii, const float* v_splats, const float* v_means2d,
                              int v_means2d_stride, const float* v_depths, const float* v_conics, const float* sh_jac,
                              float* out, void* workspace, size_t workspace_bytes, fg_stream_t stream) {
  FeatLayout fl{sh_degree, k_stored, sh_degree >= 0 ? 3 : n_color, with_depth ? 1 : 0, n_extra};
  if (N < 0 || width <= 0 || height <= 0 || !layout_ok(fl) || v_means2d_stride < 2 || !out) return FG_ERR_INVALID_ARG;
  hipStream_t s = fg_hip_stream(stream);
  if (N == 0) {
    if (hipMemsetAsync(out, 0, 19 * sizeof(float), s) != hipSuccess) return FG_ERR_LAUNCH;
    return FG_OK;
  }
  if (!means || !quats || !scales || !opacities || !viewmat || !K || !radii || !v_splats || !v_means2d || !workspace)
    return FG_ERR_INVALID_ARG;
  if (raw && (sh_degree < 0 || (k_stored > 1 && !features_rest))) return FG_ERR_INVALID_ARG;
  if (sh_degree >= 1 && !sh_jac && !colors) return FG_ERR_INVALID_ARG;
  if (workspace_bytes < fg_viewmat_bwd_workspace_bytes(N)) return FG_ERR_WORKSPACE;
  const int grid = viewmat_grid(N);
  hipLaunchKernelGGL(viewmat_bwd_kernel, dim3(grid), dim3(VM_BLOCK), 0, s, N, fl,
                     RawForm{raw ? 1 : 0, raw ? d_quats : nullptr, raw ? d_scales : nullptr, raw ? features_rest : nullptr},
                     means, quats, scales, opacities, colors, viewmat, K, width, height, eps2d, antialiased, radii, v_splats,
                     v_means2d, v_means2d_stride, v_depths, v_conics, sh_degree >= 1 ? sh_jac : nullptr,
                     static_cast<float*>(workspace));
  hipLaunchKernelGGL(viewmat_sum_kernel, dim3(1), dim3(256), 0, s, static_cast<const float*>(workspace), grid, out);
  FG_RETURN_IF_LAUNCH_FAILED();
  return FG_OK;
}
