// K1 / K7: EWA projection of 3D Gaussians to screen space, forward and backward.
//
// One lane per Gaussian; the work is ~150 flops on 44 input bytes, HBM-bound.  This file is
// compiled with -ffp-contract=off and uses only IEEE-exact operations (+ - * / sqrt) in the
// forward, in a fixed order, so that radii / means2d / depths -- and therefore the tile ids
// and 64-bit sort keys derived from them -- are reproducible bit for bit by a CPU restatement
// written in the same order (see DESIGN.md "bit-exact integer path").
//
// Replaces the projection stage implied by the rasterization(...) kwargs at
// /root/reference freegaussian/freegaussian_model.py:847-865 (viewmats, Ks, near_plane=0.01,
// far_plane=1e10, rasterize_mode, 0.3 px blur :110-119).
#include "project_math.h"

namespace {
using namespace fgp;

__global__ void __launch_bounds__(256)
project_fwd_kernel(int N, const float* __restrict__ means, const float* __restrict__ quats,
                   const float* __restrict__ scales, const float* __restrict__ viewmat,
                   const float* __restrict__ K, int width, int height, float eps2d, float near_plane,
                   float far_plane, float radius_clip, int tile_size, int tile_w, int tile_h,
                   int32_t* __restrict__ radii, float* __restrict__ means2d, float* __restrict__ depths,
                   float* __restrict__ conics, float* __restrict__ compensations,
                   int32_t* __restrict__ tiles_touched) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const Cam cam = load_cam(viewmat, K);
  const float mx = means[3 * i], my = means[3 * i + 1], mz = means[3 * i + 2];
  const float4 q = reinterpret_cast<const float4*>(quats)[i];
  const float s0 = scales[3 * i], s1 = scales[3 * i + 1], s2 = scales[3 * i + 2];
  const Fwd f = project_core(cam, mx, my, mz, q.x, q.y, q.z, q.w, s0, s1, s2, width, height, eps2d);

  bool ok = (f.pz >= near_plane) && (f.pz <= far_plane) && (f.det > 0.0f);
  ok = ok && isfinite(f.radius_f) && (f.radius_f > radius_clip);
  const float fw = (float)width, fh = (float)height;
  ok = ok && !((f.m2x + f.radius_f <= 0.0f) || (f.m2x - f.radius_f >= fw) || (f.m2y + f.radius_f <= 0.0f) ||
               (f.m2y - f.radius_f >= fh));
  int32_t radius = 0, touched = 0;
  float o_x = 0.f, o_y = 0.f, o_d = 0.f, o_a = 0.f, o_b = 0.f, o_c = 0.f, o_comp = 0.f;
  if (ok) {
    radius = (int32_t)f.radius_f;
    o_x = f.m2x; o_y = f.m2y; o_d = f.pz;
    o_a = f.conic_a; o_b = f.conic_b; o_c = f.conic_c; o_comp = f.comp;
    const float ts = (float)tile_size;
    const float r = (float)radius / ts;
    const float tx = f.m2x / ts, ty = f.m2y / ts;
    const int x0 = min(max((int)floorf(tx - r), 0), tile_w), x1 = min(max((int)ceilf(tx + r), 0), tile_w);
    const int y0 = min(max((int)floorf(ty - r), 0), tile_h), y1 = min(max((int)ceilf(ty + r), 0), tile_h);
    touched = (x1 - x0) * (y1 - y0);
  }
  radii[i] = radius;
  means2d[2 * i] = o_x;
  means2d[2 * i + 1] = o_y;
  depths[i] = o_d;
  conics[3 * i] = o_a;
  conics[3 * i + 1] = o_b;
  conics[3 * i + 2] = o_c;
  if (compensations) compensations[i] = o_comp;
  tiles_touched[i] = touched;
}

__global__ void __launch_bounds__(256)
project_bwd_kernel(int N, const float* __restrict__ means, const float* __restrict__ quats,
                   const float* __restrict__ scales, const float* __restrict__ viewmat,
                   const float* __restrict__ K, int width, int height, float eps2d,
                   const int32_t* __restrict__ radii, const float* __restrict__ v_means2d,
                   const float* __restrict__ v_depths, const float* __restrict__ v_conics,
                   const float* __restrict__ v_compensations, float* __restrict__ v_means,
                   float* __restrict__ v_quats, float* __restrict__ v_scales) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float g_m[3] = {0.f, 0.f, 0.f}, g_q[4] = {0.f, 0.f, 0.f, 0.f}, g_s[3] = {0.f, 0.f, 0.f};
  if (radii[i] > 0) {
    const Cam cam = load_cam(viewmat, K);
    const float mx = means[3 * i], my = means[3 * i + 1], mz = means[3 * i + 2];
    const float4 q = reinterpret_cast<const float4*>(quats)[i];
    const float s[3] = {scales[3 * i], scales[3 * i + 1], scales[3 * i + 2]};
    const Fwd f = project_core(cam, mx, my, mz, q.x, q.y, q.z, q.w, s[0], s[1], s[2], width, height, eps2d);
    project_backward(cam, f, s, eps2d, v_means2d[2 * i], v_means2d[2 * i + 1], v_depths[i], v_conics[3 * i],
                     v_conics[3 * i + 1], v_conics[3 * i + 2], v_compensations != nullptr,
                     v_compensations ? v_compensations[i] : 0.f, g_m, g_q, g_s);
  }
  v_means[3 * i] = g_m[0]; v_means[3 * i + 1] = g_m[1]; v_means[3 * i + 2] = g_m[2];
  reinterpret_cast<float4*>(v_quats)[i] = make_float4(g_q[0], g_q[1], g_q[2], g_q[3]);
  v_scales[3 * i] = g_s[0]; v_scales[3 * i + 1] = g_s[1]; v_scales[3 * i + 2] = g_s[2];
}

}  // namespace

extern "C" int fg_project_fwd(int N, const float* means, const float* quats, const float* scales,
                              const float* viewmat, const float* K, int width, int height, float eps2d,
                              float near_plane, float far_plane, float radius_clip, int tile_size,
                              int32_t* radii, float* means2d, float* depths, float* conics,
                              float* compensations, int32_t* tiles_touched, fg_stream_t stream) {
  if (N < 0 || width <= 0 || height <= 0 || tile_size <= 0) return FG_ERR_INVALID_ARG;
  if (N == 0) return FG_OK;
  if (!means || !quats || !scales || !viewmat || !K || !radii || !means2d || !depths || !conics || !tiles_touched)
    return FG_ERR_INVALID_ARG;
  const int tile_w = (width + tile_size - 1) / tile_size, tile_h = (height + tile_size - 1) / tile_size;
  hipLaunchKernelGGL(project_fwd_kernel, dim3((N + 255) / 256), dim3(256), 0, fg_hip_stream(stream), N, means,
                     quats, scales, viewmat, K, width, height, eps2d, near_plane, far_plane, radius_clip,
                     tile_size, tile_w, tile_h, radii, means2d, depths, conics, compensations, tiles_touched);
  FG_RETURN_IF_LAUNCH_FAILED();
  return FG_OK;
}

extern "C" int fg_project_bwd(int N, const float* means, const float* quats, const float* scales,
                              const float* viewmat, const float* K, int width, int height, float eps2d,
                              const int32_t* radii, const float* conics, const float* compensations,
                              const float* v_means2d, const float* v_depths, const float* v_conics,
                              const float* v_compensations, float* v_means, float* v_quats,
                              float* v_scales, fg_stream_t stream) {
  (void)conics;
  (void)compensations;
  if (N < 0 || width <= 0 || height <= 0) return FG_ERR_INVALID_ARG;
  if (N == 0) return FG_OK;
  if (!means || !quats || !scales || !viewmat || !K || !radii || !v_means2d || !v_depths || !v_conics ||
      !v_means || !v_quats || !v_scales)
    return FG_ERR_INVALID_ARG;
  hipLaunchKernelGGL(project_bwd_kernel, dim3((N + 255) / 256), dim3(256), 0, fg_hip_stream(stream), N, means,
                     quats, scales, viewmat, K, width, height, eps2d, radii, v_means2d, v_depths, v_conics,
                     v_compensations, v_means, v_quats, v_scales);
  FG_RETURN_IF_LAUNCH_FAILED();
  return FG_OK;
}
