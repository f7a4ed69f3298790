// Adaptive density control in three passes: the reference's
// refinement_after / split_gaussians / dup_gaussians / cull_gaussians
// (/root/reference freegaussian/freegaussian_model.py:404-571) decide per Gaussian whether to
// split, duplicate or delete it, then rebuild EVERY parameter tensor and both Adam moment tensors
// with ~12 boolean-index copies and concatenations each.  Here:
//   fg_densify_flags   one pass over the statistics -> a flag byte per Gaussian
//   (four prefix sums of the flag bits: the caller's, any scan will do)
//   fg_densify_map     -> for every row of the NEW set: the old row it is copied from, and for split
//                         children the index of their random sample
//   fg_gather_rows     one coalesced copy per tensor straight into its final place (Adam moments:
//                         zeros for new rows), no intermediate concatenated copy
//   fg_split_children  in place on the new rows: children mean += R(q) (exp(s) * z), s -= log(1.6);
//                         duplicates of split parents only shrink
// Row order of the new set = the reference's: surviving old rows in order, then the children of
// split Gaussians sample-major ([all 1st children][all 2nd children]...), then duplicates.
#include "fg_common.h"

namespace {

constexpr uint8_t F_SPLIT = 1, F_DUP = 2, F_KEEP_OLD = 4, F_KEEP_CHILD = 8, F_KEEP_DUP = 16;

struct DensifyCfg {
  int do_densify;            // splits / dups are considered at all (:414-418)
  float max_dim;             // max(H, W) of the last render (:421)
  float grad_thresh;         // densify_grad_thresh
  float size_thresh;         // densify_size_thresh
  float split_screen_size;   // < 0: screen-size splitting off (step >= stop_screen_size_at)
  float cull_alpha_thresh;
  float cull_scale_thresh;   // < 0: no "too big" culling yet (step <= refine_every*reset_alpha_every)
  float cull_screen_size;    // < 0: off
};

__global__ void __launch_bounds__(256)
densify_flags_kernel(int N, DensifyCfg cfg, const float* __restrict__ grad_norm, const float* __restrict__ vis_counts,
                     const float* __restrict__ max_2dsize, const float* __restrict__ log_scales,
                     const float* __restrict__ opacity_logits, uint8_t* __restrict__ flags) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const float s0 = expf(log_scales[3 * i]), s1 = expf(log_scales[3 * i + 1]), s2 = expf(log_scales[3 * i + 2]);
  const float smax = fmaxf(s0, fmaxf(s1, s2));
  const float m2d = max_2dsize ? max_2dsize[i] : 0.f;
  // the scales a split Gaussian -- and its children -- carry afterwards: log(exp(s)/1.6) (:548-549)
  const float c0 = expf(logf(s0 / 1.6f)), c1 = expf(logf(s1 / 1.6f)), c2 = expf(logf(s2 / 1.6f));
  const float cmax = fmaxf(c0, fmaxf(c1, c2));
  bool split = false, dup = false;
  if (cfg.do_densify) {
    const float avg = ((grad_norm[i] / vis_counts[i]) * 0.5f) * cfg.max_dim;
    const bool high = avg > cfg.grad_thresh;
    split = (smax > cfg.size_thresh) && high;
    if (cfg.split_screen_size >= 0.f) split = split || (m2d > cfg.split_screen_size);
    // `dups` is evaluated AFTER split_gaussians shrank the split rows in place (:428-431): a split
    // Gaussian whose shrunk size falls under the threshold is duplicated as well (shrunk copy)
    dup = ((split ? cmax : smax) <= cfg.size_thresh) && high;
  }
  const float alpha = 1.f / (1.f + expf(-opacity_logits[i]));
  const bool transparent = alpha < cfg.cull_alpha_thresh;
  bool big_old = false, big_new = false;
  if (cfg.cull_scale_thresh >= 0.f) {
    big_old = smax > cfg.cull_scale_thresh;
    if (cfg.cull_screen_size >= 0.f) big_old = big_old || (m2d > cfg.cull_screen_size);
    // new rows carry a fresh zero max_2Dsize entry (:443-450): only their scale can be too big
    big_new = (split ? cmax : smax) > cfg.cull_scale_thresh;
  }
  uint8_t f = 0;
  if (split) f |= F_SPLIT;
  if (dup) f |= F_DUP;
  if (!split && !transparent && !big_old) f |= F_KEEP_OLD;  // split originals are pruned (:455-464)
  if (split && !transparent && !big_new) f |= F_KEEP_CHILD;
  if (dup && !transparent && !big_new) f |= F_KEEP_DUP;
  flags[i] = f;
}

// pos_*: EXCLUSIVE prefix sums over i of the corresponding flag bit.
__global__ void __launch_bounds__(256)
densify_map_kernel(int N, const uint8_t* __restrict__ flags, const int32_t* __restrict__ pos_old,
                   const int32_t* __restrict__ pos_child, const int32_t* __restrict__ pos_dup,
                   const int32_t* __restrict__ rank_split, int n_old, int n_child, int n_split_total, int nsamps,
                   int32_t* __restrict__ src_index, int32_t* __restrict__ sample_index) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const uint8_t f = flags[i];
  if (f & F_KEEP_OLD) {
    src_index[pos_old[i]] = i;
    sample_index[pos_old[i]] = -1;
  }
  if (f & F_KEEP_CHILD) {
    for (int s = 0; s < nsamps; ++s) {
      const int j = n_old + s * n_child + pos_child[i];
      src_index[j] = i;
      sample_index[j] = s * n_split_total + rank_split[i];  // row of randn((nsamps*n_splits, 3)) (:530)
    }
  }
  if (f & F_KEEP_DUP) {
    const int j = n_old + nsamps * n_child + pos_dup[i];
    src_index[j] = i;
    sample_index[j] = (f & F_SPLIT) ? -2 : -1;  // -2: copy of a split parent -> shrunk scales
  }
}

// dst[j, :] = src[src_index[j], :] (rows of D floats), or zeros for j >= zero_from.
__global__ void __launch_bounds__(256)
gather_rows_kernel(int64_t total, int D, const float* __restrict__ src, const int32_t* __restrict__ src_index,
                   int zero_from, float* __restrict__ dst) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int64_t j = e / D;
  const int c = (int)(e - j * D);
  dst[e] = (j >= zero_from) ? 0.f : src[(int64_t)src_index[j] * D + c];
}

__global__ void __launch_bounds__(256)
split_children_kernel(int first, int count, const int32_t* __restrict__ sample_index, const float* __restrict__ samples,
                      float* __restrict__ means, float* __restrict__ log_scales, const float* __restrict__ quats) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count) return;
  const int j = first + t;
  const int si = sample_index[j];
  if (si == -1) return;  // plain duplicate
  const float ls0 = log_scales[3 * j], ls1 = log_scales[3 * j + 1], ls2 = log_scales[3 * j + 2];
  const float e0 = expf(ls0), e1 = expf(ls1), e2 = expf(ls2);
  log_scales[3 * j] = logf(e0 / 1.6f);
  log_scales[3 * j + 1] = logf(e1 / 1.6f);
  log_scales[3 * j + 2] = logf(e2 / 1.6f);
  if (si < 0) return;  // duplicate of a split parent: shrunk, not moved
  const float z0 = e0 * samples[3 * si], z1 = e1 * samples[3 * si + 1], z2 = e2 * samples[3 * si + 2];
  float w = quats[4 * j], x = quats[4 * j + 1], y = quats[4 * j + 2], z = quats[4 * j + 3];
  const float inv = 1.f / sqrtf(w * w + x * x + y * y + z * z);
  w *= inv; x *= inv; y *= inv; z *= inv;
  // rotation matrix of a unit wxyz quaternion (the reference's quat_to_rotmat, :535)
  const float r00 = 1.f - 2.f * (y * y + z * z), r01 = 2.f * (x * y - w * z), r02 = 2.f * (x * z + w * y);
  const float r10 = 2.f * (x * y + w * z), r11 = 1.f - 2.f * (x * x + z * z), r12 = 2.f * (y * z - w * x);
  const float r20 = 2.f * (x * z - w * y), r21 = 2.f * (y * z + w * x), r22 = 1.f - 2.f * (x * x + y * y);
  means[3 * j] += r00 * z0 + r01 * z1 + r02 * z2;
  means[3 * j + 1] += r10 * z0 + r11 * z1 + r12 * z2;
  means[3 * j + 2] += r20 * z0 + r21 * z1 + r22 * z2;
}

}  // namespace

namespace {
// S1: the per-iteration densification statistics of after_train_iter (/root/reference freegaussian_model.py:369-392)
// in one pass over the Gaussians: visible = radii > 0; vis_counts += visible; xys_grad_norm += visible ? |absgrad| : 0;
// max_2Dsize = max(max_2Dsize, visible ? radii / max(W, H) : 0).  (As torch operators: nine launches, 65 us at any size.)
__global__ void __launch_bounds__(256)
densify_stats_kernel(int N, const float2* __restrict__ absgrad, const int32_t* __restrict__ radii, float max_dim,
                     float* __restrict__ xys_grad_norm, float* __restrict__ vis_counts, float* __restrict__ max_2dsize) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const int r = radii[i];
  if (r <= 0) return;  // (invisible rows stay bit for bit as they were)
  const float2 a = absgrad[i];
  xys_grad_norm[i] += sqrtf(a.x * a.x + a.y * a.y);
  vis_counts[i] += 1.f;
  max_2dsize[i] = fmaxf(max_2dsize[i], (float)r / max_dim);
}
}  // namespace

extern "C" int fg_densify_stats(int N, const float* absgrad, const int32_t* radii, float max_dim, float* xys_grad_norm,
                                float* vis_counts, float* max_2dsize, fg_stream_t stream) {
  if (N < 0 || !(max_dim > 0.f)) return FG_ERR_INVALID_ARG;
  if (N == 0) return FG_OK;
  if (!absgrad || !radii || !xys_grad_norm || !vis_counts || !max_2dsize) return FG_ERR_INVALID_ARG;
  hipLaunchKernelGGL(densify_stats_kernel, dim3((N + 255) / 256), dim3(256), 0, fg_hip_stream(stream), N,
                     reinterpret_cast<const float2*>(absgrad), radii, max_dim, xys_grad_norm, vis_counts, max_2dsize);
  FG_RETURN_IF_LAUNCH_FAILED();
  return FG_OK;
}

extern "C" int fg_densify_flags(int N, int do_densify, float max_dim, float densify_grad_thresh,
                                float densify_size_thresh, float split_screen_size, float cull_alpha_thresh,
                                float cull_scale_thresh, float cull_screen_size, const float* xys_grad_norm,
                                const float* vis_counts, const float* max_2dsize, const float* log_scales,
                                const float* opacity_logits, uint8_t* flags, fg_stream_t stream) {
  if (N < 0) return FG_ERR_INVALID_ARG;
  if (N == 0) return FG_OK;
  if (!log_scales || !opacity_logits || !flags) return FG_ERR_INVALID_ARG;
  if (do_densify && (!xys_grad_norm || !vis_counts)) return FG_ERR_INVALID_ARG;
  const DensifyCfg cfg{do_densify, max_dim, densify_grad_thresh, densify_size_thresh, split_screen_size,
                       cull_alpha_thresh, cull_scale_thresh, cull_screen_size};
  hipLaunchKernelGGL(densify_flags_kernel, dim3((N + 255) / 256), dim3(256), 0, fg_hip_stream(stream), N, cfg,
                     xys_grad_norm, vis_counts, max_2dsize, log_scales, opacity_logits, flags);
  FG_RETURN_IF_LAUNCH_FAILED();
  return FG_OK;
}

extern "C" int fg_densify_map(int N, const uint8_t* flags, const int32_t* pos_old, const int32_t* pos_child,
                              const int32_t* pos_dup, const int32_t* rank_split, int n_old, int n_child,
                              int n_split_total, int n_split_samples, int32_t* src_index, int32_t* sample_index,
                              fg_stream_t stream) {
  if (N < 0 || n_old < 0 || n_child < 0 || n_split_total < 0 || n_split_samples < 0) return FG_ERR_INVALID_ARG;
  if (N == 0) return FG_OK;
  if (!flags || !pos_old || !pos_child || !pos_dup || !rank_split || !src_index || !sample_index)
    return FG_ERR_INVALID_ARG;
  hipLaunchKernelGGL(densify_map_kernel, dim3((N + 255) / 256), dim3(256), 0, fg_hip_stream(stream), N, flags, pos_old,
                     pos_child, pos_dup, rank_split, n_old, n_child, n_split_total, n_split_samples, src_index,
                     sample_index);
  FG_RETURN_IF_LAUNCH_FAILED();
  return FG_OK;
}

extern "C" int fg_gather_rows(int64_t n_rows, int row_floats, const float* src, const int32_t* src_index,
                              int64_t zero_from, float* dst, fg_stream_t stream) {
  if (n_rows < 0 || row_floats <= 0 || zero_from < 0) return FG_ERR_INVALID_ARG;
  if (n_rows == 0) return FG_OK;
  if (!dst || !src_index || (zero_from > 0 && !src)) return FG_ERR_INVALID_ARG;
  if (zero_from > 0x7fffffff) return FG_ERR_UNSUPPORTED;
  const int64_t total = n_rows * row_floats;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, fg_hip_stream(stream),
                     total, row_floats, src, src_index, (int)(zero_from < n_rows ? zero_from : n_rows), dst);
  FG_RETURN_IF_LAUNCH_FAILED();
  return FG_OK;
}

extern "C" int fg_split_children(int first_row, int n_rows, const int32_t* sample_index, const float* samples,
                                 float* means, float* log_scales, const float* quats, fg_stream_t stream) {
  if (first_row < 0 || n_rows < 0) return FG_ERR_INVALID_ARG;
  if (n_rows == 0) return FG_OK;
  if (!sample_index || !samples || !means || !log_scales || !quats) return FG_ERR_INVALID_ARG;
  hipLaunchKernelGGL(split_children_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, fg_hip_stream(stream),
                     first_row, n_rows, sample_index, samples, means, log_scales, quats);
  FG_RETURN_IF_LAUNCH_FAILED();
  return FG_OK;
}
