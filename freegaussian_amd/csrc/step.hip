// One C-ABI call per direction for the eager training step (ABI 7; footprint masks: ABI 8): fg_step_fwd = the per-Gaussian forward ->
// supertile count -> fill + job lists -> raster forward; fg_step_bwd = raster backward -> per-Gaussian backward.
//
// What it replaces: the four + two calls a host otherwise makes per view around the rasterization(...) of
// /root/reference freegaussian/freegaussian_model.py:847-868 -- each with thirty-odd marshalled arguments and a handful
// of buffer allocations in between, which is what a launch-bound step (the reference's quarter- and half-resolution
// phases, :626-633) spends its time on.  Here the host fills three small structs (the description of the step, the
// pointers of its inputs / gradients, the launch policy), asks ONCE per shape for the layout of two caller-allocated
// workspaces -- `keep` (everything the backward and the caller read: per-Gaussian outputs, records, lists, job lists,
// liveness, checkpoints, images) and `tmp` (the binning's tables and sort buffers, dead when the call returns its
// launches) -- and makes one call.  Nothing new runs on the device: the same kernels in the same order.
#include <string.h>

#include "fg_common.h"

namespace {

size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

int channels_of(const fg_step_desc& d) {
  const int ncol = d.sh_degree >= 0 ? 3 : d.n_color;
  return ncol + (d.with_depth ? 1 : 0) + d.n_extra;
}

bool desc_ok(const fg_step_desc* d) {
  if (!d || d->size < (int32_t)sizeof(fg_step_desc)) return false;
  if (d->N <= 0 || d->width <= 0 || d->height <= 0 || d->tile_size != 16 || d->capacity <= 0) return false;
  if (d->raw && d->sh_degree < 0) return false;
  const int c = channels_of(*d);
  return c >= 1 && c <= FG_MAX_CHANNELS && d->n_clamp >= 0 && d->n_clamp <= c;
}

template <typename T>
T* at(void* base, const fg_step_layout* L, int which) {
  return L->nbytes[which] > 0 ? reinterpret_cast<T*>(static_cast<char*>(base) + L->offset[which]) : nullptr;
}

}  // namespace

extern "C" int fg_step_layout_query(const fg_step_desc* d, const fg_raster_config* config, fg_step_layout* out) {
  if (!desc_ok(d) || !out) return FG_ERR_INVALID_ARG;
  const int N = d->N, W = d->width, H = d->height;
  const int tile_w = (W + 15) / 16, tile_h = (H + 15) / 16, T = tile_w * tile_h, C = channels_of(*d);
  if (!fg_stbin_supported(N, tile_w, tile_h)) return FG_ERR_UNSUPPORTED;
  memset(out, 0, sizeof(*out));
  out->channels = C;
  out->jobs_words = fg_raster_jobs_words(W, H, 16, config);
  if (out->jobs_words <= 0) return FG_ERR_UNSUPPORTED;  // classic launches (tiny images): the stage-wise entry points
  out->seg_ckpt_floats = d->want_backward && d->list_shares ? fg_raster_seg_ckpt_floats(C, W, H, 16, d->capacity, config) : 0;
  size_t n[FG_STEP_BUFFERS] = {0};
  n[FG_STEP_RADII] = (size_t)N * 4;
  n[FG_STEP_MEANS2D] = (size_t)N * 8;
  n[FG_STEP_DEPTHS] = (size_t)N * 4;
  n[FG_STEP_CONICS] = (size_t)N * 12;
  n[FG_STEP_COMP] = d->antialiased ? (size_t)N * 4 : 0;
  n[FG_STEP_TILES] = (size_t)N * 4;
  n[FG_STEP_SPLATS] = (size_t)N * FG_SPLAT_FLOATS * 4;
  n[FG_STEP_DEPTH_KEYS] = (size_t)N * 4;
  n[FG_STEP_TILE_RECTS] = (size_t)N * 8;
  n[FG_STEP_TILE_MASKS] = (d->flags & FG_STEP_NO_FOOTPRINT_MASKS) ? 0 : (size_t)N * 8;
  n[FG_STEP_SH_JAC] = d->sh_degree >= 1 && d->want_backward ? (size_t)N * FG_SH_JAC_FLOATS * 4 : 0;
  n[FG_STEP_TILE_OFFSETS] = (size_t)(T + 1) * 4;
  n[FG_STEP_LIST_OFFSETS] = (size_t)(T + 1) * 4;
  n[FG_STEP_FLATTEN_IDS] = (size_t)d->capacity * 4;
  n[FG_STEP_JOBS] = (size_t)out->jobs_words * 4 * 2;
  n[FG_STEP_LIVE] = d->want_backward ? (size_t)d->capacity * 4 : 0;
  n[FG_STEP_SEG_CKPT] = (size_t)out->seg_ckpt_floats * 4;
  n[FG_STEP_V_SPLATS] = d->want_backward ? (size_t)N * FG_SPLAT_FLOATS * 4 : 0;
  n[FG_STEP_RENDER] = (size_t)W * H * C * 4;
  n[FG_STEP_ALPHAS] = (size_t)W * H * 4;
  n[FG_STEP_LAST_IDS] = (size_t)W * H * 4;
  n[FG_STEP_CLAMP_MASK] = d->n_clamp > 0 ? (size_t)W * H : 0;
  size_t o = 0;
  for (int b = 0; b < FG_STEP_COUNT_WS; ++b) {
    out->offset[b] = (int64_t)o;
    out->nbytes[b] = (int64_t)n[b];
    o += al256(n[b]);
  }
  out->keep_bytes = (int64_t)(o ? o : 256);
  n[FG_STEP_COUNT_WS] = fg_stbin_count_workspace_bytes(N, tile_w, tile_h);
  n[FG_STEP_FILL_WS] = fg_stbin_fill_workspace_bytes(d->capacity);
  o = 0;
  for (int b = FG_STEP_COUNT_WS; b < FG_STEP_BUFFERS; ++b) {  // (offsets into `tmp`)
    out->offset[b] = (int64_t)o;
    out->nbytes[b] = (int64_t)n[b];
    o += al256(n[b]);
  }
  out->tmp_bytes = (int64_t)o;
  return FG_OK;
}

extern "C" int fg_step_fwd(const fg_step_desc* d, const fg_raster_config* config, const fg_step_io* io, void* keep,
                           void* tmp, const fg_step_layout* L, fg_stream_t stream) {
  if (!desc_ok(d) || !io || !keep || !tmp || !L) return FG_ERR_INVALID_ARG;
  const int N = d->N, W = d->width, H = d->height, C = channels_of(*d);
  if (C != L->channels || L->keep_bytes <= 0) return FG_ERR_INVALID_ARG;
  const int tile_w = (W + 15) / 16, tile_h = (H + 15) / 16;
  int32_t* radii = at<int32_t>(keep, L, FG_STEP_RADII);
  float* means2d = at<float>(keep, L, FG_STEP_MEANS2D);
  float* depths = at<float>(keep, L, FG_STEP_DEPTHS);
  float* conics = at<float>(keep, L, FG_STEP_CONICS);
  float* comp = at<float>(keep, L, FG_STEP_COMP);
  int32_t* tiles = at<int32_t>(keep, L, FG_STEP_TILES);
  float* splats = at<float>(keep, L, FG_STEP_SPLATS);
  uint32_t* depth_keys = at<uint32_t>(keep, L, FG_STEP_DEPTH_KEYS);
  int32_t* tile_rects = at<int32_t>(keep, L, FG_STEP_TILE_RECTS);
  uint64_t* tile_masks = at<uint64_t>(keep, L, FG_STEP_TILE_MASKS);
  float* sh_jac = at<float>(keep, L, FG_STEP_SH_JAC);
  int rc;
  if (d->raw)
    rc = fg_preprocess_raw_fwd(N, io->means, io->quats, io->d_quats, io->scales, io->d_scales, io->opacities, io->colors,
                               io->features_rest, d->sh_degree, d->k_stored, d->with_depth, io->extra, d->n_extra, io->viewmat,
                               io->K, W, H, d->eps2d, d->near_plane, d->far_plane, d->radius_clip, 16, d->antialiased, radii,
                               means2d, depths, conics, comp, tiles, splats, depth_keys, tile_rects, tile_masks, sh_jac, stream);
  else
    rc = fg_preprocess_fwd(N, io->means, io->quats, io->scales, io->opacities, io->colors, d->sh_degree, d->k_stored,
                           d->n_color, d->with_depth, io->extra, d->n_extra, io->viewmat, io->K, W, H, d->eps2d, d->near_plane,
                           d->far_plane, d->radius_clip, 16, d->antialiased, radii, means2d, depths, conics, comp, tiles,
                           splats, depth_keys, tile_rects, tile_masks, sh_jac, stream);
  if (rc != FG_OK) return rc;
  int32_t* tile_offsets = at<int32_t>(keep, L, FG_STEP_TILE_OFFSETS);
  int32_t* list_offsets = at<int32_t>(keep, L, FG_STEP_LIST_OFFSETS);
  void* count_ws = at<char>(tmp, L, FG_STEP_COUNT_WS);
  rc = fg_stbin_count(N, tile_rects, tile_masks, tile_w, tile_h, tile_offsets, io->count_out, count_ws,
                      (size_t)L->nbytes[FG_STEP_COUNT_WS], stream);
  if (rc != FG_OK) return rc;
  int32_t* flatten_ids = at<int32_t>(keep, L, FG_STEP_FLATTEN_IDS);
  int32_t* jobs = at<int32_t>(keep, L, FG_STEP_JOBS);
  float* seg_ckpt = at<float>(keep, L, FG_STEP_SEG_CKPT);
  rc = fg_stbin_fill_jobs(N, depth_keys, tile_rects, tile_masks, tile_w, tile_h, d->capacity, tile_offsets, count_ws, flatten_ids,
                          list_offsets, at<char>(tmp, L, FG_STEP_FILL_WS), (size_t)L->nbytes[FG_STEP_FILL_WS], W, H, 16, jobs,
                          jobs + L->jobs_words, seg_ckpt != nullptr, config, d->flags & (FG_STBIN_LONG_SEGMENTS | FG_STBIN_TEST_SMALL_SLABS), io->ckpt_need_out,
                          stream);
  if (rc != FG_OK) return rc;
  float* v_splats = at<float>(keep, L, FG_STEP_V_SPLATS);
  if (io->ev_raster_begin && hipEventRecord(static_cast<hipEvent_t>(io->ev_raster_begin), fg_hip_stream(stream)) != hipSuccess)
    return FG_ERR_LAUNCH;
  rc = fg_raster_jobs_fwd(C, W, H, 16, splats, list_offsets, flatten_ids, jobs, io->background, d->n_clamp,
                          at<float>(keep, L, FG_STEP_RENDER), at<float>(keep, L, FG_STEP_ALPHAS),
                          at<int32_t>(keep, L, FG_STEP_LAST_IDS), at<uint8_t>(keep, L, FG_STEP_CLAMP_MASK), seg_ckpt,
                          at<uint32_t>(keep, L, FG_STEP_LIVE), v_splats, v_splats ? (int64_t)N * FG_SPLAT_FLOATS : 0,
                          io->ckpt_need_out ? io->ckpt_need_out + 9 : nullptr, config, stream);
  if (io->ev_raster_end && hipEventRecord(static_cast<hipEvent_t>(io->ev_raster_end), fg_hip_stream(stream)) != hipSuccess)
    return FG_ERR_LAUNCH;
  return rc;
}

extern "C" int fg_step_bwd(const fg_step_desc* d, const fg_raster_config* config, const fg_step_io* io, void* keep,
                           const fg_step_layout* L, fg_stream_t stream) {
  if (!desc_ok(d) || !io || !keep || !L || !d->want_backward) return FG_ERR_INVALID_ARG;
  if (!io->v_render || !io->v_means || !io->v_quats || !io->v_scales || !io->v_opacities) return FG_ERR_INVALID_ARG;
  const int N = d->N, W = d->width, H = d->height, C = channels_of(*d);
  float* v_splats = at<float>(keep, L, FG_STEP_V_SPLATS);
  const float* splats = at<float>(keep, L, FG_STEP_SPLATS);
  int32_t* jobs = at<int32_t>(keep, L, FG_STEP_JOBS);
  if (io->ev_raster_begin && hipEventRecord(static_cast<hipEvent_t>(io->ev_raster_begin), fg_hip_stream(stream)) != hipSuccess)
    return FG_ERR_LAUNCH;
  int rc = fg_raster_jobs_bwd(C, W, H, 16, splats, at<int32_t>(keep, L, FG_STEP_LIST_OFFSETS),
                              at<int32_t>(keep, L, FG_STEP_FLATTEN_IDS), jobs + L->jobs_words, io->background, d->n_clamp,
                              at<uint8_t>(keep, L, FG_STEP_CLAMP_MASK), at<float>(keep, L, FG_STEP_ALPHAS),
                              at<int32_t>(keep, L, FG_STEP_LAST_IDS), io->v_render, io->v_alphas, v_splats,
                              at<float>(keep, L, FG_STEP_SEG_CKPT), at<float>(keep, L, FG_STEP_RENDER),
                              at<uint32_t>(keep, L, FG_STEP_LIVE), config, stream);
  if (io->ev_raster_end && hipEventRecord(static_cast<hipEvent_t>(io->ev_raster_end), fg_hip_stream(stream)) != hipSuccess)
    return FG_ERR_LAUNCH;
  if (rc != FG_OK) return rc;
  const int32_t* radii = at<int32_t>(keep, L, FG_STEP_RADII);
  const float* sh_jac = at<float>(keep, L, FG_STEP_SH_JAC);
  // (the xy gradient is read out of the record gradients in place: pointer + stride)
  if (d->raw) {
    if (io->v_rgb)
      return fg_preprocess_raw_bwd_factored(N, io->means, io->quats, io->d_quats, io->scales, io->d_scales, io->opacities,
                                            io->colors, io->features_rest, d->sh_degree, d->k_stored, d->with_depth, d->n_extra,
                                            io->viewmat, io->K, W, H, d->eps2d, d->antialiased, radii, v_splats, v_splats,
                                            FG_SPLAT_FLOATS, io->v_depths, io->v_conics, io->v_means, io->v_quats, io->v_d_quats,
                                            io->v_scales, io->v_d_scales, io->v_opacities, io->v_rgb, io->v_rgb_floats,
                                            io->v_extra, sh_jac, stream);
    return fg_preprocess_raw_bwd(N, io->means, io->quats, io->d_quats, io->scales, io->d_scales, io->opacities, io->colors,
                                 io->features_rest, d->sh_degree, d->k_stored, d->with_depth, d->n_extra, io->viewmat, io->K, W,
                                 H, d->eps2d, d->antialiased, radii, v_splats, v_splats, FG_SPLAT_FLOATS, io->v_depths,
                                 io->v_conics, io->v_means, io->v_quats, io->v_d_quats, io->v_scales, io->v_d_scales, io->v_opacities,
                                 io->v_colors, io->v_features_rest, io->v_extra, sh_jac, stream);
  }
  if (io->v_rgb)
    return fg_preprocess_bwd_factored(N, io->means, io->quats, io->scales, io->opacities, io->colors, d->sh_degree, d->k_stored,
                                      d->with_depth, d->n_extra, io->viewmat, io->K, W, H, d->eps2d, d->antialiased, radii,
                                      v_splats, v_splats, FG_SPLAT_FLOATS, io->v_depths, io->v_conics, io->v_means, io->v_quats,
                                      io->v_scales, io->v_opacities, io->v_rgb, io->v_rgb_floats, io->v_extra, sh_jac, stream);
  return fg_preprocess_bwd(N, io->means, io->quats, io->scales, io->opacities, io->colors, d->sh_degree, d->k_stored, d->n_color,
                           d->with_depth, d->n_extra, io->viewmat, io->K, W, H, d->eps2d, d->antialiased, radii, v_splats,
                           v_splats, FG_SPLAT_FLOATS, io->v_depths, io->v_conics, io->v_means, io->v_quats, io->v_scales, io->v_opacities,
                           io->v_colors, io->v_extra, sh_jac, stream);
}
