// Attribute-mask back-projection: the per-key-frame body of
// /root/reference preprocess/knn_gaussian.py:116-132.  Every visible Gaussian looks up the pixel
// its centre projects to, keeps itself only if its depth agrees with the rendered expected depth
// there (the reference's "HACK: filtering", :120-123) and ORs that pixel's valid 2-D attribute
// labels into its row of gaussian_mask[N,M] -- the array saved as gaussian_mask_NxM.npy and read by
// stage 2 (freegaussian_pipeline.py:45-47).  One lane per Gaussian, a gather; bool = 1 byte.
#include "fg_common.h"

namespace {

__global__ void __launch_bounds__(256)
mask_backproject_kernel(int N, const float* __restrict__ means2d, const float* __restrict__ depths,
                        const int32_t* __restrict__ radii, const float* __restrict__ depth_map, int width, int height,
                        const uint8_t* __restrict__ atrb_masks, const uint8_t* __restrict__ mask_valids, int m_stored,
                        int M, uint8_t* __restrict__ gaussian_masks) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N || radii[i] <= 0) return;
  // `.long()` truncates toward zero (:117): a centre in (-1, 0) lands on pixel 0
  const long long x = (long long)means2d[2 * i], y = (long long)means2d[2 * i + 1];
  if (x < 0 || y < 0 || x >= width || y >= height) return;  // :118
  const size_t pix = (size_t)y * width + x;
  const float d = depth_map[pix];
  const float delta = d - depths[i];                        // :121
  if (!((-d * 0.1f < delta) && (delta < d * 1.f))) return;  // :122
  const uint8_t* labels = atrb_masks + pix * m_stored;
  for (int j = 0; j < M; ++j)
    if (labels[j] && mask_valids[j]) gaussian_masks[(size_t)i * M + j] = 1;  // :127-132
}

}  // namespace

extern "C" int fg_mask_backproject(int N, const float* means2d, const float* depths, const int32_t* radii,
                                   const float* depth_map, int width, int height, const uint8_t* atrb_masks,
                                   const uint8_t* mask_valids, int n_labels_stored, int n_attributes,
                                   uint8_t* gaussian_masks, fg_stream_t stream) {
  if (N < 0 || width <= 0 || height <= 0 || n_attributes < 0 || n_labels_stored < n_attributes) return FG_ERR_INVALID_ARG;
  if (N == 0 || n_attributes == 0) return FG_OK;
  if (!means2d || !depths || !radii || !depth_map || !atrb_masks || !mask_valids || !gaussian_masks)
    return FG_ERR_INVALID_ARG;
  hipLaunchKernelGGL(mask_backproject_kernel, dim3((N + 255) / 256), dim3(256), 0, fg_hip_stream(stream), N, means2d,
                     depths, radii, depth_map, width, height, atrb_masks, mask_valids, n_labels_stored, n_attributes,
                     gaussian_masks);
  FG_RETURN_IF_LAUNCH_FAILED();
  return FG_OK;
}
