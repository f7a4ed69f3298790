// K2 / K8: view-dependent colour from real spherical harmonics, forward and backward.
//
// coeffs is [N, k_stored, 3] (the reference concatenates features_dc/features_rest into
// [N,16,3], freegaussian/freegaussian_model.py:801) -- an array of 192-byte rows.  A lane per
// Gaussian reading its row directly would touch 64 cache lines per load instruction, so a
// 256-Gaussian workgroup first streams its contiguous [256 x 3k] slab through LDS with fully
// coalesced loads (consecutive lanes -> consecutive dwords), padded to an odd row stride so
// the per-lane row reads are bank-conflict free.  The backward writes v_coeffs the same way.
#include "sh_math.h"
#include "slab_io.h"

namespace {
using namespace fgsh;

// Stream the first k3 = 3*(degree+1)^2 floats of each of the block's rows into LDS.
__device__ __forceinline__ void stage_rows_in(float* lds, const float* __restrict__ coeffs, int row0, int nrows,
                                              int k3, int row_floats) {
  const int total = nrows * k3;
  for (int e = threadIdx.x; e < total; e += BLOCK) {
    const int r = e / k3, c = e - r * k3;
    lds[r * ROW + c] = coeffs[(size_t)(row0 + r) * row_floats + c];
  }
}

__global__ void __launch_bounds__(BLOCK)
sh_fwd_kernel(int N, int degree, int k_stored, const float* __restrict__ means, const float* __restrict__ viewmat,
              const float* __restrict__ coeffs, const int32_t* __restrict__ radii, float* __restrict__ colors) {
  __shared__ float lds[BLOCK * ROW];
  const int row0 = blockIdx.x * BLOCK;
  const int nrows = min(BLOCK, N - row0);
  const int kk = (degree + 1) * (degree + 1);
  stage_rows_in(lds, coeffs, row0, nrows, 3 * kk, 3 * k_stored);
  __syncthreads();
  const int i = row0 + threadIdx.x;
  if (i >= N) return;
  float r = 0.f, g = 0.f, b = 0.f;
  if (!radii || radii[i] > 0) {
    float dx, dy, dz, inv;
    view_dir(viewmat, means[3 * i], means[3 * i + 1], means[3 * i + 2], dx, dy, dz, inv);
    float basis[16];
    sh_basis(degree, dx, dy, dz, basis);
    sh_dot(basis, lds + threadIdx.x * ROW, kk, r, g, b);
    r = fmaxf(r + 0.5f, 0.f);
    g = fmaxf(g + 0.5f, 0.f);
    b = fmaxf(b + 0.5f, 0.f);
  }
  colors[3 * i] = r;
  colors[3 * i + 1] = g;
  colors[3 * i + 2] = b;
}

__global__ void __launch_bounds__(BLOCK)
sh_bwd_kernel(int N, int degree, int k_stored, const float* __restrict__ means, const float* __restrict__ viewmat,
              const float* __restrict__ coeffs, const int32_t* __restrict__ radii,
              const float* __restrict__ colors, const float* __restrict__ v_colors,
              float* __restrict__ v_coeffs, float* __restrict__ v_means) {
  __shared__ float lds[BLOCK * ROW];
  const int row0 = blockIdx.x * BLOCK;
  const int nrows = min(BLOCK, N - row0);
  const int kk = (degree + 1) * (degree + 1);
  const int i = row0 + threadIdx.x;
  const bool need_dir = (v_means != nullptr) && degree > 0;
  if (need_dir) {
    stage_rows_in(lds, coeffs, row0, nrows, 3 * kk, 3 * k_stored);
    __syncthreads();
  }
  float vr = 0.f, vg = 0.f, vb = 0.f;
  float basis[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) basis[k] = 0.f;
  float gmx = 0.f, gmy = 0.f, gmz = 0.f;
  const bool active = (i < N) && (!radii || radii[i] > 0);
  if (active) {
    // clamp: colour = max(sh + 0.5, 0) passes gradient only where the output is positive
    vr = colors[3 * i] > 0.f ? v_colors[3 * i] : 0.f;
    vg = colors[3 * i + 1] > 0.f ? v_colors[3 * i + 1] : 0.f;
    vb = colors[3 * i + 2] > 0.f ? v_colors[3 * i + 2] : 0.f;
    float dx, dy, dz, inv;
    view_dir(viewmat, means[3 * i], means[3 * i + 1], means[3 * i + 2], dx, dy, dz, inv);
    sh_basis(degree, dx, dy, dz, basis);
    if (need_dir) {
      float bx[16], by[16], bz[16];
      sh_basis_grad(degree, dx, dy, dz, bx, by, bz);
      const float* row = lds + threadIdx.x * ROW;
      float vdx = 0.f, vdy = 0.f, vdz = 0.f;
#pragma unroll
      for (int k = 1; k < 16; ++k) {
        if (k < kk) {
          const float s = vr * row[3 * k] + vg * row[3 * k + 1] + vb * row[3 * k + 2];
          vdx += bx[k] * s; vdy += by[k] * s; vdz += bz[k] * s;
        }
      }
      // through the normalisation d = u/|u|
      const float dp = vdx * dx + vdy * dy + vdz * dz;
      gmx = (vdx - dp * dx) * inv;
      gmy = (vdy - dp * dy) * inv;
      gmz = (vdz - dp * dz) * inv;
    }
  }
  if (v_means && i < N) {
    v_means[3 * i] = gmx; v_means[3 * i + 1] = gmy; v_means[3 * i + 2] = gmz;
  }
  // v_coeffs[i][k][c] = basis[k] * v_colour[c]; stage through LDS, then stream out whole rows
  if (need_dir) __syncthreads();  // everyone is done reading coeffs from lds
  {
    float* row = lds + threadIdx.x * ROW;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float bk = (k < kk) ? basis[k] : 0.f;
      row[3 * k] = bk * vr; row[3 * k + 1] = bk * vg; row[3 * k + 2] = bk * vb;
    }
  }
  __syncthreads();
  const int row_floats = 3 * k_stored;
  const int total = nrows * row_floats;
  for (int e = threadIdx.x; e < total; e += BLOCK) {
    const int r = e / row_floats, c = e - r * row_floats;
    v_coeffs[(size_t)(row0 + r) * row_floats + c] = (c < 48) ? lds[r * ROW + c] : 0.f;
  }
}

// View-DP exchange, second half: every rank holds the masked colour gradients g_v[N,3] of ALL views
// (all-gathered, 12 B per Gaussian per view) and rebuilds the summed coefficient gradient
//   v_coeffs[i,k,:] = scale * sum_v basis_k(dir_v(i)) * g_v[i,:]
// locally -- 192 B per Gaussian that never cross xGMI.  payload: n_views blocks of view_stride
// floats; payload_floats == 3: block v = [g_v (3N floats) | camera position of view v (3 floats)]
// and the direction is normalize(means - camera); payload_floats == 6: block v = N rows of
// [g_v (3) | unit direction (3)] (per-view deformed means: every rank saw different positions).
__global__ void __launch_bounds__(BLOCK)
sh_grad_accumulate_kernel(int N, int n_views, int degree, int k_stored, const float* __restrict__ means,
                          const float* __restrict__ payload, int64_t view_stride, int payload_floats, float scale,
                          float* __restrict__ v_coeffs, float* __restrict__ v_rest) {
  __shared__ float lds[BLOCK * ROW];
  const int row0 = blockIdx.x * BLOCK;
  const int nrows = min(BLOCK, N - row0);
  const int i = row0 + threadIdx.x;
  const int kk = (degree + 1) * (degree + 1);
  float acc[48];
#pragma unroll
  for (int q = 0; q < 48; ++q) acc[q] = 0.f;
  if (i < N) {
    float mx = 0.f, my = 0.f, mz = 0.f;
    if (payload_floats == 3) { mx = means[3 * i]; my = means[3 * i + 1]; mz = means[3 * i + 2]; }
    for (int v = 0; v < n_views; ++v) {
      const float* blk = payload + (size_t)v * view_stride;
      const float* row = blk + (size_t)payload_floats * i;
      const float gr = row[0], gg = row[1], gb = row[2];
      if (gr == 0.f && gg == 0.f && gb == 0.f) continue;  // culled or untouched in that view
      float dx, dy, dz;
      if (payload_floats == 6) {
        dx = row[3]; dy = row[4]; dz = row[5];
      } else {
        const float* cam = blk + (size_t)3 * N;
        dx = mx - cam[0]; dy = my - cam[1]; dz = mz - cam[2];
        const float inv = 1.f / sqrtf(dx * dx + dy * dy + dz * dz);
        dx *= inv; dy *= inv; dz *= inv;
      }
      float basis[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) basis[k] = 0.f;
      sh_basis(degree, dx, dy, dz, basis);
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        if (k < kk) {
          acc[3 * k] += basis[k] * gr; acc[3 * k + 1] += basis[k] * gg; acc[3 * k + 2] += basis[k] * gb;
        }
      }
    }
  }
  float* row = lds + threadIdx.x * ROW;
#pragma unroll
  for (int q = 0; q < 48; ++q) row[q] = acc[q] * scale;
  __syncthreads();
  if (v_rest) {  // the model's layout: features_dc [N,3] and features_rest [N,k_stored - 1,3] as two arrays
    lds_to_slab_at(v_coeffs + (size_t)row0 * 3, lds, 0, nrows, 3, 3);
    if (k_stored > 1) lds_to_slab_at(v_rest + (size_t)row0 * 3 * (k_stored - 1), lds, 3, nrows, 3 * (k_stored - 1), 45);
  } else {
    lds_to_slab_at(v_coeffs + (size_t)row0 * 3 * k_stored, lds, 0, nrows, 3 * k_stored, 48);
  }
}

}  // namespace

namespace {
int sh_grad_accumulate(int N, int n_views, int sh_degree, int k_stored, const float* means, const float* payload,
                       int64_t view_stride, int payload_floats, float scale, float* v_coeffs, float* v_rest,
                       fg_stream_t stream) {
  if (N < 0 || n_views < 1 || sh_degree < 0 || sh_degree > 3 || k_stored < (sh_degree + 1) * (sh_degree + 1) ||
      k_stored > 16 || (payload_floats != 3 && payload_floats != 6) ||
      view_stride < (payload_floats == 3 ? (int64_t)3 * N + 3 : (int64_t)6 * N))
    return FG_ERR_INVALID_ARG;
  if (N == 0) return FG_OK;
  if ((payload_floats == 3 && !means) || !payload || !v_coeffs) return FG_ERR_INVALID_ARG;
  hipLaunchKernelGGL(sh_grad_accumulate_kernel, dim3((N + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, fg_hip_stream(stream),
                     N, n_views, sh_degree, k_stored, means, payload, view_stride, payload_floats, scale, v_coeffs, v_rest);
  FG_RETURN_IF_LAUNCH_FAILED();
  return FG_OK;
}
}  // namespace

// ---- view-DP exchange, sparse payload (round 5) ----------------------------------------------------------------------
// Only Gaussians that took part in a pixel of the view carry a colour gradient: on a captured scene under a real frustum that
// is a fraction of N (and 0.5-0.6 N even on the bench's everything-in-view cube).  The all-gathered block of a rank can then
// be [header: count, camera position | capacity rows of (id, g[3] (, direction[3]))] instead of N dense rows; every rank
// EXPANDS the gathered blocks into dense ones locally (HBM traffic, not link traffic) and rebuilds as before.
// compact: rows with a non-zero g, in id order (incl = the inclusive scan of the row flags: row i goes to slot
// incl[i] - 1); rows beyond `capacity` are dropped -- the header's count says so (count > capacity: the receiver must
// not use the block).  out: [4 + capacity * (1 + payload_floats)] floats; out[0] = count (as int bits), out[1..3] are
// the caller's (camera position).
namespace {
__global__ void __launch_bounds__(256)
payload_compact_kernel(int N, int pf, const float* __restrict__ dense, const int32_t* __restrict__ incl, int64_t capacity,
                       float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const int here = incl[i], before = i > 0 ? incl[i - 1] : 0;
  if (i == N - 1) out[0] = __int_as_float(here);
  if (here == before || (int64_t)here > capacity) return;
  float* row = out + 4 + (size_t)(here - 1) * (1 + pf);
  row[0] = __int_as_float(i);
  for (int c = 0; c < pf; ++c) row[1 + c] = dense[(size_t)i * pf + c];
}
// expand: n_views compact blocks (block_stride floats apart) -> n_views dense blocks of dense_stride floats, laid out as
// fg_sh_grad_accumulate reads them (payload_floats 3: [g (3N) | camera position (3)]; 6: N rows of [g | direction]).  The
// dense blocks must be ZERO on entry (rows without an entry stay zero = "no gradient in that view").
__global__ void __launch_bounds__(256)
payload_expand_kernel(int N, int pf, int n_views, const float* __restrict__ compact, int64_t block_stride, int64_t capacity,
                      float* __restrict__ dense, int64_t dense_stride) {
  const int v = blockIdx.y;
  const float* blk = compact + (size_t)v * block_stride;
  float* out = dense + (size_t)v * dense_stride;
  const int64_t count = min((int64_t)__float_as_int(blk[0]), capacity);
  if (pf == 3 && blockIdx.x == 0 && threadIdx.x < 3) out[(size_t)3 * N + threadIdx.x] = blk[1 + threadIdx.x];
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < count; e += (int64_t)gridDim.x * 256) {
    const float* row = blk + 4 + (size_t)e * (1 + pf);
    const int id = __float_as_int(row[0]);
    if (id < 0 || id >= N) continue;
    for (int c = 0; c < pf; ++c) out[(size_t)id * pf + c] = row[1 + c];
  }
}
}  // namespace

extern "C" int fg_payload_compact(int N, int payload_floats, const float* dense, const int32_t* incl_scan, int64_t capacity,
                                  float* out, fg_stream_t stream) {
  if (N < 0 || capacity < 0 || (payload_floats != 3 && payload_floats != 6)) return FG_ERR_INVALID_ARG;
  if (N == 0) return FG_OK;
  if (!dense || !incl_scan || !out) return FG_ERR_INVALID_ARG;
  hipLaunchKernelGGL(payload_compact_kernel, dim3((N + 255) / 256), dim3(256), 0, fg_hip_stream(stream), N, payload_floats, dense,
                     incl_scan, capacity, out);
  FG_RETURN_IF_LAUNCH_FAILED();
  return FG_OK;
}

extern "C" int fg_payload_expand(int N, int payload_floats, int n_views, const float* compact, int64_t block_stride,
                                 int64_t capacity, float* dense, int64_t dense_stride, fg_stream_t stream) {
  if (N < 0 || n_views < 1 || capacity < 0 || (payload_floats != 3 && payload_floats != 6) ||
      block_stride < 4 + capacity * (1 + payload_floats) ||
      dense_stride < (payload_floats == 3 ? (int64_t)3 * N + 3 : (int64_t)6 * N))
    return FG_ERR_INVALID_ARG;
  if (N == 0) return FG_OK;
  if (!compact || !dense) return FG_ERR_INVALID_ARG;
  const int64_t blocks = (capacity + 255) / 256;
  hipLaunchKernelGGL(payload_expand_kernel, dim3((unsigned)(blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks)), n_views), dim3(256), 0,
                     fg_hip_stream(stream), N, payload_floats, n_views, compact, block_stride, capacity, dense, dense_stride);
  FG_RETURN_IF_LAUNCH_FAILED();
  return FG_OK;
}

extern "C" int fg_sh_grad_accumulate(int N, int n_views, int sh_degree, int k_stored, const float* means,
                                     const float* payload, int64_t view_stride, int payload_floats, float scale,
                                     float* v_coeffs, fg_stream_t stream) {
  return sh_grad_accumulate(N, n_views, sh_degree, k_stored, means, payload, view_stride, payload_floats, scale, v_coeffs,
                            nullptr, stream);
}

extern "C" int fg_sh_grad_accumulate_split(int N, int n_views, int sh_degree, int k_stored, const float* means,
                                           const float* payload, int64_t view_stride, int payload_floats, float scale,
                                           float* v_features_dc, float* v_features_rest, fg_stream_t stream) {
  if (k_stored > 1 && !v_features_rest) return FG_ERR_INVALID_ARG;
  return sh_grad_accumulate(N, n_views, sh_degree, k_stored, means, payload, view_stride, payload_floats, scale,
                            v_features_dc, v_features_rest ? v_features_rest : v_features_dc, stream);
}

extern "C" int fg_sh_fwd(int N, int degree, int k_stored, const float* means, const float* viewmat,
                         const float* coeffs, const int32_t* radii, float* colors, fg_stream_t stream) {
  if (N < 0 || degree < 0 || degree > 3 || k_stored < (degree + 1) * (degree + 1)) return FG_ERR_INVALID_ARG;
  if (N == 0) return FG_OK;
  if (!means || !viewmat || !coeffs || !colors) return FG_ERR_INVALID_ARG;
  hipLaunchKernelGGL(sh_fwd_kernel, dim3((N + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, fg_hip_stream(stream), N,
                     degree, k_stored, means, viewmat, coeffs, radii, colors);
  FG_RETURN_IF_LAUNCH_FAILED();
  return FG_OK;
}

extern "C" int fg_sh_bwd(int N, int degree, int k_stored, const float* means, const float* viewmat,
                         const float* coeffs, const int32_t* radii, const float* colors,
                         const float* v_colors, float* v_coeffs, float* v_means, fg_stream_t stream) {
  if (N < 0 || degree < 0 || degree > 3 || k_stored < (degree + 1) * (degree + 1)) return FG_ERR_INVALID_ARG;
  if (N == 0) return FG_OK;
  if (!means || !viewmat || !coeffs || !colors || !v_colors || !v_coeffs) return FG_ERR_INVALID_ARG;
  hipLaunchKernelGGL(sh_bwd_kernel, dim3((N + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, fg_hip_stream(stream), N,
                     degree, k_stored, means, viewmat, coeffs, radii, colors, v_colors, v_coeffs, v_means);
  FG_RETURN_IF_LAUNCH_FAILED();
  return FG_OK;
}
