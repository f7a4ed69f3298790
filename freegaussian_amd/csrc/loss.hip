// L: the image loss of the training step, fused.
//
//   main_loss = (1 - ssim_lambda) * mean|gt - pred| + ssim_lambda * (1 - SSIM(gt, pred))
//        /root/reference freegaussian/freegaussian_model.py:965-981, with
//        SSIM = pytorch_msssim.SSIM(data_range=1.0, size_average=True, channel=3) (:22, :211): 11-tap Gaussian
//        window, sigma 1.5, separable, 'valid' borders, mean of the (H-10) x (W-10) x C map.
//
// As torch operators (five depthwise 11 x 11 convolutions and ~40 elementwise kernels forward, as many again
// backward: 202 launches) this costs 10.4 ms per step at 1920 x 1080 on an MI355X -- fourteen times the whole
// rasterizer step (scripts/loss_time.py).  Here: one launch forward, one backward, both HBM-bound.
//
//   forward   one 256-thread workgroup per 32 x 16 pixel tile and channel: the 42 x 26 input windows of pred and gt
//             in LDS, the five windowed moments (x, y, xx, yy, xy) by a horizontal (four columns per thread) then a
//             vertical (two rows per thread) 11-tap pass -- both passes are bound by instruction issue, not bytes --,
//             the SSIM map value and its three partial derivatives (d/d mu_x, d/d E[xx], d/d E[xy], the first one
//             total: including the paths through the variances) per map pixel -> maps[3, C, H-10, W-10];
//             per-workgroup partial sums of |x - y| and of the map, reduced in a fixed order by a second,
//             one-workgroup launch (no float atomics: the loss value is reproducible bit for bit).
//   backward  d mean(SSIM) / d x[p] = sum over the map pixels q whose window holds p of
//                 w(p - q) * (D_mu[q] + 2 x[p] D_xx[q] + y[p] D_xy[q]):
//             the three maps convolved with the same window ('full' borders), same tiling; plus the L1 sign term.
//
// Images are [H, W, C] as the model holds them (no permuted copies); only pred receives a gradient.
#include "fg_common.h"

namespace {

constexpr int TW = 32, TH = 16;   // tile: 32 columns x 16 rows of pixels per 256-thread workgroup (two rows per thread)
constexpr int WIN = 11;           // taps
constexpr int HALO = WIN - 1;     // 10
constexpr int INW = TW + HALO;    // 42 input columns a tile needs
constexpr int INH = TH + HALO;    // 26 input rows
constexpr int INWP = INW + 1;     // padded LDS rows
constexpr int TWP = TW + 1;
constexpr float SSIM_C1 = 0.01f * 0.01f, SSIM_C2 = 0.03f * 0.03f;  // data_range = 1

struct Window {
  float g[WIN];
};

Window gaussian_window() {
  Window w;
  double s = 0.0, v[WIN];
  for (int i = 0; i < WIN; ++i) {
    const double d = (double)(i - WIN / 2);
    v[i] = exp(-(d * d) / (2.0 * 1.5 * 1.5));
    s += v[i];
  }
  for (int i = 0; i < WIN; ++i) w.g[i] = (float)(v[i] / s);
  return w;
}

// sums of two values over the 256 threads of a workgroup (fixed order)
__device__ __forceinline__ float2 block_sum2_256(float a, float b, float2* scratch) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) {
    a += __shfl_xor(a, m);
    b += __shfl_xor(b, m);
  }
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = make_float2(a, b);
  __syncthreads();
  return make_float2((scratch[0].x + scratch[1].x) + (scratch[2].x + scratch[3].x),
                     (scratch[0].y + scratch[1].y) + (scratch[2].y + scratch[3].y));
}

__global__ void __launch_bounds__(256)
l1_ssim_fwd_kernel(int H, int W, int C, Window win, const float* __restrict__ pred, const float* __restrict__ gt,
                   float* __restrict__ maps, float* __restrict__ partials) {
  __shared__ float sx[INH][INWP], sy[INH][INWP];
  __shared__ float hm[5][INH][TWP];
  __shared__ float2 red[4];
  const int c = blockIdx.z, x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
  const int Hm = H - HALO, Wm = W - HALO;
  // input window (zero outside the image: those taps only feed map pixels outside the map)
  for (int i = threadIdx.x; i < INH * INW; i += 256) {
    const int r = i / INW, q = i - r * INW;
    const int y = y0 + r, x = x0 + q;
    float a = 0.f, b = 0.f;
    if (y < H && x < W) {
      const size_t o = ((size_t)y * W + x) * C + c;
      a = pred[o];
      b = gt[o];
    }
    sx[r][q] = a;
    sy[r][q] = b;
  }
  __syncthreads();
  // a thread's two pixels: column tx, rows 2 ty and 2 ty + 1 of the tile
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  float l1 = 0.f;
#pragma unroll
  for (int o = 0; o < 2; ++o)
    if (y0 + 2 * ty + o < H && x0 + tx < W) l1 += fabsf(sy[2 * ty + o][tx] - sx[2 * ty + o][tx]);
  // horizontal pass: 26 rows x 32 columns x five moments; a thread takes FOUR neighbouring columns of a row (fourteen
  // inputs read once, their products formed once): the pass is bound by instruction issue, not by bytes
  if (threadIdx.x < INH * (TW / 4)) {
    const int r = threadIdx.x >> 3, q = (threadIdx.x & 7) * 4;
    float m[4][5];
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int v = 0; v < 5; ++v) m[o][v] = 0.f;
#pragma unroll
    for (int j = 0; j < WIN + 3; ++j) {
      const float a = sx[r][q + j], b = sy[r][q + j];
      const float aa = a * a, bb = b * b, ab = a * b;
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const int k = j - o;
        if (k >= 0 && k < WIN) {
          const float g = win.g[k];
          m[o][0] = fmaf(g, a, m[o][0]);
          m[o][1] = fmaf(g, b, m[o][1]);
          m[o][2] = fmaf(g, aa, m[o][2]);
          m[o][3] = fmaf(g, bb, m[o][3]);
          m[o][4] = fmaf(g, ab, m[o][4]);
        }
      }
    }
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int v = 0; v < 5; ++v) hm[v][r][q + o] = m[o][v];
  }
  __syncthreads();
  // vertical pass: the map pixels (y0 + 2 ty + {0, 1}, x0 + tx): twelve rows read once for the two
  float acc[2][5];
#pragma unroll
  for (int o = 0; o < 2; ++o)
#pragma unroll
    for (int v = 0; v < 5; ++v) acc[o][v] = 0.f;
#pragma unroll
  for (int j = 0; j < WIN + 1; ++j) {
    float h[5];
#pragma unroll
    for (int v = 0; v < 5; ++v) h[v] = hm[v][2 * ty + j][tx];
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const int k = j - o;
      if (k >= 0 && k < WIN) {
#pragma unroll
        for (int v = 0; v < 5; ++v) acc[o][v] = fmaf(win.g[k], h[v], acc[o][v]);
      }
    }
  }
  float ssim_sum = 0.f;
#pragma unroll
  for (int o = 0; o < 2; ++o) {
    const int my = y0 + 2 * ty + o, mx = x0 + tx;
    if (my < Hm && mx < Wm) {
      const float mu1 = acc[o][0], mu2 = acc[o][1], exx = acc[o][2], eyy = acc[o][3], exy = acc[o][4];
      const float s1 = exx - mu1 * mu1, s2 = eyy - mu2 * mu2, s12 = exy - mu1 * mu2;
      const float A1 = 2.f * mu1 * mu2 + SSIM_C1, A2 = 2.f * s12 + SSIM_C2;
      const float B1 = mu1 * mu1 + mu2 * mu2 + SSIM_C1, B2 = s1 + s2 + SSIM_C2;
      const float iB1 = 1.f / B1, iB2 = 1.f / B2;
      const float ssim = (A1 * iB1) * (A2 * iB2);
      // partial derivatives of the map value: with respect to E[xx] (= d/d sigma_x^2), E[xy] (= d/d sigma_xy) and
      // mu_x -- the last one total, i.e. including sigma_x^2 = E[xx] - mu_x^2 and sigma_xy = E[xy] - mu_x mu_y
      const float d_xx = -ssim * iB2;
      const float d_xy = 2.f * (A1 * iB1) * iB2;
      const float d_mu = 2.f * mu2 * (A2 * iB2) * iB1 - 2.f * mu1 * ssim * iB1 - 2.f * mu1 * d_xx - mu2 * d_xy;
      const size_t plane = (size_t)Hm * Wm, idx = (size_t)c * plane + (size_t)my * Wm + mx;
      maps[idx] = d_mu;
      maps[(size_t)C * plane + idx] = d_xx;
      maps[2 * (size_t)C * plane + idx] = d_xy;
      ssim_sum += ssim;
    }
  }
  const float2 sums = block_sum2_256(l1, ssim_sum, red);
  if (threadIdx.x == 0) {
    const int b = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    reinterpret_cast<float2*>(partials)[b] = sums;
  }
}

// fixed-order reduction of the per-workgroup partial sums (float2 {|x - y|, SSIM} per workgroup) ->
// out[0] = mean |x - y|, out[1] = mean SSIM
__global__ void __launch_bounds__(1024)
l1_ssim_reduce_kernel(int nb, const float2* __restrict__ partials, float inv_l1, float inv_ssim, float* __restrict__ out) {
  __shared__ float2 red[16];
  float a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < nb; i += 4096) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = i + u * 1024;
      const float2 v = j < nb ? partials[j] : make_float2(0.f, 0.f);
      a[u] += v.x;
      b[u] += v.y;
    }
  }
  float sa = (a[0] + a[1]) + (a[2] + a[3]), sb = (b[0] + b[1]) + (b[2] + b[3]);
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) {
    sa += __shfl_xor(sa, m);
    sb += __shfl_xor(sb, m);
  }
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = make_float2(sa, sb);
  __syncthreads();
  if (threadIdx.x == 0) {
    float ta = 0.f, tb = 0.f;
    for (int w = 0; w < 16; ++w) {
      ta += red[w].x;
      tb += red[w].y;
    }
    out[0] = ta * inv_l1;
    out[1] = tb * inv_ssim;
  }
}

__global__ void __launch_bounds__(256)
l1_ssim_bwd_kernel(int H, int W, int C, Window win, const float* __restrict__ pred, const float* __restrict__ gt,
                   const float* __restrict__ maps, const float* __restrict__ v_out, float inv_l1, float inv_ssim,
                   float* __restrict__ v_pred) {
  __shared__ float sm[3][INH][INWP];
  __shared__ float hm[3][INH][TWP];
  const int c = blockIdx.z, x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
  const int Hm = H - HALO, Wm = W - HALO;
  const size_t plane = (size_t)Hm * Wm;
  // map window: the map pixels q in [p - 10, p] of the tile's pixels p (zero outside the map: 'full' borders)
  for (int i = threadIdx.x; i < INH * INW; i += 256) {
    const int r = i / INW, q = i - r * INW;
    const int y = y0 - HALO + r, x = x0 - HALO + q;
    float d0 = 0.f, d1 = 0.f, d2 = 0.f;
    if (y >= 0 && y < Hm && x >= 0 && x < Wm) {
      const size_t o = (size_t)c * plane + (size_t)y * Wm + x;
      d0 = maps[o];
      d1 = maps[(size_t)C * plane + o];
      d2 = maps[2 * (size_t)C * plane + o];
    }
    sm[0][r][q] = d0;
    sm[1][r][q] = d1;
    sm[2][r][q] = d2;
  }
  __syncthreads();
  // tile column j holds map column x0 - 10 + j; pixel column x0 + q takes columns j = q .. q + 10 with weights
  // w(p - q) = g[10 - k] = g[k] (the window is symmetric); four neighbouring columns per thread
  if (threadIdx.x < INH * (TW / 4)) {
    const int r = threadIdx.x >> 3, q = (threadIdx.x & 7) * 4;
    float m[4][3];
#pragma unroll
    for (int o = 0; o < 4; ++o) m[o][0] = m[o][1] = m[o][2] = 0.f;
#pragma unroll
    for (int j = 0; j < WIN + 3; ++j) {
      const float d0 = sm[0][r][q + j], d1 = sm[1][r][q + j], d2 = sm[2][r][q + j];
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const int k = j - o;
        if (k >= 0 && k < WIN) {
          const float g = win.g[k];
          m[o][0] = fmaf(g, d0, m[o][0]);
          m[o][1] = fmaf(g, d1, m[o][1]);
          m[o][2] = fmaf(g, d2, m[o][2]);
        }
      }
    }
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int v = 0; v < 3; ++v) hm[v][r][q + o] = m[o][v];
  }
  __syncthreads();
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  float acc[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
#pragma unroll
  for (int j = 0; j < WIN + 1; ++j) {
    const float h0 = hm[0][2 * ty + j][tx], h1 = hm[1][2 * ty + j][tx], h2 = hm[2][2 * ty + j][tx];
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const int k = j - o;
      if (k >= 0 && k < WIN) {
        const float g = win.g[k];
        acc[o][0] = fmaf(g, h0, acc[o][0]);
        acc[o][1] = fmaf(g, h1, acc[o][1]);
        acc[o][2] = fmaf(g, h2, acc[o][2]);
      }
    }
  }
  const float v_l1 = v_out[0] * inv_l1, v_ssim = v_out[1] * inv_ssim;
#pragma unroll
  for (int o = 0; o < 2; ++o) {
    const int y = y0 + 2 * ty + o, x = x0 + tx;
    if (y < H && x < W) {
      const size_t idx = ((size_t)y * W + x) * C + c;
      const float a = pred[idx], b = gt[idx];
      const float sgn = a > b ? 1.f : (a < b ? -1.f : 0.f);  // d|b - a| / da
      v_pred[idx] = v_l1 * sgn + v_ssim * (acc[o][0] + 2.f * a * acc[o][1] + b * acc[o][2]);
    }
  }
}

bool bad_shape(int H, int W, int C) { return H <= HALO || W <= HALO || C <= 0 || C > 65535; }

}  // namespace

extern "C" size_t fg_l1_ssim_workspace_floats(int height, int width, int channels) {
  if (bad_shape(height, width, channels)) return 0;
  const size_t nb = (size_t)((width + TW - 1) / TW) * ((height + TH - 1) / TH) * channels;
  return 2 * nb;
}

extern "C" int fg_l1_ssim_fwd(int height, int width, int channels, const float* pred, const float* gt, float* maps,
                              float* workspace, size_t workspace_floats, float* out, fg_stream_t stream) {
  if (bad_shape(height, width, channels)) return FG_ERR_INVALID_ARG;
  if (!pred || !gt || !maps || !workspace || !out) return FG_ERR_INVALID_ARG;
  if (workspace_floats < fg_l1_ssim_workspace_floats(height, width, channels)) return FG_ERR_WORKSPACE;
  static const Window win = gaussian_window();
  const dim3 grid((width + TW - 1) / TW, (height + TH - 1) / TH, channels);
  if (grid.y > 65535) return FG_ERR_UNSUPPORTED;
  const int nb = (int)(grid.x * grid.y * grid.z);
  hipStream_t s = fg_hip_stream(stream);
  hipLaunchKernelGGL(l1_ssim_fwd_kernel, grid, dim3(256), 0, s, height, width, channels, win, pred, gt, maps, workspace);
  const float inv_l1 = (float)(1.0 / ((double)height * width * channels));
  const float inv_ssim = (float)(1.0 / ((double)(height - HALO) * (width - HALO) * channels));
  hipLaunchKernelGGL(l1_ssim_reduce_kernel, dim3(1), dim3(1024), 0, s, nb, reinterpret_cast<const float2*>(workspace), inv_l1,
                     inv_ssim, out);
  FG_RETURN_IF_LAUNCH_FAILED();
  return FG_OK;
}

extern "C" int fg_l1_ssim_bwd(int height, int width, int channels, const float* pred, const float* gt, const float* maps,
                              const float* v_out, float* v_pred, fg_stream_t stream) {
  if (bad_shape(height, width, channels)) return FG_ERR_INVALID_ARG;
  if (!pred || !gt || !maps || !v_out || !v_pred) return FG_ERR_INVALID_ARG;
  static const Window win = gaussian_window();
  const dim3 grid((width + TW - 1) / TW, (height + TH - 1) / TH, channels);
  if (grid.y > 65535) return FG_ERR_UNSUPPORTED;
  const float inv_l1 = (float)(1.0 / ((double)height * width * channels));
  const float inv_ssim = (float)(1.0 / ((double)(height - HALO) * (width - HALO) * channels));
  hipLaunchKernelGGL(l1_ssim_bwd_kernel, grid, dim3(256), 0, fg_hip_stream(stream), height, width, channels, win, pred, gt,
                     maps, v_out, inv_l1, inv_ssim, v_pred);
  FG_RETURN_IF_LAUNCH_FAILED();
  return FG_OK;
}
