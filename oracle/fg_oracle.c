/*
 * fg_oracle.c -- plain-C CPU restatement of the Gaussian-raster hot path.
 * TEST INFRASTRUCTURE ONLY: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this.  PARITY UNPINNED w.r.t. the reference's own rasterizer (gsplat, un-vendored,
 * CUDA-only; see oracle/raster_oracle.py header): it restates the published algorithm behind
 * the call at /root/reference freegaussian/freegaussian_model.py:847-868 with the constants the
 * call sites state (tile 16 :806, near/far :859-860, blur 0.3 :110-119).
 *
 * Written independently of both the HIP kernels and the PyTorch oracle, scalar and sequential,
 * in the order a per-pixel loop naturally has.  Compile with -ffp-contract=off: projection
 * results must then agree BIT FOR BIT with oracle/raster_oracle.py (checked in
 * tests/test_oracle.py), which pins the evaluation order the integer path depends on.
 *
 * Build: make -C oracle   ->  oracle/libfg_oracle.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ALPHA_SKIP (1.0f / 255.0f)
#define ALPHA_MAX 0.999f
#define T_STOP 1e-4f

/* ---- K1 ---------------------------------------------------------------------------------- */
/* returns number of visible Gaussians */
int fgo_project(int N, const float* means, const float* quats, const float* scales, const float* vm,
                const float* K, int width, int height, float eps2d, float near_plane, float far_plane,
                float radius_clip, int32_t* radii, float* means2d, float* depths, float* conics,
                float* comps) {
  const float fx = K[0], fy = K[4], cx = K[2], cy = K[5];
  const float W[3][3] = {{vm[0], vm[1], vm[2]}, {vm[4], vm[5], vm[6]}, {vm[8], vm[9], vm[10]}};
  const float t[3] = {vm[3], vm[7], vm[11]};
  int visible = 0;
  for (int i = 0; i < N; ++i) {
    const float mx = means[3 * i], my = means[3 * i + 1], mz = means[3 * i + 2];
    float p[3];
    for (int r = 0; r < 3; ++r) p[r] = ((W[r][0] * mx + W[r][1] * my) + W[r][2] * mz) + t[r];
    float qw = quats[4 * i], qx = quats[4 * i + 1], qy = quats[4 * i + 2], qz = quats[4 * i + 3];
    const float qn = sqrtf(((qw * qw + qx * qx) + qy * qy) + qz * qz);
    qw /= qn; qx /= qn; qy /= qn; qz /= qn;
    const float x2 = qx * qx, y2 = qy * qy, z2 = qz * qz, xy = qx * qy, xz = qx * qz, yz = qy * qz;
    const float wx = qw * qx, wy = qw * qy, wz = qw * qz;
    const float R[3][3] = {{1.0f - 2.0f * (y2 + z2), 2.0f * (xy - wz), 2.0f * (xz + wy)},
                           {2.0f * (xy + wz), 1.0f - 2.0f * (x2 + z2), 2.0f * (yz - wx)},
                           {2.0f * (xz - wy), 2.0f * (yz + wx), 1.0f - 2.0f * (x2 + y2)}};
    float M[3][3], C[3][3], T[3][3], CC[3][3];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) M[r][c] = R[r][c] * scales[3 * i + c];
    for (int r = 0; r < 3; ++r)
      for (int c = r; c < 3; ++c) {
        C[r][c] = (M[r][0] * M[c][0] + M[r][1] * M[c][1]) + M[r][2] * M[c][2];
        C[c][r] = C[r][c];
      }
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) T[r][c] = (W[r][0] * C[0][c] + W[r][1] * C[1][c]) + W[r][2] * C[2][c];
    for (int r = 0; r < 3; ++r)
      for (int c = r; c < 3; ++c) {
        CC[r][c] = (T[r][0] * W[c][0] + T[r][1] * W[c][1]) + T[r][2] * W[c][2];
        CC[c][r] = CC[r][c];
      }
    const float tan_fovx = (0.5f * (float)width) / fx, tan_fovy = (0.5f * (float)height) / fy;
    const float lim_x = 1.3f * tan_fovx, lim_y = 1.3f * tan_fovy;
    const float rz = 1.0f / p[2], rz2 = rz * rz;
    const float tx = p[2] * fminf(lim_x, fmaxf(-lim_x, p[0] * rz));
    const float ty = p[2] * fminf(lim_y, fmaxf(-lim_y, p[1] * rz));
    const float ja = fx * rz, jb = -(fx * tx) * rz2, jc = fy * rz, jd = -(fy * ty) * rz2;
    const float u0 = ja * CC[0][0] + jb * CC[0][2];
    const float u1 = ja * CC[0][1] + jb * CC[1][2];
    const float u2 = ja * CC[0][2] + jb * CC[2][2];
    const float w1 = jc * CC[1][1] + jd * CC[1][2];
    const float w2 = jc * CC[1][2] + jd * CC[2][2];
    float c00 = u0 * ja + u2 * jb;
    const float c01 = u1 * jc + u2 * jd;
    float c11 = w1 * jc + w2 * jd;
    const float m2x = (fx * p[0]) * rz + cx, m2y = (fy * p[1]) * rz + cy;
    const float det_orig = c00 * c11 - c01 * c01;
    c00 = c00 + eps2d;
    c11 = c11 + eps2d;
    const float det = c00 * c11 - c01 * c01;
    const float comp = sqrtf(fmaxf(det_orig / det, 0.0f));
    const float inv_det = 1.0f / det;
    const float b = 0.5f * (c00 + c11);
    const float v1 = b + sqrtf(fmaxf(b * b - det, 0.01f));
    const float rad = ceilf(3.0f * sqrtf(v1));
    int ok = (p[2] >= near_plane) && (p[2] <= far_plane) && (det > 0.0f) && isfinite(rad) && (rad > radius_clip);
    ok = ok && !((m2x + rad <= 0.0f) || (m2x - rad >= (float)width) || (m2y + rad <= 0.0f) ||
                 (m2y - rad >= (float)height));
    if (ok) {
      radii[i] = (int32_t)rad;
      means2d[2 * i] = m2x; means2d[2 * i + 1] = m2y;
      depths[i] = p[2];
      conics[3 * i] = c11 * inv_det; conics[3 * i + 1] = -c01 * inv_det; conics[3 * i + 2] = c00 * inv_det;
      if (comps) comps[i] = comp;
      ++visible;
    } else {
      radii[i] = 0;
      means2d[2 * i] = means2d[2 * i + 1] = 0.0f;
      depths[i] = 0.0f;
      conics[3 * i] = conics[3 * i + 1] = conics[3 * i + 2] = 0.0f;
      if (comps) comps[i] = 0.0f;
    }
  }
  return visible;
}

/* ---- K3 / K4 ------------------------------------------------------------------------------ */
static void tile_rect(float mx, float my, int radius, int ts, int tw, int th, int* x0, int* y0, int* x1,
                      int* y1) {
  const float fts = (float)ts, r = (float)radius / fts, tx = mx / fts, ty = my / fts;
  int a = (int)floorf(tx - r), b = (int)ceilf(tx + r), c = (int)floorf(ty - r), d = (int)ceilf(ty + r);
  *x0 = a < 0 ? 0 : (a > tw ? tw : a);
  *x1 = b < 0 ? 0 : (b > tw ? tw : b);
  *y0 = c < 0 ? 0 : (c > th ? th : c);
  *y1 = d < 0 ? 0 : (d > th ? th : d);
}

int64_t fgo_count_isects(int N, const float* means2d, const int32_t* radii, int ts, int tw, int th,
                         int32_t* tiles_per_gauss) {
  int64_t total = 0;
  for (int i = 0; i < N; ++i) {
    int n = 0;
    if (radii[i] > 0) {
      int x0, y0, x1, y1;
      tile_rect(means2d[2 * i], means2d[2 * i + 1], radii[i], ts, tw, th, &x0, &y0, &x1, &y1);
      n = (x1 - x0) * (y1 - y0);
    }
    tiles_per_gauss[i] = n;
    total += n;
  }
  return total;
}

/* emits, then stable LSD radix sort (16-bit digits) on the full 64-bit key */
void fgo_isect_sorted(int N, const float* means2d, const int32_t* radii, const float* depths, int ts, int tw,
                      int th, int64_t n_isects, int64_t* keys, int32_t* vals, int do_sort) {
  int64_t cur = 0;
  for (int i = 0; i < N; ++i) {
    if (radii[i] <= 0) continue;
    int x0, y0, x1, y1;
    tile_rect(means2d[2 * i], means2d[2 * i + 1], radii[i], ts, tw, th, &x0, &y0, &x1, &y1);
    uint32_t db;
    memcpy(&db, &depths[i], 4);
    for (int y = y0; y < y1; ++y)
      for (int x = x0; x < x1; ++x) {
        keys[cur] = ((int64_t)(y * tw + x) << 32) | (int64_t)db;
        vals[cur] = i;
        ++cur;
      }
  }
  if (!do_sort || n_isects < 2) return;
  int64_t* k2 = (int64_t*)malloc(sizeof(int64_t) * n_isects);
  int32_t* v2 = (int32_t*)malloc(sizeof(int32_t) * n_isects);
  int64_t* cnt = (int64_t*)malloc(sizeof(int64_t) * 65537);
  int64_t *ka = keys, *kb = k2;
  int32_t *va = vals, *vb = v2;
  for (int pass = 0; pass < 4; ++pass) {
    const int sh = pass * 16;
    memset(cnt, 0, sizeof(int64_t) * 65537);
    for (int64_t j = 0; j < n_isects; ++j) cnt[(((uint64_t)ka[j]) >> sh & 0xFFFF) + 1]++;
    for (int d = 0; d < 65536; ++d) cnt[d + 1] += cnt[d];
    for (int64_t j = 0; j < n_isects; ++j) {
      const int64_t pos = cnt[((uint64_t)ka[j]) >> sh & 0xFFFF]++;
      kb[pos] = ka[j];
      vb[pos] = va[j];
    }
    int64_t* tk = ka; ka = kb; kb = tk;
    int32_t* tv = va; va = vb; vb = tv;
  }
  /* 4 passes: result is back in keys/vals */
  free(k2); free(v2); free(cnt);
}

void fgo_tile_offsets(int64_t n, const int64_t* sorted_keys, int n_tiles, int32_t* offsets) {
  int64_t j = 0;
  for (int t = 0; t <= n_tiles; ++t) {
    while (j < n && (sorted_keys[j] >> 32) < t) ++j;
    offsets[t] = (int32_t)j;
  }
}

/* ---- K5 ----------------------------------------------------------------------------------- */
void fgo_raster_fwd(int C, int width, int height, int ts, const float* means2d, const float* conics,
                    const float* feats, const float* opac, const int32_t* offsets, const int32_t* ids,
                    float* render, float* alphas, int32_t* last_ids) {
  const int tw = (width + ts - 1) / ts;
#pragma omp parallel for schedule(dynamic, 8)
  for (int y = 0; y < height; ++y)
    for (int x = 0; x < width; ++x) {
      const int tile = (y / ts) * tw + x / ts;
      const int s = offsets[tile], e = offsets[tile + 1];
      const float px = (float)x + 0.5f, py = (float)y + 0.5f;
      float T = 1.0f, acc[8] = {0};
      int last = s - 1;
      for (int j = s; j < e; ++j) {
        const int g = ids[j];
        const float dx = means2d[2 * g] - px, dy = means2d[2 * g + 1] - py;
        const float sigma = 0.5f * (conics[3 * g] * dx * dx + conics[3 * g + 2] * dy * dy) + conics[3 * g + 1] * dx * dy;
        float alpha = opac[g] * expf(-sigma);
        if (alpha > ALPHA_MAX) alpha = ALPHA_MAX;
        if (sigma < 0.0f || alpha < ALPHA_SKIP) continue;
        const float nT = T * (1.0f - alpha);
        if (nT <= T_STOP) break;
        const float vis = alpha * T;
        for (int c = 0; c < C; ++c) acc[c] += feats[(size_t)g * C + c] * vis;
        last = j;
        T = nT;
      }
      const size_t pix = (size_t)y * width + x;
      for (int c = 0; c < C; ++c) render[pix * C + c] = acc[c];
      alphas[pix] = 1.0f - T;
      last_ids[pix] = last;
    }
}

/* ---- K6: per-pixel reverse traversal; gradients accumulated in double then stored as float
 * (sequential over pixels, so the result does not depend on thread scheduling) -------------- */
void fgo_raster_bwd(int N, int C, int width, int height, int ts, const float* means2d, const float* conics,
                    const float* feats, const float* opac, const int32_t* offsets, const int32_t* ids,
                    const float* alphas, const int32_t* last_ids, const float* v_render, const float* v_alphas,
                    float* v_means2d, float* v_abs, float* v_conics, float* v_feats, float* v_opac) {
  const int tw = (width + ts - 1) / ts;
  double* g_xy = (double*)calloc((size_t)N * 2, sizeof(double));
  double* g_ab = (double*)calloc((size_t)N * 2, sizeof(double));
  double* g_co = (double*)calloc((size_t)N * 3, sizeof(double));
  double* g_ft = (double*)calloc((size_t)N * C, sizeof(double));
  double* g_op = (double*)calloc((size_t)N, sizeof(double));
  for (int y = 0; y < height; ++y)
    for (int x = 0; x < width; ++x) {
      const int tile = (y / ts) * tw + x / ts;
      const int s = offsets[tile];
      const size_t pix = (size_t)y * width + x;
      const float px = (float)x + 0.5f, py = (float)y + 0.5f;
      const float T_final = 1.0f - alphas[pix];
      float T = T_final, buf[8] = {0};
      const float* vr = v_render + pix * C;
      const float va = v_alphas[pix];
      for (int j = last_ids[pix]; j >= s; --j) {
        const int g = ids[j];
        const float dx = means2d[2 * g] - px, dy = means2d[2 * g + 1] - py;
        const float ca = conics[3 * g], cb = conics[3 * g + 1], cc = conics[3 * g + 2];
        const float sigma = 0.5f * (ca * dx * dx + cc * dy * dy) + cb * dx * dy;
        const float vis = expf(-sigma);
        float alpha = opac[g] * vis;
        if (alpha > ALPHA_MAX) alpha = ALPHA_MAX;
        if (sigma < 0.0f || alpha < ALPHA_SKIP) continue;
        const float ra = 1.0f / (1.0f - alpha);
        T *= ra;
        const float fac = alpha * T;
        float v_alpha = 0.0f;
        for (int c = 0; c < C; ++c) {
          const float f = feats[(size_t)g * C + c];
          g_ft[(size_t)g * C + c] += fac * vr[c];
          v_alpha += (f * T - buf[c] * ra) * vr[c];
          buf[c] += f * fac;
        }
        v_alpha += T_final * ra * va;
        if (opac[g] * vis <= ALPHA_MAX) {
          const float v_sigma = -opac[g] * vis * v_alpha;
          g_co[3 * g] += 0.5f * v_sigma * dx * dx;
          g_co[3 * g + 1] += v_sigma * dx * dy;
          g_co[3 * g + 2] += 0.5f * v_sigma * dy * dy;
          const float gx = v_sigma * (ca * dx + cb * dy), gy = v_sigma * (cb * dx + cc * dy);
          g_xy[2 * g] += gx; g_xy[2 * g + 1] += gy;
          g_ab[2 * g] += fabsf(gx); g_ab[2 * g + 1] += fabsf(gy);
          g_op[g] += vis * v_alpha;
        }
      }
    }
  for (size_t i = 0; i < (size_t)N * 2; ++i) { v_means2d[i] = (float)g_xy[i]; v_abs[i] = (float)g_ab[i]; }
  for (size_t i = 0; i < (size_t)N * 3; ++i) v_conics[i] = (float)g_co[i];
  for (size_t i = 0; i < (size_t)N * C; ++i) v_feats[i] = (float)g_ft[i];
  for (size_t i = 0; i < (size_t)N; ++i) v_opac[i] = (float)g_op[i];
  free(g_xy); free(g_ab); free(g_co); free(g_ft); free(g_op);
}
