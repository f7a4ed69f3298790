// Device-side math of the EWA projection (K1) and its backward (K7), shared by project.hip and
// preprocess.hip.  Translation units including this header must be compiled with
// -ffp-contract=off: the forward is the bit-exact part of the path (see project.hip).
#pragma once
#include "fg_common.h"

namespace fgp {

struct Cam {
  float W[3][3];
  float t[3];
  float fx, fy, cx, cy;
};

__device__ __forceinline__ Cam load_cam(const float* __restrict__ viewmat, const float* __restrict__ K) {
  Cam c;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) c.W[i][j] = viewmat[i * 4 + j];
    c.t[i] = viewmat[i * 4 + 3];
  }
  c.fx = K[0];
  c.fy = K[4];
  c.cx = K[2];
  c.cy = K[5];
  return c;
}

// Everything the forward computes that the backward needs again.
struct Fwd {
  float qn, w, x, y, z;  // quaternion norm and normalised components
  float R[3][3];
  float M[3][3];
  float CC[3][3];  // camera-space covariance (symmetric)
  float px, py, pz;  // camera-space mean
  float rz, rz2, tx, ty;
  bool in_x, in_y;  // FOV clamp inactive
  float ja, jb, jc, jd;
  float c00, c01, c11;  // blurred 2D covariance
  float det_orig, det;
  float m2x, m2y;
  float conic_a, conic_b, conic_c, comp;
  float radius_f;
};

__device__ __forceinline__ Fwd project_core(const Cam& cam, float mx, float my, float mz, float qw, float qx,
                                            float qy, float qz, float s0, float s1, float s2, int width,
                                            int height, float eps2d) {
#pragma clang fp contract(off)  // bit-exact forward whatever the translation unit's flags are
  Fwd f;
  const float(&W)[3][3] = cam.W;
  f.px = ((W[0][0] * mx + W[0][1] * my) + W[0][2] * mz) + cam.t[0];
  f.py = ((W[1][0] * mx + W[1][1] * my) + W[1][2] * mz) + cam.t[1];
  f.pz = ((W[2][0] * mx + W[2][1] * my) + W[2][2] * mz) + cam.t[2];

  f.qn = sqrtf(((qw * qw + qx * qx) + qy * qy) + qz * qz);
  const float w = qw / f.qn, x = qx / f.qn, y = qy / f.qn, z = qz / f.qn;
  f.w = w; f.x = x; f.y = y; f.z = z;
  const float x2 = x * x, y2 = y * y, z2 = z * z;
  const float xy = x * y, xz = x * z, yz = y * z;
  const float wx = w * x, wy = w * y, wz = w * z;
  f.R[0][0] = 1.0f - 2.0f * (y2 + z2); f.R[0][1] = 2.0f * (xy - wz); f.R[0][2] = 2.0f * (xz + wy);
  f.R[1][0] = 2.0f * (xy + wz); f.R[1][1] = 1.0f - 2.0f * (x2 + z2); f.R[1][2] = 2.0f * (yz - wx);
  f.R[2][0] = 2.0f * (xz - wy); f.R[2][1] = 2.0f * (yz + wx); f.R[2][2] = 1.0f - 2.0f * (x2 + y2);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    f.M[i][0] = f.R[i][0] * s0;
    f.M[i][1] = f.R[i][1] * s1;
    f.M[i][2] = f.R[i][2] * s2;
  }
  float C[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = i; j < 3; ++j) {
      C[i][j] = (f.M[i][0] * f.M[j][0] + f.M[i][1] * f.M[j][1]) + f.M[i][2] * f.M[j][2];
      C[j][i] = C[i][j];
    }
  float T[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) T[i][j] = (W[i][0] * C[0][j] + W[i][1] * C[1][j]) + W[i][2] * C[2][j];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = i; j < 3; ++j) {
      f.CC[i][j] = (T[i][0] * W[j][0] + T[i][1] * W[j][1]) + T[i][2] * W[j][2];
      f.CC[j][i] = f.CC[i][j];
    }

  const float tan_fovx = (0.5f * (float)width) / cam.fx;
  const float tan_fovy = (0.5f * (float)height) / cam.fy;
  const float lim_x = FG_FOV_CLAMP * tan_fovx;
  const float lim_y = FG_FOV_CLAMP * tan_fovy;
  f.rz = 1.0f / f.pz;
  f.rz2 = f.rz * f.rz;
  const float xr = f.px * f.rz, yr = f.py * f.rz;
  f.in_x = (xr <= lim_x) && (xr >= -lim_x);
  f.in_y = (yr <= lim_y) && (yr >= -lim_y);
  f.tx = f.pz * fminf(lim_x, fmaxf(-lim_x, xr));
  f.ty = f.pz * fminf(lim_y, fmaxf(-lim_y, yr));
  f.ja = cam.fx * f.rz;
  f.jb = -(cam.fx * f.tx) * f.rz2;
  f.jc = cam.fy * f.rz;
  f.jd = -(cam.fy * f.ty) * f.rz2;
  const float u0 = f.ja * f.CC[0][0] + f.jb * f.CC[0][2];
  const float u1 = f.ja * f.CC[0][1] + f.jb * f.CC[1][2];
  const float u2 = f.ja * f.CC[0][2] + f.jb * f.CC[2][2];
  const float w1 = f.jc * f.CC[1][1] + f.jd * f.CC[1][2];
  const float w2 = f.jc * f.CC[1][2] + f.jd * f.CC[2][2];
  float c00 = u0 * f.ja + u2 * f.jb;
  const float c01 = u1 * f.jc + u2 * f.jd;
  float c11 = w1 * f.jc + w2 * f.jd;
  f.m2x = (cam.fx * f.px) * f.rz + cam.cx;
  f.m2y = (cam.fy * f.py) * f.rz + cam.cy;

  f.det_orig = c00 * c11 - c01 * c01;
  c00 = c00 + eps2d;
  c11 = c11 + eps2d;
  f.det = c00 * c11 - c01 * c01;
  f.c00 = c00; f.c01 = c01; f.c11 = c11;
  f.comp = sqrtf(fmaxf(f.det_orig / f.det, 0.0f));
  const float inv_det = 1.0f / f.det;
  f.conic_a = c11 * inv_det;
  f.conic_b = -c01 * inv_det;
  f.conic_c = c00 * inv_det;
  const float b = 0.5f * (c00 + c11);
  const float v1 = b + sqrtf(fmaxf(b * b - f.det, 0.01f));
  f.radius_f = ceilf(3.0f * sqrtf(v1));
  return f;
}

// Backward of project_core for one visible Gaussian, first half: from the gradients w.r.t. means2d (vmx, vmy), depth
// (vdepth), the three conic values (vca, vcb_full, vcc) and the compensation (vcomp, used when with_comp) to the gradients
// w.r.t. the CAMERA-space mean (vp) and covariance (vCC, symmetric).  Everything in front of the world->camera transform:
// shared by the parameter gradients (project_backward) and the camera-pose gradient (viewmat.hip).
__device__ __forceinline__ void project_backward_camera(const Cam& cam, const Fwd& f, float eps2d, float vmx, float vmy,
                                                        float vdepth, float vca, float vcb_full, float vcc, bool with_comp,
                                                        float vcomp, float (&vp)[3], float (&vCC)[3][3]) {
  {
    // conic = inverse(blurred cov2d):  G_cov = -Q G_Q Q  with G_Q = [[va, vb/2],[vb/2, vc]]
    const float va = vca, vb = 0.5f * vcb_full, vc = vcc;
    const float qa = f.conic_a, qb = f.conic_b, qc = f.conic_c;
    // X = G_Q Q
    const float x00 = va * qa + vb * qb, x01 = va * qb + vb * qc;
    const float x10 = vb * qa + vc * qb, x11 = vb * qb + vc * qc;
    // G = -Q X   (symmetric)
    float g00 = -(qa * x00 + qb * x10);
    float g01 = -(qa * x01 + qb * x11);
    float g11 = -(qb * x01 + qc * x11);
    if (with_comp) {
      // comp = sqrt(det_orig/det), both functions of the unblurred covariance o = c - eps
      if (f.comp > 0.f && vcomp != 0.f) {
        const float o00 = f.c00 - eps2d, o11 = f.c11 - eps2d, o01 = f.c01;
        const float k = vcomp * 0.5f / f.comp / (f.det * f.det);
        g00 += k * (o11 * f.det - f.det_orig * f.c11);
        g11 += k * (o00 * f.det - f.det_orig * f.c00);
        // c01 appears twice in the symmetric matrix: half of d/d(o01) goes to each slot
        g01 += k * (o01 * (f.det_orig - f.det));
      }
    }
    // cov2d = J CC J^T with J = [[ja,0,jb],[0,jc,jd]]
    // v_CC = J^T G J
    {
      const float J[2][3] = {{f.ja, 0.f, f.jb}, {0.f, f.jc, f.jd}};
      const float G[2][2] = {{g00, g01}, {g01, g11}};
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          float acc = 0.f;
#pragma unroll
          for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 2; ++c) acc += J[r][a] * G[r][c] * J[c][b];
          vCC[a][b] = acc;
        }
    }
    // v_J = 2 G J CC
    float vJ[2][3];
    {
      const float J[2][3] = {{f.ja, 0.f, f.jb}, {0.f, f.jc, f.jd}};
      float JC[2][3];
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int b = 0; b < 3; ++b) JC[r][b] = J[r][0] * f.CC[0][b] + J[r][1] * f.CC[1][b] + J[r][2] * f.CC[2][b];
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        vJ[0][b] = 2.f * (g00 * JC[0][b] + g01 * JC[1][b]);
        vJ[1][b] = 2.f * (g01 * JC[0][b] + g11 * JC[1][b]);
      }
    }
    // camera-space mean gradient
    float vpx = cam.fx * f.rz * vmx;
    float vpy = cam.fy * f.rz * vmy;
    float vpz = -(cam.fx * f.px * vmx + cam.fy * f.py * vmy) * f.rz2 + vdepth;
    // through J: ja = fx rz; jc = fy rz; jb = -fx cl_x rz; jd = -fy cl_y rz (cl = clamp(p/z))
    vpz += -cam.fx * f.rz2 * vJ[0][0] - cam.fy * f.rz2 * vJ[1][1];
    {
      const float clx = f.tx * f.rz, cly = f.ty * f.rz;
      const float mxf = f.in_x ? 1.f : 0.f, myf = f.in_y ? 1.f : 0.f;
      vpx += -cam.fx * f.rz2 * mxf * vJ[0][2];
      vpz += cam.fx * f.rz2 * (clx + mxf * f.px * f.rz) * vJ[0][2];
      vpy += -cam.fy * f.rz2 * myf * vJ[1][2];
      vpz += cam.fy * f.rz2 * (cly + myf * f.py * f.rz) * vJ[1][2];
    }
    vp[0] = vpx; vp[1] = vpy; vp[2] = vpz;
  }
}

// Backward of project_core for one visible Gaussian.  Inputs: gradients w.r.t. means2d (vmx,vmy),
// depth (vdepth), the three conic values (vca,vcb,vcc) and the compensation (vcomp, used when
// with_comp).  Outputs are ADDED into g_m[3], g_q[4], g_s[3].
__device__ __forceinline__ void project_backward(const Cam& cam, const Fwd& f, const float (&s)[3], float eps2d,
                                                 float vmx, float vmy, float vdepth, float vca, float vcb_full,
                                                 float vcc, bool with_comp, float vcomp, float (&g_m)[3],
                                                 float (&g_q)[4], float (&g_s)[3]) {
  const float(&W)[3][3] = cam.W;
  {
    float vp[3], vCC[3][3];
    project_backward_camera(cam, f, eps2d, vmx, vmy, vdepth, vca, vcb_full, vcc, with_comp, vcomp, vp, vCC);
    const float vpx = vp[0], vpy = vp[1], vpz = vp[2];
    // world mean: p = W m + t
#pragma unroll
    for (int j = 0; j < 3; ++j) g_m[j] += W[0][j] * vpx + W[1][j] * vpy + W[2][j] * vpz;
    // world covariance: CC = W C W^T  ->  v_C = W^T v_CC W
    float vC[3][3];
    {
      float tmp[3][3];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) tmp[a][b] = vCC[a][0] * W[0][b] + vCC[a][1] * W[1][b] + vCC[a][2] * W[2][b];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) vC[a][b] = W[0][a] * tmp[0][b] + W[1][a] * tmp[1][b] + W[2][a] * tmp[2][b];
    }
    // C = M M^T -> v_M = 2 v_C M (v_C symmetric); M = R diag(s)
    float vR[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const float vM = 2.f * (vC[a][0] * f.M[0][b] + vC[a][1] * f.M[1][b] + vC[a][2] * f.M[2][b]);
        g_s[b] += vM * f.R[a][b];
        vR[a][b] = vM * s[b];
      }
    // rotation -> normalised quaternion
    const float w = f.w, x = f.x, y = f.y, z = f.z;
    const float vqw = 2.f * (x * (vR[2][1] - vR[1][2]) + y * (vR[0][2] - vR[2][0]) + z * (vR[1][0] - vR[0][1]));
    const float vqx = 2.f * (-2.f * x * (vR[1][1] + vR[2][2]) + y * (vR[0][1] + vR[1][0]) +
                             z * (vR[0][2] + vR[2][0]) + w * (vR[2][1] - vR[1][2]));
    const float vqy = 2.f * (x * (vR[0][1] + vR[1][0]) - 2.f * y * (vR[0][0] + vR[2][2]) +
                             z * (vR[1][2] + vR[2][1]) + w * (vR[0][2] - vR[2][0]));
    const float vqz = 2.f * (x * (vR[0][2] + vR[2][0]) + y * (vR[1][2] + vR[2][1]) -
                             2.f * z * (vR[0][0] + vR[1][1]) + w * (vR[1][0] - vR[0][1]));
    // normalisation q_hat = q / |q|
    const float dotp = vqw * w + vqx * x + vqy * y + vqz * z;
    const float inv = 1.f / f.qn;
    g_q[0] += (vqw - dotp * w) * inv;
    g_q[1] += (vqx - dotp * x) * inv;
    g_q[2] += (vqy - dotp * y) * inv;
    g_q[3] += (vqz - dotp * z) * inv;
  }
}

}  // namespace fgp
