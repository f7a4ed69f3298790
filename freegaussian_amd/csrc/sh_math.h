// Device-side math of the real spherical-harmonics colour (K2/K8), shared by sh.hip and
// preprocess.hip: 3DGS basis / sign convention, degree 0..3.
#pragma once
#include "fg_common.h"

namespace fgsh {

constexpr float C0 = 0.28209479177387814f;
constexpr float C1 = 0.4886025119029199f;
constexpr float C2_0 = 1.0925484305920792f, C2_1 = -1.0925484305920792f, C2_2 = 0.31539156525252005f,
                C2_3 = -1.0925484305920792f, C2_4 = 0.5462742152960396f;
constexpr float C3_0 = -0.5900435899266435f, C3_1 = 2.890611442640554f, C3_2 = -0.4570457994644658f,
                C3_3 = 0.3731763325901154f, C3_4 = -0.4570457994644658f, C3_5 = 1.445305721320277f,
                C3_6 = -0.5900435899266435f;

#ifndef FG_PRE_BLOCK
// Gaussians per workgroup of the per-Gaussian passes.  One wavefront: each workgroup alternates a load phase
// (its slab streams in), arithmetic and a store phase, and the more independent workgroups a CU holds
// (12 KB of LDS each) the more of them are in their load phase at any moment.  MI355X, 1M Gaussians:
// preprocess fwd / bwd 0.0721 / 0.1097 ms at 256, 0.0701 / 0.1068 at 128, 0.0687 / 0.1055 at 64; 0.106 / 0.154 at 512
#define FG_PRE_BLOCK 64
#endif
constexpr int BLOCK = FG_PRE_BLOCK;
constexpr int ROW = 49;  // 48 floats (16 bases x 3) + 1 pad -> odd stride, conflict-free

// camera position = -W^-1 t for the 3x4 world->camera transform
__device__ __forceinline__ void camera_position(const float* __restrict__ vm, float& cx, float& cy, float& cz) {
#pragma clang fp contract(off)  // (the same bits wherever it is inlined, like sh_basis below)
  const float a = vm[0], b = vm[1], c = vm[2], d = vm[4], e = vm[5], f = vm[6], g = vm[8], h = vm[9], i = vm[10];
  const float tx = vm[3], ty = vm[7], tz = vm[11];
  const float A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
  const float det = a * A + b * B + c * C;
  const float id = 1.f / det;
  // inverse = adj / det
  const float i00 = A * id, i01 = -(b * i - c * h) * id, i02 = (b * f - c * e) * id;
  const float i10 = B * id, i11 = (a * i - c * g) * id, i12 = -(a * f - c * d) * id;
  const float i20 = C * id, i21 = -(a * h - b * g) * id, i22 = (a * e - b * d) * id;
  cx = -(i00 * tx + i01 * ty + i02 * tz);
  cy = -(i10 * tx + i11 * ty + i12 * tz);
  cz = -(i20 * tx + i21 * ty + i22 * tz);
}

// unit view direction of a mean and 1 / distance
__device__ __forceinline__ void view_dir(const float* __restrict__ vm, float mx, float my, float mz, float& dx, float& dy,
                                         float& dz, float& inv) {
#pragma clang fp contract(off)
  float cx, cy, cz;
  camera_position(vm, cx, cy, cz);
  const float ux = mx - cx, uy = my - cy, uz = mz - cz;
  inv = 1.f / sqrtf(ux * ux + uy * uy + uz * uz);
  dx = ux * inv; dy = uy * inv; dz = uz * inv;
}

// basis values b[k] for k < (degree+1)^2.  No FMA contraction in here: where the compiler fuses a multiply into
// an add depends on the code around the call (2 z^2 - x^2 - y^2 next to the gradient's x^2, y^2 ...), and the fused
// and the stage-by-stage paths must give the same bits (tests: torch.equal).
__device__ __forceinline__ void sh_basis(int degree, float x, float y, float z, float (&b)[16]) {
#pragma clang fp contract(off)
  b[0] = C0;
  if (degree > 0) {
    b[1] = -C1 * y; b[2] = C1 * z; b[3] = -C1 * x;
  }
  if (degree > 1) {
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    b[4] = C2_0 * xy; b[5] = C2_1 * yz; b[6] = C2_2 * (2.f * zz - xx - yy); b[7] = C2_3 * xz;
    b[8] = C2_4 * (xx - yy);
    if (degree > 2) {
      b[9] = C3_0 * y * (3.f * xx - yy);
      b[10] = C3_1 * xy * z;
      b[11] = C3_2 * y * (4.f * zz - xx - yy);
      b[12] = C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy);
      b[13] = C3_4 * x * (4.f * zz - xx - yy);
      b[14] = C3_5 * z * (xx - yy);
      b[15] = C3_6 * x * (xx - 3.f * yy);
    }
  }
}

// colour (before + 0.5 / clamp) = sum_k basis_k coeff_k, one explicit FMA per term in the order of the bases: the
// same bits wherever it is inlined.  row: [16 x 3] coefficients of one Gaussian.
__device__ __forceinline__ void sh_dot(const float (&basis)[16], const float* row, int kk, float& r, float& g, float& b) {
#pragma clang fp contract(off)
  r = g = b = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    if (k < kk) {
      r = __builtin_fmaf(basis[k], row[3 * k], r);
      g = __builtin_fmaf(basis[k], row[3 * k + 1], g);
      b = __builtin_fmaf(basis[k], row[3 * k + 2], b);
    }
  }
}

// d basis / d(x,y,z)
__device__ __forceinline__ void sh_basis_grad(int degree, float x, float y, float z, float (&dx)[16],
                                              float (&dy)[16], float (&dz)[16]) {
#pragma unroll
  for (int k = 0; k < 16; ++k) dx[k] = dy[k] = dz[k] = 0.f;
  if (degree > 0) {
    dy[1] = -C1; dz[2] = C1; dx[3] = -C1;
  }
  if (degree > 1) {
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    dx[4] = C2_0 * y; dy[4] = C2_0 * x;
    dy[5] = C2_1 * z; dz[5] = C2_1 * y;
    dx[6] = -2.f * C2_2 * x; dy[6] = -2.f * C2_2 * y; dz[6] = 4.f * C2_2 * z;
    dx[7] = C2_3 * z; dz[7] = C2_3 * x;
    dx[8] = 2.f * C2_4 * x; dy[8] = -2.f * C2_4 * y;
    if (degree > 2) {
      dx[9] = C3_0 * 6.f * xy; dy[9] = C3_0 * (3.f * xx - 3.f * yy);
      dx[10] = C3_1 * yz; dy[10] = C3_1 * xz; dz[10] = C3_1 * xy;
      dx[11] = C3_2 * (-2.f * xy); dy[11] = C3_2 * (4.f * zz - xx - 3.f * yy); dz[11] = C3_2 * 8.f * yz;
      dx[12] = C3_3 * (-6.f * xz); dy[12] = C3_3 * (-6.f * yz); dz[12] = C3_3 * (6.f * zz - 3.f * xx - 3.f * yy);
      dx[13] = C3_4 * (4.f * zz - 3.f * xx - yy); dy[13] = C3_4 * (-2.f * xy); dz[13] = C3_4 * 8.f * xz;
      dx[14] = C3_5 * 2.f * xz; dy[14] = C3_5 * (-2.f * yz); dz[14] = C3_5 * (xx - yy);
      dx[15] = C3_6 * (3.f * xx - 3.f * yy); dy[15] = C3_6 * (-6.f * xy);
    }
  }
}

}  // namespace fgsh
