// F: screen-space flow-derivative kernels.
//
//  * fg_camera_flow : per-pixel camera flow  A(x,y) v / Z + B(x,y) w
//        /root/reference preprocess/epipolar_flow.py:272-309 (pixel centres at integer
//        coordinates, :272; infinite depth -> 0, :315-317).
//  * fg_reprojection_flow : the exact-reprojection variant (F-spec')
//        /root/reference preprocess/epipolar_flow_bp.py:268-295: lift the pixel with the depth of
//        frame 0, map it by the 3x4 matrix M the host composed from the two poses, project with K,
//        divide by the depth map of frame 1, subtract the pixel.
//  * fg_flow_fwd/bwd : per-Gaussian projection-flow Jacobian (Lemma 1,
//        /root/reference docs/index.html:256-273), with the CODE's sign convention
//        A = [[fx,0,cx-x],[0,fy,cy-y]] (epipolar_flow.py:277-282), which is -1x the page's.
//        u_gs  = A(mu) vel / Z      (the per-Gaussian term composited by T_i alpha_i)
//        u_cam = A(mu) v   / Z + B(mu) w
//    Both are elementwise over Gaussians: HBM-bound, one lane per Gaussian.
#include "fg_common.h"

namespace {

struct Intr {
  float fx, fy, cx, cy;
};

__device__ __forceinline__ Intr load_intr(const float* __restrict__ K) { return {K[0], K[4], K[2], K[5]}; }

// B rows at (x, y), code convention (epipolar_flow.py:293-298)
__device__ __forceinline__ void make_B(const Intr& k, float x, float y, float (&B)[2][3]) {
  const float xc = x - k.cx, yc = y - k.cy;
  B[0][0] = -xc * yc / k.fy;
  B[0][1] = k.fx + xc * xc / k.fx;
  B[0][2] = -yc * k.fx / k.fy;
  B[1][0] = -k.fy - yc * yc / k.fy;
  B[1][1] = xc * yc / k.fx;
  B[1][2] = xc * k.fy / k.fx;
}

__global__ void __launch_bounds__(256)
camera_flow_kernel(int width, int height, const float* __restrict__ depth, const float* __restrict__ K,
                   const float* __restrict__ veloc, const float* __restrict__ omega, float* __restrict__ flow) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= (int64_t)width * height) return;
  const int iy = (int)(p / width), ix = (int)(p - (int64_t)iy * width);
  const Intr k = load_intr(K);
  const float x = (float)ix, y = (float)iy;
  const float Z = depth[p];
  float fu = 0.f, fv = 0.f;
  if (!isinf(Z)) {
    float B[2][3];
    make_B(k, x, y, B);
    const float v0 = veloc[0], v1 = veloc[1], v2 = veloc[2];
    const float w0 = omega[0], w1 = omega[1], w2 = omega[2];
    fu = (k.fx * v0 + (k.cx - x) * v2) / Z + (B[0][0] * w0 + B[0][1] * w1 + B[0][2] * w2);
    fv = (k.fy * v1 + (k.cy - y) * v2) / Z + (B[1][0] * w0 + B[1][1] * w1 + B[1][2] * w2);
  }
  reinterpret_cast<float2*>(flow)[p] = make_float2(fu, fv);
}

// uv - xy of epipolar_flow_bp.py:271-279; `sign` = -1 writes the reference's returned
// "sceneflow" (:295), +1 the raw difference; infinite depth0 -> 0 (:285-287).
__global__ void __launch_bounds__(256)
reprojection_flow_kernel(int width, int height, const float* __restrict__ depth0,
                         const float* __restrict__ depth1, const float* __restrict__ K,
                         const float* __restrict__ M, float sign, float* __restrict__ flow) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= (int64_t)width * height) return;
  const int iy = (int)(p / width), ix = (int)(p - (int64_t)iy * width);
  const Intr k = load_intr(K);
  const float x = (float)ix, y = (float)iy;
  const float Z = depth0[p];
  float fu = 0.f, fv = 0.f;
  if (!isinf(Z)) {
    // K^-1 (x, y, 1) Z
    const float X = (x - k.cx) / k.fx * Z, Y = (y - k.cy) / k.fy * Z;
    const float qx = M[0] * X + M[1] * Y + M[2] * Z + M[3];
    const float qy = M[4] * X + M[5] * Y + M[6] * Z + M[7];
    const float qz = M[8] * X + M[9] * Y + M[10] * Z + M[11];
    const float iz1 = 1.f / depth1[p];
    fu = sign * ((k.fx * qx + k.cx * qz) * iz1 - x);
    fv = sign * ((k.fy * qy + k.cy * qz) * iz1 - y);
  }
  reinterpret_cast<float2*>(flow)[p] = make_float2(fu, fv);
}

__global__ void __launch_bounds__(256)
flow_fwd_kernel(int N, const float* __restrict__ means2d, const float* __restrict__ depths,
                const int32_t* __restrict__ radii, const float* __restrict__ vel, const float* __restrict__ K,
                const float* __restrict__ veloc, const float* __restrict__ omega, float* __restrict__ u_gs,
                float* __restrict__ u_cam) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float g0 = 0.f, g1 = 0.f, c0 = 0.f, c1 = 0.f;
  if (!radii || radii[i] > 0) {
    const Intr k = load_intr(K);
    const float2 mu = reinterpret_cast<const float2*>(means2d)[i];
    const float iz = 1.f / depths[i];
    const float a02 = k.cx - mu.x, a12 = k.cy - mu.y;
    const float vx = vel[3 * i], vy = vel[3 * i + 1], vz = vel[3 * i + 2];
    g0 = (k.fx * vx + a02 * vz) * iz;
    g1 = (k.fy * vy + a12 * vz) * iz;
    float B[2][3];
    make_B(k, mu.x, mu.y, B);
    const float v0 = veloc[0], v1 = veloc[1], v2 = veloc[2];
    const float w0 = omega[0], w1 = omega[1], w2 = omega[2];
    c0 = (k.fx * v0 + a02 * v2) * iz + (B[0][0] * w0 + B[0][1] * w1 + B[0][2] * w2);
    c1 = (k.fy * v1 + a12 * v2) * iz + (B[1][0] * w0 + B[1][1] * w1 + B[1][2] * w2);
  }
  reinterpret_cast<float2*>(u_gs)[i] = make_float2(g0, g1);
  reinterpret_cast<float2*>(u_cam)[i] = make_float2(c0, c1);
}

__global__ void __launch_bounds__(256)
flow_bwd_kernel(int N, const float* __restrict__ means2d, const float* __restrict__ depths,
                const int32_t* __restrict__ radii, const float* __restrict__ vel, const float* __restrict__ K,
                const float* __restrict__ veloc, const float* __restrict__ omega,
                const float* __restrict__ v_u_gs, const float* __restrict__ v_u_cam,
                float* __restrict__ v_means2d, float* __restrict__ v_depths, float* __restrict__ v_vel) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float gx = 0.f, gy = 0.f, gz = 0.f, gv0 = 0.f, gv1 = 0.f, gv2 = 0.f;
  if (!radii || radii[i] > 0) {
    const Intr k = load_intr(K);
    const float2 mu = reinterpret_cast<const float2*>(means2d)[i];
    const float iz = 1.f / depths[i];
    const float a02 = k.cx - mu.x, a12 = k.cy - mu.y;
    const float vx = vel[3 * i], vy = vel[3 * i + 1], vz = vel[3 * i + 2];
    const float2 vg = reinterpret_cast<const float2*>(v_u_gs)[i];
    const float2 vc = reinterpret_cast<const float2*>(v_u_cam)[i];
    const float v0 = veloc[0], v1 = veloc[1], v2 = veloc[2];
    const float w0 = omega[0], w1 = omega[1], w2 = omega[2];
    // u_gs
    gv0 = vg.x * k.fx * iz;
    gv1 = vg.y * k.fy * iz;
    gv2 = (vg.x * a02 + vg.y * a12) * iz;
    gx = -vg.x * vz * iz;
    gy = -vg.y * vz * iz;
    const float n0 = k.fx * vx + a02 * vz, n1 = k.fy * vy + a12 * vz;  // u_gs = n * iz
    const float m0 = k.fx * v0 + a02 * v2, m1 = k.fy * v1 + a12 * v2;  // translational part of u_cam
    gz = -(vg.x * n0 + vg.y * n1 + vc.x * m0 + vc.y * m1) * iz * iz;
    // u_cam translational part wrt mu
    gx += -vc.x * v2 * iz;
    gy += -vc.y * v2 * iz;
    // rotational part: d(B w)/d(mu)
    const float xc = mu.x - k.cx, yc = mu.y - k.cy;
    const float dB0x = -yc / k.fy * w0 + 2.f * xc / k.fx * w1;
    const float dB0y = -xc / k.fy * w0 - k.fx / k.fy * w2;
    const float dB1x = yc / k.fx * w1 + k.fy / k.fx * w2;
    const float dB1y = -2.f * yc / k.fy * w0 + xc / k.fx * w1;
    gx += vc.x * dB0x + vc.y * dB1x;
    gy += vc.x * dB0y + vc.y * dB1y;
  }
  reinterpret_cast<float2*>(v_means2d)[i] = make_float2(gx, gy);
  v_depths[i] = gz;
  v_vel[3 * i] = gv0; v_vel[3 * i + 1] = gv1; v_vel[3 * i + 2] = gv2;
}

}  // namespace

extern "C" int fg_camera_flow(int width, int height, const float* depth, const float* K, const float* veloc,
                              const float* omega, float* flow, fg_stream_t stream) {
  if (width <= 0 || height <= 0 || !depth || !K || !veloc || !omega || !flow) return FG_ERR_INVALID_ARG;
  const int64_t P = (int64_t)width * height;
  hipLaunchKernelGGL(camera_flow_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, fg_hip_stream(stream),
                     width, height, depth, K, veloc, omega, flow);
  FG_RETURN_IF_LAUNCH_FAILED();
  return FG_OK;
}

extern "C" int fg_reprojection_flow(int width, int height, const float* depth0, const float* depth1,
                                    const float* K, const float* M, float sign, float* flow,
                                    fg_stream_t stream) {
  if (width <= 0 || height <= 0 || !depth0 || !depth1 || !K || !M || !flow) return FG_ERR_INVALID_ARG;
  const int64_t P = (int64_t)width * height;
  hipLaunchKernelGGL(reprojection_flow_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0,
                     fg_hip_stream(stream), width, height, depth0, depth1, K, M, sign, flow);
  FG_RETURN_IF_LAUNCH_FAILED();
  return FG_OK;
}

extern "C" int fg_flow_fwd(int N, const float* means2d, const float* depths, const int32_t* radii,
                           const float* vel, const float* K, const float* veloc, const float* omega,
                           float* u_gs, float* u_cam, fg_stream_t stream) {
  if (N < 0) return FG_ERR_INVALID_ARG;
  if (N == 0) return FG_OK;
  if (!means2d || !depths || !vel || !K || !veloc || !omega || !u_gs || !u_cam) return FG_ERR_INVALID_ARG;
  hipLaunchKernelGGL(flow_fwd_kernel, dim3((N + 255) / 256), dim3(256), 0, fg_hip_stream(stream), N, means2d,
                     depths, radii, vel, K, veloc, omega, u_gs, u_cam);
  FG_RETURN_IF_LAUNCH_FAILED();
  return FG_OK;
}

extern "C" int fg_flow_bwd(int N, const float* means2d, const float* depths, const int32_t* radii,
                           const float* vel, const float* K, const float* veloc, const float* omega,
                           const float* v_u_gs, const float* v_u_cam, float* v_means2d, float* v_depths,
                           float* v_vel, fg_stream_t stream) {
  if (N < 0) return FG_ERR_INVALID_ARG;
  if (N == 0) return FG_OK;
  if (!means2d || !depths || !vel || !K || !veloc || !omega || !v_u_gs || !v_u_cam || !v_means2d || !v_depths ||
      !v_vel)
    return FG_ERR_INVALID_ARG;
  hipLaunchKernelGGL(flow_bwd_kernel, dim3((N + 255) / 256), dim3(256), 0, fg_hip_stream(stream), N, means2d,
                     depths, radii, vel, K, veloc, omega, v_u_gs, v_u_cam, v_means2d, v_depths, v_vel);
  FG_RETURN_IF_LAUNCH_FAILED();
  return FG_OK;
}
