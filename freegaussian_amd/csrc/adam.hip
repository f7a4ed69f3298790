// A: one Adam update of a dense float32 parameter tensor in ONE launch (torch.optim.Adam's defaults: no weight decay,
// no amsgrad) -- the optimizers the reference's method config attaches to the six Gaussian parameter groups
// (/root/reference freegaussian/freegaussian_config.py, via nerfstudio's AdamOptimizerConfig).  torch's own step
// takes 1.42 ms per iteration for the 59 floats x 1M Gaussians of the bench scene on an MI355X
// (scripts/train_step_bench.py) -- as long as the whole render + loss + backward; by bytes (p, g, m, v read; p, m, v
// written: 28 B per element, 1.65 GB) it is a 0.3 ms job.
//
// Same operations in the same order as torch._single_tensor_adam (each in fp32, the step-dependent scalars computed by
// the host in double as torch does):
//   m += (g - m) * (1 - beta1);   v = v * beta2 + (1 - beta2) * g * g;
//   p += -step_size * (m / (sqrt(v) / sqrt(1 - beta2^t) + eps)),   step_size = lr / (1 - beta1^t)
#include "fg_common.h"

namespace {

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, float w1, float beta2, float w2, float neg_step,
                                      float bc2_sqrt, float eps) {
  m = m + w1 * (g - m);
  v = v * beta2 + (w2 * g) * g;
  const float denom = sqrtf(v) / bc2_sqrt + eps;
  p = p + neg_step * (m / denom);
}

__global__ void __launch_bounds__(256)
adam_kernel(long long n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
            float w1, float beta2, float w2, float neg_step, float bc2_sqrt, float eps) {
  const long long n4 = n >> 2;
  const long long stride = (long long)gridDim.x * blockDim.x;
  float4* p4 = reinterpret_cast<float4*>(p);
  const float4* g4 = reinterpret_cast<const float4*>(g);
  float4* m4 = reinterpret_cast<float4*>(m);
  float4* v4 = reinterpret_cast<float4*>(v);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 pp = p4[i], mm = m4[i], vv = v4[i];
    const float4 gg = g4[i];
    adam1(pp.x, gg.x, mm.x, vv.x, w1, beta2, w2, neg_step, bc2_sqrt, eps);
    adam1(pp.y, gg.y, mm.y, vv.y, w1, beta2, w2, neg_step, bc2_sqrt, eps);
    adam1(pp.z, gg.z, mm.z, vv.z, w1, beta2, w2, neg_step, bc2_sqrt, eps);
    adam1(pp.w, gg.w, mm.w, vv.w, w1, beta2, w2, neg_step, bc2_sqrt, eps);
    p4[i] = pp;
    m4[i] = mm;
    v4[i] = vv;
  }
  // the last n mod 4 elements
  const long long t = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) adam1(p[t], g[t], m[t], v[t], w1, beta2, w2, neg_step, bc2_sqrt, eps);
}

// up to FG_ADAM_MAX_TENSORS tensors in one launch: workgroups [first[i], first[i + 1]) walk tensor i
struct AdamDesc {
  float* p;
  const float* g;
  float* m;
  float* v;
  long long n;
  float w1, beta2, w2, neg_step, bc2_sqrt, eps;
};
struct AdamBatch {
  AdamDesc d[FG_ADAM_MAX_TENSORS];
  int first[FG_ADAM_MAX_TENSORS + 1];
  int count;
};

__global__ void __launch_bounds__(256) adam_multi_kernel(AdamBatch b) {
  int t = 0;
#pragma unroll 1
  while (t + 1 < b.count && (int)blockIdx.x >= b.first[t + 1]) ++t;
  const AdamDesc& d = b.d[t];
  const long long nblk = b.first[t + 1] - b.first[t], blk = (long long)blockIdx.x - b.first[t];
  const long long n4 = d.n >> 2, stride = nblk * 256;
  float4* p4 = reinterpret_cast<float4*>(d.p);
  const float4* g4 = reinterpret_cast<const float4*>(d.g);
  float4* m4 = reinterpret_cast<float4*>(d.m);
  float4* v4 = reinterpret_cast<float4*>(d.v);
  for (long long i = blk * 256 + threadIdx.x; i < n4; i += stride) {
    float4 pp = p4[i], mm = m4[i], vv = v4[i];
    const float4 gg = g4[i];
    adam1(pp.x, gg.x, mm.x, vv.x, d.w1, d.beta2, d.w2, d.neg_step, d.bc2_sqrt, d.eps);
    adam1(pp.y, gg.y, mm.y, vv.y, d.w1, d.beta2, d.w2, d.neg_step, d.bc2_sqrt, d.eps);
    adam1(pp.z, gg.z, mm.z, vv.z, d.w1, d.beta2, d.w2, d.neg_step, d.bc2_sqrt, d.eps);
    adam1(pp.w, gg.w, mm.w, vv.w, d.w1, d.beta2, d.w2, d.neg_step, d.bc2_sqrt, d.eps);
    p4[i] = pp;
    m4[i] = mm;
    v4[i] = vv;
  }
  const long long tail = (n4 << 2) + blk * 256 + threadIdx.x;
  if (tail < d.n) adam1(d.p[tail], d.g[tail], d.m[tail], d.v[tail], d.w1, d.beta2, d.w2, d.neg_step, d.bc2_sqrt, d.eps);
}

// the step-dependent scalars of one update, as torch computes them (doubles on the host); false = invalid arguments
bool adam_scalars(double lr, double beta1, double beta2, double eps, int64_t step, AdamDesc* d) {
  if (step < 1 || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0)) return false;
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  d->neg_step = (float)(-(lr / bc1));
  d->bc2_sqrt = (float)sqrt(bc2);
  d->w1 = (float)(1.0 - beta1);
  d->w2 = (float)(1.0 - beta2);
  d->beta2 = (float)beta2;
  d->eps = (float)eps;
  return true;
}

}  // namespace

extern "C" int fg_adam_step_multi(int count, const fg_adam_tensor* tensors, fg_stream_t stream) {
  if (count < 0 || (count > 0 && !tensors)) return FG_ERR_INVALID_ARG;
  for (int base = 0; base < count; base += FG_ADAM_MAX_TENSORS) {
    AdamBatch b{};  // (count 0, first[0] 0)
    for (int i = base; i < count && i < base + FG_ADAM_MAX_TENSORS; ++i) {
      const fg_adam_tensor& t = tensors[i];
      if (t.n < 0) return FG_ERR_INVALID_ARG;
      if (t.n == 0) continue;
      if (!t.param || !t.grad || !t.exp_avg || !t.exp_avg_sq) return FG_ERR_INVALID_ARG;
      if ((reinterpret_cast<uintptr_t>(t.param) | reinterpret_cast<uintptr_t>(t.grad) | reinterpret_cast<uintptr_t>(t.exp_avg) |
           reinterpret_cast<uintptr_t>(t.exp_avg_sq)) & 15)
        return FG_ERR_INVALID_ARG;
      AdamDesc& d = b.d[b.count];
      if (!adam_scalars(t.lr, t.beta1, t.beta2, t.eps, t.step, &d)) return FG_ERR_INVALID_ARG;
      d.p = t.param;
      d.g = t.grad;
      d.m = t.exp_avg;
      d.v = t.exp_avg_sq;
      d.n = t.n;
      long long blocks = ((t.n >> 2) + 255) / 256;
      if (blocks < 1) blocks = 1;
      if (blocks > 256 * 8) blocks = 256 * 8;
      b.first[b.count + 1] = b.first[b.count] + (int)blocks;
      ++b.count;
    }
    if (b.count == 0) continue;
    for (int i = b.count; i < FG_ADAM_MAX_TENSORS; ++i) b.first[i + 1] = b.first[b.count];
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)b.first[b.count]), dim3(256), 0, fg_hip_stream(stream), b);
  }
  FG_RETURN_IF_LAUNCH_FAILED();
  return FG_OK;
}

extern "C" int fg_adam_step(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, double lr,
                            double beta1, double beta2, double eps, int64_t step, fg_stream_t stream) {
  if (n < 0 || step < 1 || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0)) return FG_ERR_INVALID_ARG;
  if (n == 0) return FG_OK;
  if (!param || !grad || !exp_avg || !exp_avg_sq) return FG_ERR_INVALID_ARG;
  if ((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(exp_avg) |
       reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15)
    return FG_ERR_INVALID_ARG;  // (16-byte aligned arrays: float4 accesses)
  // torch: bias corrections and step size as Python floats (double), handed to fp32 kernels
  // (the hyper-parameters arrive as doubles, as torch holds them: 1 - 0.999 rounded to float from the double is
  // 1.3e-5 away from 1 - float(0.999))
  AdamDesc d{};
  if (!adam_scalars(lr, beta1, beta2, eps, step, &d)) return FG_ERR_INVALID_ARG;
  const long long n4 = n >> 2;
  long long blocks = (n4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 256 * 32) blocks = 256 * 32;  // grid-stride beyond 32 workgroups per CU
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, fg_hip_stream(stream), (long long)n, param, grad, exp_avg,
                     exp_avg_sq, d.w1, d.beta2, d.w2, d.neg_step, d.bc2_sqrt, d.eps);
  FG_RETURN_IF_LAUNCH_FAILED();
  return FG_OK;
}
