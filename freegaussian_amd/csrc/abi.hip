// ABI bookkeeping + a self-test entry for the transposing wave reduction.
#include "fg_common.h"

extern "C" int fg_abi_version(void) { return FG_ABI_VERSION; }

extern "C" const char* fg_error_string(int code) {
  switch (code) {
    case FG_OK: return "ok";
    case FG_ERR_INVALID_ARG: return "invalid argument";
    case FG_ERR_LAUNCH: return "HIP launch failed";
    case FG_ERR_WORKSPACE: return "workspace too small";
    case FG_ERR_UNSUPPORTED: return "unsupported configuration";
    default: return "unknown error";
  }
}

namespace {
// in[64][16] -> out[16]: out[i] = sum over lanes of in[lane][i], written by lane 4i.
__global__ void __launch_bounds__(64) debug_reduce16_kernel(const float* __restrict__ in, float* __restrict__ out) {
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = in[threadIdx.x * 16 + i];
  const float r = fg::wave_reduce16_transposed(v);
  if ((threadIdx.x & 3) == 0) out[threadIdx.x >> 2] = r;
  // every lane of a quad must agree
  out[16 + threadIdx.x] = r;
  out[80 + threadIdx.x] = fg::wave_sum(in[threadIdx.x * 16]);
  // 12-value variant on the first 12 columns
  float w[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) w[i] = in[threadIdx.x * 16 + i];
  const float r12 = fg::wave_reduce12_transposed(w);
  if (fg::wave_reduce12_owner(threadIdx.x)) out[144 + fg::wave_reduce12_index(threadIdx.x)] = r12;
  // LDS-transposing variant: 11 rows (8 + 3 channels), every lane of the owning quads reports
  __shared__ float red[16 * fg::FG_RED_STRIDE];
  float u[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) u[i] = in[threadIdx.x * 16 + i];
  const float r11 = fg::wave_reduce_rows_lds<11>(u, red, threadIdx.x);
  out[156 + threadIdx.x] = (threadIdx.x >> 2) < 11 ? r11 : -1.f;
}
}  // namespace

// Test hook (not part of the drop-in surface): in[64*16] floats, out[220] floats.
extern "C" int fg_debug_wave_reduce16(const float* in, float* out, fg_stream_t stream) {
  if (!in || !out) return FG_ERR_INVALID_ARG;
  hipLaunchKernelGGL(debug_reduce16_kernel, dim3(1), dim3(64), 0, fg_hip_stream(stream), in, out);
  FG_RETURN_IF_LAUNCH_FAILED();
  return FG_OK;
}
