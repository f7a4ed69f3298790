// Micro-benchmark: returning global atomicAdd throughput on T counters with the access pattern
// of a tile scatter (lanes of a wave hit short runs of consecutive counters; runs are random).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void scatter_atomics(int64_t n, const int* __restrict__ tiles, int* __restrict__ cursor, int* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int t = tiles[i];
  const int pos = atomicAdd(&cursor[t], 1);
  out[i] = pos;
}
__global__ void plain_atomics(int64_t n, const int* __restrict__ tiles, int* __restrict__ cursor) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  atomicAdd(&cursor[tiles[i]], 1);  // non-returning
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 7200000;
  const int T = argc > 2 ? atoi(argv[2]) : 8160;
  const int tw = argc > 3 ? atoi(argv[3]) : 120;
  std::vector<int> h(n);
  srand(1);
  int64_t i = 0;
  while (i < n) {  // a "Gaussian": a 3x3 rect of tiles at a random place, row-major
    const int x0 = rand() % (tw - 3), y0 = rand() % (T / tw - 3);
    for (int y = 0; y < 3 && i < n; ++y)
      for (int x = 0; x < 3 && i < n; ++x) h[i++] = (y0 + y) * tw + x0 + x;
  }
  int *d_t, *d_c, *d_o;
  hipMalloc(&d_t, n * 4); hipMalloc(&d_c, T * 4); hipMalloc(&d_o, n * 4);
  hipMemcpy(d_t, h.data(), n * 4, hipMemcpyHostToDevice);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int mode = 0; mode < 2; ++mode) {
    float best = 1e9;
    for (int r = 0; r < 5; ++r) {
      hipMemset(d_c, 0, T * 4);
      hipEventRecord(a);
      if (mode == 0) hipLaunchKernelGGL(scatter_atomics, dim3((n + 255) / 256), dim3(256), 0, 0, n, d_t, d_c, d_o);
      else hipLaunchKernelGGL(plain_atomics, dim3((n + 255) / 256), dim3(256), 0, 0, n, d_t, d_c);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (ms < best) best = ms;
    }
    printf("%s atomics: n=%lld T=%d -> %.3f ms = %.1f G atomics/s\n", mode == 0 ? "returning" : "non-returning", (long long)n, T,
           best, n / best / 1e6);
  }
  return 0;
}
