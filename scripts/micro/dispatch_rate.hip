// What does it cost to DISPATCH workgroups that do (almost) nothing?  grid x block x static LDS sweeps.
// Build + run on the GPU box: hipcc -O3 --offload-arch=gfx950 scripts/micro/dispatch_rate.hip -o /tmp/dr && /tmp/dr
#include <hip/hip_runtime.h>
#include <cstdio>

template <int LDS_WORDS>
__global__ void empty_kernel(int* out, int flag) {
  __shared__ int lds[LDS_WORDS > 0 ? LDS_WORDS : 1];
  if (flag) {  // never taken: keeps the LDS array alive
    lds[threadIdx.x] = flag;
    __syncthreads();
    out[blockIdx.x] = lds[(threadIdx.x + 1) % blockDim.x];
  }
}

template <int LDS_WORDS>
void run(const char* name, int grid, int block, int* out) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(empty_kernel<LDS_WORDS>, dim3(grid), dim3(block), 0, 0, out, 0);
  hipEventRecord(a, 0);
  const int reps = 50;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(empty_kernel<LDS_WORDS>, dim3(grid), dim3(block), 0, 0, out, 0);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  printf("%-34s grid %6d block %4d lds %6d B : %7.2f us per launch, %6.1f ns per workgroup\n", name, grid, block,
         LDS_WORDS * 4, ms * 1e3 / reps, ms * 1e6 / reps / grid);
}

int main() {
  int* out;
  hipMalloc(&out, 1 << 20);
  run<0>("1 workgroup", 1, 64, out);
  run<0>("2400 x 64, no LDS", 2400, 64, out);
  run<0>("2400 x 256, no LDS", 2400, 256, out);
  run<5120>("2400 x 256, 20 KB LDS", 2400, 256, out);
  run<0>("2400 x 512, no LDS", 2400, 512, out);
  run<10240>("2400 x 512, 41 KB LDS", 2400, 512, out);
  run<0>("8160 x 256, no LDS", 8160, 256, out);
  run<5120>("8160 x 256, 20 KB LDS", 8160, 256, out);
  run<0>("16000 x 64, no LDS", 16000, 64, out);
  run<12288>("16000 x 64, 48 KB LDS", 16000, 64, out);
  run<0>("26000 x 64, no LDS", 26000, 64, out);
  run<5120>("768 x 512, 20 KB LDS", 768, 512, out);
  run<10240>("768 x 512, 41 KB LDS", 768, 512, out);
  return 0;
}
