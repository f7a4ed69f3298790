// Micro-benchmark: what does a grid-wide barrier cost on MI355X next to a kernel boundary?
//   a) K back-to-back launches of a kernel in which every workgroup does one dependent global
//      read-modify-write (the shape of a radix-sort phase at 1M keys: all latency, no bandwidth)
//   b) ONE launch of G co-resident workgroups doing the same K phases separated by a grid barrier
//      (monotonic ticket counter, agent-scope atomics, bounded spin)
// Usage: grid_barrier [workgroups=490] [phases=100] [s_sleep between polls: 1|4|16|64]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ void phase_work(int* data, int phase) {
  // one dependent round trip per thread, like a histogram / scan phase
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  data[i] = data[i] + phase;
}

__global__ void __launch_bounds__(256) one_phase(int* data, int phase) { phase_work(data, phase); }

template <int SLEEP>
__global__ void __launch_bounds__(256) all_phases(int* data, int phases, unsigned* ticket, int* timed_out) {
  for (int p = 0; p < phases; ++p) {
    phase_work(data, p);
    // grid barrier: make this workgroup's writes visible, take a ticket, wait until all arrived
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      const unsigned goal = (unsigned)(p + 1) * gridDim.x;
      __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      long spins = 0;
      while (__hip_atomic_load(ticket, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < goal) {
        __builtin_amdgcn_s_sleep(SLEEP);
        if (++spins > 20000000) { *timed_out = 1; break; }
      }
    }
    __syncthreads();
  }
}

int main(int argc, char** argv) {
  const int G = argc > 1 ? atoi(argv[1]) : 490, K = argc > 2 ? atoi(argv[2]) : 100;
  int *data, *to; unsigned* ticket;
  hipMalloc(&data, (size_t)G * 256 * 4); hipMemset(data, 0, (size_t)G * 256 * 4);
  hipMalloc(&ticket, 4); hipMalloc(&to, 4); hipMemset(to, 0, 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int sleep = argc > 3 ? atoi(argv[3]) : 1;
  float best_a = 1e9, best_b = 1e9;
  for (int r = 0; r < 5; ++r) {
    hipEventRecord(a);
    for (int p = 0; p < K; ++p) hipLaunchKernelGGL(one_phase, dim3(G), dim3(256), 0, 0, data, p);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); if (ms < best_a) best_a = ms;
    hipMemset(ticket, 0, 4);
    hipEventRecord(a);
    if (sleep >= 64) hipLaunchKernelGGL(all_phases<64>, dim3(G), dim3(256), 0, 0, data, K, ticket, to);
    else if (sleep >= 16) hipLaunchKernelGGL(all_phases<16>, dim3(G), dim3(256), 0, 0, data, K, ticket, to);
    else if (sleep >= 4) hipLaunchKernelGGL(all_phases<4>, dim3(G), dim3(256), 0, 0, data, K, ticket, to);
    else hipLaunchKernelGGL(all_phases<1>, dim3(G), dim3(256), 0, 0, data, K, ticket, to);
    hipEventRecord(b); hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b); if (ms < best_b) best_b = ms;
  }
  int h_to = 0; hipMemcpy(&h_to, to, 4, hipMemcpyDeviceToHost);
  printf("workgroups %d phases %d sleep %d | separate launches: %.2f us per phase | one launch + grid barriers: %.2f us per phase%s\n",
         G, K, sleep, best_a * 1e3 / K, best_b * 1e3 / K, h_to ? "  (SPIN TIMEOUT)" : "");
  return 0;
}
