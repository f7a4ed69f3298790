// Stable least-significant-digit radix sort of (key, uint32 value) pairs for 64-lane wavefronts,
// 8-bit digits, keys of 32 or 64 bits, only bits [0, end_bit) sorted.
//
// Per pass (3 launches):
//   hist    : each workgroup histograms its key tile in LDS               -> block_hist[block][256]
//   scan    : workgroup d turns column d of block_hist into an exclusive prefix over workgroups
//             and writes the digit total
//   scatter : each workgroup scans the 256 digit totals, recomputes stable ranks for its tile
//             with wave-ballot match masks and writes keys+values to their final slots
// Stability: wave w of a workgroup owns the w-th contiguous quarter of the tile, rounds inside a
// wave advance through consecutive 64-key groups, ranks inside a round follow the lane order.
#pragma once
#include "fg_common.h"

namespace fg_sort {

constexpr int RADIX_BITS = 8;
constexpr int RADIX = 1 << RADIX_BITS;
constexpr int BLOCK = 256;
constexpr int WAVES = BLOCK / 64;
constexpr int KEYS_PER_THREAD = 16;
constexpr int TILE = BLOCK * KEYS_PER_THREAD;  // 4096 keys per workgroup
constexpr int WAVE_SPAN = 64 * KEYS_PER_THREAD;

template <typename KeyT>
__device__ __forceinline__ unsigned digit_of(KeyT key, int shift) {
  return (unsigned)(key >> shift) & (RADIX - 1);
}

template <typename KeyT>
__global__ void __launch_bounds__(BLOCK)
hist_kernel(int64_t n, const KeyT* __restrict__ keys, int shift, uint32_t* __restrict__ block_hist) {
  __shared__ uint32_t hist[RADIX];
  hist[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * TILE;
#pragma unroll 4
  for (int k = 0; k < KEYS_PER_THREAD; ++k) {
    const int64_t i = base + k * BLOCK + threadIdx.x;
    if (i < n) atomicAdd(&hist[digit_of(keys[i], shift)], 1u);
  }
  __syncthreads();
  block_hist[(size_t)blockIdx.x * RADIX + threadIdx.x] = hist[threadIdx.x];
}

// workgroup d: exclusive prefix over workgroups of column d of block_hist (in place) and the
// column total.  Thread t owns a contiguous chunk of the column.
static __global__ void __launch_bounds__(BLOCK)
digit_scan_kernel(int nblocks, uint32_t* __restrict__ block_hist, uint32_t* __restrict__ digit_total) {
  __shared__ uint32_t wave_tot[WAVES];
  const int d = blockIdx.x, lane = fg::lane_id(), wave = threadIdx.x >> 6;
  const int chunk = (nblocks + BLOCK - 1) / BLOCK;
  const int b0 = min(nblocks, (int)threadIdx.x * chunk), b1 = min(nblocks, b0 + chunk);
  uint32_t sum = 0;
  for (int b = b0; b < b1; ++b) sum += block_hist[(size_t)b * RADIX + d];
  uint32_t incl = sum;
#pragma unroll
  for (int k = 1; k < 64; k <<= 1) {
    const uint32_t o = __shfl_up(incl, k);
    if (lane >= k) incl += o;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  uint32_t run = incl - sum, total = 0;
#pragma unroll
  for (int w = 0; w < WAVES; ++w) {
    if (w < wave) run += wave_tot[w];
    total += wave_tot[w];
  }
  for (int b = b0; b < b1; ++b) {
    uint32_t* p = block_hist + (size_t)b * RADIX + d;
    const uint32_t c = *p;
    *p = run;
    run += c;
  }
  if (threadIdx.x == 0) digit_total[d] = total;
}

template <typename KeyT>
__global__ void __launch_bounds__(BLOCK)
scatter_kernel(int64_t n, const KeyT* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
               KeyT* __restrict__ keys_out, uint32_t* __restrict__ vals_out, int shift,
               const uint32_t* __restrict__ block_hist, const uint32_t* __restrict__ digit_total) {
  __shared__ uint32_t wave_cnt[WAVES][RADIX];
  __shared__ uint32_t scan_tmp[WAVES];
  const int lane = fg::lane_id(), wave = threadIdx.x >> 6;
#pragma unroll
  for (int w = 0; w < WAVES; ++w) wave_cnt[w][threadIdx.x] = 0;
  // global slot of this workgroup's first key of digit (threadIdx.x): exclusive scan of the 256
  // digit totals + this workgroup's offset inside the digit's bucket
  uint32_t digit_start;
  {
    const uint32_t v = digit_total[threadIdx.x];
    uint32_t incl = v;
#pragma unroll
    for (int k = 1; k < 64; k <<= 1) {
      const uint32_t o = __shfl_up(incl, k);
      if (lane >= k) incl += o;
    }
    if (lane == 63) scan_tmp[wave] = incl;
    __syncthreads();
    uint32_t base = incl - v;
#pragma unroll
    for (int w = 0; w < WAVES; ++w)
      if (w < wave) base += scan_tmp[w];
    digit_start = base + block_hist[(size_t)blockIdx.x * RADIX + threadIdx.x];
  }
  __syncthreads();

  // phase 1: ranks within the wave's own 1024-key span
  const int64_t wave_base = (int64_t)blockIdx.x * TILE + (int64_t)wave * WAVE_SPAN;
  KeyT key[KEYS_PER_THREAD];
  uint32_t val[KEYS_PER_THREAD];
  uint32_t rank[KEYS_PER_THREAD];
  const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int k = 0; k < KEYS_PER_THREAD; ++k) {
    const int64_t i = wave_base + k * 64 + lane;
    const bool in = i < n;
    key[k] = in ? keys_in[i] : (KeyT)~(KeyT)0;
    val[k] = in ? vals_in[i] : 0u;
    const unsigned d = digit_of(key[k], shift);
    uint64_t peers = __ballot(in);  // lanes holding the same digit
#pragma unroll
    for (int b = 0; b < RADIX_BITS; ++b) {
      const uint64_t m = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? m : ~m;
    }
    volatile uint32_t* cnt = &wave_cnt[wave][d];  // shared by the lanes of this wave across rounds
    const uint32_t before = *cnt;
    rank[k] = before + (uint32_t)__popcll(peers & lt_mask);
    // all peers have read `before` (same wave, program order, LDS completes in order); the
    // highest peer lane publishes the new running count
    __builtin_amdgcn_wave_barrier();
    if (in && (peers >> lane) == 1ull) *cnt = before + (uint32_t)__popcll(peers);
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();

  // phase 2: start slot of each wave's keys of digit (threadIdx.x)
  {
    uint32_t run = digit_start;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
      const uint32_t c = wave_cnt[w][threadIdx.x];
      wave_cnt[w][threadIdx.x] = run;
      run += c;
    }
  }
  __syncthreads();

  // phase 3: scatter
#pragma unroll
  for (int k = 0; k < KEYS_PER_THREAD; ++k) {
    const int64_t i = wave_base + k * 64 + lane;
    if (i < n) {
      const uint32_t pos = wave_cnt[wave][digit_of(key[k], shift)] + rank[k];
      keys_out[pos] = key[k];
      vals_out[pos] = val[k];
    }
  }
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static inline int num_blocks(int64_t n) { return (int)((n + TILE - 1) / TILE); }

// scratch: alternate key/value buffers + histograms
template <typename KeyT>
static inline size_t workspace_bytes(int64_t n) {
  if (n < 0) n = 0;
  const size_t nb = (size_t)num_blocks(n) + 1;
  return align256((size_t)n * sizeof(KeyT)) + align256((size_t)n * 4) + align256(nb * RADIX * 4) +
         align256((size_t)RADIX * 4);
}

// Sorts in place (result copied back into keys/vals if it ends in the scratch copy).
template <typename KeyT>
static inline int sort_pairs(int64_t n, KeyT* keys, uint32_t* vals, int end_bit, void* workspace,
                             size_t ws_bytes, hipStream_t s) {
  if (n <= 1 || end_bit <= 0) return FG_OK;
  if (n > 0xFFFFFFFFll) return FG_ERR_UNSUPPORTED;
  if (ws_bytes < workspace_bytes<KeyT>(n)) return FG_ERR_WORKSPACE;
  const int nb = num_blocks(n);
  char* ws = static_cast<char*>(workspace);
  KeyT* keys_alt = reinterpret_cast<KeyT*>(ws);
  ws += align256((size_t)n * sizeof(KeyT));
  uint32_t* vals_alt = reinterpret_cast<uint32_t*>(ws);
  ws += align256((size_t)n * 4);
  uint32_t* block_hist = reinterpret_cast<uint32_t*>(ws);
  ws += align256(((size_t)nb + 1) * RADIX * 4);
  uint32_t* digit_total = reinterpret_cast<uint32_t*>(ws);

  KeyT *kin = keys, *kout = keys_alt;
  uint32_t *vin = vals, *vout = vals_alt;
  const int passes = (end_bit + RADIX_BITS - 1) / RADIX_BITS;
  for (int p = 0; p < passes; ++p) {
    const int shift = p * RADIX_BITS;
    hipLaunchKernelGGL(hist_kernel<KeyT>, dim3(nb), dim3(BLOCK), 0, s, n, kin, shift, block_hist);
    hipLaunchKernelGGL(digit_scan_kernel, dim3(RADIX), dim3(BLOCK), 0, s, nb, block_hist, digit_total);
    hipLaunchKernelGGL(scatter_kernel<KeyT>, dim3(nb), dim3(BLOCK), 0, s, n, kin, vin, kout, vout, shift,
                       block_hist, digit_total);
    KeyT* tk = kin; kin = kout; kout = tk;
    uint32_t* tv = vin; vin = vout; vout = tv;
  }
  if (kin != keys) {
    if (hipMemcpyAsync(keys, kin, (size_t)n * sizeof(KeyT), hipMemcpyDeviceToDevice, s) != hipSuccess)
      return FG_ERR_LAUNCH;
    if (hipMemcpyAsync(vals, vin, (size_t)n * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return FG_ERR_LAUNCH;
  }
  return hipGetLastError() == hipSuccess ? FG_OK : FG_ERR_LAUNCH;
}

}  // namespace fg_sort
