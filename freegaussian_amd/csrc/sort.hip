// K4: stable least-significant-digit radix sort of (uint64 key, uint32 value) pairs, 8-bit
// digits, written for 64-lane wavefronts.  Only the significant key bits are sorted
// (32 depth bits + ceil(log2 tiles) tile bits, e.g. 45 bits -> 6 passes at 1080p).
//
// Per pass:
//   1. hist   : each workgroup histograms its 4096-key tile in LDS        (reads 8 B/key)
//   2. rowscan: one wave per digit scans that digit's per-workgroup counts  (tiny)
//   3. scatter: each workgroup recomputes stable ranks for its tile with wave-ballot
//               match masks and writes keys+values to their final slots   (r/w 12 B/key)
// Stability comes from a blocked key->wave assignment: wave w owns the w-th contiguous
// quarter of the tile, rounds inside a wave advance through consecutive 64-key groups, and
// ranks inside a round follow the lane order.
#include "fg_common.h"

namespace {

constexpr int RADIX_BITS = 8;
constexpr int RADIX = 1 << RADIX_BITS;  // 256
constexpr int SORT_BLOCK = 256;         // threads = 4 waves
constexpr int SORT_WAVES = SORT_BLOCK / 64;
constexpr int KEYS_PER_THREAD = 16;
constexpr int SORT_TILE = SORT_BLOCK * KEYS_PER_THREAD;  // 4096 keys per workgroup
constexpr int WAVE_SPAN = 64 * KEYS_PER_THREAD;          // 1024 consecutive keys per wave

__device__ __forceinline__ unsigned digit_of(uint64_t key, int shift) {
  return (unsigned)(key >> shift) & (RADIX - 1);
}

__global__ void __launch_bounds__(SORT_BLOCK)
radix_hist_kernel(int64_t n, const uint64_t* __restrict__ keys, int shift, int nblocks,
                  uint32_t* __restrict__ block_hist /* [RADIX][nblocks] */) {
  __shared__ uint32_t hist[RADIX];
  hist[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * SORT_TILE;
#pragma unroll 4
  for (int k = 0; k < KEYS_PER_THREAD; ++k) {
    const int64_t i = base + k * SORT_BLOCK + threadIdx.x;
    if (i < n) atomicAdd(&hist[digit_of(keys[i], shift)], 1u);
  }
  __syncthreads();
  block_hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = hist[threadIdx.x];
}

// One wave per digit: exclusive scan of that digit's row of per-workgroup counts, in place;
// the row total goes to digit_total[digit].
__global__ void __launch_bounds__(SORT_BLOCK)
radix_rowscan_kernel(int nblocks, uint32_t* __restrict__ block_hist, uint32_t* __restrict__ digit_total) {
  const int lane = fg::lane_id();
  const int digit = blockIdx.x * SORT_WAVES + (threadIdx.x >> 6);
  uint32_t* row = block_hist + (size_t)digit * nblocks;
  uint32_t carry = 0;
  for (int base = 0; base < nblocks; base += 64) {
    const int i = base + lane;
    const uint32_t v = (i < nblocks) ? row[i] : 0u;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(incl, d);
      if (lane >= d) incl += o;
    }
    if (i < nblocks) row[i] = carry + incl - v;
    carry += __shfl(incl, 63);
  }
  if (lane == 0) digit_total[digit] = carry;
}

__global__ void __launch_bounds__(SORT_BLOCK)
radix_scatter_kernel(int64_t n, const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                     uint64_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, int shift, int nblocks,
                     const uint32_t* __restrict__ block_hist, const uint32_t* __restrict__ digit_total) {
  __shared__ uint32_t wave_cnt[SORT_WAVES][RADIX];  // running per-wave digit counters
  __shared__ uint32_t digit_base[RADIX];            // global start of each digit bucket
  __shared__ uint32_t scan_tmp[SORT_WAVES];
  const int lane = fg::lane_id(), wave = threadIdx.x >> 6;
#pragma unroll
  for (int w = 0; w < SORT_WAVES; ++w) wave_cnt[w][threadIdx.x] = 0;

  // exclusive scan of the 256 digit totals (one per thread)
  {
    const uint32_t v = digit_total[threadIdx.x];
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(incl, d);
      if (lane >= d) incl += o;
    }
    if (lane == 63) scan_tmp[wave] = incl;
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (int w = 0; w < SORT_WAVES; ++w)
      if (w < wave) base += scan_tmp[w];
    digit_base[threadIdx.x] = base + incl - v + block_hist[(size_t)threadIdx.x * nblocks + blockIdx.x];
  }
  __syncthreads();

  // phase 1: ranks within the wave's own 1024-key span
  const int64_t wave_base = (int64_t)blockIdx.x * SORT_TILE + (int64_t)wave * WAVE_SPAN;
  uint64_t key[KEYS_PER_THREAD];
  uint32_t val[KEYS_PER_THREAD];
  uint32_t rank[KEYS_PER_THREAD];
  const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int k = 0; k < KEYS_PER_THREAD; ++k) {
    const int64_t i = wave_base + k * 64 + lane;
    const bool in = i < n;
    key[k] = in ? keys_in[i] : ~0ull;
    val[k] = in ? vals_in[i] : 0u;
    const unsigned d = digit_of(key[k], shift);
    // lanes holding the same digit (out-of-range lanes excluded)
    uint64_t peers = __ballot(in);
#pragma unroll
    for (int b = 0; b < RADIX_BITS; ++b) {
      const uint64_t m = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? m : ~m;
    }
    // volatile: the counter is shared by the lanes of this wave across rounds
    volatile uint32_t* cnt = &wave_cnt[wave][d];
    const uint32_t before = *cnt;
    rank[k] = before + (uint32_t)__popcll(peers & lt_mask);
    // the highest peer lane publishes the new running count (all peers read `before` first:
    // same wave, program order + LDS in-order completion)
    __builtin_amdgcn_wave_barrier();
    if (in && (peers >> lane) == 1ull) *cnt = before + (uint32_t)__popcll(peers);
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();

  // phase 2: per-digit offsets of each wave inside the workgroup (thread t handles digit t)
  {
    uint32_t run = digit_base[threadIdx.x];
#pragma unroll
    for (int w = 0; w < SORT_WAVES; ++w) {
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

}  // namespace

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" size_t fg_sort_workspace_bytes(int64_t n) {
  if (n < 0) n = 0;
  const size_t nblocks = (size_t)((n + SORT_TILE - 1) / SORT_TILE) + 1;
  return align256((size_t)n * 8) + align256((size_t)n * 4) + align256(nblocks * RADIX * 4) + align256(RADIX * 4);
}

extern "C" int fg_sort_pairs(int64_t n, int64_t* keys, int32_t* vals, int end_bit, void* workspace,
                             size_t workspace_bytes, fg_stream_t stream) {
  if (n < 0 || end_bit < 0 || end_bit > 64) return FG_ERR_INVALID_ARG;
  if (n <= 1 || end_bit == 0) return FG_OK;
  if (n > 0xFFFFFFFFll) return FG_ERR_UNSUPPORTED;
  if (!keys || !vals || !workspace) return FG_ERR_INVALID_ARG;
  if (workspace_bytes < fg_sort_workspace_bytes(n)) return FG_ERR_WORKSPACE;
  const int nblocks = (int)((n + SORT_TILE - 1) / SORT_TILE);
  char* ws = static_cast<char*>(workspace);
  uint64_t* keys_alt = reinterpret_cast<uint64_t*>(ws);
  ws += align256((size_t)n * 8);
  uint32_t* vals_alt = reinterpret_cast<uint32_t*>(ws);
  ws += align256((size_t)n * 4);
  uint32_t* block_hist = reinterpret_cast<uint32_t*>(ws);
  ws += align256(((size_t)nblocks + 1) * RADIX * 4);
  uint32_t* digit_total = reinterpret_cast<uint32_t*>(ws);

  hipStream_t s = fg_hip_stream(stream);
  uint64_t* kin = reinterpret_cast<uint64_t*>(keys);
  uint32_t* vin = reinterpret_cast<uint32_t*>(vals);
  uint64_t* kout = keys_alt;
  uint32_t* vout = vals_alt;
  const int passes = (end_bit + RADIX_BITS - 1) / RADIX_BITS;
  for (int p = 0; p < passes; ++p) {
    const int shift = p * RADIX_BITS;
    hipLaunchKernelGGL(radix_hist_kernel, dim3(nblocks), dim3(SORT_BLOCK), 0, s, n, kin, shift, nblocks, block_hist);
    hipLaunchKernelGGL(radix_rowscan_kernel, dim3(RADIX / SORT_WAVES), dim3(SORT_BLOCK), 0, s, nblocks, block_hist,
                       digit_total);
    hipLaunchKernelGGL(radix_scatter_kernel, dim3(nblocks), dim3(SORT_BLOCK), 0, s, n, kin, vin, kout, vout, shift,
                       nblocks, block_hist, digit_total);
    uint64_t* tk = kin; kin = kout; kout = tk;
    uint32_t* tv = vin; vin = vout; vout = tv;
  }
  if (kin != reinterpret_cast<uint64_t*>(keys)) {
    if (hipMemcpyAsync(keys, kin, (size_t)n * 8, hipMemcpyDeviceToDevice, s) != hipSuccess) return FG_ERR_LAUNCH;
    if (hipMemcpyAsync(vals, vin, (size_t)n * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return FG_ERR_LAUNCH;
  }
  FG_RETURN_IF_LAUNCH_FAILED();
  return FG_OK;
}
