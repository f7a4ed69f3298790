// Stable least-significant-digit radix sort of (key, uint32 value) pairs for 64-lane wavefronts,
// 8-bit digits, keys of 32 or 64 bits, only bits [0, end_bit) sorted.
//
// Per pass (3 launches):
//   hist    : each workgroup histograms its key tile in LDS               -> block_hist[256][block]
//   scan    : workgroup d turns row d of block_hist into an exclusive prefix over workgroups
//             and writes the digit total
//   scatter : each workgroup scans the 256 digit totals, recomputes stable ranks for its tile
//             with wave-ballot match masks and writes keys+values to their final slots
// Stability: wave w of a workgroup owns the w-th contiguous quarter of the tile, rounds inside a
// wave advance through consecutive 64-key groups, ranks inside a round follow the lane order.
#pragma once
#include <cstdlib>

#include "fg_common.h"

namespace fg_sort {

constexpr int RADIX_BITS = 8;
constexpr int RADIX = 1 << RADIX_BITS;
constexpr int BLOCK = 256;
constexpr int WAVES = BLOCK / 64;
#ifndef FG_SORT_KPT
#define FG_SORT_KPT 16
#endif
constexpr int KEYS_PER_THREAD = FG_SORT_KPT;
constexpr int TILE = BLOCK * KEYS_PER_THREAD;  // 4096 keys per workgroup
constexpr int WAVE_SPAN = 64 * KEYS_PER_THREAD;
// Small arrays (the depth sort of ~1M Gaussians is 245 workgroups of 4096 keys on 256 CUs) are
// sorted with fewer keys per thread, i.e. more, shorter workgroups.  The kernels take the keys
// per thread as a template parameter and shadow the three constants above.
#ifndef FG_SORT_KPT_SMALL
#define FG_SORT_KPT_SMALL 8
#endif
#ifndef FG_SORT_SMALL_N
#define FG_SORT_SMALL_N (3 << 20)
#endif
static inline int keys_per_thread_for(int64_t n) { return n < FG_SORT_SMALL_N ? FG_SORT_KPT_SMALL : FG_SORT_KPT; }

// `shift` of the kernels below packs the pass's digit: bits 0-7 = bit position, bits 8-11 = digit
// width (<= RADIX_BITS).  Passes share the key bits evenly (13 tile bits = 7 + 6, not 8 + 5): the
// ranking costs one ballot round per digit bit.

template <typename KeyT>
__device__ __forceinline__ unsigned digit_of(KeyT key, int shift) {
  return (unsigned)(key >> (shift & 255)) & ((1u << (shift >> 8)) - 1u);
}

template <typename KeyT, int KPT>
__global__ void __launch_bounds__(BLOCK)
hist_kernel(int64_t n, const int64_t* __restrict__ n_dev, const KeyT* __restrict__ keys, int shift,
            uint32_t* __restrict__ block_hist) {
  constexpr int KEYS_PER_THREAD = KPT, TILE = BLOCK * KPT;
  __shared__ uint32_t hist[RADIX];
  hist[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * TILE;
  // keys are loaded up to the capacity n (in bounds); the device-side count, if any, is read in
  // the same batch of loads and applied afterwards, so its latency is not serialised in front
  KeyT key[KEYS_PER_THREAD];
#pragma unroll
  for (int k = 0; k < KEYS_PER_THREAD; ++k) {
    const int64_t i = base + k * BLOCK + threadIdx.x;
    key[k] = i < n ? keys[i] : (KeyT)0;
  }
  if (n_dev) n = min(n, *n_dev);
#pragma unroll
  for (int k = 0; k < KEYS_PER_THREAD; ++k) {
    const int64_t i = base + k * BLOCK + threadIdx.x;
    if (i < n) atomicAdd(&hist[digit_of(key[k], shift)], 1u);
  }
  __syncthreads();
  // digit-major layout [RADIX][nblocks]: the per-digit scan below then walks contiguous memory
  block_hist[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = hist[threadIdx.x];
}

// workgroup d: exclusive prefix over workgroups of row d of block_hist (in place) and the
// row total.  Thread t owns a contiguous chunk of the row.
static __global__ void __launch_bounds__(BLOCK)
digit_scan_kernel(int nblocks, uint32_t* __restrict__ block_hist, uint32_t* __restrict__ digit_total) {
  __shared__ uint32_t wave_tot[WAVES];
  const int d = blockIdx.x, lane = fg::lane_id(), wave = threadIdx.x >> 6;
  const int chunk = (nblocks + BLOCK - 1) / BLOCK;
  const int b0 = min(nblocks, (int)threadIdx.x * chunk), b1 = min(nblocks, b0 + chunk);
  uint32_t sum = 0;
  uint32_t* row = block_hist + (size_t)d * nblocks;
  for (int b = b0; b < b1; ++b) sum += row[b];
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
    const uint32_t c = row[b];
    row[b] = run;
    run += c;
  }
  if (threadIdx.x == 0) digit_total[d] = total;
}

// NBITS: digit width known at compile time (5..8), or 0 = read it from `shift` (narrow digits)
template <typename KeyT, int NBITS, int KPT>
__global__ void __launch_bounds__(BLOCK, 3)
scatter_kernel(int64_t n, const int64_t* __restrict__ n_dev, const KeyT* __restrict__ keys_in,
               const uint32_t* __restrict__ vals_in, KeyT* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
               int shift, const uint32_t* __restrict__ block_hist, const uint32_t* __restrict__ digit_total) {
  constexpr int KEYS_PER_THREAD = KPT, TILE = BLOCK * KPT, WAVE_SPAN = 64 * KPT;
  const int64_t cap = n;
  __shared__ uint32_t wave_cnt[WAVES][RADIX];
  __shared__ uint32_t scan_tmp[WAVES];
  __shared__ uint32_t global_delta[RADIX];  // (global slot) - (slot in the LDS image) per digit
  __shared__ KeyT skeys[TILE];              // the tile, re-ordered by digit
  __shared__ uint32_t svals[TILE];
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
    digit_start = base + block_hist[(size_t)threadIdx.x * gridDim.x + blockIdx.x];
  }
  __syncthreads();

  // phase 1: ranks within the wave's own 1024-key span.  Per round the lowest lane of each
  // group of equal digits adds the group size to the wave's running LDS counter with a
  // returning atomic; the 16 atomics of a lane are issued back to back (same-address LDS
  // atomics of one wave retire in issue order, so the returned values are the prefix counts)
  // and only then consumed.
  const int64_t tile_base = (int64_t)blockIdx.x * TILE;
  const int64_t wave_base = tile_base + (int64_t)wave * WAVE_SPAN;
  KeyT key[KEYS_PER_THREAD];
  uint32_t val[KEYS_PER_THREAD];
  // rank[k]: bits 0-15 = keys of the same digit before this one in the wave's span (first the
  // in-round count below this lane, or -- on a group's leader lane, whose in-round count is 0 --
  // the running count returned by the atomic), bits 16-21 = the group's leader lane
  uint32_t rank[KEYS_PER_THREAD];
  const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int k = 0; k < KEYS_PER_THREAD; ++k) {
    const int64_t i = wave_base + k * 64 + lane;
    const bool in = i < cap;  // in bounds of the buffers; the count proper is applied below
    key[k] = in ? keys_in[i] : (KeyT)~(KeyT)0;
    val[k] = in ? (vals_in ? vals_in[i] : (uint32_t)i) : 0u;  // no input values: the element's index
  }
  if (n_dev) n = min(n, *n_dev);  // device-side count: read alongside the keys, not in front of them
#pragma unroll
  for (int k = 0; k < KEYS_PER_THREAD; ++k) {
    const bool in = wave_base + k * 64 + lane < n;
    const unsigned d = digit_of(key[k], shift);
    // lanes holding the same digit: per digit bit one ballot and, per half of the lane mask, one
    // three-input bit operation  peers & ~(ballot ^ sel)  with sel = -1 where the lane's bit is set
    // (v_bitop3_b32, truth table 0x90).  As `peers &= bit ? m : ~m` the compiler spent ~10 vector
    // instructions per round on carries and selects.
    const uint64_t in_mask = __ballot(in);
    uint32_t peers_lo = (uint32_t)in_mask, peers_hi = (uint32_t)(in_mask >> 32);
#pragma unroll
    for (int b = 0; b < RADIX_BITS; ++b) {
      if (NBITS ? b >= NBITS : b >= (shift >> 8)) break;  // narrower digits need fewer rounds
      const int sel = (int)(d << (31 - b)) >> 31;
      const uint64_t m = __ballot(sel != 0);
      peers_lo = __builtin_amdgcn_bitop3_b32(peers_lo, (uint32_t)m, (uint32_t)sel, 0x90);
      peers_hi = __builtin_amdgcn_bitop3_b32(peers_hi, (uint32_t)(m >> 32), (uint32_t)sel, 0x90);
    }
    const uint64_t peers = ((uint64_t)peers_hi << 32) | peers_lo;
    const uint32_t leader = in ? (uint32_t)__builtin_ctzll(peers) : (uint32_t)lane;
    uint32_t r = (uint32_t)__popcll(peers & lt_mask);
    if (in && leader == (uint32_t)lane) r = atomicAdd(&wave_cnt[wave][d], (uint32_t)__popcll(peers));
    rank[k] = r | (leader << 16);
  }
#pragma unroll
  for (int k = 0; k < KEYS_PER_THREAD; ++k) {
    const uint32_t leader = rank[k] >> 16;
    const uint32_t before = (uint32_t)__shfl((int)(rank[k] & 0xFFFFu), (int)leader);
    rank[k] = (leader == (uint32_t)lane) ? before : before + (rank[k] & 0xFFFFu);
  }
  __syncthreads();

  // phase 2 (thread t = digit t): place the digits back to back in the LDS image
  {
    uint32_t c[WAVES], tot = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
      c[w] = wave_cnt[w][threadIdx.x];
      tot += c[w];
    }
    uint32_t incl = tot;
#pragma unroll
    for (int k = 1; k < 64; k <<= 1) {
      const uint32_t o = __shfl_up(incl, k);
      if (lane >= k) incl += o;
    }
    if (lane == 63) scan_tmp[wave] = incl;
    __syncthreads();
    uint32_t local = incl - tot;
#pragma unroll
    for (int w = 0; w < WAVES; ++w)
      if (w < wave) local += scan_tmp[w];
    global_delta[threadIdx.x] = digit_start - local;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
      wave_cnt[w][threadIdx.x] = local;
      local += c[w];
    }
  }
  __syncthreads();

  // phase 3: scatter into the LDS image
#pragma unroll
  for (int k = 0; k < KEYS_PER_THREAD; ++k) {
    if (wave_base + k * 64 + lane < n) {
      const uint32_t lp = wave_cnt[wave][digit_of(key[k], shift)] + rank[k];
      skeys[lp] = key[k];
      svals[lp] = val[k];
    }
  }
  __syncthreads();

  // phase 4: stream the image out; consecutive lanes hit consecutive slots of a bucket
  const int count = (int)min((int64_t)TILE, n - tile_base);
  for (int e = threadIdx.x; e < count; e += BLOCK) {
    const KeyT kk = skeys[e];
    const uint32_t pos = (uint32_t)e + global_delta[digit_of(kk, shift)];
    keys_out[pos] = kk;
    vals_out[pos] = svals[e];
  }
}

// one word, as a launch of our own (a hipMemsetAsync under torch's graph capture was dropped from the graph)
static __global__ void set_word_kernel(uint32_t* p, uint32_t v) { *p = v; }

// (A single-kernel variant of the passes -- ticketed tiles, chained scan with decoupled look-back
// -- was built, verified bit-exact and measured slower on MI355X: profiles/r01_onesweep.md; it was
// removed again, see the git history of this file.)

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static inline int num_blocks(int64_t n) {
  const int64_t tile = (int64_t)BLOCK * keys_per_thread_for(n);
  return (int)((n + tile - 1) / tile);
}

// bytes of the histogram region: per-workgroup digit histograms + the digit totals
static inline size_t control_bytes(int64_t n) { return align256(((size_t)num_blocks(n) + 2) * RADIX * 4); }

// scratch: alternate key/value buffers + histogram region
template <typename KeyT>
static inline size_t workspace_bytes(int64_t n) {
  if (n < 0) n = 0;
  return align256((size_t)n * sizeof(KeyT)) + align256((size_t)n * 4) + control_bytes(n);
}

// Sorts in place (result copied back into keys/vals if it ends in the scratch copy).
// With n_dev != nullptr the element count is min(n, *n_dev), read on the device: n is then only the
// capacity the launch grids and the scratch are sized for (no host round trip for the count).
// iota_vals: the values are the element indices 0..n-1 and `vals` is output only (the first pass
// numbers them itself instead of reading an array somebody had to fill).
template <typename KeyT>
static inline int sort_pairs(int64_t n, KeyT* keys, uint32_t* vals, int end_bit, void* workspace,
                             size_t ws_bytes, hipStream_t s, const int64_t* n_dev = nullptr, bool iota_vals = false) {
  if (iota_vals && (n <= 1 || end_bit <= 0)) {
    if (n == 1) {
      hipLaunchKernelGGL(set_word_kernel, dim3(1), dim3(1), 0, s, vals, 0u);
      if (hipGetLastError() != hipSuccess) return FG_ERR_LAUNCH;
    }
    return FG_OK;
  }
  if (n <= 1 || end_bit <= 0) return FG_OK;
  if (n > 0xFFFFFFFFll) return FG_ERR_UNSUPPORTED;
  if (ws_bytes < workspace_bytes<KeyT>(n)) return FG_ERR_WORKSPACE;
  const int nb = num_blocks(n);
  const bool small = keys_per_thread_for(n) != FG_SORT_KPT;
  char* ws = static_cast<char*>(workspace);
  KeyT* keys_alt = reinterpret_cast<KeyT*>(ws);
  ws += align256((size_t)n * sizeof(KeyT));
  uint32_t* vals_alt = reinterpret_cast<uint32_t*>(ws);
  ws += align256((size_t)n * 4);
  uint32_t* control = reinterpret_cast<uint32_t*>(ws);

  KeyT *kin = keys, *kout = keys_alt;
  uint32_t *vin = vals, *vout = vals_alt;
  const int passes = (end_bit + RADIX_BITS - 1) / RADIX_BITS;
  uint32_t* block_hist = control;
  uint32_t* digit_total = control + ((size_t)nb + 1) * RADIX;
  int first_bit = 0;
  for (int p = 0; p < passes; ++p) {
    const int nbits = (end_bit - first_bit + (passes - p) - 1) / (passes - p);  // even split of what is left
    const int shift = first_bit | (nbits << 8);
    first_bit += nbits;
    if (small) hipLaunchKernelGGL((hist_kernel<KeyT, FG_SORT_KPT_SMALL>), dim3(nb), dim3(BLOCK), 0, s, n, n_dev, kin, shift, block_hist);
    else hipLaunchKernelGGL((hist_kernel<KeyT, FG_SORT_KPT>), dim3(nb), dim3(BLOCK), 0, s, n, n_dev, kin, shift, block_hist);
    hipLaunchKernelGGL(digit_scan_kernel, dim3(RADIX), dim3(BLOCK), 0, s, nb, block_hist, digit_total);
#define FG_SCATTER(NB)                                                                                         \
  if (small)                                                                                                    \
    hipLaunchKernelGGL((scatter_kernel<KeyT, NB, FG_SORT_KPT_SMALL>), dim3(nb), dim3(BLOCK), 0, s, n, n_dev,    \
                       kin, vsrc, kout, vout, shift, block_hist, digit_total);                                  \
  else                                                                                                          \
    hipLaunchKernelGGL((scatter_kernel<KeyT, NB, FG_SORT_KPT>), dim3(nb), dim3(BLOCK), 0, s, n, n_dev, kin,     \
                       vsrc, kout, vout, shift, block_hist, digit_total)
    const uint32_t* vsrc = (iota_vals && p == 0) ? nullptr : vin;
    switch (nbits) {
      case 8: FG_SCATTER(8); break;
      case 7: FG_SCATTER(7); break;
      case 6: FG_SCATTER(6); break;
      case 5: FG_SCATTER(5); break;
      default: FG_SCATTER(0); break;
    }
#undef FG_SCATTER
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
