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

template <typename KeyT>
__device__ __forceinline__ unsigned digit_of(KeyT key, int shift) {
  return (unsigned)(key >> shift) & (RADIX - 1);
}

template <typename KeyT>
__global__ void __launch_bounds__(BLOCK)
hist_kernel(int64_t n, const int64_t* __restrict__ n_dev, const KeyT* __restrict__ keys, int shift,
            uint32_t* __restrict__ block_hist) {
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
__global__ void __launch_bounds__(BLOCK, 3)
scatter_kernel(int64_t n, const int64_t* __restrict__ n_dev, const KeyT* __restrict__ keys_in,
               const uint32_t* __restrict__ vals_in, KeyT* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
               int shift, const uint32_t* __restrict__ block_hist, const uint32_t* __restrict__ digit_total) {
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
    digit_start = base + block_hist[(size_t)blockIdx.x * RADIX + threadIdx.x];
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
    val[k] = in ? vals_in[i] : 0u;
  }
  if (n_dev) n = min(n, *n_dev);  // device-side count: read alongside the keys, not in front of them
#pragma unroll
  for (int k = 0; k < KEYS_PER_THREAD; ++k) {
    const bool in = wave_base + k * 64 + lane < n;
    const unsigned d = digit_of(key[k], shift);
    uint64_t peers = __ballot(in);  // lanes holding the same digit
#pragma unroll
    for (int b = 0; b < RADIX_BITS; ++b) {
      const uint64_t m = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? m : ~m;
    }
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

// ---- single-kernel passes ("onesweep": chained scan with decoupled look-back), OPTIONAL -------
// One launch per pass instead of three (off by default: see use_onesweep()).  The digit histograms of ALL passes come from one upfront
// read of the keys (global_hist_kernel).  In a pass, a workgroup takes its tile number from an
// atomic ticket (so tile t-1 is always already running when tile t waits on it), ranks its keys,
// publishes its per-digit counts in status[tile][digit] (flag in the top two bits: AGGREGATE =
// the tile's own count, INCLUSIVE = prefix over tiles 0..tile) and walks back over the
// predecessors' entries until it meets an INCLUSIVE one.  A status word is written and read as
// one 32-bit agent-scope atomic, so flag and count cannot tear.
constexpr uint32_t ST_AGGREGATE = 1u << 30, ST_INCLUSIVE = 2u << 30, ST_COUNT = (1u << 30) - 1;
constexpr int MAX_PASSES = 8;

template <typename KeyT>
__global__ void __launch_bounds__(BLOCK)
global_hist_kernel(int64_t n, const int64_t* __restrict__ n_dev, const KeyT* __restrict__ keys, int passes,
                   uint32_t* __restrict__ ghist) {
  __shared__ uint32_t hist[MAX_PASSES][RADIX];
  for (int p = 0; p < passes; ++p) hist[p][threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * TILE;
  KeyT key[KEYS_PER_THREAD];
#pragma unroll
  for (int k = 0; k < KEYS_PER_THREAD; ++k) {
    const int64_t i = base + k * BLOCK + threadIdx.x;
    key[k] = i < n ? keys[i] : (KeyT)0;
  }
  if (n_dev) n = min(n, *n_dev);
  for (int p = 0; p < passes; ++p) {
#pragma unroll
    for (int k = 0; k < KEYS_PER_THREAD; ++k) {
      const int64_t i = base + k * BLOCK + threadIdx.x;
      if (i < n) atomicAdd(&hist[p][digit_of(key[k], p * RADIX_BITS)], 1u);
    }
  }
  __syncthreads();
  for (int p = 0; p < passes; ++p) {
    const uint32_t c = hist[p][threadIdx.x];
    if (c) atomicAdd(&ghist[p * RADIX + threadIdx.x], c);
  }
}

template <typename KeyT>
__global__ void __launch_bounds__(BLOCK, 3)
onesweep_kernel(int64_t n, const int64_t* __restrict__ n_dev, const KeyT* __restrict__ keys_in,
                const uint32_t* __restrict__ vals_in, KeyT* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                int shift, const uint32_t* __restrict__ ghist, uint32_t* __restrict__ status,
                uint32_t* __restrict__ ticket) {
  __shared__ uint32_t wave_cnt[WAVES][RADIX];
  __shared__ uint32_t scan_tmp[WAVES];
  __shared__ uint32_t global_delta[RADIX];  // (global slot) - (slot in the LDS image) per digit
  __shared__ KeyT skeys[TILE];              // the tile, re-ordered by digit
  __shared__ uint32_t svals[TILE];
  __shared__ int s_tile;
  const int lane = fg::lane_id(), wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_tile = (int)atomicAdd(ticket, 1u);
#pragma unroll
  for (int w = 0; w < WAVES; ++w) wave_cnt[w][threadIdx.x] = 0;
  __syncthreads();
  const int tile = s_tile;
  const int64_t cap = n;
  const int64_t tile_base = (int64_t)tile * TILE;
  const int64_t wave_base = tile_base + (int64_t)wave * WAVE_SPAN;
  KeyT key[KEYS_PER_THREAD];
  uint32_t val[KEYS_PER_THREAD];
  uint32_t rank[KEYS_PER_THREAD];
  const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int k = 0; k < KEYS_PER_THREAD; ++k) {
    const int64_t i = wave_base + k * 64 + lane;
    const bool in = i < cap;
    key[k] = in ? keys_in[i] : (KeyT)~(KeyT)0;
    val[k] = in ? vals_in[i] : 0u;
  }
  const uint32_t gh = ghist[threadIdx.x];
  if (n_dev) n = min(n, *n_dev);
  // tiles past the count hold nothing and nobody after them does either (tickets are handed out
  // in launch order and the populated tiles are a prefix): leave without publishing
  if (tile_base >= n) return;

  // phase 1: stable ranks within the wave's own span (see scatter_kernel)
#pragma unroll
  for (int k = 0; k < KEYS_PER_THREAD; ++k) {
    const bool in = wave_base + k * 64 + lane < n;
    const unsigned d = digit_of(key[k], shift);
    uint64_t peers = __ballot(in);
#pragma unroll
    for (int b = 0; b < RADIX_BITS; ++b) {
      const uint64_t m = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? m : ~m;
    }
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

  // phase 2 (thread t = digit t): publish the tile's count, look back for the prefix over the
  // earlier tiles, then place the digits back to back in the LDS image
  {
    uint32_t c[WAVES], tot = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
      c[w] = wave_cnt[w][threadIdx.x];
      tot += c[w];
    }
    uint32_t* my = status + (size_t)tile * RADIX + threadIdx.x;
    __hip_atomic_store(my, (tile == 0 ? ST_INCLUSIVE : ST_AGGREGATE) | tot, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    uint32_t excl = 0;
    if (tile > 0) {
      for (int t = tile - 1; t >= 0; --t) {
        const uint32_t* p = status + (size_t)t * RADIX + threadIdx.x;
        uint32_t sv;
        do {
          sv = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } while ((sv >> 30) == 0u);
        excl += sv & ST_COUNT;
        if ((sv >> 30) == 2u) break;
      }
      __hip_atomic_store(my, ST_INCLUSIVE | (excl + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // exclusive scan over digits of the global histogram -> first slot of each digit's bucket
    uint32_t ginc = gh;
#pragma unroll
    for (int k = 1; k < 64; k <<= 1) {
      const uint32_t o = __shfl_up(ginc, k);
      if (lane >= k) ginc += o;
    }
    // and of the tile's own counts -> slot in the LDS image
    uint32_t incl = tot;
#pragma unroll
    for (int k = 1; k < 64; k <<= 1) {
      const uint32_t o = __shfl_up(incl, k);
      if (lane >= k) incl += o;
    }
    __shared__ uint32_t gscan_tmp[WAVES];
    if (lane == 63) {
      scan_tmp[wave] = incl;
      gscan_tmp[wave] = ginc;
    }
    __syncthreads();
    uint32_t local = incl - tot, gbase = ginc - gh;
#pragma unroll
    for (int w = 0; w < WAVES; ++w)
      if (w < wave) {
        local += scan_tmp[w];
        gbase += gscan_tmp[w];
      }
    global_delta[threadIdx.x] = gbase + excl - local;
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

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static inline int num_blocks(int64_t n) { return (int)((n + TILE - 1) / TILE); }

// bytes of the zero-initialised control region: global histograms, tickets, tile status words
static inline size_t control_bytes(int64_t n) {
  return align256(((size_t)MAX_PASSES * RADIX + MAX_PASSES + (size_t)MAX_PASSES * num_blocks(n) * RADIX) * 4);
}

// scratch: alternate key/value buffers + control region (the 3-kernel fallback uses the same
// region for its block histograms)
template <typename KeyT>
static inline size_t workspace_bytes(int64_t n) {
  if (n < 0) n = 0;
  return align256((size_t)n * sizeof(KeyT)) + align256((size_t)n * 4) + control_bytes(n);
}

// Measured on MI355X (profiles/r01_onesweep.md): the chained scan loses here.  A status word
// crosses XCDs at ~1-2 us per hop and the ~770 tiles of the first wave all start together, so the
// look-back chain costs more than the two small launches it removes (tile sort of 7.2M pairs:
// 0.225 ms vs 0.181 ms).  Kept behind FG_SORT_ONESWEEP=1 for re-measurement on other sizes.
static inline bool use_onesweep() {
  static const bool v = [] {
    const char* e = getenv("FG_SORT_ONESWEEP");
    return e && e[0] == '1';
  }();
  return v;
}

// Sorts in place (result copied back into keys/vals if it ends in the scratch copy).
// With n_dev != nullptr the element count is min(n, *n_dev), read on the device: n is then only the
// capacity the launch grids and the scratch are sized for (no host round trip for the count).
template <typename KeyT>
static inline int sort_pairs(int64_t n, KeyT* keys, uint32_t* vals, int end_bit, void* workspace,
                             size_t ws_bytes, hipStream_t s, const int64_t* n_dev = nullptr) {
  if (n <= 1 || end_bit <= 0) return FG_OK;
  if (n > 0xFFFFFFFFll) return FG_ERR_UNSUPPORTED;
  if (ws_bytes < workspace_bytes<KeyT>(n)) return FG_ERR_WORKSPACE;
  const int nb = num_blocks(n);
  char* ws = static_cast<char*>(workspace);
  KeyT* keys_alt = reinterpret_cast<KeyT*>(ws);
  ws += align256((size_t)n * sizeof(KeyT));
  uint32_t* vals_alt = reinterpret_cast<uint32_t*>(ws);
  ws += align256((size_t)n * 4);
  uint32_t* control = reinterpret_cast<uint32_t*>(ws);

  KeyT *kin = keys, *kout = keys_alt;
  uint32_t *vin = vals, *vout = vals_alt;
  const int passes = (end_bit + RADIX_BITS - 1) / RADIX_BITS;
  if (passes > MAX_PASSES) return FG_ERR_UNSUPPORTED;
  if (use_onesweep()) {
    uint32_t* ghist = control;
    uint32_t* tickets = ghist + MAX_PASSES * RADIX;
    uint32_t* status = tickets + MAX_PASSES;
    const size_t used = ((size_t)MAX_PASSES * RADIX + MAX_PASSES + (size_t)passes * nb * RADIX) * 4;
    if (hipMemsetAsync(control, 0, used, s) != hipSuccess) return FG_ERR_LAUNCH;
    hipLaunchKernelGGL(global_hist_kernel<KeyT>, dim3(nb), dim3(BLOCK), 0, s, n, n_dev, kin, passes, ghist);
    for (int p = 0; p < passes; ++p) {
      hipLaunchKernelGGL(onesweep_kernel<KeyT>, dim3(nb), dim3(BLOCK), 0, s, n, n_dev, kin, vin, kout, vout,
                         p * RADIX_BITS, ghist + p * RADIX, status + (size_t)p * nb * RADIX, tickets + p);
      KeyT* tk = kin; kin = kout; kout = tk;
      uint32_t* tv = vin; vin = vout; vout = tv;
    }
  } else {
    uint32_t* block_hist = control;
    uint32_t* digit_total = control + ((size_t)nb + 1) * RADIX;
    for (int p = 0; p < passes; ++p) {
      const int shift = p * RADIX_BITS;
      hipLaunchKernelGGL(hist_kernel<KeyT>, dim3(nb), dim3(BLOCK), 0, s, n, n_dev, kin, shift, block_hist);
      hipLaunchKernelGGL(digit_scan_kernel, dim3(RADIX), dim3(BLOCK), 0, s, nb, block_hist, digit_total);
      hipLaunchKernelGGL(scatter_kernel<KeyT>, dim3(nb), dim3(BLOCK), 0, s, n, n_dev, kin, vin, kout, vout, shift,
                         block_hist, digit_total);
      KeyT* tk = kin; kin = kout; kout = tk;
      uint32_t* tv = vin; vin = vout; vout = tv;
    }
  }
  if (kin != keys) {
    if (hipMemcpyAsync(keys, kin, (size_t)n * sizeof(KeyT), hipMemcpyDeviceToDevice, s) != hipSuccess)
      return FG_ERR_LAUNCH;
    if (hipMemcpyAsync(vals, vin, (size_t)n * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return FG_ERR_LAUNCH;
  }
  return hipGetLastError() == hipSuccess ? FG_OK : FG_ERR_LAUNCH;
}

}  // namespace fg_sort
