// K3: tile binning -- prefix scan of per-Gaussian tile counts, (tile|depth, id) emission and
// per-tile ranges of the sorted list.  Integer path: results are bit-exact by construction.
//
// Replaces the binning stage implied by tile_size=16 at
// /root/reference freegaussian/freegaussian_model.py:806,857.
#include "fg_common.h"
#include "radix_sort.h"

namespace {

constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_ITEMS = 8;                       // per thread
constexpr int SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;  // 2048 elements per workgroup

// inclusive scan across the 64 lanes of a wave (DPP row_shr + row broadcasts through SGPRs)
__device__ __forceinline__ int64_t wave_inclusive_scan(int64_t v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int64_t o = __shfl_up(v, d);
    if (fg::lane_id() >= d) v += o;
  }
  return v;
}

// workgroup-wide inclusive scan of one value per thread; returns the thread's inclusive
// prefix and the workgroup total.
__device__ __forceinline__ int64_t block_inclusive_scan(int64_t v, int64_t* wave_sums, int64_t& total) {
  const int lane = fg::lane_id(), wave = threadIdx.x >> 6;
  const int64_t incl = wave_inclusive_scan(v);
  if (lane == 63) wave_sums[wave] = incl;
  __syncthreads();
  int64_t base = 0;
  int64_t tot = 0;
#pragma unroll
  for (int w = 0; w < SCAN_BLOCK / 64; ++w) {
    const int64_t s = wave_sums[w];
    if (w < wave) base += s;
    tot += s;
  }
  total = tot;
  __syncthreads();
  return incl + base;
}

// pass 1: per-workgroup totals
__global__ void __launch_bounds__(SCAN_BLOCK)
scan_reduce_kernel(int N, const int32_t* __restrict__ in, const int32_t* __restrict__ order,
                   int64_t* __restrict__ block_sums, int32_t* __restrict__ gathered,
                   const int2* __restrict__ rect2, int32_t* __restrict__ rects_sorted) {
  __shared__ int64_t wave_sums[SCAN_BLOCK / 64];
  const int base = blockIdx.x * SCAN_TILE;
  int64_t s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    const int i = base + k * SCAN_BLOCK + threadIdx.x;
    if (i < N) {
      int32_t v;
      if (rect2) {
        // ONE 8-byte gather per Gaussian yields its tile rectangle: the count is width x height
        // and the emission kernel later reads the rectangle in depth order, coalesced (its own
        // gathers of radii / means2d by id were 15 of its 35 us)
        const int2 r = rect2[order[i]];  // {x0 | y0 << 16, w | h << 16}
        const int w = r.y & 0xFFFF, h = r.y >> 16;
        v = w * h;
        rects_sorted[i] = (r.x & 0xFFFF) | ((r.x >> 16) << 10) | (w << 20);
      } else {
        v = in[order ? order[i] : i];
      }
      s += v;
      if (gathered) gathered[i] = v;  // the second pass then reads in order instead of gathering again
    }
  }
  int64_t total;
  block_inclusive_scan(s, wave_sums, total);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// pass 2: exclusive scan of the workgroup totals, one workgroup
__global__ void __launch_bounds__(SCAN_BLOCK)
scan_blocksums_kernel(int nblocks, int64_t* __restrict__ block_sums) {
  __shared__ int64_t wave_sums[SCAN_BLOCK / 64];
  int64_t carry = 0;
  for (int base = 0; base < nblocks; base += SCAN_BLOCK) {
    const int i = base + threadIdx.x;
    const int64_t v = (i < nblocks) ? block_sums[i] : 0;
    int64_t total;
    const int64_t incl = block_inclusive_scan(v, wave_sums, total);
    if (i < nblocks) block_sums[i] = carry + incl - v;
    carry += total;
  }
}

// pass 3: inclusive scan inside each workgroup, offset by the scanned totals.  Thread t owns
// SCAN_ITEMS consecutive elements so the sequential order matches the array order.
// RAW_TOTALS: block_sums holds the per-workgroup totals of pass 1 as they are and every workgroup
// sums its predecessors itself (up to FG_SCAN_FUSED_MAX of them): one launch fewer.
template <bool RAW_TOTALS>
__global__ void __launch_bounds__(SCAN_BLOCK)
scan_apply_kernel(int N, const int32_t* __restrict__ in, const int32_t* __restrict__ order,
                  const int64_t* __restrict__ block_sums, int64_t* __restrict__ out, int64_t* count_out) {
  __shared__ int64_t wave_sums[SCAN_BLOCK / 64];
  int64_t before = 0;
  if (RAW_TOTALS) {
    int64_t part = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += SCAN_BLOCK) part += block_sums[b];
    int64_t tot;
    block_inclusive_scan(part, wave_sums, tot);
    before = tot;
  } else {
    before = block_sums[blockIdx.x];
  }
  const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  int32_t v[SCAN_ITEMS];
  int64_t s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    v[k] = (base + k < N) ? in[order ? order[base + k] : base + k] : 0;
    s += v[k];
  }
  int64_t total;
  const int64_t incl = block_inclusive_scan(s, wave_sums, total);
  int64_t run = before + incl - s;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    run += v[k];
    if (base + k < N) out[base + k] = run;
    // the grand total also goes to the caller's (host-visible) word: no copy launch for the read-back
    if (count_out && base + k == N - 1) {
      __hip_atomic_store(count_out, run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __threadfence_system();
    }
  }
}
constexpr int FG_SCAN_FUSED_MAX = 2048;  // workgroups; beyond that the one-workgroup middle pass is cheaper
// gathered (optional, N int32 of scratch): with an `order`, pass 1 leaves the gathered values there
// and pass 2 reads them back coalesced instead of repeating the random gather
void launch_scan(int N, const int32_t* in, const int32_t* order, int64_t* block_sums, int64_t* out, hipStream_t s,
                 int32_t* gathered = nullptr, const int2* rect2 = nullptr, int32_t* rects_sorted = nullptr,
                 int64_t* count_out = nullptr) {
  const int nblocks = (N + SCAN_TILE - 1) / SCAN_TILE;
  if (!order) gathered = nullptr;
  hipLaunchKernelGGL(scan_reduce_kernel, dim3(nblocks), dim3(SCAN_BLOCK), 0, s, N, in, order, block_sums, gathered,
                     rect2, rects_sorted);
  if (gathered) {
    in = gathered;
    order = nullptr;
  }
  if (nblocks <= FG_SCAN_FUSED_MAX) {
    hipLaunchKernelGGL(scan_apply_kernel<true>, dim3(nblocks), dim3(SCAN_BLOCK), 0, s, N, in, order, block_sums, out,
                       count_out);
  } else {
    hipLaunchKernelGGL(scan_blocksums_kernel, dim3(1), dim3(SCAN_BLOCK), 0, s, nblocks, block_sums);
    hipLaunchKernelGGL(scan_apply_kernel<false>, dim3(nblocks), dim3(SCAN_BLOCK), 0, s, N, in, order, block_sums, out,
                       count_out);
  }
}

// One lane per Gaussian writes its (key, id) pairs, row-major over its tile rectangle.
__global__ void __launch_bounds__(256)
tile_bin_kernel(int N, const float* __restrict__ means2d, const int32_t* __restrict__ radii,
                const float* __restrict__ depths, const int64_t* __restrict__ cum_tiles, int tile_size,
                int tile_w, int tile_h, int64_t* __restrict__ isect_ids, int32_t* __restrict__ flatten_ids) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int radius = radii[i];
  if (radius <= 0) return;
  const float ts = (float)tile_size;
  const float r = (float)radius / ts;
  const float tx = means2d[2 * i] / ts, ty = means2d[2 * i + 1] / ts;
  const int x0 = min(max((int)floorf(tx - r), 0), tile_w), x1 = min(max((int)ceilf(tx + r), 0), tile_w);
  const int y0 = min(max((int)floorf(ty - r), 0), tile_h), y1 = min(max((int)ceilf(ty + r), 0), tile_h);
  int64_t cur = (i == 0) ? 0 : cum_tiles[i - 1];
  const int64_t dbits = (int64_t)(uint32_t)__float_as_int(depths[i]);
  for (int y = y0; y < y1; ++y)
    for (int x = x0; x < x1; ++x) {
      const int64_t tile = (int64_t)y * tile_w + x;
      isect_ids[cur] = (tile << 32) | dbits;
      flatten_ids[cur] = i;
      ++cur;
    }
}

// tile_offsets[t] = first sorted index whose tile id is >= t; tile_offsets[n_tiles] = n.
__global__ void __launch_bounds__(256)
tile_ranges_kernel(int64_t n, const int64_t* __restrict__ keys, int n_tiles, int32_t* __restrict__ offsets) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n == 0) {
    if (i <= n_tiles) offsets[i] = 0;
    return;
  }
  if (i >= n) return;
  const int cur = (int)(keys[i] >> 32);
  if (i == 0) {
    for (int t = 0; t <= cur; ++t) offsets[t] = 0;
  } else {
    const int prev = (int)(keys[i - 1] >> 32);
    for (int t = prev + 1; t <= cur; ++t) offsets[t] = (int32_t)i;
  }
  if (i == n - 1) {
    for (int t = cur + 1; t <= n_tiles; ++t) offsets[t] = (int32_t)n;
  }
}

// ---- depth-first binning (DESIGN.md "binning without a 45-bit sort") ------------------------
// key = float bits of depth for visible Gaussians (depth > 0: integer order == float order),
// 0xFFFFFFFF for culled ones; value = Gaussian id.
__global__ void __launch_bounds__(256)
depth_keys_kernel(int N, const float* __restrict__ depths, const int32_t* __restrict__ radii,
                  uint32_t* __restrict__ keys, int32_t* __restrict__ order) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  keys[i] = radii[i] > 0 ? (uint32_t)__float_as_int(depths[i]) : 0xFFFFFFFFu;
  order[i] = i;
}

// The same plus the tile rectangle of every Gaussian as {x0 | y0 << 16, w | h << 16} (zeros when
// culled); the arithmetic is that of the preprocess pass, so w * h == tiles_touched.
__global__ void __launch_bounds__(256)
depth_keys_rects_kernel(int N, const float* __restrict__ depths, const int32_t* __restrict__ radii,
                        const float* __restrict__ means2d, int tile_size, int tile_w, int tile_h,
                        uint32_t* __restrict__ keys, int32_t* __restrict__ order, int2* __restrict__ rect2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int radius = radii[i];
  keys[i] = radius > 0 ? (uint32_t)__float_as_int(depths[i]) : 0xFFFFFFFFu;
  order[i] = i;
  int2 r = make_int2(0, 0);
  if (radius > 0) {
    const float ts = (float)tile_size;
    const float rr = (float)radius / ts;
    const float tx = means2d[2 * i] / ts, ty = means2d[2 * i + 1] / ts;
    const int x0 = min(max((int)floorf(tx - rr), 0), tile_w), x1 = min(max((int)ceilf(tx + rr), 0), tile_w);
    const int y0 = min(max((int)floorf(ty - rr), 0), tile_h), y1 = min(max((int)ceilf(ty + rr), 0), tile_h);
    r = make_int2(x0 | (y0 << 16), (x1 - x0) | ((y1 - y0) << 16));
  }
  rect2[i] = r;
}

// Emission in depth order, wave-cooperative: the 64 Gaussians of a wavefront own ONE contiguous
// output range [cum[k0-1], cum[k0+63]) (cum is the ordered inclusive scan), so the wave walks that
// range 64 slots at a time -- every store instruction writes 64 consecutive (tile, id) pairs --
// and each lane finds the splat that owns its slot by a 6-step binary search over the wave's
// exclusive offsets (kept in LDS).  A lane-per-Gaussian loop wrote 8-byte pieces at 64 unrelated
// addresses per instruction instead.
template <typename KeyT>
__global__ void __launch_bounds__(256)
tile_bin_ordered_kernel(int N, const float* __restrict__ means2d, const int32_t* __restrict__ radii,
                        const int32_t* __restrict__ order, const int64_t* __restrict__ cum_tiles, int tile_size,
                        int tile_w, int tile_h, KeyT* __restrict__ tile_keys,
                        int32_t* __restrict__ flatten_ids, int64_t capacity,
                        const int32_t* __restrict__ rects_sorted) {
  __shared__ int32_t s_excl[4][64];  // exclusive slot offset of each lane's splat inside the wave's range
  __shared__ int32_t s_gid[4][64];
  __shared__ int32_t s_rect[4][64];  // x0 | y0 << 10 | width << 20
  const int lane = fg::lane_id(), wave = threadIdx.x >> 6;
  const int k0 = (blockIdx.x * 4 + wave) * 64;
  if (k0 >= N) return;
  const int k = k0 + lane;
  const int64_t base = (k0 == 0) ? 0 : cum_tiles[k0 - 1];
  const int klast = min(k0 + 63, N - 1);
  const int total = (int)(cum_tiles[klast] - base);
  int gid = 0, rect = 0;
  int excl = total;  // lanes past N own nothing
  if (k < N) {
    gid = order[k];
    excl = (int)(((k == 0) ? 0 : cum_tiles[k - 1]) - base);
    if (rects_sorted) {
      rect = rects_sorted[k];  // in depth order from fg_bin_prepare_rects: no gather by id
    } else if (const int radius = radii[gid]; radius > 0) {
      const float ts = (float)tile_size;
      const float r = (float)radius / ts;
      const float tx = means2d[2 * gid] / ts, ty = means2d[2 * gid + 1] / ts;
      const int x0 = min(max((int)floorf(tx - r), 0), tile_w), x1 = min(max((int)ceilf(tx + r), 0), tile_w);
      const int y0 = min(max((int)floorf(ty - r), 0), tile_h);
      rect = x0 | (y0 << 10) | ((x1 - x0) << 20);
    }
  }
  s_excl[wave][lane] = excl;
  s_gid[wave][lane] = gid;
  s_rect[wave][lane] = rect;
  __builtin_amdgcn_wave_barrier();  // same-wave LDS traffic only: program order suffices
  for (int s0 = 0; s0 < total; s0 += 64) {
    const int slot = s0 + lane;
    if (slot < total) {
      // owner = last lane whose exclusive offset is <= slot (empty splats share their successor's
      // offset and are skipped by taking the LAST such lane)
      int lo = 0;
#pragma unroll
      for (int step = 32; step > 0; step >>= 1)
        if (s_excl[wave][lo + step] <= slot) lo += step;
      const int t = slot - s_excl[wave][lo];
      const int rc = s_rect[wave][lo];
      const int w = rc >> 20;
      // t / w without the ~30-instruction integer division: t < 2^20 is exact in fp32, one fix-up each way
      int ty = (int)((float)t * __builtin_amdgcn_rcpf((float)w));
      ty -= (ty * w > t);
      ty += ((ty + 1) * w <= t);
      const int tx = t - ty * w;
      const int64_t out = base + slot;
      if (out >= capacity) continue;  // capacity launch that guessed too low: the host redoes it
      tile_keys[out] = (KeyT)((uint32_t)(((rc >> 10) & 1023) + ty) * (uint32_t)tile_w + (uint32_t)((rc & 1023) + tx));
      flatten_ids[out] = s_gid[wave][lo];
    }
  }
}

// tile_offsets from the sorted tile keys; 16 bytes of consecutive keys per thread (4 x u32 or 8 x u16)
// + the predecessor's last key.
template <typename KeyT>
__global__ void __launch_bounds__(256)
tile_ranges32_kernel(int64_t n, const int64_t* __restrict__ n_dev, const KeyT* __restrict__ keys, int n_tiles,
                     int32_t* __restrict__ offsets) {
  constexpr int KPT = 16 / sizeof(KeyT);
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n_dev) n = min(n, *n_dev);
  if (n == 0) {
    for (int64_t t = KPT * q; t < KPT * q + KPT; ++t)
      if (t <= n_tiles) offsets[t] = 0;
    return;
  }
  const int64_t i0 = KPT * q;
  if (i0 >= n) return;
  KeyT k[KPT];
  if (i0 + KPT <= n) {
    *reinterpret_cast<uint4*>(k) = reinterpret_cast<const uint4*>(keys)[q];
  } else {
#pragma unroll
    for (int j = 0; j < KPT; ++j) k[j] = i0 + j < n ? keys[i0 + j] : (KeyT)0;
  }
  int prev = i0 == 0 ? -1 : (int)keys[i0 - 1];
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    const int64_t i = i0 + j;
    if (i >= n) break;
    const int cur = (int)k[j];
    for (int t = prev + 1; t <= cur; ++t) offsets[t] = (int32_t)i;  // i == 0: t = 0..cur get 0
    prev = cur;
    if (i == n - 1)
      for (int t = cur + 1; t <= n_tiles; ++t) offsets[t] = (int32_t)n;
  }
}

__global__ void __launch_bounds__(256)
isect_keys_kernel(int64_t n, const uint32_t* __restrict__ tile_keys, const int32_t* __restrict__ flatten_ids,
                  const float* __restrict__ depths, int64_t* __restrict__ isect_ids) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  isect_ids[i] = ((int64_t)tile_keys[i] << 32) | (int64_t)(uint32_t)__float_as_int(depths[flatten_ids[i]]);
}

}  // namespace

extern "C" size_t fg_scan_workspace_bytes(int N) {
  const size_t nblocks = ((size_t)(N > 0 ? N : 1) + SCAN_TILE - 1) / SCAN_TILE;
  return nblocks * sizeof(int64_t);
}

extern "C" int fg_scan_tiles(int N, const int32_t* tiles_touched, int64_t* cum_tiles, void* workspace,
                             size_t workspace_bytes, fg_stream_t stream) {
  if (N < 0) return FG_ERR_INVALID_ARG;
  if (N == 0) return FG_OK;
  if (!tiles_touched || !cum_tiles || !workspace) return FG_ERR_INVALID_ARG;
  if (workspace_bytes < fg_scan_workspace_bytes(N)) return FG_ERR_WORKSPACE;
  int64_t* block_sums = static_cast<int64_t*>(workspace);
  hipStream_t s = fg_hip_stream(stream);
  launch_scan(N, tiles_touched, nullptr, block_sums, cum_tiles, s);
  FG_RETURN_IF_LAUNCH_FAILED();
  return FG_OK;
}

extern "C" int fg_tile_bin(int N, const float* means2d, const int32_t* radii, const float* depths,
                           const int64_t* cum_tiles, int tile_size, int tile_w, int tile_h,
                           int64_t* isect_ids, int32_t* flatten_ids, fg_stream_t stream) {
  if (N < 0 || tile_size <= 0 || tile_w <= 0 || tile_h <= 0) return FG_ERR_INVALID_ARG;
  if (N == 0) return FG_OK;
  if (!means2d || !radii || !depths || !cum_tiles) return FG_ERR_INVALID_ARG;
  hipLaunchKernelGGL(tile_bin_kernel, dim3((N + 255) / 256), dim3(256), 0, fg_hip_stream(stream), N, means2d,
                     radii, depths, cum_tiles, tile_size, tile_w, tile_h, isect_ids, flatten_ids);
  FG_RETURN_IF_LAUNCH_FAILED();
  return FG_OK;
}

extern "C" int fg_tile_ranges(int64_t n, const int64_t* sorted_keys, int n_tiles, int32_t* tile_offsets,
                              fg_stream_t stream) {
  if (n < 0 || n_tiles <= 0 || !tile_offsets) return FG_ERR_INVALID_ARG;
  if (n > 0 && !sorted_keys) return FG_ERR_INVALID_ARG;
  const int64_t work = n > 0 ? n : (int64_t)n_tiles + 1;
  hipLaunchKernelGGL(tile_ranges_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0,
                     fg_hip_stream(stream), n, sorted_keys, n_tiles, tile_offsets);
  FG_RETURN_IF_LAUNCH_FAILED();
  return FG_OK;
}

// ---- depth-first binning entry points --------------------------------------------------------
static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" size_t fg_bin_prepare_workspace_bytes(int N) {
  const size_t n = (size_t)(N > 0 ? N : 1);
  return al256(n * 4) + fg_sort::workspace_bytes<uint32_t>((int64_t)n) + al256(fg_scan_workspace_bytes(N)) +
         al256(n * 8);  // + the rectangle pairs of fg_bin_prepare_rects
}

namespace {
// rects_sorted == nullptr: counts come from tiles_touched (fg_bin_prepare); otherwise from the
// rectangles computed here, which also go out in depth order for the emission kernel.
int bin_prepare_any(int N, const float* depths, const int32_t* radii, const int32_t* tiles_touched,
                    const float* means2d, int tile_size, int tile_w, int tile_h, int32_t* order, int64_t* cum_tiles,
                    int32_t* rects_sorted, void* workspace, size_t workspace_bytes, fg_stream_t stream) {
  if (N < 0) return FG_ERR_INVALID_ARG;
  if (N == 0) return FG_OK;
  if (!depths || !radii || !order || !cum_tiles || !workspace) return FG_ERR_INVALID_ARG;
  if (rects_sorted ? (!means2d || tile_size <= 0 || tile_w <= 0 || tile_h <= 0) : !tiles_touched) return FG_ERR_INVALID_ARG;
  if (rects_sorted && (tile_w > 1023 || tile_h > 1023)) return FG_ERR_UNSUPPORTED;  // rectangle packing
  if (workspace_bytes < fg_bin_prepare_workspace_bytes(N)) return FG_ERR_WORKSPACE;
  hipStream_t s = fg_hip_stream(stream);
  char* ws = static_cast<char*>(workspace);
  uint32_t* keys = reinterpret_cast<uint32_t*>(ws);
  ws += al256((size_t)N * 4);
  void* sort_ws = ws;
  const size_t sort_bytes = fg_sort::workspace_bytes<uint32_t>(N);
  ws += sort_bytes;
  int64_t* block_sums = reinterpret_cast<int64_t*>(ws);
  ws += al256(fg_scan_workspace_bytes(N));
  int2* rect2 = reinterpret_cast<int2*>(ws);
  if (rects_sorted)
    hipLaunchKernelGGL(depth_keys_rects_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, depths, radii, means2d,
                       tile_size, tile_w, tile_h, keys, order, rect2);
  else
    hipLaunchKernelGGL(depth_keys_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, depths, radii, keys, order);
  const int rc = fg_sort::sort_pairs<uint32_t>(N, keys, reinterpret_cast<uint32_t*>(order), 32, sort_ws, sort_bytes, s);
  if (rc != FG_OK) return rc;
  // the sort's alternate key buffer is free again: scratch for the gathered tile counts
  launch_scan(N, tiles_touched, (const int32_t*)order, block_sums, cum_tiles, s, reinterpret_cast<int32_t*>(sort_ws),
              rects_sorted ? rect2 : nullptr, rects_sorted);
  FG_RETURN_IF_LAUNCH_FAILED();
  return FG_OK;
}
}  // namespace

extern "C" int fg_bin_prepare(int N, const float* depths, const int32_t* radii, const int32_t* tiles_touched,
                              int32_t* order, int64_t* cum_tiles, void* workspace, size_t workspace_bytes,
                              fg_stream_t stream) {
  return bin_prepare_any(N, depths, radii, tiles_touched, nullptr, 0, 0, 0, order, cum_tiles, nullptr, workspace,
                         workspace_bytes, stream);
}

extern "C" int fg_bin_prepare_rects(int N, const float* depths, const int32_t* radii, const float* means2d,
                                    int tile_size, int tile_w, int tile_h, int32_t* order, int64_t* cum_tiles,
                                    int32_t* rects_sorted, void* workspace, size_t workspace_bytes,
                                    fg_stream_t stream) {
  if (!rects_sorted) return FG_ERR_INVALID_ARG;
  return bin_prepare_any(N, depths, radii, nullptr, means2d, tile_size, tile_w, tile_h, order, cum_tiles, rects_sorted,
                         workspace, workspace_bytes, stream);
}

extern "C" int fg_bin_prepare_keys(int N, uint32_t* depth_keys, const int32_t* tile_rects, int32_t* order,
                                   int64_t* cum_tiles, int32_t* rects_sorted, int64_t* count_out, void* workspace,
                                   size_t workspace_bytes, fg_stream_t stream) {
  if (N < 0) return FG_ERR_INVALID_ARG;
  if (N == 0) return FG_OK;
  if (!depth_keys || !tile_rects || !order || !cum_tiles || !rects_sorted || !workspace) return FG_ERR_INVALID_ARG;
  if (workspace_bytes < fg_bin_prepare_workspace_bytes(N)) return FG_ERR_WORKSPACE;
  hipStream_t s = fg_hip_stream(stream);
  char* ws = static_cast<char*>(workspace);
  ws += al256((size_t)N * 4);  // (the key array of the other entry points: unused here)
  void* sort_ws = ws;
  const size_t sort_bytes = fg_sort::workspace_bytes<uint32_t>(N);
  ws += sort_bytes;
  int64_t* block_sums = reinterpret_cast<int64_t*>(ws);
  const int rc = fg_sort::sort_pairs<uint32_t>(N, depth_keys, reinterpret_cast<uint32_t*>(order), 32, sort_ws,
                                               sort_bytes, s, nullptr, /*iota_vals=*/true);
  if (rc != FG_OK) return rc;
  launch_scan(N, nullptr, (const int32_t*)order, block_sums, cum_tiles, s, reinterpret_cast<int32_t*>(sort_ws),
              reinterpret_cast<const int2*>(tile_rects), rects_sorted, count_out);
  FG_RETURN_IF_LAUNCH_FAILED();
  return FG_OK;
}

extern "C" size_t fg_bin_emit_workspace_bytes(int64_t n_isects) {
  // covers both layouts: 32-bit keys in the caller's buffer, or 16-bit keys + their sort scratch
  // in here (same byte count up to alignment padding)
  return fg_sort::workspace_bytes<uint32_t>(n_isects > 0 ? n_isects : 1) + 1024;
}

namespace {

// n_dev == nullptr: exactly n_isects intersections.  Otherwise n_isects is a capacity and the
// count is read on the device from *n_dev (clamped to the capacity).
template <typename KeyT>
int bin_emit_sort_keys(int N, int64_t n_isects, const int64_t* n_dev, const float* means2d, const int32_t* radii,
                       const int32_t* order, const int64_t* cum_tiles, int tile_size, int tile_w, int tile_h,
                       KeyT* tile_keys, int32_t* flatten_ids, int32_t* tile_offsets, void* sort_ws, size_t sort_bytes,
                       hipStream_t s, const int32_t* rects_sorted) {
  const int n_tiles = tile_w * tile_h;
  if (n_isects > 0) {
    hipLaunchKernelGGL(tile_bin_ordered_kernel<KeyT>, dim3((N + 255) / 256), dim3(256), 0, s, N, means2d, radii, order,
                       cum_tiles, tile_size, tile_w, tile_h, tile_keys, flatten_ids, n_isects, rects_sorted);
    int bits = 1;
    while ((1 << bits) < n_tiles) ++bits;
    const int rc = fg_sort::sort_pairs<KeyT>(n_isects, tile_keys, reinterpret_cast<uint32_t*>(flatten_ids), bits, sort_ws,
                                             sort_bytes, s, n_dev);
    if (rc != FG_OK) return rc;
  }
  constexpr int KPT = 16 / sizeof(KeyT);
  const int64_t work = ((n_isects > (int64_t)n_tiles + 1 ? n_isects : (int64_t)n_tiles + 1) + KPT - 1) / KPT;
  hipLaunchKernelGGL(tile_ranges32_kernel<KeyT>, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, s, n_isects, n_dev,
                     tile_keys, n_tiles, tile_offsets);
  return hipGetLastError() == hipSuccess ? FG_OK : FG_ERR_LAUNCH;
}

// tile_keys == nullptr: the caller does not want the keys; they then live in the workspace as
// 16-bit values (tile ids < 65536), which cuts the traffic of emission + two sort passes from 48
// to 34 bytes per intersection.
int bin_emit_sort_any(int N, int64_t n_isects, const int64_t* n_dev, const float* means2d, const int32_t* radii,
                      const int32_t* order, const int64_t* cum_tiles, int tile_size, int tile_w, int tile_h,
                      uint32_t* tile_keys, int32_t* flatten_ids, int32_t* tile_offsets, void* workspace,
                      size_t workspace_bytes, fg_stream_t stream, const int32_t* rects_sorted) {
  if (N < 0 || n_isects < 0 || tile_size <= 0 || tile_w <= 0 || tile_h <= 0 || !tile_offsets) return FG_ERR_INVALID_ARG;
  if (tile_w > 1023 || tile_h > 1023) return FG_ERR_UNSUPPORTED;  // rectangle packing of the emit kernel
  hipStream_t s = fg_hip_stream(stream);
  if (n_isects > 0 && (!order || !cum_tiles || !flatten_ids || !workspace)) return FG_ERR_INVALID_ARG;
  if (n_isects > 0 && !rects_sorted && (!means2d || !radii)) return FG_ERR_INVALID_ARG;
  if (workspace_bytes < fg_bin_emit_workspace_bytes(n_isects)) return FG_ERR_WORKSPACE;
  if (tile_keys || tile_w * tile_h > 65536) {
    if (n_isects > 0 && !tile_keys) return FG_ERR_INVALID_ARG;  // > 65536 tiles: 32-bit keys, caller's buffer
    return bin_emit_sort_keys<uint32_t>(N, n_isects, n_dev, means2d, radii, order, cum_tiles, tile_size, tile_w, tile_h,
                                        tile_keys, flatten_ids, tile_offsets, workspace, workspace_bytes, s, rects_sorted);
  }
  char* ws = static_cast<char*>(workspace);
  uint16_t* keys16 = reinterpret_cast<uint16_t*>(ws);
  const size_t head = al256((size_t)(n_isects > 0 ? n_isects : 1) * 2);
  return bin_emit_sort_keys<uint16_t>(N, n_isects, n_dev, means2d, radii, order, cum_tiles, tile_size, tile_w, tile_h,
                                      keys16, flatten_ids, tile_offsets, ws + head, workspace_bytes - head, s, rects_sorted);
}

}  // namespace

extern "C" int fg_bin_emit_sort(int N, int64_t n_isects, const float* means2d, const int32_t* radii,
                                const int32_t* order, const int64_t* cum_tiles, const int32_t* rects_sorted,
                                int tile_size, int tile_w, int tile_h, uint32_t* tile_keys, int32_t* flatten_ids,
                                int32_t* tile_offsets, void* workspace, size_t workspace_bytes, fg_stream_t stream) {
  return bin_emit_sort_any(N, n_isects, nullptr, means2d, radii, order, cum_tiles, tile_size, tile_w, tile_h,
                           tile_keys, flatten_ids, tile_offsets, workspace, workspace_bytes, stream, rects_sorted);
}

extern "C" int fg_bin_emit_sort_capacity(int N, int64_t capacity, const float* means2d, const int32_t* radii,
                                         const int32_t* order, const int64_t* cum_tiles,
                                         const int32_t* rects_sorted, int tile_size, int tile_w, int tile_h,
                                         uint32_t* tile_keys, int32_t* flatten_ids, int32_t* tile_offsets,
                                         void* workspace, size_t workspace_bytes, fg_stream_t stream) {
  if (N <= 0 || capacity <= 0 || !cum_tiles) return FG_ERR_INVALID_ARG;
  return bin_emit_sort_any(N, capacity, cum_tiles + (N - 1), means2d, radii, order, cum_tiles, tile_size, tile_w,
                           tile_h, tile_keys, flatten_ids, tile_offsets, workspace, workspace_bytes, stream,
                           rects_sorted);
}

extern "C" int fg_isect_keys(int64_t n_isects, const uint32_t* tile_keys, const int32_t* flatten_ids,
                             const float* depths, int64_t* isect_ids, fg_stream_t stream) {
  if (n_isects < 0) return FG_ERR_INVALID_ARG;
  if (n_isects == 0) return FG_OK;
  if (!tile_keys || !flatten_ids || !depths || !isect_ids) return FG_ERR_INVALID_ARG;
  hipLaunchKernelGGL(isect_keys_kernel, dim3((unsigned)((n_isects + 255) / 256)), dim3(256), 0, fg_hip_stream(stream),
                     n_isects, tile_keys, flatten_ids, depths, isect_ids);
  FG_RETURN_IF_LAUNCH_FAILED();
  return FG_OK;
}
