// K3 + K4, banded form: count per tile -> scan -> scatter into per-tile segments -> every tile's
// segment sorted by (depth bits, Gaussian id) inside LDS.  Same lists as the reference algorithm
// (tile | depth keys, stable sort: /root/reference freegaussian/freegaussian_model.py:847-868 runs
// gsplat's isect_tiles + radix sort + isect_offset_encode behind `tile_size=16`), bit for bit -- the
// order inside a tile is a total order on (depth bits, id), so HOW the entries reach the tile does not
// matter -- in 5 launches instead of 27.
//
// Why this shape on MI355X (profiles/r03_banded_binning.md):
//  * a kernel boundary costs ~2.5 us and a kernel of this size another 3-6 us of ramp and tail, so the
//    depth-first binning (4-pass sort of N keys, ordered scan, emission, 2-pass tile sort, ranges: 26
//    launches of 5-30 us) was bound by launch count, not by bytes;
//  * the eight XCDs have private L2s.  Every XCD owns a BAND of tile rows (the same bands the raster
//    kernels walk): its workgroups count, scatter and sort only pairs of tiles in that band, so the
//    8-byte scattered stores into a tile's segment meet in ONE L2 and leave it as whole lines, and
//    the sort and the raster kernels of that XCD find the segment there;
//  * positions come from a table of per-(chunk, tile) counts (LDS histograms, written out coalesced)
//    and a column scan -- no global atomics (they run at ~60 G/s on this chip: 85 us for 5M pairs);
//    inside a workgroup the cursor of a tile is an LDS word bumped with a returning atomic.
#include "fg_common.h"

namespace {

constexpr int TB_BLOCK = 256;
constexpr int TB_CHUNK = 4096;                       // Gaussians per (chunk, band) workgroup
constexpr int TB_ROUNDS = TB_CHUNK / TB_BLOCK;       // rounds of 64 Gaussians per wavefront
constexpr int TB_SORT_MAX = 2048;                    // entries of a tile sorted in LDS (larger: through global memory)
constexpr int TB_SORT_KPT = TB_SORT_MAX / TB_BLOCK;  // 8 entries per thread
constexpr int TB_MAX_BAND_TILES = 12288;             // LDS words of the count / cursor array

struct Rows {
  int r0, r1;
};
__host__ __device__ __forceinline__ Rows band_rows(int xcd, int tile_h) {
  Rows r;
  r.r0 = (xcd * tile_h) / 8;
  r.r1 = ((xcd + 1) * tile_h) / 8;
  return r;
}
int band_tiles_max(int tile_w, int tile_h) {
  int m = 0;
  for (int x = 0; x < 8; ++x) {
    const Rows r = band_rows(x, tile_h);
    m = (r.r1 - r.r0) * tile_w > m ? (r.r1 - r.r0) * tile_w : m;
  }
  return m;
}

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(v, d);
    if (lane >= d) v += o;
  }
  return v;
}

// ---- count ------------------------------------------------------------------------------------------
// Workgroup (xcd = blockIdx % 8, chunk = blockIdx / 8): per tile of the XCD's band, how many of the chunk's
// 4096 rectangles cover it.  No pair is enumerated: every rectangle (clipped to the band) adds +1 / -1 at
// its four corners of an LDS grid and two prefix sums (along x, then along y) turn the corners into counts
// -- work per rectangle, not per (rectangle, tile) pair.
__global__ void __launch_bounds__(TB_BLOCK)
tb_count_kernel(int N, const int2* __restrict__ rects, int tile_w, int tile_h, uint32_t* __restrict__ table) {
  extern __shared__ int32_t s_grid[];  // [(rows + 1)][tile_w + 1] corner marks -> counts
  const int xcd = blockIdx.x & 7, chunk = blockIdx.x >> 3;
  const Rows br = band_rows(xcd, tile_h);
  const int nr = br.r1 - br.r0, gw = tile_w + 1, T = tile_w * tile_h;
  if (nr <= 0) return;
  for (int i = threadIdx.x; i < (nr + 1) * gw; i += TB_BLOCK) s_grid[i] = 0;
  __syncthreads();
  const int g0 = chunk * TB_CHUNK + threadIdx.x;
  int2 rc[TB_ROUNDS];
#pragma unroll
  for (int r = 0; r < TB_ROUNDS; ++r) rc[r] = (g0 + r * TB_BLOCK < N) ? rects[g0 + r * TB_BLOCK] : make_int2(0, 0);
#pragma unroll
  for (int r = 0; r < TB_ROUNDS; ++r) {
    const int w = rc[r].y & 0xFFFF, h = rc[r].y >> 16, x0 = rc[r].x & 0xFFFF, y0 = rc[r].x >> 16;
    const int ya = max(y0, br.r0) - br.r0, yb = min(y0 + h, br.r1) - br.r0;
    if (w > 0 && yb > ya) {
      atomicAdd(&s_grid[ya * gw + x0], 1);
      atomicAdd(&s_grid[ya * gw + x0 + w], -1);
      atomicAdd(&s_grid[yb * gw + x0], -1);
      atomicAdd(&s_grid[yb * gw + x0 + w], 1);
    }
  }
  __syncthreads();
  // along x: wavefront w takes rows w, w + 4, ...; 64 cells per step with a running carry
  const int lane = fg::lane_id(), wave = threadIdx.x >> 6;
  for (int row = wave; row < nr; row += TB_BLOCK / 64) {
    int carry = 0;
    for (int x = 0; x < tile_w; x += 64) {
      const int v = x + lane < tile_w ? s_grid[row * gw + x + lane] : 0;
      const int incl = (int)wave_incl_scan_u32((uint32_t)v, lane) + carry;
      if (x + lane < tile_w) s_grid[row * gw + x + lane] = incl;
      carry = __builtin_amdgcn_readlane(incl, 63);
    }
  }
  __syncthreads();
  // along y, and out: thread = column
  uint32_t* out = table + (size_t)chunk * T + br.r0 * tile_w;
  for (int x = threadIdx.x; x < tile_w; x += TB_BLOCK) {
    int run = 0;
    for (int row = 0; row < nr; ++row) {
      run += s_grid[row * gw + x];
      out[row * tile_w + x] = (uint32_t)run;
    }
  }
}

// ---- scatter ----------------------------------------------------------------------------------------
// Workgroup (xcd = blockIdx % 8, chunk = blockIdx / 8): the (depth bits, id) pair of every (Gaussian, tile)
// pair of the chunk's 4096 Gaussians inside the XCD's band of tile rows, written to the tile's segment; the
// slot comes from an LDS cursor per tile (tile start + the chunks before this one, from the scanned table).  A wavefront reads 64 rectangles per round, keeps the ones that reach the band in
// a queue (7 of 8 do not) and, whenever 64 are queued, walks their (Gaussian, tile) pairs 64 at a time:
// every lane finds the owner of its slot by binary search over the wave's exclusive counts.
__global__ void __launch_bounds__(TB_BLOCK)
tb_scatter_kernel(int N, const int2* __restrict__ rects, const uint32_t* __restrict__ depth_keys, int tile_w, int tile_h,
               const uint32_t* __restrict__ table, const int32_t* __restrict__ tile_offsets, uint64_t* __restrict__ pairs,
               long long capacity) {
  extern __shared__ uint32_t s_cnt[];  // [tiles of the band]: cursors into the tile segments
  __shared__ int4 s_q[TB_BLOCK / 64][128];
  __shared__ int32_t s_excl[TB_BLOCK / 64][64];
  const int xcd = blockIdx.x & 7, chunk = blockIdx.x >> 3;
  const Rows br = band_rows(xcd, tile_h);
  const int tb = br.r0 * tile_w, nb = (br.r1 - br.r0) * tile_w, T = tile_w * tile_h;
  const uint32_t* row = table + (size_t)chunk * T + tb;
  if ((long long)tile_offsets[T] > capacity) return;  // the guess was too small: the host repeats the call
  for (int i = threadIdx.x; i < nb; i += TB_BLOCK) s_cnt[i] = (uint32_t)tile_offsets[tb + i] + row[i];
  __syncthreads();

  const int lane = fg::lane_id(), wave = threadIdx.x >> 6;
  const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  int4* q = s_q[wave];
  int32_t* ex = s_excl[wave];
  int qn = 0;  // queued survivors (wave-uniform)

  // the first m (<= 64) queued entries: every (Gaussian, tile) pair of theirs
  auto drain = [&](int m) {
    const int4 e = q[lane];
    const int w = e.w & 0xFFFF, hc = e.w >> 16;
    const uint32_t cnt = lane < m ? (uint32_t)(w * hc) : 0u;
    const uint32_t incl = wave_incl_scan_u32(cnt, lane);
    const int total = (int)__builtin_amdgcn_readlane((int)incl, 63);
    ex[lane] = lane < m ? (int)(incl - cnt) : total;  // lanes past m own nothing
    __builtin_amdgcn_wave_barrier();
    for (int s0 = 0; s0 < total; s0 += 64) {
      const int slot = s0 + lane;
      if (slot < total) {
        int lo = 0;
#pragma unroll
        for (int step = 32; step > 0; step >>= 1)
          if (ex[lo + step] <= slot) lo += step;
        const int4 o = q[lo];
        const int t = slot - ex[lo];
        const int ow = o.w & 0xFFFF;
        // t / ow without the integer division: t < 2^20 is exact in fp32, one fix-up each way
        int ty = (int)((float)t * __builtin_amdgcn_rcpf((float)ow));
        ty -= (ty * ow > t);
        ty += ((ty + 1) * ow <= t);
        const int tx = t - ty * ow;
        const int local = ((o.z >> 16) + ty - br.r0) * tile_w + (o.z & 0xFFFF) + tx;
        const uint32_t pos = atomicAdd(&s_cnt[local], 1u);
        pairs[pos] = ((uint64_t)(uint32_t)o.y << 32) | (uint64_t)(uint32_t)o.x;
      }
    }
    __builtin_amdgcn_wave_barrier();
  };
  auto pop64 = [&]() {  // drop the 64 drained entries, keep the rest at the front
    const int rest = qn - 64;
    int4 keep = make_int4(0, 0, 0, 0);
    if (lane < rest) keep = q[64 + lane];
    __builtin_amdgcn_wave_barrier();
    if (lane < rest) q[lane] = keep;
    __builtin_amdgcn_wave_barrier();
    qn = rest;
  };

  const int g0 = chunk * TB_CHUNK + wave * (TB_CHUNK / (TB_BLOCK / 64));
  for (int r = 0; r < TB_ROUNDS; ++r) {
    const int g = g0 + r * 64 + lane;
    bool hit = false;
    int4 e = make_int4(0, 0, 0, 0);
    if (g < N) {
      const int2 rc = rects[g];  // {x0 | y0 << 16, w | h << 16}; zeros when culled
      const int w = rc.y & 0xFFFF, h = rc.y >> 16, y0 = rc.x >> 16;
      const int ya = max(y0, br.r0), yb = min(y0 + h, br.r1);
      hit = w > 0 && yb > ya;
      e = make_int4(g, 0, (rc.x & 0xFFFF) | (ya << 16), w | ((yb - ya) << 16));
    }
    if (hit) e.y = (int)depth_keys[g];
    const uint64_t bal = __ballot(hit);
    if (hit) q[qn + __popcll(bal & lt_mask)] = e;
    qn += __popcll(bal);
    __builtin_amdgcn_wave_barrier();
    if (qn >= 64) {
      drain(64);
      pop64();
    }
  }
  if (qn > 0) drain(qn);
}

// ---- scan: per tile the exclusive prefix over chunks (in place) and the tile's count ---------------------
// 16 tiles x 16 chunk slices per workgroup: the walk over a tile's column of the table is a chain of
// memory round trips, so it is cut into 16 short ones (4 tiles x 64: 38 us; this: see profiles/).
constexpr int TS_TILES = 16, TS_SLICES = TB_BLOCK / TS_TILES;
__global__ void __launch_bounds__(TB_BLOCK)
tb_scan_kernel(int T, int n_chunks, uint32_t* __restrict__ table, int32_t* __restrict__ tile_offsets) {
  __shared__ uint32_t part[TS_SLICES][TS_TILES];
  const int tl = threadIdx.x % TS_TILES, qd = threadIdx.x / TS_TILES, t = blockIdx.x * TS_TILES + tl;
  const int cq = (n_chunks + TS_SLICES - 1) / TS_SLICES, c0 = min(qd * cq, n_chunks), c1 = min(c0 + cq, n_chunks);
  constexpr int U = 16;
  uint32_t v[U];
  uint32_t s = 0;
  // (slices of up to U chunks stay in registers between the two phases; longer ones are re-read)
  const bool small = c1 - c0 <= U;
  if (t < T) {
    if (small) {
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = c0 + u < c1 ? table[(size_t)(c0 + u) * T + t] : 0u;
#pragma unroll
      for (int u = 0; u < U; ++u) s += v[u];
    } else {
      for (int c = c0; c < c1; ++c) s += table[(size_t)c * T + t];
    }
  }
  part[qd][tl] = s;
  __syncthreads();
  uint32_t run = 0, total = 0;
#pragma unroll
  for (int k = 0; k < TS_SLICES; ++k) {
    const uint32_t p = part[k][tl];
    if (k < qd) run += p;
    total += p;
  }
  if (t < T) {
    if (small) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (c0 + u < c1) table[(size_t)(c0 + u) * T + t] = run;
        run += v[u];
      }
    } else {
      for (int c = c0; c < c1; ++c) {
        const uint32_t x = table[(size_t)c * T + t];
        table[(size_t)c * T + t] = run;
        run += x;
      }
    }
    if (qd == 0) tile_offsets[t + 1] = (int32_t)total;  // the tile's count, for the pass below
  }
}

// ... and the counts become tile_offsets[T + 1]: ONE workgroup, its own launch.  (As the tail of the kernel
// above -- the last workgroup to arrive, found with a ticket -- every one of the 510 workgroups paid a
// device-scope release, i.e. an L2 write-back, before taking its ticket: 65 us for a kernel that moves 16 MB.)
__global__ void __launch_bounds__(TB_BLOCK)
tb_offsets_kernel(int T, int32_t* __restrict__ tile_offsets, int64_t* __restrict__ count_out) {
  __shared__ uint32_t wave_tot[TB_BLOCK / 64];
  // inclusive scan of the counts in place, 4096 at a time: coalesced loads into LDS (16 independent loads per
  // thread -- a chain of single loads here cost 60 us), every thread scans 16 consecutive values, block scan
  __shared__ uint32_t buf[TB_BLOCK * 16];
  const int lane = fg::lane_id(), wave = threadIdx.x >> 6;
  uint32_t carry = 0;
  for (int base = 0; base < T; base += TB_BLOCK * 16) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int i = base + k * TB_BLOCK + threadIdx.x;
      buf[k * TB_BLOCK + threadIdx.x] = i < T ? (uint32_t)tile_offsets[i + 1] : 0u;
    }
    __syncthreads();
    uint32_t v[16], sum = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      v[k] = buf[threadIdx.x * 16 + k];
      sum += v[k];
    }
    const uint32_t incl = wave_incl_scan_u32(sum, lane);
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t run = carry + incl - sum, all = 0;
#pragma unroll
    for (int k = 0; k < TB_BLOCK / 64; ++k) {
      if (k < wave) run += wave_tot[k];
      all += wave_tot[k];
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      run += v[k];
      buf[threadIdx.x * 16 + k] = run;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int i = base + k * TB_BLOCK + threadIdx.x;
      if (i < T) tile_offsets[i + 1] = (int32_t)buf[k * TB_BLOCK + threadIdx.x];
    }
    carry += all;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    tile_offsets[0] = 0;
    if (count_out) {  // the list length straight into the caller's host-visible word (no copy launch)
      __hip_atomic_store(count_out, (int64_t)carry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __threadfence_system();
    }
  }
}

// ---- per-tile sort ------------------------------------------------------------------------------------
// lanes of the wave holding the same digit (one ballot + two v_bitop3 per digit bit, radix_sort.h)
__device__ __forceinline__ uint64_t same_digit_lanes(unsigned d, int nbits, bool in) {
  const uint64_t in_mask = __ballot(in);
  uint32_t lo = (uint32_t)in_mask, hi = (uint32_t)(in_mask >> 32);
  for (int b = 0; b < nbits; ++b) {
    const int sel = (int)(d << (31 - b)) >> 31;
    const uint64_t m = __ballot(sel != 0);
    lo = __builtin_amdgcn_bitop3_b32(lo, (uint32_t)m, (uint32_t)sel, 0x90);
    hi = __builtin_amdgcn_bitop3_b32(hi, (uint32_t)(m >> 32), (uint32_t)sel, 0x90);
  }
  return ((uint64_t)hi << 32) | lo;
}
// slot of this lane's element: the wave's running cursor of its digit (an LDS word holding the next free
// slot; the group's lowest lane advances it by the group size) + the lane's rank inside the group
__device__ __forceinline__ uint32_t stable_slot(unsigned d, int nbits, bool in, uint32_t* cursor, int lane,
                                                uint64_t lt_mask) {
  const uint64_t peers = same_digit_lanes(d, nbits, in);
  const int leader = in ? (int)__builtin_ctzll(peers) : lane;
  uint32_t base = 0;
  if (in && leader == lane) base = atomicAdd(&cursor[d], (uint32_t)__popcll(peers));
  base = (uint32_t)__shfl((int)base, leader);
  return base + (uint32_t)__popcll(peers & lt_mask);
}
// wave_cnt[w][digit] counts -> start slots (digit-major, waves in order inside a digit); thread = digit
__device__ __forceinline__ void digit_starts(uint32_t (*wave_cnt)[256], uint32_t* scan_tmp) {
  const int lane = fg::lane_id(), wave = threadIdx.x >> 6;
  uint32_t c[TB_BLOCK / 64], tot = 0;
#pragma unroll
  for (int w = 0; w < TB_BLOCK / 64; ++w) {
    c[w] = wave_cnt[w][threadIdx.x];
    tot += c[w];
  }
  const uint32_t incl = wave_incl_scan_u32(tot, lane);
  if (lane == 63) scan_tmp[wave] = incl;
  __syncthreads();
  uint32_t run = incl - tot;
#pragma unroll
  for (int w = 0; w < TB_BLOCK / 64; ++w)
    if (w < wave) run += scan_tmp[w];
#pragma unroll
  for (int w = 0; w < TB_BLOCK / 64; ++w) {
    wave_cnt[w][threadIdx.x] = run;
    run += c[w];
  }
  __syncthreads();
}
__device__ __forceinline__ void block_min_max(uint32_t lo, uint32_t hi, uint32_t* red, uint32_t& kmin, uint32_t& kmax) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) {
    lo = min(lo, (uint32_t)__shfl_xor((int)lo, m));
    hi = max(hi, (uint32_t)__shfl_xor((int)hi, m));
  }
  const int wave = threadIdx.x >> 6;
  if (fg::lane_id() == 0) {
    red[wave] = lo;
    red[4 + wave] = hi;
  }
  __syncthreads();
  kmin = min(min(red[0], red[1]), min(red[2], red[3]));
  kmax = max(max(red[4], red[5]), max(red[6], red[7]));
  __syncthreads();
}
// Position of element i inside its RUN -- the entries whose sorted key bits (depth bits minus the tile's
// smallest, shifted right by `low`) are equal; they are adjacent after the passes, in arrival order -- by the
// full (depth bits, id) order: the number of smaller elements of the run.  Runs are exact depth ties when
// every differing bit was sorted (low = 0) and short groups otherwise (see tb_sort_kernel).
template <typename Ptr>
__device__ __forceinline__ int run_position(Ptr cur, int n, int i, uint64_t e, uint32_t kmin, int low) {
  const uint32_t key = ((uint32_t)(e >> 32) - kmin) >> low;
  const bool left = i > 0 && (((uint32_t)(cur[i - 1] >> 32) - kmin) >> low) == key;
  const bool right = i + 1 < n && (((uint32_t)(cur[i + 1] >> 32) - kmin) >> low) == key;
  if (!left && !right) return i;
  int a = i;
  while (a > 0 && (((uint32_t)(cur[a - 1] >> 32) - kmin) >> low) == key) --a;
  int smaller = 0;
  for (int j = a; j < n; ++j) {
    const uint64_t o = cur[j];
    if ((((uint32_t)(o >> 32) - kmin) >> low) != key) break;
    smaller += o < e;
  }
  return a + smaller;
}
// How many key bits the passes sort: all `bits` that differ inside the tile up to 16 (two 8-bit passes); of
// more only the TOP 16 -- the rest is settled by run_position.  A tile of ~600 entries spread over 65536
// values of the sorted bits has runs of one or two entries; the 1M / 1080p scene has 24 differing bits, i.e.
// two passes instead of three.  (Entries crowded into few values -- a wall plus one far outlier -- make long
// runs: slower, never wrong.)
__device__ __forceinline__ int unsorted_low_bits(int bits) { return bits > 16 ? bits - 16 : 0; }

// A tile too long for LDS: the same passes through global memory (src <-> alt, both L2-resident), two
// sweeps per pass (per-wave digit counts; stable slots).  One workgroup; rare (thousands of entries).
__device__ void sort_tile_global(uint64_t* a, uint64_t* b, int n, int32_t* ids_out,
                                 uint32_t (*wave_cnt)[256], uint32_t* scan_tmp, uint32_t* red) {
  const int lane = fg::lane_id(), wave = threadIdx.x >> 6;
  const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  uint32_t lo = 0xFFFFFFFFu, hi = 0u;
  for (int i = threadIdx.x; i < n; i += TB_BLOCK) {
    const uint32_t k = (uint32_t)(a[i] >> 32);
    lo = min(lo, k);
    hi = max(hi, k);
  }
  uint32_t kmin, kmax;
  block_min_max(lo, hi, red, kmin, kmax);
  const uint32_t range = kmax - kmin;
  const int bits = range ? 32 - __builtin_clz(range) : 0, low = unsorted_low_bits(bits), passes = (bits - low + 7) / 8;
  const int span = ((n + 4 * 64 - 1) / (4 * 64)) * 64;  // a wavefront's contiguous share
  const int w0 = min(wave * span, n), w1 = min(w0 + span, n);
  uint64_t* src = a;
  uint64_t* dst = b;
  int first = low;
  for (int p = 0; p < passes; ++p) {
    const int nbits = (bits - first + (passes - p) - 1) / (passes - p);
    const uint32_t mask = (1u << nbits) - 1u;
#pragma unroll
    for (int w = 0; w < TB_BLOCK / 64; ++w) wave_cnt[w][threadIdx.x] = 0;
    __syncthreads();
    for (int i = w0 + lane; i < w1; i += 64)
      atomicAdd(&wave_cnt[wave][(((uint32_t)(src[i] >> 32) - kmin) >> first) & mask], 1u);
    __syncthreads();
    digit_starts(wave_cnt, scan_tmp);
    for (int i0 = w0; i0 < w1; i0 += 64) {
      const int i = i0 + lane;
      const bool in = i < w1;
      const uint64_t e = in ? src[i] : 0ull;
      const unsigned d = (((uint32_t)(e >> 32) - kmin) >> first) & mask;
      const uint32_t slot = stable_slot(d, nbits, in, wave_cnt[wave], lane, lt_mask);
      if (in) dst[slot] = e;
    }
    __threadfence();  // the next pass reads what other wavefronts wrote, through this CU's L1
    __syncthreads();
    uint64_t* t = src;
    src = dst;
    dst = t;
    first += nbits;
  }
  for (int i = threadIdx.x; i < n; i += TB_BLOCK) {
    const uint64_t e = src[i];
    ids_out[run_position(src, n, i, e, kmin, low)] = (int32_t)(uint32_t)e;
  }
}

// One workgroup per tile (XCD = the tile's band).  Up to 2048 entries live in registers (8 per thread)
// and one LDS image: per pass every wavefront counts the digits of its contiguous share, the counts
// become start slots (digit-major, wavefronts in order: stable), and every element is written to its
// slot of the image and read back in order.  Keys are depth bits minus the tile's smallest: the passes
// cover only bits that differ inside the tile, at most the top 16 of them (unsorted_low_bits).
__global__ void __launch_bounds__(TB_BLOCK)
tb_sort_kernel(int tile_w, int tile_h, const int32_t* __restrict__ tile_offsets, uint64_t* __restrict__ pairs,
               uint64_t* __restrict__ pairs_alt, long long capacity, int32_t* __restrict__ flatten_ids,
               int32_t* __restrict__ list_offsets) {
  __shared__ uint64_t img[TB_SORT_MAX];
  __shared__ uint32_t wave_cnt[TB_BLOCK / 64][256];
  __shared__ uint32_t scan_tmp[TB_BLOCK / 64];
  __shared__ uint32_t red[8];
  const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
  const Rows br = band_rows(xcd, tile_h);
  if (k >= (br.r1 - br.r0) * tile_w) return;
  const int tile = br.r0 * tile_w + k, T = tile_w * tile_h;
  const int total = tile_offsets[T];
  const bool over = (long long)total > capacity;
  const int off = tile_offsets[tile], n = tile_offsets[tile + 1] - off;
  // the ranges the CONSUMERS of flatten_ids read: the tile ranges when the list fits, empty lists when it
  // does not (nothing was filled; whoever was enqueued speculatively behind this call walks nothing)
  if (threadIdx.x == 0) {
    list_offsets[tile] = over ? 0 : off;
    if (tile == T - 1) list_offsets[T] = over ? 0 : total;
  }
  if (over || n <= 0) return;
  if (n > TB_SORT_MAX) {
    sort_tile_global(pairs + off, pairs_alt + off, n, flatten_ids + off, wave_cnt, scan_tmp, red);
    return;
  }
  const int lane = fg::lane_id(), wave = threadIdx.x >> 6;
  const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const int R = (n + TB_BLOCK - 1) / TB_BLOCK;  // rounds; wavefront w owns [w R 64, (w + 1) R 64)
  const int ibase = wave * R * 64 + lane;
  uint64_t e[TB_SORT_KPT];
  uint32_t lo = 0xFFFFFFFFu, hi = 0u;
#pragma unroll
  for (int q = 0; q < TB_SORT_KPT; ++q) {
    e[q] = 0;
    if (q < R) {
      const int i = ibase + q * 64;
      if (i < n) {
        e[q] = pairs[off + i];
        lo = min(lo, (uint32_t)(e[q] >> 32));
        hi = max(hi, (uint32_t)(e[q] >> 32));
      }
    }
  }
  uint32_t kmin, kmax;
  block_min_max(lo, hi, red, kmin, kmax);
  const uint32_t range = kmax - kmin;
  const int bits = range ? 32 - __builtin_clz(range) : 0, low = unsorted_low_bits(bits), passes = (bits - low + 7) / 8;
  int first = low;
  for (int p = 0; p < passes; ++p) {
    const int nbits = (bits - first + (passes - p) - 1) / (passes - p);
    const uint32_t mask = (1u << nbits) - 1u;
#pragma unroll
    for (int w = 0; w < TB_BLOCK / 64; ++w) wave_cnt[w][threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < TB_SORT_KPT; ++q)
      if (q < R && ibase + q * 64 < n) atomicAdd(&wave_cnt[wave][(((uint32_t)(e[q] >> 32) - kmin) >> first) & mask], 1u);
    __syncthreads();
    digit_starts(wave_cnt, scan_tmp);
    // stable slots: the group's lowest lane advances the wave's cursor of the digit by the group size; all
    // the atomics of a lane are issued before any result is consumed (same-address LDS atomics of one
    // wavefront retire in issue order): one LDS round trip on the chain instead of R
    uint32_t rk[TB_SORT_KPT];
#pragma unroll
    for (int q = 0; q < TB_SORT_KPT; ++q) {
      rk[q] = 0;
      if (q < R) {
        const bool in = ibase + q * 64 < n;
        const unsigned d = (((uint32_t)(e[q] >> 32) - kmin) >> first) & mask;
        const uint64_t peers = same_digit_lanes(d, nbits, in);
        const uint32_t leader = in ? (uint32_t)__builtin_ctzll(peers) : (uint32_t)lane;
        uint32_t r = (uint32_t)__popcll(peers & lt_mask);
        if (in && leader == (uint32_t)lane) r = atomicAdd(&wave_cnt[wave][d], (uint32_t)__popcll(peers));
        rk[q] = r | (leader << 16);
      }
    }
#pragma unroll
    for (int q = 0; q < TB_SORT_KPT; ++q) {
      if (q < R) {
        const uint32_t leader = rk[q] >> 16;
        const uint32_t before = (uint32_t)__shfl((int)(rk[q] & 0xFFFFu), (int)leader);
        const uint32_t slot = (leader == (uint32_t)lane) ? before : before + (rk[q] & 0xFFFFu);
        if (ibase + q * 64 < n) img[slot] = e[q];
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < TB_SORT_KPT; ++q)
      if (q < R && ibase + q * 64 < n) e[q] = img[ibase + q * 64];
    first += nbits;
  }
  if (passes == 0) {  // every entry has the same depth bits: the image is the input, ties decide
#pragma unroll
    for (int q = 0; q < TB_SORT_KPT; ++q)
      if (q < R && ibase + q * 64 < n) img[ibase + q * 64] = e[q];
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < TB_SORT_KPT; ++q) {
    if (q < R) {
      const int i = ibase + q * 64;
      if (i < n) flatten_ids[off + run_position(img, n, i, e[q], kmin, low)] = (int32_t)(uint32_t)e[q];
    }
  }
}

size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
int n_chunks_of(int N) { return (N + TB_CHUNK - 1) / TB_CHUNK; }

}  // namespace

extern "C" int fg_tilebin_supported(int tile_w, int tile_h) {
  if (tile_w <= 0 || tile_h <= 0 || tile_w > 1023 || tile_h > 1023) return 0;
  return band_tiles_max(tile_w, tile_h) <= TB_MAX_BAND_TILES ? 1 : 0;
}

extern "C" size_t fg_tilebin_count_workspace_bytes(int N, int tile_w, int tile_h) {
  if (N < 0 || tile_w <= 0 || tile_h <= 0) return 0;
  return al256((size_t)n_chunks_of(N > 0 ? N : 1) * (size_t)tile_w * tile_h * 4) + 256;
}

extern "C" int fg_tilebin_count(int N, const int32_t* tile_rects, int tile_w, int tile_h, int32_t* tile_offsets,
                                int64_t* count_out, void* workspace, size_t workspace_bytes, fg_stream_t stream) {
  if (N <= 0 || tile_w <= 0 || tile_h <= 0 || !tile_rects || !tile_offsets || !workspace) return FG_ERR_INVALID_ARG;
  if (!fg_tilebin_supported(tile_w, tile_h)) return FG_ERR_UNSUPPORTED;
  if (workspace_bytes < fg_tilebin_count_workspace_bytes(N, tile_w, tile_h)) return FG_ERR_WORKSPACE;
  hipStream_t s = fg_hip_stream(stream);
  const int T = tile_w * tile_h, nc = n_chunks_of(N);
  uint32_t* table = static_cast<uint32_t*>(workspace);
  const size_t lds = (size_t)((tile_h + 7) / 8 + 1) * (tile_w + 1) * 4;
  hipLaunchKernelGGL(tb_count_kernel, dim3(8 * nc), dim3(TB_BLOCK), lds, s, N, reinterpret_cast<const int2*>(tile_rects),
                     tile_w, tile_h, table);
  hipLaunchKernelGGL(tb_scan_kernel, dim3((T + TS_TILES - 1) / TS_TILES), dim3(TB_BLOCK), 0, s, T, nc, table,
                     tile_offsets);
  hipLaunchKernelGGL(tb_offsets_kernel, dim3(1), dim3(TB_BLOCK), 0, s, T, tile_offsets, count_out);
  FG_RETURN_IF_LAUNCH_FAILED();
  return FG_OK;
}

extern "C" size_t fg_tilebin_fill_workspace_bytes(int64_t capacity) {
  return 2 * al256((size_t)(capacity > 0 ? capacity : 1) * 8);
}

extern "C" int fg_tilebin_fill(int N, const uint32_t* depth_keys, const int32_t* tile_rects, int tile_w, int tile_h,
                               int64_t capacity, const int32_t* tile_offsets, const void* count_workspace,
                               int32_t* flatten_ids, int32_t* list_offsets, void* workspace, size_t workspace_bytes,
                               fg_stream_t stream) {
  if (N <= 0 || capacity <= 0 || tile_w <= 0 || tile_h <= 0) return FG_ERR_INVALID_ARG;
  if (!depth_keys || !tile_rects || !tile_offsets || !count_workspace || !flatten_ids || !list_offsets || !workspace)
    return FG_ERR_INVALID_ARG;
  if (!fg_tilebin_supported(tile_w, tile_h)) return FG_ERR_UNSUPPORTED;
  if (workspace_bytes < fg_tilebin_fill_workspace_bytes(capacity)) return FG_ERR_WORKSPACE;
  hipStream_t s = fg_hip_stream(stream);
  const int nc = n_chunks_of(N);
  uint64_t* pairs = static_cast<uint64_t*>(workspace);
  uint64_t* pairs_alt = reinterpret_cast<uint64_t*>(static_cast<char*>(workspace) + al256((size_t)capacity * 8));
  uint32_t* table = static_cast<uint32_t*>(const_cast<void*>(count_workspace));
  const size_t lds = (size_t)band_tiles_max(tile_w, tile_h) * 4;
  hipLaunchKernelGGL(tb_scatter_kernel, dim3(8 * nc), dim3(TB_BLOCK), lds, s, N,
                     reinterpret_cast<const int2*>(tile_rects), depth_keys, tile_w, tile_h, table, tile_offsets, pairs,
                     (long long)capacity);
  hipLaunchKernelGGL(tb_sort_kernel, dim3(8 * band_tiles_max(tile_w, tile_h)), dim3(TB_BLOCK), 0, s, tile_w, tile_h,
                     tile_offsets, pairs, pairs_alt, (long long)capacity, flatten_ids, list_offsets);
  FG_RETURN_IF_LAUNCH_FAILED();
  return FG_OK;
}
