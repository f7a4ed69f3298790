// K3 + K4, supertile form: count -> scan -> scatter -> sort per segment, with a SUPERTILE of 2 x 2
// tiles (32 x 32 pixels).  A Gaussian is scattered once per supertile its rectangle touches (2.4 on the
// 1M / 1080p scene) instead of once per tile (6.0), every supertile's entries are sorted ONCE by (depth
// bits, id), and the four tile lists are read off the sorted run in order -- an entry goes to the tiles
// its rectangle covers.  Same lists as the reference algorithm (tile | depth keys, stable sort: behind
// `tile_size=16` of /root/reference freegaussian/freegaussian_model.py:847-868), bit for bit: inside a
// tile the order is the total order (depth bits, id) whatever route the entries took.
//
//   count   one workgroup per chunk of 4096 Gaussians: corner marks of the chunk's rectangles on the tile grid and
//           the supertile grid (LDS, the whole image when it fits -- 1080p does --, else in bands of rows with the
//           rectangles kept in registers), two prefix sums each -> table_t[chunk][T], table_s[chunk][S]
//   columns per tile the sum over chunks; per supertile the exclusive prefix over chunks and the sum
//   offsets one workgroup per array: tile_offsets[T + 1], st_offsets[S + 1], the list length to the host
//   scatter one workgroup per chunk: one 8-byte element per (Gaussian, supertile) pair to the slot an LDS cursor
//           of the supertile hands out
//   sort    one workgroup per supertile: sort, then emit the four tile lists
//
// Every rectangle is read ONCE by count and once by scatter.  (Round 3 began with workgroups per (XCD band, chunk)
// so that a segment's scattered stores met in one L2: count and scatter then each pulled the rectangles through
// all eight L2s -- 62 + 95 MB of fabric reads for 12 MB of data, and that, not the stores, bounded both kernels.)
// The sort deals supertiles to the XCDs round-robin.
#include <atomic>

#include "fg_common.h"
#include "jobs_build.h"

namespace {

constexpr int SB_BLOCK = 256;                        // columns kernel
constexpr int SB_CHUNK = 4096;                       // Gaussians per workgroup of count and scatter AT MOST ...
constexpr int SC_BLOCK = 1024, SC_WAVES = SC_BLOCK / 64, SC_PER = SB_CHUNK / SC_BLOCK;
// ... `per` x 1024 of them (per = 1, 2, 4 by N, chunk_rounds): a trained scene of 257k Gaussians is 63 chunks of 4096 -- a
// quarter of the chip's CUs busy, count 20 us and scatter 35 us against 21 / 29 for four times as many Gaussians --, and
// 251 of 1024.  The [chunks][T] tables keep their size: ~250-320 chunks at every N up to 1.3M.
__host__ __device__ __forceinline__ int chunk_rounds(int N) { return N <= 320 * 1024 ? 1 : (N <= 640 * 1024 ? 2 : SC_PER); }
#ifndef FG_SB_SMALL_WAVES
#define FG_SB_SMALL_WAVES 8
#endif
#ifndef FG_SB_SMALL_KPT
#define FG_SB_SMALL_KPT 6
#endif
constexpr int SB_SMALL_WAVES = FG_SB_SMALL_WAVES, SB_LARGE_WAVES = 16;  // wavefronts per workgroup of the two sort launches
// (a supertile holds 1250 entries on average, up to ~2300, on the 1M / 1080p scene; 64 x wavefronts x elements per thread
// are sorted in LDS -- 3072 / 8064 for the two launches --, longer segments through global memory)
constexpr int SB_SMALL_KPT = FG_SB_SMALL_KPT, SB_LARGE_KPT = 8;  // elements per thread
constexpr int SB_LARGE_GRID = 512;                   // persistent workgroups of the large launch (two per CU)
constexpr int SB_SMALL_BUCKET_BITS = 11, SB_LARGE_BUCKET_BITS = 12;  // the counting pass of the LDS sorts
constexpr int SB_MAX_LDS_WORDS = 12288;              // grid words of the count kernel / cursors of the scatter
constexpr int SB_SCATTER_LDS_BYTES = 124 * 1024;     // dynamic LDS of the staging scatter (+ 29 KB static: one workgroup per CU)
constexpr int SB_MIN_STAGE = 4096;                   // staging buffers smaller than this are not worth the second sweep
#ifndef FG_SB_SKEW_MAX
#define FG_SB_SKEW_MAX 96
#endif
constexpr int SB_SKEW_MAX = FG_SB_SKEW_MAX;                      // elements in one bucket of an LDS sort's counting pass beyond which the segment is re-bucketed by splitters
// LONG segments (more elements than the large launch sorts in LDS: a dense cluster over one supertile) are split by
// SAMPLE SORT into buckets of ~LG_T elements that the LDS sort then takes one by one (FG_STBIN_LONG_SEGMENTS):
constexpr int SB_LONG_MIN = 64 * 16 * 8 - 256;       // = SortShared<SB_LARGE_WAVES, SB_LARGE_KPT, ..>::MAXN (static_assert below)
// ... and (long mode) every segment of more than SB_LONG_SPLIT elements -- the small launch's LDS capacity -- goes that way:
// the large launch sorts one segment per 1024-thread workgroup, two workgroups per CU; where hundreds of segments are of
// that size the bucket passes -- more, smaller workgroups -- are faster (fill 0.207 -> 0.137 ms with half of the Gaussians
// in a ball of 0.4, 0.224 -> 0.173 with 80 % in a ball of 0.2; profiles/r04_binning.md).  The large launch's own sorts
// remain what runs without the flag.
#ifndef FG_SB_LONG_SPLIT
#define FG_SB_LONG_SPLIT 3072
#endif
constexpr int SB_LONG_SPLIT = FG_SB_LONG_SPLIT;
static_assert(SB_LONG_SPLIT <= SB_LONG_MIN && SB_LONG_SPLIT >= 2 * 1536, "long segments: at least two buckets, at most the large sort's capacity");
constexpr int LG_T = 1536;                           // target bucket size: half of the small LDS sort's capacity (sb_long_sort_kernel)
constexpr int LG_KMAX = 1008;                        // buckets per segment at most (splitters + counters in LDS)
constexpr int LG_SA = 32;                            // samples per bucket (fewer when LG_SA * k exceeds one LDS sort): a bucket twice its target is a 1e-6 event
constexpr int LG_BLOCK = 512, LG_PER = 8, LG_CHUNK = LG_BLOCK * LG_PER;  // count / scatter passes over a long segment
constexpr int LG_GRID = 1024, LG_SORT_GRID = 1024;   // persistent workgroups of those passes / of the bucket sort
constexpr int SB_LONG_OVER_GRID = 64;                // workgroups of sb_long_overflow_kernel
constexpr int SB_LONG_SLAB = SB_LONG_SPLIT;          // elements of `scratch` a bucket owns: what the LDS sort of sb_long_sort_kernel takes (2 x LG_T)
__host__ __device__ __forceinline__ int long_buckets(int n) {
  const int k = (n + LG_T - 1) / LG_T;
  return k > LG_KMAX ? LG_KMAX : (k < 2 ? 2 : k);
}
// (cap: what the sampler's LDS sort takes; a long segment has more elements than that)
__host__ __device__ __forceinline__ int long_samples(int k, int cap) {
  const int s = LG_SA * k;
  return s > cap ? cap : s;
}
// per bucket of every long segment (fill workspace, behind the two element arrays)
struct LongTables {
  uint64_t* split;     // [b] = smallest element of bucket b + 1 (bucket k - 1: unused)
  uint32_t* cnt;       // elements per bucket
  uint32_t* cursor;    // scatter cursors
  uint4* tcnt;         // elements per bucket that go to tile j of the supertile
  // what a work item needs of its segment, one load (build_segment_lists): {first bucket slot, first element, elements,
  // chunk in the segment} per chunk of the count / scatter passes, {first bucket slot, first element, supertile, bucket in the
  // segment} per bucket slot of the sort
  int4* chunk_seg;
  int4* bucket_seg;
  // [0] = skewed whole segments the small sort launch left, [1] = buckets beyond the LDS sort's capacity; the two lists
  // follow: -(supertile + 1) at [2 ..], bucket slots at [2 + over_cap ..]
  int32_t* over_list;
  int over_cap;
  uint64_t* arena;     // sb_long_overflow_kernel: SB_LONG_OVER_GRID x 7936 elements, a stretch per workgroup
};
__device__ __forceinline__ int n_slots_cap(const LongTables& lt) { return lt.over_cap; }

struct Geo {
  int tile_w, tile_h, sw, sh;
};
__host__ __device__ __forceinline__ Geo geo_of(int tile_w, int tile_h) {
  Geo g;
  g.tile_w = tile_w;
  g.tile_h = tile_h;
  g.sw = (tile_w + 1) >> 1;
  g.sh = (tile_h + 1) >> 1;
  return g;
}
// Supertile rows per LDS pass: count holds (2 rows + 1) x (tile_w + 1) tile marks and (rows + 1) x (sw + 1)
// supertile marks, scatter rows x sw cursors.  0 = does not fit at all.
__host__ __device__ __forceinline__ int count_band_rows(const Geo& g) {
  const int per = 2 * (g.tile_w + 1) + (g.sw + 1), fixed = (g.tile_w + 1) + (g.sw + 1);
  const int rows = (SB_MAX_LDS_WORDS - fixed) / per;
  return rows < g.sh ? (rows > 0 ? rows : 0) : g.sh;
}
__host__ __device__ __forceinline__ int scatter_band_rows(const Geo& g) {
  const int rows = SB_MAX_LDS_WORDS / g.sw;
  return rows < g.sh ? rows : g.sh;
}

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(v, d);
    if (lane >= d) v += o;
  }
  return v;
}

// corner marks -> counts, both grids in the same two phases: prefix along x (a wavefront per row, 64 cells a
// step), then along y with the result written out (thread = column); grids [rows + 1][cols + 1]
// (rowwise: the marks are differences along x only -- a pair per run of blocks a footprint mask sets in a row --, so the
// second phase copies the rows out instead of summing down the columns)
__device__ __forceinline__ void marks_to_counts(int32_t* gt, int ntr, int tile_w, uint32_t* __restrict__ out_t,
                                                int32_t* gs, int nsr, int sw, uint32_t* __restrict__ out_s,
                                                bool rowwise = false) {
  const int lane = fg::lane_id(), wave = threadIdx.x >> 6;
  for (int r = wave; r < ntr + nsr; r += SC_WAVES) {
    const bool tiles = r < ntr;
    int32_t* row = tiles ? gt + r * (tile_w + 1) : gs + (r - ntr) * (sw + 1);
    const int cols = tiles ? tile_w : sw;
    int carry = 0;
    for (int x = 0; x < cols; x += 64) {
      const int v = x + lane < cols ? row[x + lane] : 0;
      const int incl = (int)wave_incl_scan((uint32_t)v, lane) + carry;
      if (x + lane < cols) row[x + lane] = incl;
      carry = __builtin_amdgcn_readlane(incl, 63);
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < tile_w + sw; c += SC_BLOCK) {
    const bool tiles = c < tile_w;
    const int x = tiles ? c : c - tile_w, cols = tiles ? tile_w : sw, rows = tiles ? ntr : nsr;
    const int32_t* col = (tiles ? gt : gs) + x;
    uint32_t* out = (tiles ? out_t : out_s) + x;
    int run = 0, row = 0;
    if (rowwise) {
      for (; row + 4 <= rows; row += 4) {
        int v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = col[(row + k) * (cols + 1)];
#pragma unroll
        for (int k = 0; k < 4; ++k) out[(size_t)(row + k) * cols] = (uint32_t)v[k];
      }
      for (; row < rows; ++row) out[(size_t)row * cols] = (uint32_t)col[row * (cols + 1)];
      continue;
    }
    for (; row + 4 <= rows; row += 4) {
      int v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = col[(row + k) * (cols + 1)];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        run += v[k];
        out[(size_t)(row + k) * cols] = (uint32_t)run;
      }
    }
    for (; row < rows; ++row) {
      run += col[row * (cols + 1)];
      out[(size_t)row * cols] = (uint32_t)run;
    }
  }
}

// ---- count --------------------------------------------------------------------------------------------
// runs of set bits among the low 8 bits of `bits`: f(first, one past the last)
template <typename F>
__device__ __forceinline__ void for_bit_runs(uint32_t bits, F f) {
  bits &= 0xFFu;
  if (bits == 0u) return;
  {  // one run -- what a convex footprint gives in every row -- without the loop
    const int s = __builtin_ctz(bits), e = 32 - __builtin_clz(bits);
    if (bits == (((1u << (e - s)) - 1u) << s)) {
      f(s, e);
      return;
    }
  }
  while (bits) {
    const int s = __builtin_ctz(bits);
    const int len = __builtin_ctz(~(bits >> s));
    f(s, s + len);
    bits &= ~(((1u << len) - 1u) << s);
  }
}
// masks (nullable): fg::footprint_mask of every rectangle -- which blocks of it the ellipse reaches; counted (and
// scattered, sb_scatter_kernel) are the set blocks' tiles only.  NULL: whole rectangles.
__global__ void __launch_bounds__(SC_BLOCK)
sb_count_kernel(int N, int per, const int2* __restrict__ rects, const unsigned long long* __restrict__ masks, int tile_w,
                int tile_h, int band_rows, uint32_t* __restrict__ table_t, uint32_t* __restrict__ table_s,
                unsigned long long* __restrict__ area_part) {
  extern __shared__ int32_t s_grid[];  // tile grid [(tile rows + 1)][tile_w + 1], then supertile grid
  // (masks: the wavefront's round of rectangles compacted to its owners of supertile rows, as in sb_scatter_kernel)
  __shared__ int4 s_cq[SC_WAVES][64];          // {rect.x, rect.y, footprint mask low, high}
  __shared__ uint32_t s_cqx[SC_WAVES][64];     // items before this owner << 10 | its first supertile row
  __shared__ unsigned long long s_cmarks[SC_WAVES];
  const Geo g = geo_of(tile_w, tile_h);
  const int chunk = blockIdx.x;
  const int gwt = tile_w + 1, gws = g.sw + 1;
  const int T = tile_w * tile_h, S = g.sw * g.sh;
  const int lane = fg::lane_id(), wave = threadIdx.x >> 6;
  // masks: a wavefront owns 256 consecutive Gaussians (64 a round); whole rectangles: thread-strided over the chunk
  const int chunk_n = per * SC_BLOCK;
  const int g0 = masks ? chunk * chunk_n + wave * (chunk_n / SC_WAVES) + lane : chunk * chunk_n + (int)threadIdx.x;
  const int gstep = masks ? 64 : SC_BLOCK;
  int2 rc[SC_PER];
  unsigned long long mk[SC_PER];
#pragma unroll
  for (int r = 0; r < SC_PER; ++r) {
    const bool in = r < per && g0 + r * gstep < N;
    rc[r] = in ? rects[g0 + r * gstep] : make_int2(0, 0);
    mk[r] = in && masks ? masks[g0 + r * gstep] : 0ull;
  }
  const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  {
    // The chunk's RECTANGLE AREA (tiles), beside the counts: with footprint masks the list length falls short of it by what
    // the masks drop -- the host keeps the masks for a shape only while that is worth their price (count_out[14])
    __shared__ unsigned long long s_area[SC_WAVES];
    unsigned long long area = 0;
#pragma unroll
    for (int r = 0; r < SC_PER; ++r) area += (unsigned long long)((rc[r].y & 0xFFFF) * (rc[r].y >> 16));
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) area += __shfl_xor(area, m);
    if (lane == 0) s_area[wave] = area;
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long all = 0;
#pragma unroll
      for (int w = 0; w < SC_WAVES; ++w) all += s_area[w];
      area_part[chunk] = all;
    }
  }
  for (int sr0 = 0; sr0 < g.sh; sr0 += band_rows) {  // (one pass when the image's grids fit the LDS)
    const int sr1 = min(sr0 + band_rows, g.sh), tr0 = 2 * sr0, tr1 = min(2 * sr1, tile_h);
    const int ntr = tr1 - tr0, nsr = sr1 - sr0;
    int32_t* gt = s_grid;
    int32_t* gs = s_grid + (ntr + 1) * gwt;
    for (int i = threadIdx.x; i < (ntr + 1) * gwt + (nsr + 1) * gws; i += SC_BLOCK) s_grid[i] = 0;
    __syncthreads();
    if (masks) {
      // One ITEM = one supertile row of one rectangle: each of the row's two tile rows marks its runs of set blocks on the
      // tile grid, the two together theirs on the supertile grid -- differences along x ONLY (marks_to_counts rowwise), two
      // LDS atomics a run.  The items of a round's 64 rectangles are DEALT to the lanes (round 5 looped over the rows per
      // lane: a wall splat next to the lens has 34 supertile rows at 1080p and the wavefront's other 63 lanes waited for
      // it -- 25.7 us for 257k Gaussians of a trained scene against 21 us for the bench scene's million).
      int4* q = s_cq[wave];
      uint32_t* qx = s_cqx[wave];
      unsigned long long* marks = &s_cmarks[wave];
#pragma unroll
      for (int r = 0; r < SC_PER; ++r) {
        if (r >= per) break;
        const int w = rc[r].y & 0xFFFF, h = rc[r].y >> 16, y0 = rc[r].x >> 16;
        const int Ra = max(y0 >> 1, sr0), Rb = min(((y0 + h - 1) >> 1) + 1, sr1);
        const bool hit = w > 0 && h > 0 && Rb > Ra;
        const uint32_t cnt = hit ? (uint32_t)(Rb - Ra) : 0u;
        const uint32_t incl = wave_incl_scan(cnt, lane);
        const int total = (int)__builtin_amdgcn_readlane((int)incl, 63);
        const uint64_t hm = __ballot(hit);
        const int ci = __popcll(hm & lt_mask);
        if (hit) {
          q[ci] = make_int4(rc[r].x, rc[r].y, (int)(uint32_t)mk[r], (int)(uint32_t)(mk[r] >> 32));
          qx[ci] = ((incl - cnt) << 10) | (uint32_t)Ra;  // (rows < 1024: fg_stbin_supported; items of a round <= 64 * 512)
        }
        __builtin_amdgcn_wave_barrier();
        for (int s0 = 0; s0 < total; s0 += 64) {
          // owner of slot s0 + lane: the owners that ended before the window (a ballot) + the end marks below this lane's
          // bit (sb_scatter_kernel's scheme)
          if (lane == 0) *marks = 0ull;
          __builtin_amdgcn_wave_barrier();
          const int endpos = (int)incl - 1 - s0;
          if (hit && endpos >= 0 && endpos < 64) atomicOr(marks, 1ull << endpos);
          __builtin_amdgcn_wave_barrier();
          const uint64_t em = __hip_atomic_load(marks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
          const int before = __popcll(__ballot(hit && (int)incl <= s0));
          const int slot = s0 + lane;
          if (slot < total) {
            const int lo = before + __popcll(em & lt_mask);
            const int4 o = q[lo];
            const uint32_t ox = qx[lo];
            const unsigned long long m = ((unsigned long long)(uint32_t)o.w << 32) | (uint32_t)o.z;
            const int ow = o.y & 0xFFFF, oh = o.y >> 16, ox0 = o.x & 0xFFFF, oy0 = o.x >> 16;
            const int R = (int)(ox & 1023u) + (slot - (int)(ox >> 10));
            const int sh = 31 - __builtin_clz((unsigned)fg::footprint_block(ow, oh)), bs = 1 << sh;
            uint32_t both = 0;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              const int ty = 2 * R + i;
              if (ty < oy0 || ty >= oy0 + oh) continue;
              const uint32_t bits = (uint32_t)(m >> (8 * ((ty - oy0) >> sh))) & 0xFFu;
              both |= bits;
              for_bit_runs(bits, [&](int a0, int a1) {
                atomicAdd(&gt[(ty - tr0) * gwt + ox0 + a0 * bs], 1);
                atomicAdd(&gt[(ty - tr0) * gwt + min(ox0 + a1 * bs, ox0 + ow)], -1);
              });
            }
            for_bit_runs(both, [&](int a0, int a1) {
              atomicAdd(&gs[(R - sr0) * gws + ((ox0 + a0 * bs) >> 1)], 1);
              atomicAdd(&gs[(R - sr0) * gws + ((min(ox0 + a1 * bs, ox0 + ow) - 1) >> 1) + 1], -1);
            });
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    } else {
#pragma unroll
      for (int r = 0; r < SC_PER; ++r) {
        const int w = rc[r].y & 0xFFFF, h = rc[r].y >> 16, x0 = rc[r].x & 0xFFFF, y0 = rc[r].x >> 16;
        const int ya = max(y0, tr0), yb = min(y0 + h, tr1);
        if (w > 0 && yb > ya) {
          const int ra = ya - tr0, rb = yb - tr0;
          atomicAdd(&gt[ra * gwt + x0], 1);
          atomicAdd(&gt[ra * gwt + x0 + w], -1);
          atomicAdd(&gt[rb * gwt + x0], -1);
          atomicAdd(&gt[rb * gwt + x0 + w], 1);
          const int sxa = x0 >> 1, sxb = ((x0 + w - 1) >> 1) + 1;
          const int sya = (ya >> 1) - sr0, syb = ((yb - 1) >> 1) + 1 - sr0;
          atomicAdd(&gs[sya * gws + sxa], 1);
          atomicAdd(&gs[sya * gws + sxb], -1);
          atomicAdd(&gs[syb * gws + sxa], -1);
          atomicAdd(&gs[syb * gws + sxb], 1);
        }
      }
    }
    __syncthreads();
    marks_to_counts(gt, ntr, tile_w, table_t + (size_t)chunk * T + tr0 * tile_w, gs, nsr, g.sw,
                    table_s + (size_t)chunk * S + sr0 * g.sw, masks != nullptr);
    __syncthreads();
  }
}

// ---- columns: 16 columns x 16 chunk slices per workgroup.  Workgroups [0, wg_t) sum the tile columns,
// the rest turn the supertile columns into exclusive prefixes over chunks (in place) + the sum -----------
constexpr int SC_COLS = 16, SC_SLICES = SB_BLOCK / SC_COLS;
__global__ void __launch_bounds__(SB_BLOCK)
sb_columns_kernel(int T, int S, int n_chunks, int wg_t, const uint32_t* __restrict__ table_t,
                  uint32_t* __restrict__ table_s, int32_t* __restrict__ tile_offsets, int32_t* __restrict__ st_offsets) {
  __shared__ uint32_t part[SC_SLICES][SC_COLS];
  const bool tiles = (int)blockIdx.x < wg_t;
  const int n = tiles ? T : S;
  const int cl = threadIdx.x % SC_COLS, qd = threadIdx.x / SC_COLS;
  const int col = ((int)blockIdx.x - (tiles ? 0 : wg_t)) * SC_COLS + cl;
  const int cq = (n_chunks + SC_SLICES - 1) / SC_SLICES, c0 = min(qd * cq, n_chunks), c1 = min(c0 + cq, n_chunks);
  const uint32_t* src = tiles ? table_t : table_s;
  constexpr int U = 16;
  uint32_t v[U], s = 0;
  const bool small = c1 - c0 <= U;
  if (col < n) {
    if (small) {
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = c0 + u < c1 ? src[(size_t)(c0 + u) * n + col] : 0u;
#pragma unroll
      for (int u = 0; u < U; ++u) s += v[u];
    } else {
      for (int c = c0; c < c1; ++c) s += src[(size_t)c * n + col];
    }
  }
  part[qd][cl] = s;
  __syncthreads();
  uint32_t run = 0, total = 0;
#pragma unroll
  for (int k = 0; k < SC_SLICES; ++k) {
    const uint32_t p = part[k][cl];
    if (k < qd) run += p;
    total += p;
  }
  if (col >= n) return;
  if (tiles) {
    if (qd == 0) tile_offsets[col + 1] = (int32_t)total;  // counts; sb_offsets_kernel scans them
    return;
  }
  if (small) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (c0 + u < c1) table_s[(size_t)(c0 + u) * n + col] = run;
      run += v[u];
    }
  } else {
    for (int c = c0; c < c1; ++c) {
      const uint32_t x = table_s[(size_t)c * n + col];
      table_s[(size_t)c * n + col] = run;
      run += x;
    }
  }
  if (qd == 0) st_offsets[col + 1] = (int32_t)total;
}

// ---- offsets: inclusive scans of the counts at [1 .. n] in place, [0] = 0; one workgroup per array.  1024
// threads x 8 values: the 8160 tiles of a 1080p frame are one trip through memory, not two ----------------------
constexpr int SO_BLOCK = 1024, SO_WAVES = SO_BLOCK / 64, SO_PER = 8;
__device__ __forceinline__ uint64_t scan_counts_in_place(int n, int32_t* __restrict__ offs, uint32_t* buf,
                                                         uint32_t* wave_tot, uint32_t* largest = nullptr,
                                                         uint32_t over_thr = 0xFFFFFFFFu, uint32_t* n_over = nullptr) {
  const int lane = fg::lane_id(), wave = threadIdx.x >> 6;
  uint32_t carry = 0, big = 0, over = 0;
  uint64_t total = 0;  // (the offsets are 32-bit; the total is reported in full so that the host can refuse a list it cannot index)
  for (int base = 0; base < n; base += SO_BLOCK * SO_PER) {
    uint32_t in[SO_PER];
#pragma unroll
    for (int k = 0; k < SO_PER; ++k) {
      const int i = base + k * SO_BLOCK + threadIdx.x;
      in[k] = i < n ? (uint32_t)offs[i + 1] : 0u;
    }
#pragma unroll
    for (int k = 0; k < SO_PER; ++k) {
      buf[k * SO_BLOCK + threadIdx.x] = in[k];
      big = max(big, in[k]);
      over += in[k] > over_thr ? 1u : 0u;
    }
    __syncthreads();
    uint32_t v[SO_PER], sum = 0;
#pragma unroll
    for (int k = 0; k < SO_PER; k += 4) {
      const uint4 x = *reinterpret_cast<const uint4*>(&buf[threadIdx.x * SO_PER + k]);
      v[k] = x.x; v[k + 1] = x.y; v[k + 2] = x.z; v[k + 3] = x.w;
    }
#pragma unroll
    for (int k = 0; k < SO_PER; ++k) sum += v[k];
    const uint32_t incl = wave_incl_scan(sum, lane);
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t run = carry + incl - sum, all = 0;
#pragma unroll
    for (int k = 0; k < SO_WAVES; ++k) {
      const uint32_t t = wave_tot[k];
      if (k < wave) run += t;
      all += t;
    }
#pragma unroll
    for (int k = 0; k < SO_PER; ++k) {
      run += v[k];
      v[k] = run;
    }
#pragma unroll
    for (int k = 0; k < SO_PER; k += 4)
      *reinterpret_cast<uint4*>(&buf[threadIdx.x * SO_PER + k]) = make_uint4(v[k], v[k + 1], v[k + 2], v[k + 3]);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SO_PER; ++k) {
      const int i = base + k * SO_BLOCK + threadIdx.x;
      if (i < n) offs[i + 1] = (int32_t)buf[k * SO_BLOCK + threadIdx.x];
    }
    carry += all;
    total += all;
    __syncthreads();
  }
  if (threadIdx.x == 0) offs[0] = 0;
  if (largest) {  // the largest count, for the caller's one thread that wants it (thread 0)
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) big = max(big, (uint32_t)__shfl_xor((int)big, m));
    __syncthreads();
    if (lane == 0) wave_tot[wave] = big;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SO_WAVES; ++k) big = max(big, wave_tot[k]);
    *largest = big;
  }
  if (n_over) {  // how many counts exceed over_thr (for thread 0)
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) over += (uint32_t)__shfl_xor((int)over, m);
    __syncthreads();
    if (lane == 0) wave_tot[wave] = over;
    __syncthreads();
    over = 0;
#pragma unroll
    for (int k = 0; k < SO_WAVES; ++k) over += wave_tot[k];
    *n_over = over;
  }
  return total;
}
__global__ void __launch_bounds__(SO_BLOCK)
sb_offsets_kernel(int T, int S, int32_t* __restrict__ tile_offsets, int32_t* __restrict__ st_offsets,
                  int64_t* __restrict__ count_out, const unsigned long long* __restrict__ area_part, int n_chunks) {
  __shared__ alignas(16) uint32_t buf[SO_BLOCK * SO_PER];
  __shared__ uint32_t wave_tot[SO_WAVES];
  if (blockIdx.x == 1) {  // (two workgroups: the two scans side by side)
    uint32_t longest = 0, n_over = 0;
    scan_counts_in_place(S, st_offsets, buf, wave_tot, &longest, (uint32_t)SB_LONG_SPLIT, &n_over);
    if (count_out) {  // the rectangles' area: the sum of the chunks' partial sums (a few hundred words)
      __shared__ unsigned long long s_sum[SO_WAVES];
      unsigned long long a = 0;
      for (int c = threadIdx.x; c < n_chunks; c += SO_BLOCK) a += area_part[c];
#pragma unroll
      for (int m = 1; m < 64; m <<= 1) a += __shfl_xor(a, m);
      if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = a;
      __syncthreads();
      if (threadIdx.x == 0) {
        unsigned long long all = 0;
#pragma unroll
        for (int w = 0; w < SO_WAVES; ++w) all += s_sum[w];
        __hip_atomic_store(count_out + 14, (int64_t)all, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    if (threadIdx.x == 0 && count_out) {  // the longest supertile segment, beside the list length (the host's path choice)
      // ... and how many segments are beyond the small launch's LDS sort: many of them are better off with the bucket passes
      __hip_atomic_store(count_out + 3, (int64_t)n_over, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(count_out + 1, (int64_t)longest, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __threadfence_system();
    }
    return;
  }
  uint32_t longest_tile = 0;
  const uint64_t total = scan_counts_in_place(T, tile_offsets, buf, wave_tot, &longest_tile);
  if (threadIdx.x == 0 && count_out) {  // the list length straight into the caller's host-visible word
    // (and the longest tile list: a host enables the heavy-tile forward -- fg_raster_config::heavy_tiles -- from it)
    __hip_atomic_store(count_out + 2, (int64_t)longest_tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(count_out, (int64_t)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
  }
}

// ---- which segments the small sort launch leaves to the others ---------------------------------------------------
// large_list: [0] = count, then the supertiles (any order) whose segment is longer than small_max -- and, with
// long_mode, not longer than SB_LONG_MIN: those are the large launch's LDS sorts.  long_list (long_mode): the segments
// beyond SB_LONG_MIN in supertile order as int4 {supertile, first chunk, first bucket, elements} behind a header
// {count, chunks, buckets, 0} and in front of a sentinel {-1, chunks, buckets, 0}: the work items of the sample sort's
// passes (chunks of LG_CHUNK elements; long_buckets(n) buckets) find their segment by bisection.
template <int NTH>
__device__ __forceinline__ void build_segment_lists(int S, const int32_t* __restrict__ st_offsets, int small_max,
                                                    int32_t* __restrict__ large_list, int4* __restrict__ long_list,
                                                    bool long_mode, int4* __restrict__ chunk_seg,
                                                    int4* __restrict__ bucket_seg, int32_t* __restrict__ over_list = nullptr) {
  constexpr int NWV = NTH / 64;
  __shared__ int s_large;
  if (long_mode && over_list && threadIdx.x == 0) over_list[0] = over_list[1] = 0;  // (the sort launches append to them)
  __shared__ uint32_t s_tot[3][NWV];
  const int lane = fg::lane_id(), wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_large = 0;
  __syncthreads();
  const int per = (S + NTH - 1) / NTH, i0 = min((int)threadIdx.x * per, S), i1 = min(i0 + per, S);
  uint32_t mine[3] = {0, 0, 0};  // long segments, their chunks, their buckets in [i0, i1)
  for (int i = i0; i < i1; ++i) {
    const int n = st_offsets[i + 1] - st_offsets[i];
    if (long_mode && n > SB_LONG_SPLIT) {
      mine[0] += 1;
      mine[1] += (uint32_t)((n + LG_CHUNK - 1) / LG_CHUNK);
      mine[2] += (uint32_t)long_buckets(n);
    } else if (n > small_max) {
      large_list[1 + atomicAdd(&s_large, 1)] = i;
    }
  }
  uint32_t incl[3];
#pragma unroll
  for (int f = 0; f < 3; ++f) {
    incl[f] = wave_incl_scan(mine[f], lane);
    if (lane == 63) s_tot[f][wave] = incl[f];
  }
  __syncthreads();
  if (threadIdx.x == 0) large_list[0] = s_large;
  if (!long_mode) {
    if (threadIdx.x == 0) long_list[0] = make_int4(0, 0, 0, 0);
    return;
  }
  uint32_t run[3], all[3];
#pragma unroll
  for (int f = 0; f < 3; ++f) {
    run[f] = incl[f] - mine[f];
    all[f] = 0;
#pragma unroll
    for (int w = 0; w < NWV; ++w) {
      const uint32_t t = s_tot[f][w];
      if (w < wave) run[f] += t;
      all[f] += t;
    }
  }
  for (int i = i0; i < i1; ++i) {
    const int n = st_offsets[i + 1] - st_offsets[i];
    if (n > SB_LONG_SPLIT) {
      long_list[1 + run[0]] = make_int4(i, (int)run[1], (int)run[2], n);
      // (the passes' work items find their segment with ONE load instead of a bisection of dependent loads)
      const int nch = (n + LG_CHUNK - 1) / LG_CHUNK, nb = long_buckets(n);
      const int off = st_offsets[i];
      for (int c = 0; c < nch; ++c) chunk_seg[run[1] + c] = make_int4((int)run[2], off, n, c);
      for (int b = 0; b < nb; ++b) bucket_seg[run[2] + b] = make_int4((int)run[2], off, i, b);
      run[0] += 1;
      run[1] += (uint32_t)nch;
      run[2] += (uint32_t)nb;
    }
  }
  if (threadIdx.x == 0) {
    long_list[0] = make_int4((int)all[0], (int)all[1], (int)all[2], 0);
    long_list[1 + all[0]] = make_int4(-1, (int)all[1], (int)all[2], 0);
  }
}
// ---- scatter --------------------------------------------------------------------------------------------
// One workgroup per chunk; a wavefront takes 64 Gaussians a round: every lane finds the owner of its slot among
// the round's (Gaussian, supertile) pairs and writes the element depth bits << 32 | id << 4 | tile mask (bit j:
// the rectangle covers tile j = 2 * (row in the supertile) + column) to the slot an LDS cursor of the supertile
// hands out (segment start + the chunks before this one, from the scanned table).  Order = (depth bits, id): the
// mask sits below the id and never decides.
// STAGE (whenever the whole image's cursors and a staging buffer fit the LDS): the chunk's elements are first put
// in LDS ordered by supertile -- the cursors then count from a LOCAL exclusive scan of the chunk's own counts --
// and go out in one sweep, a run of ~5 consecutive addresses per (chunk, supertile) instead of one scattered
// 8-byte store per element (2.45M of those left the L2 as 32-byte writes each: 70 MB for 20 MB of elements and
// +13 us over coalesced stores, profiles/r03_binning.md section 4).  Elements beyond the staging buffer's `stage_cap`
// (a chunk of huge rectangles) are stored directly, as without STAGE.
template <bool STAGE>
__global__ void __launch_bounds__(SC_BLOCK)
sb_scatter_kernel(int N, int per, const int2* __restrict__ rects, const unsigned long long* __restrict__ masks,
                  const uint32_t* __restrict__ depth_keys, int tile_w,
                  int tile_h, int band_rows, int stage_cap, const uint32_t* __restrict__ table_s,
                  const int32_t* __restrict__ tile_offsets, const int32_t* __restrict__ st_offsets,
                  uint64_t* __restrict__ entries, long long capacity, int small_max, int32_t* __restrict__ large_list,
                  int4* __restrict__ long_list, int long_mode, int4* __restrict__ chunk_seg,
                  int4* __restrict__ bucket_seg, int job_blocks, fgjobs::JobBuild jb, int32_t* __restrict__ over_list) {
  // fg_stbin_fill_jobs: the LAST job_blocks workgroups build the raster launches' job lists from the (exact) tile
  // ranges -- each the forward's and the backward's list of one XCD band -- beside the scatter, with no launch of
  // their own.  (STAGE only: one workgroup per CU by LDS, so the builder's 118 registers cost this kernel nothing;
  // in the small-segment sort launch they halved the occupancy, 54 -> 62 us, and at the head of the large-segment
  // one the builder outlasts the launch on light scenes.)
  // (STAGE: one more workgroup at the very end builds the segment lists for the sort launches -- off chunk 0's path)
  if constexpr (STAGE) {
    const int first = (int)gridDim.x - 1 - job_blocks;
    if ((int)blockIdx.x == (int)gridDim.x - 1) {
      const Geo gg = geo_of(tile_w, tile_h);
      if ((long long)tile_offsets[tile_w * tile_h] <= capacity)
        build_segment_lists<SC_BLOCK>(gg.sw * gg.sh, st_offsets, small_max, large_list, long_list, long_mode != 0, chunk_seg,
                                      bucket_seg, over_list);
      return;
    }
    if ((int)blockIdx.x >= first) {
      // (the forward's list here; the backward's -- not needed before the backward -- by eight workgroups at the end of
      // the small-segment sort launch: two builds one after the other made these workgroups the launch's longest, 36 us
      // on a clustered scene against 26 for the scatter itself)
      extern __shared__ uint32_t s_dyn[];  // (this launch's staging buffer: 132 KB, unused by this workgroup)
      fgjobs::build_jobs_block<SC_BLOCK>((int)blockIdx.x - first, jb, tile_offsets, s_dyn);
      return;
    }
  }
  extern __shared__ uint32_t s_cur[];  // [supertiles of the pass] (STAGE: + destination deltas + staging buffer)
  __shared__ uint32_t s_wave_tot[SC_WAVES];
  __shared__ int4 s_q[SC_WAVES][64];
  __shared__ unsigned long long s_qm[SC_WAVES][64];  // the owners' footprint masks
  __shared__ int32_t s_excl[SC_WAVES][64];
  __shared__ unsigned long long s_marks[SC_WAVES];
  const Geo g = geo_of(tile_w, tile_h);
  const int chunk = blockIdx.x, S = g.sw * g.sh;
  if ((long long)tile_offsets[tile_w * tile_h] > capacity) return;  // the guess was too small: the host repeats the call
  const int lane = fg::lane_id(), wave = threadIdx.x >> 6;
  if constexpr (!STAGE) {
    if (chunk == 0)
      build_segment_lists<SC_BLOCK>(S, st_offsets, small_max, large_list, long_list, long_mode != 0, chunk_seg, bucket_seg,
                                    over_list);
  }
  const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  int4* q = s_q[wave];  // {id, depth bits, rect.x, rect.y}
  unsigned long long* qm = s_qm[wave];
  int32_t* ex = s_excl[wave];
  unsigned long long* marks = &s_marks[wave];
  // the wavefront's rectangles and depth bits: all loads in flight together
  const int chunk_n = per * SC_BLOCK;
  const int g0 = chunk * chunk_n + wave * (chunk_n / SC_WAVES) + lane;
  int2 rc[SC_PER];
  uint32_t dk[SC_PER];
  unsigned long long mk[SC_PER];
#pragma unroll
  for (int r = 0; r < SC_PER; ++r) {
    const bool in = r < per && g0 + r * 64 < N;
    rc[r] = in ? rects[g0 + r * 64] : make_int2(0, 0);
    dk[r] = in ? depth_keys[g0 + r * 64] : 0u;
    mk[r] = in && masks ? masks[g0 + r * 64] : 0ull;
  }
  // STAGE: s_cur = local cursors, s_gd[st] = (global slot) - (local slot) of the chunk's run in supertile st
  const int s_pad = (S + 1) & ~1;
  uint32_t* s_gd = s_cur + s_pad;
  uint64_t* stage = reinterpret_cast<uint64_t*>(s_gd + s_pad);
  uint16_t* stage_st = reinterpret_cast<uint16_t*>(stage + stage_cap);
  uint32_t chunk_pairs = 0;
  if (STAGE) {
    const uint32_t* row = table_s + (size_t)chunk * S;
    const bool last = (chunk + 1) * chunk_n >= N;
    auto count_of = [&](int i) {  // this chunk's pairs in supertile i: the next chunk's prefix minus this one's
      return (last ? (uint32_t)(st_offsets[i + 1] - st_offsets[i]) : row[S + i]) - row[i];
    };
    const int per = (S + SC_BLOCK - 1) / SC_BLOCK, i0 = min((int)threadIdx.x * per, S), i1 = min(i0 + per, S);
    uint32_t sum = 0;
    for (int i = i0; i < i1; ++i) sum += count_of(i);
    const uint32_t incl = wave_incl_scan(sum, lane);
    if (lane == 63) s_wave_tot[wave] = incl;
    __syncthreads();
    uint32_t base = incl - sum;
#pragma unroll
    for (int w = 0; w < SC_WAVES; ++w) {
      const uint32_t t = s_wave_tot[w];
      if (w < wave) base += t;
      chunk_pairs += t;
    }
    for (int i = i0; i < i1; ++i) {
      s_cur[i] = base;
      s_gd[i] = (uint32_t)st_offsets[i] + row[i] - base;
      base += count_of(i);
    }
  }
  for (int sr0 = 0; sr0 < g.sh; sr0 += band_rows) {  // (one pass when the image's cursors fit the LDS)
    const int sr1 = min(sr0 + band_rows, g.sh), tr0 = 2 * sr0, tr1 = min(2 * sr1, tile_h);
    const int sb0 = sr0 * g.sw, nbs = (sr1 - sr0) * g.sw;
    if (!STAGE) {
      const uint32_t* row = table_s + (size_t)chunk * S + sb0;
      for (int i = threadIdx.x; i < nbs; i += SC_BLOCK) s_cur[i] = (uint32_t)st_offsets[sb0 + i] + row[i];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SC_PER; ++r) {
      if (r >= per) break;
      const int w = rc[r].y & 0xFFFF, h = rc[r].y >> 16, x0 = rc[r].x & 0xFFFF, y0 = rc[r].x >> 16;
      const int ya = max(y0, tr0), yb = min(y0 + h, tr1);
      const bool hit = w > 0 && yb > ya;
      const int sxa = x0 >> 1, snx = ((x0 + w - 1) >> 1) + 1 - sxa;
      const int sya = ya >> 1, sny = ((yb - 1) >> 1) + 1 - sya;
      const uint32_t cnt = hit ? (uint32_t)(snx * sny) : 0u;
      const uint32_t incl = wave_incl_scan(cnt, lane);
      const int total = (int)__builtin_amdgcn_readlane((int)incl, 63);
      // the round's Gaussians with pairs, compacted: owner k = the k-th of them
      const uint64_t hm = __ballot(hit);
      const int ci = __popcll(hm & lt_mask);
      if (hit) {
        q[ci] = make_int4(g0 + r * 64, (int)dk[r], rc[r].x, rc[r].y);
        if (masks) qm[ci] = mk[r];
        ex[ci] = (int)(incl - cnt);
      }
      __builtin_amdgcn_wave_barrier();
      for (int s0 = 0; s0 < total; s0 += 64) {
        // Owner of slot s0 + lane = the number of owners whose pairs end before it: the ones that ended before the
        // window (a ballot) + the END MARKS below this lane's bit in a 64-bit word the owners ending inside the
        // window OR together in LDS.  (A binary search over the exclusive counts was six dependent LDS reads.)
        if (lane == 0) *marks = 0ull;
        __builtin_amdgcn_wave_barrier();
        const int endpos = (int)incl - 1 - s0;
        if (hit && endpos >= 0 && endpos < 64) atomicOr(marks, 1ull << endpos);
        __builtin_amdgcn_wave_barrier();
        const uint64_t m = __hip_atomic_load(marks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        const int before = __popcll(__ballot(hit && (int)incl <= s0));
        const int slot = s0 + lane;
        if (slot < total) {
          const int lo = before + __popcll(m & lt_mask);
          const int4 o = q[lo];
          const int t = slot - ex[lo];
          const int ow = o.w & 0xFFFF, oh = o.w >> 16, ox0 = o.z & 0xFFFF, oy0 = o.z >> 16;
          const int oxa = ox0 >> 1, onx = ((ox0 + ow - 1) >> 1) + 1 - oxa;
          const int oya = max(oy0, tr0) >> 1;
          int ty = (int)((float)t * __builtin_amdgcn_rcpf((float)onx));
          ty -= (ty * onx > t);
          ty += ((ty + 1) * onx <= t);
          const int tx = t - ty * onx;
          const int local = (oya + ty - sr0) * g.sw + oxa + tx;
          // which of the supertile's four tiles the rectangle covers (bit j: tile 2 * row + column)
          const int c0 = 2 * (oxa + tx), r0 = 2 * (oya + ty);
          const uint32_t cols = (uint32_t)(c0 >= ox0) | ((uint32_t)(c0 + 1 < ox0 + ow) << 1);
          const uint32_t rows = (uint32_t)(r0 >= oy0) | ((uint32_t)(r0 + 1 < oy0 + oh) << 1);
          uint32_t mask = ((rows & 1u) ? cols : 0u) | ((rows & 2u) ? cols << 2 : 0u);
          if (masks) {
            // ... and the footprint mask's blocks hold (the pairs sb_count_kernel counted: a pair none of whose tiles is
            // reached has no element)
            const unsigned long long om = qm[lo];
            const int sh = 31 - __builtin_clz((unsigned)fg::footprint_block(ow, oh));
            // (columns / rows outside the rectangle are masked off above: their shifted indices may be anything in 0..7)
            const int bx0 = ((c0 - ox0) >> sh) & 7, bx1 = ((c0 + 1 - ox0) >> sh) & 7;
            const int by0 = ((r0 - oy0) >> sh) & 7, by1 = ((r0 + 1 - oy0) >> sh) & 7;
            const uint32_t ra = (uint32_t)(om >> (8 * by0)), rb = (uint32_t)(om >> (8 * by1));
            mask &= ((ra >> bx0) & 1u) | (((ra >> bx1) & 1u) << 1) | (((rb >> bx0) & 1u) << 2) | (((rb >> bx1) & 1u) << 3);
          }
          if (mask != 0u) {
            const uint32_t pos = atomicAdd(&s_cur[local], 1u);
            const uint64_t element = ((uint64_t)(uint32_t)o.y << 32) | ((uint64_t)(uint32_t)o.x << 4) | mask;
            if (!STAGE) {
              entries[pos] = element;
            } else if (pos < (uint32_t)stage_cap) {
              stage[pos] = element;
              stage_st[pos] = (uint16_t)local;
            } else {
              entries[pos + s_gd[local]] = element;
            }
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
  }
  if (STAGE) {  // the staged elements out, ordered by supertile: consecutive lanes, runs of consecutive addresses
    const uint32_t n_staged = min(chunk_pairs, (uint32_t)stage_cap);
    for (uint32_t i = threadIdx.x; i < n_staged; i += SC_BLOCK) entries[i + s_gd[stage_st[i]]] = stage[i];
  }
}

// ---- per-supertile sort + emission ------------------------------------------------------------------------
// An element: depth bits << 32 | id << 4 | tile mask (bit j: the rectangle covers tile j = 2 * (row in the
// supertile) + column).  Order = (depth bits, id): the mask sits below the id and never decides.
__device__ __forceinline__ uint64_t same_digit_lanes(unsigned d, int nbits, bool in) {
  const uint64_t in_mask = __ballot(in);
  uint32_t lo = (uint32_t)in_mask, hi = (uint32_t)(in_mask >> 32);
  for (int bit = 0; bit < nbits; ++bit) {
    const int sel = (int)(d << (31 - bit)) >> 31;
    const uint64_t m = __ballot(sel != 0);
    lo = __builtin_amdgcn_bitop3_b32(lo, (uint32_t)m, (uint32_t)sel, 0x90);
    hi = __builtin_amdgcn_bitop3_b32(hi, (uint32_t)(m >> 32), (uint32_t)sel, 0x90);
  }
  return ((uint64_t)hi << 32) | lo;
}
// wave_cnt[w][digit] counts -> start slots (digit-major, wavefronts in order inside a digit); thread = digit
template <int NW>
__device__ __forceinline__ void digit_starts(uint32_t (*wave_cnt)[256], uint32_t* scan_tmp) {
  const int lane = fg::lane_id(), wave = threadIdx.x >> 6;
  const bool own = threadIdx.x < 256;  // threads 0..255 own a digit each (wavefronts 0..3)
  uint32_t c[NW], tot = 0, incl = 0;
  if (own) {
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      c[w] = wave_cnt[w][threadIdx.x];
      tot += c[w];
    }
    incl = wave_incl_scan(tot, lane);
    if (lane == 63) scan_tmp[wave] = incl;
  }
  __syncthreads();
  if (own) {
    uint32_t run = incl - tot;
#pragma unroll
    for (int w = 0; w < 4; ++w)
      if (w < wave) run += scan_tmp[w];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      wave_cnt[w][threadIdx.x] = run;
      run += c[w];
    }
  }
  __syncthreads();
}
template <int NW>
__device__ __forceinline__ void block_min_max(uint32_t lo, uint32_t hi, uint32_t* red, uint32_t& kmin, uint32_t& kmax) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) {
    lo = min(lo, (uint32_t)__shfl_xor((int)lo, m));
    hi = max(hi, (uint32_t)__shfl_xor((int)hi, m));
  }
  const int wave = threadIdx.x >> 6;
  if (fg::lane_id() == 0) {
    red[wave] = lo;
    red[NW + wave] = hi;
  }
  __syncthreads();
  kmin = red[0];
  kmax = red[NW];
#pragma unroll
  for (int w = 1; w < NW; ++w) {
    kmin = min(kmin, red[w]);
    kmax = max(kmax, red[NW + w]);
  }
  __syncthreads();
}
// Position of element i inside its RUN -- the elements whose sorted key bits (depth bits minus the
// segment's smallest, shifted right by `low`) are equal; adjacent after the passes, in arrival order -- by
// the full order: the number of smaller elements of the run.
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
// The global-memory passes sort all key bits that differ inside the segment up to 16 (two 8-bit passes), of
// more only the TOP 16; run_position settles the rest.
__device__ __forceinline__ int unsorted_low_bits(int bits) { return bits > 16 ? bits - 16 : 0; }

// The tile lists of a supertile from its n elements in final order (cur: LDS or global), read in order by the
// workgroup: wavefront w owns the contiguous share [w span, (w + 1) span), counts its elements per tile, the
// counts become bases (wavefronts in order), and a second walk writes every element's id to base + rank.
template <int NW, typename Ptr>
__device__ __forceinline__ void emit_tiles(Ptr cur, int n, const int* tile_base, int32_t* __restrict__ flatten_ids,
                                           uint32_t (*cnt)[4]) {
  const int lane = fg::lane_id(), wave = threadIdx.x >> 6;
  const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const int span = ((n + (64 * NW) - 1) / (64 * NW)) * 64;
  const int w0 = min(wave * span, n), w1 = min(w0 + span, n);
  uint32_t c[4] = {0, 0, 0, 0};
  for (int i0 = w0; i0 < w1; i0 += 64) {
    const int i = i0 + lane;
    const uint32_t m = i < w1 ? (uint32_t)cur[i] & 15u : 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) c[j] += (uint32_t)__popcll(__ballot((m >> j) & 1u));
  }
  if (lane < 4) cnt[wave][lane] = c[lane];
  __syncthreads();
  uint32_t base[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    base[j] = (uint32_t)tile_base[j];
    for (int w = 0; w < NW; ++w)
      if (w < wave) base[j] += cnt[w][j];
  }
  for (int i0 = w0; i0 < w1; i0 += 64) {
    const int i = i0 + lane;
    const uint64_t e = i < w1 ? cur[i] : 0ull;
    const uint32_t m = (uint32_t)e & 15u, id = (uint32_t)(e >> 4) & 0x0FFFFFFFu;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint64_t bal = __ballot((m >> j) & 1u);
      if ((m >> j) & 1u) flatten_ids[base[j] + (uint32_t)__popcll(bal & lt_mask)] = (int32_t)id;
      base[j] += (uint32_t)__popcll(bal);
    }
  }
}

// A supertile too long for LDS: the same passes through global memory (a <-> b, both L2-resident), two sweeps
// per pass (per-wavefront digit counts; stable slots), then the run fix into the other buffer.  Returns the
// buffer holding the final order.  One workgroup; rare (thousands of entries).
template <int NW>
__device__ uint64_t* sort_segment_global(uint64_t* a, uint64_t* b, int n, uint32_t (*wave_cnt)[256], uint32_t* scan_tmp,
                                         uint32_t* red) {
  const int lane = fg::lane_id(), wave = threadIdx.x >> 6;
  const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  uint32_t lo = 0xFFFFFFFFu, hi = 0u;
  for (int i = threadIdx.x; i < n; i += (64 * NW)) {
    const uint32_t k = (uint32_t)(a[i] >> 32);
    lo = min(lo, k);
    hi = max(hi, k);
  }
  uint32_t kmin, kmax;
  block_min_max<NW>(lo, hi, red, kmin, kmax);
  const uint32_t range = kmax - kmin;
  const int bits = range ? 32 - __builtin_clz(range) : 0, low = unsorted_low_bits(bits), passes = (bits - low + 7) / 8;
  const int span = ((n + (64 * NW) - 1) / (64 * NW)) * 64;
  const int w0 = min(wave * span, n), w1 = min(w0 + span, n);
  uint64_t* src = a;
  uint64_t* dst = b;
  int first = low;
  for (int p = 0; p < passes; ++p) {
    const int nbits = (bits - first + (passes - p) - 1) / (passes - p);
    const uint32_t mask = (1u << nbits) - 1u;
    for (int i = threadIdx.x; i < NW * 256; i += (64 * NW)) (&wave_cnt[0][0])[i] = 0;
    __syncthreads();
    for (int i = w0 + lane; i < w1; i += 64)
      atomicAdd(&wave_cnt[wave][(((uint32_t)(src[i] >> 32) - kmin) >> first) & mask], 1u);
    __syncthreads();
    digit_starts<NW>(wave_cnt, scan_tmp);
    for (int i0 = w0; i0 < w1; i0 += 64) {
      const int i = i0 + lane;
      const bool in = i < w1;
      const uint64_t e = in ? src[i] : 0ull;
      const unsigned d = (((uint32_t)(e >> 32) - kmin) >> first) & mask;
      const uint64_t peers = same_digit_lanes(d, nbits, in);
      const int leader = in ? (int)__builtin_ctzll(peers) : lane;
      uint32_t slot = 0;
      if (in && leader == lane) slot = atomicAdd(&wave_cnt[wave][d], (uint32_t)__popcll(peers));
      slot = (uint32_t)__shfl((int)slot, leader) + (uint32_t)__popcll(peers & lt_mask);
      if (in) dst[slot] = e;
    }
    __threadfence();  // the next pass reads what other wavefronts wrote, through this CU's L1
    __syncthreads();
    uint64_t* t = src;
    src = dst;
    dst = t;
    first += nbits;
  }
  for (int i = threadIdx.x; i < n; i += (64 * NW)) {
    const uint64_t e = src[i];
    dst[run_position(src, n, i, e, kmin, low)] = e;
  }
  __threadfence();
  __syncthreads();
  return dst;
}

// ---- the LDS sort + emission of one segment by a GROUP of NW wavefronts (a whole workgroup, or one of several
// groups of a workgroup running in lockstep: every barrier below is a workgroup barrier, so all groups of a
// workgroup must call this the same number of times -- with n = 0 when they have nothing to do) ----------------
template <int NW, int KPT, int BB>
struct SortShared {
  // (the large variant gives up 256 elements so that two workgroups -- 2 x 80 KB -- fit a CU's 160 KB of LDS)
  static constexpr int MAXN = 64 * NW * KPT - (NW > 8 ? 256 : 0);
  static constexpr int NSP = 127;  // splitters of the skew path (sort_emit_lds)
  uint64_t img[MAXN];
  uint64_t sp[NSP + 1];
  alignas(16) uint32_t bucket[(1 << BB) + 4];  // counts -> exclusive bases; [1 << BB] = n
  uint32_t red[2 * NW];
  uint32_t wtot[NW];
  uint32_t scan_tmp[4];
  alignas(16) uint32_t tcnt[NW][4];
};

// ONE counting pass on the top BB of the key bits that differ inside the segment, then every bucket is put in
// order by a full (depth bits, id) comparison among its own elements.  Because the buckets get sorted anyway the
// counting pass need not be stable: an element's slot is its bucket's base + the value a returning LDS atomic on
// the bucket's counter handed it -- no ballots, no per-wavefront counters.  ~1250 elements over 2048 buckets: a
// bucket holds one or two.  (Exact depth ties crowd one bucket: every element still ranks itself in O(bucket).)
// Then the tile lists: wavefront w owns the contiguous share [w R 64, (w + 1) R 64) of the sorted run; ONE walk
// takes every element's rank among the wavefront's elements of each tile (ballots), the per-wavefront counts
// become bases, the ids go out.
// Rounds q = 0 .. KPT - 1 come in blocks: a block (4 rounds, or 3) runs iff the segment reaches it (a scalar branch) and is
// straight-line code with a per-lane `valid` bit per round inside -- its loads are in flight together.  (With
// `if (q < R && i < n) { load; use }` per round every load was followed by its own wait: KPT dependent round trips.)
#define SB_FOR_ROUNDS(q)                                 \
  _Pragma("unroll") for (int qb_ = 0; qb_ < KPT; qb_ += QB) \
    if (qb_ < R)                                         \
      _Pragma("unroll") for (int q = qb_; q < qb_ + QB; ++q)

// (src_at(i): element i of the input, i < n; EMIT = false: stop with the n elements in order in sh.img;
// SKEW = true: a skewed segment -- see below -- is re-bucketed by splitters here; false: it is sorted as it is unless
// `defer_skew`, then nothing is written and the function returns true: the caller hands the segment to a launch that can)
template <int NW, int KPT, int BB, bool EMIT = true, bool SKEW = true, typename SrcAt>
__device__ __forceinline__ bool sort_emit_lds(SortShared<NW, KPT, BB>& sh, int gt, SrcAt src_at, int n,
                                              const int* tile_base, int32_t* __restrict__ flatten_ids,
                                              bool defer_skew = false) {
  constexpr int NT = 64 * NW, NB = 1 << BB, PER = NB / NT;
  static_assert(NB % NT == 0 && PER >= 1 && (PER % 4 == 0 || PER < 4), "buckets per thread");
  constexpr int QB = KPT % 4 == 0 ? 4 : 3;
  static_assert(KPT % QB == 0, "rounds come in blocks");
  const int lane = gt & 63, gw = gt >> 6;
  const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const int R = (n + NT - 1) / NT;  // rounds; wavefront w owns [w R 64, (w + 1) R 64)
  const int ibase = gw * R * 64 + lane;
  uint32_t valid = 0;
#pragma unroll
  for (int q = 0; q < KPT; ++q) valid |= (uint32_t)(q < R && ibase + q * 64 < n) << q;
  uint64_t e[KPT];
#pragma unroll
  for (int q = 0; q < KPT; ++q) e[q] = 0;
  SB_FOR_ROUNDS(q) e[q] = src_at((valid >> q) & 1u ? ibase + q * 64 : 0);  // (R > 0 here: element 0 exists)
  uint32_t lo = 0xFFFFFFFFu, hi = 0u;
  SB_FOR_ROUNDS(q) {
    const bool v = (valid >> q) & 1u;
    e[q] = v ? e[q] : 0ull;
    lo = v ? min(lo, (uint32_t)(e[q] >> 32)) : lo;
    hi = v ? max(hi, (uint32_t)(e[q] >> 32)) : hi;
  }
  if constexpr (PER >= 4) {
#pragma unroll
    for (int k = 0; k < PER; k += 4) *reinterpret_cast<uint4*>(&sh.bucket[gt * PER + k]) = make_uint4(0, 0, 0, 0);
  } else {
#pragma unroll
    for (int k = 0; k < PER; ++k) sh.bucket[gt * PER + k] = 0;
  }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) {
    lo = min(lo, (uint32_t)__shfl_xor((int)lo, m));
    hi = max(hi, (uint32_t)__shfl_xor((int)hi, m));
  }
  if (lane == 0) {
    sh.red[gw] = lo;
    sh.red[NW + gw] = hi;
  }
  __syncthreads();
  uint32_t kmin = sh.red[0], kmax = sh.red[NW];
#pragma unroll
  for (int w = 1; w < NW; ++w) {
    kmin = min(kmin, sh.red[w]);
    kmax = max(kmax, sh.red[NW + w]);
  }
  const uint32_t range = n > 0 ? kmax - kmin : 0u;
  const int bits = range ? 32 - __builtin_clz(range) : 0, low = bits > BB ? bits - BB : 0;
  uint32_t rk[KPT], bk[KPT];  // an element's bucket and the slot inside it the counting pass handed out
#pragma unroll
  for (int q = 0; q < KPT; ++q) rk[q] = bk[q] = 0;
  SB_FOR_ROUNDS(q) {
    bk[q] = ((uint32_t)(e[q] >> 32) - kmin) >> low;
    if ((valid >> q) & 1u) rk[q] = atomicAdd(&sh.bucket[bk[q]], 1u);
  }
  __syncthreads();
  // counts -> exclusive bases: thread t owns PER consecutive buckets; returns the fullest bucket's count
  auto scan_buckets = [&]() -> uint32_t {
    uint32_t c[PER], tot = 0, mx = 0;
    if constexpr (PER >= 4) {
#pragma unroll
      for (int k = 0; k < PER; k += 4) {
        const uint4 v = *reinterpret_cast<const uint4*>(&sh.bucket[gt * PER + k]);
        c[k] = v.x; c[k + 1] = v.y; c[k + 2] = v.z; c[k + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int k = 0; k < PER; ++k) c[k] = sh.bucket[gt * PER + k];
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      tot += c[k];
      mx = max(mx, c[k]);
    }
    const uint32_t incl = wave_incl_scan(tot, lane);
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, m));
    if (lane == 63) {
      sh.wtot[gw] = incl;
      sh.red[gw] = mx;
    }
    __syncthreads();
    uint32_t run = incl - tot;
    mx = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      if (w < gw) run += sh.wtot[w];
      mx = max(mx, sh.red[w]);
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const uint32_t x = c[k];
      c[k] = run;
      run += x;
    }
    if constexpr (PER >= 4) {
#pragma unroll
      for (int k = 0; k < PER; k += 4) *reinterpret_cast<uint4*>(&sh.bucket[gt * PER + k]) = make_uint4(c[k], c[k + 1], c[k + 2], c[k + 3]);
    } else {
#pragma unroll
      for (int k = 0; k < PER; ++k) sh.bucket[gt * PER + k] = c[k];
    }
    if (gt == NT - 1) sh.bucket[NB] = run;  // = n
    __syncthreads();
    return mx;
  };
  const uint32_t fullest = scan_buckets();
  // SKEW: equal-width buckets fail a segment that is a sparse spread of depths plus a dense cluster (the rim of a
  // ball of splats: half of the elements within a few hundred ulps) or a run of exact ties -- one bucket then holds
  // hundreds of elements and the ranking below is quadratic in that (1500 in a bucket: 80 us for ONE workgroup).
  // Such a segment is bucketed again by SPLITTERS: a regular sample of up to 127 elements, ranked all against all,
  // and every element bisects them (full 64-bit comparisons: ties and skew alike end up ~n / 128 to a bucket).
  if (!SKEW && defer_skew && fullest > SB_SKEW_MAX) return true;  // (uniform across the group)
  if (SKEW && fullest > SB_SKEW_MAX) {
    using Sh = SortShared<NW, KPT, BB>;
    const int ns = min(Sh::NSP, n >> 2);  // (n > SB_SKEW_MAX here)
    SB_FOR_ROUNDS(q)
      if ((valid >> q) & 1u) sh.img[ibase + q * 64] = e[q];
    if constexpr (PER >= 4) {
#pragma unroll
      for (int k = 0; k < PER; k += 4) *reinterpret_cast<uint4*>(&sh.bucket[gt * PER + k]) = make_uint4(0, 0, 0, 0);
    } else {
#pragma unroll
      for (int k = 0; k < PER; ++k) sh.bucket[gt * PER + k] = 0;
    }
    __syncthreads();
    uint64_t x = 0;
    if (gt < ns) {
      x = sh.img[(gt * n) / ns];  // (gt < 128, n < 2^13)
      sh.sp[gt] = x;
    }
    __syncthreads();
    int smaller = 0;
    if (gt < ns)
      for (int u = 0; u < ns; ++u) smaller += sh.sp[u] < x;  // (all lanes read one address: a broadcast)
    __syncthreads();
    if (gt < ns) sh.sp[smaller] = x;  // (unique elements: a permutation)
    __syncthreads();
    SB_FOR_ROUNDS(q) {
      uint32_t b = 0;
#pragma unroll
      for (int step = 64; step >= 1; step >>= 1) {
        const uint32_t t = b + (uint32_t)step;
        const uint64_t v = sh.sp[min(t, (uint32_t)ns) - 1];
        b = (t <= (uint32_t)ns && v <= e[q]) ? t : b;
      }
      bk[q] = b;
      if ((valid >> q) & 1u) rk[q] = atomicAdd(&sh.bucket[b], 1u);
    }
    __syncthreads();
    (void)scan_buckets();
  }
  SB_FOR_ROUNDS(q)
    if ((valid >> q) & 1u) sh.img[sh.bucket[bk[q]] + rk[q]] = e[q];
  __syncthreads();
  // the order inside every bucket: an element's place = the bucket's base + the number of smaller elements in it
  int pos[KPT];
#pragma unroll
  for (int q = 0; q < KPT; ++q) pos[q] = -1;
  SB_FOR_ROUNDS(q) {
    if ((valid >> q) & 1u) {
      const int s0 = (int)sh.bucket[bk[q]], s1 = (int)sh.bucket[bk[q] + 1];
      if (s1 - s0 > 1) {
        int smaller = 0;
        for (int j = s0; j < s1; ++j) smaller += sh.img[j] < e[q];
        pos[q] = s0 + smaller;
      }
    }
  }
  __syncthreads();
  SB_FOR_ROUNDS(q)
    if (pos[q] >= 0) sh.img[pos[q]] = e[q];  // (elements alone in their bucket are in place)
  __syncthreads();
  if constexpr (!EMIT) return false;
  // ---- emission ----
  uint32_t idm[KPT], ranks[KPT];
  uint32_t pre[KPT][4], c[4] = {0, 0, 0, 0};  // (wave-uniform: scalar registers)
#pragma unroll
  for (int q = 0; q < KPT; ++q) idm[q] = 0;
  SB_FOR_ROUNDS(q)
    if ((valid >> q) & 1u) idm[q] = (uint32_t)sh.img[ibase + q * 64];  // id << 4 | mask
  SB_FOR_ROUNDS(q) {
    ranks[q] = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint64_t bal = __ballot((idm[q] >> j) & 1u);
      ranks[q] |= (uint32_t)__popcll(bal & lt_mask) << (8 * j);
      pre[q][j] = c[j];
      c[j] += (uint32_t)__popcll(bal);
    }
  }
  if (lane == 0) *reinterpret_cast<uint4*>(sh.tcnt[gw]) = make_uint4(c[0], c[1], c[2], c[3]);
  __syncthreads();
  uint32_t base[4];
  {
    uint4 t = make_uint4(0, 0, 0, 0);
    if (lane < gw) t = *reinterpret_cast<const uint4*>(sh.tcnt[lane]);  // (gw <= NW - 1 < 64)
    uint32_t v[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int m = 1; m < (NW > 1 ? NW : 2); m <<= 1) v[j] += (uint32_t)__shfl_xor((int)v[j], m);
      base[j] = (uint32_t)tile_base[j] + (uint32_t)__builtin_amdgcn_readfirstlane((int)v[j]);
    }
  }
  SB_FOR_ROUNDS(q) {
    const uint32_t id = idm[q] >> 4;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if ((idm[q] >> j) & 1u) flatten_ids[base[j] + pre[q][j] + ((ranks[q] >> (8 * j)) & 0xFFu)] = (int32_t)id;
  }
  return false;
}
#undef SB_FOR_ROUNDS

// Two launches share the supertiles: the SMALL one (512 threads, up to 3072 elements, 32 KB of LDS) has a workgroup
// per supertile and takes all of them on the 1M / 1080p scene; the LARGE one (1024 threads, up to 8064 elements) is
// SB_LARGE_GRID persistent workgroups over the list of longer segments the scatter kernel left (empty: they return);
// beyond its capacity a segment goes through global memory.
// first list slot of the supertile's four tiles (0 for tiles outside the image)
__device__ __forceinline__ void supertile_tile_bases(int st, int tile_w, int tile_h, const int32_t* __restrict__ tile_offsets,
                                                     int (&tile_base)[4], int (&tile_id)[4]) {
  const Geo g = geo_of(tile_w, tile_h);
  const int sy = st / g.sw, sx = st - sy * g.sw;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int tx = 2 * sx + (j & 1), ty = 2 * sy + (j >> 1);
    const bool inside = tx < tile_w && ty < tile_h;
    tile_id[j] = inside ? ty * tile_w + tx : -1;
    tile_base[j] = inside ? tile_offsets[tile_id[j]] : 0;
  }
}

// (returns true when the small launch leaves a skewed segment to a later one -- `defer_skew`, sort_emit_lds)
template <int NW, int KPT, int BB, bool SMALL>
__device__ __forceinline__ bool sort_supertile(SortShared<NW, KPT, BB>& sh, int st, int tile_w, int tile_h,
                                               const int32_t* __restrict__ tile_offsets,
                                               const int32_t* __restrict__ st_offsets, uint64_t* __restrict__ entries,
                                               uint64_t* __restrict__ scratch, bool over, int total,
                                               int32_t* __restrict__ flatten_ids, int32_t* __restrict__ list_offsets,
                                               bool defer_skew = false) {
  constexpr int MAXN = SortShared<NW, KPT, BB>::MAXN;  // elements sorted in LDS by this variant
  uint32_t(*wave_cnt)[256] = reinterpret_cast<uint32_t(*)[256]>(sh.bucket);
  static_assert((1 << BB) >= NW * 256, "the fallback's counters live in the bucket array");
  const int T = tile_w * tile_h;
  int tile_base[4], tile_id[4];
  supertile_tile_bases(st, tile_w, tile_h, tile_offsets, tile_base, tile_id);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    // the ranges the CONSUMERS of flatten_ids read: the tile ranges when the list fits, empty lists when it
    // does not (nothing is filled then; whoever was enqueued speculatively behind this call walks nothing)
    if (SMALL && tile_id[j] >= 0 && threadIdx.x == 0) {
      list_offsets[tile_id[j]] = over ? 0 : tile_base[j];
      if (tile_id[j] == T - 1) list_offsets[T] = over ? 0 : total;
    }
  }
  const int off = st_offsets[st], n = st_offsets[st + 1] - off;
  if (over || n <= 0) return false;
  if (SMALL && n > MAXN) return false;  // (on the large launch's list, or a long segment)
  if constexpr (!SMALL) {
    if (n > MAXN) {
      const uint64_t* fin = sort_segment_global<NW>(entries + off, scratch + off, n, wave_cnt, sh.scan_tmp, sh.red);
      emit_tiles<NW>(fin, n, tile_base, flatten_ids, sh.tcnt);
      return false;
    }
  }
  const uint64_t* __restrict__ src = entries + off;
  // (the small launch keeps its registers -- four workgroups per CU -- and leaves skew to others or, without
  // FG_STBIN_LONG_SEGMENTS, sorts the crowded bucket the quadratic way)
  return sort_emit_lds<NW, KPT, BB, true, !SMALL>(sh, (int)threadIdx.x, [src](int i) { return src[i]; }, n, tile_base,
                                                  flatten_ids, defer_skew);
}

// fg_stbin_fill_jobs without the staged scatter (tile grids beyond 65535 supertiles: the builders have no launch to ride in --
// inside the large-segment sort they cost it 35 spilled registers): the raster launches' job lists by a launch of their own
__global__ void __launch_bounds__(1024)
sb_build_jobs_kernel(fgjobs::JobBuild jb, const int32_t* __restrict__ tile_offsets) {
  __shared__ uint32_t s_rows[fgjobs::FG_BAND_MAX_ROWS];
  fgjobs::build_jobs_block<1024>((int)blockIdx.x, jb, tile_offsets, s_rows);
}

// ---- long segments: sample sort ------------------------------------------------------------------------------------
// A segment of n > SB_LONG_MIN elements (tens of thousands of splats over one 32 x 32-pixel supertile: a dense
// cluster) is cut into k = long_buckets(n) buckets by k - 1 SPLITTERS taken from a sorted regular sample of the
// segment; elements are unique 64-bit values (depth bits | id | mask), so any distribution of depths -- ties
// included -- splits into buckets of about n / k elements.  Passes, every one of them many workgroups wide:
//   sample   (a work item of the large sort launch) one workgroup per segment sorts LG_SA * k samples in LDS, writes the
//            splitters, zeroes the bucket counters
//   count    per chunk of LG_CHUNK elements: bucket of every element by bisection over the splitters in LDS, counts
//            per bucket and per (bucket, tile of the supertile) -> global counters
//   scatter  per chunk: the same buckets; a run per (chunk, bucket) reserved with ONE returning atomic on the bucket's
//            cursor; elements to `scratch` at segment start + bucket start + run + rank.  Order inside a bucket: any.
//   sort     (work items of the SMALL sort launch, beside the supertiles it takes whole) per bucket: where it starts
//            and what the buckets in front send to each of the four tiles (a block reduction over the counters), then
//            the LDS sort + emission used for whole segments.  (A bucket beyond the LDS capacity -- the sample was
//            unlucky by a factor of three -- goes through global memory in one workgroup, as whole segments did before.)
// Same lists, bit for bit: inside a tile the order is the total order on (depth bits, id).
template <int NW, int KPT, int BB>
__device__ __forceinline__ void sample_long_segment(SortShared<NW, KPT, BB>& sh, const uint64_t* __restrict__ src, int n,
                                                    int bucket_base, const LongTables& lt) {
  constexpr int NT = 64 * NW;
  const int k = long_buckets(n), s = long_samples(k, SortShared<NW, KPT, BB>::MAXN);
  const uint64_t step = ((uint64_t)n << 16) / (uint64_t)s;  // sample i = element floor(i n / s): distinct positions
  sort_emit_lds<NW, KPT, BB, false>(sh, (int)threadIdx.x, [src, step](int i) { return src[((uint64_t)i * step) >> 16]; }, s,
                                    nullptr, nullptr);
  for (int t = threadIdx.x; t < k; t += NT) {
    if (t + 1 < k) lt.split[bucket_base + t] = sh.img[(int)(((long long)(t + 1) * s) / k)];
    lt.cnt[bucket_base + t] = 0u;
    lt.cursor[bucket_base + t] = 0u;
    lt.tcnt[bucket_base + t] = make_uint4(0, 0, 0, 0);
  }
  __syncthreads();  // (sh.img is the next work item's)
}

__global__ void __launch_bounds__(64 * SB_LARGE_WAVES)
sb_sort_large_kernel(int tile_w, int tile_h, const int32_t* __restrict__ tile_offsets,
                     const int32_t* __restrict__ st_offsets, const int32_t* __restrict__ large_list,
                     const int4* __restrict__ long_list, LongTables lt, uint64_t* __restrict__ entries,
                     uint64_t* __restrict__ scratch, long long capacity, int32_t* __restrict__ flatten_ids) {
  __shared__ SortShared<SB_LARGE_WAVES, SB_LARGE_KPT, SB_LARGE_BUCKET_BITS> sh;
  static_assert(SortShared<SB_LARGE_WAVES, SB_LARGE_KPT, SB_LARGE_BUCKET_BITS>::MAXN == SB_LONG_MIN, "long = beyond this launch's LDS sort");
  constexpr int job_blocks = 0;
  const int total = tile_offsets[tile_w * tile_h];
  if ((long long)total > capacity) return;  // (the scatter kernel wrote no list then)
  // work items: the long segments' sample step first (the head of a chain of three more launches), then the LDS sorts
  const int n_long = long_list[0].x, count = large_list[0];
  for (int k = (int)blockIdx.x - job_blocks; k < n_long + count; k += (int)gridDim.x - job_blocks) {
    if (k < n_long) {
      const int4 ls = long_list[1 + k];
      sample_long_segment(sh, entries + st_offsets[ls.x], ls.w, ls.z, lt);
    } else {
      sort_supertile<SB_LARGE_WAVES, SB_LARGE_KPT, SB_LARGE_BUCKET_BITS, false>(
          sh, large_list[1 + k - n_long], tile_w, tile_h, tile_offsets, st_offsets, entries, scratch, false, total,
          flatten_ids, nullptr);
    }
    __syncthreads();
  }
}

// bucket of element e among k: the number of splitters <= e (s_split[0 .. k - 1) ascending, in LDS)
__device__ __forceinline__ int long_bucket_of(const uint64_t* s_split, int k, uint64_t e) {
  int b = 0;
#pragma unroll
  for (int step = 512; step >= 1; step >>= 1) {
    const int t = b + step;
    const uint64_t v = s_split[min(t, k - 1) - 1];  // (k >= 2)
    b = (t <= k - 1 && v <= e) ? t : b;
  }
  return b;
}
static_assert(LG_KMAX <= 1024, "long_bucket_of bisects 10 levels");

static_assert(LG_CHUNK < 65536, "16-bit per-chunk tile counters");

// COUNT AND SCATTER IN ONE PASS (round 6; rounds 4-5: a count launch, then a scatter launch that needed the counts' prefix
// sums for the buckets' starts).  Every bucket owns a SLAB of SB_LONG_SLAB elements of `scratch` -- twice its target size,
// what the LDS sort takes -- at (first bucket slot of the segment + bucket) x SB_LONG_SLAB: a chunk reserves a run per
// (chunk, bucket) with ONE returning atomic on the bucket's cursor and stores at slab + run + rank; the cursor is the
// bucket's element count when the launch ends.  An element beyond the slab (the sample was unlucky by a factor of two: a
// 1e-6 event per bucket) is NOT stored; the cursor still counts it and sb_long_overflow_kernel sends the bucket's whole SEGMENT
// through global memory from `entries`, which this pass leaves intact.  Per (bucket, tile of the supertile) counts ride
// along as before.  Order inside a bucket: any.
__global__ void __launch_bounds__(LG_BLOCK)
sb_long_scatter_kernel(const int32_t* __restrict__ tile_offsets, int n_tiles, long long capacity,
                       const int32_t* __restrict__ st_offsets, const int4* __restrict__ long_list,
                       const uint64_t* __restrict__ entries, uint64_t* __restrict__ scratch, LongTables lt, int slab_limit) {
  __shared__ uint64_t s_split[LG_KMAX];
  __shared__ uint32_t s_hist[LG_KMAX], s_base[LG_KMAX];
  __shared__ unsigned long long s_tc[LG_KMAX];  // four 16-bit counters: elements of this chunk per tile of the supertile
  if ((long long)tile_offsets[n_tiles] > capacity) return;
  const int4 hdr = long_list[0];
  int cur = -1;
  for (int w = blockIdx.x; w < hdr.y; w += gridDim.x) {
    const int4 ci = lt.chunk_seg[w];  // {first bucket slot, first element, elements, chunk in the segment}
    const int slot0 = ci.x, n = ci.z, k = long_buckets(n), c = ci.w;
    const uint64_t* __restrict__ src = entries + ci.y;
    if (slot0 != cur) {
      __syncthreads();
      for (int t = threadIdx.x; t < k - 1; t += LG_BLOCK) s_split[t] = lt.split[slot0 + t];
      cur = slot0;
    }
    for (int t = threadIdx.x; t < k; t += LG_BLOCK) {
      s_hist[t] = 0u;
      s_tc[t] = 0ull;
    }
    __syncthreads();
    uint64_t e[LG_PER];
    int bk[LG_PER];
    uint32_t rk[LG_PER];
#pragma unroll
    for (int q = 0; q < LG_PER; ++q) {
      const int i = c * LG_CHUNK + q * LG_BLOCK + (int)threadIdx.x;
      e[q] = src[min(i, n - 1)];
    }
#pragma unroll
    for (int q = 0; q < LG_PER; ++q) {
      const int i = c * LG_CHUNK + q * LG_BLOCK + (int)threadIdx.x;
      bk[q] = -1;
      rk[q] = 0u;
      if (i < n) {
        bk[q] = long_bucket_of(s_split, k, e[q]);
        rk[q] = atomicAdd(&s_hist[bk[q]], 1u);
        const uint32_t m = (uint32_t)e[q] & 15u;
        atomicAdd(&s_tc[bk[q]], (unsigned long long)(m & 1u) | ((unsigned long long)((m >> 1) & 1u) << 16) |
                                    ((unsigned long long)((m >> 2) & 1u) << 32) | ((unsigned long long)((m >> 3) & 1u) << 48));
      }
    }
    __syncthreads();
    // one run per (chunk, bucket): s_base[b] = the run's first slot in the bucket's slab
    for (int t = threadIdx.x; t < k; t += LG_BLOCK) {
      const uint32_t h = s_hist[t];
      if (h) {
        s_base[t] = atomicAdd(&lt.cursor[slot0 + t], h);
        const unsigned long long tc = s_tc[t];
        uint32_t* tg = reinterpret_cast<uint32_t*>(&lt.tcnt[slot0 + t]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t v = (uint32_t)(tc >> (16 * j)) & 0xFFFFu;
          if (v) atomicAdd(&tg[j], v);
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < LG_PER; ++q)
      if (bk[q] >= 0) {
        const uint32_t pos = s_base[bk[q]] + rk[q];
        if (pos < (uint32_t)slab_limit) scratch[(size_t)(slot0 + bk[q]) * SB_LONG_SLAB + pos] = e[q];
      }
    __syncthreads();
  }
}

// The SMALL sort launch: one workgroup per supertile, whole segments up to 3072 elements.  (Its registers -- four
// workgroups per CU -- are the headline scene's binning time: the long segments' buckets have a launch of their own.)
// long_mode: a skewed segment (sort_emit_lds) is left on over_list as -(supertile + 1) for sb_long_sort_kernel.
__global__ void __launch_bounds__(64 * SB_SMALL_WAVES)
sb_sort_small_kernel(int tile_w, int tile_h, const int32_t* __restrict__ tile_offsets,
                     const int32_t* __restrict__ st_offsets, uint64_t* __restrict__ entries,
                     uint64_t* __restrict__ scratch, long long capacity, int32_t* __restrict__ flatten_ids,
                     int32_t* __restrict__ list_offsets, int32_t* __restrict__ over_list, int job_blocks,
                     fgjobs::JobBuild jb, const int4* __restrict__ long_list, LongTables lt) {
  __shared__ SortShared<SB_SMALL_WAVES, SB_SMALL_KPT, SB_SMALL_BUCKET_BITS> sh;
  // supertile = workgroup id: neighbours go to different XCDs.  (By band the long segments of a centre-weighted
  // image all land on the two or three XCDs that own the middle rows.)
  // fg_stbin_fill_jobs: the backward's job list, one XCD's a workgroup, in FRONT of the supertiles (behind them they
  // would start when the launch is nearly over: 26 -> 30 us); the builder's ~60 registers fit this launch's 64, its row
  // sums lie in the sort's element buffer
  if ((int)blockIdx.x < job_blocks) {
    static_assert(sizeof(sh.img) >= fgjobs::FG_BAND_MAX_ROWS * sizeof(uint32_t), "row sums in the element buffer");
    fgjobs::build_jobs_block<64 * SB_SMALL_WAVES>((int)blockIdx.x + 8, jb, tile_offsets, reinterpret_cast<uint32_t*>(sh.img));
    return;
  }
  const int st = (int)blockIdx.x - job_blocks;
  if (st >= ((tile_w + 1) >> 1) * ((tile_h + 1) >> 1)) return;
  const int total = tile_offsets[tile_w * tile_h];
  const bool skewed = sort_supertile<SB_SMALL_WAVES, SB_SMALL_KPT, SB_SMALL_BUCKET_BITS, true>(
      sh, st, tile_w, tile_h, tile_offsets, st_offsets, entries, scratch, (long long)total > capacity, total, flatten_ids,
      list_offsets, over_list != nullptr);
  if (skewed && threadIdx.x == 0) over_list[2 + atomicAdd(over_list, 1)] = -(st + 1);
  // long mode: a segment beyond this launch's LDS sort is a LONG segment -- its own workgroup, which has nothing else to
  // do, takes the sample step (splitters from a sorted regular sample; it used to be a work item of the large-segment
  // launch, which long mode now does without: every segment beyond SB_LONG_SPLIT = this launch's capacity is long)
  if (over_list && (long long)total <= capacity) {
    using Sh = SortShared<SB_SMALL_WAVES, SB_SMALL_KPT, SB_SMALL_BUCKET_BITS>;
    static_assert(Sh::MAXN == SB_LONG_SPLIT, "in long mode no segment is left for the large launch");
    const int n = st_offsets[st + 1] - st_offsets[st];
    if (n > Sh::MAXN) {
      __shared__ int s_long;
      if (threadIdx.x == 0) s_long = -1;
      __syncthreads();
      const int n_long = long_list[0].x;  // (the list is in supertile order; a look at every entry is one round trip)
      for (int t = threadIdx.x; t < n_long; t += 64 * SB_SMALL_WAVES)
        if (long_list[1 + t].x == st) s_long = t;
      __syncthreads();
      if (s_long >= 0) {
        const int4 ls = long_list[1 + s_long];
        sample_long_segment(sh, entries + st_offsets[st], ls.w, ls.z, lt);
      }
    }
  }
}

// The buckets of the long segments (slot = work item) and the skewed segments the small launch left: LG_SORT_GRID
// persistent workgroups, the small launch's LDS sort with the splitter path for skew.  A bucket whose cursor ran beyond its
// slab (the sample was unlucky by a factor of two: with 32 samples per bucket a 1e-6 event per bucket) goes on over_list's
// second list for sb_long_overflow_kernel.
// (eight wavefronts per SIMD = four workgroups per CU, as the small launch: 9 spilled registers on the skew path)
__global__ void __launch_bounds__(64 * SB_SMALL_WAVES) __attribute__((amdgpu_waves_per_eu(8, 8)))
sb_long_sort_kernel(int tile_w, int tile_h, long long capacity, const int32_t* __restrict__ tile_offsets,
                    const int32_t* __restrict__ st_offsets, const int4* __restrict__ long_list, LongTables lt,
                    const uint64_t* __restrict__ entries, const uint64_t* __restrict__ scratch,
                    int32_t* __restrict__ flatten_ids, int slab_limit) {
  using Sh = SortShared<SB_SMALL_WAVES, SB_SMALL_KPT, SB_SMALL_BUCKET_BITS>;
  __shared__ Sh sh;
  __shared__ int s_tile_base[4];
  constexpr int NT = 64 * SB_SMALL_WAVES;
  static_assert(2 * LG_T <= Sh::MAXN && SB_LONG_SLAB == Sh::MAXN, "a bucket's slab is what the LDS sort takes: twice the target");
  if ((long long)tile_offsets[tile_w * tile_h] > capacity) return;
  const int lane = fg::lane_id(), wave = threadIdx.x >> 6;
  const int n_slots = long_list[0].z, n_items = n_slots + lt.over_list[0];
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const uint64_t* src;
    int n, st;
    uint32_t pre[4] = {0, 0, 0, 0};
    if (item >= n_slots) {  // a whole (skewed) segment
      st = -lt.over_list[2 + item - n_slots] - 1;
      const int off = st_offsets[st];
      n = st_offsets[st + 1] - off;
      src = entries + off;
    } else {
      // a bucket: where it starts in each of the four tile lists = sums over the buckets in front of it
      const int4 bi = lt.bucket_seg[item];  // {first bucket slot, first element, supertile, bucket in the segment}
      st = bi.z;
      const int b = bi.w;
      for (int t = threadIdx.x; t < b; t += NT) {
        const uint4 tc = lt.tcnt[bi.x + t];
        pre[0] += tc.x; pre[1] += tc.y; pre[2] += tc.z; pre[3] += tc.w;
      }
#pragma unroll
      for (int f = 0; f < 4; ++f) {
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) pre[f] += (uint32_t)__shfl_xor((int)pre[f], m);
        if (lane == 0) sh.bucket[f * SB_SMALL_WAVES + wave] = pre[f];
      }
      __syncthreads();
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        pre[f] = 0;
#pragma unroll
        for (int w = 0; w < SB_SMALL_WAVES; ++w) pre[f] += sh.bucket[f * SB_SMALL_WAVES + w];
      }
      n = (int)lt.cursor[item];  // (the scatter pass's cursor: the bucket's element count)
      src = scratch + (size_t)item * SB_LONG_SLAB;
      if (n > slab_limit) {
        if (threadIdx.x == 0) lt.over_list[2 + n_slots_cap(lt) + atomicAdd(lt.over_list + 1, 1)] = item;
        n = 0;
      }
    }
    {  // the tile bases wait in LDS for the emission (registers: three workgroups per CU)
      int tile_base[4], tile_id[4];
      supertile_tile_bases(st, tile_w, tile_h, tile_offsets, tile_base, tile_id);
      __syncthreads();  // (sh.bucket is the sort's from here; s_tile_base is free again)
      if (threadIdx.x < 4) s_tile_base[threadIdx.x] = tile_base[threadIdx.x] + (int)pre[threadIdx.x];
    }
    if (n > 0)  // (uniform; the sort's first barrier publishes s_tile_base)
      sort_emit_lds<SB_SMALL_WAVES, SB_SMALL_KPT, SB_SMALL_BUCKET_BITS>(sh, (int)threadIdx.x, [src](int i) { return src[i]; }, n,
                                                                         s_tile_base, flatten_ids);
    __syncthreads();
  }
}

// The buckets that outgrew their slabs (over_list's second list; normally none: 5 us of an empty grid -- a bucket of twice its
// target size is a 1e-4 ... 1e-6 event, seen about once in a hundred calls on a scene with a hundred buckets), one workgroup each.
// The scatter pass dropped what did not fit the slab but left `entries` intact: the bucket's elements are gathered from its
// segment again -- those between the bucket's two splitters -- into the workgroup's own stretch of a small arena (64 x 7936
// elements of the workspace), then sorted and emitted like any bucket, by the large launch's LDS sort (7936 elements).  A bucket beyond even that -- the sample missed it by a factor of five -- sends its
// whole segment through global memory and emits all of the segment's tile lists again (the same values where its other buckets
// had written theirs).  That sort permutes the segment's `entries` in place, so it is decided per SEGMENT from the cursors
// before anything is gathered: no workgroup re-gathers a bucket of a segment another one is sorting whole.  (Measured and dropped: this as the tail of sb_long_sort_kernel's last workgroup, found by a ticket --
// 1024 same-address returning atomics and the global sort's registers in that kernel: 34 -> 218 us.)
__global__ void __launch_bounds__(64 * SB_LARGE_WAVES)
sb_long_overflow_kernel(int tile_w, int tile_h, long long capacity, const int32_t* __restrict__ tile_offsets,
                        const int32_t* __restrict__ st_offsets, LongTables lt, uint64_t* __restrict__ entries,
                        uint64_t* __restrict__ scratch, int32_t* __restrict__ flatten_ids, int whole_limit) {
  using Sh = SortShared<SB_LARGE_WAVES, SB_LARGE_KPT, SB_LARGE_BUCKET_BITS>;
  __shared__ Sh sh;
  __shared__ int s_tile_base[4];
  __shared__ uint32_t s_n;
  __shared__ int s_mine;
  constexpr int NT = 64 * SB_LARGE_WAVES;
  if ((long long)tile_offsets[tile_w * tile_h] > capacity) return;
  const int lane = fg::lane_id(), wave = threadIdx.x >> 6;
  const int count = lt.over_list[1];
  uint32_t(*wave_cnt)[256] = reinterpret_cast<uint32_t(*)[256]>(sh.bucket);
  for (int v = blockIdx.x; v < count; v += gridDim.x) {
    const int slot = lt.over_list[2 + n_slots_cap(lt) + v];
    const int4 bi = lt.bucket_seg[slot];  // {first bucket slot, first element, supertile, bucket in the segment}
    const int st = bi.z, b = bi.w, off = bi.y, n_seg = st_offsets[st + 1] - st_offsets[st], k = long_buckets(n_seg);
    const uint64_t lo = b > 0 ? lt.split[bi.x + b - 1] : 0ull, hi = b + 1 < k ? lt.split[bi.x + b] : ~0ull;
    uint64_t* tmp = lt.arena + (size_t)blockIdx.x * Sh::MAXN;
    int tile_base[4], tile_id[4];
    supertile_tile_bases(st, tile_w, tile_h, tile_offsets, tile_base, tile_id);
    // A segment with ANY bucket beyond the arena's stretch goes through global memory whole, once (the claim: lt.cnt of its
    // first bucket slot, zeroed by the sample step), and that sort permutes `entries` in place: no workgroup may gather a
    // bucket of that segment from `entries` meanwhile.  The scatter pass's cursors hold every bucket's count, so every
    // workgroup that meets the segment decides the same way before touching it.
    // (whole_limit < 0, FG_STBIN_TEST_SMALL_SLABS: from LG_T + 192 elements in even supertiles -- both paths run in the tests)
    const uint32_t limit = whole_limit >= 0 ? (uint32_t)whole_limit : (st & 1) ? (uint32_t)Sh::MAXN : (uint32_t)(LG_T + 192);
    int big = 0;
    for (int t = threadIdx.x; t < k; t += NT) big |= lt.cursor[bi.x + t] > limit ? 1 : 0;
    if (threadIdx.x == 0) s_n = 0u;
    const int whole = __syncthreads_or(big);
    if (whole) {
      if (threadIdx.x == 0) s_mine = atomicCAS(lt.cnt + bi.x, 0u, 0xFFFFFFFFu) == 0u ? 1 : 0;
      __syncthreads();
      if (s_mine) {
        const uint64_t* fin = sort_segment_global<SB_LARGE_WAVES>(entries + off, scratch + (size_t)bi.x * SB_LONG_SLAB, n_seg, wave_cnt,
                                                                   sh.scan_tmp, sh.red);
        emit_tiles<SB_LARGE_WAVES>(fin, n_seg, tile_base, flatten_ids, sh.tcnt);
      }
      __syncthreads();
      continue;
    }
    // (bucket of e = the number of splitters <= e: bucket b holds lo <= e < hi; the last bucket has no upper splitter)
    for (int i = threadIdx.x; i < n_seg; i += NT) {
      const uint64_t e = entries[(size_t)off + i];
      if (e >= lo && (b + 1 >= k || e < hi)) {
        const uint32_t pos = atomicAdd(&s_n, 1u);
        if (pos < (uint32_t)Sh::MAXN) tmp[pos] = e;
      }
    }
    __syncthreads();
    const int n_b = min((int)s_n, Sh::MAXN);  // (= the bucket's cursor: the scatter pass's rule, <= whole_limit <= MAXN)
    uint32_t pre[4] = {0, 0, 0, 0};
    for (int t = threadIdx.x; t < b; t += NT) {
      const uint4 tc = lt.tcnt[bi.x + t];
      pre[0] += tc.x; pre[1] += tc.y; pre[2] += tc.z; pre[3] += tc.w;
    }
#pragma unroll
    for (int f = 0; f < 4; ++f) {
#pragma unroll
      for (int m = 1; m < 64; m <<= 1) pre[f] += (uint32_t)__shfl_xor((int)pre[f], m);
      if (lane == 0) sh.bucket[f * SB_LARGE_WAVES + wave] = pre[f];
    }
    __syncthreads();
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      pre[f] = 0;
#pragma unroll
      for (int w = 0; w < SB_LARGE_WAVES; ++w) pre[f] += sh.bucket[f * SB_LARGE_WAVES + w];
    }
    __syncthreads();
    if (threadIdx.x < 4) s_tile_base[threadIdx.x] = tile_base[threadIdx.x] + (int)pre[threadIdx.x];
    const uint64_t* src = tmp;
    sort_emit_lds<SB_LARGE_WAVES, SB_LARGE_KPT, SB_LARGE_BUCKET_BITS>(sh, (int)threadIdx.x, [src](int i) { return src[i]; }, n_b,
                                                                       s_tile_base, flatten_ids);
    __syncthreads();
  }
}

size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
int n_chunks_of(int N) {
  const int c = chunk_rounds(N) * SC_BLOCK;
  return (N + c - 1) / c;
}

struct CountWs {
  uint32_t *table_t, *table_s;
  int32_t* st_offsets;
  int32_t* large_list;  // [0] = how many supertiles the small sort launch leaves to the large one, then their indices
  int4* long_list;      // build_segment_lists
  unsigned long long* area_part;  // [chunks]: the rectangles' area per chunk (sb_count_kernel -> sb_offsets_kernel)
  size_t bytes;
};
// (layouts are computed on offsets: a size query passes no base, and nullptr + offset is undefined behaviour)
template <typename T>
T* ws_at(void* base, size_t offset) {
  return base ? reinterpret_cast<T*>(static_cast<char*>(base) + offset) : nullptr;
}
CountWs count_ws(void* base, int N, const Geo& g) {
  const size_t nc = (size_t)n_chunks_of(N > 0 ? N : 1), T = (size_t)g.tile_w * g.tile_h, S = (size_t)g.sw * g.sh;
  CountWs w;
  size_t o = 0;
  w.table_t = ws_at<uint32_t>(base, o);
  o += al256(nc * T * 4);
  w.table_s = ws_at<uint32_t>(base, o);
  o += al256(nc * S * 4);
  w.st_offsets = ws_at<int32_t>(base, o);
  o += al256((S + 1) * 4);
  w.large_list = ws_at<int32_t>(base, o);
  o += al256((S + 1) * 4);
  w.long_list = ws_at<int4>(base, o);
  o += al256((S + 2) * 16);
  w.area_part = ws_at<unsigned long long>(base, o);
  o += al256(nc * 8);
  w.bytes = o;
  return w;
}
size_t count_lds_bytes(const Geo& g) {
  const int nsr = count_band_rows(g);
  return ((size_t)(2 * nsr + 1) * (g.tile_w + 1) + (size_t)(nsr + 1) * (g.sw + 1)) * 4;
}

}  // namespace

extern "C" int fg_stbin_supported(int N, int tile_w, int tile_h) {
  if (N < 0 || N >= (1 << 28) || tile_w <= 0 || tile_h <= 0 || tile_w > 1023 || tile_h > 1023) return 0;
  const Geo g = geo_of(tile_w, tile_h);
  return count_band_rows(g) > 0 && scatter_band_rows(g) > 0 ? 1 : 0;
}

extern "C" size_t fg_stbin_count_workspace_bytes(int N, int tile_w, int tile_h) {
  if (N < 0 || tile_w <= 0 || tile_h <= 0) return 0;
  return count_ws(nullptr, N, geo_of(tile_w, tile_h)).bytes;
}

extern "C" int fg_stbin_count(int N, const int32_t* tile_rects, const uint64_t* tile_masks, int tile_w, int tile_h,
                              int32_t* tile_offsets, int64_t* count_out, void* workspace, size_t workspace_bytes,
                              fg_stream_t stream) {
  if (N <= 0 || tile_w <= 0 || tile_h <= 0 || !tile_rects || !tile_offsets || !workspace) return FG_ERR_INVALID_ARG;
  if (!fg_stbin_supported(N, tile_w, tile_h)) return FG_ERR_UNSUPPORTED;
  const Geo g = geo_of(tile_w, tile_h);
  const CountWs w = count_ws(workspace, N, g);
  if (workspace_bytes < w.bytes) return FG_ERR_WORKSPACE;
  hipStream_t s = fg_hip_stream(stream);
  const int T = tile_w * tile_h, S = g.sw * g.sh, nc = n_chunks_of(N);
  hipLaunchKernelGGL(sb_count_kernel, dim3(nc), dim3(SC_BLOCK), count_lds_bytes(g), s, N, chunk_rounds(N),
                     reinterpret_cast<const int2*>(tile_rects), reinterpret_cast<const unsigned long long*>(tile_masks), tile_w,
                     tile_h, count_band_rows(g), w.table_t, w.table_s, w.area_part);
  const int wg_t = (T + SC_COLS - 1) / SC_COLS, wg_s = (S + SC_COLS - 1) / SC_COLS;
  hipLaunchKernelGGL(sb_columns_kernel, dim3(wg_t + wg_s), dim3(SB_BLOCK), 0, s, T, S, nc, wg_t, w.table_t, w.table_s,
                     tile_offsets, w.st_offsets);
  hipLaunchKernelGGL(sb_offsets_kernel, dim3(2), dim3(SO_BLOCK), 0, s, T, S, tile_offsets, w.st_offsets, count_out, w.area_part, nc);
  FG_RETURN_IF_LAUNCH_FAILED();
  return FG_OK;
}

namespace {
// buckets of all long segments together: sum of ceil(n / LG_T) over segments of more than SB_LONG_MIN elements
size_t long_buckets_max(size_t capacity) { return capacity / LG_T + capacity / SB_LONG_SPLIT + 2; }  // (sum of ceil(n / LG_T))
struct FillWs {
  uint64_t *entries, *scratch;
  LongTables lt;
  size_t bytes;
};
FillWs fill_ws(void* base, size_t capacity) {
  const size_t kb = long_buckets_max(capacity);
  FillWs w;
  size_t o = 0;
  w.entries = ws_at<uint64_t>(base, o);
  o += al256(capacity * 8);
  w.scratch = ws_at<uint64_t>(base, o);
  // (the long segments' buckets own SLABS of SB_LONG_SLAB elements here, a slab per bucket slot: 3 x capacity at most)
  o += al256((kb * (size_t)SB_LONG_SLAB > capacity ? kb * (size_t)SB_LONG_SLAB : capacity) * 8);
  w.lt.split = ws_at<uint64_t>(base, o);
  o += al256(kb * 8);
  w.lt.tcnt = ws_at<uint4>(base, o);
  o += al256(kb * 16);
  w.lt.cnt = ws_at<uint32_t>(base, o);
  o += al256(kb * 4);
  w.lt.cursor = ws_at<uint32_t>(base, o);
  o += al256(kb * 4);
  w.lt.bucket_seg = ws_at<int4>(base, o);
  o += al256(kb * 16);
  w.lt.over_cap = (int)(capacity / SB_SKEW_MAX + 2);  // (a skewed segment has more elements than SB_SKEW_MAX)
  w.lt.over_list = ws_at<int32_t>(base, o);
  o += al256((2 + (size_t)w.lt.over_cap + kb) * 4);
  w.lt.chunk_seg = ws_at<int4>(base, o);
  o += al256((capacity / LG_CHUNK + capacity / SB_LONG_SPLIT + 2) * 16);
  w.lt.arena = ws_at<uint64_t>(base, o);
  o += al256((size_t)SB_LONG_OVER_GRID * SB_LONG_MIN * 8);
  w.bytes = o;
  return w;
}
}  // namespace

extern "C" size_t fg_stbin_fill_workspace_bytes(int64_t capacity) {
  return fill_ws(nullptr, (size_t)(capacity > 0 ? capacity : 1)).bytes;
}

namespace {

int stbin_fill(int N, const uint32_t* depth_keys, const int32_t* tile_rects, const uint64_t* tile_masks, int tile_w, int tile_h,
               int64_t capacity,
               const int32_t* tile_offsets, const void* count_workspace, int32_t* flatten_ids, int32_t* list_offsets,
               void* workspace, size_t workspace_bytes, const fgjobs::JobBuild* jobs, int flags, fg_stream_t stream) {
  if (N <= 0 || capacity <= 0 || tile_w <= 0 || tile_h <= 0) return FG_ERR_INVALID_ARG;
  if (!depth_keys || !tile_rects || !tile_offsets || !count_workspace || !flatten_ids || !list_offsets || !workspace)
    return FG_ERR_INVALID_ARG;
  if (flags & ~(FG_STBIN_LONG_SEGMENTS | FG_STBIN_TEST_SMALL_SLABS)) return FG_ERR_INVALID_ARG;
  if (!fg_stbin_supported(N, tile_w, tile_h)) return FG_ERR_UNSUPPORTED;
  if (workspace_bytes < fg_stbin_fill_workspace_bytes(capacity)) return FG_ERR_WORKSPACE;
  hipStream_t s = fg_hip_stream(stream);
  const Geo g = geo_of(tile_w, tile_h);
  const CountWs w = count_ws(const_cast<void*>(count_workspace), N, g);
  const FillWs fw = fill_ws(workspace, (size_t)capacity);
  const int nc = n_chunks_of(N);
  uint64_t* entries = fw.entries;
  uint64_t* scratch = fw.scratch;
  const int band_rows = scatter_band_rows(g), S = g.sw * g.sh, T = tile_w * tile_h;
  const int long_mode = (flags & FG_STBIN_LONG_SEGMENTS) ? 1 : 0;
  constexpr int small_max = SortShared<SB_SMALL_WAVES, SB_SMALL_KPT, SB_SMALL_BUCKET_BITS>::MAXN;
  // staging buffer: whatever the LDS leaves beside the two [S] arrays, if that is worth it (10 bytes per element)
  const int s_pad = (S + 1) & ~1;
  int stage_cap = band_rows == g.sh && S <= 65535 ? ((int)SB_SCATTER_LDS_BYTES - 8 * s_pad) / 10 : 0;
  stage_cap = stage_cap >= SB_MIN_STAGE ? (stage_cap & ~63) : 0;
  bool want_jobs = jobs && (jobs->jobs_fwd || jobs->jobs_bwd), bwd_jobs_in_sort = false;
  if (stage_cap) {
    // dynamic LDS beyond 64 KB is an opt-in per function AND per device: remember which devices have it
    static std::atomic<uint64_t> attr_devices{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
      stage_cap = 0;
    } else if (!((attr_devices.load(std::memory_order_relaxed) >> dev) & 1u)) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(sb_scatter_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)SB_SCATTER_LDS_BYTES) == hipSuccess) {
        attr_devices.fetch_or(1ull << dev, std::memory_order_relaxed);
      } else {
        (void)hipGetLastError();
        stage_cap = 0;  // the un-staged scatter needs no opt-in
      }
    }
  }
  if (stage_cap) {
    const size_t lds = (size_t)8 * s_pad + (size_t)10 * stage_cap;
    // (the job lists and the segment lists ride here; without the staged scatter: in the large-segment sort launch / chunk 0)
    hipLaunchKernelGGL(sb_scatter_kernel<true>, dim3(nc + (want_jobs ? 8 : 0) + 1), dim3(SC_BLOCK), lds, s, N, chunk_rounds(N),
                       reinterpret_cast<const int2*>(tile_rects), reinterpret_cast<const unsigned long long*>(tile_masks),
                       depth_keys, tile_w, tile_h, band_rows, stage_cap, w.table_s,
                       tile_offsets, w.st_offsets, entries, (long long)capacity, small_max, w.large_list, w.long_list,
                       long_mode, fw.lt.chunk_seg, fw.lt.bucket_seg, want_jobs ? 8 : 0, want_jobs ? *jobs : fgjobs::JobBuild{},
                       fw.lt.over_list);
    want_jobs = false;
    bwd_jobs_in_sort = jobs && jobs->jobs_bwd;
  } else {
    hipLaunchKernelGGL(sb_scatter_kernel<false>, dim3(nc), dim3(SC_BLOCK), (size_t)band_rows * g.sw * 4, s, N, chunk_rounds(N),
                       reinterpret_cast<const int2*>(tile_rects), reinterpret_cast<const unsigned long long*>(tile_masks),
                       depth_keys, tile_w, tile_h, band_rows, 0, w.table_s,
                       tile_offsets, w.st_offsets, entries, (long long)capacity, small_max, w.large_list, w.long_list,
                       long_mode, fw.lt.chunk_seg, fw.lt.bucket_seg, 0, fgjobs::JobBuild{}, fw.lt.over_list);
  }
  if (want_jobs) hipLaunchKernelGGL(sb_build_jobs_kernel, dim3(fgjobs::FG_JOB_BLOCKS), dim3(1024), 0, s, *jobs, tile_offsets);
  // (long mode: every segment beyond the small launch's capacity is a long segment -- no large launch; the small launch's
  // workgroups of those segments take their sample step, the bucket passes follow it)
  if (!long_mode)
    hipLaunchKernelGGL(sb_sort_large_kernel, dim3(S < SB_LARGE_GRID ? S : SB_LARGE_GRID), dim3(64 * SB_LARGE_WAVES), 0, s,
                       tile_w, tile_h, tile_offsets, w.st_offsets, w.large_list, w.long_list, fw.lt, entries, scratch,
                       (long long)capacity, flatten_ids);
  hipLaunchKernelGGL(sb_sort_small_kernel, dim3(S + (bwd_jobs_in_sort ? 8 : 0)), dim3(64 * SB_SMALL_WAVES), 0, s, tile_w, tile_h,
                     tile_offsets, w.st_offsets, entries, scratch, (long long)capacity, flatten_ids, list_offsets,
                     long_mode ? fw.lt.over_list : nullptr, bwd_jobs_in_sort ? 8 : 0,
                     bwd_jobs_in_sort ? *jobs : fgjobs::JobBuild{}, w.long_list, fw.lt);
  // (FG_STBIN_TEST_SMALL_SLABS: a bucket counts as overflowing from LG_T + 64 elements -- two of five do -- and a segment of an
  // even supertile with a bucket beyond LG_T + 192 goes through global memory whole, so that tests reach both paths of
  // sb_long_overflow_kernel, which a 1e-6 event would not)
  const int slab_limit = (flags & FG_STBIN_TEST_SMALL_SLABS) ? LG_T + 64 : SB_LONG_SLAB;
  if (long_mode) {  // (three launches behind the small sort -- round 5: four, a count pass in front of the scatter)
    hipLaunchKernelGGL(sb_long_scatter_kernel, dim3(LG_GRID), dim3(LG_BLOCK), 0, s, tile_offsets, T, (long long)capacity,
                       w.st_offsets, w.long_list, entries, scratch, fw.lt, slab_limit);
    hipLaunchKernelGGL(sb_long_sort_kernel, dim3(LG_SORT_GRID), dim3(64 * SB_SMALL_WAVES), 0, s, tile_w, tile_h,
                       (long long)capacity, tile_offsets, w.st_offsets, w.long_list, fw.lt, entries, scratch, flatten_ids, slab_limit);
    hipLaunchKernelGGL(sb_long_overflow_kernel, dim3(SB_LONG_OVER_GRID), dim3(64 * SB_LARGE_WAVES), 0, s, tile_w, tile_h, (long long)capacity,
                       tile_offsets, w.st_offsets, fw.lt, entries, scratch, flatten_ids,
                       (flags & FG_STBIN_TEST_SMALL_SLABS) ? -1 : SortShared<SB_LARGE_WAVES, SB_LARGE_KPT, SB_LARGE_BUCKET_BITS>::MAXN);
  }
  FG_RETURN_IF_LAUNCH_FAILED();
  return FG_OK;
}

}  // namespace

extern "C" int fg_stbin_fill(int N, const uint32_t* depth_keys, const int32_t* tile_rects, const uint64_t* tile_masks,
                             int tile_w, int tile_h,
                             int64_t capacity, const int32_t* tile_offsets, const void* count_workspace,
                             int32_t* flatten_ids, int32_t* list_offsets, void* workspace, size_t workspace_bytes,
                             int flags, fg_stream_t stream) {
  return stbin_fill(N, depth_keys, tile_rects, tile_masks, tile_w, tile_h, capacity, tile_offsets, count_workspace, flatten_ids,
                    list_offsets, workspace, workspace_bytes, nullptr, flags, stream);
}

extern "C" int fg_stbin_fill_jobs(int N, const uint32_t* depth_keys, const int32_t* tile_rects, const uint64_t* tile_masks,
                                  int tile_w, int tile_h, int64_t capacity, const int32_t* tile_offsets, const void* count_workspace,
                                  int32_t* flatten_ids, int32_t* list_offsets, void* workspace, size_t workspace_bytes,
                                  int width, int height, int tile_size, int32_t* jobs_fwd, int32_t* jobs_bwd,
                                  int bwd_list_shares, const fg_raster_config* config, int flags, int64_t* ckpt_need_out,
                                  fg_stream_t stream) {
  if (width <= 0 || height <= 0 || tile_size <= 0) return FG_ERR_INVALID_ARG;
  if ((width + tile_size - 1) / tile_size != tile_w || (height + tile_size - 1) / tile_size != tile_h)
    return FG_ERR_INVALID_ARG;
  fgjobs::JobBuild jb;
  const int rc = fgjobs::plan_jobs(width, height, tile_size, jobs_fwd, jobs_bwd, bwd_list_shares, config, &jb);
  if (rc != FG_OK) return rc;
  jb.need_out = reinterpret_cast<long long*>(ckpt_need_out);
  return stbin_fill(N, depth_keys, tile_rects, tile_masks, tile_w, tile_h, capacity, tile_offsets, count_workspace, flatten_ids,
                    list_offsets, workspace, workspace_bytes, &jb, flags, stream);
}
