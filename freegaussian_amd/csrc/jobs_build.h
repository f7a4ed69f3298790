// Job lists of the raster launches: the device code that builds them and the plan the host derives from a
// fg_raster_config.  Shared by raster.hip (fg_raster_build_jobs: a launch of its own) and stbin.hip
// (fg_stbin_fill_jobs: eight extra workgroups at the end of the scatter launch, so that no launch
// stands between the sorted lists and the raster forward).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/fgraster.h"

namespace fgjobs {

// The eight XCDs' shares of the tile grid in the mixed launches: an nx x (8 / nx) arrangement of
// rectangles, one per XCD, each walked column-major (a tile and its vertical neighbours, which share
// most of their splats, run back to back in one L2).  nx = 1 (whole-row bands) is what runs; the other
// arrangements are an A/B knob (band_nx).
struct Band {
  int c0, ncols, r0, nrows;
};
__host__ __device__ __forceinline__ Band band_of_xcd(int xcd, int tile_w, int tile_h, int nx) {
  const int ny = 8 / nx, rx = xcd % nx, ry = xcd / nx;
  Band b;
  b.c0 = (rx * tile_w) / nx;
  b.ncols = ((rx + 1) * tile_w) / nx - b.c0;
  b.r0 = (ry * tile_h) / ny;
  b.nrows = ((ry + 1) * tile_h) / ny - b.r0;
  return b;
}
__device__ __forceinline__ int band_tile(const Band& b, int idx, int tile_w) {
  const int col = idx / b.nrows;
  return (b.r0 + idx - col * b.nrows) * tile_w + b.c0 + col;
}
// Whole-row bands (nx = 1) of the job lists are SPANS of the row-major tile sequence: [t0, t1) may begin and end inside
// a row, so that the eight XCDs hold equal numbers of tiles (68 tile rows of a 1080p frame are 8 or 9 rows a band: the
// XCDs with 9 finished 4 % after the others in both launches, profiles/r04_job_timeline.md).  Order inside a span:
// column-major over its rows (a tile and its vertical neighbours back to back in one L2), the columns ragged at the top
// and at the bottom.
struct Span {
  int t0, t1;
};
__device__ __forceinline__ int span_tile(const Span& s, int idx, int tile_w) {
  // column-major over the span's rows r0 .. r1 (r1 = the row of its last tile): column c holds the complete rows between
  // them, row r0 if c >= c0 and row r1 if c <= c1 -- three runs of columns of constant height
  const int r0 = s.t0 / tile_w, c0 = s.t0 - r0 * tile_w;
  const int last = s.t1 - 1, r1 = last / tile_w, c1 = last - r1 * tile_w;
  if (r0 == r1) return s.t0 + idx;  // (within one row)
  const int mid = r1 - r0 - 1;  // complete rows
  // runs: [0, a) , [a, b) , [b, W) with a = min(c0, c1 + 1), b = max(c0, c1 + 1)
  const int a = min(c0, c1 + 1), b = max(c0, c1 + 1);
  const int h0 = mid + 1;                              // columns < a: row r1 only (c <= c1) ... or row r0 only (c >= c0): see below
  const int h1 = mid + (c0 <= c1 + 1 ? 2 : 0);         // columns in [a, b): both partial rows (c0 <= c <= c1) or neither
  const int h2 = mid + 1;                              // columns >= b
  // columns < a have c < c0 (no row r0) and c <= c1 (row r1) when a = c0 <= c1 + 1; when a = c1 + 1 < c0 they have c <= c1
  // too: in both cases exactly row r1 of the two.  Columns >= b have c >= c0 (row r0) and c > c1 (no row r1).
  int col, k;  // k: position inside the column, top to bottom
  const int n0 = a * h0, n1 = (b - a) * h1;
  if (idx < n0) {
    col = idx / h0;
    k = idx - col * h0;
    return (r0 + 1 + k) * tile_w + col;  // rows r0 + 1 .. r1
  }
  if (idx < n0 + n1) {
    const int j = idx - n0;
    col = a + j / h1;
    k = j - (col - a) * h1;
    return (c0 <= c1 + 1 ? r0 + k : r0 + 1 + k) * tile_w + col;  // rows r0 .. r1, or the complete rows only
  }
  const int j = idx - n0 - n1;
  col = b + j / h2;
  k = j - (col - b) * h2;
  return (r0 + k) * tile_w + col;  // rows r0 .. r1 - 1
}

// Job lists (fg_raster_build_jobs): job sizes chosen by POSITION as above and by CONTENT -- a tile
// whose list is longer than total * a4 / 65536 becomes four single-strip jobs, longer than
// total * a2 / 65536 two two-strip jobs.  On a scene with half of the Gaussians in a small ball (longest list 20x the
// mean, scripts/clustered_check.py) whole-tile jobs for those tiles made the forward 0.81 ms; every
// tile in quarters 0.39 ms.  Layout of a list (int32): [0..7] jobs per XCD, then 8 segments of
// `cap` = 8 x (tiles of the largest XCD band) entries, entry = tile << 3 | (strip + 1) in the XCD's
// column-major order.  (Giving every tile four workgroups and letting the unused ones return was
// tried first: the empty workgroups in front of live ones cost the forward 0.218 -> 0.275 ms, and
// with the live one always in slot 0 every whole-tile job landed on the same SIMD of its CU.)
// forward lists: bit 30 of an entry = "the backward will not cut this tile's list into shares": the
// forward then writes no compositing checkpoints for it
constexpr int FG_JOB_NO_CKPT = 1 << 30;
// HEAVY tiles of the forward (raster.hip, raster_fwd_body MODE 1 / 2 / 3; JobParams::heavy_len): in the main list a
// heavy tile is four single-strip jobs with bit 28 set -- they walk only the first FG_HEAVY_PREFIX entries of the list
// (most long lists saturate their pixels well inside that: nothing more happens to such a tile) and leave the state there.
// Two more lists of the same shape follow the main one, for the two launches behind it: the LOCAL jobs -- [0..7] jobs per
// XCD, 8 segments of FG_LOCAL_CAP entries, entry = tile << 8 | part: batches [part, part + 1) x heavy_batches_per_job
// of the list BEHIND the prefix, every 64-entry batch composited by itself -- and the COMBINE jobs -- 8 segments of
// FG_HEAVY_CAP entries, four per heavy tile: tile << 3 | (strip + 1).
constexpr int FG_JOB_PREFIX = 1 << 28;
#ifndef FG_HEAVY_MIN_BATCHES
#define FG_HEAVY_MIN_BATCHES 4
#endif
#ifndef FG_HEAVY_BWD_PARTS
#define FG_HEAVY_BWD_PARTS 16
#endif
#ifndef FG_HEAVY_PREFIX_ENTRIES
#define FG_HEAVY_PREFIX_ENTRIES 2048
#endif
constexpr int FG_HEAVY_PREFIX = FG_HEAVY_PREFIX_ENTRIES;  // entries the four strip jobs of a heavy tile walk serially
constexpr int FG_LOCAL_CAP = 8192;         // local jobs per XCD at most
constexpr int FG_HEAVY_CAP = 2048;         // combine jobs per XCD at most (512 heavy tiles in an XCD's band)
constexpr int FG_LOCAL_WORDS = 8 + 8 * FG_LOCAL_CAP, FG_HEAVY_WORDS = 8 + 8 * FG_HEAVY_CAP;
constexpr int FG_SEG_ENTRIES_H = 64;       // (= FG_SEG_ENTRIES of raster.hip: a batch)
// CHECKPOINT GRID of a tile's list (round 5): a checkpoint in front of every 64th entry for the list's first `fine` entries,
// in front of every 128th behind them (fine: a multiple of 64; FG_SEG_FINE_NEVER = every 64th throughout, rounds 2-4).  The
// backward wants shares of ~100 entries: a list of the tail (400-700 entries) needs the fine grid for its five shares, a
// list of thousands has segments to spare -- and they are most of the slots (and of the forward's checkpoint writes).
constexpr int FG_SEG_FINE_NEVER = 1 << 30;
__host__ __device__ __forceinline__ int seg_index(int rel, int fine) {  // the segment entry `rel` (from the list's start) lies in
  return rel < fine ? rel >> 6 : (fine >> 6) + ((rel - fine) >> 7);
}
__host__ __device__ __forceinline__ int seg_bound(int c, int fine) {  // first entry of segment c
  const int nf = fine >> 6;
  return c <= nf ? c << 6 : fine + ((c - nf) << 7);
}
__host__ __device__ __forceinline__ int seg_count(int n, int fine) {  // segments of a list of n entries = its checkpoint slots
  return n <= fine ? (n + 63) >> 6 : (fine >> 6) + ((n - fine + 127) >> 7);
}
__host__ __device__ __forceinline__ bool seg_starts_at(int rel, int fine) {  // a segment starts at entry `rel` (a multiple of 64)
  return rel < fine || ((rel - fine) & 127) == 0;
}
__host__ __device__ __forceinline__ int heavy_batches_per_job(int len) {
  // (at most 64 jobs per tile: the longest lists -- a dense cluster seen end on -- are the ones the prefix finishes)
  const int nb = (len - FG_HEAVY_PREFIX + FG_SEG_ENTRIES_H - 1) / FG_SEG_ENTRIES_H, per = (nb + 63) / 64;
  return per > FG_HEAVY_MIN_BATCHES ? per : FG_HEAVY_MIN_BATCHES;
}
__host__ __device__ __forceinline__ int heavy_local_jobs(int len) {
  const int nb = (len - FG_HEAVY_PREFIX + FG_SEG_ENTRIES_H - 1) / FG_SEG_ENTRIES_H, per = heavy_batches_per_job(len);
  return (nb + per - 1) / per;
}
struct JobParams {
  int tail4, tail2, s4, s2;
  int max_jobs;  // workgroups per XCD of the launch that will read the list
  // list shares instead of pixel strips (the segmented backward, struct Segments): the last seg_tail
  // tiles of the sequence become seg_parts jobs each, a tile longer than the s2 threshold as many
  // jobs (up to 16) as make its shares about half that threshold long; entry = tile << 8 | part << 4 |
  // (parts - 1), the part numbers rotated by the tile's position
  int seg_parts, seg_tail;
  // graded tail: the last seg_tail2 (<= seg_tail) tiles get seg_parts2 (>= seg_parts) jobs -- the jobs
  // that run while the launch drains are the shortest ones
  int seg_parts2, seg_tail2;
  // forward lists: a tile with a list longer than this is a HEAVY tile (0 = off); backward share lists: such a tile
  // gets up to 64 shares (16 otherwise)
  int heavy_len;
  // forward lists: what is left of a heavy tile behind its prefix jobs goes to WIDE jobs (raster.hip raster_fwd_wide_kernel):
  // four (tile, strip) entries in the heavy tiles' list, no local jobs
  int heavy_wide;
};
// (thr_h: the heavy threshold of this build round, 0x7fffffff = none)
__device__ __forceinline__ int job_count(const JobParams& p, int idx, int n, int tail4, int tail2, int thr4,
                                         int thr2, int len, int thr_h = 0x7fffffff) {
  if (p.seg_parts > 1) {
    int c = idx >= n - min(p.seg_tail, n) ? p.seg_parts : 1;
    if (idx >= n - min(p.seg_tail2, n)) c = max(c, p.seg_parts2);
    if (len > thr2) {
      const int share = max(thr2 >> 1, 1);
      c = max(c, min(len > thr_h ? FG_HEAVY_BWD_PARTS : 16, (len + share - 1) / share));
    }
    return c;
  }
  if (len > thr_h) return 4;  // (a heavy tile: four strip jobs over the list's first FG_HEAVY_PREFIX / FG_WIDE_PREFIX entries)
  int level = idx >= n - tail4 ? 2 : (idx >= n - tail4 - tail2 ? 1 : 0);
  level = max(level, len > thr4 ? 2 : (len > thr2 ? 1 : 0));
  return 1 << level;
}
// what one build needs on the device (kernel argument, by value)
struct JobBuild {
  int tile_w, tile_h, nx, cap;
  JobParams pf, pb;
  int32_t* jobs_fwd;
  int32_t* jobs_bwd;
  // balanced row bands (nx == 1): an XCD's band holds up to rows_limit tile rows (what the launches' workgroups per XCD
  // can take; 0 = the equal bands of band_of_xcd)
  int main_words;  // words of the main list: the heavy tiles' local and combine lists start there (forward list, heavy_len > 0)
  int rows_limit;
  int balance_percent;  // ... when the heaviest equal span's cost exceeds this many percent of the mean's; 0: equal ROW bands
                        // (rounds 1-3, A/B); -1: equal spans, the cost is not looked at (a host that knows the scene is even)
  // COMPACT CHECKPOINT SLOTS (list shares; slot_budget > 0): the checkpoint buffer has 8 x slot_budget slots, an XCD's
  // band owns slot_budget of them, and a tile the backward may cut into shares (or a heavy tile) gets ceil(len / 64)
  // consecutive ones -- granted from the END of the band's sequence (the positional tail first) while they last; a
  // tile without slots runs unsplit, without checkpoints.  slot_tab[tile] = first slot or -1, written into BOTH lists
  // at word tab_offset (every workgroup of a build computes the same grants from the same ranges).  need_out
  // (nullable; pinned host memory or device, NINE words): word xcd = the slots the band's candidates would take together,
  // for the host to size the next buffer by; word 8 = what the cost pass of the shares decided (1 by cost, 0 equal, -1 not run).  slot_budget = 0: no table, slot index by formula (raster.hip seg_slot_base).
  int slot_budget, tab_offset;
  long long* need_out;
  int seg_fine;  // the checkpoint grid (seg_count: slots per tile)
};
constexpr int FG_BAND_MAX_ROWS = 1024;
#ifndef FG_BAND_COST_CAP4
#define FG_BAND_COST_CAP4 12  // a tile's list length counts up to this many quarters of the mean (a longer list saturates)
// (round 4: 10, tuned on lists that still held the entries the footprint masks drop since round 5 -- 12 % of a round splat's:
// the mean fell, the saturated tiles' real cost did not; re-swept, profiles/r05_clustered_thresholds.md: 80 % of the
// Gaussians in a ball of 0.2, backward 0.47 -> 0.36 ms)
#endif

// The eight XCDs' row bands by CONTENT: equal shares of the tiles' expected walking cost instead of equal numbers of
// rows.  A cluster of splats under one XCD's band made that XCD the launch (half of the Gaussians in a ball: three XCDs
// finish at 390 / 570 us, forward / backward, five at 280 / 420 and idle: profiles/r04_job_timeline.md section 2); a
// workgroup taking jobs from a shared cursor instead costs every job a same-address atomic (57 ns each, serialised:
// forward 0.20 -> 0.38 ms, profiles/r04_job_stealing.md).  Cost of a tile = its list length, capped at 2.5 times the
// mean (a long list saturates its pixels and is not walked to its end; the forward alone would like 6 x, the backward
// 2.5-3 x: half of the Gaussians in a ball of 0.4, forward / backward ms at 6 x 0.261 / 0.395, 4 x 0.288 / 0.370, 3 x
// 0.288 / 0.339, 2.5 x 0.283 / 0.334, 2 x 0.283 / 0.336 -- one set of bands serves both lists, the checkpoint slots are
// granted per band) + a quarter of the mean (the job itself).  All
// sixteen workgroups of a build compute the same boundaries from the same tile ranges.  The equal bands stay when the
// heaviest of them is within balance_percent of the mean (the launch policy is tuned on them; on the uniform bench scene the
// cost model's bands were 3% slower than the equal ones).  row0[0 .. 8] in LDS.
template <int NTH>
__device__ __forceinline__ void balanced_row_bands(const JobBuild& jb, const int32_t* __restrict__ tile_offsets, int* tile0,
                                                   uint32_t* s_roww) {  // s_roww: FG_BAND_MAX_ROWS words of LDS
  __shared__ int row0[9];
  const int tile_w = jb.tile_w, tile_h = jb.tile_h, T = tile_w * tile_h;
  const bool uniform = jb.nx != 1 || jb.balance_percent <= 0 || jb.rows_limit <= 0 || tile_h < 16 || tile_h > FG_BAND_MAX_ROWS;
  if (!uniform) {
    for (int r = threadIdx.x; r < tile_h; r += NTH) s_roww[r] = 0u;
    __syncthreads();
    const uint32_t total = (uint32_t)tile_offsets[T], mean = total / (uint32_t)T;
    const uint32_t cap = (uint32_t)FG_BAND_COST_CAP4 * mean / 4u + 16u, fixed = mean / 4u + 4u;
    // (four tiles per thread a trip, all loads in flight together; a thread's tiles are consecutive: mostly one row)
    for (int t0 = 4 * (int)threadIdx.x; t0 < T; t0 += 4 * NTH) {
      int32_t o[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) o[k] = tile_offsets[min(t0 + k, T)];
      int row = t0 / tile_w, next = (row + 1) * tile_w;
      uint32_t acc = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (t0 + k >= T) break;
        if (t0 + k == next) {
          atomicAdd(&s_roww[row], acc);
          acc = 0;
          ++row;
          next += tile_w;
        }
        acc += min((uint32_t)(o[k + 1] - o[k]), cap) + fixed;
      }
      atomicAdd(&s_roww[row], acc);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    bool by_rows = false;  // the boundaries are rows (balanced by cost) / equal numbers of tiles
    if (jb.nx == 1 && jb.balance_percent == 0) {  // balance_bands = 0: the equal row bands of rounds 1-3 (A/B)
      for (int x = 0; x < 8; ++x) row0[x] = band_of_xcd(x, tile_w, tile_h, 1).r0;
      row0[8] = tile_h;
      by_rows = true;
    } else if (!uniform) {
      unsigned long long W = 0, cum = 0, heaviest = 0;
      for (int x = 0; x < 8; ++x) {  // the equal spans' shares: left alone unless one of them is well above the mean
        // (a span's cost from the row sums: its partial first and last rows count by the fraction of the row it holds)
        const long long t0 = (long long)x * T / 8, t1 = (long long)(x + 1) * T / 8;
        unsigned long long w8 = 0;  // in 1/tile_w of a row's cost
        for (int r = (int)(t0 / tile_w); r <= (int)((t1 - 1) / tile_w); ++r) {
          const long long a = r * (long long)tile_w > t0 ? r * (long long)tile_w : t0;
          const long long b = (r + 1) * (long long)tile_w < t1 ? (r + 1) * (long long)tile_w : t1;
          w8 += (unsigned long long)s_roww[r] * (unsigned long long)(b - a);
        }
        const unsigned long long w = w8 / (unsigned)tile_w;
        W += w;
        heaviest = w > heaviest ? w : heaviest;
        row0[x] = band_of_xcd(x, tile_w, tile_h, 1).r0;
      }
      row0[8] = tile_h;
      int r = 0;
      by_rows = 8 * heaviest * 100 > W * (unsigned)jb.balance_percent;
      if (by_rows)
      for (int x = 0; x < 8; ++x) {
        row0[x] = r;
        const int left = tile_h - r, others = 7 - x;
        const int most = min(jb.rows_limit, left - others);                 // leave a row for every XCD behind
        const int least = max(1, left - others * jb.rows_limit);           // ... and no more than they can take
        const unsigned long long target = W * (unsigned)(x + 1) / 8u;
        int n = 0;
        while (n < most && (n < least || cum + s_roww[r + n] / 2u <= target)) cum += s_roww[r + n++];
        r += x == 7 ? left : n;
      }
    }
    for (int x = 0; x <= 8; ++x) tile0[x] = by_rows ? row0[x] * tile_w : (int)((long long)x * T / 8);
    // [9]: what the cost pass decided (1: bands by cost, 0: the equal spans stood), -1 when it did not run
    tile0[9] = uniform || jb.balance_percent <= 0 ? -1 : (by_rows ? 1 : 0);
  }
  __syncthreads();
}
constexpr int FG_JOB_BLOCKS = 16;  // workgroups of one build: 8 XCD bands x (forward list, backward list)

// host: fill `out` for fg_raster_build_jobs' arguments (raster.hip, where the launch policy lives);
// FG_OK and out->jobs_fwd == out->jobs_bwd == nullptr when there is nothing to build
__attribute__((visibility("hidden"))) int plan_jobs(int width, int height, int tile_size, int32_t* jobs_fwd, int32_t* jobs_bwd, int bwd_list_shares,
              const fg_raster_config* config, JobBuild* out);

// workgroup `block` (0 .. FG_JOB_BLOCKS) of a build, NTH threads (a multiple of 64, all of the workgroup)
// (reuse_bands: a workgroup's second call -- its other list -- takes the band boundaries its first call left in LDS)
// (scratch: FG_BAND_MAX_ROWS words of LDS the caller can spare for the duration of the call)
template <int NTH>
__device__ __forceinline__ void build_jobs_block(int block, const JobBuild& jb, const int32_t* __restrict__ tile_offsets,
                                                 uint32_t* scratch, bool reuse_bands = false) {
  constexpr int NWV = NTH / 64;
  __shared__ int wave_tot[NWV];
  __shared__ int carry;
  __shared__ int s_row0[10];
  const int xcd = block & 7;
  const bool bwd = block >= 8;
  const int tile_w = jb.tile_w, tile_h = jb.tile_h, cap = jb.cap;
  const JobParams &pf = jb.pf, &pb = jb.pb;
  int32_t* jobs = bwd ? jb.jobs_bwd : jb.jobs_fwd;
  if (!jobs) return;
  const JobParams p = bwd ? pb : pf;
  const Band band = band_of_xcd(xcd, tile_w, tile_h, jb.nx);  // (nx != 1: rectangles, an A/B knob)
  Span span{0, 0};
  if (jb.nx == 1) {
    if (!reuse_bands) balanced_row_bands<NTH>(jb, tile_offsets, s_row0, scratch);
    span = Span{s_row0[xcd], s_row0[xcd + 1]};
  }
  // (the host may skip the cost pass for a shape whose last calls all kept the equal spans: word 8 of need_out)
  if (jb.nx == 1 && !bwd && xcd == 0 && jb.need_out && threadIdx.x == 0 && !reuse_bands)
    __hip_atomic_store(jb.need_out + 8, (long long)s_row0[9], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const bool spans = jb.nx == 1;
  // INTERLEAVED shares (balance_percent == -2, fg_raster_config::balance_bands = 3): the 2 x 2-tile blocks of the image (the
  // binning's supertiles: four tiles that share most of their splats) dealt to the XCDs round-robin in raster order -- every
  // XCD a uniform sample of the image, balanced on ANY content without a cost model.  (Round 6, profiles/r06_xcd_shares.md:
  // the cost bands price a tile by its list length capped at three times the mean; on 17 layouts they had not been tuned on
  // they lost up to 12 % to plain equal spans and won up to 40 % against them -- and a trained scene ran 10 % faster with
  // interleaved blocks than with either.)  A block at an odd grid's right / bottom edge holds tiles outside the image:
  // tile_at = -1, no job.
  const bool inter = spans && jb.balance_percent == -2;
  const int sw2 = (tile_w + 1) >> 1, n_blocks = sw2 * ((tile_h + 1) >> 1);
  const int n = inter ? 4 * ((n_blocks - xcd + 7) / 8) : (spans ? span.t1 - span.t0 : band.nrows * band.ncols);
  auto tile_at = [&](int idx) {
    if (inter) {
      const int g = (idx >> 2) * 8 + xcd, sy = g / sw2, sx = g - sy * sw2;
      const int ty = 2 * sy + ((idx >> 1) & 1), tx = 2 * sx + (idx & 1);
      return tx < tile_w && ty < tile_h ? ty * tile_w + tx : -1;
    }
    return spans ? span_tile(span, idx, tile_w) : band_tile(band, idx, tile_w);
  };
  const int total = tile_offsets[tile_w * tile_h];
  const int tail4 = min(p.tail4, n), tail2 = min(p.tail2, n - tail4);
  // thresholds in 1/65536 of the total list length (64-bit product: total can exceed 2^31 / 65536)
  int thr4 = p.s4 ? (int)(((int64_t)total * p.s4) >> 16) : 0x7fffffff;
  int thr2 = p.s2 ? (int)(((int64_t)total * p.s2) >> 16) : 0x7fffffff;
  const int thr2_b = pb.s2 ? (int)(((int64_t)total * pb.s2) >> 16) : 0x7fffffff;  // the backward's, as given
  int thr_h = p.heavy_len > 0 ? p.heavy_len : 0x7fffffff;
  __shared__ long long heavy_tot[NWV];
  __shared__ int heavy_n, local_n;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // Compact checkpoint slots.  A CANDIDATE is a tile the backward may cut into shares (by its un-raised content
  // threshold: a superset of what its own list ends up splitting) or a heavy tile (by the threshold as given); it needs
  // ceil(len / 64) slots.  Normally the band's candidates fit its budget together: then every candidate has slots, the
  // first-slot table is written by the list pass below (a second prefix sum riding in the same shuffles) and the only
  // extra work is a sum in the first round of the fit loop.  When they do not fit, a pass of its own grants the slots
  // from the END of the sequence (the positional tail first) and leaves the table for the other passes to read.
  int32_t* const slot_tab = jb.slot_budget > 0 && pb.seg_parts > 1 ? jobs + jb.tab_offset : nullptr;
  const bool track = slot_tab || (jb.need_out && !bwd && pb.seg_parts > 1);
  const int hl = pf.heavy_len > 0 ? pf.heavy_len : 0x7fffffff;
  __shared__ int need_tot[NWV];
  __shared__ int carry_need;
  bool all_granted = true;  // (uniform)
  auto candidate = [&](int idx, int len) {
    return len > 0 && (len > hl || job_count(pb, idx, n, 0, 0, 0x7fffffff, thr2_b, len) > 1);
  };
  // has this tile checkpoint slots?  (no table: the question is not asked -- the forward's flag goes by its own rule)
  auto has_slots = [&](int idx, int tile, int len) {
    return !slot_tab || (all_granted ? candidate(idx, len) : slot_tab[tile] >= 0);
  };
  // the list must fit the launch's workgroups: raise the content thresholds (x1.5 per round) until
  // it does; the positional jobs alone always fit
  for (int round = 0; round < 12; ++round) {
    int mine = 0, need = 0;
    for (int idx = threadIdx.x; idx < n; idx += NTH) {
      const int tile = tile_at(idx);
      if (tile < 0) continue;
      const int len = tile_offsets[tile + 1] - tile_offsets[tile];
      const bool slots = has_slots(idx, tile, len);
      mine += p.seg_parts > 1 && !slots ? 1 : job_count(p, idx, n, tail4, tail2, thr4, thr2, len, slots ? thr_h : 0x7fffffff);
      if (track && round == 0 && candidate(idx, len)) need += seg_count(len, jb.seg_fine);
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
      mine += __shfl_xor(mine, m);
      need += __shfl_xor(need, m);
    }
    __syncthreads();
    if (lane == 0) {
      wave_tot[wave] = mine;
      need_tot[wave] = need;
    }
    __syncthreads();
    int all = 0;
#pragma unroll
    for (int w = 0; w < NWV; ++w) all += wave_tot[w];
    if (track && round == 0 && all_granted) {
      int all_need = 0;
#pragma unroll
      for (int w = 0; w < NWV; ++w) all_need += need_tot[w];
      // (stored early: the write to pinned host memory completes while the lists are being written)
      if (!bwd && jb.need_out && threadIdx.x == 0)
        __hip_atomic_store(jb.need_out + xcd, (long long)all_need, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (slot_tab && all_need > jb.slot_budget) {
        // not enough slots for all: grants from the end of the sequence while they last
        if (threadIdx.x == 0) carry = 0;
        __syncthreads();
        for (int base = 0; base < n; base += NTH) {
          const int idx = n - 1 - (base + (int)threadIdx.x);
          int nd = 0, tile = 0;
          if (idx >= 0) {
            tile = tile_at(idx);
            if (tile >= 0) {
              const int len = tile_offsets[tile + 1] - tile_offsets[tile];
              nd = candidate(idx, len) ? seg_count(len, jb.seg_fine) : 0;
            }
          }
          int incl = nd;
#pragma unroll
          for (int k = 1; k < 64; k <<= 1) {
            const int o = __shfl_up(incl, k);
            if (lane >= k) incl += o;
          }
          if (lane == 63) wave_tot[wave] = incl;
          __syncthreads();
          int pos = carry + incl - nd;
#pragma unroll
          for (int w = 0; w < NWV; ++w)
            if (w < wave) pos += wave_tot[w];
          if (idx >= 0 && tile >= 0) slot_tab[tile] = nd > 0 && pos + nd <= jb.slot_budget ? xcd * jb.slot_budget + pos : -1;
          __syncthreads();
          if (threadIdx.x == NTH - 1) carry = pos + nd;
          __syncthreads();
        }
        all_granted = false;
        --round;  // this round again, counted by the grants (the thresholds stay)
        continue;
      }
    }
    bool fits = all <= p.max_jobs;  // uniform across the workgroup
    if (!bwd && thr_h != 0x7fffffff) {
      // (a pass of its own: the build rides in a 1024-thread launch, 128 registers)
      long long heavy = 0;  // heavy tiles | their local jobs << 32 (a band of an 8K frame holds more than 4096 tiles)
      for (int idx = threadIdx.x; idx < n; idx += NTH) {
        const int tile = tile_at(idx);
        if (tile < 0) continue;
        const int len = tile_offsets[tile + 1] - tile_offsets[tile];
        if (len > thr_h && has_slots(idx, tile, len)) heavy += 1ll + ((long long)(p.heavy_wide ? 0 : heavy_local_jobs(len)) << 32);
      }
#pragma unroll
      for (int m = 1; m < 64; m <<= 1) heavy += __shfl_xor(heavy, m);
      __syncthreads();
      if (lane == 0) heavy_tot[wave] = heavy;
      __syncthreads();
      long long all_heavy = 0;
#pragma unroll
      for (int w = 0; w < NWV; ++w) all_heavy += heavy_tot[w];
      fits = fits && 4 * (all_heavy & 0xFFFFFFFFll) <= FG_HEAVY_CAP && (all_heavy >> 32) <= FG_LOCAL_CAP;
    }
    if (fits) break;
    thr4 = thr4 > 0x50000000 ? 0x7fffffff : thr4 + (thr4 >> 1) + 1;
    thr2 = thr2 > 0x50000000 ? 0x7fffffff : thr2 + (thr2 >> 1) + 1;
    thr_h = thr_h > 0x50000000 ? 0x7fffffff : thr_h + (thr_h >> 1) + 1;
    if (round == 10) thr4 = thr2 = thr_h = 0x7fffffff;
  }
  // (the backward's list gives heavy tiles finer shares: by the forward's threshold as given -- its own list is built
  // by another workgroup, whose raised threshold this one does not see; more shares than heavy tiles need is harmless)
  if (threadIdx.x == 0) heavy_n = local_n = 0;
  if (threadIdx.x == 0) carry = carry_need = 0;
  __syncthreads();
  const bool write_tab = slot_tab && all_granted;  // (else the grant pass has written it)
  int32_t* seg = jobs + 8 + (size_t)xcd * cap;
  for (int base = 0; base < n; base += NTH) {
    const int idx = base + (int)threadIdx.x;
    int cnt = 0, tile = 0, flag = 0, need = 0;
    bool heavy_tile = false;
    if (idx < n && (tile = tile_at(idx)) >= 0) {
      const int len = tile_offsets[tile + 1] - tile_offsets[tile];
      const bool slots = has_slots(idx, tile, len);
      cnt = p.seg_parts > 1 && !slots ? 1 : job_count(p, idx, n, tail4, tail2, thr4, thr2, len, slots ? thr_h : 0x7fffffff);
      heavy_tile = !bwd && len > thr_h && slots;
      if (write_tab && slots) need = seg_count(len, jb.seg_fine);
      // forward lists: will the backward (list shares, its un-raised content threshold: a superset of
      // what its own list ends up splitting) run this tile as ONE job?  Then no checkpoints are needed.
      // (compact slots: a tile without slots is such a tile)
      if (!bwd && pb.seg_parts > 1 && (slot_tab ? !slots : job_count(pb, idx, n, 0, 0, 0x7fffffff, thr2_b, len) <= 1))
        flag = FG_JOB_NO_CKPT;
    }
    // two prefix sums in one: the jobs in front of this tile's (low word), the slots in front of its own (high word)
    long long incl = (long long)cnt | (long long)need << 32;
    const long long own = incl;
#pragma unroll
    for (int k = 1; k < 64; k <<= 1) {
      const long long o = __shfl_up(incl, k);
      if (lane >= k) incl += o;
    }
    if (lane == 63) {
      wave_tot[wave] = (int)(incl & 0xFFFFFFFFll);
      need_tot[wave] = (int)(incl >> 32);
    }
    __syncthreads();
    int pos = carry + (int)((incl - own) & 0xFFFFFFFFll), npos = carry_need + (int)((incl - own) >> 32);
#pragma unroll
    for (int w = 0; w < NWV; ++w)
      if (w < wave) {
        pos += wave_tot[w];
        npos += need_tot[w];
      }
    if (write_tab && idx < n && tile >= 0) slot_tab[tile] = need > 0 ? xcd * jb.slot_budget + npos : -1;
    if (p.seg_parts > 1) {
      for (int j = 0; j < cnt; ++j) seg[pos + j] = tile << 12 | ((j + idx) % cnt) << 6 | (cnt - 1);
    } else if (heavy_tile) {
#pragma unroll
      for (int j = 0; j < 4; ++j) seg[pos + j] = tile << 3 | (j + 1) | FG_JOB_PREFIX;  // (checkpoints: always)
    } else if (cnt == 1) seg[pos] = tile << 3 | flag;  // strip -1
    else if (cnt == 2) { seg[pos] = tile << 3 | 5 | flag; seg[pos + 1] = tile << 3 | 6 | flag; }  // strip 4, 5
    else if (cnt == 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) seg[pos + j] = tile << 3 | (j + 1) | flag;  // strip 0..3
    }
    __syncthreads();
    if (threadIdx.x == NTH - 1) {
      carry = pos + cnt;
      carry_need = npos + need;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) jobs[xcd] = carry;
  if (!bwd && p.heavy_len <= 0 && jb.tab_offset > jb.main_words && threadIdx.x == 0) {
    // the policy has heavy tiles on (the two lists are in the layout and launch_fwd_mixed reads their counts) but this
    // build runs without them (no list shares: no checkpoint buffer): empty lists, not uninitialised ones
    jobs[jb.main_words + xcd] = 0;
    jobs[jb.main_words + FG_LOCAL_WORDS + xcd] = 0;
  }
  if (!bwd && p.heavy_len > 0) {
    // the heavy tiles' local jobs and their four combine jobs each, on the lists of the two launches behind the main one
    // (a pass of its own: registers, see above; any order)
    int32_t* lc = jobs + jb.main_words;
    int32_t* hv = lc + FG_LOCAL_WORDS;
    for (int idx = threadIdx.x; idx < n; idx += NTH) {
      const int tile = tile_at(idx);
      if (tile < 0) continue;
      const int len = tile_offsets[tile + 1] - tile_offsets[tile];
      if (len > thr_h && has_slots(idx, tile, len)) {
        const int nl = p.heavy_wide ? 0 : heavy_local_jobs(len);
        const int l0 = atomicAdd(&local_n, nl), h = atomicAdd(&heavy_n, 1);
        for (int j = 0; j < nl; ++j) lc[8 + xcd * FG_LOCAL_CAP + l0 + j] = tile << 8 | j;
#pragma unroll
        for (int j = 0; j < 4; ++j) hv[8 + xcd * FG_HEAVY_CAP + 4 * h + j] = tile << 3 | (j + 1);
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      lc[xcd] = local_n;
      hv[xcd] = 4 * heavy_n;
    }
  }
}

}  // namespace fgjobs
