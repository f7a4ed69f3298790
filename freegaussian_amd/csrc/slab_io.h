// Coalesced staging of row slabs through LDS, shared by preprocess.hip and sh.hip.
#pragma once
#include "sh_math.h"

// Slabs are read once / written once per launch: non-temporal accesses keep them from evicting
// the record arrays the raster kernels re-read from L2 (FG_SLAB_NT=0 for the A/B).
#ifndef FG_SLAB_NT
#define FG_SLAB_NT 1
#endif
#if FG_SLAB_NT
typedef float fg_f4_native __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 fg_slab_load_nt(const float4* p) {
  const fg_f4_native v = __builtin_nontemporal_load(reinterpret_cast<const fg_f4_native*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void fg_slab_store_nt(float4* p, float4 v) {
  fg_f4_native n = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(n, reinterpret_cast<fg_f4_native*>(p));
}
#define FG_SLAB_LOAD(p) fg_slab_load_nt(p)
#define FG_SLAB_STORE(p, v) fg_slab_store_nt((p), (v))
#else
#define FG_SLAB_LOAD(p) (*(p))
#define FG_SLAB_STORE(p, v) (*(p) = (v))
#endif

namespace fgsh {

// Coalesced copy of a contiguous [nrows x row_floats] slab into padded LDS rows (stride STRIDE) at
// column lds_col0, keeping the first use_floats columns of each row.  16-byte loads whenever the
// slab is a whole number of float4, for ANY row length: all loads of a lane are issued first, the
// (row, column) of an element comes from a multiply-shift division (exact for rows <= 48 floats
// and slabs <= 2^14 elements), row wrap inside a float4 is branch-free and dropped columns land
// in the row's last pad slot (column STRIDE-1), so there is no divergent code between load and store.
constexpr int slab_max_q(int maxrows) { return (maxrows * 48 / 4 + BLOCK - 1) / BLOCK; }  // float4 per lane of a slab
constexpr int SLAB_MAX_Q = slab_max_q(BLOCK);

// row_live (LDS, one byte per row, nullable): rows whose flag is 0 are not fetched (their LDS rows
// keep whatever they held; nobody reads them) -- culled Gaussians' 192-byte coefficient rows are a
// sixth of the forward's traffic on the 1M / 1080p scene.
// MAXROWS: the most rows a call stages (a workgroup's 64, or 32 when it stages its rows in two halves).
template <int STRIDE = ROW, int MAXROWS = BLOCK>
__device__ __forceinline__ void slab_to_lds_at(float* lds, int lds_col0, const float* __restrict__ src, int nrows,
                                               int row_floats, int use_floats, const uint8_t* row_live = nullptr) {
  constexpr int SLAB_MAX_Q = slab_max_q(MAXROWS);
  const int total = nrows * row_floats;
  if ((row_floats & 3) == 0 && (use_floats & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
    // rows of whole float4s (the [16 x 3] coefficient row, the 16-float record): a float4 never straddles rows --
    // (row, float4-in-row) from one multiply-shift, four stores at consecutive columns; ~10 instead of ~26 vector
    // instructions per float4 of the generic path below
    const float4* src4 = reinterpret_cast<const float4*>(src);
    const int rq = row_floats >> 2, uq = use_floats >> 2, total4 = nrows * rq;
    const unsigned magic = ((1u << 20) + rq - 1) / rq;
    float4 v[SLAB_MAX_Q];
    bool fetch[SLAB_MAX_Q];
#pragma unroll
    for (int t = 0; t < SLAB_MAX_Q; ++t) {
      const int q = threadIdx.x + t * BLOCK;
      const int r = (int)(((unsigned)q * magic) >> 20);
      fetch[t] = q < total4 && q - r * rq < uq && (!row_live || row_live[r < nrows ? r : 0]);
      if (fetch[t]) v[t] = FG_SLAB_LOAD(&src4[q]);
    }
#pragma unroll
    for (int t = 0; t < SLAB_MAX_Q; ++t) {
      if (fetch[t]) {
        const int q = threadIdx.x + t * BLOCK;
        const int r = (int)(((unsigned)q * magic) >> 20);
        float* d = lds + r * STRIDE + lds_col0 + 4 * (q - r * rq);
        d[0] = v[t].x; d[1] = v[t].y; d[2] = v[t].z; d[3] = v[t].w;
      }
    }
  } else if ((total & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
    const float4* src4 = reinterpret_cast<const float4*>(src);
    const int total4 = total / 4;
    const unsigned magic = ((1u << 20) + row_floats - 1) / row_floats;
    const bool partial = use_floats < row_floats;  // (wave-uniform)
    float4 v[SLAB_MAX_Q];
    bool fetch[SLAB_MAX_Q];
#pragma unroll
    for (int t = 0; t < SLAB_MAX_Q; ++t) {
      const int q = threadIdx.x + t * BLOCK;
      fetch[t] = q < total4;
      if ((row_live || partial) && fetch[t]) {  // a float4 touches at most two rows (rows are >= 3 floats)
        const int e = 4 * q;
        const int r = (int)(((unsigned)e * magic) >> 20);
        const int r2 = (int)(((unsigned)(e + 3) * magic) >> 20);
        // columns in use: the float4 starts inside them, or runs over into the next row's first columns
        // (SH degree < 3 -- the reference's first 3000 steps -- uses 12 / 48 / 108 of a row's 192 bytes: the
        // rest of the row is not fetched)
        if (partial) fetch[t] = (e - r * row_floats < use_floats) | (r2 != r);
        if (row_live) fetch[t] = fetch[t] & (row_live[r] | row_live[r2 < nrows ? r2 : r]);
      }
      if (fetch[t]) v[t] = FG_SLAB_LOAD(&src4[q]);
    }
#pragma unroll
    for (int t = 0; t < SLAB_MAX_Q; ++t) {
      const int q = threadIdx.x + t * BLOCK;
      if (fetch[t]) {
        const int e = 4 * q;
        const int r = (int)(((unsigned)e * magic) >> 20), c = e - r * row_floats;
        const float vv[4] = {v[t].x, v[t].y, v[t].z, v[t].w};
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const bool wrap = c + jj >= row_floats;
          const int cc = wrap ? c + jj - row_floats : c + jj;
          const int row_base = (wrap ? r + 1 : r) * STRIDE;
          lds[row_base + (cc < use_floats ? lds_col0 + cc : STRIDE - 1)] = vv[jj];
        }
      }
    }
  } else {
    const int tot = nrows * use_floats;
    for (int e = threadIdx.x; e < tot; e += BLOCK) {
      const int r = e / use_floats, c = e - r * use_floats;
      if (!row_live || row_live[r]) lds[r * STRIDE + lds_col0 + c] = src[(size_t)r * row_floats + c];
    }
  }
}

// The inverse: padded LDS rows (from column lds_col0) out to a contiguous [nrows x row_floats]
// slab; columns >= lds_cols are written as zero.
// NT: the slab is not read again on the GPU soon (gradient outputs) -> non-temporal stores; the
// splat records, which the raster kernels gather right afterwards, keep the default policy.
template <int STRIDE = ROW, bool NT = true, int MAXROWS = BLOCK>
__device__ __forceinline__ void lds_to_slab_at(float* __restrict__ dst, const float* lds, int lds_col0, int nrows,
                                               int row_floats, int lds_cols) {
  constexpr int SLAB_MAX_Q = slab_max_q(MAXROWS);
  const int total = nrows * row_floats;
  if ((row_floats & 3) == 0 && (lds_cols & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
    // rows of whole float4s: see slab_to_lds_at
    float4* dst4 = reinterpret_cast<float4*>(dst);
    const int rq = row_floats >> 2, lq = lds_cols >> 2, total4 = nrows * rq;
    const unsigned magic = ((1u << 20) + rq - 1) / rq;
#pragma unroll
    for (int t = 0; t < SLAB_MAX_Q; ++t) {
      const int q = threadIdx.x + t * BLOCK;
      if (q < total4) {
        const int r = (int)(((unsigned)q * magic) >> 20), cq = q - r * rq;
        const float* s = lds + r * STRIDE + (cq < lq ? lds_col0 + 4 * cq : 0);
        float4 o = make_float4(s[0], s[1], s[2], s[3]);
        if (cq >= lq) o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (NT) FG_SLAB_STORE(&dst4[q], o);
        else dst4[q] = o;
      }
    }
  } else if ((total & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
    float4* dst4 = reinterpret_cast<float4*>(dst);
    const int total4 = total / 4;
    const unsigned magic = ((1u << 20) + row_floats - 1) / row_floats;
#pragma unroll
    for (int t = 0; t < SLAB_MAX_Q; ++t) {
      const int q = threadIdx.x + t * BLOCK;
      if (q < total4) {
        const int e = 4 * q;
        const int r = (int)(((unsigned)e * magic) >> 20), c = e - r * row_floats;
        float vv[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const bool wrap = c + jj >= row_floats;
          const int cc = wrap ? c + jj - row_floats : c + jj;
          const int row_base = (wrap ? r + 1 : r) * STRIDE;
          const float x = lds[row_base + (cc < lds_cols ? lds_col0 + cc : STRIDE - 1)];
          vv[jj] = cc < lds_cols ? x : 0.f;
        }
        if (NT) FG_SLAB_STORE(&dst4[q], make_float4(vv[0], vv[1], vv[2], vv[3]));
        else dst4[q] = make_float4(vv[0], vv[1], vv[2], vv[3]);
      }
    }
  } else {
    for (int e = threadIdx.x; e < total; e += BLOCK) {
      const int r = e / row_floats, c = e - r * row_floats;
      dst[e] = (c < lds_cols) ? lds[r * STRIDE + lds_col0 + c] : 0.f;
    }
  }
}

}  // namespace fgsh
