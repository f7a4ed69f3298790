// Shared device helpers for the gfx950 kernels (wave = 64 lanes everywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fgraster.h"

#define FG_WAVE 64

#define FG_RETURN_IF_LAUNCH_FAILED()                  \
  do {                                                \
    if (hipGetLastError() != hipSuccess) return FG_ERR_LAUNCH; \
  } while (0)

static inline hipStream_t fg_hip_stream(fg_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// blending constants (build constants; see DESIGN.md "constants")
#define FG_ALPHA_SKIP (1.0f / 255.0f)
#define FG_ALPHA_MAX 0.999f
#define FG_T_STOP 1e-4f
#define FG_FOV_CLAMP 1.3f

namespace fg {

__device__ __forceinline__ int lane_id() { return __lane_id(); }

// ---- where can a splat reach alpha >= 1/255? ------------------------------------------------
// alpha = o exp(-sigma) >= 1/255  <=>  sigma <= tau = ln(255 o); the ellipse sigma <= tau of the conic
// (a, b, c) has the axis-aligned half extents sqrt(2 tau c / det), sqrt(2 tau a / det).  The extents
// are inflated (x1.0005 + 0.02 px) so that rounding -- of the 1-ulp hardware log / rcp / sqrt used
// here and of the pixel test itself -- can only ever ADD work: whatever lies outside them would have
// failed the per-pixel alpha test anyway, so culling by them never changes a result.  ONE function
// for the two users (the strip masks of the raster kernels, the tight tile rectangles of the
// binning), so both cull by identical numbers.
//   returns 0: the splat reaches 1/255 nowhere;  1: ex / ey valid;  2: no culling possible (NaN
//   opacity -- the reference behaviour is to propagate it --, degenerate conic)
__device__ __forceinline__ int alpha_extent(float o, float a, float b, float c, float& ex, float& ey) {
  ex = ey = 0.f;
  if (!(o == o)) return 2;
  const float t255 = 255.f * o;
  if (!(t255 >= 1.f)) return 0;
  const float det = a * c - b * b;
  if (!(det > 0.f)) return 2;
  // t255 >= 1 here: the bare v_log_f32 (log2, 1 ulp) needs no denormal scaling; 2 ln(x) = 2 ln2 log2(x)
  const float tau2 = 1.3862943611f * __builtin_amdgcn_logf(t255) + 1e-4f;
  const float rdet = __builtin_amdgcn_rcpf(det);
  ex = __builtin_amdgcn_sqrtf(tau2 * c * rdet);
  ey = __builtin_amdgcn_sqrtf(tau2 * a * rdet);
  if (!(ex == ex) || !(ey == ey)) return 2;
  // (multiply, then add: two literal operands -- as one fused multiply-add the 0.02 sat in a VGPR)
  ex = __fadd_rn(__fmul_rn(ex, 1.0005f), 0.02f);
  ey = __fadd_rn(__fmul_rn(ey, 1.0005f), 0.02f);
  return 1;
}
// FOOTPRINT MASK (round 5).  The footprint rectangle is the axis-aligned box of the ellipse sigma <= ln(255 o): tight for a
// round splat, mostly empty for a needle lying diagonally (trained scenes are full of those: a rectangle of 12 x 12 tiles
// of which the ellipse touches 20).  The mask says which parts of the rectangle the ELLIPSE reaches: the rectangle's
// w x h tiles are cut into at most 8 x 8 blocks of bs x bs tiles (bs = 1 up to 8 x 8 tiles, then 2, 4, ...), bit 8 by + bx
// is set iff some pixel centre of block (bx, by) lies inside the (inflated) ellipse.  Exact for the block -- for a block row,
// the x-range of the ellipse over the row's strip of pixel centres is closed-form (the ellipse cut by a strip is convex) --
// and conservative the way alpha_extent is: tau and the ranges are inflated so that rounding only ever keeps more.  The
// binning counts and scatters the set blocks' tiles only: lists stay an order-preserving subsequence of the reference's,
// every dropped entry is one no pixel of the tile would have taken.
__host__ __device__ __forceinline__ int footprint_block(int w, int h) {
  const int m = w > h ? w : h;
  int bs = 1;
  while (8 * bs < m) bs <<= 1;
  return bs;
}
// (x0, y0, w, h: the footprint rectangle in tiles, w, h > 0; ts: the tile side in pixels)
__device__ __forceinline__ uint64_t footprint_mask(float o, float a, float b, float c, float gx, float gy, int x0, int y0,
                                                   int w, int h, float ts) {
  const int bs = footprint_block(w, h);
  const int nbx = (w + bs - 1) / bs, nby = (h + bs - 1) / bs;
  const uint32_t row_all = (1u << nbx) - 1u;
  uint64_t all = 0;
  for (int by = 0; by < nby; ++by) all |= (uint64_t)row_all << (8 * by);
  // det by a compensated difference of products: for a long diagonal needle the blur leaves det ~1e-4 against a c ~ 3, and
  // the plain a c - b b loses 1e-3 ... 1e-2 of it -- more than the 0.1 % inflation of tau2 below keeps
  const float t255 = 255.f * o, bb = b * b, det = __builtin_fmaf(a, c, -bb) + __builtin_fmaf(-b, b, bb);
  if (!(o == o) || !(det > 0.f) || !(a > 0.f) || !(c > 0.f) || !(t255 >= 1.f)) return all;  // (no culling possible: alpha_extent kinds 2 / 0)
  // tau2 = 2 ln(255 o), inflated by 0.1 % + 2e-3 (the pixel test's own rounding is ~1e-5 of it)
  // one tile wide or high: the rectangle was tightened against the ellipse's extents axis by axis, and a connected shape
  // that reaches the first and the last tile of a row of tiles crosses the ones in between
  if (w == 1 || h == 1) return all;
  // (hardware reciprocals, 1 ulp: the inflations below are thousands of ulps)
  const float tau2 = (1.3862943611f * __builtin_amdgcn_logf(t255)) * 1.001f + 2e-3f;
  const float rdet = __builtin_amdgcn_rcpf(det), ra = __builtin_amdgcn_rcpf(a);
  const float ex = __builtin_amdgcn_sqrtf(tau2 * c * rdet), ey = __builtin_amdgcn_sqrtf(tau2 * a * rdet);
  if (!(ex == ex) || !(ey == ey)) return all;
  const float dy_right = -b * ex * __builtin_amdgcn_rcpf(c);  // where the ellipse is widest to the right (to the left: at -dy_right)
  const float at = a * tau2, m = 0.02f, rspan = __builtin_amdgcn_rcpf(ts * (float)bs), org = ts * (float)x0;
  // The x-range of the ellipse over a block row's strip: x_right(dy) = (-b dy + sqrt(a tau2 - det dy^2)) / a is concave, so
  // its maximum over the strip is at the ellipse's own rightmost point when that lies inside, else at the strip edge nearer
  // to it; likewise x_left.  The strips are taken from tile boundary to tile boundary (half a pixel more than the pixel
  // centres on either side: conservative), so that a boundary's cross-section -- ONE square root -- serves the row above
  // and the row below.
  auto cross = [&](float dy, float& xl, float& xr) {  // the ellipse's x-range at height dy (relative to the centre)
    const float d = fminf(fmaxf(dy, -ey), ey);  // (beyond the top / bottom: the tangent point)
    const float root = __builtin_amdgcn_sqrtf(fmaxf(at - det * d * d, 0.f));
    xr = (-b * d + root) * ra;
    xl = (-b * d - root) * ra;
  };
  uint64_t mask = 0;
  float y_top = ts * (float)y0 - gy, tl, tr;  // (the half pixel between a tile boundary and its first pixel centres is the margin in y)
  cross(y_top, tl, tr);
  for (int by = 0; by < nby; ++by) {
    const float y_bot = ts * (float)min(y0 + (by + 1) * bs, y0 + h) - gy;
    float bl, br;
    cross(y_bot, bl, br);
    const bool reached = !(y_top > ey * 1.0005f || y_bot < -ey * 1.0005f);  // the ellipse reaches the strip at all
    float xr = fmaxf(tr, br), xl = fminf(tl, bl);
    if (dy_right >= y_top && dy_right <= y_bot) xr = ex;     // its rightmost point lies inside the strip
    if (-dy_right >= y_top && -dy_right <= y_bot) xl = -ex;  // ... its leftmost
    y_top = y_bot;
    tl = bl;
    tr = br;
    if (!reached) continue;
    xr = gx + xr + (fabsf(xr) * 0.0005f + m);
    xl = gx + xl - (fabsf(xl) * 0.0005f + m);
    // blocks whose pixel centres [ts (x0 + bx bs) + 0.5, ts (x0 + (bx + 1) bs) - 0.5] meet [xl, xr]
    const float fa = ceilf((xl + 0.5f - org) * rspan) - 1.f, fb = floorf((xr - 0.5f - org) * rspan);
    if (!(fa == fa) || !(fb == fb)) {  // (NaN: keep the whole row)
      mask |= (uint64_t)row_all << (8 * by);
      continue;
    }
    const int ba = (int)fmaxf(fa, 0.f), bb = (int)fminf(fb, (float)(nbx - 1));
    if (bb < ba) continue;
    mask |= (uint64_t)((row_all >> (nbx - 1 - (bb - ba))) << ba) << (8 * by);
  }
  return mask;
}
// does the extent [g - e, g + e] reach a pixel centre of [lo + 0.5, lo + span - 0.5]?  (the comparison
// form both users share; span = 16 for a tile side, 4 for a strip)
__device__ __forceinline__ bool extent_reaches(float g, float e, float lo, float span) {
  return !(g + e < lo + 0.5f || g - e > lo + (span - 0.5f));
}
// the same against bounds the caller precomputed: first = lo + 0.5f, last = lo + (span - 0.5f)
__device__ __forceinline__ bool extent_reaches_bounds(float g, float e, float first, float last) {
  return !(g + e < first || g - e > last);
}
// a wave-uniform float, held in a scalar register (gfx950 has no scalar float ALU: a uniform value
// computed by vector instructions otherwise occupies a VGPR for as long as it lives)
// (inline asm: the builtin is folded away when the compiler can prove the value uniform, and the
// value then stays in the vector register its arithmetic produced it in)
// The s_nops are the wait states gfx950 needs around it and hipcc does not insert around asm
// statements: 1 between a vector write of the VGPR and the v_readfirstlane that reads it (without it
// the read returned the register's OLD value: every strip mask came out empty), 2 between the SGPR
// write and a vector instruction reading that SGPR.
__device__ __forceinline__ float uniform(float v) {
  float s;
  asm("s_nop 0\n\tv_readfirstlane_b32 %0, %1\n\ts_nop 1" : "=s"(s) : "v"(v));
  return s;
}

// ---- DPP / permlane cross-lane moves (no LDS traffic) ------------------------------------
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf, bool BOUND = true>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, BANK_MASK, BOUND));
}
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ float dpp_mov_keep(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old),
                                                               __builtin_bit_cast(int, v), CTRL, ROW_MASK,
                                                               BANK_MASK, false));
}
#define FG_DPP_QUAD_XOR1 0xB1  // quad_perm [1,0,3,2]
#define FG_DPP_QUAD_XOR2 0x4E  // quad_perm [2,3,0,1]
#define FG_DPP_ROW_SHL4 0x104
#define FG_DPP_ROW_SHR4 0x114
#define FG_DPP_ROW_ROR8 0x128

// Sum of v over the 64 lanes, result in every lane.
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_mov<FG_DPP_QUAD_XOR1>(v);
  v += dpp_mov<FG_DPP_QUAD_XOR2>(v);
  v += dpp_mov<0x124>(v);  // row_ror:4
  v += dpp_mov<FG_DPP_ROW_ROR8>(v);
  // rows of 16 now hold their sum in every lane; combine the 4 rows through SGPRs
  float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return (a + b) + (c + d);
}

__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v = max(v, __shfl_xor(v, m));
  return v;
}

// a' = [a.lo32, b.lo32], b' = [a.hi32, b.hi32]; returns a' + b'.  Inline asm: the builtin's
// second result is mis-extracted by ROCm 7.2's hipcc when bit-cast to float (both operands of
// the add became the first result).  The s_nops cover the VALU->permlane-swap wait states that
// hipcc does not insert around asm statements.
__device__ __forceinline__ float swap32_add(float a, float b) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return a + b;
}
// rows of 16 lanes: a' = [a.r0, b.r0, a.r2, b.r2], b' = [a.r1, b.r1, a.r3, b.r3]
__device__ __forceinline__ float swap16_add(float a, float b) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return a + b;
}

// Transposing wave reduction: 16 per-lane values -> after the call lane l holds, in the
// return value, the 64-lane sum of value index (l >> 2).  ~37 VALU instructions and no LDS.
__device__ __forceinline__ float wave_reduce16_transposed(float (&v)[16]) {
  const int lane = lane_id();
  // step 1: lanes l <-> l^32 (v_permlane32_swap): 16 -> 8 registers, bit5 selects j / j+8
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    v[j] = swap32_add(v[j], v[j + 8]);
  }
  // step 2: lanes l <-> l^16 (v_permlane16_swap): 8 -> 4, bit4 selects j / j+4
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v[j] = swap16_add(v[j], v[j + 4]);
  }
  // step 3: lanes l <-> l^8 (DPP row_ror:8): 4 -> 2, bit3 selects j / j+2
  const bool b3 = (lane & 8) != 0;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float keep = b3 ? v[j + 2] : v[j];
    float send = b3 ? v[j] : v[j + 2];
    v[j] = keep + dpp_mov<FG_DPP_ROW_ROR8>(send);
  }
  // step 4: lanes l <-> l^4 (row_shl:4 into banks 0,2; row_shr:4 into banks 1,3): 2 -> 1
  const bool b2 = (lane & 4) != 0;
  {
    float keep = b2 ? v[1] : v[0];
    float send = b2 ? v[0] : v[1];
    float got = dpp_mov_keep<FG_DPP_ROW_SHL4, 0xf, 0x5>(0.f, send);
    got = dpp_mov_keep<FG_DPP_ROW_SHR4, 0xf, 0xA>(got, send);
    v[0] = keep + got;
  }
  // step 5: sum the quad
  float r = v[0];
  r += dpp_mov<FG_DPP_QUAD_XOR1>(r);
  r += dpp_mov<FG_DPP_QUAD_XOR2>(r);
  return r;
}

// 12-value variant (enough for <= 4 composited channels): 29 VALU ops.  After the call the lanes
// with (lane & 3) == 0 and not (bit2 && bit3) hold the 64-lane sum of value index
//   6*bit5 + 3*bit4 + (bit2 ? 2 : bit3)          (wave_reduce12_index(lane)).
__device__ __forceinline__ float wave_reduce12_transposed(float (&v)[16]) {
  const int lane = lane_id();
#pragma unroll
  for (int j = 0; j < 6; ++j) v[j] = swap32_add(v[j], v[j + 6]);  // bit5 selects j / j+6
#pragma unroll
  for (int j = 0; j < 3; ++j) v[j] = swap16_add(v[j], v[j + 3]);  // bit4 selects j / j+3
  const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0;
  // bit3 selects v0 / v1; v2 is summed over the xor-8 partner as it is
  const float keep3 = b3 ? v[1] : v[0], send3 = b3 ? v[0] : v[1];
  const float A = keep3 + dpp_mov<FG_DPP_ROW_ROR8>(send3);
  const float B = v[2] + dpp_mov<FG_DPP_ROW_ROR8>(v[2]);
  // bit2 selects A / B
  const float keep2 = b2 ? B : A, send2 = b2 ? A : B;
  float got = dpp_mov_keep<FG_DPP_ROW_SHL4, 0xf, 0x5>(0.f, send2);
  got = dpp_mov_keep<FG_DPP_ROW_SHR4, 0xf, 0xA>(got, send2);
  float r = keep2 + got;
  r += dpp_mov<FG_DPP_QUAD_XOR1>(r);
  r += dpp_mov<FG_DPP_QUAD_XOR2>(r);
  return r;
}
__device__ __forceinline__ int wave_reduce12_index(int lane) {
  return 6 * (lane >> 5) + 3 * ((lane >> 4) & 1) + ((lane & 4) ? 2 : ((lane >> 3) & 1));
}
__device__ __forceinline__ bool wave_reduce12_owner(int lane) { return (lane & 3) == 0 && (lane & 12) != 12; }

// LDS-transposing wave reduction of ROWS per-lane values (ROWS <= 16).  Measured on gfx950
// (scripts/micro/valu_rate.hip, profiles/r01_valu_issue_rates.md): a v_permlane*_swap holds a
// SIMD's vector issue for 8 clocks and every DPP operation for 4 (neither overlaps with other
// vector work), while a plain v_add_f32 costs 2 -- the register-only butterflies above spend
// ~130-170 issue clocks per reduction.  Here every lane stores its ROWS values column-wise
// (row q at buf[q * FG_RED_STRIDE + lane]: conflict-free), then lane l = 4q + part sums 16
// consecutive floats of row q with four ds_read_b128 (stride 68 floats keeps the four b128
// phases on disjoint banks) and 15 full-rate adds; two quad DPP adds finish.  The LDS pipe does
// the data movement beside the vector ALU.  After the call lane l holds the 64-lane sum of value
// (l >> 2) when (l >> 2) < ROWS; one wavefront per buffer, no barrier (a wavefront's LDS
// operations execute in order).
constexpr int FG_RED_STRIDE = 68;
template <int ROWS>
__device__ __forceinline__ float wave_reduce_rows_lds(const float (&v)[16], float* buf, int lane) {
#pragma unroll
  for (int q = 0; q < ROWS; ++q) buf[q * FG_RED_STRIDE + lane] = v[q];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  float r = 0.f;
  if ((lane >> 2) < ROWS) {
    const float4* src = reinterpret_cast<const float4*>(buf + (lane >> 2) * FG_RED_STRIDE + (lane & 3) * 16);
    const float4 a = src[0], b = src[1], c = src[2], d = src[3];
    r = (((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w))) +
        (((c.x + c.y) + (c.z + c.w)) + ((d.x + d.y) + (d.z + d.w)));
    r += dpp_mov<FG_DPP_QUAD_XOR1>(r);
    r += dpp_mov<FG_DPP_QUAD_XOR2>(r);
  }
  return r;
}

// The same with the store addresses kept in registers by the caller: wr[i] = LDS byte address of
// buf[4 * i * FG_RED_STRIDE + lane] (i < (ROWS + 3) / 4), made opaque once per job (lds_opaque) -- inside a
// per-entry loop the compiler otherwise rebuilds them with a v_add each, every entry (rows beyond the 255-dword
// reach of ds_write2_b32's offsets).
typedef __attribute__((address_space(3))) float lds_float_t;
__device__ __forceinline__ uint32_t lds_opaque(const void* p) {
  uint32_t a = (uint32_t)(uintptr_t)p;  // (the low half of a shared-memory flat address is the LDS offset)
  asm volatile("" : "+v"(a));
  return a;
}
template <int ROWS>
__device__ __forceinline__ float wave_reduce_rows_lds(const float (&v)[16], const uint32_t (&wr)[(ROWS + 3) / 4],
                                                       const float* buf, int lane) {
#pragma unroll
  for (int q = 0; q < ROWS; ++q) reinterpret_cast<lds_float_t*>(wr[q >> 2])[(q & 3) * FG_RED_STRIDE] = v[q];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  float r = 0.f;
  if ((lane >> 2) < ROWS) {
    const float4* src = reinterpret_cast<const float4*>(buf + (lane >> 2) * FG_RED_STRIDE + (lane & 3) * 16);
    const float4 a = src[0], b = src[1], c = src[2], d = src[3];
    r = (((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w))) +
        (((c.x + c.y) + (c.z + c.w)) + ((d.x + d.y) + (d.z + d.w)));
    r += dpp_mov<FG_DPP_QUAD_XOR1>(r);
    r += dpp_mov<FG_DPP_QUAD_XOR2>(r);
  }
  return r;
}

}  // namespace fg
