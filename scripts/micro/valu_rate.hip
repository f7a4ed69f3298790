// Micro-benchmark: vector-ALU issue rate of one SIMD as a function of the number of resident
// wavefronts, per instruction kind.  Answers: how many clocks does a wave64 instruction cost
// when 1 / 2 / 4 / 8 wavefronts share a SIMD, and which instructions run at the full fp32 rate?
//   hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int ITERS = 4096;
constexpr int UNROLL = 16;  // instructions per loop body

#define BODY16(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7) I(8) I(9) I(10) I(11) I(12) I(13) I(14) I(15)

// Every kernel: 16 independent chains a[0..15] (or one dependent chain), operands b, c (VGPR),
// s (SGPR float), m (SGPR pair lane mask), p[] (64-bit VGPR pairs).
#define KERNEL(NAME, STMT)                                                              \
  __global__ void __launch_bounds__(256) NAME(float* out, float seed, float sarg) {     \
    float a[16];                                                                        \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) a[i] = seed + (float)(threadIdx.x + i); \
    float b = seed * 0.5f, c = seed * 0.25f;                                            \
    const float s = __builtin_amdgcn_readfirstlane(sarg);                               \
    unsigned long long m = __ballot(threadIdx.x & 1);                                   \
    double p[8];                                                                        \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) p[i] = (double)seed + i;              \
    for (int it = 0; it < ITERS; ++it) { BODY16(STMT) }                                 \
    float r = 0.f;                                                                      \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) r += a[i];                           \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) r += (float)p[i];                     \
    if (r == 12345.678f) out[threadIdx.x] = r + b + c + s + (float)m;                   \
  }

#define S_FMA(k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b), "v"(c));
#define S_FMA_DEP(k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[0]) : "v"(b), "v"(c));
#define S_FMAC(k) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define S_FMA_SGPR(k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[k]) : "s"(s), "v"(c));
#define S_FMA_NEG(k) asm volatile("v_fma_f32 %0, -%1, |%2|, %0" : "+v"(a[k]) : "v"(b), "v"(c));
#define S_ADD(k) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[k]) : "v"(b));
#define S_ADD_SGPR(k) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[k]) : "s"(s));
#define S_SUB(k) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[k]) : "v"(b));
#define S_MUL(k) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[k]) : "v"(b));
#define S_MUL_E64(k) asm volatile("v_mul_f32_e64 %0, %1, %0" : "+v"(a[k]) : "v"(b));
#define S_MAX(k) asm volatile("v_max_f32 %0, %1, %0" : "+v"(a[k]) : "v"(b));
#define S_MIN(k) asm volatile("v_min_f32 %0, %1, %0" : "+v"(a[k]) : "v"(b));
#define S_MOV(k) asm volatile("v_mov_b32 %0, %1" : "=v"(a[k]) : "v"(b));
#define S_MOV64(k) asm volatile("v_mov_b64 %0, %1" : "=v"(p[k & 7]) : "v"(p[(k + 1) & 7]));
#define S_AND(k) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[k]) : "v"(b));
#define S_ADDU(k) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[k]) : "v"(b));
#define S_LSHL(k) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[k]));
#define S_CMP_VCC(k) asm volatile("v_cmp_le_f32 vcc, %0, %1" : : "v"(a[k]), "v"(b) : "vcc");
#define S_CMP_SGPR(k) asm volatile("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(m) : "v"(a[k]), "v"(b));
#define S_CND_VCC(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b) : "vcc");
#define S_CND_SGPR(k) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "s"(m));
#define S_EXP(k) asm volatile("v_exp_f32 %0, %0" : "+v"(a[k]));
#define S_RCP(k) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]));
#define S_PKFMA(k) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(p[k & 7]) : "v"(p[(k + 1) & 7]));
#define S_PKMUL(k) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[k & 7]) : "v"(p[(k + 1) & 7]));
#define S_PKADD(k) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[k & 7]) : "v"(p[(k + 1) & 7]));
#define S_DPP_MOV(k) asm volatile("v_mov_b32_dpp %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(a[k]));
#define S_DPP_ADD(k) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(a[k]));
#define S_DPP_QUAD(k) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[k]));
#define S_SWAP32(k) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[k]), "+v"(a[(k + 8) & 15]));
#define S_SWAP16(k) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[k]), "+v"(a[(k + 8) & 15]));
#define S_MULLO(k) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define S_MAD24(k) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b), "v"(c));
#define S_CVT(k) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[k]));
#define S_MIX(k) if ((k & 3) == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(a[k])); else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b), "v"(c));
#define S_MIX2(k) if (k & 1) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "s"(m)); else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b), "v"(c));
#define S_READFL(k) { int t_; asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(t_) : "v"(a[k])); }
#define S_MAX3(k) asm volatile("v_max3_f32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b), "v"(c));
#define S_FMAAK(k) asm volatile("v_fmaak_f32 %0, %1, %0, 0x3fb8aa3b" : "+v"(a[k]) : "v"(b));
#define S_MUL_LIT(k) asm volatile("v_mul_f32 %0, 0x3fb8aa3b, %0" : "+v"(a[k]));
#define S_MUL_INL(k) asm volatile("v_mul_f32 %0, 0.5, %0" : "+v"(a[k]));
#define S_MIX_MAX(k) if (k & 1) asm volatile("v_max_f32 %0, %1, %0" : "+v"(a[k]) : "v"(b)); else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b), "v"(c));
#define S_MIX_DPP(k) if (k & 1) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(a[k])); else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b), "v"(c));
#define S_MIX_CMP(k) if (k & 1) asm volatile("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(m) : "v"(a[k]), "v"(b)); else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b), "v"(c));
#define S_MIX_CND_DPP(k) if (k & 1) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(a[k])); else asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "s"(m));
#define S_MIX_EXP1(k) if (k & 1) asm volatile("v_exp_f32 %0, %0" : "+v"(a[k])); else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b), "v"(c));
#define S_BPERM(k) asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(8)" : "+v"(a[k]) : "v"(b));
#define S_SWIZZLE(k) asm volatile("ds_swizzle_b32 %0, %0 offset:0x041F\n\ts_waitcnt lgkmcnt(8)" : "+v"(a[k]));
#define S_MIX_BPERM(k) if (k & 1) asm volatile("ds_swizzle_b32 %0, %0 offset:0x041F\n\ts_waitcnt lgkmcnt(8)" : "+v"(a[k])); else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b), "v"(c));
#define S_OR(k) asm volatile("v_or_b32 %0, %1, %0" : "+v"(a[k]) : "v"(b));
#define S_SUBU(k) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(a[k]) : "v"(b));
#define S_ASHR(k) asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(a[k]));
#define S_BFE(k) asm volatile("v_bfe_u32 %0, %0, 1, 8" : "+v"(a[k]));
#define S_MED3(k) asm volatile("v_med3_f32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b), "v"(c));
#define S_SALU(k) { unsigned t_ = (unsigned)m; asm volatile("s_add_u32 %0, %0, 1" : "+s"(t_)); m = t_; }

KERNEL(k_fma, S_FMA) KERNEL(k_fma_dep, S_FMA_DEP) KERNEL(k_fmac, S_FMAC) KERNEL(k_fma_sgpr, S_FMA_SGPR)
KERNEL(k_fma_neg, S_FMA_NEG) KERNEL(k_add, S_ADD) KERNEL(k_add_sgpr, S_ADD_SGPR) KERNEL(k_sub, S_SUB)
KERNEL(k_mul, S_MUL) KERNEL(k_mul_e64, S_MUL_E64) KERNEL(k_max, S_MAX) KERNEL(k_min, S_MIN) KERNEL(k_mov, S_MOV)
KERNEL(k_mov64, S_MOV64) KERNEL(k_and, S_AND) KERNEL(k_addu, S_ADDU) KERNEL(k_lshl, S_LSHL)
KERNEL(k_cmp_vcc, S_CMP_VCC) KERNEL(k_cmp_sgpr, S_CMP_SGPR) KERNEL(k_cnd_vcc, S_CND_VCC) KERNEL(k_cnd_sgpr, S_CND_SGPR)
KERNEL(k_exp, S_EXP) KERNEL(k_rcp, S_RCP) KERNEL(k_pkfma, S_PKFMA) KERNEL(k_pkmul, S_PKMUL) KERNEL(k_pkadd, S_PKADD)
KERNEL(k_dpp_mov, S_DPP_MOV) KERNEL(k_dpp_add, S_DPP_ADD) KERNEL(k_dpp_quad, S_DPP_QUAD)
KERNEL(k_swap32, S_SWAP32) KERNEL(k_swap16, S_SWAP16) KERNEL(k_mullo, S_MULLO) KERNEL(k_mad24, S_MAD24)
KERNEL(k_cvt, S_CVT) KERNEL(k_mix, S_MIX) KERNEL(k_mix2, S_MIX2) KERNEL(k_readfl, S_READFL) KERNEL(k_max3, S_MAX3)
KERNEL(k_fmaak, S_FMAAK) KERNEL(k_mul_lit, S_MUL_LIT) KERNEL(k_mul_inl, S_MUL_INL)
KERNEL(k_mix_max, S_MIX_MAX) KERNEL(k_mix_dpp, S_MIX_DPP) KERNEL(k_mix_cmp, S_MIX_CMP) KERNEL(k_mix_cnd_dpp, S_MIX_CND_DPP) KERNEL(k_mix_exp1, S_MIX_EXP1)
KERNEL(k_bperm, S_BPERM) KERNEL(k_swizzle, S_SWIZZLE) KERNEL(k_mix_bperm, S_MIX_BPERM) KERNEL(k_or, S_OR) KERNEL(k_subu, S_SUBU) KERNEL(k_ashr, S_ASHR) KERNEL(k_bfe, S_BFE) KERNEL(k_med3, S_MED3)

typedef void (*kern_t)(float*, float, float);

void run(const char* name, kern_t k, float* d_out, int n_cu) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  printf("%-30s", name);
  for (int waves_per_simd : {1, 2, 4, 8}) {
    // 256-thread workgroups = 4 wavefronts = one per SIMD of a CU; waves_per_simd workgroups per CU
    const int grid = n_cu * waves_per_simd;
    float best = 1e9f;
    for (int r = 0; r < 4; ++r) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d_out, 1.0f, 2.0f);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    const double instr_per_simd = (double)waves_per_simd * ITERS * UNROLL;
    const double ns_per_instr = best * 1e6 / instr_per_simd;
    printf("  %dw %6.3f ns", waves_per_simd, ns_per_instr);
  }
  printf("\n");
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const int n_cu = prop.multiProcessorCount;
  printf("%s: %d CUs, clock %d kHz\n", prop.name, n_cu, prop.clockRate);
  float* d_out;
  (void)hipMalloc(&d_out, 4096);
  printf("ns per wave64 instruction per SIMD (all SIMDs busy), by wavefronts resident per SIMD; 2 clk @2.4 GHz = 0.833 ns\n");
  if (argc > 1) {  // counter-calibration mode for rocprofv3 --pmc: one launch each at 8 wavefronts per SIMD
#define ONE(k) hipLaunchKernelGGL(k, dim3(n_cu * 8), dim3(256), 0, 0, d_out, 1.0f, 2.0f); (void)hipDeviceSynchronize();
    ONE(k_fma) ONE(k_max) ONE(k_exp) ONE(k_swap32) ONE(k_mix_max) ONE(k_cnd_sgpr) ONE(k_dpp_add) ONE(k_fma_dep)
    return 0;
  }
#define RUN(k) run(#k, k, d_out, n_cu);
  RUN(k_fma) RUN(k_fma_dep) RUN(k_fmac) RUN(k_fma_sgpr) RUN(k_fma_neg) RUN(k_fmaak) RUN(k_add) RUN(k_add_sgpr) RUN(k_sub)
  RUN(k_mul) RUN(k_mul_e64) RUN(k_mul_lit) RUN(k_mul_inl) RUN(k_max) RUN(k_min) RUN(k_max3) RUN(k_mov) RUN(k_mov64) RUN(k_and) RUN(k_addu) RUN(k_lshl)
  RUN(k_cnd_vcc) RUN(k_cnd_sgpr) RUN(k_exp) RUN(k_rcp) RUN(k_pkfma) RUN(k_pkmul) RUN(k_pkadd)
  RUN(k_dpp_mov) RUN(k_dpp_add) RUN(k_dpp_quad) RUN(k_mullo) RUN(k_mad24) RUN(k_cvt) RUN(k_mix) RUN(k_mix2)
  RUN(k_cmp_vcc) RUN(k_cmp_sgpr) RUN(k_readfl) RUN(k_swap32) RUN(k_swap16)
  RUN(k_mix_max) RUN(k_mix_dpp) RUN(k_mix_cmp) RUN(k_mix_cnd_dpp) RUN(k_mix_exp1) RUN(k_or) RUN(k_subu) RUN(k_ashr) RUN(k_bfe) RUN(k_med3)
  RUN(k_bperm) RUN(k_swizzle) RUN(k_mix_bperm)
  return 0;
}
