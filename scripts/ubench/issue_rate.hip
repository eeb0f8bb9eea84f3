// Micro-benchmark behind bench.py's `roofline.peak`: how many wave-instructions per cycle ONE SIMD of gfx950 issues, by
// instruction kind and by wavefronts resident on the SIMD (1, 2, 4, 8), with INDEPENDENT instruction streams (8 register
// chains per wavefront, so neither a dependent-issue latency nor a lone wavefront's own cadence is what is measured, only
// the port).  VERDICT r02 weak #5: the guide gives v_fma_f32 (wave64) 2 cycles on a SIMD-32 with more than one
// wavefront; round 1's micro-benchmark ran dependent integer chains at 1 and 4 wavefronts only.
//
// Method: every kernel is `iters` trips of 64 asm instructions of one kind (8 chains x 8), 256-lane blocks (one wavefront per
// SIMD per block), grid = 256 CUs x waves/SIMD blocks.  Wall time from HIP events over a launch of >= 20 ms; the shader
// clock of the same launch from s_memtime / s_memrealtime (100 MHz) of block 0.  cycles per wave-instruction per SIMD =
//   wall x clock / (iters x 64 x waves_per_simd).
// Build: hipcc --offload-arch=gfx950 -O2 -o issue_rate.bin issue_rate.hip      Output: profiles/r03_issue_rate_ubench.txt
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

// eight independent chains r0..r7; INS(d, a, b) expands to one instruction that writes chain register d
#define EIGHT(INS) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)
#define SIXTYFOUR(INS) EIGHT(INS) EIGHT(INS) EIGHT(INS) EIGHT(INS) EIGHT(INS) EIGHT(INS) EIGHT(INS) EIGHT(INS)

struct Clock { unsigned long long t0, t1, r0, r1; };

#define KERNEL(NAME, BODY, DECL, SINK)                                                                         \
    __global__ void __launch_bounds__(256) NAME(uint32_t *out, uint32_t seed, int iters, Clock *clk) {        \
        DECL                                                                                                   \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();    \
        for (int i = 0; i < iters; ++i) { BODY }                                                               \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();    \
        SINK                                                                                                   \
        if (threadIdx.x == 0 && blockIdx.x == 0) *clk = Clock{t0, t1, r0, r1};                                 \
    }

// ---- 32-bit integer / float VALU: chains are uint32_t r[8]
#define DECL_U32 uint32_t r[8]; for (int k = 0; k < 8; ++k) r[k] = (threadIdx.x + 1u) * (2654435761u + k) + seed; uint32_t c1 = seed | 1u, c2 = seed * 3u + 7u; (void)c1; (void)c2;
#define SINK_U32 uint32_t acc = 0; for (int k = 0; k < 8; ++k) acc ^= r[k]; out[blockIdx.x * 256 + threadIdx.x] = acc;

#define I_FMA(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[k]) : "v"(c1), "v"(c2));
#define I_ADD(k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[k]) : "v"(c1));
#define I_XOR(k) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[k]) : "v"(c1));
#define I_BITOP3(k) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(r[k]) : "v"(c1), "v"(c2));
#define I_MAD24(k) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r[k]) : "v"(c1), "v"(c2));
#define I_MAD16(k) asm volatile("v_mad_u32_u16 %0, %1, %0, %2 op_sel:[1,0,0,0]" : "+v"(r[k]) : "v"(c1), "v"(c2));
#define I_MULHI(k) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(r[k]) : "v"(c1));
#define I_MULLO(k) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[k]) : "v"(c1));
#define I_ANDOR(k) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r[k]) : "v"(c1), "v"(c2));
#define I_LSHLOR(k) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(r[k]) : "v"(c1));
#define I_BFE(k) asm volatile("v_bfe_u32 %0, %0, %1, 1" : "+v"(r[k]) : "v"(c1));
#define I_BCNT(k) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(r[k]) : "v"(c1));
#define I_AND(k) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r[k]) : "v"(c1));
#define I_OR(k) asm volatile("v_or_b32 %0, %0, %1" : "+v"(r[k]) : "v"(c1));
#define I_SUB(k) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(r[k]) : "v"(c1));
#define I_SUBC(k) asm volatile("v_sub_u32_e32 %0, 16, %0" : "+v"(r[k]));
#define I_LSHL(k) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(r[k]));
#define I_LSHR(k) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(r[k]) : "v"(c1));
#define I_MIN(k) asm volatile("v_min_u32 %0, %0, %1" : "+v"(r[k]) : "v"(c1));
#define I_ADDSDWA(k) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "+v"(r[k]) : "v"(c1));
#define I_DOT2(k) asm volatile("v_dot2_u32_u16 %0, %0, %1, %2" : "+v"(r[k]) : "v"(c1), "v"(c2));
#define I_DOT4(k) asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(r[k]) : "v"(c1), "v"(c2));
#define I_DOT8(k) asm volatile("v_dot8_u32_u4 %0, %0, %1, %2" : "+v"(r[k]) : "v"(c1), "v"(c2));
#define I_LSHLADD(k) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(r[k]) : "v"(c1));
#define I_ADD3(k) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r[k]) : "v"(c1), "v"(c2));
#define I_PERM(k) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r[k]) : "v"(c1), "v"(c2));
#define I_CNDVCC(k) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(r[k]) : "v"(c1) : );
#define I_SWAP16(k) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(r[k]), "+v"(r[(k + 1) & 7]));
#define I_MULU24(k) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r[k]) : "v"(c1));
#define I_CVTF(k) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(r[k]));
#define I_MAXF(k) asm volatile("v_max_f32 %0, %0, %1" : "+v"(r[k]) : "v"(c1));
#define I_MULF(k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[k]) : "v"(c1));
#define I_ADDF(k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[k]) : "v"(c1));
#define I_CVT64(k) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[k]) : "v"(r[k]));
#define I_CMPF64(k) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(d[k]), "v"(dc) : "vcc");
#define I_DSW8(k) asm volatile("ds_write_b8 %0, %1 offset:%2" : : "v"(la), "v"(r[k]), "n"(k * 80) : "memory");
#define I_DSADD64(k) asm volatile("ds_add_u64 %0, %1 offset:%2" : : "v"(la8), "v"(d[k]), "n"(k * 512) : "memory");
#define I_DSB128(k) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q4[k]) : "v"(la16), "n"(k * 1024));
#define I_MOV(k) asm volatile("v_mov_b32 %0, %1" : "+v"(r[k]) : "v"(c1));
#define I_CNDMASK(k) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(r[k]) : "v"(c1), "s"(msk));
#define I_ADDC(k) asm volatile("v_addc_co_u32 %0, vcc, %0, %0, %1" : "+v"(r[k]) : "s"(msk) : "vcc");
#define I_CMP(k) asm volatile("v_cmp_ge_u32 vcc, %0, %1" : : "v"(r[k]), "v"(c1) : "vcc");
#define I_CMPS(k) asm volatile("v_cmp_ge_u32 %0, %1, %2" : "=s"(sm[k]) : "v"(r[k]), "v"(c1));
#define I_EXP(k) asm volatile("v_exp_f32 %0, %0" : "+v"(r[k]));
#define I_PKADD(k) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d[k]) : "v"(dc));
#define I_PKFMA(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(d[k]) : "v"(dc));
#define I_MAD64(k) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(d[k]) : "v"(r[k]), "v"(c1) : "vcc");
#define I_FMA64(k) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[k]) : "v"(dc));
#define I_ADD64(k) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[k]) : "v"(dc));
#define I_MUL64(k) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[k]) : "v"(dc));
#define I_SWAP32(k) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(r[k]), "+v"(r[(k + 1) & 7]));
#define I_DPP(k) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r[k]) : "v"(c1));
#define I_READLANE(k) asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(ss[k]) : "v"(r[k]));
#define I_SADD(k) asm volatile("s_add_u32 %0, %0, %1" : "+s"(ss[k]) : "s"(seed) : "scc");
#define I_SMUL(k) asm volatile("s_mul_i32 %0, %0, %1" : "+s"(ss[k]) : "s"(seed));
#define I_SMULHI(k) asm volatile("s_mul_hi_u32 %0, %0, %1" : "+s"(ss[k]) : "s"(seed));
#define I_SXOR(k) asm volatile("s_xor_b32 %0, %0, %1" : "+s"(ss[k]) : "s"(seed) : "scc");
#define I_SBCNT(k) asm volatile("s_bcnt1_i32_b64 %0, %1" : "=s"(ss[k]) : "s"(msk) : "scc");
// the sweep's own mix: per neighbour 1 address add + 1 multiply-add, Philox (mad_u64 + bitop3), compare, popcount
#define I_MIXA(k) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(d[k]) : "v"(r[k]), "v"(c1) : "vcc"); asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(r[k]) : "v"(c1), "v"(c2));
// VALU and SALU side by side (do the two ports issue in the same cycle from different wavefronts?)
#define I_VS(k) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[k]) : "v"(c1)); asm volatile("s_xor_b32 %0, %0, %1" : "+s"(ss[k]) : "s"(seed) : "scc");
// LDS: conflict-free byte / dword / 8-byte reads at per-lane addresses; no wait inside the trip (the chain register only feeds the address)
#define I_DSU8(k) asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(r[k]) : "v"(la), "n"(k * 80));
#define I_DSB32(k) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r[k]) : "v"(la4), "n"(k * 256));
#define I_DSB64(k) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d[k]) : "v"(la8), "n"(k * 512));
#define I_DSVALU(k) asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(q[k]) : "v"(la), "n"(k * 80)); asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[k]) : "v"(c1)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[k]) : "v"(c2)); asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[k]) : "v"(c2));

#define DECL_MSK DECL_U32 unsigned long long msk = 0x5a5a5a5a12345678ull ^ seed;
#define DECL_SM DECL_U32 unsigned long long sm[8];
#define SINK_SM uint32_t acc = 0; for (int k = 0; k < 8; ++k) acc ^= r[k] ^ (uint32_t)sm[k]; out[blockIdx.x * 256 + threadIdx.x] = acc;
#define DECL_D DECL_U32 double d[8]; for (int k = 0; k < 8; ++k) d[k] = 1.0 + 1e-9 * (double)r[k]; double dc = 1.0000001; (void)dc;
#define SINK_D double acc = 0; for (int k = 0; k < 8; ++k) acc += d[k]; uint32_t a2 = 0; for (int k = 0; k < 8; ++k) a2 ^= r[k]; out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)__double_as_longlong(acc) ^ a2;
#define DECL_S DECL_U32 uint32_t ss[8]; for (int k = 0; k < 8; ++k) ss[k] = seed + k; unsigned long long msk = 0x5a5a5a5a12345678ull ^ seed; (void)msk;
#define SINK_S uint32_t acc = 0; for (int k = 0; k < 8; ++k) acc ^= r[k] ^ ss[k]; out[blockIdx.x * 256 + threadIdx.x] = acc;
#define DECL_L DECL_U32 __shared__ uint32_t sh[4096]; for (int k = threadIdx.x; k < 4096; k += 256) sh[k] = k * seed; __syncthreads(); \
    const uint32_t la16 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)sh + (threadIdx.x & 63u) * 16u; (void)la16; typedef uint32_t u4v __attribute__((ext_vector_type(4))); u4v q4[8]; (void)q4; \
    const uint32_t la = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)sh + (threadIdx.x & 63u), la4 = la * 4u - 3u * (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)sh, \
                   la8 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)sh + (threadIdx.x & 63u) * 8u; (void)la; (void)la4; (void)la8; double d[8]; uint32_t q[8]; (void)d; (void)q;
#define SINK_L asm volatile("s_waitcnt lgkmcnt(0)"); uint32_t acc = 0; for (int k = 0; k < 8; ++k) acc ^= r[k]; out[blockIdx.x * 256 + threadIdx.x] = acc;
#define WAITL asm volatile("s_waitcnt lgkmcnt(0)");

KERNEL(k_fma, SIXTYFOUR(I_FMA), DECL_U32, SINK_U32)
KERNEL(k_add, SIXTYFOUR(I_ADD), DECL_U32, SINK_U32)
KERNEL(k_xor, SIXTYFOUR(I_XOR), DECL_U32, SINK_U32)
KERNEL(k_bitop3, SIXTYFOUR(I_BITOP3), DECL_U32, SINK_U32)
KERNEL(k_mad24, SIXTYFOUR(I_MAD24), DECL_U32, SINK_U32)
KERNEL(k_mad16, SIXTYFOUR(I_MAD16), DECL_U32, SINK_U32)
KERNEL(k_mulhi, SIXTYFOUR(I_MULHI), DECL_U32, SINK_U32)
KERNEL(k_mullo, SIXTYFOUR(I_MULLO), DECL_U32, SINK_U32)
KERNEL(k_andor, SIXTYFOUR(I_ANDOR), DECL_U32, SINK_U32)
KERNEL(k_lshlor, SIXTYFOUR(I_LSHLOR), DECL_U32, SINK_U32)
KERNEL(k_bfe, SIXTYFOUR(I_BFE), DECL_U32, SINK_U32)
KERNEL(k_bcnt, SIXTYFOUR(I_BCNT), DECL_U32, SINK_U32)
KERNEL(k_mov, SIXTYFOUR(I_MOV), DECL_U32, SINK_U32)
KERNEL(k_and, SIXTYFOUR(I_AND), DECL_U32, SINK_U32)
KERNEL(k_or, SIXTYFOUR(I_OR), DECL_U32, SINK_U32)
KERNEL(k_sub, SIXTYFOUR(I_SUB), DECL_U32, SINK_U32)
KERNEL(k_subc, SIXTYFOUR(I_SUBC), DECL_U32, SINK_U32)
KERNEL(k_lshl, SIXTYFOUR(I_LSHL), DECL_U32, SINK_U32)
KERNEL(k_lshr, SIXTYFOUR(I_LSHR), DECL_U32, SINK_U32)
KERNEL(k_min, SIXTYFOUR(I_MIN), DECL_U32, SINK_U32)
KERNEL(k_addsdwa, SIXTYFOUR(I_ADDSDWA), DECL_U32, SINK_U32)
KERNEL(k_dot2, SIXTYFOUR(I_DOT2), DECL_U32, SINK_U32)
KERNEL(k_dot4, SIXTYFOUR(I_DOT4), DECL_U32, SINK_U32)
KERNEL(k_dot8, SIXTYFOUR(I_DOT8), DECL_U32, SINK_U32)
KERNEL(k_lshladd, SIXTYFOUR(I_LSHLADD), DECL_U32, SINK_U32)
KERNEL(k_add3, SIXTYFOUR(I_ADD3), DECL_U32, SINK_U32)
KERNEL(k_perm, SIXTYFOUR(I_PERM), DECL_U32, SINK_U32)
KERNEL(k_cndvcc, SIXTYFOUR(I_CNDVCC), DECL_U32, SINK_U32)
KERNEL(k_swap16, SIXTYFOUR(I_SWAP16), DECL_U32, SINK_U32)
KERNEL(k_mulu24, SIXTYFOUR(I_MULU24), DECL_U32, SINK_U32)
KERNEL(k_cvtf, SIXTYFOUR(I_CVTF), DECL_U32, SINK_U32)
KERNEL(k_maxf, SIXTYFOUR(I_MAXF), DECL_U32, SINK_U32)
KERNEL(k_mulf, SIXTYFOUR(I_MULF), DECL_U32, SINK_U32)
KERNEL(k_addf, SIXTYFOUR(I_ADDF), DECL_U32, SINK_U32)
KERNEL(k_cvt64, SIXTYFOUR(I_CVT64), DECL_D, SINK_D)
KERNEL(k_cmpf64, SIXTYFOUR(I_CMPF64), DECL_D, SINK_D)
KERNEL(k_ds_write_b8, SIXTYFOUR(I_DSW8) WAITL, DECL_L, SINK_L)
KERNEL(k_ds_add_u64, SIXTYFOUR(I_DSADD64) WAITL, DECL_L for (int k = 0; k < 8; ++k) d[k] = 1.0;, SINK_L)
KERNEL(k_ds_read_b128, SIXTYFOUR(I_DSB128) WAITL, DECL_L, SINK_L for (int k = 0; k < 8; ++k) out[blockIdx.x * 256 + threadIdx.x] ^= q4[k].x;)
KERNEL(k_cndmask, SIXTYFOUR(I_CNDMASK), DECL_MSK, SINK_U32)
KERNEL(k_addc, SIXTYFOUR(I_ADDC), DECL_MSK, SINK_U32)
KERNEL(k_cmp_vcc, SIXTYFOUR(I_CMP), DECL_U32, SINK_U32)
KERNEL(k_cmp_sgpr, SIXTYFOUR(I_CMPS), DECL_SM, SINK_SM)
KERNEL(k_exp, SIXTYFOUR(I_EXP), DECL_U32, SINK_U32)
KERNEL(k_pk_add_f32, SIXTYFOUR(I_PKADD), DECL_D, SINK_D)
KERNEL(k_pk_fma_f32, SIXTYFOUR(I_PKFMA), DECL_D, SINK_D)
KERNEL(k_mad_u64_u32, SIXTYFOUR(I_MAD64), DECL_D, SINK_D)
KERNEL(k_fma_f64, SIXTYFOUR(I_FMA64), DECL_D, SINK_D)
KERNEL(k_add_f64, SIXTYFOUR(I_ADD64), DECL_D, SINK_D)
KERNEL(k_mul_f64, SIXTYFOUR(I_MUL64), DECL_D, SINK_D)
KERNEL(k_permlane32_swap, SIXTYFOUR(I_SWAP32), DECL_U32, SINK_U32)
KERNEL(k_mov_dpp, SIXTYFOUR(I_DPP), DECL_U32, SINK_U32)
KERNEL(k_readfirstlane, SIXTYFOUR(I_READLANE), DECL_S, SINK_S)
KERNEL(k_s_add, SIXTYFOUR(I_SADD), DECL_S, SINK_S)
KERNEL(k_s_mul, SIXTYFOUR(I_SMUL), DECL_S, SINK_S)
KERNEL(k_s_mul_hi, SIXTYFOUR(I_SMULHI), DECL_S, SINK_S)
KERNEL(k_s_xor, SIXTYFOUR(I_SXOR), DECL_S, SINK_S)
KERNEL(k_s_bcnt1_b64, SIXTYFOUR(I_SBCNT), DECL_S, SINK_S)
KERNEL(k_philox_pair, SIXTYFOUR(I_MIXA), DECL_D, SINK_D)      // 128 instructions per trip
KERNEL(k_valu_salu_pair, SIXTYFOUR(I_VS), DECL_S, SINK_S)     // 64 VALU + 64 SALU per trip
KERNEL(k_ds_read_u8, SIXTYFOUR(I_DSU8) WAITL, DECL_L, SINK_L)
KERNEL(k_ds_read_b32, SIXTYFOUR(I_DSB32) WAITL, DECL_L, SINK_L)
KERNEL(k_ds_read_b64, SIXTYFOUR(I_DSB64) WAITL, DECL_L, SINK_L)
KERNEL(k_ds_u8_plus_3valu, SIXTYFOUR(I_DSVALU) WAITL, DECL_L, SINK_L) // 64 LDS + 192 VALU per trip

typedef void (*kern_t)(uint32_t *, uint32_t, int, Clock *);
struct Case { const char *name; kern_t k; int per_trip; const char *note; };

int main(int argc, char **argv) {
    const Case cases[] = {
        {"v_fma_f32", k_fma, 64, ""}, {"v_add_u32", k_add, 64, ""}, {"v_xor_b32", k_xor, 64, ""}, {"v_bitop3_b32", k_bitop3, 64, "Philox xor3"},
        {"v_mad_u32_u24", k_mad24, 64, "configuration index"}, {"v_mad_u32_u16 op_sel", k_mad16, 64, "packed records"},
        {"v_mul_hi_u32", k_mulhi, 64, ""}, {"v_mul_lo_u32", k_mullo, 64, ""}, {"v_mad_u64_u32", k_mad_u64_u32, 64, "Philox product"},
        {"v_and_or_b32", k_andor, 64, "row mask select"}, {"v_lshl_or_b32", k_lshlor, 64, ""}, {"v_bfe_u32", k_bfe, 64, ""}, {"v_bcnt_u32_b32", k_bcnt, 64, "ones of a step slot"},
        {"v_mov_b32", k_mov, 64, ""}, {"v_and_b32", k_and, 64, ""}, {"v_or_b32", k_or, 64, ""}, {"v_sub_u32", k_sub, 64, ""}, {"v_sub_u32 (16 - x)", k_subc, 64, "zeros of a step slot"},
        {"v_lshlrev_b32", k_lshl, 64, ""}, {"v_lshrrev_b32", k_lshr, 64, ""}, {"v_min_u32", k_min, 64, ""}, {"v_add_u32_sdwa WORD_0", k_addsdwa, 64, "state row address"},
        {"v_dot2_u32_u16", k_dot2, 64, "2 neighbours per instruction"}, {"v_dot4_u32_u8", k_dot4, 64, "4 neighbours per instruction"}, {"v_dot8_u32_u4", k_dot8, 64, ""},
        {"v_lshl_add_u32", k_lshladd, 64, ""}, {"v_add3_u32", k_add3, 64, ""}, {"v_perm_b32", k_perm, 64, ""}, {"v_cndmask_b32 (vcc, VOP2)", k_cndvcc, 64, ""},
        {"v_permlane16_swap", k_swap16, 64, "draw transpose"}, {"v_mul_u32_u24", k_mulu24, 64, ""},
        {"v_cvt_f32_u32", k_cvtf, 64, ""}, {"v_max_f32", k_maxf, 64, ""}, {"v_mul_f32", k_mulf, 64, ""}, {"v_add_f32", k_addf, 64, ""}, {"v_cvt_f64_f32", k_cvt64, 64, "cond.h"}, {"v_cmp_lt_f64 -> vcc", k_cmpf64, 64, "cond.h"},
        {"ds_write_b8", k_ds_write_b8, 64, "LDS state commit"}, {"ds_add_u64 (no return)", k_ds_add_u64, 64, "LDS histogram"}, {"ds_read_b128", k_ds_read_b128, 64, "LDS records"}, {"v_cndmask_b32 (sgpr mask)", k_cndmask, 64, ""}, {"v_addc_co_u32 (sgpr carry-in)", k_addc, 64, ""},
        {"v_cmp_ge_u32 -> vcc", k_cmp_vcc, 64, "threshold compare"}, {"v_cmp_ge_u32 -> sgpr pair", k_cmp_sgpr, 64, ""},
        {"v_exp_f32", k_exp, 64, "transcendental"}, {"v_pk_add_f32", k_pk_add_f32, 64, "k_sweep_pw4 row sums"}, {"v_pk_fma_f32", k_pk_fma_f32, 64, ""},
        {"v_fma_f64", k_fma_f64, 64, "cond.h"}, {"v_add_f64", k_add_f64, 64, "cond.h"}, {"v_mul_f64", k_mul_f64, 64, "cond.h"},
        {"v_permlane32_swap", k_permlane32_swap, 64, "draw transpose"}, {"v_mov_b32 dpp quad_perm", k_mov_dpp, 64, ""},
        {"v_readfirstlane_b32", k_readfirstlane, 64, ""},
        {"s_add_u32", k_s_add, 64, "SALU"}, {"s_xor_b32", k_s_xor, 64, "SALU"}, {"s_mul_i32", k_s_mul, 64, "SALU"}, {"s_mul_hi_u32", k_s_mul_hi, 64, "SALU"}, {"s_bcnt1_i32_b64", k_s_bcnt1_b64, 64, "SALU"},
        {"v_mad_u64_u32 + v_bitop3 (Philox mix)", k_philox_pair, 128, ""}, {"v_xor_b32 + s_xor_b32 interleaved", k_valu_salu_pair, 128, "VALU + SALU ports together"},
        {"ds_read_u8 (conflict-free)", k_ds_read_u8, 64, "LDS, wait once per 64"}, {"ds_read_b32 (conflict-free)", k_ds_read_b32, 64, "LDS"}, {"ds_read_b64", k_ds_read_b64, 64, "LDS"},
        {"ds_read_u8 + 3 VALU interleaved", k_ds_u8_plus_3valu, 256, "LDS + VALU ports together"},
    };
    const char *only = argc > 1 ? argv[1] : nullptr;
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    std::printf("# device %s, %d CUs, nominal clock %.0f MHz\n", prop.gcnArchName, cus, prop.clockRate / 1000.0);
    std::printf("# cycles per wave-instruction per SIMD (independent streams); < 4 means a SIMD issues more than one per 4 cycles\n");
    std::printf("%-40s %8s | %9s %9s %9s %9s | %s\n", "instruction", "clk GHz", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD", "8 w/SIMD", "note");
    uint32_t *out; Clock *clk;
    CHK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    CHK(hipMalloc(&clk, sizeof(Clock)));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (const Case &c : cases) {
        if (only && !std::strstr(c.name, only)) continue;
        double cyc[4] = {0, 0, 0, 0}, ghz = 0;
        int wi = 0;
        for (int w : {1, 2, 4, 8}) {
            const int blocks = cus * w;
            int iters = 2000;
            c.k<<<blocks, 256>>>(out, 1, 200, clk); // warm-up
            CHK(hipDeviceSynchronize());
            float ms = 0;
            for (int tries = 0; tries < 6; ++tries) { // grow until the launch takes >= 20 ms
                CHK(hipEventRecord(e0));
                c.k<<<blocks, 256>>>(out, 2, iters, clk);
                CHK(hipEventRecord(e1));
                CHK(hipDeviceSynchronize());
                CHK(hipEventElapsedTime(&ms, e0, e1));
                if (ms >= 20.f) break;
                iters *= 4;
            }
            Clock h;
            CHK(hipMemcpy(&h, clk, sizeof h, hipMemcpyDeviceToHost));
            const double g = (double)(h.t1 - h.t0) / ((double)(h.r1 - h.r0) * 10.0); // s_memrealtime ticks at 100 MHz -> GHz
            ghz = g;
            cyc[wi++] = ms * 1e-3 * g * 1e9 / ((double)iters * c.per_trip * w);
        }
        std::printf("%-40s %8.3f | %9.2f %9.2f %9.2f %9.2f | %s\n", c.name, ghz, cyc[0], cyc[1], cyc[2], cyc[3], c.note);
        std::fflush(stdout);
    }
    return 0;
}
