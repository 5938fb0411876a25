// valu_rate.hip -- micro-benchmark: issue rate of the integer VALU instructions the
// evaluator kernels are made of, on gfx950.  Prints cycles per wave-instruction
// per SIMD with every CU busy (8 independent chains per wave, 8 waves per SIMD).
//   hipcc --offload-arch=gfx950 -O2 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32;

#define CHAINS8(OP)            \
  OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)

#define KERNEL(NAME, OPMACRO)                                                         \
  __global__ void __launch_bounds__(256) NAME(u32 *out, int iters, u32 seed)          \
  {                                                                                   \
    u32 t = threadIdx.x + blockIdx.x * blockDim.x;                                    \
    u32 a0 = t * 3 + seed, a1 = t * 5 + 1, a2 = t * 7 + 2, a3 = t * 11 + 3;            \
    u32 a4 = t * 13 + 4, a5 = t * 17 + 5, a6 = t * 19 + 6, a7 = t * 23 + 7;            \
    u32 b = t ^ 0x9e3779b9u, c = seed * 0x01010101u + 0x04050607u;                     \
    for (int i = 0; i < iters; i++) {                                                 \
      CHAINS8(OPMACRO) CHAINS8(OPMACRO) CHAINS8(OPMACRO) CHAINS8(OPMACRO)             \
    }                                                                                 \
    out[t] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ b ^ c;                            \
  }

#define OP_XOR(x) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_ADD(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_SUB(x) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_LSHR(x) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(x));
#define OP_ANDOR(x) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define OP_OR3(x) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define OP_BITOP3(x) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xb4" : "+v"(x) : "v"(b), "v"(c));
#define OP_BFI(x) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(x) : "v"(b), "v"(c));
#define OP_PERM(x) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define OP_ALIGNBYTE(x) asm volatile("v_alignbyte_b32 %0, %0, %1, 3" : "+v"(x) : "v"(b));
#define OP_SAD(x) asm volatile("v_sad_u8 %0, %1, 0, %0" : "+v"(x) : "v"(b));
#define OP_DOT4(x) asm volatile("v_dot4_u32_u8 %0, %1, %1, %0" : "+v"(x) : "v"(b));
#define OP_PKSUB(x) asm volatile("v_pk_sub_u16 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_PKADD(x) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_LSHLADD(x) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(x) : "v"(b));
#define OP_MIN(x) asm volatile("v_min_u32 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_FFBH(x) asm volatile("v_ffbh_u32 %0, %0" : "+v"(x));
#define OP_BCNT(x) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(x) : "v"(b));
#define OP_MULLO(x) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_MUL24(x) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_MAD24(x) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define OP_CNDMASK(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(b));
#define OP_CMP(x) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(b) : "vcc");
#define OP_MOVDPP(x) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x));
#define OP_ORDPP(x) asm volatile("v_or_b32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(x));
#define OP_SDWA(x) asm volatile("v_or_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(x) : "v"(b));

#define OP_XNOR(x) asm volatile("v_xnor_b32 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_BFE(x) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(x));
#define OP_MSAD(x) asm volatile("v_msad_u8 %0, %1, %2, %0" : "+v"(x) : "v"(b), "v"(c));
#define OP_SADHI(x) asm volatile("v_sad_hi_u8 %0, %1, %2, %0" : "+v"(x) : "v"(b), "v"(c));
#define OP_SUBSDWA(x) asm volatile("v_sub_u32_sdwa %0, %0, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_0" : "+v"(x) : "v"(b));

#define OP_AND(x) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_OR(x) asm volatile("v_or_b32 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_LSHL(x) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(x));
#define OP_MOV(x) asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(b));
#define OP_ANDLIT(x) asm volatile("v_and_b32 %0, 0x7f7f7f7f, %0" : "+v"(x));
#define OP_XORLIT(x) asm volatile("v_xor_b32 %0, 0x80808081, %0" : "+v"(x));
#define OP_ANDSGPR(x) asm volatile("v_and_b32 %0, %1, %0" : "+v"(x) : "s"(seed));
#define OP_BITOP3LIT(x) asm volatile("v_bitop3_b32 %0, %0, %1, 0x80808080 bitop3:0xb4" : "+v"(x) : "v"(b));
#define OP_BITOP3SGPR(x) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xb4" : "+v"(x) : "v"(b), "s"(seed));
#define OP_CND2(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(b) : );
#define OP_CMPONLY(x) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_xor_b32 %0, %0, %1" : "+v"(x) : "v"(b) : "vcc");
#define OP_CMPS(x) asm volatile("v_cmp_lt_u32 s[20:21], %0, %1\n\tv_xor_b32 %0, %0, %1" : "+v"(x) : "v"(b) : "s20", "s21");
#define OP_SUBLIT(x) asm volatile("v_subrev_u32 %0, 0x01010101, %0" : "+v"(x));
#define OP_ADD3(x) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define OP_LSHLOR(x) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(x) : "v"(b));
KERNEL(k_and, OP_AND)
KERNEL(k_or, OP_OR)
KERNEL(k_lshl, OP_LSHL)
KERNEL(k_mov, OP_MOV)
KERNEL(k_andlit, OP_ANDLIT)
KERNEL(k_xorlit, OP_XORLIT)
KERNEL(k_andsgpr, OP_ANDSGPR)
KERNEL(k_bitop3sgpr, OP_BITOP3SGPR)
KERNEL(k_cmponly, OP_CMPONLY)
KERNEL(k_cmps, OP_CMPS)
KERNEL(k_sublit, OP_SUBLIT)
KERNEL(k_add3, OP_ADD3)
KERNEL(k_lshlor, OP_LSHLOR)
KERNEL(k_xor, OP_XOR)
KERNEL(k_add, OP_ADD)
KERNEL(k_sub, OP_SUB)
KERNEL(k_lshr, OP_LSHR)
KERNEL(k_andor, OP_ANDOR)
KERNEL(k_or3, OP_OR3)
KERNEL(k_bitop3, OP_BITOP3)
KERNEL(k_bfi, OP_BFI)
KERNEL(k_perm, OP_PERM)
KERNEL(k_alignbyte, OP_ALIGNBYTE)
KERNEL(k_sad, OP_SAD)
KERNEL(k_dot4, OP_DOT4)
KERNEL(k_pksub, OP_PKSUB)
KERNEL(k_pkadd, OP_PKADD)
KERNEL(k_lshladd, OP_LSHLADD)
KERNEL(k_min, OP_MIN)
KERNEL(k_ffbh, OP_FFBH)
KERNEL(k_bcnt, OP_BCNT)
KERNEL(k_mullo, OP_MULLO)
KERNEL(k_mul24, OP_MUL24)
KERNEL(k_mad24, OP_MAD24)
KERNEL(k_cndmask, OP_CNDMASK)
KERNEL(k_cmp_cnd, OP_CMP)
KERNEL(k_movdpp, OP_MOVDPP)
KERNEL(k_ordpp, OP_ORDPP)
KERNEL(k_sdwa, OP_SDWA)
KERNEL(k_xnor, OP_XNOR)
KERNEL(k_bfe, OP_BFE)
KERNEL(k_msad, OP_MSAD)
KERNEL(k_sadhi, OP_SADHI)
KERNEL(k_subsdwa, OP_SUBSDWA)

typedef void (*kern_t)(u32 *, int, u32);
struct Entry { const char *name; kern_t k; int per_op; };

int main()
{
  const int blocks = 256 * 8, threads = 256, iters = 4000;
  u32 *out;
  hipMalloc(&out, (size_t)blocks * threads * 4);
  Entry es[] = {
      {"v_and_b32", k_and, 1}, {"v_or_b32", k_or, 1}, {"v_lshlrev_b32", k_lshl, 1}, {"v_mov_b32", k_mov, 1},
      {"v_and_b32 literal", k_andlit, 1}, {"v_xor_b32 literal", k_xorlit, 1}, {"v_and_b32 sgpr", k_andsgpr, 1},
      {"v_bitop3 sgpr", k_bitop3sgpr, 1}, {"v_cmp(vcc)+v_xor", k_cmponly, 2},
      {"v_cmp(sgpr)+v_xor", k_cmps, 2}, {"v_sub_u32 literal", k_sublit, 1}, {"v_add3_u32", k_add3, 1}, {"v_lshl_or_b32", k_lshlor, 1},
      {"v_xor_b32", k_xor, 1}, {"v_add_u32", k_add, 1}, {"v_sub_u32", k_sub, 1}, {"v_lshrrev_b32", k_lshr, 1},
      {"v_and_or_b32", k_andor, 1}, {"v_or3_b32", k_or3, 1}, {"v_bitop3_b32", k_bitop3, 1}, {"v_bfi_b32", k_bfi, 1},
      {"v_xnor_b32", k_xnor, 1}, {"v_bfe_u32", k_bfe, 1},
      {"v_perm_b32", k_perm, 1}, {"v_alignbyte_b32", k_alignbyte, 1}, {"v_sad_u8", k_sad, 1}, {"v_msad_u8", k_msad, 1},
      {"v_sad_hi_u8", k_sadhi, 1}, {"v_dot4_u32_u8", k_dot4, 1},
      {"v_pk_sub_u16", k_pksub, 1}, {"v_pk_add_u16", k_pkadd, 1}, {"v_lshl_add_u32", k_lshladd, 1},
      {"v_min_u32", k_min, 1}, {"v_ffbh_u32", k_ffbh, 1}, {"v_bcnt_u32_b32", k_bcnt, 1}, {"v_mul_lo_u32", k_mullo, 1},
      {"v_mul_u32_u24", k_mul24, 1}, {"v_mad_u32_u24", k_mad24, 1}, {"v_cndmask_b32", k_cndmask, 1},
      {"v_cmp+v_cndmask", k_cmp_cnd, 2}, {"v_mov_b32_dpp", k_movdpp, 1}, {"v_or_b32_dpp", k_ordpp, 1},
      {"v_or_b32_sdwa", k_sdwa, 1}, {"v_sub_u32_sdwa(byte)", k_subsdwa, 1},
  };
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  int clk_khz = 0;
  hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
  printf("reported clock %d kHz; %d blocks x %d threads, %d iters x 32 ops\n", clk_khz, blocks, threads, iters);
  for (auto &e : es) {
    hipLaunchKernelGGL(e.k, dim3(blocks), dim3(threads), 0, 0, out, 100, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(e.k, dim3(blocks), dim3(threads), 0, 0, out, iters, 1u);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD: blocks*4 waves * iters*32*per_op / 1024 SIMDs
    double winst = (double)blocks * 4.0 * iters * 32.0 * e.per_op / 1024.0;
    double ns_per = ms * 1e6 / winst;
    printf("%-22s %8.3f ms  %6.3f ns/wave-instr/SIMD  = %5.2f cycles @2.4GHz\n", e.name, ms, ns_per, ns_per * 2.4);
  }
  return 0;
}
