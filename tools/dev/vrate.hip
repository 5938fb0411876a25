// Development: issue rate of the vector instructions the group code is made of, at W waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/dev/vrate.hip -o tools/dev/vrate && tools/dev/vrate
// Prints, per instruction, the time of a loop of 8 independent chains relative to v_add_u32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define OP8(S) \
  asm volatile(S : "+v"(a0) : "v"(b), "v"(c), "s"(s0), "s"(m64)); asm volatile(S : "+v"(a1) : "v"(b), "v"(c), "s"(s0), "s"(m64)); \
  asm volatile(S : "+v"(a2) : "v"(b), "v"(c), "s"(s0), "s"(m64)); asm volatile(S : "+v"(a3) : "v"(b), "v"(c), "s"(s0), "s"(m64)); \
  asm volatile(S : "+v"(a4) : "v"(b), "v"(c), "s"(s0), "s"(m64)); asm volatile(S : "+v"(a5) : "v"(b), "v"(c), "s"(s0), "s"(m64)); \
  asm volatile(S : "+v"(a6) : "v"(b), "v"(c), "s"(s0), "s"(m64)); asm volatile(S : "+v"(a7) : "v"(b), "v"(c), "s"(s0), "s"(m64));

#define KERNEL(NAME, S) \
__global__ void __launch_bounds__(1024) k_##NAME(unsigned *out, int iters, unsigned seed) { \
  unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
  unsigned b = seed * 2654435761u + threadIdx.x, c = seed ^ 0x0c0d0e0fu; unsigned s0 = __builtin_amdgcn_readfirstlane(seed | 0x01020304u); \
  unsigned long long m64 = __builtin_amdgcn_read_exec() ^ (0x5555aaaaull * seed); asm volatile("s_mov_b64 vcc, %0" :: "s"(m64) : "vcc"); \
  if ((seed >> 16) != 0u && (threadIdx.x & 63u) >= (seed >> 16)) return;      /* only the first (seed >> 16) lanes of every wave run */ \
  for (int i = 0; i < iters; i++) { OP8(S) OP8(S) OP8(S) OP8(S) } \
  if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345u) out[0] = a0; }

KERNEL(add,      "v_add_u32 %0, %0, %1")
KERNEL(snop,     "s_nop 0")
KERNEL(xor_same, "v_xor_b32 %0, %0, %0")
KERNEL(perm_c,   "v_perm_b32 %0, %0, %0, %3")
KERNEL(dot4_same,"v_dot4_u32_u8 %0, %0, %0, %0")
KERNEL(and_,     "v_and_b32 %0, %0, %1")
KERNEL(and_s,    "v_and_b32 %0, %3, %0")
KERNEL(xor_,     "v_xor_b32 %0, %0, %1")
KERNEL(mov,      "v_mov_b32 %0, %1")
KERNEL(sub,      "v_sub_u32 %0, %0, %1")
KERNEL(lshr,     "v_lshrrev_b32 %0, 3, %0")
KERNEL(bitop3,   "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")
KERNEL(bitop3_s, "v_bitop3_b32 %0, %0, %1, %3 bitop3:0x96")
KERNEL(perm,     "v_perm_b32 %0, %0, %1, %2")
KERNEL(perm_s,   "v_perm_b32 %0, %0, %1, %3")
KERNEL(sad,      "v_sad_u8 %0, %1, %2, %0")
KERNEL(sad0,     "v_sad_u8 %0, %1, 0, %0")
KERNEL(dot4,     "v_dot4_u32_u8 %0, %1, %2, %0")
KERNEL(or3,      "v_or3_b32 %0, %0, %1, %2")
KERNEL(and_or,   "v_and_or_b32 %0, %0, %1, %2")
KERNEL(add3,     "v_add3_u32 %0, %0, %1, %2")
KERNEL(lshl_add, "v_lshl_add_u32 %0, %0, 2, %1")
KERNEL(lshl_or,  "v_lshl_or_b32 %0, %0, 2, %1")
KERNEL(alignbyte,"v_alignbyte_b32 %0, %0, %1, 1")
KERNEL(alignbit, "v_alignbit_b32 %0, %0, %1, 5")
KERNEL(cndmask,  "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL(cnd_e64,  "v_cndmask_b32_e64 %0, %0, %1, %4")
KERNEL(cnd_c,    "v_cndmask_b32_e64 %0, 0, %0, %4")
KERNEL(and_lit,  "v_and_b32 %0, 0x7f7f7f7f, %0")
KERNEL(and_inl,  "v_and_b32 %0, 15, %0")
KERNEL(xor_s,    "v_xor_b32 %0, %3, %0")
KERNEL(add_s,    "v_add_u32 %0, %3, %0")
KERNEL(mov_s,    "v_mov_b32 %0, %3")
KERNEL(mov_lit,  "v_mov_b32 %0, 0x12345678")
KERNEL(lshr_s,   "v_lshrrev_b32 %0, %3, %0")
KERNEL(lshl,     "v_lshlrev_b32 %0, 3, %0")
KERNEL(or_,      "v_or_b32 %0, %0, %1")
KERNEL(bitop3_i, "v_bitop3_b32 %0, %0, %1, 15 bitop3:0x96")
KERNEL(addco,    "v_add_co_u32 %0, vcc, %0, %1")
KERNEL(subrev,   "v_subrev_u32 %0, %0, %1")
KERNEL(ashr,     "v_ashrrev_i32 %0, 3, %0")
KERNEL(not_,     "v_not_b32 %0, %0")
KERNEL(bfrev,    "v_bfrev_b32 %0, %0")
KERNEL(ffbl,     "v_ffbl_b32 %0, %0")
KERNEL(max_,     "v_max_u32 %0, %0, %1")
KERNEL(xnor,     "v_xnor_b32 %0, %0, %1")
KERNEL(bfi,      "v_bfi_b32 %0, %0, %1, %2")
KERNEL(mbcnt,    "v_mbcnt_lo_u32_b32 %0, %1, %0")
KERNEL(rfl,      "v_readfirstlane_b32 s10, %0")
KERNEL(cmp_e64,  "v_cmp_eq_u32_e64 s[10:11], %0, %1")
KERNEL(bcnt,     "v_bcnt_u32_b32 %0, %1, %0")
KERNEL(ffbh,     "v_ffbh_u32 %0, %0")
KERNEL(mul24,    "v_mul_u32_u24 %0, %0, %1")
KERNEL(mad24,    "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL(mullo,    "v_mul_lo_u32 %0, %0, %1")
KERNEL(bfe,      "v_bfe_u32 %0, %0, 3, 5")
KERNEL(min_,     "v_min_u32 %0, %0, %1")
KERNEL(max3,     "v_max3_u32 %0, %0, %1, %2")
KERNEL(xad,      "v_xad_u32 %0, %0, %1, %2")
KERNEL(pkadd16,  "v_pk_add_u16 %0, %0, %1")
KERNEL(pksub16,  "v_pk_sub_u16 %0, %0, %1")
KERNEL(pkmax16,  "v_pk_max_u16 %0, %0, %1")
KERNEL(dot8u4,   "v_dot8_u32_u4 %0, %1, %2, %0")
KERNEL(dot2u16,  "v_dot2_u32_u16 %0, %1, %2, %0")
KERNEL(msad,     "v_msad_u8 %0, %1, %2, %0")
KERNEL(cmp_eq,   "v_cmp_eq_u32 vcc, %0, %1")
KERNEL(sdwa_or,  "v_or_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD")
KERNEL(add_dpp,  "v_add_u32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL(mov_dpp,  "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")

struct Ent { const char *name; void (*fn)(unsigned *, int, unsigned); };
#define E(NAME) { #NAME, k_##NAME }
static Ent ents[] = { E(add), E(snop), E(xor_same), E(perm_c), E(dot4_same), E(and_), E(and_s), E(xor_), E(mov), E(sub), E(lshr), E(bitop3), E(bitop3_s), E(perm), E(perm_s), E(sad), E(sad0),
  E(dot4), E(or3), E(and_or), E(add3), E(lshl_add), E(lshl_or), E(alignbyte), E(alignbit), E(cndmask), E(cnd_e64), E(cnd_c), E(and_lit), E(and_inl), E(xor_s), E(add_s), E(mov_s), E(mov_lit), E(lshr_s), E(lshl), E(or_), E(bitop3_i), E(addco), E(subrev), E(ashr), E(not_), E(bfrev), E(ffbl), E(max_), E(xnor), E(bfi), E(mbcnt), E(bcnt), E(ffbh), E(mul24), E(mad24),
  E(mullo), E(bfe), E(min_), E(max3), E(xad), E(pkadd16), E(pksub16), E(pkmax16), E(dot8u4), E(dot2u16), E(msad), E(cmp_eq),
  E(sdwa_or), E(add_dpp), E(mov_dpp) };

int main(int argc, char **argv) {
  int waves_per_simd = argc > 1 ? atoi(argv[1]) : 4;
  int iters = argc > 2 ? atoi(argv[2]) : 4000;
  unsigned *out; CHK(hipMalloc(&out, 64));
  hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
  int cus = p.multiProcessorCount;
  int threads = 64 * 4 * waves_per_simd;             // one workgroup per CU, W waves on each SIMD
  if (threads > 1024) threads = 1024;
  int wg_per_cu = (64 * 4 * waves_per_simd) / threads;
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  double base = 0;
  if (argc > 3) {      // sustained mode: one instruction, back to back for argv[4] seconds (power sampling from outside)
    double secs = argc > 4 ? atof(argv[4]) : 2.0;
    const unsigned seed_arg = 12345u | ((argc > 5 ? (unsigned)atoi(argv[5]) : 0u) << 16);      // argv[5]: active lanes per wave
    for (auto &e : ents) {
      if (strcmp(e.name, argv[3])) continue;
      hipLaunchKernelGGL(e.fn, dim3(cus * wg_per_cu), dim3(threads), 0, 0, out, iters, seed_arg);
      CHK(hipDeviceSynchronize());
      CHK(hipEventRecord(e0));
      hipLaunchKernelGGL(e.fn, dim3(cus * wg_per_cu), dim3(threads), 0, 0, out, iters, seed_arg);
      CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
      float ms1; CHK(hipEventElapsedTime(&ms1, e0, e1));
      int n = (int)(secs * 1000.0 / ms1) + 1;
      CHK(hipEventRecord(e0));
      for (int r = 0; r < n; r++) hipLaunchKernelGGL(e.fn, dim3(cus * wg_per_cu), dim3(threads), 0, 0, out, iters, seed_arg);
      CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
      float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); ms /= n;
      printf("SUSTAIN %s %.4f ms per launch, %.4f ns per wave-instruction per SIMD, %.3f G wave-instructions/s chip-wide\n", e.name, ms,
             ms * 1e6 / ((double)iters * 32 * waves_per_simd), (double)iters * 32 * waves_per_simd * 4 * cus / (ms * 1e6));
    }
    return 0;
  }
  printf("waves/SIMD %d, %d CUs, %d threads x %d workgroups, %d x 32 instructions per wave\n", waves_per_simd, cus, threads, cus * wg_per_cu, iters);
  for (int rep = 0; rep < 2; rep++)
    for (auto &e : ents) {
      hipLaunchKernelGGL(e.fn, dim3(cus * wg_per_cu), dim3(threads), 0, 0, out, iters, 12345u);   // warm
      CHK(hipEventRecord(e0));
      for (int r = 0; r < 3; r++) hipLaunchKernelGGL(e.fn, dim3(cus * wg_per_cu), dim3(threads), 0, 0, out, iters, 12345u);
      CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
      float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
      double ns_per_instr_per_simd = ms * 1e6 / ((double)iters * 32 * waves_per_simd);
      if (!strcmp(e.name, "add")) base = ns_per_instr_per_simd;
      if (rep == 1) printf("%-10s %7.3f ms  %6.3f ns per wave-instruction per SIMD  x%.2f of v_add_u32\n", e.name, ms, ns_per_instr_per_simd, ns_per_instr_per_simd / base);
    }
  return 0;
}
