// mpc_kernel_common.h -- device helpers shared by the gfx950 kernels: SWAR byte arithmetic
// on 32-bit words, the common encoder's row classification, per-workgroup LDS statistics.
#pragma once
#ifndef __HIPCC_RTC__      /* (hiprtc, mpc_jit.h: the runtime's declarations and the fixed-width types are built in) */
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

#include "mpc_device.h"

typedef unsigned long long u64;
typedef uint32_t u32;

// ---------------------------------------------------------------------------
// MPC_TESTING (libmpc_hip_test.so only, cal_22-mpc_amd/build.py): the kernels count how often their alternative
// routes ran -- lines set aside into the LDS queues, drain groups, paired / plain blocks, switches between the two --
// in MPC_ROUTE_COUNT extra words behind the raw statistics of the handle, which mpc_test_routes() (test library
// only) reads; and the launch grid can be capped with the MPC_TEST_GRID environment variable.  The product library
// has neither: the counting code is compiled out and the variable is not read.
// ---------------------------------------------------------------------------
#ifndef MPC_TESTING
#define MPC_TESTING 0
#endif
enum {
  MPC_RT_VPC_DEFERRED = 0,   // lines set aside into a wave's queue
  MPC_RT_VPC_DRAINS,         // drain groups (up to 64 queued lines each)
  MPC_RT_VPC_PAIRED_BLOCKS,  // 128-line blocks evaluated as even lines / odd lines
  MPC_RT_VPC_PLAIN_BLOCKS,   // 128-line blocks evaluated as first half / second half
  MPC_RT_VPC_TO_PAIRED,      // a wave switched from plain to paired groups
  MPC_RT_VPC_TO_PLAIN,       // a wave went back to plain groups to probe
  MPC_RT_VPC_TAIL_GROUPS,    // groups behind the last whole block (plain loads)
  MPC_RT_BDI_DEFERRED,
  MPC_RT_BDI_DRAINS,
  MPC_ROUTE_COUNT = 16
};
// called by ONE lane of a wave
__device__ __forceinline__ void route_add(u64 *routes, int which, u32 v)
{
#if MPC_TESTING
  if (v) atomicAdd(&routes[which], (u64)v);
#else
  (void)routes; (void)which; (void)v;
#endif
}

#define H80 0x80808080u
#define L7F 0x7f7f7f7fu

// ---------------------------------------------------------------------------
// SWAR byte arithmetic on 32-bit words
// ---------------------------------------------------------------------------
__device__ __forceinline__ u32 bsub(u32 a, u32 b)   // per-byte (a - b) mod 256
{
  return ((a | H80) - (b & L7F)) ^ ((a ^ ~b) & H80);
}
__device__ __forceinline__ u32 badd(u32 a, u32 b)   // per-byte (a + b) mod 256
{
  return ((a & L7F) + (b & L7F)) ^ ((a ^ b) & H80);
}
__device__ __forceinline__ u32 fold8(u32 x)          // OR of the 4 bytes
{
  x |= x >> 16;
  x |= x >> 8;
  return x & 0xffu;
}
// v_bitop3_b32: any boolean function of three words in ONE instruction that issues at the fast VALU
// rate on gfx950 (like v_xor / v_add; v_or3, v_and_or, v_bfi, v_xnor, v_cndmask issue 1.6x slower:
// profiles/r01_valu_issue_rates_gfx950.txt).  TT = truth table with a = 0xF0, b = 0xCC, c = 0xAA.
#define BO_A 0xF0u
#define BO_B 0xCCu
#define BO_C 0xAAu
template <unsigned TT>
__device__ __forceinline__ u32 bitop3(u32 a, u32 b, u32 c) { return __builtin_amdgcn_bitop3_b32(a, b, c, TT); }
__device__ __forceinline__ u32 or3(u32 a, u32 b, u32 c) { return bitop3<(BO_A | BO_B | BO_C)>(a, b, c); }
__device__ __forceinline__ u32 and_or(u32 a, u32 b, u32 c) { return bitop3<((BO_A & BO_B) | BO_C)>(a, b, c); }          // (a & b) | c
__device__ __forceinline__ u32 mask_sel(u32 m, u32 a, u32 b) { return bitop3<0xCAu>(m, a, b); }   // m ? a : b, bitwise
__device__ __forceinline__ u32 xor3(u32 a, u32 b, u32 c) { return bitop3<(BO_A ^ BO_B ^ BO_C)>(a, b, c); }
__device__ __forceinline__ u32 maj3(u32 a, u32 b, u32 c) { return bitop3<((BO_A & BO_B) | (BO_A & BO_C) | (BO_B & BO_C))>(a, b, c); }
__device__ __forceinline__ u32 perm(u32 hi, u32 lo, u32 sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
__device__ __forceinline__ u32 alignbyte(u32 hi, u32 lo, u32 n) { return __builtin_amdgcn_alignbyte(hi, lo, n); }
__device__ __forceinline__ u32 sum_bytes(u32 x, u32 acc) { return __builtin_amdgcn_sad_u8(x, 0u, acc); }
__device__ __forceinline__ u32 sum_sq_bytes(u32 x, u32 acc) { return __builtin_amdgcn_udot4(x, x, acc, false); }

// ---------------------------------------------------------------------------
// common encoder on 8 rows of one 16-column group (bit j <-> plane 7-j).  t[4] = transformed
// residue bytes of the group's 16 columns.  Returns the bits of the non-zero rows.
// ---------------------------------------------------------------------------
__device__ __forceinline__ u32 encode_rows(const u32 t[4], u32 &zero_mask)
{
  // per plane: S = any bit set, T = set in >= 2 columns, U = in >= 3 columns
  u32 s01 = t[0] | t[1], t01 = t[0] & t[1];
  u32 s23 = t[2] | t[3], t23 = t[2] & t[3];
  u32 S = s01 | s23;
  u32 T = t01 | t23 | (s01 & s23);
  u32 U = (t01 & s23) | (s01 & t23);
  {   // fold the 4 byte lanes: counts add up across bytes
    u32 S2 = S >> 16, T2 = T >> 16, U2 = U >> 16;
    u32 nU = U | U2 | (T & S2) | (S & T2);
    u32 nT = T | T2 | (S & S2);
    S |= S2; T = nT; U = nU;
  }
  {
    u32 S2 = S >> 8, T2 = T >> 8, U2 = U >> 8;
    u32 nU = U | U2 | (T & S2) | (S & T2);
    u32 nT = T | T2 | (S & S2);
    S |= S2; T = nT; U = nU;
  }
  S &= 0xffu; T &= 0xffu; U &= 0xffu;
  const u32 Sf = fold8(s01);   // columns 0..7
  const u32 Sb = fold8(s23);   // columns 8..15
  // two ones in adjacent columns c, c+1 (c = 0..14)
  u32 A = (t[0] & alignbyte(t[1], t[0], 1)) | (t[1] & alignbyte(t[2], t[1], 1)) |
          (t[2] & alignbyte(t[3], t[2], 1)) | (t[3] & (t[3] >> 8));
  A = fold8(A);
  const u32 single = S & ~T;             // exactly one 1: 7 bits
  const u32 two = T & ~U & A;            // exactly two, adjacent: 8 bits
  const u32 rest = S & ~single & ~two;
  const u32 both = Sf & Sb;
  const u32 half = rest & ~both;         // one 8-column half empty: 12 bits
  const u32 full = rest & both;          // 17 bits
  zero_mask = ~S & 0xffu;
  return 7u * __popc(single) + 8u * __popc(two) + 12u * __popc(half) + 17u * __popc(full);
}

// ---------------------------------------------------------------------------
// statistics: per-workgroup LDS accumulators, flushed once per workgroup
// ---------------------------------------------------------------------------
struct WgStats {
  u32 *hist;   // [K * bins] (LDS)
  u64 *sums;   // [2 * K]    (LDS): sum_r, sum_r2
};

__device__ __forceinline__ void stats_init(const WgStats &s, int K, int bins, int tid, int nthreads)
{
  for (int i = tid; i < K * bins; i += nthreads) s.hist[i] = 0;
  for (int i = tid; i < 2 * K; i += nthreads) s.sums[i] = 0;
  __syncthreads();
}
__device__ __forceinline__ void stats_init(const WgStats &s, int K, int bins) { stats_init(s, K, bins, threadIdx.x, blockDim.x); }

// tid / nthreads: the caller's thread index and workgroup size.  (A kernel that passes them as lane + 64 * wave and a
// constant need not keep threadIdx.x and the dispatch packet's pointer alive from its first instruction to its last:
// the lane kernel spilled exactly those.)
__device__ __forceinline__ void stats_flush(const WgStats &s, int K, int bins, u64 *g, int tid, int nthreads)
{
  __syncthreads();
  for (int i = tid; i < 2 * K; i += nthreads) {
    u64 v = s.sums[i];
    if (v) atomicAdd(&g[i], v);
  }
  for (int i = tid; i < K * bins; i += nthreads) {
    u32 v = s.hist[i];
    if (v) atomicAdd(&g[2 * K + i], (u64)v);
  }
}
__device__ __forceinline__ void stats_flush(const WgStats &s, int K, int bins, u64 *g) { stats_flush(s, K, bins, g, threadIdx.x, blockDim.x); }


// LDS bytes of the per-workgroup statistics (sums then histogram)
__host__ __device__ static inline size_t vpc_stats_smem(int K, int bins)
{
  return 16 * (size_t)((2 * K * 8 + 15) / 16) + 16 * (size_t)((K * bins * 4 + 15) / 16);
}

