// mpc_ring.h -- lines from global memory straight into LDS (global_load_lds_dwordx4) and out of it one line per
// lane: the helpers of the per-wave line rings (mpc_vpc_lane.hip explains the design; mpc_kernels.hip uses the same
// ring for the BDI baseline).  gfx950 only.
#pragma once
#include "mpc_kernel_common.h"

#ifndef MPC_RING_NT
#define MPC_RING_NT 1   /* non-temporal DMA loads */
#endif

template <int NQ>
__device__ __forceinline__ u32 ring_swz(u32 line) { return (line >> (NQ == 2 ? 3 : (NQ == 4 ? 2 : 1))) & (u32)(NQ - 1); }

// byte offset (from the group's first byte) of the 16-byte unit that lane `lane` of DMA instruction j fetches,
// less what the request's base and the instruction's offset field add (1 KiB per instruction)
template <int NQ>
__device__ __forceinline__ u32 ring_src_off(int j, u32 lane)
{
  const u32 q = 64u * (u32)j + lane, line = q / NQ, pos = q % NQ;
  return 16u * (line * NQ + (pos ^ ring_swz<NQ>(line))) - 1024u * (u32)j;
}

// LDS byte offset (inside a stage) of piece k of line `ll` (0..63)
template <int NQ>
__device__ __forceinline__ u32 ring_rd_off(u32 ll, int k) { return 16u * (ll * NQ + ((u32)k ^ ring_swz<NQ>(ll))); }

#if MPC_RING_NT
#define MPC_GLDS_AUX " nt"
#else
#define MPC_GLDS_AUX ""
#endif
// up to 4 KiB (NP pieces of 1 KiB) from global memory into LDS at lds_dst; M0 carries the LDS base and is put back
// (the compiler reserves it and does not see it change inside the statement).  No "memory" clobber: nothing the
// compiler knows about is written (the stage is only ever read by ring_read's statement, and volatile statements
// keep their order); with one, every table read behind it would stop being a scalar load.
template <int NP>
__device__ __forceinline__ void ring_glds(const u32 *lane_off, const void *gbase, u32 lds_dst)
{
  u32 keep;
  if constexpr (NP == 4)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %5" MPC_GLDS_AUX "\n\tglobal_load_lds_dwordx4 %2, %5 offset:1024" MPC_GLDS_AUX "\n\t"
                 "global_load_lds_dwordx4 %3, %5 offset:2048" MPC_GLDS_AUX "\n\tglobal_load_lds_dwordx4 %4, %5 offset:3072" MPC_GLDS_AUX "\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_off[0]), "v"(lane_off[1]), "v"(lane_off[2]), "v"(lane_off[3]), "s"(gbase), "s"(lds_dst));
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %3" MPC_GLDS_AUX "\n\tglobal_load_lds_dwordx4 %2, %3 offset:1024" MPC_GLDS_AUX "\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_off[0]), "v"(lane_off[1]), "s"(gbase), "s"(lds_dst));
}

// one group of 64 lines (NQ KiB) into the stage at lds_dst
template <int NQ>
__device__ __forceinline__ void ring_request(const u32 (&lane_off)[NQ], const uint4 *gbase_, u32 lds_dst_)
{
  // wave-uniform by construction; said so explicitly (the asm statements need scalar registers)
  const u64 gb = reinterpret_cast<u64>(gbase_);
  const u32 gb_hi = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(gb >> 32)), gb_lo = (u32)__builtin_amdgcn_readfirstlane((int)(u32)gb);
  const uint4 *gbase = reinterpret_cast<const uint4 *>(((u64)gb_hi << 32) | (u64)gb_lo);      // (the builtin returns a signed int)
  const u32 lds_dst = (u32)__builtin_amdgcn_readfirstlane((int)lds_dst_);
  if constexpr (NQ == 8) {
    ring_glds<4>(&lane_off[0], gbase, lds_dst);
    ring_glds<4>(&lane_off[4], gbase + 256, lds_dst + 4096u);
  } else {
    ring_glds<NQ>(&lane_off[0], gbase, lds_dst);
  }
}

// The streaming loop's state is wave-uniform by construction, but the compiler's divergence analysis gives up on values
// that travel through the group code by reference (it then keeps them in vector registers and branches with EXEC masks:
// +40 vector instructions per group, measured).  uni() says so explicitly: the value of the first active lane, in a scalar
// register.
__device__ __forceinline__ u32 uni(u32 x) { return (u32)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ bool uni(bool x) { return __builtin_amdgcn_readfirstlane((int)x) != 0; }

// wait until at most N of the wave's vector-memory operations (the DMA instructions, in issue order) are outstanding
template <int N>
__device__ __forceinline__ void ring_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N)); }

// the line at LDS addresses a[0..NQ-1] (one 16-byte piece each) into v; the reads and their wait are one statement
template <int NQ>
__device__ __forceinline__ void ring_read(uint4 (&v)[NQ], const u32 (&a)[NQ])
{
  typedef u32 v4 __attribute__((ext_vector_type(4)));
  v4 r[NQ];
  if constexpr (NQ == 2)
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(r[0]), "=&v"(r[1]) : "v"(a[0]), "v"(a[1]));
  else if constexpr (NQ == 4)
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]));
  else
    asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %9\n\tds_read_b128 %2, %10\n\tds_read_b128 %3, %11\n\t"
                 "ds_read_b128 %4, %12\n\tds_read_b128 %5, %13\n\tds_read_b128 %6, %14\n\tds_read_b128 %7, %15\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]));
#pragma unroll
  for (int k = 0; k < NQ; k++) v[k] = make_uint4(r[k].x, r[k].y, r[k].z, r[k].w);
}


