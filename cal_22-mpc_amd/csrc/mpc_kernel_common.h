// mpc_kernel_common.h -- device helpers shared by the gfx950 kernels: DPP cross-lane
// primitives for the lanes that hold one line, SWAR byte arithmetic on 32-bit words,
// per-workgroup LDS statistics.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mpc_device.h"

typedef unsigned long long u64;
typedef uint32_t u32;

#define H80 0x80808080u
#define L7F 0x7f7f7f7fu

// ---------------------------------------------------------------------------
// DPP helpers
// ---------------------------------------------------------------------------
#define QP(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))
#define DPP_ROW_SHL(n) (0x100 + (n))
#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_ROW_HALF_MIRROR 0x141

template <int CTRL>
__device__ __forceinline__ u32 dpp(u32 v)
{
  return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}

// Cross-lane operations inside the LPL lanes that hold one line.
template <int LPL> struct Grp;

template <> struct Grp<4> {
  static __device__ __forceinline__ u32 prev(u32 v) { return dpp<QP(0, 0, 1, 2)>(v); }   // lane g-1 (g=0: self)
  static __device__ __forceinline__ u32 next(u32 v) { return dpp<QP(1, 2, 3, 3)>(v); }   // lane g+1 (last: self)
  static __device__ __forceinline__ u32 first(u32 v) { return dpp<QP(0, 0, 0, 0)>(v); }
  static __device__ __forceinline__ u32 last(u32 v) { return dpp<QP(3, 3, 3, 3)>(v); }
  static __device__ __forceinline__ u32 red_or(u32 v) { v |= dpp<QP(1, 0, 3, 2)>(v); v |= dpp<QP(2, 3, 0, 1)>(v); return v; }
  static __device__ __forceinline__ u32 red_add(u32 v) { v += dpp<QP(1, 0, 3, 2)>(v); v += dpp<QP(2, 3, 0, 1)>(v); return v; }
  static __device__ __forceinline__ u32 red_min(u32 v) { v = min(v, dpp<QP(1, 0, 3, 2)>(v)); v = min(v, dpp<QP(2, 3, 0, 1)>(v)); return v; }
};

template <> struct Grp<2> {
  static __device__ __forceinline__ u32 prev(u32 v) { return dpp<QP(0, 0, 2, 2)>(v); }
  static __device__ __forceinline__ u32 next(u32 v) { return dpp<QP(1, 1, 3, 3)>(v); }
  static __device__ __forceinline__ u32 first(u32 v) { return dpp<QP(0, 0, 2, 2)>(v); }
  static __device__ __forceinline__ u32 last(u32 v) { return dpp<QP(1, 1, 3, 3)>(v); }
  static __device__ __forceinline__ u32 red_or(u32 v) { return v | dpp<QP(1, 0, 3, 2)>(v); }
  static __device__ __forceinline__ u32 red_add(u32 v) { return v + dpp<QP(1, 0, 3, 2)>(v); }
  static __device__ __forceinline__ u32 red_min(u32 v) { return min(v, dpp<QP(1, 0, 3, 2)>(v)); }
};

template <> struct Grp<8> {
  // groups are the aligned 8-lane halves of a 16-lane DPP row
  static __device__ __forceinline__ u32 prev(u32 v) { return dpp<DPP_ROW_SHR(1)>(v); }   // g=0: other group / 0 (caller masks)
  static __device__ __forceinline__ u32 next(u32 v) { return dpp<DPP_ROW_SHL(1)>(v); }   // g=7: other group / 0 (caller masks)
  static __device__ __forceinline__ u32 first(u32 v)
  {
    u32 q = dpp<QP(0, 0, 0, 0)>(v);          // lane 0 of each quad
    u32 s = dpp<DPP_ROW_SHR(4)>(q);          // the quad below
    return (threadIdx.x & 4) ? s : q;
  }
  static __device__ __forceinline__ u32 last(u32 v)
  {
    u32 q = dpp<QP(3, 3, 3, 3)>(v);
    u32 s = dpp<DPP_ROW_SHL(4)>(q);
    return (threadIdx.x & 4) ? q : s;
  }
  static __device__ __forceinline__ u32 red_or(u32 v) { v |= dpp<QP(1, 0, 3, 2)>(v); v |= dpp<QP(2, 3, 0, 1)>(v); v |= dpp<DPP_ROW_HALF_MIRROR>(v); return v; }
  static __device__ __forceinline__ u32 red_add(u32 v) { v += dpp<QP(1, 0, 3, 2)>(v); v += dpp<QP(2, 3, 0, 1)>(v); v += dpp<DPP_ROW_HALF_MIRROR>(v); return v; }
  static __device__ __forceinline__ u32 red_min(u32 v) { v = min(v, dpp<QP(1, 0, 3, 2)>(v)); v = min(v, dpp<QP(2, 3, 0, 1)>(v)); v = min(v, dpp<DPP_ROW_HALF_MIRROR>(v)); return v; }
};

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
__device__ __forceinline__ u32 perm(u32 hi, u32 lo, u32 sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
__device__ __forceinline__ u32 alignbyte(u32 hi, u32 lo, u32 n) { return __builtin_amdgcn_alignbyte(hi, lo, n); }
__device__ __forceinline__ u32 sum_bytes(u32 x, u32 acc) { return __builtin_amdgcn_sad_u8(x, 0u, acc); }
__device__ __forceinline__ u32 sum_sq_bytes(u32 x, u32 acc) { return __builtin_amdgcn_udot4(x, x, acc, false); }

// ---------------------------------------------------------------------------
// statistics: per-workgroup LDS accumulators, flushed once per workgroup
// ---------------------------------------------------------------------------
struct WgStats {
  u32 *hist;   // [K * bins] (LDS)
  u64 *sums;   // [2 * K]    (LDS): sum_r, sum_r2
};

__device__ __forceinline__ void stats_init(const WgStats &s, int K, int bins)
{
  for (int i = threadIdx.x; i < K * bins; i += blockDim.x) s.hist[i] = 0;
  for (int i = threadIdx.x; i < 2 * K; i += blockDim.x) s.sums[i] = 0;
  __syncthreads();
}

__device__ __forceinline__ void stats_flush(const WgStats &s, int K, int bins, u64 *g)
{
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * K; i += blockDim.x) {
    u64 v = s.sums[i];
    if (v) atomicAdd(&g[i], v);
  }
  for (int i = threadIdx.x; i < K * bins; i += blockDim.x) {
    u32 v = s.hist[i];
    if (v) atomicAdd(&g[2 * K + i], (u64)v);
  }
}


// LDS bytes of the per-workgroup statistics (sums then histogram)
__host__ __device__ static inline size_t vpc_stats_smem(int K, int bins)
{
  return 16 * (size_t)((2 * K * 8 + 15) / 16) + 16 * (size_t)((K * bins * 4 + 15) / 16);
}

