// Development only: how well do HBM streaming and a fixed amount of vector arithmetic per group of 64 lines overlap
// under (a) the product kernel's register double buffer (one 64-byte line per lane, next group in flight) and
// (b) a per-wave ring of S 4-KiB stages in LDS filled by global_load_lds_dwordx4 (coalesced, optional nt), read out
// transposed to one line per lane.
//   hipcc -O3 --offload-arch=gfx950 -o tools/dev/ringbw tools/dev/ringbw.hip && tools/dev/ringbw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u32;
typedef unsigned long long u64;

// REPS x 48 fast-rate vector instructions on the 16 words of a line
template <int REPS>
__device__ __forceinline__ u32 work(const uint4 (&v)[4])
{
  u32 c[16];
#pragma unroll
  for (int i = 0; i < 4; i++) { c[4 * i] = v[i].x; c[4 * i + 1] = v[i].y; c[4 * i + 2] = v[i].z; c[4 * i + 3] = v[i].w; }
#pragma unroll
  for (int r = 0; r < REPS; r++) {
#pragma unroll
    for (int e = 0; e < 16; e++) c[e] = (c[e] ^ (c[(e + 1) & 15] >> 1)) + c[(e + 5) & 15];
  }
  u32 a = 0;
#pragma unroll
  for (int e = 0; e < 16; e++) a ^= c[e];
  return a;
}

template <int REPS, int WAVES>
__global__ void __launch_bounds__(256, WAVES) k_regs(const uint4 *__restrict__ p, u32 n_lines, u32 *sink)
{
  u32 acc = 0;
  const u32 lane = threadIdx.x & 63u, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const u32 stride = gridDim.x * 256u;
  u32 g = (blockIdx.x * 4u + w) * 64u;
  uint4 a[4], b[4];
  auto fetch = [&](uint4 (&d)[4], u32 g0) {
    const uint4 *s = p + (u64)min(g0 + lane, n_lines - 1u) * 4u;
#pragma unroll
    for (int k = 0; k < 4; k++) d[k] = s[k];
  };
  if (g < n_lines) fetch(a, g);
  while (g < n_lines) {
    fetch(b, g + stride);
    acc ^= work<REPS>(a);
    g += stride;
    if (g >= n_lines) break;
    fetch(a, g + stride);
    acc ^= work<REPS>(b);
    g += stride;
  }
  if (acc == 0x9e3779b9u) *sink = acc;
}

__device__ __forceinline__ u32 swz(u32 e) { const u32 line = e >> 2; return (line << 2) | ((e & 3u) ^ ((line >> 2) & 3u)); }

// one group of 64 lines (4 KiB) from global memory straight into an LDS stage: four 1-KiB pieces, the piece offset
// in the instruction's offset field (it moves the global and the LDS address alike)
template <int AUX>
__device__ __forceinline__ void glds_group(u32 lane_off, const void *gbase, u32 lds_dst)
{
  u32 keep;
  if constexpr (AUX == 0)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072\n\t"
                 "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(lane_off), "s"(gbase), "s"(lds_dst) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2 nt\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024 nt\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:2048 nt\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072 nt\n\t"
                 "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(lane_off), "s"(gbase), "s"(lds_dst) : "memory");
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

// S stages per wave; T threads per workgroup; WAVES = launch bound (waves per SIMD)
template <int REPS, int S, int T, int WAVES, int AUX>
__global__ void __launch_bounds__(T, WAVES) k_ring(const uint4 *__restrict__ p, u32 n_lines, u32 *sink)
{
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];
  u32 acc = 0;
  const u32 lane = threadIdx.x & 63u, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr u32 WPB = T / 64;
  const u32 stride = gridDim.x * WPB * 64u;
  u32 g = (blockIdx.x * WPB + w) * 64u;
  uint4 *ring = smem + w * (S * 256);
  const u32 ring_lds = (u32)(uintptr_t)(__attribute__((address_space(3))) void *)ring;   // LDS byte address
  // lane l of instruction k deposits at unit k*64 + l the 16-byte unit that belongs there: unit k*64 + u0(l)
  const u32 lane_off = 16u * ((lane & ~3u) | ((lane & 3u) ^ ((lane >> 4) & 3u)));
  u32 rd_addr[4];
#pragma unroll
  for (int k = 0; k < 4; k++) rd_addr[k] = ring_lds + 16u * swz(lane * 4 + k);
  auto request = [&](u32 g0, int stage) {
    const u32 gg = min(g0, n_lines - 64u);            // past the end: re-read the last group (never evaluated)
    glds_group<AUX>(lane_off, p + (u64)gg * 4u, ring_lds + (u32)stage * 4096u);
  };
#pragma unroll
  for (int s = 0; s < S; s++) request(g + s * stride, s);
  int stage = 0;
  while (g < n_lines) {
#pragma unroll
    for (int s = 0; s < S; s++) {                      // unrolled over the stages: the stage index is a constant
      if (g < n_lines) {
        wait_vm<4 * (S - 1)>();
        uint4 c[4];
        typedef u32 v4 __attribute__((ext_vector_type(4)));
        v4 r0, r1, r2, r3;
        // four 16-byte reads + their wait in one statement (the compiler would split the reads otherwise)
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
                     : "v"(rd_addr[0] + s * 4096u), "v"(rd_addr[1] + s * 4096u), "v"(rd_addr[2] + s * 4096u), "v"(rd_addr[3] + s * 4096u)
                     : "memory");
        c[0] = make_uint4(r0.x, r0.y, r0.z, r0.w); c[1] = make_uint4(r1.x, r1.y, r1.z, r1.w);
        c[2] = make_uint4(r2.x, r2.y, r2.z, r2.w); c[3] = make_uint4(r3.x, r3.y, r3.z, r3.w);
        request(g + S * stride, s);
        __builtin_amdgcn_sched_barrier(0);       // the request is issued before the arithmetic, not after it
        acc ^= work<REPS>(c);
        g += stride;
      }
    }
  }
  wait_vm<0>();
  (void)stage;
  if (acc == 0x9e3779b9u) *sink = acc;
}

template <typename F>
static double time_ms(F launch)
{
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); launch();
  hipDeviceSynchronize();
  float sum = 0;
  for (int r = 0; r < 6; r++) {
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    sum += ms;
  }
  return sum / 6;
}

int main()
{
  const u64 bytes = 16ull << 30;
  const u32 n_lines = (u32)(bytes / 64);
  uint4 *d; u32 *sink;
  if (hipMalloc(&d, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&sink, 4);
  hipMemset(d, 0x5a, bytes);
  hipDeviceSynchronize();
  double t;
#define RUN(NAME, KERN, GRID, T, SMEM)                                                                   \
  {                                                                                                      \
    if ((SMEM) > 65536) hipFuncSetAttribute((const void *)KERN, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(SMEM)); \
    t = time_ms([&] { hipLaunchKernelGGL(KERN, dim3(GRID), dim3(T), SMEM, 0, d, n_lines, sink); });      \
    hipError_t e = hipGetLastError();                                                                    \
    printf("%-52s %7.3f ms  %7.1f GB/s  %s\n", NAME, t, bytes / t / 1e6, e == hipSuccess ? "" : hipGetErrorString(e)); \
    fflush(stdout);                                                                                      \
  }
#define SET(R)                                                                                           \
  RUN("regs  4x256/CU 4w/SIMD                 reps " #R, (k_regs<R, 4>), 256 * 32, 256, 0)               \
  RUN("regs  3 w/SIMD                         reps " #R, (k_regs<R, 3>), 256 * 24, 256, 0)               \
  RUN("ring S=1 256thr 4w/SIMD aux0           reps " #R, (k_ring<R, 1, 256, 4, 0>), 256 * 32, 256, 4 * 1 * 4096) \
  RUN("ring S=1 256thr 4w/SIMD nt             reps " #R, (k_ring<R, 1, 256, 4, 1>), 256 * 32, 256, 4 * 1 * 4096) \
  RUN("ring S=2 256thr 4w/SIMD aux0           reps " #R, (k_ring<R, 2, 256, 4, 0>), 256 * 32, 256, 4 * 2 * 4096) \
  RUN("ring S=2 256thr 4w/SIMD nt             reps " #R, (k_ring<R, 2, 256, 4, 1>), 256 * 32, 256, 4 * 2 * 4096) \
  RUN("ring S=2 1024thr 4w/SIMD nt (1 WG/CU)  reps " #R, (k_ring<R, 2, 1024, 4, 1>), 256 * 1, 1024, 16 * 2 * 4096) \
  RUN("ring S=2 1024thr nt, 4 WG per CU queued reps " #R, (k_ring<R, 2, 1024, 4, 1>), 256 * 4, 1024, 16 * 2 * 4096) \
  RUN("ring S=2 512thr 4w/SIMD nt (2 WG/CU)   reps " #R, (k_ring<R, 2, 512, 4, 1>), 256 * 16, 512, 8 * 2 * 4096) \
  RUN("ring S=3 256thr 3w/SIMD nt             reps " #R, (k_ring<R, 3, 256, 3, 1>), 256 * 24, 256, 4 * 3 * 4096) \
  RUN("ring S=2 256thr 5w/SIMD nt (needs <=96 VGPR) reps " #R, (k_ring<R, 2, 256, 5, 1>), 256 * 40, 256, 4 * 2 * 4096) \
  RUN("ring S=4 256thr 2w/SIMD nt             reps " #R, (k_ring<R, 4, 256, 2, 1>), 256 * 16, 256, 4 * 4 * 4096)
  SET(0)
  SET(6)
  SET(9)
  SET(13)
  return 0;
}
