// Development only: streaming-read bandwidth of a few access patterns on one MI355X (16 GiB buffer).
//   hipcc -O3 --offload-arch=gfx950 -o tools/dev/membw tools/dev/membw.hip && tools/dev/membw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u32;
typedef unsigned long long u64;

template <int MODE>   // 0 plain, 1 nontemporal
__device__ __forceinline__ uint4 ld(const uint4 *p)
{
  if constexpr (MODE == 1) {
    typedef u32 v4 __attribute__((ext_vector_type(4)));
    const v4 t = __builtin_nontemporal_load(reinterpret_cast<const v4 *>(p));
    return make_uint4(t.x, t.y, t.z, t.w);
  } else return *p;
}

// A: coalesced grid-stride, UNR independent loads per iteration
template <int MODE, int UNR>
__global__ void __launch_bounds__(256) k_coalesced(const uint4 *__restrict__ p, u64 n16, u32 *sink)
{
  u32 acc = 0;
  const u64 stride = (u64)gridDim.x * 256u;
  u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
  for (; i + (UNR - 1) * stride < n16; i += UNR * stride) {
    uint4 v[UNR];
#pragma unroll
    for (int k = 0; k < UNR; k++) v[k] = ld<MODE>(p + i + k * stride);
#pragma unroll
    for (int k = 0; k < UNR; k++) acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
  }
  for (; i < n16; i += stride) { uint4 v = ld<MODE>(p + i); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x9e3779b9u) *sink = acc;
}

// D: one 64-byte line per lane (4 x dwordx4 at 64-byte lane stride), next group prefetched
template <int MODE>
__global__ void __launch_bounds__(256) k_lines(const uint4 *__restrict__ p, u64 n_lines, u32 *sink)
{
  u32 acc = 0;
  const u64 stride = (u64)gridDim.x * 256u;
  u64 line = (u64)blockIdx.x * 256u + threadIdx.x;
  uint4 a[4], b[4];
  if (line < n_lines) {
#pragma unroll
    for (int k = 0; k < 4; k++) a[k] = ld<MODE>(p + line * 4 + k);
  }
  while (line < n_lines) {
    const u64 nx = line + stride < n_lines ? line + stride : line;
#pragma unroll
    for (int k = 0; k < 4; k++) b[k] = ld<MODE>(p + nx * 4 + k);
#pragma unroll
    for (int k = 0; k < 4; k++) acc ^= a[k].x ^ a[k].y ^ a[k].z ^ a[k].w;
#pragma unroll
    for (int k = 0; k < 4; k++) a[k] = b[k];
    line += stride;
  }
  if (acc == 0x9e3779b9u) *sink = acc;
}

// F: coalesced loads (instruction k reads 16-byte units k*64 + lane of the group), transposed to one line per lane
// through the wave's 4 KiB of LDS (swizzled: conflict-free on both sides), next group prefetched
__device__ __forceinline__ u32 swz(u32 e) { const u32 line = e >> 2; return (line << 2) | ((e & 3u) ^ ((line >> 2) & 3u)); }

template <int MODE>
__global__ void __launch_bounds__(256, 4) k_lines_lds(const uint4 *__restrict__ p, u64 n_lines, u32 *sink)
{
  __shared__ uint4 buf[4][256];
  u32 acc = 0;
  const u32 lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
  const u64 stride = (u64)gridDim.x * 256u;                 // lines per grid step
  u64 g = ((u64)blockIdx.x * 4u + w) * 64u;                 // first line of the wave's group
  uint4 a[4], b[4], c[4];
  if (g < n_lines) {
#pragma unroll
    for (int k = 0; k < 4; k++) a[k] = ld<MODE>(p + g * 4 + k * 64 + lane);
  }
  while (g < n_lines) {
    const u64 nx = g + stride < n_lines ? g + stride : g;
#pragma unroll
    for (int k = 0; k < 4; k++) b[k] = ld<MODE>(p + nx * 4 + k * 64 + lane);
#pragma unroll
    for (int k = 0; k < 4; k++) buf[w][swz(k * 64 + lane)] = a[k];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 4; k++) c[k] = buf[w][swz(lane * 4 + k)];
#pragma unroll
    for (int k = 0; k < 4; k++) acc ^= c[k].x ^ (c[k].y + k) ^ c[k].z ^ c[k].w;
#pragma unroll
    for (int k = 0; k < 4; k++) a[k] = b[k];
    g += stride;
  }
  if (acc == 0x9e3779b9u) *sink = acc;
}

// G: the group goes straight from global memory to the wave's LDS stage (global_load_lds_dwordx4: lane l of
// instruction k deposits its 16 bytes at unit k*64 + l), no registers in between; the swizzle sits on the global
// side (position q receives the unit that belongs there).  One stage per wave: the next group is requested right
// after the current one has been read out, and arrives while the current one is "evaluated".
__device__ __forceinline__ u32 unswz(u32 q) { const u32 line = q >> 2; return (line << 2) | ((q & 3u) ^ ((line >> 2) & 3u)); }   // an involution

template <int AUX>
__global__ void __launch_bounds__(256, 4) k_lines_dma(const uint4 *__restrict__ p, u64 n_lines, u32 *sink)
{
  __shared__ uint4 buf[4][256];
  u32 acc = 0;
  const u32 lane = threadIdx.x & 63u, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const u64 stride = (u64)gridDim.x * 256u;
  u64 g = ((u64)blockIdx.x * 4u + w) * 64u;
  auto request = [&](u64 g0) {
#pragma unroll
    for (int k = 0; k < 4; k++)
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(p + g0 * 4 + unswz(k * 64 + lane)),
                                       (void __attribute__((address_space(3))) *)(&buf[w][k * 64]), 16, 0, AUX);
  };
  if (g < n_lines) request(g);
  while (g < n_lines) {
    uint4 c[4];
    __builtin_amdgcn_s_waitcnt(0x0f70);        // vmcnt(0): the group has landed
#pragma unroll
    for (int k = 0; k < 4; k++) c[k] = buf[w][swz(lane * 4 + k)];
    __builtin_amdgcn_s_waitcnt(0xc07f);        // lgkmcnt(0): read out, the stage is free
    const u64 nx = g + stride < n_lines ? g + stride : g;
    request(nx);
#pragma unroll
    for (int k = 0; k < 4; k++) acc ^= c[k].x ^ (c[k].y + k) ^ c[k].z ^ c[k].w;
    g += stride;
  }
  if (acc == 0x9e3779b9u) *sink = acc;
}

template <typename F>
static double time_ms(F launch)
{
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); launch();
  hipDeviceSynchronize();
  float best = 1e9f, sum = 0;
  for (int r = 0; r < 6; r++) {
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best; sum += ms;
  }
  return sum / 6;
}

int main()
{
  const u64 bytes = 16ull << 30, n16 = bytes / 16, n_lines = bytes / 64;
  uint4 *d; u32 *sink;
  if (hipMalloc(&d, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&sink, 4);
  hipMemset(d, 0x5a, bytes);
  hipDeviceSynchronize();
  for (int per_cu : {4, 8, 16, 32, 64}) {
    const int grid = 256 * per_cu;
    double t;
#define RUN(NAME, ...) t = time_ms([&] { __VA_ARGS__; }); printf("grid %5d  %-28s %7.3f ms  %7.1f GB/s\n", grid, NAME, t, bytes / t / 1e6);
    RUN("coalesced x1", hipLaunchKernelGGL((k_coalesced<0, 1>), dim3(grid), dim3(256), 0, 0, d, n16, sink))
    RUN("coalesced x4", hipLaunchKernelGGL((k_coalesced<0, 4>), dim3(grid), dim3(256), 0, 0, d, n16, sink))
    RUN("coalesced x4 nontemporal", hipLaunchKernelGGL((k_coalesced<1, 4>), dim3(grid), dim3(256), 0, 0, d, n16, sink))
    RUN("line per lane", hipLaunchKernelGGL((k_lines<0>), dim3(grid), dim3(256), 0, 0, d, n_lines, sink))
    RUN("coalesced -> LDS -> lines", hipLaunchKernelGGL((k_lines_lds<0>), dim3(grid), dim3(256), 0, 0, d, n_lines, sink))
    RUN("coalesced nt -> LDS -> lines", hipLaunchKernelGGL((k_lines_lds<1>), dim3(grid), dim3(256), 0, 0, d, n_lines, sink))
    RUN("global -> LDS direct aux 0", hipLaunchKernelGGL(k_lines_dma<0>, dim3(grid), dim3(256), 0, 0, d, n_lines, sink))
    RUN("global -> LDS direct aux 2", hipLaunchKernelGGL(k_lines_dma<2>, dim3(grid), dim3(256), 0, 0, d, n_lines, sink))
    RUN("global -> LDS direct aux 1", hipLaunchKernelGGL(k_lines_dma<1>, dim3(grid), dim3(256), 0, 0, d, n_lines, sink))
    RUN("global -> LDS direct aux 3", hipLaunchKernelGGL(k_lines_dma<3>, dim3(grid), dim3(256), 0, 0, d, n_lines, sink))
    RUN("line per lane nontemporal", hipLaunchKernelGGL((k_lines<1>), dim3(grid), dim3(256), 0, 0, d, n_lines, sink))
  }
  return 0;
}
