// Development only: workgroups per CU the runtime reports for 256-thread kernels by dynamic LDS size, and a census
// (how many workgroups of a grid were resident on one CU at the same time).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256, 4) k(unsigned *out, unsigned *cu_count, int spin)
{
  extern __shared__ unsigned char smem[];
  smem[threadIdx.x] = 1;
  __syncthreads();
  // count concurrent residents per CU: hardware id of the CU from HW_REG_HW_ID / XCC_ID
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const unsigned cu = ((xcc & 0xf) << 8) | ((hw >> 8) & 0xff);     // se/sh/cu bits + xcc
  if (threadIdx.x == 0) {
    unsigned now = atomicAdd(&cu_count[cu], 1u) + 1u;
    atomicMax(&out[0], now);
    for (int i = 0; i < spin; i++) __builtin_amdgcn_s_sleep(100);
    atomicSub(&cu_count[cu], 1u);
  }
  __syncthreads();
  if (smem[threadIdx.x] == 7) out[1] = 1;
}
int main()
{
  unsigned *out, *cnt;
  hipMalloc(&out, 8); hipMalloc(&cnt, 4096 * 4);
  for (int kb : {16, 32, 36, 38, 39, 40, 41, 42, 44, 48, 51, 52, 53, 54, 56, 64, 72, 80}) {
    const int smem = kb * 1024;
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int nb = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, 256, smem);
    hipMemset(out, 0, 8); hipMemset(cnt, 0, 4096 * 4);
    hipLaunchKernelGGL(k, dim3(256 * 8), dim3(256), smem, 0, out, cnt, 2000);
    hipDeviceSynchronize();
    unsigned h[2]; hipMemcpy(h, out, 8, hipMemcpyDeviceToHost);
    printf("dynamic LDS %3d KiB: API says %d workgroups/CU, census max resident per CU id %u\n", kb, nb, h[0]);
  }
  return 0;
}
