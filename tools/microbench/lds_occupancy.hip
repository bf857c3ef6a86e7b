// How many workgroups run concurrently as a function of the dynamic LDS they declare (gfx950)?
// Each workgroup spins a fixed VALU loop; total time / single-workgroup time = number of rounds.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ void __launch_bounds__(512) spin(unsigned* out, int iters) {
  extern __shared__ unsigned lds[];
  unsigned x = threadIdx.x + blockIdx.x;
  lds[threadIdx.x] = x;
  __syncthreads();
  for (int i = 0; i < iters; ++i) x = x * 1664525u + 1013904223u;
  out[blockIdx.x * blockDim.x + threadIdx.x] = x + lds[(threadIdx.x + 1) & 511];
}
int main() {
  unsigned* d; CK(hipMalloc(&d, 4096 * 512 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int sizes[] = {4096, 32768, 65536, 65540, 81920, 98304, 131072, 155680, 163840};
  for (int lds : sizes) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(spin), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    for (int blocks : {1, 8, 32, 64, 128, 256, 512}) {
      hipLaunchKernelGGL(spin, dim3(blocks), dim3(512), lds, 0, d, 200000);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(spin, dim3(blocks), dim3(512), lds, 0, d, 200000);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("{\"lds_bytes\": %d, \"blocks\": %d, \"ms\": %.3f}\n", lds, blocks, ms);
    }
  }
  return 0;
}
