// Instruction-issue-rate microbenchmark for gfx950 (MI355X).
// Measures the sustained chip-wide rate of the integer / fp64 multiply
// instructions a 255-bit Montgomery multiplier can be built from, so that the
// VRF kernels can be priced against a *measured* VALU ceiling (SURVEY.md §8d:
// "measure; do not assume an issue rate").
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int ITERS = 16384;
constexpr int UNROLL = 8;   // independent chains per thread

__global__ void k_mad_u64_u32(uint64_t* out, uint32_t a, uint32_t b) {
  uint64_t acc[UNROLL];
  uint32_t x = a + threadIdx.x, y = b + blockIdx.x;
  for (int j = 0; j < UNROLL; ++j) acc[j] = j + x;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int j = 0; j < UNROLL; ++j)
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[j]) : "v"(x), "v"(y) : "vcc");
  }
  uint64_t s = 0;
  for (int j = 0; j < UNROLL; ++j) s += acc[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_mul_lo(uint64_t* out, uint32_t a, uint32_t b) {
  uint32_t acc[UNROLL];
  uint32_t y = b + blockIdx.x;
  for (int j = 0; j < UNROLL; ++j) acc[j] = j + a + threadIdx.x;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int j = 0; j < UNROLL; ++j)
      asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(acc[j]) : "v"(y));
  }
  uint64_t s = 0;
  for (int j = 0; j < UNROLL; ++j) s += acc[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_mul_hi(uint64_t* out, uint32_t a, uint32_t b) {
  uint32_t acc[UNROLL];
  uint32_t y = b + blockIdx.x;
  for (int j = 0; j < UNROLL; ++j) acc[j] = j + a + threadIdx.x;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int j = 0; j < UNROLL; ++j)
      asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(acc[j]) : "v"(y));
  }
  uint64_t s = 0;
  for (int j = 0; j < UNROLL; ++j) s += acc[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_mad_u32_u24(uint64_t* out, uint32_t a, uint32_t b) {
  uint32_t acc[UNROLL];
  uint32_t y = b + blockIdx.x;
  for (int j = 0; j < UNROLL; ++j) acc[j] = j + a + threadIdx.x;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int j = 0; j < UNROLL; ++j)
      asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(acc[j]) : "v"(y));
  }
  uint64_t s = 0;
  for (int j = 0; j < UNROLL; ++j) s += acc[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_add_co(uint64_t* out, uint32_t a, uint32_t b) {
  uint32_t acc[UNROLL];
  uint32_t y = b + blockIdx.x;
  for (int j = 0; j < UNROLL; ++j) acc[j] = j + a + threadIdx.x;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int j = 0; j < UNROLL; ++j)
      asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(acc[j]) : "v"(y) : "vcc");
  }
  uint64_t s = 0;
  for (int j = 0; j < UNROLL; ++j) s += acc[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_add3(uint64_t* out, uint32_t a, uint32_t b) {
  uint32_t acc[UNROLL];
  uint32_t y = b + blockIdx.x;
  for (int j = 0; j < UNROLL; ++j) acc[j] = j + a + threadIdx.x;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int j = 0; j < UNROLL; ++j)
      asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(acc[j]) : "v"(y));
  }
  uint64_t s = 0;
  for (int j = 0; j < UNROLL; ++j) s += acc[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_fma_f64(uint64_t* out, uint32_t a, uint32_t b) {
  double acc[UNROLL];
  double x = 1.0 + 1e-9 * (a + threadIdx.x), y = 1e-12 * (b + blockIdx.x);
  for (int j = 0; j < UNROLL; ++j) acc[j] = j + x;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int j = 0; j < UNROLL; ++j)
      asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[j]) : "v"(x), "v"(y));
  }
  double s = 0;
  for (int j = 0; j < UNROLL; ++j) s += acc[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)s;
}

__global__ void k_fma_f32(uint64_t* out, uint32_t a, uint32_t b) {
  float acc[UNROLL];
  float x = 1.0f + 1e-6f * (a + threadIdx.x), y = 1e-6f * (b + blockIdx.x);
  for (int j = 0; j < UNROLL; ++j) acc[j] = j + x;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int j = 0; j < UNROLL; ++j)
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[j]) : "v"(x), "v"(y));
  }
  float s = 0;
  for (int j = 0; j < UNROLL; ++j) s += acc[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)s;
}

#define K32(NAME, ASM)                                                         \
__global__ void NAME(uint64_t* out, uint32_t a, uint32_t b) {                  \
  uint32_t acc[UNROLL];                                                        \
  uint32_t y = b + blockIdx.x;                                                 \
  for (int j = 0; j < UNROLL; ++j) acc[j] = j + a + threadIdx.x;               \
  for (int i = 0; i < ITERS; ++i) {                                            \
    _Pragma("unroll") for (int j = 0; j < UNROLL; ++j)                         \
      asm volatile(ASM : "+v"(acc[j]) : "v"(y));                               \
  }                                                                            \
  uint64_t s = 0;                                                              \
  for (int j = 0; j < UNROLL; ++j) s += acc[j];                                \
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;                              \
}
#define K64(NAME, ASM)                                                         \
__global__ void NAME(uint64_t* out, uint32_t a, uint32_t b) {                  \
  uint64_t acc[UNROLL];                                                        \
  uint64_t y = b + blockIdx.x;                                                 \
  for (int j = 0; j < UNROLL; ++j) acc[j] = j + a + threadIdx.x;               \
  for (int i = 0; i < ITERS; ++i) {                                            \
    _Pragma("unroll") for (int j = 0; j < UNROLL; ++j)                         \
      asm volatile(ASM : "+v"(acc[j]) : "v"(y));                               \
  }                                                                            \
  uint64_t s = 0;                                                              \
  for (int j = 0; j < UNROLL; ++j) s += acc[j];                                \
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;                              \
}
K32(k_and, "v_and_b32 %0, %0, %1")
K32(k_alignbit, "v_alignbit_b32 %0, %0, %1, 29")
K32(k_lshl_add_u32, "v_lshl_add_u32 %0, %0, 2, %1")
K32(k_xad, "v_xad_u32 %0, %0, %1, %1")
K32(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
K32(k_sub, "v_sub_u32 %0, %0, %1")
K32(k_lshrrev_b32, "v_lshrrev_b32 %0, 3, %0")
K32(k_bfe, "v_bfe_u32 %0, %0, 3, 20")
K32(k_mov, "v_mov_b32 %0, %1")
K64(k_lshrrev_b64, "v_lshrrev_b64 %0, 29, %0")
K64(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 0, %1")
K64(k_mov_b64, "v_mov_b64 %0, %1")

template <typename K>
static double run(const char* name, K kern, uint64_t* d_out, int blocks, int threads) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d_out, 3u, 5u);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d_out, 3u, 5u);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  double lane_ops = (double)blocks * threads * ITERS * UNROLL;
  double rate = lane_ops / (best * 1e-3);
  printf("{\"instr\": \"%s\", \"ms\": %.4f, \"lane_ops_per_s\": %.4e, \"blocks\": %d, \"threads\": %d}\n",
         name, best, rate, blocks, threads);
  return rate;
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_khz\": %d}\n", prop.name, cus, prop.clockRate);
  int threads = 256;
  for (int wpc : {8, 16}) {           // waves per CU
    int blocks = cus * wpc / 4;
    uint64_t* d_out; CK(hipMalloc(&d_out, (size_t)blocks * threads * 8));
    printf("# waves/CU = %d\n", wpc);
    run("v_mad_u64_u32", k_mad_u64_u32, d_out, blocks, threads);
    run("v_mul_lo_u32", k_mul_lo, d_out, blocks, threads);
    run("v_mul_hi_u32", k_mul_hi, d_out, blocks, threads);
    run("v_mad_u32_u24", k_mad_u32_u24, d_out, blocks, threads);
    run("v_addc_co_u32", k_add_co, d_out, blocks, threads);
    run("v_add3_u32", k_add3, d_out, blocks, threads);
    run("v_and_b32", k_and, d_out, blocks, threads);
    run("v_alignbit_b32", k_alignbit, d_out, blocks, threads);
    run("v_lshl_add_u32", k_lshl_add_u32, d_out, blocks, threads);
    run("v_xad_u32", k_xad, d_out, blocks, threads);
    run("v_cndmask_b32", k_cndmask, d_out, blocks, threads);
    run("v_sub_u32", k_sub, d_out, blocks, threads);
    run("v_lshrrev_b32", k_lshrrev_b32, d_out, blocks, threads);
    run("v_bfe_u32", k_bfe, d_out, blocks, threads);
    run("v_mov_b32", k_mov, d_out, blocks, threads);
    run("v_lshrrev_b64", k_lshrrev_b64, d_out, blocks, threads);
    run("v_lshl_add_u64", k_lshl_add_u64, d_out, blocks, threads);
    run("v_mov_b64", k_mov_b64, d_out, blocks, threads);
    run("v_fma_f64", k_fma_f64, d_out, blocks, threads);
    run("v_fma_f32", k_fma_f32, d_out, blocks, threads);
    CK(hipFree(d_out));
  }
  return 0;
}
