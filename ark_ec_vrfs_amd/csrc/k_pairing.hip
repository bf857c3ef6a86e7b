// k_pairing.hip -- BLS12-381 pairing-product check kernels:
// e(P0, Q0) * e(P1, Q1) == 1 per item, the KZG equation at the end of `ring::Verifier::verify`
// (/root/reference src/lib.rs:14 `ring`).
//   k_pairing_check2_quad (k_pairing_quad.hip): one item per DPP quad, tower arithmetic in registers -- the
//                           path vrfhip_pairing_check_batch runs.
//   k_pairing_check2      : one item per lane (bls12.cuh); kept as the cross-check of the quad arithmetic
//                           and selectable with VRFHIP_PAIRING=lane.
#include "kernels.h"
#include "bls12.cuh"
#include <cstdlib>
#include <cstring>

namespace vrf {

__global__ void __launch_bounds__(64) k_pairing_check2(size_t n, const uint8_t* g1, const uint8_t* g2,
                                                        size_t g2_stride, uint8_t* status) {
  size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  uint32_t w1[48], w2[96];
  const uint32_t* p1 = reinterpret_cast<const uint32_t*>(g1 + i * 192);
  const uint32_t* p2 = reinterpret_cast<const uint32_t*>(g2 + i * g2_stride);
  for (int k = 0; k < 48; ++k) w1[k] = p1[k];
  for (int k = 0; k < 96; ++k) w2[k] = p2[k];
  status[i] = (uint8_t)bls::pairing_check2_item(w1, w2);
}

void launch_pairing_check2_quad(size_t n, const uint8_t* g1, const uint8_t* g2, size_t g2_stride, uint8_t* status,
                                hipStream_t st);

void launch_pairing_check2(size_t n, const uint8_t* g1, const uint8_t* g2, size_t g2_stride, uint8_t* status,
                           hipStream_t st) {
  if (!n) return;
  const char* mode = getenv("VRFHIP_PAIRING");
  if (mode && !strcmp(mode, "lane")) {
    hipLaunchKernelGGL(k_pairing_check2, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, n, g1, g2, g2_stride, status);
    return;
  }
  launch_pairing_check2_quad(n, g1, g2, g2_stride, status, st);
}

}  // namespace vrf
