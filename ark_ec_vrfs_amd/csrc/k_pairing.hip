// k_pairing.hip -- BLS12-381 pairing-product check kernels:
// e(P0, Q0) * e(P1, Q1) == 1 per item, the KZG equation at the end of `ring::Verifier::verify`
// (/root/reference src/lib.rs:14 `ring`).
//   k_pairing_check2_quad (k_pairing_quad.hip): one item per DPP quad, tower arithmetic in registers -- the
//                           path vrfhip_pairing_check_batch runs.
//   k_pairing_check2_oct  (k_pairing_oct.hip): one item per 8 lanes, Fp2 split over lane pairs -- the throughput path.
//   k_pairing_check2      : one item per lane (bls12.cuh); kept as the cross-check of the distributed arithmetic
//                           (layout PAIRING_LANE, reachable through vrfhip_debug_set only).
#include "kernels.h"
#include "bls12.cuh"

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

// shared G2 points: the 2 x 68 Miller-loop lines, one lane per pair (the one-lane tower: ~3 ms of latency, so the
// buffer also keeps the 96 words of the points its lines belong to and a valid word -- a verifier's SRS does
// not change between calls, and a call with the same points finds the lines in place)
__global__ void __launch_bounds__(64) k_pairing_prepare_g2(const uint8_t* g2, uint32_t* prep) {
  if (blockIdx.x != 0) return;
  const uint32_t* p2 = reinterpret_cast<const uint32_t*>(g2);
  uint32_t* key = prep + bls::G2_PREP_WORDS;
  const int t = (int)threadIdx.x;
  const bool same = key[96] == 1u && key[t] == p2[t] && (t >= 32 || key[64 + t] == p2[64 + t]);
  if (__all(same ? 1 : 0)) return;
  if (t < 2) {
    uint32_t w2[96];
    for (int k = 0; k < 96; ++k) w2[k] = p2[k];
    bls::pairing_prepare_g2_pair(w2, prep, t);
  }
  __threadfence();
  key[t] = p2[t];
  if (t < 32) key[64 + t] = p2[64 + t];
  if (t == 0) key[96] = 1u;
}

size_t pairing_prep_bytes() { return (size_t)(bls::G2_PREP_WORDS + 96 + 1) * sizeof(uint32_t); }

void launch_pairing_check2_quad(size_t n, const uint8_t* g1, const uint8_t* g2, size_t g2_stride, uint8_t* status,
                                hipStream_t st);
void launch_pairing_check2_quad_prepared(size_t n, const uint8_t* g1, const uint32_t* prep, uint8_t* status, hipStream_t st);
void launch_pairing_check2_row_prepared(size_t n, const uint8_t* g1, const uint32_t* prep, uint8_t* status, hipStream_t st, int tri);
void launch_pairing_check2_oct(size_t n, const uint8_t* g1, const uint8_t* g2, size_t g2_stride, uint8_t* status, hipStream_t st);
void launch_pairing_check2_oct_prepared(size_t n, const uint8_t* g1, const uint32_t* prep, uint8_t* status, hipStream_t st);
// Layout by default: ONE ITEM PER 8 LANES at every batch size.  Measured (profiles/r04/pairing_layouts_time_small.log):
// prepared lines, 8 .. 4096 items: 3.6 ms per launch against 3.9-4.1 (one item per wave), 5.3-5.5 (per row), 9.1 (per quad);
// per-item G2 points, 8 .. 4096 items: 4.6-4.7 ms (lines kernel + Miller kernel) against 10.9 (per quad).  A single wave of
// the 8-lane kernels is a shorter chain than any of the older layouts, so the row / wave / quad kernels (rounds 1-3) are
// no longer chosen by size; they stay selectable (vrfhip_debug_set) as independent formulations for the tests.
void launch_pairing_check2(size_t n, const uint8_t* g1, const uint8_t* g2, size_t g2_stride, uint8_t* status,
                           hipStream_t st, uint32_t* prep, int layout) {
  if (!n) return;
  const int mode = layout & 0xff;
  if (mode == PAIRING_LANE) {
    hipLaunchKernelGGL(k_pairing_check2, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, n, g1, g2, g2_stride, status);
    return;
  }
  if (g2_stride == 0 && prep && !(layout & PAIRING_NOPREP)) {
    hipLaunchKernelGGL(k_pairing_prepare_g2, dim3(1), dim3(64), 0, st, g2, prep);
    if (mode == PAIRING_ROW || mode == PAIRING_TRI) launch_pairing_check2_row_prepared(n, g1, prep, status, st, mode == PAIRING_TRI);
    else if (mode == PAIRING_QUAD) launch_pairing_check2_quad_prepared(n, g1, prep, status, st);
    else launch_pairing_check2_oct_prepared(n, g1, prep, status, st);
    return;
  }
  // per-item G2 points reach this function only for the one-kernel forms (api.hip runs the default two-kernel path itself)
  if (mode == PAIRING_QUAD) launch_pairing_check2_quad(n, g1, g2, g2_stride, status, st);
  else launch_pairing_check2_oct(n, g1, g2, g2_stride, status, st);
}

}  // namespace vrf
