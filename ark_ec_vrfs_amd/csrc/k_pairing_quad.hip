// k_pairing_quad.hip -- BLS12-381 pairing-product check, one item per DPP quad (bls12_quad.cuh); SURVEY.md
// section 8 row a11, `ring::Verifier::verify` tail (/root/reference src/lib.rs:14 `ring`).  A translation unit
// of its own: the out-of-line tower functions it calls inherit the kernels' register budget only if no differently
// constrained kernel shares them.  The check kernels declare ONE wave per SIMD (amdgpu_waves_per_eu(1, 1)): the tower
// needs more than 256 live registers either way, and with the whole 512-entry file the overflow goes to the accumulation
// registers instead of scratch memory -- measured 12.3 -> 11.7 ms (per item) and 9.8 -> 9.3 ms (prepared lines) at 2^14,
// the same 4 % up to 2^17 items (two waves per SIMD never paid: the time is linear in the batch from 2^14 on).
#include "kernels.h"
#include "bls12.cuh"
#include "bls12_quad.cuh"

namespace vrf {

constexpr int PAIR_BLOCK = 128;      // 32 items per workgroup

__global__ void __launch_bounds__(PAIR_BLOCK) __attribute__((amdgpu_waves_per_eu(1, 1))) k_pairing_check2_quad(size_t n, const uint8_t* g1, const uint8_t* g2,
                                                                     size_t g2_stride, uint8_t* status) {
  const size_t lane = (size_t)blockIdx.x * PAIR_BLOCK + threadIdx.x;
  const size_t item = lane >> 2;
  const int q = (int)(lane & 3);
  if (item >= n) return;                       // whole quads leave together
  const uint32_t st = bls::pairing_check2_quad(reinterpret_cast<const uint32_t*>(g1 + item * 192),
                                               reinterpret_cast<const uint32_t*>(g2 + item * g2_stride), q);
  if (q == 0) status[item] = (uint8_t)st;
}

// Test-only: the quad tower operations against the one-lane operations of bls12.cuh on the same operands.
// in: n x 2 x 12 field elements of 48 bytes (little-endian, reduced mod p by the loader); status[i] = bit mask
// of the operations whose results differ (0 = all equal): 1 mul, 2 sqr, 4 cyclotomic sqr, 8 mul_by_014,
// 16 frobenius, 32 conj/inverse/gather-scatter round trip.
__global__ void __launch_bounds__(PAIR_BLOCK) k_pairing_quad_selftest(size_t n, const uint8_t* in, uint8_t* status) {
  using namespace bls;
  const size_t lane = (size_t)blockIdx.x * PAIR_BLOCK + threadIdx.x;
  const size_t item = lane >> 2;
  const int q = (int)(lane & 3);
  if (item >= n) return;
  const uint32_t* w = reinterpret_cast<const uint32_t*>(in + item * 2 * 576);
  Fp12 x, y;
  Fp2* xs[6] = {&x.c0.c0, &x.c0.c1, &x.c0.c2, &x.c1.c0, &x.c1.c1, &x.c1.c2};
  Fp2* ys[6] = {&y.c0.c0, &y.c0.c1, &y.c0.c2, &y.c1.c0, &y.c1.c1, &y.c1.c2};
  for (int k = 0; k < 6; ++k) {
    fp_from_words(xs[k]->a, w + 24 * k); fp_from_words(xs[k]->b, w + 24 * k + 12);
    fp_from_words(ys[k]->a, w + 144 + 24 * k); fp_from_words(ys[k]->b, w + 144 + 24 * k + 12);
  }
  const Q12 xq = q12_scatter(&x, q), yq = q12_scatter(&y, q);
  auto same = [](const Fp12* a, const Fp12* b) {
    return fp2_eq(a->c0.c0, b->c0.c0) && fp2_eq(a->c0.c1, b->c0.c1) && fp2_eq(a->c0.c2, b->c0.c2) &&
           fp2_eq(a->c1.c0, b->c1.c0) && fp2_eq(a->c1.c1, b->c1.c1) && fp2_eq(a->c1.c2, b->c1.c2);
  };
  uint32_t bad = 0;
  Fp12 ref, got;
  fp12_mul(&ref, &x, &y);
  q12_gather(&got, fp12_mul_q(xq, yq, q));
  if (!same(&ref, &got)) bad |= 1;
  fp12_sqr(&ref, &x);
  q12_gather(&got, fp12_sqr_q(xq, q));
  if (!same(&ref, &got)) bad |= 2;
  fp12_cyclotomic_sqr(&ref, &x);
  q12_gather(&got, fp12_cyclotomic_sqr_q(xq, q));
  if (!same(&ref, &got)) bad |= 4;
  ref = x;
  fp12_mul_by_014(&ref, &y.c0.c0, &y.c0.c1, &y.c1.c2);
  q12_gather(&got, fp12_mul_by_014_q(xq, y.c0.c0, y.c0.c1, y.c1.c2, q));
  if (!same(&ref, &got)) bad |= 8;
  fp12_frob(&ref, &x);
  q12_gather(&got, fp12_frob_q(xq, q));
  if (!same(&ref, &got)) bad |= 16;
  fp12_conj(&ref, &x);
  q12_gather(&got, fp12_conj_q(xq));
  if (!same(&ref, &got)) bad |= 32;
  q12_gather(&got, xq);
  if (!same(&x, &got)) bad |= 32;
  // all four lanes hold the same verdict; OR them anyway so that a lane-dependent slip shows
  bad |= (uint32_t)qperm_i32<QP_BC1>((int)bad) | (uint32_t)qperm_i32<QP_BC2>((int)bad);
  if (q == 0) status[item] = (uint8_t)bad;
}

__global__ void __launch_bounds__(PAIR_BLOCK) __attribute__((amdgpu_waves_per_eu(1, 1))) k_pairing_check2_quad_prepared(size_t n, const uint8_t* g1, const uint32_t* prep,
                                                                              uint8_t* status) {
  const size_t lane = (size_t)blockIdx.x * PAIR_BLOCK + threadIdx.x;
  const size_t item = lane >> 2;
  const int q = (int)(lane & 3);
  if (item >= n) return;
  const uint32_t st = bls::pairing_check2_quad_prepared(reinterpret_cast<const uint32_t*>(g1 + item * 192), prep, q);
  if (q == 0) status[item] = (uint8_t)st;
}

void launch_pairing_check2_quad_prepared(size_t n, const uint8_t* g1, const uint32_t* prep, uint8_t* status, hipStream_t st) {
  if (!n) return;
  const size_t lanes = 4 * n;
  hipLaunchKernelGGL(k_pairing_check2_quad_prepared, dim3((unsigned)((lanes + PAIR_BLOCK - 1) / PAIR_BLOCK)), dim3(PAIR_BLOCK),
                     0, st, n, g1, prep, status);
}

void launch_pairing_check2_quad(size_t n, const uint8_t* g1, const uint8_t* g2, size_t g2_stride, uint8_t* status,
                                hipStream_t st) {
  if (!n) return;
  const size_t lanes = 4 * n;
  hipLaunchKernelGGL(k_pairing_check2_quad, dim3((unsigned)((lanes + PAIR_BLOCK - 1) / PAIR_BLOCK)), dim3(PAIR_BLOCK), 0,
                     st, n, g1, g2, g2_stride, status);
}

void launch_pairing_quad_selftest(size_t n, const uint8_t* in, uint8_t* status, hipStream_t st) {
  if (!n) return;
  const size_t lanes = 4 * n;
  hipLaunchKernelGGL(k_pairing_quad_selftest, dim3((unsigned)((lanes + PAIR_BLOCK - 1) / PAIR_BLOCK)), dim3(PAIR_BLOCK),
                     0, st, n, in, status);
}

}  // namespace vrf
