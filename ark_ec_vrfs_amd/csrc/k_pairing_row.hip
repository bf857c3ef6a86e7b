// k_pairing_row.hip -- BLS12-381 pairing-product check with ONE ITEM PER 16-LANE ROW (bls12_row.cuh): the low-latency
// form of k_pairing_check2_quad_prepared for the handful of checks that decide a batch (vrfhip_pairing_check_batch_rlc).
// SURVEY.md section 8 rows a11 / f3, `ring::Verifier::verify` tail (/root/reference src/lib.rs:14 `ring`).
#include "kernels.h"
#include "bls12.cuh"
#include "bls12_row.cuh"

namespace vrf {

constexpr int ROW_BLOCK = 64;        // 4 items per wave

// TRI = false: one item per 16-lane row, 4 items per wave.  TRI = true: one item per wave, rows 0..2 each form one of the
// three Fp products of every Fp2 product (bls12_row.cuh row_fp2_mul); row 3 leaves at once.
template <bool TRI>
__global__ void __launch_bounds__(ROW_BLOCK) __attribute__((amdgpu_waves_per_eu(1, 1))) k_pairing_check2_row_prepared(size_t n, const uint8_t* g1, const uint32_t* prep,
                                                                            uint8_t* status) {
  const size_t lane = (size_t)blockIdx.x * ROW_BLOCK + threadIdx.x;
  const size_t item = TRI ? (size_t)blockIdx.x : lane >> 4;
  if (item >= n || (TRI && threadIdx.x >= 48)) return;       // whole rows leave together
  const bls::RowCtx c = bls::row_ctx((int)(threadIdx.x & 63));
  const uint32_t st = bls::pairing_check2_row_prepared<TRI>(reinterpret_cast<const uint32_t*>(g1 + item * 192), prep, c);
  if ((threadIdx.x & (TRI ? 63 : 15)) == 0) status[item] = (uint8_t)st;
}

// Test-only: the row tower operations against the one-lane operations of bls12.cuh; ORs into status[i] (format of
// k_pairing_quad_selftest): 64 = product / squaring / sparse product differ, 128 = cyclotomic squaring / Frobenius differ.
template <bool TRI>
__global__ void __launch_bounds__(ROW_BLOCK) k_pairing_row_selftest(size_t n, const uint8_t* in, uint8_t* status) {
  using namespace bls;
  const size_t lane = (size_t)blockIdx.x * ROW_BLOCK + threadIdx.x;
  const size_t item = TRI ? (size_t)blockIdx.x : lane >> 4;
  if (item >= n || (TRI && threadIdx.x >= 48)) return;
  const RowCtx c = row_ctx((int)(threadIdx.x & 63));
  const int q = c.q;
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
  q12_gather(&got, fp12_mul_row<TRI>(xq, yq, c));
  if (!same(&ref, &got)) bad |= 64;
  fp12_sqr(&ref, &x);
  q12_gather(&got, fp12_sqr_row<TRI>(xq, c));
  if (!same(&ref, &got)) bad |= 64;
  ref = x;
  fp12_mul_by_014(&ref, &y.c0.c0, &y.c0.c1, &y.c1.c2);
  q12_gather(&got, fp12_mul_by_014_row<TRI>(xq, y.c0.c0, y.c0.c1, y.c1.c2, c));
  if (!same(&ref, &got)) bad |= 64;
  fp12_cyclotomic_sqr(&ref, &x);
  q12_gather(&got, fp12_cyclotomic_sqr_row<TRI>(xq, c));
  if (!same(&ref, &got)) bad |= 128;
  fp12_frob(&ref, &x);
  q12_gather(&got, fp12_frob_row<TRI>(xq, c));
  if (!same(&ref, &got)) bad |= 128;
  // every lane of the row must agree: OR the verdicts of the three quads (a quad-dependent slip shows)
  bad |= (uint32_t)__builtin_amdgcn_ds_bpermute(c.src[0], (int)bad) | (uint32_t)__builtin_amdgcn_ds_bpermute(c.src[1], (int)bad) |
         (uint32_t)__builtin_amdgcn_ds_bpermute(c.src[2], (int)bad);
  bad |= (uint32_t)qperm_i32<QP_BC1>((int)bad) | (uint32_t)qperm_i32<QP_BC2>((int)bad);
  if (TRI) {
    // the three rows must agree as well
    bad |= (uint32_t)__builtin_amdgcn_ds_bpermute(c.srck[0], (int)bad) | (uint32_t)__builtin_amdgcn_ds_bpermute(c.srck[1], (int)bad) |
           (uint32_t)__builtin_amdgcn_ds_bpermute(c.srck[2], (int)bad);
  }
  if ((threadIdx.x & (TRI ? 63 : 15)) == 0) status[item] |= (uint8_t)bad;
}

// Up to this many checks run one item per WAVE (three rows: Fp2 products split over the rows too); then one item per row.
constexpr size_t PAIRING_TRI_MAX_ITEMS = 1024;   // measured: 4.1-4.3 ms up to 1024 items (one wave each), 8.6 ms at 2048; rows 5.3 ms
// want_tri: 1 one item per wave, 0 one item per row, -1 by batch size
void launch_pairing_check2_row_prepared(size_t n, const uint8_t* g1, const uint32_t* prep, uint8_t* status, hipStream_t st, int want_tri) {
  if (!n) return;
  const bool tri = want_tri >= 0 ? want_tri != 0 : n <= PAIRING_TRI_MAX_ITEMS;
  if (tri) {
    hipLaunchKernelGGL(k_pairing_check2_row_prepared<true>, dim3((unsigned)n), dim3(ROW_BLOCK), 0, st, n, g1, prep, status);
    return;
  }
  const size_t lanes = 16 * n;
  hipLaunchKernelGGL(k_pairing_check2_row_prepared<false>, dim3((unsigned)((lanes + ROW_BLOCK - 1) / ROW_BLOCK)), dim3(ROW_BLOCK), 0,
                     st, n, g1, prep, status);
}

void launch_pairing_row_selftest(size_t n, const uint8_t* in, uint8_t* status, hipStream_t st) {
  if (!n) return;
  const size_t lanes = 16 * n;
  hipLaunchKernelGGL(k_pairing_row_selftest<false>, dim3((unsigned)((lanes + ROW_BLOCK - 1) / ROW_BLOCK)), dim3(ROW_BLOCK), 0, st,
                     n, in, status);
  hipLaunchKernelGGL(k_pairing_row_selftest<true>, dim3((unsigned)n), dim3(ROW_BLOCK), 0, st, n, in, status);
}

}  // namespace vrf
