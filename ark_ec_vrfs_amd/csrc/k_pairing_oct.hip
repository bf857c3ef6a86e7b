// k_pairing_oct.hip -- BLS12-381 pairing-product check, one item per 8 lanes (bls12_oct.cuh); SURVEY.md section 8 row a11,
// `ring::Verifier::verify` tail (/root/reference src/lib.rs:14 `ring`).  The throughput path of
// vrfhip_pairing_check_batch from 2^13 items on: an Fp2 is split over a lane pair, an Fp12 is 28 registers per lane, the
// kernels are declared TWO waves per SIMD (256 registers) so that BASELINE.json's 2^14 items -- 2048 waves -- are resident
// at once and the plain 32-bit instructions of one wave fill the issue slots the other's 64-bit multiply-adds leave.
#include "kernels.h"
#include "bls12.cuh"
#include "bls12_oct.cuh"

namespace vrf {

constexpr int OCT_BLOCK = 128;       // 16 items per workgroup

__global__ void __launch_bounds__(OCT_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_pairing_check2_oct(size_t n, const uint8_t* g1, const uint8_t* g2, size_t g2_stride, uint8_t* status) {
  const size_t lane = (size_t)blockIdx.x * OCT_BLOCK + threadIdx.x;
  const size_t item = lane >> 3;
  if (item >= n) return;                       // whole items leave together
  bls::oct::Ln ln;
  bls::oct::lanes_of(ln, (int)(threadIdx.x & 63));
  const uint32_t st = bls::oct::pairing_check2_oct(reinterpret_cast<const uint32_t*>(g1 + item * 192),
                                                   reinterpret_cast<const uint32_t*>(g2 + item * g2_stride), ln);
  if ((lane & 7) == 0) status[item] = (uint8_t)st;
}

__global__ void __launch_bounds__(OCT_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_pairing_check2_oct_prepared(size_t n, const uint8_t* g1, const uint32_t* prep, uint8_t* status) {
  const size_t lane = (size_t)blockIdx.x * OCT_BLOCK + threadIdx.x;
  const size_t item = lane >> 3;
  if (item >= n) return;
  bls::oct::Ln ln;
  bls::oct::lanes_of(ln, (int)(threadIdx.x & 63));
  const uint32_t st = bls::oct::pairing_check2_oct_prepared(reinterpret_cast<const uint32_t*>(g1 + item * 192), prep, ln);
  if ((lane & 7) == 0) status[item] = (uint8_t)st;
}

// The per-item path in two kernels (bls12_oct.cuh): lines of both pairs -> HBM, then the prepared-lines Miller loop over them
__global__ void __launch_bounds__(OCT_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_pairing_lines_oct(size_t n, const uint8_t* g1, const uint8_t* g2, size_t g2_stride, uint32_t* lines, uint32_t* item_flags) {
  const size_t lane = (size_t)blockIdx.x * OCT_BLOCK + threadIdx.x;
  const size_t item = lane >> 3;
  if (item >= n) return;
  bls::oct::Ln ln;
  bls::oct::lanes_of(ln, (int)(threadIdx.x & 63));
  bls::oct::pairing_lines_oct(reinterpret_cast<const uint32_t*>(g1 + item * 192), reinterpret_cast<const uint32_t*>(g2 + item * g2_stride),
                              lines, item_flags, n, item, ln);
}
__global__ void __launch_bounds__(OCT_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_pairing_check2_oct_lines(size_t n, const uint32_t* lines, const uint32_t* item_flags, uint8_t* status) {
  const size_t lane = (size_t)blockIdx.x * OCT_BLOCK + threadIdx.x;
  const size_t item = lane >> 3;
  if (item >= n) return;
  bls::oct::Ln ln;
  bls::oct::lanes_of(ln, (int)(threadIdx.x & 63));
  const uint32_t st = bls::oct::pairing_check2_oct_lines(lines, item_flags, n, item, ln);
  if ((lane & 7) == 0) status[item] = (uint8_t)st;
}

// Test-only: the oct tower operations against the one-lane operations of bls12.cuh on the same operands, and the
// cross-lane moves themselves.  in: n x 2 x 12 field elements of 48 bytes (little-endian, reduced mod p by the loader);
// status[i] = bit mask of what differs (0 = all equal): 1 mul, 2 sqr, 4 cyclotomic sqr (of x^((p^6-1)(p^2+1))),
// 8 mul_by_014, 16 frobenius, 32 inverse, 64 conj / scatter round trip, 128 a cross-lane move.
__global__ void __launch_bounds__(OCT_BLOCK) k_pairing_oct_selftest(size_t n, const uint8_t* in, uint8_t* status) {
  using namespace bls;
  using namespace bls::oct;
  const size_t lane = (size_t)blockIdx.x * OCT_BLOCK + threadIdx.x;
  const size_t item = lane >> 3;
  if (item >= n) return;
  Ln ln;
  lanes_of(ln, (int)(threadIdx.x & 63));
  uint32_t bad = 0;
  {
    const int me = (int)(threadIdx.x & 63);
    if (xp_i32(me) != (me ^ 4)) bad |= 128;
    if (xq_i32<QP_ROT1>(me) != (ln.j < 3 ? (me & ~3) + (ln.j + 1) % 3 : me)) bad |= 128;
    if (xq_i32<QP_PAIRSWAP>(me) != (me ^ 1)) bad |= 128;
    FpS a, k;
    for (int i = 0; i < NLB; ++i) { a.v[i] = 100 * me + i; k.v[i] = -7; }
    const FpS u = xp_h1(a), w = xp_h0(k, a);
    for (int i = 0; i < NLB; ++i) {
      if (u.v[i] != 100 * (ln.h ? (me ^ 4) : me) + i) bad |= 128;
      if (w.v[i] != (ln.h ? -7 : 100 * (me ^ 4) + i)) bad |= 128;
    }
    if (!x_all8(true, ln) || x_all8((me & 7) != 5, ln)) bad |= 128;
  }
  const uint32_t* w = reinterpret_cast<const uint32_t*>(in + item * 2 * 576);
  Fp12 x, y;
  Fp2* xs[6] = {&x.c0.c0, &x.c0.c1, &x.c0.c2, &x.c1.c0, &x.c1.c1, &x.c1.c2};
  Fp2* ys[6] = {&y.c0.c0, &y.c0.c1, &y.c0.c2, &y.c1.c0, &y.c1.c1, &y.c1.c2};
  for (int k = 0; k < 6; ++k) {
    fp_from_words(xs[k]->a, w + 24 * k); fp_from_words(xs[k]->b, w + 24 * k + 12);
    fp_from_words(ys[k]->a, w + 144 + 24 * k); fp_from_words(ys[k]->b, w + 144 + 24 * k + 12);
  }
  const O12 xo = o12_scatter(&x, ln), yo = o12_scatter(&y, ln);
  Fp12 ref;
  fp12_mul(&ref, &x, &y);
  if (!o12_same(o12_mul(xo, yo, ln), &ref, ln)) bad |= 1;
  fp12_sqr(&ref, &x);
  if (!o12_same(o12_sqr(xo, ln), &ref, ln)) bad |= 2;
  ref = x;
  fp12_mul_by_014(&ref, &y.c0.c0, &y.c0.c1, &y.c1.c2);
  {
    const FpS l0 = ln.h ? y.c0.c0.b : y.c0.c0.a, l1 = ln.h ? y.c0.c1.b : y.c0.c1.a, l4 = ln.h ? y.c1.c2.b : y.c1.c2.a;
    if (!o12_same(o12_mul_by_014<true>(xo, l0, l1, l4, ln), &ref, ln)) bad |= 8;
    if (!o12_same(o12_mul_by_014<false>(xo, l0, l1, l4, ln), &ref, ln)) bad |= 8;
  }
  fp12_frob(&ref, &x);
  if (!o12_same(o12_frob(xo, ln), &ref, ln)) bad |= 16;
  fp12_inv(&ref, &x);
  const O12 xi = o12_inv(xo, ln);
  if (!o12_same(xi, &ref, ln)) bad |= 32;
  {
    // x^((p^6 - 1)(p^2 + 1)) lies in the cyclotomic subgroup: formed in the oct layout, squared both ways
    const O12 t = o12_mul(o12_conj(xo), xi, ln);
    const O12 c = o12_mul(o12_frob(o12_frob(t, ln), ln), t, ln);
    Fp12 t0, t1, t2, xc;
    fp12_conj(&t0, &x); fp12_mul(&t2, &t0, &ref);
    fp12_frob(&t0, &t2); fp12_frob(&t1, &t0); fp12_mul(&xc, &t1, &t2);
    if (!o12_same(c, &xc, ln)) bad |= 64;
    fp12_cyclotomic_sqr(&ref, &xc);
    if (!o12_same(o12_cyclotomic_sqr(c, ln), &ref, ln)) bad |= 4;
  }
  fp12_conj(&ref, &x);
  if (!o12_same(o12_conj(xo), &ref, ln) || !o12_same(xo, &x, ln)) bad |= 64;
  // every lane of the item holds the combined verdicts of the comparisons; OR the move checks across the item
  const unsigned long long mv = __builtin_amdgcn_ballot_w64((bad & 128) != 0);
  if ((mv >> ln.base) & 0xffull) bad |= 128;
  if ((lane & 7) == 0) status[item] = (uint8_t)bad;
}

void launch_pairing_check2_oct(size_t n, const uint8_t* g1, const uint8_t* g2, size_t g2_stride, uint8_t* status, hipStream_t st) {
  if (!n) return;
  const size_t lanes = 8 * n;
  hipLaunchKernelGGL(k_pairing_check2_oct, dim3((unsigned)((lanes + OCT_BLOCK - 1) / OCT_BLOCK)), dim3(OCT_BLOCK), 0, st, n, g1, g2,
                     g2_stride, status);
}

// per-item G2 points through the two kernels above; ws: pairing_oct_lines_bytes(chunk) of device memory, chunk = the number
// of items one pass may hold (the batch is walked in chunks of that size)
size_t pairing_oct_lines_bytes(size_t items) { return items * (bls::oct::oct_lines_words_per_item() + 4) * sizeof(uint32_t); }
void launch_pairing_check2_oct_split(size_t n, const uint8_t* g1, const uint8_t* g2, size_t g2_stride, uint8_t* status, void* ws,
                                     size_t chunk, hipStream_t st) {
  for (size_t base = 0; base < n; base += chunk) {
    const size_t m = n - base < chunk ? n - base : chunk;
    uint32_t* lines = static_cast<uint32_t*>(ws);
    uint32_t* flags = lines + m * bls::oct::oct_lines_words_per_item();
    const unsigned blocks = (unsigned)((8 * m + OCT_BLOCK - 1) / OCT_BLOCK);
    hipLaunchKernelGGL(k_pairing_lines_oct, dim3(blocks), dim3(OCT_BLOCK), 0, st, m, g1 + base * 192, g2 + base * g2_stride, g2_stride,
                       lines, flags);
    hipLaunchKernelGGL(k_pairing_check2_oct_lines, dim3(blocks), dim3(OCT_BLOCK), 0, st, m, lines, flags, status + base);
  }
}

void launch_pairing_check2_oct_prepared(size_t n, const uint8_t* g1, const uint32_t* prep, uint8_t* status, hipStream_t st) {
  if (!n) return;
  const size_t lanes = 8 * n;
  hipLaunchKernelGGL(k_pairing_check2_oct_prepared, dim3((unsigned)((lanes + OCT_BLOCK - 1) / OCT_BLOCK)), dim3(OCT_BLOCK), 0, st, n,
                     g1, prep, status);
}

void launch_pairing_oct_selftest(size_t n, const uint8_t* in, uint8_t* status, hipStream_t st) {
  if (!n) return;
  const size_t lanes = 8 * n;
  hipLaunchKernelGGL(k_pairing_oct_selftest, dim3((unsigned)((lanes + OCT_BLOCK - 1) / OCT_BLOCK)), dim3(OCT_BLOCK), 0, st, n, in,
                     status);
}

}  // namespace vrf
