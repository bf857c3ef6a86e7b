// bls12.cuh -- BLS12-381 pairing check for the ring-VRF tail.
//
// Replaces ark_ec::pairing::Pairing::{multi_miller_loop, final_exponentiation} on ark-bls12-381,
// reached from /root/reference through `ring` (src/lib.rs:14): the KZG check
// e(A1, B1) * e(A2, B2) == 1 that ends `ring::Verifier::verify`.
//
// Field: Fp (381 bit) as 14 limbs x 28 bits, Montgomery radix 2^392, the same lazy-limb scheme as
// fe.cuh (64-bit column accumulators, compile-time (L, V) bounds; here L1*L2 <= 16 is allowed, so
// whole Karatsuba operand sums stay unreduced).  Tower Fp2 = Fp[u]/(u^2+1),
// Fp6 = Fp2[v]/(v^3-(1+u)), Fp12 = Fp6[w]/(w^2-v), M-type twist.  Fp12 values (168 words) do not
// fit the register file next to their operands; the tower functions are ordinary by-pointer C++
// and the compiler keeps the big values in the per-lane scratch segment (dword-interleaved across
// lanes, i.e. coalesced), while every Fp2-level operation runs in registers.
//
// Algorithm (validated against an independent affine formulation in oracle/bls_oracle.py):
// optimal ate Miller loop over |x| with homogeneous projective G2 steps and 0-1-4 sparse lines,
// conjugation for x < 0, easy part (p^6-1)(p^2+1), hard part through
// 3*(p^4-p^2+1)/r = (x-1)^2 (x+p)(x^2+p^2-1) + 3.  The result is e(P,Q)^3; the check "== 1" is
// unaffected (gcd(3, r) = 1).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "constants.gen.h"

#ifndef VRF_HD
#define VRF_HD __host__ __device__ __forceinline__
#endif
#define VRF_HD_NOINLINE static __host__ __device__ __attribute__((noinline))

namespace bls {

constexpr int NLB = 14;
constexpr int LWB = 28;
constexpr uint32_t MASKB = (1u << LWB) - 1;
constexpr int STORE_V = 12;          // storage bound: |value| < 12 p, limbs normalised (L = 1)

// Signed lazy representation: value = sum v[i] * 2^(28 i) with int32 limbs, |v[i]| < L * (2^28 + 2^12)
// and |value| < V * p (negative values allowed).  add / sub / neg are limb-wise with no bias words.
// Montgomery product: |(a*b + m*p)/R| < p * (1 + Va*Vb * p/R), p/R = 1/2521.
constexpr int mul_v(int v1, int v2) { return 1 + (v1 * v2 + 2499) / 2500; }

template <int L, int V>
struct Fp {
  static_assert(L >= 1 && L <= 7, "limb bound out of range");
  static_assert(V >= 1 && V <= 1024, "value bound out of range");
  int32_t v[NLB];
  Fp() = default;
  template <int L2, int V2>
  VRF_HD Fp(const Fp<L2, V2>& o) {
    static_assert(L2 <= L && V2 <= V, "narrowing Fp conversion");
#pragma unroll
    for (int i = 0; i < NLB; ++i) v[i] = o.v[i];
  }
};
using FpN = Fp<1, 2>;
using FpS = Fp<1, STORE_V>;

VRF_HD Fp<1, 1> fp_const(const uint32_t (&c)[NLB]) {
  Fp<1, 1> r;
#pragma unroll
  for (int i = 0; i < NLB; ++i) r.v[i] = (int32_t)c[i];
  return r;
}
VRF_HD Fp<1, 1> fp_zero() {
  Fp<1, 1> r;
#pragma unroll
  for (int i = 0; i < NLB; ++i) r.v[i] = 0;
  return r;
}
VRF_HD Fp<1, 1> fp_one() { return fp_const(vrfk::BLS_ONE_M); }

// weak normalisation: limbs 0..12 in [0, 2^28 + 8), the (signed) top limb carries the sign
template <int L, int V>
VRF_HD Fp<1, V> fp_norm(const Fp<L, V>& a) {
  Fp<1, V> r;
  r.v[0] = a.v[0] & (int32_t)MASKB;
#pragma unroll
  for (int i = 1; i < NLB - 1; ++i) r.v[i] = (a.v[i] & (int32_t)MASKB) + (a.v[i - 1] >> LWB);
  r.v[NLB - 1] = a.v[NLB - 1] + (a.v[NLB - 2] >> LWB);
  return r;
}
// limb bound of a sum: operands are normalised first when the lazy sum would not fit int32 limbs
constexpr int lsum(int l1, int l2) { return l1 + l2 <= 7 ? l1 + l2 : 2; }
template <int L1, int V1, int L2, int V2>
VRF_HD Fp<lsum(L1, L2), V1 + V2> fp_add(const Fp<L1, V1>& a, const Fp<L2, V2>& b) {
  Fp<lsum(L1, L2), V1 + V2> r;
  if constexpr (L1 + L2 <= 7) {
#pragma unroll
    for (int i = 0; i < NLB; ++i) r.v[i] = a.v[i] + b.v[i];
  } else {
    auto x = fp_norm(a);
    auto y = fp_norm(b);
#pragma unroll
    for (int i = 0; i < NLB; ++i) r.v[i] = x.v[i] + y.v[i];
  }
  return r;
}
template <int L1, int V1, int L2, int V2>
VRF_HD Fp<lsum(L1, L2), V1 + V2> fp_sub(const Fp<L1, V1>& a, const Fp<L2, V2>& b) {
  Fp<lsum(L1, L2), V1 + V2> r;
  if constexpr (L1 + L2 <= 7) {
#pragma unroll
    for (int i = 0; i < NLB; ++i) r.v[i] = a.v[i] - b.v[i];
  } else {
    auto x = fp_norm(a);
    auto y = fp_norm(b);
#pragma unroll
    for (int i = 0; i < NLB; ++i) r.v[i] = x.v[i] - y.v[i];
  }
  return r;
}
template <int L, int V>
VRF_HD Fp<L, V> fp_neg(const Fp<L, V>& a) {
  Fp<L, V> r;
#pragma unroll
  for (int i = 0; i < NLB; ++i) r.v[i] = -a.v[i];
  return r;
}
template <int L, int V>
VRF_HD Fp<lsum(L, L), 2 * V> fp_dbl(const Fp<L, V>& a) { return fp_add(a, a); }
template <int L, int V>
VRF_HD Fp<L, V> fp_select(bool c, const Fp<L, V>& a, const Fp<L, V>& b) {
  Fp<L, V> r;
#pragma unroll
  for (int i = 0; i < NLB; ++i) r.v[i] = c ? a.v[i] : b.v[i];
  return r;
}

VRF_HD int64_t smad(int32_t a, int32_t b, int64_t c) { return (int64_t)a * b + c; }

template <int L1, int V1, int L2, int V2>
VRF_HD Fp<1, mul_v(V1, V2)> fp_mul(const Fp<L1, V1>& a, const Fp<L2, V2>& b) {
  static_assert(L1 * L2 <= 8, "fp_mul: signed 64-bit column accumulator could overflow");
  Fp<1, mul_v(V1, V2)> r;
  int32_t m[NLB];
  int64_t acc = 0;
#pragma unroll
  for (int k = 0; k < NLB; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) acc = smad(a.v[i], b.v[k - i], acc);
#pragma unroll
    for (int i = 0; i < k; ++i) acc = smad(m[i], (int32_t)vrfk::BLS_P28[k - i], acc);
    m[k] = (int32_t)(((uint32_t)acc * vrfk::BLS_PINV28) & MASKB);
    acc = smad(m[k], (int32_t)vrfk::BLS_P28[0], acc);
    acc >>= LWB;
  }
#pragma unroll
  for (int k = NLB; k < 2 * NLB - 1; ++k) {
#pragma unroll
    for (int i = k - NLB + 1; i < NLB; ++i) acc = smad(a.v[i], b.v[k - i], acc);
#pragma unroll
    for (int i = k - NLB + 1; i < NLB; ++i) acc = smad(m[i], (int32_t)vrfk::BLS_P28[k - i], acc);
    r.v[k - NLB] = (int32_t)((uint32_t)acc & MASKB);
    acc >>= LWB;
  }
  r.v[NLB - 1] = (int32_t)acc;
  return r;
}
// a*b + c*d with ONE Montgomery reduction (lazy reduction across the two products of an Fp2 component: ac - bd or
// ad + bc): 2 x 196 product multiply-adds + 196 for the reduction instead of 2 x 392.  The column accumulators take two
// products per term, hence the tighter limb condition.
constexpr int mul2_v(int v1, int v2, int v3, int v4) { return 1 + (v1 * v2 + v3 * v4 + 2499) / 2500; }
template <int L1, int V1, int L2, int V2, int L3, int V3, int L4, int V4>
VRF_HD Fp<1, mul2_v(V1, V2, V3, V4)> fp_mul2(const Fp<L1, V1>& a, const Fp<L2, V2>& b, const Fp<L3, V3>& c, const Fp<L4, V4>& d) {
  static_assert(L1 * L2 + L3 * L4 <= 8, "fp_mul2: signed 64-bit column accumulator could overflow");
  Fp<1, mul2_v(V1, V2, V3, V4)> r;
  int32_t m[NLB];
  int64_t acc = 0;
#pragma unroll
  for (int k = 0; k < NLB; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) { acc = smad(a.v[i], b.v[k - i], acc); acc = smad(c.v[i], d.v[k - i], acc); }
#pragma unroll
    for (int i = 0; i < k; ++i) acc = smad(m[i], (int32_t)vrfk::BLS_P28[k - i], acc);
    m[k] = (int32_t)(((uint32_t)acc * vrfk::BLS_PINV28) & MASKB);
    acc = smad(m[k], (int32_t)vrfk::BLS_P28[0], acc);
    acc >>= LWB;
  }
#pragma unroll
  for (int k = NLB; k < 2 * NLB - 1; ++k) {
#pragma unroll
    for (int i = k - NLB + 1; i < NLB; ++i) { acc = smad(a.v[i], b.v[k - i], acc); acc = smad(c.v[i], d.v[k - i], acc); }
#pragma unroll
    for (int i = k - NLB + 1; i < NLB; ++i) acc = smad(m[i], (int32_t)vrfk::BLS_P28[k - i], acc);
    r.v[k - NLB] = (int32_t)((uint32_t)acc & MASKB);
    acc >>= LWB;
  }
  r.v[NLB - 1] = (int32_t)acc;
  return r;
}
// u1 c1 + w1 d1 + u2 c2 + w2 d2 + u3 c3 + w3 d3 with ONE Montgomery reduction: a component of a sum of three Fp2
// products (a column of a sparse Fp12 product).  Six products and the m p row fill the 64-bit column accumulators exactly
// as far as one product of limb bounds 2 x 4 does (fp_mul), so every operand must have normalised limbs.
constexpr int mul6_v(int va, int vb, int vc, int vy) { return 1 + (2 * vy * (va + vb + vc) + 2499) / 2500; }
template <int VA, int VB, int VC, int VY>
VRF_HD Fp<1, mul6_v(VA, VB, VC, VY)> fp_mul6(const Fp<1, VA>& u1, const Fp<1, VA>& w1, const Fp<1, VB>& u2, const Fp<1, VB>& w2,
                                             const Fp<1, VC>& u3, const Fp<1, VC>& w3, const Fp<1, VY>& c1, const Fp<1, VY>& d1,
                                             const Fp<1, VY>& c2, const Fp<1, VY>& d2, const Fp<1, VY>& c3, const Fp<1, VY>& d3) {
  Fp<1, mul6_v(VA, VB, VC, VY)> r;
  int32_t m[NLB];
  int64_t acc = 0;
#pragma unroll
  for (int k = 0; k < NLB; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) {
      acc = smad(u1.v[i], c1.v[k - i], acc); acc = smad(w1.v[i], d1.v[k - i], acc);
      acc = smad(u2.v[i], c2.v[k - i], acc); acc = smad(w2.v[i], d2.v[k - i], acc);
      acc = smad(u3.v[i], c3.v[k - i], acc); acc = smad(w3.v[i], d3.v[k - i], acc);
    }
#pragma unroll
    for (int i = 0; i < k; ++i) acc = smad(m[i], (int32_t)vrfk::BLS_P28[k - i], acc);
    m[k] = (int32_t)(((uint32_t)acc * vrfk::BLS_PINV28) & MASKB);
    acc = smad(m[k], (int32_t)vrfk::BLS_P28[0], acc);
    acc >>= LWB;
  }
#pragma unroll
  for (int k = NLB; k < 2 * NLB - 1; ++k) {
#pragma unroll
    for (int i = k - NLB + 1; i < NLB; ++i) {
      acc = smad(u1.v[i], c1.v[k - i], acc); acc = smad(w1.v[i], d1.v[k - i], acc);
      acc = smad(u2.v[i], c2.v[k - i], acc); acc = smad(w2.v[i], d2.v[k - i], acc);
      acc = smad(u3.v[i], c3.v[k - i], acc); acc = smad(w3.v[i], d3.v[k - i], acc);
    }
#pragma unroll
    for (int i = k - NLB + 1; i < NLB; ++i) acc = smad(m[i], (int32_t)vrfk::BLS_P28[k - i], acc);
    r.v[k - NLB] = (int32_t)((uint32_t)acc & MASKB);
    acc >>= LWB;
  }
  r.v[NLB - 1] = (int32_t)acc;
  return r;
}
template <int L, int V>
VRF_HD Fp<1, mul_v(V, V)> fp_sqr(const Fp<L, V>& a) {
  static_assert(L * L <= 8, "");
  Fp<1, mul_v(V, V)> r;
  int32_t m[NLB], a2[NLB];
#pragma unroll
  for (int i = 0; i < NLB; ++i) a2[i] = a.v[i] * 2;
  int64_t acc = 0;
#pragma unroll
  for (int k = 0; k < NLB; ++k) {
#pragma unroll
    for (int i = 0; 2 * i < k; ++i) acc = smad(a2[i], a.v[k - i], acc);
    if ((k & 1) == 0) acc = smad(a.v[k / 2], a.v[k / 2], acc);
#pragma unroll
    for (int i = 0; i < k; ++i) acc = smad(m[i], (int32_t)vrfk::BLS_P28[k - i], acc);
    m[k] = (int32_t)(((uint32_t)acc * vrfk::BLS_PINV28) & MASKB);
    acc = smad(m[k], (int32_t)vrfk::BLS_P28[0], acc);
    acc >>= LWB;
  }
#pragma unroll
  for (int k = NLB; k < 2 * NLB - 1; ++k) {
#pragma unroll
    for (int i = k - NLB + 1; 2 * i < k; ++i) acc = smad(a2[i], a.v[k - i], acc);
    if ((k & 1) == 0) acc = smad(a.v[k / 2], a.v[k / 2], acc);
#pragma unroll
    for (int i = k - NLB + 1; i < NLB; ++i) acc = smad(m[i], (int32_t)vrfk::BLS_P28[k - i], acc);
    r.v[k - NLB] = (int32_t)((uint32_t)acc & MASKB);
    acc >>= LWB;
  }
  r.v[NLB - 1] = (int32_t)acc;
  return r;
}

// cheap value reduction: x - round(x / p) * p with the quotient estimated from the top limb
// (p / 2^364 = 106513.94; RECIP = round(2^44 / 106513.94)).  |result| < 0.52 p, limbs L = 1.
constexpr int64_t RECIP44 = 165165572;   // round(2^44 / 106513.9397)
template <int L, int V>
VRF_HD Fp<1, 1> fp_reduce(const Fp<L, V>& a) {
  Fp<1, V> n = fp_norm(a);
  int32_t q = (int32_t)(((int64_t)n.v[NLB - 1] * RECIP44 + ((int64_t)1 << 43)) >> 44);
  Fp<1, 1> r;
  int64_t c = 0;
#pragma unroll
  for (int i = 0; i < NLB - 1; ++i) {
    int64_t t = smad(q, (int32_t)vrfk::BLS_P28[i], c);
    r.v[i] = n.v[i] - (int32_t)((uint32_t)t & MASKB);
    c = t >> LWB;
  }
  int64_t t = smad(q, (int32_t)vrfk::BLS_P28[NLB - 1], c);
  r.v[NLB - 1] = n.v[NLB - 1] - (int32_t)t;
  return r;
}
// bring any lazy value into storage form (L = 1, |value| < STORE_V * p)
template <int L, int V>
VRF_HD FpS fp_fit(const Fp<L, V>& a) {
  if constexpr (V <= STORE_V) {
    if constexpr (L == 1) return FpS(a);
    else return FpS(fp_norm(a));
  } else {
    return FpS(fp_reduce(a));
  }
}

// canonical representative in [0, p) of the same residue, exact 28-bit limbs
template <int L, int V>
VRF_HD void fp_canon_limbs(uint32_t out[NLB], const Fp<L, V>& a) {
  Fp<1, 1> r = fp_reduce(a);                     // |r| < 0.52 p
  int32_t x[NLB];
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < NLB - 1; ++i) {            // r + p, exact carries: value in (0.48 p, 1.52 p)
    int32_t t = r.v[i] + (int32_t)vrfk::BLS_P28[i] + c;
    x[i] = t & (int32_t)MASKB;
    c = t >> LWB;
  }
  x[NLB - 1] = r.v[NLB - 1] + (int32_t)vrfk::BLS_P28[NLB - 1] + c;
  int32_t d[NLB];
  int32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < NLB; ++i) {
    int32_t t = x[i] - (int32_t)vrfk::BLS_P28[i] - borrow;
    borrow = (t >> 31) & 1;
    d[i] = (i < NLB - 1) ? (t & (int32_t)MASKB) : t;
  }
#pragma unroll
  for (int i = 0; i < NLB; ++i) out[i] = (uint32_t)(borrow ? x[i] : d[i]);
}
template <int L, int V>
VRF_HD bool fp_is_zero(const Fp<L, V>& a) {
  uint32_t c[NLB];
  fp_canon_limbs(c, a);
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < NLB; ++i) o |= c[i];
  return o == 0;
}
template <int L1, int V1, int L2, int V2>
VRF_HD bool fp_eq(const Fp<L1, V1>& a, const Fp<L2, V2>& b) {
  return fp_is_zero(fp_sub(fp_norm(a), fp_norm(b)));
}

// 48-byte little-endian integer (12 u32 words) <-> Montgomery.  Returns false if >= p.
VRF_HD bool fp_from_words(FpS& out, const uint32_t w[12]) {
  bool lt = false, decided = false;
#pragma unroll
  for (int i = 11; i >= 0; --i) {
    uint32_t pi = vrfk::BLS_P_WORDS[i];
    if (!decided && w[i] != pi) { lt = w[i] < pi; decided = true; }
  }
  Fp<1, 8> t;                                     // 2^384 < 8 p
#pragma unroll
  for (int i = 0; i < NLB; ++i) {
    const int bit = i * LWB, j = bit >> 5, s = bit & 31;
    uint32_t lo = w[j] >> s;
    if (s > 4 && j + 1 < 12) lo |= w[j + 1] << (32 - s);
    t.v[i] = (int32_t)((i < NLB - 1) ? (lo & MASKB) : lo);
  }
  out = fp_mul(t, fp_const(vrfk::BLS_R2));
  return lt;
}
template <int L, int V>
VRF_HD void fp_to_words(uint32_t w[12], const Fp<L, V>& a) {
  Fp<1, 1> one;
#pragma unroll
  for (int i = 0; i < NLB; ++i) one.v[i] = (i == 0);
  auto t = fp_mul(fp_norm(a), one);              // a / R
  uint32_t x[NLB];
  fp_canon_limbs(x, t);
#pragma unroll
  for (int j = 0; j < 12; ++j) {
    const int bit = j * 32, i = bit / LWB, s = bit % LWB;
    uint32_t v = x[i] >> s;
    if (i + 1 < NLB) v |= x[i + 1] << (LWB - s);
    w[j] = v;
  }
}

// a^(p-2) with a 3-bit fixed window (uniform exponent => scalar loop)
VRF_HD_NOINLINE void fp_inv_pow(FpS* out, const FpS* a) {
  FpN t[8];
  t[1] = fp_mul(*a, fp_one());
  t[2] = fp_sqr(t[1]);
  for (int i = 3; i < 8; ++i) t[i] = fp_mul(t[i - 1], t[1]);
  FpN acc = fp_one();
  bool started = false;
  for (int w = 126; w >= 0; --w) {                 // 381 bits = 127 windows of 3
    const int bit = 3 * w;
    uint32_t d = vrfk::BLS_EXP_INV[bit >> 5] >> (bit & 31);
    if ((bit & 31) > 29 && (bit >> 5) + 1 < 12) d |= vrfk::BLS_EXP_INV[(bit >> 5) + 1] << (32 - (bit & 31));
    d &= 7;
    if (started) { acc = fp_sqr(acc); acc = fp_sqr(acc); acc = fp_sqr(acc); }
    if (d != 0) {
      FpN s = t[1];
      for (int j = 2; j < 8; ++j) if (d == (uint32_t)j) s = t[j];
      acc = started ? fp_mul(acc, s) : s;
      started = true;
    }
  }
  *out = acc;
}

// a wave-wide "any lane still working" (host build: the lane itself)
VRF_HD bool bls_any(bool v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_ballot_w64(v) != 0;
#else
  return v;
#endif
}

// Inverse by POSITIVE DIVSTEPS with cofactors (the loop of fe.cuh's Jacobi symbol, here on 14 x 28-bit limbs): f = p,
// g = the canonical Montgomery image a~; every step keeps f odd and replaces g by (g + [g odd] f) / 2, swapping first
// when the counter says so -- sums, never differences, so f and g stay non-negative.  28 steps read only the low 32 bits of
// f and g and yield a 2x2 matrix of entries < 2^29; applying it to (f, g) divides by 2^28 exactly, applying it to the
// cofactors (d, e) -- f = d a~, g = e a~ (mod p) -- adds the multiple of p that makes the division exact.  f = 1 ends a
// lane: d = a~^-1.  About 45 rounds of ~700 instructions against ~500 field products (3 x 10^5 instructions) for
// a^(p-2): the inversion was 35 % of k_g1_final and 6-8 % of a pairing check.  0 -> 0, as the power gives.
constexpr int FPINV_K = LWB;            // steps per round = limb width
constexpr int FPINV_MAX_ROUNDS = 72;    // observed <= 50; a lane that is not done after these falls back to the power
VRF_HD_NOINLINE void fp_inv(FpS* out, const FpS* a) {
  uint32_t f[NLB], g[NLB], d[NLB], e[NLB];
  fp_canon_limbs(g, *a);
  uint32_t nz = 0;
#pragma unroll
  for (int i = 0; i < NLB; ++i) { f[i] = vrfk::BLS_P28[i]; d[i] = 0; e[i] = (i == 0); nz |= g[i]; }
  int32_t eta = -1;
  bool done = nz == 0;
#pragma unroll 1
  for (int round = 0; round < FPINV_MAX_ROUNDS; ++round) {
    if (!bls_any(!done)) break;
    uint32_t f0 = f[0] | (f[1] << LWB), g0 = g[0] | (g[1] << LWB);
    uint32_t u = 1, v = 0, q = 0, r = 1;
    int32_t et = eta;
#pragma unroll
    for (int i = 0; i < FPINV_K; ++i) {
      const uint32_t m = 0u - (g0 & 1u);                       // g odd
      const uint32_t sw = m & (uint32_t)(et >> 31);            // ... and eta < 0: swap
      uint32_t t = (f0 ^ g0) & sw; f0 ^= t; g0 ^= t;
      t = (u ^ q) & sw; u ^= t; q ^= t;
      t = (v ^ r) & sw; v ^= t; r ^= t;
      et = (et ^ (int32_t)sw) - (int32_t)sw;
      g0 += f0 & m; q += u & m; r += v & m;
      g0 >>= 1; u <<= 1; v <<= 1; et -= 1;
    }
    // (f, g) <- (u f + v g, q f + r g) / 2^28: the low limb of both sums is zero by construction
    uint32_t nf[NLB], ng[NLB], nd[NLB], ne[NLB];
    uint64_t af = ((uint64_t)u * f[0] + (uint64_t)v * g[0]) >> LWB, ag = ((uint64_t)q * f[0] + (uint64_t)r * g[0]) >> LWB;
    // (d, e) <- (u d + v e + md p, q d + r e + me p) / 2^28 with md, me = -(low limb) / p mod 2^28
    const uint64_t td = (uint64_t)u * d[0] + (uint64_t)v * e[0], te = (uint64_t)q * d[0] + (uint64_t)r * e[0];
    const uint32_t md = ((uint32_t)td * vrfk::BLS_PINV28) & MASKB, me = ((uint32_t)te * vrfk::BLS_PINV28) & MASKB;
    uint64_t ad = (td + (uint64_t)md * vrfk::BLS_P28[0]) >> LWB, ae = (te + (uint64_t)me * vrfk::BLS_P28[0]) >> LWB;
#pragma unroll
    for (int i = 1; i < NLB; ++i) {
      af += (uint64_t)u * f[i] + (uint64_t)v * g[i];
      ag += (uint64_t)q * f[i] + (uint64_t)r * g[i];
      ad += (uint64_t)u * d[i] + (uint64_t)v * e[i] + (uint64_t)md * vrfk::BLS_P28[i];
      ae += (uint64_t)q * d[i] + (uint64_t)r * e[i] + (uint64_t)me * vrfk::BLS_P28[i];
      nf[i - 1] = (uint32_t)af & MASKB; af >>= LWB;
      ng[i - 1] = (uint32_t)ag & MASKB; ag >>= LWB;
      nd[i - 1] = (uint32_t)ad & MASKB; ad >>= LWB;
      ne[i - 1] = (uint32_t)ae & MASKB; ae >>= LWB;
    }
    nf[NLB - 1] = (uint32_t)af; ng[NLB - 1] = (uint32_t)ag; nd[NLB - 1] = (uint32_t)ad; ne[NLB - 1] = (uint32_t)ae;
    uint32_t rest = 0;
#pragma unroll
    for (int i = 1; i < NLB; ++i) rest |= nf[i];
    const bool now_one = nf[0] == 1u && rest == 0;
#pragma unroll
    for (int i = 0; i < NLB; ++i) {          // a finished lane keeps its state
      f[i] = done ? f[i] : nf[i]; g[i] = done ? g[i] : ng[i];
      d[i] = done ? d[i] : nd[i]; e[i] = done ? e[i] : ne[i];
    }
    eta = done ? eta : et;
    done = done || now_one;
  }
  if (bls_any(!done)) {                      // rounds exhausted (not observed): the power, for the whole wave
    FpS slow;
    fp_inv_pow(&slow, a);
    if (!done) { *out = slow; return; }
  }
  // d = a~^-1 = a^-1 / R as an integer < (rounds + 1) p: two Montgomery products by R^2 bring it to a^-1 R
  Fp<1, 128> dd;
#pragma unroll
  for (int i = 0; i < NLB; ++i) dd.v[i] = (int32_t)(nz == 0 ? 0u : d[i]);
  *out = fp_fit(fp_mul(fp_fit(fp_mul(dd, fp_const(vrfk::BLS_R2))), fp_const(vrfk::BLS_R2)));
}

// ------------------------------------------------------------------------------------ Fp2
// a + b u, u^2 = -1.  Fp2<L, V>: both components carry the bound (L, V).  Stored values are
// Fp2S = Fp2<1, STORE_V>; everything in between is lazily typed.
template <int L, int V>
struct Fp2T { Fp<L, V> a, b; };
using Fp2 = Fp2T<1, STORE_V>;

VRF_HD Fp2 fp2_zero() { Fp2 r; r.a = fp_zero(); r.b = fp_zero(); return r; }
VRF_HD Fp2 fp2_one() { Fp2 r; r.a = fp_one(); r.b = fp_zero(); return r; }
template <int L, int V>
VRF_HD Fp2 fp2_fit(const Fp2T<L, V>& x) { Fp2 r; r.a = fp_fit(x.a); r.b = fp_fit(x.b); return r; }
template <int L, int V>
VRF_HD Fp2T<1, V> fp2_norm(const Fp2T<L, V>& x) { Fp2T<1, V> r; r.a = fp_norm(x.a); r.b = fp_norm(x.b); return r; }

template <int L1, int V1, int L2, int V2>
VRF_HD Fp2T<lsum(L1, L2), V1 + V2> fp2_add(const Fp2T<L1, V1>& x, const Fp2T<L2, V2>& y) {
  Fp2T<lsum(L1, L2), V1 + V2> r; r.a = fp_add(x.a, y.a); r.b = fp_add(x.b, y.b); return r;
}
template <int L1, int V1, int L2, int V2>
VRF_HD Fp2T<lsum(L1, L2), V1 + V2> fp2_sub(const Fp2T<L1, V1>& x, const Fp2T<L2, V2>& y) {
  Fp2T<lsum(L1, L2), V1 + V2> r; r.a = fp_sub(x.a, y.a); r.b = fp_sub(x.b, y.b); return r;
}
template <int L, int V>
VRF_HD Fp2T<L, V> fp2_neg(const Fp2T<L, V>& x) { Fp2T<L, V> r; r.a = fp_neg(x.a); r.b = fp_neg(x.b); return r; }
template <int L, int V>
VRF_HD Fp2T<L, V> fp2_conj(const Fp2T<L, V>& x) { Fp2T<L, V> r; r.a = x.a; r.b = fp_neg(x.b); return r; }
template <int L, int V>
VRF_HD Fp2T<lsum(L, L), 2 * V> fp2_dbl(const Fp2T<L, V>& x) { return fp2_add(x, x); }
template <int L, int V>
VRF_HD Fp2T<lsum(L, L), 2 * V> fp2_mul_xi(const Fp2T<L, V>& x) {     // * (1 + u): (a - b) + (a + b) u
  Fp2T<lsum(L, L), 2 * V> r; r.a = fp_sub(x.a, x.b); r.b = fp_add(x.a, x.b); return r;
}
// (a + bu)(c + du) = (ac - bd) + ((a+b)(c+d) - ac - bd) u; operands are normalised first if lazy
template <int L1, int V1, int L2, int V2>
VRF_HD auto fp2_mul(const Fp2T<L1, V1>& x0, const Fp2T<L2, V2>& y0) {
  auto x = fp2_norm(x0);
  auto y = fp2_norm(y0);
  auto ac = fp_mul(x.a, y.a);
  auto bd = fp_mul(x.b, y.b);
  auto s = fp_mul(fp_add(x.a, x.b), fp_add(y.a, y.b));
  constexpr int VA = 2 * mul_v(V1, V2), VB = mul_v(2 * V1, 2 * V2) + 2 * mul_v(V1, V2);
  constexpr int VO = VA > VB ? VA : VB;
  Fp2T<3, VO> r;
  r.a = fp_sub(ac, bd);
  r.b = fp_sub(fp_sub(s, ac), bd);
  return r;
}
template <int L, int V>
VRF_HD auto fp2_sqr(const Fp2T<L, V>& x0) {       // (a+b)(a-b) + 2ab u
  auto x = fp2_norm(x0);
  auto t = fp_mul(fp_add(x.a, x.b), fp_sub(x.a, x.b));
  auto ab = fp_mul(x.a, x.b);
  constexpr int VO = mul_v(2 * V, 2 * V) > 2 * mul_v(V, V) ? mul_v(2 * V, 2 * V) : 2 * mul_v(V, V);
  Fp2T<2, VO> r;
  r.a = t;
  r.b = fp_dbl(ab);
  return r;
}
template <int L, int V, int LK, int VK>
VRF_HD auto fp2_mul_fp(const Fp2T<L, V>& x0, const Fp<LK, VK>& k) {
  auto x = fp2_norm(x0);
  auto kn = fp_norm(k);
  Fp2T<1, mul_v(V, VK)> r; r.a = fp_mul(x.a, kn); r.b = fp_mul(x.b, kn); return r;
}
template <int L, int V>
VRF_HD bool fp2_is_zero(const Fp2T<L, V>& x) { return fp_is_zero(x.a) && fp_is_zero(x.b); }
template <int L1, int V1, int L2, int V2>
VRF_HD bool fp2_eq(const Fp2T<L1, V1>& x, const Fp2T<L2, V2>& y) { return fp_eq(x.a, y.a) && fp_eq(x.b, y.b); }
VRF_HD_NOINLINE void fp2_inv(Fp2* out, const Fp2* x) {
  FpS n = fp_fit(fp_add(fp_sqr(x->a), fp_sqr(x->b)));
  FpS d;
  fp_inv(&d, &n);
  out->a = fp_fit(fp_mul(x->a, d));
  out->b = fp_fit(fp_neg(fp_mul(x->b, d)));
}

// ------------------------------------------------------------------------------------ Fp6 / Fp12
// Stored towers hold Fp2 (storage form).  The mul/sqr bodies keep their intermediates lazily typed
// and fit only the outputs.
struct Fp6 { Fp2 c0, c1, c2; };
struct Fp12 { Fp6 c0, c1; };

VRF_HD void fp6_add(Fp6* r, const Fp6* x, const Fp6* y) {
  r->c0 = fp2_fit(fp2_add(x->c0, y->c0)); r->c1 = fp2_fit(fp2_add(x->c1, y->c1)); r->c2 = fp2_fit(fp2_add(x->c2, y->c2));
}
VRF_HD void fp6_sub(Fp6* r, const Fp6* x, const Fp6* y) {
  r->c0 = fp2_fit(fp2_sub(x->c0, y->c0)); r->c1 = fp2_fit(fp2_sub(x->c1, y->c1)); r->c2 = fp2_fit(fp2_sub(x->c2, y->c2));
}
VRF_HD void fp6_neg(Fp6* r, const Fp6* x) {
  r->c0 = fp2_neg(x->c0); r->c1 = fp2_neg(x->c1); r->c2 = fp2_neg(x->c2);
}
VRF_HD void fp6_mul_v(Fp6* r, const Fp6* x) {          // * v (in place safe)
  Fp2 t = fp2_fit(fp2_mul_xi(x->c2));
  Fp2 c0 = x->c0, c1 = x->c1;
  r->c0 = t; r->c1 = c0; r->c2 = c1;
}
VRF_HD_NOINLINE void fp6_mul(Fp6* r, const Fp6* x, const Fp6* y) {
  auto v0 = fp2_mul(x->c0, y->c0);
  auto v1 = fp2_mul(x->c1, y->c1);
  auto v2 = fp2_mul(x->c2, y->c2);
  auto t0 = fp2_sub(fp2_sub(fp2_mul(fp2_add(x->c1, x->c2), fp2_add(y->c1, y->c2)), v1), v2);
  auto t1 = fp2_sub(fp2_sub(fp2_mul(fp2_add(x->c0, x->c1), fp2_add(y->c0, y->c1)), v0), v1);
  auto t2 = fp2_sub(fp2_sub(fp2_mul(fp2_add(x->c0, x->c2), fp2_add(y->c0, y->c2)), v0), v2);
  Fp2 c0 = fp2_fit(fp2_add(v0, fp2_mul_xi(fp2_norm(t0))));
  Fp2 c1 = fp2_fit(fp2_add(t1, fp2_mul_xi(fp2_norm(v2))));
  Fp2 c2 = fp2_fit(fp2_add(t2, v1));
  r->c0 = c0; r->c1 = c1; r->c2 = c2;
}
// sparse: y = (c0, c1, 0)
VRF_HD_NOINLINE void fp6_mul_by_01(Fp6* r, const Fp6* x, const Fp2* c0, const Fp2* c1) {
  auto aa = fp2_mul(x->c0, *c0);
  auto bb = fp2_mul(x->c1, *c1);
  auto u1 = fp2_sub(fp2_mul(fp2_add(x->c1, x->c2), *c1), bb);
  auto t3 = fp2_add(fp2_sub(fp2_mul(fp2_add(x->c0, x->c2), *c0), aa), bb);
  auto t2 = fp2_sub(fp2_sub(fp2_mul(fp2_add(x->c0, x->c1), fp2_add(*c0, *c1)), aa), bb);
  Fp2 o0 = fp2_fit(fp2_add(fp2_mul_xi(fp2_norm(u1)), aa));
  Fp2 o1 = fp2_fit(t2);
  Fp2 o2 = fp2_fit(t3);
  r->c0 = o0; r->c1 = o1; r->c2 = o2;
}
// sparse: y = (0, c1, 0)
VRF_HD_NOINLINE void fp6_mul_by_1(Fp6* r, const Fp6* x, const Fp2* c1) {
  Fp2 t0 = fp2_fit(fp2_mul_xi(fp2_norm(fp2_mul(x->c2, *c1))));
  Fp2 t1 = fp2_fit(fp2_mul(x->c0, *c1));
  Fp2 t2 = fp2_fit(fp2_mul(x->c1, *c1));
  r->c0 = t0; r->c1 = t1; r->c2 = t2;
}
VRF_HD_NOINLINE void fp6_inv(Fp6* r, const Fp6* x) {
  Fp2 t0 = fp2_fit(fp2_sub(fp2_sqr(x->c0), fp2_mul_xi(fp2_norm(fp2_mul(x->c1, x->c2)))));
  Fp2 t1 = fp2_fit(fp2_sub(fp2_mul_xi(fp2_norm(fp2_sqr(x->c2))), fp2_mul(x->c0, x->c1)));
  Fp2 t2 = fp2_fit(fp2_sub(fp2_sqr(x->c1), fp2_mul(x->c0, x->c2)));
  auto inner = fp2_fit(fp2_add(fp2_mul(x->c2, t1), fp2_mul(x->c1, t2)));
  Fp2 d = fp2_fit(fp2_add(fp2_mul(x->c0, t0), fp2_mul_xi(inner)));
  Fp2 di;
  fp2_inv(&di, &d);
  r->c0 = fp2_fit(fp2_mul(t0, di)); r->c1 = fp2_fit(fp2_mul(t1, di)); r->c2 = fp2_fit(fp2_mul(t2, di));
}

VRF_HD void fp12_one(Fp12* r) {
  r->c0.c0 = fp2_one(); r->c0.c1 = fp2_zero(); r->c0.c2 = fp2_zero();
  r->c1.c0 = fp2_zero(); r->c1.c1 = fp2_zero(); r->c1.c2 = fp2_zero();
}
VRF_HD_NOINLINE void fp12_mul(Fp12* r, const Fp12* x, const Fp12* y) {
  Fp6 t0, t1, s, sx, sy;
  fp6_mul(&t0, &x->c0, &y->c0);
  fp6_mul(&t1, &x->c1, &y->c1);
  fp6_add(&sx, &x->c0, &x->c1);
  fp6_add(&sy, &y->c0, &y->c1);
  fp6_mul(&s, &sx, &sy);
  fp6_sub(&s, &s, &t0);
  fp6_sub(&r->c1, &s, &t1);
  fp6_mul_v(&t1, &t1);
  fp6_add(&r->c0, &t0, &t1);
}
VRF_HD_NOINLINE void fp12_sqr(Fp12* r, const Fp12* x) {
  Fp6 ab, s0, s1, t;
  fp6_mul(&ab, &x->c0, &x->c1);
  fp6_add(&s0, &x->c0, &x->c1);
  fp6_mul_v(&t, &x->c1);
  fp6_add(&s1, &x->c0, &t);
  fp6_mul(&s0, &s0, &s1);                 // (a0 + a1)(a0 + v a1)
  fp6_sub(&s0, &s0, &ab);
  fp6_mul_v(&t, &ab);
  fp6_sub(&r->c0, &s0, &t);
  fp6_add(&r->c1, &ab, &ab);
}
// f *= (c0 + c1 v + c4 v w)
VRF_HD_NOINLINE void fp12_mul_by_014(Fp12* f, const Fp2* c0, const Fp2* c1, const Fp2* c4) {
  Fp6 aa, bb, s;
  fp6_mul_by_01(&aa, &f->c0, c0, c1);
  fp6_mul_by_1(&bb, &f->c1, c4);
  Fp2 o = fp2_fit(fp2_add(*c1, *c4));
  fp6_add(&s, &f->c1, &f->c0);
  fp6_mul_by_01(&s, &s, c0, &o);
  fp6_sub(&s, &s, &aa);
  fp6_sub(&f->c1, &s, &bb);
  fp6_mul_v(&bb, &bb);
  fp6_add(&f->c0, &bb, &aa);
}
// Granger-Scott squaring for elements of the cyclotomic subgroup (after the easy part of the final
// exponentiation): three Fp4 squarings, 6 Fp2 products instead of 12.
template <int L1, int V1, int L2, int V2>
VRF_HD void fp4_square(Fp2& t0, Fp2& t1, const Fp2T<L1, V1>& a, const Fp2T<L2, V2>& b) {
  auto tmp = fp2_mul(a, b);
  auto tn = fp2_norm(tmp);
  auto s = fp2_mul(fp2_add(a, b), fp2_add(fp2_mul_xi(b), a));
  t0 = fp2_fit(fp2_sub(fp2_sub(s, tn), fp2_mul_xi(tn)));
  t1 = fp2_fit(fp2_dbl(tn));
}
VRF_HD_NOINLINE void fp12_cyclotomic_sqr(Fp12* r, const Fp12* x) {
  const Fp2 r0 = x->c0.c0, r4 = x->c0.c1, r3 = x->c0.c2, r2 = x->c1.c0, r1 = x->c1.c1, r5 = x->c1.c2;
  Fp2 t0, t1, t2, t3, t4, t5;
  fp4_square(t0, t1, r0, r1);
  fp4_square(t2, t3, r2, r3);
  fp4_square(t4, t5, r4, r5);
  auto three = [](const Fp2& t) { return fp2_add(fp2_dbl(t), t); };
  r->c0.c0 = fp2_fit(fp2_sub(three(t0), fp2_dbl(r0)));                       // z0 = 3 t0 - 2 r0
  r->c1.c1 = fp2_fit(fp2_add(three(t1), fp2_dbl(r1)));                       // z1 = 3 t1 + 2 r1
  r->c1.c0 = fp2_fit(fp2_add(three(fp2_fit(fp2_mul_xi(t5))), fp2_dbl(r2)));  // z2 = 3 xi t5 + 2 r2
  r->c0.c2 = fp2_fit(fp2_sub(three(t4), fp2_dbl(r3)));                       // z3 = 3 t4 - 2 r3
  r->c0.c1 = fp2_fit(fp2_sub(three(t2), fp2_dbl(r4)));                       // z4 = 3 t2 - 2 r4
  r->c1.c2 = fp2_fit(fp2_add(three(t3), fp2_dbl(r5)));                       // z5 = 3 t3 + 2 r5
}

VRF_HD void fp12_conj(Fp12* r, const Fp12* x) {
  r->c0 = x->c0;
  fp6_neg(&r->c1, &x->c1);
}
VRF_HD_NOINLINE void fp12_inv(Fp12* r, const Fp12* x) {
  Fp6 t0, t1, d;
  fp6_mul(&t0, &x->c0, &x->c0);
  fp6_mul(&t1, &x->c1, &x->c1);
  fp6_mul_v(&t1, &t1);
  fp6_sub(&t0, &t0, &t1);
  fp6_inv(&d, &t0);
  fp6_mul(&r->c0, &x->c0, &d);
  fp6_mul(&t1, &x->c1, &d);
  fp6_neg(&r->c1, &t1);
}
VRF_HD Fp2 gamma_const(int i) {
  Fp2 g;
  switch (i) {
    case 1: g.a = fp_const(vrfk::BLS_GAMMA1_RE_M); g.b = fp_const(vrfk::BLS_GAMMA1_IM_M); break;
    case 2: g.a = fp_const(vrfk::BLS_GAMMA2_RE_M); g.b = fp_const(vrfk::BLS_GAMMA2_IM_M); break;
    case 3: g.a = fp_const(vrfk::BLS_GAMMA3_RE_M); g.b = fp_const(vrfk::BLS_GAMMA3_IM_M); break;
    case 4: g.a = fp_const(vrfk::BLS_GAMMA4_RE_M); g.b = fp_const(vrfk::BLS_GAMMA4_IM_M); break;
    default: g.a = fp_const(vrfk::BLS_GAMMA5_RE_M); g.b = fp_const(vrfk::BLS_GAMMA5_IM_M); break;
  }
  return g;
}
// f^p: coefficient of w^i -> conj(coefficient) * xi^(i (p-1)/6)
VRF_HD_NOINLINE void fp12_frob(Fp12* r, const Fp12* x) {
  Fp2 o0 = fp2_conj(x->c0.c0);                                          // w^0
  Fp2 o1 = fp2_fit(fp2_mul(fp2_conj(x->c1.c0), gamma_const(1)));        // w^1
  Fp2 o2 = fp2_fit(fp2_mul(fp2_conj(x->c0.c1), gamma_const(2)));        // w^2
  Fp2 o3 = fp2_fit(fp2_mul(fp2_conj(x->c1.c1), gamma_const(3)));        // w^3
  Fp2 o4 = fp2_fit(fp2_mul(fp2_conj(x->c0.c2), gamma_const(4)));        // w^4
  Fp2 o5 = fp2_fit(fp2_mul(fp2_conj(x->c1.c2), gamma_const(5)));        // w^5
  r->c0.c0 = o0; r->c1.c0 = o1; r->c0.c1 = o2; r->c1.c1 = o3; r->c0.c2 = o4; r->c1.c2 = o5;
}
VRF_HD bool fp12_is_one(const Fp12* x) {
  return fp2_eq(x->c0.c0, fp2_one()) && fp2_is_zero(x->c0.c1) && fp2_is_zero(x->c0.c2) &&
         fp2_is_zero(x->c1.c0) && fp2_is_zero(x->c1.c1) && fp2_is_zero(x->c1.c2);
}

// ------------------------------------------------------------------------------------ Miller loop
struct G2Proj { Fp2 X, Y, Z; };
struct G1Aff { FpS x, y; };
struct G2Aff { Fp2 x, y; };

VRF_HD Fp2 twist_b() {            // b' = 4 (1 + u)
  FpS four = fp_fit(fp_dbl(fp_dbl(fp_one())));
  Fp2 r; r.a = four; r.b = four;
  return r;
}
// doubling step; line = (c0) + (c1 * x_P) v + (c4 * y_P) v w
VRF_HD_NOINLINE void g2_double_step(G2Proj* T, Fp2* c0, Fp2* c1, Fp2* c4) {
  const auto inv2 = fp_const(vrfk::BLS_INV2_M);
  Fp2 a = fp2_fit(fp2_mul_fp(fp2_mul(T->X, T->Y), inv2));
  Fp2 b = fp2_fit(fp2_sqr(T->Y));
  Fp2 c = fp2_fit(fp2_sqr(T->Z));
  Fp2 e = fp2_fit(fp2_mul(twist_b(), fp2_add(fp2_dbl(c), c)));
  Fp2 f = fp2_fit(fp2_add(fp2_dbl(e), e));
  Fp2 g = fp2_fit(fp2_mul_fp(fp2_add(b, f), inv2));
  Fp2 h = fp2_fit(fp2_sub(fp2_sqr(fp2_add(T->Y, T->Z)), fp2_add(b, c)));
  Fp2 j = fp2_fit(fp2_sqr(T->X));
  Fp2 e2 = fp2_fit(fp2_sqr(e));
  T->X = fp2_fit(fp2_mul(a, fp2_sub(b, f)));
  T->Y = fp2_fit(fp2_sub(fp2_sqr(g), fp2_add(fp2_dbl(e2), e2)));
  T->Z = fp2_fit(fp2_mul(b, h));
  *c0 = fp2_fit(fp2_sub(b, e));
  *c1 = fp2_fit(fp2_neg(fp2_add(fp2_dbl(j), j)));
  *c4 = h;
}
VRF_HD_NOINLINE void g2_add_step(G2Proj* T, const G2Aff* Q, Fp2* c0, Fp2* c1, Fp2* c4) {
  Fp2 theta = fp2_fit(fp2_sub(T->Y, fp2_mul(Q->y, T->Z)));
  Fp2 lam = fp2_fit(fp2_sub(T->X, fp2_mul(Q->x, T->Z)));
  Fp2 c = fp2_fit(fp2_sqr(theta));
  Fp2 d = fp2_fit(fp2_sqr(lam));
  Fp2 e = fp2_fit(fp2_mul(lam, d));
  Fp2 f = fp2_fit(fp2_mul(T->Z, c));
  Fp2 g = fp2_fit(fp2_mul(T->X, d));
  Fp2 h = fp2_fit(fp2_sub(fp2_add(e, f), fp2_dbl(g)));
  Fp2 x3 = fp2_fit(fp2_mul(lam, h));
  Fp2 y3 = fp2_fit(fp2_sub(fp2_mul(theta, fp2_sub(g, h)), fp2_mul(e, T->Y)));
  Fp2 z3 = fp2_fit(fp2_mul(T->Z, e));
  T->X = x3; T->Y = y3; T->Z = z3;
  *c0 = fp2_fit(fp2_sub(fp2_mul(theta, Q->x), fp2_mul(lam, Q->y)));
  *c1 = fp2_neg(theta);
  *c4 = lam;
}

constexpr uint64_t X_ABS = 0xD201000000010000ULL;

// f = prod_i f_{|x|,Q_i}(P_i), conjugated.  skip[i]: pair i contributes 1 (a point at infinity).
template <int NPAIRS>
VRF_HD_NOINLINE void miller_loop(Fp12* f, const G1Aff* P, const G2Aff* Q, const bool* skip) {
  G2Proj T[NPAIRS];
  for (int i = 0; i < NPAIRS; ++i) { T[i].X = Q[i].x; T[i].Y = Q[i].y; T[i].Z = fp2_one(); }
  fp12_one(f);
  for (int bit = 62; bit >= 0; --bit) {
    Fp12 t;
    fp12_sqr(&t, f);
    *f = t;
    for (int i = 0; i < NPAIRS; ++i) {
      Fp2 c0, c1, c4;
      g2_double_step(&T[i], &c0, &c1, &c4);
      if (!skip[i]) {
        c1 = fp2_fit(fp2_mul_fp(c1, P[i].x));
        c4 = fp2_fit(fp2_mul_fp(c4, P[i].y));
        fp12_mul_by_014(f, &c0, &c1, &c4);
      }
    }
    if ((X_ABS >> bit) & 1) {
      for (int i = 0; i < NPAIRS; ++i) {
        Fp2 c0, c1, c4;
        g2_add_step(&T[i], &Q[i], &c0, &c1, &c4);
        if (!skip[i]) {
          c1 = fp2_fit(fp2_mul_fp(c1, P[i].x));
          c4 = fp2_fit(fp2_mul_fp(c4, P[i].y));
          fp12_mul_by_014(f, &c0, &c1, &c4);
        }
      }
    }
  }
  Fp12 t;
  fp12_conj(&t, f);
  *f = t;
}

// f^x, x = -X_ABS, for f in the cyclotomic subgroup (inverse = conjugate)
VRF_HD_NOINLINE void exp_by_x(Fp12* r, const Fp12* f) {
  Fp12 acc = *f, t;
  for (int bit = 62; bit >= 0; --bit) {
    fp12_cyclotomic_sqr(&t, &acc);
    acc = t;
    if ((X_ABS >> bit) & 1) {
      fp12_mul(&t, &acc, f);
      acc = t;
    }
  }
  fp12_conj(r, &acc);
}

// f^(3 (p^12 - 1)/r)
VRF_HD_NOINLINE void final_exponentiation(Fp12* r, const Fp12* f) {
  Fp12 t0, t1, f2, y0, y1, y2, y3;
  fp12_conj(&t0, f);
  fp12_inv(&t1, f);
  fp12_mul(&f2, &t0, &t1);            // f^(p^6 - 1)
  fp12_frob(&t0, &f2);
  fp12_frob(&t1, &t0);
  fp12_mul(&t0, &t1, &f2);            // ^(p^2 + 1)
  f2 = t0;
  exp_by_x(&t0, &f2); fp12_conj(&t1, &f2); fp12_mul(&y0, &t0, &t1);        // ^(x - 1)
  exp_by_x(&t0, &y0); fp12_conj(&t1, &y0); fp12_mul(&y1, &t0, &t1);        // ^(x - 1)^2
  exp_by_x(&t0, &y1); fp12_frob(&t1, &y1); fp12_mul(&y2, &t0, &t1);        // ^(x + p)
  exp_by_x(&t0, &y2); exp_by_x(&t1, &t0);                                  // y2^(x^2)
  fp12_frob(&t0, &y2); fp12_frob(&y3, &t0);                                // y2^(p^2)
  fp12_mul(&t0, &t1, &y3);
  fp12_conj(&t1, &y2);
  fp12_mul(&y3, &t0, &t1);                                                 // ^(x^2 + p^2 - 1)
  fp12_cyclotomic_sqr(&t0, &f2);
  fp12_mul(&t1, &t0, &f2);                                                 // f2^3
  fp12_mul(r, &y3, &t1);
}

// ------------------------------------------------------------------------------------ item
enum : uint32_t { PST_OK = 0, PST_FAIL = 1, PST_INVALID = 2 };

// G1: x || y, 48-byte little-endian each; all-zero = point at infinity.  G2: x.c0 || x.c1 ||
// y.c0 || y.c1.  Returns false on a coordinate >= p or a point off its curve.
VRF_HD bool g1_load(G1Aff& P, bool& inf, const uint32_t* w /*24 words*/) {
  uint32_t any = 0;
  for (int i = 0; i < 24; ++i) any |= w[i];
  inf = any == 0;
  const bool okx = fp_from_words(P.x, w), oky = fp_from_words(P.y, w + 12);
  bool ok = okx && oky;
  // y^2 = x^3 + 4
  auto four = fp_dbl(fp_dbl(fp_one()));
  auto rhs = fp_add(fp_mul(fp_sqr(P.x), P.x), four);
  ok = ok && (inf || fp_eq(fp_sqr(P.y), rhs));
  return ok;
}
VRF_HD bool g2_load(G2Aff& Q, bool& inf, const uint32_t* w /*48 words*/) {
  uint32_t any = 0;
  for (int i = 0; i < 48; ++i) any |= w[i];
  inf = any == 0;
  const bool ok0 = fp_from_words(Q.x.a, w), ok1 = fp_from_words(Q.x.b, w + 12);
  const bool ok2 = fp_from_words(Q.y.a, w + 24), ok3 = fp_from_words(Q.y.b, w + 36);
  bool ok = ok0 && ok1 && ok2 && ok3;
  auto rhs = fp2_add(fp2_mul(fp2_sqr(Q.x), Q.x), twist_b());
  ok = ok && (inf || fp2_eq(fp2_sqr(Q.y), rhs));
  return ok;
}

// e(P0, Q0) * e(P1, Q1) == 1 ?   g1: 2 x 24 words, g2: 2 x 48 words
VRF_HD_NOINLINE uint32_t pairing_check2_item(const uint32_t* g1, const uint32_t* g2) {
  G1Aff P[2];
  G2Aff Q[2];
  bool skip[2];
  bool ok = true;
  for (int i = 0; i < 2; ++i) {
    bool i1, i2;
    ok = g1_load(P[i], i1, g1 + 24 * i) && ok;
    ok = g2_load(Q[i], i2, g2 + 48 * i) && ok;
    skip[i] = i1 || i2;
  }
  Fp12 f, e;
  miller_loop<2>(&f, P, Q, skip);
  final_exponentiation(&e, &f);
  bool one = fp12_is_one(&e);
  if (!ok) return PST_INVALID;
  return one ? PST_OK : PST_FAIL;
}

// ---- shared G2 points (the SRS case of a KZG check: every item pairs against the same Q0, Q1) ----
// The G2 walk of the Miller loop does not depend on the G1 points: its 68 lines per pair are computed ONCE
// (pairing_prepare_g2_pair, one lane per pair) and every item only scales them by its own (x_P, y_P).
constexpr int G2_LINES = 63 + 5;                       // doublings for bits 62..0, additions at the 5 set bits
constexpr int G2_LINE_WORDS = 3 * 2 * NLB;             // c0, c1, c4 as Fp2 limbs
constexpr int G2_PREP_WORDS = 2 * G2_LINES * G2_LINE_WORDS + 4;   // two pairs + flags (ok0, inf0, ok1, inf1)

VRF_HD void fp2_store_words(uint32_t* dst, const Fp2& x) {
  for (int i = 0; i < NLB; ++i) { dst[i] = (uint32_t)x.a.v[i]; dst[NLB + i] = (uint32_t)x.b.v[i]; }
}
VRF_HD Fp2 fp2_load_words(const uint32_t* src) {
  Fp2 x;
  for (int i = 0; i < NLB; ++i) { x.a.v[i] = (int32_t)src[i]; x.b.v[i] = (int32_t)src[NLB + i]; }
  return x;
}
// lines of pair `pi` (g2: 2 x 48 words) in Miller-loop order, then its flags
VRF_HD_NOINLINE void pairing_prepare_g2_pair(const uint32_t* g2, uint32_t* prep, int pi) {
  G2Aff Q;
  bool inf;
  const bool ok = g2_load(Q, inf, g2 + 48 * pi);
  G2Proj T;
  T.X = Q.x; T.Y = Q.y; T.Z = fp2_one();
  uint32_t* dst = prep + (size_t)pi * G2_LINES * G2_LINE_WORDS;
  for (int bit = 62; bit >= 0; --bit) {
    Fp2 c0, c1, c4;
    g2_double_step(&T, &c0, &c1, &c4);
    fp2_store_words(dst, c0); fp2_store_words(dst + 2 * NLB, c1); fp2_store_words(dst + 4 * NLB, c4);
    dst += G2_LINE_WORDS;
    if ((X_ABS >> bit) & 1) {
      g2_add_step(&T, &Q, &c0, &c1, &c4);
      fp2_store_words(dst, c0); fp2_store_words(dst + 2 * NLB, c1); fp2_store_words(dst + 4 * NLB, c4);
      dst += G2_LINE_WORDS;
    }
  }
  uint32_t* flags = prep + (size_t)2 * G2_LINES * G2_LINE_WORDS;
  flags[2 * pi] = ok ? 1u : 0u;
  flags[2 * pi + 1] = inf ? 1u : 0u;
}

}  // namespace bls
