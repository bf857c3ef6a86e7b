// fe.cuh -- Fq arithmetic for gfx950: 9 limbs x 29 bits, Montgomery radix R = 2^261.
//
// Replaces ark_ff::Fp<MontBackend, 4> (the type behind `BaseField`, /root/reference
// src/lib.rs:15) on the VRF hot path.  Design notes:
//
//  * gfx950 has a full-rate 32x32+64 multiply-add (v_mad_u64_u32, measured at the same issue
//    rate as v_add: profiles/r01_instr_rate_microbench.jsonl) but that instruction has no
//    carry-IN.  A saturated 2^32 radix therefore pays an add-with-carry per product.  With
//    29-bit limbs nine products plus nine reduction products fit a 64-bit accumulator with
//    no carry handling at all: a field multiplication is 81 + 72 v_mad_u64_u32 plus one
//    shift/mask per column.
//  * Limbs are *lazy*: additions and subtractions are nine independent 32-bit adds with no
//    carry propagation and no modular reduction.  Every value carries two compile-time
//    bounds: L (each limb < L * (2^29 + 2^13)) and V (value < V * q).  fe_mul static_asserts
//    that its 64-bit column accumulators cannot overflow (L1 * L2 <= 6) and derives the
//    output bound, so an overflow is a compile error, not a wrong proof.
//  * q = 1 (mod 2^32) => -q^-1 = -1 (mod 2^29) and q's low limb is 1: the Montgomery
//    quotient digit is (-acc) & mask, no multiply.
//
// Everything is __host__ __device__ so the same source can be unit-tested on the CPU by
// tests/hostsim (test tooling only; never linked into libvrfhip.so).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "field.h"
#include "vrf_types.h"

VRF_NS_BEGIN

using vrfk::LMASK;
constexpr int NL = 9;
constexpr int LW = 29;

// out value bound of a Montgomery product: (a*b + m*q)/R < q * (1 + Va*Vb*q/R), q/R = MULV_NUM / 1e5 (0.014152 for
// BLS12-381 Fr); the pseudo-Mersenne product of 2^255 - 19 always comes out below 2^255 + 2^30 < 2q
constexpr int mul_v(int v1, int v2) {
  return vrfk::FIELD_KIND == 2 ? 2 : 1 + (v1 * v2 * vrfk::MULV_NUM + 99999) / 100000;
}

template <int L, int V>
struct Fe {
  static_assert(L >= 1 && L <= 7, "limb bound out of range");
  static_assert(V >= 1 && V <= vrfk::VMAX, "value bound out of range (top limb must stay < 2^29)");
  uint32_t v[NL];
  Fe() = default;
  template <int L2, int V2>
  VRF_HD Fe(const Fe<L2, V2>& o) {
    static_assert(L2 <= L && V2 <= V, "narrowing Fe conversion");
#pragma unroll
    for (int i = 0; i < NL; ++i) v[i] = o.v[i];
  }
};

using FeN = Fe<1, 2>;   // storage type: a Montgomery product (limbs < 2^29, value < 2q)
using FeP = Fe<1, 5>;   // point-coordinate type (closure bound of the TE formulas, te.cuh)

template <int L, int V>
VRF_HD Fe<L, V> fe_load(const uint32_t* p) {
  Fe<L, V> r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.v[i] = p[i];
  return r;
}
VRF_HD FeN fe_const(const uint32_t (&c)[NL]) {
  FeN r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.v[i] = c[i];
  return r;
}
template <int L, int V>
VRF_HD void fe_store(uint32_t* p, const Fe<L, V>& a) {
#pragma unroll
  for (int i = 0; i < NL; ++i) p[i] = a.v[i];
}

VRF_HD FeN fe_zero() {
  FeN r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.v[i] = 0;
  return r;
}
VRF_HD FeN fe_one() { return fe_const(vrfk::ONE_M); }

// ------------------------------------------------------------------ lazy add / sub
template <int L1, int V1, int L2, int V2>
VRF_HD Fe<L1 + L2, V1 + V2> fe_add(const Fe<L1, V1>& a, const Fe<L2, V2>& b) {
  Fe<L1 + L2, V1 + V2> r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.v[i] = a.v[i] + b.v[i];
  return r;
}

template <int K, int LB>
struct Bias;
#define VRF_BIAS(K_, LB_)                                            \
  template <>                                                        \
  struct Bias<K_, LB_> {                                             \
    static VRF_HD uint32_t at(int i) { return vrfk::BIAS_L##LB_##_K##K_[i]; } \
  };
VRF_BIAS(4, 1) VRF_BIAS(8, 1) VRF_BIAS(16, 1) VRF_BIAS(32, 1) VRF_BIAS(64, 1)
VRF_BIAS(4, 2) VRF_BIAS(8, 2) VRF_BIAS(16, 2) VRF_BIAS(32, 2) VRF_BIAS(64, 2)
#undef VRF_BIAS

constexpr int bias_k(int v) { return v < 4 ? 4 : v < 8 ? 8 : v < 16 ? 16 : v < 32 ? 32 : 64; }

// a - b (mod q) computed as a + K*q - b limb-wise.  The bias word has limbs >= LB*(2^29+2^13)
// and a top limb >= V2*q / 2^232, so no limb ever borrows.
template <int L1, int V1, int L2, int V2>
VRF_HD Fe<L1 + L2 + 1, V1 + bias_k(V2)> fe_sub(const Fe<L1, V1>& a, const Fe<L2, V2>& b) {
  static_assert(L2 <= 2, "subtrahend must have L <= 2 (normalise first)");
  constexpr int K = bias_k(V2);   // K >= V2 + 1
  Fe<L1 + L2 + 1, V1 + K> r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.v[i] = a.v[i] + (Bias<K, L2>::at(i) - b.v[i]);
  return r;
}

template <int L, int V>
VRF_HD Fe<L + 1, bias_k(V)> fe_neg(const Fe<L, V>& b) {
  static_assert(L <= 2, "fe_neg operand must have L <= 2");
  constexpr int K = bias_k(V);
  Fe<L + 1, K> r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.v[i] = Bias<K, L>::at(i) - b.v[i];
  return r;
}

// small multiples (shift-adds, one v_lshl_add_u32 per limb)
template <int L, int V>
VRF_HD Fe<2 * L, 2 * V> fe_dbl(const Fe<L, V>& a) {
  Fe<2 * L, 2 * V> r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.v[i] = a.v[i] << 1;
  return r;
}
template <int L, int V>
VRF_HD Fe<5 * L, 5 * V> fe_mul5(const Fe<L, V>& a) {
  Fe<5 * L, 5 * V> r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.v[i] = (a.v[i] << 2) + a.v[i];
  return r;
}

// weak normalisation: one parallel carry pass; limbs < 2^29 + 8 afterwards (L = 1).
template <int L, int V>
VRF_HD Fe<1, V> fe_norm(const Fe<L, V>& a) {
  Fe<1, V> r;
  r.v[0] = a.v[0] & LMASK;
#pragma unroll
  for (int i = 1; i < NL - 1; ++i) r.v[i] = (a.v[i] & LMASK) + (a.v[i - 1] >> LW);
  r.v[NL - 1] = a.v[NL - 1] + (a.v[NL - 2] >> LW);
  return r;
}

#if VRF_FIELD == 3
// Weak reduction for P-256, where a 256-bit modulus leaves the lazy sums only V <= 32 of headroom (2^261 / 2^256):
// any value < 32 q comes back below 2^256 + 2^229 < 2 q with exact limbs.  k = floor(v / 2^256) <= 31 and
// v - k q = (v mod 2^256) + k (2^224 - 2^192 - 2^96 + 1): one exact carry pass to read k, one signed pass to add the
// four terms.  ~60 32-bit instructions, a third of a product by one (the only reduction the other fields need).
template <int L, int V>
VRF_HD Fe<1, 2> fe_wred(const Fe<L, V>& a) {
  uint32_t x[NL];
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < NL - 1; ++i) {
    const uint32_t t = a.v[i] + c;
    x[i] = t & LMASK;
    c = t >> LW;
  }
  x[NL - 1] = a.v[NL - 1] + c;
  const int32_t k = (int32_t)(x[NL - 1] >> 24);          // bit 256 is bit 24 of limb 8
  x[NL - 1] &= 0x00ffffffu;
  Fe<1, 2> r;
  int32_t cc = k;                                        // + k at bit 0
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    int32_t t = (int32_t)x[i] + cc;
    if (i == 3) t -= k << 9;                             // - k 2^96  (96 = 3 * 29 + 9)
    if (i == 6) t -= k << 18;                            // - k 2^192 (192 = 6 * 29 + 18)
    if (i == 7) t += k << 21;                            // + k 2^224 (224 = 7 * 29 + 21)
    r.v[i] = (i < NL - 1) ? ((uint32_t)t & LMASK) : (uint32_t)t;
    cc = t >> LW;
  }
  return r;
}
#endif

template <int L, int V>
VRF_HD Fe<L, V> fe_select(bool c, const Fe<L, V>& a, const Fe<L, V>& b) {   // c ? a : b
  Fe<L, V> r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.v[i] = c ? a.v[i] : b.v[i];
  return r;
}

// ------------------------------------------------------------------ multiply
VRF_HD uint64_t mad(uint32_t a, uint32_t b, uint64_t c) { return (uint64_t)a * b + c; }
// Three digit rules, chosen at compile time by the field (field.h, vrfk::FIELD_KIND):
//
//  kind 0 (BLS12-381 Fr, q = 1 mod 2^29): the Montgomery digit of a column t is m = -t mod 2^29 and the column leaves the
//  carry (t + m) >> 29 = (t >> 29) + (t mod 2^29 != 0) = (t + 2^29 - 1) >> 29.  Carrying the bias 2^29 - 1 in the
//  accumulator turns "negate, mask, add m back (64-bit)" into one v_bitop3 (~t & mask) and no addition:
//  17 fewer instructions per product (measured on gfx950: 1.64e11 -> 1.75e11 products/s).  The bias is far
//  below the headroom the L1 * L2 <= 6 assertion leaves (about 2^59).
//
//  kind 1 (BN254 Fr, general q): m = t * (-q^-1) mod 2^29 -- NINV29 = 2^28 - 1 there, a shift and a subtraction -- and
//  the column takes m * q[0] like every other limb of q: 9 more multiply-adds per product than kind 0.
//
//  kind 2 (2^255 - 19): no Montgomery form at all (R = 1).  The 17 columns of the plain product are carried into 29-bit
//  digits, then 2^261 = 1216 (mod q) folds digit k + 9 onto digit k in a second carry pass, and what is left above bit
//  255 comes back as 19 * carry into limb 0: 81 + 9 multiply-adds instead of 153.  Output < 2^255 + 2^30.
VRF_HD constexpr uint64_t mont_bias() { return vrfk::FIELD_KIND == 0 ? (uint64_t)LMASK : 0; }
VRF_HD uint32_t mont_digit(uint32_t col) {
  if constexpr (vrfk::FIELD_KIND == 0) return ~col & LMASK;          // the accumulator carries the bias LMASK
  else return (col * vrfk::NINV29) & LMASK;
}
constexpr uint32_t PM_FOLD = 1216;      // 2^261 mod (2^255 - 19)
constexpr int PM_TOPBITS = 23;          // bit 255 is bit 23 of limb 8

// second pass of the pseudo-Mersenne product: d = the plain product in 18 digits (d[17] < 2^29 because the operands are
// < 64 q); digit k + 9 folds onto digit k with the factor 2^261 = 1216, and what then lies above bit 255 (bit 23 of
// limb 8) comes back as 19 * carry into limb 0.  Out: limbs < 2^29 + 2, value < 2^255 + 2^30 < 2q.
template <int V>
VRF_HD void pm_fold(Fe<1, V>& r, const uint32_t (&d)[2 * NL]) {
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < NL - 1; ++k) {
    acc = mad(d[k + NL], PM_FOLD, acc + d[k]);            // < 2^41
    r.v[k] = (uint32_t)acc & LMASK;
    acc >>= LW;
  }
  acc = mad(d[2 * NL - 1], PM_FOLD, acc + d[NL - 1]);
  r.v[NL - 1] = (uint32_t)acc & ((1u << PM_TOPBITS) - 1);
  const uint32_t c = (uint32_t)(acc >> PM_TOPBITS);       // < 2^18
  const uint32_t r0 = r.v[0] + 19u * c;                   // < 2^30
  r.v[0] = r0 & LMASK;
  r.v[1] += r0 >> LW;
}

template <int L1, int V1, int L2, int V2>
VRF_HD Fe<1, mul_v(V1, V2)> fe_mul(const Fe<L1, V1>& a, const Fe<L2, V2>& b) {
  static_assert(L1 * L2 <= 6, "fe_mul: 64-bit column accumulator could overflow");
  Fe<1, mul_v(V1, V2)> r;
  if constexpr (vrfk::FIELD_KIND == 2) {
    uint32_t d[2 * NL];                                   // the plain product in 29-bit digits: < 2^522
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * NL - 1; ++k) {
#pragma unroll
      for (int i = (k < NL ? 0 : k - NL + 1); i <= (k < NL ? k : NL - 1); ++i) acc = mad(a.v[i], b.v[k - i], acc);
      d[k] = (uint32_t)acc & LMASK;
      acc >>= LW;
    }
    d[2 * NL - 1] = (uint32_t)acc;
    pm_fold(r, d);
    return r;
  } else {
  uint32_t m[NL];
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    acc += mont_bias();
#pragma unroll
    for (int i = 0; i <= k; ++i) acc = mad(a.v[i], b.v[k - i], acc);
#pragma unroll
    for (int i = 0; i < k; ++i) acc = mad(m[i], vrfk::Q29[k - i], acc);
    m[k] = mont_digit((uint32_t)acc);
    if constexpr (vrfk::FIELD_KIND != 0) acc = mad(m[k], vrfk::Q29[0], acc);
    acc >>= LW;                        // kind 0: == (column + m[k] * Q29[0]) >> LW, Q29[0] == 1 (see mont_bias)
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; ++k) {
#pragma unroll
    for (int i = k - NL + 1; i < NL; ++i) acc = mad(a.v[i], b.v[k - i], acc);
#pragma unroll
    for (int i = k - NL + 1; i < NL; ++i) acc = mad(m[i], vrfk::Q29[k - i], acc);
    r.v[k - NL] = (uint32_t)acc & LMASK;
    acc >>= LW;
  }
  r.v[NL - 1] = (uint32_t)acc;
  return r;
  }
}

template <int L, int V>
VRF_HD Fe<1, mul_v(V, V)> fe_sqr(const Fe<L, V>& a) {
  static_assert(L * L <= 6, "fe_sqr: 64-bit column accumulator could overflow");
  Fe<1, mul_v(V, V)> r;
  uint32_t a2[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) a2[i] = a.v[i] << 1;
  if constexpr (vrfk::FIELD_KIND == 2) {
    uint32_t d[2 * NL];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * NL - 1; ++k) {
#pragma unroll
      for (int i = (k < NL ? 0 : k - NL + 1); 2 * i < k; ++i) acc = mad(a2[i], a.v[k - i], acc);
      if ((k & 1) == 0) acc = mad(a.v[k / 2], a.v[k / 2], acc);
      d[k] = (uint32_t)acc & LMASK;
      acc >>= LW;
    }
    d[2 * NL - 1] = (uint32_t)acc;
    pm_fold(r, d);
    return r;
  } else {
  uint32_t m[NL];
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    acc += mont_bias();
#pragma unroll
    for (int i = 0; 2 * i < k; ++i) acc = mad(a2[i], a.v[k - i], acc);
    if ((k & 1) == 0) acc = mad(a.v[k / 2], a.v[k / 2], acc);
#pragma unroll
    for (int i = 0; i < k; ++i) acc = mad(m[i], vrfk::Q29[k - i], acc);
    m[k] = mont_digit((uint32_t)acc);
    if constexpr (vrfk::FIELD_KIND != 0) acc = mad(m[k], vrfk::Q29[0], acc);
    acc >>= LW;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; ++k) {
#pragma unroll
    for (int i = k - NL + 1; 2 * i < k; ++i) acc = mad(a2[i], a.v[k - i], acc);
    if ((k & 1) == 0) acc = mad(a.v[k / 2], a.v[k / 2], acc);
#pragma unroll
    for (int i = k - NL + 1; i < NL; ++i) acc = mad(m[i], vrfk::Q29[k - i], acc);
    r.v[k - NL] = (uint32_t)acc & LMASK;
    acc >>= LW;
  }
  r.v[NL - 1] = (uint32_t)acc;
  return r;
  }
}

// ------------------------------------------------------------------ canonical forms
// Exact carry propagation + one conditional subtraction of q.  Input value must be < 2q.
template <int L>
VRF_HD void fe_reduce_once(uint32_t out[NL], const Fe<L, 2>& a) {
  uint32_t x[NL], d[NL];
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < NL - 1; ++i) {
    uint32_t t = a.v[i] + c;
    x[i] = t & LMASK;
    c = t >> LW;
  }
  x[NL - 1] = a.v[NL - 1] + c;
  uint32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    uint32_t t = x[i] - vrfk::Q29[i] - borrow;
    borrow = t >> 31;
    d[i] = (i < NL - 1) ? (t & LMASK) : t;
  }
#pragma unroll
  for (int i = 0; i < NL; ++i) out[i] = borrow ? x[i] : d[i];
}

// canonical Montgomery image: the unique representative in [0, q) of the same residue.
template <int L, int V>
VRF_HD FeN fe_canon(const Fe<L, V>& a) {
  static_assert(L <= 6, "");
  FeN t = fe_mul(a, fe_one());     // a * R / R = a, value < q(1 + V*0.0142) < 2q
  FeN r;
  fe_reduce_once(r.v, t);
  return r;
}

template <int L, int V>
VRF_HD bool fe_is_zero(const Fe<L, V>& a) {
  FeN c = fe_canon(a);
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) o |= c.v[i];
  return o == 0;
}

template <int L1, int V1, int L2, int V2>
VRF_HD bool fe_eq(const Fe<L1, V1>& a, const Fe<L2, V2>& b) {
  FeN x = fe_canon(a), y = fe_canon(b);
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) o |= x.v[i] ^ y.v[i];
  return o == 0;
}

// 8 x u32 little-endian integer (< 2^256) -> 9 x 29 plain limbs
VRF_HD void u256_to_limbs(uint32_t out[NL], const uint32_t w[8]) {
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int bit = i * LW, j = bit >> 5, s = bit & 31;
    uint32_t lo = w[j] >> s;
    if (s > 3 && j + 1 < 8) lo |= w[j + 1] << (32 - s);
    out[i] = (i < NL - 1) ? (lo & LMASK) : lo;
  }
}
// 9 x 29 exact limbs (value < 2^256) -> 8 x u32
VRF_HD void limbs_to_u256(uint32_t w[8], const uint32_t x[NL]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int bit = j * 32, i = bit / LW, s = bit % LW;
    uint32_t v = x[i] >> s;
    if (i + 1 < NL) v |= x[i + 1] << (LW - s);
    if (LW - s + LW < 32 && i + 2 < NL) v |= x[i + 2] << (2 * LW - s);
    w[j] = v;
  }
}

// integer < 2^256 (need not be < q) -> Montgomery
VRF_HD FeN fe_from_u256(const uint32_t w[8]) {
  Fe<1, vrfk::V256> t;                // 2^256 < 2.21 q (BLS12-381 Fr), 5.3 q (BN254 Fr)
  u256_to_limbs(t.v, w);
  return fe_mul(t, fe_const(vrfk::R2_29));
}
// Montgomery -> canonical integer in [0, q) as 8 x u32
template <int L, int V>
VRF_HD void fe_to_u256(uint32_t w[8], const Fe<L, V>& a) {
  static_assert(L <= 6, "");
  FeN one;
#pragma unroll
  for (int i = 0; i < NL; ++i) one.v[i] = (i == 0);
  FeN t = fe_mul(a, one);             // a / R, value < q + tiny
  uint32_t c[NL];
  fe_reduce_once(c, t);
  limbs_to_u256(w, c);
}

// Coordinates at the ABI come as canonical integers or -- VRFHIP_FLAG_COORDS_MONT256 -- as arkworks' in-memory field
// elements: the Montgomery image x 2^256 mod q, four little-endian u64.  Both are integers < q on the wire.
//   in : words -> Fe (x 2^261) and the canonical words of x (the sign of x, the encoding of y need them)
//   out: Fe -> words
VRF_HD FeN fe_from_abi(uint32_t canon[8], const uint32_t w[8], bool mont256) {
  Fe<1, vrfk::V256> t;
  u256_to_limbs(t.v, w);
  FeN k;
#pragma unroll
  for (int i = 0; i < NL; ++i) k.v[i] = mont256 ? vrfk::TWO266_29[i] : vrfk::R2_29[i];
  const FeN r = fe_mul(t, k);          // canonical: x R^2 / R; Montgomery-256: (x 2^256) 2^266 / 2^261
  if (mont256) {
    fe_to_u256(canon, r);
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) canon[i] = w[i];
  }
  return r;
}
template <int L, int V>
VRF_HD void fe_to_mont256(uint32_t w[8], const Fe<L, V>& a) {
  static_assert(L <= 6, "");
  FeN t = fe_mul(a, fe_const(vrfk::TWO256_29));      // (x 2^261) 2^256 / 2^261 = x 2^256, value < q + tiny
  uint32_t c[NL];
  fe_reduce_once(c, t);
  limbs_to_u256(w, c);
}
// canonical words of x -> the words of its Montgomery-256 image (outputs that were produced canonically)
VRF_HD void u256_canon_to_mont256(uint32_t w[8]) {
  Fe<1, vrfk::V256> t;
  u256_to_limbs(t.v, w);
  FeN k;
#pragma unroll
  for (int i = 0; i < NL; ++i) k.v[i] = vrfk::R2_29[i];
  fe_to_mont256(w, fe_mul(t, k));
}

// 512-bit integer (16 x u32 LE) mod q -> Montgomery (hash_to_field, 48-byte inputs fit)
VRF_HD Fe<1, 4> fe_from_u512(const uint32_t w[16]) {
  Fe<1, vrfk::V256> lo, hi;
  u256_to_limbs(lo.v, w);
  u256_to_limbs(hi.v, w + 8);
  FeN a = fe_mul(lo, fe_const(vrfk::R2_29));
  FeN b = fe_mul(hi, fe_const(vrfk::TWO256_R2));
  return fe_norm(fe_add(a, b));
}

// ------------------------------------------------------------------ fixed exponentiation
// x^e by a generated sliding-window program (constants.gen.h POW_*_PROG, tools/gen_constants.py): word 0
// selects the start value among the odd powers x^1..x^15, every later word is (squarings << 4) | index of
// the odd power to multiply by (15 = none).  The program is the same for every lane: scalar control flow.
// x^(q-2): 253 squarings + 60 multiplications; x^((t-1)/2): 220 + 52 (fixed 3-bit windows took 85 / 80).
template <int LEN>
VRF_HD FeN fe_pow_prog(const FeN& x, const uint16_t (&prog)[LEN]) {
  FeN t[8];
  t[0] = x;
  const FeN x2 = fe_sqr(x);
#pragma unroll
  for (int i = 1; i < 8; ++i) t[i] = fe_mul(t[i - 1], x2);
  auto pick = [&](uint32_t idx) {
    FeN s = t[0];
#pragma unroll
    for (int j = 1; j < 8; ++j)
      if (idx == (uint32_t)j) s = t[j];
    return s;
  };
  FeN acc = pick(prog[0] & 15u);
#pragma unroll 1
  for (int i = 1; i < LEN; ++i) {
    const uint32_t op = prog[i], nsq = op >> 4, idx = op & 15u;
#pragma unroll 1
    for (uint32_t k = 0; k < nsq; ++k) acc = fe_sqr(acc);
    if (idx != 15u) acc = fe_mul(acc, pick(idx));
  }
  return acc;
}

template <int L, int V>
VRF_HD FeN fe_inv_pow(const Fe<L, V>& a) {   // a^(q-2); 0 -> 0
  FeN x = fe_mul(a, fe_one());
  return fe_pow_prog(x, vrfk::POW_INV_PROG);
}

// ------------------------------------------------------------------ square root
// Table-driven Tonelli-Shanks.  q - 1 = 2^S t, g = Z^t generates the 2^S-torsion (Z the field's non-residue: 5, 2, 5),
// h = 1/g.  w^t = g^e; the discrete logarithm e is read off digit by digit: a power of w^t that depends on the digits
// found so far and the next one only, corrected by table entries h^(j 2^level), lands in the 2^8-torsion, where a
// perfect hash of the canonical low limb (lut = SQRT_LUT) gives the digit.  Then sqrt(w) = w^((t+1)/2) h^(e/2).
//   S = 32 (BLS12-381 Fr): byte digits; tables k = 0..3 at levels 0, 8, 16, 24.  220 + 24 squarings.
//   S = 28 (BN254 Fr): digits 8 | 8 | 8 | 4; the same four tables plus levels 4 and 12 (k = 4, 5).
//   S = 2 (2^255 - 19): the 4-torsion is four constants, no table.
//   S = 1 (P-256): the exponentiation is the root.
// Returns is_square(w) and sets `root` to sqrt(w) if w is a square, else to sqrt(Z*w) (Z doubles as the Elligator
// non-residue of the Bandersnatch suite).  Constant shape.
VRF_HD uint32_t sqrt_lut_index(const SqrtTables& T, const FeN& y) {
  FeN c = fe_canon(y);
  uint32_t h = (c.v[0] * vrfk::SQRT_LUT_MULT) >> (32 - vrfk::SQRT_LUT_BITS);
  return T.lut[h];
}
VRF_HD FeN sqrt_tbl(const SqrtTables& T, int k, uint32_t j) {
  return fe_load<1, 2>(T.P + ((size_t)k * 256 + j) * NL);
}
VRF_HD bool limbs_eq(const FeN& a, const uint32_t (&c)[NL]) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) o |= a.v[i] ^ c[i];
  return o == 0;
}

// root = w^((t+1)/2) h^(e/2) (e even) or sqrt(Z w) = Z^((t+1)/2) w^((t+1)/2) h^((e+1)/2) (e odd); returns e even
VRF_HD bool sqrt_finish(FeN& root, const FeN& x0, uint32_t e, const SqrtTables& T) {
  bool odd = e & 1;
  uint32_t half = (e >> 1) + (e & 1);               // in [0, 2^31]
  FeN r = fe_mul(x0, sqrt_tbl(T, 0, half & 255));
  r = fe_mul(r, sqrt_tbl(T, 1, (half >> 8) & 255));
  r = fe_mul(r, sqrt_tbl(T, 2, (half >> 16) & 255));
  r = fe_mul(r, sqrt_tbl(T, 3, half >> 24));
  FeN rz = fe_mul(r, fe_const(vrfk::SQRT_CZ_M));
  root = fe_select(odd, rz, r);
  return !odd;    // w == 0: b = 0, lut garbage, but x0 = 0 => root = 0; caller treats 0 as square
}

template <int L, int V>
VRF_HD bool fe_sqrt_or_zsqrt(FeN& root, const Fe<L, V>& w_in, const SqrtTables& T) {
  FeN w = fe_mul(w_in, fe_one());
  FeN v = fe_pow_prog(w, vrfk::POW_SQRT_PROG);        // w^((t-1)/2)
  FeN x0 = fe_mul(w, v);                            // w^((t+1)/2)
  FeN b = fe_mul(x0, v);                            // w^t, in the 2^S-torsion
  if constexpr (vrfk::SQRT_S == 1) {
    // q = 3 (mod 4), Z = -1: x0 = w^((q+1)/4) squares to w chi(w), i.e. it is sqrt(w) or sqrt(Z w) as it stands
    root = x0;
    const FeN bc = fe_canon(b);                       // chi(w): 1, -1 or 0
    uint32_t nz = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) nz |= bc.v[i];
    return limbs_eq(bc, vrfk::ONE_M) || nz == 0;
  } else if constexpr (vrfk::SQRT_S == 2) {
#if VRF_FIELD == 1
    uint32_t e;
    // x0 = w^((q+3)/8); b = x0^2 / w is 1, -1 (root x0 / sqrt(-1)) or a primitive 4th root of unity (w a non-residue)
    const FeN bc = fe_canon(b);
    e = limbs_eq(bc, vrfk::SQRT_G1_M) ? 1u : limbs_eq(bc, vrfk::SQRT_G2_M) ? 2u : limbs_eq(bc, vrfk::SQRT_G3_M) ? 3u : 0u;
    const bool odd = e & 1;
    const uint32_t half = (e >> 1) + (e & 1);       // 0, 1, 1, 2
    const FeN hm = fe_select(half == 0, fe_one(), fe_select(half == 1, fe_const(vrfk::SQRT_H1_M), fe_const(vrfk::SQRT_H2_M)));
    FeN r = fe_mul(x0, hm);
    FeN rz = fe_mul(r, fe_const(vrfk::SQRT_CZ_M));
    root = fe_select(odd, rz, r);
    return !odd;
#endif
  } else if constexpr (vrfk::SQRT_S == 28) {
    FeN b4 = b;
    for (int i = 0; i < 4; ++i) b4 = fe_sqr(b4);
    FeN b12 = b4;
    for (int i = 0; i < 8; ++i) b12 = fe_sqr(b12);
    FeN b20 = b12;
    for (int i = 0; i < 8; ++i) b20 = fe_sqr(b20);
    // b = g^e, e = e0 + e1 2^8 + e2 2^16 + e3 2^24, e3 < 16; the 2^8-torsion is generated by g^(2^20)
    uint32_t e0 = sqrt_lut_index(T, b20);
    uint32_t e1 = sqrt_lut_index(T, fe_mul(b12, sqrt_tbl(T, 5, e0)));
    uint32_t e2 = sqrt_lut_index(T, fe_mul(fe_mul(b4, sqrt_tbl(T, 4, e0)), sqrt_tbl(T, 5, e1)));
    uint32_t e3 = sqrt_lut_index(
        T, fe_mul(fe_mul(fe_mul(b, sqrt_tbl(T, 0, e0)), sqrt_tbl(T, 1, e1)), sqrt_tbl(T, 2, e2))) >> 4;
    return sqrt_finish(root, x0, e0 | (e1 << 8) | (e2 << 16) | (e3 << 24), T);
  } else {
    static_assert(vrfk::SQRT_S <= 2 || vrfk::SQRT_S == 28 || vrfk::SQRT_S == 32, "no square-root plan for this 2-adicity");
    FeN b8 = b;
    for (int i = 0; i < 8; ++i) b8 = fe_sqr(b8);
    FeN b16 = b8;
    for (int i = 0; i < 8; ++i) b16 = fe_sqr(b16);
    FeN b24 = b16;
    for (int i = 0; i < 8; ++i) b24 = fe_sqr(b24);
    // b = g^e, e = e0 + e1 2^8 + e2 2^16 + e3 2^24
    uint32_t e0 = sqrt_lut_index(T, b24);
    uint32_t e1 = sqrt_lut_index(T, fe_mul(b16, sqrt_tbl(T, 2, e0)));
    uint32_t e2 = sqrt_lut_index(T, fe_mul(fe_mul(b8, sqrt_tbl(T, 1, e0)), sqrt_tbl(T, 2, e1)));
    uint32_t e3 = sqrt_lut_index(
        T, fe_mul(fe_mul(fe_mul(b, sqrt_tbl(T, 0, e0)), sqrt_tbl(T, 1, e1)), sqrt_tbl(T, 2, e2)));
    return sqrt_finish(root, x0, e0 | (e1 << 8) | (e2 << 16) | (e3 << 24), T);
  }
}


// Low byte of the 2-adic discrete logarithm: w = (odd-order part) * g^e with g the generator of the
// 2^S-torsion behind the square-root tables; returns e mod 256.  w (non-zero) is a 2^k-th power, k <= 8, iff
// 2^k divides the result.  One fixed exponentiation (220 + 24 squarings for S = 32), constant shape.
template <int L, int V>
VRF_HD uint32_t fe_dlog2_low8(const Fe<L, V>& w_in, const SqrtTables& T) {
  static_assert(L > 0 && vrfk::SQRT_S >= 8, "the field has no 2^8-torsion");
  FeN w = fe_mul(w_in, fe_one());
  FeN v = fe_pow_prog(w, vrfk::POW_SQRT_PROG);        // w^((t-1)/2)
  FeN b = fe_mul(fe_mul(w, v), v);                  // w^t, in the 2^S-torsion
  for (int i = 0; i < vrfk::SQRT_S - 8; ++i) b = fe_sqr(b);       // g^(e 2^(S-8)): depends on e mod 2^8 only
  return sqrt_lut_index(T, b);
}

// ------------------------------------------------------------------ quadratic character (Jacobi symbol)
// Is w a non-zero square?  The Euler criterion costs an exponentiation (~255 squarings, ~58 k instruction
// slots).  Here: the Jacobi symbol (g / q) by "positive divsteps" (Bernstein-Yang safegcd with the sum in
// place of the difference, so f and g stay non-negative and quadratic reciprocity applies as is):
//     g odd and eta < 0 : swap (f, g), eta = -eta       sign flips iff f = g = 3 (mod 4)
//     g odd             : g += f                        (g / f) depends on g mod f only
//     then              : g /= 2, eta -= 1              sign flips iff f = 3, 5 (mod 8)
// The pair converges to f = g = gcd; the symbol is known once f = 1.  Every decision reads the low bits of f
// and g and the counter eta only, so JAC_K = 29 steps run on the low 32-bit words (one VGPR each) and yield a
// 2x2 matrix of non-negative 30-bit entries; applying it to the 9-limb numbers divides by 2^29 exactly, i.e.
// drops one 29-bit limb.  The step is branch-free and every lane of a wave runs the same shape; 24..30 rounds
// for 255-bit inputs (measured on 3e4 random values; a wave pays the maximum of its lanes, ~29).  ~1000 cheap
// 32-bit VOP2 instructions + 36 multiply-adds per round: about a third of the exponentiation.
// chi(R) = chi(2)^261 = 1 because q = 1 (mod 8) for both Montgomery fields (and R = 1 for 2^255 - 19): the symbol of
// the stored image is the symbol of the value.
static_assert(vrfk::CHI_R == 1, "jacobi_limbs callers assume chi(R) = 1");
constexpr int JAC_K = 29;
constexpr int JAC_MAX_ROUNDS = 40;      // rounds needed: <= 30 in 3e4 samples, tail decays ~50x per round

// wave-level "any lane" on the device, identity on the host build
VRF_HD bool vrf_any(bool x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __any((int)x) != 0;
#else
  return x;
#endif
}

// x: exact 29-bit limbs of an integer in [0, 2^256).  Returns +1 / -1, 0 if x = 0 (mod q), 2 if the rounds ran out.
VRF_HD int jacobi_limbs(const uint32_t x[NL]) {
  uint32_t f[NL], g[NL];
  uint32_t nz = 0, dq = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) { f[i] = vrfk::Q29[i]; g[i] = x[i]; nz |= x[i]; dq |= x[i] ^ vrfk::Q29[i]; }
  const bool zero = nz == 0 || dq == 0;          // canonical inputs are < 2q: the multiples of q are 0 and q
  int32_t eta = -1;
  uint32_t jac = 0;                               // bit 1 carries the sign
  bool done = zero;
#pragma unroll 1
  for (int round = 0; round < JAC_MAX_ROUNDS; ++round) {
    if (!vrf_any(!done)) break;
    uint32_t f0 = f[0] | (f[1] << LW), g0 = g[0] | (g[1] << LW);
    uint32_t u = 1, v = 0, q = 0, r = 1, jl = jac;
    int32_t e = eta;
#pragma unroll
    for (int i = 0; i < JAC_K; ++i) {
      const uint32_t m = 0u - (g0 & 1u);                       // g odd
      const uint32_t sw = m & (uint32_t)(e >> 31);             // ... and eta < 0: swap
      jl ^= f0 & g0 & sw;
      uint32_t t = (f0 ^ g0) & sw; f0 ^= t; g0 ^= t;
      t = (u ^ q) & sw; u ^= t; q ^= t;
      t = (v ^ r) & sw; v ^= t; r ^= t;
      e = (e ^ (int32_t)sw) - (int32_t)sw;
      g0 += f0 & m; q += u & m; r += v & m;
      g0 >>= 1; u <<= 1; v <<= 1; e -= 1;
      jl ^= f0 ^ (f0 >> 1);
    }
    // (f, g) <- (u f + v g, q f + r g) / 2^29: the low limb of both sums is zero by construction
    uint32_t nf[NL], ng[NL];
    uint64_t af = mad(u, f[0], (uint64_t)v * g[0]) >> LW, ag = mad(q, f[0], (uint64_t)r * g[0]) >> LW;
#pragma unroll
    for (int i = 1; i < NL; ++i) {
      af = mad(u, f[i], mad(v, g[i], af));
      ag = mad(q, f[i], mad(r, g[i], ag));
      nf[i - 1] = (uint32_t)af & LMASK; af >>= LW;
      ng[i - 1] = (uint32_t)ag & LMASK; ag >>= LW;
    }
    nf[NL - 1] = (uint32_t)af; ng[NL - 1] = (uint32_t)ag;
    uint32_t rest = 0;
#pragma unroll
    for (int i = 1; i < NL; ++i) rest |= nf[i];
    const bool now_one = nf[0] == 1u && rest == 0;
    // lanes that had finished keep their state (f = 1 is not absorbing under further steps)
#pragma unroll
    for (int i = 0; i < NL; ++i) { f[i] = done ? f[i] : nf[i]; g[i] = done ? g[i] : ng[i]; }
    eta = done ? eta : e;
    jac = done ? jac : jl;
    done = done || now_one;
  }
  if (zero) return 0;
  if (!done) return 2;
  return (jac & 2u) ? -1 : 1;
}

// w is a non-zero square.  The fall-back (rounds exhausted; not observed) is the exponentiation.
template <int L, int V>
VRF_HD bool fe_is_nonzero_square(const Fe<L, V>& w, const SqrtTables& T) {
  const FeN c = fe_canon(w);
  const int j = jacobi_limbs(c.v);
  if (j == 2) {
    FeN root;
    return fe_sqrt_or_zsqrt(root, c, T) && !fe_is_zero(c);
  }
  return j == 1;
}

// w is a square (zero included)
template <int L, int V>
VRF_HD bool fe_is_square_or_zero(const Fe<L, V>& w, const SqrtTables& T) {
  const FeN c = fe_canon(w);
  const int j = jacobi_limbs(c.v);
  if (j == 2) {
    FeN root;
    return fe_sqrt_or_zsqrt(root, c, T) || fe_is_zero(c);
  }
  return j >= 0;
}

// ------------------------------------------------------------------ inversion
// 1 / a by the positive divsteps of the Jacobi symbol above with the cofactors kept: f = q, g = the canonical Montgomery
// image a~, and f = d a~, g = e a~ (mod q) throughout.  The round's 2x2 matrix (entries <= 2^29, rows summing to at most
// 2^29) is applied to (f, g) -- an exact division by 2^29 -- and to (d, e), where the multiple of q that makes the division
// exact is (low limb) * (-q^-1 mod 2^29): -(low limb) when q = 1 (mod 2^29).  f = 1 ends a lane with d = a~^-1 < (rounds + 1) q.  About 27 rounds of
// ~650 instructions (~18 k) against 255 squarings + 30 products (~60 k) for a^(q-2): one inversion per 8 proofs in the big
// batches, but up to three per proof when a batch is small enough for one proof per lane.  0 -> 0, as the power gives.
// CT = true (the prover's inversions: the Z coordinates of sk*H, k*H, k*G depend on secrets): every lane runs all
// JAC_MAX_ROUNDS rounds, so the time no longer follows the operand.  Verification and decoding invert public
// data and keep the early exit (~27 rounds instead of 40).
template <bool CT = false, int L, int V>
VRF_HD FeN fe_inv(const Fe<L, V>& a) {
  const FeN c = fe_canon(a);
  uint32_t f[NL], g[NL], d[NL], e[NL];
  uint32_t nz = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) { f[i] = vrfk::Q29[i]; g[i] = c.v[i]; d[i] = 0; e[i] = (i == 0); nz |= c.v[i]; }
  int32_t eta = -1;
  bool done = nz == 0;
#pragma unroll 1
  for (int round = 0; round < JAC_MAX_ROUNDS; ++round) {
    if (!CT && !vrf_any(!done)) break;
    uint32_t f0 = f[0] | (f[1] << LW), g0 = g[0] | (g[1] << LW);
    uint32_t u = 1, v = 0, q = 0, r = 1;
    int32_t et = eta;
#pragma unroll
    for (int i = 0; i < JAC_K; ++i) {
      const uint32_t m = 0u - (g0 & 1u);                       // g odd
      const uint32_t sw = m & (uint32_t)(et >> 31);            // ... and eta < 0: swap
      uint32_t t = (f0 ^ g0) & sw; f0 ^= t; g0 ^= t;
      t = (u ^ q) & sw; u ^= t; q ^= t;
      t = (v ^ r) & sw; v ^= t; r ^= t;
      et = (et ^ (int32_t)sw) - (int32_t)sw;
      g0 += f0 & m; q += u & m; r += v & m;
      g0 >>= 1; u <<= 1; v <<= 1; et -= 1;
    }
    uint32_t nf[NL], ng[NL], nd[NL], ne[NL];
    uint64_t af = mad(u, f[0], (uint64_t)v * g[0]) >> LW, ag = mad(q, f[0], (uint64_t)r * g[0]) >> LW;
    const uint64_t td = mad(u, d[0], (uint64_t)v * e[0]), te = mad(q, d[0], (uint64_t)r * e[0]);
    // the multiple of q that clears the low limb: -(low limb) when q = 1 (mod 2^29), (low limb) * (-q^-1) in general
    const uint32_t md = ((uint32_t)td * vrfk::NINV29) & LMASK, me = ((uint32_t)te * vrfk::NINV29) & LMASK;
    uint64_t ad = mad(md, vrfk::Q29[0], td) >> LW, ae = mad(me, vrfk::Q29[0], te) >> LW;
#pragma unroll
    for (int i = 1; i < NL; ++i) {
      af = mad(u, f[i], mad(v, g[i], af));
      ag = mad(q, f[i], mad(r, g[i], ag));
      ad = mad(u, d[i], mad(v, e[i], mad(md, vrfk::Q29[i], ad)));
      ae = mad(q, d[i], mad(r, e[i], mad(me, vrfk::Q29[i], ae)));
      nf[i - 1] = (uint32_t)af & LMASK; af >>= LW;
      ng[i - 1] = (uint32_t)ag & LMASK; ag >>= LW;
      nd[i - 1] = (uint32_t)ad & LMASK; ad >>= LW;
      ne[i - 1] = (uint32_t)ae & LMASK; ae >>= LW;
    }
    nf[NL - 1] = (uint32_t)af; ng[NL - 1] = (uint32_t)ag; nd[NL - 1] = (uint32_t)ad; ne[NL - 1] = (uint32_t)ae;
    uint32_t rest = 0;
#pragma unroll
    for (int i = 1; i < NL; ++i) rest |= nf[i];
    const bool now_one = nf[0] == 1u && rest == 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) {           // a finished lane keeps its state
      f[i] = done ? f[i] : nf[i]; g[i] = done ? g[i] : ng[i];
      d[i] = done ? d[i] : nd[i]; e[i] = done ? e[i] : ne[i];
    }
    eta = done ? eta : et;
    done = done || now_one;
  }
  if (vrf_any(!done)) {                       // rounds exhausted (not observed): the power, for the whole wave
    const FeN slow = fe_inv_pow(a);
    if (!done) return slow;
  }
  // d < (JAC_MAX_ROUNDS + 1) q = 41 q.  A 256-bit q (P-256, VMAX = 32) puts that above the typed range: there the top
  // limb is < 2^30 (declared as L = 2) and the first product below is by the constant R^2 mod q < q, so its value is
  // < q (1 + 41 q / R) < 2.3 q, inside what mul_v(32, 2) = 3 claims for it.
  Fe<(vrfk::VMAX < 64 ? 2 : 1), (vrfk::VMAX < 64 ? vrfk::VMAX : 64)> dd;
#pragma unroll
  for (int i = 0; i < NL; ++i) dd.v[i] = nz == 0 ? 0u : d[i];
  // d = a^-1 / R as an integer: two Montgomery products by R^2 give a^-1 R
  return fe_mul(fe_mul(dd, fe_const(vrfk::R2_29)), fe_const(vrfk::R2_29));
}

VRF_NS_END
