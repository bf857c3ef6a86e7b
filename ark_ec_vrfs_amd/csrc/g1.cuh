// g1.cuh -- BLS12-381 G1 (y^2 = x^3 + 4 over Fp381) group law for the multi-scalar multiplication behind the
// batched pairing check (the KZG-aggregation step either side of the pairing tail of
// `ring::Verifier::verify`, /root/reference src/lib.rs:14 `ring`).  Replaces ark_ec short-Weierstrass
// `Projective` add / double for ark-bls12-381's G1 on that path.
//
// Homogeneous projective coordinates with the COMPLETE formulas of Renes-Costello-Batina (2016) for a = 0
// (Algorithms 7, 8, 9; b3 = 3 b = 12).  #E(Fp) = h r is odd (no point of order 2), so they have no exceptional
// cases at all: identity, doubling through the addition law and P + (-P) need no branches -- which matters in a
// bucket method, where a bench that tiles a few items lands equal points in one bucket.  Mixed addition 11 M,
// addition 12 M, doubling 6 M + 2 S, on the signed lazy 14 x 28-bit limbs of bls12.cuh with its typed bounds.
#pragma once
#include "bls12.cuh"

namespace bls {

struct G1P {   // (X : Y : Z), identity = (0 : 1 : 0)
  FpS X, Y, Z;
};
constexpr int G1P_WORDS = 3 * NLB;      // 42 words = 168 bytes
constexpr int G1A_WORDS = 2 * NLB;      // affine Montgomery (x, y): 28 words

VRF_HD G1P g1_identity() {
  G1P r;
  r.X = fp_zero(); r.Y = fp_one(); r.Z = fp_zero();
  return r;
}
VRF_HD void g1p_store(uint32_t* m, const G1P& p) {
#pragma unroll
  for (int i = 0; i < NLB; ++i) { m[i] = (uint32_t)p.X.v[i]; m[NLB + i] = (uint32_t)p.Y.v[i]; m[2 * NLB + i] = (uint32_t)p.Z.v[i]; }
}
VRF_HD G1P g1p_load(const uint32_t* m) {
  G1P p;
#pragma unroll
  for (int i = 0; i < NLB; ++i) { p.X.v[i] = (int32_t)m[i]; p.Y.v[i] = (int32_t)m[NLB + i]; p.Z.v[i] = (int32_t)m[2 * NLB + i]; }
  return p;
}

// 12 x = b3 x as shift-adds on normalised limbs
template <int L, int V>
VRF_HD auto fp_mul12(const Fp<L, V>& a) {
  auto n = fp_norm(a);                    // L = 1
  auto x4 = fp_dbl(fp_dbl(n));            // L = 4
  auto x8 = fp_norm(fp_dbl(x4));          // normalised
  return fp_norm(fp_add(x8, x4));         // (1, 12 V)
}

// P + Q, both projective (Algorithm 7, a = 0)
VRF_HD G1P g1_add(const G1P& p, const G1P& q) {
  auto t0 = fp_mul(p.X, q.X);
  auto t1 = fp_mul(p.Y, q.Y);
  auto t2 = fp_mul(p.Z, q.Z);
  auto t3 = fp_norm(fp_sub(fp_mul(fp_add(p.X, p.Y), fp_add(q.X, q.Y)), fp_add(t0, t1)));     // X1 Y2 + X2 Y1
  auto t4 = fp_norm(fp_sub(fp_mul(fp_add(p.Y, p.Z), fp_add(q.Y, q.Z)), fp_add(t1, t2)));     // Y1 Z2 + Y2 Z1
  auto y3 = fp_norm(fp_sub(fp_mul(fp_add(p.X, p.Z), fp_add(q.X, q.Z)), fp_add(t0, t2)));     // X1 Z2 + X2 Z1
  auto t0_3 = fp_norm(fp_add(fp_dbl(t0), t0));                                                  // 3 X1 X2
  auto t2b = fp_reduce(fp_mul12(t2));                                                           // b3 Z1 Z2
  auto z3 = fp_norm(fp_add(t1, t2b));
  auto t1m = fp_norm(fp_sub(t1, t2b));
  auto y3b = fp_reduce(fp_mul12(y3));                                                           // b3 (X1 Z2 + X2 Z1)
  G1P r;
  r.X = fp_fit(fp_sub(fp_mul(t3, t1m), fp_mul(t4, y3b)));
  r.Y = fp_fit(fp_add(fp_mul(t1m, z3), fp_mul(y3b, t0_3)));
  r.Z = fp_fit(fp_add(fp_mul(z3, t4), fp_mul(t0_3, t3)));
  return r;
}

// P + (+/-)(x2, y2) with an affine second operand that is NOT the point at infinity (Algorithm 8, a = 0)
VRF_HD G1P g1_madd(const G1P& p, const FpS& x2, const FpS& y2_in, bool neg) {
  const FpS y2 = fp_select(neg, fp_neg(y2_in), y2_in);
  auto t0 = fp_mul(p.X, x2);
  auto t1 = fp_mul(p.Y, y2);
  auto t3 = fp_norm(fp_sub(fp_mul(fp_add(x2, y2), fp_add(p.X, p.Y)), fp_add(t0, t1)));         // X1 y2 + x2 Y1
  auto t4 = fp_norm(fp_add(fp_mul(y2, p.Z), p.Y));                                              // y2 Z1 + Y1
  auto y3 = fp_norm(fp_add(fp_mul(x2, p.Z), p.X));                                              // x2 Z1 + X1
  auto t0_3 = fp_norm(fp_add(fp_dbl(t0), t0));
  auto t2b = fp_reduce(fp_mul12(p.Z));                                                          // b3 Z1
  auto z3 = fp_norm(fp_add(t1, t2b));
  auto t1m = fp_norm(fp_sub(t1, t2b));
  auto y3b = fp_reduce(fp_mul12(y3));
  G1P r;
  r.X = fp_fit(fp_sub(fp_mul(t3, t1m), fp_mul(t4, y3b)));
  r.Y = fp_fit(fp_add(fp_mul(t1m, z3), fp_mul(y3b, t0_3)));
  r.Z = fp_fit(fp_add(fp_mul(z3, t4), fp_mul(t0_3, t3)));
  return r;
}

// 2 P (Algorithm 9, a = 0)
VRF_HD G1P g1_dbl(const G1P& p) {
  auto t0 = fp_sqr(p.Y);
  auto z8 = fp_norm(fp_dbl(fp_dbl(fp_dbl(t0))));                  // 8 Y^2
  auto t1 = fp_mul(p.Y, p.Z);
  auto t2 = fp_reduce(fp_mul12(fp_sqr(p.Z)));                     // b3 Z^2
  auto x3 = fp_mul(t2, z8);
  auto y3 = fp_norm(fp_add(t0, t2));
  auto z3 = fp_mul(t1, z8);
  auto t2_3 = fp_norm(fp_add(fp_dbl(t2), t2));                    // 3 b3 Z^2
  auto t0m = fp_norm(fp_sub(t0, t2_3));
  auto xy = fp_mul(p.X, p.Y);
  G1P r;
  r.X = fp_fit(fp_dbl(fp_mul(t0m, xy)));
  r.Y = fp_fit(fp_add(x3, fp_mul(t0m, y3)));
  r.Z = fp_fit(z3);
  return r;
}

VRF_HD bool g1_is_identity(const G1P& p) { return fp_is_zero(p.Z); }

// affine coordinates of a finite point (one inversion); (0, 0) for the identity, as the wire format encodes it
VRF_HD void g1_to_affine(FpS& x, FpS& y, const G1P& p) {
  FpS zi;
  fp_inv(&zi, &p.Z);
  x = fp_fit(fp_mul(p.X, zi));
  y = fp_fit(fp_mul(p.Y, zi));
}

}  // namespace bls
