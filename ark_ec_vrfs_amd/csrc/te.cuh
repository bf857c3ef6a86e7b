// te.cuh -- twisted Edwards group law a*x^2 + y^2 = 1 + d*x^2*y^2 in extended coordinates.
//
// Replaces ark_ec::twisted_edwards::{Affine, Projective} add / double / mul_bigint behind
// `AffinePoint` (/root/reference src/lib.rs:15) for the two suites on the hot path:
// Bandersnatch (a = -5) and JubJub (a = -1).  Formulas: add-2008-hwcd (unified) and
// dbl-2008-hwcd, rewritten so that a = -ANEG never needs a negation.  Every intermediate
// carries its (L, V) bound (fe.cuh); the closure bound for point coordinates is FeP = Fe<1,5>.
//
// Exceptional cases: a is a non-square for both curves, so the unified law is complete on the
// prime-order subgroup (and on the identity); inputs are required to lie in it, exactly as the
// reference's `AffinePoint` values do after arkworks' checked deserialisation.
#pragma once
#include "fe.cuh"

VRF_NS_BEGIN

// A curve tag states: a (as A_SIGN * ANEG: -5, -1, -1, +1), the cofactor, d and the default bases as field constants,
// and the subgroup order r for fr.cuh.  Only the curves of this translation unit's base field exist (field.h).
#if VRF_FIELD == 0
struct CurveBS {   // Bandersnatch
  static constexpr int ANEG = 5;
  static constexpr bool A_PLUS_ONE = false;
  static constexpr int COFACTOR_LOG2 = 2;
  static VRF_HD FeN d() { return fe_const(vrfk::BS_D_M); }
  static VRF_HD FeN aneg_m() { return fe_const(vrfk::FIVE_M); }
  static VRF_HD FeN gx() { return fe_const(vrfk::BS_GX_M); }
  static VRF_HD FeN gy() { return fe_const(vrfk::BS_GY_M); }
  static VRF_HD FeN bx() { return fe_const(vrfk::BS_BX_M); }
  static VRF_HD FeN by() { return fe_const(vrfk::BS_BY_M); }
  template <int L, int V>
  static VRF_HD Fe<5 * L, 5 * V> mul_aneg(const Fe<L, V>& a) { return fe_mul5(a); }
  static VRF_HD uint32_t r32(int i) { return vrfk::BS_R32[i]; }
  static VRF_HD uint32_t r_r1(int i) { return vrfk::BS_R_R1[i]; }
  static VRF_HD uint32_t r_r2(int i) { return vrfk::BS_R_R2[i]; }
  static constexpr uint32_t R_NINV32 = vrfk::BS_R_NINV32;
};

struct CurveJJ {   // JubJub (a = -1)
  static constexpr int ANEG = 1;
  static constexpr bool A_PLUS_ONE = false;
  static constexpr int COFACTOR_LOG2 = 3;
  static VRF_HD FeN d() { return fe_const(vrfk::JJ_D_M); }
  static VRF_HD FeN aneg_m() { return fe_const(vrfk::ONE_M); }
  static VRF_HD FeN gx() { return fe_const(vrfk::JJ_GX_M); }
  static VRF_HD FeN gy() { return fe_const(vrfk::JJ_GY_M); }
  static VRF_HD FeN bx() { return fe_const(vrfk::JJ_BX_M); }
  static VRF_HD FeN by() { return fe_const(vrfk::JJ_BY_M); }
  template <int L, int V>
  static VRF_HD Fe<L, V> mul_aneg(const Fe<L, V>& a) { return a; }
  static VRF_HD uint32_t r32(int i) { return vrfk::JJ_R32[i]; }
  static VRF_HD uint32_t r_r1(int i) { return vrfk::JJ_R_R1[i]; }
  static VRF_HD uint32_t r_r2(int i) { return vrfk::JJ_R_R2[i]; }
  static constexpr uint32_t R_NINV32 = vrfk::JJ_R_NINV32;
};
#elif VRF_FIELD == 1
struct CurveED {   // Ed25519 (RFC 8032): a = -1, cofactor 8
  static constexpr int ANEG = 1;
  static constexpr bool A_PLUS_ONE = false;
  static constexpr int COFACTOR_LOG2 = 3;
  static VRF_HD FeN d() { return fe_const(vrfk::ED_D_M); }
  static VRF_HD FeN aneg_m() { return fe_const(vrfk::ONE_M); }
  static VRF_HD FeN gx() { return fe_const(vrfk::ED_GX_M); }
  static VRF_HD FeN gy() { return fe_const(vrfk::ED_GY_M); }
  static VRF_HD FeN bx() { return fe_const(vrfk::ED_BX_M); }
  static VRF_HD FeN by() { return fe_const(vrfk::ED_BY_M); }
  template <int L, int V>
  static VRF_HD Fe<L, V> mul_aneg(const Fe<L, V>& a) { return a; }
  static VRF_HD uint32_t r32(int i) { return vrfk::ED_R32[i]; }
  static VRF_HD uint32_t r_r1(int i) { return vrfk::ED_R_R1[i]; }
  static VRF_HD uint32_t r_r2(int i) { return vrfk::ED_R_R2[i]; }
  static constexpr uint32_t R_NINV32 = vrfk::ED_R_NINV32;
};
#elif VRF_FIELD == 2
struct CurveBJ {   // Baby-JubJub as ark-ed-on-bn254 states it: a = +1, cofactor 8
  static constexpr int ANEG = -1;               // -a
  static constexpr bool A_PLUS_ONE = true;      // the group law below switches to its a = 1 form
  static constexpr int COFACTOR_LOG2 = 3;
  static VRF_HD FeN d() { return fe_const(vrfk::BJ_D_M); }
  static VRF_HD FeN gx() { return fe_const(vrfk::BJ_GX_M); }
  static VRF_HD FeN gy() { return fe_const(vrfk::BJ_GY_M); }
  static VRF_HD FeN bx() { return fe_const(vrfk::BJ_BX_M); }
  static VRF_HD FeN by() { return fe_const(vrfk::BJ_BY_M); }
  static VRF_HD uint32_t r32(int i) { return vrfk::BJ_R32[i]; }
  static VRF_HD uint32_t r_r1(int i) { return vrfk::BJ_R_R1[i]; }
  static VRF_HD uint32_t r_r2(int i) { return vrfk::BJ_R_R2[i]; }
  static constexpr uint32_t R_NINV32 = vrfk::BJ_R_NINV32;
};
#endif

struct PtE {   // extended projective: x = X/Z, y = Y/Z, T = X*Y/Z
  FeP X, Y, Z, T;
};
struct PtC {   // cached table entry: (X, Y, Z, d*T)
  FeP X, Y, Z;
  FeN dT;
};
constexpr int PTC_WORDS = 4 * NL;   // 36 words = 144 bytes

VRF_HD PtE te_identity() {
  PtE r;
  r.X = fe_zero(); r.Y = fe_one(); r.Z = fe_one(); r.T = fe_zero();
  return r;
}
VRF_HD PtC te_identity_cached() {
  PtC r;
  r.X = fe_zero(); r.Y = fe_one(); r.Z = fe_one(); r.dT = fe_zero();
  return r;
}
VRF_HD PtE te_from_affine(const FeP& x, const FeP& y) {
  PtE r;
  r.X = x; r.Y = y; r.Z = fe_one(); r.T = fe_mul(x, y);
  return r;
}

template <class C>
VRF_HD PtC te_to_cached(const PtE& p) {
  PtC r;
  r.X = p.X; r.Y = p.Y; r.Z = p.Z;
  r.dT = fe_mul(p.T, C::d());
  return r;
}
VRF_HD void ptc_store(uint32_t* m, const PtC& c) {
  fe_store(m, c.X); fe_store(m + NL, c.Y); fe_store(m + 2 * NL, c.Z); fe_store(m + 3 * NL, c.dT);
}
VRF_HD PtC ptc_load(const uint32_t* m) {
  PtC c;
  c.X = fe_load<1, 5>(m); c.Y = fe_load<1, 5>(m + NL); c.Z = fe_load<1, 5>(m + 2 * NL);
  c.dT = fe_load<1, 2>(m + 3 * NL);
  return c;
}

// 2P.  need_t = false skips the T coordinate (legal when the next operation is a doubling);
// it is wave-uniform at every call site, so the branch is scalar.
template <class C>
VRF_HD PtE te_dbl(const PtE& p, bool need_t) {
  auto A = fe_sqr(p.X);                               // (1,2)
  auto B = fe_sqr(p.Y);                               // (1,2)
  auto S = fe_sqr(fe_add(p.X, p.Y));                  // (1,3)
  auto ZZ = fe_sqr(p.Z);                              // (1,2)
  PtE r;
  if constexpr (C::A_PLUS_ONE) {
    // a = 1: E = 2XY = S - A - B, G = A + B, F = G - 2Z^2, H = A - B; X3 = E F, Y3 = G H, Z3 = F G, T3 = E H
    auto G = fe_add(A, B);                            // (2,4)
    auto E = fe_norm(fe_sub(S, G));                   // (1,11)
    auto F = fe_norm(fe_sub(G, fe_norm(fe_dbl(ZZ)))); // (1,12)
    auto H = fe_norm(fe_sub(A, B));                   // (1,6)
    r.X = fe_mul(E, F);
    r.Y = fe_mul(G, H);
    r.Z = fe_mul(F, G);
    r.T = fe_zero();
    if (need_t) r.T = fe_mul(E, H);
  } else {
    auto E = fe_norm(fe_sub(fe_add(A, B), S));          // A + B - S            (1,8)
    auto aA = C::mul_aneg(A);                           // -a*A
    auto H = fe_add(aA, B);                             // -a*A + B             (<=6,12)
    auto G = fe_norm(fe_sub(aA, B));                    // -a*A - B             (1,14)
    auto F = fe_add(G, fe_dbl(ZZ));                     // G + 2Z^2             (3,18)
    r.X = fe_mul(E, F);
    r.Y = fe_mul(G, H);
    r.Z = fe_mul(F, G);
    r.T = fe_zero();
    if (need_t) r.T = fe_mul(E, H);
  }
  return r;
}

// H = B - a*A of the addition laws: B + ANEG*A, or B - A for a = 1
template <class C, int L1, int V1, int L2, int V2>
VRF_HD auto te_b_minus_aa(const Fe<L1, V1>& B, const Fe<L2, V2>& A) {
  if constexpr (C::A_PLUS_ONE) return fe_norm(fe_sub(B, A));
  else return fe_norm(fe_add(B, C::mul_aneg(A)));
}

// a x^2 + y^2 - 1 from x^2, y^2: the left side of the curve equation a x^2 + y^2 - 1 = d x^2 y^2 (on-curve tests)
template <class C>
VRF_HD auto te_curve_lhs(const FeN& x2, const FeN& y2) {
  if constexpr (C::A_PLUS_ONE) return fe_norm(fe_sub(fe_add(y2, x2), fe_one()));
  else return fe_norm(fe_add(y2, fe_neg(fe_norm(fe_add(C::mul_aneg(x2), fe_one())))));    // y^2 - (ANEG x^2 + 1)
}

// conditional negation of a storage-type element; result type covers both branches
template <int L, int V>
VRF_HD Fe<L + 1, (V > bias_k(V) ? V : bias_k(V))> fe_cneg(bool neg, const Fe<L, V>& a) {
  using R = Fe<L + 1, (V > bias_k(V) ? V : bias_k(V))>;
  R n = fe_neg(a);
  R p = a;
  return fe_select(neg, n, p);
}

// P + (+/-)Q with Q a cached entry (projective).  Unified: also correct for P == Q and for
// either operand being the identity.  need_t = false skips the T coordinate (legal when the next
// operation is a doubling or the result is only read as X, Y, Z); wave-uniform at every call site.
template <class C>
VRF_HD PtE te_add_cached(const PtE& p, const PtC& q, bool neg, bool need_t = true) {
  auto X2 = fe_cneg(neg, q.X);                        // (2,8)
  auto dT2 = fe_cneg(neg, q.dT);                      // (2,4)
  auto A = fe_mul(p.X, X2);                           // (1,2)
  auto B = fe_mul(p.Y, q.Y);                          // (1,2)
  auto Cc = fe_mul(p.T, dT2);                         // (1,2)
  auto D = fe_mul(p.Z, q.Z);                          // (1,2)
  auto S = fe_mul(fe_add(p.X, p.Y), fe_add(X2, q.Y)); // (2,10)x(3,13) -> (1,3)
  auto E = fe_norm(fe_sub(S, fe_add(A, B)));          // (1,11)
  auto F = fe_sub(D, Cc);                             // (3,6)
  auto G = fe_add(D, Cc);                             // (2,4)
  auto H = te_b_minus_aa<C>(B, A);                    // B - a*A              (1,12)
  PtE r;
  r.X = fe_mul(E, F);
  r.Y = fe_mul(G, H);
  r.Z = fe_mul(F, G);
  r.T = fe_zero();
  if (need_t) r.T = fe_mul(E, H);
  return r;
}

// P + (+/-)Q with Q affine: (x, y, d*x*y), Z2 = 1.
struct PtA {
  FeN x, y, dt;
};
constexpr int PTA_WORDS = 3 * NL;   // 27 words = 108 bytes
VRF_HD PtA pta_load(const uint32_t* m) {
  PtA a;
  a.x = fe_load<1, 2>(m); a.y = fe_load<1, 2>(m + NL); a.dt = fe_load<1, 2>(m + 2 * NL);
  return a;
}
VRF_HD void pta_store(uint32_t* m, const PtA& a) {
  fe_store(m, a.x); fe_store(m + NL, a.y); fe_store(m + 2 * NL, a.dt);
}
VRF_HD PtA pta_identity() {
  PtA a;
  a.x = fe_zero(); a.y = fe_one(); a.dt = fe_zero();
  return a;
}

template <class C>
VRF_HD PtE te_add_affine(const PtE& p, const PtA& q, bool neg) {
  auto X2 = fe_cneg(neg, q.x);                        // (2,4)
  auto dT2 = fe_cneg(neg, q.dt);                      // (2,4)
  auto A = fe_mul(p.X, X2);
  auto B = fe_mul(p.Y, q.y);
  auto Cc = fe_mul(p.T, dT2);
  auto S = fe_mul(fe_add(p.X, p.Y), fe_add(X2, q.y));
  auto E = fe_norm(fe_sub(S, fe_add(A, B)));
  auto F = fe_sub(p.Z, Cc);                           // (3,9)
  auto G = fe_add(p.Z, Cc);                           // (2,7)
  auto H = te_b_minus_aa<C>(B, A);
  PtE r;
  r.X = fe_mul(E, F);
  r.Y = fe_mul(G, H);
  r.Z = fe_mul(F, G);
  r.T = fe_mul(E, H);
  return r;
}

#if VRF_FIELD == 0
// Bandersnatch GLV endomorphism psi(x, y) = (c(1 - y^2)/(xy), b(y^2 + b)/(y^2 - b)); on the
// prime-order subgroup psi(P) = LAMBDA * P.  Projective: 2S + 8M.  x*y = 0 happens in the
// subgroup only for the identity, which maps to the identity.
template <class C>
VRF_HD PtE te_psi(const PtE& p) {
  const FeN b = fe_const(vrfk::BS_PSI_B_M), c = fe_const(vrfk::BS_PSI_C_M);
  auto YY = fe_sqr(p.Y);                      // (1,2)
  auto ZZ = fe_sqr(p.Z);
  auto XY = fe_mul(p.X, p.Y);
  auto bZZ = fe_mul(ZZ, b);
  auto A = fe_sub(ZZ, YY);                    // Z^2 - Y^2            (3,6)
  auto Bv = fe_sub(YY, bZZ);                  // Y^2 - b Z^2          (3,6)
  auto Cv = fe_add(YY, bZZ);                  // Y^2 + b Z^2          (2,4)
  auto cA = fe_mul(A, c);
  auto bC = fe_mul(Cv, b);
  PtE r;
  r.X = fe_mul(cA, Bv);
  r.Y = fe_mul(bC, XY);
  r.Z = fe_mul(XY, Bv);
  r.T = fe_mul(cA, bC);
  bool exc = fe_is_zero(r.Z);
  PtE id = te_identity();
  r.X = fe_select(exc, id.X, r.X);
  r.Y = fe_select(exc, id.Y, r.Y);
  r.Z = fe_select(exc, id.Z, r.Z);
  r.T = fe_select(exc, id.T, r.T);
  return r;
}

#else
template <class C>
VRF_HD PtE te_psi(const PtE& p);      // no endomorphism in this field's suites: named only in discarded branches
#endif  // VRF_FIELD == 0

// general extended + extended (used off the hot loop: table building, h2c)
template <class C>
VRF_HD PtE te_add(const PtE& p, const PtE& q) {
  return te_add_cached<C>(p, te_to_cached<C>(q), false);
}

// -------------------------------------------------------------------------------- codec
// ArkworksCodec: y little-endian, bit 255 = (x > q - x) -- or x mod 2, see te_x_sign.
VRF_HD bool u256_gt(const uint32_t a[8], const uint32_t (&b)[8]) {   // a > b
  bool gt = false, decided = false;
#pragma unroll
  for (int i = 7; i >= 0; --i) {
    if (!decided && a[i] != b[i]) { gt = a[i] > b[i]; decided = true; }
  }
  return gt;
}
VRF_HD bool u256_ge(const uint32_t a[8], const uint32_t (&b)[8]) {   // a >= b
  bool gt = true, decided = false;
#pragma unroll
  for (int i = 7; i >= 0; --i) {
    if (!decided && a[i] != b[i]) { gt = a[i] > b[i]; decided = true; }
  }
  return gt;
}

// affine (Montgomery) -> 32-byte compressed encoding as 8 LE u32 words
// the sign flag of a compressed point from the canonical words of x: arkworks' "x > q - x", or RFC 8032's x mod 2 when the
// suite descriptor says so (SS_SIGN_PARITY; wave-uniform)
VRF_HD bool te_x_sign(const uint32_t xw[8], uint32_t sflags) {
  return (sflags & SS_SIGN_PARITY) ? (xw[0] & 1u) != 0 : u256_gt(xw, vrfk::QM1H32);
}
VRF_HD void te_encode_affine(uint32_t out[8], const FeN& x, const FeN& y, uint32_t sflags) {
  uint32_t xw[8];
  fe_to_u256(xw, x);
  fe_to_u256(out, y);
  if (te_x_sign(xw, sflags)) out[7] |= 0x80000000u;
}

VRF_NS_END
