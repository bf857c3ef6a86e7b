// sw.cuh -- short-Weierstrass group law for secp256r1 (y^2 = x^3 - 3x + b over the NIST P-256 prime, prime order,
// cofactor 1): the curve arithmetic behind `suites::secp256r1` (/root/reference src/lib.rs:14), where upstream runs
// ark_ec::short_weierstrass::{Affine, Projective} (`AffinePoint`, src/lib.rs:15).
//
// Homogeneous projective coordinates with the COMPLETE formulas of Renes, Costello and Batina ("Complete addition
// formulas for prime order elliptic curves", Eurocrypt 2016, algorithms 4 and 6 for a = -3): one straight-line
// sequence adds any two points -- equal, opposite, the point at infinity -- so every lane of a wave runs the same shape
// whatever the (attacker-chosen) inputs are, and a verifier's answer on P = +-Q corner cases is the group law's answer
// by construction.  arkworks' Jacobian formulas branch on those cases; a GPU lane cannot branch for free.
//
// Bounds: the 256-bit modulus leaves the typed lazy limbs (fe.cuh) V <= 32 instead of 64, which the add/sub chains of the
// formulas overrun; fe_wred (the Solinas-shaped weak reduction, ~1/3 of a product) is placed where the compiler's bound
// arithmetic says so.  Every function takes and returns coordinates as FeN (limbs < 2^29, value < 2p).
#pragma once
#include "fe.cuh"
#include "fr.cuh"

#if VRF_FIELD != 3
#error "sw.cuh is the secp256r1 group law: compile with -DVRF_FIELD=3"
#endif

VRF_NS_BEGIN

struct CurveP256 {
  static VRF_HD FeN b() { return fe_const(vrfk::P256_B_M); }
  static VRF_HD uint32_t r32(int i) { return vrfk::P256_R32[i]; }
  static VRF_HD uint32_t r_r1(int i) { return vrfk::P256_R_R1[i]; }
  static VRF_HD uint32_t r_r2(int i) { return vrfk::P256_R_R2[i]; }
  static constexpr uint32_t R_NINV32 = vrfk::P256_R_NINV32;
};

struct PtW {          // homogeneous projective (X : Y : Z); the point at infinity is (0 : 1 : 0)
  FeN X, Y, Z;
};
constexpr int PTW_WORDS = 3 * NL;     // 27

VRF_HD PtW sw_identity() {
  PtW p;
  p.X = fe_zero(); p.Y = fe_one(); p.Z = fe_zero();
  return p;
}
VRF_HD PtW sw_from_affine(const FeN& x, const FeN& y) {
  PtW p;
  p.X = x; p.Y = y; p.Z = fe_one();
  return p;
}
VRF_HD PtW sw_select(bool c, const PtW& a, const PtW& b) {
  PtW r;
  r.X = fe_select(c, a.X, b.X); r.Y = fe_select(c, a.Y, b.Y); r.Z = fe_select(c, a.Z, b.Z);
  return r;
}
VRF_HD PtW sw_cneg(bool neg, const PtW& p) {
  PtW r = p;
  r.Y = fe_select(neg, fe_wred(fe_neg(p.Y)), p.Y);
  return r;
}
VRF_HD void ptw_store(uint32_t* p, size_t stride, const PtW& a) {      // word w of the point at p[w * stride]
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    p[(size_t)i * stride] = a.X.v[i];
    p[(size_t)(NL + i) * stride] = a.Y.v[i];
    p[(size_t)(2 * NL + i) * stride] = a.Z.v[i];
  }
}
VRF_HD PtW ptw_load(const uint32_t* p, size_t stride) {
  PtW a;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    a.X.v[i] = p[(size_t)i * stride];
    a.Y.v[i] = p[(size_t)(NL + i) * stride];
    a.Z.v[i] = p[(size_t)(2 * NL + i) * stride];
  }
  return a;
}

// P + Q, complete (RCB 2016 algorithm 4, a = -3): 12 products + 2 by b, 5 weak reductions.
VRF_HD PtW sw_add(const PtW& p, const PtW& q) {
  const FeN b = CurveP256::b();
  const FeN t0 = fe_mul(p.X, q.X), t1 = fe_mul(p.Y, q.Y), t2 = fe_mul(p.Z, q.Z);
  const auto t3 = fe_sub(fe_mul(fe_add(p.X, p.Y), fe_add(q.X, q.Y)), fe_add(t0, t1));     // X1 Y2 + X2 Y1   (4, 10)
  const auto t4 = fe_sub(fe_mul(fe_add(p.Y, p.Z), fe_add(q.Y, q.Z)), fe_add(t1, t2));     // Y1 Z2 + Y2 Z1
  const auto y3 = fe_sub(fe_mul(fe_add(p.X, p.Z), fe_add(q.X, q.Z)), fe_add(t0, t2));     // X1 Z2 + X2 Z1
  const FeN a1 = fe_wred(fe_sub(y3, fe_mul(t2, b)));                                      // X1Z2 + X2Z1 - b Z1Z2
  const auto a3 = fe_add(fe_dbl(a1), a1);                                                  // 3 (...)         (3, 6)
  const auto z3 = fe_norm(fe_sub(t1, fe_norm(a3)));                                        // Y1Y2 - 3(...)   (1, 10)
  const auto x3 = fe_norm(fe_add(t1, a3));                                                 // Y1Y2 + 3(...)   (1, 8)
  const auto t2x3 = fe_norm(fe_add(fe_dbl(t2), t2));                                       // 3 Z1Z2          (1, 6)
  const FeN bb = fe_wred(fe_sub(fe_sub(fe_mul(y3, b), t2x3), t0));                         // b(X1Z2+X2Z1) - 3 Z1Z2 - X1X2
  const auto y3b = fe_add(fe_dbl(bb), bb);                                                 // 3 (...)         (3, 6)
  const auto t0b = fe_norm(fe_sub(fe_add(fe_dbl(t0), t0), t2x3));                          // 3 X1X2 - 3 Z1Z2 (1, 14)
  const auto t1b = fe_mul(fe_norm(t4), y3b);
  const auto t2b = fe_mul(t0b, fe_norm(y3b));
  PtW r;
  r.Y = fe_wred(fe_add(fe_mul(x3, z3), t2b));
  r.X = fe_wred(fe_sub(fe_mul(fe_norm(t3), x3), t1b));
  r.Z = fe_wred(fe_add(fe_mul(fe_norm(t4), z3), fe_mul(fe_norm(t3), t0b)));
  return r;
}

// 2 P, complete (RCB 2016 algorithm 6, a = -3): 3 squarings + 8 products + 2 by b, 5 weak reductions.
VRF_HD PtW sw_dbl(const PtW& p) {
  const FeN b = CurveP256::b();
  const FeN t0 = fe_sqr(p.X), t1 = fe_sqr(p.Y), t2 = fe_sqr(p.Z);
  const auto xy2 = fe_dbl(fe_mul(p.X, p.Y));                                               // 2 X Y           (2, 4)
  const auto xz2 = fe_dbl(fe_mul(p.X, p.Z));                                               // 2 X Z
  const FeN ya = fe_wred(fe_sub(fe_mul(t2, b), xz2));                                      // b Z^2 - 2 X Z
  const auto yb = fe_add(fe_dbl(ya), ya);                                                  // 3 (...)         (3, 6)
  const auto xa = fe_norm(fe_sub(t1, fe_norm(yb)));                                        // Y^2 - 3(...)    (1, 10)
  const auto yc = fe_norm(fe_add(t1, yb));                                                 // Y^2 + 3(...)    (1, 8)
  const auto yd = fe_mul(xa, yc);
  const auto xb = fe_mul(xa, xy2);
  const auto t2x3 = fe_norm(fe_add(fe_dbl(t2), t2));                                       // 3 Z^2           (1, 6)
  const FeN zc = fe_wred(fe_sub(fe_sub(fe_mul(xz2, b), t2x3), t0));                        // 2b X Z - 3 Z^2 - X^2
  const auto zd = fe_add(fe_dbl(zc), zc);                                                  // 3 (...)         (3, 6)
  const auto t0b = fe_norm(fe_sub(fe_add(fe_dbl(t0), t0), t2x3));                          // 3 X^2 - 3 Z^2   (1, 14)
  const auto t0c = fe_mul(t0b, fe_norm(zd));
  const auto yz2 = fe_dbl(fe_mul(p.Y, p.Z));                                               // 2 Y Z           (2, 4)
  PtW r;
  r.Y = fe_wred(fe_add(yd, t0c));
  r.X = fe_wred(fe_sub(xb, fe_mul(yz2, zd)));
  r.Z = fe_wred(fe_dbl(fe_dbl(fe_mul(yz2, t1))));                                          // 8 Y^3 Z
  return r;
}

// ---- runs of doublings in Jacobian coordinates ----
// The complete doubling above costs 13 products; dbl-2001-b on Jacobian coordinates (x = X/Z^2, y = Y/Z^3; Bernstein-Lange
// EFD, a = -3) costs 3 + 5 squarings and has no exceptional case on a curve without 2-torsion: the point at infinity is
// (0 : Y != 0 : 0) and stays of that shape (Y -> -8 Y^4).  The ladders therefore double four times in Jacobian form between
// the (complete, homogeneous) additions of a window; the two changes of coordinates cost 2 products + 1 squaring each.
struct PtJ {
  FeN X, Y, Z;
};
// exact-limbed values below 2p (what fe_wred returns): zero mod p is the integer 0 or p
VRF_HD bool fe_is_zero_exact(const FeN& a) {
  uint32_t z = 0, q = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) { z |= a.v[i]; q |= a.v[i] ^ vrfk::Q29[i]; }
  return z == 0 || q == 0;
}
VRF_HD PtJ sw_to_jac(const PtW& p) {                 // (X : Y : Z) -> (X Z, Y Z^2, Z); infinity (0 : Y : 0) -> (0, 1, 0)
  const FeN zz = fe_sqr(p.Z);
  PtJ r;
  r.X = fe_mul(p.X, p.Z);
  r.Y = fe_select(fe_is_zero_exact(p.Z), fe_one(), fe_mul(p.Y, zz));
  r.Z = p.Z;
  return r;
}
VRF_HD PtW sw_from_jac(const PtJ& p) {               // (X, Y, Z) -> (X Z : Y : Z^3)
  const FeN zz = fe_sqr(p.Z);
  PtW r;
  r.X = fe_mul(p.X, p.Z);
  r.Y = p.Y;
  r.Z = fe_mul(zz, p.Z);
  return r;
}
VRF_HD PtJ sw_dbl_jac(const PtJ& p) {
  const FeN delta = fe_sqr(p.Z), gamma = fe_sqr(p.Y);
  const FeN b4 = fe_mul(fe_dbl(p.X), fe_dbl(gamma));                                      // 4 X Y^2
  const FeN t = fe_mul(fe_sub(p.X, delta), fe_add(p.X, delta));                           // X^2 - Z^4
  const auto alpha = fe_norm(fe_add(fe_dbl(t), t));                                       // 3 (...)        (1, 6)
  PtJ r;
  const auto x3 = fe_sub(fe_sqr(alpha), fe_dbl(b4));                                      // alpha^2 - 8 X Y^2
  r.X = fe_wred(x3);
  r.Z = fe_wred(fe_sub(fe_sqr(fe_add(p.Y, p.Z)), fe_add(gamma, delta)));                  // (Y + Z)^2 - Y^2 - Z^2 = 2 Y Z
  const auto g8 = fe_dbl(fe_sqr(fe_dbl(gamma)));                                          // 8 Y^4          (2, 4)
  r.Y = fe_wred(fe_sub(fe_mul(alpha, fe_sub(b4, r.X)), g8));                              // alpha (4 X Y^2 - X3) - 8 Y^4
  return r;
}
// 16 P on a homogeneous point: the ladders' step between windows
VRF_HD PtW sw_dbl4(const PtW& p) {
  PtJ j = sw_to_jac(p);
#pragma unroll 1
  for (int k = 0; k < 4; ++k) j = sw_dbl_jac(j);
  return sw_from_jac(j);
}

// y^2 = x^3 - 3x + b
VRF_HD FeN sw_rhs(const FeN& x) {
  const FeN x2 = fe_sqr(x);
  const FeN x3 = fe_mul(x2, x);
  const auto x3x = fe_add(fe_dbl(x), x);                                                   // 3x (3, 6)
  return fe_wred(fe_add(fe_sub(x3, fe_norm(x3x)), CurveP256::b()));
}
VRF_HD bool sw_on_curve(const FeN& x, const FeN& y) { return fe_eq(fe_sqr(y), sw_rhs(x)); }

// ---- signed radix-16 window tables: j * P for j = 1..8, projective ----
// An entry is 27 words padded to 28 (112 B, 16-byte aligned) and a table is CONTIGUOUS per item: a lookup is indexed by
// a digit of the item's own scalar, so the 64 lanes of a wave read 64 different entries anyway, and what matters is that
// each lane's 27 words share two cache lines.  (The first layout spread a table word-major over the batch like the other
// workspace regions: every word of every lookup then pulled its own 64-byte line, 116 GB of HBM traffic per 2^20
// verifications where the tables hold 2.7 -- profiles/r03.)
constexpr int SW_WIN = 8;
constexpr int SW_ENTRY_WORDS = 28;
constexpr int SW_TABLE_WORDS = SW_WIN * SW_ENTRY_WORDS;      // 224

VRF_HD void sw_build_table(uint32_t* tab, size_t stride, const PtW& p) {
  PtW acc = p;
  ptw_store(tab, stride, acc);
#pragma unroll 1
  for (int j = 1; j < SW_WIN; ++j) {
    acc = sw_add(acc, p);
    ptw_store(tab + (size_t)j * SW_ENTRY_WORDS * stride, stride, acc);
  }
}
VRF_HD PtW sw_lookup(const uint32_t* tab, size_t stride, int digit) {      // digit in [-8, 8]
  const int mag = digit < 0 ? -digit : digit;
  const int idx = mag > 0 ? mag - 1 : 0;
  PtW e = ptw_load(tab + (size_t)idx * SW_ENTRY_WORDS * stride, stride);
  e = sw_select(mag == 0, sw_identity(), e);
  return sw_cneg(digit < 0, e);
}

VRF_NS_END
