// bls12_quad.cuh -- the BLS12-381 pairing check of bls12.cuh with ONE ITEM PER DPP QUAD (SURVEY.md section 8
// row a11; replaces ark_ec::pairing::Pairing::{multi_miller_loop, final_exponentiation}, reached from
// /root/reference through `ring`, src/lib.rs:14).
//
// Why: one lane per item keeps every Fp12 temporary in the per-lane scratch segment (10.5 KB per lane,
// 512 registers => one wave per SIMD); measured at 2^16 items the kernel moves ~190 GB of scratch traffic
// for 1.45e10 VALU wave-instructions: it is HBM-bound at 45 % of the VALU ceiling, and the 2^14 batch of
// BASELINE.json configs[4] fills one wave per CU.  Here lane j < 3 of a quad holds the v^j column of
// every Fp6: an Fp12 is 2 Fp2 = 56 registers per lane, the tower lives in registers, and a quad offers
// four times the lanes to a small batch.
//
//   Fp6 product (x0,x1,x2)(y0,y1,y2), Karatsuba: lane j forms p_j = x_j y_j (local) and the cross product
//   of the OTHER two columns q_j = (x_a + x_b)(y_a + y_b), c_j = q_j - p_a - p_b.  Then
//       r0 = p0 + xi c0,   r1 = c2 + xi p2,   r2 = c1 + p1:
//   lane 0 is local, lanes 1 and 2 swap (p, c).  Two Fp2 products per lane instead of six; operands and
//   results travel by DPP quad_perm moves (VALU, no LDS).
//   Fp12 = Fp6[w]/(w^2 - v): both halves of a column sit in the same lane, so the Karatsuba sums of
//   fp12 mul / sqr are local; *v is a rotation of the columns with xi applied in lane 0.
//   Granger-Scott cyclotomic squaring: the three Fp4 squarings run one per lane.
//   Frobenius is column-local.  The line functions of the Miller loop are computed two lanes per (P, Q)
//   pair (g2_step_q: five rounds of one Fp2 product per lane for a doubling, seven for an addition) and
//   broadcast to the quad.  Code is kept small (the I-cache holds ~64 KB and one Fp2 product is ~14 KB):
//   every level loops over ONE instance of the level below, operands picked by selects.
#pragma once
#include "bls12.cuh"
#include <type_traits>

namespace bls {

// quad_perm controls: lane i reads lane perm[i]; lane 3 always reads itself
constexpr int QP_ROT1 = 1 | (2 << 2) | (0 << 4) | (3 << 6);     // j <- (j+1) % 3
constexpr int QP_ROT2 = 2 | (0 << 2) | (1 << 4) | (3 << 6);     // j <- (j+2) % 3
constexpr int QP_SWAP12 = 0 | (2 << 2) | (1 << 4) | (3 << 6);   // 1 <-> 2
constexpr int QP_SWAP01 = 1 | (0 << 2) | (2 << 4) | (3 << 6);   // 0 <-> 1
constexpr int QP_BC0 = 0x00, QP_BC1 = 0x55, QP_BC2 = 0xaa;      // broadcast lane k

template <int CTRL>
__device__ __forceinline__ int32_t qperm_i32(int32_t v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}
template <int CTRL, int L, int V>
__device__ __forceinline__ Fp<L, V> qperm(const Fp<L, V>& a) {
  Fp<L, V> r;
#pragma unroll
  for (int i = 0; i < NLB; ++i) r.v[i] = qperm_i32<CTRL>(a.v[i]);
  return r;
}
template <int CTRL, int L, int V>
__device__ __forceinline__ Fp2T<L, V> qperm(const Fp2T<L, V>& a) {
  Fp2T<L, V> r;
  r.a = qperm<CTRL>(a.a);
  r.b = qperm<CTRL>(a.b);
  return r;
}
template <int L, int V>
__device__ __forceinline__ Fp2T<L, V> fp2_sel(bool c, const Fp2T<L, V>& x, const Fp2T<L, V>& y) {   // c ? x : y
  Fp2T<L, V> r;
  r.a = fp_select(c, x.a, y.a);
  r.b = fp_select(c, x.b, y.b);
  return r;
}
template <int LO, int VO, int L, int V>
__device__ __forceinline__ Fp2T<LO, VO> fp2_widen(const Fp2T<L, V>& x) {
  Fp2T<LO, VO> r;
  r.a = Fp<LO, VO>(x.a);
  r.b = Fp<LO, VO>(x.b);
  return r;
}

struct Q12 { Fp2 c0, c1; };          // this lane's column of c0 and of c1

// x * v for a column-distributed Fp6: (x0, x1, x2) -> (xi x2, x0, x1)
__device__ __forceinline__ Fp2 fp6_mul_v_q(const Fp2& x, int q) {
  Fp2 u = qperm<QP_ROT2>(x);
  Fp2 xu = fp2_fit(fp2_mul_xi(u));
  return fp2_sel(q == 0, xu, u);
}

// column-distributed Fp6 product.  Intermediates keep their natural bounds (normalised limbs, values up to
// 54 p); only the result is brought back to storage form: one quotient-estimate reduction per component.
__device__ __forceinline__ Fp2 fp6_mul_q(const Fp2& x, const Fp2& y, int q) {
  using W = Fp2T<2, 2 * STORE_V>;
  const W xo = fp2_add(qperm<QP_ROT1>(x), qperm<QP_ROT2>(x));     // the other two columns
  const W yo = fp2_add(qperm<QP_ROT1>(y), qperm<QP_ROT2>(y));
  using M = decltype(fp2_norm(fp2_mul(xo, yo)));
  M p, cr;
  p.a = fp_zero(); p.b = fp_zero(); cr = p;
#pragma unroll 1
  for (int r = 0; r < 2; ++r) {
    const W a = fp2_sel(r != 0, xo, fp2_widen<2, 2 * STORE_V>(x));
    const W b = fp2_sel(r != 0, yo, fp2_widen<2, 2 * STORE_V>(y));
    const M m = fp2_norm(fp2_mul(a, b));
    p = fp2_sel(r == 0, m, p);
    cr = fp2_sel(r != 0, m, cr);
  }
  const auto c = fp2_norm(fp2_sub(fp2_sub(cr, qperm<QP_ROT1>(p)), qperm<QP_ROT2>(p)));
  using C = std::remove_const_t<decltype(c)>;
  C pw;
  pw.a = p.a; pw.b = p.b;                                          // widening
  const C ps = qperm<QP_SWAP12>(pw), cs = qperm<QP_SWAP12>(c);
  // lane 0: ps + xi cs ; lane 1: cs + xi ps ; lane 2: cs + ps
  const C A = fp2_sel(q == 0, ps, cs), B = fp2_sel(q == 0, cs, ps);
  const auto xiB = fp2_mul_xi(B);
  using X = std::remove_const_t<decltype(xiB)>;
  X Bw;
  Bw.a = B.a; Bw.b = B.b;                                          // widening
  return fp2_fit(fp2_add(A, fp2_sel(q == 2, Bw, xiB)));
}

__device__ __forceinline__ Q12 fp12_mul_q(const Q12& x, const Q12& y, int q) {
  const Fp2 sx = fp2_fit(fp2_add(x.c0, x.c1)), sy = fp2_fit(fp2_add(y.c0, y.c1));
  Fp2 t0 = fp2_zero(), t1 = fp2_zero(), s = fp2_zero();
#pragma unroll 1
  for (int r = 0; r < 3; ++r) {
    const Fp2 a = fp2_sel(r == 0, x.c0, fp2_sel(r == 1, x.c1, sx));
    const Fp2 b = fp2_sel(r == 0, y.c0, fp2_sel(r == 1, y.c1, sy));
    const Fp2 m = fp6_mul_q(a, b, q);
    t0 = fp2_sel(r == 0, m, t0);
    t1 = fp2_sel(r == 1, m, t1);
    s = fp2_sel(r == 2, m, s);
  }
  Q12 o;
  o.c1 = fp2_fit(fp2_sub(fp2_sub(s, t0), t1));
  o.c0 = fp2_fit(fp2_add(t0, fp6_mul_v_q(t1, q)));
  return o;
}

// complex squaring: c0' = (c0 + c1)(c0 + v c1) - ab - v ab, c1' = 2ab, ab = c0 c1
__device__ __forceinline__ Q12 fp12_sqr_q(const Q12& x, int q) {
  const Fp2 s0 = fp2_fit(fp2_add(x.c0, x.c1));
  const Fp2 s1 = fp2_fit(fp2_add(x.c0, fp6_mul_v_q(x.c1, q)));
  Fp2 ab = fp2_zero(), m2 = fp2_zero();
#pragma unroll 1
  for (int r = 0; r < 2; ++r) {
    const Fp2 a = fp2_sel(r == 0, x.c0, s0), b = fp2_sel(r == 0, x.c1, s1);
    const Fp2 m = fp6_mul_q(a, b, q);
    ab = fp2_sel(r == 0, m, ab);
    m2 = fp2_sel(r != 0, m, m2);
  }
  Q12 o;
  o.c0 = fp2_fit(fp2_sub(fp2_sub(m2, ab), fp6_mul_v_q(ab, q)));
  o.c1 = fp2_fit(fp2_dbl(ab));
  return o;
}

// f * (c0 + c1 v + c4 v w); the line coefficients are replicated in the quad.  Five products per lane:
// two sparse Fp6 products (c0, c1, 0) and f.c1 * (c4 v) = v * (f.c1 scaled by c4), column-local.
__device__ __forceinline__ Q12 fp12_mul_by_014_q(const Q12& f, const Fp2& c0, const Fp2& c1, const Fp2& c4, int q) {
  const Fp2 z = fp2_zero();
  const Fp2 o = fp2_fit(fp2_add(c1, c4));
  const Fp2 y01 = fp2_sel(q == 0, c0, fp2_sel(q == 1, c1, z));   // (c0, c1, 0)
  const Fp2 y0o = fp2_sel(q == 0, c0, fp2_sel(q == 1, o, z));    // (c0, c1 + c4, 0)
  const Fp2 fs = fp2_fit(fp2_add(f.c0, f.c1));
  Fp2 aa = z, s = z;
#pragma unroll 1
  for (int r = 0; r < 2; ++r) {
    const Fp2 a = fp2_sel(r == 0, f.c0, fs);
    const Fp2 b = fp2_sel(r == 0, y01, y0o);
    const Fp2 m = fp6_mul_q(a, b, q);
    aa = fp2_sel(r == 0, m, aa);
    s = fp2_sel(r != 0, m, s);
  }
  const Fp2 bb = fp6_mul_v_q(fp2_fit(fp2_mul(f.c1, c4)), q);
  Q12 r;
  r.c1 = fp2_fit(fp2_sub(fp2_sub(s, aa), bb));
  r.c0 = fp2_fit(fp2_add(fp6_mul_v_q(bb, q), aa));
  return r;
}

// Granger-Scott squaring in the cyclotomic subgroup: lane 0 squares the Fp4 (c0.c0, c1.c1), lane 1
// (c0.c1, c1.c2), lane 2 (c1.c0, c0.c2)
__device__ __forceinline__ Q12 fp12_cyclotomic_sqr_q(const Q12& x, int q) {
  const Fp2 g = qperm<QP_ROT1>(x.c1);
  const Fp2 a = fp2_sel(q == 2, g, x.c0), b = fp2_sel(q == 2, x.c0, g);
  using W = Fp2T<3, 3 * STORE_V>;
  const W s1 = fp2_widen<3, 3 * STORE_V>(fp2_add(a, b));
  const W s2 = fp2_add(fp2_mul_xi(b), a);
  Fp2 tmp = fp2_zero(), s = fp2_zero();
#pragma unroll 1
  for (int r = 0; r < 2; ++r) {
    const W u = fp2_sel(r == 0, fp2_widen<3, 3 * STORE_V>(a), s1);
    const W v = fp2_sel(r == 0, fp2_widen<3, 3 * STORE_V>(b), s2);
    const Fp2 m = fp2_fit(fp2_mul(u, v));
    tmp = fp2_sel(r == 0, m, tmp);
    s = fp2_sel(r != 0, m, s);
  }
  const Fp2 T0 = fp2_fit(fp2_sub(fp2_sub(s, tmp), fp2_mul_xi(tmp)));
  const Fp2 T1 = fp2_fit(fp2_dbl(tmp));
  const Fp2 u0 = qperm<QP_SWAP12>(T0);
  const Fp2 u1r = qperm<QP_SWAP01>(T1);
  const Fp2 u1 = fp2_sel(q == 0, fp2_fit(fp2_mul_xi(u1r)), u1r);
  auto three = [](const Fp2& t) { return fp2_add(fp2_dbl(t), t); };
  Q12 o;
  o.c0 = fp2_fit(fp2_sub(three(u0), fp2_dbl(x.c0)));
  o.c1 = fp2_fit(fp2_add(three(u1), fp2_dbl(x.c1)));
  return o;
}

__device__ __forceinline__ Q12 fp12_conj_q(const Q12& x) {
  Q12 o;
  o.c0 = x.c0;
  o.c1 = fp2_neg(x.c1);
  return o;
}

// f^p: lane j holds the coefficients of w^(2j) and w^(2j+1)
__device__ __forceinline__ Q12 fp12_frob_q(const Q12& x, int q) {
  const Fp2 g0 = fp2_sel(q == 0, fp2_one(), fp2_sel(q == 1, gamma_const(2), gamma_const(4)));
  const Fp2 g1 = fp2_sel(q == 0, gamma_const(1), fp2_sel(q == 1, gamma_const(3), gamma_const(5)));
  Fp2 o0 = fp2_zero(), o1 = fp2_zero();
#pragma unroll 1
  for (int r = 0; r < 2; ++r) {
    const Fp2 a = fp2_conj(fp2_sel(r == 0, x.c0, x.c1)), b = fp2_sel(r == 0, g0, g1);
    const Fp2 m = fp2_fit(fp2_mul(a, b));
    o0 = fp2_sel(r == 0, m, o0);
    o1 = fp2_sel(r != 0, m, o1);
  }
  Q12 o;
  o.c0 = o0; o.c1 = o1;
  return o;
}

// column form <-> a whole Fp12 replicated in every lane of the quad
__device__ __forceinline__ void q12_gather(Fp12* full, const Q12& x) {
  full->c0.c0 = qperm<QP_BC0>(x.c0); full->c0.c1 = qperm<QP_BC1>(x.c0); full->c0.c2 = qperm<QP_BC2>(x.c0);
  full->c1.c0 = qperm<QP_BC0>(x.c1); full->c1.c1 = qperm<QP_BC1>(x.c1); full->c1.c2 = qperm<QP_BC2>(x.c1);
}
__device__ __forceinline__ Q12 q12_scatter(const Fp12* full, int q) {
  Q12 o;
  o.c0 = fp2_sel(q == 0, full->c0.c0, fp2_sel(q == 1, full->c0.c1, full->c0.c2));
  o.c1 = fp2_sel(q == 0, full->c1.c0, fp2_sel(q == 1, full->c1.c1, full->c1.c2));
  return o;
}
__device__ __forceinline__ Q12 q12_one(int q) {
  Q12 o;
  o.c0 = fp2_sel(q == 0, fp2_one(), fp2_zero());
  o.c1 = fp2_zero();
  return o;
}

// f^x, x = -X_ABS, f in the cyclotomic subgroup
__device__ __attribute__((noinline)) Q12 exp_by_x_q(const Q12& f, int q) {
  Q12 acc = f;
#pragma unroll 1
  for (int bit = 62; bit >= 0; --bit) {
    acc = fp12_cyclotomic_sqr_q(acc, q);
    if ((X_ABS >> bit) & 1) acc = fp12_mul_q(acc, f, q);
  }
  return fp12_conj_q(acc);
}

// f^(3 (p^12 - 1)/r), same chain as final_exponentiation of bls12.cuh
__device__ __forceinline__ Q12 final_exponentiation_q(const Q12& f, int q) {
  Q12 f2;
  {
    // easy part: the inversion runs on the gathered value, replicated in every lane
    Fp12 full, inv;
    q12_gather(&full, f);
    fp12_inv(&inv, &full);
    Q12 t = fp12_mul_q(fp12_conj_q(f), q12_scatter(&inv, q), q);          // f^(p^6 - 1)
    f2 = fp12_mul_q(fp12_frob_q(fp12_frob_q(t, q), q), t, q);              // ^(p^2 + 1)
  }
  Q12 y = fp12_mul_q(exp_by_x_q(f2, q), fp12_conj_q(f2), q);              // ^(x - 1)
  y = fp12_mul_q(exp_by_x_q(y, q), fp12_conj_q(y), q);                    // ^(x - 1)^2
  y = fp12_mul_q(exp_by_x_q(y, q), fp12_frob_q(y, q), q);                 // ^(x + p)        = y2
  Q12 t = exp_by_x_q(exp_by_x_q(y, q), q);                                // y2^(x^2)
  t = fp12_mul_q(t, fp12_frob_q(fp12_frob_q(y, q), q), q);                // * y2^(p^2)
  y = fp12_mul_q(t, fp12_conj_q(y), q);                                   // ^(x^2 + p^2 - 1) = y3
  t = fp12_mul_q(fp12_cyclotomic_sqr_q(f2, q), f2, q);                    // f2^3
  return fp12_mul_q(y, t, q);
}

// ---- Miller loop steps, two lanes per (P, Q) pair: lanes 0,1 of the quad own pair 0, lanes 2,3 pair 1; the
// G2 state T = (X, Y, Z) is replicated in both lanes, h = lane & 1 says which product of a round a lane forms.
// Each round both lanes of a pair form one Fp2 product (operands chosen by h) and swap the results inside the
// pair: five rounds for a doubling, seven for an addition.  Homogeneous projective formulas of bls12.cuh (g2_double_step / g2_add_step) with
// the halvings scaled away: T3 is multiplied by 4 (projective point: harmless) and the lines keep their
// scale up to factors of Fp2, which the final exponentiation kills.
constexpr int QP_PAIRSWAP = 1 | (0 << 2) | (3 << 4) | (2 << 6);
struct G2Line { Fp2 c0, c1, c4; };

__device__ __forceinline__ Fp2 fp2_small12(const Fp2& x) {    // 12 x, storage form
  const Fp2 t = fp2_fit(fp2_add(fp2_dbl(x), x));              // 3x
  return fp2_fit(fp2_dbl(fp2_dbl(t)));
}

// squaring round: both halves square their operand
template <int L1, int V1>
__device__ __forceinline__ void pair_round_sqr(Fp2& pa, Fp2& pb, const Fp2T<L1, V1>& u, bool hb) {
  const Fp2 m = fp2_fit(fp2_sqr(u));
  const Fp2 mo = qperm<QP_PAIRSWAP>(m);
  pa = fp2_sel(hb, mo, m);
  pb = fp2_sel(hb, m, mo);
}
// one round: this lane multiplies (u, v); returns the product of half 0 in pa and of half 1 in pb
template <int L1, int V1, int L2, int V2>
__device__ __forceinline__ void pair_round(Fp2& pa, Fp2& pb, const Fp2T<L1, V1>& u, const Fp2T<L2, V2>& v, bool hb) {
  const Fp2 m = fp2_fit(fp2_mul(u, v));
  const Fp2 mo = qperm<QP_PAIRSWAP>(m);
  pa = fp2_sel(hb, mo, m);
  pb = fp2_sel(hb, m, mo);
}

// T <- 2T and the tangent line at T, scaled by the G1 coordinates of this lane's pair
__device__ __forceinline__ G2Line g2_double_q(G2Proj& T, const G1Aff& P, int h) {
  const bool hb = h != 0;
  const Fp2 X = T.X, Y = T.Y, Z = T.Z;
  Fp2 pyc;                                                 // (Py, 0) as an Fp2 factor
  pyc.a = P.y; pyc.b = fp_zero();
  Fp2 xy, b, c, syz, j, e2, t1, t2, t3, c4s;
  pair_round(xy, b, fp2_sel(hb, Y, X), Y, hb);                                   // XY | Y^2
  {
    const auto yz = fp2_add(Y, Z);
    const auto u = fp2_sel(hb, yz, fp2_widen<2, 2 * STORE_V>(Z));
    pair_round_sqr(c, syz, u, hb);                                               // Z^2 | (Y+Z)^2
  }
  const Fp2 e = fp2_fit(fp2_mul_xi(fp2_small12(c)));                             // b' * 3c = 12 xi c
  const Fp2 hh = fp2_fit(fp2_sub(syz, fp2_add(b, c)));                           // (Y+Z)^2 - b - c
  {
    const Fp2 u = fp2_sel(hb, e, X);
    pair_round_sqr(j, e2, u, hb);                                                // X^2 | e^2
  }
  {
    const Fp2 f3 = fp2_fit(fp2_add(fp2_dbl(e), e));
    const Fp2 bmf = fp2_fit(fp2_sub(b, f3)), bpf = fp2_fit(fp2_add(b, f3));
    pair_round(t1, t2, fp2_sel(hb, bpf, xy), fp2_sel(hb, bpf, bmf), hb);         // XY (b - f) | (b + f)^2
  }
  pair_round(t3, c4s, fp2_sel(hb, hh, b), fp2_sel(hb, pyc, hh), hb);             // b h | h Py
  T.X = fp2_fit(fp2_dbl(t1));                                   // 2 XY (b - f)            (x4 overall)
  T.Y = fp2_fit(fp2_sub(t2, fp2_small12(e2)));                  // (b + f)^2 - 12 e^2
  T.Z = fp2_fit(fp2_dbl(fp2_dbl(t3)));                          // 4 b h
  G2Line L;
  L.c0 = fp2_fit(fp2_sub(b, e));
  L.c1 = fp2_fit(fp2_mul_fp(fp2_neg(fp2_add(fp2_dbl(j), j)), P.x));              // -3 X^2 Px
  L.c4 = c4s;
  return L;
}

// T <- T + Q and the line through T and Q, scaled by the G1 coordinates
__device__ __forceinline__ G2Line g2_add_q(G2Proj& T, const G2Aff& Q, const G1Aff& P, int h) {
  const bool hb = h != 0;
  const Fp2 X = T.X, Y = T.Y, Z = T.Z;
  Fp2 pyc;
  pyc.a = P.y; pyc.b = fp_zero();
  Fp2 yz, xz, c, d, e, f, g, tq, lq, z3, x3, y3a, ey, c4s;
  pair_round(yz, xz, fp2_sel(hb, Q.x, Q.y), Z, hb);                              // Qy Z | Qx Z
  const Fp2 th = fp2_fit(fp2_sub(Y, yz)), lam = fp2_fit(fp2_sub(X, xz));
  {
    const Fp2 u = fp2_sel(hb, lam, th);
    pair_round_sqr(c, d, u, hb);                                                 // theta^2 | lambda^2
  }
  pair_round(e, f, fp2_sel(hb, Z, lam), fp2_sel(hb, c, d), hb);                  // lambda d | Z c
  pair_round(g, tq, fp2_sel(hb, th, X), fp2_sel(hb, Q.x, d), hb);                // X d | theta Qx
  pair_round(lq, z3, fp2_sel(hb, Z, lam), fp2_sel(hb, e, Q.y), hb);              // lambda Qy | Z e
  {
    const Fp2 hv = fp2_fit(fp2_sub(fp2_add(e, f), fp2_dbl(g)));
    const Fp2 gmh = fp2_fit(fp2_sub(g, hv));
    pair_round(x3, y3a, fp2_sel(hb, th, lam), fp2_sel(hb, gmh, hv), hb);         // lambda h | theta (g - h)
  }
  pair_round(ey, c4s, fp2_sel(hb, lam, e), fp2_sel(hb, pyc, Y), hb);             // e Y | lambda Py
  T.X = x3;
  T.Y = fp2_fit(fp2_sub(y3a, ey));
  T.Z = z3;
  G2Line L;
  L.c0 = fp2_fit(fp2_sub(tq, lq));                                               // theta Qx - lambda Qy
  L.c1 = fp2_fit(fp2_mul_fp(fp2_neg(th), P.x));
  L.c4 = c4s;
  return L;
}

// One item per quad.  g1: 2 x 24 words, g2: 2 x 48 words (formats of pairing_check2_item).  Returns the
// status in every lane.
__device__ __attribute__((noinline)) uint32_t pairing_check2_quad(const uint32_t* g1, const uint32_t* g2, int q) {
  const int pi = q >> 1, h = q & 1;
  G1Aff P;
  G2Aff Q;
  bool i1, i2;
  bool ok = g1_load(P, i1, g1 + 24 * pi);
  ok = g2_load(Q, i2, g2 + 48 * pi) && ok;
  const int my_skip = (i1 || i2) ? 1 : 0, my_ok = ok ? 1 : 0;
  const int skip0 = qperm_i32<QP_BC0>(my_skip), skip1 = qperm_i32<QP_BC2>(my_skip);
  const bool all_ok = qperm_i32<QP_BC0>(my_ok) != 0 && qperm_i32<QP_BC2>(my_ok) != 0;
  G2Proj T;
  T.X = Q.x; T.Y = Q.y; T.Z = fp2_one();
  Q12 f = q12_one(q);
#pragma unroll 1
  for (int bit = 62; bit >= 0; --bit) {
    f = fp12_sqr_q(f, q);
    const int nsteps = ((X_ABS >> bit) & 1) ? 2 : 1;
#pragma unroll 1
    for (int step = 0; step < nsteps; ++step) {
      G2Line L;
      if (step == 0) L = g2_double_q(T, P, h);
      else L = g2_add_q(T, Q, P, h);
#pragma unroll 1
      for (int i = 0; i < 2; ++i) {
        const Fp2 l0 = fp2_sel(i == 0, qperm<QP_BC0>(L.c0), qperm<QP_BC2>(L.c0));
        const Fp2 l1 = fp2_sel(i == 0, qperm<QP_BC0>(L.c1), qperm<QP_BC2>(L.c1));
        const Fp2 l4 = fp2_sel(i == 0, qperm<QP_BC0>(L.c4), qperm<QP_BC2>(L.c4));
        const bool skip = (i == 0 ? skip0 : skip1) != 0;
        if (!skip) f = fp12_mul_by_014_q(f, l0, l1, l4, q);
      }
    }
  }
  f = fp12_conj_q(f);
  const Q12 e = final_exponentiation_q(f, q);
  Fp12 full;
  q12_gather(&full, e);
  const bool one = fp12_is_one(&full);
  if (!all_ok) return PST_INVALID;
  return one ? PST_OK : PST_FAIL;
}

// One item per quad against prepared G2 lines (pairing_prepare_g2_pair).  g1: 2 x 24 words.  An iteration costs
// the squaring, the two sparse products and one Fp2-by-Fp product per lane (lane 2i scales c1 of pair i by
// x_P, lane 2i+1 scales c4 by y_P) instead of also the five to seven rounds of a G2 step.
__device__ __attribute__((noinline)) uint32_t pairing_check2_quad_prepared(const uint32_t* g1, const uint32_t* prep, int q) {
  const int pi = q >> 1;
  G1Aff P;
  bool i1;
  const bool ok1 = g1_load(P, i1, g1 + 24 * pi);
  const uint32_t* flags = prep + (size_t)2 * G2_LINES * G2_LINE_WORDS;
  const bool ok = ok1 && flags[2 * pi] != 0;
  const int my_skip = (i1 || flags[2 * pi + 1] != 0) ? 1 : 0, my_ok = ok ? 1 : 0;
  const int skip0 = qperm_i32<QP_BC0>(my_skip), skip1 = qperm_i32<QP_BC2>(my_skip);
  const bool all_ok = qperm_i32<QP_BC0>(my_ok) != 0 && qperm_i32<QP_BC2>(my_ok) != 0;
  const FpS scale = (q & 1) ? P.y : P.x;
  const uint32_t* my_line = prep + (size_t)pi * G2_LINES * G2_LINE_WORDS + ((q & 1) ? 4 * NLB : 2 * NLB);
  const uint32_t* line0 = prep;
  Q12 f = q12_one(q);
#pragma unroll 1
  for (int bit = 62; bit >= 0; --bit) {
    f = fp12_sqr_q(f, q);
    const int nsteps = ((X_ABS >> bit) & 1) ? 2 : 1;
#pragma unroll 1
    for (int step = 0; step < nsteps; ++step) {
      const Fp2 scaled = fp2_fit(fp2_mul_fp(fp2_load_words(my_line), scale));
#pragma unroll 1
      for (int i = 0; i < 2; ++i) {
        const Fp2 l0 = fp2_load_words(line0 + (size_t)i * G2_LINES * G2_LINE_WORDS);
        const Fp2 l1 = fp2_sel(i == 0, qperm<QP_BC0>(scaled), qperm<QP_BC2>(scaled));
        const Fp2 l4 = fp2_sel(i == 0, qperm<QP_BC1>(scaled), qperm<0xff>(scaled));
        const bool skip = (i == 0 ? skip0 : skip1) != 0;
        if (!skip) f = fp12_mul_by_014_q(f, l0, l1, l4, q);
      }
      my_line += G2_LINE_WORDS;
      line0 += G2_LINE_WORDS;
    }
  }
  f = fp12_conj_q(f);
  const Q12 e = final_exponentiation_q(f, q);
  Fp12 full;
  q12_gather(&full, e);
  const bool one = fp12_is_one(&full);
  if (!all_ok) return PST_INVALID;
  return one ? PST_OK : PST_FAIL;
}

}  // namespace bls
