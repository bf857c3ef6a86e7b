// te_sw_map.cuh -- [ref src/lib.rs:14 `utils`: te_sw_map::{te_to_sw, sw_to_te}] the point map between a twisted-Edwards
// curve a x^2 + y^2 = 1 + d x^2 y^2 and the short-Weierstrass form of its Montgomery model
//   B v^2 = u^3 + A u^2 + u,  A = 2(a + d)/(a - d),  B = 4/(a - d)      (arkworks' MontCurveConfig of the same curve)
//   TE -> Mont (u, v) = ((1 + y)/(1 - y), (1 + y)/((1 - y) x)),   Mont -> SW ((u + A/3)/B, v/B).
// Upstream divides four times per point; here A, B, 1/3 never appear -- substituting them and clearing denominators:
//   TE -> SW:  D = 12 x (1 - y),   X = (3 (1 + y)(a - d) + 2 (a + d)(1 - y)) x / D,   Y = 3 (1 + y)(a - d) / D
//   SW -> TE:  x = (6 X - (a + d)) / (6 Y),   y = (12 X - 5 a + d) / (12 X + a - 5 d)
// one inversion each.  false = upstream's None (`inverse()?` of zero): x = 0 or y = 1 going out (the identity and the point
// of order 2 have no affine image), Y = 0 or B X - A/3 = -1 coming back.  Like upstream (`new_unchecked`) neither direction
// asks whether the input is on its curve.  oracle/vrf_oracle.py te_to_sw / sw_to_te is the restatement the tests compare with.
#pragma once
#include "te.cuh"

VRF_NS_BEGIN

template <int L, int V>
VRF_HD FeN fe_full(const Fe<L, V>& a) { return fe_mul(a, fe_one()); }      // any lazy sum -> a storage-type element

template <class C>
VRF_HD FeN te_coeff_a() {
  if constexpr (C::A_PLUS_ONE) return fe_one();
  else return fe_full(fe_neg(C::aneg_m()));
}
template <int L, int V>
VRF_HD FeN fe_times3(const Fe<L, V>& a) {
  const FeN n = fe_full(a);
  return fe_full(fe_add(fe_dbl(n), n));
}

template <class C>
VRF_HD bool te_to_sw(FeN& sx, FeN& sy, const FeN& x, const FeN& y) {
  const FeN one = fe_one(), a = te_coeff_a<C>(), d = C::d();
  const FeN amd = fe_full(fe_sub(a, d)), apd2 = fe_full(fe_dbl(fe_add(a, d)));      // a - d, 2 (a + d)
  const FeN n = fe_full(fe_add(one, y)), m = fe_full(fe_sub(one, y));               // 1 + y, 1 - y
  const FeN xm3 = fe_times3(fe_mul(x, m));
  const FeN D = fe_full(fe_dbl(fe_dbl(xm3)));                                       // 12 x (1 - y)
  const bool some = !fe_is_zero(D);
  const FeN iD = fe_inv(D);
  const FeN namd3 = fe_times3(fe_mul(n, amd));                                      // 3 (1 + y)(a - d)
  const FeN num = fe_full(fe_add(namd3, fe_mul(apd2, m)));
  sx = fe_mul(fe_mul(num, x), iD);
  sy = fe_mul(namd3, iD);
  return some;
}

template <class C>
VRF_HD bool sw_to_te(FeN& tx, FeN& ty, const FeN& x, const FeN& y) {
  const FeN a = te_coeff_a<C>(), d = C::d();
  const FeN x6 = fe_full(fe_dbl(fe_times3(x))), y6 = fe_full(fe_dbl(fe_times3(y)));
  const FeN x12 = fe_full(fe_dbl(x6));
  const FeN a5 = fe_full(fe_mul5(a)), d5 = fe_full(fe_mul5(d));
  const FeN num_v = fe_full(fe_sub(x6, fe_full(fe_add(a, d))));                     // 6 X - (a + d)
  const FeN num_w = fe_full(fe_add(fe_sub(x12, a5), d));                            // 12 X - 5 a + d
  const FeN den_w = fe_full(fe_sub(fe_full(fe_add(x12, a)), d5));                   // 12 X + a - 5 d
  const FeN den = fe_mul(y6, den_w);
  const bool some = !fe_is_zero(den);
  const FeN i = fe_inv(den);
  tx = fe_mul(fe_mul(num_v, den_w), i);
  ty = fe_mul(fe_mul(num_w, y6), i);
  return some;
}

VRF_NS_END
