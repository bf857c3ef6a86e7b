// bls12_row.cuh -- the quad pairing arithmetic of bls12_quad.cuh spread over THREE quads of a 16-lane row, for the
// case where ONE pairing check decides a whole batch (vrfhip_pairing_check_batch_rlc: two G1 MSMs, then a single
// check whose latency -- 10 ms on one quad -- was 2/3 of the call).  [SURVEY.md section 8 row a11 / f3; replaces
// `Pairing::multi_miller_loop` + `final_exponentiation`, reached from /root/reference through `ring`, src/lib.rs:14]
//
// Every quad of the row holds the same column-distributed Fp12 (Q12).  The tower operations of bls12_quad.cuh are
// loops over "rounds" whose operands are picked by selects (three Fp6 products in an Fp12 product, two in a
// squaring or a sparse product, two Fp2 products per lane in a cyclotomic squaring or a Frobenius): here round r is
// run by quad r, all quads in the same instructions, and the results travel between the quads with ds_bpermute (the
// LDS crossbar, no memory).  Fp12 product 6 -> 2 rounds, squaring 4 -> 2, sparse product 5 -> 2 (the f.c1 * c4 v term
// becomes a third sparse Fp6 product), cyclotomic squaring and Frobenius 2 -> 1.  Lanes 12..15 of a row (quad 3) run
// along as a copy of quad 2.
#pragma once
#include "bls12_quad.cuh"

namespace bls {

struct RowCtx {
  int q;          // column inside the quad (0..2; 3 = the quad's idle lane)
  int g;          // quad inside the row (0..2; 3 mirrors 2)
  int k;          // TRI layout only: the row inside the item's three rows = which Fp product of an Fp2 product (0..2)
  int src[3];     // byte address (lane * 4) of the lane with the same column in quad 0, 1, 2 of this row
  int srck[3];    // TRI layout only: the lane at the same place in row 0, 1, 2 of the item
};
__device__ __forceinline__ RowCtx row_ctx(int lane_in_wave) {
  RowCtx c;
  c.q = lane_in_wave & 3;
  c.g = (lane_in_wave >> 2) & 3;
  c.k = (lane_in_wave >> 4) & 3;
  const int base = (lane_in_wave & ~15) | c.q;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    c.src[k] = (base | (4 * k)) << 2;
    c.srck[k] = ((lane_in_wave & 15) | (16 * k)) << 2;       // one item per wave: rows 0..2 of the wave
  }
  return c;
}

// TRI: an Fp2 product on the three rows of an item -- row 0 forms a0 b0, row 1 a1 b1, row 2 (a0 + a1)(b0 + b1), the three
// Fp products cross the rows with ds_bpermute and every row recombines them (Karatsuba, as fp2_mul).  The rows hold
// identical copies of everything else, so all linear work is simply repeated.  One Fp product of latency per Fp2 product
// instead of three.
template <bool TRI, int L1, int V1, int L2, int V2>
__device__ __forceinline__ auto row_fp2_mul(const Fp2T<L1, V1>& x0, const Fp2T<L2, V2>& y0, const RowCtx& c) {
  if constexpr (!TRI) {
    return fp2_mul(x0, y0);
  } else {
    const auto x = fp2_norm(x0);
    const auto y = fp2_norm(y0);
    const auto xs = fp_add(x.a, x.b);
    const auto ys = fp_add(y.a, y.b);
    using XS = std::remove_const_t<decltype(xs)>;
    using YS = std::remove_const_t<decltype(ys)>;
    const XS xo = fp_select(c.k == 0, XS(x.a), fp_select(c.k == 1, XS(x.b), xs));
    const YS yo = fp_select(c.k == 0, YS(y.a), fp_select(c.k == 1, YS(y.b), ys));
    const auto m = fp_mul(xo, yo);
    using M = std::remove_const_t<decltype(m)>;
    M ac, bd, sm;
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
      ac.v[i] = __builtin_amdgcn_ds_bpermute(c.srck[0], m.v[i]);
      bd.v[i] = __builtin_amdgcn_ds_bpermute(c.srck[1], m.v[i]);
      sm.v[i] = __builtin_amdgcn_ds_bpermute(c.srck[2], m.v[i]);
    }
    const auto ra = fp_sub(ac, bd);
    const auto rb = fp_sub(fp_sub(sm, ac), bd);
    using RB = std::remove_const_t<decltype(rb)>;
    Fp2T<3, 3 * mul_v(2 * V1, 2 * V2)> r;
    static_assert(std::is_same<RB, Fp<3, 3 * mul_v(2 * V1, 2 * V2)>>::value, "bound of the recombined product");
    r.a = ra;
    r.b = rb;
    return r;
  }
}

// fp6_mul_q with the Fp2 products of row_fp2_mul
template <bool TRI>
__device__ __forceinline__ Fp2 fp6_mul_r(const Fp2& x, const Fp2& y, const RowCtx& cx) {
  if constexpr (!TRI) {
    return fp6_mul_q(x, y, cx.q);
  } else {
    const int q = cx.q;
    using W = Fp2T<2, 2 * STORE_V>;
    const W xo = fp2_add(qperm<QP_ROT1>(x), qperm<QP_ROT2>(x));     // the other two columns
    const W yo = fp2_add(qperm<QP_ROT1>(y), qperm<QP_ROT2>(y));
    using M = decltype(fp2_norm(row_fp2_mul<true>(xo, yo, cx)));
    M p, cr;
    p.a = fp_zero(); p.b = fp_zero(); cr = p;
#pragma unroll 1
    for (int r = 0; r < 2; ++r) {
      const W a = fp2_sel(r != 0, xo, fp2_widen<2, 2 * STORE_V>(x));
      const W b = fp2_sel(r != 0, yo, fp2_widen<2, 2 * STORE_V>(y));
      const M m = fp2_norm(row_fp2_mul<true>(a, b, cx));
      p = fp2_sel(r == 0, m, p);
      cr = fp2_sel(r != 0, m, cr);
    }
    const auto c = fp2_norm(fp2_sub(fp2_sub(cr, qperm<QP_ROT1>(p)), qperm<QP_ROT2>(p)));
    using C = std::remove_const_t<decltype(c)>;
    C pw;
    pw.a = p.a; pw.b = p.b;
    const C ps = qperm<QP_SWAP12>(pw), cs = qperm<QP_SWAP12>(c);
    const C A = fp2_sel(q == 0, ps, cs), B = fp2_sel(q == 0, cs, ps);
    const auto xiB = fp2_mul_xi(B);
    using X = std::remove_const_t<decltype(xiB)>;
    X Bw;
    Bw.a = B.a; Bw.b = B.b;
    return fp2_fit(fp2_add(A, fp2_sel(q == 2, Bw, xiB)));
  }
}

// the value quad K of the row holds (same column)
template <int K, int L, int V>
__device__ __forceinline__ Fp2T<L, V> rowq(const Fp2T<L, V>& a, const RowCtx& c) {
  Fp2T<L, V> r;
#pragma unroll
  for (int i = 0; i < NLB; ++i) {
    r.a.v[i] = __builtin_amdgcn_ds_bpermute(c.src[K], a.a.v[i]);
    r.b.v[i] = __builtin_amdgcn_ds_bpermute(c.src[K], a.b.v[i]);
  }
  return r;
}

template <bool TRI>
__device__ __forceinline__ Q12 fp12_mul_row(const Q12& x, const Q12& y, const RowCtx& c) {
  const Fp2 sx = fp2_fit(fp2_add(x.c0, x.c1)), sy = fp2_fit(fp2_add(y.c0, y.c1));
  const Fp2 a = fp2_sel(c.g == 0, x.c0, fp2_sel(c.g == 1, x.c1, sx));
  const Fp2 b = fp2_sel(c.g == 0, y.c0, fp2_sel(c.g == 1, y.c1, sy));
  const Fp2 m = fp6_mul_r<TRI>(a, b, c);
  const Fp2 t0 = rowq<0>(m, c), t1 = rowq<1>(m, c), s = rowq<2>(m, c);
  Q12 o;
  o.c1 = fp2_fit(fp2_sub(fp2_sub(s, t0), t1));
  o.c0 = fp2_fit(fp2_add(t0, fp6_mul_v_q(t1, c.q)));
  return o;
}

template <bool TRI>
__device__ __forceinline__ Q12 fp12_sqr_row(const Q12& x, const RowCtx& c) {
  const Fp2 s0 = fp2_fit(fp2_add(x.c0, x.c1));
  const Fp2 s1 = fp2_fit(fp2_add(x.c0, fp6_mul_v_q(x.c1, c.q)));
  const Fp2 m = fp6_mul_r<TRI>(fp2_sel(c.g == 0, x.c0, s0), fp2_sel(c.g == 0, x.c1, s1), c);
  const Fp2 ab = rowq<0>(m, c), m2 = rowq<1>(m, c);
  Q12 o;
  o.c0 = fp2_fit(fp2_sub(fp2_sub(m2, ab), fp6_mul_v_q(ab, c.q)));
  o.c1 = fp2_fit(fp2_dbl(ab));
  return o;
}

// f * (c0 + c1 v + c4 v w): quad 0 forms f.c0 * (c0, c1, 0), quad 1 (f.c0 + f.c1) * (c0, c1 + c4, 0), quad 2
// f.c1 * (0, c4, 0) = v * (f.c1 scaled by c4) -- three sparse Fp6 products in the time of one
template <bool TRI>
__device__ __forceinline__ Q12 fp12_mul_by_014_row(const Q12& f, const Fp2& c0, const Fp2& c1, const Fp2& c4,
                                                   const RowCtx& c) {
  const Fp2 z = fp2_zero();
  const Fp2 o = fp2_fit(fp2_add(c1, c4));
  const Fp2 y01 = fp2_sel(c.q == 0, c0, fp2_sel(c.q == 1, c1, z));
  const Fp2 y0o = fp2_sel(c.q == 0, c0, fp2_sel(c.q == 1, o, z));
  const Fp2 y4 = fp2_sel(c.q == 1, c4, z);
  const Fp2 fs = fp2_fit(fp2_add(f.c0, f.c1));
  const Fp2 a = fp2_sel(c.g == 0, f.c0, fp2_sel(c.g == 1, fs, f.c1));
  const Fp2 b = fp2_sel(c.g == 0, y01, fp2_sel(c.g == 1, y0o, y4));
  const Fp2 m = fp6_mul_r<TRI>(a, b, c);
  const Fp2 aa = rowq<0>(m, c), s = rowq<1>(m, c), bb = rowq<2>(m, c);
  Q12 r;
  r.c1 = fp2_fit(fp2_sub(fp2_sub(s, aa), bb));
  r.c0 = fp2_fit(fp2_add(fp6_mul_v_q(bb, c.q), aa));
  return r;
}

template <bool TRI>
__device__ __forceinline__ Q12 fp12_cyclotomic_sqr_row(const Q12& x, const RowCtx& c) {
  const int q = c.q;
  const Fp2 g = qperm<QP_ROT1>(x.c1);
  const Fp2 a = fp2_sel(q == 2, g, x.c0), b = fp2_sel(q == 2, x.c0, g);
  using W = Fp2T<3, 3 * STORE_V>;
  const W s1 = fp2_widen<3, 3 * STORE_V>(fp2_add(a, b));
  const W s2 = fp2_add(fp2_mul_xi(b), a);
  const W u = fp2_sel(c.g == 0, fp2_widen<3, 3 * STORE_V>(a), s1);
  const W v = fp2_sel(c.g == 0, fp2_widen<3, 3 * STORE_V>(b), s2);
  const Fp2 m = fp2_fit(row_fp2_mul<TRI>(u, v, c));
  const Fp2 tmp = rowq<0>(m, c), s = rowq<1>(m, c);
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

template <bool TRI>
__device__ __forceinline__ Q12 fp12_frob_row(const Q12& x, const RowCtx& c) {
  const int q = c.q;
  const Fp2 g0 = fp2_sel(q == 0, fp2_one(), fp2_sel(q == 1, gamma_const(2), gamma_const(4)));
  const Fp2 g1 = fp2_sel(q == 0, gamma_const(1), fp2_sel(q == 1, gamma_const(3), gamma_const(5)));
  const Fp2 a = fp2_conj(fp2_sel(c.g == 0, x.c0, x.c1)), b = fp2_sel(c.g == 0, g0, g1);
  const Fp2 m = fp2_fit(row_fp2_mul<TRI>(a, b, c));
  Q12 o;
  o.c0 = rowq<0>(m, c);
  o.c1 = rowq<1>(m, c);
  return o;
}

template <bool TRI>
__device__ __attribute__((noinline)) Q12 exp_by_x_row(const Q12& f, const RowCtx& c) {
  Q12 acc = f;
#pragma unroll 1
  for (int bit = 62; bit >= 0; --bit) {
    acc = fp12_cyclotomic_sqr_row<TRI>(acc, c);
    if ((X_ABS >> bit) & 1) acc = fp12_mul_row<TRI>(acc, f, c);
  }
  return fp12_conj_q(acc);
}

// same chain as final_exponentiation_q
template <bool TRI>
__device__ __forceinline__ Q12 final_exponentiation_row(const Q12& f, const RowCtx& c) {
  const int q = c.q;
  Q12 f2;
  {
    Fp12 full, inv;
    q12_gather(&full, f);
    fp12_inv(&inv, &full);
    Q12 t = fp12_mul_row<TRI>(fp12_conj_q(f), q12_scatter(&inv, q), c);
    f2 = fp12_mul_row<TRI>(fp12_frob_row<TRI>(fp12_frob_row<TRI>(t, c), c), t, c);
  }
  Q12 y = fp12_mul_row<TRI>(exp_by_x_row<TRI>(f2, c), fp12_conj_q(f2), c);
  y = fp12_mul_row<TRI>(exp_by_x_row<TRI>(y, c), fp12_conj_q(y), c);
  y = fp12_mul_row<TRI>(exp_by_x_row<TRI>(y, c), fp12_frob_row<TRI>(y, c), c);
  Q12 t = exp_by_x_row<TRI>(exp_by_x_row<TRI>(y, c), c);
  t = fp12_mul_row<TRI>(t, fp12_frob_row<TRI>(fp12_frob_row<TRI>(y, c), c), c);
  y = fp12_mul_row<TRI>(t, fp12_conj_q(y), c);
  t = fp12_mul_row<TRI>(fp12_cyclotomic_sqr_row<TRI>(f2, c), f2, c);
  return fp12_mul_row<TRI>(y, t, c);
}

// pairing_check2_quad_prepared with the tower on three quads; the line scaling stays inside each quad (replicated)
template <bool TRI>
__device__ __attribute__((noinline)) uint32_t pairing_check2_row_prepared(const uint32_t* g1, const uint32_t* prep,
                                                                          const RowCtx& c) {
  const int q = c.q;
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
    f = fp12_sqr_row<TRI>(f, c);
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
        if (!skip) f = fp12_mul_by_014_row<TRI>(f, l0, l1, l4, c);
      }
      my_line += G2_LINE_WORDS;
      line0 += G2_LINE_WORDS;
    }
  }
  f = fp12_conj_q(f);
  const Q12 e = final_exponentiation_row<TRI>(f, c);
  Fp12 full;
  q12_gather(&full, e);
  const bool one = fp12_is_one(&full);
  if (!all_ok) return PST_INVALID;
  return one ? PST_OK : PST_FAIL;
}

}  // namespace bls
