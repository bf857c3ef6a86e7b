// vrf_core.cuh -- per-item VRF algorithms (one item == one GPU lane).
//
// Each function names the ark-vrf interface it replaces; the only place those names exist in
// /root/reference is the re-export list src/lib.rs:13-17.  The bit-exact recipe followed is
// SURVEY.md Appendix A.
#pragma once
#include "fe.cuh"
#include "fr.cuh"
#include "sha512.cuh"
#include "te.cuh"

VRF_NS_BEGIN

enum : uint32_t { ST_OK = 0, ST_VERIFICATION_FAILURE = 1, ST_INVALID_DATA = 2 };

// BytesViewLite, DevTables (shared, read-only device tables built once per context): vrf_types.h

constexpr int WIN_ENTRIES = 8;                       // signed radix-16: |digit| in 1..8
constexpr int WIN_TABLE_WORDS = WIN_ENTRIES * PTC_WORDS;   // 288 words = 1152 B per base

// ------------------------------------------------------------------------ suites
// [ref src/lib.rs:16 `Suite`, :14 `suites`]  What is compiled in is the ARITHMETIC of a suite: the curve (a, d,
// cofactor, r), whether it has the GLV endomorphism, how subgroup membership is decided and which hash-to-curve
// construction runs.  Everything a `Suite` impl states as data -- SUITE_ID, the hash-to-curve DST, the generator
// and the Pedersen blinding base -- comes from the context's descriptor (vrfhip_suite_desc): the byte strings
// (SuiteStr, fe.cuh), and the fixed-base tables built from the descriptor's points at context creation.
// How prime-order subgroup membership of a decoded point is decided (in_prime_subgroup)
enum : int { SUBGROUP_2DESCENT = 0, SUBGROUP_TATE8 = 1, SUBGROUP_ORDER = 2, SUBGROUP_HALVE_TATE4 = 3 };

#if VRF_FIELD == 0
struct SuiteBS : CurveBS {
  static constexpr bool HAS_GLV = true;        // Bandersnatch endomorphism (te_psi, glv_decompose_bs)
  static constexpr int SUBGROUP = SUBGROUP_2DESCENT;   // E(Fq) = Z2 x Z2 x Zr: the prime-order subgroup is 2E
  static constexpr bool H2C_ELL2 = true;       // Input::new = Elligator 2 (else try-and-increment)
};

// JubJub: a = -1, cofactor 8, try-and-increment hash-to-curve, no GLV.
struct SuiteJJ : CurveJJ {
  static constexpr bool HAS_GLV = false;
  static constexpr int SUBGROUP = SUBGROUP_TATE8;   // cyclic 8-torsion, 8 | q - 1: Tate pairing with a point of order 8
  static constexpr bool H2C_ELL2 = false;
  static VRF_HD FeN tate_a3() { return fe_const(vrfk::JJ_TATE_A3_M); }
  static VRF_HD FeN tate_b() { return fe_const(vrfk::JJ_TATE_B_M); }
  static VRF_HD FeN tate_c4() { return fe_const(vrfk::JJ_TATE_C4_M); }
  static VRF_HD FeN tate_c8() { return fe_const(vrfk::JJ_TATE_C8_M); }
  static VRF_HD FeN tate_y4() { return fe_const(vrfk::JJ_TATE_Y4_M); }
  static VRF_HD FeN tate_y8() { return fe_const(vrfk::JJ_TATE_Y8_M); }
  static VRF_HD FeN tate_lam4() { return fe_const(vrfk::JJ_TATE_LAM4_M); }
  static VRF_HD FeN tate_lam8() { return fe_const(vrfk::JJ_TATE_LAM8_M); }
};
#elif VRF_FIELD == 1
// Ed25519 (`suites::ed25519`, upstream "Ed25519_SHA-512_TAI"): a = -1, cofactor 8, try-and-increment, no GLV.  The
// rational 2-power torsion is cyclic of order 8 but 8 does not divide q - 1 (q = 5 mod 8), so the order-8 Tate pairing
// does not exist over Fq: membership is one halving and the order-4 pairing (subgroup_by_halving_tate4 below; arkworks'
// own test, r * P = O, stays compiled as subgroup_by_order and the tests hold the two against each other).
struct SuiteED : CurveED {
  static constexpr bool HAS_GLV = false;
  static constexpr int SUBGROUP = SUBGROUP_HALVE_TATE4;
  static constexpr bool H2C_ELL2 = false;
};
#elif VRF_FIELD == 2
// Baby-JubJub (`suites::baby_jubjub`, upstream "BabyJubJub_SHA-512_TAI"): a = 1, cofactor 8, try-and-increment, no GLV.
struct SuiteBJ : CurveBJ {
  static constexpr bool HAS_GLV = false;
  static constexpr int SUBGROUP = SUBGROUP_TATE8;   // same torsion structure as JubJub; q - 1 has 2-adicity 28
  static constexpr bool H2C_ELL2 = false;
  static VRF_HD FeN tate_a3() { return fe_const(vrfk::BJ_TATE_A3_M); }
  static VRF_HD FeN tate_b() { return fe_const(vrfk::BJ_TATE_B_M); }
  static VRF_HD FeN tate_c4() { return fe_const(vrfk::BJ_TATE_C4_M); }
  static VRF_HD FeN tate_c8() { return fe_const(vrfk::BJ_TATE_C8_M); }
  static VRF_HD FeN tate_y4() { return fe_const(vrfk::BJ_TATE_Y4_M); }
  static VRF_HD FeN tate_y8() { return fe_const(vrfk::BJ_TATE_Y8_M); }
  static VRF_HD FeN tate_lam4() { return fe_const(vrfk::BJ_TATE_LAM4_M); }
  static VRF_HD FeN tate_lam8() { return fe_const(vrfk::BJ_TATE_LAM8_M); }
};
#endif

VRF_HD void put_suite_id(Sha512& h, const SuiteStr& ss) { sha512_put_packed(h, ss.suite_id_w, ss.suite_id_len); }

// ------------------------------------------------------------------------ batch inversion
template <bool CT = false, int N, int L, int V>
VRF_HD void fe_batch_inv(FeN (&out)[N], const Fe<L, V> (&in)[N]) {
  FeN pre[N];
  pre[0] = fe_mul(in[0], fe_one());
#pragma unroll
  for (int i = 1; i < N; ++i) pre[i] = fe_mul(pre[i - 1], in[i]);
  FeN acc = fe_inv<CT>(pre[N - 1]);
#pragma unroll
  for (int i = N - 1; i > 0; --i) {
    out[i] = fe_mul(acc, pre[i - 1]);
    acc = fe_mul(acc, in[i]);
  }
  out[0] = acc;
}

// ------------------------------------------------------------------------ point decoding
// [ref src/lib.rs:14 `codec`] ArkworksCodec::point_decode without the subgroup check
// (SURVEY.md A.1).  Split in two so that callers can share one inversion across points.
struct DecodeA {
  FeN y, den;
  Fe<1, 6> num;
  bool flag, ok;
};
template <class C>
VRF_HD DecodeA decode_phase_a(const uint32_t enc[8]) {
  DecodeA r;
  uint32_t w[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) w[i] = enc[i];
  r.flag = (w[7] >> 31) != 0;
  w[7] &= 0x7fffffffu;
  r.ok = !u256_ge(w, vrfk::Q32);
  r.y = fe_from_u256(w);
  FeN y2 = fe_sqr(r.y);
  r.num = fe_norm(fe_sub(y2, fe_one()));                                 // y^2 - 1
  FeN den;                                                               // d*y^2 - a
  if constexpr (C::A_PLUS_ONE) den = fe_canon(fe_sub(fe_mul(y2, C::d()), fe_one()));
  else den = fe_canon(fe_add(fe_mul(y2, C::d()), C::aneg_m()));
  bool dz = true;
#pragma unroll
  for (int i = 0; i < NL; ++i) dz = dz && (den.v[i] == 0);
  r.ok = r.ok && !dz;
  r.den = fe_select(dz, fe_one(), den);
  return r;
}
// x from w = num/den; returns validity
template <class C>
VRF_HD bool decode_phase_b(Fe<1, 4>& x_out, const DecodeA& a, const FeN& den_inv, const SqrtTables& T) {
  FeN w = fe_mul(a.num, den_inv);
  FeN root;
  bool sq = fe_sqrt_or_zsqrt(root, w, T);
  uint32_t xw[8];
  fe_to_u256(xw, root);
  uint32_t nz = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) nz |= xw[i];
  sq = sq || (nz == 0);
  bool greater = te_x_sign(xw, T.str.flags);
  x_out = fe_norm(fe_cneg(greater != a.flag, root));
  // x == 0: -0 = K*q, still a valid representation of zero.  arkworks accepts x = 0 with either flag; RFC 8032's decoding
  // (SS_SIGN_PARITY) rejects x = 0 with the flag set
  if ((T.str.flags & SS_SIGN_PARITY) && nz == 0 && a.flag) sq = false;
  return a.ok && sq;
}

// Prime-order subgroup membership of a decoded curve point by 2-descent, for curves with full rational
// 2-torsion and cofactor 4 (Bandersnatch): the subgroup is 2E, and on the Montgomery model
// B v^2 = u (u - e2)(u - e3), u = (1 + y)/(1 - y), a point lies in 2E iff B u, B (u - e2), B (u - e3) are
// squares.  Their product is a square, so two tests decide; cleared of the denominator 1 - y:
//     B (1 - y^2)   and   B (1 - y) ((1 + e2) y + (1 - e2))   are non-zero squares   (or y = 1: identity).
// Two Jacobi symbols (fe.cuh: ~0.04 M instruction slots) instead of r*P = O (~0.6 M).  [codec: arkworks'
// checked deserialisation, `codec` src/lib.rs:14]; tests compare with r*P = O on every coset.
#if VRF_FIELD == 0
template <class S>
VRF_HD bool subgroup_by_2descent(const FeN& y, const SqrtTables& T) {
  const FeN one = fe_one();
  const bool is_identity = fe_eq(y, one);
  const auto omy = fe_norm(fe_sub(one, y));                                           // 1 - y
  const FeN t1 = fe_mul(fe_mul(omy, fe_add(one, y)), fe_const(vrfk::BS_DESC_B_M));    // B (1 - y^2)
  const auto lin = fe_norm(fe_add(fe_mul(y, fe_const(vrfk::BS_DESC_1PE2_M)), fe_const(vrfk::BS_DESC_1ME2_M)));
  const FeN t2 = fe_mul(fe_mul(omy, lin), fe_const(vrfk::BS_DESC_B_M));
  const bool s1 = fe_is_nonzero_square(t1, T);
  const bool s2 = fe_is_nonzero_square(t2, T);
  return is_identity || (s1 && s2);
}
#endif

// check_mask bits (= the complement of include/vrfhip.h VRFHIP_FLAG_PREVALIDATED_*): which point classes get
// the prime-order-subgroup test when they are decoded
enum : uint32_t { CHK_PUBLIC = 1, CHK_INPUT = 2, CHK_OUTPUT = 4, CHK_PROOF = 8,
                  CHK_CT_TABLES = 0x100 };      // provers: window-table lookups read all 8 entries (VRFHIP_FLAG_CT_TABLES)

// ------------------------------------------------------------------------ window tables
// multiples 1..8 of an affine point, cached form, written to `tab` (WIN_TABLE_WORDS words).
// One unified-add call site in a loop (the unified law also doubles).
template <class C>
VRF_HD void build_win_table_from(uint32_t* tab, PtE acc) {
  PtC c1 = te_to_cached<C>(acc);
  ptc_store(tab, c1);
#pragma unroll 1
  for (int j = 1; j < WIN_ENTRIES; ++j) {
    acc = te_add_cached<C>(acc, c1, false);
    ptc_store(tab + j * PTC_WORDS, te_to_cached<C>(acc));
  }
}
// the pair of tables GLV needs: multiples of P and of psi(P) (2 * WIN_TABLE_WORDS words).
// One add call site: a two-trip loop over {P, psi(P)}.
template <class C>
VRF_HD void build_glv_tables(uint32_t* tab, const FeP& x, const FeP& y) {
  PtE p = te_from_affine(x, y);
  if constexpr (!C::HAS_GLV) {
    build_win_table_from<C>(tab, p);            // suites without an endomorphism: one 253-bit table
    return;
  }
  PtE q = te_psi<C>(p);
#pragma unroll 1
  for (int t = 0; t < 2; ++t) {
    PtE acc;
    acc.X = fe_select(t == 0, p.X, q.X); acc.Y = fe_select(t == 0, p.Y, q.Y);
    acc.Z = fe_select(t == 0, p.Z, q.Z); acc.T = fe_select(t == 0, p.T, q.T);
    build_win_table_from<C>(tab + t * WIN_TABLE_WORDS, acc);
  }
}
template <class C>
VRF_HD void build_win_table(uint32_t* tab, const FeP& x, const FeP& y) {
  PtE acc = te_from_affine(x, y);
  PtC c1 = te_to_cached<C>(acc);
  ptc_store(tab, c1);
#pragma unroll 1
  for (int j = 1; j < WIN_ENTRIES; ++j) {
    acc = te_add_cached<C>(acc, c1, false);
    ptc_store(tab + j * PTC_WORDS, te_to_cached<C>(acc));
  }
}

VRF_HD PtC win_lookup(const uint32_t* tab, int digit) {   // |digit| in 0..8
  int mag = digit < 0 ? -digit : digit;
  int idx = mag > 0 ? mag - 1 : 0;
  PtC e = ptc_load(tab + idx * PTC_WORDS);
  PtC id = te_identity_cached();
  bool z = mag == 0;
  e.X = fe_select(z, id.X, e.X);
  e.Y = fe_select(z, id.Y, e.Y);
  e.Z = fe_select(z, id.Z, e.Z);
  e.dT = fe_select(z, id.dT, e.dT);
  return e;
}

// The same lookup with a memory access pattern that does not depend on the digit: all eight entries are read and the
// wanted one is kept by masks.  For the provers' per-proof tables, whose digits are digits of sk or of the nonce k
// (VRFHIP_FLAG_CT_TABLES; arkworks' own mul_bigint is not constant-time either -- this is hardening).
template <bool CT>
VRF_HD PtC win_lookup_t(const uint32_t* tab, int digit) {
  if constexpr (!CT) {
    return win_lookup(tab, digit);
  } else {
    const int mag = digit < 0 ? -digit : digit;
    PtC e = te_identity_cached();
#pragma unroll 1
    for (int j = 0; j < WIN_ENTRIES; ++j) {
      const PtC t = ptc_load(tab + j * PTC_WORDS);
      const bool hit = mag == j + 1;
      e.X = fe_select(hit, t.X, e.X);
      e.Y = fe_select(hit, t.Y, e.Y);
      e.Z = fe_select(hit, t.Z, e.Z);
      e.dT = fe_select(hit, t.dT, e.dT);
    }
    return e;
  }
}

VRF_HD void sel8(uint32_t out[8], bool c, const uint32_t a[8], const uint32_t b[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = c ? a[i] : b[i];
}

// sa*A + sb*B by Straus with signed radix-16 digits; negB flips the sign of the B terms.
// reca / recb are recoded scalars (scalar_recode_signed4).  One te_dbl and one te_add call
// site: the loop body is the whole hot path of IETF verification.
// topB: the highest window of B's scalar that can be non-zero (challenge_top_window below; 63 = all of them).
template <class C>
VRF_HD PtE straus2(const uint32_t* tabA, const uint32_t reca[8], const uint32_t* tabB,
                   const uint32_t recb[8], bool negB, int topB = 63) {
  PtE acc = te_identity();
#pragma unroll 1
  for (int w = 63; w >= 0; --w) {
    if (w != 63) {
#pragma unroll 1
      for (int j = 0; j < 4; ++j) acc = te_dbl<C>(acc, j == 3);
    }
    const int terms = w <= topB ? 2 : 1;
#pragma unroll 1
    for (int t = 0; t < terms; ++t) {
      uint32_t rec[8];
      sel8(rec, t != 0, recb, reca);
      const uint32_t* tab = t ? tabB : tabA;
      int d = scalar_digit4(rec, w);
      acc = te_add_cached<C>(acc, win_lookup(tab, d), (d < 0) != (t != 0 && negB), t + 1 < terms || w == 0);
    }
  }
  return acc;
}

// A challenge is CHALLENGE_LEN bytes of a hash: c < 2^(8 len), and its signed radix-16 recoding has no non-zero digit
// above window 2 len (window 2 len itself holds the recoding's carry).  Ladders over c start there: with the RFC's
// 16-byte challenges that is half of the doublings of -c*Y.  A proof carrying a larger c (it has 32 bytes to put one
// in) gets the ladder of its low windows: some other point, whose challenge hash -- below 2^(8 len) -- still
// cannot equal that c; the verdict is the one the full ladder gives.
VRF_HD int challenge_top_window(const SuiteStr& ss) { return ss.challenge_len < 32u ? (int)(2u * ss.challenge_len) : 63; }

// GLV Straus: sum_{t<4} (+/-) k_t * P_t with 128-bit k_t, signed radix-16, 32 windows, 128
// doublings.  tabs[t] are window tables, rec[t] recoded magnitudes, neg[t] the term's sign.
struct Straus4 {
  const uint32_t* tab[4];
  uint32_t rec[4][4];
  bool neg[4];
};
template <class C, int NT = 4, bool CT = false>
VRF_HD PtE straus4(const Straus4& q) {
  PtE acc = te_identity();
#pragma unroll 1
  for (int w = 31; w >= 0; --w) {
    if (w != 31) {
#pragma unroll 1
      for (int j = 0; j < 4; ++j) acc = te_dbl<C>(acc, j == 3);
    }
#pragma unroll 1
    for (int t = 0; t < NT; ++t) {
      uint32_t rec[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        rec[i] = t == 0 ? q.rec[0][i] : t == 1 ? q.rec[1][i] : t == 2 ? q.rec[2][i] : q.rec[3][i];
      const uint32_t* tab = t == 0 ? q.tab[0] : t == 1 ? q.tab[1] : t == 2 ? q.tab[2] : q.tab[3];
      bool neg = t == 0 ? q.neg[0] : t == 1 ? q.neg[1] : t == 2 ? q.neg[2] : q.neg[3];
      int d = scalar_digit4_128(rec, w);
      acc = te_add_cached<C>(acc, win_lookup_t<CT>(tab, d), (d < 0) != neg, t != NT - 1 || w == 0);   // doublings follow
    }
  }
  return acc;
}

// k*P for one window table (prove: Gamma = sk*H, kH)
template <class C, bool CT = false>
VRF_HD PtE win_mul(const uint32_t* tab, const uint32_t rec[8], bool negate = false, int top = 63) {
  PtE acc = te_identity();
#pragma unroll 1
  for (int w = top; w >= 0; --w) {
    if (w != top) {
#pragma unroll 1
      for (int j = 0; j < 4; ++j) acc = te_dbl<C>(acc, j == 3);
    }
    int d = scalar_digit4(rec, w);
    acc = te_add_cached<C>(acc, win_lookup_t<CT>(tab, d), (d < 0) != negate, w == 0);
  }
  return acc;
}

// Prime-order subgroup membership for a curve whose rational 2-power torsion is cyclic of order 8 (JubJub: one
// rational point of order 2, cofactor 8): P lies in 8E iff the reduced Tate pairing with a generator T8 of the
// 8-torsion is trivial, t_8(T8, P) = f_{8,T8}(P)^((q-1)/8) = 1.  f is three doubling steps of Miller's loop with
// FIXED lines (T8, T4 = 2 T8, T2 = 4 T8 are constants: tools/gen_constants.py derives them and the formula below,
// and checks it against r*P = O), evaluated without inversions modulo 8th powers:
//   Yn = 1 + y, W = (1 - y) x, V2 = (1 + y) x, Xn = V2 + (A/3) W,
//   V4 = Xn - c4 W, L4 = Yn - y4 W - lam4 V4, V8 = Xn - c8 W, L8 = Yn - y8 W - lam8 V8,
//   f = L8^4 L4^2 V4^4 (B V2 W)^7,
// and f is an 8th power iff 8 divides the 2-adic discrete logarithm of f (one fixed exponentiation, ~60 k
// instruction slots; arkworks' own test, r*P = O, is a 252-bit scalar multiplication: ~330 k).  f = 0 only on
// small-order points; the identity is accepted explicitly.
template <class S>
VRF_HD bool subgroup_by_tate8(const FeN& x, const FeN& y, const SqrtTables& T) {
  const FeN one = fe_one();
  const bool is_identity = fe_is_zero(x) && fe_eq(y, one);
  const FeN Yn = fe_mul(fe_add(one, y), one);                              // 1 + y (reduced)
  const FeN W = fe_mul(fe_sub(one, y), x);                                 // (1 - y) x
  const FeN V2 = fe_mul(Yn, x);                                            // (1 + y) x
  const FeN Xn = fe_mul(fe_add(V2, fe_mul(W, S::tate_a3())), one);
  const FeN V4 = fe_mul(fe_sub(Xn, fe_mul(W, S::tate_c4())), one);
  const FeN V8 = fe_mul(fe_sub(Xn, fe_mul(W, S::tate_c8())), one);
  const FeN t4 = fe_mul(fe_add(fe_mul(W, S::tate_y4()), fe_mul(V4, S::tate_lam4())), one);
  const FeN t8 = fe_mul(fe_add(fe_mul(W, S::tate_y8()), fe_mul(V8, S::tate_lam8())), one);
  const FeN L4 = fe_mul(fe_sub(Yn, t4), one);
  const FeN L8 = fe_mul(fe_sub(Yn, t8), one);
  const FeN Z = fe_mul(fe_mul(V2, W), S::tate_b());
  const FeN Z2 = fe_sqr(Z), Z4 = fe_sqr(Z2);
  const FeN Z7 = fe_mul(fe_mul(Z4, Z2), Z);
  const FeN L8_4 = fe_sqr(fe_sqr(L8)), V4_4 = fe_sqr(fe_sqr(V4)), L4_2 = fe_sqr(L4);
  const FeN f = fe_mul(fe_mul(L8_4, L4_2), fe_mul(V4_4, Z7));
  const bool eighth_power = !fe_is_zero(f) && (fe_dlog2_low8(f, T) & 7u) == 0;
  return is_identity || eighth_power;
}

// Prime-order subgroup membership of a decoded point (x, y) [ref src/lib.rs:14 `codec`: arkworks' checked
// deserialisation].  Bandersnatch: 2-descent on y (two Jacobi symbols).  JubJub, Baby-JubJub: Tate pairing with the
// 8-torsion.  Ed25519: one halving + the order-4 Tate pairing.
// r * P = O by double-and-add over the bits of the (fixed) subgroup order: scalar control flow, every lane the same
// shape.  arkworks' own `is_in_correct_subgroup_assuming_on_curve`; used where no cheaper character exists (Ed25519:
// 252 doublings + 63 additions, the order 2^252 + 2^124.4.. is sparse at the top).
template <class S>
VRF_HD bool subgroup_by_order(const FeN& x, const FeN& y) {
  const PtC pc = te_to_cached<S>(te_from_affine(x, y));
  PtE acc = te_identity();
#pragma unroll 1
  for (int i = 255; i >= 0; --i) {
    const bool bit = (S::r32(i >> 5) >> (i & 31)) & 1u;          // wave-uniform
    acc = te_dbl<S>(acc, bit);
    if (bit) acc = te_add_cached<S>(acc, pc, false);
  }
  return fe_is_zero(acc.X) && fe_eq(acc.Y, acc.Z);
}

#if VRF_FIELD == 1
// w^((q-1)/4) == 1: w is a non-zero 4th power.  q = 2^255 - 19 = 5 (mod 8): the exponent is the t of the square-root
// plan (q - 1 = 4 t), so this is the root's exponentiation without the root.
VRF_HD bool fe_is_nonzero_fourth_power(const FeN& w) {
  const FeN v = fe_pow_prog(w, vrfk::POW_SQRT_PROG);        // w^((t-1)/2)
  const FeN b = fe_mul(fe_mul(w, v), v);                    // w^t
  return limbs_eq(fe_canon(b), vrfk::ONE_M);
}

// Prime-order subgroup membership on Ed25519 without the 252-bit multiplication.  The rational 2-power torsion is cyclic
// of order 8 but q = 5 (mod 8): Fq holds the 4th roots of unity and not the 8th, so the order-8 Tate pairing that decides
// JubJub and Baby-JubJub does not exist here.  What exists is one level of 2-descent and the order-4 pairing:
//   on v^2 = u^3 + A u^2 + u (u = (1+y)/(1-y), v = c u/x, c = sqrt(-(A+2))) the only rational point of order 2 is (0,0),
//   P is in 2E iff u is a square, and a half Q of P is rational: through the 2-isogeny with kernel (0,0) and its dual,
//     s = sqrt(u), X = A + 2u +- 2v/s (the one of the two that is a square), T = X - A, x_Q = (T + sqrt(T^2 - 4))/2,
//     Y' = 8 X^2 v / ((A^2 - 4) - X^2), y_Q = Y' x_Q^2 / (1 - x_Q^2).
//   Q is defined up to (0,0), which lies in 4E, so P is in 8E iff Q is in 4E iff the reduced Tate pairing with
//   T4 = (1, y4), y4 = sqrt(A+2), is trivial: f_{4,T4}(Q) = x_Q l^2 modulo 4th powers, l = y_Q - y4 x_Q (the tangent at T4
//   passes through (0,0)), i.e. chi_4(x_Q l^2) = 1.
// No quotient is ever formed: s, sqrt(T^2 - 4) are roots of numerators over known denominators and everything else only
// feeds a character.  With n = 1+y, m = 1-y, s' = sqrt(n m), Td = m x s':
//   Tn = 2 n (x s' +- c m), Xn = A Td + Tn (sign: chi(Xn Td) = 1), xn = Tn + sqrt(Tn^2 - 4 Td^2), xd = 2 Td,
//   Bd = Td ((A^2-4) Td^2 - Xn^2)(xd^2 - xn^2), Bn = 8c Xn^2 n s' xn xd - y4 Bd,   accept iff chi_4(xn^3 xd (Bn Bd)^2) = 1.
// Three fixed exponentiations and one Jacobi symbol (~1050 products) against 252 doublings + 63 additions (~2600);
// tools/gen_constants.py (Curve.halve_tate4) derives the constants and checks this formula against r*P = O on every
// coset of the 8-torsion; tests/test_new_suites.py does the same through the compiled code.
template <class S>
VRF_HD bool subgroup_by_halving_tate4(const FeN& x, const FeN& y, const SqrtTables& T) {
  const FeN one = fe_one();
  const bool is_identity = fe_is_zero(x) && fe_eq(y, one);
  const FeN n = fe_mul(fe_add(one, y), one), m = fe_mul(fe_sub(one, y), one);
  const FeN N = fe_mul(n, m);                                              // 1 - y^2: chi(N) = chi(u)
  FeN sp;
  bool ok = fe_sqrt_or_zsqrt(sp, N, T) && !fe_is_zero(N);                  // u = 0 is the point of order 2
  const FeN xs = fe_mul(x, sp), cm = fe_mul(m, fe_const(vrfk::ED_HALVE_C_M));
  const FeN Td = fe_mul(fe_mul(m, x), sp);
  const FeN Tnp = fe_mul(fe_dbl(n), fe_add(xs, cm)), Tnm = fe_mul(fe_dbl(n), fe_sub(xs, cm));
  const FeN ATd = fe_mul(Td, fe_const(vrfk::ED_HALVE_A_M));
  const bool plus = fe_is_nonzero_square(fe_mul(fe_add(ATd, Tnp), Td), T);
  const FeN Tn = fe_select(plus, Tnp, Tnm);
  const FeN Xn = fe_mul(fe_add(ATd, Tn), one);
  const FeN xd = fe_mul(fe_dbl(Td), one);
  const FeN Dn = fe_mul(fe_sub(Tn, xd), fe_add(Tn, xd));                   // Tn^2 - 4 Td^2
  FeN rD;
  ok = fe_sqrt_or_zsqrt(rD, Dn, T) && ok;
  const FeN xn = fe_mul(fe_add(Tn, rD), one);
  const FeN Xn2 = fe_sqr(Xn), xn2 = fe_sqr(xn);
  const FeN K1 = fe_mul(fe_sub(fe_mul(fe_sqr(Td), fe_const(vrfk::ED_HALVE_BP_M)), Xn2), one);
  const FeN K2 = fe_mul(fe_sub(fe_sqr(xd), xn2), one);
  const FeN Bd = fe_mul(fe_mul(Td, K1), K2);
  const FeN t8 = fe_mul(fe_mul(fe_mul(Xn2, fe_const(vrfk::ED_HALVE_C8_M)), fe_mul(n, sp)), fe_mul(xn, xd));
  const FeN Bn = fe_mul(fe_sub(t8, fe_mul(Bd, fe_const(vrfk::ED_HALVE_Y4_M))), one);
  const FeN W = fe_mul(fe_mul(fe_mul(xn2, xn), xd), fe_sqr(fe_mul(Bn, Bd)));
  ok = fe_is_nonzero_fourth_power(W) && ok;                                // W = 0 fails: 0^t = 0
  return is_identity || ok;
}
#endif

template <class S>
VRF_HD bool in_prime_subgroup(const FeN& x, const FeN& y, const SqrtTables& T) {
  if constexpr (S::SUBGROUP == SUBGROUP_HALVE_TATE4) {
#if VRF_FIELD == 1
    return subgroup_by_halving_tate4<S>(x, y, T);
#else
    return false;
#endif
  } else if constexpr (S::SUBGROUP == SUBGROUP_ORDER) return subgroup_by_order<S>(x, y);
  else if constexpr (S::SUBGROUP == SUBGROUP_TATE8) return subgroup_by_tate8<S>(x, y, T);
  else {
#if VRF_FIELD == 0
    return subgroup_by_2descent<S>(y, T);
#else
    return false;
#endif
  }
}

// ---- fixed-base tables of the suite's generators G and B: signed GCB-bit windows, no doublings ----
// k = sum_w d_w 2^(GCB w), d_w in [-2^(GCB-1), 2^(GCB-1)); row w holds j * 2^(GCB w) * Base for j = 1..2^(GCB-1)
// as affine-cached entries.  GCB = 16 on the device: 16 mixed additions per scalar from a 56.6 MB table per
// base (HBM / MALL resident, next entry prefetched while the current addition runs) instead of 32 from an
// L2-resident 8-bit comb.  The host simulation compiles the same code with GCB = 8 (-DVRF_GCOMB_BITS=8).
#ifndef VRF_GCOMB_BITS
#define VRF_GCOMB_BITS 16
#endif
constexpr int GCB = VRF_GCOMB_BITS;
static_assert(GCB == 8 || GCB == 16, "window must divide 32");
constexpr int GC_ROWS = 256 / GCB;
constexpr int GC_COLS = 1 << (GCB - 1);
constexpr size_t GCOMB_WORDS = (size_t)GC_ROWS * GC_COLS * PTA_WORDS;
constexpr int GC_SEG = GC_COLS < 256 ? GC_COLS : 256;            // entries built by one lane (one inversion)
constexpr int GC_SEGS = GC_COLS / GC_SEG;

// signed digit of window w with the carry of the windows below; k < 2^253 so the top window cannot overflow
VRF_HD int gcomb_digit(const uint32_t k[8], int w, uint32_t& carry) {
  const int bit = w * GCB;
  uint32_t word = k[0];
#pragma unroll
  for (int i = 1; i < 8; ++i)
    if ((bit >> 5) == i) word = k[i];
  uint32_t chunk = ((word >> (bit & 31)) & ((1u << GCB) - 1)) + carry;
  carry = chunk >= (1u << (GCB - 1)) ? 1u : 0u;
  return (int)chunk - (int)(carry << GCB);
}
VRF_HD PtA gcomb_entry(const uint32_t* tab, int w, int d) {
  const int mag = d < 0 ? -d : d;
  PtA e = pta_load(tab + ((size_t)w * GC_COLS + (mag ? mag - 1 : 0)) * PTA_WORDS);
  const PtA id = pta_identity();
  const bool z = mag == 0;
  e.x = fe_select(z, id.x, e.x);
  e.y = fe_select(z, id.y, e.y);
  e.dt = fe_select(z, id.dt, e.dt);
  return e;
}
// acc +/- k*Base
template <class C>
VRF_HD PtE gcomb_add(PtE acc, const uint32_t* tab, const uint32_t k[8], bool neg = false) {
  uint32_t carry = 0;
  int d = gcomb_digit(k, 0, carry);
  PtA e = gcomb_entry(tab, 0, d);
#pragma unroll 1
  for (int w = 0; w < GC_ROWS; ++w) {
    const PtA cur = e;
    const bool sgn = d < 0;
    if (w + 1 < GC_ROWS) {                       // next entry in flight during this addition
      d = gcomb_digit(k, w + 1, carry);
      e = gcomb_entry(tab, w + 1, d);
    }
    acc = te_add_affine<C>(acc, cur, sgn != neg);
  }
  return acc;
}
template <class C>
VRF_HD PtE gcomb_mul(const uint32_t* tab, const uint32_t k[8]) { return gcomb_add<C>(te_identity(), tab, k); }
// k1*G + k2*B (one add call site)
template <class C>
VRF_HD PtE gcomb_mul2(const uint32_t* tab1, const uint32_t k1[8], const uint32_t* tab2, const uint32_t k2[8]) {
  PtE acc = te_identity();
#pragma unroll 1
  for (int t = 0; t < 2; ++t) {
    uint32_t k[8];
    sel8(k, t != 0, k2, k1);
    acc = gcomb_add<C>(acc, t ? tab2 : tab1, k);
  }
  return acc;
}

// count entries start, start + step, start + 2 step, ... made affine with ONE inversion (Montgomery's trick;
// prefix: count x 9 words owned by the caller)
template <class C>
VRF_HD void comb_chain(uint32_t* row, uint32_t* prefix, PtE acc, const PtC& step, int count) {
  FeN run = fe_one();
#pragma unroll 1
  for (int j = 0; j < count; ++j) {
    uint32_t* slot = row + (size_t)j * PTA_WORDS;
    fe_store(slot, acc.X); fe_store(slot + NL, acc.Y); fe_store(slot + 2 * NL, acc.Z);
    fe_store(prefix + (size_t)j * NL, run);
    run = fe_mul(run, acc.Z);
    acc = te_add_cached<C>(acc, step, false);
  }
  FeN inv = fe_inv(run);
#pragma unroll 1
  for (int j = count - 1; j >= 0; --j) {
    uint32_t* slot = row + (size_t)j * PTA_WORDS;
    const FeP X = fe_load<1, 5>(slot), Y = fe_load<1, 5>(slot + NL), Z = fe_load<1, 5>(slot + 2 * NL);
    const FeN zi = fe_mul(inv, fe_load<1, 2>(prefix + (size_t)j * NL));
    inv = fe_mul(inv, Z);
    PtA a;
    a.x = fe_mul(X, zi);
    a.y = fe_mul(Y, zi);
    a.dt = fe_mul(fe_mul(a.x, a.y), C::d());
    pta_store(slot, a);
  }
}
// segment `seg` of row w of a generator table: entries j = seg*GC_SEG + 1 .. (seg+1)*GC_SEG
template <class C>
VRF_HD void gcomb_build_segment(uint32_t* tab, uint32_t* prefix, const FeN& x, const FeN& y, int w, int seg) {
  PtE base = te_from_affine(x, y);
#pragma unroll 1
  for (int i = 0; i < GCB * w; ++i) base = te_dbl<C>(base, true);        // 2^(GCB w) * Base
  const PtC step = te_to_cached<C>(base);
  PtE big = base;
#pragma unroll 1
  for (int i = 1; i < GC_SEG; i <<= 1) big = te_dbl<C>(big, true);        // GC_SEG * base
  const PtC bigc = te_to_cached<C>(big);
  PtE start = te_identity();                                              // seg * big, double and add
#pragma unroll 1
  for (int b = 15; b >= 0; --b) {
    start = te_dbl<C>(start, true);
    PtE sum = te_add_cached<C>(start, bigc, false);
    const bool bit = (seg >> b) & 1;
    start.X = fe_select(bit, sum.X, start.X); start.Y = fe_select(bit, sum.Y, start.Y);
    start.Z = fe_select(bit, sum.Z, start.Z); start.T = fe_select(bit, sum.T, start.T);
  }
  start = te_add_cached<C>(start, step, false);                           // (seg * GC_SEG + 1) * base
  comb_chain<C>(tab + ((size_t)w * GC_COLS + (size_t)seg * GC_SEG) * PTA_WORDS, prefix, start, step, GC_SEG);
}

// k*Base from an 8-bit unsigned comb [32][255] of affine entries (key sets: 881 KB per public key)
template <class C>
VRF_HD PtE comb_mul(const uint32_t* comb, const uint32_t k[8]) {
  PtE acc = te_identity();
#pragma unroll 1
  for (int w = 0; w < 32; ++w) {
    uint32_t word = k[0];
#pragma unroll
    for (int i = 1; i < 8; ++i)
      if ((w >> 2) == i) word = k[i];
    uint32_t d = (word >> ((w & 3) * 8)) & 255u;
    uint32_t idx = d ? d - 1 : 0;
    PtA e = pta_load(comb + ((size_t)w * 255 + idx) * PTA_WORDS);
    PtA id = pta_identity();
    bool z = d == 0;
    e.x = fe_select(z, id.x, e.x);
    e.y = fe_select(z, id.y, e.y);
    e.dt = fe_select(z, id.dt, e.dt);
    acc = te_add_affine<C>(acc, e, false);
  }
  return acc;
}

// acc +/- k*Base from a comb table (continues an accumulator)
template <class C>
VRF_HD PtE comb_add(PtE acc, const uint32_t* comb, const uint32_t k[8], bool neg = false) {
#pragma unroll 1
  for (int w = 0; w < 32; ++w) {
    uint32_t word = k[0];
#pragma unroll
    for (int i = 1; i < 8; ++i)
      if ((w >> 2) == i) word = k[i];
    uint32_t d = (word >> ((w & 3) * 8)) & 255u;
    uint32_t idx = d ? d - 1 : 0;
    PtA e = pta_load(comb + ((size_t)w * 255 + idx) * PTA_WORDS);
    PtA id = pta_identity();
    bool z = d == 0;
    e.x = fe_select(z, id.x, e.x);
    e.y = fe_select(z, id.y, e.y);
    e.dt = fe_select(z, id.dt, e.dt);
    acc = te_add_affine<C>(acc, e, neg);
  }
  return acc;
}
// ------------------------------------------------------------------------ table init helpers
// k*P by branch-free double-and-add (one-time table construction only)
template <class C>
VRF_HD PtE te_mul_slow(const PtE& base, const uint32_t k[8]) {
  PtE acc = te_identity();
  PtC bc = te_to_cached<C>(base);
#pragma unroll 1
  for (int i = 255; i >= 0; --i) {
    acc = te_dbl<C>(acc, true);
    uint32_t word = k[0];
#pragma unroll
    for (int j = 1; j < 8; ++j)
      if ((i >> 5) == j) word = k[j];
    bool bit = (word >> (i & 31)) & 1u;
    PtE sum = te_add_cached<C>(acc, bc, false);
    acc.X = fe_select(bit, sum.X, acc.X);
    acc.Y = fe_select(bit, sum.Y, acc.Y);
    acc.Z = fe_select(bit, sum.Z, acc.Z);
    acc.T = fe_select(bit, sum.T, acc.T);
  }
  return acc;
}
// ------------------------------------------------------------------------ key sets
// Context-resident fixed-base comb of a PUBLIC KEY (keyed verification: many proofs per key).  One lane builds
// row w of a key: base_w = 256^w * Y by 8w doublings, entries j * base_w, j = 1..255, by a chain of unified
// additions; the 255 projective entries are made affine with ONE inversion (Montgomery's trick; prefix
// products in `prefix`, 255 x 9 words owned by the lane).  ~1.5 M instructions per row instead of ~280 M for
// 255 independent double-and-add multiplications.
constexpr int COMB_ROWS = 32, COMB_COLS = 255;
constexpr size_t COMB_WORDS = (size_t)COMB_ROWS * COMB_COLS * PTA_WORDS;      // per base: 881,280 bytes
template <class C>
VRF_HD void comb_build_row(uint32_t* row /*[255][27]*/, uint32_t* prefix /*[255][9]*/, const FeN& x, const FeN& y, int w) {
  PtE base = te_from_affine(x, y);
#pragma unroll 1
  for (int i = 0; i < 8 * w; ++i) base = te_dbl<C>(base, i == 8 * w - 1);
  comb_chain<C>(row, prefix, base, te_to_cached<C>(base), COMB_COLS);
}

// ------------------------------------------------------------------------ challenge
// [ref src/lib.rs:14,16 `Suite::challenge` / utils::challenge_rfc_9381]  SURVEY.md A.4:
// c = int_be(SHA512(suite_id || 0x02 || enc(P1..P5) || ad || 0x00)[0..CHALLENGE_LEN]) mod r   (32 bytes for the
// Bandersnatch / JubJub / Baby-JubJub suites, 16 for Ed25519)
// The transcript hashes take `point_encode` of the TYPED point.  arkworks decodes a compressed point with x = 0 (y = 1 or
// y = q - 1) whatever its sign flag says and encodes it with the flag clear, so wire bytes are brought to that form
// before they are hashed; every other accepted encoding is already canonical.  (parity on crafted inputs.)
VRF_HD void enc_canonical(uint32_t w[8]) {
  const uint32_t top = w[7] & 0x7fffffffu;
  uint32_t d1 = (w[0] ^ 1u) | top, dm = (w[0] ^ (vrfk::Q32[0] - 1u)) | (top ^ vrfk::Q32[7]);
#pragma unroll
  for (int i = 1; i < 7; ++i) { d1 |= w[i]; dm |= w[i] ^ vrfk::Q32[i]; }
  if (d1 == 0 || dm == 0) w[7] = top;
}

// w >> (8 * nbytes), nbytes wave-uniform in 0..31
VRF_HD void u256_shr_bytes(uint32_t w[8], uint32_t nbytes) {
  const uint32_t ws = nbytes >> 2, bs = (nbytes & 3u) * 8u;
  uint32_t t[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if ((uint32_t)j == i + ws) lo = w[j];
      if ((uint32_t)j == i + ws + 1) hi = w[j];
    }
    t[i] = bs ? (lo >> bs) | (hi << (32u - bs)) : lo;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) w[i] = t[i];
}

template <class S>
VRF_HD void challenge5(uint32_t c_out[8], const uint32_t (&pts)[5][8], const uint8_t* ad,
                       uint32_t ad_len, const SuiteStr& ss) {
  Sha512 h;
  sha512_init(h);
  put_suite_id(h, ss);
  sha512_put_byte(h, 0x02);
  uint64_t w[20];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    uint32_t e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = pts[i][j];
    enc_canonical(e);
    sha512_words_le32x8(w + 4 * i, e);
  }
  sha512_put_words(h, w);
  sha512_put_bytes(h, ad, ad_len);
  sha512_put_byte(h, 0x00);
  sha512_final(h);
  uint32_t be[8];
  if (ss.flags & SS_CHALLENGE_LE) {
    // the same leading bytes read as a little-endian integer (RFC 9381's edwards suites): the digest as it sits in memory,
    // cut after CHALLENGE_LEN bytes
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t w = sha512_word_mem(h, j), have = ss.challenge_len > 4u * j ? ss.challenge_len - 4u * j : 0u;
      be[j] = have >= 4u ? w : have == 0u ? 0u : (w & ((1u << (8u * have)) - 1u));
    }
  } else {
    sha512_be256(be, h);
    if (ss.challenge_len != 32u) u256_shr_bytes(be, 32u - ss.challenge_len);    // `Suite::CHALLENGE_LEN` leading bytes only
  }
  fr_reduce256<S>(c_out, be);
}

// The c field of a proof.  With CHALLENGE_LEN = 32 it is a scalar, decoded mod r as upstream's scalar_decode does (s stays
// strict).  With a shorter challenge upstream's wire format has CHALLENGE_LEN bytes for it, so a 32-byte field holding more
// cannot be a proof string: it is compared as it stands and never equals a recomputed challenge (< 2^(8 len)) -- c + r no
// longer verifies (malleability of the 16-byte-challenge suites at this ABI).
template <class S>
VRF_HD void proof_challenge_decode(uint32_t out[8], const uint32_t c[8], const SuiteStr& ss) {
  if (ss.challenge_len < 32u) {
#pragma unroll
    for (int j = 0; j < 8; ++j) out[j] = c[j];
  } else {
    fr_reduce256<S>(out, c);
  }
}

// ------------------------------------------------------------------------ IETF verify
// [ref src/lib.rs:14 `ietf::Verifier::verify`]  U = s*G - c*Y, V = s*H - c*Gamma, accept iff
// challenge(Y, H, Gamma, U, V, ad) == c.  Three stages (three kernels):
//   decode : decompress Y, H, Gamma with one shared inversion; radix-16 window tables -> HBM
//   straus : U and V, one lane each (2 lanes per proof)                               -> HBM
//   finish : shared inversion of Z_U, Z_V; encode; SHA-512 challenge; compare.
constexpr int UV_WORDS = 3 * NL;          // X, Y, Z of one projective result

VRF_HD FeN fe_sel3(int p, const FeN& a, const FeN& b, const FeN& c) {
  return fe_select(p == 0, a, fe_select(p == 1, b, c));
}

constexpr int VERIFY_TABS = 6;   // Y, psi Y, H, psi H, Gamma, psi Gamma
// tabs: 6 * WIN_TABLE_WORDS words (GLV table pairs of Y, H, Gamma).  Returns validity.
template <class S>
VRF_HD bool verify_decode_item(const DevTables& T, const uint32_t pk[8], const uint32_t hh[8],
                               const uint32_t gamma[8], uint32_t* tabs, uint32_t check_mask = 0) {
  DecodeA a0 = decode_phase_a<S>(pk), a1 = decode_phase_a<S>(hh), a2 = decode_phase_a<S>(gamma);
  FeN dens[3] = {a0.den, a1.den, a2.den}, dinv[3];
  fe_batch_inv(dinv, dens);
  bool valid = true;
#pragma unroll 1
  for (int p = 0; p < 3; ++p) {
    DecodeA a;
    a.y = fe_sel3(p, a0.y, a1.y, a2.y);
    a.den = fe_sel3(p, a0.den, a1.den, a2.den);
    a.num = fe_select(p == 0, a0.num, fe_select(p == 1, a1.num, a2.num));
    a.flag = p == 0 ? a0.flag : (p == 1 ? a1.flag : a2.flag);
    a.ok = p == 0 ? a0.ok : (p == 1 ? a1.ok : a2.ok);
    FeN di = fe_sel3(p, dinv[0], dinv[1], dinv[2]);
    Fe<1, 4> x;
    valid = decode_phase_b<S>(x, a, di, T.sq) && valid;
    build_glv_tables<S>(tabs + p * 2 * WIN_TABLE_WORDS, x, a.y);
    if ((check_mask >> p) & 1u) valid = in_prime_subgroup<S>(fe_mul(x, fe_one()), a.y, T.sq) && valid;
  }
  return valid;
}

// ---- several proofs per lane: the inversion of decode and of finish is shared by K proofs ----
// (Montgomery's trick across 3K denominators / 2K Z coordinates; prefix products are parked in the
// lane's own slice of the HBM workspace, so the loops are rolled and register use stays low.)
constexpr int VERIFY_K = 8;                 // max proofs per lane in decode and finish (runtime K <= this)
constexpr int DEC_SLOT = 4 * NL;            // per point: y | num | den | prefix  (36 words)

// points: base pointers of the pk / H / Gamma arrays; items [first, first + K) ∩ [0, n).
// tabs_base / scratch_base / flags: workspace arrays indexed by item.
// NP = 3: pk, H, Gamma (table slots 0, 1, 2); NP = 2 (keyed verification: the key's tables are context
// resident): H, Gamma (pk is not read, slots 1, 2); NP = 2 with SKIP_H (verification from alpha: H came out of
// hash-to-curve in this call and its tables are already in slot 1): pk, Gamma (slots 0, 2).
template <class S, int NP = 3, bool SKIP_H = false>
VRF_HD void verify_decode_multi(int K, const DevTables& T, size_t first, size_t n, const uint8_t* pk,
                                const uint8_t* hh, const uint8_t* gamma, uint32_t* tabs_base,
                                uint32_t* scratch_base, uint8_t* flags, uint32_t check_mask = 0) {
  // scratch: K * 108 words owned by this lane = 3K point slots of 36 words
  uint32_t* scr = scratch_base + first * (3 * DEC_SLOT);
  FeN run = fe_one();
  uint64_t bits = 0;                         // per point: bit 2j = sign flag, bit 2j+1 = ok
#pragma unroll 1
  for (int j = 0; j < NP * K; ++j) {
    const size_t item = first + j / NP;
    const int p = SKIP_H ? (j % NP) * 2 : j % NP + (3 - NP);
    if (item < n) {
      const uint8_t* src = p == 0 ? pk : (p == 1 ? hh : gamma);
      uint32_t enc[8];
      const uint32_t* w = reinterpret_cast<const uint32_t*>(src + item * 32);
#pragma unroll
      for (int k = 0; k < 8; ++k) enc[k] = w[k];
      DecodeA a = decode_phase_a<S>(enc);
      uint32_t* slot = scr + j * DEC_SLOT;
      fe_store(slot, a.y);
      fe_store(slot + NL, a.num);
      fe_store(slot + 2 * NL, a.den);
      fe_store(slot + 3 * NL, run);
      run = fe_mul(run, a.den);
      bits |= (uint64_t)((a.flag ? 1u : 0u) | (a.ok ? 2u : 0u)) << (2 * j);
    }
  }
  FeN inv = fe_inv(run);
  uint32_t valid_mask = 0xffffffffu;
#pragma unroll 1
  for (int j = NP * K - 1; j >= 0; --j) {
    const size_t item = first + j / NP;
    const int p = SKIP_H ? (j % NP) * 2 : j % NP + (3 - NP);
    if (item < n) {
      const uint32_t* slot = scr + j * DEC_SLOT;
      DecodeA a;
      a.y = fe_load<1, 2>(slot);
      a.num = fe_load<1, 6>(slot + NL);
      a.den = fe_load<1, 2>(slot + 2 * NL);
      FeN prefix = fe_load<1, 2>(slot + 3 * NL);
      a.flag = (bits >> (2 * j)) & 1;
      a.ok = (bits >> (2 * j + 1)) & 1;
      FeN di = fe_mul(inv, prefix);
      inv = fe_mul(inv, a.den);
      Fe<1, 4> x;
      bool ok = decode_phase_b<S>(x, a, di, T.sq);
      uint32_t* tab = tabs_base + item * (VERIFY_TABS * WIN_TABLE_WORDS) + p * 2 * WIN_TABLE_WORDS;
      build_glv_tables<S>(tab, x, a.y);
      if ((check_mask >> p) & 1u) ok = in_prime_subgroup<S>(fe_mul(x, fe_one()), a.y, T.sq) && ok;    // bit p: pk, H, Gamma
      if (!ok) valid_mask &= ~(1u << (j / NP));
    }
  }
#pragma unroll 1
  for (int k = 0; k < K; ++k)
    if (first + k < n) flags[first + k] = (valid_mask >> k) & 1;
}

// finish for K proofs per lane: one inversion for 2K Z coordinates.  uv_base: [item][2][27] words inside
// the pts region; the 54 spare words of each item's pts slot hold the prefix products.
template <class S>
VRF_HD void verify_finish_multi(int K, size_t first, size_t n, uint32_t* pts_base, int pts_stride,
                                const uint8_t* pk, const uint8_t* hh, const uint8_t* gamma,
                                const uint32_t* enc_aux, int aux_stride, const uint8_t* c_arr,
                                const uint8_t* s_arr, const BytesViewLite& ad, const uint8_t* flags,
                                uint8_t* status, const SuiteStr& ss, const uint32_t* key_index = nullptr,
                                size_t n_keys = 0) {
  FeN run = fe_one();
#pragma unroll 1
  for (int j = 0; j < 2 * K; ++j) {
    const size_t item = first + j / 2;
    if (item < n) {
      uint32_t* slot = pts_base + item * pts_stride;
      FeP z = fe_load<1, 5>(slot + (j & 1) * UV_WORDS + 2 * NL);
      fe_store(slot + 2 * UV_WORDS + (j & 1) * NL, run);
      run = fe_mul(run, z);
    }
  }
  FeN inv = fe_inv(run);
  uint32_t encv[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) encv[k] = 0;
#pragma unroll 1
  for (int j = 2 * K - 1; j >= 0; --j) {
    const size_t item = first + j / 2;
    if (item < n) {
      const uint32_t* slot = pts_base + item * pts_stride;
      const uint32_t* uv = slot + (j & 1) * UV_WORDS;
      FeN prefix = fe_load<1, 2>(slot + 2 * UV_WORDS + (j & 1) * NL);
      FeP z = fe_load<1, 5>(uv + 2 * NL);
      FeN zi = fe_mul(inv, prefix);
      inv = fe_mul(inv, z);
      uint32_t e[8];
      te_encode_affine(e, fe_mul(fe_load<1, 5>(uv), zi), fe_mul(fe_load<1, 5>(uv + NL), zi), ss.flags);
      if (j & 1) {
#pragma unroll
        for (int k = 0; k < 8; ++k) encv[k] = e[k];          // V first (reverse order), U completes the item
      } else {
        uint32_t pts[5][8], c[8], sc[8];
        if (enc_aux) {
          const uint32_t* aux = enc_aux + item * aux_stride;
#pragma unroll
          for (int k = 0; k < 8; ++k) { pts[0][k] = aux[k]; pts[1][k] = aux[8 + k]; pts[2][k] = aux[16 + k]; }
        } else {
          // keyed verification: pk holds the key set's encodings, key_index picks the item's key
          size_t pk_row = item;
          if (key_index) pk_row = key_index[item] < n_keys ? key_index[item] : 0;
          const uint32_t* w0 = reinterpret_cast<const uint32_t*>(pk + pk_row * 32);
          const uint32_t* w1 = reinterpret_cast<const uint32_t*>(hh + item * 32);
          const uint32_t* w2 = reinterpret_cast<const uint32_t*>(gamma + item * 32);
#pragma unroll
          for (int k = 0; k < 8; ++k) { pts[0][k] = w0[k]; pts[1][k] = w1[k]; pts[2][k] = w2[k]; }
        }
        const uint32_t* wc = reinterpret_cast<const uint32_t*>(c_arr + item * 32);
        const uint32_t* ws = reinterpret_cast<const uint32_t*>(s_arr + item * 32);
#pragma unroll
        for (int k = 0; k < 8; ++k) { pts[3][k] = e[k]; pts[4][k] = encv[k]; c[k] = wc[k]; sc[k] = ws[k]; }
        const uint8_t* adp; uint32_t adl;
        bytes_lite_get(ad, item, adp, adl);
        uint32_t c2[8], cr[8];
        challenge5<S>(c2, pts, adp, adl, ss);
        proof_challenge_decode<S>(cr, c, ss);
        uint32_t diff = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) diff |= c2[k] ^ cr[k];
        bool valid = flags[item] != 0 && fr_is_canonical<S>(sc);
        status[item] = (uint8_t)(!valid ? ST_INVALID_DATA : (diff == 0 ? ST_OK : ST_VERIFICATION_FAILURE));
      }
    }
  }
}

// Same stage for callers that hold affine points in memory (arkworks `Affine { x, y }`): x || y as
// 32-byte little-endian canonical integers, no square roots.  Validity = coordinates < q and the
// point is on the curve.
template <class S>
VRF_HD bool verify_decode_affine_item(uint32_t enc_out[3][8], const uint32_t (&xy)[3][16], uint32_t* tabs,
                                      const SqrtTables& T, uint32_t check_mask = 0, bool mont256 = false) {
  bool valid = true;
#pragma unroll 1
  for (int p = 0; p < 3; ++p) {
    uint32_t xin[8], yin[8], xw[8], yw[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      xin[j] = p == 0 ? xy[0][j] : p == 1 ? xy[1][j] : xy[2][j];
      yin[j] = p == 0 ? xy[0][8 + j] : p == 1 ? xy[1][8 + j] : xy[2][8 + j];
    }
    valid = valid && !u256_ge(xin, vrfk::Q32) && !u256_ge(yin, vrfk::Q32);
    FeN x = fe_from_abi(xw, xin, mont256), y = fe_from_abi(yw, yin, mont256);     // xw, yw: canonical words
    // a x^2 + y^2 = 1 + d x^2 y^2   <=>   a x^2 + y^2 - 1 = d (x y)^2
    FeN x2 = fe_sqr(x), y2 = fe_sqr(y), xyv = fe_mul(x, y);
    auto lhs = te_curve_lhs<S>(x2, y2);
    valid = fe_eq(lhs, fe_mul(fe_mul(fe_sqr(xyv), S::d()), fe_one())) && valid;
    build_glv_tables<S>(tabs + p * 2 * WIN_TABLE_WORDS, x, y);
    if ((check_mask >> p) & 1u) valid = in_prime_subgroup<S>(x, y, T) && valid;
    uint32_t e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = yw[j];
    if (te_x_sign(xw, T.str.flags)) e[7] |= 0x80000000u;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (p == 0) enc_out[0][j] = e[j];
      if (p == 1) enc_out[1][j] = e[j];
      if (p == 2) enc_out[2][j] = e[j];
    }
  }
  return valid;
}

// HALF 0: U = s*G - c*Y = comb(G, s) - (c1*Y + c2*psi Y)   (2-table GLV Straus + 16-bit signed fixed-base table)
// HALF 1: V = s*H - c*Gamma                                 (4-table GLV Straus over tabs[2..5])
// GLV: k = k1 + k2*lambda with 128-bit halves; the c terms are subtracted.  The two halves run as
// separate launches so that every wave executes one shape.
template <class S, int HALF>
VRF_HD void verify_straus_item(uint32_t* out_uv, const DevTables& T, const uint32_t* tabs,
                               const uint32_t c[8], const uint32_t s[8]) {
  if constexpr (!S::HAS_GLV) {
    // plain 253-bit Straus: U = comb(G, s) - c*Y ; V = s*H - c*Gamma (tables at slots 0, 2, 4)
    uint32_t recs[8], recc[8];
    scalar_recode_signed4(recs, s);
    scalar_recode_signed4(recc, c);
    PtE r;
    if (HALF == 0) {
      r = win_mul<S>(tabs, recc, true, challenge_top_window(T.sq.str));
      r = gcomb_add<S>(r, T.g_comb, s);
    } else {
      r = straus2<S>(tabs + 2 * WIN_TABLE_WORDS, recs, tabs + 4 * WIN_TABLE_WORDS, recc, true, challenge_top_window(T.sq.str));
    }
    fe_store(out_uv, r.X);
    fe_store(out_uv + NL, r.Y);
    fe_store(out_uv + 2 * NL, r.Z);
    return;
  } else {
  Straus4 q;
  GlvHalf h[4];
  glv_decompose_bs(h[0], h[1], c);
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    scalar_recode_signed4_128(q.rec[t], h[t].mag);
    q.neg[t] = !h[t].neg;
  }
  PtE r;
  if (HALF == 0) {
    q.tab[0] = tabs; q.tab[1] = tabs + WIN_TABLE_WORDS;
    q.tab[2] = tabs; q.tab[3] = tabs;
#pragma unroll
    for (int i = 0; i < 4; ++i) { q.rec[2][i] = 0; q.rec[3][i] = 0; }
    q.neg[2] = false; q.neg[3] = false;
    r = straus4<S, 2>(q);
    r = gcomb_add<S>(r, T.g_comb, s);
  } else {
    glv_decompose_bs(h[2], h[3], s);
#pragma unroll
    for (int t = 2; t < 4; ++t) {
      scalar_recode_signed4_128(q.rec[t], h[t].mag);
      q.neg[t] = h[t].neg;
    }
    q.tab[0] = tabs + 4 * WIN_TABLE_WORDS; q.tab[1] = tabs + 5 * WIN_TABLE_WORDS;
    q.tab[2] = tabs + 2 * WIN_TABLE_WORDS; q.tab[3] = tabs + 3 * WIN_TABLE_WORDS;
    r = straus4<S, 4>(q);
  }
  fe_store(out_uv, r.X);
  fe_store(out_uv + NL, r.Y);
  fe_store(out_uv + 2 * NL, r.Z);
  }
}

template <class S>
VRF_HD uint32_t verify_finish_item(const uint32_t* uv, const uint32_t pk[8], const uint32_t hh[8],
                                   const uint32_t gamma[8], const uint32_t c[8],
                                   const uint32_t s[8], bool valid, const uint8_t* ad,
                                   uint32_t ad_len, const SuiteStr& ss) {
  valid = valid && fr_is_canonical<S>(s);
  uint32_t cr[8];
  proof_challenge_decode<S>(cr, c, ss);
  FeP zin[2] = {fe_load<1, 5>(uv + 2 * NL), fe_load<1, 5>(uv + UV_WORDS + 2 * NL)};
  FeN zi[2];
  fe_batch_inv(zi, zin);
  uint32_t pts[5][8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { pts[0][i] = pk[i]; pts[1][i] = hh[i]; pts[2][i] = gamma[i]; }
  te_encode_affine(pts[3], fe_mul(fe_load<1, 5>(uv), zi[0]), fe_mul(fe_load<1, 5>(uv + NL), zi[0]), ss.flags);
  te_encode_affine(pts[4], fe_mul(fe_load<1, 5>(uv + UV_WORDS), zi[1]),
                   fe_mul(fe_load<1, 5>(uv + UV_WORDS + NL), zi[1]), ss.flags);
  uint32_t c2[8];
  challenge5<S>(c2, pts, ad, ad_len, ss);
  uint32_t diff = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) diff |= c2[i] ^ cr[i];
  if (!valid) return ST_INVALID_DATA;
  return diff == 0 ? ST_OK : ST_VERIFICATION_FAILURE;
}

// ------------------------------------------------------------------------ hash to curve
// [ref src/lib.rs:15-16 `Input::new` -> utils::hash_to_curve_ell2_rfc_9380]  SURVEY.md A.3
// expand_message_xmd(SHA-512) with arkworks' 48-byte Z_pad, two field elements, Elligator 2
// on the Montgomery model, map to twisted Edwards, add, clear cofactor.
VRF_HD void put_dst_prime(Sha512& h, const SuiteStr& ss) { sha512_put_packed(h, ss.dst_prime_w, ss.dst_prime_len); }

// expand_message_xmd(SHA-512) to 96 bytes with arkworks' 48-byte Z_pad: uniform = b1 || b2[0..32];
// b1, b2 as the eight big-endian 64-bit words of each digest
template <class S>
VRF_HD void expand_message_xmd96(uint64_t (&hb)[2][8], const uint8_t* msg, uint32_t msg_len, const SuiteStr& ss) {
  Sha512 b0;
  sha512_init(b0);
#pragma unroll
  for (int i = 0; i < 6; ++i) sha512_put(b0, 0, 8);      // Z_pad: 48 zero bytes
  sha512_put_bytes(b0, msg, msg_len);
  sha512_put_byte(b0, 0x00);
  sha512_put_byte(b0, 0x60);                             // len_in_bytes = 96
  sha512_put_byte(b0, 0x00);
  put_dst_prime(b0, ss);
  sha512_final(b0);
  uint64_t h0[8], h1[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) h0[i] = b0.h[i];
  // b1 = H(b0 || 0x01 || DST'), b2 = H((b0 ^ b1) || 0x02 || DST'): one hashing site, two trips
#pragma unroll
  for (int i = 0; i < 8; ++i) h1[i] = 0;
#pragma unroll 1
  for (int t = 0; t < 2; ++t) {
    Sha512 b;
    sha512_init(b);
#pragma unroll
    for (int i = 0; i < 8; ++i) sha512_put(b, h0[i] ^ h1[i], 8);
    sha512_put_byte(b, (uint8_t)(t + 1));
    put_dst_prime(b, ss);
    sha512_final(b);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (t == 0) { h1[i] = b.h[i]; hb[0][i] = b.h[i]; } else { hb[1][i] = b.h[i]; }
    }
  }
}

template <class S>
VRF_HD void hash_to_field2(Fe<1, 4>& u0, Fe<1, 4>& u1, const uint8_t* msg, uint32_t msg_len, const SuiteStr& ss) {
  uint64_t hb[2][8];
  expand_message_xmd96<S>(hb, msg, msg_len, ss);
  // uniform = b1 (64 B) || b2[0..32];  u0 = BE(b1.h[0..6]);  u1 = BE(b1.h[6..8] || b2.h[0..4])
  uint32_t w0[16], w1[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { w0[i] = 0; w1[i] = 0; }
#pragma unroll
  for (int j = 0; j < 6; ++j) {          // 64-bit BE words, most significant first
    uint64_t a = hb[0][j];
    uint64_t b = (j < 2) ? hb[0][6 + j] : hb[1][j - 2];
    w0[2 * (5 - j)] = (uint32_t)a; w0[2 * (5 - j) + 1] = (uint32_t)(a >> 32);
    w1[2 * (5 - j)] = (uint32_t)b; w1[2 * (5 - j) + 1] = (uint32_t)(b >> 32);
  }
  u0 = fe_from_u512(w0);
  u1 = fe_from_u512(w1);
}

VRF_HD bool fe_parity(const FeN& a) {       // canonical integer is odd
  uint32_t w[8];
  fe_to_u256(w, a);
  return w[0] & 1;
}

#if VRF_FIELD == 0
// Elligator 2 for one u, given D = 1 + Z*u^2 (already replaced by 1 if zero) and 1/D.
template <class S>
VRF_HD PtE ell2_map(const Fe<1, 4>& u, const FeN& Dinv, const SqrtTables& T) {
  const FeN JK = fe_const(vrfk::BS_ELL2_JK_M), K2I = fe_const(vrfk::BS_ELL2_K2I_M);
  const FeN K = fe_const(vrfk::BS_ELL2_K_M);
  FeN x1 = fe_mul(fe_const(vrfk::BS_ELL2_NJK_M), Dinv);              // -(J/K) / D
  FeN t = fe_mul(fe_add(x1, JK), x1);                                // gx1 = ((x1+J/K)x1 + 1/K^2)x1
  FeN gx1 = fe_mul(fe_add(t, K2I), x1);
  FeN root;
  bool sq = fe_sqrt_or_zsqrt(root, gx1, T);
  uint32_t rw[8];
  fe_to_u256(rw, root);
  uint32_t nz = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) nz |= rw[i];
  sq = sq || (nz == 0);
  Fe<1, 8> x2 = fe_norm(fe_neg(fe_norm(fe_add(x1, JK))));           // -x1 - J/K
  FeN y2 = fe_mul(root, u);                                          // sqrt(gx2) = u * sqrt(Z*gx1)
  Fe<1, 8> x1w = x1;
  Fe<1, 8> x = fe_select(sq, x1w, x2);
  FeN y = fe_select(sq, root, y2);
  bool odd = fe_parity(y);
  auto ys = fe_norm(fe_cneg(odd != sq, y));    // square: want y odd ; non-square: want y even
  FeN s = fe_mul(x, K), tt = fe_mul(ys, K);
  auto sp1 = fe_add(s, fe_one());              // (2,4)
  auto sm1 = fe_sub(s, fe_one());              // (3,6)
  PtE p;
  p.X = fe_mul(s, sp1);
  p.Y = fe_mul(sm1, tt);
  p.Z = fe_mul(tt, sp1);
  p.T = fe_mul(s, sm1);
  bool exc = fe_is_zero(p.Z);
  PtE id = te_identity();
  p.X = fe_select(exc, id.X, p.X);
  p.Y = fe_select(exc, id.Y, p.Y);
  p.Z = fe_select(exc, id.Z, p.Z);
  p.T = fe_select(exc, id.T, p.T);
  return p;
}

template <class S>
VRF_HD PtE hash_to_curve_ell2(const uint8_t* msg, uint32_t msg_len, const SqrtTables& T) {
  Fe<1, 4> u[2];
  hash_to_field2<S>(u[0], u[1], msg, msg_len, T.str);
  FeN D[2], Di[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    FeN d = fe_canon(fe_add(fe_mul5(fe_sqr(u[i])), fe_one()));       // 1 + Z*u^2, Z = 5
    bool dz = true;
#pragma unroll
    for (int k = 0; k < NL; ++k) dz = dz && (d.v[k] == 0);
    D[i] = fe_select(dz, fe_one(), d);
  }
  fe_batch_inv(Di, D);
  PtE q0 = te_identity();
  PtE acc = te_identity();
#pragma unroll 1
  for (int i = 0; i < 2; ++i) {
    Fe<1, 4> ui = fe_select(i == 0, u[0], u[1]);
    FeN di = fe_select(i == 0, Di[0], Di[1]);
    PtE q = ell2_map<S>(ui, di, T);
    if (i == 0) q0 = q; else acc = te_add<S>(q0, q);
  }
#pragma unroll 1
  for (int i = 0; i < S::COFACTOR_LOG2; ++i) acc = te_dbl<S>(acc, true);
  return acc;
}

#else
template <class S>
VRF_HD PtE hash_to_curve_ell2(const uint8_t* msg, uint32_t msg_len, const SqrtTables& T);   // Elligator suites: field 0 only
#endif

// [ref src/lib.rs:14 `utils::hash_to_curve_tai_rfc_9381`]  SURVEY.md A.6 (unpinned): for ctr = 0..255:
// h = SHA512(suite_id || 0x01 || data || ctr || 0x00); decode h[0..32] as a point; clear the
// cofactor; first non-identity result wins.  Lanes iterate until they succeed (about two trips).
// candidate encoding of attempt `ctr`: the first 32 bytes of SHA512(suite_id || 0x01 || data || ctr || 0x00)
template <class S>
VRF_HD void tai_candidate(uint32_t enc[8], const uint8_t* msg, uint32_t msg_len, uint32_t ctr, const SuiteStr& ss) {
  Sha512 h;
  sha512_init(h);
  put_suite_id(h, ss);
  sha512_put_byte(h, 0x01);
  sha512_put_bytes(h, msg, msg_len);
  sha512_put_byte(h, (uint8_t)ctr);
  sha512_put_byte(h, 0x00);
  sha512_final(h);
#pragma unroll
  for (int j = 0; j < 8; ++j) enc[j] = sha512_word_mem(h, j);
}
// Does attempt `ctr` decode to a curve point?  (y < q, denominator non-zero, (y^2-1)(d y^2-a) a square: the
// same verdict as decode_phase_b on (y^2-1)/(d y^2-a), without the inversion and without a square root.)  k_tai_find uses it to hand
// hash_to_curve_tai a starting counter; a candidate of small order passes here and is rejected there.
template <class S>
VRF_HD bool tai_attempt_decodes(const uint8_t* msg, uint32_t msg_len, uint32_t ctr, const SqrtTables& T) {
  uint32_t enc[8];
  tai_candidate<S>(enc, msg, msg_len, ctr, T.str);
  DecodeA a = decode_phase_a<S>(enc);
  return a.ok && fe_is_square_or_zero(fe_mul(a.num, a.den), T);       // Jacobi symbol: no exponentiation
}
// The same verdict in two steps, for k_tai_find: the cheap half (hash, y < q, denominator non-zero) leaves the value whose
// quadratic character decides; the expensive half is fe_is_square_or_zero of it.
template <class S>
VRF_HD bool tai_attempt_candidate(FeN& w, const uint8_t* msg, uint32_t msg_len, uint32_t ctr, const SqrtTables& T) {
  uint32_t enc[8];
  tai_candidate<S>(enc, msg, msg_len, ctr, T.str);
  DecodeA a = decode_phase_a<S>(enc);
  w = fe_mul(a.num, a.den);
  return a.ok;
}

// start: first counter to try (0, or the hint of k_tai_find: every smaller counter is known not to decode)
template <class S>
VRF_HD PtE hash_to_curve_tai(const uint8_t* msg, uint32_t msg_len, const SqrtTables& T, uint32_t start = 0) {
  PtE res = te_identity();
  bool done = false;
#pragma unroll 1
  for (uint32_t ctr = start; ctr < 256 && !done; ++ctr) {
    uint32_t enc[8];
    tai_candidate<S>(enc, msg, msg_len, ctr, T.str);
    DecodeA a = decode_phase_a<S>(enc);
    FeN di = fe_inv(a.den);
    Fe<1, 4> x;
    bool ok = decode_phase_b<S>(x, a, di, T);
    PtE p = te_from_affine(x, a.y);
#pragma unroll 1
    for (int i = 0; i < S::COFACTOR_LOG2; ++i) p = te_dbl<S>(p, true);
    bool is_id = fe_is_zero(p.X) && fe_eq(p.Y, p.Z);
    if (ok && !is_id) {
      res = p;
      done = true;
    }
  }
  return res;
}

// [ref src/lib.rs:15-16 `Input::new` / `Suite::data_to_point`]
template <class S>
VRF_HD PtE data_to_point(const uint8_t* msg, uint32_t msg_len, const SqrtTables& T, uint32_t tai_start = 0) {
  if constexpr (S::H2C_ELL2) return hash_to_curve_ell2<S>(msg, msg_len, T);
  else return hash_to_curve_tai<S>(msg, msg_len, T, tai_start);
}

// ------------------------------------------------------------------------ nonce
// [ref src/lib.rs:14,16 `Suite::nonce` / utils::nonce_rfc_8032]  SURVEY.md A.4:
// k = int_le(SHA512(SHA512(sk_le32)[32..64] || enc(H))) mod r
template <class S>
VRF_HD void nonce_rfc8032(uint32_t k[8], const uint32_t sk[8], const uint32_t h_enc[8]) {
  Sha512 a;
  sha512_init(a);
  sha512_put_le32x8(a, sk);
  sha512_final(a);
  Sha512 b;
  sha512_init(b);
#pragma unroll
  for (int i = 4; i < 8; ++i) sha512_put(b, a.h[i], 8);
  sha512_put_le32x8(b, h_enc);
  sha512_final(b);
  uint32_t le[16];
  sha512_le512(le, b);
  fr_reduce512<S>(k, le);
}

#if VRF_FIELD == 0
// Elligator-2 denominator D = 1 + Z*u^2 (replaced by 1 if zero)
VRF_HD FeN ell2_den(const Fe<1, 4>& u) {
  FeN d = fe_canon(fe_add(fe_mul5(fe_sqr(u)), fe_one()));
  bool dz = true;
#pragma unroll
  for (int k = 0; k < NL; ++k) dz = dz && (d.v[k] == 0);
  return fe_select(dz, fe_one(), d);
}
#endif

// affine coordinates of a projective point (one inversion; CT: fixed shape, for points derived from secrets)
template <bool CT = false>
VRF_HD void te_to_affine(FeN& x, FeN& y, const PtE& p) {
  FeN zi = fe_inv<CT>(p.Z);
  x = fe_mul(p.X, zi);
  y = fe_mul(p.Y, zi);
}

// ------------------------------------------------------------------------ IETF prove
// [ref src/lib.rs:14 `ietf::Prover::prove` + src/lib.rs:16 `Secret::{public, output}`]
// From (sk, msg): H = hash_to_curve(msg), pk = sk*G, Gamma = sk*H, k = nonce(sk, H),
// c = challenge(pk, H, Gamma, kG, kH), s = k + c*sk.  Three stages:
//   prepare: H (hash-to-curve or decode), enc(H), nonce k, window table of H     -> HBM
//   mul    : lane 0: (sk*H, sk*G) ; lane 1: (k*H, k*G)                           -> HBM
//   finish : shared inversion of the four Z, encodings, challenge, s.
constexpr int PROVE_PTS_WORDS = 4 * UV_WORDS;   // [half][win|comb][X,Y,Z]

// ---- prove: the two variable-base products of a proof, sk*H and k*H, share their base ----
// Without an endomorphism (JubJub) a 253-bit product costs 252 accumulator doublings.  Doubling the BASE instead --
// tables {H, 2^64 H, 2^128 H, 2^192 H}, 192 base doublings once per proof -- leaves each product 60 accumulator
// doublings + 64 additions (four streams of 16 signed radix-16 digits, one per table): 312 doublings + 149
// additions per proof instead of 504 + 128 (k_prove_mul 34.5 -> 18.4 ms at 2^20).  Any digit split gives the same
// group element: results stay bit-exact.
// With GLV (Bandersnatch) the same trick -- {H, 2^64 H, psi H, psi 2^64 H}, 64-bit quarters -- was built and
// measured, and withdrawn: it saves 128 accumulator doublings per proof in a kernel that runs at 93 % of the VALU
// ceiling and pays 64 base doublings + two more tables in a stage that does not (its table stores are uncoalesced
// 144-byte records: 0.76 of the ceiling at best): 34.5 ms per 2^20 proofs against 32.3 ms.  Bandersnatch keeps
// the table pair {H, psi H} and two streams of 32 digits.
template <class S> struct ProveLayout {
  static constexpr int TABS = S::HAS_GLV ? 2 : 4;        // tables per proof
  static constexpr int DIGITS = S::HAS_GLV ? 32 : 16;    // signed radix-16 digits per stream
  static constexpr int TAB_WORDS = TABS * WIN_TABLE_WORDS;   // the proof's stride in the table region
};
template <class S>
VRF_HD void build_prove_tables(uint32_t* tab, const FeP& x, const FeP& y) {
  if constexpr (S::HAS_GLV) {
    build_glv_tables<S>(tab, x, y);                      // {H, psi H}
  } else {
    PtE p = te_from_affine(x, y);
#pragma unroll 1
    for (int j = 0; j < ProveLayout<S>::TABS; ++j) {
      if (j) {
#pragma unroll 1
        for (int i = 0; i < 64; ++i) p = te_dbl<S>(p, i == 63);
      }
      build_win_table_from<S>(tab + j * WIN_TABLE_WORDS, p);
    }
  }
}
// scalar * H from the tables of build_prove_tables: stream t reads digits DIGITS * t + w of a 64-digit signed radix-16
// string (GLV: k1's 32 digits then k2's 32 digits; else the 253-bit scalar) and adds from table t
template <class S, bool CT = false>
VRF_HD PtE var_base_mul(const uint32_t* tab, const uint32_t scalar[8]);
template <class S, bool CT = false>
VRF_HD PtE prove_var_mul(const uint32_t* tab, const uint32_t scalar[8]) {
  if constexpr (S::HAS_GLV) return var_base_mul<S, CT>(tab, scalar);     // two-table GLV Straus over {H, psi H}
  constexpr int NT = ProveLayout<S>::TABS, ND = ProveLayout<S>::DIGITS;
  uint32_t rec[8];
  const bool neg_lo = false, neg_hi = false;
  scalar_recode_signed4(rec, scalar);
  PtE acc = te_identity();
#pragma unroll 1
  for (int w = ND - 1; w >= 0; --w) {
    if (w != ND - 1) {
#pragma unroll 1
      for (int j = 0; j < 4; ++j) acc = te_dbl<S>(acc, j == 3);
    }
#pragma unroll 1
    for (int t = 0; t < NT; ++t) {
      const int d = scalar_digit4(rec, ND * t + w);
      const bool neg = 2 * t < NT ? neg_lo : neg_hi;
      acc = te_add_cached<S>(acc, win_lookup_t<CT>(tab + t * WIN_TABLE_WORDS, d), (d < 0) != neg, t != NT - 1 || w == 0);
    }
  }
  return acc;
}

// returns validity (always true for the hash-to-curve path; decode may fail)
// [ref src/lib.rs:14 `pedersen::PedersenSuite::blinding`]  SURVEY.md A.5:
// b = int_be(SHA512(suite_id || 0xCC || sk_le32 || enc(H) || ad || 0x00)) mod r  (all 64 bytes)
template <class S>
VRF_HD void pedersen_blinding(uint32_t b[8], const uint32_t sk[8], const uint32_t h_enc[8],
                              const uint8_t* ad, uint32_t ad_len, const SuiteStr& ss) {
  Sha512 h;
  sha512_init(h);
  put_suite_id(h, ss);
  sha512_put_byte(h, 0xCC);
  uint64_t w[8];
  sha512_words_le32x8(w, sk);
  sha512_words_le32x8(w + 4, h_enc);
  sha512_put_words(h, w);
  sha512_put_bytes(h, ad, ad_len);
  sha512_put_byte(h, 0x00);
  sha512_final(h);
  uint32_t be[16];
  sha512_be512(be, h);
  fr_reduce512<S>(b, be);
}

template <class S>
VRF_HD bool prove_prepare_item(uint32_t h_enc[8], uint32_t k[8], uint32_t* tab, const DevTables& T,
                               const uint32_t sk[8], const uint8_t* msg, uint32_t msg_len,
                               const uint32_t* h_given, uint32_t tai_start = 0, uint32_t check_mask = 0) {
  FeN x, y;
  bool valid = fr_is_canonical<S>(sk);
  if (h_given) {
    DecodeA a = decode_phase_a<S>(h_given);
    FeN di = fe_inv(a.den);
    Fe<1, 4> xx;
    valid = decode_phase_b<S>(xx, a, di, T.sq) && valid;
    x = fe_mul(xx, fe_one());
    y = a.y;
  } else {
    PtE hp = data_to_point<S>(msg, msg_len, T.sq, tai_start);
    te_to_affine(x, y, hp);
  }
  te_encode_affine(h_enc, x, y, T.sq.str.flags);
  nonce_rfc8032<S>(k, sk, h_enc);
  build_prove_tables<S>(tab, x, y);        // ProveLayout<S>::TAB_WORDS words
  if (h_given && (check_mask & CHK_INPUT)) valid = in_prime_subgroup<S>(x, y, T.sq) && valid;   // a given H is wire data
  return valid;
}

// scalar2 != nullptr (Pedersen): the fixed-base part is scalar*G + scalar2*B
// prepare for K proofs per lane (hash-to-curve path): two shared inversions per lane instead of 2K
// (the 2K Elligator denominators, then the K projective Z of H).  scratch: the lane's K pts slots
// (108 words per item): u0 | u1 | D0 | D1 | pre0 | pre1 | X | Y | Z | preZ.
constexpr int PROVE_K = 8;
#if VRF_FIELD == 0
template <class S>
VRF_HD void prove_prepare_multi(int K, const DevTables& T, size_t first, size_t n, const uint8_t* sk_arr,
                                const BytesViewLite& msgs, uint32_t* tabs_base, uint32_t* pts_base,
                                uint32_t* aux_base, int aux_stride, uint8_t* flags) {
  static_assert(S::H2C_ELL2, "multi-proof prepare is the Elligator path");
  FeN run = fe_one();
#pragma unroll 1
  for (int j = 0; j < K; ++j) {
    const size_t item = first + j;
    if (item < n) {
      const uint8_t* m; uint32_t len;
      bytes_lite_get(msgs, item, m, len);
      Fe<1, 4> u0, u1;
      hash_to_field2<S>(u0, u1, m, len, T.sq.str);
      uint32_t* slot = pts_base + item * PROVE_PTS_WORDS;
#pragma unroll 1
      for (int t = 0; t < 2; ++t) {
        Fe<1, 4> u = fe_select(t == 0, u0, u1);
        FeN D = ell2_den(u);
        fe_store(slot + t * NL, u);
        fe_store(slot + (2 + t) * NL, D);
        fe_store(slot + (4 + t) * NL, run);
        run = fe_mul(run, D);
      }
    }
  }
  FeN inv = fe_inv(run);
  FeN runz = fe_one();
#pragma unroll 1
  for (int j = K - 1; j >= 0; --j) {
    const size_t item = first + j;
    if (item < n) {
      uint32_t* slot = pts_base + item * PROVE_PTS_WORDS;
      PtE q1 = te_identity(), acc = te_identity();
#pragma unroll 1
      for (int t = 1; t >= 0; --t) {
        Fe<1, 4> u = fe_load<1, 4>(slot + t * NL);
        FeN D = fe_load<1, 2>(slot + (2 + t) * NL);
        FeN pre = fe_load<1, 2>(slot + (4 + t) * NL);
        FeN di = fe_mul(inv, pre);
        inv = fe_mul(inv, D);
        PtE q = ell2_map<S>(u, di, T.sq);
        if (t == 1) q1 = q; else acc = te_add<S>(q, q1);
      }
#pragma unroll 1
      for (int i = 0; i < S::COFACTOR_LOG2; ++i) acc = te_dbl<S>(acc, true);
      fe_store(slot + 6 * NL, acc.X); fe_store(slot + 7 * NL, acc.Y); fe_store(slot + 8 * NL, acc.Z);
    }
  }
  // second shared inversion: the Z coordinates (forward prefix, backward unwind)
#pragma unroll 1
  for (int j = 0; j < K; ++j) {
    const size_t item = first + j;
    if (item < n) {
      uint32_t* slot = pts_base + item * PROVE_PTS_WORDS;
      fe_store(slot + 9 * NL, runz);
      runz = fe_mul(runz, fe_load<1, 5>(slot + 8 * NL));
    }
  }
  FeN invz = fe_inv(runz);
#pragma unroll 1
  for (int j = K - 1; j >= 0; --j) {
    const size_t item = first + j;
    if (item < n) {
      uint32_t* slot = pts_base + item * PROVE_PTS_WORDS;
      FeP Z = fe_load<1, 5>(slot + 8 * NL);
      FeN zi = fe_mul(invz, fe_load<1, 2>(slot + 9 * NL));
      invz = fe_mul(invz, Z);
      FeN x = fe_mul(fe_load<1, 5>(slot + 6 * NL), zi), y = fe_mul(fe_load<1, 5>(slot + 7 * NL), zi);
      uint32_t sk[8], h_enc[8], k[8];
      const uint32_t* w = reinterpret_cast<const uint32_t*>(sk_arr + item * 32);
#pragma unroll
      for (int i = 0; i < 8; ++i) sk[i] = w[i];
      te_encode_affine(h_enc, x, y, T.sq.str.flags);
      nonce_rfc8032<S>(k, sk, h_enc);
      build_prove_tables<S>(tabs_base + item * ProveLayout<S>::TAB_WORDS, x, y);
      uint32_t* aux = aux_base + item * aux_stride;
#pragma unroll
      for (int i = 0; i < 8; ++i) { aux[i] = h_enc[i]; aux[8 + i] = k[i]; }
      flags[item] = fr_is_canonical<S>(sk) ? 1 : 0;
    }
  }
}
#else
template <class S>
VRF_HD void prove_prepare_multi(int K, const DevTables& T, size_t first, size_t n, const uint8_t* sk_arr,
                                const BytesViewLite& msgs, uint32_t* tabs_base, uint32_t* pts_base,
                                uint32_t* aux_base, int aux_stride, uint8_t* flags);   // the Elligator path: field 0 only
#endif

// scalar * P from the GLV table pair {P, psi P} (or the single 253-bit table of a suite without endomorphism)
template <class S, bool CT>
VRF_HD PtE var_base_mul(const uint32_t* tab, const uint32_t scalar[8]) {
  PtE w;
  if constexpr (S::HAS_GLV) {
    Straus4 q;
    GlvHalf h[2];
    glv_decompose_bs(h[0], h[1], scalar);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      scalar_recode_signed4_128(q.rec[t], h[t].mag);
      q.neg[t] = h[t].neg;
    }
    q.tab[0] = tab; q.tab[1] = tab + WIN_TABLE_WORDS; q.tab[2] = tab; q.tab[3] = tab;
#pragma unroll
    for (int i = 0; i < 4; ++i) { q.rec[2][i] = 0; q.rec[3][i] = 0; }
    q.neg[2] = false; q.neg[3] = false;
    w = straus4<S, 2, CT>(q);
  } else {
    uint32_t rec[8];
    scalar_recode_signed4(rec, scalar);
    w = win_mul<S, CT>(tab, rec);
  }
  return w;
}

template <class S, bool CT = false>
VRF_HD void prove_mul_item(uint32_t* out /*2*UV_WORDS*/, const DevTables& T, const uint32_t* tab,
                           const uint32_t scalar[8], const uint32_t* scalar2) {
  const PtE w = prove_var_mul<S, CT>(tab, scalar);
  fe_store(out, w.X); fe_store(out + NL, w.Y); fe_store(out + 2 * NL, w.Z);
  PtE c;
  if (scalar2) {
    uint32_t k2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) k2[i] = scalar2[i];
    c = gcomb_mul2<S>(T.g_comb, scalar, T.b_comb, k2);
  } else {
    c = gcomb_mul<S>(T.g_comb, scalar);
  }
  fe_store(out + UV_WORDS, c.X); fe_store(out + UV_WORDS + NL, c.Y);
  fe_store(out + UV_WORDS + 2 * NL, c.Z);
}

// pts: [sk*H, sk*G (+b*B), k*H, k*G (+kb*B)] projective.  Writes gamma, c, s, pk (= pk_com for
// Pedersen) and the encodings of k*G (+kb*B) = R and k*H = Ok.
template <class S>
VRF_HD void prove_finish_item(uint32_t gamma_out[8], uint32_t c_out[8], uint32_t s_out[8],
                              uint32_t pk_out[8], uint32_t r_out[8], uint32_t ok_out[8],
                              const uint32_t* pts_in, const uint32_t h_enc[8],
                              const uint32_t sk[8], const uint32_t k[8], const uint8_t* ad,
                              uint32_t ad_len, const SuiteStr& ss) {
  FeP zin[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) zin[i] = fe_load<1, 5>(pts_in + i * UV_WORDS + 2 * NL);
  FeN zi[4];
  fe_batch_inv<true>(zi, zin);                  // secret-dependent Z: fixed-shape inversion
  uint32_t enc[4][8];
#pragma unroll 1
  for (int i = 0; i < 4; ++i) {
    FeN z = fe_select(i == 0, zi[0], fe_select(i == 1, zi[1], fe_select(i == 2, zi[2], zi[3])));
    uint32_t e[8];
    te_encode_affine(e, fe_mul(fe_load<1, 5>(pts_in + i * UV_WORDS), z),
                     fe_mul(fe_load<1, 5>(pts_in + i * UV_WORDS + NL), z), ss.flags);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (i == 0) enc[0][j] = e[j];
      if (i == 1) enc[1][j] = e[j];
      if (i == 2) enc[2][j] = e[j];
      if (i == 3) enc[3][j] = e[j];
    }
  }
  uint32_t pts[5][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    pts[0][j] = enc[1][j];      // pk = sk*G
    pts[1][j] = h_enc[j];
    pts[2][j] = enc[0][j];      // Gamma = sk*H
    pts[3][j] = enc[3][j];      // k*G
    pts[4][j] = enc[2][j];      // k*H
  }
  uint32_t c[8], cs[8], s[8];
  challenge5<S>(c, pts, ad, ad_len, ss);
  fr_mul<S>(cs, c, sk);
  fr_add<S>(s, cs, k);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    gamma_out[j] = pts[2][j]; c_out[j] = c[j]; s_out[j] = s[j]; pk_out[j] = pts[0][j];
    r_out[j] = pts[3][j]; ok_out[j] = pts[4][j];
  }
}

// ------------------------------------------------------------------------ Pedersen verify
// [ref src/lib.rs:14 `pedersen::Verifier::verify`]  SURVEY.md A.5:
//   c = challenge(pk_com, H, Gamma, R, Ok, ad);  accept iff  s*H - c*Gamma == Ok  and
//   s*G + sb*B - c*pk_com == R.
// decode : decompress H, Gamma, pk_com, R, Ok (one shared inversion), tables of H, Gamma, pk_com,
//          challenge c                                                             -> HBM
// straus : lane A: s*H - c*Gamma ; lane B: s*G - c*pk_com + sb*B                   -> HBM
// finish : projective equality with the affine Ok and R.
constexpr int PED_OK_OFF = 2 * UV_WORDS;            // affine Ok (x, y) in the pts region
constexpr int PED_R_OFF = 2 * UV_WORDS + 2 * NL;    // affine R (x, y)

VRF_HD FeN fe_sel5(int p, const FeN (&a)[5]) {
  return fe_select(p == 0, a[0], fe_select(p == 1, a[1], fe_select(p == 2, a[2], fe_select(p == 3, a[3], a[4]))));
}

// enc: [H, Gamma, pk_com, R, Ok].  tabs: GLV table pairs of H, Gamma, pk_com (6 tables).  pts: receives affine Ok, R.
template <class S>
VRF_HD bool pedersen_verify_decode_item(uint32_t c_out[8], const DevTables& T,
                                        const uint32_t (&enc)[5][8], const uint8_t* ad,
                                        uint32_t ad_len, uint32_t* tabs, uint32_t* pts, uint32_t check_mask = 0) {
  FeN ys[5], dens[5], dinv[5];
  Fe<1, 6> nums[5];
  bool flags[5], oks[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    DecodeA a = decode_phase_a<S>(enc[i]);
    ys[i] = a.y; dens[i] = a.den; nums[i] = a.num; flags[i] = a.flag; oks[i] = a.ok;
  }
  fe_batch_inv(dinv, dens);
  bool valid = true;
#pragma unroll 1
  for (int p = 0; p < 5; ++p) {
    DecodeA a;
    a.y = fe_sel5(p, ys);
    a.den = fe_sel5(p, dens);
    a.num = fe_select(p == 0, nums[0], fe_select(p == 1, nums[1], fe_select(p == 2, nums[2],
                      fe_select(p == 3, nums[3], nums[4]))));
    a.flag = p == 0 ? flags[0] : p == 1 ? flags[1] : p == 2 ? flags[2] : p == 3 ? flags[3] : flags[4];
    a.ok = p == 0 ? oks[0] : p == 1 ? oks[1] : p == 2 ? oks[2] : p == 3 ? oks[3] : oks[4];
    FeN di = fe_sel5(p, dinv);
    Fe<1, 4> x;
    valid = decode_phase_b<S>(x, a, di, T.sq) && valid;
    // check_mask: H is an input, Gamma an output, pk_com / R / Ok proof points
    if (check_mask & (p == 0 ? CHK_INPUT : p == 1 ? CHK_OUTPUT : CHK_PROOF))
      valid = in_prime_subgroup<S>(fe_mul(x, fe_one()), a.y, T.sq) && valid;
    if (p < 3) {
      build_glv_tables<S>(tabs + p * 2 * WIN_TABLE_WORDS, x, a.y);
    } else {
      uint32_t* dst = pts + (p == 3 ? PED_R_OFF : PED_OK_OFF);
      fe_store(dst, x);
      fe_store(dst + NL, a.y);
    }
  }
  uint32_t cp[5][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    cp[0][j] = enc[2][j]; cp[1][j] = enc[0][j]; cp[2][j] = enc[1][j]; cp[3][j] = enc[3][j]; cp[4][j] = enc[4][j];
  }
  challenge5<S>(c_out, cp, ad, ad_len, T.sq.str);
  return valid;
}

// half 0: s*H - c*Gamma (4-table GLV Straus) ; half 1: -c*pk_com (2-table GLV Straus) + s*G + sb*B
// (two fixed-base combs).  tabs: GLV table pairs of H, Gamma, pk_com.
template <class S, int HALF>
VRF_HD void pedersen_verify_straus_item(uint32_t* out_uv, const DevTables& T, const uint32_t* tabs,
                                        const uint32_t c[8], const uint32_t s[8],
                                        const uint32_t sb[8]) {
  if constexpr (!S::HAS_GLV) {
    uint32_t recs[8], recc[8];
    scalar_recode_signed4(recs, s);
    scalar_recode_signed4(recc, c);
    PtE r;
    if (HALF == 0) {
      r = straus2<S>(tabs, recs, tabs + 2 * WIN_TABLE_WORDS, recc, true, challenge_top_window(T.sq.str));   // s*H - c*Gamma
    } else {
      r = win_mul<S>(tabs + 4 * WIN_TABLE_WORDS, recc, true, challenge_top_window(T.sq.str));                // -c*pk_com
      r = gcomb_add<S>(r, T.g_comb, s);
      r = gcomb_add<S>(r, T.b_comb, sb);
    }
    fe_store(out_uv, r.X);
    fe_store(out_uv + NL, r.Y);
    fe_store(out_uv + 2 * NL, r.Z);
    return;
  } else {
  Straus4 q;
  GlvHalf h[4];
  glv_decompose_bs(h[0], h[1], c);
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    scalar_recode_signed4_128(q.rec[t], h[t].mag);
    q.neg[t] = !h[t].neg;                      // the c terms are subtracted
  }
  PtE r;
  if (HALF == 0) {
    glv_decompose_bs(h[2], h[3], s);
#pragma unroll
    for (int t = 2; t < 4; ++t) {
      scalar_recode_signed4_128(q.rec[t], h[t].mag);
      q.neg[t] = h[t].neg;
    }
    q.tab[0] = tabs + 2 * WIN_TABLE_WORDS; q.tab[1] = tabs + 3 * WIN_TABLE_WORDS;   // Gamma pair
    q.tab[2] = tabs; q.tab[3] = tabs + WIN_TABLE_WORDS;                             // H pair
    r = straus4<S, 4>(q);
  } else {
    q.tab[0] = tabs + 4 * WIN_TABLE_WORDS; q.tab[1] = tabs + 5 * WIN_TABLE_WORDS;   // pk_com pair
    q.tab[2] = tabs; q.tab[3] = tabs;
#pragma unroll
    for (int i = 0; i < 4; ++i) { q.rec[2][i] = 0; q.rec[3][i] = 0; }
    q.neg[2] = false; q.neg[3] = false;
    r = straus4<S, 2>(q);
    r = gcomb_add<S>(r, T.g_comb, s);
    r = gcomb_add<S>(r, T.b_comb, sb);
  }
  fe_store(out_uv, r.X);
  fe_store(out_uv + NL, r.Y);
  fe_store(out_uv + 2 * NL, r.Z);
  }
}

// projective P (X, Y, Z) equals affine (x, y)
VRF_HD bool proj_eq_affine(const uint32_t* P, const uint32_t* xy) {
  FeP X = fe_load<1, 5>(P), Y = fe_load<1, 5>(P + NL), Z = fe_load<1, 5>(P + 2 * NL);
  Fe<1, 4> x = fe_load<1, 4>(xy);
  FeN y = fe_load<1, 2>(xy + NL);
  return fe_eq(X, fe_mul(x, Z)) && fe_eq(Y, fe_mul(y, Z)) && !fe_is_zero(Z);
}

template <class S>
VRF_HD uint32_t pedersen_verify_finish_item(const uint32_t* pts, const uint32_t s[8],
                                            const uint32_t sb[8], bool valid) {
  valid = valid && fr_is_canonical<S>(s) && fr_is_canonical<S>(sb);
  bool ok = proj_eq_affine(pts, pts + PED_OK_OFF) && proj_eq_affine(pts + UV_WORDS, pts + PED_R_OFF);
  if (!valid) return ST_INVALID_DATA;
  return ok ? ST_OK : ST_VERIFICATION_FAILURE;
}

// Shared inversion of the 4K projective Z of K proofs: writes the affine encodings of the four points
// of every item (sk*H, sk*G(+bB), k*H, k*G(+kbB)) to enc_out[item][4][8].  Prefix products are parked in
// the items' (now idle) table slots.
constexpr int PROVE_ENC_OFF = 64;     // word offset of the 4 x 8-word encodings inside an item's table slot
constexpr int PROVE_X_OFF = 32;       // the canonical x of encoding j sits PROVE_X_OFF words after it (affine outputs)
template <class S>
VRF_HD void prove_encode_multi(int K, size_t first, size_t n, const uint32_t* pts_base, uint32_t* tabs_base,
                               int tabs_stride, uint32_t sflags) {
  FeN run = fe_one();
#pragma unroll 1
  for (int j = 0; j < 4 * K; ++j) {
    const size_t item = first + j / 4;
    if (item < n) {
      fe_store(tabs_base + item * tabs_stride + (j & 3) * NL, run);
      run = fe_mul(run, fe_load<1, 5>(pts_base + item * PROVE_PTS_WORDS + (j & 3) * UV_WORDS + 2 * NL));
    }
  }
  FeN inv = fe_inv<true>(run);                  // Z of sk*H, k*H, k*G: fixed-shape inversion
#pragma unroll 1
  for (int j = 4 * K - 1; j >= 0; --j) {
    const size_t item = first + j / 4;
    if (item < n) {
      const uint32_t* pt = pts_base + item * PROVE_PTS_WORDS + (j & 3) * UV_WORDS;
      FeP Z = fe_load<1, 5>(pt + 2 * NL);
      FeN zi = fe_mul(inv, fe_load<1, 2>(tabs_base + item * tabs_stride + (j & 3) * NL));
      inv = fe_mul(inv, Z);
      uint32_t e[8], xw[8];
      fe_to_u256(xw, fe_mul(fe_load<1, 5>(pt), zi));
      fe_to_u256(e, fe_mul(fe_load<1, 5>(pt + NL), zi));
      if (te_x_sign(xw, sflags)) e[7] |= 0x80000000u;             // te_encode_affine, keeping the canonical x
      uint32_t* dst = tabs_base + item * tabs_stride + PROVE_ENC_OFF + (j & 3) * 8;
#pragma unroll
      for (int k = 0; k < 8; ++k) { dst[k] = e[k]; dst[PROVE_X_OFF + k] = xw[k]; }
    }
  }
}

// challenge + response from already encoded points: enc = [sk*H, pk(_com), k*H, R]
template <class S>
VRF_HD void prove_respond_item(uint32_t c_out[8], uint32_t s_out[8], const uint32_t* enc, const uint32_t h_enc[8],
                               const uint32_t sk[8], const uint32_t k[8], const uint8_t* ad, uint32_t ad_len,
                               const SuiteStr& ss) {
  uint32_t pts[5][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    pts[0][j] = enc[8 + j];       // pk = sk*G (+ b*B)
    pts[1][j] = h_enc[j];
    pts[2][j] = enc[j];           // Gamma = sk*H
    pts[3][j] = enc[24 + j];      // k*G (+ kb*B)
    pts[4][j] = enc[16 + j];      // k*H
  }
  uint32_t cs[8];
  challenge5<S>(c_out, pts, ad, ad_len, ss);
  fr_mul<S>(cs, c_out, sk);
  fr_add<S>(s_out, cs, k);
}

// RFC 9381 proof_to_hash hashes cofactor * Gamma (SS_HASH_COFACTOR; upstream does not): decode, COFACTOR_LOG2 doublings,
// encode.  Returns false if gamma does not decode (the caller reports an all-zero hash).
template <class S>
VRF_HD bool output_cofactor_encoding(uint32_t enc[8], const uint32_t gamma[8], const SqrtTables& T) {
  DecodeA a = decode_phase_a<S>(gamma);
  FeN di = fe_inv(a.den);
  Fe<1, 4> xx;
  const bool ok = decode_phase_b<S>(xx, a, di, T);
  PtE p = te_from_affine(xx, a.y);
#pragma unroll 1
  for (int i = 0; i < S::COFACTOR_LOG2; ++i) p = te_dbl<S>(p, true);
  FeN x, y;
  te_to_affine(x, y, p);
  te_encode_affine(enc, x, y, T.str.flags);
  return ok;
}

// [ref src/lib.rs:15 `Output::hash` / utils::point_to_hash_rfc_9381]  SURVEY.md A.4:
// beta = SHA512(suite_id || 0x03 || enc(Gamma) || 0x00)   (no cofactor multiplication)
template <class S>
VRF_HD void output_hash_item(uint32_t out16[16], const uint32_t gamma[8], const SuiteStr& ss) {
  Sha512 h;
  sha512_init(h);
  put_suite_id(h, ss);
  sha512_put_byte(h, 0x03);
  uint64_t w[4];
  sha512_words_le32x8(w, gamma);
  sha512_put_words(h, w);
  sha512_put_byte(h, 0x00);
  sha512_final(h);
#pragma unroll
  for (int j = 0; j < 16; ++j) out16[j] = sha512_word_mem(h, j);
}

// [ref src/lib.rs:16 `Secret::from_seed`]  SURVEY.md A.2: sk = int_le(SHA512(seed)) mod r
template <class S>
VRF_HD void secret_from_seed_item(uint32_t sk[8], const uint8_t* seed, uint32_t seed_len) {
  Sha512 h;
  sha512_init(h);
  sha512_put_bytes(h, seed, seed_len);
  sha512_final(h);
  uint32_t le[16];
  sha512_le512(le, h);
  fr_reduce512<S>(sk, le);
}

// [ref src/lib.rs:16 `Secret::public`]  pk = sk*G via the fixed-base comb, encoded.
template <class S>
VRF_HD void public_from_secret_item(uint32_t pk[8], const DevTables& T, const uint32_t sk[8]) {
  PtE p = gcomb_mul<S>(T.g_comb, sk);
  FeN x, y;
  te_to_affine<true>(x, y, p);
  te_encode_affine(pk, x, y, T.sq.str.flags);
}

VRF_NS_END
