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

namespace vrf {

enum : uint32_t { ST_OK = 0, ST_VERIFICATION_FAILURE = 1, ST_INVALID_DATA = 2 };

// Shared, read-only device tables (built once per context, see kernels.hip: k_init_tables)
struct DevTables {
  SqrtTables sq;
  const uint32_t* g_win;     // [8][PTC_WORDS]       j*G, j = 1..8, cached form (Straus table of G)
  const uint32_t* g_comb;    // [32][255][PTA_WORDS] j*256^w*G affine (fixed-base comb)
  const uint32_t* b_comb;    // same for the Pedersen blinding base
};

constexpr int WIN_ENTRIES = 8;                       // signed radix-16: |digit| in 1..8
constexpr int WIN_TABLE_WORDS = WIN_ENTRIES * PTC_WORDS;   // 288 words = 1152 B per base

// ------------------------------------------------------------------------ suite byte strings
struct SuiteBS : CurveBS {
  static constexpr int SUITE_ID_LEN = 25;
  static VRF_HD uint8_t suite_id(int i) {
    constexpr char s[] = "Bandersnatch_SHA-512_ELL2";
    return (uint8_t)s[i];
  }
  static constexpr int DST_LEN = 64;
  static VRF_HD uint8_t dst(int i) {
    constexpr char s[] = "ECVRF_Bandersnatch_XMD:SHA-512_ELL2_RO_Bandersnatch_SHA-512_ELL2";
    return (uint8_t)s[i];
  }
};

template <class S>
VRF_HD void put_suite_id(Sha512& h) {
#pragma unroll
  for (int i = 0; i < S::SUITE_ID_LEN; ++i) sha512_put_byte(h, S::suite_id(i));
}

// ------------------------------------------------------------------------ batch inversion
template <int N, int L, int V>
VRF_HD void fe_batch_inv(FeN (&out)[N], const Fe<L, V> (&in)[N]) {
  FeN pre[N];
  pre[0] = fe_mul(in[0], fe_one());
#pragma unroll
  for (int i = 1; i < N; ++i) pre[i] = fe_mul(pre[i - 1], in[i]);
  FeN acc = fe_inv(pre[N - 1]);
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
  FeN den = fe_canon(fe_add(fe_mul(y2, C::d()), C::aneg_m()));           // d*y^2 - a
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
  bool greater = u256_gt(xw, vrfk::QM1H32);
  x_out = fe_norm(fe_cneg(greater != a.flag, root));
  // x == 0: -0 = K*q, still a valid representation of zero
  return a.ok && sq;
}

// ------------------------------------------------------------------------ window tables
// multiples 1..8 of an affine point, cached form, written to `tab` (WIN_TABLE_WORDS words)
template <class C>
VRF_HD void build_win_table(uint32_t* tab, const FeP& x, const FeP& y) {
  PtE p1 = te_from_affine(x, y);
  PtC c1 = te_to_cached<C>(p1);
  ptc_store(tab + 0 * PTC_WORDS, c1);
  PtE p2 = te_dbl<C, true>(p1);
  ptc_store(tab + 1 * PTC_WORDS, te_to_cached<C>(p2));
  PtE p3 = te_add_cached<C>(p2, c1, false);
  ptc_store(tab + 2 * PTC_WORDS, te_to_cached<C>(p3));
  PtE p4 = te_dbl<C, true>(p2);
  ptc_store(tab + 3 * PTC_WORDS, te_to_cached<C>(p4));
  PtE p6 = te_dbl<C, true>(p3);
  ptc_store(tab + 5 * PTC_WORDS, te_to_cached<C>(p6));
  PtE p5 = te_add_cached<C>(p4, c1, false);
  ptc_store(tab + 4 * PTC_WORDS, te_to_cached<C>(p5));
  PtE p7 = te_add_cached<C>(p6, c1, false);
  ptc_store(tab + 6 * PTC_WORDS, te_to_cached<C>(p7));
  PtE p8 = te_dbl<C, true>(p4);
  ptc_store(tab + 7 * PTC_WORDS, te_to_cached<C>(p8));
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

// sa*A + sb*B by Straus with signed radix-16 digits; negB flips the sign of the B terms.
// reca / recb are recoded scalars (scalar_recode_signed4).
template <class C>
VRF_HD PtE straus2(const uint32_t* tabA, const uint32_t reca[8], const uint32_t* tabB,
                   const uint32_t recb[8], bool negB) {
  PtE acc = te_identity();
  for (int w = 63; w >= 0; --w) {
    if (w != 63) {
      for (int j = 0; j < 3; ++j) acc = te_dbl<C, false>(acc);
      acc = te_dbl<C, true>(acc);
    }
    int da = scalar_digit4(reca, w);
    int db = scalar_digit4(recb, w);
    acc = te_add_cached<C>(acc, win_lookup(tabA, da), da < 0);
    acc = te_add_cached<C>(acc, win_lookup(tabB, db), (db < 0) != negB);
  }
  return acc;
}

// k*P for one window table (used by prove: Gamma = sk*H, kH)
template <class C>
VRF_HD PtE win_mul(const uint32_t* tab, const uint32_t rec[8]) {
  PtE acc = te_identity();
  for (int w = 63; w >= 0; --w) {
    if (w != 63) {
      for (int j = 0; j < 3; ++j) acc = te_dbl<C, false>(acc);
      acc = te_dbl<C, true>(acc);
    }
    int d = scalar_digit4(rec, w);
    acc = te_add_cached<C>(acc, win_lookup(tab, d), d < 0);
  }
  return acc;
}

// k*Base from an 8-bit fixed-base comb table [32][255] of affine entries
template <class C>
VRF_HD PtE comb_mul(const uint32_t* comb, const uint32_t k[8]) {
  PtE acc = te_identity();
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

// ------------------------------------------------------------------------ table init helpers
// k*P by branch-free double-and-add (one-time table construction only)
template <class C>
VRF_HD PtE te_mul_slow(const PtE& base, const uint32_t k[8]) {
  PtE acc = te_identity();
  PtC bc = te_to_cached<C>(base);
  for (int i = 255; i >= 0; --i) {
    acc = te_dbl<C, true>(acc);
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
// comb entry (w, j): (j * 256^w) * Base as an affine cached entry, j in 1..255
template <class C>
VRF_HD void comb_entry(uint32_t* out, const FeN& bx, const FeN& by, int w, int j) {
  uint32_t k[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) k[i] = ((w >> 2) == i) ? ((uint32_t)j << ((w & 3) * 8)) : 0u;
  PtE p = te_mul_slow<C>(te_from_affine(bx, by), k);
  FeN zi = fe_inv(p.Z);
  PtA a;
  a.x = fe_mul(p.X, zi);
  a.y = fe_mul(p.Y, zi);
  a.dt = fe_mul(fe_mul(a.x, a.y), C::d());
  pta_store(out, a);
}

// ------------------------------------------------------------------------ challenge
// [ref src/lib.rs:14,16 `Suite::challenge` / utils::challenge_rfc_9381]  SURVEY.md A.4:
// c = int_be(SHA512(suite_id || 0x02 || enc(P1..P5) || ad || 0x00)[0..32]) mod r
template <class S>
VRF_HD void challenge5(uint32_t c_out[8], const uint32_t (&pts)[5][8], const uint8_t* ad,
                       uint32_t ad_len) {
  Sha512 h;
  sha512_init(h);
  put_suite_id<S>(h);
  sha512_put_byte(h, 0x02);
#pragma unroll
  for (int i = 0; i < 5; ++i) sha512_put_le32x8(h, pts[i]);
  sha512_put_bytes(h, ad, ad_len);
  sha512_put_byte(h, 0x00);
  sha512_final(h);
  uint32_t be[8];
  sha512_be256(be, h);
  fr_reduce256<S>(c_out, be);
}

// ------------------------------------------------------------------------ IETF verify
// [ref src/lib.rs:14 `ietf::Verifier::verify`]  U = s*G - c*Y, V = s*H - c*Gamma, accept iff
// challenge(Y, H, Gamma, U, V, ad) == c.   `scratch`: 3 * WIN_TABLE_WORDS words, private.
template <class S>
VRF_HD uint32_t ietf_verify_item(const DevTables& T, const uint32_t pk[8], const uint32_t hh[8],
                                 const uint32_t gamma[8], const uint32_t c[8], const uint32_t s[8],
                                 const uint8_t* ad, uint32_t ad_len, uint32_t* scratch) {
  bool valid = fr_is_canonical<S>(c) && fr_is_canonical<S>(s);
  // decode the three points with one shared inversion
  DecodeA a0 = decode_phase_a<S>(pk), a1 = decode_phase_a<S>(hh), a2 = decode_phase_a<S>(gamma);
  FeN dens[3] = {a0.den, a1.den, a2.den}, dinv[3];
  fe_batch_inv(dinv, dens);
  Fe<1, 4> x;
  uint32_t* tabY = scratch;
  uint32_t* tabH = scratch + WIN_TABLE_WORDS;
  uint32_t* tabG = scratch + 2 * WIN_TABLE_WORDS;
  valid = decode_phase_b<S>(x, a0, dinv[0], T.sq) && valid;
  build_win_table<S>(tabY, x, a0.y);
  valid = decode_phase_b<S>(x, a1, dinv[1], T.sq) && valid;
  build_win_table<S>(tabH, x, a1.y);
  valid = decode_phase_b<S>(x, a2, dinv[2], T.sq) && valid;
  build_win_table<S>(tabG, x, a2.y);

  uint32_t recs[8], recc[8];
  scalar_recode_signed4(recs, s);
  scalar_recode_signed4(recc, c);
  PtE U = straus2<S>(T.g_win, recs, tabY, recc, true);
  PtE V = straus2<S>(tabH, recs, tabG, recc, true);

  FeP zin[2] = {U.Z, V.Z};
  FeN zi[2];
  fe_batch_inv(zi, zin);
  uint32_t pts[5][8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { pts[0][i] = pk[i]; pts[1][i] = hh[i]; pts[2][i] = gamma[i]; }
  te_encode_affine(pts[3], fe_mul(U.X, zi[0]), fe_mul(U.Y, zi[0]));
  te_encode_affine(pts[4], fe_mul(V.X, zi[1]), fe_mul(V.Y, zi[1]));
  uint32_t c2[8];
  challenge5<S>(c2, pts, ad, ad_len);
  uint32_t diff = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) diff |= c2[i] ^ c[i];
  if (!valid) return ST_INVALID_DATA;
  return diff == 0 ? ST_OK : ST_VERIFICATION_FAILURE;
}

// ------------------------------------------------------------------------ hash to curve
// [ref src/lib.rs:15-16 `Input::new` -> utils::hash_to_curve_ell2_rfc_9380]  SURVEY.md A.3
// expand_message_xmd(SHA-512) with arkworks' 48-byte Z_pad, two field elements, Elligator 2
// on the Montgomery model, map to twisted Edwards, add, clear cofactor.
template <class S>
VRF_HD void put_dst_prime(Sha512& h) {
#pragma unroll
  for (int i = 0; i < S::DST_LEN; ++i) sha512_put_byte(h, S::dst(i));
  sha512_put_byte(h, (uint8_t)S::DST_LEN);
}

template <class S>
VRF_HD void hash_to_field2(Fe<1, 4>& u0, Fe<1, 4>& u1, const uint8_t* msg, uint32_t msg_len) {
  Sha512 b0;
  sha512_init(b0);
#pragma unroll
  for (int i = 0; i < 6; ++i) sha512_put(b0, 0, 8);      // Z_pad: 48 zero bytes
  sha512_put_bytes(b0, msg, msg_len);
  sha512_put_byte(b0, 0x00);
  sha512_put_byte(b0, 0x60);                             // len_in_bytes = 96
  sha512_put_byte(b0, 0x00);
  put_dst_prime<S>(b0);
  sha512_final(b0);
  Sha512 b1;
  sha512_init(b1);
#pragma unroll
  for (int i = 0; i < 8; ++i) sha512_put(b1, b0.h[i], 8);
  sha512_put_byte(b1, 0x01);
  put_dst_prime<S>(b1);
  sha512_final(b1);
  Sha512 b2;
  sha512_init(b2);
#pragma unroll
  for (int i = 0; i < 8; ++i) sha512_put(b2, b0.h[i] ^ b1.h[i], 8);
  sha512_put_byte(b2, 0x02);
  put_dst_prime<S>(b2);
  sha512_final(b2);
  // uniform = b1 (64 B) || b2[0..32].  u0 = BE(uniform[0..48]) = b1.h[0..6] ; u1 = BE(b1.h[6..8] || b2.h[0..4])
  uint32_t w0[16], w1[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { w0[i] = 0; w1[i] = 0; }
#pragma unroll
  for (int j = 0; j < 6; ++j) {          // 64-bit BE words, most significant first
    uint64_t a = b1.h[j];
    uint64_t b = (j < 2) ? b1.h[6 + j] : b2.h[j - 2];
    w0[2 * (5 - j)] = (uint32_t)a; w0[2 * (5 - j) + 1] = (uint32_t)(a >> 32);
    w1[2 * (5 - j)] = (uint32_t)b; w1[2 * (5 - j) + 1] = (uint32_t)(b >> 32);
  }
  u0 = fe_from_u512(w0);
  u1 = fe_from_u512(w1);
}

struct Ell2A {           // Elligator-2 phase a: everything before the inversion of D
  Fe<1, 4> u;
  FeN D;
};
template <class S>
VRF_HD Ell2A ell2_phase_a(const Fe<1, 4>& u) {
  Ell2A r;
  r.u = u;
  FeN u2 = fe_sqr(u);
  FeN D = fe_canon(fe_add(fe_mul5(u2), fe_one()));       // 1 + Z*u^2, Z = 5
  bool dz = true;
#pragma unroll
  for (int i = 0; i < NL; ++i) dz = dz && (D.v[i] == 0);
  r.D = fe_select(dz, fe_one(), D);
  return r;
}
VRF_HD bool fe_parity(const FeN& a) {       // canonical integer is odd
  uint32_t w[8];
  fe_to_u256(w, a);
  return w[0] & 1;
}
template <class S>
VRF_HD PtE ell2_phase_b(const Ell2A& a, const FeN& Dinv, const SqrtTables& T) {
  const FeN JK = fe_const(vrfk::BS_ELL2_JK_M), K2I = fe_const(vrfk::BS_ELL2_K2I_M);
  const FeN K = fe_const(vrfk::BS_ELL2_K_M);
  FeN x1 = fe_mul(fe_const(vrfk::BS_ELL2_NJK_M), Dinv);              // -(J/K) / D
  // gx1 = ((x1 + J/K) * x1 + 1/K^2) * x1
  FeN t = fe_mul(fe_add(x1, JK), x1);
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
  FeN y2 = fe_mul(root, a.u);                                        // sqrt(gx2) = u * sqrt(Z*gx1)
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
  Fe<1, 4> u0, u1;
  hash_to_field2<S>(u0, u1, msg, msg_len);
  Ell2A a0 = ell2_phase_a<S>(u0), a1 = ell2_phase_a<S>(u1);
  FeN ds[2] = {a0.D, a1.D}, di[2];
  fe_batch_inv(di, ds);
  PtE q0 = ell2_phase_b<S>(a0, di[0], T);
  PtE q1 = ell2_phase_b<S>(a1, di[1], T);
  PtE h = te_add<S>(q0, q1);
  for (int i = 0; i < S::COFACTOR_LOG2; ++i) h = te_dbl<S, true>(h);
  return h;
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

// ------------------------------------------------------------------------ IETF prove
// [ref src/lib.rs:14 `ietf::Prover::prove` + src/lib.rs:16 `Secret::{public, output}`]
// From (sk, H): pk = sk*G, Gamma = sk*H, k = nonce, c = challenge(pk, H, Gamma, kG, kH), s = k + c*sk.
// H is given projective (from hash-to-curve) or decoded by the caller.  scratch: WIN_TABLE_WORDS.
template <class S>
VRF_HD void ietf_prove_core(uint32_t gamma_out[8], uint32_t c_out[8], uint32_t s_out[8],
                            uint32_t h_out[8], uint32_t pk_out[8], const DevTables& T,
                            const uint32_t sk[8], const FeN& hx, const FeN& hy,
                            const uint8_t* ad, uint32_t ad_len, uint32_t* scratch) {
  uint32_t h_enc[8];
  te_encode_affine(h_enc, hx, hy);
  uint32_t k[8];
  nonce_rfc8032<S>(k, sk, h_enc);
  build_win_table<S>(scratch, hx, hy);
  uint32_t rec[8];
  scalar_recode_signed4(rec, sk);
  PtE G1 = win_mul<S>(scratch, rec);            // Gamma = sk*H
  scalar_recode_signed4(rec, k);
  PtE KH = win_mul<S>(scratch, rec);            // k*H
  PtE PK = comb_mul<S>(T.g_comb, sk);           // sk*G
  PtE KG = comb_mul<S>(T.g_comb, k);            // k*G
  FeP zin[4] = {G1.Z, KH.Z, PK.Z, KG.Z};
  FeN zi[4];
  fe_batch_inv(zi, zin);
  uint32_t pts[5][8];
  te_encode_affine(pts[0], fe_mul(PK.X, zi[2]), fe_mul(PK.Y, zi[2]));
#pragma unroll
  for (int i = 0; i < 8; ++i) pts[1][i] = h_enc[i];
  te_encode_affine(pts[2], fe_mul(G1.X, zi[0]), fe_mul(G1.Y, zi[0]));
  te_encode_affine(pts[3], fe_mul(KG.X, zi[3]), fe_mul(KG.Y, zi[3]));
  te_encode_affine(pts[4], fe_mul(KH.X, zi[1]), fe_mul(KH.Y, zi[1]));
  uint32_t c[8], cs[8], s[8];
  challenge5<S>(c, pts, ad, ad_len);
  fr_mul<S>(cs, c, sk);
  fr_add<S>(s, cs, k);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    gamma_out[i] = pts[2][i]; c_out[i] = c[i]; s_out[i] = s[i]; h_out[i] = h_enc[i];
    pk_out[i] = pts[0][i];
  }
}

// affine coordinates of a projective point (one inversion)
VRF_HD void te_to_affine(FeN& x, FeN& y, const PtE& p) {
  FeN zi = fe_inv(p.Z);
  x = fe_mul(p.X, zi);
  y = fe_mul(p.Y, zi);
}

// [ref src/lib.rs:15 `Output::hash` / utils::point_to_hash_rfc_9381]  SURVEY.md A.4:
// beta = SHA512(suite_id || 0x03 || enc(Gamma) || 0x00)   (no cofactor multiplication)
template <class S>
VRF_HD void output_hash_item(uint32_t out16[16], const uint32_t gamma[8]) {
  Sha512 h;
  sha512_init(h);
  put_suite_id<S>(h);
  sha512_put_byte(h, 0x03);
  sha512_put_le32x8(h, gamma);
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

}  // namespace vrf
