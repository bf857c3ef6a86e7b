// bsw_core.cuh -- `suites::bandersnatch_sw` (/root/reference src/lib.rs:14; upstream "Bandersnatch_SW_SHA-512_TAI"): the
// Bandersnatch group presented on its short-Weierstrass model y^2 = x^3 + a' x + b' -- arkworks' SWAffine behind
// `AffinePoint` (src/lib.rs:15) -- with ArkworksCodec's 33-byte compressed points, try-and-increment hash-to-curve, the
// RFC 8032 nonce and the RFC 9381 challenge.
//
// MI355X-first: the GROUP is the same group as the twisted-Edwards suite's, so nothing here adds points on the Weierstrass
// curve.  A point crosses `utils::te_sw_map` (te_sw_map.cuh) once on its way in and once on its way out, and all arithmetic
// in between -- window tables, the GLV Straus loops, the fixed-base combs, the 2-descent subgroup test -- is the
// twisted-Edwards suite's, kernels included (k_prove_mul, verify_straus_item, pedersen_verify_straus_item).  What the suite
// owns is the codec and every hash that absorbs an encoded point:
//   in : x < q, flags; y = sqrt(x^3 + a' x + b') picked by the flag; (x, y) -> Edwards (one inversion)
//   out: Edwards projective (X : Y : Z) -> sx = N X / D, sy = 3 (Z + Y)(a - d) Z / D with N = 3 (Z + Y)(a - d) + 2 (a + d)(Z - Y)
//        and D = 12 X (Z - Y): ONE inversion per point gives both coordinates (te_sw_map with x = X/Z, y = Y/Z substituted).
// The oracle (oracle/bsw_oracle.py) does the opposite -- chord-and-tangent on the Weierstrass curve, no map -- so that
// agreement between the two is a statement about the map as well as the scheme.
//
// Wire format [ref src/lib.rs:14 `codec`: ArkworksCodec, ark_ec::short_weierstrass::Affine serialization, as recalled]:
// 33 bytes = x little-endian || one flag byte (bit 7: y is the larger of {y, q - y}; bit 6: infinity, written with x = 0).
// Reading: both flags = error; x >= q = error; the low six bits of the flag byte and, for infinity, x itself are ignored.
// Hashes take the canonical re-encoding (flag byte 0x00 / 0x80 / 0x40, infinity with x = 0).
//
// Not representable on the Edwards side: the two points of order 2 that the Edwards model has at infinity.  They, and sums
// with them, lie outside the prime-order subgroup, so a checked decode (the default) rejects them like upstream; with the
// subgroup test switched off (VRFHIP_FLAG_PREVALIDATED_*) a point with y = 0 is reported InvalidData here where upstream
// would compute with it.
#pragma once
#include "vrf_core.cuh"
#include "te_sw_map.cuh"
#include "constants_bsw.gen.h"

#if VRF_FIELD != 0
#error "bsw_core.cuh is the short-Weierstrass Bandersnatch suite: compile with -DVRF_FIELD=0"
#endif

VRF_NS_BEGIN

using BswS = SuiteBS;                         // the arithmetic
struct SuiteBW : SuiteBS {};                  // tag for the codec-dependent specialisations (tai_attempt_candidate)
constexpr int BSW_PT = 33;
constexpr uint32_t BSW_NEG = 0x80u, BSW_INF = 0x40u;

struct Enc33 {
  uint32_t w[8];     // x, little-endian words
  uint32_t fl;       // flag byte
};

VRF_HD Enc33 load33(const uint8_t* base, size_t i) {
  const uint8_t* p = base + i * BSW_PT;
  Enc33 e;
#pragma unroll
  for (int k = 0; k < 8; ++k)
    e.w[k] = (uint32_t)p[4 * k] | ((uint32_t)p[4 * k + 1] << 8) | ((uint32_t)p[4 * k + 2] << 16) | ((uint32_t)p[4 * k + 3] << 24);
  e.fl = p[32];
  return e;
}
VRF_HD void store33(uint8_t* base, size_t i, const Enc33& e, bool ok = true) {
  uint8_t* p = base + i * BSW_PT;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const uint32_t w = ok ? e.w[k] : 0u;
    p[4 * k] = (uint8_t)w; p[4 * k + 1] = (uint8_t)(w >> 8); p[4 * k + 2] = (uint8_t)(w >> 16); p[4 * k + 3] = (uint8_t)(w >> 24);
  }
  p[32] = ok ? (uint8_t)e.fl : (uint8_t)0;
}
// the encoding `point_encode` gives the point these bytes decode to (for bytes that decode)
VRF_HD Enc33 enc33_canonical(const Enc33& e) {
  Enc33 r;
  const uint32_t fl = e.fl & 0xC0u;
#pragma unroll
  for (int k = 0; k < 8; ++k) r.w[k] = fl == BSW_INF ? 0u : e.w[k];
  r.fl = fl;
  return r;
}
VRF_HD Enc33 enc33_infinity() {
  Enc33 r;
#pragma unroll
  for (int k = 0; k < 8; ++k) r.w[k] = 0;
  r.fl = BSW_INF;
  return r;
}

VRF_HD FeN bsw_rhs(const FeN& x) {            // x^3 + a' x + b'
  const FeN a2 = fe_const(vrfk_bsw::A_M), b2 = fe_const(vrfk_bsw::B_M);
  return fe_full(fe_add(fe_mul(fe_add(fe_sqr(x), a2), x), b2));
}

// ---- decode: wire -> Edwards affine ----
// sx, sy: the Weierstrass coordinates (for callers that hand them out); tx, ty: the Edwards image ((0, 1) for infinity).
// false: flags, x >= q, not on the curve, or y = 0 (see the header).
template <class C>
VRF_HD bool bsw_decode(FeN& tx, FeN& ty, FeN& sx, FeN& sy, bool& inf, const Enc33& e, const SqrtTables& T) {
  const uint32_t fl = e.fl & 0xC0u;
  bool ok = fl != 0xC0u && !u256_ge(e.w, vrfk::Q32);
  inf = fl == BSW_INF;
  sx = fe_from_u256(e.w);
  FeN root;
  bool sq = fe_sqrt_or_zsqrt(root, bsw_rhs(sx), T);
  uint32_t yw[8];
  fe_to_u256(yw, root);
  uint32_t nz = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) nz |= yw[i];
  sq = sq || nz == 0;
  const bool larger = u256_gt(yw, vrfk::QM1H32);
  sy = fe_full(fe_cneg(larger != (fl == BSW_NEG), root));
  FeN mx, my;
  const bool some = sw_to_te<C>(mx, my, sx, sy);
  tx = fe_select(inf, fe_zero(), mx);
  ty = fe_select(inf, fe_one(), my);
  return ok && (inf || (sq && some));
}
template <class C>
VRF_HD bool bsw_decode(FeN& tx, FeN& ty, const Enc33& e, const SqrtTables& T) {
  FeN sx, sy;
  bool inf;
  return bsw_decode<C>(tx, ty, sx, sy, inf, e, T);
}

// ---- the way in, split so that several points share ONE inversion (Montgomery's trick across their map denominators) ----
// Phase A leaves sx | sy | den | prefix (the product of the denominators in front of this point) in a 36-word slot of the
// caller's scratch and multiplies the running product; phase B, called in REVERSE order with inv = 1 / (product of all the
// denominators so far), gives the point's Edwards coordinates and strips its denominator from inv.
constexpr int BSW_SLOT = 4 * NL;
constexpr uint32_t BSW_A_OK = 1u, BSW_A_INF = 2u;
VRF_HD uint32_t bsw_in_a_store(uint32_t* slot, FeN& run, const FeN& sx, const FeN& sy, bool ok, bool inf) {
  const FeN a = te_coeff_a<BswS>(), d = BswS::d();
  const FeN x12 = fe_full(fe_dbl(fe_dbl(fe_times3(sx))));
  const FeN den_w = fe_full(fe_sub(fe_full(fe_add(x12, a)), fe_full(fe_mul5(d))));           // 12 X + a - 5 d
  const FeN den = fe_mul(fe_full(fe_dbl(fe_times3(sy))), den_w);                                // 6 Y (12 X + a - 5 d)
  const bool some = !fe_is_zero(den);
  const FeN dd = fe_select(some && !inf, den, fe_one());
  fe_store(slot, sx); fe_store(slot + NL, sy); fe_store(slot + 2 * NL, dd); fe_store(slot + 3 * NL, run);
  run = fe_mul(run, dd);
  return ((ok && (inf || some)) ? BSW_A_OK : 0u) | (inf ? BSW_A_INF : 0u);
}
// from the wire
VRF_HD uint32_t bsw_in_a(uint32_t* slot, FeN& run, const Enc33& e, const SqrtTables& T) {
  const uint32_t fl = e.fl & 0xC0u;
  bool ok = fl != 0xC0u && !u256_ge(e.w, vrfk::Q32);
  const bool inf = fl == BSW_INF;
  const FeN sx = fe_from_u256(e.w);
  FeN root;
  bool sq = fe_sqrt_or_zsqrt(root, bsw_rhs(sx), T);
  uint32_t yw[8];
  fe_to_u256(yw, root);
  uint32_t nz = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) nz |= yw[i];
  sq = sq || nz == 0;
  const bool larger = u256_gt(yw, vrfk::QM1H32);
  const FeN sy = fe_full(fe_cneg(larger != (fl == BSW_NEG), root));
  return bsw_in_a_store(slot, run, sx, sy, ok && (inf || sq), inf);
}
VRF_HD void bsw_in_b(FeN& tx, FeN& ty, FeN& inv, const uint32_t* slot, bool inf) {
  const FeN sx = fe_load<1, 2>(slot), sy = fe_load<1, 2>(slot + NL), den = fe_load<1, 2>(slot + 2 * NL);
  const FeN prefix = fe_load<1, 2>(slot + 3 * NL);
  const FeN i = fe_mul(inv, prefix);
  inv = fe_mul(inv, den);
  const FeN a = te_coeff_a<BswS>(), d = BswS::d();
  const FeN x6 = fe_full(fe_dbl(fe_times3(sx))), y6 = fe_full(fe_dbl(fe_times3(sy)));
  const FeN x12 = fe_full(fe_dbl(x6));
  const FeN num_v = fe_full(fe_sub(x6, fe_full(fe_add(a, d))));                                // 6 X - (a + d)
  const FeN num_w = fe_full(fe_add(fe_sub(x12, fe_full(fe_mul5(a))), d));                       // 12 X - 5 a + d
  const FeN den_w = fe_full(fe_sub(fe_full(fe_add(x12, a)), fe_full(fe_mul5(d))));            // 12 X + a - 5 d
  tx = fe_select(inf, fe_zero(), fe_mul(fe_mul(num_v, den_w), i));
  ty = fe_select(inf, fe_one(), fe_mul(fe_mul(num_w, y6), i));
}

// ---- encode: Edwards projective -> wire ----
// the Weierstrass image of (X : Y : Z) as two numerators over one denominator (den = 1 where the image has none: the
// neutral element, and the point of order 2 on the Edwards y axis whose image is (nx / den, 0))
struct SwFrac {
  FeN nx, ny, den;
  bool inf;
};
template <class C>
VRF_HD SwFrac bsw_frac(const FeP& X, const FeP& Y, const FeP& Z) {
  const FeN a = te_coeff_a<C>(), d = C::d();
  const FeN amd = fe_full(fe_sub(a, d)), apd2 = fe_full(fe_dbl(fe_add(a, d)));      // a - d, 2 (a + d)
  const FeN n = fe_full(fe_add(Z, Y)), m = fe_full(fe_sub(Z, Y));
  const bool xz = fe_is_zero(X), mz = fe_is_zero(m);
  const FeN namd3 = fe_times3(fe_mul(n, amd));                                      // 3 (Z + Y)(a - d)
  const FeN num = fe_full(fe_add(namd3, fe_mul(apd2, m)));
  const FeN xs = fe_select(xz, fe_one(), fe_full(X));
  SwFrac r;
  r.inf = mz;                                   // Y = Z happens on the curve only at (0, 1)
  r.nx = fe_mul(num, xs);
  r.ny = fe_select(xz, fe_zero(), fe_mul(namd3, Z));
  r.den = fe_select(mz, fe_one(), fe_full(fe_dbl(fe_dbl(fe_times3(fe_mul(xs, m))))));   // 12 X (Z - Y)
  return r;
}
// with 1 / den; sxw (nullable): the canonical words of both coordinates for x || y outputs
VRF_HD Enc33 bsw_encode_frac(const SwFrac& f, const FeN& den_inv, uint32_t* syw = nullptr) {
  Enc33 e;
  uint32_t yw[8];
  fe_to_u256(e.w, fe_mul(f.nx, den_inv));
  fe_to_u256(yw, fe_mul(f.ny, den_inv));
  e.fl = u256_gt(yw, vrfk::QM1H32) ? BSW_NEG : 0u;
  if (f.inf) e = enc33_infinity();
  if (syw) {
#pragma unroll
    for (int k = 0; k < 8; ++k) syw[k] = f.inf ? 0u : yw[k];
  }
  return e;
}
template <class C, bool CT = false>
VRF_HD Enc33 bsw_encode(const PtE& p) {
  const SwFrac f = bsw_frac<C>(p.X, p.Y, p.Z);
  return bsw_encode_frac(f, fe_inv<CT>(f.den));
}

// ---- hashing encodings ----
VRF_HD uint32_t enc33_byte(const Enc33& e, int j) { return j == 32 ? (e.fl & 0xffu) : (e.w[j >> 2] >> (8 * (j & 3))) & 0xffu; }
// N encodings as a big-endian packed byte string (sha512_put_packed)
template <int N>
VRF_HD void enc33_pack(uint64_t (&w)[(33 * N + 7) / 8], const Enc33 (&pts)[N]) {
#pragma unroll
  for (int k = 0; k < (33 * N + 7) / 8; ++k) w[k] = 0;
#pragma unroll
  for (int p = 0; p < N; ++p) {
#pragma unroll
    for (int j = 0; j < 33; ++j) {
      const int pos = 33 * p + j;
      w[pos >> 3] |= (uint64_t)enc33_byte(pts[p], j) << (56 - 8 * (pos & 7));
    }
  }
}
VRF_HD void sha512_put_enc33(Sha512& h, const Enc33& e) {
  uint64_t w[4];
  sha512_words_le32x8(w, e.w);
#pragma unroll
  for (int i = 0; i < 4; ++i) sha512_put(h, w[i], 8);
  sha512_put_byte(h, (uint8_t)e.fl);
}

// [ref src/lib.rs:14,16 `Suite::nonce` / utils::nonce_rfc_8032]  k = int_le(SHA512(SHA512(sk_le32)[32..64] || enc(H))) mod r
VRF_HD void bsw_nonce(uint32_t k[8], const uint32_t sk[8], const Enc33& h_enc) {
  Sha512 a;
  sha512_init(a);
  sha512_put_le32x8(a, sk);
  sha512_final(a);
  Sha512 b;
  sha512_init(b);
#pragma unroll
  for (int i = 4; i < 8; ++i) sha512_put(b, a.h[i], 8);
  sha512_put_enc33(b, h_enc);
  sha512_final(b);
  uint32_t le[16];
  sha512_le512(le, b);
  fr_reduce512<BswS>(k, le);
}

// [ref src/lib.rs:14,16 `Suite::challenge` / utils::challenge_rfc_9381]
// c = int_be(SHA512(suite_id || 0x02 || enc(P1..P5) || ad || 0x00)[0..CHALLENGE_LEN]) mod r
VRF_HD void bsw_challenge5(uint32_t c_out[8], const Enc33 (&pts)[5], const uint8_t* ad, uint32_t ad_len, const SuiteStr& ss) {
  Sha512 h;
  sha512_init(h);
  put_suite_id(h, ss);
  sha512_put_byte(h, 0x02);
  uint64_t w[21];
  enc33_pack<5>(w, pts);
  sha512_put_packed(h, w, 165);
  sha512_put_bytes(h, ad, ad_len);
  sha512_put_byte(h, 0x00);
  sha512_final(h);
  uint32_t be[8];
  sha512_be256(be, h);
  if (ss.challenge_len != 32u) u256_shr_bytes(be, 32u - ss.challenge_len);
  fr_reduce256<BswS>(c_out, be);
}

// [ref src/lib.rs:14 `pedersen::PedersenSuite::blinding`]  b = int_be(SHA512(suite_id || 0xCC || sk_le32 || enc(H) || ad || 0x00)) mod r
VRF_HD void bsw_blinding(uint32_t b[8], const uint32_t sk[8], const Enc33& h_enc, const uint8_t* ad, uint32_t ad_len,
                         const SuiteStr& ss) {
  Sha512 h;
  sha512_init(h);
  put_suite_id(h, ss);
  sha512_put_byte(h, 0xCC);
  uint64_t w[4];
  sha512_words_le32x8(w, sk);
  sha512_put_words(h, w);
  sha512_put_enc33(h, h_enc);
  sha512_put_bytes(h, ad, ad_len);
  sha512_put_byte(h, 0x00);
  sha512_final(h);
  uint32_t be[16];
  sha512_be512(be, h);
  fr_reduce512<BswS>(b, be);
}

// [ref src/lib.rs:15 `Output::hash` / utils::point_to_hash_rfc_9381]  beta = SHA512(suite_id || 0x03 || enc(Gamma) || 0x00)
VRF_HD void bsw_output_hash(uint32_t out16[16], const Enc33& gamma, const SuiteStr& ss) {
  Sha512 h;
  sha512_init(h);
  put_suite_id(h, ss);
  sha512_put_byte(h, 0x03);
  sha512_put_enc33(h, gamma);
  sha512_put_byte(h, 0x00);
  sha512_final(h);
#pragma unroll
  for (int j = 0; j < 16; ++j) out16[j] = sha512_word_mem(h, j);
}

// ---- [ref src/lib.rs:14 `utils::hash_to_curve_tai_rfc_9381`] ----
// candidate of attempt `ctr`: the first 33 bytes of SHA512(suite_id || 0x01 || data || ctr || 0x00), read as a compressed point
VRF_HD Enc33 bsw_tai_candidate(const uint8_t* msg, uint32_t msg_len, uint32_t ctr, const SuiteStr& ss) {
  Sha512 h;
  sha512_init(h);
  put_suite_id(h, ss);
  sha512_put_byte(h, 0x01);
  sha512_put_bytes(h, msg, msg_len);
  sha512_put_byte(h, (uint8_t)ctr);
  sha512_put_byte(h, 0x00);
  sha512_final(h);
  Enc33 e;
#pragma unroll
  for (int j = 0; j < 8; ++j) e.w[j] = sha512_word_mem(h, j);
  e.fl = sha512_word_mem(h, 8) & 0xffu;
  return e;
}
// k_tai_find's cheap half for this codec: flags say "a finite point", x < q; w = the value whose quadratic character decides.
// (The infinity flag decodes, but to the neutral element, which hash-to-curve skips: it is no candidate.)
template <>
VRF_HD bool tai_attempt_candidate<SuiteBW>(FeN& w, const uint8_t* msg, uint32_t msg_len, uint32_t ctr, const SqrtTables& T) {
  const Enc33 e = bsw_tai_candidate(msg, msg_len, ctr, T.str);
  w = bsw_rhs(fe_from_u256(e.w));
  return (e.fl & BSW_INF) == 0u && !u256_ge(e.w, vrfk::Q32);
}
// start: first counter to try (k_tai_find's hint: every smaller counter is known not to give a point).  The neutral
// element comes back when all 256 attempts fail.
VRF_HD PtE bsw_hash_to_curve_tai(const uint8_t* msg, uint32_t msg_len, const SqrtTables& T, uint32_t start = 0) {
  PtE res = te_identity();
  bool done = false;
#pragma unroll 1
  for (uint32_t ctr = start; ctr < 256 && !done; ++ctr) {
    const Enc33 e = bsw_tai_candidate(msg, msg_len, ctr, T.str);
    FeN tx, ty;
    const bool ok = bsw_decode<BswS>(tx, ty, e, T);
    PtE p = te_from_affine(tx, ty);
#pragma unroll 1
    for (int i = 0; i < BswS::COFACTOR_LOG2; ++i) p = te_dbl<BswS>(p, true);
    const bool is_id = fe_is_zero(p.X) && fe_eq(p.Y, p.Z);
    if (ok && !is_id) {
      res = p;
      done = true;
    }
  }
  return res;
}

VRF_NS_END
