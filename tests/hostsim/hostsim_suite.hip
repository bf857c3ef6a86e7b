// tests/hostsim -- TEST TOOLING ONLY: host build of the device headers for ONE base field's try-and-increment suite
// (-DVRF_FIELD=1: Ed25519 over 2^255 - 19, -DVRF_FIELD=2: Baby-JubJub over BN254 Fr).  Field arithmetic, square roots,
// the Jacobi symbol, decoding with the subgroup test, hash-to-curve and the per-item prove / verify pipelines exactly as
// the kernels compose them.  Never linked into libvrfhip.so.
#include "../../ark_ec_vrfs_amd/csrc/vrf_core.cuh"
#include <cstring>
#include <vector>
using namespace vrf;
#if VRF_FIELD == 1
typedef SuiteED SX;
static const char* kSuiteId = "Ed25519_SHA-512_TAI";
static const uint32_t kChallengeLen = 16;
#elif VRF_FIELD == 2
typedef SuiteBJ SX;
static const char* kSuiteId = "BabyJubJub_SHA-512_TAI";
static const uint32_t kChallengeLen = 32;
#else
#error "hostsim_suite.hip is for the fields added in round 3"
#endif
namespace {
static SuiteStr hx_make_str(const uint8_t* id, uint32_t id_len, uint32_t challenge_len, uint32_t flags) {
  SuiteStr s{};
  s.challenge_len = challenge_len;
  s.flags = flags;
  s.suite_id_len = id_len;
  for (uint32_t i = 0; i < id_len; ++i) s.suite_id_w[i >> 3] |= (uint64_t)id[i] << (56 - 8 * (i & 7));
  return s;
}
SuiteStr g_hx_str = hx_make_str((const uint8_t*)kSuiteId, (uint32_t)strlen(kSuiteId), kChallengeLen, 0);
struct HostTablesX {
  std::vector<uint32_t> g_win, g_comb, b_comb;
  DevTables t;
  HostTablesX() { build(SX::gx(), SX::gy(), SX::bx(), SX::by()); }
  void build(const FeN& gx, const FeN& gy, const FeN& bx, const FeN& by) {
    g_win.assign(2 * WIN_TABLE_WORDS, 0);
    build_glv_tables<SX>(g_win.data(), gx, gy);
    g_comb.assign(GCOMB_WORDS, 0); b_comb.assign(GCOMB_WORDS, 0);
    std::vector<uint32_t> prefix((size_t)GC_SEG * NL);
    for (int which = 0; which < 2; ++which)
    for (int w = 0; w < GC_ROWS; ++w)
    for (int seg = 0; seg < GC_SEGS; ++seg)       // the device's own table builder (k_init_gcomb runs it per lane)
      gcomb_build_segment<SX>(which ? b_comb.data() : g_comb.data(), prefix.data(),
                              which ? bx : gx, which ? by : gy, w, seg);
    t.sq.P = vrfk_tables::SQRT_P; t.sq.lut = vrfk_tables::SQRT_LUT; t.sq.str = g_hx_str;
    t.g_win = g_win.data(); t.g_comb = g_comb.data(); t.b_comb = b_comb.data();
  }
};
HostTablesX& HX() { static HostTablesX h; h.t.sq.str = g_hx_str; return h; }
static FeN in(const uint8_t* b) { uint32_t w[8]; memcpy(w, b, 32); return fe_from_u256(w); }
template <int L, int V> static void out(uint8_t* b, const Fe<L, V>& a) { uint32_t w[8]; fe_to_u256(w, a); memcpy(b, w, 32); }
}
static uint32_t g_check_mask_x = 0;   // CHK_* bits for the decode stages (0 = on-curve only)
extern "C" {
void hx_set_check_mask(uint32_t m) { g_check_mask_x = m; }
void hx_init() { (void)HX(); }
// ---- field layer ----
void hx_fe_mul(const uint8_t* a, const uint8_t* b, uint8_t* r) { out(r, fe_mul(in(a), in(b))); }
void hx_fe_sqr(const uint8_t* a, uint8_t* r) { out(r, fe_sqr(in(a))); }
void hx_fe_add(const uint8_t* a, const uint8_t* b, uint8_t* r) { out(r, fe_add(in(a), in(b))); }
void hx_fe_sub(const uint8_t* a, const uint8_t* b, uint8_t* r) { out(r, fe_sub(in(a), in(b))); }
void hx_fe_inv(const uint8_t* a, uint8_t* r) { out(r, fe_inv(in(a))); }
void hx_fe_inv_pow(const uint8_t* a, uint8_t* r) { out(r, fe_inv_pow(in(a))); }
// stress the lazy bounds: ((a+b)*(a-b) + 5*a*b - b) * (a - (a*b + b)) etc.
void hx_fe_lazy(const uint8_t* a_, const uint8_t* b_, uint8_t* r) {
  FeN a = in(a_), b = in(b_);
  auto s = fe_add(a, b);              // (2,4)
  auto d = fe_sub(a, b);              // (3,6)
  auto p = fe_mul(s, d);              // L 6
  auto ab = fe_mul(a, b);
  auto t = fe_add(fe_mul5(ab), p);    // (6,12)
  auto tn = fe_norm(t);
  auto u2 = fe_sub(tn, b);            // (3, 16)
  auto w = fe_sub(a, fe_add(ab, b));  // subtrahend L=2: (4, 2+8)
  out(r, fe_mul(fe_norm(u2), w));
}
// worst-case limb bounds through the product: operands as wide as the type system allows (L1 * L2 = 6, V = 64)
void hx_fe_wide(const uint8_t* a_, const uint8_t* b_, uint8_t* r) {
  FeN a = in(a_), b = in(b_);
  Fe<2, 60> x; Fe<3, 60> y;
  // x = 30 a (two lazy doublings of 15 a ...): built by repeated lazy additions so that limbs really grow
  auto a2 = fe_add(a, a);                 // (2,4)
  auto b3 = fe_add(fe_add(b, b), b);      // (3,6)
  for (int i = 0; i < NL; ++i) { x.v[i] = a2.v[i]; y.v[i] = b3.v[i]; }
  out(r, fe_mul(x, y));                   // = 6 a b
}
int hx_fe_sqrt(const uint8_t* a, uint8_t* r) {
  FeN root; bool sq = fe_sqrt_or_zsqrt(root, in(a), HX().t.sq); out(r, root); return sq;
}
int hx_fe_jacobi(const uint8_t* a) { FeN c = fe_canon(in(a)); return jacobi_limbs(c.v); }
int hx_fe_is_nonzero_square(const uint8_t* a) { return fe_is_nonzero_square(in(a), HX().t.sq) ? 1 : 0; }
void hx_from_u512(const uint8_t* w64, uint8_t* r) { uint32_t w[16]; memcpy(w, w64, 64); out(r, fe_from_u512(w)); }
// Montgomery-256 coordinate format (VRFHIP_FLAG_COORDS_MONT256): words -> canonical words and back
void hx_mont256_roundtrip(const uint8_t* w_in, uint8_t* canon_out, uint8_t* back_out) {
  uint32_t w[8], c[8]; memcpy(w, w_in, 32);
  FeN v = fe_from_abi(c, w, true); memcpy(canon_out, c, 32);
  uint32_t b[8]; fe_to_mont256(b, v); memcpy(back_out, b, 32);
}
// ---- curve layer ----
int hx_decode(const uint8_t* enc, uint8_t* x, uint8_t* y) {
  uint32_t w[8]; memcpy(w, enc, 32);
  DecodeA a = decode_phase_a<SX>(w);
  FeN di = fe_inv(a.den);
  Fe<1,4> xx; bool ok = decode_phase_b<SX>(xx, a, di, HX().t.sq);
  out(x, xx); out(y, a.y); return ok;
}
// decode + prime-order subgroup test: 0 = in the subgroup, 2 = not decodable / not in it
int hx_decode_checked(const uint8_t* enc) {
  uint32_t w[8]; memcpy(w, enc, 32);
  DecodeA a = decode_phase_a<SX>(w);
  FeN di = fe_inv(a.den);
  Fe<1, 4> x; bool ok = decode_phase_b<SX>(x, a, di, HX().t.sq);
  ok = ok && in_prime_subgroup<SX>(fe_mul(x, fe_one()), a.y, HX().t.sq);
  return ok ? 0 : 2;
}
// the suite's subgroup test and arkworks' own (r * P = O through the compiled group law) on an affine point: bit 0 / bit 1
int hx_subgroup_both(const uint8_t* x, const uint8_t* y) {
  const FeN xx = in(x), yy = in(y);
  return (in_prime_subgroup<SX>(xx, yy, HX().t.sq) ? 1 : 0) | (subgroup_by_order<SX>(xx, yy) ? 2 : 0);
}
// (x1, y1) + (x2, y2) and 2 (x1, y1) through the extended-coordinate laws, affine out
void hx_point_add(const uint8_t* x1, const uint8_t* y1, const uint8_t* x2, const uint8_t* y2, uint8_t* ox, uint8_t* oy) {
  PtE p = te_from_affine(in(x1), in(y1)), q = te_from_affine(in(x2), in(y2));
  PtE r = te_add<SX>(p, q);
  FeN x, y; te_to_affine(x, y, r); out(ox, x); out(oy, y);
}
void hx_point_dbl(const uint8_t* x1, const uint8_t* y1, uint8_t* ox, uint8_t* oy) {
  PtE r = te_dbl<SX>(te_from_affine(in(x1), in(y1)), true);
  FeN x, y; te_to_affine(x, y, r); out(ox, x); out(oy, y);
  // the T coordinate must be consistent: T Z = X Y
  if (!fe_eq(fe_mul(r.T, r.Z), fe_mul(r.X, r.Y))) memset(ox, 0xee, 32);
}
// k * (x, y) by the window table the verifiers use
void hx_scalar_mul(const uint8_t* k, const uint8_t* xb, const uint8_t* yb, uint8_t* ox, uint8_t* oy) {
  std::vector<uint32_t> tab(2 * WIN_TABLE_WORDS);
  build_glv_tables<SX>(tab.data(), in(xb), in(yb));
  uint32_t kw[8]; memcpy(kw, k, 32);
  PtE r = var_base_mul<SX>(tab.data(), kw);
  FeN x, y; te_to_affine(x, y, r); out(ox, x); out(oy, y);
}
// a descriptor for the host build: suite string, challenge length, flags, generator and blinding base (x || y little-endian)
void hx_configure(const uint8_t* id, uint32_t id_len, uint32_t challenge_len, uint32_t flags, const uint8_t* g_xy,
                  const uint8_t* b_xy) {
  g_hx_str = hx_make_str(id, id_len, challenge_len, flags);
  HX().build(in(g_xy), in(g_xy + 32), in(b_xy), in(b_xy + 32));
}
void hx_hash_to_curve(const uint8_t* msg, uint32_t len, uint8_t* outp) {
  PtE h = data_to_point<SX>(msg, len, HX().t.sq);
  FeN x, y; te_to_affine(x, y, h);
  uint32_t e[8]; te_encode_affine(e, x, y, g_hx_str.flags); memcpy(outp, e, 32);
}
int hx_tai_first_decodable(const uint8_t* msg, uint32_t len) {
  for (uint32_t ctr = 0; ctr < 256; ++ctr)
    if (tai_attempt_decodes<SX>(msg, len, ctr, HX().t.sq)) return (int)ctr;
  return 255;
}
void hx_output_hash(const uint8_t* g, uint8_t* outp) {
  uint32_t gw[8], o[16]; memcpy(gw, g, 32);
  enc_canonical(gw);
  if (g_hx_str.flags & SS_HASH_COFACTOR) {       // as k_output_hash
    uint32_t e[8]; (void)output_cofactor_encoding<SX>(e, gw, HX().t.sq); memcpy(gw, e, 32);
  }
  output_hash_item<SX>(o, gw, g_hx_str); memcpy(outp, o, 64);
}
void hx_secret_from_seed(const uint8_t* seed, uint32_t len, uint8_t* outp) {
  uint32_t sk[8]; secret_from_seed_item<SX>(sk, seed, len); memcpy(outp, sk, 32);
}
void hx_public(const uint8_t* sk, uint8_t* outp) {
  uint32_t skw[8], pk[8]; memcpy(skw, sk, 32); public_from_secret_item<SX>(pk, HX().t, skw); memcpy(outp, pk, 32);
}
// out: gamma | c | s | pk | h  (IETF, pedersen = 0)  or  gamma | pk_com | r | ok | s | sb | blinding (pedersen = 1)
int hx_prove(int pedersen, const uint8_t* sk, const uint8_t* msg, uint32_t len, const uint8_t* ad, uint32_t ad_len, uint8_t* outp) {
  uint32_t skw[8]; memcpy(skw, sk, 32);
  uint32_t h_enc[8], k[8], kb[8], b[8], o[6][8], sb[8];
  std::vector<uint32_t> tab(ProveLayout<SX>::TAB_WORDS), pts(PROVE_PTS_WORDS);
  bool valid = prove_prepare_item<SX>(h_enc, k, tab.data(), HX().t, skw, msg, len, nullptr);
  if (pedersen) { pedersen_blinding<SX>(b, skw, h_enc, ad, ad_len, g_hx_str); nonce_rfc8032<SX>(kb, b, h_enc); }
  prove_mul_item<SX>(pts.data(), HX().t, tab.data(), skw, pedersen ? b : nullptr);
  prove_mul_item<SX>(pts.data() + 2 * UV_WORDS, HX().t, tab.data(), k, pedersen ? kb : nullptr);
  prove_finish_item<SX>(o[0], o[1], o[2], o[3], o[4], o[5], pts.data(), h_enc, skw, k, ad, ad_len, g_hx_str);
  if (pedersen) {
    uint32_t cb[8]; fr_mul<SX>(cb, o[1], b); fr_add<SX>(sb, cb, kb);
    memcpy(outp, o[0], 32); memcpy(outp + 32, o[3], 32); memcpy(outp + 64, o[4], 32); memcpy(outp + 96, o[5], 32);
    memcpy(outp + 128, o[2], 32); memcpy(outp + 160, sb, 32); memcpy(outp + 192, b, 32);
  } else {
    memcpy(outp, o[0], 32); memcpy(outp + 32, o[1], 32); memcpy(outp + 64, o[2], 32); memcpy(outp + 96, o[3], 32); memcpy(outp + 128, h_enc, 32);
  }
  return valid;
}
uint32_t hx_ietf_verify(const uint8_t* pk, const uint8_t* h, const uint8_t* g, const uint8_t* c, const uint8_t* s,
                        const uint8_t* ad, uint32_t ad_len) {
  uint32_t w[5][8];
  memcpy(w[0], pk, 32); memcpy(w[1], h, 32); memcpy(w[2], g, 32); memcpy(w[3], c, 32); memcpy(w[4], s, 32);
  std::vector<uint32_t> tabs(VERIFY_TABS * WIN_TABLE_WORDS), uv(2 * UV_WORDS);
  bool valid = verify_decode_item<SX>(HX().t, w[0], w[1], w[2], tabs.data(), g_check_mask_x);
  uint32_t c2[8], s2[8], cr[8];
  fr_reduce256<SX>(cr, w[3]);                    // as k_verify.hip load_cs_reduced
  bool canon = fr_is_canonical<SX>(w[4]);
  for (int j = 0; j < 8; ++j) { c2[j] = canon ? cr[j] : 0; s2[j] = canon ? w[4][j] : 0; }
  verify_straus_item<SX, 0>(uv.data(), HX().t, tabs.data(), c2, s2);
  verify_straus_item<SX, 1>(uv.data() + UV_WORDS, HX().t, tabs.data(), c2, s2);
  return verify_finish_item<SX>(uv.data(), w[0], w[1], w[2], w[3], w[4], valid, ad, ad_len, g_hx_str);
}
uint32_t hx_pedersen_verify(const uint8_t* h, const uint8_t* g, const uint8_t* proof160, const uint8_t* ad, uint32_t ad_len) {
  uint32_t enc[5][8], s[8], sb[8], c[8];
  memcpy(enc[0], h, 32); memcpy(enc[1], g, 32); memcpy(enc[2], proof160, 32); memcpy(enc[3], proof160 + 32, 32);
  memcpy(enc[4], proof160 + 64, 32); memcpy(s, proof160 + 96, 32); memcpy(sb, proof160 + 128, 32);
  std::vector<uint32_t> tabs(VERIFY_TABS * WIN_TABLE_WORDS), pts(PROVE_PTS_WORDS);
  bool valid = pedersen_verify_decode_item<SX>(c, HX().t, enc, ad, ad_len, tabs.data(), pts.data(), g_check_mask_x);
  uint32_t s2[8], sb2[8];
  bool canon = fr_is_canonical<SX>(s) && fr_is_canonical<SX>(sb);
  for (int j = 0; j < 8; ++j) { s2[j] = canon ? s[j] : 0; sb2[j] = canon ? sb[j] : 0; }
  pedersen_verify_straus_item<SX, 0>(pts.data(), HX().t, tabs.data(), c, s2, sb2);
  pedersen_verify_straus_item<SX, 1>(pts.data() + UV_WORDS, HX().t, tabs.data(), c, s2, sb2);
  return pedersen_verify_finish_item<SX>(pts.data(), s, sb, valid);
}
}
