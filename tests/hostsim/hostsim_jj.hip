// tests/hostsim -- TEST TOOLING ONLY: host build of the device headers for the JubJub suite.
#include "../../ark_ec_vrfs_amd/csrc/vrf_core.cuh"
#include <cstring>
#include <vector>
using namespace vrf;
typedef SuiteJJ SJ;
namespace {
static SuiteStr hj_make_str(const uint8_t* id, uint32_t id_len) {
  SuiteStr s{};
  s.challenge_len = 32;
  s.suite_id_len = id_len;
  for (uint32_t i = 0; i < id_len; ++i) s.suite_id_w[i >> 3] |= (uint64_t)id[i] << (56 - 8 * (i & 7));
  return s;
}
SuiteStr g_hj_str = hj_make_str((const uint8_t*)"JubJub_SHA-512_TAI", 18);
struct HostTablesJ {
  std::vector<uint32_t> g_win, g_comb, b_comb;
  DevTables t;
  HostTablesJ() { build(SJ::gx(), SJ::gy(), SJ::bx(), SJ::by()); }
  void build(const FeN& gx, const FeN& gy, const FeN& bx, const FeN& by) {
    g_win.assign(2 * WIN_TABLE_WORDS, 0);
    build_glv_tables<SJ>(g_win.data(), gx, gy);
    g_comb.assign(GCOMB_WORDS, 0); b_comb.assign(GCOMB_WORDS, 0);
    std::vector<uint32_t> prefix((size_t)GC_SEG * NL);
    for (int which = 0; which < 2; ++which)
    for (int w = 0; w < GC_ROWS; ++w)
    for (int seg = 0; seg < GC_SEGS; ++seg)       // the device's own table builder (k_init_gcomb runs it per lane)
      gcomb_build_segment<SJ>(which ? b_comb.data() : g_comb.data(), prefix.data(),
                              which ? bx : gx, which ? by : gy, w, seg);
    t.sq.P = vrfk_tables::SQRT_P; t.sq.lut = vrfk_tables::SQRT_LUT; t.sq.str = g_hj_str;
    t.g_win = g_win.data(); t.g_comb = g_comb.data(); t.b_comb = b_comb.data();
  }
};
HostTablesJ& HJ() { static HostTablesJ h; h.t.sq.str = g_hj_str; return h; }
}
static uint32_t g_check_mask_jj = 0;   // CHK_* bits for the decode stages (0 = on-curve only)
extern "C" {
void hj_set_check_mask(uint32_t m) { g_check_mask_jj = m; }
void hj_init() { (void)HJ(); }
// decode + prime-order subgroup test (Tate pairing with the 8-torsion): 0 = in the subgroup, 2 = not decodable / not in it
int hj_decode_checked(const uint8_t* enc) {
  uint32_t w[8]; memcpy(w, enc, 32);
  DecodeA a = decode_phase_a<SJ>(w);
  FeN di = fe_inv(a.den);
  Fe<1, 4> x; bool ok = decode_phase_b<SJ>(x, a, di, HJ().t.sq);
  ok = ok && in_prime_subgroup<SJ>(fe_mul(x, fe_one()), a.y, HJ().t.sq);
  return ok ? 0 : 2;
}
// a descriptor for the host build: suite string + generator and blinding base (x || y little-endian)
void hj_configure(const uint8_t* id, uint32_t id_len, const uint8_t* g_xy, const uint8_t* b_xy) {
  g_hj_str = hj_make_str(id, id_len);
  auto in = [](const uint8_t* b) { uint32_t w[8]; memcpy(w, b, 32); return fe_from_u256(w); };
  HJ().build(in(g_xy), in(g_xy + 32), in(b_xy), in(b_xy + 32));
}
void hj_hash_to_curve(const uint8_t* msg, uint32_t len, uint8_t* out) {
  PtE h = data_to_point<SJ>(msg, len, HJ().t.sq);
  FeN x, y; te_to_affine(x, y, h);
  uint32_t e[8]; te_encode_affine(e, x, y, 0); memcpy(out, e, 32);
}
// out: gamma | c | s | pk | h  (IETF, pedersen = 0)  or  gamma | pk_com | r | ok | s | sb | blinding (pedersen = 1)
// first counter whose candidate decodes (the verdict k_tai_find uses), and hash-to-curve started from a hint
int hj_tai_first_decodable(const uint8_t* msg, uint32_t len) {
  for (uint32_t ctr = 0; ctr < 256; ++ctr)
    if (tai_attempt_decodes<SuiteJJ>(msg, len, ctr, HJ().t.sq)) return (int)ctr;
  return 255;
}
void hj_hash_to_curve_from(const uint8_t* msg, uint32_t len, uint32_t start, uint8_t* out) {
  PtE h = hash_to_curve_tai<SuiteJJ>(msg, len, HJ().t.sq, start);
  FeN x, y;
  te_to_affine(x, y, h);
  uint32_t e[8];
  te_encode_affine(e, x, y, 0);
  memcpy(out, e, 32);
}
int hj_prove(int pedersen, const uint8_t* sk, const uint8_t* msg, uint32_t len, const uint8_t* ad, uint32_t ad_len, uint8_t* out) {
  uint32_t skw[8]; memcpy(skw, sk, 32);
  uint32_t h_enc[8], k[8], kb[8], b[8], o[6][8], sb[8];
  std::vector<uint32_t> tab(ProveLayout<SJ>::TAB_WORDS), pts(PROVE_PTS_WORDS);
  bool valid = prove_prepare_item<SJ>(h_enc, k, tab.data(), HJ().t, skw, msg, len, nullptr);
  if (pedersen) { pedersen_blinding<SJ>(b, skw, h_enc, ad, ad_len, g_hj_str); nonce_rfc8032<SJ>(kb, b, h_enc); }
  prove_mul_item<SJ>(pts.data(), HJ().t, tab.data(), skw, pedersen ? b : nullptr);
  prove_mul_item<SJ>(pts.data() + 2 * UV_WORDS, HJ().t, tab.data(), k, pedersen ? kb : nullptr);
  prove_finish_item<SJ>(o[0], o[1], o[2], o[3], o[4], o[5], pts.data(), h_enc, skw, k, ad, ad_len, g_hj_str);
  if (pedersen) {
    uint32_t cb[8]; fr_mul<SJ>(cb, o[1], b); fr_add<SJ>(sb, cb, kb);
    memcpy(out, o[0], 32); memcpy(out + 32, o[3], 32); memcpy(out + 64, o[4], 32); memcpy(out + 96, o[5], 32);
    memcpy(out + 128, o[2], 32); memcpy(out + 160, sb, 32); memcpy(out + 192, b, 32);
  } else {
    memcpy(out, o[0], 32); memcpy(out + 32, o[1], 32); memcpy(out + 64, o[2], 32); memcpy(out + 96, o[3], 32); memcpy(out + 128, h_enc, 32);
  }
  return valid;
}
uint32_t hj_ietf_verify(const uint8_t* pk, const uint8_t* h, const uint8_t* g, const uint8_t* c, const uint8_t* s,
                        const uint8_t* ad, uint32_t ad_len) {
  uint32_t w[5][8];
  memcpy(w[0], pk, 32); memcpy(w[1], h, 32); memcpy(w[2], g, 32); memcpy(w[3], c, 32); memcpy(w[4], s, 32);
  std::vector<uint32_t> tabs(VERIFY_TABS * WIN_TABLE_WORDS), uv(2 * UV_WORDS);
  bool valid = verify_decode_item<SJ>(HJ().t, w[0], w[1], w[2], tabs.data(), g_check_mask_jj);
  uint32_t c2[8], s2[8];
  bool canon = fr_is_canonical<SJ>(w[3]) && fr_is_canonical<SJ>(w[4]);
  for (int j = 0; j < 8; ++j) { c2[j] = canon ? w[3][j] : 0; s2[j] = canon ? w[4][j] : 0; }
  verify_straus_item<SJ, 0>(uv.data(), HJ().t, tabs.data(), c2, s2);
  verify_straus_item<SJ, 1>(uv.data() + UV_WORDS, HJ().t, tabs.data(), c2, s2);
  return verify_finish_item<SJ>(uv.data(), w[0], w[1], w[2], w[3], w[4], valid, ad, ad_len, g_hj_str);
}
uint32_t hj_pedersen_verify(const uint8_t* h, const uint8_t* g, const uint8_t* proof160, const uint8_t* ad, uint32_t ad_len) {
  uint32_t enc[5][8], s[8], sb[8], c[8];
  memcpy(enc[0], h, 32); memcpy(enc[1], g, 32); memcpy(enc[2], proof160, 32); memcpy(enc[3], proof160 + 32, 32);
  memcpy(enc[4], proof160 + 64, 32); memcpy(s, proof160 + 96, 32); memcpy(sb, proof160 + 128, 32);
  std::vector<uint32_t> tabs(VERIFY_TABS * WIN_TABLE_WORDS), pts(PROVE_PTS_WORDS);
  bool valid = pedersen_verify_decode_item<SJ>(c, HJ().t, enc, ad, ad_len, tabs.data(), pts.data(), g_check_mask_jj);
  uint32_t s2[8], sb2[8];
  bool canon = fr_is_canonical<SJ>(s) && fr_is_canonical<SJ>(sb);
  for (int j = 0; j < 8; ++j) { s2[j] = canon ? s[j] : 0; sb2[j] = canon ? sb[j] : 0; }
  pedersen_verify_straus_item<SJ, 0>(pts.data(), HJ().t, tabs.data(), c, s2, sb2);
  pedersen_verify_straus_item<SJ, 1>(pts.data() + UV_WORDS, HJ().t, tabs.data(), c, s2, sb2);
  return pedersen_verify_finish_item<SJ>(pts.data(), s, sb, valid);
}
}
