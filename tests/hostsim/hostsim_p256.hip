// tests/hostsim -- TEST TOOLING ONLY: host build of the secp256r1 device headers (-DVRF_FIELD=3): the P-256 field, the
// complete short-Weierstrass law, SHA-256 / HMAC, the Sec1 codec, try-and-increment, the RFC 6979 nonce and the per-item
// prove / verify steps exactly as k_p256.hip composes them.  Never linked into libvrfhip.so.
#include "../../ark_ec_vrfs_amd/csrc/p256_core.cuh"
#include <cstring>
#include <vector>
using namespace vrf;
namespace {
SuiteStr make_str(const uint8_t* id, uint32_t id_len, uint32_t challenge_len) {
  SuiteStr s{};
  s.challenge_len = challenge_len;
  s.suite_id_len = id_len;
  for (uint32_t i = 0; i < id_len; ++i) s.suite_id_w[i >> 3] |= (uint64_t)id[i] << (56 - 8 * (i & 7));
  return s;
}
const uint8_t kId[1] = {0x01};
SuiteStr g_str = make_str(kId, 1, 16);
std::vector<uint32_t>& comb() {
  static std::vector<uint32_t> c;
  if (c.empty()) {
    c.assign(P256_COMB_WORDS, 0);
    for (int w = 0; w < P256_COMB_ROWS; ++w) p256_comb_build_row(c.data(), w, fe_const(vrfk::P256_GX_M), fe_const(vrfk::P256_GY_M));
  }
  return c;
}
std::vector<uint32_t> g_comb_b;          // comb of the Pedersen blinding base (hp_set_blinding_base)
FeN in(const uint8_t* b) { uint32_t w[8]; load_be256(w, b); return fe_from_u256(w); }
template <int L, int V> void out(uint8_t* b, const Fe<L, V>& a) { uint32_t w[8]; fe_to_u256(w, a); store_be256(b, w); }
// affine big-endian x || y (all zero = the point at infinity)
PtW pin(const uint8_t* xy) {
  bool z = true;
  for (int i = 0; i < 64; ++i) z = z && xy[i] == 0;
  return z ? sw_identity() : sw_from_affine(in(xy), in(xy + 32));
}
void pout(uint8_t* xy, const PtW& p) {
  if (fe_is_zero(p.Z)) { memset(xy, 0, 64); return; }
  FeN zi = fe_inv(p.Z);
  out(xy, fe_mul(p.X, zi)); out(xy + 32, fe_mul(p.Y, zi));
}
}
extern "C" {
void hp_set_suite(const uint8_t* id, uint32_t id_len, uint32_t challenge_len) { g_str = make_str(id, id_len, challenge_len); }
// ---- field (big-endian 32-byte integers) ----
void hp_fe_mul(const uint8_t* a, const uint8_t* b, uint8_t* r) { out(r, fe_mul(in(a), in(b))); }
void hp_fe_sqr(const uint8_t* a, uint8_t* r) { out(r, fe_sqr(in(a))); }
void hp_fe_sub(const uint8_t* a, const uint8_t* b, uint8_t* r) { out(r, fe_sub(in(a), in(b))); }
void hp_fe_inv(const uint8_t* a, uint8_t* r) { out(r, fe_inv(in(a))); }
// lazily accumulated value k * a (k <= 31, as a chain of additions) through the weak reduction
void hp_fe_wred_chain(const uint8_t* a, int k, uint8_t* r) {
  FeN x = in(a);
  Fe<7, 32> acc; for (int i = 0; i < NL; ++i) acc.v[i] = 0;
  for (int j = 0; j < k; ++j) { for (int i = 0; i < NL; ++i) acc.v[i] += x.v[i]; if ((j & 3) == 3) { auto t = fe_norm(acc); for (int i = 0; i < NL; ++i) acc.v[i] = t.v[i]; } }
  out(r, fe_wred(acc));
}
int hp_fe_sqrt(const uint8_t* a, uint8_t* r) { FeN root; SqrtTables none{}; bool sq = fe_sqrt_or_zsqrt(root, in(a), none); out(r, root); return sq; }
int hp_fe_jacobi(const uint8_t* a) { FeN c = fe_canon(in(a)); return jacobi_limbs(c.v); }
// ---- scalars mod n (big-endian) ----
void hp_fr_mul_add(const uint8_t* a, const uint8_t* b, const uint8_t* c, uint8_t* r) {
  uint32_t x[8], y[8], z[8], t[8], o[8];
  (void)p256_scalar_decode(x, a); (void)p256_scalar_decode(y, b); (void)p256_scalar_decode(z, c);
  fr_mul<CurveP256>(t, x, y); fr_add<CurveP256>(o, t, z); store_be256(r, o);
}
// ---- group law (affine big-endian x || y; all zero = infinity) ----
void hp_add(const uint8_t* a, const uint8_t* b, uint8_t* r) { pout(r, sw_add(pin(a), pin(b))); }
void hp_dbl(const uint8_t* a, uint8_t* r) { pout(r, sw_dbl(pin(a))); }
// 16 P through the Jacobian doubling run, from a point with Z = z (z = 0: the point at infinity as (0 : 1 : 0) scaled)
void hp_dbl4(const uint8_t* a, const uint8_t* z, uint8_t* r) {
  PtW p = pin(a); FeN u = in(z);
  p.X = fe_wred(fe_mul(p.X, u)); p.Y = fe_wred(fe_mul(p.Y, u)); p.Z = fe_wred(fe_mul(p.Z, u));
  pout(r, sw_dbl4(p));
}
// the same through non-trivial Z (both operands scaled)
void hp_add_scaled(const uint8_t* a, const uint8_t* b, const uint8_t* za, const uint8_t* zb, uint8_t* r) {
  PtW p = pin(a), q = pin(b); FeN u = in(za), v = in(zb);
  p.X = fe_mul(p.X, u); p.Y = fe_mul(p.Y, u); p.Z = fe_mul(p.Z, u);
  q.X = fe_mul(q.X, v); q.Y = fe_mul(q.Y, v); q.Z = fe_mul(q.Z, v);
  pout(r, sw_add(p, q));
}
void hp_mul(const uint8_t* k_be, const uint8_t* a, uint8_t* r) {
  std::vector<uint32_t> tab(SW_TABLE_WORDS);
  sw_build_table(tab.data(), 1, pin(a));
  uint32_t k[8]; load_be256(k, k_be);
  pout(r, sw_win_mul(tab.data(), 1, k, false));
}
// k * P through the prover's four tables (P, 2^64 P, 2^128 P, 2^192 P) and the folded, patched recoding
void hp_mul_quad(const uint8_t* k_be, const uint8_t* a, uint8_t* r) {
  std::vector<uint32_t> tab(SW_QUAD_TABLES * SW_TABLE_WORDS);
  sw_build_quad_tables(tab.data(), 1, in(a), in(a + 32));
  uint32_t k[8]; load_be256(k, k_be);
  pout(r, sw_quad_mul(tab.data(), 1, k));
}
void hp_mul_base(const uint8_t* k_be, uint8_t* r) { uint32_t k[8]; load_be256(k, k_be); pout(r, sw_comb_mul(comb().data(), k)); }
// ---- hashes ----
void hp_sha256(const uint8_t* m, uint32_t len, uint8_t* digest) {
  Sha256 s; sha256_init(s); sha256_put_bytes(s, m, len); sha256_final(s);
  uint32_t w[8]; sha256_be256(w, s); store_be256(digest, w);
}
void hp_hmac256(const uint8_t* key32, const uint8_t* m, uint32_t len, uint8_t* mac) {
  uint32_t k[8], o[8]; load_be256(k, key32);
  Sha256 s; hmac256_begin(s, k); sha256_put_bytes(s, m, len); hmac256_end(o, s, k); store_be256(mac, o);
}
// ---- codec / suite functions ----
int hp_decode(const uint8_t* enc33, uint8_t* xy) { FeN x, y; bool ok = sec1_decode(x, y, enc33); out(xy, x); out(xy + 32, y); return ok; }
int hp_hash_to_curve(const uint8_t* data, uint32_t len, uint8_t* enc33) {
  FeN x, y; uint32_t xw[8]; bool ok = p256_hash_to_curve(x, y, xw, data, len, g_str);
  if (ok) sec1_store(enc33, 2, xw); else memset(enc33, 0, 33);
  return ok;
}
void hp_nonce(const uint8_t* sk_be, const uint8_t* h33, uint8_t* k_be) {
  uint32_t sk[8], k[8], xw[8]; (void)p256_scalar_decode(sk, sk_be); load_be256(xw, h33 + 1);
  p256_nonce(k, sk, h33[0], xw); store_be256(k_be, k);
}
void hp_output_hash(const uint8_t* g33, uint8_t* beta) {
  uint32_t xw[8], o[8]; load_be256(xw, g33 + 1); p256_output_hash(o, g33[0], xw, g_str); store_be256(beta, o);
}
void hp_secret_from_seed(const uint8_t* seed, uint32_t len, uint8_t* sk_be) { uint32_t sk[8]; p256_secret_from_seed(sk, seed, len); store_be256(sk_be, sk); }
// ---- the per-item pipelines as the kernels stage them ----
// out: pk 33 | h 33 | gamma 33 | c 32 | s 32 | k 32 | u 33 | v 33
int hp_prove(const uint8_t* sk_be, const uint8_t* msg, uint32_t msg_len, const uint8_t* h_given, const uint8_t* ad, uint32_t ad_len, uint8_t* o) {
  uint32_t sk[8], k[8]; FeN hx, hy; Sec1W henc;
  bool ok = p256_prove_prepare_item(sk, k, hx, hy, henc, sk_be, msg, msg_len, h_given, g_str);
  if (!ok) return 0;
  std::vector<uint32_t> tab(SW_QUAD_TABLES * SW_TABLE_WORDS);
  sw_build_quad_tables(tab.data(), 1, hx, hy);
  const PtW res[4] = {sw_comb_mul(comb().data(), sk), sw_quad_mul(tab.data(), 1, sk), sw_comb_mul(comb().data(), k), sw_quad_mul(tab.data(), 1, k)};
  Sec1W pk, gamma; uint32_t c[8], s[8];
  p256_prove_finish_item(pk, gamma, c, s, res, henc, sk, k, ad, ad_len, g_str);
  sec1_store(o, pk.tag, pk.xw); sec1_store(o + 33, henc.tag, henc.xw); sec1_store(o + 66, gamma.tag, gamma.xw);
  store_be256(o + 99, c); store_be256(o + 131, s); store_be256(o + 163, k);
  Sec1W uv[2]; const PtW r2[2] = {res[2], res[3]}; sw_to_sec1<false>(uv, r2);
  sec1_store(o + 195, uv[0].tag, uv[0].xw); sec1_store(o + 228, uv[1].tag, uv[1].xw);
  return 1;
}
// 0 = ok, 1 = does not verify, 2 = invalid data
int hp_verify(const uint8_t* pk, const uint8_t* h, const uint8_t* gamma, const uint8_t* c_be, const uint8_t* s_be, const uint8_t* ad, uint32_t ad_len) {
  FeN x[3], y[3]; Sec1W enc[3]; uint32_t c[8], s[8];
  if (!p256_verify_decode_item(x, y, enc, c, s, pk, h, gamma, c_be, s_be, g_str.challenge_len)) return 2;
  std::vector<uint32_t> ty(SW_TABLE_WORDS), th(SW_TABLE_WORDS), tg(SW_TABLE_WORDS);
  sw_build_table(ty.data(), 1, sw_from_affine(x[0], y[0]));
  sw_build_table(th.data(), 1, sw_from_affine(x[1], y[1]));
  sw_build_table(tg.data(), 1, sw_from_affine(x[2], y[2]));
  PtW U = sw_comb_minus_win(comb().data(), ty.data(), 1, s, c, sw_challenge_windows(g_str));
  PtW V = sw_straus_sc(th.data(), tg.data(), 1, s, c, sw_challenge_windows(g_str));
  return p256_verify_finish_item(U, V, enc, c, ad, ad_len, g_str);
}
// ---- Pedersen, staged as k_p256.hip stages it ----
void hp_set_blinding_base(const uint8_t* xy) {       // affine big-endian x || y
  g_comb_b.assign(P256_COMB_WORDS, 0);
  for (int w = 0; w < P256_COMB_ROWS; ++w) p256_comb_build_row(g_comb_b.data(), w, in(xy), in(xy + 32));
}
// out: gamma 33 | pk_com 33 | r 33 | ok 33 | s 32 | sb 32 | blinding 32 | h 33
int hp_ped_prove(const uint8_t* sk_be, const uint8_t* msg, uint32_t msg_len, const uint8_t* ad, uint32_t ad_len, uint8_t* o) {
  uint32_t sk[8], k[8], b[8], kb[8]; FeN hx, hy; Sec1W henc;
  if (!p256_prove_prepare_item(sk, k, hx, hy, henc, sk_be, msg, msg_len, nullptr, g_str)) return 0;
  p256_blinding(b, sk, henc, ad, ad_len, g_str);
  p256_nonce(kb, b, henc.tag, henc.xw);
  std::vector<uint32_t> tab(SW_QUAD_TABLES * SW_TABLE_WORDS);
  sw_build_quad_tables(tab.data(), 1, hx, hy);
  const PtW res[4] = {sw_add(sw_comb_mul(comb().data(), sk), sw_comb_mul(g_comb_b.data(), b)), sw_quad_mul(tab.data(), 1, sk),
                      sw_add(sw_comb_mul(comb().data(), k), sw_comb_mul(g_comb_b.data(), kb)), sw_quad_mul(tab.data(), 1, k)};
  Sec1W enc[4]; uint32_t s[8], sb[8];
  p256_ped_prove_finish_item(enc, s, sb, res, henc, sk, k, b, kb, ad, ad_len, g_str);
  sec1_store(o, enc[1].tag, enc[1].xw); sec1_store(o + 33, enc[0].tag, enc[0].xw); sec1_store(o + 66, enc[2].tag, enc[2].xw);
  sec1_store(o + 99, enc[3].tag, enc[3].xw); store_be256(o + 132, s); store_be256(o + 164, sb); store_be256(o + 196, b);
  sec1_store(o + 228, henc.tag, henc.xw);
  return 1;
}
int hp_ped_verify(const uint8_t* h, const uint8_t* gamma, const uint8_t* pk_com, const uint8_t* r, const uint8_t* ok, const uint8_t* s_be,
                  const uint8_t* sb_be, const uint8_t* ad, uint32_t ad_len) {
  FeN x[5], y[5]; uint32_t c[8], s[8], sb[8];
  if (!p256_ped_verify_decode_item(x, y, c, s, sb, h, gamma, pk_com, r, ok, s_be, sb_be, ad, ad_len, g_str)) return 2;
  std::vector<uint32_t> th(SW_TABLE_WORDS), tg(SW_TABLE_WORDS), tp(SW_TABLE_WORDS);
  sw_build_table(th.data(), 1, sw_from_affine(x[0], y[0]));
  sw_build_table(tg.data(), 1, sw_from_affine(x[1], y[1]));
  sw_build_table(tp.data(), 1, sw_from_affine(x[2], y[2]));
  const int cw = sw_challenge_windows(g_str);
  const bool e1 = p256_ped_verify_eq_h(th.data(), tg.data(), 1, x[4], y[4], s, c, cw);
  const bool e2 = p256_ped_verify_eq_g(comb().data(), g_comb_b.data(), tp.data(), 1, x[3], y[3], s, sb, c, cw);
  return (e1 && e2) ? 0 : 1;
}
}
