// tests/hostsim -- TEST TOOLING ONLY.
// Compiles the __host__ __device__ arithmetic headers of ark_ec_vrfs_amd/csrc for the host so
// that the exact device source can be unit-tested on a machine without a GPU.  It is never
// linked into libvrfhip.so and nothing in the product path can reach it.
#include "../../ark_ec_vrfs_amd/csrc/vrf_core.cuh"
#include <cstring>
#include <vector>
using namespace vrf;

extern SuiteStr g_hs_str;      // hostsim_fe.hip
static SqrtTables host_tables() {
  SqrtTables t; t.P = vrfk_tables::SQRT_P; t.lut = vrfk_tables::SQRT_LUT; t.str = g_hs_str; return t;
}
static FeN in(const uint8_t* b) { uint32_t w[8]; memcpy(w, b, 32); return fe_from_u256(w); }
template <int L, int V> static void out(uint8_t* b, const Fe<L, V>& a) { uint32_t w[8]; fe_to_u256(w, a); memcpy(b, w, 32); }
namespace {
struct HostTables {
  std::vector<uint32_t> g_win, g_comb, b_comb;
  DevTables t;
  FeN gx, gy;
  HostTables() { build(SuiteBS::gx(), SuiteBS::gy(), SuiteBS::bx(), SuiteBS::by()); }
  // the tables a context builds from its descriptor's generator and blinding base (k_init_gwin / k_init_gcomb)
  void build(const FeN& gx_, const FeN& gy_, const FeN& bx, const FeN& by) {
    gx = gx_; gy = gy_;
    g_win.assign(2 * WIN_TABLE_WORDS, 0);
    build_glv_tables<SuiteBS>(g_win.data(), gx, gy);
    g_comb.assign(GCOMB_WORDS, 0); b_comb.assign(GCOMB_WORDS, 0);
    std::vector<uint32_t> prefix((size_t)GC_SEG * NL);
    for (int which = 0; which < 2; ++which)
    for (int w = 0; w < GC_ROWS; ++w)
    for (int seg = 0; seg < GC_SEGS; ++seg)       // the device's own table builder (k_init_gcomb runs it per lane)
      gcomb_build_segment<SuiteBS>(which ? b_comb.data() : g_comb.data(), prefix.data(),
                              which ? bx : gx, which ? by : gy, w, seg);
    t.sq = host_tables(); t.g_win = g_win.data(); t.g_comb = g_comb.data(); t.b_comb = b_comb.data();
  }
};
HostTables& HT() { static HostTables h; h.t.sq.str = g_hs_str; return h; }      // strings may change between calls
}
static uint32_t g_check_mask_p = 0;    // CHK_* bits for the decode stages (0 = on-curve only)
extern "C" {
void hs_set_check_mask_prove(uint32_t m) { g_check_mask_p = m; }
void hs_init() { (void)HT(); }
// descriptor points (x || y, 32-byte little-endian each) -> rebuild the fixed-base tables
void hs_set_bases(const uint8_t* g_xy, const uint8_t* b_xy) { HT().build(in(g_xy), in(g_xy + 32), in(b_xy), in(b_xy + 32)); }
// entry (w, j) of the generator table built by gcomb_build_segment against j * 2^(GCB w) * G by the ladder
int hs_comb_entry_check(int w, int j) {
  if (w >= GC_ROWS || j < 1 || j > GC_COLS) return -1;
  uint32_t k[8] = {0};
  const int bit = w * GCB;
  k[bit >> 5] = (uint32_t)j << (bit & 31);
  PtE p = te_mul_slow<SuiteBS>(te_from_affine(HT().gx, HT().gy), k);
  FeN zi = fe_inv(p.Z);
  FeN x = fe_mul(p.X, zi), y = fe_mul(p.Y, zi), dt = fe_mul(fe_mul(x, y), SuiteBS::d());
  const uint32_t* ref = HT().g_comb.data() + ((size_t)w * GC_COLS + (j - 1)) * PTA_WORDS;
  return fe_eq(x, fe_load<1, 2>(ref)) && fe_eq(y, fe_load<1, 2>(ref + NL)) && fe_eq(dt, fe_load<1, 2>(ref + 2 * NL));
}
int hs_gcomb_geometry(int* rows, int* cols) { *rows = GC_ROWS; *cols = GC_COLS; return GCB; }
void hs_hash_to_curve(const uint8_t* msg, uint32_t len, uint8_t* out) {
  PtE h = hash_to_curve_ell2<SuiteBS>(msg, len, HT().t.sq);
  FeN x, y; te_to_affine(x, y, h);
  uint32_t e[8]; te_encode_affine(e, x, y, 0); memcpy(out, e, 32);
}
static int prove_any(bool pedersen, const uint8_t* sk, const uint8_t* msg, uint32_t len, const uint8_t* h_given,
                     const uint8_t* ad, uint32_t ad_len, uint32_t o[6][8], uint32_t h_enc[8], uint32_t sb[8], uint32_t b[8]) {
  uint32_t skw[8]; memcpy(skw, sk, 32);
  uint32_t hg[8]; if (h_given) memcpy(hg, h_given, 32);
  uint32_t k[8], kb[8];
  std::vector<uint32_t> tab(ProveLayout<SuiteBS>::TAB_WORDS), pts(PROVE_PTS_WORDS);
  bool valid = prove_prepare_item<SuiteBS>(h_enc, k, tab.data(), HT().t, skw, msg, len, h_given ? hg : nullptr, 0, g_check_mask_p);
  if (pedersen) { pedersen_blinding<SuiteBS>(b, skw, h_enc, ad, ad_len, g_hs_str); nonce_rfc8032<SuiteBS>(kb, b, h_enc); }
  prove_mul_item<SuiteBS>(pts.data(), HT().t, tab.data(), skw, pedersen ? b : nullptr);
  prove_mul_item<SuiteBS>(pts.data() + 2 * UV_WORDS, HT().t, tab.data(), k, pedersen ? kb : nullptr);
  // o: gamma, c, s, pk, r, ok
  prove_finish_item<SuiteBS>(o[0], o[1], o[2], o[3], o[4], o[5], pts.data(), h_enc, skw, k, ad, ad_len, g_hs_str);
  if (pedersen) { uint32_t cb[8]; fr_mul<SuiteBS>(cb, o[1], b); fr_add<SuiteBS>(sb, cb, kb); }
  return valid;
}
int hs_ietf_prove(const uint8_t* sk, const uint8_t* msg, uint32_t len, const uint8_t* h_given,
                  const uint8_t* ad, uint32_t ad_len,
                  uint8_t* gamma, uint8_t* c, uint8_t* s, uint8_t* h, uint8_t* pk) {
  uint32_t o[6][8], h_enc[8], sb[8], b[8];
  int valid = prove_any(false, sk, msg, len, h_given, ad, ad_len, o, h_enc, sb, b);
  memcpy(gamma, o[0], 32); memcpy(c, o[1], 32); memcpy(s, o[2], 32); memcpy(h, h_enc, 32); memcpy(pk, o[3], 32);
  return valid;
}
// out: gamma | pk_com | r | ok | s | sb | blinding  (7 x 32 B)
int hs_pedersen_prove(const uint8_t* sk, const uint8_t* msg, uint32_t len, const uint8_t* ad, uint32_t ad_len,
                      uint8_t* out) {
  uint32_t o[6][8], h_enc[8], sb[8], b[8];
  int valid = prove_any(true, sk, msg, len, nullptr, ad, ad_len, o, h_enc, sb, b);
  memcpy(out, o[0], 32); memcpy(out + 32, o[3], 32); memcpy(out + 64, o[4], 32); memcpy(out + 96, o[5], 32);
  memcpy(out + 128, o[2], 32); memcpy(out + 160, sb, 32); memcpy(out + 192, b, 32);
  return valid;
}
void hs_public(const uint8_t* sk, uint8_t* pk) {
  uint32_t skw[8], o[8]; memcpy(skw, sk, 32); public_from_secret_item<SuiteBS>(o, HT().t, skw); memcpy(pk, o, 32);
}
uint32_t hs_pedersen_verify(const uint8_t* h, const uint8_t* g, const uint8_t* proof160, const uint8_t* ad, uint32_t ad_len) {
  uint32_t enc[5][8], s[8], sb[8], c[8];
  memcpy(enc[0], h, 32); memcpy(enc[1], g, 32); memcpy(enc[2], proof160, 32); memcpy(enc[3], proof160 + 32, 32);
  memcpy(enc[4], proof160 + 64, 32); memcpy(s, proof160 + 96, 32); memcpy(sb, proof160 + 128, 32);
  std::vector<uint32_t> tabs(VERIFY_TABS * WIN_TABLE_WORDS), pts(PROVE_PTS_WORDS);
  bool valid = pedersen_verify_decode_item<SuiteBS>(c, HT().t, enc, ad, ad_len, tabs.data(), pts.data(), g_check_mask_p);
  uint32_t s2[8], sb2[8];
  bool canon = fr_is_canonical<SuiteBS>(s) && fr_is_canonical<SuiteBS>(sb);
  for (int j = 0; j < 8; ++j) { s2[j] = canon ? s[j] : 0; sb2[j] = canon ? sb[j] : 0; }
  pedersen_verify_straus_item<SuiteBS, 0>(pts.data(), HT().t, tabs.data(), c, s2, sb2);
  pedersen_verify_straus_item<SuiteBS, 1>(pts.data() + UV_WORDS, HT().t, tabs.data(), c, s2, sb2);
  return pedersen_verify_finish_item<SuiteBS>(pts.data(), s, sb, valid);
}
// batch prove through the K-per-lane prepare, exactly as the kernels run it; out: n x (gamma | c | s | pk | h)
void hs_ietf_prove_multi(uint32_t n, const uint8_t* sk, const uint8_t* msgs, uint32_t msg_len, const uint8_t* ad,
                         uint32_t ad_len, uint8_t* out) {
  std::vector<uint32_t> tabs((size_t)n * ProveLayout<SuiteBS>::TAB_WORDS), pts((size_t)n * PROVE_PTS_WORDS), aux((size_t)n * 32);
  std::vector<uint8_t> flags(n);
  BytesViewLite mv; mv.blob = msgs; mv.off = nullptr; mv.len = msg_len; mv.stride = msg_len;
  for (size_t first = 0; first < n; first += PROVE_K)
    prove_prepare_multi<SuiteBS>(PROVE_K, HT().t, first, n, sk, mv, tabs.data(), pts.data(), aux.data(), 32, flags.data());
  for (size_t i = 0; i < n; ++i) {
    uint32_t skw[8], k[8];
    memcpy(skw, sk + 32 * i, 32); memcpy(k, aux.data() + 32 * i + 8, 32);
    prove_mul_item<SuiteBS>(pts.data() + i * PROVE_PTS_WORDS, HT().t, tabs.data() + i * ProveLayout<SuiteBS>::TAB_WORDS, skw, nullptr);
    prove_mul_item<SuiteBS>(pts.data() + i * PROVE_PTS_WORDS + 2 * UV_WORDS, HT().t, tabs.data() + i * ProveLayout<SuiteBS>::TAB_WORDS, k, nullptr);
  }
  for (size_t first = 0; first < n; first += PROVE_K)
    prove_encode_multi<SuiteBS>(PROVE_K, first, n, pts.data(), tabs.data(), ProveLayout<SuiteBS>::TAB_WORDS, 0);
  for (size_t i = 0; i < n; ++i) {
    uint32_t skw[8], h_enc[8], k[8], c[8], s2[8];
    memcpy(skw, sk + 32 * i, 32); memcpy(h_enc, aux.data() + 32 * i, 32); memcpy(k, aux.data() + 32 * i + 8, 32);
    const uint32_t* enc = tabs.data() + i * ProveLayout<SuiteBS>::TAB_WORDS + PROVE_ENC_OFF;
    prove_respond_item<SuiteBS>(c, s2, enc, h_enc, skw, k, ad, ad_len, g_hs_str);
    memcpy(out + 160 * i, enc, 32); memcpy(out + 160 * i + 32, c, 32); memcpy(out + 160 * i + 64, s2, 32);
    memcpy(out + 160 * i + 96, enc + 8, 32); memcpy(out + 160 * i + 128, h_enc, 32);
  }
}
}
