// tests/hostsim -- TEST TOOLING ONLY.
// Compiles the __host__ __device__ arithmetic headers of ark_ec_vrfs_amd/csrc for the host so
// that the exact device source can be unit-tested on a machine without a GPU.  It is never
// linked into libvrfhip.so and nothing in the product path can reach it.
#include "../../ark_ec_vrfs_amd/csrc/vrf_core.cuh"
#include <cstring>
#include <vector>
using namespace vrf;

static SqrtTables host_tables() {
  SqrtTables t; t.P = vrfk_tables::SQRT_P; t.lut = vrfk_tables::SQRT_LUT; return t;
}
static FeN in(const uint8_t* b) { uint32_t w[8]; memcpy(w, b, 32); return fe_from_u256(w); }
template <int L, int V> static void out(uint8_t* b, const Fe<L, V>& a) { uint32_t w[8]; fe_to_u256(w, a); memcpy(b, w, 32); }
namespace {
struct HostTables {
  std::vector<uint32_t> g_win, g_comb;
  DevTables t;
  HostTables() {
    g_win.resize(WIN_TABLE_WORDS);
    build_win_table<SuiteBS>(g_win.data(), SuiteBS::gx(), SuiteBS::gy());
    g_comb.resize((size_t)32 * 255 * PTA_WORDS);
    // comb by repeated addition instead of 8160 scalar multiplications (host is slow):
    for (int w = 0; w < 32; ++w) {
      uint32_t k[8] = {0}; k[w >> 2] = 1u << ((w & 3) * 8);
      PtE base = te_mul_slow<SuiteBS>(te_from_affine(SuiteBS::gx(), SuiteBS::gy()), k);
      PtC bc = te_to_cached<SuiteBS>(base);
      PtE acc = base;
      for (int j = 1; j <= 255; ++j) {
        FeN zi = fe_inv(acc.Z);
        PtA a; a.x = fe_mul(acc.X, zi); a.y = fe_mul(acc.Y, zi);
        a.dt = fe_mul(fe_mul(a.x, a.y), SuiteBS::d());
        pta_store(g_comb.data() + ((size_t)w * 255 + (j - 1)) * PTA_WORDS, a);
        acc = te_add_cached<SuiteBS>(acc, bc, false);
      }
    }
    t.sq = host_tables(); t.g_win = g_win.data(); t.g_comb = g_comb.data(); t.b_comb = nullptr;
  }
};
HostTables& HT() { static HostTables h; return h; }
}
extern "C" {
void hs_init() { (void)HT(); }
// single comb entry through the device init path (checks comb_entry against the additive build)
int hs_comb_entry_check(int w, int j) {
  uint32_t e[PTA_WORDS];
  comb_entry<SuiteBS>(e, SuiteBS::gx(), SuiteBS::gy(), w, j);
  // compare canonical values
  const uint32_t* ref = HT().g_comb.data() + ((size_t)w * 255 + (j - 1)) * PTA_WORDS;
  for (int c = 0; c < 3; ++c)
    if (!fe_eq(fe_load<1, 2>(e + c * NL), fe_load<1, 2>(ref + c * NL))) return 0;
  return 1;
}
void hs_hash_to_curve(const uint8_t* msg, uint32_t len, uint8_t* out) {
  PtE h = hash_to_curve_ell2<SuiteBS>(msg, len, HT().t.sq);
  FeN x, y; te_to_affine(x, y, h);
  uint32_t e[8]; te_encode_affine(e, x, y); memcpy(out, e, 32);
}
int hs_ietf_prove(const uint8_t* sk, const uint8_t* msg, uint32_t len, const uint8_t* h_given,
                  const uint8_t* ad, uint32_t ad_len,
                  uint8_t* gamma, uint8_t* c, uint8_t* s, uint8_t* h, uint8_t* pk) {
  uint32_t skw[8]; memcpy(skw, sk, 32);
  uint32_t hg[8]; if (h_given) memcpy(hg, h_given, 32);
  uint32_t h_enc[8], k[8], o[4][8];
  std::vector<uint32_t> tab(WIN_TABLE_WORDS), pts(PROVE_PTS_WORDS);
  bool valid = prove_prepare_item<SuiteBS>(h_enc, k, tab.data(), HT().t, skw, msg, len, h_given ? hg : nullptr);
  prove_mul_item<SuiteBS>(pts.data(), HT().t, tab.data(), skw);
  prove_mul_item<SuiteBS>(pts.data() + 2 * UV_WORDS, HT().t, tab.data(), k);
  prove_finish_item<SuiteBS>(o[0], o[1], o[2], o[3], pts.data(), h_enc, skw, k, ad, ad_len);
  memcpy(gamma, o[0], 32); memcpy(c, o[1], 32); memcpy(s, o[2], 32); memcpy(h, h_enc, 32); memcpy(pk, o[3], 32);
  return valid;
}
void hs_public(const uint8_t* sk, uint8_t* pk) {
  uint32_t skw[8], o[8]; memcpy(skw, sk, 32); public_from_secret_item<SuiteBS>(o, HT().t, skw); memcpy(pk, o, 32);
}
}
