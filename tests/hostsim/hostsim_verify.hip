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
  std::vector<uint32_t> g_win, g_comb;
  DevTables t;
  HostTables() { build(SuiteBS::gx(), SuiteBS::gy()); }
  void build(const FeN& gx, const FeN& gy) {
    g_win.assign(2 * WIN_TABLE_WORDS, 0);
    build_glv_tables<SuiteBS>(g_win.data(), gx, gy);
    g_comb.assign(GCOMB_WORDS, 0);
    std::vector<uint32_t> prefix((size_t)GC_SEG * NL);
    for (int w = 0; w < GC_ROWS; ++w)
    for (int seg = 0; seg < GC_SEGS; ++seg)       // the device's own table builder (k_init_gcomb runs it per lane)
      gcomb_build_segment<SuiteBS>(g_comb.data(), prefix.data(), gx, gy, w, seg);
    t.sq = host_tables(); t.g_win = g_win.data(); t.g_comb = g_comb.data(); t.b_comb = nullptr;
  }
};
HostTables& HT() { static HostTables h; h.t.sq.str = g_hs_str; return h; }      // strings may change between calls
}
static uint32_t g_check_mask = 0;      // CHK_* bits applied by the decode stages below (0 = on-curve only)
extern "C" {
void hs_set_check_mask(uint32_t m) { g_check_mask = m; }
void hs_set_generator_verify(const uint8_t* g_xy) { HT().build(in(g_xy), in(g_xy + 32)); }
uint32_t hs_ietf_verify(const uint8_t* pk, const uint8_t* h, const uint8_t* g, const uint8_t* c,
                        const uint8_t* s, const uint8_t* ad, uint32_t ad_len) {
  uint32_t w[5][8];
  memcpy(w[0], pk, 32); memcpy(w[1], h, 32); memcpy(w[2], g, 32); memcpy(w[3], c, 32); memcpy(w[4], s, 32);
  std::vector<uint32_t> tabs(VERIFY_TABS * WIN_TABLE_WORDS), uv(2 * UV_WORDS);
  bool valid = verify_decode_item<SuiteBS>(HT().t, w[0], w[1], w[2], tabs.data(), g_check_mask);
  verify_straus_item<SuiteBS, 0>(uv.data(), HT().t, tabs.data(), w[3], w[4]);
  verify_straus_item<SuiteBS, 1>(uv.data() + UV_WORDS, HT().t, tabs.data(), w[3], w[4]);
  return verify_finish_item<SuiteBS>(uv.data(), w[0], w[1], w[2], w[3], w[4], valid, ad, ad_len, g_hs_str);
}
// GLV pieces: k -> (k1, k2) as 2 x (16 B magnitude, 1 B sign) ; psi(P) encoded
void hs_glv_decompose(const uint8_t* k, uint8_t* out34) {
  uint32_t kw[8]; memcpy(kw, k, 32);
  GlvHalf a, b; glv_decompose_bs(a, b, kw);
  memcpy(out34, a.mag, 16); out34[16] = a.neg; memcpy(out34 + 17, b.mag, 16); out34[33] = b.neg;
}
int hs_psi(const uint8_t* enc, uint8_t* out) {
  uint32_t w[8]; memcpy(w, enc, 32);
  DecodeA a = decode_phase_a<SuiteBS>(w);
  FeN di = fe_inv(a.den);
  Fe<1,4> x; bool ok = decode_phase_b<SuiteBS>(x, a, di, HT().t.sq);
  PtE q = te_psi<SuiteBS>(te_from_affine(x, a.y));
  FeN qx, qy; te_to_affine(qx, qy, q);
  uint32_t e[8]; te_encode_affine(e, qx, qy, 0); memcpy(out, e, 32);
  return ok;
}
uint32_t hs_ietf_verify_affine(const uint8_t* pk_xy, const uint8_t* h_xy, const uint8_t* g_xy, const uint8_t* c,
                               const uint8_t* s, const uint8_t* ad, uint32_t ad_len) {
  uint32_t xy[3][16], enc[3][8], cw[8], sw[8];
  memcpy(xy[0], pk_xy, 64); memcpy(xy[1], h_xy, 64); memcpy(xy[2], g_xy, 64); memcpy(cw, c, 32); memcpy(sw, s, 32);
  std::vector<uint32_t> tabs(VERIFY_TABS * WIN_TABLE_WORDS), uv(2 * UV_WORDS);
  bool valid = verify_decode_affine_item<SuiteBS>(enc, xy, tabs.data(), HT().t.sq, g_check_mask);
  verify_straus_item<SuiteBS, 0>(uv.data(), HT().t, tabs.data(), cw, sw);
  verify_straus_item<SuiteBS, 1>(uv.data() + UV_WORDS, HT().t, tabs.data(), cw, sw);
  return verify_finish_item<SuiteBS>(uv.data(), enc[0], enc[1], enc[2], cw, sw, valid, ad, ad_len, g_hs_str);
}
// the K-proofs-per-lane pipeline exactly as the kernels run it (decode_multi -> straus -> finish_multi)
void hs_ietf_verify_multi(uint32_t n, const uint8_t* pk, const uint8_t* h, const uint8_t* g, const uint8_t* c,
                          const uint8_t* s, const uint8_t* ad, uint32_t ad_len, uint8_t* status) {
  std::vector<uint32_t> tabs((size_t)n * VERIFY_TABS * WIN_TABLE_WORDS), pts((size_t)n * PROVE_PTS_WORDS);
  std::vector<uint8_t> flags(n);
  for (size_t first = 0; first < n; first += VERIFY_K)
    verify_decode_multi<SuiteBS>(VERIFY_K, HT().t, first, n, pk, h, g, tabs.data(), pts.data(), flags.data(), g_check_mask);
  for (size_t i = 0; i < n; ++i) {
    uint32_t cw[8], sw[8]; memcpy(cw, c + 32 * i, 32); memcpy(sw, s + 32 * i, 32);
    if (!fr_is_canonical<SuiteBS>(cw) || !fr_is_canonical<SuiteBS>(sw)) { memset(cw, 0, 32); memset(sw, 0, 32); }
    const uint32_t* t = tabs.data() + i * VERIFY_TABS * WIN_TABLE_WORDS;
    verify_straus_item<SuiteBS, 1>(pts.data() + i * PROVE_PTS_WORDS + UV_WORDS, HT().t, t, cw, sw);
    verify_straus_item<SuiteBS, 0>(pts.data() + i * PROVE_PTS_WORDS, HT().t, t, cw, sw);
  }
  BytesViewLite adv; adv.blob = ad; adv.off = nullptr; adv.len = ad_len; adv.stride = 0;
  for (size_t first = 0; first < n; first += VERIFY_K)
    verify_finish_multi<SuiteBS>(VERIFY_K, first, n, pts.data(), PROVE_PTS_WORDS, pk, h, g, nullptr, 0, c, s, adv, flags.data(), status, g_hs_str);
}
}

// ---- MSM digit recoding and batched-verification weights (msm.cuh) ----
#include "../../ark_ec_vrfs_amd/csrc/msm.cuh"
#include "../../ark_ec_vrfs_amd/csrc/digest.cuh"
extern "C" {
void hs_msm_digits(int suite, const uint8_t* k, int negate, int zero, int16_t* out23) {
  uint32_t w[8]; memcpy(w, k, 32);
  int16_t d[MSM_W];
  if (suite == 2) msm_write_digits<SuiteJJ>(d, 1, 0, w, negate != 0, zero != 0);
  else msm_write_digits<SuiteBS>(d, 1, 0, w, negate != 0, zero != 0);
  memcpy(out23, d, sizeof d);
}
void hs_rlc_weights(const uint8_t* seed, const uint8_t* root, uint64_t index, uint8_t* z32, uint8_t* zp32) {
  uint32_t z[8], zp[8];
  rlc_weights<SuiteBS>(z, zp, seed, root, index);
  memcpy(z32, z, 32); memcpy(zp32, zp, 32);
}
// digest.cuh on the host: leaves then 128-ary node levels, exactly the kernels' loop (k_digest.hip)
void hs_batch_digest(uint64_t n, uint64_t index0, int n_arr, const uint8_t* const* arrs, const uint32_t* widths,
                     const uint8_t* ad, const uint32_t* ad_off, uint32_t ad_len, uint8_t* root) {
  DigestSrc s{};
  s.n_arr = n_arr;
  for (int j = 0; j < n_arr; ++j) { s.p[j] = arrs[j]; s.w[j] = widths[j]; }
  if (ad) { s.ad.blob = ad; s.ad.off = ad_off; s.ad.len = ad_len; s.ad.stride = 0; }
  std::vector<uint8_t> level(n * 32), next;
  for (uint64_t i = 0; i < n; ++i) digest_leaf(level.data() + 32 * i, s, i, index0 + i);
  uint64_t m = n;
  do {
    uint64_t nodes = (m + DIGEST_FAN - 1) / DIGEST_FAN;
    next.assign(nodes * 32, 0);
    for (uint64_t t = 0; t < nodes; ++t) {
      uint64_t cnt = m - t * DIGEST_FAN < (uint64_t)DIGEST_FAN ? m - t * DIGEST_FAN : (uint64_t)DIGEST_FAN;
      digest_node(next.data() + 32 * t, level.data() + 32 * DIGEST_FAN * t, (uint32_t)cnt);
    }
    level.swap(next);
    m = nodes;
  } while (m > 1);
  memcpy(root, level.data(), 32);
}
uint64_t hs_rlc_index(int p, uint64_t n, uint64_t i) { return rlc_index(p, n, i); }
}

