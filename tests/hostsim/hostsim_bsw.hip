// tests/hostsim -- TEST TOOLING ONLY: host build of the codec-bound functions of `suites::bandersnatch_sw` (csrc/bsw_core.cuh):
// the 33-byte decode (square root, te_sw_map, 2-descent subgroup test), the projective -> wire encode, the transcript hashes
// over 33-byte encodings and try-and-increment -- the functions k_bsw.hip composes.  Never linked into libvrfhip.so.
#include "../../ark_ec_vrfs_amd/csrc/bsw_core.cuh"
#include <cstring>
using namespace vrf;
namespace {
SuiteStr make_str(const char* id, uint32_t challenge_len) {
  SuiteStr s{};
  s.challenge_len = challenge_len;
  s.suite_id_len = (uint32_t)strlen(id);
  for (uint32_t i = 0; i < s.suite_id_len; ++i) s.suite_id_w[i >> 3] |= (uint64_t)(uint8_t)id[i] << (56 - 8 * (i & 7));
  return s;
}
SqrtTables tables() {
  SqrtTables t;
  t.P = vrfk_tables::SQRT_P; t.lut = vrfk_tables::SQRT_LUT;
  t.str = make_str("Bandersnatch_SW_SHA-512_TAI", 32);
  return t;
}
FeN in(const uint8_t* b) { uint32_t w[8]; memcpy(w, b, 32); return fe_from_u256(w); }
template <int L, int V> void out(uint8_t* b, const Fe<L, V>& a) { uint32_t w[8]; fe_to_u256(w, a); memcpy(b, w, 32); }
}
extern "C" {
// bit 0: decodes; bit 1: the point at infinity; bit 2: in the prime-order subgroup.  te_xy / sw_xy: x || y little-endian
int hb_decode(const uint8_t* enc, uint8_t* te_xy, uint8_t* sw_xy) {
  const SqrtTables T = tables();
  FeN tx, ty, sx, sy;
  bool inf;
  const bool ok = bsw_decode<BswS>(tx, ty, sx, sy, inf, load33(enc, 0), T);
  out(te_xy, tx); out(te_xy + 32, ty); out(sw_xy, sx); out(sw_xy + 32, sy);
  const bool sub = ok && in_prime_subgroup<BswS>(tx, ty, T);
  return (ok ? 1 : 0) | (inf ? 2 : 0) | (sub ? 4 : 0);
}
// the Edwards point (x z : y z : z) -> wire
void hb_encode(const uint8_t* te_xy, const uint8_t* z_, uint8_t* enc) {
  const FeN x = in(te_xy), y = in(te_xy + 32), z = in(z_);
  PtE p;
  p.X = fe_mul(x, z); p.Y = fe_mul(y, z); p.Z = z; p.T = fe_mul(fe_mul(x, y), z);
  store33(enc, 0, bsw_encode<BswS>(p));
}
// n <= 8 points through the two-phase way in (ONE inversion for all of them, as k_bsw_verify_decode / k_bsw_rlc_decode run it):
// flags[i] as hb_decode bits 0 and 1, te_xy[i] the Edwards coordinates
void hb_decode_shared(const uint8_t* encs, int n, uint8_t* te_xy, uint8_t* flags) {
  const SqrtTables T = tables();
  uint32_t scr[8 * BSW_SLOT];
  uint32_t st[8];
  FeN run = fe_one();
  for (int p = 0; p < n; ++p) st[p] = bsw_in_a(scr + p * BSW_SLOT, run, load33(encs, p), T);
  FeN inv = fe_inv(run);
  for (int p = n - 1; p >= 0; --p) {
    FeN x, y;
    bsw_in_b(x, y, inv, scr + p * BSW_SLOT, (st[p] & BSW_A_INF) != 0);
    out(te_xy + 64 * p, x); out(te_xy + 64 * p + 32, y);
    flags[p] = (uint8_t)(((st[p] & BSW_A_OK) ? 1 : 0) | ((st[p] & BSW_A_INF) ? 2 : 0));
  }
}
void hb_canonical(const uint8_t* enc, uint8_t* o) { store33(o, 0, enc33_canonical(load33(enc, 0))); }
void hb_challenge(const uint8_t* pts, const uint8_t* ad, uint32_t ad_len, uint8_t* c) {
  Enc33 e[5];
  for (int i = 0; i < 5; ++i) e[i] = load33(pts, i);
  uint32_t w[8];
  bsw_challenge5(w, e, ad, ad_len, tables().str);
  memcpy(c, w, 32);
}
void hb_nonce(const uint8_t* sk, const uint8_t* h, uint8_t* k) {
  uint32_t s[8], w[8];
  memcpy(s, sk, 32);
  bsw_nonce(w, s, load33(h, 0));
  memcpy(k, w, 32);
}
void hb_blinding(const uint8_t* sk, const uint8_t* h, const uint8_t* ad, uint32_t ad_len, uint8_t* b) {
  uint32_t s[8], w[8];
  memcpy(s, sk, 32);
  bsw_blinding(w, s, load33(h, 0), ad, ad_len, tables().str);
  memcpy(b, w, 32);
}
void hb_output_hash(const uint8_t* g, uint8_t* o) {
  uint32_t w[16];
  bsw_output_hash(w, enc33_canonical(load33(g, 0)), tables().str);
  memcpy(o, w, 64);
}
// hash-to-curve; start = first counter tried.  ctr_hint (nullable) receives the first counter whose candidate passes the
// two halves of k_tai_find's test (the hint the kernels start from), 256 if none
void hb_hash_to_curve(const uint8_t* msg, uint32_t len, uint32_t start, uint8_t* enc, uint32_t* ctr_hint) {
  const SqrtTables T = tables();
  store33(enc, 0, bsw_encode<BswS>(bsw_hash_to_curve_tai(msg, len, T, start)));
  if (ctr_hint) {
    *ctr_hint = 256;
    for (uint32_t c = 0; c < 256; ++c) {
      FeN w;
      if (tai_attempt_candidate<SuiteBW>(w, msg, len, c, T) && fe_is_square_or_zero(w, T)) { *ctr_hint = c; break; }
    }
  }
}
}
