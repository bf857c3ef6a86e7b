// tests/hostsim -- TEST TOOLING ONLY.
// Compiles the __host__ __device__ arithmetic headers of ark_ec_vrfs_amd/csrc for the host so
// that the exact device source can be unit-tested on a machine without a GPU.  It is never
// linked into libvrfhip.so and nothing in the product path can reach it.
#include "../../ark_ec_vrfs_amd/csrc/vrf_core.cuh"
#include <cstring>
#include <vector>
using namespace vrf;

// suite strings of the host build (defaults: the built-in Bandersnatch descriptor); shared by the Bandersnatch units
static void hs_pack(uint64_t* w, const uint8_t* b, size_t n) {
  for (size_t i = 0; i < n; ++i) w[i >> 3] |= (uint64_t)b[i] << (56 - 8 * (i & 7));
}
static SuiteStr hs_make_str(const uint8_t* id, uint32_t id_len, const uint8_t* dst, uint32_t dst_len) {
  SuiteStr s{};
  s.challenge_len = 32;
  s.suite_id_len = id_len; hs_pack(s.suite_id_w, id, id_len);
  if (dst_len) {
    uint8_t dp[129]; memcpy(dp, dst, dst_len); dp[dst_len] = (uint8_t)dst_len;
    s.dst_prime_len = dst_len + 1; hs_pack(s.dst_prime_w, dp, dst_len + 1);
  }
  return s;
}
SuiteStr g_hs_str = hs_make_str((const uint8_t*)"Bandersnatch_SHA-512_ELL2", 25,
                                (const uint8_t*)"ECVRF_Bandersnatch_XMD:SHA-512_ELL2_RO_Bandersnatch_SHA-512_ELL2", 64);
static SqrtTables host_tables() {
  SqrtTables t; t.P = vrfk_tables::SQRT_P; t.lut = vrfk_tables::SQRT_LUT; t.str = g_hs_str; return t;
}
static FeN in(const uint8_t* b) { uint32_t w[8]; memcpy(w, b, 32); return fe_from_u256(w); }
template <int L, int V> static void out(uint8_t* b, const Fe<L, V>& a) { uint32_t w[8]; fe_to_u256(w, a); memcpy(b, w, 32); }
extern "C" {
// descriptor strings (vrfhip_suite_desc.suite_id / h2c_dst) for every Bandersnatch entry point of the host build
void hs_set_suite_strings(const uint8_t* id, uint32_t id_len, const uint8_t* dst, uint32_t dst_len) {
  g_hs_str = hs_make_str(id, id_len, dst, dst_len);
}
void hs_fe_mul(const uint8_t* a, const uint8_t* b, uint8_t* r) { out(r, fe_mul(in(a), in(b))); }
void hs_fe_sqr(const uint8_t* a, uint8_t* r) { out(r, fe_sqr(in(a))); }
void hs_fe_add(const uint8_t* a, const uint8_t* b, uint8_t* r) { out(r, fe_add(in(a), in(b))); }
void hs_fe_sub(const uint8_t* a, const uint8_t* b, uint8_t* r) { out(r, fe_sub(in(a), in(b))); }
void hs_fe_inv(const uint8_t* a, uint8_t* r) { out(r, fe_inv(in(a))); }
// stress the lazy bounds: ((a+b)*(a-b) + 5*a*b - b) * (a - (a*b + b)) etc.
void hs_fe_lazy(const uint8_t* a_, const uint8_t* b_, uint8_t* r) {
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
int hs_fe_sqrt(const uint8_t* a, uint8_t* r) {
  FeN root; bool sq = fe_sqrt_or_zsqrt(root, in(a), host_tables()); out(r, root); return sq;
}
// quadratic character by positive divsteps (fe.cuh): +1 / -1 / 0, 2 = rounds exhausted
int hs_fe_jacobi(const uint8_t* a) { FeN c = fe_canon(in(a)); return jacobi_limbs(c.v); }
int hs_fe_is_nonzero_square(const uint8_t* a) { return fe_is_nonzero_square(in(a), host_tables()) ? 1 : 0; }
int hs_fe_eq(const uint8_t* a, const uint8_t* b) { return fe_eq(in(a), fe_norm(fe_add(in(b), fe_zero()))); }
void hs_h2f(const uint8_t* msg, uint32_t len, uint8_t* u0, uint8_t* u1) {
  Fe<1,4> a, b; hash_to_field2<SuiteBS>(a, b, msg, len, g_hs_str); out(u0, a); out(u1, b);
}
void hs_sha512(const uint8_t* msg, uint32_t len, uint8_t* out) {
  Sha512 h; sha512_init(h); sha512_put_bytes(h, msg, len); sha512_final(h);
  for (int j = 0; j < 16; ++j) { uint32_t w = sha512_word_mem(h, j); memcpy(out + 4 * j, &w, 4); }
}
void hs_output_hash(const uint8_t* g, uint8_t* out) {
  uint32_t gw[8], o[16]; memcpy(gw, g, 32); output_hash_item<SuiteBS>(o, gw, g_hs_str); memcpy(out, o, 64);
}
void hs_secret_from_seed(const uint8_t* seed, uint32_t len, uint8_t* out) {
  uint32_t sk[8]; secret_from_seed_item<SuiteBS>(sk, seed, len); memcpy(out, sk, 32);
}
// decode + subgroup test by 2-descent: 0 = in the prime-order subgroup, 2 = not decodable or not in it
int hs_decode_checked(const uint8_t* enc) {
  uint32_t w[8]; memcpy(w, enc, 32);
  DecodeA a = decode_phase_a<SuiteBS>(w);
  FeN di = fe_inv(a.den);
  Fe<1,4> xx; bool ok = decode_phase_b<SuiteBS>(xx, a, di, host_tables());
  ok = ok && subgroup_by_2descent<SuiteBS>(a.y, host_tables());
  return ok ? 0 : 2;
}
// key-set comb row w of the point (x, y): entry j-1 = j * 256^w * P as affine x || y (64 bytes each, 255 entries)
void hs_comb_row(const uint8_t* xb, const uint8_t* yb, int w, uint8_t* out) {
  std::vector<uint32_t> row(COMB_COLS * PTA_WORDS), prefix(COMB_COLS * NL);
  comb_build_row<SuiteBS>(row.data(), prefix.data(), in(xb), in(yb), w);
  for (int j = 0; j < COMB_COLS; ++j) {
    PtA a = pta_load(row.data() + (size_t)j * PTA_WORDS);
    ::out(out + 64 * j, a.x); ::out(out + 64 * j + 32, a.y);
    // the cached product d*x*y must match too
    FeN dt = fe_mul(fe_mul(a.x, a.y), SuiteBS::d());
    if (!fe_eq(dt, a.dt)) memset(out + 64 * j, 0xee, 64);
  }
}
int hs_decode(const uint8_t* enc, uint8_t* x, uint8_t* y) {
  uint32_t w[8]; memcpy(w, enc, 32);
  DecodeA a = decode_phase_a<SuiteBS>(w);
  FeN di = fe_inv(a.den);
  Fe<1,4> xx; bool ok = decode_phase_b<SuiteBS>(xx, a, di, host_tables());
  out(x, xx); out(y, a.y); return ok;
}
}
