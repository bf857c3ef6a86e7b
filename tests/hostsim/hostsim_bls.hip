// tests/hostsim -- TEST TOOLING ONLY: host build of the BLS12-381 pairing headers.
#include "../../ark_ec_vrfs_amd/csrc/bls12.cuh"
#include "../../ark_ec_vrfs_amd/csrc/g1.cuh"
#include <cstring>
#include <vector>
using namespace bls;
static FpS inw(const uint8_t* b) { uint32_t w[12]; memcpy(w, b, 48); FpS r; fp_from_words(r, w); return r; }
template <class A> static void outw(uint8_t* b, const A& a) { uint32_t w[12]; fp_to_words(w, a); memcpy(b, w, 48); }
static Fp2 in2(const uint8_t* b) { Fp2 r; r.a = inw(b); r.b = inw(b + 48); return r; }
static void out2(uint8_t* b, const Fp2& x) { outw(b, x.a); outw(b + 48, x.b); }
static void in12(Fp12* f, const uint8_t* b) {
  Fp2* c[6] = {&f->c0.c0, &f->c0.c1, &f->c0.c2, &f->c1.c0, &f->c1.c1, &f->c1.c2};
  for (int i = 0; i < 6; ++i) *c[i] = in2(b + 96 * i);
}
static void out12(uint8_t* b, const Fp12* f) {
  const Fp2* c[6] = {&f->c0.c0, &f->c0.c1, &f->c0.c2, &f->c1.c0, &f->c1.c1, &f->c1.c2};
  for (int i = 0; i < 6; ++i) out2(b + 96 * i, *c[i]);
}
extern "C" {
void hb_fp_mul(const uint8_t* a, const uint8_t* b, uint8_t* r) { outw(r, fp_mul(inw(a), inw(b))); }
void hb_fp_sqr(const uint8_t* a, uint8_t* r) { outw(r, fp_sqr(inw(a))); }
void hb_fp_lazy(const uint8_t* a_, const uint8_t* b_, uint8_t* r) {     // ((a-b)*(a+b) - 3ab) reduced, times (b - a - a)
  FpS a = inw(a_), b = inw(b_);
  auto d = fp_sub(a, b), s = fp_add(a, b);
  auto p = fp_mul(d, s);
  auto ab = fp_mul(a, b);
  auto t = fp_sub(p, fp_add(fp_dbl(ab), ab));
  auto u = fp_reduce(t);
  auto w = fp_sub(fp_sub(b, a), a);
  outw(r, fp_mul(u, fp_norm(w)));
}
void hb_fp_inv(const uint8_t* a, uint8_t* r) { FpS x = inw(a), y; fp_inv(&y, &x); outw(r, y); }
int hb_fp_eq(const uint8_t* a, const uint8_t* b) { return fp_eq(inw(a), fp_add(inw(b), fp_zero())); }
void hb_fp12_mul(const uint8_t* a, const uint8_t* b, uint8_t* r) { Fp12 x, y, z; in12(&x, a); in12(&y, b); fp12_mul(&z, &x, &y); out12(r, &z); }
void hb_fp12_sqr(const uint8_t* a, uint8_t* r) { Fp12 x, z; in12(&x, a); fp12_sqr(&z, &x); out12(r, &z); }
void hb_fp12_inv(const uint8_t* a, uint8_t* r) { Fp12 x, z; in12(&x, a); fp12_inv(&z, &x); out12(r, &z); }
void hb_fp12_frob(const uint8_t* a, uint8_t* r) { Fp12 x, z; in12(&x, a); fp12_frob(&z, &x); out12(r, &z); }
void hb_fp12_mul014(const uint8_t* a, const uint8_t* c, uint8_t* r) {
  Fp12 x; in12(&x, a); Fp2 c0 = in2(c), c1 = in2(c + 96), c4 = in2(c + 192); fp12_mul_by_014(&x, &c0, &c1, &c4); out12(r, &x);
}
// g1: 96 B, g2: 192 B -> Miller loop value (576 B) and final exponentiation (576 B)
int hb_pairing(const uint8_t* g1, const uint8_t* g2, uint8_t* ml, uint8_t* fe) {
  uint32_t w1[24], w2[48]; memcpy(w1, g1, 96); memcpy(w2, g2, 192);
  G1Aff P; G2Aff Q; bool i1, i2;
  bool ok = g1_load(P, i1, w1) & g2_load(Q, i2, w2);
  bool skip = i1 || i2;
  Fp12 f, e;
  miller_loop<1>(&f, &P, &Q, &skip);
  out12(ml, &f);
  final_exponentiation(&e, &f);
  out12(fe, &e);
  return ok;
}
// the shared-G2 route: lines from pairing_prepare_g2_pair, scaled per item the way the quad kernel does
// (c1 * x_P, c4 * y_P), absorbed in Miller-loop order
uint32_t hb_pairing_check2_prepared(const uint8_t* g1x2, const uint8_t* g2x2) {
  uint32_t w1[48], w2[96]; memcpy(w1, g1x2, 192); memcpy(w2, g2x2, 384);
  std::vector<uint32_t> prep(G2_PREP_WORDS);
  pairing_prepare_g2_pair(w2, prep.data(), 0);
  pairing_prepare_g2_pair(w2, prep.data(), 1);
  const uint32_t* flags = prep.data() + (size_t)2 * G2_LINES * G2_LINE_WORDS;
  G1Aff P[2];
  bool skip[2], ok = true;
  for (int i = 0; i < 2; ++i) {
    bool i1;
    ok = g1_load(P[i], i1, w1 + 24 * i) && ok;
    ok = ok && flags[2 * i] != 0;
    skip[i] = i1 || flags[2 * i + 1] != 0;
  }
  Fp12 f, t;
  fp12_one(&f);
  int k = 0;
  for (int bit = 62; bit >= 0; --bit) {
    fp12_sqr(&t, &f); f = t;
    const int nsteps = ((X_ABS >> bit) & 1) ? 2 : 1;
    for (int step = 0; step < nsteps; ++step, ++k)
      for (int i = 0; i < 2; ++i) {
        const uint32_t* L = prep.data() + ((size_t)i * G2_LINES + k) * G2_LINE_WORDS;
        Fp2 c0 = fp2_load_words(L);
        Fp2 c1 = fp2_fit(fp2_mul_fp(fp2_load_words(L + 2 * NLB), P[i].x));
        Fp2 c4 = fp2_fit(fp2_mul_fp(fp2_load_words(L + 4 * NLB), P[i].y));
        if (!skip[i]) fp12_mul_by_014(&f, &c0, &c1, &c4);
      }
  }
  if (k != G2_LINES) return 99;
  fp12_conj(&t, &f); f = t;
  Fp12 e;
  final_exponentiation(&e, &f);
  if (!ok) return PST_INVALID;
  return fp12_is_one(&e) ? PST_OK : PST_FAIL;
}
uint32_t hb_pairing_check2(const uint8_t* g1x2, const uint8_t* g2x2) {
  uint32_t w1[48], w2[96]; memcpy(w1, g1x2, 192); memcpy(w2, g2x2, 384);
  return pairing_check2_item(w1, w2);
}
// ---- G1 group law (g1.cuh): points as x || y (96 B, all-zero = infinity) ----
static G1P g1_in(const uint8_t* b) {
  bool any = false; for (int i = 0; i < 96; ++i) any = any || b[i];
  if (!any) return g1_identity();
  G1P p; p.X = inw(b); p.Y = inw(b + 48); p.Z = fp_one(); return p;
}
static void g1_out(uint8_t* b, const G1P& p) {
  if (g1_is_identity(p)) { memset(b, 0, 96); return; }
  FpS x, y; g1_to_affine(x, y, p); outw(b, x); outw(b + 48, y);
}
// op 0: projective add (both operands rescaled by an odd factor so that Z != 1), 1: mixed add, 2: mixed add of -b, 3: doubling of a
void hb_g1_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* r) {
  G1P p = g1_in(a);
  if (op == 0) {
    G1P q = g1_in(b);
    FpS k = inw(a + 48);                       // any non-zero scale
    FpS one = fp_one();
    if (fp_is_zero(k)) k = one;
    p.X = fp_fit(fp_mul(p.X, k)); p.Y = fp_fit(fp_mul(p.Y, k)); p.Z = fp_fit(fp_mul(p.Z, k));
    auto k2 = fp_fit(fp_add(k, one));
    q.X = fp_fit(fp_mul(q.X, k2)); q.Y = fp_fit(fp_mul(q.Y, k2)); q.Z = fp_fit(fp_mul(q.Z, k2));
    g1_out(r, g1_add(p, q));
  } else if (op == 3) {
    g1_out(r, g1_dbl(p));
  } else {
    g1_out(r, g1_madd(p, inw(b), inw(b + 48), op == 2));
  }
}
// chains of the three laws: acc = sum_i (+/-) P_i by mixed additions into a running projective sum, then doubled k times
void hb_g1_chain(uint32_t n, const uint8_t* pts, const uint8_t* signs, uint32_t dbl, uint8_t* r) {
  G1P acc = g1_identity();
  for (uint32_t i = 0; i < n; ++i) acc = g1_madd(acc, inw(pts + 96 * i), inw(pts + 96 * i + 48), signs[i] != 0);
  for (uint32_t i = 0; i < dbl; ++i) acc = g1_dbl(acc);
  g1_out(r, acc);
}
}
