// tests/hostsim -- TEST TOOLING ONLY: host build of bls12_oct.cuh (one pairing item per 8 lanes).  The eight lanes of an
// item run as eight threads that meet at a barrier in every cross-lane move (bls12_oct.cuh, host branch); results are
// compared with the one-lane tower of bls12.cuh, which tests/test_bls_pairing.py holds against the Python oracle.
#include "../../ark_ec_vrfs_amd/csrc/bls12.cuh"
#include "../../ark_ec_vrfs_amd/csrc/bls12_oct.cuh"
#include <cstring>
#include <functional>
#include <thread>
#include <vector>
using namespace bls;
using namespace bls::oct;

static FpS inw(const uint8_t* b) { uint32_t w[12]; memcpy(w, b, 48); FpS r; fp_from_words(r, w); return r; }
static Fp2 in2(const uint8_t* b) { Fp2 r; r.a = inw(b); r.b = inw(b + 48); return r; }
static void in12(Fp12* f, const uint8_t* b) {
  Fp2* c[6] = {&f->c0.c0, &f->c0.c1, &f->c0.c2, &f->c1.c0, &f->c1.c1, &f->c1.c2};
  for (int i = 0; i < 6; ++i) *c[i] = in2(b + 96 * i);
}
// run fn(lane coordinates) on the 8 lanes of one item; returns the 8 results
static std::vector<uint32_t> run8(const std::function<uint32_t(const Ln&)>& fn) {
  HostGroup grp;
  std::vector<uint32_t> out(8, 0xffffffffu);
  std::vector<std::thread> th;
  for (int l = 0; l < 8; ++l)
    th.emplace_back([&, l]() {
      t_grp = &grp;
      t_lane = l;
      Ln ln;
      lanes_of(ln, l);
      out[l] = fn(ln);
    });
  for (auto& t : th) t.join();
  return out;
}
static uint32_t agree(const std::vector<uint32_t>& v) {
  for (int l = 1; l < 8; ++l) if (v[l] != v[0]) return 0x80000000u | (uint32_t)l;
  return v[0];
}
extern "C" {
// the cross-lane moves themselves: lane l holds l in limb 0; bit k of the result = move k is wrong in some lane
uint32_t hb_oct_moves() {
  auto r = run8([&](const Ln& ln) -> uint32_t {
    const int me = 4 * ln.h + ln.j;
    uint32_t bad = 0;
    if (xp_i32(me) != (me ^ 4)) bad |= 1;
    if (xq_i32<QP_ROT1>(me) != (ln.j < 3 ? 4 * ln.h + (ln.j + 1) % 3 : me)) bad |= 2;
    if (xq_i32<QP_PAIRSWAP>(me) != (me ^ 1)) bad |= 4;
    FpS a; for (int i = 0; i < NLB; ++i) a.v[i] = 100 * me + i;
    FpS k; for (int i = 0; i < NLB; ++i) k.v[i] = -7;
    const FpS u = xp_h1(a), w = xp_h0(k, a);
    for (int i = 0; i < NLB; ++i) {
      if (u.v[i] != 100 * (ln.h ? (me ^ 4) : me) + i) bad |= 8;
      if (w.v[i] != (ln.h ? -7 : 100 * (me ^ 4) + i)) bad |= 16;
    }
    if (!x_all8(true, ln) || x_all8(me != 5, ln)) bad |= 32;
    return bad;
  });
  uint32_t o = 0;
  for (auto v : r) o |= v;
  return o;
}
// x, y: 576 B each (12 field elements, the order of hostsim_bls.hip); c: 3 x 96 B line coefficients.  Bit mask of the
// tower operations whose oct result differs from the one-lane result: 1 mul, 2 sqr, 4 cyclotomic sqr (on x made
// cyclotomic first), 8 mul_by_014, 16 frobenius, 32 inverse, 64 conj / scatter round trip, 128 lanes disagree
uint32_t hb_oct_selftest(const uint8_t* xb, const uint8_t* yb, const uint8_t* cb) {
  Fp12 x, y;
  in12(&x, xb); in12(&y, yb);
  const Fp2 c0 = in2(cb), c1 = in2(cb + 96), c4 = in2(cb + 192);
  Fp12 r_mul, r_sqr, r_cyc, r_014, r_frob, r_inv, r_conj, xc;
  fp12_mul(&r_mul, &x, &y);
  fp12_sqr(&r_sqr, &x);
  {   // x^((p^6 - 1)(p^2 + 1)) lies in the cyclotomic subgroup
    Fp12 t0, t1, t2;
    fp12_conj(&t0, &x); fp12_inv(&t1, &x); fp12_mul(&t2, &t0, &t1);
    fp12_frob(&t0, &t2); fp12_frob(&t1, &t0); fp12_mul(&xc, &t1, &t2);
  }
  fp12_cyclotomic_sqr(&r_cyc, &xc);
  r_014 = x; fp12_mul_by_014(&r_014, &c0, &c1, &c4);
  fp12_frob(&r_frob, &x);
  fp12_inv(&r_inv, &x);
  fp12_conj(&r_conj, &x);
  auto r = run8([&](const Ln& ln) -> uint32_t {
    const O12 xo = o12_scatter(&x, ln), yo = o12_scatter(&y, ln), xco = o12_scatter(&xc, ln);
    const FpS l0 = ln.h ? c0.b : c0.a, l1 = ln.h ? c1.b : c1.a, l4 = ln.h ? c4.b : c4.a;
    uint32_t bad = 0;
    if (!o12_same(o12_mul(xo, yo, ln), &r_mul, ln)) bad |= 1;
    if (!o12_same(o12_sqr(xo, ln), &r_sqr, ln)) bad |= 2;
    if (!o12_same(o12_cyclotomic_sqr(xco, ln), &r_cyc, ln)) bad |= 4;
    if (!o12_same(o12_mul_by_014<true>(xo, l0, l1, l4, ln), &r_014, ln)) bad |= 8;
    if (!o12_same(o12_mul_by_014<false>(xo, l0, l1, l4, ln), &r_014, ln)) bad |= 8;
    if (!o12_same(o12_frob(xo, ln), &r_frob, ln)) bad |= 16;
    if (!o12_same(o12_inv(xo, ln), &r_inv, ln)) bad |= 32;
    if (!o12_same(o12_conj(xo), &r_conj, ln) || !o12_same(xo, &x, ln)) bad |= 64;
    return bad;
  });
  uint32_t o = 0;
  for (auto v : r) o |= (v & 0x7f);
  if (agree(r) & 0x80000000u) o |= 128;
  return o;
}
uint32_t hb_oct_pairing_check2(const uint8_t* g1x2, const uint8_t* g2x2) {
  uint32_t w1[48], w2[96]; memcpy(w1, g1x2, 192); memcpy(w2, g2x2, 384);
  return agree(run8([&](const Ln& ln) -> uint32_t { return pairing_check2_oct(w1, w2, ln); }));
}
// the per-item path in two passes: the lines of both pairs into a buffer, then the Miller loop over them
uint32_t hb_oct_pairing_check2_split(const uint8_t* g1x2, const uint8_t* g2x2) {
  uint32_t w1[48], w2[96]; memcpy(w1, g1x2, 192); memcpy(w2, g2x2, 384);
  std::vector<uint32_t> lines(oct_lines_words_per_item()), flags(4, 0xdeadu);
  run8([&](const Ln& ln) -> uint32_t { pairing_lines_oct(w1, w2, lines.data(), flags.data(), 1, 0, ln); return 0; });
  return agree(run8([&](const Ln& ln) -> uint32_t { return pairing_check2_oct_lines(lines.data(), flags.data(), 1, 0, ln); }));
}
uint32_t hb_oct_pairing_check2_prepared(const uint8_t* g1x2, const uint8_t* g2x2) {
  uint32_t w1[48], w2[96]; memcpy(w1, g1x2, 192); memcpy(w2, g2x2, 384);
  std::vector<uint32_t> prep(G2_PREP_WORDS);
  pairing_prepare_g2_pair(w2, prep.data(), 0);
  pairing_prepare_g2_pair(w2, prep.data(), 1);
  return agree(run8([&](const Ln& ln) -> uint32_t { return pairing_check2_oct_prepared(w1, prep.data(), ln); }));
}
}
