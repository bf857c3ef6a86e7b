// bls12_oct.cuh -- the BLS12-381 pairing check of bls12.cuh with ONE ITEM PER 8 LANES (SURVEY.md section 8 row a11;
// replaces ark_ec::pairing::Pairing::{multi_miller_loop, final_exponentiation}, reached from /root/reference through
// `ring`, src/lib.rs:14).
//
// Why (round 4): the quad layout of bls12_quad.cuh holds an Fp2 per lane (an Fp12 = 56 registers per lane, 512-register
// kernel, ONE wave per SIMD at BASELINE.json's 2^14 items) and pays three full Montgomery products plus the Karatsuba
// sums per Fp2 product: 44 % of its instructions are multiply-adds, and a lone wave issues every instruction at 4 cycles.
// Here an Fp2 value a + b u is SPLIT over a lane pair -- lane h = 0 holds a, lane h = 1 holds b -- so
//   * an Fp2 product is ONE lazy double product per lane: h = 0 forms ac - bd, h = 1 forms ad + bc, 2 x 196 product
//     multiply-adds and ONE Montgomery reduction (fp_mul2) instead of 3 x (196 + 196), with no Karatsuba sums;
//   * every linear operation (add, sub, negate, the tower's recombinations) touches 14 limbs per lane instead of 28;
//   * an Fp12 is 2 Fp = 28 registers per lane: the kernel fits 256 registers, so 2^14 items x 8 lanes = 2048 waves run
//     TWO per SIMD, where the plain 32-bit instructions (adds, masks, moves, DPP) issue at twice the rate of a lone wave.
// Lane map inside a 16-lane DPP row: item = lane >> 3, h = (lane >> 2) & 1, j = lane & 3.  Quad h of an item holds
// component h; lane j < 3 of a quad holds the v^j column of every Fp6 (as in bls12_quad.cuh), j = 3 is spare in the tower
// and works in the Miller loop's G2 steps.  Column moves are DPP quad_perm (one instruction per limb, both quads alike);
// the partner lane (lane ^ 4) is reached by row_shl:4 / row_shr:4 with bank masks, which also fuse the "which half am I"
// selects of the double product into the move.
//
// The host build (tests/hostsim) runs the same source with the eight lanes of one item as eight threads that meet at a
// barrier in every cross-lane move: the lane algebra is checked on the CPU against the one-lane tower before it sees a GPU.
#pragma once
#include "bls12.cuh"
#include <type_traits>
#if !defined(__HIP_DEVICE_COMPILE__)
#include <atomic>
#include <sched.h>
#endif

namespace bls {
namespace oct {

// quad_perm controls: lane i reads lane perm[i]
constexpr int QP_ROT1 = 1 | (2 << 2) | (0 << 4) | (3 << 6);     // j <- (j+1) % 3
constexpr int QP_ROT2 = 2 | (0 << 2) | (1 << 4) | (3 << 6);     // j <- (j+2) % 3
constexpr int QP_SWAP12 = 0 | (2 << 2) | (1 << 4) | (3 << 6);   // 1 <-> 2
constexpr int QP_SWAP01 = 1 | (0 << 2) | (2 << 4) | (3 << 6);   // 0 <-> 1
constexpr int QP_BC0 = 0x00, QP_BC1 = 0x55, QP_BC2 = 0xaa, QP_BC3 = 0xff;
constexpr int QP_PAIRSWAP = 1 | (0 << 2) | (3 << 4) | (2 << 6); // 0 <-> 1, 2 <-> 3
constexpr int DPP_ROW_SHL4 = 0x104;                              // lane i reads lane i + 4 of its row
constexpr int DPP_ROW_SHR4 = 0x114;                              // lane i reads lane i - 4 of its row
constexpr int BANK_H0 = 0x5, BANK_H1 = 0xa;                      // the quads of a row that hold h = 0 / h = 1

struct Ln { int j, h, base; };       // column, component, first lane of the item inside the wave

#if !defined(__HIP_DEVICE_COMPILE__)
// ---- host emulation: the 8 lanes of ONE item are 8 threads; a cross-lane move is store, barrier, load, barrier ----
struct HostGroup {
  int32_t slot[8][16];
  std::atomic<int> count{0};
  std::atomic<int> gen{0};
  void barrier() {
    const int g = gen.load(std::memory_order_acquire);
    if (count.fetch_add(1, std::memory_order_acq_rel) == 7) {
      count.store(0, std::memory_order_relaxed);
      gen.store(g + 1, std::memory_order_release);
    } else {
      int spins = 0;
      while (gen.load(std::memory_order_acquire) == g)
        if (++spins > 200) { sched_yield(); spins = 0; }
    }
  }
};
inline thread_local HostGroup* t_grp = nullptr;
inline thread_local int t_lane = 0;
inline void host_xchg(int32_t* out, const int32_t* in, int n, int src_lane) {
  for (int i = 0; i < n; ++i) t_grp->slot[t_lane][i] = in[i];
  t_grp->barrier();
  for (int i = 0; i < n; ++i) out[i] = t_grp->slot[src_lane][i];
  t_grp->barrier();
}
inline int host_qsrc(int ctrl) { return (t_lane & 4) | ((ctrl >> (2 * (t_lane & 3))) & 3); }
#endif

// ---- cross-lane moves ----
template <int CTRL>
VRF_HD int32_t xq_i32(int32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
#else
  int32_t r;
  host_xchg(&r, &v, 1, host_qsrc(CTRL));
  return r;
#endif
}
VRF_HD int32_t xp_i32(int32_t v) {                      // the value of lane ^ 4
#if defined(__HIP_DEVICE_COMPILE__)
  const int32_t t = __builtin_amdgcn_update_dpp(v, v, DPP_ROW_SHL4, 0xf, BANK_H0, false);
  return __builtin_amdgcn_update_dpp(t, v, DPP_ROW_SHR4, 0xf, BANK_H1, false);
#else
  int32_t r;
  host_xchg(&r, &v, 1, t_lane ^ 4);
  return r;
#endif
}
template <int CTRL, int L, int V>
VRF_HD Fp<L, V> xq(const Fp<L, V>& a) {
  Fp<L, V> r;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
  for (int i = 0; i < NLB; ++i) r.v[i] = __builtin_amdgcn_update_dpp(0, a.v[i], CTRL, 0xf, 0xf, false);
#else
  host_xchg(r.v, a.v, NLB, host_qsrc(CTRL));
#endif
  return r;
}
template <int L, int V>
VRF_HD Fp<L, V> xp(const Fp<L, V>& a) {                 // partner's value
  Fp<L, V> r;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
  for (int i = 0; i < NLB; ++i) {
    const int32_t t = __builtin_amdgcn_update_dpp(a.v[i], a.v[i], DPP_ROW_SHL4, 0xf, BANK_H0, false);
    r.v[i] = __builtin_amdgcn_update_dpp(t, a.v[i], DPP_ROW_SHR4, 0xf, BANK_H1, false);
  }
#else
  host_xchg(r.v, a.v, NLB, t_lane ^ 4);
#endif
  return r;
}
template <int L, int V>
VRF_HD Fp<L, V> xp_h1(const Fp<L, V>& own) {            // h ? partner(own) : own   -- one move
  Fp<L, V> r;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
  for (int i = 0; i < NLB; ++i) r.v[i] = __builtin_amdgcn_update_dpp(own.v[i], own.v[i], DPP_ROW_SHR4, 0xf, BANK_H1, false);
#else
  host_xchg(r.v, own.v, NLB, t_lane ^ 4);
  if (!(t_lane & 4)) r = own;
#endif
  return r;
}
template <int L, int V>
VRF_HD Fp<L, V> xp_h0(const Fp<L, V>& keep, const Fp<L, V>& src) {   // h ? keep : partner(src)   -- one move
  Fp<L, V> r;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
  for (int i = 0; i < NLB; ++i) r.v[i] = __builtin_amdgcn_update_dpp(keep.v[i], src.v[i], DPP_ROW_SHL4, 0xf, BANK_H0, false);
#else
  host_xchg(r.v, src.v, NLB, t_lane ^ 4);
  if (t_lane & 4) r = keep;
#endif
  return r;
}
// true in every lane of the item iff v holds in all eight
VRF_HD bool x_all8(bool v, const Ln& ln) {
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned long long m = __builtin_amdgcn_ballot_w64(v);
  return ((m >> ln.base) & 0xffull) == 0xffull;
#else
  (void)ln;
  int32_t mine = v ? 1 : 0, got;
  bool all = true;
  for (int k = 0; k < 8; ++k) { host_xchg(&got, &mine, 1, k); all = all && got != 0; }
  return all;
#endif
}

// ---- Fp2 over a lane pair: this lane's component.  L-bounds are brought to what a double product allows. ----
template <bool NORM, int L, int V>
VRF_HD auto fp_norm_if(const Fp<L, V>& x) {
  if constexpr (NORM) return fp_norm(x);
  else return x;
}
template <int L1, int V1, int L2, int V2>
VRF_HD Fp<1, mul2_v(V1, V2, V1, V2)> o2_mul(const Fp<L1, V1>& x0, const Fp<L2, V2>& y0) {
  constexpr bool NX = L1 * L2 > 4 && L1 > 1;             // a double product takes L1 * L2 <= 4
  constexpr bool NY = (NX ? 1 : L1) * L2 > 4;
  const auto x = fp_norm_if<NX>(x0);
  const auto y = fp_norm_if<NY>(y0);
  using X = std::remove_const_t<decltype(x)>;
  const X U = xp_h1(x);                                  // h = 0: a        h = 1: a (the partner's)
  const X W = xp_h0(x, X(fp_neg(x)));                    // h = 0: -b       h = 1: b (own)
  const auto D = xp(y);                                  // h = 0: d        h = 1: c
  return fp_mul2(U, y, W, D);                            // h = 0: ac - bd  h = 1: ad + bc
}
// x * (re + im u) for a constant
template <int L1, int V1, int VC>
VRF_HD auto o2_mul_const(const Fp<L1, V1>& x0, const Fp<1, VC>& re, const Fp<1, VC>& im, const Ln& ln) {
  const auto x = fp_norm_if<(L1 > 4)>(x0);
  using X = std::remove_const_t<decltype(x)>;
  const X U = xp_h1(x);
  const X W = xp_h0(x, X(fp_neg(x)));
  const Fp<1, VC> C = fp_select(ln.h != 0, im, re), D = fp_select(ln.h != 0, re, im);
  return fp_mul2(U, C, W, D);
}
template <int L, int V>
VRF_HD Fp<1, mul_v(2 * V, 2 * V)> o2_sqr(const Fp<L, V>& x0, const Ln& ln) {   // (a+b)(a-b) | 2ab
  const auto x = fp_norm_if<(L > 1)>(x0);
  const auto P = xp(x);
  const Fp<2, 2 * V> s = fp_add(x, P), d = fp_sub(x, P), p2 = fp_add(P, P), xw = Fp<2, 2 * V>(x);
  const Fp<2, 2 * V> A = fp_select(ln.h != 0, xw, s), B = fp_select(ln.h != 0, p2, d);
  return fp_mul(A, B);
}
template <int L, int V>
VRF_HD Fp<lsum(L, L), 2 * V> o2_mul_xi(const Fp<L, V>& x, const Ln& ln) {      // (a - b) | (b + a)
  const Fp<L, V> P = xp(x);
  return fp_add(x, fp_select(ln.h != 0, P, fp_neg(P)));
}
template <int L, int V>
VRF_HD Fp<L, V> o2_conj(const Fp<L, V>& x, const Ln& ln) { return fp_select(ln.h != 0, fp_neg(x), x); }

VRF_HD FpS o_zero() { return FpS(fp_zero()); }
VRF_HD FpS o2_one(const Ln& ln) { return fp_select(ln.h == 0, FpS(fp_one()), FpS(fp_zero())); }
template <int L, int V>
VRF_HD bool o2_is_zero(const Fp<L, V>& x) {           // both components zero?  (every lane of the pair gets the answer)
  const int32_t z = fp_is_zero(x) ? 1 : 0, zp = xp_i32(z);    // the move is unconditional: every lane takes part
  return z != 0 && zp != 0;
}

#if defined(__HIP_DEVICE_COMPILE__)
#define OCT_NOINLINE __device__ __attribute__((noinline))
#else
#define OCT_NOINLINE inline
#endif
// ---- Fp6: column j of (x0, x1, x2) in lane j, components over h ----
using In6 = Fp<2, 36>;        // what the column product accepts: a stored value, a sum of two, c0 + v c1
using Out6 = Fp<3, 36>;       // what it returns (limbs up to 3 units, value below 36 p)

// x * v: (x0, x1, x2) -> (xi x2, x0, x1)
template <int L, int V>
VRF_HD Fp<lsum(L, L), 2 * V> o6_mul_v(const Fp<L, V>& x, const Ln& ln) {
  const Fp<L, V> u = xq<QP_ROT2>(x);
  const auto xu = o2_mul_xi(u, ln);
  using R = Fp<lsum(L, L), 2 * V>;
  return fp_select(ln.j == 0, xu, R(u));
}

// p_j = x_j y_j and c_j = x_a y_b + x_b y_a (a, b the other two columns)
VRF_HD void o6_pc(Fp<1, 3>& p, Fp<1, 12>& c, const In6& x, const In6& y) {
  const auto xo = fp_norm(fp_add(xq<QP_ROT1>(x), xq<QP_ROT2>(x)));        // V 72
  const auto yo = fp_norm(fp_add(xq<QP_ROT1>(y), xq<QP_ROT2>(y)));
  p = o2_mul(x, y);                                                       // V 3
  const auto cr = o2_mul(xo, yo);                                         // V 6
  c = fp_norm(fp_sub(fp_sub(cr, xq<QP_ROT1>(p)), xq<QP_ROT2>(p)));        // V 12
}
// Column-distributed Fp6 product: r0 = p0 + xi c0, r1 = c2 + xi p2, r2 = c1 + p1.  Four double products per lane pair
// and column; the result keeps its natural bound, callers reduce where they store.
OCT_NOINLINE Out6 o6_mul(In6 x, In6 y, Ln ln) {
  Fp<1, 3> p;
  Fp<1, 12> c;
  o6_pc(p, c, x, y);
  const Fp<1, 12> ps = xq<QP_SWAP12>(Fp<1, 12>(p)), cs = xq<QP_SWAP12>(c);
  const Fp<1, 12> A = fp_select(ln.j == 0, ps, cs), B = fp_select(ln.j == 0, cs, ps);
  const Fp<2, 24> xiB = o2_mul_xi(B, ln);
  return Out6(fp_add(A, fp_select(ln.j == 2, Fp<2, 24>(B), xiB)));
}

struct O12 { FpS c0, c1; };          // this lane's component of its column of c0 and of c1

VRF_HD O12 o12_mul(const O12& x, const O12& y, const Ln& ln) {
  const Out6 t0 = o6_mul(In6(x.c0), In6(y.c0), ln);
  const Out6 t1 = o6_mul(In6(x.c1), In6(y.c1), ln);
  const Out6 s = o6_mul(In6(fp_add(x.c0, x.c1)), In6(fp_add(y.c0, y.c1)), ln);
  O12 o;
  o.c1 = fp_fit(fp_sub(fp_sub(s, t0), t1));
  o.c0 = fp_fit(fp_add(t0, o6_mul_v(fp_norm(t1), ln)));
  return o;
}
// complex squaring: c0' = (c0 + c1)(c0 + v c1) - ab - v ab, c1' = 2ab, ab = c0 c1
VRF_HD O12 o12_sqr(const O12& x, const Ln& ln) {
  const Out6 ab = o6_mul(In6(x.c0), In6(x.c1), ln);
  const auto s1 = fp_norm(fp_add(x.c0, o6_mul_v(x.c1, ln)));               // V 36
  const Out6 m2 = o6_mul(In6(fp_add(x.c0, x.c1)), In6(s1), ln);
  const auto abn = fp_norm(ab);
  O12 o;
  o.c0 = fp_fit(fp_sub(fp_sub(m2, abn), o6_mul_v(abn, ln)));
  o.c1 = fp_fit(fp_dbl(abn));
  return o;
}
// the two operands a lane derives from its share x of a left factor: (U, W) = (a, -b) for h = 0, (a, b) for h = 1
template <int V>
struct Pre { Fp<1, V> U, W; };
template <int V>
VRF_HD Pre<V> o2_pre(const Fp<1, V>& x) {
  Pre<V> r;
  r.U = xp_h1(x);
  r.W = xp_h0(x, fp_neg(x));
  return r;
}
// f * (c0 + c1 v + c4 v w); the line coefficients (this lane's component) are the same in every column.  With
// f = f0 + f1 w:  c0' = f0 c0 + (f0 v) c1 + (f1 v^2) c4,  c1' = f1 c0 + (f1 v) c1 + (f0 v) c4  -- each column of each half is
// a sum of three Fp2 products, formed as ONE six-product accumulation with one reduction (fp_mul6): 2 x 1372 multiply-adds
// where the Karatsuba route (two column products and one Fp2 product) took 2940 and twice the glue.
VRF_HD O12 o12_mul_by_014_fused(const O12& f, const FpS& c0, const FpS& c1, const FpS& c4, const Ln& ln) {
  const auto f0v = fp_norm(o6_mul_v(f.c0, ln));                             // V 24
  const auto f1v = fp_norm(o6_mul_v(f.c1, ln));                             // V 24
  const auto f1vv = fp_norm(o6_mul_v(f1v, ln));                             // V 48
  const FpS d0 = xp(c0), d1 = xp(c1), d4 = xp(c4);
  const auto p0 = o2_pre(f.c0), p1 = o2_pre(f.c1);
  const auto p0v = o2_pre(f0v), p1v = o2_pre(f1v);
  const auto p1vv = o2_pre(f1vv);
  O12 r;
  r.c0 = fp_fit(fp_mul6(p0.U, p0.W, p0v.U, p0v.W, p1vv.U, p1vv.W, c0, d0, c1, d1, c4, d4));
  r.c1 = fp_fit(fp_mul6(p1.U, p1.W, p1v.U, p1v.W, Fp<1, 48>(p0v.U), Fp<1, 48>(p0v.W), c0, d0, c1, d1, c4, d4));
  return r;
}
// The same product through two column products and one Fp2 product (Karatsuba over w): more instructions executed but
// less code, for the kernel whose loop also holds the G2 steps (with the fused form inlined beside them the loop outgrew
// the instruction cache: 8.97 -> 9.27 ms per 2^14 checks, while the prepared-lines kernel went 7.30 -> 6.68 ms).
VRF_HD O12 o12_mul_by_014_karatsuba(const O12& f, const FpS& c0, const FpS& c1, const FpS& c4, const Ln& ln) {
  const FpS z = o_zero();
  const In6 y01 = In6(fp_select(ln.j == 0, c0, fp_select(ln.j == 1, c1, z)));                         // (c0, c1, 0)
  const In6 y0o = fp_select(ln.j == 0, In6(c0), fp_select(ln.j == 1, In6(fp_add(c1, c4)), In6(z)));   // (c0, c1 + c4, 0)
  const Out6 aa = o6_mul(In6(f.c0), y01, ln);
  const Out6 s = o6_mul(In6(fp_add(f.c0, f.c1)), y0o, ln);
  const auto bb = fp_norm(o6_mul_v(o2_mul(f.c1, c4), ln));                  // f.c1 * (c4 v)
  O12 r;
  r.c1 = fp_fit(fp_sub(fp_sub(s, aa), bb));
  r.c0 = fp_fit(fp_add(o6_mul_v(bb, ln), aa));
  return r;
}
template <bool FUSED>
VRF_HD O12 o12_mul_by_014(const O12& f, const FpS& c0, const FpS& c1, const FpS& c4, const Ln& ln) {
  if constexpr (FUSED) return o12_mul_by_014_fused(f, c0, c1, c4, ln);
  else return o12_mul_by_014_karatsuba(f, c0, c1, c4, ln);
}
// Granger-Scott squaring in the cyclotomic subgroup: column 0 squares the Fp4 (c0.c0, c1.c1), column 1 (c0.c1, c1.c2),
// column 2 (c1.c0, c0.c2)
VRF_HD O12 o12_cyclotomic_sqr(const O12& x, const Ln& ln) {
  const FpS g = xq<QP_ROT1>(x.c1);
  const FpS a = fp_select(ln.j == 2, g, x.c0), b = fp_select(ln.j == 2, x.c0, g);
  const auto tmp = o2_mul(a, b);                                           // V 2
  const auto s2 = fp_norm(fp_add(o2_mul_xi(b, ln), a));                     // V 36
  const auto s = o2_mul(fp_add(a, b), s2);                                  // V 2
  const auto T0 = fp_norm(fp_sub(fp_sub(s, tmp), o2_mul_xi(tmp, ln)));      // V 8
  const auto T1 = fp_dbl(tmp);                                              // L 2, V 4
  const auto u0 = xq<QP_SWAP12>(T0);
  const auto u1r = xq<QP_SWAP01>(T1);
  const auto u1x = o2_mul_xi(u1r, ln);                                      // L 4, V 8
  using U1 = std::remove_const_t<decltype(u1x)>;
  const auto u1 = fp_norm(fp_select(ln.j == 0, u1x, U1(u1r)));
  O12 o;
  o.c0 = fp_fit(fp_sub(fp_add(fp_dbl(u0), u0), fp_dbl(x.c0)));
  o.c1 = fp_fit(fp_add(fp_add(fp_dbl(u1), u1), fp_dbl(x.c1)));
  return o;
}
VRF_HD O12 o12_conj(const O12& x) {
  O12 o;
  o.c0 = x.c0;
  o.c1 = fp_neg(x.c1);
  return o;
}
// f^p: column j holds the coefficients of w^(2j) and w^(2j+1)
VRF_HD O12 o12_frob(const O12& x, const Ln& ln) {
  const Fp2 g1 = gamma_const(1), g2 = gamma_const(2), g3 = gamma_const(3), g4 = gamma_const(4), g5 = gamma_const(5);
  const FpS one_re = FpS(fp_one()), zero = o_zero();
  const FpS g0re = fp_select(ln.j == 0, one_re, fp_select(ln.j == 1, g2.a, g4.a));
  const FpS g0im = fp_select(ln.j == 0, zero, fp_select(ln.j == 1, g2.b, g4.b));
  const FpS g1re = fp_select(ln.j == 0, g1.a, fp_select(ln.j == 1, g3.a, g5.a));
  const FpS g1im = fp_select(ln.j == 0, g1.b, fp_select(ln.j == 1, g3.b, g5.b));
  O12 o;
  o.c0 = fp_fit(o2_mul_const(o2_conj(x.c0, ln), g0re, g0im, ln));
  o.c1 = fp_fit(o2_mul_const(o2_conj(x.c1, ln), g1re, g1im, ln));
  return o;
}
VRF_HD O12 o12_one(const Ln& ln) {
  O12 o;
  o.c0 = fp_select(ln.j == 0 && ln.h == 0, FpS(fp_one()), o_zero());
  o.c1 = o_zero();
  return o;
}
VRF_HD bool o12_is_one(const O12& x, const Ln& ln) {
  const bool mine = fp_eq(x.c0, fp_select(ln.j == 0 && ln.h == 0, FpS(fp_one()), o_zero())) && fp_is_zero(x.c1);
  return x_all8(mine || ln.j == 3, ln);
}

// ---- inversion, all of it in the distributed form ----
// 1 / (a + b u) = (a - b u) / (a^2 + b^2), the value replicated wherever x is
VRF_HD FpS o2_inv(const FpS& x, const Ln& ln) {
  const auto sq = fp_sqr(x);
  FpS n = fp_fit(fp_add(sq, xp(sq))), ni;
  fp_inv(&ni, &n);
  return fp_fit(fp_mul(o2_conj(x, ln), ni));
}
// (t0, t1, t2) = (x0^2 - xi x1 x2, xi x2^2 - x0 x1, x1^2 - x0 x2) doubled; x t = norm in column 0; 1/x = t / norm
VRF_HD FpS o6_inv(const FpS& x, const Ln& ln) {
  Fp<1, 3> p;
  Fp<1, 12> c;
  o6_pc(p, c, In6(x), In6(x));                                             // p_j = x_j^2, c_j = 2 x_a x_b
  const Fp<2, 6> ps = xq<QP_SWAP12>(fp_dbl(p));
  const Fp<1, 12> cs = xq<QP_SWAP12>(c);
  // column 0: 2 p0 - xi c0      column 1: xi 2 p2 - c2      column 2: 2 p1 - c1
  const auto xi_c = o2_mul_xi(cs, ln);                                      // L 2, V 24
  const auto xi_p = o2_mul_xi(ps, ln);                                      // L 4, V 12
  using W = Fp<6, 36>;
  const W t0 = W(fp_sub(ps, xi_c)), t1 = W(fp_sub(xi_p, cs)), t2 = W(fp_sub(ps, cs));
  const FpS t = fp_fit(fp_select(ln.j == 0, t0, fp_select(ln.j == 1, t1, t2)));
  const FpS nrm = xq<QP_BC0>(fp_fit(o6_mul(In6(x), In6(t), ln)));           // x t = (norm, 0, 0)
  return fp_fit(o2_mul(t, o2_inv(nrm, ln)));
}
VRF_HD O12 o12_inv(const O12& x, const Ln& ln) {
  const Out6 t0 = o6_mul(In6(x.c0), In6(x.c0), ln);
  const Out6 t1 = o6_mul(In6(x.c1), In6(x.c1), ln);
  const FpS d = fp_fit(fp_sub(t0, o6_mul_v(fp_norm(t1), ln)));              // c0^2 - v c1^2
  const FpS di = o6_inv(d, ln);
  O12 o;
  o.c0 = fp_fit(o6_mul(In6(x.c0), In6(di), ln));
  o.c1 = fp_fit(fp_neg(o6_mul(In6(x.c1), In6(di), ln)));
  return o;
}

// f^x, x = -X_ABS, f in the cyclotomic subgroup
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __attribute__((noinline))
#else
inline
#endif
O12 exp_by_x_o(O12 f, Ln ln) {
  O12 acc = f;
#pragma unroll 1
  for (int bit = 62; bit >= 0; --bit) {
    acc = o12_cyclotomic_sqr(acc, ln);
    if ((X_ABS >> bit) & 1) acc = o12_mul(acc, f, ln);
  }
  return o12_conj(acc);
}

// f^(3 (p^12 - 1)/r), the chain of bls12.cuh's final_exponentiation
VRF_HD O12 final_exponentiation_o(const O12& f, const Ln& ln) {
  O12 f2;
  {
    const O12 t = o12_mul(o12_conj(f), o12_inv(f, ln), ln);                 // f^(p^6 - 1)
    f2 = o12_mul(o12_frob(o12_frob(t, ln), ln), t, ln);                     // ^(p^2 + 1)
  }
  O12 y = o12_mul(exp_by_x_o(f2, ln), o12_conj(f2), ln);                    // ^(x - 1)
  y = o12_mul(exp_by_x_o(y, ln), o12_conj(y), ln);                          // ^(x - 1)^2
  y = o12_mul(exp_by_x_o(y, ln), o12_frob(y, ln), ln);                      // ^(x + p)        = y2
  O12 t = exp_by_x_o(exp_by_x_o(y, ln), ln);                                // y2^(x^2)
  t = o12_mul(t, o12_frob(o12_frob(y, ln), ln), ln);                        // * y2^(p^2)
  y = o12_mul(t, o12_conj(y), ln);                                          // ^(x^2 + p^2 - 1) = y3
  t = o12_mul(o12_cyclotomic_sqr(f2, ln), f2, ln);                          // f2^3
  return o12_mul(y, t, ln);
}

// ---- Miller loop steps.  Columns 0,1 of both quads own pair 0, columns 2,3 pair 1: four lanes per (P, Q) pair, s = j & 1
// says which Fp2 product of a round a lane pair forms, h which component of it.  The G2 state T = (X, Y, Z) is replicated
// over s.  Formulas and scalings: g2_double_q / g2_add_q of bls12_quad.cuh (homogeneous projective, halvings scaled away).
struct G2T { FpS X, Y, Z; };          // components
struct G2L { FpS c0, c1, c4; };
struct G1A { FpS x, y; };             // a G1 point (whole Fp coordinates, the same in both h)
struct G2A { FpS x, y; };             // components of the affine G2 point

OCT_NOINLINE Fp<1, 3> o2_mul_s(Fp<2, 24> x, Fp<2, 24> y) { return o2_mul(x, y); }

template <int L1, int V1, int L2, int V2>
VRF_HD void pair_round(FpS& pa, FpS& pb, const Fp<L1, V1>& u, const Fp<L2, V2>& v, bool sb) {
  const FpS m = FpS(o2_mul_s(Fp<2, 24>(u), Fp<2, 24>(v)));
  const FpS mo = xq<QP_PAIRSWAP>(m);
  pa = fp_select(sb, mo, m);
  pb = fp_select(sb, m, mo);
}
template <int L1, int V1>
VRF_HD void pair_round_sqr(FpS& pa, FpS& pb, const Fp<L1, V1>& u, bool sb, const Ln& ln) {
  const FpS m = fp_fit(o2_sqr(u, ln));
  const FpS mo = xq<QP_PAIRSWAP>(m);
  pa = fp_select(sb, mo, m);
  pb = fp_select(sb, m, mo);
}
VRF_HD FpS o_small12(const FpS& x) {                        // 12 x, storage form
  const FpS t = fp_fit(fp_add(fp_dbl(x), x));
  return fp_fit(fp_dbl(fp_dbl(t)));
}

// T <- 2T and the tangent line at T, scaled by the G1 coordinates of this lane's pair
VRF_HD G2L g2_double_o(G2T& T, const G1A& P, const Ln& ln) {
  const bool sb = (ln.j & 1) != 0;
  const FpS X = T.X, Y = T.Y, Z = T.Z;
  const FpS pyc = fp_select(ln.h == 0, P.y, o_zero());      // (Py, 0) as an Fp2 factor
  FpS xy, b, c, syz, j, e2, t1, t2, t3, c4s;
  pair_round(xy, b, fp_select(sb, Y, X), Y, sb);                                  // XY | Y^2
  {
    const auto yz = fp_norm(fp_add(Y, Z));
    pair_round_sqr(c, syz, fp_select(sb, yz, Fp<1, 2 * STORE_V>(Z)), sb, ln);      // Z^2 | (Y+Z)^2
  }
  const FpS e = fp_fit(o2_mul_xi(o_small12(c), ln));                               // b' * 3c = 12 xi c
  const FpS hh = fp_fit(fp_sub(syz, fp_add(b, c)));                                // (Y+Z)^2 - b - c
  pair_round_sqr(j, e2, fp_select(sb, e, X), sb, ln);                              // X^2 | e^2
  {
    const FpS f3 = fp_fit(fp_add(fp_dbl(e), e));
    const FpS bmf = fp_fit(fp_sub(b, f3)), bpf = fp_fit(fp_add(b, f3));
    pair_round(t1, t2, fp_select(sb, bpf, xy), fp_select(sb, bpf, bmf), sb);       // XY (b - f) | (b + f)^2
  }
  pair_round(t3, c4s, fp_select(sb, hh, b), fp_select(sb, pyc, hh), sb);           // b h | h Py
  T.X = fp_fit(fp_dbl(t1));
  T.Y = fp_fit(fp_sub(t2, o_small12(e2)));
  T.Z = fp_fit(fp_dbl(fp_dbl(t3)));
  G2L L;
  L.c0 = fp_fit(fp_sub(b, e));
  L.c1 = fp_fit(fp_mul(fp_norm(fp_neg(fp_add(fp_dbl(j), j))), P.x));               // -3 X^2 Px
  L.c4 = c4s;
  return L;
}
// T <- T + Q and the line through T and Q, scaled by the G1 coordinates
VRF_HD G2L g2_add_o(G2T& T, const G2A& Q, const G1A& P, const Ln& ln) {
  const bool sb = (ln.j & 1) != 0;
  const FpS X = T.X, Y = T.Y, Z = T.Z;
  const FpS pyc = fp_select(ln.h == 0, P.y, o_zero());
  FpS yz, xz, c, d, e, f, g, tq, lq, z3, x3, y3a, ey, c4s;
  pair_round(yz, xz, fp_select(sb, Q.x, Q.y), Z, sb);                              // Qy Z | Qx Z
  const FpS th = fp_fit(fp_sub(Y, yz)), lam = fp_fit(fp_sub(X, xz));
  pair_round_sqr(c, d, fp_select(sb, lam, th), sb, ln);                            // theta^2 | lambda^2
  pair_round(e, f, fp_select(sb, Z, lam), fp_select(sb, c, d), sb);                // lambda d | Z c
  pair_round(g, tq, fp_select(sb, th, X), fp_select(sb, Q.x, d), sb);              // X d | theta Qx
  pair_round(lq, z3, fp_select(sb, Z, lam), fp_select(sb, e, Q.y), sb);            // lambda Qy | Z e
  {
    const FpS hv = fp_fit(fp_sub(fp_add(e, f), fp_dbl(g)));
    const FpS gmh = fp_fit(fp_sub(g, hv));
    pair_round(x3, y3a, fp_select(sb, th, lam), fp_select(sb, gmh, hv), sb);       // lambda h | theta (g - h)
  }
  pair_round(ey, c4s, fp_select(sb, lam, e), fp_select(sb, pyc, Y), sb);           // e Y | lambda Py
  T.X = x3;
  T.Y = fp_fit(fp_sub(y3a, ey));
  T.Z = z3;
  G2L L;
  L.c0 = fp_fit(fp_sub(tq, lq));                                                   // theta Qx - lambda Qy
  L.c1 = fp_fit(fp_mul(fp_neg(th), P.x));
  L.c4 = c4s;
  return L;
}

// this lane's component of the G2 point of its pair: words x.c0 | x.c1 | y.c0 | y.c1 (12 each).  ok / inf are the
// whole point's (both components), the same in the four lanes of the pair.
VRF_HD bool g2_load_o(G2A& Q, bool& inf, const uint32_t* w /*48 words*/, const Ln& ln) {
  const uint32_t* wx = w + 12 * ln.h;
  const uint32_t* wy = w + 24 + 12 * ln.h;
  uint32_t any = 0;
  for (int i = 0; i < 12; ++i) any |= wx[i] | wy[i];
  const int32_t z = any == 0 ? 1 : 0, zp = xp_i32(z);         // unconditional: both lanes of the pair take part
  inf = z != 0 && zp != 0;
  const bool okx = fp_from_words(Q.x, wx), oky = fp_from_words(Q.y, wy);
  // y^2 = x^3 + 4 (1 + u)
  const FpS four = fp_fit(fp_dbl(fp_dbl(fp_one())));
  const auto x2 = fp_fit(o2_sqr(Q.x, ln));
  const auto rhs = fp_add(o2_mul(x2, Q.x), four);
  const auto lhs = o2_sqr(Q.y, ln);
  const int32_t mine = (okx && oky && (inf || fp_eq(lhs, rhs))) ? 1 : 0, theirs = xp_i32(mine);
  return mine != 0 && theirs != 0;
}

VRF_HD void lanes_of(Ln& ln, int lane_in_wave) {
  ln.j = lane_in_wave & 3;
  ln.h = (lane_in_wave >> 2) & 1;
  ln.base = lane_in_wave & ~7;
}

// One item per 8 lanes.  g1: 2 x 24 words, g2: 2 x 48 words (formats of pairing_check2_item).  Returns the status in
// every lane of the item.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __attribute__((noinline))
#else
inline
#endif
uint32_t pairing_check2_oct(const uint32_t* g1, const uint32_t* g2, Ln ln) {
  const int pi = ln.j >> 1;
  G1A P;
  G2A Q;
  bool i1, i2;
  {
    G1Aff P1;
    bool ok1 = g1_load(P1, i1, g1 + 24 * pi);
    P.x = P1.x; P.y = P1.y;
    const bool ok2 = g2_load_o(Q, i2, g2 + 48 * pi, ln);
    const int my_skip = (i1 || i2) ? 1 : 0;
    const int skip0 = xq_i32<QP_BC0>(my_skip), skip1 = xq_i32<QP_BC2>(my_skip);
    const bool all_ok = x_all8(ok1 && ok2, ln);
    G2T T;
    T.X = Q.x; T.Y = Q.y; T.Z = o2_one(ln);
    O12 f = o12_one(ln);
#pragma unroll 1
    for (int bit = 62; bit >= 0; --bit) {
      f = o12_sqr(f, ln);
      const int nsteps = ((X_ABS >> bit) & 1) ? 2 : 1;
#pragma unroll 1
      for (int step = 0; step < nsteps; ++step) {
        G2L L;
        if (step == 0) L = g2_double_o(T, P, ln);
        else L = g2_add_o(T, Q, P, ln);
#pragma unroll 1
        for (int i = 0; i < 2; ++i) {
          const FpS l0 = fp_select(i == 0, xq<QP_BC0>(L.c0), xq<QP_BC2>(L.c0));
          const FpS l1 = fp_select(i == 0, xq<QP_BC0>(L.c1), xq<QP_BC2>(L.c1));
          const FpS l4 = fp_select(i == 0, xq<QP_BC0>(L.c4), xq<QP_BC2>(L.c4));
          const bool skip = (i == 0 ? skip0 : skip1) != 0;
          if (!skip) f = o12_mul_by_014<false>(f, l0, l1, l4, ln);
        }
      }
    }
    f = o12_conj(f);
    const O12 e = final_exponentiation_o(f, ln);
    const bool one = o12_is_one(e, ln);
    if (!all_ok) return PST_INVALID;
    return one ? PST_OK : PST_FAIL;
  }
}

// ---- per-item G2 points in two kernels: the lines first, then the prepared-lines Miller loop ----
// With the G2 steps, the squaring and two sparse products in ONE loop the loop body outgrows the instruction cache (the
// fused sparse product made the single kernel slower, 8.97 -> 9.27 ms per 2^14 checks, while it made the prepared-lines
// kernel faster).  So the per-item path runs the G2 walk on its own -- all eight lanes busy, four per (P, Q) pair -- and
// leaves the 68 lines of each pair, already scaled by (x_P, y_P), in HBM: 45.7 KB per item, written and read once, item
// index fastest so that the eight items of a wave touch consecutive words.  The second kernel is the prepared-lines loop
// with per-item lines.  word(k, pair, coef, h, limb) of item i sits at lines[((((k * 2 + pair) * 3 + coef) * 2 + h) * 14 + limb) * n + i].
constexpr int OCT_LINE_WORDS = 2 * 3 * 2 * NLB;            // one Miller step of one item: two pairs x (c0, c1, c4) x two components
constexpr size_t oct_lines_words_per_item() { return (size_t)G2_LINES * OCT_LINE_WORDS; }
VRF_HD size_t oct_line_off(int k, int pair, int coef, int h) { return ((((size_t)k * 2 + pair) * 3 + coef) * 2 + h) * NLB; }

// item_flags: 4 words per item: ok0, skip0, ok1, skip1 (ok: both points of the pair decoded and lie on their curves;
// skip: one of them is the point at infinity -- the pair contributes 1)
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __attribute__((noinline))
#else
inline
#endif
void pairing_lines_oct(const uint32_t* g1, const uint32_t* g2, uint32_t* lines, uint32_t* item_flags, size_t n, size_t item, Ln ln) {
  const int pi = ln.j >> 1, sl = ln.j & 1;
  G1A P;
  G2A Q;
  bool i1, i2;
  G1Aff P1;
  const bool ok1 = g1_load(P1, i1, g1 + 24 * pi);
  P.x = P1.x; P.y = P1.y;
  const bool ok2 = g2_load_o(Q, i2, g2 + 48 * pi, ln);
  if (sl == 0 && ln.h == 0) {
    item_flags[4 * item + 2 * pi] = (ok1 && ok2) ? 1u : 0u;
    item_flags[4 * item + 2 * pi + 1] = (i1 || i2) ? 1u : 0u;
  }
  G2T T;
  T.X = Q.x; T.Y = Q.y; T.Z = o2_one(ln);
  auto store = [&](int k, const G2L& L) {
    // the two lanes of a pair that share a component split the three coefficients: s = 0 writes c0 and c1, s = 1 writes c4
    uint32_t* base = lines + item;
    if (sl == 0) {
      const size_t o0 = oct_line_off(k, pi, 0, ln.h), o1 = oct_line_off(k, pi, 1, ln.h);
#pragma unroll
      for (int i = 0; i < NLB; ++i) { base[(o0 + i) * n] = (uint32_t)L.c0.v[i]; base[(o1 + i) * n] = (uint32_t)L.c1.v[i]; }
    } else {
      const size_t o4 = oct_line_off(k, pi, 2, ln.h);
#pragma unroll
      for (int i = 0; i < NLB; ++i) base[(o4 + i) * n] = (uint32_t)L.c4.v[i];
    }
  };
  int k = 0;
#pragma unroll 1
  for (int bit = 62; bit >= 0; --bit) {
    store(k++, g2_double_o(T, P, ln));
    if ((X_ABS >> bit) & 1) store(k++, g2_add_o(T, Q, P, ln));
  }
}

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __attribute__((noinline))
#else
inline
#endif
uint32_t pairing_check2_oct_lines(const uint32_t* lines, const uint32_t* item_flags, size_t n, size_t item, Ln ln) {
  const bool all_ok = item_flags[4 * item] != 0 && item_flags[4 * item + 2] != 0;
  const bool skip0 = item_flags[4 * item + 1] != 0, skip1 = item_flags[4 * item + 3] != 0;
  const uint32_t* base = lines + item;
  auto load = [&](int k, int pair, int coef) {
    FpS x;
    const size_t o = oct_line_off(k, pair, coef, ln.h);
#pragma unroll
    for (int i = 0; i < NLB; ++i) x.v[i] = (int32_t)base[(o + i) * n];
    return x;
  };
  O12 f = o12_one(ln);
  int k = 0;
#pragma unroll 1
  for (int bit = 62; bit >= 0; --bit) {
    f = o12_sqr(f, ln);
    const int nsteps = ((X_ABS >> bit) & 1) ? 2 : 1;
#pragma unroll 1
    for (int step = 0; step < nsteps; ++step, ++k) {
#pragma unroll 1
      for (int i = 0; i < 2; ++i) {
        const FpS l0 = load(k, i, 0), l1 = load(k, i, 1), l4 = load(k, i, 2);
        const bool skip = i == 0 ? skip0 : skip1;
        if (!skip) f = o12_mul_by_014<true>(f, l0, l1, l4, ln);
      }
    }
  }
  f = o12_conj(f);
  const O12 e = final_exponentiation_o(f, ln);
  const bool one = o12_is_one(e, ln);
  if (!all_ok) return PST_INVALID;
  return one ? PST_OK : PST_FAIL;
}

// One item per 8 lanes against prepared G2 lines (pairing_prepare_g2_pair).  g1: 2 x 24 words.  Lane pair (column j) of
// pair j >> 1 scales c1 by x_P (j even) or c4 by y_P (j odd), component-wise.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __attribute__((noinline))
#else
inline
#endif
uint32_t pairing_check2_oct_prepared(const uint32_t* g1, const uint32_t* prep, Ln ln) {
  const int pi = ln.j >> 1;
  G1Aff P;
  bool i1;
  const bool ok1 = g1_load(P, i1, g1 + 24 * pi);
  const uint32_t* flags = prep + (size_t)2 * G2_LINES * G2_LINE_WORDS;
  const bool ok = ok1 && flags[2 * pi] != 0;
  const int my_skip = (i1 || flags[2 * pi + 1] != 0) ? 1 : 0;
  const int skip0 = xq_i32<QP_BC0>(my_skip), skip1 = xq_i32<QP_BC2>(my_skip);
  const bool all_ok = x_all8(ok, ln);
  const FpS scale = (ln.j & 1) ? P.y : P.x;
  const uint32_t* my_line = prep + (size_t)pi * G2_LINES * G2_LINE_WORDS + ((ln.j & 1) ? 4 * NLB : 2 * NLB) + ln.h * NLB;
  const uint32_t* line0 = prep + ln.h * NLB;
  O12 f = o12_one(ln);
  auto load = [](const uint32_t* src) {
    FpS x;
#pragma unroll
    for (int i = 0; i < NLB; ++i) x.v[i] = (int32_t)src[i];
    return x;
  };
#pragma unroll 1
  for (int bit = 62; bit >= 0; --bit) {
    f = o12_sqr(f, ln);
    const int nsteps = ((X_ABS >> bit) & 1) ? 2 : 1;
#pragma unroll 1
    for (int step = 0; step < nsteps; ++step) {
      const FpS scaled = fp_fit(fp_mul(load(my_line), scale));
#pragma unroll 1
      for (int i = 0; i < 2; ++i) {
        const FpS l0 = load(line0 + (size_t)i * G2_LINES * G2_LINE_WORDS);
        const FpS l1 = fp_select(i == 0, xq<QP_BC0>(scaled), xq<QP_BC2>(scaled));
        const FpS l4 = fp_select(i == 0, xq<QP_BC1>(scaled), xq<QP_BC3>(scaled));
        const bool skip = (i == 0 ? skip0 : skip1) != 0;
        if (!skip) f = o12_mul_by_014<true>(f, l0, l1, l4, ln);
      }
      my_line += G2_LINE_WORDS;
      line0 += G2_LINE_WORDS;
    }
  }
  f = o12_conj(f);
  const O12 e = final_exponentiation_o(f, ln);
  const bool one = o12_is_one(e, ln);
  if (!all_ok) return PST_INVALID;
  return one ? PST_OK : PST_FAIL;
}

// whole Fp12 (one-lane form) <-> this lane's share; for the self tests
VRF_HD O12 o12_scatter(const Fp12* full, const Ln& ln) {
  const Fp2 a0 = ln.j == 0 ? full->c0.c0 : (ln.j == 1 ? full->c0.c1 : full->c0.c2);
  const Fp2 a1 = ln.j == 0 ? full->c1.c0 : (ln.j == 1 ? full->c1.c1 : full->c1.c2);
  O12 o;
  o.c0 = ln.h ? a0.b : a0.a;
  o.c1 = ln.h ? a1.b : a1.a;
  return o;
}
// does this lane's share equal the matching component of `full`?  (all eight lanes get the combined answer)
VRF_HD bool o12_same(const O12& x, const Fp12* full, const Ln& ln) {
  const O12 w = o12_scatter(full, ln);
  return x_all8(ln.j == 3 || (fp_eq(x.c0, w.c0) && fp_eq(x.c1, w.c1)), ln);
}

}  // namespace oct
}  // namespace bls
