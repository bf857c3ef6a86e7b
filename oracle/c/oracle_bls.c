/* oracle_bls.c -- plain-C CPU restatement of the BLS12-381 pairing-product check.  TEST INFRASTRUCTURE ONLY.
 *
 * It is the checker and the reported CPU baseline ("port") of BASELINE.json configs[4], never the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle_vrf.so.
 *
 * What it restates: the tail of `ring::Verifier::verify` (/root/reference src/lib.rs:14 `ring`), i.e.
 * ark_ec::pairing::Pairing::{multi_miller_loop, final_exponentiation} on ark-bls12-381, from the published construction
 * (optimal ate pairing, |x| = 0xd201000000010000, x < 0; Fp2 = Fp[u]/(u^2+1), Fp6 = Fp2[v]/(v^3 - (1+u)),
 * Fp12 = Fp6[w]/(w^2 - v); M-type sextic twist) -- the same construction as oracle/bls_oracle.py, function by function
 * (miller_projective, final_exp_chain), which tests/test_bls_pairing.py holds it against on the fixture and on random items.
 * Parity status: unpinned by the reference (no vectors); pinned by algebra through the Python
 * oracle (bilinearity, e^r = 1, two formulations).
 *
 * Shape: 6 x 64-bit saturated limbs, unsigned __int128 CIOS Montgomery (R = 2^384) like ark_ff::MontBackend<_, 6>;
 * Karatsuba in Fp2 / Fp6 / Fp12; square-and-multiply for f^|x|.  A few milliseconds per two-pair check on one core.
 */
#include <pthread.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t v[6]; } fp;
typedef struct { fp a, b; } fp2;
typedef struct { fp2 c[3]; } fp6;
typedef struct { fp6 c0, c1; } fp12;

static const uint64_t BP[6] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
                               0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
static const uint64_t X_ABS = 0xd201000000010000ULL;
static uint64_t B_NINV;
static fp B_ONE, B_R2, B_INV2;
static fp2 GAMMA[6], TWIST_B;      /* xi^(i (p-1)/6); b' = 4 (1 + u) */
static pthread_once_t b_once = PTHREAD_ONCE_INIT;

/* ------------------------------------------------------------------ Fp */
static int b_cmp(const uint64_t a[6], const uint64_t b[6]) {
  for (int i = 5; i >= 0; --i) { if (a[i] < b[i]) return -1; if (a[i] > b[i]) return 1; }
  return 0;
}
static uint64_t b_add6(uint64_t r[6], const uint64_t a[6], const uint64_t b[6]) {
  u128 c = 0;
  for (int i = 0; i < 6; ++i) { c += (u128)a[i] + b[i]; r[i] = (uint64_t)c; c >>= 64; }
  return (uint64_t)c;
}
static uint64_t b_sub6(uint64_t r[6], const uint64_t a[6], const uint64_t b[6]) {
  uint64_t borrow = 0;
  for (int i = 0; i < 6; ++i) { u128 t = (u128)a[i] - b[i] - borrow; r[i] = (uint64_t)t; borrow = (uint64_t)(t >> 64) & 1; }
  return borrow;
}
static void fp_add(fp* r, const fp* a, const fp* b) {
  uint64_t t[6], d[6];
  uint64_t c = b_add6(t, a->v, b->v), bo = b_sub6(d, t, BP);
  memcpy(r->v, (c || !bo) ? d : t, 48);
}
static void fp_sub(fp* r, const fp* a, const fp* b) {
  uint64_t t[6];
  if (b_sub6(t, a->v, b->v)) b_add6(t, t, BP);
  memcpy(r->v, t, 48);
}
static void fp_neg(fp* r, const fp* a) { fp z; memset(&z, 0, sizeof z); fp_sub(r, &z, a); }
static void fp_mul(fp* r, const fp* a, const fp* b) {
  uint64_t t[8] = {0};
  for (int i = 0; i < 6; ++i) {
    u128 c = 0;
    for (int j = 0; j < 6; ++j) { c += (u128)a->v[j] * b->v[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
    c += t[6]; t[6] = (uint64_t)c; t[7] = (uint64_t)(c >> 64);
    uint64_t m = t[0] * B_NINV;
    c = (u128)m * BP[0] + t[0]; c >>= 64;
    for (int j = 1; j < 6; ++j) { c += (u128)m * BP[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
    c += t[6]; t[5] = (uint64_t)c; t[6] = t[7] + (uint64_t)(c >> 64);
  }
  uint64_t d[6];
  uint64_t bo = b_sub6(d, t, BP);
  memcpy(r->v, (t[6] || !bo) ? d : t, 48);
}
static void fp_sqr(fp* r, const fp* a) { fp_mul(r, a, a); }
static int fp_is_zero(const fp* a) { uint64_t o = 0; for (int i = 0; i < 6; ++i) o |= a->v[i]; return o == 0; }
static int fp_eq(const fp* a, const fp* b) { return b_cmp(a->v, b->v) == 0; }
static void fp_from_int(fp* r, const uint64_t a[6]) { fp t; memcpy(t.v, a, 48); fp_mul(r, &t, &B_R2); }
static void fp_to_int(uint64_t r[6], const fp* a) { fp one = {{1, 0, 0, 0, 0, 0}}, t; fp_mul(&t, a, &one); memcpy(r, t.v, 48); }
static void fp_pow(fp* r, const fp* a, const uint64_t e[6]) {
  fp acc = B_ONE;
  for (int i = 383; i >= 0; --i) { fp_sqr(&acc, &acc); if ((e[i >> 6] >> (i & 63)) & 1) fp_mul(&acc, &acc, a); }
  *r = acc;
}
static void fp_inv(fp* r, const fp* a) { uint64_t e[6], two[6] = {2, 0, 0, 0, 0, 0}; b_sub6(e, BP, two); fp_pow(r, a, e); }

/* ------------------------------------------------------------------ Fp2 = Fp[u]/(u^2 + 1) */
static void f2_add(fp2* r, const fp2* a, const fp2* b) { fp_add(&r->a, &a->a, &b->a); fp_add(&r->b, &a->b, &b->b); }
static void f2_sub(fp2* r, const fp2* a, const fp2* b) { fp_sub(&r->a, &a->a, &b->a); fp_sub(&r->b, &a->b, &b->b); }
static void f2_neg(fp2* r, const fp2* a) { fp_neg(&r->a, &a->a); fp_neg(&r->b, &a->b); }
static void f2_dbl(fp2* r, const fp2* a) { f2_add(r, a, a); }
static void f2_mul(fp2* r, const fp2* x, const fp2* y) {
  fp t0, t1, t2, s0, s1;
  fp_mul(&t0, &x->a, &y->a); fp_mul(&t1, &x->b, &y->b);
  fp_add(&s0, &x->a, &x->b); fp_add(&s1, &y->a, &y->b); fp_mul(&t2, &s0, &s1);
  fp_sub(&r->a, &t0, &t1);
  fp_sub(&t2, &t2, &t0); fp_sub(&r->b, &t2, &t1);
}
static void f2_sqr(fp2* r, const fp2* x) {
  fp s, d, m;
  fp_add(&s, &x->a, &x->b); fp_sub(&d, &x->a, &x->b); fp_mul(&m, &x->a, &x->b);
  fp_mul(&r->a, &s, &d); fp_add(&r->b, &m, &m);
}
static void f2_mul_fp(fp2* r, const fp2* x, const fp* k) { fp_mul(&r->a, &x->a, k); fp_mul(&r->b, &x->b, k); }
static void f2_conj(fp2* r, const fp2* x) { r->a = x->a; fp_neg(&r->b, &x->b); }
static void f2_mul_xi(fp2* r, const fp2* x) { fp a, b; fp_sub(&a, &x->a, &x->b); fp_add(&b, &x->a, &x->b); r->a = a; r->b = b; }
static void f2_inv(fp2* r, const fp2* x) {
  fp n, t, d; fp_sqr(&n, &x->a); fp_sqr(&t, &x->b); fp_add(&n, &n, &t); fp_inv(&d, &n);
  fp_mul(&r->a, &x->a, &d); fp_mul(&t, &x->b, &d); fp_neg(&r->b, &t);
}
static int f2_is_zero(const fp2* x) { return fp_is_zero(&x->a) && fp_is_zero(&x->b); }
static int f2_eq(const fp2* x, const fp2* y) { return fp_eq(&x->a, &y->a) && fp_eq(&x->b, &y->b); }
static void f2_zero(fp2* r) { memset(r, 0, sizeof *r); }
static void f2_one(fp2* r) { r->a = B_ONE; memset(&r->b, 0, sizeof r->b); }

/* ------------------------------------------------------------------ Fp6 = Fp2[v]/(v^3 - xi) */
static void f6_add(fp6* r, const fp6* a, const fp6* b) { for (int i = 0; i < 3; ++i) f2_add(&r->c[i], &a->c[i], &b->c[i]); }
static void f6_sub(fp6* r, const fp6* a, const fp6* b) { for (int i = 0; i < 3; ++i) f2_sub(&r->c[i], &a->c[i], &b->c[i]); }
static void f6_neg(fp6* r, const fp6* a) { for (int i = 0; i < 3; ++i) f2_neg(&r->c[i], &a->c[i]); }
static void f6_mul(fp6* r, const fp6* a, const fp6* b) {      /* Karatsuba: 6 Fp2 products */
  fp2 v0, v1, v2, t0, t1, t2, s0, s1;
  f2_mul(&v0, &a->c[0], &b->c[0]); f2_mul(&v1, &a->c[1], &b->c[1]); f2_mul(&v2, &a->c[2], &b->c[2]);
  f2_add(&s0, &a->c[1], &a->c[2]); f2_add(&s1, &b->c[1], &b->c[2]); f2_mul(&t0, &s0, &s1);
  f2_sub(&t0, &t0, &v1); f2_sub(&t0, &t0, &v2); f2_mul_xi(&t0, &t0); f2_add(&t0, &t0, &v0);          /* c0 */
  f2_add(&s0, &a->c[0], &a->c[1]); f2_add(&s1, &b->c[0], &b->c[1]); f2_mul(&t1, &s0, &s1);
  f2_sub(&t1, &t1, &v0); f2_sub(&t1, &t1, &v1); f2_mul_xi(&s0, &v2); f2_add(&t1, &t1, &s0);           /* c1 */
  f2_add(&s0, &a->c[0], &a->c[2]); f2_add(&s1, &b->c[0], &b->c[2]); f2_mul(&t2, &s0, &s1);
  f2_sub(&t2, &t2, &v0); f2_sub(&t2, &t2, &v2); f2_add(&t2, &t2, &v1);                                /* c2 */
  r->c[0] = t0; r->c[1] = t1; r->c[2] = t2;
}
static void f6_mul_v(fp6* r, const fp6* a) { fp2 t; f2_mul_xi(&t, &a->c[2]); fp2 c0 = a->c[0], c1 = a->c[1]; r->c[0] = t; r->c[1] = c0; r->c[2] = c1; }
static void f6_inv(fp6* r, const fp6* a) {
  fp2 t0, t1, t2, s, d;
  f2_sqr(&t0, &a->c[0]); f2_mul(&s, &a->c[1], &a->c[2]); f2_mul_xi(&s, &s); f2_sub(&t0, &t0, &s);
  f2_sqr(&t1, &a->c[2]); f2_mul_xi(&t1, &t1); f2_mul(&s, &a->c[0], &a->c[1]); f2_sub(&t1, &t1, &s);
  f2_sqr(&t2, &a->c[1]); f2_mul(&s, &a->c[0], &a->c[2]); f2_sub(&t2, &t2, &s);
  fp2 u0, u1; f2_mul(&u0, &a->c[2], &t1); f2_mul(&u1, &a->c[1], &t2); f2_add(&u0, &u0, &u1); f2_mul_xi(&u0, &u0);
  f2_mul(&d, &a->c[0], &t0); f2_add(&d, &d, &u0); f2_inv(&d, &d);
  f2_mul(&r->c[0], &t0, &d); f2_mul(&r->c[1], &t1, &d); f2_mul(&r->c[2], &t2, &d);
}
static int f6_eq(const fp6* a, const fp6* b) { return f2_eq(&a->c[0], &b->c[0]) && f2_eq(&a->c[1], &b->c[1]) && f2_eq(&a->c[2], &b->c[2]); }

/* ------------------------------------------------------------------ Fp12 = Fp6[w]/(w^2 - v) */
static void f12_one(fp12* r) { memset(r, 0, sizeof *r); f2_one(&r->c0.c[0]); }
static void f12_mul(fp12* r, const fp12* a, const fp12* b) {  /* Karatsuba: 3 Fp6 products */
  fp6 v0, v1, s0, s1, t;
  f6_mul(&v0, &a->c0, &b->c0); f6_mul(&v1, &a->c1, &b->c1);
  f6_add(&s0, &a->c0, &a->c1); f6_add(&s1, &b->c0, &b->c1); f6_mul(&t, &s0, &s1);
  f6_sub(&t, &t, &v0); f6_sub(&t, &t, &v1);
  f6_mul_v(&s0, &v1); f6_add(&r->c0, &v0, &s0);
  r->c1 = t;
}
static void f12_sqr(fp12* r, const fp12* a) { f12_mul(r, a, a); }
static void f12_conj(fp12* r, const fp12* a) { r->c0 = a->c0; f6_neg(&r->c1, &a->c1); }
static void f12_inv(fp12* r, const fp12* a) {
  fp6 t0, t1, d;
  f6_mul(&t0, &a->c0, &a->c0); f6_mul(&t1, &a->c1, &a->c1); f6_mul_v(&t1, &t1); f6_sub(&d, &t0, &t1); f6_inv(&d, &d);
  f6_mul(&r->c0, &a->c0, &d); f6_mul(&t0, &a->c1, &d); f6_neg(&r->c1, &t0);
}
static int f12_is_one(const fp12* a) { fp12 o; f12_one(&o); return f6_eq(&a->c0, &o.c0) && f6_eq(&a->c1, &o.c1); }
/* c0 + c1 v + c4 v w */
static void f12_from_014(fp12* r, const fp2* c0, const fp2* c1, const fp2* c4) {
  memset(r, 0, sizeof *r); r->c0.c[0] = *c0; r->c0.c[1] = *c1; r->c1.c[1] = *c4;
}
/* f^p: coefficients of w^0..w^5 are (c0.c0, c1.c0, c0.c1, c1.c1, c0.c2, c1.c2); each conjugated and scaled by xi^(i (p-1)/6) */
static void f12_frob(fp12* r, const fp12* f) {
  const fp2* cs[6] = {&f->c0.c[0], &f->c1.c[0], &f->c0.c[1], &f->c1.c[1], &f->c0.c[2], &f->c1.c[2]};
  fp2 out[6];
  for (int i = 0; i < 6; ++i) { fp2 t; f2_conj(&t, cs[i]); f2_mul(&out[i], &t, &GAMMA[i]); }
  r->c0.c[0] = out[0]; r->c1.c[0] = out[1]; r->c0.c[1] = out[2]; r->c1.c[1] = out[3]; r->c0.c[2] = out[4]; r->c1.c[2] = out[5];
}
static void f12_exp_by_x(fp12* r, const fp12* f) {           /* f^x, x = -|x|: f in the cyclotomic subgroup, inverse = conjugate */
  fp12 acc = *f;
  for (int i = 62; i >= 0; --i) { f12_sqr(&acc, &acc); if ((X_ABS >> i) & 1) f12_mul(&acc, &acc, f); }
  f12_conj(r, &acc);
}

/* ------------------------------------------------------------------ init */
static void b_init(void) {
  uint64_t x = 1;
  for (int i = 0; i < 6; ++i) x *= 2 - BP[0] * x;
  B_NINV = (uint64_t)0 - x;
  uint64_t t[6] = {1, 0, 0, 0, 0, 0};
  for (int i = 0; i < 768; ++i) {
    uint64_t d[6]; uint64_t c = b_add6(t, t, t), bo = b_sub6(d, t, BP);
    if (c || !bo) memcpy(t, d, 48);
    if (i == 383) memcpy(B_ONE.v, t, 48);
  }
  memcpy(B_R2.v, t, 48);
  fp two; fp_add(&two, &B_ONE, &B_ONE); fp_inv(&B_INV2, &two);
  /* (p - 1) / 6 by long division, then xi^((p-1)/6) */
  uint64_t e[6], pm1[6], one[6] = {1, 0, 0, 0, 0, 0};
  b_sub6(pm1, BP, one);
  u128 rem = 0;
  for (int i = 5; i >= 0; --i) { u128 cur = (rem << 64) | pm1[i]; e[i] = (uint64_t)(cur / 6); rem = cur % 6; }
  fp2 xi; xi.a = B_ONE; xi.b = B_ONE;
  fp2 g; f2_one(&g);
  for (int i = 383; i >= 0; --i) { f2_sqr(&g, &g); if ((e[i >> 6] >> (i & 63)) & 1) f2_mul(&g, &g, &xi); }
  f2_one(&GAMMA[0]);
  for (int i = 1; i < 6; ++i) f2_mul(&GAMMA[i], &GAMMA[i - 1], &g);
  fp four; fp_add(&four, &two, &two);
  TWIST_B.a = four; TWIST_B.b = four;
}
static void b_ensure(void) { pthread_once(&b_once, b_init); }

/* ------------------------------------------------------------------ points: wire format <-> field elements */
static int load48(uint64_t w[6], const uint8_t* b) {         /* returns 1 if < p */
  for (int i = 0; i < 6; ++i) { uint64_t x = 0; for (int j = 7; j >= 0; --j) x = (x << 8) | b[8 * i + j]; w[i] = x; }
  return b_cmp(w, BP) < 0;
}
static void store48(uint8_t* b, const fp* a) {
  uint64_t w[6]; fp_to_int(w, a);
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 8; ++j) b[8 * i + j] = (uint8_t)(w[i] >> (8 * j));
}
static int all_zero(const uint8_t* b, size_t n) { uint8_t o = 0; for (size_t i = 0; i < n; ++i) o |= b[i]; return o == 0; }
typedef struct { fp x, y; int inf; } g1pt;
typedef struct { fp2 x, y; int inf; } g2pt;
/* 0 ok, 2 invalid (coordinate >= p or off the curve); an all-zero encoding is the point at infinity */
static int g1_decode(g1pt* p, const uint8_t b[96]) {
  p->inf = all_zero(b, 96);
  if (p->inf) return 0;
  uint64_t w[6]; int ok = 1;
  ok &= load48(w, b); fp_from_int(&p->x, w); ok &= load48(w, b + 48); fp_from_int(&p->y, w);
  if (!ok) return 2;
  fp l, r, four; fp_sqr(&l, &p->y); fp_sqr(&r, &p->x); fp_mul(&r, &r, &p->x);
  fp_add(&four, &B_ONE, &B_ONE); fp_add(&four, &four, &four); fp_add(&r, &r, &four);
  return fp_eq(&l, &r) ? 0 : 2;
}
static int g2_decode(g2pt* p, const uint8_t b[192]) {
  p->inf = all_zero(b, 192);
  if (p->inf) return 0;
  uint64_t w[6]; int ok = 1;
  ok &= load48(w, b); fp_from_int(&p->x.a, w); ok &= load48(w, b + 48); fp_from_int(&p->x.b, w);
  ok &= load48(w, b + 96); fp_from_int(&p->y.a, w); ok &= load48(w, b + 144); fp_from_int(&p->y.b, w);
  if (!ok) return 2;
  fp2 l, r; f2_sqr(&l, &p->y); f2_sqr(&r, &p->x); f2_mul(&r, &r, &p->x); f2_add(&r, &r, &TWIST_B);
  return f2_eq(&l, &r) ? 0 : 2;
}

/* ------------------------------------------------------------------ Miller loop (homogeneous projective, as bls_oracle.py) */
typedef struct { fp2 X, Y, Z; } g2proj;
/* T <- 2T; line = (Y^2 - 3b'Z^2) + (-3X^2 x_P) v + (2YZ y_P) v w */
static void g2_double_step(g2proj* T, fp2* c0, fp2* c1, fp2* c4) {
  fp2 a, b, c, e, f, g, h, j, t;
  f2_mul(&a, &T->X, &T->Y); f2_mul_fp(&a, &a, &B_INV2);
  f2_sqr(&b, &T->Y); f2_sqr(&c, &T->Z);
  f2_dbl(&t, &c); f2_add(&t, &t, &c); f2_mul(&e, &TWIST_B, &t);           /* e = b' * 3c */
  f2_dbl(&f, &e); f2_add(&f, &f, &e);                                     /* f = 3e */
  f2_add(&g, &b, &f); f2_mul_fp(&g, &g, &B_INV2);
  f2_add(&h, &T->Y, &T->Z); f2_sqr(&h, &h); f2_add(&t, &b, &c); f2_sub(&h, &h, &t);     /* 2YZ */
  f2_sqr(&j, &T->X);
  fp2 X3, Y3, Z3, e2;
  f2_sub(&t, &b, &f); f2_mul(&X3, &a, &t);
  f2_sqr(&Y3, &g); f2_sqr(&e2, &e); f2_dbl(&t, &e2); f2_add(&t, &t, &e2); f2_sub(&Y3, &Y3, &t);
  f2_mul(&Z3, &b, &h);
  f2_sub(c0, &b, &e);
  f2_dbl(&t, &j); f2_add(&t, &t, &j); f2_neg(c1, &t);
  *c4 = h;
  T->X = X3; T->Y = Y3; T->Z = Z3;
}
/* T <- T + Q; line = (theta x_Q - lam y_Q) + (-theta x_P) v + (lam y_P) v w */
static void g2_add_step(g2proj* T, const g2pt* Q, fp2* c0, fp2* c1, fp2* c4) {
  fp2 theta, lam, c, d, e, f, g, h, t, X3, Y3, Z3;
  f2_mul(&t, &Q->y, &T->Z); f2_sub(&theta, &T->Y, &t);
  f2_mul(&t, &Q->x, &T->Z); f2_sub(&lam, &T->X, &t);
  f2_sqr(&c, &theta); f2_sqr(&d, &lam); f2_mul(&e, &lam, &d); f2_mul(&f, &T->Z, &c); f2_mul(&g, &T->X, &d);
  f2_add(&h, &e, &f); f2_dbl(&t, &g); f2_sub(&h, &h, &t);
  f2_mul(&X3, &lam, &h);
  f2_sub(&t, &g, &h); f2_mul(&Y3, &theta, &t); f2_mul(&t, &e, &T->Y); f2_sub(&Y3, &Y3, &t);
  f2_mul(&Z3, &T->Z, &e);
  fp2 u0, u1; f2_mul(&u0, &theta, &Q->x); f2_mul(&u1, &lam, &Q->y); f2_sub(c0, &u0, &u1);
  f2_neg(c1, &theta);
  *c4 = lam;
  T->X = X3; T->Y = Y3; T->Z = Z3;
}
static void miller2(fp12* f, const g1pt P[2], const g2pt Q[2]) {
  int use[2]; g2proj T[2];
  for (int i = 0; i < 2; ++i) { use[i] = !P[i].inf && !Q[i].inf; T[i].X = Q[i].x; T[i].Y = Q[i].y; f2_one(&T[i].Z); }
  f12_one(f);
  for (int bit = 62; bit >= 0; --bit) {
    f12_sqr(f, f);
    for (int i = 0; i < 2; ++i) if (use[i]) {
      fp2 c0, c1, c4; fp12 l;
      g2_double_step(&T[i], &c0, &c1, &c4);
      f2_mul_fp(&c1, &c1, &P[i].x); f2_mul_fp(&c4, &c4, &P[i].y);
      f12_from_014(&l, &c0, &c1, &c4); f12_mul(f, f, &l);
    }
    if ((X_ABS >> bit) & 1)
      for (int i = 0; i < 2; ++i) if (use[i]) {
        fp2 c0, c1, c4; fp12 l;
        g2_add_step(&T[i], &Q[i], &c0, &c1, &c4);
        f2_mul_fp(&c1, &c1, &P[i].x); f2_mul_fp(&c4, &c4, &P[i].y);
        f12_from_014(&l, &c0, &c1, &c4); f12_mul(f, f, &l);
      }
  }
  f12_conj(f, f);                                 /* x < 0 */
}
/* f^(3 (p^12-1)/r): easy part, then 3 lambda = (x-1)^2 (x+p) (x^2+p^2-1) + 3 */
static void final_exp_chain(fp12* r, const fp12* f) {
  fp12 f1, f2, y0, y1, y2, y3, t, u;
  f12_conj(&t, f); f12_inv(&u, f); f12_mul(&f1, &t, &u);                    /* ^(p^6 - 1) */
  f12_frob(&t, &f1); f12_frob(&t, &t); f12_mul(&f2, &t, &f1);              /* ^(p^2 + 1) */
  f12_exp_by_x(&t, &f2); f12_conj(&u, &f2); f12_mul(&y0, &t, &u);          /* ^(x - 1) */
  f12_exp_by_x(&t, &y0); f12_conj(&u, &y0); f12_mul(&y1, &t, &u);          /* ^(x - 1)^2 */
  f12_exp_by_x(&t, &y1); f12_frob(&u, &y1); f12_mul(&y2, &t, &u);          /* ^(x + p) */
  f12_exp_by_x(&t, &y2); f12_exp_by_x(&t, &t); f12_frob(&u, &y2); f12_frob(&u, &u); f12_mul(&y3, &t, &u);
  f12_conj(&u, &y2); f12_mul(&y3, &y3, &u);                                  /* ^(x^2 + p^2 - 1) */
  f12_sqr(&t, &f2); f12_mul(&t, &t, &f2); f12_mul(r, &y3, &t);
}

/* ------------------------------------------------------------------ exported API */
/* e(P0, Q0) e(P1, Q1) == 1 for one item (g1: 192 B, g2: 384 B, include/vrfhip.h vrfhip_pairing_check_batch):
 * 0 = product is one, 1 = it is not, 2 = InvalidData (coordinate >= p or point off its curve). */
int oracle_pairing_check2(const uint8_t g1[192], const uint8_t g2[384]) {
  b_ensure();
  g1pt P[2]; g2pt Q[2];
  int bad = 0;
  for (int i = 0; i < 2; ++i) { bad |= g1_decode(&P[i], g1 + 96 * i); bad |= g2_decode(&Q[i], g2 + 192 * i); }
  if (bad) return 2;
  fp12 f, e; miller2(&f, P, Q); final_exp_chain(&e, &f);
  return f12_is_one(&e) ? 0 : 1;
}
typedef struct { size_t lo, hi; const uint8_t *g1, *g2; size_t g2_stride; uint8_t* st; } bjob;
static void* b_run(void* arg) {
  bjob* j = (bjob*)arg;
  for (size_t i = j->lo; i < j->hi; ++i) j->st[i] = (uint8_t)oracle_pairing_check2(j->g1 + 192 * i, j->g2 + j->g2_stride * i);
  return NULL;
}
/* g2_stride: 384 (a pair per item) or 0 (one shared pair) */
void oracle_pairing_check2_batch(size_t n, const uint8_t* g1, const uint8_t* g2, size_t g2_stride, uint8_t* status, int threads) {
  b_ensure();
  if (threads < 1) threads = 1;
  if ((size_t)threads > n) threads = n ? (int)n : 1;
  pthread_t th[64]; bjob jobs[64];
  if (threads > 64) threads = 64;
  for (int t = 0; t < threads; ++t) {
    jobs[t] = (bjob){n * (size_t)t / (size_t)threads, n * (size_t)(t + 1) / (size_t)threads, g1, g2, g2_stride, status};
    if (threads == 1) b_run(&jobs[t]); else pthread_create(&th[t], NULL, b_run, &jobs[t]);
  }
  if (threads > 1) for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
}

/* k * P on G1 / G2 (affine double-and-add, test-vector construction): points in the wire format, k 32 B little-endian.
 * Returns 2 if the input does not decode. */
static void g1_add_aff(g1pt* r, const g1pt* a, const g1pt* b) {
  if (a->inf) { *r = *b; return; }
  if (b->inf) { *r = *a; return; }
  fp lam, t, u;
  if (fp_eq(&a->x, &b->x)) {
    fp s; fp_add(&s, &a->y, &b->y);
    if (fp_is_zero(&s)) { memset(r, 0, sizeof *r); r->inf = 1; return; }
    fp_sqr(&t, &a->x); fp_add(&u, &t, &t); fp_add(&t, &u, &t); fp_add(&u, &a->y, &a->y); fp_inv(&u, &u); fp_mul(&lam, &t, &u);
  } else {
    fp_sub(&t, &b->y, &a->y); fp_sub(&u, &b->x, &a->x); fp_inv(&u, &u); fp_mul(&lam, &t, &u);
  }
  g1pt o; o.inf = 0;
  fp_sqr(&o.x, &lam); fp_sub(&o.x, &o.x, &a->x); fp_sub(&o.x, &o.x, &b->x);
  fp_sub(&t, &a->x, &o.x); fp_mul(&o.y, &lam, &t); fp_sub(&o.y, &o.y, &a->y);
  *r = o;
}
static void g2_add_aff(g2pt* r, const g2pt* a, const g2pt* b) {
  if (a->inf) { *r = *b; return; }
  if (b->inf) { *r = *a; return; }
  fp2 lam, t, u;
  if (f2_eq(&a->x, &b->x)) {
    fp2 s; f2_add(&s, &a->y, &b->y);
    if (f2_is_zero(&s)) { memset(r, 0, sizeof *r); r->inf = 1; return; }
    f2_sqr(&t, &a->x); f2_dbl(&u, &t); f2_add(&t, &u, &t); f2_dbl(&u, &a->y); f2_inv(&u, &u); f2_mul(&lam, &t, &u);
  } else {
    f2_sub(&t, &b->y, &a->y); f2_sub(&u, &b->x, &a->x); f2_inv(&u, &u); f2_mul(&lam, &t, &u);
  }
  g2pt o; o.inf = 0;
  f2_sqr(&o.x, &lam); f2_sub(&o.x, &o.x, &a->x); f2_sub(&o.x, &o.x, &b->x);
  f2_sub(&t, &a->x, &o.x); f2_mul(&o.y, &lam, &t); f2_sub(&o.y, &o.y, &a->y);
  *r = o;
}
/* Jacobian double / mixed add for y^2 = x^3 + b (a = 0), written once over a field given by its operations: k * P costs
 * ~4 k field products instead of 256 inversions (the soak tests build thousands of distinct items). */
#define JAC_MUL(NAME, FE, PT, MUL, SQR, ADD, SUB, INV, ISZERO, EQ, ONE)                                              \
  static void NAME(PT* r, const PT* p, const uint8_t k[32]) {                                                         \
    FE X, Y, Z; int inf = 1;                                                                                          \
    memset(&X, 0, sizeof X); memset(&Y, 0, sizeof Y); memset(&Z, 0, sizeof Z);                                        \
    if (p->inf) { *r = *p; return; }                                                                                  \
    for (int i = 255; i >= 0; --i) {                                                                                  \
      if (!inf) { /* dbl-2009-l */                                                                                    \
        FE A, B, C, D, E, F, t, X3, Y3, Z3;                                                                           \
        SQR(&A, &X); SQR(&B, &Y); SQR(&C, &B);                                                                        \
        ADD(&t, &X, &B); SQR(&t, &t); SUB(&t, &t, &A); SUB(&t, &t, &C); ADD(&D, &t, &t);                              \
        ADD(&E, &A, &A); ADD(&E, &E, &A); SQR(&F, &E);                                                                \
        SUB(&X3, &F, &D); SUB(&X3, &X3, &D);                                                                          \
        SUB(&t, &D, &X3); MUL(&Y3, &E, &t); ADD(&t, &C, &C); ADD(&t, &t, &t); ADD(&t, &t, &t); SUB(&Y3, &Y3, &t);     \
        MUL(&Z3, &Y, &Z); ADD(&Z3, &Z3, &Z3);                                                                         \
        X = X3; Y = Y3; Z = Z3;                                                                                       \
      }                                                                                                               \
      if ((k[i >> 3] >> (i & 7)) & 1) {                                                                               \
        if (inf) { X = p->x; Y = p->y; ONE(&Z); inf = 0; continue; }                                                  \
        /* madd: U2 = x Z^2, S2 = y Z^3 */                                                                            \
        FE Z2, U2, S2, H, Rr, t, H2, H3, V, X3, Y3, Z3;                                                               \
        SQR(&Z2, &Z); MUL(&U2, &p->x, &Z2); MUL(&S2, &p->y, &Z2); MUL(&S2, &S2, &Z);                                  \
        SUB(&H, &U2, &X); SUB(&Rr, &S2, &Y);                                                                          \
        if (ISZERO(&H)) {                                                                                             \
          if (ISZERO(&Rr)) { /* P + P never happens for k < r with P of order r except through doubling */            \
            FE A, B, C, D, E, F, X4, Y4, Z4;                                                                          \
            SQR(&A, &X); SQR(&B, &Y); SQR(&C, &B);                                                                    \
            ADD(&t, &X, &B); SQR(&t, &t); SUB(&t, &t, &A); SUB(&t, &t, &C); ADD(&D, &t, &t);                          \
            ADD(&E, &A, &A); ADD(&E, &E, &A); SQR(&F, &E);                                                            \
            SUB(&X4, &F, &D); SUB(&X4, &X4, &D);                                                                      \
            SUB(&t, &D, &X4); MUL(&Y4, &E, &t); ADD(&t, &C, &C); ADD(&t, &t, &t); ADD(&t, &t, &t); SUB(&Y4, &Y4, &t); \
            MUL(&Z4, &Y, &Z); ADD(&Z4, &Z4, &Z4);                                                                     \
            X = X4; Y = Y4; Z = Z4;                                                                                   \
          } else { inf = 1; }                                                                                         \
          continue;                                                                                                   \
        }                                                                                                             \
        SQR(&H2, &H); MUL(&H3, &H2, &H); MUL(&V, &X, &H2);                                                            \
        SQR(&X3, &Rr); SUB(&X3, &X3, &H3); SUB(&X3, &X3, &V); SUB(&X3, &X3, &V);                                      \
        SUB(&t, &V, &X3); MUL(&Y3, &Rr, &t); MUL(&t, &Y, &H3); SUB(&Y3, &Y3, &t);                                     \
        MUL(&Z3, &Z, &H);                                                                                             \
        X = X3; Y = Y3; Z = Z3;                                                                                       \
      }                                                                                                               \
    }                                                                                                                 \
    memset(r, 0, sizeof *r);                                                                                          \
    if (inf || ISZERO(&Z)) { r->inf = 1; return; }                                                                    \
    FE zi, zi2, zi3; INV(&zi, &Z); SQR(&zi2, &zi); MUL(&zi3, &zi2, &zi);                                              \
    MUL(&r->x, &X, &zi2); MUL(&r->y, &Y, &zi3); r->inf = 0;                                                           \
  }
static void fp_set_one(fp* r) { *r = B_ONE; }
JAC_MUL(g1_mul_jac, fp, g1pt, fp_mul, fp_sqr, fp_add, fp_sub, fp_inv, fp_is_zero, fp_eq, fp_set_one)
JAC_MUL(g2_mul_jac, fp2, g2pt, f2_mul, f2_sqr, f2_add, f2_sub, f2_inv, f2_is_zero, f2_eq, f2_one)

int oracle_g1_mul(const uint8_t k[32], const uint8_t in[96], uint8_t out[96]) {
  b_ensure();
  g1pt p, acc;
  if (g1_decode(&p, in)) return 2;
  g1_mul_jac(&acc, &p, k);
  memset(out, 0, 96);
  if (!acc.inf) { store48(out, &acc.x); store48(out + 48, &acc.y); }
  return 0;
}
int oracle_g2_mul(const uint8_t k[32], const uint8_t in[192], uint8_t out[192]) {
  b_ensure();
  g2pt p, acc;
  if (g2_decode(&p, in)) return 2;
  g2_mul_jac(&acc, &p, k);
  memset(out, 0, 192);
  if (!acc.inf) { store48(out, &acc.x.a); store48(out + 48, &acc.x.b); store48(out + 96, &acc.y.a); store48(out + 144, &acc.y.b); }
  return 0;
}
/* out = a + b on G1 (affine law, every special case) -- the soak tests' tampering ("swapped", "shifted" points) */
int oracle_g1_add(const uint8_t a[96], const uint8_t b[96], uint8_t out[96]) {
  b_ensure();
  g1pt p, q, r;
  if (g1_decode(&p, a) || g1_decode(&q, b)) return 2;
  g1_add_aff(&r, &p, &q);
  memset(out, 0, 96);
  if (!r.inf) { store48(out, &r.x); store48(out + 48, &r.y); }
  return 0;
}
