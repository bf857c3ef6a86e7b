/* oracle_p256.c -- CPU restatement in plain C (TEST INFRASTRUCTURE ONLY) of the secp256r1 suite of ark-vrf
 * (`suites::secp256r1`, /root/reference src/lib.rs:14; "P256_SHA256_TAI" = RFC 9381 ECVRF-P256-SHA256-TAI).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load this; the product (libvrfhip.so) never does.
 * It follows oracle/sw_oracle.py function by function -- which RFC 9381 Appendix B.1 pins (tests/test_secp256r1.py holds the
 * two against each other and against the vectors) -- with the arithmetic a CPU library would use: 4 x 64-bit Montgomery
 * limbs, Jacobian coordinates, double-and-add.  It exists because the Python oracle manages a few proofs per second and the
 * parity samples and the CPU baseline of the bench want thousands.
 *
 * [ref src/lib.rs:14 `ietf::{Prover, Verifier}`, `utils::{hash_to_curve_tai_rfc_9381, nonce_rfc_6979, challenge_rfc_9381,
 * point_to_hash_rfc_9381}`, `codec::Sec1Codec`; src/lib.rs:16 `Secret::from_seed`] */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t m[4]; uint64_t ninv; uint64_t r2[4]; uint64_t one[4]; } field;
typedef struct { uint64_t v[4]; } fe;          /* Montgomery form */

static const uint64_t P_M[4] = {0xffffffffffffffffULL, 0x00000000ffffffffULL, 0x0000000000000000ULL, 0xffffffff00000001ULL};
static const uint64_t N_M[4] = {0xf3b9cac2fc632551ULL, 0xbce6faada7179e84ULL, 0xffffffffffffffffULL, 0xffffffff00000000ULL};
static const uint64_t B_I[4] = {0x3bce3c3e27d2604bULL, 0x651d06b0cc53b0f6ULL, 0xb3ebbd55769886bcULL, 0x5ac635d8aa3a93e7ULL};
static const uint64_t GX_I[4] = {0xf4a13945d898c296ULL, 0x77037d812deb33a0ULL, 0xf8bce6e563a440f2ULL, 0x6b17d1f2e12c4247ULL};
static const uint64_t GY_I[4] = {0xcbb6406837bf51f5ULL, 0x2bce33576b315eceULL, 0x8ee7eb4a7c0f9e16ULL, 0x4fe342e2fe1a7f9bULL};
static field FP, FN;
static fe B_M, GX_M, GY_M, THREE_M;
static pthread_once_t g_once = PTHREAD_ONCE_INIT;

static int cmp4(const uint64_t a[4], const uint64_t b[4]) {
  for (int i = 3; i >= 0; --i) if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
  return 0;
}
static uint64_t add4(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
  u128 c = 0;
  for (int i = 0; i < 4; ++i) { c += (u128)a[i] + b[i]; r[i] = (uint64_t)c; c >>= 64; }
  return (uint64_t)c;
}
static uint64_t sub4(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
  uint64_t br = 0;
  for (int i = 0; i < 4; ++i) { u128 d = (u128)a[i] - b[i] - br; r[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
  return br;
}
static int is_zero4(const uint64_t a[4]) { return (a[0] | a[1] | a[2] | a[3]) == 0; }
static void f_add(const field* F, uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
  uint64_t t[4], c = add4(t, a, b);
  if (c || cmp4(t, F->m) >= 0) sub4(t, t, F->m);
  memcpy(r, t, 32);
}
static void f_sub(const field* F, uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
  uint64_t t[4];
  if (sub4(t, a, b)) add4(t, t, F->m);
  memcpy(r, t, 32);
}
static void f_mul(const field* F, uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {   /* CIOS, moduli up to 2^256 */
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) {
    u128 c = 0;
    for (int j = 0; j < 4; ++j) { c += (u128)a[j] * b[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
    uint64_t m = t[0] * F->ninv;
    c = ((u128)m * F->m[0] + t[0]) >> 64;
    for (int j = 1; j < 4; ++j) { c += (u128)m * F->m[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
  }
  if (t[4] || cmp4(t, F->m) >= 0) sub4(t, t, F->m);
  memcpy(r, t, 32);
}
static void f_pow(const field* F, uint64_t r[4], const uint64_t a[4], const uint64_t e[4]) {
  uint64_t acc[4], base[4];
  memcpy(acc, F->one, 32); memcpy(base, a, 32);
  for (int i = 255; i >= 0; --i) {
    f_mul(F, acc, acc, acc);
    if ((e[i >> 6] >> (i & 63)) & 1) f_mul(F, acc, acc, base);
  }
  memcpy(r, acc, 32);
}
static void field_init(field* F, const uint64_t m[4]) {
  memcpy(F->m, m, 32);
  uint64_t x = 1;
  for (int i = 0; i < 6; ++i) x *= 2 - m[0] * x;      /* m^-1 mod 2^64 */
  F->ninv = (uint64_t)0 - x;
  /* 2^256 mod m and 2^512 mod m by doubling */
  uint64_t t[4] = {1, 0, 0, 0};
  for (int i = 0; i < 512; ++i) {
    uint64_t c = add4(t, t, t);
    if (c || cmp4(t, m) >= 0) sub4(t, t, m);
    if (i == 255) memcpy(F->one, t, 32);
  }
  memcpy(F->r2, t, 32);
}
static void to_mont(const field* F, uint64_t r[4], const uint64_t a[4]) { f_mul(F, r, a, F->r2); }
static void from_mont(const field* F, uint64_t r[4], const uint64_t a[4]) { uint64_t o[4] = {1, 0, 0, 0}; f_mul(F, r, a, o); }
/* any 256-bit integer mod m */
static void reduce256(const field* F, uint64_t r[4], const uint64_t a[4]) { uint64_t t[4]; to_mont(F, t, a); from_mont(F, r, t); }

static void do_init(void) {
  field_init(&FP, P_M); field_init(&FN, N_M);
  to_mont(&FP, B_M.v, B_I); to_mont(&FP, GX_M.v, GX_I); to_mont(&FP, GY_M.v, GY_I);
  uint64_t three[4] = {3, 0, 0, 0};
  to_mont(&FP, THREE_M.v, three);
}
static void ensure_init(void) { pthread_once(&g_once, do_init); }

#define qmul(r, a, b) f_mul(&FP, (r)->v, (a)->v, (b)->v)
#define qadd(r, a, b) f_add(&FP, (r)->v, (a)->v, (b)->v)
#define qsub(r, a, b) f_sub(&FP, (r)->v, (a)->v, (b)->v)
static void qinv(fe* r, const fe* a) { uint64_t e[4]; uint64_t two[4] = {2, 0, 0, 0}; sub4(e, P_M, two); f_pow(&FP, r->v, a->v, e); }
static int qsqrt(fe* r, const fe* a) {           /* p = 3 (mod 4) */
  uint64_t e[4], one[4] = {1, 0, 0, 0};
  add4(e, P_M, one);                              /* p + 1 < 2^256 */
  for (int i = 0; i < 4; ++i) e[i] = (e[i] >> 2) | (i < 3 ? e[i + 1] << 62 : 0);
  fe t, c;
  f_pow(&FP, t.v, a->v, e);
  qmul(&c, &t, &t);
  *r = t;
  return cmp4(c.v, a->v) == 0;
}

/* Jacobian (X, Y, Z), x = X / Z^2, y = Y / Z^3; Z = 0 is the point at infinity */
typedef struct { fe X, Y, Z; } jac;
static void jac_inf(jac* p) { memcpy(p->X.v, FP.one, 32); memcpy(p->Y.v, FP.one, 32); memset(p->Z.v, 0, 32); }
static int jac_is_inf(const jac* p) { return is_zero4(p->Z.v); }
static void jac_dbl(jac* r, const jac* p) {      /* dbl-2001-b, a = -3 */
  if (jac_is_inf(p)) { *r = *p; return; }
  fe delta, gamma, beta, alpha, t0, t1, t2;
  qmul(&delta, &p->Z, &p->Z); qmul(&gamma, &p->Y, &p->Y); qmul(&beta, &p->X, &gamma);
  qsub(&t0, &p->X, &delta); qadd(&t1, &p->X, &delta); qmul(&t2, &t0, &t1); qmul(&alpha, &t2, &THREE_M);
  fe X3, Y3, Z3, b4, b8;
  qadd(&b4, &beta, &beta); qadd(&b4, &b4, &b4); qadd(&b8, &b4, &b4);
  qmul(&X3, &alpha, &alpha); qsub(&X3, &X3, &b8);
  qadd(&t0, &p->Y, &p->Z); qmul(&Z3, &t0, &t0); qsub(&Z3, &Z3, &gamma); qsub(&Z3, &Z3, &delta);
  qsub(&t0, &b4, &X3); qmul(&Y3, &alpha, &t0);
  qmul(&t1, &gamma, &gamma); qadd(&t1, &t1, &t1); qadd(&t1, &t1, &t1); qadd(&t1, &t1, &t1);
  qsub(&Y3, &Y3, &t1);
  r->X = X3; r->Y = Y3; r->Z = Z3;
}
static void jac_add(jac* r, const jac* p, const jac* q) {   /* add-2007-bl with the special cases spelled out */
  if (jac_is_inf(p)) { *r = *q; return; }
  if (jac_is_inf(q)) { *r = *p; return; }
  fe z1z1, z2z2, u1, u2, s1, s2, h, i, j, rr, v, t;
  qmul(&z1z1, &p->Z, &p->Z); qmul(&z2z2, &q->Z, &q->Z);
  qmul(&u1, &p->X, &z2z2); qmul(&u2, &q->X, &z1z1);
  qmul(&t, &q->Z, &z2z2); qmul(&s1, &p->Y, &t);
  qmul(&t, &p->Z, &z1z1); qmul(&s2, &q->Y, &t);
  qsub(&h, &u2, &u1); qsub(&rr, &s2, &s1);
  if (is_zero4(h.v)) {
    if (is_zero4(rr.v)) { jac_dbl(r, p); return; }
    jac_inf(r); return;
  }
  qadd(&i, &h, &h); qmul(&i, &i, &i); qmul(&j, &h, &i); qadd(&rr, &rr, &rr); qmul(&v, &u1, &i);
  fe X3, Y3, Z3;
  qmul(&X3, &rr, &rr); qsub(&X3, &X3, &j); qsub(&X3, &X3, &v); qsub(&X3, &X3, &v);
  qsub(&t, &v, &X3); qmul(&Y3, &rr, &t); qmul(&t, &s1, &j); qadd(&t, &t, &t); qsub(&Y3, &Y3, &t);
  qadd(&Z3, &p->Z, &q->Z); qmul(&Z3, &Z3, &Z3); qsub(&Z3, &Z3, &z1z1); qsub(&Z3, &Z3, &z2z2); qmul(&Z3, &Z3, &h);
  r->X = X3; r->Y = Y3; r->Z = Z3;
}
static void jac_neg(jac* r, const jac* p) { fe z; memset(&z, 0, sizeof z); *r = *p; qsub(&r->Y, &z, &p->Y); }
static void jac_mul(jac* r, const jac* p, const uint64_t k[4]) {
  jac acc; jac_inf(&acc);
  for (int i = 255; i >= 0; --i) {
    jac_dbl(&acc, &acc);
    if ((k[i >> 6] >> (i & 63)) & 1) jac_add(&acc, &acc, p);
  }
  *r = acc;
}
static void jac_from_affine(jac* p, const fe* x, const fe* y) { p->X = *x; p->Y = *y; memcpy(p->Z.v, FP.one, 32); }

/* ---- bytes ---- */
static void load_be(uint64_t w[4], const uint8_t* b) {
  for (int i = 0; i < 4; ++i) { uint64_t v = 0; for (int k = 0; k < 8; ++k) v = (v << 8) | b[8 * (3 - i) + k]; w[i] = v; }
}
static void store_be(uint8_t* b, const uint64_t w[4]) {
  for (int i = 0; i < 4; ++i) for (int k = 0; k < 8; ++k) b[8 * (3 - i) + k] = (uint8_t)(w[i] >> (56 - 8 * k));
}
/* Sec1Codec::point_encode: 33 bytes, or the single byte 0x00 for the point at infinity; returns the length */
static size_t sec1_encode(uint8_t out[33], const jac* p) {
  if (jac_is_inf(p)) { out[0] = 0; return 1; }
  fe zi, zi2, zi3, x, y;
  qinv(&zi, &p->Z); qmul(&zi2, &zi, &zi); qmul(&zi3, &zi2, &zi); qmul(&x, &p->X, &zi2); qmul(&y, &p->Y, &zi3);
  uint64_t xi[4], yi[4];
  from_mont(&FP, xi, x.v); from_mont(&FP, yi, y.v);
  out[0] = (uint8_t)(2 + (yi[0] & 1));
  store_be(out + 1, xi);
  return 33;
}
static int sec1_decode(jac* p, const uint8_t in[33]) {
  if (in[0] != 2 && in[0] != 3) return 0;
  uint64_t xi[4];
  load_be(xi, in + 1);
  if (cmp4(xi, P_M) >= 0) return 0;
  fe x, y, t, rhs;
  to_mont(&FP, x.v, xi);
  qmul(&t, &x, &x); qmul(&rhs, &t, &x); qmul(&t, &x, &THREE_M); qsub(&rhs, &rhs, &t); qadd(&rhs, &rhs, &B_M);
  if (!qsqrt(&y, &rhs)) return 0;
  uint64_t yi[4];
  from_mont(&FP, yi, y.v);
  if ((yi[0] & 1) != (uint64_t)(in[0] & 1)) { fe z; memset(&z, 0, sizeof z); qsub(&y, &z, &y); }
  jac_from_affine(p, &x, &y);
  return 1;
}

/* ---- SHA-256 / HMAC ---- */
static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
typedef struct { uint32_t h[8]; uint8_t buf[64]; size_t len, total; } sha256_ctx;
#define ROR32(x, n) (((x) >> (n)) | ((x) << (32 - (n))))
static void sha256_block(sha256_ctx* c, const uint8_t* p) {
  uint32_t w[64], a[8];
  for (int i = 0; i < 16; ++i) w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
  for (int i = 16; i < 64; ++i) {
    uint32_t s0 = ROR32(w[i - 15], 7) ^ ROR32(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = ROR32(w[i - 2], 17) ^ ROR32(w[i - 2], 19) ^ (w[i - 2] >> 10);
    w[i] = w[i - 16] + s0 + w[i - 7] + s1;
  }
  memcpy(a, c->h, 32);
  for (int i = 0; i < 64; ++i) {
    uint32_t S1 = ROR32(a[4], 6) ^ ROR32(a[4], 11) ^ ROR32(a[4], 25), ch = (a[4] & a[5]) ^ (~a[4] & a[6]);
    uint32_t t1 = a[7] + S1 + ch + K256[i] + w[i];
    uint32_t S0 = ROR32(a[0], 2) ^ ROR32(a[0], 13) ^ ROR32(a[0], 22), mj = (a[0] & a[1]) ^ (a[0] & a[2]) ^ (a[1] & a[2]);
    uint32_t t2 = S0 + mj;
    a[7] = a[6]; a[6] = a[5]; a[5] = a[4]; a[4] = a[3] + t1; a[3] = a[2]; a[2] = a[1]; a[1] = a[0]; a[0] = t1 + t2;
  }
  for (int i = 0; i < 8; ++i) c->h[i] += a[i];
}
static void sha256_init(sha256_ctx* c) {
  static const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  memcpy(c->h, iv, 32); c->len = 0; c->total = 0;
}
static void sha256_update(sha256_ctx* c, const void* data, size_t n) {
  const uint8_t* p = (const uint8_t*)data;
  c->total += n;
  while (n) {
    size_t k = 64 - c->len; if (k > n) k = n;
    memcpy(c->buf + c->len, p, k); c->len += k; p += k; n -= k;
    if (c->len == 64) { sha256_block(c, c->buf); c->len = 0; }
  }
}
static void sha256_final(sha256_ctx* c, uint8_t out[32]) {
  uint64_t bits = (uint64_t)c->total * 8;
  uint8_t pad = 0x80; sha256_update(c, &pad, 1);
  uint8_t z = 0; while (c->len != 56) sha256_update(c, &z, 1);
  uint8_t lb[8]; for (int i = 0; i < 8; ++i) lb[i] = (uint8_t)(bits >> (56 - 8 * i));
  sha256_update(c, lb, 8);
  for (int i = 0; i < 8; ++i) { out[4 * i] = (uint8_t)(c->h[i] >> 24); out[4 * i + 1] = (uint8_t)(c->h[i] >> 16); out[4 * i + 2] = (uint8_t)(c->h[i] >> 8); out[4 * i + 3] = (uint8_t)c->h[i]; }
}
static void hmac256(uint8_t out[32], const uint8_t key[32], const uint8_t* m, size_t n) {
  uint8_t k[64], inner[32];
  sha256_ctx c;
  memset(k, 0x36, 64); for (int i = 0; i < 32; ++i) k[i] ^= key[i];
  sha256_init(&c); sha256_update(&c, k, 64); sha256_update(&c, m, n); sha256_final(&c, inner);
  memset(k, 0x5c, 64); for (int i = 0; i < 32; ++i) k[i] ^= key[i];
  sha256_init(&c); sha256_update(&c, k, 64); sha256_update(&c, inner, 32); sha256_final(&c, out);
}

/* ---- suite functions (oracle/sw_oracle.py) ---- */
static const uint8_t SUITE_ID = 0x01;
static int hash_to_curve_tai(jac* out, uint8_t enc[33], const uint8_t* data, size_t len) {
  for (int ctr = 0; ctr < 256; ++ctr) {
    sha256_ctx c; uint8_t b[2] = {SUITE_ID, 0x01}, t[2] = {(uint8_t)ctr, 0x00};
    sha256_init(&c); sha256_update(&c, b, 2); sha256_update(&c, data, len); sha256_update(&c, t, 2);
    enc[0] = 0x02; sha256_final(&c, enc + 1);
    if (sec1_decode(out, enc)) return 1;
  }
  return 0;
}
static void nonce_rfc6979(uint64_t k[4], const uint64_t sk[4], const uint8_t* h_enc, size_t h_len) {
  uint8_t h1[32], x[32], V[32], K[32], buf[32 + 1 + 32 + 32];
  sha256_ctx c; sha256_init(&c); sha256_update(&c, h_enc, h_len); sha256_final(&c, h1);
  store_be(x, sk);
  memset(V, 1, 32); memset(K, 0, 32);
  for (int sep = 0; sep < 2; ++sep) {
    memcpy(buf, V, 32); buf[32] = (uint8_t)sep; memcpy(buf + 33, x, 32); memcpy(buf + 65, h1, 32);
    hmac256(K, K, buf, 97); hmac256(V, K, V, 32);
  }
  hmac256(V, K, V, 32);
  uint64_t t[4]; load_be(t, V); reduce256(&FN, k, t);
}
static void challenge(uint64_t c_out[4], const uint8_t* encs[5], const size_t lens[5], const uint8_t* ad, size_t ad_len) {
  sha256_ctx c; uint8_t b[2] = {SUITE_ID, 0x02}, z = 0, h[32], w[32];
  sha256_init(&c); sha256_update(&c, b, 2);
  for (int i = 0; i < 5; ++i) sha256_update(&c, encs[i], lens[i]);
  sha256_update(&c, ad, ad_len); sha256_update(&c, &z, 1); sha256_final(&c, h);
  memset(w, 0, 32); memcpy(w + 16, h, 16);
  load_be(c_out, w);
}

int p256_secret_from_seed(const uint8_t* seed, size_t len, uint8_t sk_be[32]) {
  ensure_init();
  uint8_t h[32]; sha256_ctx c; sha256_init(&c); sha256_update(&c, seed, len); sha256_final(&c, h);
  uint64_t w[4], r[4];
  for (int i = 0; i < 4; ++i) { uint64_t v = 0; for (int k = 7; k >= 0; --k) v = (v << 8) | h[8 * i + k]; w[i] = v; }   /* little-endian */
  reduce256(&FN, r, w); store_be(sk_be, r);
  return 0;
}
int p256_public_from_secret(const uint8_t sk_be[32], uint8_t pk[33]) {
  ensure_init();
  uint64_t t[4], k[4]; load_be(t, sk_be); reduce256(&FN, k, t);
  jac g, r; jac_from_affine(&g, &GX_M, &GY_M); jac_mul(&r, &g, k);
  return sec1_encode(pk, &r) == 33 ? 0 : 2;
}
int p256_hash_to_curve(const uint8_t* data, size_t len, uint8_t out[33]) {
  ensure_init();
  jac h; return hash_to_curve_tai(&h, out, data, len) ? 0 : 2;
}
int p256_output_hash(const uint8_t gamma[33], uint8_t out[32]) {
  sha256_ctx c; uint8_t b[2] = {SUITE_ID, 0x03}, z = 0;
  sha256_init(&c); sha256_update(&c, b, 2); sha256_update(&c, gamma, 33); sha256_update(&c, &z, 1); sha256_final(&c, out);
  return 0;
}
int p256_point_decode(const uint8_t in[33]) { ensure_init(); jac p; return sec1_decode(&p, in) ? 0 : 2; }

/* outputs: gamma 33, c 32 (big-endian, 16 significant), s 32, pk 33 (nullable), h 33 (nullable).  0 = ok, 2 = invalid data */
int p256_ietf_prove(const uint8_t sk_be[32], const uint8_t* msg, size_t msg_len, const uint8_t* h_given, const uint8_t* ad,
                    size_t ad_len, uint8_t gamma[33], uint8_t c_be[32], uint8_t s_be[32], uint8_t* pk_out, uint8_t* h_out) {
  ensure_init();
  uint64_t t[4], sk[4], k[4];
  load_be(t, sk_be); reduce256(&FN, sk, t);
  if (cmp4(t, N_M) >= 0) return 2;               /* a secret key is a canonical scalar (as the Edwards suites hold it) */
  jac H, G, Y, Gam, U, V;
  uint8_t henc[33], pk[33], uenc[33], venc[33];
  if (h_given) { if (!sec1_decode(&H, h_given)) return 2; memcpy(henc, h_given, 33); }
  else if (!hash_to_curve_tai(&H, henc, msg, msg_len)) return 2;
  jac_from_affine(&G, &GX_M, &GY_M);
  jac_mul(&Y, &G, sk); jac_mul(&Gam, &H, sk);
  nonce_rfc6979(k, sk, henc, 33);
  jac_mul(&U, &G, k); jac_mul(&V, &H, k);
  size_t lens[5] = {sec1_encode(pk, &Y), 33, sec1_encode(gamma, &Gam), sec1_encode(uenc, &U), sec1_encode(venc, &V)};
  const uint8_t* encs[5] = {pk, henc, gamma, uenc, venc};
  uint64_t c[4], cm[4], skm[4], prod[4], km[4], sm[4], s[4];
  challenge(c, encs, lens, ad, ad_len);
  to_mont(&FN, cm, c); to_mont(&FN, skm, sk); f_mul(&FN, prod, cm, skm); to_mont(&FN, km, k); f_add(&FN, sm, prod, km);
  from_mont(&FN, s, sm);
  store_be(c_be, c); store_be(s_be, s);
  if (lens[0] != 33) memset(pk + 1, 0, 32);
  if (lens[2] != 33) memset(gamma + 1, 0, 32);
  if (pk_out) memcpy(pk_out, pk, 33);
  if (h_out) memcpy(h_out, henc, 33);
  return 0;
}
/* 0 = ok, 1 = does not verify, 2 = invalid data */
int p256_ietf_verify(const uint8_t pk[33], const uint8_t h[33], const uint8_t gamma[33], const uint8_t c_be[32],
                     const uint8_t s_be[32], const uint8_t* ad, size_t ad_len) {
  ensure_init();
  jac Y, H, Gam, G, t1, t2, U, V;
  if (!sec1_decode(&Y, pk) || !sec1_decode(&H, h) || !sec1_decode(&Gam, gamma)) return 2;
  uint64_t t[4], c[4], s[4], cc[4];
  load_be(t, c_be); reduce256(&FN, c, t);
  load_be(t, s_be); reduce256(&FN, s, t);
  if (cmp4(t, N_M) >= 0) return 2;               /* RFC 9381 5.4.4: s >= q is INVALID (upstream deserialises s strictly) */
  jac_from_affine(&G, &GX_M, &GY_M);
  jac_mul(&t1, &G, s); jac_mul(&t2, &Y, c); jac_neg(&t2, &t2); jac_add(&U, &t1, &t2);
  jac_mul(&t1, &H, s); jac_mul(&t2, &Gam, c); jac_neg(&t2, &t2); jac_add(&V, &t1, &t2);
  uint8_t uenc[33], venc[33];
  size_t lens[5] = {33, 33, 33, sec1_encode(uenc, &U), sec1_encode(venc, &V)};
  const uint8_t* encs[5] = {pk, h, gamma, uenc, venc};
  challenge(cc, encs, lens, ad, ad_len);
  load_be(t, c_be);                              /* CHALLENGE_LEN = 16: the field as it stands (oracle_vrf.c oracle_ietf_verify) */
  return cmp4(cc, t) == 0 ? 0 : 1;
}

/* ---- Pedersen VRF (oracle/sw_oracle.py pedersen_*; unpinned on this suite) ---- */
static jac BB_;                                   /* the blinding base the tests set (p256_set_blinding_base) */
static int have_bb = 0;
int p256_set_blinding_base(const uint8_t bx_be[32], const uint8_t by_be[32]) {
  ensure_init();
  uint64_t xi[4], yi[4]; fe x, y;
  load_be(xi, bx_be); load_be(yi, by_be);
  to_mont(&FP, x.v, xi); to_mont(&FP, y.v, yi);
  jac_from_affine(&BB_, &x, &y);
  have_bb = 1;
  return 0;
}
static void blinding(uint64_t b[4], const uint64_t sk[4], const uint8_t henc[33], const uint8_t* ad, size_t ad_len) {
  sha256_ctx c; uint8_t pre[2] = {SUITE_ID, 0xCC}, z = 0, skb[32], h[32];
  store_be(skb, sk);
  sha256_init(&c); sha256_update(&c, pre, 2); sha256_update(&c, skb, 32); sha256_update(&c, henc, 33);
  sha256_update(&c, ad, ad_len); sha256_update(&c, &z, 1); sha256_final(&c, h);
  uint64_t t[4]; load_be(t, h); reduce256(&FN, b, t);
}
static void n_muladd(uint64_t out[4], const uint64_t a[4], const uint64_t b[4], const uint64_t c[4]) {   /* a b + c mod n */
  uint64_t am[4], bm[4], cm[4], p[4], r[4];
  to_mont(&FN, am, a); to_mont(&FN, bm, b); to_mont(&FN, cm, c);
  f_mul(&FN, p, am, bm); f_add(&FN, r, p, cm); from_mont(&FN, out, r);
}
/* outputs: gamma, pk_com, r, ok 33 each; s, sb 32; blinding 32 (nullable); h_out 33 (nullable) */
int p256_pedersen_prove(const uint8_t sk_be[32], const uint8_t* msg, size_t msg_len, const uint8_t* h_given, const uint8_t* ad,
                        size_t ad_len, uint8_t gamma[33], uint8_t pk_com[33], uint8_t r_out[33], uint8_t ok_out[33], uint8_t s_be[32],
                        uint8_t sb_be[32], uint8_t* blinding_out, uint8_t* h_out) {
  ensure_init();
  if (!have_bb) return 2;
  uint64_t t[4], sk[4], k[4], kb[4], b[4];
  load_be(t, sk_be); reduce256(&FN, sk, t);
  if (cmp4(t, N_M) >= 0) return 2;
  jac H, G, Gam, t1, t2, PC, R, OK;
  uint8_t henc[33];
  if (h_given) { if (!sec1_decode(&H, h_given)) return 2; memcpy(henc, h_given, 33); }
  else if (!hash_to_curve_tai(&H, henc, msg, msg_len)) return 2;
  jac_from_affine(&G, &GX_M, &GY_M);
  blinding(b, sk, henc, ad, ad_len);
  nonce_rfc6979(k, sk, henc, 33); nonce_rfc6979(kb, b, henc, 33);
  jac_mul(&Gam, &H, sk);
  jac_mul(&t1, &G, sk); jac_mul(&t2, &BB_, b); jac_add(&PC, &t1, &t2);
  jac_mul(&t1, &G, k); jac_mul(&t2, &BB_, kb); jac_add(&R, &t1, &t2);
  jac_mul(&OK, &H, k);
  size_t lens[5] = {sec1_encode(pk_com, &PC), 33, sec1_encode(gamma, &Gam), sec1_encode(r_out, &R), sec1_encode(ok_out, &OK)};
  const uint8_t* encs[5] = {pk_com, henc, gamma, r_out, ok_out};
  uint64_t c[4], s[4], sb[4];
  challenge(c, encs, lens, ad, ad_len);
  n_muladd(s, c, sk, k); n_muladd(sb, c, b, kb);
  store_be(s_be, s); store_be(sb_be, sb);
  if (blinding_out) store_be(blinding_out, b);
  if (h_out) memcpy(h_out, henc, 33);
  return 0;
}
int p256_pedersen_verify(const uint8_t h[33], const uint8_t gamma[33], const uint8_t pk_com[33], const uint8_t r[33],
                         const uint8_t ok[33], const uint8_t s_be[32], const uint8_t sb_be[32], const uint8_t* ad, size_t ad_len) {
  ensure_init();
  if (!have_bb) return 2;
  jac H, Gam, PC, R, OK, G, t1, t2, lhs, rhs;
  if (!sec1_decode(&H, h) || !sec1_decode(&Gam, gamma) || !sec1_decode(&PC, pk_com) || !sec1_decode(&R, r) || !sec1_decode(&OK, ok)) return 2;
  uint64_t t[4], s[4], sb[4], c[4];
  load_be(t, s_be); reduce256(&FN, s, t); if (cmp4(t, N_M) >= 0) return 2;
  load_be(t, sb_be); reduce256(&FN, sb, t); if (cmp4(t, N_M) >= 0) return 2;
  size_t lens[5] = {33, 33, 33, 33, 33};
  const uint8_t* encs[5] = {pk_com, h, gamma, r, ok};
  challenge(c, encs, lens, ad, ad_len);
  uint8_t e1[33], e2[33];
  jac_from_affine(&G, &GX_M, &GY_M);
  jac_mul(&t1, &Gam, c); jac_add(&lhs, &t1, &OK); jac_mul(&rhs, &H, s);           /* c Gamma + Ok == s H */
  size_t l1 = sec1_encode(e1, &lhs), l2 = sec1_encode(e2, &rhs);
  if (l1 != l2 || memcmp(e1, e2, l1)) return 1;
  jac_mul(&t1, &PC, c); jac_add(&lhs, &t1, &R);                                   /* c pk_com + R == s G + sb B */
  jac_mul(&t1, &G, s); jac_mul(&t2, &BB_, sb); jac_add(&rhs, &t1, &t2);
  l1 = sec1_encode(e1, &lhs); l2 = sec1_encode(e2, &rhs);
  return (l1 == l2 && !memcmp(e1, e2, l1)) ? 0 : 1;
}

/* `VariableBaseMSM::msm` restated the plain way: sum_i k_i P_i by one double-and-add per term (arkworks' msm is a bucket method;
 * the sum is the same group element).  bases: n x 64 B affine x || y, 32-byte LITTLE-endian integers (all-zero = the point at
 * infinity), scalars: n x 32 B big-endian (< n).  out33: Sec1 (0x00 + 32 zero bytes for the point at infinity), out_xy: x || y
 * little-endian (all-zero for it).  Returns 0, or 2 on a coordinate >= p, a point off the curve or a scalar >= n (outputs
 * zeroed) -- the statuses of vrfhip_msm on the secp256r1 suite.  [ref /root/reference src/lib.rs:14 `reexports`; BASELINE.json] */
int p256_msm(size_t n, const uint8_t* bases_xy, const uint8_t* scalars_be, uint8_t out33[33], uint8_t out_xy[64]) {
  ensure_init();
  jac acc;
  jac_inf(&acc);
  int bad = 0;
  for (size_t i = 0; i < n; ++i) {
    const uint8_t* b = bases_xy + 64 * i;
    uint64_t xw[4], yw[4], k[4];
    for (int j = 0; j < 4; ++j) {
      xw[j] = 0; yw[j] = 0;
      for (int t = 7; t >= 0; --t) { xw[j] = (xw[j] << 8) | b[8 * j + t]; yw[j] = (yw[j] << 8) | b[32 + 8 * j + t]; }
    }
    load_be(k, scalars_be + 32 * i);
    if (cmp4(k, N_M) >= 0 || cmp4(xw, P_M) >= 0 || cmp4(yw, P_M) >= 0) { bad = 1; continue; }
    if (is_zero4(xw) && is_zero4(yw)) continue;                  /* infinity: adds nothing */
    fe x, y, l, r, t;
    to_mont(&FP, x.v, xw); to_mont(&FP, y.v, yw);
    qmul(&l, &y, &y);
    qmul(&t, &x, &x); qmul(&r, &t, &x);
    qsub(&r, &r, &x); qsub(&r, &r, &x); qsub(&r, &r, &x);
    qadd(&r, &r, &B_M);
    if (memcmp(l.v, r.v, 32) != 0) { bad = 1; continue; }
    jac p, kp;
    jac_from_affine(&p, &x, &y);
    jac_mul(&kp, &p, k);
    jac_add(&acc, &acc, &kp);
  }
  memset(out33, 0, 33);
  if (out_xy) memset(out_xy, 0, 64);
  if (bad) return 2;
  if (jac_is_inf(&acc)) return 0;
  sec1_encode(out33, &acc);
  if (out_xy) {
    /* affine x, y from the Sec1 string and the parity of y */
    jac q;
    sec1_decode(&q, out33);
    uint64_t xa[4], ya[4];
    from_mont(&FP, xa, q.X.v); from_mont(&FP, ya, q.Y.v);
    for (int j = 0; j < 4; ++j)
      for (int t = 0; t < 8; ++t) { out_xy[8 * j + t] = (uint8_t)(xa[j] >> (8 * t)); out_xy[32 + 8 * j + t] = (uint8_t)(ya[j] >> (8 * t)); }
  }
  return 0;
}

/* ---- batch drivers ---- */
typedef struct {
  int kind; size_t lo, hi;
  const uint8_t *a0, *a1, *a2, *a3, *a4, *a5, *a6, *ad; size_t ad_len, msg_len;
  uint8_t *o0, *o1, *o2, *o3, *o4, *o5, *o6, *o7, *st;
} pjob;
static void* p_run(void* arg) {
  pjob* j = (pjob*)arg;
  for (size_t i = j->lo; i < j->hi; ++i) {
    if (j->kind == 0)
      j->st[i] = (uint8_t)p256_ietf_verify(j->a0 + 33 * i, j->a1 + 33 * i, j->a2 + 33 * i, j->a3 + 32 * i, j->a4 + 32 * i, j->ad, j->ad_len);
    else if (j->kind == 2)          /* a0..a4: h, gamma, pk_com, r, ok; a5, a6: s, sb */
      j->st[i] = (uint8_t)p256_pedersen_verify(j->a0 + 33 * i, j->a1 + 33 * i, j->a2 + 33 * i, j->a3 + 33 * i, j->a4 + 33 * i,
                                               j->a5 + 32 * i, j->a6 + 32 * i, j->ad, j->ad_len);
    else if (j->kind == 3) {        /* o0..o3: gamma, pk_com, r, ok; o4, o5: s, sb; o6: blinding; o7: h */
      int rc = p256_pedersen_prove(j->a0 + 32 * i, j->a1 ? j->a1 + j->msg_len * i : NULL, j->msg_len, j->a2 ? j->a2 + 33 * i : NULL,
                                   j->ad, j->ad_len, j->o0 + 33 * i, j->o1 + 33 * i, j->o2 + 33 * i, j->o3 + 33 * i, j->o4 + 32 * i,
                                   j->o5 + 32 * i, j->o6 ? j->o6 + 32 * i : NULL, j->o7 ? j->o7 + 33 * i : NULL);
      if (j->st) j->st[i] = (uint8_t)rc;
    } else {
      int rc = p256_ietf_prove(j->a0 + 32 * i, j->a1 ? j->a1 + j->msg_len * i : NULL, j->msg_len, j->a2 ? j->a2 + 33 * i : NULL, j->ad,
                               j->ad_len, j->o0 + 33 * i, j->o1 + 32 * i, j->o2 + 32 * i, j->o3 ? j->o3 + 33 * i : NULL,
                               j->o4 ? j->o4 + 33 * i : NULL);
      if (j->st) j->st[i] = (uint8_t)rc;
    }
  }
  return NULL;
}
static void p_batch(pjob base, size_t n, int threads) {
  ensure_init();
  if (threads <= 1 || n < 2) { base.lo = 0; base.hi = n; p_run(&base); return; }
  if ((size_t)threads > n) threads = (int)n;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * threads);
  pjob* jobs = (pjob*)malloc(sizeof(pjob) * threads);
  for (int t = 0; t < threads; ++t) {
    jobs[t] = base; jobs[t].lo = n * t / threads; jobs[t].hi = n * (t + 1) / threads;
    pthread_create(&th[t], NULL, p_run, &jobs[t]);
  }
  for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
  free(th); free(jobs);
}
void p256_ietf_verify_batch(size_t n, const uint8_t* pk, const uint8_t* h, const uint8_t* gamma, const uint8_t* c, const uint8_t* s,
                            const uint8_t* ad, size_t ad_len, uint8_t* status, int threads) {
  pjob j; memset(&j, 0, sizeof j);
  j.kind = 0; j.a0 = pk; j.a1 = h; j.a2 = gamma; j.a3 = c; j.a4 = s; j.ad = ad; j.ad_len = ad_len; j.st = status;
  p_batch(j, n, threads);
}
void p256_ietf_prove_batch(size_t n, const uint8_t* sk, const uint8_t* msg, size_t msg_len, const uint8_t* h_given, const uint8_t* ad,
                           size_t ad_len, uint8_t* gamma, uint8_t* c, uint8_t* s, uint8_t* pk_out, uint8_t* h_out, uint8_t* status,
                           int threads) {
  pjob j; memset(&j, 0, sizeof j);
  j.kind = 1; j.a0 = sk; j.a1 = msg; j.msg_len = msg_len; j.a2 = h_given; j.ad = ad; j.ad_len = ad_len;
  j.o0 = gamma; j.o1 = c; j.o2 = s; j.o3 = pk_out; j.o4 = h_out; j.st = status;
  p_batch(j, n, threads);
}
void p256_pedersen_verify_batch(size_t n, const uint8_t* h, const uint8_t* gamma, const uint8_t* pk_com, const uint8_t* r,
                                const uint8_t* ok, const uint8_t* s, const uint8_t* sb, const uint8_t* ad, size_t ad_len,
                                uint8_t* status, int threads) {
  pjob j; memset(&j, 0, sizeof j);
  j.kind = 2; j.a0 = h; j.a1 = gamma; j.a2 = pk_com; j.a3 = r; j.a4 = ok; j.a5 = s; j.a6 = sb; j.ad = ad; j.ad_len = ad_len; j.st = status;
  p_batch(j, n, threads);
}
void p256_pedersen_prove_batch(size_t n, const uint8_t* sk, const uint8_t* msg, size_t msg_len, const uint8_t* h_given,
                               const uint8_t* ad, size_t ad_len, uint8_t* gamma, uint8_t* pk_com, uint8_t* r, uint8_t* ok, uint8_t* s,
                               uint8_t* sb, uint8_t* blinding_out, uint8_t* h_out, uint8_t* status, int threads) {
  pjob j; memset(&j, 0, sizeof j);
  j.kind = 3; j.a0 = sk; j.a1 = msg; j.msg_len = msg_len; j.a2 = h_given; j.ad = ad; j.ad_len = ad_len;
  j.o0 = gamma; j.o1 = pk_com; j.o2 = r; j.o3 = ok; j.o4 = s; j.o5 = sb; j.o6 = blinding_out; j.o7 = h_out; j.st = status;
  p_batch(j, n, threads);
}
