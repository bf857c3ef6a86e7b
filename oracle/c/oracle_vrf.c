/* oracle_vrf.c -- plain-C CPU restatement of the EC-VRF hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * It is the checker and the reported CPU baseline ("port"), never the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle_vrf.so.
 *
 * Parity status: the reference checkout is a re-export shim (/root/reference src/lib.rs:13-17)
 * with no arithmetic and no vectors, so by its own contents parity is UNPINNED.  This file
 * restates the published algorithms ark-vrf 0.1.x implements (RFC 9381 sec. 5, RFC 9380
 * sec. 5.3.1 / 6.7.1 / 6.8.2, RFC 8032-style nonce, arkworks compressed twisted-Edwards
 * serialisation) as written down in SURVEY.md Appendix A, and is pinned field by field on the
 * upstream `bandersnatch_sha-512_ell2` vectors of SURVEY.md Appendix B
 * (tests/golden/bandersnatch_sha512_ell2_kat.json) and on the Python oracle
 * (oracle/vrf_oracle.py).
 *
 * Shape follows arkworks' CPU path, not the GPU design: 4 x 64-bit saturated limbs with
 * unsigned __int128 CIOS Montgomery (R = 2^256) like ark_ff::MontBackend, MSB-first
 * double-and-add like `mul_bigint`, four independent scalar multiplications per verify,
 * loop-form Tonelli-Shanks.  [ref src/lib.rs:LINE name] tags name the interface restated.
 */
#include <pthread.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "params.h"

typedef unsigned __int128 u128;
typedef struct { uint64_t v[4]; } fp;          /* Montgomery form */
typedef struct { uint64_t m[4]; uint64_t ninv; uint64_t r2[4]; uint64_t one[4]; } field;
/* A base field with what Tonelli-Shanks needs: q - 1 = 2^s t, c = z^t for a non-residue z, (t + 1) / 2. */
typedef struct { field f; fp ts_c; uint64_t ts_t[4], ts_e[4]; int s; } basefield;

/* Suite descriptor [ref src/lib.rs:16 `Suite`]: Bandersnatch_SHA-512_ELL2 (pinned by the KATs) and
 * JubJub_SHA-512_TAI (recollection only: parity unpinned). */
typedef struct {
  basefield* bf;                 /* base field: BLS12-381 Fr (Bandersnatch, JubJub), 2^255 - 19 (Ed25519), BN254 Fr (Baby-JubJub) */
  field fr;                      /* scalar field */
  fp d, a, gx, gy, bx, by;       /* curve a x^2 + y^2 = 1 + d x^2 y^2, generator, blinding base */
  const char* suite_id; size_t suite_id_len;
  const char* h2c_dst; size_t h2c_dst_len;     /* RFC 9380 DST (Elligator suites) */
  int h2c_tai, cofactor_log2;
  int challenge_len;             /* `Suite::CHALLENGE_LEN`: leading bytes of the challenge hash */
  int flags;                     /* vrfhip_suite_desc.flags: 1 sign = x mod 2 (RFC 8032), 2 challenge little-endian,
                                    4 output hash of cofactor * Gamma (RFC 9381); 0 for upstream's suites */
} suite_t;
static basefield BF_BLS, BF_25519, BF_BN254;
static suite_t SUITE_BS, SUITE_JJ, SUITE_ED, SUITE_BJ, SUITE_CUSTOM;
static char CUSTOM_ID[64], CUSTOM_DST[128];
static suite_t* S_ = &SUITE_BS;   /* current suite (tests select it with oracle_set_suite) */
#define FR (S_->fr)
#define BS_D_M (S_->d)
#define BS_A_M (S_->a)
#define BS_GX_M (S_->gx)
#define BS_GY_M (S_->gy)
#define BS_BX_M (S_->bx)
#define BS_BY_M (S_->by)
#define FQ (S_->bf->f)
#define TS_C_M (S_->bf->ts_c)     /* z^t, generator of the 2^s-torsion (Tonelli-Shanks) */
#define TS_T (S_->bf->ts_t)       /* t = (q-1)/2^s */
#define TS_E (S_->bf->ts_e)       /* (t+1)/2 */
static fp ELL_J_M, ELL_K_M, ELL_Z_M, ELL_JK_M, ELL_K2I_M;      /* Bandersnatch only (BLS12-381 Fr images) */
static int g_init_done = 0;
static pthread_once_t g_once = PTHREAD_ONCE_INIT;

/* ------------------------------------------------------------------ big-integer helpers */
static int cmp4(const uint64_t a[4], const uint64_t b[4]) {
  for (int i = 3; i >= 0; --i) { if (a[i] < b[i]) return -1; if (a[i] > b[i]) return 1; }
  return 0;
}
static uint64_t add4(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
  u128 c = 0;
  for (int i = 0; i < 4; ++i) { c += (u128)a[i] + b[i]; r[i] = (uint64_t)c; c >>= 64; }
  return (uint64_t)c;
}
static uint64_t sub4(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
  uint64_t borrow = 0;
  for (int i = 0; i < 4; ++i) {
    u128 t = (u128)a[i] - b[i] - borrow;
    r[i] = (uint64_t)t; borrow = (uint64_t)(t >> 64) & 1;
  }
  return borrow;
}
static int is_zero4(const uint64_t a[4]) { return (a[0] | a[1] | a[2] | a[3]) == 0; }

/* ------------------------------------------------------------------ Montgomery field
 * [ref src/lib.rs:15 `BaseField` / `ScalarField`] ark_ff::Fp<MontBackend<_, 4>, 4> */
static void f_add(const field* F, uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
  uint64_t t[4], d[4];
  uint64_t c = add4(t, a, b);
  uint64_t bo = sub4(d, t, F->m);
  if (c || !bo) memcpy(r, d, 32); else memcpy(r, t, 32);
}
static void f_sub(const field* F, uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
  uint64_t t[4];
  if (sub4(t, a, b)) add4(t, t, F->m);
  memcpy(r, t, 32);
}
static void f_mul(const field* F, uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) {
    u128 c = 0;
    for (int j = 0; j < 4; ++j) { c += (u128)a[j] * b[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
    uint64_t m = t[0] * F->ninv;
    c = (u128)m * F->m[0] + t[0]; c >>= 64;
    for (int j = 1; j < 4; ++j) { c += (u128)m * F->m[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
  }
  uint64_t d[4];
  uint64_t bo = sub4(d, t, F->m);
  if (t[4] || !bo) memcpy(r, d, 32); else memcpy(r, t, 32);
}
static void f_to_mont(const field* F, uint64_t r[4], const uint64_t a[4]) { f_mul(F, r, a, F->r2); }
static void f_from_mont(const field* F, uint64_t r[4], const uint64_t a[4]) {
  uint64_t one[4] = {1, 0, 0, 0};
  f_mul(F, r, a, one);
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
static void f_inv(const field* F, uint64_t r[4], const uint64_t a[4]) {
  uint64_t e[4], two[4] = {2, 0, 0, 0};
  sub4(e, F->m, two);
  f_pow(F, r, a, e);
}
static void field_init(field* F, const uint64_t m[4]) {
  memcpy(F->m, m, 32);
  uint64_t x = 1;                               /* -m^-1 mod 2^64 by Newton iteration */
  for (int i = 0; i < 6; ++i) x *= 2 - m[0] * x;
  F->ninv = (uint64_t)0 - x;
  uint64_t t[4] = {1, 0, 0, 0};                 /* 2^k mod m by modular doubling */
  for (int i = 0; i < 512; ++i) {
    uint64_t c = add4(t, t, t), d[4];
    uint64_t bo = sub4(d, t, m);
    if (c || !bo) memcpy(t, d, 32);
    if (i == 255) memcpy(F->one, t, 32);
  }
  memcpy(F->r2, t, 32);
}

/* Fq wrappers */
static void q_add(fp* r, const fp* a, const fp* b) { f_add(&FQ, r->v, a->v, b->v); }
static void q_sub(fp* r, const fp* a, const fp* b) { f_sub(&FQ, r->v, a->v, b->v); }
static void q_mul(fp* r, const fp* a, const fp* b) { f_mul(&FQ, r->v, a->v, b->v); }
static void q_sqr(fp* r, const fp* a) { f_mul(&FQ, r->v, a->v, a->v); }
static void q_inv(fp* r, const fp* a) { f_inv(&FQ, r->v, a->v); }
static void q_neg(fp* r, const fp* a) { fp z; memset(&z, 0, sizeof z); f_sub(&FQ, r->v, z.v, a->v); }
static void q_one(fp* r) { memcpy(r->v, FQ.one, 32); }
static void q_zero(fp* r) { memset(r, 0, sizeof *r); }
static int q_is_zero(const fp* a) { return is_zero4(a->v); }
static int q_eq(const fp* a, const fp* b) { return cmp4(a->v, b->v) == 0; }
static void q_from_int(fp* r, const uint64_t a[4]) { f_to_mont(&FQ, r->v, a); }
static void q_to_int(uint64_t r[4], const fp* a) { f_from_mont(&FQ, r, a->v); }
static void q_from_u64(fp* r, uint64_t x) { uint64_t a[4] = {x, 0, 0, 0}; q_from_int(r, a); }

/* Legendre symbol: 1 square (non-zero), 0 zero, -1 non-square */
static int q_legendre(const fp* a) {
  if (q_is_zero(a)) return 0;
  uint64_t e[4], one[4] = {1, 0, 0, 0};
  sub4(e, FQ.m, one);
  for (int i = 0; i < 4; ++i) e[i] = (e[i] >> 1) | (i < 3 ? e[i + 1] << 63 : 0);
  fp t; f_pow(&FQ, t.v, a->v, e);
  return cmp4(t.v, FQ.one) == 0 ? 1 : -1;
}
/* Tonelli-Shanks, loop form (q - 1 = 2^s * t).  Returns 0 if a is not a square. */
static int q_sqrt(fp* r, const fp* a) {
  if (q_is_zero(a)) { q_zero(r); return 1; }
  if (q_legendre(a) != 1) return 0;
  fp c = TS_C_M, tt, rr;
  f_pow(&FQ, tt.v, a->v, TS_T);
  f_pow(&FQ, rr.v, a->v, TS_E);
  int m = S_->bf->s;
  while (cmp4(tt.v, FQ.one) != 0) {
    int i = 0; fp t2 = tt;
    while (cmp4(t2.v, FQ.one) != 0) { q_sqr(&t2, &t2); ++i; }
    fp b = c;
    for (int j = 0; j < m - i - 1; ++j) q_sqr(&b, &b);
    m = i; q_sqr(&c, &b); q_mul(&tt, &tt, &c); q_mul(&rr, &rr, &b);
  }
  *r = rr;
  return 1;
}

/* ------------------------------------------------------------------ twisted Edwards group
 * [ref src/lib.rs:15 `AffinePoint`] ark_ec::twisted_edwards, extended coordinates */
typedef struct { fp X, Y, Z, T; } pt;
static void pt_identity(pt* p) { q_zero(&p->X); q_one(&p->Y); q_one(&p->Z); q_zero(&p->T); }
static void pt_from_affine(pt* p, const fp* x, const fp* y) { p->X = *x; p->Y = *y; q_one(&p->Z); q_mul(&p->T, x, y); }
static void pt_add(pt* r, const pt* p, const pt* q) {          /* add-2008-hwcd */
  fp A, B, C, D, E, F, G, H, t0, t1;
  q_mul(&A, &p->X, &q->X); q_mul(&B, &p->Y, &q->Y);
  q_mul(&C, &p->T, &q->T); q_mul(&C, &C, &BS_D_M);
  q_mul(&D, &p->Z, &q->Z);
  q_add(&t0, &p->X, &p->Y); q_add(&t1, &q->X, &q->Y); q_mul(&E, &t0, &t1); q_sub(&E, &E, &A); q_sub(&E, &E, &B);
  q_sub(&F, &D, &C); q_add(&G, &D, &C);
  q_mul(&t0, &BS_A_M, &A); q_sub(&H, &B, &t0);
  q_mul(&r->X, &E, &F); q_mul(&r->Y, &G, &H); q_mul(&r->T, &E, &H); q_mul(&r->Z, &F, &G);
}
static void pt_double(pt* r, const pt* p) {                   /* dbl-2008-hwcd */
  fp A, B, C, D, E, F, G, H, t0;
  q_sqr(&A, &p->X); q_sqr(&B, &p->Y); q_sqr(&C, &p->Z); q_add(&C, &C, &C);
  q_mul(&D, &BS_A_M, &A);
  q_add(&t0, &p->X, &p->Y); q_sqr(&E, &t0); q_sub(&E, &E, &A); q_sub(&E, &E, &B);
  q_add(&G, &D, &B); q_sub(&F, &G, &C); q_sub(&H, &D, &B);
  q_mul(&r->X, &E, &F); q_mul(&r->Y, &G, &H); q_mul(&r->T, &E, &H); q_mul(&r->Z, &F, &G);
}
static void pt_neg(pt* r, const pt* p) { *r = *p; q_neg(&r->X, &p->X); q_neg(&r->T, &p->T); }
/* [ark_ec mul_bigint] MSB-first double-and-add over a 256-bit scalar */
static void pt_mul(pt* r, const pt* p, const uint64_t k[4]) {
  pt acc; pt_identity(&acc);
  int started = 0;
  for (int i = 255; i >= 0; --i) {
    if (started) pt_double(&acc, &acc);
    if ((k[i >> 6] >> (i & 63)) & 1) { pt_add(&acc, &acc, p); started = 1; }
  }
  *r = acc;
}
static void pt_to_affine(fp* x, fp* y, const pt* p) {
  fp zi; q_inv(&zi, &p->Z); q_mul(x, &p->X, &zi); q_mul(y, &p->Y, &zi);
}
static int pt_is_identity(const pt* p) { return q_is_zero(&p->X) && q_eq(&p->Y, &p->Z); }

/* ------------------------------------------------------------------ codec
 * [ref src/lib.rs:14 `codec`] ArkworksCodec, SURVEY.md A.1 */
static void load_le(uint64_t w[4], const uint8_t b[32]) {
  for (int i = 0; i < 4; ++i) { uint64_t x = 0; for (int j = 7; j >= 0; --j) x = (x << 8) | b[8 * i + j]; w[i] = x; }
}
static void store_le(uint8_t b[32], const uint64_t w[4]) {
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) b[8 * i + j] = (uint8_t)(w[i] >> (8 * j));
}
static void point_encode(uint8_t out[32], const fp* x, const fp* y) {
  uint64_t xi[4], yi[4], nx[4];
  q_to_int(xi, x); q_to_int(yi, y);
  store_le(out, yi);
  if (S_->flags & 1) { if (xi[0] & 1) out[31] |= 0x80; return; }           /* RFC 8032: x mod 2 */
  if (!is_zero4(xi)) { sub4(nx, FQ.m, xi); if (cmp4(xi, nx) > 0) out[31] |= 0x80; }
}
static int point_decode(fp* x, fp* y, const uint8_t in[32]) {
  uint8_t raw[32]; memcpy(raw, in, 32);
  int flag = raw[31] >> 7; raw[31] &= 0x7f;
  uint64_t yi[4]; load_le(yi, raw);
  if (cmp4(yi, FQ.m) >= 0) return 0;
  q_from_int(y, yi);
  fp y2, num, den, one, x2, r;
  q_one(&one); q_sqr(&y2, y);
  q_sub(&num, &one, &y2);
  q_mul(&den, &BS_D_M, &y2); q_sub(&den, &BS_A_M, &den);          /* a - d*y^2 */
  if (q_is_zero(&den)) return 0;
  q_inv(&den, &den); q_mul(&x2, &num, &den);
  if (!q_sqrt(&r, &x2)) return 0;
  uint64_t ri[4], ni[4]; q_to_int(ri, &r);
  fp nr; q_neg(&nr, &r); q_to_int(ni, &nr);
  if (S_->flags & 1) {                          /* RFC 8032 5.1.3: the root with the flagged parity; x = 0 with the flag fails */
    if (is_zero4(ri) && flag) return 0;
    *x = ((int)(ri[0] & 1) == flag) ? r : nr;
    return 1;
  }
  int r_is_greater = cmp4(ri, ni) > 0;
  *x = (r_is_greater == flag) ? r : nr;
  return 1;
}

/* ------------------------------------------------------------------ SHA-512 (FIPS 180-4) */
static const uint64_t K512[80] = {
  0x428a2f98d728ae22ULL,0x7137449123ef65cdULL,0xb5c0fbcfec4d3b2fULL,0xe9b5dba58189dbbcULL,0x3956c25bf348b538ULL,
  0x59f111f1b605d019ULL,0x923f82a4af194f9bULL,0xab1c5ed5da6d8118ULL,0xd807aa98a3030242ULL,0x12835b0145706fbeULL,
  0x243185be4ee4b28cULL,0x550c7dc3d5ffb4e2ULL,0x72be5d74f27b896fULL,0x80deb1fe3b1696b1ULL,0x9bdc06a725c71235ULL,
  0xc19bf174cf692694ULL,0xe49b69c19ef14ad2ULL,0xefbe4786384f25e3ULL,0x0fc19dc68b8cd5b5ULL,0x240ca1cc77ac9c65ULL,
  0x2de92c6f592b0275ULL,0x4a7484aa6ea6e483ULL,0x5cb0a9dcbd41fbd4ULL,0x76f988da831153b5ULL,0x983e5152ee66dfabULL,
  0xa831c66d2db43210ULL,0xb00327c898fb213fULL,0xbf597fc7beef0ee4ULL,0xc6e00bf33da88fc2ULL,0xd5a79147930aa725ULL,
  0x06ca6351e003826fULL,0x142929670a0e6e70ULL,0x27b70a8546d22ffcULL,0x2e1b21385c26c926ULL,0x4d2c6dfc5ac42aedULL,
  0x53380d139d95b3dfULL,0x650a73548baf63deULL,0x766a0abb3c77b2a8ULL,0x81c2c92e47edaee6ULL,0x92722c851482353bULL,
  0xa2bfe8a14cf10364ULL,0xa81a664bbc423001ULL,0xc24b8b70d0f89791ULL,0xc76c51a30654be30ULL,0xd192e819d6ef5218ULL,
  0xd69906245565a910ULL,0xf40e35855771202aULL,0x106aa07032bbd1b8ULL,0x19a4c116b8d2d0c8ULL,0x1e376c085141ab53ULL,
  0x2748774cdf8eeb99ULL,0x34b0bcb5e19b48a8ULL,0x391c0cb3c5c95a63ULL,0x4ed8aa4ae3418acbULL,0x5b9cca4f7763e373ULL,
  0x682e6ff3d6b2b8a3ULL,0x748f82ee5defb2fcULL,0x78a5636f43172f60ULL,0x84c87814a1f0ab72ULL,0x8cc702081a6439ecULL,
  0x90befffa23631e28ULL,0xa4506cebde82bde9ULL,0xbef9a3f7b2c67915ULL,0xc67178f2e372532bULL,0xca273eceea26619cULL,
  0xd186b8c721c0c207ULL,0xeada7dd6cde0eb1eULL,0xf57d4f7fee6ed178ULL,0x06f067aa72176fbaULL,0x0a637dc5a2c898a6ULL,
  0x113f9804bef90daeULL,0x1b710b35131c471bULL,0x28db77f523047d84ULL,0x32caab7b40c72493ULL,0x3c9ebe0a15c9bebcULL,
  0x431d67c49c100d4cULL,0x4cc5d4becb3e42b6ULL,0x597f299cfc657e2aULL,0x5fcb6fab3ad6faecULL,0x6c44198c4a475817ULL};
typedef struct { uint64_t h[8]; uint8_t buf[128]; size_t len, total; } sha512_ctx;
#define ROR(x, n) (((x) >> (n)) | ((x) << (64 - (n))))
static void sha512_block(sha512_ctx* c, const uint8_t* p) {
  uint64_t w[80], a, b, cc, d, e, f, g, h;
  for (int i = 0; i < 16; ++i) { uint64_t x = 0; for (int j = 0; j < 8; ++j) x = (x << 8) | p[8 * i + j]; w[i] = x; }
  for (int i = 16; i < 80; ++i) {
    uint64_t s0 = ROR(w[i - 15], 1) ^ ROR(w[i - 15], 8) ^ (w[i - 15] >> 7);
    uint64_t s1 = ROR(w[i - 2], 19) ^ ROR(w[i - 2], 61) ^ (w[i - 2] >> 6);
    w[i] = w[i - 16] + s0 + w[i - 7] + s1;
  }
  a = c->h[0]; b = c->h[1]; cc = c->h[2]; d = c->h[3]; e = c->h[4]; f = c->h[5]; g = c->h[6]; h = c->h[7];
  for (int i = 0; i < 80; ++i) {
    uint64_t t1 = h + (ROR(e, 14) ^ ROR(e, 18) ^ ROR(e, 41)) + ((e & f) ^ (~e & g)) + K512[i] + w[i];
    uint64_t t2 = (ROR(a, 28) ^ ROR(a, 34) ^ ROR(a, 39)) + ((a & b) ^ (a & cc) ^ (b & cc));
    h = g; g = f; f = e; e = d + t1; d = cc; cc = b; b = a; a = t1 + t2;
  }
  c->h[0] += a; c->h[1] += b; c->h[2] += cc; c->h[3] += d; c->h[4] += e; c->h[5] += f; c->h[6] += g; c->h[7] += h;
}
static void sha512_init(sha512_ctx* c) {
  static const uint64_t iv[8] = {0x6a09e667f3bcc908ULL,0xbb67ae8584caa73bULL,0x3c6ef372fe94f82bULL,0xa54ff53a5f1d36f1ULL,
                                 0x510e527fade682d1ULL,0x9b05688c2b3e6c1fULL,0x1f83d9abfb41bd6bULL,0x5be0cd19137e2179ULL};
  memcpy(c->h, iv, sizeof iv); c->len = 0; c->total = 0;
}
static void sha512_update(sha512_ctx* c, const void* data, size_t n) {
  const uint8_t* p = (const uint8_t*)data;
  c->total += n;
  while (n) {
    size_t k = 128 - c->len; if (k > n) k = n;
    memcpy(c->buf + c->len, p, k); c->len += k; p += k; n -= k;
    if (c->len == 128) { sha512_block(c, c->buf); c->len = 0; }
  }
}
static void sha512_final(sha512_ctx* c, uint8_t out[64]) {
  uint64_t bits = (uint64_t)c->total * 8;
  uint8_t pad = 0x80; size_t t = c->total;
  sha512_update(c, &pad, 1);
  uint8_t z = 0;
  while (c->len != 112) sha512_update(c, &z, 1);
  uint8_t lenb[16] = {0};
  for (int i = 0; i < 8; ++i) lenb[15 - i] = (uint8_t)(bits >> (8 * i));
  sha512_update(c, lenb, 16);
  c->total = t;
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) out[8 * i + j] = (uint8_t)(c->h[i] >> (56 - 8 * j));
}

/* ------------------------------------------------------------------ scalar helpers (mod r) */
static void r_from_bytes_wide(uint64_t out[4], const uint8_t* b, size_t n, int big_endian) {
  /* integer of n <= 64 bytes reduced mod r: Horner over bytes in Montgomery form */
  uint64_t acc[4] = {0, 0, 0, 0}, c256[4] = {256, 0, 0, 0}, m256[4];
  f_to_mont(&FR, m256, c256);
  for (size_t i = 0; i < n; ++i) {
    uint8_t byte = big_endian ? b[i] : b[n - 1 - i];
    uint64_t d[4] = {byte, 0, 0, 0}, dm[4];
    f_to_mont(&FR, dm, d);
    f_mul(&FR, acc, acc, m256);
    f_add(&FR, acc, acc, dm);
  }
  f_from_mont(&FR, out, acc);
}
static void r_muladd(uint64_t out[4], const uint64_t a[4], const uint64_t b[4], const uint64_t c[4]) { /* a*b + c */
  uint64_t am[4], bm[4], cm[4], t[4];
  f_to_mont(&FR, am, a); f_to_mont(&FR, bm, b); f_to_mont(&FR, cm, c);
  f_mul(&FR, t, am, bm); f_add(&FR, t, t, cm); f_from_mont(&FR, out, t);
}

/* ------------------------------------------------------------------ suite glue */
#define SUITE_ID (S_->suite_id)
#define SUITE_ID_LEN (S_->suite_id_len)
static const char H2C_DST[] = "ECVRF_Bandersnatch_XMD:SHA-512_ELL2_RO_Bandersnatch_SHA-512_ELL2";

static void basefield_init(basefield* b, const uint64_t m[4], uint64_t z) {
  field_init(&b->f, m);
  uint64_t one[4] = {1, 0, 0, 0}, zz[4] = {z, 0, 0, 0};
  sub4(b->ts_t, m, one);
  b->s = 0;
  while (!(b->ts_t[0] & 1)) { for (int i = 0; i < 4; ++i) b->ts_t[i] = (b->ts_t[i] >> 1) | (i < 3 ? b->ts_t[i + 1] << 63 : 0); b->s++; }
  add4(b->ts_e, b->ts_t, one);
  for (int i = 0; i < 4; ++i) b->ts_e[i] = (b->ts_e[i] >> 1) | (i < 3 ? b->ts_e[i + 1] << 63 : 0);
  fp zm; f_to_mont(&b->f, zm.v, zz);
  f_pow(&b->f, b->ts_c.v, zm.v, b->ts_t);      /* z is a non-residue: 5 (BLS12-381 Fr, BN254 Fr), 2 (2^255 - 19) */
}
static void suite_point(suite_t* s, fp* dst, const uint64_t v[4]) { f_to_mont(&s->bf->f, dst->v, v); }
static void do_init(void) {
  basefield_init(&BF_BLS, P_Q, 5); basefield_init(&BF_25519, P_ED_Q, 2); basefield_init(&BF_BN254, P_BJ_Q, 5);
  SUITE_BS.bf = &BF_BLS; SUITE_JJ.bf = &BF_BLS; SUITE_ED.bf = &BF_25519; SUITE_BJ.bf = &BF_BN254;
  SUITE_BS.challenge_len = SUITE_JJ.challenge_len = SUITE_BJ.challenge_len = 32; SUITE_ED.challenge_len = 16;
  S_ = &SUITE_BS;
  fp five, fone; q_from_u64(&five, 5); q_from_u64(&fone, 1);
  field_init(&SUITE_BS.fr, P_R_ORDER);
  q_from_int(&SUITE_BS.d, P_BS_D); q_neg(&SUITE_BS.a, &five);
  q_from_int(&SUITE_BS.gx, P_BS_GX); q_from_int(&SUITE_BS.gy, P_BS_GY);
  q_from_int(&SUITE_BS.bx, P_BS_BX); q_from_int(&SUITE_BS.by, P_BS_BY);
  SUITE_BS.suite_id = "Bandersnatch_SHA-512_ELL2"; SUITE_BS.suite_id_len = 25; SUITE_BS.h2c_tai = 0; SUITE_BS.cofactor_log2 = 2;
  SUITE_BS.h2c_dst = H2C_DST; SUITE_BS.h2c_dst_len = 64;
  field_init(&SUITE_JJ.fr, P_JJ_R_ORDER);
  q_from_int(&SUITE_JJ.d, P_JJ_D); q_neg(&SUITE_JJ.a, &fone);
  q_from_int(&SUITE_JJ.gx, P_JJ_GX); q_from_int(&SUITE_JJ.gy, P_JJ_GY);
  q_from_int(&SUITE_JJ.bx, P_JJ_BX); q_from_int(&SUITE_JJ.by, P_JJ_BY);
  SUITE_JJ.suite_id = "JubJub_SHA-512_TAI"; SUITE_JJ.suite_id_len = 18; SUITE_JJ.h2c_tai = 1; SUITE_JJ.cofactor_log2 = 3;
  q_from_int(&ELL_J_M, P_BS_J); q_from_int(&ELL_K_M, P_BS_K); q_from_u64(&ELL_Z_M, 5);
  fp ki; q_inv(&ki, &ELL_K_M); q_mul(&ELL_JK_M, &ELL_J_M, &ki); q_sqr(&ELL_K2I_M, &ki);
  /* Ed25519: a = -1; Baby-JubJub: a = +1 (each in its own base field) */
  field_init(&SUITE_ED.fr, P_ED_R_ORDER);
  { uint64_t o1[4] = {1, 0, 0, 0}; fp t; suite_point(&SUITE_ED, &t, o1); fp z; memset(&z, 0, sizeof z); f_sub(&BF_25519.f, SUITE_ED.a.v, z.v, t.v);
    suite_point(&SUITE_BJ, &SUITE_BJ.a, o1); }
  suite_point(&SUITE_ED, &SUITE_ED.d, P_ED_D); suite_point(&SUITE_ED, &SUITE_ED.gx, P_ED_GX); suite_point(&SUITE_ED, &SUITE_ED.gy, P_ED_GY);
  suite_point(&SUITE_ED, &SUITE_ED.bx, P_ED_BX); suite_point(&SUITE_ED, &SUITE_ED.by, P_ED_BY);
  SUITE_ED.suite_id = "Ed25519_SHA-512_TAI"; SUITE_ED.suite_id_len = 19; SUITE_ED.h2c_tai = 1; SUITE_ED.cofactor_log2 = 3;
  field_init(&SUITE_BJ.fr, P_BJ_R_ORDER);
  suite_point(&SUITE_BJ, &SUITE_BJ.d, P_BJ_D); suite_point(&SUITE_BJ, &SUITE_BJ.gx, P_BJ_GX); suite_point(&SUITE_BJ, &SUITE_BJ.gy, P_BJ_GY);
  suite_point(&SUITE_BJ, &SUITE_BJ.bx, P_BJ_BX); suite_point(&SUITE_BJ, &SUITE_BJ.by, P_BJ_BY);
  SUITE_BJ.suite_id = "BabyJubJub_SHA-512_TAI"; SUITE_BJ.suite_id_len = 22; SUITE_BJ.h2c_tai = 1; SUITE_BJ.cofactor_log2 = 3;
  g_init_done = 1;
}
static void ensure_init(void) { pthread_once(&g_once, do_init); }

/* [ref src/lib.rs:14 `utils::hash_to_curve_ell2_rfc_9380`] SURVEY.md A.3 */
static void elligator2(pt* out, const fp* u) {
  fp jk = ELL_JK_M, k2i = ELL_K2I_M, den, one, x1, gx1, x2, gx2, t, y, x, s, tt;
  q_one(&one);
  q_sqr(&t, u); q_mul(&t, &t, &ELL_Z_M); q_add(&den, &one, &t);
  if (q_is_zero(&den)) den = one;
  q_inv(&den, &den); q_neg(&x1, &jk); q_mul(&x1, &x1, &den);
  /* g(x) = x^3 + (J/K) x^2 + x/K^2 */
  q_sqr(&t, &x1); q_mul(&gx1, &t, &x1); q_mul(&t, &t, &jk); q_add(&gx1, &gx1, &t); q_mul(&t, &x1, &k2i); q_add(&gx1, &gx1, &t);
  q_neg(&x2, &x1); q_sub(&x2, &x2, &jk);
  q_sqr(&t, &x2); q_mul(&gx2, &t, &x2); q_mul(&t, &t, &jk); q_add(&gx2, &gx2, &t); q_mul(&t, &x2, &k2i); q_add(&gx2, &gx2, &t);
  uint64_t yi[4];
  if (q_legendre(&gx1) >= 0) {
    x = x1; q_sqrt(&y, &gx1); q_to_int(yi, &y);
    if (!(yi[0] & 1)) q_neg(&y, &y);            /* want y odd */
  } else {
    x = x2; q_sqrt(&y, &gx2); q_to_int(yi, &y);
    if (yi[0] & 1) q_neg(&y, &y);               /* want y even */
  }
  q_mul(&s, &x, &ELL_K_M); q_mul(&tt, &y, &ELL_K_M);
  fp sp1, sm1, chk;
  q_add(&sp1, &s, &one); q_sub(&sm1, &s, &one); q_mul(&chk, &tt, &sp1);
  if (q_is_zero(&chk)) { pt_identity(out); return; }
  fp v, w, ti;
  q_inv(&ti, &tt); q_mul(&v, &s, &ti);
  q_inv(&ti, &sp1); q_mul(&w, &sm1, &ti);
  pt_from_affine(out, &v, &w);
}
static int point_decode(fp* x, fp* y, const uint8_t in[32]);
static void pt_double(pt* r, const pt* p);
static int pt_is_identity(const pt* p);
/* [ref src/lib.rs:14 `utils::hash_to_curve_tai_rfc_9381`] SURVEY.md A.6 (unpinned) */
static int hash_to_curve_tai(pt* out, const uint8_t* msg, size_t len) {
  for (int ctr = 0; ctr < 256; ++ctr) {
    uint8_t h[64], one = 1, cb = (uint8_t)ctr, zero = 0; sha512_ctx c;
    sha512_init(&c); sha512_update(&c, SUITE_ID, SUITE_ID_LEN); sha512_update(&c, &one, 1);
    sha512_update(&c, msg, len); sha512_update(&c, &cb, 1); sha512_update(&c, &zero, 1); sha512_final(&c, h);
    fp x, y;
    if (!point_decode(&x, &y, h)) continue;
    pt p; pt_from_affine(&p, &x, &y);
    for (int i = 0; i < S_->cofactor_log2; ++i) pt_double(&p, &p);
    if (pt_is_identity(&p)) continue;
    *out = p;
    return 1;
  }
  pt_identity(out);
  return 0;
}
static void hash_to_curve(pt* out, const uint8_t* msg, size_t len) {
  if (S_->h2c_tai) { hash_to_curve_tai(out, msg, len); return; }
  uint8_t dstp[129]; const size_t dl = S_->h2c_dst_len;
  memcpy(dstp, S_->h2c_dst, dl); dstp[dl] = (uint8_t)dl;
  uint8_t zpad[48] = {0}, lib[3] = {0x00, 0x60, 0x00}, b0[64], b1[64], b2[64], x[64];
  sha512_ctx c;
  sha512_init(&c); sha512_update(&c, zpad, 48); sha512_update(&c, msg, len); sha512_update(&c, lib, 3);
  sha512_update(&c, dstp, dl + 1); sha512_final(&c, b0);
  uint8_t one = 1, two = 2;
  sha512_init(&c); sha512_update(&c, b0, 64); sha512_update(&c, &one, 1); sha512_update(&c, dstp, dl + 1); sha512_final(&c, b1);
  for (int i = 0; i < 64; ++i) x[i] = b0[i] ^ b1[i];
  sha512_init(&c); sha512_update(&c, x, 64); sha512_update(&c, &two, 1); sha512_update(&c, dstp, dl + 1); sha512_final(&c, b2);
  uint8_t uni[96]; memcpy(uni, b1, 64); memcpy(uni + 64, b2, 32);
  fp u[2];
  for (int k = 0; k < 2; ++k) {              /* 48-byte big-endian integer mod q, Horner */
    fp acc, c256; q_zero(&acc); q_from_u64(&c256, 256);
    for (int i = 0; i < 48; ++i) { fp d; q_from_u64(&d, uni[48 * k + i]); q_mul(&acc, &acc, &c256); q_add(&acc, &acc, &d); }
    u[k] = acc;
  }
  pt q0, q1, s; elligator2(&q0, &u[0]); elligator2(&q1, &u[1]);
  pt_add(&s, &q0, &q1); pt_double(&s, &s); pt_double(&s, &s);   /* cofactor 4 */
  *out = s;
}

/* [ref src/lib.rs:14,16 `Suite::nonce`] SURVEY.md A.4 */
static void nonce(uint64_t k[4], const uint8_t sk_le[32], const uint8_t h_enc[32]) {
  uint8_t h1[64], h2[64]; sha512_ctx c;
  sha512_init(&c); sha512_update(&c, sk_le, 32); sha512_final(&c, h1);
  sha512_init(&c); sha512_update(&c, h1 + 32, 32); sha512_update(&c, h_enc, 32); sha512_final(&c, h2);
  r_from_bytes_wide(k, h2, 64, 0);
}
/* [ref src/lib.rs:14,16 `Suite::challenge`] SURVEY.md A.4 */
/* `point_encode` of the typed point a wire encoding decodes to: arkworks accepts x = 0 (y = 1 or y = q - 1) with the sign
 * flag set and encodes it with the flag clear; every other accepted encoding is already canonical. */
static void enc_canonical(uint8_t out[32], const uint8_t in[32]) {
  memcpy(out, in, 32);
  uint8_t t[32]; memcpy(t, in, 32); t[31] &= 0x7f;
  uint64_t v[4], qm1[4]; load_le(v, t);
  memcpy(qm1, FQ.m, sizeof qm1); qm1[0] -= 1;                 /* q is odd */
  if ((v[0] == 1 && !v[1] && !v[2] && !v[3]) || cmp4(v, qm1) == 0) out[31] &= 0x7f;
}
static void challenge(uint64_t c_out[4], const uint8_t pts[5][32], const uint8_t* ad, size_t ad_len) {
  uint8_t h[64], two = 2, zero = 0; sha512_ctx c;
  sha512_init(&c); sha512_update(&c, SUITE_ID, SUITE_ID_LEN); sha512_update(&c, &two, 1);
  for (int i = 0; i < 5; ++i) { uint8_t e[32]; enc_canonical(e, pts[i]); sha512_update(&c, e, 32); }
  sha512_update(&c, ad, ad_len); sha512_update(&c, &zero, 1); sha512_final(&c, h);
  r_from_bytes_wide(c_out, h, (size_t)S_->challenge_len, !(S_->flags & 2));   /* CHALLENGE_LEN leading bytes, big-endian upstream */
}

/* Checked deserialisation [ref src/lib.rs:14 `codec`: arkworks validates on-curve AND subgroup membership when
 * `Public` / `Input` / `Output` / proof points are decoded].  g_check_mask selects which point classes get the
 * r*P == O test (bits as include/vrfhip.h VRFHIP_FLAG_PREVALIDATED_*: 1 public key, 2 input, 4 output, 8 proof
 * points); 0 = on-curve only (the caller vouches for the subgroup).  Default: everything checked, as upstream. */
static int g_check_mask = 15;
static void pt_mul(pt* r, const pt* p, const uint64_t k[4]);
static int point_decode_chk(fp* x, fp* y, const uint8_t in[32], int bit) {
  if (!point_decode(x, y, in)) return 0;
  if (g_check_mask & bit) {
    pt p, rp; pt_from_affine(&p, x, y); pt_mul(&rp, &p, FR.m);
    if (!pt_is_identity(&rp)) return 0;
  }
  return 1;
}

/* ------------------------------------------------------------------ exported API */
/* 1 = Bandersnatch_SHA-512_ELL2 (default), 2 = JubJub_SHA-512_TAI, 3 = Ed25519_SHA-512_TAI, 4 = BabyJubJub_SHA-512_TAI.
 * Process-global: tests only. */
void oracle_set_check_mask(int mask) { g_check_mask = mask; }
/* A suite from a descriptor (include/vrfhip.h vrfhip_suite_desc): curve 1 = Bandersnatch (Elligator 2), 2 = JubJub
 * (try-and-increment); suite string, hash-to-curve DST, generator and blinding base (x || y, 32-byte little-endian)
 * supplied by the caller.  Selects it as the current suite.  Returns -1 on a bad argument (the points are not
 * validated here: the product does that). */
int oracle_set_suite_desc2(int curve, const uint8_t* suite_id, size_t id_len, const uint8_t* dst, size_t dst_len,
                           const uint8_t g_xy[64], const uint8_t b_xy[64], int challenge_len, int flags);
int oracle_set_suite_desc(int curve, const uint8_t* suite_id, size_t id_len, const uint8_t* dst, size_t dst_len,
                          const uint8_t g_xy[64], const uint8_t b_xy[64]) {
  return oracle_set_suite_desc2(curve, suite_id, id_len, dst, dst_len, g_xy, b_xy, curve == 3 ? 16 : 32, 0);
}
/* curve: 1 Bandersnatch, 2 JubJub, 3 Ed25519, 4 Baby-JubJub (vrfhip_curve); challenge_len 1..32; flags as suite_t.flags */
int oracle_set_suite_desc2(int curve, const uint8_t* suite_id, size_t id_len, const uint8_t* dst, size_t dst_len,
                           const uint8_t g_xy[64], const uint8_t b_xy[64], int challenge_len, int flags) {
  ensure_init();
  if (curve < 1 || curve > 4 || id_len == 0 || id_len > sizeof CUSTOM_ID || dst_len > sizeof CUSTOM_DST) return -1;
  if (challenge_len < 1 || challenge_len > 32 || (flags & ~7)) return -1;
  SUITE_CUSTOM = curve == 1 ? SUITE_BS : curve == 2 ? SUITE_JJ : curve == 3 ? SUITE_ED : SUITE_BJ;
  SUITE_CUSTOM.challenge_len = challenge_len; SUITE_CUSTOM.flags = flags;
  S_ = &SUITE_CUSTOM;                          /* the conversions below run in the suite's base field */
  memcpy(CUSTOM_ID, suite_id, id_len); SUITE_CUSTOM.suite_id = CUSTOM_ID; SUITE_CUSTOM.suite_id_len = id_len;
  if (dst_len) memcpy(CUSTOM_DST, dst, dst_len);
  SUITE_CUSTOM.h2c_dst = CUSTOM_DST; SUITE_CUSTOM.h2c_dst_len = dst_len;
  uint64_t t[4];
  load_le(t, g_xy); q_from_int(&SUITE_CUSTOM.gx, t); load_le(t, g_xy + 32); q_from_int(&SUITE_CUSTOM.gy, t);
  load_le(t, b_xy); q_from_int(&SUITE_CUSTOM.bx, t); load_le(t, b_xy + 32); q_from_int(&SUITE_CUSTOM.by, t);
  S_ = &SUITE_CUSTOM;
  return 0;
}
int oracle_set_suite(int id) {
  ensure_init();
  if (id == 1) S_ = &SUITE_BS; else if (id == 2) S_ = &SUITE_JJ; else if (id == 3) S_ = &SUITE_ED;
  else if (id == 4) S_ = &SUITE_BJ; else return -1;
  return 0;
}
int oracle_secret_from_seed(const uint8_t* seed, size_t len, uint8_t sk_out[32]) {
  ensure_init();
  uint8_t h[64]; sha512_ctx c; sha512_init(&c); sha512_update(&c, seed, len); sha512_final(&c, h);
  uint64_t sk[4]; r_from_bytes_wide(sk, h, 64, 0); store_le(sk_out, sk);
  return 0;
}
int oracle_public_from_secret(const uint8_t sk_le[32], uint8_t pk_out[32]) {
  ensure_init();
  uint64_t sk[4]; load_le(sk, sk_le);
  pt g, p; pt_from_affine(&g, &BS_GX_M, &BS_GY_M); pt_mul(&p, &g, sk);
  fp x, y; pt_to_affine(&x, &y, &p); point_encode(pk_out, &x, &y);
  return 0;
}
int oracle_hash_to_curve(const uint8_t* msg, size_t len, uint8_t out[32]) {
  ensure_init();
  pt h; hash_to_curve(&h, msg, len);
  fp x, y; pt_to_affine(&x, &y, &h); point_encode(out, &x, &y);
  return 0;
}
int oracle_output_hash(const uint8_t gamma[32], uint8_t out[64]) {
  ensure_init();
  uint8_t three = 3, zero = 0; sha512_ctx c;
  uint8_t e[32]; enc_canonical(e, gamma);       /* `Output::hash` encodes the typed point */
  if (S_->flags & 4) {                          /* RFC 9381 proof_to_hash: cofactor * Gamma */
    fp x, y; pt p;
    if (!point_decode(&x, &y, gamma)) return 2;
    pt_from_affine(&p, &x, &y);
    for (int i = 0; i < S_->cofactor_log2; ++i) pt_double(&p, &p);
    pt_to_affine(&x, &y, &p); point_encode(e, &x, &y);
  }
  sha512_init(&c); sha512_update(&c, SUITE_ID, SUITE_ID_LEN); sha512_update(&c, &three, 1); sha512_update(&c, e, 32);
  sha512_update(&c, &zero, 1); sha512_final(&c, out);
  return 0;
}
/* returns 0 ok / 2 invalid; subgroup != 0 adds the r*P == O check (arkworks checked decode) */
int oracle_point_decode(const uint8_t in[32], int subgroup, uint8_t xy_out[64]) {
  ensure_init();
  fp x, y;
  if (!point_decode(&x, &y, in)) return 2;
  if (subgroup) { pt p, rp; pt_from_affine(&p, &x, &y); pt_mul(&rp, &p, FR.m); if (!pt_is_identity(&rp)) return 2; }
  if (xy_out) { uint64_t t[4]; q_to_int(t, &x); store_le(xy_out, t); q_to_int(t, &y); store_le(xy_out + 32, t); }
  return 0;
}
/* [ref src/lib.rs:14 `ietf::Prover::prove`]; h_given != NULL skips hash-to-curve */
int oracle_ietf_prove(const uint8_t sk_le[32], const uint8_t* msg, size_t msg_len, const uint8_t* h_given,
                      const uint8_t* ad, size_t ad_len, uint8_t gamma_out[32], uint8_t c_out[32],
                      uint8_t s_out[32], uint8_t pk_out[32], uint8_t h_out[32]) {
  ensure_init();
  uint64_t sk[4]; load_le(sk, sk_le);
  if (cmp4(sk, FR.m) >= 0) return 2;
  pt H, G, Gm, PK, KG, KH; fp x, y;
  uint8_t pts[5][32];
  if (h_given) {
    if (!point_decode_chk(&x, &y, h_given, 2)) return 2;
    pt_from_affine(&H, &x, &y); memcpy(pts[1], h_given, 32);
  }
  else { hash_to_curve(&H, msg, msg_len); pt_to_affine(&x, &y, &H); point_encode(pts[1], &x, &y); pt_from_affine(&H, &x, &y); }
  pt_from_affine(&G, &BS_GX_M, &BS_GY_M);
  pt_mul(&PK, &G, sk); pt_to_affine(&x, &y, &PK); point_encode(pts[0], &x, &y);
  pt_mul(&Gm, &H, sk); pt_to_affine(&x, &y, &Gm); point_encode(pts[2], &x, &y);
  uint64_t k[4], c[4], s[4]; nonce(k, sk_le, pts[1]);
  pt_mul(&KG, &G, k); pt_to_affine(&x, &y, &KG); point_encode(pts[3], &x, &y);
  pt_mul(&KH, &H, k); pt_to_affine(&x, &y, &KH); point_encode(pts[4], &x, &y);
  challenge(c, pts, ad, ad_len);
  r_muladd(s, c, sk, k);
  memcpy(gamma_out, pts[2], 32); store_le(c_out, c); store_le(s_out, s);
  if (pk_out) memcpy(pk_out, pts[0], 32);
  if (h_out) memcpy(h_out, pts[1], 32);
  return 0;
}
/* [ref src/lib.rs:14 `ietf::Verifier::verify`]: four independent mul_bigint, as upstream */
int oracle_ietf_verify(const uint8_t pk[32], const uint8_t h[32], const uint8_t gamma[32],
                       const uint8_t c_le[32], const uint8_t s_le[32], const uint8_t* ad, size_t ad_len) {
  ensure_init();
  uint64_t c[4], s[4], c2[4];
  r_from_bytes_wide(c, c_le, 32, 0);           /* `Proof::c`: scalar_decode = from_le_bytes_mod_order */
  load_le(s, s_le);                            /* `Proof::s`: canonical deserialisation, strict */
  if (cmp4(s, FR.m) >= 0) return 2;
  fp x, y; pt Y, H, Gm, G, sG, cY, sH, cG, U, V, n;
  if (!point_decode_chk(&x, &y, pk, 1)) return 2;
  pt_from_affine(&Y, &x, &y);
  if (!point_decode_chk(&x, &y, h, 2)) return 2;
  pt_from_affine(&H, &x, &y);
  if (!point_decode_chk(&x, &y, gamma, 4)) return 2;
  pt_from_affine(&Gm, &x, &y);
  pt_from_affine(&G, &BS_GX_M, &BS_GY_M);
  pt_mul(&sG, &G, s); pt_mul(&cY, &Y, c); pt_neg(&n, &cY); pt_add(&U, &sG, &n);
  pt_mul(&sH, &H, s); pt_mul(&cG, &Gm, c); pt_neg(&n, &cG); pt_add(&V, &sH, &n);
  uint8_t pts[5][32];
  memcpy(pts[0], pk, 32); memcpy(pts[1], h, 32); memcpy(pts[2], gamma, 32);
  pt_to_affine(&x, &y, &U); point_encode(pts[3], &x, &y);
  pt_to_affine(&x, &y, &V); point_encode(pts[4], &x, &y);
  challenge(c2, pts, ad, ad_len);
  if (S_->challenge_len < 32) {
    /* upstream writes a short challenge with CHALLENGE_LEN bytes: a 32-byte field holding more is no proof string and is
     * compared as it stands -- it never equals a recomputed challenge (c + r no longer verifies) */
    uint64_t raw[4];
    load_le(raw, c_le);
    return cmp4(raw, c2) == 0 ? 0 : 1;
  }
  return cmp4(c, c2) == 0 ? 0 : 1;
}

/* [ref src/lib.rs:14 `pedersen::PedersenSuite::blinding`] SURVEY.md A.5 */
static void blinding(uint64_t b[4], const uint8_t sk_le[32], const uint8_t h_enc[32], const uint8_t* ad, size_t ad_len) {
  uint8_t h[64], cc = 0xCC, zero = 0; sha512_ctx c;
  sha512_init(&c); sha512_update(&c, SUITE_ID, SUITE_ID_LEN); sha512_update(&c, &cc, 1); sha512_update(&c, sk_le, 32);
  sha512_update(&c, h_enc, 32); sha512_update(&c, ad, ad_len); sha512_update(&c, &zero, 1); sha512_final(&c, h);
  r_from_bytes_wide(b, h, 64, 1);
}
/* [ref src/lib.rs:14 `pedersen::Prover::prove`]; proof160 = pk_com | R | Ok | s | sb */
int oracle_pedersen_prove(const uint8_t sk_le[32], const uint8_t* msg, size_t msg_len, const uint8_t* h_given,
                          const uint8_t* ad, size_t ad_len, uint8_t gamma_out[32], uint8_t proof160[160],
                          uint8_t blinding_out[32], uint8_t h_out[32]) {
  ensure_init();
  uint64_t sk[4]; load_le(sk, sk_le);
  if (cmp4(sk, FR.m) >= 0) return 2;
  pt H, G, B, t0, t1, P; fp x, y;
  uint8_t pts[5][32];
  if (h_given) {
    if (!point_decode_chk(&x, &y, h_given, 2)) return 2;
    memcpy(pts[1], h_given, 32);
  } else { hash_to_curve(&H, msg, msg_len); pt_to_affine(&x, &y, &H); point_encode(pts[1], &x, &y); }
  pt_from_affine(&H, &x, &y);
  pt_from_affine(&G, &BS_GX_M, &BS_GY_M); pt_from_affine(&B, &BS_BX_M, &BS_BY_M);
  uint64_t b[4], k[4], kb[4], c[4], s[4], sb[4]; uint8_t b_le[32];
  blinding(b, sk_le, pts[1], ad, ad_len); store_le(b_le, b);
  nonce(k, sk_le, pts[1]); nonce(kb, b_le, pts[1]);
  pt_mul(&P, &H, sk); pt_to_affine(&x, &y, &P); point_encode(pts[2], &x, &y);                       /* Gamma */
  pt_mul(&t0, &G, sk); pt_mul(&t1, &B, b); pt_add(&P, &t0, &t1); pt_to_affine(&x, &y, &P); point_encode(pts[0], &x, &y); /* pk_com */
  pt_mul(&t0, &G, k); pt_mul(&t1, &B, kb); pt_add(&P, &t0, &t1); pt_to_affine(&x, &y, &P); point_encode(pts[3], &x, &y); /* R */
  pt_mul(&P, &H, k); pt_to_affine(&x, &y, &P); point_encode(pts[4], &x, &y);                        /* Ok */
  challenge(c, pts, ad, ad_len);
  r_muladd(s, c, sk, k); r_muladd(sb, c, b, kb);
  memcpy(gamma_out, pts[2], 32);
  memcpy(proof160, pts[0], 32); memcpy(proof160 + 32, pts[3], 32); memcpy(proof160 + 64, pts[4], 32);
  store_le(proof160 + 96, s); store_le(proof160 + 128, sb);
  if (blinding_out) memcpy(blinding_out, b_le, 32);
  if (h_out) memcpy(h_out, pts[1], 32);
  return 0;
}
/* [ref src/lib.rs:14 `pedersen::Verifier::verify`] */
int oracle_pedersen_verify(const uint8_t h[32], const uint8_t gamma[32], const uint8_t proof160[160],
                           const uint8_t* ad, size_t ad_len) {
  ensure_init();
  uint64_t s[4], sb[4], c[4];
  load_le(s, proof160 + 96); load_le(sb, proof160 + 128);
  if (cmp4(s, FR.m) >= 0 || cmp4(sb, FR.m) >= 0) return 2;
  fp x, y; pt H, Gm, PC, R, Ok, G, B, l, r1, t0, t1;
  if (!point_decode_chk(&x, &y, h, 2)) return 2;
  pt_from_affine(&H, &x, &y);
  if (!point_decode_chk(&x, &y, gamma, 4)) return 2;
  pt_from_affine(&Gm, &x, &y);
  if (!point_decode_chk(&x, &y, proof160, 8)) return 2;
  pt_from_affine(&PC, &x, &y);
  if (!point_decode_chk(&x, &y, proof160 + 32, 8)) return 2;
  pt_from_affine(&R, &x, &y);
  if (!point_decode_chk(&x, &y, proof160 + 64, 8)) return 2;
  pt_from_affine(&Ok, &x, &y);
  uint8_t pts[5][32];
  memcpy(pts[0], proof160, 32); memcpy(pts[1], h, 32); memcpy(pts[2], gamma, 32);
  memcpy(pts[3], proof160 + 32, 32); memcpy(pts[4], proof160 + 64, 32);
  challenge(c, pts, ad, ad_len);
  pt_from_affine(&G, &BS_GX_M, &BS_GY_M); pt_from_affine(&B, &BS_BX_M, &BS_BY_M);
  /* Ok + c*Gamma == s*H */
  pt_mul(&t0, &Gm, c); pt_add(&l, &Ok, &t0); pt_mul(&r1, &H, s);
  fp ax, ay, bx, by; pt_to_affine(&ax, &ay, &l); pt_to_affine(&bx, &by, &r1);
  if (!q_eq(&ax, &bx) || !q_eq(&ay, &by)) return 1;
  /* R + c*pk_com == s*G + sb*B */
  pt_mul(&t0, &PC, c); pt_add(&l, &R, &t0); pt_mul(&t0, &G, s); pt_mul(&t1, &B, sb); pt_add(&r1, &t0, &t1);
  pt_to_affine(&ax, &ay, &l); pt_to_affine(&bx, &by, &r1);
  if (!q_eq(&ax, &bx) || !q_eq(&ay, &by)) return 1;
  return 0;
}

/* ---- batch drivers (static partition over pthreads; threads <= 1 runs inline) ---- */
typedef struct {
  int kind; size_t lo, hi;
  const uint8_t *a0, *a1, *a2, *a3, *a4; const uint8_t* ad; size_t ad_len; size_t msg_len;
  uint8_t *o0, *o1, *o2, *o3, *o4, *st;
} job;
static void* run_job(void* arg) {
  job* j = (job*)arg;
  for (size_t i = j->lo; i < j->hi; ++i) {
    if (j->kind == 0) {
      j->st[i] = (uint8_t)oracle_ietf_verify(j->a0 + 32 * i, j->a1 + 32 * i, j->a2 + 32 * i, j->a3 + 32 * i,
                                             j->a4 + 32 * i, j->ad, j->ad_len);
    } else if (j->kind == 2) {
      j->st[i] = (uint8_t)oracle_pedersen_verify(j->a0 + 32 * i, j->a1 + 32 * i, j->a2 + 160 * i, j->ad, j->ad_len);
    } else if (j->kind == 3) {
      int rc = oracle_pedersen_prove(j->a0 + 32 * i, j->a1 ? j->a1 + j->msg_len * i : NULL, j->msg_len,
                                     j->a2 ? j->a2 + 32 * i : NULL, j->ad, j->ad_len, j->o0 + 32 * i,
                                     j->o1 + 160 * i, j->o2 ? j->o2 + 32 * i : NULL, j->o4 ? j->o4 + 32 * i : NULL);
      if (j->st) j->st[i] = (uint8_t)rc;
    } else {
      int rc = oracle_ietf_prove(j->a0 + 32 * i, j->a1 ? j->a1 + j->msg_len * i : NULL, j->msg_len,
                                 j->a2 ? j->a2 + 32 * i : NULL, j->ad, j->ad_len, j->o0 + 32 * i, j->o1 + 32 * i,
                                 j->o2 + 32 * i, j->o3 ? j->o3 + 32 * i : NULL, j->o4 ? j->o4 + 32 * i : NULL);
      if (j->st) j->st[i] = (uint8_t)rc;
    }
  }
  return NULL;
}
static void run_batch(job base, size_t n, int threads) {
  ensure_init();
  if (threads <= 1 || n < 2) { base.lo = 0; base.hi = n; run_job(&base); return; }
  if ((size_t)threads > n) threads = (int)n;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * threads);
  job* jobs = (job*)malloc(sizeof(job) * threads);
  for (int t = 0; t < threads; ++t) {
    jobs[t] = base; jobs[t].lo = n * t / threads; jobs[t].hi = n * (t + 1) / threads;
    pthread_create(&th[t], NULL, run_job, &jobs[t]);
  }
  for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
  free(th); free(jobs);
}
void oracle_ietf_verify_batch(size_t n, const uint8_t* pk, const uint8_t* h, const uint8_t* gamma, const uint8_t* c,
                              const uint8_t* s, const uint8_t* ad, size_t ad_len, uint8_t* status, int threads) {
  job j; memset(&j, 0, sizeof j);
  j.kind = 0; j.a0 = pk; j.a1 = h; j.a2 = gamma; j.a3 = c; j.a4 = s; j.ad = ad; j.ad_len = ad_len; j.st = status;
  run_batch(j, n, threads);
}
/* fixed-stride messages (msg_len bytes each) or pre-hashed inputs */
void oracle_ietf_prove_batch(size_t n, const uint8_t* sk, const uint8_t* msg, size_t msg_len, const uint8_t* h_given,
                             const uint8_t* ad, size_t ad_len, uint8_t* gamma, uint8_t* c, uint8_t* s,
                             uint8_t* pk_out, uint8_t* h_out, uint8_t* status, int threads) {
  job j; memset(&j, 0, sizeof j);
  j.kind = 1; j.a0 = sk; j.a1 = msg; j.msg_len = msg_len; j.a2 = h_given; j.ad = ad; j.ad_len = ad_len;
  j.o0 = gamma; j.o1 = c; j.o2 = s; j.o3 = pk_out; j.o4 = h_out; j.st = status;
  run_batch(j, n, threads);
}
void oracle_pedersen_verify_batch(size_t n, const uint8_t* h, const uint8_t* gamma, const uint8_t* proof160,
                                  const uint8_t* ad, size_t ad_len, uint8_t* status, int threads) {
  job j; memset(&j, 0, sizeof j);
  j.kind = 2; j.a0 = h; j.a1 = gamma; j.a2 = proof160; j.ad = ad; j.ad_len = ad_len; j.st = status;
  run_batch(j, n, threads);
}
void oracle_pedersen_prove_batch(size_t n, const uint8_t* sk, const uint8_t* msg, size_t msg_len, const uint8_t* h_given,
                                 const uint8_t* ad, size_t ad_len, uint8_t* gamma, uint8_t* proof160,
                                 uint8_t* blinding_out, uint8_t* h_out, uint8_t* status, int threads) {
  job j; memset(&j, 0, sizeof j);
  j.kind = 3; j.a0 = sk; j.a1 = msg; j.msg_len = msg_len; j.a2 = h_given; j.ad = ad; j.ad_len = ad_len;
  j.o0 = gamma; j.o1 = proof160; j.o2 = blinding_out; j.o4 = h_out; j.st = status;
  run_batch(j, n, threads);
}
/* Batch digest hashed into the weights of the random-linear-combination checks (ark_ec_vrfs_amd/csrc/digest.cuh states the
 * definition): leaf_i = SHA-512("vrfhip-leaf-v1" || u64_le(index0 + i) || a_0[i] || ... || ad_i || u32_le(|ad_i|))[0..32];
 * node = SHA-512("vrfhip-node-v1" || u32_le(count) || children)[0..32] over runs of 16; root = the last level's one node.
 * arrs[j] + i * strides[j] is item i of array j (widths[j] bytes).  ad_off != NULL: per-item strings ad[ad_off[i]..ad_off[i+1]);
 * else every item carries ad[0..ad_len).  n >= 1. */
void oracle_batch_digest(size_t n, uint64_t index0, int n_arr, const uint8_t* const* arrs, const uint32_t* widths,
                         const uint32_t* strides, const uint8_t* ad, const uint32_t* ad_off, uint32_t ad_len,
                         uint8_t root[32]) {
  uint8_t* level = (uint8_t*)malloc(n * 32);
  for (size_t i = 0; i < n; ++i) {
    sha512_ctx hc; uint8_t dg[64], idx[8], al[4];
    uint64_t gi = index0 + i;
    for (int k = 0; k < 8; ++k) idx[k] = (uint8_t)(gi >> (8 * k));
    sha512_init(&hc); sha512_update(&hc, "vrfhip-leaf-v1", 14); sha512_update(&hc, idx, 8);
    for (int j = 0; j < n_arr; ++j) sha512_update(&hc, arrs[j] + i * (size_t)strides[j], widths[j]);
    const uint8_t* a = ad; uint32_t len = ad_len;
    if (ad_off) { a = ad + ad_off[i]; len = ad_off[i + 1] - ad_off[i]; }
    if (len) sha512_update(&hc, a, len);
    for (int k = 0; k < 4; ++k) al[k] = (uint8_t)(len >> (8 * k));
    sha512_update(&hc, al, 4); sha512_final(&hc, dg);
    memcpy(level + 32 * i, dg, 32);
  }
  size_t m = n;
  do {
    size_t nodes = (m + 15) / 16;
    for (size_t t = 0; t < nodes; ++t) {
      uint32_t cnt = (uint32_t)(m - t * 16 < 16 ? m - t * 16 : 16);
      sha512_ctx hc; uint8_t dg[64], cb[4];
      for (int k = 0; k < 4; ++k) cb[k] = (uint8_t)(cnt >> (8 * k));
      sha512_init(&hc); sha512_update(&hc, "vrfhip-node-v1", 14); sha512_update(&hc, cb, 4);
      sha512_update(&hc, level + 32 * 16 * t, 32 * (size_t)cnt); sha512_final(&hc, dg);
      memcpy(level + 32 * t, dg, 32);      /* node t overwrites children it has already consumed (t <= 16 t) */
    }
    m = nodes;
  } while (m > 1);
  memcpy(root, level, 32);
  free(level);
}
/* Batched Pedersen verification by random linear combination, the slow way:
 * for every proof the two defects D1 = s*H - c*Gamma - Ok and D2 = s*G + sb*B - c*pk_com - R are
 * computed by double-and-add, weighted by (z_i, z'_i) = the two little-endian 16-byte halves of
 * SHA-512("vrfhip-rlc-v2" || seed || batch digest || u64_le(index0 + i)), low three bits forced to 001, and summed; the digest
 * (oracle_batch_digest) covers H, Gamma, pk_com, R, Ok, s, sb and ad of the n proofs as ONE launch group.  status[i] = 0 (in
 * the sum) or 2 (undecodable: left out).  Returns 0 if the sum is the neutral element, else 1. */
int oracle_pedersen_rlc_check(size_t n, const uint8_t* h, const uint8_t* gamma, const uint8_t* proof160,
                              const uint8_t* ad, size_t ad_len, const uint8_t seed[32], uint64_t index0,
                              uint8_t* status) {
  ensure_init();
  pt acc; pt_identity(&acc);
  pt G, B; pt_from_affine(&G, &BS_GX_M, &BS_GY_M); pt_from_affine(&B, &BS_BX_M, &BS_BY_M);
  uint8_t root[32];
  memset(root, 0, 32);
  if (n) {
    const uint8_t* arrs[7] = {h, gamma, proof160, proof160 + 32, proof160 + 64, proof160 + 96, proof160 + 128};
    const uint32_t widths[7] = {32, 32, 32, 32, 32, 32, 32}, strides[7] = {32, 32, 160, 160, 160, 160, 160};
    oracle_batch_digest(n, index0, 7, arrs, widths, strides, ad, NULL, (uint32_t)ad_len, root);
  }
  for (size_t i = 0; i < n; ++i) {
    const uint8_t* pr = proof160 + 160 * i;
    uint64_t s[4], sb[4], c[4];
    load_le(s, pr + 96); load_le(sb, pr + 128);
    fp x, y; pt H, Gm, PC, R, Ok;
    int ok = cmp4(s, FR.m) < 0 && cmp4(sb, FR.m) < 0;
    ok = ok && point_decode_chk(&x, &y, h + 32 * i, 2);
    if (ok) pt_from_affine(&H, &x, &y);
    ok = ok && point_decode_chk(&x, &y, gamma + 32 * i, 4);
    if (ok) pt_from_affine(&Gm, &x, &y);
    ok = ok && point_decode_chk(&x, &y, pr, 8);
    if (ok) pt_from_affine(&PC, &x, &y);
    ok = ok && point_decode_chk(&x, &y, pr + 32, 8);
    if (ok) pt_from_affine(&R, &x, &y);
    ok = ok && point_decode_chk(&x, &y, pr + 64, 8);
    if (ok) pt_from_affine(&Ok, &x, &y);
    status[i] = ok ? 0 : 2;
    if (!ok) continue;
    uint8_t pts[5][32];
    memcpy(pts[0], pr, 32); memcpy(pts[1], h + 32 * i, 32); memcpy(pts[2], gamma + 32 * i, 32);
    memcpy(pts[3], pr + 32, 32); memcpy(pts[4], pr + 64, 32);
    challenge(c, pts, ad, ad_len);
    sha512_ctx hc; uint8_t dg[64], idx[8];
    uint64_t gi = index0 + i;
    for (int k = 0; k < 8; ++k) idx[k] = (uint8_t)(gi >> (8 * k));
    sha512_init(&hc); sha512_update(&hc, "vrfhip-rlc-v2", 13); sha512_update(&hc, seed, 32);
    sha512_update(&hc, root, 32); sha512_update(&hc, idx, 8); sha512_final(&hc, dg);
    uint64_t z[4] = {0, 0, 0, 0}, zp[4] = {0, 0, 0, 0};
    for (int k = 0; k < 8; ++k) {
      z[0] |= (uint64_t)dg[k] << (8 * k); z[1] |= (uint64_t)dg[8 + k] << (8 * k);
      zp[0] |= (uint64_t)dg[16 + k] << (8 * k); zp[1] |= (uint64_t)dg[24 + k] << (8 * k);
    }
    z[0] = (z[0] & ~(uint64_t)7) | 1; zp[0] = (zp[0] & ~(uint64_t)7) | 1;   /* weights = 1 (mod 8) */
    pt t0, t1, d1, d2, n0;
    pt_mul(&t0, &H, s); pt_mul(&t1, &Gm, c); pt_neg(&n0, &t1); pt_add(&d1, &t0, &n0);
    pt_neg(&n0, &Ok); pt_add(&d1, &d1, &n0);
    pt_mul(&t0, &G, s); pt_mul(&t1, &B, sb); pt_add(&d2, &t0, &t1);
    pt_mul(&t1, &PC, c); pt_neg(&n0, &t1); pt_add(&d2, &d2, &n0);
    pt_neg(&n0, &R); pt_add(&d2, &d2, &n0);
    pt_mul(&t0, &d1, z); pt_add(&acc, &acc, &t0);
    pt_mul(&t0, &d2, zp); pt_add(&acc, &acc, &t0);
  }
  return pt_is_identity(&acc) ? 0 : 1;
}
/* [ark_ec VariableBaseMSM::msm] naive reference: sum_i k_i * P_i, affine x||y inputs (64 B LE each).
 * Returns 0 / 2 (coordinate >= q, off-curve point or scalar >= r). */
int oracle_msm(size_t n, const uint8_t* xy, const uint8_t* scalars, uint8_t out_enc[32], uint8_t out_xy[64]) {
  ensure_init();
  pt acc; pt_identity(&acc);
  for (size_t i = 0; i < n; ++i) {
    uint64_t xi[4], yi[4], k[4];
    load_le(xi, xy + 64 * i); load_le(yi, xy + 64 * i + 32); load_le(k, scalars + 32 * i);
    if (cmp4(xi, FQ.m) >= 0 || cmp4(yi, FQ.m) >= 0 || cmp4(k, FR.m) >= 0) return 2;
    fp x, y, x2, y2, l, r1, one; q_from_int(&x, xi); q_from_int(&y, yi); q_one(&one);
    q_sqr(&x2, &x); q_sqr(&y2, &y);
    q_mul(&l, &BS_A_M, &x2); q_add(&l, &l, &y2);
    q_mul(&r1, &x2, &y2); q_mul(&r1, &r1, &BS_D_M); q_add(&r1, &r1, &one);
    if (!q_eq(&l, &r1)) return 2;
    pt p, kp; pt_from_affine(&p, &x, &y); pt_mul(&kp, &p, k); pt_add(&acc, &acc, &kp);
  }
  fp x, y; pt_to_affine(&x, &y, &acc);
  point_encode(out_enc, &x, &y);
  if (out_xy) { uint64_t t[4]; q_to_int(t, &x); store_le(out_xy, t); q_to_int(t, &y); store_le(out_xy + 32, t); }
  return 0;
}
/* test hooks for the field layer */
void oracle_fq_mul(const uint8_t a[32], const uint8_t b[32], uint8_t r[32]) {
  ensure_init();
  uint64_t x[4], y[4]; load_le(x, a); load_le(y, b);
  fp xm, ym; q_from_int(&xm, x); q_from_int(&ym, y); q_mul(&xm, &xm, &ym); q_to_int(x, &xm); store_le(r, x);
}
void oracle_sha512(const uint8_t* m, size_t n, uint8_t out[64]) {
  sha512_ctx c; sha512_init(&c); sha512_update(&c, m, n); sha512_final(&c, out);
}

/* `suites::bandersnatch_sw` on its own curve model: shares this file's field, hash and thread helpers */
#include "oracle_bsw.c"
