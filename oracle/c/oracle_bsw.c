/* oracle_bsw.c -- CPU restatement (TEST INFRASTRUCTURE ONLY) of `suites::bandersnatch_sw` (/root/reference src/lib.rs:14;
 * upstream "Bandersnatch_SW_SHA-512_TAI"), the native twin of oracle/bsw_oracle.py.  Compiled as part of oracle_vrf.c
 * (#included at its end: it uses that file's Montgomery field, SHA-512 and thread helpers).
 *
 * Parity status: UNPINNED (no vector of this suite on this machine) -- see oracle/bsw_oracle.py for what it rests on.  Like
 * the Python oracle, and unlike the device code, the arithmetic is done ON the short-Weierstrass curve y^2 = x^3 + a' x + b'
 * (Jacobian chord-and-tangent, MSB-first double-and-add, r * P = O subgroup test: arkworks' shape); the device runs the
 * twisted-Edwards model and maps at the codec.  tests/test_bandersnatch_sw.py holds the two oracles against each other.
 *
 * Wire format (ArkworksCodec over short_weierstrass::Affine, as recalled): 33 bytes = x little-endian || flag byte (0x80: y is the
 * larger of {y, q - y}; 0x40: infinity, x = 0); both flags / x >= q = error; the low six flag-byte bits and, for infinity, x are
 * not looked at; hashes take the canonical re-encoding. */

typedef struct { fp X, Y, Z; } jpt;                  /* Jacobian (X/Z^2, Y/Z^3); Z = 0: the point at infinity */
static fp BW_A, BW_B, BW_GX, BW_GY, BW_BX, BW_BY;
static pthread_once_t bw_once = PTHREAD_ONCE_INIT;
static const char BW_SUITE_ID[] = "Bandersnatch_SW_SHA-512_TAI";
#define BW_ID_LEN 27
#define BQF (&BF_BLS.f)
#define BRF (&SUITE_BS.fr)

static void bmul(fp* r, const fp* a, const fp* b) { f_mul(BQF, r->v, a->v, b->v); }
static void bsqr(fp* r, const fp* a) { f_mul(BQF, r->v, a->v, a->v); }
static void badd(fp* r, const fp* a, const fp* b) { f_add(BQF, r->v, a->v, b->v); }
static void bsub(fp* r, const fp* a, const fp* b) { f_sub(BQF, r->v, a->v, b->v); }
static void binv(fp* r, const fp* a) { f_inv(BQF, r->v, a->v); }
static void bneg(fp* r, const fp* a) { fp z; memset(&z, 0, sizeof z); f_sub(BQF, r->v, z.v, a->v); }
static void bu64(fp* r, uint64_t x) { uint64_t a[4] = {x, 0, 0, 0}; f_to_mont(BQF, r->v, a); }
static int bisz(const fp* a) { return is_zero4(a->v); }
static int beq(const fp* a, const fp* b) { return cmp4(a->v, b->v) == 0; }

/* te_sw_map: TE (x, y) -> Montgomery (u, v) = ((1 + y)/(1 - y), u / x) -> SW ((u + A/3)/B, v/B) */
static void bw_from_te(fp* sx, fp* sy, const fp* x, const fp* y, const fp* A3, const fp* Binv) {
  fp one, n, m, u, v, t;
  bu64(&one, 1); badd(&n, &one, y); bsub(&m, &one, y);
  binv(&t, &m); bmul(&u, &n, &t);
  binv(&t, x); bmul(&v, &u, &t);
  badd(&t, &u, A3); bmul(sx, &t, Binv); bmul(sy, &v, Binv);
}
static void bw_do_init(void) {
  ensure_init();
  /* A = 2 (a + d)/(a - d), B = 4/(a - d); a' = (3 - A^2)/(3 B^2), b' = (2 A^3 - 9 A)/(27 B^3) */
  fp amd, apd, t, A, B, A2, A3v, B2, B3, c3, c9, c27, c2, c4, third, Binv;
  bsub(&amd, &SUITE_BS.a, &SUITE_BS.d); badd(&apd, &SUITE_BS.a, &SUITE_BS.d);
  binv(&t, &amd);
  bu64(&c2, 2); bu64(&c3, 3); bu64(&c4, 4); bu64(&c9, 9); bu64(&c27, 27);
  bmul(&A, &apd, &t); bmul(&A, &A, &c2); bmul(&B, &c4, &t);
  bsqr(&A2, &A); bmul(&A3v, &A2, &A); bsqr(&B2, &B); bmul(&B3, &B2, &B);
  fp num, den;
  bsub(&num, &c3, &A2); bmul(&den, &c3, &B2); binv(&den, &den); bmul(&BW_A, &num, &den);
  bmul(&num, &c2, &A3v); bmul(&t, &c9, &A); bsub(&num, &num, &t); bmul(&den, &c27, &B3); binv(&den, &den); bmul(&BW_B, &num, &den);
  binv(&third, &c3); bmul(&third, &third, &A); binv(&Binv, &B);
  bw_from_te(&BW_GX, &BW_GY, &SUITE_BS.gx, &SUITE_BS.gy, &third, &Binv);
  bw_from_te(&BW_BX, &BW_BY, &SUITE_BS.bx, &SUITE_BS.by, &third, &Binv);
}
static void bw_init(void) { pthread_once(&bw_once, bw_do_init); }

/* ---- group law  [ref src/lib.rs:15 `AffinePoint`: ark_ec::short_weierstrass] ---- */
static void j_inf(jpt* p) { bu64(&p->X, 1); bu64(&p->Y, 1); memset(&p->Z, 0, sizeof p->Z); }
static void j_aff(jpt* p, const fp* x, const fp* y) { p->X = *x; p->Y = *y; bu64(&p->Z, 1); }
static void j_dbl(jpt* r, const jpt* p) {
  if (bisz(&p->Z) || bisz(&p->Y)) { j_inf(r); return; }
  fp YY, S, M, Z2, Z4, t, X3, Y3, Z3, c;
  bsqr(&YY, &p->Y); bmul(&S, &p->X, &YY); badd(&S, &S, &S); badd(&S, &S, &S);                 /* 4 X Y^2 */
  bsqr(&M, &p->X); badd(&t, &M, &M); badd(&M, &t, &M);                                          /* 3 X^2 */
  bsqr(&Z2, &p->Z); bsqr(&Z4, &Z2); bmul(&t, &BW_A, &Z4); badd(&M, &M, &t);
  bsqr(&X3, &M); bsub(&X3, &X3, &S); bsub(&X3, &X3, &S);
  bsub(&t, &S, &X3); bmul(&Y3, &M, &t); bsqr(&c, &YY); badd(&c, &c, &c); badd(&c, &c, &c); badd(&c, &c, &c); bsub(&Y3, &Y3, &c);
  bmul(&Z3, &p->Y, &p->Z); badd(&Z3, &Z3, &Z3);
  r->X = X3; r->Y = Y3; r->Z = Z3;
}
static void j_add(jpt* r, const jpt* p, const jpt* q) {
  if (bisz(&p->Z)) { *r = *q; return; }
  if (bisz(&q->Z)) { *r = *p; return; }
  fp Z1Z1, Z2Z2, U1, U2, S1, S2, H, R, HH, HHH, V, t, X3, Y3, Z3;
  bsqr(&Z1Z1, &p->Z); bsqr(&Z2Z2, &q->Z);
  bmul(&U1, &p->X, &Z2Z2); bmul(&U2, &q->X, &Z1Z1);
  bmul(&S1, &p->Y, &q->Z); bmul(&S1, &S1, &Z2Z2); bmul(&S2, &q->Y, &p->Z); bmul(&S2, &S2, &Z1Z1);
  if (beq(&U1, &U2)) { if (beq(&S1, &S2)) j_dbl(r, p); else j_inf(r); return; }
  bsub(&H, &U2, &U1); bsub(&R, &S2, &S1); bsqr(&HH, &H); bmul(&HHH, &H, &HH); bmul(&V, &U1, &HH);
  bsqr(&X3, &R); bsub(&X3, &X3, &HHH); bsub(&X3, &X3, &V); bsub(&X3, &X3, &V);
  bsub(&t, &V, &X3); bmul(&Y3, &R, &t); bmul(&t, &S1, &HHH); bsub(&Y3, &Y3, &t);
  bmul(&Z3, &H, &p->Z); bmul(&Z3, &Z3, &q->Z);
  r->X = X3; r->Y = Y3; r->Z = Z3;
}
static void j_neg(jpt* r, const jpt* p) { *r = *p; bneg(&r->Y, &p->Y); }
static void j_mul(jpt* r, const jpt* p, const uint64_t k[4]) {       /* MSB-first double-and-add */
  jpt acc; j_inf(&acc);
  for (int i = 255; i >= 0; --i) {
    j_dbl(&acc, &acc);
    if ((k[i >> 6] >> (i & 63)) & 1) j_add(&acc, &acc, p);
  }
  *r = acc;
}
/* returns 0 for the point at infinity */
static int j_affine(fp* x, fp* y, const jpt* p) {
  if (bisz(&p->Z)) return 0;
  fp zi, z2, z3;
  binv(&zi, &p->Z); bsqr(&z2, &zi); bmul(&z3, &z2, &zi);
  bmul(x, &p->X, &z2); bmul(y, &p->Y, &z3);
  return 1;
}
static int j_eq(const jpt* a, const jpt* b) {
  fp ax, ay, bx, by;
  const int fa = j_affine(&ax, &ay, a), fb = j_affine(&bx, &by, b);
  if (!fa || !fb) return fa == fb;
  return beq(&ax, &bx) && beq(&ay, &by);
}
static int j_in_subgroup(const jpt* p) { jpt t; j_mul(&t, p, BRF->m); return bisz(&t.Z); }

/* ---- codec ---- */
static void bw_encode(uint8_t out[33], const jpt* p) {
  fp x, y;
  memset(out, 0, 33);
  if (!j_affine(&x, &y, p)) { out[32] = 0x40; return; }
  uint64_t xi[4], yi[4], ny[4]; fp n;
  f_from_mont(BQF, xi, x.v); f_from_mont(BQF, yi, y.v); bneg(&n, &y); f_from_mont(BQF, ny, n.v);
  store_le(out, xi);
  out[32] = cmp4(yi, ny) > 0 ? 0x80 : 0x00;
}
static int bw_sqrt(fp* r, const fp* a) {             /* Tonelli-Shanks over the BLS12-381 scalar field; 0: no root */
  if (bisz(a)) { memset(r, 0, sizeof *r); return 1; }
  uint64_t e[4], one[4] = {1, 0, 0, 0};
  sub4(e, BQF->m, one);
  for (int i = 0; i < 4; ++i) e[i] = (e[i] >> 1) | (i < 3 ? e[i + 1] << 63 : 0);
  fp t; f_pow(BQF, t.v, a->v, e);
  if (cmp4(t.v, BQF->one) != 0) return 0;
  fp c = BF_BLS.ts_c, tt, rr;
  f_pow(BQF, tt.v, a->v, BF_BLS.ts_t);
  f_pow(BQF, rr.v, a->v, BF_BLS.ts_e);
  int m = BF_BLS.s;
  while (cmp4(tt.v, BQF->one) != 0) {
    int i = 0; fp t2 = tt;
    while (cmp4(t2.v, BQF->one) != 0) { bsqr(&t2, &t2); ++i; }
    fp b = c;
    for (int j = 0; j < m - i - 1; ++j) bsqr(&b, &b);
    m = i; bsqr(&c, &b); bmul(&tt, &tt, &c); bmul(&rr, &rr, &b);
  }
  *r = rr;
  return 1;
}
/* `deserialize_compressed_unchecked`: 1 = decodes (on the curve, any subgroup) */
static int bw_decode(jpt* p, const uint8_t in[33]) {
  const uint8_t fl = in[32] & 0xC0;
  uint64_t xi[4]; load_le(xi, in);
  if (fl == 0xC0 || cmp4(xi, BQF->m) >= 0) return 0;
  if (fl == 0x40) { j_inf(p); return 1; }
  fp x, rhs, t, y, ny;
  f_to_mont(BQF, x.v, xi);
  bsqr(&t, &x); badd(&t, &t, &BW_A); bmul(&rhs, &t, &x); badd(&rhs, &rhs, &BW_B);
  if (!bw_sqrt(&y, &rhs)) return 0;
  bneg(&ny, &y);
  uint64_t yi[4], nyi[4];
  f_from_mont(BQF, yi, y.v); f_from_mont(BQF, nyi, ny.v);
  const int y_is_larger = cmp4(yi, nyi) > 0;
  if ((fl == 0x80) != y_is_larger) y = ny;
  j_aff(p, &x, &y);
  return 1;
}
static int bw_decode_chk(jpt* p, const uint8_t in[33], int bit) {
  if (!bw_decode(p, in)) return 0;
  if ((g_check_mask & bit) && !j_in_subgroup(p)) return 0;
  return 1;
}
static void bw_canonical(uint8_t out[33], const uint8_t in[33]) {
  memcpy(out, in, 33);
  out[32] &= 0xC0;
  if (out[32] == 0x40) memset(out, 0, 32);
}

/* ---- scalars mod r, hashes ---- */
static void bw_r_wide(uint64_t out[4], const uint8_t* b, size_t n, int big_endian) {
  uint64_t acc[4] = {0, 0, 0, 0}, c256[4] = {256, 0, 0, 0}, m256[4];
  f_to_mont(BRF, m256, c256);
  for (size_t i = 0; i < n; ++i) {
    uint8_t byte = big_endian ? b[i] : b[n - 1 - i];
    uint64_t d[4] = {byte, 0, 0, 0}, dm[4];
    f_to_mont(BRF, dm, d);
    f_mul(BRF, acc, acc, m256);
    f_add(BRF, acc, acc, dm);
  }
  f_from_mont(BRF, out, acc);
}
static void bw_r_muladd(uint64_t out[4], const uint64_t a[4], const uint64_t b[4], const uint64_t c[4]) {
  uint64_t am[4], bm[4], cm[4], t[4];
  f_to_mont(BRF, am, a); f_to_mont(BRF, bm, b); f_to_mont(BRF, cm, c);
  f_mul(BRF, t, am, bm); f_add(BRF, t, t, cm); f_from_mont(BRF, out, t);
}
static void bw_nonce(uint64_t k[4], const uint8_t sk_le[32], const uint8_t h_enc[33]) {
  uint8_t h1[64], h2[64]; sha512_ctx c;
  sha512_init(&c); sha512_update(&c, sk_le, 32); sha512_final(&c, h1);
  sha512_init(&c); sha512_update(&c, h1 + 32, 32); sha512_update(&c, h_enc, 33); sha512_final(&c, h2);
  bw_r_wide(k, h2, 64, 0);
}
static void bw_challenge(uint64_t c_out[4], const uint8_t pts[5][33], const uint8_t* ad, size_t ad_len) {
  uint8_t h[64], two = 2, zero = 0; sha512_ctx c;
  sha512_init(&c); sha512_update(&c, BW_SUITE_ID, BW_ID_LEN); sha512_update(&c, &two, 1);
  for (int i = 0; i < 5; ++i) { uint8_t e[33]; bw_canonical(e, pts[i]); sha512_update(&c, e, 33); }
  sha512_update(&c, ad, ad_len); sha512_update(&c, &zero, 1); sha512_final(&c, h);
  bw_r_wide(c_out, h, 32, 1);
}
static void bw_blinding(uint64_t b[4], const uint8_t sk_le[32], const uint8_t h_enc[33], const uint8_t* ad, size_t ad_len) {
  uint8_t h[64], cc = 0xCC, zero = 0; sha512_ctx c;
  sha512_init(&c); sha512_update(&c, BW_SUITE_ID, BW_ID_LEN); sha512_update(&c, &cc, 1); sha512_update(&c, sk_le, 32);
  sha512_update(&c, h_enc, 33); sha512_update(&c, ad, ad_len); sha512_update(&c, &zero, 1); sha512_final(&c, h);
  bw_r_wide(b, h, 64, 1);
}
/* [ref src/lib.rs:14 `utils::hash_to_curve_tai_rfc_9381`]: 1 = found */
static int bw_hash_to_curve(jpt* out, const uint8_t* msg, size_t len) {
  for (int ctr = 0; ctr < 256; ++ctr) {
    uint8_t h[64], one = 1, cb = (uint8_t)ctr, zero = 0; sha512_ctx c;
    sha512_init(&c); sha512_update(&c, BW_SUITE_ID, BW_ID_LEN); sha512_update(&c, &one, 1);
    sha512_update(&c, msg, len); sha512_update(&c, &cb, 1); sha512_update(&c, &zero, 1); sha512_final(&c, h);
    jpt p;
    if (!bw_decode(&p, h)) continue;
    j_dbl(&p, &p); j_dbl(&p, &p);                     /* cofactor 4 */
    if (bisz(&p.Z)) continue;
    *out = p;
    return 1;
  }
  j_inf(out);
  return 0;
}

/* ---- entry points (ctypes: oracle/c_oracle.py) ---- */
int oracle_bsw_constants(uint8_t ab[64], uint8_t g[64], uint8_t bb[64]) {
  bw_init();
  uint64_t t[4];
  f_from_mont(BQF, t, BW_A.v); store_le(ab, t); f_from_mont(BQF, t, BW_B.v); store_le(ab + 32, t);
  f_from_mont(BQF, t, BW_GX.v); store_le(g, t); f_from_mont(BQF, t, BW_GY.v); store_le(g + 32, t);
  f_from_mont(BQF, t, BW_BX.v); store_le(bb, t); f_from_mont(BQF, t, BW_BY.v); store_le(bb + 32, t);
  return 0;
}
int oracle_bsw_secret_public(const uint8_t* seed, size_t len, uint8_t sk_out[32], uint8_t pk_out[33]) {
  bw_init();
  uint8_t h[64]; sha512_ctx c; uint64_t sk[4];
  sha512_init(&c); sha512_update(&c, seed, len); sha512_final(&c, h);
  bw_r_wide(sk, h, 64, 0); store_le(sk_out, sk);
  if (pk_out) { jpt G, P; j_aff(&G, &BW_GX, &BW_GY); j_mul(&P, &G, sk); bw_encode(pk_out, &P); }
  return 0;
}
int oracle_bsw_hash_to_curve(const uint8_t* msg, size_t len, uint8_t out[33]) {
  bw_init();
  jpt p; const int ok = bw_hash_to_curve(&p, msg, len);
  bw_encode(out, &p);
  return ok ? 0 : 2;
}
int oracle_bsw_output_hash(const uint8_t gamma[33], uint8_t out[64]) {
  bw_init();
  uint8_t e[33], three = 3, zero = 0; sha512_ctx c;
  bw_canonical(e, gamma);
  sha512_init(&c); sha512_update(&c, BW_SUITE_ID, BW_ID_LEN); sha512_update(&c, &three, 1); sha512_update(&c, e, 33);
  sha512_update(&c, &zero, 1); sha512_final(&c, out);
  return 0;
}
/* 0 = valid point of the prime-order subgroup (subgroup != 0) / of the curve; xy_out (nullable): x || y little-endian */
int oracle_bsw_point_decode(const uint8_t in[33], int subgroup, uint8_t xy_out[64]) {
  bw_init();
  jpt p;
  if (!bw_decode(&p, in) || (subgroup && !j_in_subgroup(&p))) return 2;
  if (xy_out) {
    fp x, y; uint64_t t[4];
    memset(xy_out, 0, 64);
    if (j_affine(&x, &y, &p)) { f_from_mont(BQF, t, x.v); store_le(xy_out, t); f_from_mont(BQF, t, y.v); store_le(xy_out + 32, t); }
  }
  return 0;
}
/* [ref src/lib.rs:14 `ietf::Prover::prove`]; h_given (nullable): the input as a 33-byte point instead of the message */
int oracle_bsw_ietf_prove(const uint8_t sk_le[32], const uint8_t* msg, size_t msg_len, const uint8_t* h_given, const uint8_t* ad,
                          size_t ad_len, uint8_t gamma_out[33], uint8_t c_out[32], uint8_t s_out[32], uint8_t* pk_out,
                          uint8_t* h_out) {
  bw_init();
  uint64_t sk[4]; load_le(sk, sk_le);
  if (cmp4(sk, BRF->m) >= 0) return 2;
  jpt H, G, P;
  uint8_t pts[5][33];
  if (h_given) { if (!bw_decode_chk(&H, h_given, 2)) return 2; bw_canonical(pts[1], h_given); }
  else { bw_hash_to_curve(&H, msg, msg_len); bw_encode(pts[1], &H); }
  j_aff(&G, &BW_GX, &BW_GY);
  uint64_t k[4], c[4], s[4];
  bw_nonce(k, sk_le, pts[1]);
  j_mul(&P, &G, sk); bw_encode(pts[0], &P);
  j_mul(&P, &H, sk); bw_encode(pts[2], &P);
  j_mul(&P, &G, k); bw_encode(pts[3], &P);
  j_mul(&P, &H, k); bw_encode(pts[4], &P);
  bw_challenge(c, pts, ad, ad_len);
  bw_r_muladd(s, c, sk, k);
  memcpy(gamma_out, pts[2], 33); store_le(c_out, c); store_le(s_out, s);
  if (pk_out) memcpy(pk_out, pts[0], 33);
  if (h_out) memcpy(h_out, pts[1], 33);
  return 0;
}
/* [ref src/lib.rs:14 `ietf::Verifier::verify`] 0 verified, 1 VerificationFailure, 2 InvalidData */
int oracle_bsw_ietf_verify(const uint8_t pk[33], const uint8_t h[33], const uint8_t gamma[33], const uint8_t c_le[32],
                           const uint8_t s_le[32], const uint8_t* ad, size_t ad_len) {
  bw_init();
  uint64_t s[4], c[4], c2[4];
  load_le(s, s_le);
  if (cmp4(s, BRF->m) >= 0) return 2;
  bw_r_wide(c, c_le, 32, 0);                          /* `Proof::c` mod r */
  jpt Y, H, Gm, G, sG, cY, sH, cG, n, U, V;
  if (!bw_decode_chk(&Y, pk, 1) || !bw_decode_chk(&H, h, 2) || !bw_decode_chk(&Gm, gamma, 4)) return 2;
  j_aff(&G, &BW_GX, &BW_GY);
  j_mul(&sG, &G, s); j_mul(&cY, &Y, c); j_neg(&n, &cY); j_add(&U, &sG, &n);
  j_mul(&sH, &H, s); j_mul(&cG, &Gm, c); j_neg(&n, &cG); j_add(&V, &sH, &n);
  uint8_t pts[5][33];
  memcpy(pts[0], pk, 33); memcpy(pts[1], h, 33); memcpy(pts[2], gamma, 33);
  bw_encode(pts[3], &U); bw_encode(pts[4], &V);
  bw_challenge(c2, pts, ad, ad_len);
  return cmp4(c, c2) == 0 ? 0 : 1;
}
/* [ref src/lib.rs:14 `pedersen::Prover::prove`]; proof163 = pk_com | R | Ok (33 bytes each) | s | sb */
int oracle_bsw_pedersen_prove(const uint8_t sk_le[32], const uint8_t* msg, size_t msg_len, const uint8_t* h_given, const uint8_t* ad,
                              size_t ad_len, uint8_t gamma_out[33], uint8_t proof163[163], uint8_t* blinding_out, uint8_t* h_out) {
  bw_init();
  uint64_t sk[4]; load_le(sk, sk_le);
  if (cmp4(sk, BRF->m) >= 0) return 2;
  jpt H, G, B, t0, t1, P;
  uint8_t pts[5][33];
  if (h_given) { if (!bw_decode_chk(&H, h_given, 2)) return 2; bw_canonical(pts[1], h_given); }
  else { bw_hash_to_curve(&H, msg, msg_len); bw_encode(pts[1], &H); }
  j_aff(&G, &BW_GX, &BW_GY); j_aff(&B, &BW_BX, &BW_BY);
  uint64_t b[4], k[4], kb[4], c[4], s[4], sb[4]; uint8_t b_le[32];
  bw_blinding(b, sk_le, pts[1], ad, ad_len); store_le(b_le, b);
  bw_nonce(k, sk_le, pts[1]); bw_nonce(kb, b_le, pts[1]);
  j_mul(&P, &H, sk); bw_encode(pts[2], &P);
  j_mul(&t0, &G, sk); j_mul(&t1, &B, b); j_add(&P, &t0, &t1); bw_encode(pts[0], &P);
  j_mul(&t0, &G, k); j_mul(&t1, &B, kb); j_add(&P, &t0, &t1); bw_encode(pts[3], &P);
  j_mul(&P, &H, k); bw_encode(pts[4], &P);
  bw_challenge(c, pts, ad, ad_len);
  bw_r_muladd(s, c, sk, k); bw_r_muladd(sb, c, b, kb);
  memcpy(gamma_out, pts[2], 33);
  memcpy(proof163, pts[0], 33); memcpy(proof163 + 33, pts[3], 33); memcpy(proof163 + 66, pts[4], 33);
  store_le(proof163 + 99, s); store_le(proof163 + 131, sb);
  if (blinding_out) memcpy(blinding_out, b_le, 32);
  if (h_out) memcpy(h_out, pts[1], 33);
  return 0;
}
/* [ref src/lib.rs:14 `pedersen::Verifier::verify`] */
int oracle_bsw_pedersen_verify(const uint8_t h[33], const uint8_t gamma[33], const uint8_t proof163[163], const uint8_t* ad,
                               size_t ad_len) {
  bw_init();
  uint64_t s[4], sb[4], c[4];
  load_le(s, proof163 + 99); load_le(sb, proof163 + 131);
  if (cmp4(s, BRF->m) >= 0 || cmp4(sb, BRF->m) >= 0) return 2;
  jpt H, Gm, PC, R, Ok, G, B, l, r1, t0, t1;
  if (!bw_decode_chk(&H, h, 2) || !bw_decode_chk(&Gm, gamma, 4) || !bw_decode_chk(&PC, proof163, 8) ||
      !bw_decode_chk(&R, proof163 + 33, 8) || !bw_decode_chk(&Ok, proof163 + 66, 8)) return 2;
  uint8_t pts[5][33];
  memcpy(pts[0], proof163, 33); memcpy(pts[1], h, 33); memcpy(pts[2], gamma, 33);
  memcpy(pts[3], proof163 + 33, 33); memcpy(pts[4], proof163 + 66, 33);
  bw_challenge(c, pts, ad, ad_len);
  j_aff(&G, &BW_GX, &BW_GY); j_aff(&B, &BW_BX, &BW_BY);
  j_mul(&t0, &Gm, c); j_add(&l, &Ok, &t0); j_mul(&r1, &H, s);                 /* Ok + c Gamma == s H */
  if (!j_eq(&l, &r1)) return 1;
  j_mul(&t0, &PC, c); j_add(&l, &R, &t0);                                      /* R + c pk_com == s G + sb B */
  j_mul(&t0, &G, s); j_mul(&t1, &B, sb); j_add(&r1, &t0, &t1);
  return j_eq(&l, &r1) ? 0 : 1;
}

/* ---- batch drivers: static partition over pthreads ---- */
typedef struct {
  int kind; size_t lo, hi;
  const uint8_t *a0, *a1, *a2, *a3, *a4; const uint8_t* ad; size_t ad_len, msg_len;
  uint8_t *o0, *o1, *o2, *o3, *o4, *st;
} bw_job;
static void* bw_run(void* arg) {
  bw_job* j = (bw_job*)arg;
  for (size_t i = j->lo; i < j->hi; ++i) {
    int rc;
    if (j->kind == 0)
      rc = oracle_bsw_ietf_verify(j->a0 + 33 * i, j->a1 + 33 * i, j->a2 + 33 * i, j->a3 + 32 * i, j->a4 + 32 * i, j->ad, j->ad_len);
    else if (j->kind == 1)
      rc = oracle_bsw_ietf_prove(j->a0 + 32 * i, j->a1 ? j->a1 + j->msg_len * i : NULL, j->msg_len, j->a2 ? j->a2 + 33 * i : NULL,
                                 j->ad, j->ad_len, j->o0 + 33 * i, j->o1 + 32 * i, j->o2 + 32 * i, j->o3 ? j->o3 + 33 * i : NULL,
                                 j->o4 ? j->o4 + 33 * i : NULL);
    else if (j->kind == 2)
      rc = oracle_bsw_pedersen_verify(j->a0 + 33 * i, j->a1 + 33 * i, j->a2 + 163 * i, j->ad, j->ad_len);
    else
      rc = oracle_bsw_pedersen_prove(j->a0 + 32 * i, j->a1 ? j->a1 + j->msg_len * i : NULL, j->msg_len, j->a2 ? j->a2 + 33 * i : NULL,
                                     j->ad, j->ad_len, j->o0 + 33 * i, j->o1 + 163 * i, j->o2 ? j->o2 + 32 * i : NULL,
                                     j->o4 ? j->o4 + 33 * i : NULL);
    if (j->st) j->st[i] = (uint8_t)rc;
  }
  return NULL;
}
static void bw_batch(bw_job base, size_t n, int threads) {
  bw_init();
  if (threads <= 1 || n < 2) { base.lo = 0; base.hi = n; bw_run(&base); return; }
  if ((size_t)threads > n) threads = (int)n;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * threads);
  bw_job* jobs = (bw_job*)malloc(sizeof(bw_job) * threads);
  for (int t = 0; t < threads; ++t) {
    jobs[t] = base; jobs[t].lo = n * t / threads; jobs[t].hi = n * (t + 1) / threads;
    pthread_create(&th[t], NULL, bw_run, &jobs[t]);
  }
  for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
  free(th); free(jobs);
}
void oracle_bsw_ietf_verify_batch(size_t n, const uint8_t* pk, const uint8_t* h, const uint8_t* gamma, const uint8_t* c,
                                  const uint8_t* s, const uint8_t* ad, size_t ad_len, uint8_t* status, int threads) {
  bw_job j; memset(&j, 0, sizeof j);
  j.kind = 0; j.a0 = pk; j.a1 = h; j.a2 = gamma; j.a3 = c; j.a4 = s; j.ad = ad; j.ad_len = ad_len; j.st = status;
  bw_batch(j, n, threads);
}
void oracle_bsw_ietf_prove_batch(size_t n, const uint8_t* sk, const uint8_t* msg, size_t msg_len, const uint8_t* h_given,
                                 const uint8_t* ad, size_t ad_len, uint8_t* gamma, uint8_t* c, uint8_t* s, uint8_t* pk_out,
                                 uint8_t* h_out, uint8_t* status, int threads) {
  bw_job j; memset(&j, 0, sizeof j);
  j.kind = 1; j.a0 = sk; j.a1 = msg; j.msg_len = msg_len; j.a2 = h_given; j.ad = ad; j.ad_len = ad_len;
  j.o0 = gamma; j.o1 = c; j.o2 = s; j.o3 = pk_out; j.o4 = h_out; j.st = status;
  bw_batch(j, n, threads);
}
void oracle_bsw_pedersen_verify_batch(size_t n, const uint8_t* h, const uint8_t* gamma, const uint8_t* proof163, const uint8_t* ad,
                                      size_t ad_len, uint8_t* status, int threads) {
  bw_job j; memset(&j, 0, sizeof j);
  j.kind = 2; j.a0 = h; j.a1 = gamma; j.a2 = proof163; j.ad = ad; j.ad_len = ad_len; j.st = status;
  bw_batch(j, n, threads);
}
void oracle_bsw_pedersen_prove_batch(size_t n, const uint8_t* sk, const uint8_t* msg, size_t msg_len, const uint8_t* h_given,
                                     const uint8_t* ad, size_t ad_len, uint8_t* gamma, uint8_t* proof163, uint8_t* blinding_out,
                                     uint8_t* h_out, uint8_t* status, int threads) {
  bw_job j; memset(&j, 0, sizeof j);
  j.kind = 3; j.a0 = sk; j.a1 = msg; j.msg_len = msg_len; j.a2 = h_given; j.ad = ad; j.ad_len = ad_len;
  j.o0 = gamma; j.o1 = proof163; j.o2 = blinding_out; j.o4 = h_out; j.st = status;
  bw_batch(j, n, threads);
}
