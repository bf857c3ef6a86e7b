// p256_core.cuh -- the secp256r1 suite ("P256_SHA256_TAI", RFC 9381 suite 0x01; upstream `suites::secp256r1`,
// /root/reference src/lib.rs:14): Sec1 codec, try-and-increment hash-to-curve, RFC 6979 nonce, challenge, output hash and
// the per-item prove / verify steps of the IETF and Pedersen schemes, on top of sw.cuh (group law) and sha256.cuh.
//
// Wire format of this suite at the C ABI (include/vrfhip.h): points are 33-byte Sec1 compressed strings (`Sec1Codec`),
// scalars 32-byte BIG-endian integers (`int_to_string` = I2OSP); the challenge is 16 bytes on the wire, carried here in
// a 32-byte big-endian field like every scalar.
//
// Pinned by RFC 9381 Appendix B.1 (tests/golden/rfc9381_p256_sha256_tai.json): the three published examples' H, k, U, V,
// pi and beta come out of these functions bit for bit (tests/test_secp256r1.py, on the host build and through the GPU).
#pragma once
#include "sha256.cuh"
#include "sw.cuh"

VRF_NS_BEGIN

constexpr int SEC1_LEN = 33;

VRF_HD bool u256_ge_q(const uint32_t a[8]) {   // a >= p
  bool ge = true, decided = false;
#pragma unroll
  for (int i = 7; i >= 0; --i) {
    const uint32_t b = vrfk::Q32[i];
    if (!decided && a[i] != b) { ge = a[i] > b; decided = true; }
  }
  return ge;
}
// 32 big-endian bytes <-> 8 little-endian u32 words
VRF_HD void load_be256(uint32_t w[8], const uint8_t* p) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const uint8_t* q = p + 4 * (7 - j);
    w[j] = ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | q[3];
  }
}
VRF_HD void store_be256(uint8_t* p, const uint32_t w[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    uint8_t* q = p + 4 * (7 - j);
    q[0] = (uint8_t)(w[j] >> 24); q[1] = (uint8_t)(w[j] >> 16); q[2] = (uint8_t)(w[j] >> 8); q[3] = (uint8_t)w[j];
  }
}

// y from x and the parity bit: the root of x^3 - 3x + b whose lowest bit is `odd`.  false: x is not on the curve.
VRF_HD bool sw_lift_x(FeN& y, const FeN& x, bool odd) {
  const SqrtTables none{};                       // q = 3 (mod 4): the root is one exponentiation, no tables
  FeN root;
  const bool sq = fe_sqrt_or_zsqrt(root, sw_rhs(x), none);
  uint32_t yw[8];
  fe_to_u256(yw, root);
  y = fe_select(((yw[0] & 1u) != 0) != odd, fe_wred(fe_neg(root)), root);
  return sq;
}

// [ref src/lib.rs:14 `codec`] Sec1Codec::point_decode of a 33-byte string: 0x02 / 0x03 || x (big-endian), x < p, on the
// curve.  The cofactor is 1: being on the curve is being in the group.  (The one-byte encoding 0x00 of the point at
// infinity has no 33-byte form: a key, input or output at infinity is not representable at this ABI, and upstream
// rejects such a `Public` / `Input` anyway.)
VRF_HD bool sec1_decode(FeN& x, FeN& y, const uint8_t* enc) {
  const uint32_t tag = enc[0];
  uint32_t xw[8];
  load_be256(xw, enc + 1);
  const bool ok = (tag == 2u || tag == 3u) && !u256_ge_q(xw);
  x = fe_from_u256(xw);
  return sw_lift_x(y, x, (tag & 1u) != 0) && ok;
}

// affine (x, y) -> tag and the big-endian integer x as 8 LE words (and y's words, for callers that hand out x || y)
VRF_HD uint32_t sec1_words(uint32_t xw[8], const FeN& x, const FeN& y, uint32_t* yw_out = nullptr) {
  uint32_t yw[8];
  fe_to_u256(xw, x);
  fe_to_u256(yw, y);
  if (yw_out) {
#pragma unroll
    for (int i = 0; i < 8; ++i) yw_out[i] = yw[i];
  }
  return 2u + (yw[0] & 1u);
}
// x || y as the ABI's 64-byte pairs: 32-byte little-endian canonical integers, or (mont256) arkworks' in-memory limbs
// x 2^256 mod p.  tag 0 (the point at infinity) is written as all zeros.
VRF_HD void xy_store(uint8_t* out, uint32_t tag, const uint32_t xw[8], const uint32_t yw[8], bool mont256) {
  uint32_t a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = tag ? xw[i] : 0u; b[i] = tag ? yw[i] : 0u; }
  if (mont256) { u256_canon_to_mont256(a); u256_canon_to_mont256(b); }
  uint32_t* o = reinterpret_cast<uint32_t*>(out);
#pragma unroll
  for (int i = 0; i < 8; ++i) { o[i] = a[i]; o[8 + i] = b[i]; }
}
VRF_HD void sec1_store(uint8_t* out, uint32_t tag, const uint32_t xw[8]) {
  out[0] = (uint8_t)tag;
  store_be256(out + 1, xw);
}
// a point of the transcript: tag 0 = the point at infinity, which Sec1Codec::point_encode writes as the single byte 0x00
VRF_HD void sha256_put_sec1(Sha256& h, uint32_t tag, const uint32_t xw[8]) {
  sha256_put_byte(h, (uint8_t)tag);
  if (tag != 0) sha256_put_be256(h, xw);
}

VRF_HD void p256_put_suite(Sha256& h, const SuiteStr& ss) { sha256_put_packed64(h, ss.suite_id_w, ss.suite_id_len); }

// [ref src/lib.rs:14 `utils`] hash_to_curve_tai_rfc_9381 (RFC 9381 5.4.1.1): H = string_to_point(0x02 || Hash(suite ||
// 0x01 || data || ctr || 0x00)) for the first ctr in 0..255 that decodes; cofactor 1.  `data` is what the caller passes
// to `Input::new` (RFC 9381's own use: encode_to_curve_salt || alpha with the salt = the public key string).
// one attempt: the candidate x of counter `ctr` (as words) and whether it is the abscissa of a point.  `pre` has absorbed
// suite || 0x01 || data.  The decision is the Jacobi symbol of x^3 - 3x + b -- a third of the exponentiation a root costs.
VRF_HD bool p256_tai_attempt(uint32_t w[8], const Sha256& pre, uint32_t ctr) {
  Sha256 h = pre;
  sha256_put_byte(h, (uint8_t)ctr);
  sha256_put_byte(h, 0x00);
  sha256_final(h);
  sha256_be256(w, h);
  if (u256_ge_q(w)) return false;
  const FeN xc = fe_from_u256(w);
  const FeN rc = fe_canon(sw_rhs(xc));
  const int j = jacobi_limbs(rc.v);
  if (j == 2) {                                          // rounds exhausted (not observed): decide by the root itself
    FeN yt;
    return sw_lift_x(yt, xc, false);
  }
  return j == 1 || j == 0;
}
VRF_HD void p256_tai_prefix(Sha256& pre, const uint8_t* data, uint32_t len, const SuiteStr& ss) {
  sha256_init(pre);
  p256_put_suite(pre, ss);
  sha256_put_byte(pre, 0x01);
  sha256_put_bytes(pre, data, len);
}
// `start`: the first counter to try (0, or the counter k_p256_tai_find already found to decode).  A wave runs as many
// trips as its unluckiest lane (about 7 of the 64 where one lane expects 2), so a trip only decides and the root is
// taken once, after; the batch prover goes further and finds the counters with a work queue (k_p256.hip).
VRF_HD bool p256_hash_to_curve(FeN& x, FeN& y, uint32_t xw[8], const uint8_t* data, uint32_t len, const SuiteStr& ss,
                               uint32_t start = 0) {
  Sha256 pre;
  p256_tai_prefix(pre, data, len, ss);
  bool found = false;
  x = fe_zero(); y = fe_zero();
#pragma unroll 1
  for (uint32_t ctr = start; ctr < 256 && !found; ++ctr) {
    uint32_t w[8];
    if (!p256_tai_attempt(w, pre, ctr)) continue;
    x = fe_from_u256(w); found = true;
#pragma unroll
    for (int i = 0; i < 8; ++i) xw[i] = w[i];
  }
  if (found) (void)sw_lift_x(y, x, false);               // 0x02: the even root
  return found;
}

// [ref src/lib.rs:14 `utils`] nonce_rfc_6979 (RFC 9381 5.4.2.1 / RFC 6979 3.2 with HMAC-SHA-256), as upstream runs it:
// h1 = Hash(point_to_string(H)) enters the HMAC input as it is (RFC 6979's bits2octets would reduce it mod n first: the
// two differ when h1 >= n, 2^-32 of the inputs) and the first candidate T = V is taken mod n (RFC 6979 retries when
// T >= n or T = 0: again 2^-32).  DESIGN.md section 2 lists this among the laxities held the upstream way.
VRF_HD void p256_nonce(uint32_t k[8], const uint32_t sk[8], uint32_t htag, const uint32_t hxw[8]) {
  Sha256 s;
  sha256_init(s);
  sha256_put_sec1(s, htag, hxw);
  sha256_final(s);
  uint32_t h1[8], V[8], K[8];
  sha256_be256(h1, s);
#pragma unroll
  for (int i = 0; i < 8; ++i) { V[i] = 0x01010101u; K[i] = 0; }
#pragma unroll 1
  for (uint32_t sep = 0; sep < 2; ++sep) {
    hmac256_begin(s, K);                                // K = HMAC_K(V || sep || int2octets(x) || h1)
    sha256_put_be256(s, V);
    sha256_put_byte(s, (uint8_t)sep);
    sha256_put_be256(s, sk);
    sha256_put_be256(s, h1);
    uint32_t Kn[8];
    hmac256_end(Kn, s, K);
#pragma unroll
    for (int i = 0; i < 8; ++i) K[i] = Kn[i];
    hmac256_begin(s, K);                                // V = HMAC_K(V)
    sha256_put_be256(s, V);
    hmac256_end(V, s, K);
  }
  hmac256_begin(s, K);
  sha256_put_be256(s, V);
  hmac256_end(V, s, K);
  fr_reduce256<CurveP256>(k, V);
}

// [ref src/lib.rs:14 `utils`] challenge_rfc_9381 (RFC 9381 5.4.3): the first `challenge_len` bytes of Hash(suite || 0x02
// || pk || H || Gamma || U || V || ad || 0x00) as a big-endian integer.
struct Sec1W {
  uint32_t tag;
  uint32_t xw[8];
};
VRF_HD void p256_challenge(uint32_t c[8], const Sec1W (&pts)[5], const uint8_t* ad, uint32_t ad_len, const SuiteStr& ss) {
  Sha256 h;
  sha256_init(h);
  p256_put_suite(h, ss);
  sha256_put_byte(h, 0x02);
#pragma unroll 1
  for (int j = 0; j < 5; ++j) {
    Sec1W p = pts[0];
#pragma unroll
    for (int k = 1; k < 5; ++k)
      if (j == k) p = pts[k];
    sha256_put_sec1(h, p.tag, p.xw);
  }
  sha256_put_bytes(h, ad, ad_len);
  sha256_put_byte(h, 0x00);
  sha256_final(h);
  uint32_t w[8];
  sha256_be256(w, h);
  // the first L bytes of the digest = the top 8 L bits of the 256-bit big-endian integer
  const uint32_t sh = 256u - 8u * ss.challenge_len;     // 0..248, a multiple of 8
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint32_t bit = 32u * (uint32_t)i + sh, j = bit >> 5, r = bit & 31u;
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (j == (uint32_t)k) lo = w[k];
      if (j + 1 == (uint32_t)k) hi = w[k];
    }
    c[i] = bit >= 256u ? 0u : (r ? (lo >> r) | (hi << (32u - r)) : lo);
  }
}

// [ref src/lib.rs:14 `utils`] point_to_hash_rfc_9381: Hash(suite || 0x03 || point_to_string(Gamma) || 0x00) (cofactor 1)
VRF_HD void p256_output_hash(uint32_t out[8], uint32_t tag, const uint32_t xw[8], const SuiteStr& ss) {
  Sha256 h;
  sha256_init(h);
  p256_put_suite(h, ss);
  sha256_put_byte(h, 0x03);
  sha256_put_sec1(h, tag, xw);
  sha256_put_byte(h, 0x00);
  sha256_final(h);
  sha256_be256(out, h);
}

// [ref src/lib.rs:16 `Secret::from_seed`] sk = Hash(seed) read little-endian, mod n (upstream's from_le_bytes_mod_order of
// the hasher's output, whatever the codec's endianness); 0 -> 1 is not reachable from a hash in practice
VRF_HD void p256_secret_from_seed(uint32_t sk[8], const uint8_t* seed, uint32_t len) {
  Sha256 h;
  sha256_init(h);
  sha256_put_bytes(h, seed, len);
  sha256_final(h);
  uint32_t w[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {                         // digest byte 4j.. -> little-endian word j
    const uint32_t v = h.h[j];
    w[j] = (v >> 24) | ((v >> 8) & 0xff00u) | ((v << 8) & 0xff0000u) | (v << 24);
  }
  fr_reduce256<CurveP256>(sk, w);
}

// ---- scalar multiplication ----
// Signed radix-16 digits of a full 256-bit scalar: k + 0x88..8 has nibbles d_w + 8 with d_w in [-8, 7]; the order of
// secp256r1 is just below 2^256, so -- unlike the 253-bit orders of the Edwards suites -- the addition can carry out of
// the top nibble: that carry is digit 64 (0 or 1) and every ladder below has 65 windows.
VRF_HD uint32_t sw_recode(uint32_t rec[8], const uint32_t k[8]) {
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint64_t x = (uint64_t)k[i] + 0x88888888u + c;
    rec[i] = (uint32_t)x;
    c = (uint32_t)(x >> 32);
  }
  return c;
}
VRF_HD int sw_digit(const uint32_t rec[8], uint32_t top, int w) { return w == 64 ? (int)top : scalar_digit4(rec, w); }
constexpr int SW_WINDOWS = 65;

// k * P against a window table (sw_build_table), negated if `negate`
VRF_HD PtW sw_win_mul(const uint32_t* tab, size_t stride, const uint32_t k[8], bool negate) {
  uint32_t rec[8];
  const uint32_t top = sw_recode(rec, k);
  PtW acc = sw_identity();
#pragma unroll 1
  for (int w = SW_WINDOWS - 1; w >= 0; --w) {
    if (w != SW_WINDOWS - 1) {
      acc = sw_dbl4(acc);
    }
    const int d = sw_digit(rec, top, w);
    acc = sw_add(acc, sw_lookup(tab, stride, negate ? -d : d));
  }
  return acc;
}

// ---- the prover's two multiplications of ONE base (Gamma = sk H, V = k H) ----
// Both scalars multiply the same H, so the doublings are spent once, on H, instead of twice, on the accumulators: four
// window tables of H, 2^64 H, 2^128 H, 2^192 H (192 Jacobian doublings + 4 x 7 additions), then each scalar is a 4-way
// Straus sum over 16 windows -- 60 doublings + 64 additions instead of 256 + 65.  (The Edwards provers do the same:
// vrf_core.cuh ProveLayout.)  Per proof 4980 product-equivalents for the two H ladders instead of 6980.
// A scalar enters as |k'| <= n/2 with its sign (k > n/2 becomes n - k, the result negated), so |k'| < 2^255 and the signed
// radix-16 recoding can carry out of the top nibble only as (digit 64, digit 63) = (1, -8), which is digit 63 = +8: the
// 64 digits then split into four rows of 16, row j against the table of 2^(64 j) H.
constexpr int SW_QUAD_TABLES = 4;
VRF_HD void sw_build_quad_tables(uint32_t* tabs, size_t stride, const FeN& x, const FeN& y) {
  PtJ base;
  base.X = x; base.Y = y; base.Z = fe_one();
#pragma unroll 1
  for (int j = 0; j < SW_QUAD_TABLES; ++j) {
    sw_build_table(tabs + (size_t)j * SW_TABLE_WORDS * stride, stride, sw_from_jac(base));
    if (j + 1 < SW_QUAD_TABLES) {
#pragma unroll 1
      for (int k = 0; k < 64; ++k) base = sw_dbl_jac(base);
    }
  }
}
// |k| and sign with |k| <= n/2 (k < n)
VRF_HD bool sw_scalar_fold(uint32_t out[8], const uint32_t k[8]) {
  // n - k < k  <=>  the subtraction (n - k) - k does not borrow ... decided on the halves: compare k with n - k
  uint32_t nk[8];
  uint32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint64_t d = (uint64_t)CurveP256::r32(i) - k[i] - borrow;
    nk[i] = (uint32_t)d;
    borrow = (uint32_t)(d >> 63);
  }
  bool lt = false, decided = false;                       // nk < k
#pragma unroll
  for (int i = 7; i >= 0; --i)
    if (!decided && nk[i] != k[i]) { lt = nk[i] < k[i]; decided = true; }
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = lt ? nk[i] : k[i];
  return lt;
}
// sw_lookup with an access pattern that does not depend on the digit: all eight entries are read, the wanted one kept by
// masks (the prover's tables are indexed by digits of sk and of the nonce: VRFHIP_FLAG_CT_TABLES)
VRF_HD PtW sw_lookup_ct(const uint32_t* tab, size_t stride, int digit) {
  const int mag = digit < 0 ? -digit : digit;
  PtW e = sw_identity();
#pragma unroll 1
  for (int j = 0; j < SW_WIN; ++j) e = sw_select(mag == j + 1, ptw_load(tab + (size_t)j * SW_ENTRY_WORDS * stride, stride), e);
  return sw_cneg(digit < 0, e);
}
VRF_HD PtW sw_quad_mul(const uint32_t* tabs, size_t stride, const uint32_t k[8], bool ct = false) {
  uint32_t mag[8], rec[8];
  const bool neg = sw_scalar_fold(mag, k);
  const uint32_t top = sw_recode(rec, mag);
  PtW acc = sw_identity();
#pragma unroll 1
  for (int i = 15; i >= 0; --i) {
    if (i != 15) acc = sw_dbl4(acc);
#pragma unroll 1
    for (int j = 0; j < SW_QUAD_TABLES; ++j) {
      int d = scalar_digit4(rec, 16 * j + i);
      if (top && j == 3 && i == 15) d = 8;               // (digit 64, digit 63) = (1, -8)  ->  digit 63 = +8
      const uint32_t* tj = tabs + (size_t)j * SW_TABLE_WORDS * stride;
      acc = sw_add(acc, ct ? sw_lookup_ct(tj, stride, neg ? -d : d) : sw_lookup(tj, stride, neg ? -d : d));
    }
  }
  return acc;
}

// The fixed-base comb of the generator: signed radix-256 digits, row w holds j * 256^w * G for j = 1..128 (projective,
// entries of SW_ENTRY_WORDS words), so k * G is 33 additions and no doubling (a radix-16 comb took 65).  33 x 128 x 112 B =
// 473 KB, resident in L2; built once per context, one lane per row (p256_comb_build_row).
constexpr int P256_COMB_ROWS = 33;          // 32 byte digits + the carry out of the top one
constexpr int P256_COMB_WIN = 128;
constexpr size_t P256_COMB_ROW_WORDS = (size_t)P256_COMB_WIN * SW_ENTRY_WORDS;
constexpr size_t P256_COMB_WORDS = (size_t)P256_COMB_ROWS * P256_COMB_ROW_WORDS;
VRF_HD void p256_comb_build_row(uint32_t* comb, int w, const FeN& gx, const FeN& gy) {
  PtW base = sw_from_affine(gx, gy);
#pragma unroll 1
  for (int k = 0; k < 2 * w; ++k) base = sw_dbl4(base);              // 256^w * G
  uint32_t* row = comb + (size_t)w * P256_COMB_ROW_WORDS;
  PtW acc = base;
  ptw_store(row, 1, acc);
#pragma unroll 1
  for (int jj = 1; jj < P256_COMB_WIN; ++jj) {
    acc = sw_add(acc, base);
    ptw_store(row + (size_t)jj * SW_ENTRY_WORDS, 1, acc);
  }
}
// k + 0x80..80 has bytes d_w + 128 with d_w in [-128, 127]; the carry out of the top byte is digit 32
VRF_HD uint32_t sw_recode8(uint32_t rec[8], const uint32_t k[8]) {
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint64_t x = (uint64_t)k[i] + 0x80808080u + c;
    rec[i] = (uint32_t)x;
    c = (uint32_t)(x >> 32);
  }
  return c;
}
// rows: how many rows of the comb the scalar can reach (a challenge of L bytes: L + 1; default: all 33)
VRF_HD PtW sw_comb_mul(const uint32_t* comb, const uint32_t k[8], int rows = P256_COMB_ROWS) {
  uint32_t rec[8];
  const uint32_t top = sw_recode8(rec, k);
  PtW acc = sw_identity();
#pragma unroll 1
  for (int w = 0; w < rows; ++w) {
    uint32_t word = rec[0];
#pragma unroll
    for (int i = 1; i < 8; ++i)
      if ((w >> 2) == i) word = rec[i];
    const int d = w == 32 ? (int)top : (int)((word >> ((w & 3) * 8)) & 255u) - 128;
    const int mag = d < 0 ? -d : d;
    PtW e = ptw_load(comb + (size_t)w * P256_COMB_ROW_WORDS + (size_t)(mag > 0 ? mag - 1 : 0) * SW_ENTRY_WORDS, 1);
    e = sw_select(mag == 0, sw_identity(), e);
    acc = sw_add(acc, sw_cneg(d < 0, e));
  }
  return acc;
}

// Windows the challenge can occupy: it is the first `challenge_len` bytes of a hash, so digits 0 .. 2 challenge_len (the
// last one is the recoding's carry); 33 for the suite's 16 bytes, all 65 when a descriptor asks for 32.  A proof whose c
// field is larger than that cannot verify whatever the ladders compute (the recomputed challenge is below the bound).
VRF_HD int sw_challenge_windows(const SuiteStr& ss) { return (int)(2u * ss.challenge_len + 1u); }

// s * P - c * Q with c the challenge (cw windows): Straus over the two tables: 256 doublings, 65 + cw additions
VRF_HD PtW sw_straus_sc(const uint32_t* tabP, const uint32_t* tabQ, size_t stride, const uint32_t s[8], const uint32_t c[8], int cw) {
  uint32_t srec[8], crec[8];
  const uint32_t stop = sw_recode(srec, s);
  const uint32_t ctop = sw_recode(crec, c);
  PtW acc = sw_identity();
#pragma unroll 1
  for (int w = SW_WINDOWS - 1; w >= 0; --w) {
    if (w != SW_WINDOWS - 1) {
      acc = sw_dbl4(acc);
    }
    acc = sw_add(acc, sw_lookup(tabP, stride, sw_digit(srec, stop, w)));
    if (w < cw)                                          // wave-uniform
      acc = sw_add(acc, sw_lookup(tabQ, stride, -sw_digit(crec, ctop, w)));
  }
  return acc;
}
// s * G - c * Q: the generator's half comes from the comb after the ladder (no doublings for it)
VRF_HD PtW sw_comb_minus_win(const uint32_t* comb, const uint32_t* tabQ, size_t stride, const uint32_t s[8], const uint32_t c[8], int cw) {
  uint32_t crec[8];
  const uint32_t ctop = sw_recode(crec, c);
  PtW acc = sw_identity();
#pragma unroll 1
  for (int w = cw - 1; w >= 0; --w) {
    if (w != cw - 1) {
      acc = sw_dbl4(acc);
    }
    acc = sw_add(acc, sw_lookup(tabQ, stride, -sw_digit(crec, ctop, w)));
  }
  return sw_add(acc, sw_comb_mul(comb, s));
}

// ---- projective -> Sec1 with one shared inversion ----
template <bool CT, int N>
VRF_HD void sw_to_sec1(Sec1W (&out)[N], const PtW (&p)[N], uint32_t (*yw)[8] = nullptr) {
  FeN z[N], pre[N];
  bool inf[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    inf[i] = fe_is_zero(p[i].Z);
    z[i] = fe_select(inf[i], fe_one(), p[i].Z);
    pre[i] = i == 0 ? z[0] : fe_mul(pre[i - 1], z[i]);
  }
  FeN acc = fe_inv<CT>(pre[N - 1]);
#pragma unroll
  for (int i = N - 1; i >= 0; --i) {
    const FeN zi = i == 0 ? acc : fe_mul(acc, pre[i - 1]);
    acc = fe_mul(acc, z[i]);
    const uint32_t tag = sec1_words(out[i].xw, fe_mul(p[i].X, zi), fe_mul(p[i].Y, zi), yw ? yw[i] : nullptr);
    out[i].tag = inf[i] ? 0u : tag;
  }
}

// 32-byte big-endian scalar -> words mod n (`Sec1Codec::scalar_decode` = from_be_bytes_mod_order); returns whether the
// integer was canonical (< n).  As for the Edwards suites (and RFC 9381 5.4.4: "if s >= q, output INVALID"), the proof's s
// and a secret key must be canonical; the challenge c is taken mod n, which is what upstream's `Proof` decoding does.
VRF_HD bool p256_scalar_decode(uint32_t out[8], const uint8_t* be) {
  uint32_t w[8];
  load_be256(w, be);
  fr_reduce256<CurveP256>(out, w);
  return fr_is_canonical<CurveP256>(w);
}

// ---- IETF verify, per item [ref src/lib.rs:14 `ietf::Verifier::verify`, RFC 9381 5.3] ----
// stage 1: decode pk, H, Gamma (33-byte Sec1) and the proof scalars; false = InvalidData
// The c field: a scalar (mod n) when CHALLENGE_LEN = 32; with a shorter challenge a field holding more than CHALLENGE_LEN
// bytes is no proof string -- it is kept as it stands and never equals a recomputed challenge (vrf_core.cuh
// proof_challenge_decode)
VRF_HD void p256_challenge_decode(uint32_t c[8], const uint8_t* cb, uint32_t challenge_len) {
  if (challenge_len < 32u) load_be256(c, cb);
  else (void)p256_scalar_decode(c, cb);
}
VRF_HD bool p256_verify_decode_item(FeN (&x)[3], FeN (&y)[3], Sec1W (&enc)[3], uint32_t c[8], uint32_t s[8],
                                    const uint8_t* pk, const uint8_t* h, const uint8_t* gamma, const uint8_t* cb,
                                    const uint8_t* sb, uint32_t challenge_len) {
  bool ok = true;
#pragma unroll 1
  for (int j = 0; j < 3; ++j) {
    const uint8_t* e = j == 0 ? pk : j == 1 ? h : gamma;
    FeN xx, yy;
    ok = sec1_decode(xx, yy, e) && ok;
    Sec1W w;
    w.tag = e[0];
    load_be256(w.xw, e + 1);
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (j == k) { x[k] = xx; y[k] = yy; enc[k] = w; }
  }
  p256_challenge_decode(c, cb, challenge_len);
  return p256_scalar_decode(s, sb) && ok;
}
// stage 1 for callers that hold the points as arkworks `Affine { x, y }`: pk, H, Gamma as 64-byte x || y (little-endian
// canonical integers, or arkworks' Montgomery limbs when mont256).  No square root: a coordinate >= p or a point off the
// curve is InvalidData; the encodings the challenge hashes are rebuilt from (x, parity of y).
VRF_HD bool p256_verify_decode_affine_item(FeN (&x)[3], FeN (&y)[3], Sec1W (&enc)[3], uint32_t c[8], uint32_t s[8],
                                           const uint8_t* pk_xy, const uint8_t* h_xy, const uint8_t* gamma_xy, const uint8_t* cb,
                                           const uint8_t* sb, bool mont256, uint32_t challenge_len) {
  bool ok = true;
#pragma unroll 1
  for (int j = 0; j < 3; ++j) {
    const uint32_t* e = reinterpret_cast<const uint32_t*>(j == 0 ? pk_xy : j == 1 ? h_xy : gamma_xy);
    uint32_t xin[8], yin[8], xc[8], yc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { xin[i] = e[i]; yin[i] = e[8 + i]; }
    const FeN xx = fe_from_abi(xc, xin, mont256), yy = fe_from_abi(yc, yin, mont256);
    ok = !u256_ge_q(xin) && !u256_ge_q(yin) && sw_on_curve(xx, yy) && ok;
    Sec1W w;
    w.tag = 2u + (yc[0] & 1u);
#pragma unroll
    for (int i = 0; i < 8; ++i) w.xw[i] = xc[i];
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (j == k) { x[k] = xx; y[k] = yy; enc[k] = w; }
  }
  p256_challenge_decode(c, cb, challenge_len);
  return p256_scalar_decode(s, sb) && ok;
}
// stage 3: c' = challenge(pk, H, Gamma, U, V, ad) == c.  1 = the proof does not verify
VRF_HD uint8_t p256_verify_finish_item(const PtW& U, const PtW& V, const Sec1W (&enc)[3], const uint32_t c[8],
                                       const uint8_t* ad, uint32_t ad_len, const SuiteStr& ss) {
  const PtW uv[2] = {U, V};
  Sec1W w[2];
  sw_to_sec1<false>(w, uv);
  const Sec1W pts[5] = {enc[0], enc[1], enc[2], w[0], w[1]};
  uint32_t cc[8];
  p256_challenge(cc, pts, ad, ad_len, ss);
  uint32_t diff = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) diff |= cc[i] ^ c[i];
  return diff ? 1 : 0;
}

// ---- IETF prove, per item [ref src/lib.rs:14 `ietf::Prover::prove`, RFC 9381 5.1] ----
// stage 1: sk, H = Input::new(msg) (or the given, decoded H), nonce k.  false = InvalidData (no H within 256 tries, an
// undecodable given H)
VRF_HD bool p256_prove_prepare_item(uint32_t sk[8], uint32_t k[8], FeN& hx, FeN& hy, Sec1W& henc, const uint8_t* sk_be,
                                    const uint8_t* msg, uint32_t msg_len, const uint8_t* h_given, const SuiteStr& ss,
                                    uint32_t tai_start = 0) {
  const bool sk_ok = p256_scalar_decode(sk, sk_be);
  bool ok;
  if (h_given) {
    ok = sec1_decode(hx, hy, h_given);
    henc.tag = h_given[0];
    load_be256(henc.xw, h_given + 1);
  } else {
    ok = p256_hash_to_curve(hx, hy, henc.xw, msg, msg_len, ss, tai_start);
    henc.tag = 2u;
  }
  p256_nonce(k, sk, henc.tag, henc.xw);
  return ok && sk_ok;
}
// stage 3: res = {pk = sk G, Gamma = sk H, U = k G, V = k H}: c = challenge, s = k + c sk (mod n)
VRF_HD void p256_prove_finish_item(Sec1W& pk, Sec1W& gamma, uint32_t c[8], uint32_t s[8], const PtW (&res)[4],
                                   const Sec1W& henc, const uint32_t sk[8], const uint32_t k[8], const uint8_t* ad,
                                   uint32_t ad_len, const SuiteStr& ss, uint32_t (*yw)[8] = nullptr) {
  Sec1W w[4];
  sw_to_sec1<true>(w, res, yw);
  pk = w[0]; gamma = w[1];
  const Sec1W pts[5] = {w[0], henc, w[1], w[2], w[3]};
  p256_challenge(c, pts, ad, ad_len, ss);
  uint32_t cs[8];
  fr_mul<CurveP256>(cs, c, sk);
  fr_add<CurveP256>(s, cs, k);
}

// ---- Pedersen VRF, per item [ref src/lib.rs:14 `pedersen::{Prover, Verifier}`; SURVEY.md Appendix A.5 with this suite's
// codec and hash].  Unpinned on this suite: no published vector, and upstream's BLINDING_BASE is not known here (the
// descriptor carries the base; the built-in one is a nothing-up-my-sleeve point, tools/gen_constants.py). ----
// `PedersenSuite::blinding`: Hash(suite || 0xCC || scalar_encode(sk) || point_encode(H) || ad || 0x00) mod n
VRF_HD void p256_blinding(uint32_t b[8], const uint32_t sk[8], const Sec1W& henc, const uint8_t* ad, uint32_t ad_len, const SuiteStr& ss) {
  Sha256 h;
  sha256_init(h);
  p256_put_suite(h, ss);
  sha256_put_byte(h, 0xCC);
  sha256_put_be256(h, sk);
  sha256_put_sec1(h, henc.tag, henc.xw);
  sha256_put_bytes(h, ad, ad_len);
  sha256_put_byte(h, 0x00);
  sha256_final(h);
  uint32_t w[8];
  sha256_be256(w, h);
  fr_reduce256<CurveP256>(b, w);
}
// prove, stage 3: res = {pk_com = sk G + b B, Gamma = sk H, R = k G + kb B, Ok = k H}
VRF_HD void p256_ped_prove_finish_item(Sec1W (&enc)[4], uint32_t s[8], uint32_t sb[8], const PtW (&res)[4], const Sec1W& henc,
                                       const uint32_t sk[8], const uint32_t k[8], const uint32_t b[8], const uint32_t kb[8],
                                       const uint8_t* ad, uint32_t ad_len, const SuiteStr& ss, uint32_t (*yw)[8] = nullptr) {
  sw_to_sec1<true>(enc, res, yw);
  const Sec1W pts[5] = {enc[0], henc, enc[1], enc[2], enc[3]};
  uint32_t c[8], t[8];
  p256_challenge(c, pts, ad, ad_len, ss);
  fr_mul<CurveP256>(t, c, sk);
  fr_add<CurveP256>(s, t, k);
  fr_mul<CurveP256>(t, c, b);
  fr_add<CurveP256>(sb, t, kb);
}
// verify, stage 1: decode H, Gamma, pk_com, R, Ok and the scalars; c = challenge(pk_com, H, Gamma, R, Ok, ad).
// x / y / enc order: H, Gamma, pk_com, R, Ok.  false = InvalidData (a point off the curve, s or sb >= n)
VRF_HD bool p256_ped_verify_decode_item(FeN (&x)[5], FeN (&y)[5], uint32_t c[8], uint32_t s[8], uint32_t sb[8], const uint8_t* h,
                                        const uint8_t* gamma, const uint8_t* pk_com, const uint8_t* r, const uint8_t* ok_pt,
                                        const uint8_t* s_be, const uint8_t* sb_be, const uint8_t* ad, uint32_t ad_len,
                                        const SuiteStr& ss) {
  bool ok = true;
  Sec1W enc[5];
#pragma unroll 1
  for (int j = 0; j < 5; ++j) {
    const uint8_t* e = j == 0 ? h : j == 1 ? gamma : j == 2 ? pk_com : j == 3 ? r : ok_pt;
    FeN xx, yy;
    ok = sec1_decode(xx, yy, e) && ok;
    Sec1W w;
    w.tag = e[0];
    load_be256(w.xw, e + 1);
#pragma unroll
    for (int k = 0; k < 5; ++k)
      if (j == k) { x[k] = xx; y[k] = yy; enc[k] = w; }
  }
  const Sec1W pts[5] = {enc[2], enc[0], enc[1], enc[3], enc[4]};
  p256_challenge(c, pts, ad, ad_len, ss);
  const bool s_ok = p256_scalar_decode(s, s_be), sb_ok = p256_scalar_decode(sb, sb_be);
  return ok && s_ok && sb_ok;
}
// verify, stage 2: the two equations as "is the point at infinity":  s H - c Gamma - Ok  and  s G + sb B - c pk_com - R
VRF_HD bool sw_is_infinity(const PtW& p) { return fe_is_zero(p.Z); }
VRF_HD bool p256_ped_verify_eq_h(const uint32_t* tabH, const uint32_t* tabG, size_t stride, const FeN& okx, const FeN& oky,
                                 const uint32_t s[8], const uint32_t c[8], int cw) {
  const PtW v = sw_straus_sc(tabH, tabG, stride, s, c, cw);
  return sw_is_infinity(sw_add(v, sw_cneg(true, sw_from_affine(okx, oky))));
}
VRF_HD bool p256_ped_verify_eq_g(const uint32_t* comb_g, const uint32_t* comb_b, const uint32_t* tabP, size_t stride, const FeN& rx,
                                 const FeN& ry, const uint32_t s[8], const uint32_t sb[8], const uint32_t c[8], int cw) {
  PtW u = sw_comb_minus_win(comb_g, tabP, stride, s, c, cw);
  u = sw_add(u, sw_comb_mul(comb_b, sb));
  return sw_is_infinity(sw_add(u, sw_cneg(true, sw_from_affine(rx, ry))));
}

VRF_NS_END
