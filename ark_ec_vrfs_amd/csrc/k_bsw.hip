// k_bsw.hip -- kernels of `suites::bandersnatch_sw` (/root/reference src/lib.rs:14): the codec-bound stages of the IETF and
// Pedersen schemes on the short-Weierstrass presentation of Bandersnatch (bsw_core.cuh).  The stages between them are the
// twisted-Edwards suite's: the provers' multiplication stage IS launch_prove_stage2_bs (k_prove.hip), and the verifiers'
// Straus stages wrap the same per-item functions (verify_straus_item, pedersen_verify_straus_item) -- so the workspace
// (window tables, projective intermediates, aux words, flag bytes) has the layout k_prove.hip / k_verify.hip /
// k_pedersen.hip give it, and the launch arguments are theirs (vrf_types.h), with 33-byte point rows.
#include "bsw.h"
#include "bsw_core.cuh"
#include "msm.cuh"
#include "tai_find.cuh"

VRF_NS_BEGIN

void launch_prove_stage2_bs(const ProveArgs& a, hipStream_t st);      // k_prove.hip (suite 1, stage 2)

namespace {

// flag byte of an item in the workspace: bit 0 = inputs valid; bits 6-7 = flag byte of enc(H) (its x words sit in aux[0..8))
VRF_HD Enc33 aux_h_enc(const uint32_t* aux, uint8_t flag) {
  Enc33 e;
#pragma unroll
  for (int j = 0; j < 8; ++j) e.w[j] = aux[j];
  e.fl = flag & 0xC0u;
  return e;
}

// x || y (64 bytes, canonical little-endian) of an encoded finite point; zeros for infinity or an invalid item
VRF_HD void store_xy64(uint8_t* base, size_t i, const Enc33& e, const uint32_t yw[8], bool ok) {
  uint32_t* p = reinterpret_cast<uint32_t*>(base + i * 64);
  const bool fin = ok && (e.fl & BSW_INF) == 0u;
#pragma unroll
  for (int k = 0; k < 8; ++k) { p[k] = fin ? e.w[k] : 0u; p[8 + k] = fin ? yw[k] : 0u; }
}

// ---- Secret::from_seed / Secret::public ----
__global__ void __launch_bounds__(BLOCK, 2) k_bsw_secret_from_seed(size_t n, const uint8_t* seeds, uint32_t seed_len,
                                                                   uint8_t* sk_out, uint8_t* pk_out, DevTables T) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t sk[8];
  secret_from_seed_item<BswS>(sk, seeds + i * (size_t)seed_len, seed_len);
  store32(sk_out, i, sk);
  if (pk_out) store33(pk_out, i, bsw_encode<BswS, true>(gcomb_mul<BswS>(T.g_comb, sk)));
}

// ---- Input::new ----
__global__ void __launch_bounds__(BLOCK, 2) k_bsw_hash_to_curve(size_t n, BytesView msg, uint8_t* points, DevTables T,
                                                                const uint8_t* tai_ctr) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  const uint8_t* m; uint32_t len;
  bytes_get(msg, i, m, len);
  store33(points, i, bsw_encode<BswS>(bsw_hash_to_curve_tai(m, len, T.sq, tai_ctr ? tai_ctr[i] : 0u)));
}

// ---- Output::hash ----
__global__ void __launch_bounds__(BLOCK) k_bsw_output_hash(size_t n, const uint8_t* gamma, uint8_t* hash, SqrtTables T) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t o[16];
  bsw_output_hash(o, enc33_canonical(load33(gamma, i)), T.str);      // `Output::hash` encodes the typed point
  uint32_t* p = reinterpret_cast<uint32_t*>(hash + i * 64);
#pragma unroll
  for (int j = 0; j < 16; ++j) p[j] = o[j];
}

// ---- codec: checked point decoding; xy (nullable) receives the Weierstrass x || y ----
__global__ void __launch_bounds__(BLOCK) k_bsw_point_validate(size_t n, const uint8_t* pts, uint8_t* xy, uint8_t* status,
                                                              DevTables T) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  FeN tx, ty, sx, sy;
  bool inf;
  bool ok = bsw_decode<BswS>(tx, ty, sx, sy, inf, load33(pts, i), T.sq);
  ok = ok && in_prime_subgroup<BswS>(tx, ty, T.sq);
  status[i] = ok ? ST_OK : ST_INVALID_DATA;
  if (xy) {
    uint32_t xw[8], yw[8];
    fe_to_u256(xw, sx);
    fe_to_u256(yw, sy);
    uint32_t* p = reinterpret_cast<uint32_t*>(xy + i * 64);
#pragma unroll
    for (int j = 0; j < 8; ++j) { p[j] = ok && !inf ? xw[j] : 0u; p[8 + j] = ok && !inf ? yw[j] : 0u; }
  }
}

// ---- provers, stage 1: H (given, or hashed from the message), enc(H), nonce(s), blinding, window tables of H ----
__global__ void __launch_bounds__(BLOCK, 2) k_bsw_prove_prepare(ProveArgs a) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= a.n) return;
  uint32_t sk[8], k[8];
  load32(sk, a.sk, i);
  bool valid = fr_is_canonical<BswS>(sk);
  FeN x, y;
  Enc33 h_enc;
  if (a.h_given) {
    const Enc33 e = load33(a.h_given, i);
    valid = bsw_decode<BswS>(x, y, e, a.T.sq) && valid;
    if (a.check_mask & CHK_INPUT) valid = in_prime_subgroup<BswS>(x, y, a.T.sq) && valid;     // a given H is wire data
    h_enc = enc33_canonical(e);
  } else {
    const uint8_t* msg; uint32_t msg_len;
    bytes_get(a.msg, i, msg, msg_len);
    const PtE hp = bsw_hash_to_curve_tai(msg, msg_len, a.T.sq, a.ws.flags[i]);     // k_tai_find left the first candidate's counter
    const SwFrac f = bsw_frac<BswS>(hp.X, hp.Y, hp.Z);
    const FeN dens[2] = {fe_full(hp.Z), f.den};
    FeN inv[2];
    fe_batch_inv(inv, dens);
    x = fe_mul(hp.X, inv[0]);
    y = fe_mul(hp.Y, inv[0]);
    h_enc = bsw_encode_frac(f, inv[1]);
  }
  bsw_nonce(k, sk, h_enc);
  build_prove_tables<BswS>(a.ws.tabs + i * ProveLayout<BswS>::TAB_WORDS, x, y);
  uint32_t* aux = a.ws.aux + i * AUX_WORDS;
#pragma unroll
  for (int j = 0; j < 8; ++j) { aux[j] = h_enc.w[j]; aux[8 + j] = k[j]; }
  if (a.pedersen) {
    const uint8_t* ad; uint32_t ad_len;
    bytes_get(a.ad, i, ad, ad_len);
    uint32_t b[8], kb[8];
    bsw_blinding(b, sk, h_enc, ad, ad_len, a.T.sq.str);
    bsw_nonce(kb, b, h_enc);
#pragma unroll
    for (int j = 0; j < 8; ++j) { aux[16 + j] = b[j]; aux[24 + j] = kb[j]; }
  }
  a.ws.flags[i] = (uint8_t)((valid ? 1u : 0u) | (h_enc.fl & 0xC0u));
}

// ---- provers, stage 3: the four products [sk*H, sk*G (+ b*B), k*H, k*G (+ kb*B)] -> encodings, challenge, responses ----
__global__ void __launch_bounds__(BLOCK, 2) k_bsw_prove_finish(ProveArgs a) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= a.n) return;
  const uint32_t* pts = a.ws.pts + i * PROVE_PTS_WORDS;
  SwFrac f[4];
  FeN dens[4], inv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f[j] = bsw_frac<BswS>(fe_load<1, 5>(pts + j * UV_WORDS), fe_load<1, 5>(pts + j * UV_WORDS + NL),
                          fe_load<1, 5>(pts + j * UV_WORDS + 2 * NL));
    dens[j] = f[j].den;
  }
  fe_batch_inv<true>(inv, dens);                 // secret-dependent values: fixed-shape inversion
  Enc33 e[4];
  uint32_t syw[4][8];                              // canonical y words (x || y outputs)
#pragma unroll
  for (int j = 0; j < 4; ++j) e[j] = bsw_encode_frac(f[j], inv[j], syw[j]);
  const uint32_t* aux = a.ws.aux + i * AUX_WORDS;
  const uint8_t flag = a.ws.flags[i];
  const bool ok = (flag & 1u) != 0;
  const Enc33 h_enc = aux_h_enc(aux, flag);
  uint32_t sk[8], k[8], c[8], cs[8], s[8];
  load32(sk, a.sk, i);
#pragma unroll
  for (int j = 0; j < 8; ++j) k[j] = aux[8 + j];
  const uint8_t* ad; uint32_t ad_len;
  bytes_get(a.ad, i, ad, ad_len);
  const Enc33 cp[5] = {e[1], h_enc, e[0], e[3], e[2]};        // pk (or pk_com), H, Gamma, k*G (+ kb*B), k*H
  bsw_challenge5(c, cp, ad, ad_len, a.T.sq.str);
  fr_mul<BswS>(cs, c, sk);
  fr_add<BswS>(s, cs, k);
  if (a.pedersen) {
    uint32_t b[8], kb[8], cb[8], sb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { b[j] = aux[16 + j]; kb[j] = aux[24 + j]; }
    fr_mul<BswS>(cb, c, b);
    fr_add<BswS>(sb, cb, kb);
    if (!ok) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { sb[j] = 0; b[j] = 0; }
    }
    if (a.out_affine) { store_xy64(a.r_out, i, e[3], syw[3], ok); store_xy64(a.ok_out, i, e[2], syw[2], ok); }
    else { store33(a.r_out, i, e[3], ok); store33(a.ok_out, i, e[2], ok); }
    store32(a.sb_out, i, sb);
    if (a.blinding_out) store32(a.blinding_out, i, b);
  }
  if (!ok) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { c[j] = 0; s[j] = 0; }
  }
  store32(a.s, i, s);
  if (a.c) store32(a.c, i, c);
  if (a.out_affine) {
    // x || y of the Weierstrass points instead of the encodings: the caller builds typed values without a square root
    store_xy64(a.gamma, i, e[0], syw[0], ok);
    if (a.pk_out) store_xy64(a.pk_out, i, e[1], syw[1], ok);
  } else {
    store33(a.gamma, i, e[0], ok);
    if (a.pk_out) store33(a.pk_out, i, e[1], ok);
  }
  if (a.h_out) store33(a.h_out, i, h_enc);
  if (a.status) a.status[i] = ok ? ST_OK : ST_INVALID_DATA;
}

// Phase A (bsw_core.cuh) of a point handed over as Weierstrass x || y (64 bytes; mont256: arkworks' in-memory Montgomery
// limbs): range and curve checks instead of the square root.  ce: the encoding `point_encode` gives it.
VRF_HD uint32_t bsw_in_a_xy(uint32_t* slot, FeN& run, Enc33& ce, const uint8_t* src, size_t i, bool mont256) {
  const uint32_t* w = reinterpret_cast<const uint32_t*>(src + i * 64);
  uint32_t xin[8], yin[8], yw[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { xin[j] = w[j]; yin[j] = w[8 + j]; }
  bool ok = !u256_ge(xin, vrfk::Q32) && !u256_ge(yin, vrfk::Q32);
  const FeN sx = fe_from_abi(ce.w, xin, mont256), sy = fe_from_abi(yw, yin, mont256);      // ce.w, yw: canonical words
  ok = fe_eq(fe_sqr(sy), bsw_rhs(sx)) && ok;
  ce.fl = u256_gt(yw, vrfk::QM1H32) ? BSW_NEG : 0u;
  return bsw_in_a_store(slot, run, sx, sy, ok, false);
}

// ---- IETF verify, stage 1: decode pk, H, Gamma; GLV window tables; canonical encodings for the challenge ----
// aux: 3 x 8 words of x, then one word holding the three flag bytes.
// a.affine_in (1: canonical, 2: Montgomery-256): the three points come as Weierstrass x || y (64 bytes): range and curve
// checks instead of the square root.  a.key_index (keyed verification): pk is not read -- the key set holds its canonical
// encoding, its validity and its comb (the U half runs k_bsw_verify_comb_u).
__global__ void __launch_bounds__(BLOCK, 2) k_bsw_verify_decode(VerifyArgs a) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= a.n) return;
  uint32_t* aux = a.ws.aux + i * AUX_WORDS;
  uint32_t* tabs = a.ws.tabs + i * (VERIFY_TABS * WIN_TABLE_WORDS);
  uint32_t* scr = a.ws.pts + i * PROVE_PTS_WORDS;       // 3 slots of 36 words: the item's own slice, idle until the Straus stage
  static_assert(PROVE_PTS_WORDS >= 3 * BSW_SLOT, "decode scratch");
  const int p0 = a.key_index ? 1 : 0;
  // phase A: square roots (or range / curve checks of x || y), map denominators, their running product
  FeN run = fe_one();
  uint32_t bits = 0, fls = 0;
#pragma unroll 1
  for (int p = p0; p < 3; ++p) {
    const uint8_t* src = p == 0 ? a.pk : p == 1 ? a.h : a.gamma;
    Enc33 ce;
    uint32_t st;
    if (a.affine_in) {
      st = bsw_in_a_xy(scr + p * BSW_SLOT, run, ce, src, i, a.affine_in == 2);
    } else {
      const Enc33 e = load33(src, i);
      st = bsw_in_a(scr + p * BSW_SLOT, run, e, a.T.sq);
      ce = enc33_canonical(e);
    }
    bits |= st << (2 * p);
#pragma unroll
    for (int j = 0; j < 8; ++j) aux[8 * p + j] = ce.w[j];
    fls |= ce.fl << (8 * p);
  }
  // phase B: ONE inversion for the item's points; Edwards coordinates, subgroup test, tables
  FeN inv = fe_inv(run);
  bool valid = true;
#pragma unroll 1
  for (int p = 2; p >= p0; --p) {
    const uint32_t st = (bits >> (2 * p)) & 3u;
    FeN x, y;
    bsw_in_b(x, y, inv, scr + p * BSW_SLOT, (st & BSW_A_INF) != 0);
    valid = valid && (st & BSW_A_OK) != 0;
    if ((a.check_mask >> p) & 1u) valid = in_prime_subgroup<BswS>(x, y, a.T.sq) && valid;       // bit p: pk, H, Gamma
    build_glv_tables<BswS>(tabs + p * 2 * WIN_TABLE_WORDS, x, y);
  }
  if (a.key_index) {
    const uint32_t key = a.key_index[i];
    const bool key_ok = key < a.n_keys && a.key_valid[key] != 0;
    valid = valid && key_ok;
    const Enc33 ke = load33(a.pk, key_ok ? key : 0);          // a.pk: the key set's canonical encodings
#pragma unroll
    for (int j = 0; j < 8; ++j) aux[j] = ke.w[j];
    fls |= ke.fl & 0xC0u;
  }
  aux[24] = fls;
  a.ws.flags[i] = valid ? 1 : 0;
}

// the proof's scalars for the Straus stages: c mod r, s as is (a non-canonical s is reported by the finish stage)
VRF_HD void bsw_load_cs(uint32_t c[8], uint32_t s[8], const uint8_t* c_arr, const uint8_t* s_arr, size_t i) {
  uint32_t cw[8];
  load32(cw, c_arr, i); load32(s, s_arr, i);
  fr_reduce256<BswS>(c, cw);
  const bool s_ok = fr_is_canonical<BswS>(s);
#pragma unroll
  for (int j = 0; j < 8; ++j) { c[j] = s_ok ? c[j] : 0u; s[j] = s_ok ? s[j] : 0u; }
}

// stage 2: HALF 1: V = s*H - c*Gamma; HALF 0: U = s*G - c*pk (separate launches: every wave runs one shape)
template <int HALF>
__global__ void __launch_bounds__(BLOCK) k_bsw_verify_straus(VerifyArgs a) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= a.n) return;
  uint32_t c[8], s[8];
  bsw_load_cs(c, s, a.c, a.s, i);
  verify_straus_item<BswS, HALF>(a.ws.pts + i * PROVE_PTS_WORDS + HALF * UV_WORDS, a.T,
                                 a.ws.tabs + i * (VERIFY_TABS * WIN_TABLE_WORDS), c, s);
}

// stage 2, keyed U half: U = s*G - c*Y from two fixed-base combs (k_verify_comb_u's arithmetic)
__global__ void __launch_bounds__(BLOCK) k_bsw_verify_comb_u(VerifyArgs a) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= a.n) return;
  uint32_t c[8], s[8];
  bsw_load_cs(c, s, a.c, a.s, i);
  const uint32_t key = a.key_index[i] < a.n_keys ? a.key_index[i] : 0;
  PtE r = gcomb_mul<BswS>(a.T.g_comb, s);
  r = comb_add<BswS>(r, a.key_combs + (size_t)key * COMB_WORDS, c, true);
  uint32_t* out = a.ws.pts + i * PROVE_PTS_WORDS;
  fe_store(out, r.X); fe_store(out + NL, r.Y); fe_store(out + 2 * NL, r.Z);
}

// ---- key sets: validated keys (encodings canonicalised in place), their Edwards coordinates for the comb build ----
__global__ void __launch_bounds__(BLOCK) k_bsw_keyset_decode(size_t n, uint8_t* pks, uint32_t* xy, uint8_t* valid, DevTables T) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  const Enc33 e = load33(pks, i);
  FeN x, y;
  bool ok = bsw_decode<BswS>(x, y, e, T.sq);
  ok = ok && in_prime_subgroup<BswS>(x, y, T.sq);
  // an invalid key gets the identity's tables: every proof that names it is reported InvalidData anyway
  fe_store(xy + i * 2 * NL, fe_select(ok, x, fe_zero()));
  fe_store(xy + i * 2 * NL + NL, fe_select(ok, y, fe_one()));
  valid[i] = ok ? 1 : 0;
  store33(pks, i, enc33_canonical(e));
}
__global__ void __launch_bounds__(64, 2) k_bsw_keyset_comb(size_t n_keys, const uint32_t* xy, uint32_t* combs, uint32_t* prefix) {
  size_t t = (size_t)blockIdx.x * 64 + threadIdx.x;           // one lane per (key, row)
  if (t >= n_keys * COMB_ROWS) return;
  const size_t key = t / COMB_ROWS;
  const int w = (int)(t % COMB_ROWS);
  const FeN x = fe_load<1, 2>(xy + key * 2 * NL), y = fe_load<1, 2>(xy + key * 2 * NL + NL);
  comb_build_row<BswS>(combs + key * COMB_WORDS + (size_t)w * COMB_COLS * PTA_WORDS, prefix + t * COMB_COLS * NL, x, y, w);
}

// ---- MSM over caller-supplied Weierstrass bases: the Edwards sum (x || y, canonical) -> 33-byte encoding + Weierstrass x || y ----
// (a base without an Edwards image -- y = 0, a coordinate >= q -- left te_sw_map as the row (0, 0), which the Edwards MSM
// reports: it is not on that curve)
__global__ void k_bsw_msm_out(const uint8_t* te_xy, uint8_t* out33, uint8_t* out_xy, uint8_t* status) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const uint8_t bad = status[0];
  const uint32_t* w = reinterpret_cast<const uint32_t*>(te_xy);
  uint32_t xw[8], yw[8], cw[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { xw[j] = w[j]; yw[j] = w[8 + j]; }
  PtE p = te_from_affine(fe_from_abi(cw, xw, false), fe_from_abi(cw, yw, false));
  const SwFrac f = bsw_frac<BswS>(p.X, p.Y, p.Z);
  uint32_t syw[8];
  const Enc33 e = bsw_encode_frac(f, fe_inv(f.den), syw);
  const bool ok = bad == 0;
  store33(out33, 0, ok ? e : enc33_infinity());
  if (out_xy) store_xy64(out_xy, 0, e, syw, ok);
  status[0] = ok ? ST_OK : ST_INVALID_DATA;
}

// stage 3: encode U, V (one inversion); challenge; compare
__global__ void __launch_bounds__(BLOCK, 2) k_bsw_verify_finish(VerifyArgs a) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= a.n) return;
  const uint32_t* pts = a.ws.pts + i * PROVE_PTS_WORDS;
  SwFrac f[2];
  FeN dens[2], inv[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    f[j] = bsw_frac<BswS>(fe_load<1, 5>(pts + j * UV_WORDS), fe_load<1, 5>(pts + j * UV_WORDS + NL),
                          fe_load<1, 5>(pts + j * UV_WORDS + 2 * NL));
    dens[j] = f[j].den;
  }
  fe_batch_inv(inv, dens);
  const uint32_t* aux = a.ws.aux + i * AUX_WORDS;
  Enc33 cp[5];
#pragma unroll
  for (int p = 0; p < 3; ++p) {
#pragma unroll
    for (int j = 0; j < 8; ++j) cp[p].w[j] = aux[8 * p + j];
    cp[p].fl = (aux[24] >> (8 * p)) & 0xffu;
  }
  cp[3] = bsw_encode_frac(f[0], inv[0]);
  cp[4] = bsw_encode_frac(f[1], inv[1]);
  const uint8_t* ad; uint32_t ad_len;
  bytes_lite_get(a.ad, i, ad, ad_len);
  uint32_t c[8], sc[8], c2[8], cr[8];
  load32(c, a.c, i); load32(sc, a.s, i);
  bsw_challenge5(c2, cp, ad, ad_len, a.T.sq.str);
  proof_challenge_decode<BswS>(cr, c, a.T.sq.str);
  uint32_t diff = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) diff |= c2[j] ^ cr[j];
  const bool valid = a.ws.flags[i] != 0 && fr_is_canonical<BswS>(sc);
  a.status[i] = (uint8_t)(!valid ? ST_INVALID_DATA : (diff == 0 ? ST_OK : ST_VERIFICATION_FAILURE));
}

// ---- Pedersen verify, stage 1: decode H, Gamma, pk_com, R, Ok; tables of the first three; affine R, Ok; challenge ----
__global__ void __launch_bounds__(BLOCK, 2) k_bsw_ped_verify_decode(PedersenVerifyArgs a) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= a.n) return;
  uint32_t* tabs = a.ws.tabs + i * (VERIFY_TABS * WIN_TABLE_WORDS);
  uint32_t* pts = a.ws.pts + i * PROVE_PTS_WORDS;
  Enc33 enc[5];
  bool valid = true;
  // The five points share ONE map inversion (bsw_core.cuh).  Their five 36-word slots sit where pk_com's tables go
  // (tabs[4..5]: 576 words); the chain starts with pk_com, so the reverse pass meets it LAST -- Ok and R have gone to the
  // results region, H's and Gamma's tables to tabs[0..3], and pk_com's slot is in registers before its tables overwrite it.
  uint32_t* scr = tabs + 4 * WIN_TABLE_WORDS;
  static_assert(2 * WIN_TABLE_WORDS >= 5 * BSW_SLOT, "decode scratch");
  FeN run = fe_one();
  uint32_t bits = 0;
#pragma unroll 1
  for (int k = 0; k < 5; ++k) {
    const int p = k == 0 ? 2 : k == 1 ? 0 : k == 2 ? 1 : k;             // chain order: pk_com, H, Gamma, R, Ok
    const Enc33 e = load33(p == 0 ? a.h : p == 1 ? a.gamma : p == 2 ? a.pk_com : p == 3 ? a.r : a.ok, i);
    bits |= bsw_in_a(scr + k * BSW_SLOT, run, e, a.T.sq) << (2 * k);
    const Enc33 ce = enc33_canonical(e);
#pragma unroll
    for (int q = 0; q < 5; ++q)
      if (q == p) enc[q] = ce;
  }
  FeN inv = fe_inv(run);
#pragma unroll 1
  for (int k = 4; k >= 0; --k) {
    const int p = k == 0 ? 2 : k == 1 ? 0 : k == 2 ? 1 : k;
    const uint32_t st = (bits >> (2 * k)) & 3u;
    FeN x, y;
    bsw_in_b(x, y, inv, scr + k * BSW_SLOT, (st & BSW_A_INF) != 0);
    valid = valid && (st & BSW_A_OK) != 0;
    // check_mask: H is an input, Gamma an output, pk_com / R / Ok proof points
    if (a.check_mask & (p == 0 ? CHK_INPUT : p == 1 ? CHK_OUTPUT : CHK_PROOF)) valid = in_prime_subgroup<BswS>(x, y, a.T.sq) && valid;
    if (p < 3) {
      build_glv_tables<BswS>(tabs + p * 2 * WIN_TABLE_WORDS, x, y);
    } else {
      uint32_t* dst = pts + (p == 3 ? PED_R_OFF : PED_OK_OFF);
      fe_store(dst, x);
      fe_store(dst + NL, y);
    }
  }
  const uint8_t* ad; uint32_t ad_len;
  bytes_get(a.ad, i, ad, ad_len);
  const Enc33 cp[5] = {enc[2], enc[0], enc[1], enc[3], enc[4]};      // pk_com, H, Gamma, R, Ok
  uint32_t c[8];
  bsw_challenge5(c, cp, ad, ad_len, a.T.sq.str);
  uint32_t* aux = a.ws.aux + i * AUX_WORDS;
#pragma unroll
  for (int j = 0; j < 8; ++j) aux[j] = c[j];
  a.ws.flags[i] = valid ? 1 : 0;
}

// HALF 0: s*H - c*Gamma ; HALF 1: s*G - c*pk_com + sb*B
template <int HALF>
__global__ void __launch_bounds__(BLOCK) k_bsw_ped_verify_straus(PedersenVerifyArgs a) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= a.n) return;
  uint32_t c[8], s[8], sb[8];
  load32(s, a.s, i); load32(sb, a.sb, i);
#pragma unroll
  for (int j = 0; j < 8; ++j) c[j] = a.ws.aux[i * AUX_WORDS + j];
  if (!fr_is_canonical<BswS>(s) || !fr_is_canonical<BswS>(sb)) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[j] = 0; sb[j] = 0; }
  }
  pedersen_verify_straus_item<BswS, HALF>(a.ws.pts + i * PROVE_PTS_WORDS + HALF * UV_WORDS, a.T,
                                          a.ws.tabs + i * (VERIFY_TABS * WIN_TABLE_WORDS), c, s, sb);
}

__global__ void __launch_bounds__(BLOCK) k_bsw_ped_verify_finish(PedersenVerifyArgs a) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= a.n) return;
  uint32_t s[8], sb[8];
  load32(s, a.s, i); load32(sb, a.sb, i);
  a.status[i] = (uint8_t)pedersen_verify_finish_item<BswS>(a.ws.pts + i * PROVE_PTS_WORDS, s, sb, a.ws.flags[i] != 0);
}

// ---- batched Pedersen verification (random linear combination, k_rlc.hip's scheme): the decode stage for this codec ----
// One lane per proof: five decodes into the MSM layout (Montgomery affine-cached Edwards points), the challenge over the
// canonical 33-byte encodings, weights, scalars and window digits; the two fixed-base scalars as 64-bit limb columns.  The
// weights are k_rlc.hip's (rlc_weights: seed, digest of the launch group's wire bytes, index).
__global__ void __launch_bounds__(BLOCK, 2) k_bsw_rlc_decode(RlcArgs a) {
  const size_t item = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  const size_t n = a.n, N = a.L.n;
  uint64_t cols[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) cols[j] = 0;
  if (item < n) {
    Enc33 enc[5];
    bool valid = true;
    uint32_t* scr = a.scratch + item * (size_t)a.scratch_stride;        // 5 slots of 36 words (bsw_core.cuh)
    FeN run = fe_one();
    uint32_t bits = 0;
#pragma unroll 1
    for (int p = 0; p < 5; ++p) {
      const uint8_t* src = p == 0 ? a.h : p == 1 ? a.gamma : p == 2 ? a.pk_com : p == 3 ? a.r : a.ok;
      Enc33 ce;
      uint32_t st;
      if (a.affine_in) {                          // Weierstrass x || y: the encoding comes out canonical
        st = bsw_in_a_xy(scr + p * BSW_SLOT, run, ce, src, item, a.affine_in == 2);
      } else {
        const Enc33 e = load33(src, item);
        st = bsw_in_a(scr + p * BSW_SLOT, run, e, a.T.sq);
        ce = enc33_canonical(e);
      }
      bits |= st << (2 * p);
#pragma unroll
      for (int q = 0; q < 5; ++q)
        if (q == p) enc[q] = ce;
    }
    FeN inv = fe_inv(run);                        // one inversion for the proof's five points
#pragma unroll 1
    for (int p = 4; p >= 0; --p) {
      const uint32_t st = (bits >> (2 * p)) & 3u;
      PtA pa;
      bsw_in_b(pa.x, pa.y, inv, scr + p * BSW_SLOT, (st & BSW_A_INF) != 0);
      bool ok = (st & BSW_A_OK) != 0;
      if (a.check_mask & (p == 0 ? CHK_INPUT : p == 1 ? CHK_OUTPUT : CHK_PROOF)) ok = in_prime_subgroup<BswS>(pa.x, pa.y, a.T.sq) && ok;
      valid = valid && ok;
      pa.dt = fe_mul(fe_mul(pa.x, pa.y), BswS::d());
      pta_store(a.L.pts + rlc_index(p, n, item) * MSM_PTA_STRIDE, pa);
    }
    uint32_t s[8], sb[8];
    load32(s, a.s, item); load32(sb, a.sb, item);
    valid = valid && fr_is_canonical<BswS>(s) && fr_is_canonical<BswS>(sb);
    if (!valid) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { s[j] = 0; sb[j] = 0; }
    }
    const uint8_t* adp; uint32_t adl;
    bytes_get(a.ad, item, adp, adl);
    const Enc33 cp[5] = {enc[2], enc[0], enc[1], enc[3], enc[4]};      // pk_com, H, Gamma, R, Ok
    uint32_t c[8], z[8], zp[8], t[8];
    bsw_challenge5(c, cp, adp, adl, a.T.sq.str);
    rlc_weights<BswS>(z, zp, a.seed, a.root, a.index0 + item);
    fr_mul<BswS>(t, z, s);
    msm_write_digits<BswS>(a.L.digits, N, rlc_index(0, n, item), t, false, !valid);     // + (z s) H
    fr_mul<BswS>(t, z, c);
    msm_write_digits<BswS>(a.L.digits, N, rlc_index(1, n, item), t, true, !valid);      // - (z c) Gamma
    fr_mul<BswS>(t, zp, c);
    msm_write_digits<BswS>(a.L.digits, N, rlc_index(2, n, item), t, true, !valid);      // - (z' c) pk_com
    msm_write_digits<BswS>(a.L.digits, N, rlc_index(3, n, item), zp, true, !valid);     // - z' R   (128 bits)
    msm_write_digits<BswS>(a.L.digits, N, rlc_index(4, n, item), z, true, !valid);      // - z Ok   (128 bits)
    fr_mul<BswS>(t, zp, s);
#pragma unroll
    for (int j = 0; j < 8; ++j) cols[j] += t[j];
    fr_mul<BswS>(t, zp, sb);
#pragma unroll
    for (int j = 0; j < 8; ++j) cols[8 + j] += t[j];
    a.status[item] = (uint8_t)(valid ? ST_OK : ST_INVALID_DATA);
  }
  // wave reduction of the fixed-base limb columns, one atomic per column and wave
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    uint64_t v = cols[j];
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) v += __shfl_down(v, sft, 64);
    if ((threadIdx.x & 63) == 0 && v != 0)
      atomicAdd(reinterpret_cast<unsigned long long*>(a.fixed_cols) + j, (unsigned long long)v);
  }
}
// one lane: columns -> scalars mod r; G and B (entry 1 * 256^0 of the fixed-base combs) become points 3n, 3n + 1
__global__ void k_bsw_rlc_fixed(RlcArgs a) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const size_t N = a.L.n;
#pragma unroll 1
  for (int f = 0; f < 2; ++f) {
    uint32_t wide[16];
    uint64_t carry = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      uint64_t acc = (j < 8 ? a.fixed_cols[f * 8 + j] : 0) + carry;   // columns < 2^52, carry < 2^32
      wide[j] = (uint32_t)acc;
      carry = acc >> 32;
    }
    uint32_t k[8];
    fr_reduce512<BswS>(k, wide);
    const uint32_t* src = f == 0 ? a.T.g_comb : a.T.b_comb;
    uint32_t* dst = a.L.pts + (3 * a.n + f) * MSM_PTA_STRIDE;
    for (int j = 0; j < PTA_WORDS; ++j) dst[j] = src[j];
    msm_write_digits<BswS>(a.L.digits, N, 3 * a.n + f, k, false, false);
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------------- launchers
void launch_bsw_secret_from_seed(size_t n, const uint8_t* seeds, uint32_t seed_len, uint8_t* sk, uint8_t* pk, DevTables T,
                                 hipStream_t st) {
  if (n) hipLaunchKernelGGL(k_bsw_secret_from_seed, grid_for(n), dim3(BLOCK), 0, st, n, seeds, seed_len, sk, pk, T);
}
// tai_ctr ([n] bytes) + queue: scratch of the counter search (tai_find.cuh); without them each lane loops on its own item
void launch_bsw_hash_to_curve(size_t n, BytesView msg, uint8_t* points, DevTables T, hipStream_t st, uint8_t* tai_ctr,
                              unsigned long long* queue) {
  if (!n) return;
  const uint8_t* ctr = nullptr;
  if (tai_ctr && queue) { launch_tai_find_t<SuiteBW>(n, msg, tai_ctr, T.sq, queue, st); ctr = tai_ctr; }
  hipLaunchKernelGGL(k_bsw_hash_to_curve, grid_for(n), dim3(BLOCK), 0, st, n, msg, points, T, ctr);
}
void launch_bsw_output_hash(size_t n, const uint8_t* gamma, uint8_t* hash, DevTables T, hipStream_t st) {
  if (n) hipLaunchKernelGGL(k_bsw_output_hash, grid_for(n), dim3(BLOCK), 0, st, n, gamma, hash, T.sq);
}
void launch_bsw_point_validate(size_t n, const uint8_t* pts, uint8_t* xy, uint8_t* status, DevTables T, hipStream_t st) {
  if (n) hipLaunchKernelGGL(k_bsw_point_validate, grid_for(n), dim3(BLOCK), 0, st, n, pts, xy, status, T);
}
// IETF and (a.pedersen) Pedersen proving.  ev: as launch_ietf_prove
void launch_bsw_prove(const ProveArgs& a, hipStream_t st, hipEvent_t* ev) {
  if (a.n == 0) return;
  if (ev) (void)hipEventRecord(ev[0], st);
  if (!a.h_given) launch_tai_find_t<SuiteBW>(a.n, a.msg, a.ws.flags, a.T.sq, a.tai_queue, st);
  hipLaunchKernelGGL(k_bsw_prove_prepare, grid_for(a.n), dim3(BLOCK), 0, st, a);
  if (ev) (void)hipEventRecord(ev[1], st);
  launch_prove_stage2_bs(a, st);
  if (ev) { (void)hipEventRecord(ev[2], st); (void)hipEventRecord(ev[3], st); }
  hipLaunchKernelGGL(k_bsw_prove_finish, grid_for(a.n), dim3(BLOCK), 0, st, a);
  if (ev) (void)hipEventRecord(ev[4], st);
}
void launch_bsw_ietf_verify(const VerifyArgs& a, hipStream_t st, hipEvent_t* ev) {
  if (a.n == 0) return;
  if (ev) (void)hipEventRecord(ev[0], st);
  hipLaunchKernelGGL(k_bsw_verify_decode, grid_for(a.n), dim3(BLOCK), 0, st, a);
  if (ev) (void)hipEventRecord(ev[1], st);
  hipLaunchKernelGGL(k_bsw_verify_straus<1>, grid_for(a.n), dim3(BLOCK), 0, st, a);
  if (ev) (void)hipEventRecord(ev[2], st);
  if (a.key_index) hipLaunchKernelGGL(k_bsw_verify_comb_u, grid_for(a.n), dim3(BLOCK), 0, st, a);
  else hipLaunchKernelGGL(k_bsw_verify_straus<0>, grid_for(a.n), dim3(BLOCK), 0, st, a);
  if (ev) (void)hipEventRecord(ev[3], st);
  hipLaunchKernelGGL(k_bsw_verify_finish, grid_for(a.n), dim3(BLOCK), 0, st, a);
  if (ev) (void)hipEventRecord(ev[4], st);
}
// pks: [n_keys][33] device memory, canonicalised in place; xy: [n_keys][18] words, prefix: [n_keys * 32][255][9] words of scratch
void launch_bsw_keyset_build(size_t n_keys, uint8_t* pks, uint32_t* xy, uint8_t* valid, uint32_t* combs, uint32_t* prefix, DevTables T,
                             hipStream_t st) {
  if (!n_keys) return;
  hipLaunchKernelGGL(k_bsw_keyset_decode, grid_for(n_keys), dim3(BLOCK), 0, st, n_keys, pks, xy, valid, T);
  hipLaunchKernelGGL(k_bsw_keyset_comb, dim3((unsigned)((n_keys * COMB_ROWS + 63) / 64)), dim3(64), 0, st, n_keys, xy, combs, prefix);
}
// after launch_te_sw_map (bases -> Edwards) and launch_msm_coords (the Edwards sum te_xy, status[0])
void launch_bsw_msm_out(const uint8_t* te_xy, uint8_t* out33, uint8_t* out_xy, uint8_t* status, hipStream_t st) {
  hipLaunchKernelGGL(k_bsw_msm_out, dim3(1), dim3(64), 0, st, te_xy, out33, out_xy, status);
}
void launch_bsw_pedersen_verify(const PedersenVerifyArgs& a, hipStream_t st, hipEvent_t* ev) {
  if (a.n == 0) return;
  if (ev) (void)hipEventRecord(ev[0], st);
  hipLaunchKernelGGL(k_bsw_ped_verify_decode, grid_for(a.n), dim3(BLOCK), 0, st, a);
  if (ev) (void)hipEventRecord(ev[1], st);
  hipLaunchKernelGGL(k_bsw_ped_verify_straus<0>, grid_for(a.n), dim3(BLOCK), 0, st, a);
  if (ev) (void)hipEventRecord(ev[2], st);
  hipLaunchKernelGGL(k_bsw_ped_verify_straus<1>, grid_for(a.n), dim3(BLOCK), 0, st, a);
  if (ev) (void)hipEventRecord(ev[3], st);
  hipLaunchKernelGGL(k_bsw_ped_verify_finish, grid_for(a.n), dim3(BLOCK), 0, st, a);
  if (ev) (void)hipEventRecord(ev[4], st);
}
// n x 64 B Weierstrass x || y -> n x 33 B encodings (a failed x || y batch falls back to the per-proof kernels); a coordinate
// >= q gives an undecodable string (both flags)
__global__ void __launch_bounds__(BLOCK) k_bsw_affine_compress(size_t n, const uint8_t* xy, uint8_t* enc, int mont256) {
  const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  const uint32_t* w = reinterpret_cast<const uint32_t*>(xy + i * 64);
  Enc33 e;
  uint32_t xin[8], yin[8], yw[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { xin[j] = w[j]; yin[j] = w[8 + j]; }
  const bool bad = u256_ge(xin, vrfk::Q32) || u256_ge(yin, vrfk::Q32);
  (void)fe_from_abi(e.w, xin, mont256 != 0);         // e.w, yw: the canonical words
  (void)fe_from_abi(yw, yin, mont256 != 0);
  e.fl = bad ? 0xC0u : (u256_gt(yw, vrfk::QM1H32) ? BSW_NEG : 0u);
  store33(enc, i, e);
}
void launch_bsw_affine_compress(size_t n, const uint8_t* xy, int mont256, uint8_t* enc, hipStream_t st) {
  if (n) hipLaunchKernelGGL(k_bsw_affine_compress, grid_for(n), dim3(BLOCK), 0, st, n, xy, enc, mont256);
}
// enqueues decode + MSM (k_msm.hip's bucket and final kernels); fail_flag[0] becomes 1 if the batch equation does not hold.
// a.affine_in: the five point arrays are Weierstrass x || y; a.scratch: >= 180 words per proof; a.k_lane is not read.  ev (nullable, 5 events): start | decode | buckets | final | final
void launch_bsw_pedersen_rlc(const RlcArgs& a, uint8_t* fail_flag, hipStream_t st, hipEvent_t* ev) {
  if (a.n == 0) return;
  (void)hipMemsetAsync(a.L.flags, 0, 256, st);
  (void)hipMemsetAsync(a.fixed_cols, 0, 16 * sizeof(uint64_t), st);
  if (ev) (void)hipEventRecord(ev[0], st);
  hipLaunchKernelGGL(k_bsw_rlc_decode, grid_for(a.n), dim3(BLOCK), 0, st, a);
  hipLaunchKernelGGL(k_bsw_rlc_fixed, dim3(1), dim3(64), 0, st, a);
  if (ev) (void)hipEventRecord(ev[1], st);
  launch_msm_core(SUITE_BS, a.L, nullptr, nullptr, nullptr, fail_flag, st, ev ? ev + 2 : nullptr);
}

VRF_NS_END
