// k_misc.hip -- table construction and the small building-block kernels.
#include "kernels.h"
#include "te_sw_map.cuh"
#include "tai_find.cuh"

VRF_NS_BEGIN
#include "te_sw_map.inc"

// ---- one-time table construction (context creation) ----
// gb_xy: the descriptor's generator and Pedersen blinding base, x || y as 32-byte little-endian canonical
// integers (2 x 64 bytes).  k_init_bases validates them (coordinates < q, on the curve, prime-order subgroup,
// not the identity: flags[which] = 1) and leaves their Montgomery coordinates in mont[which][2][9].
template <class S>
__global__ void k_init_bases(const uint8_t* gb_xy, uint32_t* mont, uint8_t* flags, SqrtTables T) {
  const int which = threadIdx.x;
  if (blockIdx.x != 0 || which >= 2) return;
  uint32_t xw[8], yw[8];
  load32(xw, gb_xy, 2 * which); load32(yw, gb_xy, 2 * which + 1);
  bool ok = !u256_ge(xw, vrfk::Q32) && !u256_ge(yw, vrfk::Q32);
  const FeN x = fe_from_u256(xw), y = fe_from_u256(yw);
  // a x^2 + y^2 = 1 + d x^2 y^2   <=>   y^2 - ANEG x^2 - 1 = d (x y)^2
  const FeN x2 = fe_sqr(x), y2 = fe_sqr(y), xyv = fe_mul(x, y);
  auto lhs = te_curve_lhs<S>(x2, y2);
  ok = ok && fe_eq(lhs, fe_mul(fe_sqr(xyv), S::d())) && !fe_is_zero(x);
  uint32_t r[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = S::r32(j);
  const PtE rp = te_mul_slow<S>(te_from_affine(x, y), r);          // one-time: r * P = O by the plain ladder
  ok = ok && fe_is_zero(rp.X) && fe_eq(rp.Y, rp.Z);
  fe_store(mont + which * 2 * NL, x);
  fe_store(mont + which * 2 * NL + NL, y);
  flags[which] = ok ? 1 : 0;
}
template <class S>
__global__ void k_init_gwin(uint32_t* g_win, const uint32_t* mont) {
  if (blockIdx.x == 0 && threadIdx.x == 0) build_glv_tables<S>(g_win, fe_load<1, 2>(mont), fe_load<1, 2>(mont + NL));
}
// one lane per (base, row, segment) of the two generator tables (gcomb_build_segment); prefix: GC_SEG x 9 words per lane
template <class S>
__global__ void __launch_bounds__(64, 2) k_init_gcomb(uint32_t* g_comb, uint32_t* b_comb, uint32_t* prefix,
                                                      const uint32_t* mont) {
  const int t = blockIdx.x * 64 + threadIdx.x;
  if (t >= 2 * GC_ROWS * GC_SEGS) return;
  const int which = t / (GC_ROWS * GC_SEGS), r = t % (GC_ROWS * GC_SEGS);
  const int w = r / GC_SEGS, seg = r % GC_SEGS;
  const FeN bx = fe_load<1, 2>(mont + which * 2 * NL), by = fe_load<1, 2>(mont + which * 2 * NL + NL);
  gcomb_build_segment<S>(which ? b_comb : g_comb, prefix + (size_t)t * GC_SEG * NL, bx, by, w, seg);
}
template <class S>
static void init_tables_t(uint32_t* g_win, uint32_t* g_comb, uint32_t* b_comb, uint32_t* prefix, const uint8_t* gb_xy,
                          uint32_t* mont, uint8_t* flags, SqrtTables T, hipStream_t st) {
  hipLaunchKernelGGL(k_init_bases<S>, dim3(1), dim3(64), 0, st, gb_xy, mont, flags, T);
  hipLaunchKernelGGL(k_init_gwin<S>, dim3(1), dim3(64), 0, st, g_win, (const uint32_t*)mont);
  hipLaunchKernelGGL(k_init_gcomb<S>, dim3((2 * GC_ROWS * GC_SEGS + 63) / 64), dim3(64), 0, st, g_comb, b_comb, prefix,
                     (const uint32_t*)mont);
}
void launch_init_tables(int suite, uint32_t* g_win, uint32_t* g_comb, uint32_t* b_comb, uint32_t* prefix,
                        const uint8_t* gb_xy, uint32_t* mont, uint8_t* flags, SqrtTables T, hipStream_t st) {
  VRF_DISPATCH_SUITE(suite, init_tables_t<S>(g_win, g_comb, b_comb, prefix, gb_xy, mont, flags, T, st));
}

// ---- Input::new ----
template <class S>
__global__ void __launch_bounds__(BLOCK, 2) k_hash_to_curve(size_t n, BytesView msg, uint8_t* points, DevTables T,
                                                             const uint8_t* tai_ctr) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  const uint8_t* m; uint32_t len;
  bytes_get(msg, i, m, len);
  PtE h = data_to_point<S>(m, len, T.sq, tai_ctr ? tai_ctr[i] : 0u);      // try-and-increment: k_tai_find's counter
  FeN x, y;
  te_to_affine(x, y, h);
  uint32_t e[8];
  te_encode_affine(e, x, y, T.sq.str.flags);
  store32(points, i, e);
}
// tai_ctr ([n] bytes) + queue: scratch for the try-and-increment suites' counter search (tai_find.cuh); without them every
// lane loops until its own item succeeds and a wave pays for its unluckiest lane
void launch_hash_to_curve(int suite, size_t n, BytesView msg, uint8_t* points, DevTables T, hipStream_t st, uint8_t* tai_ctr,
                          unsigned long long* queue) {
  if (!n) return;
  VRF_DISPATCH_SUITE(suite, {
    const uint8_t* ctr = nullptr;
    if constexpr (!S::H2C_ELL2) {
      if (tai_ctr && queue) { launch_tai_find_t<S>(n, msg, tai_ctr, T.sq, queue, st); ctr = tai_ctr; }
    }
    hipLaunchKernelGGL(k_hash_to_curve<S>, grid_for(n), dim3(BLOCK), 0, st, n, msg, points, T, ctr);
  });
}

// ---- Output::hash ----
template <class S>
__global__ void __launch_bounds__(BLOCK) k_output_hash(size_t n, const uint8_t* gamma, uint8_t* hash, SqrtTables T) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t g[8], o[16];
  load32(g, gamma, i);
  enc_canonical(g);                              // `Output::hash` encodes the typed point
  bool ok = true;
  if (T.str.flags & SS_HASH_COFACTOR) {          // RFC 9381 proof_to_hash: cofactor * Gamma (wave-uniform branch)
    uint32_t e[8];
    ok = output_cofactor_encoding<S>(e, g, T);
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = e[j];
  }
  output_hash_item<S>(o, g, T.str);
  if (!ok) {
#pragma unroll
    for (int j = 0; j < 16; ++j) o[j] = 0;       // an output that does not decode has no hash
  }
  uint32_t* p = reinterpret_cast<uint32_t*>(hash + i * 64);
#pragma unroll
  for (int j = 0; j < 16; ++j) p[j] = o[j];
}
void launch_output_hash(int suite, size_t n, const uint8_t* gamma, uint8_t* hash, DevTables T, hipStream_t st) {
  if (n) VRF_DISPATCH_SUITE(suite, hipLaunchKernelGGL(k_output_hash<S>, grid_for(n), dim3(BLOCK), 0, st, n, gamma, hash, T.sq));
}

// ---- Secret::from_seed / Secret::public ----
template <class S>
__global__ void __launch_bounds__(BLOCK, 2) k_secret_from_seed(size_t n, const uint8_t* seeds, uint32_t seed_len,
                                                             uint8_t* sk_out, uint8_t* pk_out, DevTables T) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t sk[8];
  secret_from_seed_item<S>(sk, seeds + i * (size_t)seed_len, seed_len);
  store32(sk_out, i, sk);
  if (pk_out) {
    uint32_t pk[8];
    public_from_secret_item<S>(pk, T, sk);
    store32(pk_out, i, pk);
  }
}
void launch_secret_from_seed(int suite, size_t n, const uint8_t* seeds, uint32_t seed_len, uint8_t* sk,
                             uint8_t* pk, DevTables T, hipStream_t st) {
  if (n) VRF_DISPATCH_SUITE(suite, hipLaunchKernelGGL(k_secret_from_seed<S>, grid_for(n), dim3(BLOCK), 0, st, n, seeds, seed_len, sk, pk, T));
}

// ---- codec: checked point decoding (on curve + prime-order subgroup) ----
template <class S>
__global__ void __launch_bounds__(BLOCK) k_point_validate(size_t n, const uint8_t* pts, uint8_t* xy,
                                                           uint8_t* status, uint32_t* tabs, DevTables T) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t e[8];
  load32(e, pts, i);
  DecodeA a = decode_phase_a<S>(e);
  FeN di = fe_inv(a.den);
  Fe<1, 4> x;
  bool ok = decode_phase_b<S>(x, a, di, T.sq);
  ok = ok && in_prime_subgroup<S>(fe_mul(x, fe_one()), a.y, T.sq);
  status[i] = ok ? ST_OK : ST_INVALID_DATA;
  if (xy) {
    uint32_t xw[8], yw[8];
    fe_to_u256(xw, x);
    fe_to_u256(yw, a.y);
    if (!ok) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { xw[j] = 0; yw[j] = 0; }
    }
    uint32_t* p = reinterpret_cast<uint32_t*>(xy + i * 64);
#pragma unroll
    for (int j = 0; j < 8; ++j) { p[j] = xw[j]; p[8 + j] = yw[j]; }
  }
}
void launch_point_validate(int suite, size_t n, const uint8_t* pts, uint8_t* xy, uint8_t* status,
                           uint32_t* tabs, DevTables T, hipStream_t st) {
  if (n) VRF_DISPATCH_SUITE(suite, hipLaunchKernelGGL(k_point_validate<S>, grid_for(n), dim3(BLOCK), 0, st, n, pts, xy, status, tabs, T));
}

// ---- `utils::te_sw_map`: twisted-Edwards <-> short-Weierstrass, x || y in, x || y out (te_sw_map.cuh) ----
template <class S>
__global__ void __launch_bounds__(BLOCK) k_te_sw_map(size_t n, int to_te, int mont256, const uint8_t* in_xy, uint8_t* out_xy,
                                                      uint8_t* status) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  const uint32_t* p = reinterpret_cast<const uint32_t*>(in_xy + i * 64);
  uint32_t xin[8], yin[8], cw[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { xin[j] = p[j]; yin[j] = p[8 + j]; }
  bool ok = !u256_ge(xin, vrfk::Q32) && !u256_ge(yin, vrfk::Q32);
  const FeN x = fe_from_abi(cw, xin, mont256 != 0), y = fe_from_abi(cw, yin, mont256 != 0);
  FeN ox, oy;
  if (to_te) ok = sw_to_te<S>(ox, oy, x, y) && ok;          // wave-uniform
  else ok = te_to_sw<S>(ox, oy, x, y) && ok;
  uint32_t xw[8], yw[8];
  if (mont256) { fe_to_mont256(xw, ox); fe_to_mont256(yw, oy); }
  else { fe_to_u256(xw, ox); fe_to_u256(yw, oy); }
  uint32_t* o = reinterpret_cast<uint32_t*>(out_xy + i * 64);
#pragma unroll
  for (int j = 0; j < 8; ++j) { o[j] = ok ? xw[j] : 0u; o[8 + j] = ok ? yw[j] : 0u; }
  status[i] = ok ? ST_OK : ST_INVALID_DATA;
}
void launch_te_sw_map(int suite, size_t n, int to_te, int mont256, const uint8_t* in_xy, uint8_t* out_xy, uint8_t* status,
                      hipStream_t st) {
  if (n) VRF_DISPATCH_SUITE(suite, hipLaunchKernelGGL(k_te_sw_map<S>, grid_for(n), dim3(BLOCK), 0, st, n, to_te, mont256, in_xy, out_xy, status));
}

// ---- key sets (keyed verification): validated keys and their fixed-base combs, context resident ----
template <class S>
__global__ void __launch_bounds__(BLOCK) k_keyset_decode(size_t n, const uint8_t* pks, uint32_t* xy, uint8_t* valid,
                                                          DevTables T) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t e[8];
  load32(e, pks, i);
  DecodeA a = decode_phase_a<S>(e);
  FeN di = fe_inv(a.den);
  Fe<1, 4> x;
  bool ok = decode_phase_b<S>(x, a, di, T.sq);
  FeN xn = fe_mul(x, fe_one());
  ok = ok && in_prime_subgroup<S>(xn, a.y, T.sq);
  // an invalid key gets the identity's tables: every proof that names it is reported InvalidData anyway
  FeN ky = fe_select(ok, a.y, fe_one());
  xn = fe_select(ok, xn, fe_zero());
  fe_store(xy + i * 2 * NL, xn);
  fe_store(xy + i * 2 * NL + NL, ky);
  valid[i] = ok ? 1 : 0;
}
template <class S>
__global__ void __launch_bounds__(64, 2) k_keyset_comb(size_t n_keys, const uint32_t* xy, uint32_t* combs, uint32_t* prefix) {
  size_t t = (size_t)blockIdx.x * 64 + threadIdx.x;           // one lane per (key, row)
  if (t >= n_keys * COMB_ROWS) return;
  const size_t key = t / COMB_ROWS;
  const int w = (int)(t % COMB_ROWS);
  const FeN x = fe_load<1, 2>(xy + key * 2 * NL), y = fe_load<1, 2>(xy + key * 2 * NL + NL);
  comb_build_row<S>(combs + key * COMB_WORDS + (size_t)w * COMB_COLS * PTA_WORDS, prefix + t * COMB_COLS * NL, x, y, w);
}
void launch_keyset_build(int suite, size_t n_keys, const uint8_t* pks, uint32_t* xy, uint8_t* valid, uint32_t* combs,
                         uint32_t* prefix, DevTables T, hipStream_t st) {
  if (!n_keys) return;
  VRF_DISPATCH_SUITE(suite, {
    hipLaunchKernelGGL(k_keyset_decode<S>, grid_for(n_keys), dim3(BLOCK), 0, st, n_keys, pks, xy, valid, T);
    hipLaunchKernelGGL(k_keyset_comb<S>, dim3((unsigned)((n_keys * COMB_ROWS + 63) / 64)), dim3(64), 0, st, n_keys, xy,
                       combs, prefix);
  });
}

// ---- test primitives: the group law, scalar multiplication, SHA-512 and XMD on their own ----
// a + b of two decoded points (no subgroup test: the law itself is what is tested)
template <class S>
__global__ void __launch_bounds__(BLOCK) k_test_point_add(size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out,
                                                           uint8_t* status, DevTables T) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t ea[8], eb[8], e[8];
  load32(ea, a, i); load32(eb, b, i);
  DecodeA da = decode_phase_a<S>(ea), db = decode_phase_a<S>(eb);
  FeN dens[2] = {da.den, db.den}, dinv[2];
  fe_batch_inv(dinv, dens);
  Fe<1, 4> xa, xb;
  bool ok = decode_phase_b<S>(xa, da, dinv[0], T.sq);
  ok = decode_phase_b<S>(xb, db, dinv[1], T.sq) && ok;
  PtE r = te_add<S>(te_from_affine(xa, da.y), te_from_affine(xb, db.y));
  FeN x, y;
  te_to_affine(x, y, r);
  te_encode_affine(e, x, y, T.sq.str.flags);
  if (!ok) {
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = 0;
  }
  store32(out, i, e);
  status[i] = ok ? ST_OK : ST_INVALID_DATA;
}
// k * P by the variable-base path of the provers (GLV Straus on Bandersnatch, 253-bit windows on JubJub)
template <class S>
__global__ void __launch_bounds__(BLOCK) k_test_scalar_mul(size_t n, const uint8_t* k, const uint8_t* p, uint8_t* out,
                                                            uint8_t* status, uint32_t* tabs, DevTables T) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t kw[8], ep[8], e[8];
  load32(kw, k, i); load32(ep, p, i);
  DecodeA d = decode_phase_a<S>(ep);
  FeN di = fe_inv(d.den);
  Fe<1, 4> x;
  bool ok = decode_phase_b<S>(x, d, di, T.sq) && fr_is_canonical<S>(kw);
  if (!fr_is_canonical<S>(kw)) {
#pragma unroll
    for (int j = 0; j < 8; ++j) kw[j] = 0;
  }
  uint32_t* tab = tabs + i * (2 * WIN_TABLE_WORDS);
  build_glv_tables<S>(tab, x, d.y);
  PtE r = var_base_mul<S>(tab, kw);
  FeN rx, ry;
  te_to_affine(rx, ry, r);
  te_encode_affine(e, rx, ry, T.sq.str.flags);
  if (!ok) {
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = 0;
  }
  store32(out, i, e);
  status[i] = ok ? ST_OK : ST_INVALID_DATA;
}
// which = 0: SHA-512 (64 B per item); which = 1: expand_message_xmd to 96 bytes with the context's DST
template <class S>
__global__ void __launch_bounds__(BLOCK) k_test_hash(size_t n, BytesView msg, uint8_t* out, int which, SuiteStr ss) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  const uint8_t* m; uint32_t len;
  bytes_get(msg, i, m, len);
  if (which == 0) {
    Sha512 h;
    sha512_init(h);
    sha512_put_bytes(h, m, len);
    sha512_final(h);
    uint32_t* o = reinterpret_cast<uint32_t*>(out + i * 64);
#pragma unroll
    for (int j = 0; j < 16; ++j) o[j] = sha512_word_mem(h, j);
  } else {
    uint64_t hb[2][8];
    expand_message_xmd96<S>(hb, m, len, ss);
    uint32_t* o = reinterpret_cast<uint32_t*>(out + i * 96);
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const uint64_t w = j < 8 ? hb[0][j] : hb[1][j - 8];
      o[2 * j] = bswap32((uint32_t)(w >> 32));
      o[2 * j + 1] = bswap32((uint32_t)w);
    }
  }
}
void launch_test_point_add(int suite, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out, uint8_t* status,
                           DevTables T, hipStream_t st) {
  if (n) VRF_DISPATCH_SUITE(suite, hipLaunchKernelGGL(k_test_point_add<S>, grid_for(n), dim3(BLOCK), 0, st, n, a, b, out, status, T));
}
void launch_test_scalar_mul(int suite, size_t n, const uint8_t* k, const uint8_t* p, uint8_t* out, uint8_t* status,
                            uint32_t* tabs, DevTables T, hipStream_t st) {
  if (n) VRF_DISPATCH_SUITE(suite, hipLaunchKernelGGL(k_test_scalar_mul<S>, grid_for(n), dim3(BLOCK), 0, st, n, k, p, out, status, tabs, T));
}
void launch_test_hash(int suite, size_t n, BytesView msg, uint8_t* out, int which, DevTables T, hipStream_t st) {
  if (n) VRF_DISPATCH_SUITE(suite, hipLaunchKernelGGL(k_test_hash<S>, grid_for(n), dim3(BLOCK), 0, st, n, msg, out, which, T.sq.str));
}

// ---- test primitive: Fq multiplication ----
__global__ void __launch_bounds__(BLOCK) k_fq_mul(size_t n, const uint8_t* a, const uint8_t* b, uint8_t* r) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t x[8], y[8], z[8];
  load32(x, a, i); load32(y, b, i);
  fe_to_u256(z, fe_mul(fe_from_u256(x), fe_from_u256(y)));
  store32(r, i, z);
}
void launch_fq_mul(size_t n, const uint8_t* a, const uint8_t* b, uint8_t* r, hipStream_t st) {
  if (n) hipLaunchKernelGGL(k_fq_mul, grid_for(n), dim3(BLOCK), 0, st, n, a, b, r);
}

// ---- x || y outputs in arkworks' in-memory form (VRFHIP_FLAG_COORDS_MONT256): canonical words -> x 2^256 mod q, in place.
// An all-zero pair (a failed item, the neutral MSM result of an invalid input) stays all-zero.
__global__ void __launch_bounds__(BLOCK) k_xy_to_mont256(size_t n, uint8_t* xy) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t* p = reinterpret_cast<uint32_t*>(xy + i * 64);
  uint32_t xw[8], yw[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { xw[j] = p[j]; yw[j] = p[8 + j]; }
  u256_canon_to_mont256(xw);
  u256_canon_to_mont256(yw);
#pragma unroll
  for (int j = 0; j < 8; ++j) { p[j] = xw[j]; p[8 + j] = yw[j]; }
}
void launch_xy_to_mont256(size_t n, uint8_t* xy, hipStream_t st) {
  if (n && xy) hipLaunchKernelGGL(k_xy_to_mont256, grid_for(n), dim3(BLOCK), 0, st, n, xy);
}
// the other direction, in place (words >= q come out reduced: the callers have rejected those items already)
__global__ void __launch_bounds__(BLOCK) k_xy_from_mont256(size_t n, uint8_t* xy) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t* p = reinterpret_cast<uint32_t*>(xy + i * 64);
  uint32_t xw[8], yw[8], cx[8], cy[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { xw[j] = p[j]; yw[j] = p[8 + j]; }
  (void)fe_from_abi(cx, xw, true);
  (void)fe_from_abi(cy, yw, true);
#pragma unroll
  for (int j = 0; j < 8; ++j) { p[j] = cx[j]; p[8 + j] = cy[j]; }
}
void launch_xy_from_mont256(size_t n, uint8_t* xy, hipStream_t st) {
  if (n && xy) hipLaunchKernelGGL(k_xy_from_mont256, grid_for(n), dim3(BLOCK), 0, st, n, xy);
}

// ---- host copies of this field's tables and built-in descriptor points, for the C ABI (api.hip) ----
const uint32_t* field_sqrt_p(size_t* bytes) { *bytes = sizeof(vrfk_tables::SQRT_P); return vrfk_tables::SQRT_P; }
const uint8_t* field_sqrt_lut(size_t* bytes) { *bytes = sizeof(vrfk_tables::SQRT_LUT); return vrfk_tables::SQRT_LUT; }
bool field_default_points(int suite, uint8_t g_xy[64], uint8_t b_xy[64]) {
  const uint8_t *g = nullptr, *b = nullptr;
#if VRF_FIELD == 0
  if (suite == SUITE_BS) { g = vrfk_tables::BS_G_XY; b = vrfk_tables::BS_B_XY; }
  if (suite == SUITE_JJ) { g = vrfk_tables::JJ_G_XY; b = vrfk_tables::JJ_B_XY; }
#elif VRF_FIELD == 1
  if (suite == SUITE_ED) { g = vrfk_tables::ED_G_XY; b = vrfk_tables::ED_B_XY; }
#else
  if (suite == SUITE_BJ) { g = vrfk_tables::BJ_G_XY; b = vrfk_tables::BJ_B_XY; }
#endif
  if (!g) return false;
  for (int i = 0; i < 64; ++i) { g_xy[i] = g[i]; b_xy[i] = b[i]; }
  return true;
}

VRF_NS_END
