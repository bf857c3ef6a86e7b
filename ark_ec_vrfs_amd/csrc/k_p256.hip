// k_p256.hip -- kernels of the secp256r1 suite ("P256_SHA256_TAI"; `suites::secp256r1`, /root/reference src/lib.rs:14):
// IETF and Pedersen prove / verify in three stages each, hash-to-curve, output hash, key derivation, point validation.
// Compiled once, with -DVRF_FIELD=3 (csrc/field.h): the NIST P-256 prime under the same 9 x 29-bit typed limbs as the
// Edwards suites' fields, with the short-Weierstrass law of sw.cuh in place of te.cuh.
//
// Shape: one item per lane.  The stages are separate kernels because their register needs differ by a factor of three
// (the ladders hold a projective accumulator, a table entry and the temporaries of the complete addition; the hash
// stages hold a SHA-256 block); what passes between them lies word-major in the context's workspace (p256.h), so every
// inter-stage load and store is coalesced.  The ladders' window tables live there too, contiguous per item (224 words per
// table): written once and read 65 (+33) times, an entry at a time, by the lane that wrote them.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include "p256.h"
#include "p256_core.cuh"

VRF_NS_BEGIN
namespace {

constexpr int P256_BLOCK = 128;
// p256.h states the workspace record sizes as plain numbers (the C ABI's translation unit does not see the device headers)
static_assert(p256::WS_TAB_WORDS == 4 * SW_TABLE_WORDS && SW_QUAD_TABLES == 4 && p256::WS_PTS_WORDS == 4 * PTW_WORDS && p256::WS_AFF_WORDS == 10 * NL &&
              p256::WS_ENC_WORDS == 3 * 9, "p256.h workspace layout");

__device__ __forceinline__ void ws_store_fe(uint32_t* base, size_t cap, size_t i, int w0, const FeN& a) {
#pragma unroll
  for (int k = 0; k < NL; ++k) base[(size_t)(w0 + k) * cap + i] = a.v[k];
}
__device__ __forceinline__ FeN ws_load_fe(const uint32_t* base, size_t cap, size_t i, int w0) {
  FeN a;
#pragma unroll
  for (int k = 0; k < NL; ++k) a.v[k] = base[(size_t)(w0 + k) * cap + i];
  return a;
}
__device__ __forceinline__ void ws_store8(uint32_t* base, size_t cap, size_t i, int w0, const uint32_t (&a)[8]) {
#pragma unroll
  for (int k = 0; k < 8; ++k) base[(size_t)(w0 + k) * cap + i] = a[k];
}
__device__ __forceinline__ void ws_load8(uint32_t (&a)[8], const uint32_t* base, size_t cap, size_t i, int w0) {
#pragma unroll
  for (int k = 0; k < 8; ++k) a[k] = base[(size_t)(w0 + k) * cap + i];
}
__device__ __forceinline__ void ws_store_enc(uint32_t* base, size_t cap, size_t i, int slot, const Sec1W& e) {
  base[(size_t)(slot * 9) * cap + i] = e.tag;
#pragma unroll
  for (int k = 0; k < 8; ++k) base[(size_t)(slot * 9 + 1 + k) * cap + i] = e.xw[k];
}
__device__ __forceinline__ Sec1W ws_load_enc(const uint32_t* base, size_t cap, size_t i, int slot) {
  Sec1W e;
  e.tag = base[(size_t)(slot * 9) * cap + i];
#pragma unroll
  for (int k = 0; k < 8; ++k) e.xw[k] = base[(size_t)(slot * 9 + 1 + k) * cap + i];
  return e;
}
__device__ __forceinline__ uint32_t* ws_tab(uint32_t* tabs, size_t cap, size_t i, int slot) {
  return tabs + ((size_t)slot * cap + i) * SW_TABLE_WORDS;      // contiguous per item (sw.cuh)
}
// the prover's four tables of one base, contiguous per item (the same region, cut as [cap][4 x 224])
__device__ __forceinline__ uint32_t* ws_quad_tab(uint32_t* tabs, size_t i) {
  return tabs + i * (size_t)(SW_QUAD_TABLES * SW_TABLE_WORDS);
}
__device__ __forceinline__ uint32_t* ws_pt(uint32_t* pts, size_t cap, size_t i, int slot) {
  return pts + (size_t)slot * PTW_WORDS * cap + i;
}

// ------------------------------------------------------------------------------------------------ context tables
// One lane per row of the comb (33 rows of 128 entries: p256_core.cuh); lane 0 also reports whether the generator is a
// point of the curve.
__global__ void k_p256_init_comb(uint32_t* comb, const uint8_t* gen_xy, uint8_t* ok) {
  const int w = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (w >= P256_COMB_ROWS) return;
  uint32_t xw[8], yw[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    xw[j] = (uint32_t)gen_xy[4 * j] | ((uint32_t)gen_xy[4 * j + 1] << 8) | ((uint32_t)gen_xy[4 * j + 2] << 16) | ((uint32_t)gen_xy[4 * j + 3] << 24);
    yw[j] = (uint32_t)gen_xy[32 + 4 * j] | ((uint32_t)gen_xy[33 + 4 * j] << 8) | ((uint32_t)gen_xy[34 + 4 * j] << 16) | ((uint32_t)gen_xy[35 + 4 * j] << 24);
  }
  const FeN x = fe_from_u256(xw), y = fe_from_u256(yw);
  if (w == 0) ok[0] = (!u256_ge_q(xw) && !u256_ge_q(yw) && sw_on_curve(x, y)) ? 1 : 0;
  p256_comb_build_row(comb, w, x, y);
}

// ------------------------------------------------------------------------------------------------ key sets
// `Public` keys a verifier meets again and again: each is decoded once and gets the rows of a radix-256 comb that a
// challenge can reach (challenge_len + 1 rows of 128 entries: 244 KB for the suite's 16-byte challenges), so that
// -c * Y in U = s G - c Y is 17 additions and no doubling, with no table to build per proof.  One lane per (key, row).
__global__ void __launch_bounds__(64) k_p256_keyset_build(size_t n_keys, const uint8_t* pks33, uint32_t* aff, uint8_t* valid,
                                                          uint32_t* combs, int rows) {
  const size_t t = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (t >= n_keys * (size_t)rows) return;
  const size_t key = t / rows;
  const int w = (int)(t % rows);
  FeN x, y;
  const bool ok = sec1_decode(x, y, pks33 + key * SEC1_LEN);
  if (w == 0) {
    valid[key] = ok ? 1 : 0;
#pragma unroll
    for (int j = 0; j < NL; ++j) { aff[key * 18 + j] = x.v[j]; aff[key * 18 + NL + j] = y.v[j]; }
  }
  // an undecodable key still gets a (never used) row of some point: the generator's place holder keeps the lanes uniform
  const FeN gx = fe_select(ok, x, fe_one()), gy = fe_select(ok, y, fe_one());
  p256_comb_build_row(combs + key * (size_t)rows * P256_COMB_ROW_WORDS, w, gx, gy);
}

// ------------------------------------------------------------------------------------------------ IETF verify
__global__ void __launch_bounds__(P256_BLOCK) k_p256_verify_decode(p256::VerifyArgs a) {
  const size_t i = (size_t)blockIdx.x * P256_BLOCK + threadIdx.x;
  if (i >= a.n) return;
  const size_t cap = a.ws.cap;
  FeN x[3], y[3];
  Sec1W enc[3];
  uint32_t c[8], s[8];
  if (a.key_index) {
    // keyed: Y comes from the key set (decoded when the set was made); H and Gamma are wire data as ever
    const size_t k = a.key_index[i] < a.n_keys ? a.key_index[i] : 0;
    bool ok = a.key_index[i] < a.n_keys && a.key_valid[k] != 0;
    const uint8_t* ke = a.key_enc + k * SEC1_LEN;
    enc[0].tag = ke[0];
    load_be256(enc[0].xw, ke + 1);
#pragma unroll
    for (int j = 0; j < NL; ++j) { x[0].v[j] = a.key_aff[k * 18 + j]; y[0].v[j] = a.key_aff[k * 18 + NL + j]; }
#pragma unroll 1
    for (int j = 1; j < 3; ++j) {
      const uint8_t* e = (j == 1 ? a.h : a.gamma) + i * SEC1_LEN;
      FeN xx, yy;
      ok = sec1_decode(xx, yy, e) && ok;
      Sec1W w;
      w.tag = e[0];
      load_be256(w.xw, e + 1);
      if (j == 1) { x[1] = xx; y[1] = yy; enc[1] = w; } else { x[2] = xx; y[2] = yy; enc[2] = w; }
    }
    p256_challenge_decode(c, a.c + i * 32, a.str.challenge_len);
    ok = p256_scalar_decode(s, a.s + i * 32) && ok;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      ws_store_fe(a.ws.aff, cap, i, j * 18, x[j]);
      ws_store_fe(a.ws.aff, cap, i, j * 18 + 9, y[j]);
      ws_store_enc(a.ws.enc, cap, i, j, enc[j]);
    }
    ws_store8(a.ws.sc, cap, i, 0, c);
    ws_store8(a.ws.sc, cap, i, 8, s);
    a.ws.flags[i] = ok ? 1 : 0;
    return;
  }
  const bool ok = a.affine_in
                      ? p256_verify_decode_affine_item(x, y, enc, c, s, a.pk + i * 64, a.h + i * 64, a.gamma + i * 64, a.c + i * 32,
                                                       a.s + i * 32, a.affine_in == 2, a.str.challenge_len)
                      : p256_verify_decode_item(x, y, enc, c, s, a.pk + i * SEC1_LEN, a.h + i * SEC1_LEN, a.gamma + i * SEC1_LEN,
                                                a.c + i * 32, a.s + i * 32, a.str.challenge_len);
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    ws_store_fe(a.ws.aff, cap, i, j * 18, x[j]);
    ws_store_fe(a.ws.aff, cap, i, j * 18 + 9, y[j]);
    ws_store_enc(a.ws.enc, cap, i, j, enc[j]);
  }
  ws_store8(a.ws.sc, cap, i, 0, c);
  ws_store8(a.ws.sc, cap, i, 8, s);
  a.ws.flags[i] = ok ? 1 : 0;
}

// WHICH = 0: U = s G - c Y (comb + one table); 1: V = s H - c Gamma (two tables).  Two kernels rather than one launch with
// blockIdx.y choosing: the V ladder needs every register there is (two table entries in flight), the U ladder does not,
// and a shared body would run both at the V ladder's occupancy.
// amdgpu_waves_per_eu(3): left alone the register allocator takes all 256 VGPRs and 34 AGPRs for these bodies -- one wave
// per SIMD, 0.78 of the issue slots -- although 168 registers and 104 B of scratch serve them as well (three waves).
template <int WHICH>
__global__ void __launch_bounds__(P256_BLOCK) __attribute__((amdgpu_waves_per_eu(3, 3))) k_p256_verify_mul(p256::VerifyArgs a) {
  const size_t i = (size_t)blockIdx.x * P256_BLOCK + threadIdx.x;
  if (i >= a.n || !a.ws.flags[i]) return;
  const size_t cap = a.ws.cap;
  uint32_t c[8], s[8];
  ws_load8(c, a.ws.sc, cap, i, 0);
  ws_load8(s, a.ws.sc, cap, i, 8);
  if constexpr (WHICH == 0) {
    if (a.key_index) {
      // keyed: U = s G - c Y from two combs: 33 + key_rows additions, no doubling, no table (wave-uniform branch)
      const size_t k = a.key_index[i] < a.n_keys ? a.key_index[i] : 0;
      const PtW cy = sw_comb_mul(a.key_combs + k * (size_t)a.key_rows * P256_COMB_ROW_WORDS, c, a.key_rows);
      ptw_store(ws_pt(a.ws.pts, cap, i, 0), cap, sw_add(sw_comb_mul(a.comb, s), sw_cneg(true, cy)));
      return;
    }
    uint32_t* ty = ws_tab(a.ws.tabs, cap, i, 0);
    sw_build_table(ty, 1, sw_from_affine(ws_load_fe(a.ws.aff, cap, i, 0), ws_load_fe(a.ws.aff, cap, i, 9)));
    ptw_store(ws_pt(a.ws.pts, cap, i, 0), cap, sw_comb_minus_win(a.comb, ty, 1, s, c, sw_challenge_windows(a.str)));
  } else {
    uint32_t* th = ws_tab(a.ws.tabs, cap, i, 1);
    uint32_t* tg = ws_tab(a.ws.tabs, cap, i, 2);
    sw_build_table(th, 1, sw_from_affine(ws_load_fe(a.ws.aff, cap, i, 18), ws_load_fe(a.ws.aff, cap, i, 27)));
    sw_build_table(tg, 1, sw_from_affine(ws_load_fe(a.ws.aff, cap, i, 36), ws_load_fe(a.ws.aff, cap, i, 45)));
    ptw_store(ws_pt(a.ws.pts, cap, i, 1), cap, sw_straus_sc(th, tg, 1, s, c, sw_challenge_windows(a.str)));
  }
}

__global__ void __launch_bounds__(P256_BLOCK) k_p256_verify_finish(p256::VerifyArgs a) {
  const size_t i = (size_t)blockIdx.x * P256_BLOCK + threadIdx.x;
  if (i >= a.n) return;
  const size_t cap = a.ws.cap;
  if (!a.ws.flags[i]) { a.status[i] = 2; return; }
  const PtW U = ptw_load(ws_pt(a.ws.pts, cap, i, 0), cap), V = ptw_load(ws_pt(a.ws.pts, cap, i, 1), cap);
  const Sec1W enc[3] = {ws_load_enc(a.ws.enc, cap, i, 0), ws_load_enc(a.ws.enc, cap, i, 1), ws_load_enc(a.ws.enc, cap, i, 2)};
  uint32_t c[8];
  ws_load8(c, a.ws.sc, cap, i, 0);
  const uint8_t* ad;
  uint32_t ad_len;
  bytes_lite_get(a.ad, i, ad, ad_len);
  a.status[i] = p256_verify_finish_item(U, V, enc, c, ad, ad_len, a.str);
}

// ------------------------------------------------------------------------------------------------ IETF prove
// Stage 0: every item's first counter whose candidate decodes.  The lanes of a persistent wave draw items from a global
// queue: a lane whose attempt succeeded records the counter and takes the next item at once, so a wave performs about
// two attempts per item instead of the seven its unluckiest lane would impose (the same schedule as k_tai_find of the
// Edwards suites, k_prove.hip).  ctr_out = the item's flag byte; the prepare stage starts its search there.
__global__ void __launch_bounds__(64, 2) k_p256_tai_find(size_t n, BytesViewLite msg, uint8_t* ctr_out, SuiteStr str,
                                                         unsigned long long* queue) {
  constexpr size_t NONE = ~size_t(0);
  const int lane = threadIdx.x;
  size_t item = NONE;
  uint32_t ctr = 0;
  bool drained = false;                                    // the queue has no items left (wave-uniform)
  while (true) {
    const bool need = item == NONE && !drained;
    const unsigned long long mask = __ballot(need);
    if (mask) {
      const uint32_t cnt = (uint32_t)__popcll(mask);
      const uint32_t rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
      unsigned long long base = 0;
      if (lane == 0) base = atomicAdd(queue, (unsigned long long)cnt);
      base = __shfl(base, 0, 64);
      if (need && base + rank < n) { item = (size_t)(base + rank); ctr = 0; }
      if (base + cnt >= n) drained = true;
    }
    if (!__any(item != NONE)) break;                       // every lane idle and nothing left to draw
    if (item != NONE) {
      const uint8_t* m;
      uint32_t len;
      bytes_lite_get(msg, item, m, len);
      Sha256 pre;
      p256_tai_prefix(pre, m, len, str);
      uint32_t w[8];
      if (p256_tai_attempt(w, pre, ctr) || ctr == 255) {
        ctr_out[item] = (uint8_t)ctr;
        item = NONE;
      } else {
        ++ctr;
      }
    }
  }
}

// PED = 1: the Pedersen prover (a compile-time switch: the IETF kernels keep the code and the registers they had)
template <int PED>
__global__ void __launch_bounds__(P256_BLOCK) k_p256_prove_prepare(p256::ProveArgs a) {
  const size_t i = (size_t)blockIdx.x * P256_BLOCK + threadIdx.x;
  if (i >= a.n) return;
  const size_t cap = a.ws.cap;
  const uint8_t* msg = nullptr;
  uint32_t msg_len = 0;
  if (!a.h_given) bytes_lite_get(a.msg, i, msg, msg_len);
  uint32_t sk[8], k[8];
  FeN hx, hy;
  Sec1W henc;
  const uint32_t tai_start = a.h_given ? 0u : a.ws.flags[i];          // k_p256_tai_find left the counter there
  const bool ok = p256_prove_prepare_item(sk, k, hx, hy, henc, a.sk + i * 32, msg, msg_len,
                                          a.h_given ? a.h_given + i * SEC1_LEN : nullptr, a.str, tai_start);
  ws_store8(a.ws.sc, cap, i, 0, sk);
  ws_store8(a.ws.sc, cap, i, 8, k);
  if constexpr (PED != 0) {                               // blinding factor b and its nonce kb
    const uint8_t* ad;
    uint32_t ad_len;
    bytes_lite_get(a.ad, i, ad, ad_len);
    uint32_t b[8], kb[8];
    p256_blinding(b, sk, henc, ad, ad_len, a.str);
    p256_nonce(kb, b, henc.tag, henc.xw);
    ws_store8(a.ws.sc, cap, i, 16, b);
    ws_store8(a.ws.sc, cap, i, 24, kb);
  }
  ws_store_fe(a.ws.aff, cap, i, 0, hx);
  ws_store_fe(a.ws.aff, cap, i, 9, hy);
  ws_store_enc(a.ws.enc, cap, i, 0, henc);
  a.ws.flags[i] = ok ? 1 : 0;
}

// The four tables both H ladders of the next stage walk (p256_core.cuh sw_build_quad_tables): 192 doublings + 28
// additions per proof.  A kernel of its own: the hashing of the prepare stage keeps a SHA-256 state in scratch and runs two
// waves per SIMD, this is pure curve arithmetic and runs three.
__global__ void __launch_bounds__(P256_BLOCK) __attribute__((amdgpu_waves_per_eu(3, 3))) k_p256_prove_tables(p256::ProveArgs a) {
  const size_t i = (size_t)blockIdx.x * P256_BLOCK + threadIdx.x;
  if (i >= a.n || !a.ws.flags[i]) return;
  const size_t cap = a.ws.cap;
  sw_build_quad_tables(ws_quad_tab(a.ws.tabs, i), 1, ws_load_fe(a.ws.aff, cap, i, 0), ws_load_fe(a.ws.aff, cap, i, 9));
}

// blockIdx.y = 0: pk = sk G, 1: Gamma = sk H, 2: U = k G, 3: V = k H; the two H jobs walk the tables of H, 2^64 H, 2^128 H,
// 2^192 H that the prepare stage built (60 doublings each instead of 256).  Pedersen: job 0 is pk_com = sk G + b B, job 2 is
// R = k G + kb B (a second comb walk over the blinding base's table), jobs 1 and 3 are Gamma and Ok = k H.
template <int PED>
__global__ void __launch_bounds__(P256_BLOCK) __attribute__((amdgpu_waves_per_eu(3, 3))) k_p256_prove_mul(p256::ProveArgs a) {
  const size_t i = (size_t)blockIdx.x * P256_BLOCK + threadIdx.x;
  if (i >= a.n || !a.ws.flags[i]) return;
  const size_t cap = a.ws.cap;
  const int job = blockIdx.y;
  uint32_t k[8];
  ws_load8(k, a.ws.sc, cap, i, (job & 2) ? 8 : 0);
  PtW r;
  if (job & 1) {
    r = sw_quad_mul(ws_quad_tab(a.ws.tabs, i), 1, k, a.ct_tables != 0);
  } else {
    r = sw_comb_mul(a.comb, k);
    if constexpr (PED != 0) {
      uint32_t kb[8];
      ws_load8(kb, a.ws.sc, cap, i, (job & 2) ? 24 : 16);
      r = sw_add(r, sw_comb_mul(a.comb_b, kb));
    }
  }
  ptw_store(ws_pt(a.ws.pts, cap, i, job), cap, r);
}

template <int PED>
__global__ void __launch_bounds__(P256_BLOCK) k_p256_prove_finish(p256::ProveArgs a) {
  const size_t i = (size_t)blockIdx.x * P256_BLOCK + threadIdx.x;
  if (i >= a.n) return;
  const size_t cap = a.ws.cap;
  const size_t pw = a.out_affine ? 64 : SEC1_LEN;          // width of the points this stage hands out
  if (!a.ws.flags[i]) {
    if (a.status) a.status[i] = 2;
    for (size_t k = 0; k < pw; ++k) {
      a.gamma[i * pw + k] = 0;
      if (a.pk_out) a.pk_out[i * pw + k] = 0;
      if (PED != 0) { a.r_out[i * pw + k] = 0; a.ok_out[i * pw + k] = 0; }
    }
    for (int k = 0; k < SEC1_LEN; ++k)
      if (a.h_out) a.h_out[i * SEC1_LEN + k] = 0;
    for (int k = 0; k < 32; ++k) {
      a.s[i * 32 + k] = 0;
      if (PED != 0) { a.sb_out[i * 32 + k] = 0; if (a.blinding_out) a.blinding_out[i * 32 + k] = 0; }
      else a.c[i * 32 + k] = 0;
    }
    return;
  }
  const PtW res[4] = {ptw_load(ws_pt(a.ws.pts, cap, i, 0), cap), ptw_load(ws_pt(a.ws.pts, cap, i, 1), cap),
                      ptw_load(ws_pt(a.ws.pts, cap, i, 2), cap), ptw_load(ws_pt(a.ws.pts, cap, i, 3), cap)};
  const Sec1W henc = ws_load_enc(a.ws.enc, cap, i, 0);
  uint32_t sk[8], k[8], c[8], s[8];
  ws_load8(sk, a.ws.sc, cap, i, 0);
  ws_load8(k, a.ws.sc, cap, i, 8);
  const uint8_t* ad;
  uint32_t ad_len;
  bytes_lite_get(a.ad, i, ad, ad_len);
  if constexpr (PED != 0) {
    uint32_t b[8], kb[8], sb[8];
    ws_load8(b, a.ws.sc, cap, i, 16);
    ws_load8(kb, a.ws.sc, cap, i, 24);
    Sec1W enc[4];
    uint32_t yw[4][8];
    p256_ped_prove_finish_item(enc, s, sb, res, henc, sk, k, b, kb, ad, ad_len, a.str, a.out_affine ? yw : nullptr);
    uint8_t* dst[4] = {a.pk_out, a.gamma, a.r_out, a.ok_out};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (a.out_affine) xy_store(dst[j] + i * 64, enc[j].tag, enc[j].xw, yw[j], a.out_affine == 2);
      else sec1_store(dst[j] + i * SEC1_LEN, enc[j].tag, enc[j].xw);
    }
    store_be256(a.s + i * 32, s);
    store_be256(a.sb_out + i * 32, sb);
    if (a.blinding_out) store_be256(a.blinding_out + i * 32, b);
    if (a.h_out) sec1_store(a.h_out + i * SEC1_LEN, henc.tag, henc.xw);
    if (a.status) a.status[i] = 0;
  } else {
    Sec1W pk, gamma;
    uint32_t yw[4][8];
    p256_prove_finish_item(pk, gamma, c, s, res, henc, sk, k, ad, ad_len, a.str, a.out_affine ? yw : nullptr);
    store_be256(a.c + i * 32, c);
    store_be256(a.s + i * 32, s);
    if (a.out_affine) {
      xy_store(a.gamma + i * 64, gamma.tag, gamma.xw, yw[1], a.out_affine == 2);
      if (a.pk_out) xy_store(a.pk_out + i * 64, pk.tag, pk.xw, yw[0], a.out_affine == 2);
    } else {
      sec1_store(a.gamma + i * SEC1_LEN, gamma.tag, gamma.xw);
      if (a.pk_out) sec1_store(a.pk_out + i * SEC1_LEN, pk.tag, pk.xw);
    }
    if (a.h_out) sec1_store(a.h_out + i * SEC1_LEN, henc.tag, henc.xw);
    if (a.status) a.status[i] = 0;
  }
}

// ------------------------------------------------------------------------------------------------ Pedersen verify
__global__ void __launch_bounds__(P256_BLOCK) k_p256_ped_verify_decode(p256::PedVerifyArgs a) {
  const size_t i = (size_t)blockIdx.x * P256_BLOCK + threadIdx.x;
  if (i >= a.n) return;
  const size_t cap = a.ws.cap;
  const uint8_t* ad;
  uint32_t ad_len;
  bytes_lite_get(a.ad, i, ad, ad_len);
  FeN x[5], y[5];
  uint32_t c[8], s[8], sb[8];
  const bool ok = p256_ped_verify_decode_item(x, y, c, s, sb, a.h + i * SEC1_LEN, a.gamma + i * SEC1_LEN, a.pk_com + i * SEC1_LEN,
                                              a.r + i * SEC1_LEN, a.ok + i * SEC1_LEN, a.s + i * 32, a.sb + i * 32, ad, ad_len, a.str);
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    ws_store_fe(a.ws.aff, cap, i, j * 18, x[j]);
    ws_store_fe(a.ws.aff, cap, i, j * 18 + 9, y[j]);
  }
  ws_store8(a.ws.sc, cap, i, 0, c);
  ws_store8(a.ws.sc, cap, i, 8, s);
  ws_store8(a.ws.sc, cap, i, 16, sb);
  a.ws.flags[i] = ok ? 1 : 0;
}
// WHICH = 0: s H - c Gamma - Ok = O (two tables); 1: s G + sb B - c pk_com - R = O (two comb walks, one table).  The
// verdict of each equation goes to word WHICH of the item's result record.
template <int WHICH>
__global__ void __launch_bounds__(P256_BLOCK) __attribute__((amdgpu_waves_per_eu(3, 3))) k_p256_ped_verify_mul(p256::PedVerifyArgs a) {
  const size_t i = (size_t)blockIdx.x * P256_BLOCK + threadIdx.x;
  if (i >= a.n || !a.ws.flags[i]) return;
  const size_t cap = a.ws.cap;
  const int cw = sw_challenge_windows(a.str);
  uint32_t c[8], s[8];
  ws_load8(c, a.ws.sc, cap, i, 0);
  ws_load8(s, a.ws.sc, cap, i, 8);
  auto pt = [&](int j) { return sw_from_affine(ws_load_fe(a.ws.aff, cap, i, j * 18), ws_load_fe(a.ws.aff, cap, i, j * 18 + 9)); };
  bool holds;
  if constexpr (WHICH == 0) {
    uint32_t* th = ws_tab(a.ws.tabs, cap, i, 0);
    uint32_t* tg = ws_tab(a.ws.tabs, cap, i, 1);
    sw_build_table(th, 1, pt(0));
    sw_build_table(tg, 1, pt(1));
    holds = p256_ped_verify_eq_h(th, tg, 1, ws_load_fe(a.ws.aff, cap, i, 72), ws_load_fe(a.ws.aff, cap, i, 81), s, c, cw);
  } else {
    uint32_t sb[8];
    ws_load8(sb, a.ws.sc, cap, i, 16);
    uint32_t* tp = ws_tab(a.ws.tabs, cap, i, 2);
    sw_build_table(tp, 1, pt(2));
    holds = p256_ped_verify_eq_g(a.comb, a.comb_b, tp, 1, ws_load_fe(a.ws.aff, cap, i, 54), ws_load_fe(a.ws.aff, cap, i, 63), s, sb, c, cw);
  }
  a.ws.pts[(size_t)WHICH * cap + i] = holds ? 1u : 0u;
}
__global__ void __launch_bounds__(P256_BLOCK) k_p256_ped_verify_finish(p256::PedVerifyArgs a) {
  const size_t i = (size_t)blockIdx.x * P256_BLOCK + threadIdx.x;
  if (i >= a.n) return;
  const size_t cap = a.ws.cap;
  a.status[i] = !a.ws.flags[i] ? 2 : (a.ws.pts[i] && a.ws.pts[cap + i]) ? 0 : 1;
}

// ------------------------------------------------------------------------------------------------ building blocks
__global__ void __launch_bounds__(P256_BLOCK) k_p256_hash_to_curve(size_t n, BytesViewLite msg, uint8_t* out, SuiteStr str,
                                                                    const uint8_t* tai_ctr) {
  const size_t i = (size_t)blockIdx.x * P256_BLOCK + threadIdx.x;
  if (i >= n) return;
  const uint8_t* m;
  uint32_t len;
  bytes_lite_get(msg, i, m, len);
  FeN x, y;
  uint32_t xw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const bool ok = p256_hash_to_curve(x, y, xw, m, len, str, tai_ctr ? tai_ctr[i] : 0u);     // k_p256_tai_find's counter
  sec1_store(out + i * SEC1_LEN, ok ? 2u : 0u, xw);       // 256 failed attempts (2^-256): an all-zero string
}

__global__ void __launch_bounds__(P256_BLOCK) k_p256_output_hash(size_t n, const uint8_t* gamma, uint8_t* out, SuiteStr str) {
  const size_t i = (size_t)blockIdx.x * P256_BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t xw[8], h[8];
  load_be256(xw, gamma + i * SEC1_LEN + 1);
  p256_output_hash(h, gamma[i * SEC1_LEN], xw, str);
  store_be256(out + i * 32, h);
}

__global__ void __launch_bounds__(P256_BLOCK) k_p256_secret_from_seed(size_t n, const uint8_t* seeds, uint32_t seed_len, uint8_t* sk_out,
                                                                      uint8_t* pk_out, const uint32_t* comb) {
  const size_t i = (size_t)blockIdx.x * P256_BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t sk[8];
  p256_secret_from_seed(sk, seeds + i * (size_t)seed_len, seed_len);
  store_be256(sk_out + i * 32, sk);
  if (pk_out) {
    const PtW p[1] = {sw_comb_mul(comb, sk)};
    Sec1W w[1];
    sw_to_sec1<true>(w, p);
    sec1_store(pk_out + i * SEC1_LEN, w[0].tag, w[0].xw);
  }
}

// [ref src/lib.rs:14 `codec`] point_decode: 0 = a point of the curve (cofactor 1: of the group), 2 = InvalidData.
// xy_out (nullable): x || y as 32-byte LITTLE-endian canonical integers, the form every *_xy array of this ABI has
__global__ void __launch_bounds__(P256_BLOCK) k_p256_point_validate(size_t n, const uint8_t* pts, uint8_t* xy_out, int mont256,
                                                                    uint8_t* status) {
  const size_t i = (size_t)blockIdx.x * P256_BLOCK + threadIdx.x;
  if (i >= n) return;
  FeN x, y;
  const bool ok = sec1_decode(x, y, pts + i * SEC1_LEN);
  status[i] = ok ? 0 : 2;
  if (xy_out) {
    uint32_t xw[8], yw[8];
    fe_to_u256(xw, x);
    fe_to_u256(yw, y);
    xy_store(xy_out + i * 64, ok ? 2u : 0u, xw, yw, mont256 != 0);
  }
}

inline unsigned blocks_for(size_t n) { return (unsigned)((n + P256_BLOCK - 1) / P256_BLOCK); }

}  // namespace
VRF_NS_END

namespace vrf {
namespace p256 {

size_t comb_bytes() { return P256_COMB_WORDS * sizeof(uint32_t); }

void default_generator(uint8_t xy[64]) { std::memcpy(xy, vrfk_tables::P256_G_XY, 64); }
void default_blinding_base(uint8_t xy[64]) { std::memcpy(xy, vrfk_tables::P256_B_XY, 64); }

void launch_init_comb(uint32_t* comb, const uint8_t* d_gen_xy, uint8_t* d_ok, hipStream_t st) {
  hipLaunchKernelGGL(k_p256_init_comb, dim3(1), dim3(64), 0, st, comb, d_gen_xy, d_ok);
}

size_t key_comb_bytes(int rows) { return (size_t)rows * P256_COMB_ROW_WORDS * sizeof(uint32_t); }
void launch_keyset_build(size_t n_keys, const uint8_t* pks33, uint32_t* aff, uint8_t* valid, uint32_t* combs, int rows, hipStream_t st) {
  if (!n_keys) return;
  const size_t lanes = n_keys * (size_t)rows;
  hipLaunchKernelGGL(k_p256_keyset_build, dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, st, n_keys, pks33, aff, valid, combs, rows);
}
void launch_verify(const VerifyArgs& a, hipStream_t st, hipEvent_t* ev) {
  const unsigned g = blocks_for(a.n);
  if (ev) (void)hipEventRecord(ev[0], st);
  hipLaunchKernelGGL(k_p256_verify_decode, dim3(g), dim3(P256_BLOCK), 0, st, a);
  if (ev) (void)hipEventRecord(ev[1], st);
  hipLaunchKernelGGL(k_p256_verify_mul<1>, dim3(g), dim3(P256_BLOCK), 0, st, a);
  if (ev) (void)hipEventRecord(ev[2], st);
  hipLaunchKernelGGL(k_p256_verify_mul<0>, dim3(g), dim3(P256_BLOCK), 0, st, a);
  if (ev) (void)hipEventRecord(ev[3], st);
  hipLaunchKernelGGL(k_p256_verify_finish, dim3(g), dim3(P256_BLOCK), 0, st, a);
  if (ev) (void)hipEventRecord(ev[4], st);
}

void launch_prove(const ProveArgs& a, hipStream_t st, hipEvent_t* ev) {
  const unsigned g = blocks_for(a.n);
  if (ev) (void)hipEventRecord(ev[0], st);
  if (!a.h_given) {
    (void)hipMemsetAsync(a.tai_queue, 0, sizeof(unsigned long long), st);
    const size_t waves = std::min<size_t>((a.n + 63) / 64, 4096);          // persistent: 4 waves per SIMD
    hipLaunchKernelGGL(k_p256_tai_find, dim3((unsigned)waves), dim3(64), 0, st, a.n, a.msg, a.ws.flags, a.str, a.tai_queue);
  }
  if (a.pedersen) hipLaunchKernelGGL(k_p256_prove_prepare<1>, dim3(g), dim3(P256_BLOCK), 0, st, a);
  else hipLaunchKernelGGL(k_p256_prove_prepare<0>, dim3(g), dim3(P256_BLOCK), 0, st, a);
  hipLaunchKernelGGL(k_p256_prove_tables, dim3(g), dim3(P256_BLOCK), 0, st, a);
  if (ev) (void)hipEventRecord(ev[1], st);
  if (a.pedersen) hipLaunchKernelGGL(k_p256_prove_mul<1>, dim3(g, 4), dim3(P256_BLOCK), 0, st, a);
  else hipLaunchKernelGGL(k_p256_prove_mul<0>, dim3(g, 4), dim3(P256_BLOCK), 0, st, a);
  if (ev) { (void)hipEventRecord(ev[2], st); (void)hipEventRecord(ev[3], st); }
  if (a.pedersen) hipLaunchKernelGGL(k_p256_prove_finish<1>, dim3(g), dim3(P256_BLOCK), 0, st, a);
  else hipLaunchKernelGGL(k_p256_prove_finish<0>, dim3(g), dim3(P256_BLOCK), 0, st, a);
  if (ev) (void)hipEventRecord(ev[4], st);
}

void launch_pedersen_verify(const PedVerifyArgs& a, hipStream_t st, hipEvent_t* ev) {
  const unsigned g = blocks_for(a.n);
  if (ev) (void)hipEventRecord(ev[0], st);
  hipLaunchKernelGGL(k_p256_ped_verify_decode, dim3(g), dim3(P256_BLOCK), 0, st, a);
  if (ev) (void)hipEventRecord(ev[1], st);
  hipLaunchKernelGGL(k_p256_ped_verify_mul<0>, dim3(g), dim3(P256_BLOCK), 0, st, a);
  if (ev) (void)hipEventRecord(ev[2], st);
  hipLaunchKernelGGL(k_p256_ped_verify_mul<1>, dim3(g), dim3(P256_BLOCK), 0, st, a);
  if (ev) (void)hipEventRecord(ev[3], st);
  hipLaunchKernelGGL(k_p256_ped_verify_finish, dim3(g), dim3(P256_BLOCK), 0, st, a);
  if (ev) (void)hipEventRecord(ev[4], st);
}

// tai_ctr ([n] bytes) + queue: scratch for the work-queue counter search the prover uses; without them every lane loops until
// its own item succeeds
void launch_hash_to_curve(size_t n, BytesViewLite msg, uint8_t* points33, const SuiteStr& str, hipStream_t st, uint8_t* tai_ctr,
                          unsigned long long* queue) {
  if (!n) return;
  if (tai_ctr && queue) {
    (void)hipMemsetAsync(queue, 0, sizeof(unsigned long long), st);
    const size_t waves = std::min<size_t>((n + 63) / 64, 4096);
    hipLaunchKernelGGL(k_p256_tai_find, dim3((unsigned)waves), dim3(64), 0, st, n, msg, tai_ctr, str, queue);
  } else {
    tai_ctr = nullptr;
  }
  hipLaunchKernelGGL(k_p256_hash_to_curve, dim3(blocks_for(n)), dim3(P256_BLOCK), 0, st, n, msg, points33, str, tai_ctr);
}
void launch_output_hash(size_t n, const uint8_t* gamma33, uint8_t* hash32, const SuiteStr& str, hipStream_t st) {
  hipLaunchKernelGGL(k_p256_output_hash, dim3(blocks_for(n)), dim3(P256_BLOCK), 0, st, n, gamma33, hash32, str);
}
void launch_secret_from_seed(size_t n, const uint8_t* seeds, uint32_t seed_len, uint8_t* sk32, uint8_t* pk33, const uint32_t* comb,
                             hipStream_t st) {
  hipLaunchKernelGGL(k_p256_secret_from_seed, dim3(blocks_for(n)), dim3(P256_BLOCK), 0, st, n, seeds, seed_len, sk32, pk33, comb);
}
void launch_point_validate(size_t n, const uint8_t* points33, uint8_t* xy_out, int xy_mont256, uint8_t* status, hipStream_t st) {
  hipLaunchKernelGGL(k_p256_point_validate, dim3(blocks_for(n)), dim3(P256_BLOCK), 0, st, n, points33, xy_out, xy_mont256, status);
}

}  // namespace p256
}  // namespace vrf
