// k_p256_msm.hip -- multi-scalar multiplication on secp256r1 and the batched Pedersen verifier on top of it
// (`suites::secp256r1` x `VariableBaseMSM::msm` / `pedersen::Verifier`, /root/reference src/lib.rs:14; SURVEY.md section 8 rows
// a14, f2, f4).  Compiled with -DVRF_FIELD=3 beside k_p256.hip; launch interface in p256.h.
//
// Pippenger with signed 10-bit windows and the window's 512 buckets in LDS (54 KiB), the schedule of k_msm.hip / k_msm_g1.hip:
//   k_pm_prep      : x || y (little-endian canonical or arkworks Montgomery limbs) -> Montgomery 9 x 29-bit limbs, on-curve
//                    test; big-endian scalars < n -> folded (k or n - k) signed digits
//   k_pm_buckets   : one workgroup per (window, point group): counting sort by bucket (LDS atomics, wave-shuffle scan), the
//                    sorted list cut into 512 equal chunks, every lane folds its chunk with the COMPLETE addition of sw.cuh
//                    (a bucket may meet P and -P, or P twice: no exceptional case to branch on), list entries fetched two
//                    trips ahead and points one trip ahead of their use; heads merged; sum_j j B_j by suffix scan + tree
//   k_pm_final     : per window the sum of its groups, 10 w Jacobian doublings, tree-sum, then Sec1 / x || y -- or, for the
//                    batched verifier, only "is it the point at infinity"
// Batched Pedersen verification (k_pm_rlc_*): with 128-bit weights z_i, z'_i drawn from a digest of every input byte,
//   sum_i z_i (s_i H_i - c_i Gamma_i - Ok_i) + z'_i (s_i G + sb_i B - c_i pk_com_i - R_i) = O
// is ONE MSM over 5n + 2 points instead of 4n scalar multiplications.  The curve has cofactor 1: decoding is the whole
// validation, and the soundness error is 2^-128 per batch.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include "p256.h"
#include "p256_core.cuh"

VRF_NS_BEGIN
namespace {

using p256::MsmL;
constexpr int PM_C = p256::MSM_C, PM_BUCKETS = 1 << (PM_C - 1), PM_BLOCK = PM_BUCKETS, PM_W = p256::MSM_W;
constexpr int PM_PT = PTW_WORDS;                       // 27 words: projective bucket / partial sum
constexpr int PM_AFF = p256::MSM_AFF_STRIDE;           // 32-word slot of an affine point (18 used): one cache line per gather
constexpr int PM_IDX_BITS = 21;
constexpr uint32_t PM_IDX_MASK = (1u << PM_IDX_BITS) - 1, PM_NONE = 0xffffffffu;
static_assert(PM_W * PM_C >= 257 && 2 * NL <= PM_AFF, "window count / slot size");

__device__ __forceinline__ void pm_store(uint32_t* dst, const PtW& p) { ptw_store(dst, 1, p); }
__device__ __forceinline__ PtW pm_load(const uint32_t* src) { return ptw_load(src, 1); }

// signed radix-2^10 digits of k (< n, little-endian words), folded to min(k, n - k) with the sign flipped; `negate` flips
// the term, `zero` drops the point
__device__ void pm_write_digits(int16_t* digits, size_t n, size_t i, const uint32_t k_in[8], bool negate, bool zero) {
  uint32_t k[8], nk[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) k[j] = zero ? 0u : k_in[j];
  {
    uint32_t borrow = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint64_t d = (uint64_t)CurveP256::r32(j) - k[j] - borrow;
      nk[j] = (uint32_t)d;
      borrow = (uint32_t)(d >> 63);
    }
  }
  bool flip = false, decided = false;
#pragma unroll
  for (int j = 7; j >= 0; --j)
    if (!decided && nk[j] != k[j]) { flip = nk[j] < k[j]; decided = true; }
  flip = flip && !zero;
#pragma unroll
  for (int j = 0; j < 8; ++j) k[j] = flip ? nk[j] : k[j];
  const bool neg = flip != negate;
  uint32_t carry = 0;
#pragma unroll 1
  for (int w = 0; w < PM_W; ++w) {
    const int bit = w * PM_C, wi = bit >> 5, sh = bit & 31;
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j == wi) lo = k[j];
      if (j == wi + 1) hi = k[j];
    }
    uint32_t v = (sh ? ((lo >> sh) | (hi << (32 - sh))) : lo) & ((1u << PM_C) - 1);
    v += carry;
    int d = (int)v;
    carry = 0;
    if (v > (uint32_t)PM_BUCKETS) { d = (int)v - (1 << PM_C); carry = 1; }
    digits[(size_t)w * n + i] = (int16_t)(neg ? -d : d);
  }
}
__device__ __forceinline__ void pm_store_affine(uint32_t* slot, const FeN& x, const FeN& y) {
#pragma unroll
  for (int j = 0; j < NL; ++j) { slot[j] = x.v[j]; slot[NL + j] = y.v[j]; }
}

// ------------------------------------------------------------------------------- prep (plain MSM)
__global__ void __launch_bounds__(128) k_pm_prep(MsmL L, const uint8_t* xy, const uint8_t* scalars_be, int mont256) {
  const size_t i = (size_t)blockIdx.x * 128 + threadIdx.x;
  if (i >= L.n) return;
  const uint32_t* e = reinterpret_cast<const uint32_t*>(xy + i * 64);
  uint32_t xin[8], yin[8], xc[8], yc[8], k[8], kw[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { xin[j] = e[j]; yin[j] = e[8 + j]; }
  const FeN x = fe_from_abi(xc, xin, mont256 != 0), y = fe_from_abi(yc, yin, mont256 != 0);
  load_be256(kw, scalars_be + i * 32);
  fr_reduce256<CurveP256>(k, kw);
  // all-zero x || y is the point at infinity (it takes no part); anything else must be on the curve
  uint32_t any = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) any |= xin[j] | yin[j];
  const bool inf = any == 0;
  const bool ok = fr_is_canonical<CurveP256>(kw) && !u256_ge_q(xin) && !u256_ge_q(yin) && (inf || sw_on_curve(x, y));
  if (!ok) L.flags[0] = 2;
  pm_store_affine(L.pts + i * PM_AFF, x, y);
  pm_write_digits(L.digits, L.n, i, k, false, !ok || inf);
}

// P + (x2, y2) for any P (the point at infinity included) and an AFFINE second operand (RCB 2016 algorithm 5, a = -3):
// 11 products + 2 by b -- sw_add with Z2 = 1 spelt out, which drops the Z1 Z2 product and turns the two Karatsuba-shaped
// cross terms into one product each.  The bucket loop adds nothing but affine input points.
VRF_HD PtW sw_madd(const PtW& p, const FeN& x2, const FeN& y2) {
  const FeN b = CurveP256::b();
  const FeN t0 = fe_mul(p.X, x2), t1 = fe_mul(p.Y, y2);
  const FeN t2 = p.Z;
  const auto t3 = fe_sub(fe_mul(fe_add(p.X, p.Y), fe_add(x2, y2)), fe_add(t0, t1));      // X1 Y2 + X2 Y1
  const auto t4 = fe_add(fe_mul(y2, p.Z), p.Y);                                           // Y1 + Y2 Z1
  const auto y3 = fe_add(fe_mul(x2, p.Z), p.X);                                           // X1 + X2 Z1
  const FeN a1 = fe_wred(fe_sub(y3, fe_mul(t2, b)));
  const auto a3 = fe_add(fe_dbl(a1), a1);
  const auto z3 = fe_norm(fe_sub(t1, fe_norm(a3)));
  const auto x3 = fe_norm(fe_add(t1, a3));
  const auto t2x3 = fe_norm(fe_add(fe_dbl(t2), t2));
  const FeN bb = fe_wred(fe_sub(fe_sub(fe_mul(y3, b), t2x3), t0));
  const auto y3b = fe_add(fe_dbl(bb), bb);
  const auto t0b = fe_norm(fe_sub(fe_add(fe_dbl(t0), t0), t2x3));
  const auto t1b = fe_mul(fe_norm(t4), y3b);
  const auto t2b = fe_mul(t0b, fe_norm(y3b));
  PtW r;
  r.Y = fe_wred(fe_add(fe_mul(x3, z3), t2b));
  r.X = fe_wred(fe_sub(fe_mul(fe_norm(t3), x3), t1b));
  r.Z = fe_wred(fe_add(fe_mul(fe_norm(t4), z3), fe_mul(fe_norm(t3), t0b)));
  return r;
}

// ------------------------------------------------------------------------------- buckets
__global__ void __launch_bounds__(PM_BLOCK) k_pm_buckets(MsmL L) {
  extern __shared__ uint32_t lds[];
  uint32_t* bucket = lds;                                   // [512][27]
  uint32_t* counts = lds + PM_BUCKETS * PM_PT;              // [512]
  uint32_t* cursor = counts + PM_BUCKETS;                   // [512] scatter cursors, later head bucket ids
  uint32_t* wsum = cursor + PM_BUCKETS;                     // [8]
  const int t = threadIdx.x;
  const int wg = blockIdx.x;
  // the grid holds the low windows' workgroups (MSM_W_SHORT x groups) and then the high windows' (x groups_hi): the windows a
  // bare 128-bit weight cannot reach hold the full-size scalars only, in fewer groups of about the same size
  const int n_lo = p256::MSM_W_SHORT * L.groups;
  const bool high = wg >= n_lo;
  const int w = high ? p256::MSM_W_SHORT + (wg - n_lo) / L.groups_hi : wg / L.groups;
  const int g = high ? (wg - n_lo) % L.groups_hi : wg % L.groups;
  const size_t span = high ? L.per_group_hi : L.per_group, end = high ? L.n_long : L.n;
  const size_t lo = (size_t)g * span;
  const size_t hi = lo + span < end ? lo + span : end;
  const uint32_t cnt_all = lo < hi ? (uint32_t)(hi - lo) : 0u;
  const int16_t* dig = L.digits + (size_t)w * L.n + lo;
  const uint32_t* P = L.pts + lo * PM_AFF;
  uint32_t* list = L.lists + (size_t)wg * L.list_cap;
  uint32_t* heads = L.heads + (size_t)wg * PM_BLOCK * PM_PT;
  pm_store(bucket + t * PM_PT, sw_identity());
  // 1. histogram
  counts[t] = 0;
  __syncthreads();
  for (uint32_t j0 = t; j0 < cnt_all; j0 += 4 * PM_BLOCK) {
    int d[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t j = j0 + u * PM_BLOCK;
      d[u] = j < cnt_all ? dig[j] : 0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (d[u] != 0) atomicAdd(&counts[(d[u] < 0 ? -d[u] : d[u]) - 1], 1u);
  }
  __syncthreads();
  // 2. exclusive scan of the 512 counts
  uint32_t m;
  {
    const uint32_t v = counts[t];
    uint32_t incl = v;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
      const uint32_t u = __shfl_up(incl, s, 64);
      if ((t & 63) >= s) incl += u;
    }
    if ((t & 63) == 63) wsum[t >> 6] = incl;
    __syncthreads();
    uint32_t wbase = 0, total = 0;
    for (int k = 0; k < PM_BLOCK / 64; ++k) {
      const uint32_t x = wsum[k];
      if (k < (t >> 6)) wbase += x;
      total += x;
    }
    m = total;
    cursor[t] = wbase + incl - v;
  }
  __syncthreads();
  // 3. scatter into bucket order, transposed: entry i of lane l sits at list[i * 512 + l]
  const uint32_t chunk = (m + PM_BLOCK - 1) / PM_BLOCK;
  for (uint32_t j = t; j < cnt_all; j += PM_BLOCK) {
    const int d = dig[j];
    if (d != 0) {
      const uint32_t b = (uint32_t)(d < 0 ? -d : d) - 1;
      const uint32_t pos = atomicAdd(&cursor[b], 1u);
      const uint32_t lane = pos / chunk, i = pos - lane * chunk;
      list[(size_t)i * PM_BLOCK + lane] = j | (b << PM_IDX_BITS) | (d < 0 ? 0x80000000u : 0u);
    }
  }
  __threadfence_block();
  __syncthreads();
  // 4. every lane folds its chunk (same trip count for all lanes); entries two trips ahead, points one trip ahead
  uint32_t head_b = PM_NONE;
  {
    const uint32_t my0 = (uint32_t)t * chunk;
    const uint32_t cnt = my0 >= m ? 0u : (m - my0 < chunk ? m - my0 : chunk);
    PtW acc = sw_identity();
    uint32_t cur = PM_NONE;
    bool first_run = true;
    uint32_t* myhead = heads + (size_t)t * PM_PT;
    auto load_xy = [&](FeN& x, FeN& y, uint32_t e) {
      const uint32_t* src = P + (size_t)(e & PM_IDX_MASK) * PM_AFF;
#pragma unroll
      for (int j = 0; j < NL; ++j) { x.v[j] = src[j]; y.v[j] = src[NL + j]; }
    };
    uint32_t ent_n = 0, ent_nn = 0;
    FeN x_n = fe_zero(), y_n = fe_one();
    if (cnt > 0) ent_n = list[t];
    if (cnt > 1) ent_nn = list[(size_t)PM_BLOCK + t];
    if (cnt > 0) load_xy(x_n, y_n, ent_n);
#pragma unroll 1
    for (uint32_t i = 0; i < chunk; ++i) {
      if (i < cnt) {
        const uint32_t ent = ent_n;
        const FeN x = x_n, y = y_n;
        ent_n = ent_nn;
        if (i + 2 < cnt) ent_nn = list[(size_t)(i + 2) * PM_BLOCK + t];
        if (i + 1 < cnt) load_xy(x_n, y_n, ent_n);
        const uint32_t b = (ent >> PM_IDX_BITS) & (PM_BUCKETS - 1);
        if (b != cur) {
          if (cur != PM_NONE) {
            if (first_run) { pm_store(myhead, acc); head_b = cur; first_run = false; }
            else pm_store(bucket + cur * PM_PT, acc);
          }
          cur = b;
          acc = sw_identity();
        }
        acc = sw_madd(acc, x, fe_select((ent >> 31) != 0, fe_wred(fe_neg(y)), y));
      }
    }
    if (cnt > 0) {
      if (first_run) { pm_store(myhead, acc); head_b = cur; }
      else pm_store(bucket + cur * PM_PT, acc);
    }
  }
  cursor[t] = head_b;
  __threadfence_block();
  __syncthreads();
  // 5. merge the heads: the first lane of each chain of equal head buckets adds the chain to the bucket
  if (head_b != PM_NONE && (t == 0 || cursor[t - 1] != head_b)) {
    PtW h = pm_load(heads + (size_t)t * PM_PT);
    for (int k = t + 1; k < PM_BLOCK && cursor[k] == head_b; ++k) h = sw_add(h, pm_load(heads + (size_t)k * PM_PT));
    uint32_t* slot = bucket + head_b * PM_PT;
    pm_store(slot, sw_add(pm_load(slot), h));
  }
  __syncthreads();
  // 6. R = sum_j j B_j: lane t holds B_{t+1}; suffix scan S_t = sum_{u >= t} B_{u+1}, then R = sum_t S_t
  PtW cur = pm_load(bucket + t * PM_PT);
  uint32_t* stage = lds;
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll 1
    for (int k = 0; k < 9; ++k) {
      const int s = pass == 0 ? (1 << k) : (PM_BLOCK >> (k + 1));
      __syncthreads();
      pm_store(stage + t * PM_PT, cur);
      __syncthreads();
      const bool active = pass == 0 ? (t + s < PM_BLOCK) : (t < s);
      if (active) cur = sw_add(cur, pm_load(stage + (t + s) * PM_PT));
    }
  }
  if (t == 0) pm_store(L.part + ((size_t)w * L.groups + g) * PM_PT, cur);
}

// ------------------------------------------------------------------------------- final
// One lane per window: the sum of its groups, then 10 w doublings (Jacobian runs: 8 products each, no exceptional case on
// a curve without 2-torsion), a tree over the 26 windows, and the result as Sec1 / x || y -- or the verdict alone.
__global__ void __launch_bounds__(64) k_pm_final(MsmL L, uint8_t* out33, uint8_t* out_xy, int xy_mont256, uint8_t* status1,
                                                  uint8_t* fail_flag) {
  __shared__ uint32_t stage[32 * PM_PT];
  const int t = threadIdx.x;
  PtW acc = sw_identity();
  if (t < PM_W) {
    const int ng = t < p256::MSM_W_SHORT ? L.groups : L.groups_hi;
    for (int g = 0; g < ng; ++g) acc = sw_add(acc, pm_load(L.part + ((size_t)t * L.groups + g) * PM_PT));
  }
  {
    PtJ j = sw_to_jac(acc);
    const int nd = t < PM_W ? PM_C * t : 0;
#pragma unroll 1
    for (int k = 0; k < PM_C * (PM_W - 1); ++k)
      if (k < nd) j = sw_dbl_jac(j);
    acc = sw_from_jac(j);
  }
#pragma unroll 1
  for (int s = 16; s >= 1; s >>= 1) {
    __syncthreads();
    if (t < 32) pm_store(stage + t * PM_PT, acc);
    __syncthreads();
    if (t < s) acc = sw_add(acc, pm_load(stage + (t + s) * PM_PT));
  }
  if (t != 0) return;
  const bool bad = L.flags[0] != 0;
  const bool inf = sw_is_infinity(acc);
  if (fail_flag && (!inf || bad)) fail_flag[0] = 1;
  if (status1) status1[0] = bad ? 2 : 0;
  if (!out33 && !out_xy) return;
  uint32_t xw[8], yw[8];
  uint32_t tag = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) { xw[j] = 0; yw[j] = 0; }
  if (!inf && !bad) {
    const FeN zi = fe_inv<false>(acc.Z);
    tag = sec1_words(xw, fe_mul(acc.X, zi), fe_mul(acc.Y, zi), yw);
  }
  // the point at infinity (and an invalid input): Sec1 0x00 followed by zeros, x || y all zero
  if (out33) sec1_store(out33, tag, xw);
  if (out_xy) xy_store(out_xy, tag, xw, yw, xy_mont256 != 0);
}

// ------------------------------------------------------------------------------- batched Pedersen verifier
// MSM index of point class p (H, Gamma, pk_com | G, B | Ok, R) of proof i: full-size scalars first, then the two fixed
// bases, then the classes whose scalars are the bare 128-bit weights
__device__ __forceinline__ size_t pm_rlc_index(int p, size_t n, size_t i) { return (size_t)p * n + i + (p >= 3 ? 2 : 0); }

__global__ void __launch_bounds__(128) k_pm_rlc_decode(p256::RlcArgs a) {
  const size_t i = (size_t)blockIdx.x * 128 + threadIdx.x;
  if (i >= a.n) return;
  const uint8_t* ad;
  uint32_t ad_len;
  bytes_lite_get(a.ad, i, ad, ad_len);
  FeN x[5], y[5];
  uint32_t c[8], s[8], sb[8];
  bool ok;
  if (a.affine_in) {
    // typed callers: the five points as x || y (no square root); the Sec1 strings the challenge hashes are rebuilt from
    // (x, parity of y); a coordinate >= p or a point off the curve is InvalidData
    ok = true;
    Sec1W enc[5];
#pragma unroll 1
    for (int j = 0; j < 5; ++j) {
      const uint8_t* src = j == 0 ? a.h : j == 1 ? a.gamma : j == 2 ? a.pk_com : j == 3 ? a.r : a.ok;
      const uint32_t* e = reinterpret_cast<const uint32_t*>(src + i * 64);
      uint32_t xin[8], yin[8], xc[8], yc[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) { xin[k] = e[k]; yin[k] = e[8 + k]; }
      const FeN xx = fe_from_abi(xc, xin, a.affine_in == 2), yy = fe_from_abi(yc, yin, a.affine_in == 2);
      ok = !u256_ge_q(xin) && !u256_ge_q(yin) && sw_on_curve(xx, yy) && ok;
      Sec1W w;
      w.tag = 2u + (yc[0] & 1u);
#pragma unroll
      for (int k = 0; k < 8; ++k) w.xw[k] = xc[k];
#pragma unroll
      for (int k = 0; k < 5; ++k)
        if (j == k) { x[k] = xx; y[k] = yy; enc[k] = w; }
    }
    const Sec1W pts[5] = {enc[2], enc[0], enc[1], enc[3], enc[4]};
    p256_challenge(c, pts, ad, ad_len, a.str);
    const bool s_ok = p256_scalar_decode(s, a.s + i * 32), sb_ok = p256_scalar_decode(sb, a.sb + i * 32);
    ok = ok && s_ok && sb_ok;
  } else {
    ok = p256_ped_verify_decode_item(x, y, c, s, sb, a.h + i * SEC1_LEN, a.gamma + i * SEC1_LEN, a.pk_com + i * SEC1_LEN,
                                     a.r + i * SEC1_LEN, a.ok + i * SEC1_LEN, a.s + i * 32, a.sb + i * 32, ad, ad_len, a.str);
  }
  a.status[i] = ok ? 0 : 2;
  // z, z' = the two halves of SHA-256("vrfhip-p256-rlc-v1" || seed || batch digest || u64_le(index)), forced odd (non-zero)
  Sha256 hh;
  sha256_init(hh);
  constexpr char tag[] = "vrfhip-p256-rlc-v1";
#pragma unroll
  for (int j = 0; j < 18; ++j) sha256_put_byte(hh, (uint8_t)tag[j]);
  sha256_put_bytes(hh, a.seed, 32);
  sha256_put_bytes(hh, a.root, 32);
  const uint64_t idx = a.index0 + i;
#pragma unroll
  for (int j = 0; j < 8; ++j) sha256_put_byte(hh, (uint8_t)(idx >> (8 * j)));
  sha256_final(hh);
  uint32_t z[8], zp[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) { z[j] = hh.h[j]; zp[j] = hh.h[4 + j]; z[4 + j] = 0; zp[4 + j] = 0; }
  z[0] |= 1u; zp[0] |= 1u;
  uint32_t zs[8], zc[8], zpc[8], zps[8], zpsb[8];
  fr_mul<CurveP256>(zs, z, s);
  fr_mul<CurveP256>(zc, z, c);
  fr_mul<CurveP256>(zpc, zp, c);
  fr_mul<CurveP256>(zps, zp, s);
  fr_mul<CurveP256>(zpsb, zp, sb);
  const MsmL& L = a.L;
  const size_t n = a.n;
  // x / y order of the decode: H, Gamma, pk_com, R, Ok
  pm_store_affine(L.pts + pm_rlc_index(0, n, i) * PM_AFF, x[0], y[0]);       // + z s   H
  pm_write_digits(L.digits, L.n, pm_rlc_index(0, n, i), zs, false, !ok);
  pm_store_affine(L.pts + pm_rlc_index(1, n, i) * PM_AFF, x[1], y[1]);       // - z c   Gamma
  pm_write_digits(L.digits, L.n, pm_rlc_index(1, n, i), zc, true, !ok);
  pm_store_affine(L.pts + pm_rlc_index(2, n, i) * PM_AFF, x[2], y[2]);       // - z' c  pk_com
  pm_write_digits(L.digits, L.n, pm_rlc_index(2, n, i), zpc, true, !ok);
  pm_store_affine(L.pts + pm_rlc_index(3, n, i) * PM_AFF, x[4], y[4]);       // - z     Ok
  pm_write_digits(L.digits, L.n, pm_rlc_index(3, n, i), z, true, !ok);
  pm_store_affine(L.pts + pm_rlc_index(4, n, i) * PM_AFF, x[3], y[3]);       // - z'    R
  pm_write_digits(L.digits, L.n, pm_rlc_index(4, n, i), zp, true, !ok);
  // sum z' s and sum z' sb: 32-bit limbs into 64-bit columns (2^20 terms of 2^32 stay far below 2^64)
  if (ok) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      atomicAdd(reinterpret_cast<unsigned long long*>(a.fixed_cols + j), (unsigned long long)zps[j]);
      atomicAdd(reinterpret_cast<unsigned long long*>(a.fixed_cols + 8 + j), (unsigned long long)zpsb[j]);
    }
  }
}
// the two fixed bases: scalars = the column sums mod n
__global__ void k_pm_rlc_fixed(p256::RlcArgs a) {
  const int f = threadIdx.x;
  if (f >= 2) return;
  // columns -> a 320-bit integer -> mod n (two folds of the top words through 2^256 mod n = r_r1)
  uint32_t w[16];
  unsigned long long carry = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const unsigned long long v = a.fixed_cols[8 * f + j] + carry;
    w[j] = (uint32_t)v;
    carry = v >> 32;
  }
  w[8] = (uint32_t)carry; w[9] = (uint32_t)(carry >> 32);
#pragma unroll
  for (int j = 10; j < 16; ++j) w[j] = 0;
  uint32_t k[8];
  fr_reduce512<CurveP256>(k, w);
  const uint8_t* src = f == 0 ? a.gen_xy : a.b_xy;
  uint32_t xin[8], yin[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    xin[j] = (uint32_t)src[4 * j] | ((uint32_t)src[4 * j + 1] << 8) | ((uint32_t)src[4 * j + 2] << 16) | ((uint32_t)src[4 * j + 3] << 24);
    yin[j] = (uint32_t)src[32 + 4 * j] | ((uint32_t)src[33 + 4 * j] << 8) | ((uint32_t)src[34 + 4 * j] << 16) | ((uint32_t)src[35 + 4 * j] << 24);
  }
  const size_t idx = 3 * a.n + f;
  pm_store_affine(a.L.pts + idx * PM_AFF, fe_from_u256(xin), fe_from_u256(yin));
  pm_write_digits(a.L.digits, a.L.n, idx, k, false, false);
}

size_t pad256(size_t x) { return (x + 255) & ~size_t(255); }

void launch_core(const MsmL& L, uint8_t* out33, uint8_t* out_xy, int xy_mont256, uint8_t* status1, uint8_t* fail_flag, hipStream_t st,
                 hipEvent_t* ev) {
  const size_t lds_bytes = ((size_t)PM_BUCKETS * PM_PT + 2 * PM_BUCKETS + 16) * 4;
  const unsigned wgs = (unsigned)(p256::MSM_W_SHORT * L.groups + (PM_W - p256::MSM_W_SHORT) * L.groups_hi);
  hipLaunchKernelGGL(k_pm_buckets, dim3(wgs), dim3(PM_BLOCK), lds_bytes, st, L);
  if (ev) (void)hipEventRecord(ev[2], st);
  hipLaunchKernelGGL(k_pm_final, dim3(1), dim3(64), 0, st, L, out33, out_xy, xy_mont256, status1, fail_flag);
  if (ev) { (void)hipEventRecord(ev[3], st); (void)hipEventRecord(ev[4], st); }
}

__global__ void __launch_bounds__(128) k_pm_affine_compress(size_t n, const uint8_t* xy, uint8_t* enc, int mont256) {
  const size_t i = (size_t)blockIdx.x * 128 + threadIdx.x;
  if (i >= n) return;
  const uint32_t* w = reinterpret_cast<const uint32_t*>(xy + i * 64);
  uint32_t xin[8], yin[8], xw[8], yw[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { xin[j] = w[j]; yin[j] = w[8 + j]; }
  const bool bad = u256_ge_q(xin) || u256_ge_q(yin);
  (void)fe_from_abi(xw, xin, mont256 != 0);         // xw, yw: the canonical words
  (void)fe_from_abi(yw, yin, mont256 != 0);
  sec1_store(enc + i * SEC1_LEN, bad ? 0xffu : 2u + (yw[0] & 1u), xw);
}
}  // namespace
VRF_NS_END

namespace vrf {
namespace p256 {

int msm_groups(size_t n, size_t n_long, int cus) {
  // The bucket kernel holds 176 registers: two waves per SIMD, so ONE 512-lane workgroup per CU whatever its LDS (measured:
  // 208 workgroups ran as one round, 260 as two -- profiles/r04/knob_sweep.log).  Workgroups = MSM_W_SHORT g in the low windows
  // + (MSM_W - MSM_W_SHORT) g n_long / n in the high ones: one round of the chip for mid-sized inputs, two for large ones.
  const int rounds = n >= (size_t(1) << 22) ? 2 : 1;
  const double share = n ? (double)(n_long < n ? n_long : n) / (double)n : 1.0;
  int g = (int)(rounds * cus / (MSM_W_SHORT + (MSM_W - MSM_W_SHORT) * share));
  if (g < 1) g = 1;
  auto wgs = [&](int gg) { return MSM_W_SHORT * gg + (MSM_W - MSM_W_SHORT) * (int)(share * gg + 0.999); };
  while (g > 1 && wgs(g) > rounds * cus) --g;
  const size_t max_g = (n + 4095) / 4096;          // at least 8 points per lane and group
  if ((size_t)g > max_g) g = (int)max_g;
  if (g < 1) g = 1;
  const size_t min_g = (n + (size_t(1) << PM_IDX_BITS) - 1) >> PM_IDX_BITS;
  if ((size_t)g < min_g) g = (int)min_g;
  return g;
}
size_t msm_workspace_bytes(size_t n, int groups) {
  const size_t per_group = (n + groups - 1) / groups, list_cap = per_group + PM_BLOCK, wgs = (size_t)MSM_W * groups;
  return pad256(n * PM_AFF * 4) + pad256((size_t)MSM_W * n * 2) + pad256(wgs * list_cap * 4) + pad256(wgs * PM_BLOCK * PM_PT * 4) +
         pad256(wgs * PM_PT * 4) + 256 + 256;
}
MsmL msm_layout(size_t n, size_t n_long, int groups, void* ws) {
  MsmL L;
  L.n = n; L.groups = groups;
  L.n_long = n_long < n ? n_long : n;
  L.per_group = (n + groups - 1) / groups;
  int gh = (int)(((double)L.n_long / (double)(n ? n : 1)) * groups + 0.999);
  if (gh < 1) gh = 1;
  if (gh > groups) gh = groups;
  L.groups_hi = gh;
  L.per_group_hi = (L.n_long + gh - 1) / gh;
  if (L.per_group_hi > L.per_group) { L.groups_hi = groups; L.per_group_hi = (L.n_long + groups - 1) / groups; }
  L.list_cap = L.per_group + PM_BLOCK;
  const size_t wgs = (size_t)MSM_W * groups;
  uint8_t* p = static_cast<uint8_t*>(ws);
  L.pts = reinterpret_cast<uint32_t*>(p); p += pad256(n * PM_AFF * 4);
  L.digits = reinterpret_cast<int16_t*>(p); p += pad256((size_t)MSM_W * n * 2);
  L.lists = reinterpret_cast<uint32_t*>(p); p += pad256(wgs * L.list_cap * 4);
  L.heads = reinterpret_cast<uint32_t*>(p); p += pad256(wgs * PM_BLOCK * PM_PT * 4);
  L.part = reinterpret_cast<uint32_t*>(p); p += pad256(wgs * PM_PT * 4);
  L.flags = p; p += 256;
  L.cols = reinterpret_cast<unsigned long long*>(p);
  return L;
}

void launch_msm(size_t n, const uint8_t* xy, int mont256, const uint8_t* scalars_be, uint8_t* out33, uint8_t* out_xy, uint8_t* status1,
                void* ws, int groups, hipStream_t st) {
  MsmL L = msm_layout(n ? n : 1, n ? n : 1, groups, ws);
  L.n = n; L.n_long = n;
  (void)hipMemsetAsync(L.flags, 0, 256, st);
  if (n == 0) {                                    // the empty sum: the point at infinity
    if (out33) (void)hipMemsetAsync(out33, 0, 33, st);
    if (out_xy) (void)hipMemsetAsync(out_xy, 0, 64, st);
    if (status1) (void)hipMemsetAsync(status1, 0, 1, st);
    return;
  }
  hipLaunchKernelGGL(k_pm_prep, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, st, L, xy, scalars_be, mont256);
  launch_core(L, out33, out_xy, mont256, status1, nullptr, st, nullptr);
}

void launch_pedersen_rlc(const RlcArgs& a_in, uint8_t* fail_flag, hipStream_t st, hipEvent_t* ev) {
  if (!a_in.n) return;
  RlcArgs a = a_in;
  a.fixed_cols = a.L.cols;
  (void)hipMemsetAsync(a.L.flags, 0, 512, st);     // flags and the 2 x 8 columns behind them
  if (ev) (void)hipEventRecord(ev[0], st);
  hipLaunchKernelGGL(k_pm_rlc_decode, dim3((unsigned)((a.n + 127) / 128)), dim3(128), 0, st, a);
  hipLaunchKernelGGL(k_pm_rlc_fixed, dim3(1), dim3(64), 0, st, a);
  if (ev) (void)hipEventRecord(ev[1], st);
  launch_core(a.L, nullptr, nullptr, 0, nullptr, fail_flag, st, ev);
}

void launch_affine_compress(size_t n, const uint8_t* xy, int mont256, uint8_t* enc33, hipStream_t st) {
  if (n) hipLaunchKernelGGL(k_pm_affine_compress, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, st, n, xy, enc33, mont256);
}

}  // namespace p256
}  // namespace vrf
