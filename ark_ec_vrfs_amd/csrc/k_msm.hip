// k_msm.hip -- variable-base multi-scalar multiplication on the suite curve (SURVEY.md section 8
// row a14).  Replaces ark_ec::scalar_mul::VariableBaseMSM::msm for twisted-Edwards affine bases
// (named in BASELINE.json north_star; reached from /root/reference through `reexports`,
// src/lib.rs:14).
//
// Pippenger with signed 11-bit windows (23 windows x 1024 buckets).  Scalars k > r/2 are replaced
// by r - k with the point's sign flipped, so k < 2^252 and the top window never carries out (a
// 24th carry-only window would put half of all points into a single bucket).
//   k_msm_prep    : bases -> Montgomery affine-cached (x, y, d*x*y); scalars -> signed digits
//   k_msm_buckets : one workgroup per (window, point-group).  Per slice of 16384 points the
//                   workgroup counting-sorts the slice by bucket in LDS (LDS atomics, block scan),
//                   then every lane sums the lists of the two buckets it owns with its
//                   accumulators kept in registers.  After the last slice the 1024 buckets are
//                   reduced to sum_j j*B_j inside the workgroup: pair-local sums, a suffix scan and
//                   a tree reduction over 512 lanes staged through LDS.
//   k_msm_final   : sums the groups of every window, then Horner over the 24 windows.
#include "kernels.h"
#include <cstdlib>

namespace vrf {

constexpr int MSM_C = 11;                         // window bits
constexpr int MSM_W = 23;                         // windows: 11*23 = 253 bits; scalars are folded to < r/2
constexpr int MSM_BUCKETS = 1 << (MSM_C - 1);     // 1024 signed buckets
constexpr int MSM_BLOCK = 512;                    // lanes per workgroup; each owns 2 buckets
constexpr int MSM_SLICE = 16384;                  // points sorted per pass
constexpr int MSM_PT_WORDS = 4 * NL;              // extended point staged in LDS: X, Y, Z, T

// ------------------------------------------------------------------------------- prep
__global__ void __launch_bounds__(BLOCK) k_msm_prep(size_t n, const uint8_t* xy, const uint8_t* scalars,
                                                     uint32_t* pts, int16_t* digits, uint8_t* flags) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  const uint32_t* p = reinterpret_cast<const uint32_t*>(xy + i * 64);
  uint32_t xw[8], yw[8], k[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { xw[j] = p[j]; yw[j] = p[8 + j]; }
  load32(k, scalars, i);
  bool ok = !u256_ge(xw, vrfk::Q32) && !u256_ge(yw, vrfk::Q32) && fr_is_canonical<SuiteBS>(k);
  PtA a;
  a.x = fe_from_u256(xw);
  a.y = fe_from_u256(yw);
  a.dt = fe_mul(fe_mul(a.x, a.y), SuiteBS::d());
  // on-curve check: a*x^2 + y^2 == 1 + d*x^2*y^2  <=>  y^2 - 5 x^2 - 1 - (d x y) * (x y) == 0
  FeN x2 = fe_sqr(a.x), y2 = fe_sqr(a.y), xy_ = fe_mul(a.x, a.y);
  auto lhs = fe_norm(fe_add(y2, fe_neg(fe_norm(fe_add(fe_mul5(x2), fe_one())))));
  ok = ok && fe_eq(lhs, fe_mul(a.dt, xy_));
  pta_store(pts + i * PTA_WORDS, a);
  if (!ok) flags[0] = 1;
  // fold: k > r - k  ->  use r - k and flip the sign of every digit
  uint32_t nk[8];
  {
    uint32_t borrow = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      uint64_t d = (uint64_t)SuiteBS::r32(j) - k[j] - borrow;
      nk[j] = (uint32_t)d;
      borrow = (uint32_t)(d >> 63);
    }
  }
  bool flip = false, decided = false;
#pragma unroll
  for (int j = 7; j >= 0; --j)
    if (!decided && nk[j] != k[j]) { flip = nk[j] < k[j]; decided = true; }
  flip = flip && ok;
#pragma unroll
  for (int j = 0; j < 8; ++j) k[j] = flip ? nk[j] : k[j];
  // signed radix-2^11 digits in [-1023, 1024]
  uint32_t carry = 0;
#pragma unroll 1
  for (int w = 0; w < MSM_W; ++w) {
    int bit = w * MSM_C, wi = bit >> 5, sh = bit & 31;
    uint32_t lo = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) if (j == wi) lo = k[j];
    uint32_t hi = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) if (j == wi + 1) hi = k[j];
    uint32_t v = (sh ? ((lo >> sh) | (hi << (32 - sh))) : lo) & ((1u << MSM_C) - 1);
    if (wi >= 8) v = 0;
    v += carry;
    int d = (int)v;
    carry = 0;
    if (v > (uint32_t)MSM_BUCKETS) { d = (int)v - (1 << MSM_C); carry = 1; }
    digits[(size_t)w * n + i] = (int16_t)(flip ? -d : d);
  }
}

// ------------------------------------------------------------------------------- buckets
VRF_HD void lds_store_pt(uint32_t* dst, const PtE& p) {
  fe_store(dst, p.X); fe_store(dst + NL, p.Y); fe_store(dst + 2 * NL, p.Z); fe_store(dst + 3 * NL, p.T);
}
VRF_HD PtE lds_load_pt(const uint32_t* src) {
  PtE p;
  p.X = fe_load<1, 5>(src); p.Y = fe_load<1, 5>(src + NL); p.Z = fe_load<1, 5>(src + 2 * NL);
  p.T = fe_load<1, 5>(src + 3 * NL);
  return p;
}

__global__ void __launch_bounds__(MSM_BLOCK) k_msm_buckets(size_t n, const uint32_t* pts, const int16_t* digits,
                                                            uint32_t* lists, uint32_t* out, int groups, int dbg) {
  extern __shared__ uint32_t lds[];
  uint32_t* bucket = lds;                                   // [1024][36] bucket accumulators (147,456 B)
  uint32_t* counts = lds + MSM_BUCKETS * MSM_PT_WORDS;      // [1024] bucket sizes, then exclusive offsets
  uint32_t* cursor = counts + MSM_BUCKETS;                  // [1024] scatter cursors
  uint32_t* wsum = cursor + MSM_BUCKETS;                    // [8] per-wave totals for the block scan
  const int w = blockIdx.x, g = blockIdx.y, t = threadIdx.x;
  const size_t per_group = (n + groups - 1) / groups;
  const size_t lo = (size_t)g * per_group, hi = lo + per_group < n ? lo + per_group : n;
  const int16_t* dig = digits + (size_t)w * n;
  uint32_t* list = lists + ((size_t)w * groups + g) * MSM_SLICE;
  // lane t owns buckets 2t and 2t+1 (digit magnitudes 2t+1 and 2t+2), resident in LDS
  {
    PtE id = te_identity();
    lds_store_pt(bucket + (2 * t) * MSM_PT_WORDS, id);
    lds_store_pt(bucket + (2 * t + 1) * MSM_PT_WORDS, id);
  }
  for (size_t base = lo; base < hi; base += MSM_SLICE) {
    const int m = (int)(hi - base < (size_t)MSM_SLICE ? hi - base : (size_t)MSM_SLICE);
    // 1. histogram of the slice
    counts[t] = 0; counts[t + MSM_BLOCK] = 0;
    __syncthreads();
    for (int j = t; j < m; j += MSM_BLOCK) {
      int d = dig[base + j];
      if (d != 0) atomicAdd(&counts[(d < 0 ? -d : d) - 1], 1u);
    }
    __syncthreads();
    // 2. exclusive scan over 1024 counts: lane t scans its pair, waves scan by shuffles
    uint32_t c0 = counts[2 * t], c1 = counts[2 * t + 1];
    uint32_t v = c0 + c1, incl = v;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
      uint32_t u = __shfl_up(incl, s, 64);
      if ((t & 63) >= s) incl += u;
    }
    if ((t & 63) == 63) wsum[t >> 6] = incl;
    __syncthreads();
    uint32_t wbase = 0;
    for (int k = 0; k < (t >> 6); ++k) wbase += wsum[k];
    uint32_t excl = wbase + incl - v;
    __syncthreads();
    cursor[2 * t] = excl; cursor[2 * t + 1] = excl + c0;
    __syncthreads();
    // 3. scatter point indices (sign in bit 31) into bucket order
    for (int j = t; j < m; j += MSM_BLOCK) {
      int d = dig[base + j];
      if (d != 0) {
        uint32_t pos = atomicAdd(&cursor[(d < 0 ? -d : d) - 1], 1u);
        list[pos] = (uint32_t)j | (d < 0 ? 0x80000000u : 0u);
      }
    }
    __threadfence_block();
    __syncthreads();
    // 4. each lane folds the lists of its two buckets into the LDS accumulators
#pragma unroll 1
    for (int b = 0; b < 2; ++b) {
      const uint32_t q0 = b ? excl + c0 : excl, q1 = b ? excl + c0 + c1 : excl + c0;
      uint32_t* slot = bucket + (2 * t + b) * MSM_PT_WORDS;
      PtE acc = lds_load_pt(slot);
      for (uint32_t q = q0; q < q1 && !(dbg & 1); ++q) {
        uint32_t ent = list[q];
        PtA pa = pta_load(pts + (base + (ent & 0x7fffffffu)) * PTA_WORDS);
        acc = te_add_affine<SuiteBS>(acc, pa, (ent >> 31) != 0);
      }
      lds_store_pt(slot, acc);
    }
    __syncthreads();
  }
  // ---- bucket reduction: R = sum_j j*B_j over j = 1..1024, lane t holds B_{2t+1}, B_{2t+2} ----
  // S_t = B_{2t+1} + B_{2t+2};  L_t = S_t + B_{2t+2};  R = sum_t L_t + 2 sum_{t>=1} Suf_t,
  // Suf_t = sum_{u>=t} S_u.  The staging area re-uses the bucket storage, one slot per lane.
  PtE S, V;
  {
    PtE b1 = lds_load_pt(bucket + (2 * t + 1) * MSM_PT_WORDS);
    S = te_add<SuiteBS>(lds_load_pt(bucket + (2 * t) * MSM_PT_WORDS), b1);
    V = te_add<SuiteBS>(S, b1);                   // L_t
  }
  __syncthreads();
  uint32_t* stage = lds;                          // [512][36]
  // pass 0: suffix scan of S (Hillis-Steele);  pass 1: tree reduction of V = L + 2*Suf
#pragma unroll 1
  for (int pass = 0; pass < 2 && !(dbg & 2); ++pass) {
    PtE cur = pass == 0 ? S : V;
#pragma unroll 1
    for (int k = 0; k < 9; ++k) {
      const int s = pass == 0 ? (1 << k) : (MSM_BLOCK >> (k + 1));
      __syncthreads();
      lds_store_pt(stage + t * MSM_PT_WORDS, cur);
      __syncthreads();
      const bool active = pass == 0 ? (t + s < MSM_BLOCK) : (t < s);
      if (active) cur = te_add<SuiteBS>(cur, lds_load_pt(stage + (t + s) * MSM_PT_WORDS));
    }
    if (pass == 0) {
      if (t >= 1) V = te_add<SuiteBS>(V, te_dbl<SuiteBS>(cur, true));
    } else {
      V = cur;
    }
  }
  if (t == 0) lds_store_pt(out + ((size_t)w * groups + g) * MSM_PT_WORDS, V);
}

// ------------------------------------------------------------------------------- final
__global__ void __launch_bounds__(64) k_msm_final(const uint32_t* part, int groups, uint8_t* out_enc,
                                                   uint8_t* out_xy, const uint8_t* flags, uint8_t* status) {
  __shared__ uint32_t stage[32 * MSM_PT_WORDS];
  const int t = threadIdx.x;
  // lane w < 23: R_w = sum over groups, then 2^(11 w) * R_w by 11 w doublings (lanes run in lockstep,
  // the critical path is the top window's 242 doublings instead of a serial 253-doubling Horner)
  PtE acc = te_identity();
  if (t < MSM_W) {
    acc = lds_load_pt(part + (size_t)t * groups * MSM_PT_WORDS);
    for (int g = 1; g < groups; ++g)
      acc = te_add<SuiteBS>(acc, lds_load_pt(part + ((size_t)t * groups + g) * MSM_PT_WORDS));
    const int nd = MSM_C * t;
    for (int j = 0; j < nd; ++j) acc = te_dbl<SuiteBS>(acc, j == nd - 1);
  }
  // tree-sum of 32 lanes through LDS
#pragma unroll 1
  for (int s = 16; s >= 1; s >>= 1) {
    __syncthreads();
    if (t < 32) lds_store_pt(stage + t * MSM_PT_WORDS, acc);
    __syncthreads();
    if (t < s) acc = te_add<SuiteBS>(acc, lds_load_pt(stage + (t + s) * MSM_PT_WORDS));
  }
  if (t == 0) {
    FeN x, y;
    te_to_affine(x, y, acc);
    uint32_t e[8], xw[8], yw[8];
    te_encode_affine(e, x, y);
    fe_to_u256(xw, x); fe_to_u256(yw, y);
    bool bad = flags[0] != 0;
    if (bad) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { e[j] = 0; xw[j] = 0; yw[j] = 0; }
    }
    store32(out_enc, 0, e);
    if (out_xy) {
      uint32_t* p = reinterpret_cast<uint32_t*>(out_xy);
#pragma unroll
      for (int j = 0; j < 8; ++j) { p[j] = xw[j]; p[8 + j] = yw[j]; }
    }
    status[0] = bad ? ST_INVALID_DATA : ST_OK;
  }
}

size_t msm_workspace_bytes(size_t n, int groups) {
  size_t pts = n * PTA_WORDS * 4, dig = (size_t)MSM_W * n * 2;
  size_t lists = (size_t)MSM_W * groups * MSM_SLICE * 4, part = (size_t)MSM_W * groups * MSM_PT_WORDS * 4;
  auto pad = [](size_t x) { return (x + 255) & ~size_t(255); };
  return pad(pts) + pad(dig) + pad(lists) + pad(part) + 256;
}
int msm_groups(size_t n, int cus) {
  // one workgroup per CU and window pass; at least one slice of work per group
  int g = cus / MSM_W;
  if (g < 1) g = 1;
  size_t max_g = (n + MSM_SLICE - 1) / MSM_SLICE;
  if ((size_t)g > max_g) g = (int)max_g;
  if (g < 1) g = 1;
  return g;
}

static bool g_msm_attr_set = false;
void launch_msm(size_t n, const uint8_t* xy, const uint8_t* scalars, uint8_t* out_enc, uint8_t* out_xy,
                uint8_t* status, void* ws, int groups, hipStream_t st) {
  auto pad = [](size_t x) { return (x + 255) & ~size_t(255); };
  uint8_t* p = static_cast<uint8_t*>(ws);
  uint32_t* pts = reinterpret_cast<uint32_t*>(p); p += pad(n * PTA_WORDS * 4);
  int16_t* digits = reinterpret_cast<int16_t*>(p); p += pad((size_t)MSM_W * n * 2);
  uint32_t* lists = reinterpret_cast<uint32_t*>(p); p += pad((size_t)MSM_W * groups * MSM_SLICE * 4);
  uint32_t* part = reinterpret_cast<uint32_t*>(p); p += pad((size_t)MSM_W * groups * MSM_PT_WORDS * 4);
  uint8_t* flags = p;
  (void)hipMemsetAsync(flags, 0, 256, st);
  hipLaunchKernelGGL(k_msm_prep, grid_for(n), dim3(BLOCK), 0, st, n, xy, scalars, pts, digits, flags);
  size_t lds_bytes = ((size_t)MSM_BUCKETS * MSM_PT_WORDS + 2 * MSM_BUCKETS + 8) * 4;   // 155,680 B of 160 KiB
  if (!g_msm_attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_msm_buckets),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    g_msm_attr_set = true;
  }
  hipLaunchKernelGGL(k_msm_buckets, dim3(MSM_W, groups), dim3(MSM_BLOCK), lds_bytes, st, n, pts, digits, lists,
                     part, groups, getenv("VRFHIP_MSM_DBG") ? atoi(getenv("VRFHIP_MSM_DBG")) : 0);
  hipLaunchKernelGGL(k_msm_final, dim3(1), dim3(64), 0, st, part, groups, out_enc, out_xy, flags, status);
}

}  // namespace vrf
