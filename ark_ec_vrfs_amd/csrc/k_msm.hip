// k_msm.hip -- variable-base multi-scalar multiplication on the suite curve (SURVEY.md section 8
// row a14).  Replaces ark_ec::scalar_mul::VariableBaseMSM::msm for twisted-Edwards affine bases
// (named in BASELINE.json north_star; reached from /root/reference through `reexports`,
// src/lib.rs:14).
//
// Pippenger with signed 11-bit windows (msm.cuh).
//   k_msm_prep    : bases -> Montgomery affine-cached (x, y, d*x*y); scalars -> signed digits
//   k_msm_buckets : one workgroup per (window, point-group); the 1024 buckets of the window live in LDS.
//                   The workgroup counting-sorts its points by bucket (LDS atomics + wave-shuffle block
//                   scan), then the sorted list is cut into 512 EQUAL chunks, one per lane: every lane
//                   performs the same number of mixed additions whatever the digit distribution.  A
//                   lane keeps the running sum of the current bucket in registers and flushes it when
//                   the bucket changes: runs that start inside the chunk go straight to their LDS
//                   bucket (exactly one lane owns the start of a bucket), the first run of a chunk --
//                   possibly the continuation of the previous lane's bucket -- is parked as a "head"
//                   and merged afterwards by the first lane of each chain.  Then the 1024 buckets are
//                   reduced to sum_j j*B_j inside the workgroup: pair-local sums, a suffix scan and a
//                   tree reduction over 512 lanes staged through LDS.
//   k_msm_final   : sums the groups of every window, shifts window w by 11*w doublings (one DPP quad per
//                   window, the quad's lanes share each doubling), tree-sums the windows, encodes.
#include "kernels.h"
#include "msm.cuh"
#include <cstdlib>

VRF_NS_BEGIN

// ------------------------------------------------------------------------------- prep
template <class S>
__global__ void __launch_bounds__(BLOCK) k_msm_prep(size_t n, const uint8_t* xy, const uint8_t* scalars,
                                                     uint32_t* pts, int16_t* digits, uint8_t* flags, int mont256) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  const uint32_t* p = reinterpret_cast<const uint32_t*>(xy + i * 64);
  uint32_t xw[8], yw[8], k[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { xw[j] = p[j]; yw[j] = p[8 + j]; }
  load32(k, scalars, i);
  bool ok = !u256_ge(xw, vrfk::Q32) && !u256_ge(yw, vrfk::Q32) && fr_is_canonical<S>(k);
  PtA a;
  uint32_t cw[8];
  a.x = fe_from_abi(cw, xw, mont256 != 0);
  a.y = fe_from_abi(cw, yw, mont256 != 0);
  a.dt = fe_mul(fe_mul(a.x, a.y), S::d());
  // on-curve check: a*x^2 + y^2 == 1 + d*x^2*y^2  <=>  y^2 - ANEG x^2 - 1 - (d x y) * (x y) == 0
  FeN x2 = fe_sqr(a.x), y2 = fe_sqr(a.y), xy_ = fe_mul(a.x, a.y);
  auto lhs = te_curve_lhs<S>(x2, y2);
  ok = ok && fe_eq(lhs, fe_mul(a.dt, xy_));
  pta_store(pts + i * MSM_PTA_STRIDE, a);
  if (!ok) flags[0] = 1;
  msm_write_digits<S>(digits, n, i, k, false, !ok);
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

constexpr uint32_t MSM_NONE = 0xffffffffu;

template <class S>
__global__ void __launch_bounds__(MSM_BLOCK) k_msm_buckets(MsmLayout L) {
  extern __shared__ uint32_t lds[];
  uint32_t* bucket = lds;                                   // [MSM_BUCKETS][36] bucket accumulators (147,456 B)
  uint32_t* counts = lds + MSM_BUCKETS * MSM_PT_WORDS;      // [1024] bucket sizes
  uint32_t* cursor = counts + MSM_BUCKETS;                  // [1024] scatter cursors, later head bucket ids
  uint32_t* wsum = cursor + MSM_BUCKETS;                    // [8] per-wave totals for the block scan
  const int t = threadIdx.x;
  // low windows (all points) first, then the high windows (only the points with full-size scalars)
  // GROUP-major: consecutive workgroups are the windows of ONE point group, so the workgroups that run at the same time
  // gather from the same few slices of the point array and find each other's lines in the Infinity Cache / L2 (window-major
  // order streamed the whole array from HBM once per window: 16.6 GB per batched-Pedersen launch, 2.2 TB/s of random 128-B
  // reads -- the bound behind the 0.72 of the issue slots)
  const int wg = blockIdx.x;
  const int n_lo_wgs = MSM_W_SHORT * L.groups;
  const bool low = wg < n_lo_wgs;
  const int w = low ? wg % MSM_W_SHORT : MSM_W_SHORT + (wg - n_lo_wgs) % (MSM_W - MSM_W_SHORT);
  const int g = low ? wg / MSM_W_SHORT : (wg - n_lo_wgs) / (MSM_W - MSM_W_SHORT);
  const size_t span = low ? L.per_group : L.per_group_hi, end = low ? L.n : L.n_long;
  const size_t lo = (size_t)g * span;
  const size_t hi = lo + span < end ? lo + span : end;
  const uint32_t cnt_all = lo < hi ? (uint32_t)(hi - lo) : 0u;
  const int16_t* dig = L.digits + (size_t)w * L.n + lo;
  const uint32_t* P = L.pts + lo * MSM_PTA_STRIDE;
  uint32_t* list = L.lists + (size_t)wg * L.list_cap;
  uint32_t* heads = L.heads + (size_t)wg * MSM_BLOCK * MSM_PT_WORDS;
  {
    PtE id = te_identity();
#pragma unroll
    for (int u = 0; u < MSM_BPL; ++u) lds_store_pt(bucket + (MSM_BPL * t + u) * MSM_PT_WORDS, id);
  }
  // 1. histogram of the group's digits
#pragma unroll
  for (int u = 0; u < MSM_BPL; ++u) counts[t + u * MSM_BLOCK] = 0;
  __syncthreads();
  // four independent digit loads per trip keep the memory pipe busy (a lane's trip is latency-bound)
  for (uint32_t j0 = t; j0 < cnt_all; j0 += 4 * MSM_BLOCK) {
    int d[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t j = j0 + u * MSM_BLOCK;
      d[u] = j < cnt_all ? dig[j] : 0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (d[u] != 0) atomicAdd(&counts[(d[u] < 0 ? -d[u] : d[u]) - 1], 1u);
  }
  __syncthreads();
  // 2. exclusive scan over 1024 counts: lane t scans its pair, waves scan by shuffles
  uint32_t m;
  {
    uint32_t c0 = counts[MSM_BPL * t], c1 = MSM_BPL == 2 ? counts[MSM_BPL * t + MSM_BPL - 1] : 0u;
    uint32_t v = c0 + c1, incl = v;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
      uint32_t u = __shfl_up(incl, s, 64);
      if ((t & 63) >= s) incl += u;
    }
    if ((t & 63) == 63) wsum[t >> 6] = incl;
    __syncthreads();
    uint32_t wbase = 0, total = 0;
    for (int k = 0; k < MSM_BLOCK / 64; ++k) {
      uint32_t x = wsum[k];
      if (k < (t >> 6)) wbase += x;
      total += x;
    }
    m = total;
    uint32_t excl = wbase + incl - v;
    cursor[MSM_BPL * t] = excl;
    if (MSM_BPL == 2) cursor[MSM_BPL * t + MSM_BPL - 1] = excl + c0;
  }
  __syncthreads();
  // 3. scatter into bucket order, transposed so that entry i of lane l sits at list[i*512 + l]
  const uint32_t chunk = (m + MSM_BLOCK - 1) / MSM_BLOCK;
  for (uint32_t j0 = t; j0 < cnt_all; j0 += 4 * MSM_BLOCK) {
    int dd[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t j = j0 + u * MSM_BLOCK;
      dd[u] = j < cnt_all ? dig[j] : 0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int d = dd[u];
      const uint32_t j = j0 + u * MSM_BLOCK;
      if (d != 0) {
        uint32_t b = (uint32_t)(d < 0 ? -d : d) - 1;
        uint32_t pos = atomicAdd(&cursor[b], 1u);
        uint32_t lane = pos / chunk, i = pos - lane * chunk;
        list[(size_t)i * MSM_BLOCK + lane] = j | (b << MSM_IDX_BITS) | (d < 0 ? 0x80000000u : 0u);
      }
    }
  }
  __threadfence_block();
  __syncthreads();
  // 4. every lane folds its chunk; same trip count for all lanes
  uint32_t head_b = MSM_NONE;
  {
    const uint32_t my0 = (uint32_t)t * chunk;
    const uint32_t cnt = my0 >= m ? 0u : (m - my0 < chunk ? m - my0 : chunk);
    PtE acc = te_identity();
    uint32_t cur = MSM_NONE;
    bool first_run = true;
    uint32_t* myhead = heads + (size_t)t * MSM_PT_WORDS;
    // Two loads feed an addition -- the list entry, then the point it names -- and the second address depends on the first:
    // entries are fetched TWO trips ahead and points one trip ahead, so that neither latency is exposed (with both one
    // trip ahead the gather waited for its own index: 0.70 of the issue slots).
    uint32_t ent_n = 0, ent_nn = 0;
    PtA pa_n = pta_identity();
    if (cnt > 0) ent_n = list[t];
    if (cnt > 1) ent_nn = list[(size_t)MSM_BLOCK + t];
    if (cnt > 0) pa_n = pta_load(P + (size_t)(ent_n & MSM_IDX_MASK) * MSM_PTA_STRIDE);
#pragma unroll 1
    for (uint32_t i = 0; i < chunk; ++i) {
      if (i < cnt) {
        const uint32_t ent = ent_n;
        const PtA pa = pa_n;
        ent_n = ent_nn;
        if (i + 2 < cnt) ent_nn = list[(size_t)(i + 2) * MSM_BLOCK + t];
        if (i + 1 < cnt) pa_n = pta_load(P + (size_t)(ent_n & MSM_IDX_MASK) * MSM_PTA_STRIDE);
        const uint32_t b = (ent >> MSM_IDX_BITS) & (MSM_BUCKETS - 1);
        if (b != cur) {
          if (cur != MSM_NONE) {
            if (first_run) { lds_store_pt(myhead, acc); head_b = cur; first_run = false; }
            else lds_store_pt(bucket + cur * MSM_PT_WORDS, acc);
          }
          cur = b;
          acc = te_identity();
        }
        acc = te_add_affine<S>(acc, pa, (ent >> 31) != 0);
      }
    }
    if (cnt > 0) {
      if (first_run) { lds_store_pt(myhead, acc); head_b = cur; }
      else lds_store_pt(bucket + cur * MSM_PT_WORDS, acc);
    }
  }
  cursor[t] = head_b;
  __threadfence_block();
  __syncthreads();
  // 5. merge the heads: the first lane of each chain of equal head buckets adds the chain to the bucket
  if (head_b != MSM_NONE && (t == 0 || cursor[t - 1] != head_b)) {
    PtE h = lds_load_pt(heads + (size_t)t * MSM_PT_WORDS);
    for (int k = t + 1; k < MSM_BLOCK && cursor[k] == head_b; ++k)
      h = te_add<S>(h, lds_load_pt(heads + (size_t)k * MSM_PT_WORDS));
    uint32_t* slot = bucket + head_b * MSM_PT_WORDS;
    lds_store_pt(slot, te_add<S>(lds_load_pt(slot), h));
  }
  __syncthreads();
  // ---- bucket reduction: R = sum_j j*B_j.  One bucket per lane (10-bit windows): lane t holds B_{t+1}; a suffix scan gives
  // S_t = sum_{u >= t} B_{u+1} and R = sum_t S_t.  Two per lane (11-bit): S_t = B_{2t+1} + B_{2t+2}, L_t = S_t + B_{2t+2},
  // R = sum_t L_t + 2 sum_{t >= 1} Suf_t with Suf the suffix sums of S.  The staging area re-uses the bucket storage. ----
  PtE Ssum, V;
  if (MSM_BPL == 2) {
    PtE b1 = lds_load_pt(bucket + (2 * t + 1) * MSM_PT_WORDS);
    Ssum = te_add<S>(lds_load_pt(bucket + (2 * t) * MSM_PT_WORDS), b1);
    V = te_add<S>(Ssum, b1);                      // L_t
  } else {
    Ssum = lds_load_pt(bucket + t * MSM_PT_WORDS);
    V = te_identity();
  }
  __syncthreads();
  uint32_t* stage = lds;                          // [MSM_BLOCK][36]
  // pass 0: suffix scan of S (Hillis-Steele);  pass 1: tree reduction
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    PtE cur = pass == 0 ? Ssum : V;
#pragma unroll 1
    for (int k = 0; k < MSM_LOG2_BLOCK; ++k) {
      const int s = pass == 0 ? (1 << k) : (MSM_BLOCK >> (k + 1));
      __syncthreads();
      lds_store_pt(stage + t * MSM_PT_WORDS, cur);
      __syncthreads();
      const bool active = pass == 0 ? (t + s < MSM_BLOCK) : (t < s);
      if (active) cur = te_add<S>(cur, lds_load_pt(stage + (t + s) * MSM_PT_WORDS));
    }
    if (pass == 0) {
      if (MSM_BPL == 2) {
        if (t >= 1) V = te_add<S>(V, te_dbl<S>(cur, true));
      } else {
        V = cur;                                  // S_t: the tree below sums them
      }
    } else {
      V = cur;
    }
  }
  if (t == 0) lds_store_pt(L.part + ((size_t)w * L.groups + g) * MSM_PT_WORDS, V);
}

// ------------------------------------------------------------------------------- final
// Quad-parallel point arithmetic: the four lanes of a DPP quad hold the same point and share one
// doubling -- the four squarings, then the four products, of dbl-2008-hwcd run one per lane, and the
// results travel inside the quad by DPP quad_perm broadcasts (VALU moves, no LDS).  The critical path of
// a doubling drops from 4S + 4M to 1S + 1M; that path (242 doublings for the top window) is the whole
// run time of this single-workgroup kernel.
template <int K>
VRF_HD uint32_t quad_bcast_u32(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, K * 0x55, 0xf, 0xf, false);   // quad_perm:[K,K,K,K]
#else
  return v;
#endif
}
template <int K, int L, int V>
VRF_HD Fe<L, V> quad_bcast(const Fe<L, V>& a) {
  Fe<L, V> r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.v[i] = quad_bcast_u32<K>(a.v[i]);
  return r;
}
// bounds of the operands are proved per call site; the typed multiplier is bypassed
template <int L, int V>
VRF_HD const Fe<1, 1>& fe_unchecked(const Fe<L, V>& a) { return reinterpret_cast<const Fe<1, 1>&>(a); }
template <int LO, int VO, int L, int V>
VRF_HD const Fe<LO, VO>& fe_assume(const Fe<L, V>& a) { return reinterpret_cast<const Fe<LO, VO>&>(a); }

// 2P with the work split over the quad; q = lane index inside the quad.  In: every lane holds (X, Y, Z).
// Out: every lane holds (X, Y, Z, T) of 2P.
template <class C>
VRF_HD PtE te_dbl_quad(const PtE& p, int q) {
  // lane q squares one of X, Y, X+Y, Z                              operand bound (2,10)
  Fe<2, 10> xy = fe_add(p.X, p.Y), opnd = xy;
#pragma unroll
  for (int i = 0; i < NL; ++i) opnd.v[i] = q == 0 ? p.X.v[i] : q == 1 ? p.Y.v[i] : q == 2 ? xy.v[i] : p.Z.v[i];
  auto sq = fe_sqr(opnd);                                            // typed (1,3) for the widest operand
  // lanes 0, 1, 3 squared a coordinate (value < 5q): their squares are < 2q, as in te_dbl
  const FeN A = fe_assume<1, 2>(quad_bcast<0>(sq)), B = fe_assume<1, 2>(quad_bcast<1>(sq)),
            ZZ = fe_assume<1, 2>(quad_bcast<3>(sq));
  const auto Sx = quad_bcast<2>(sq);                                 // (X+Y)^2              (1,3)
  // lane 0: X = E*F, lane 1: Y = G*H, lane 2: Z = F*G, lane 3: T = E*H.  All four factors are normalised
  // (L = 1) and each product obeys the value bounds of te_dbl, so the result fits FeP.
  Fe<1, 1> lhs, rhs;
  auto place = [&](const auto& E, const auto& F, const auto& G, const auto& H) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      lhs.v[i] = (q == 0 || q == 3) ? E.v[i] : q == 1 ? G.v[i] : F.v[i];
      rhs.v[i] = q == 0 ? F.v[i] : q == 2 ? G.v[i] : H.v[i];
    }
  };
  if constexpr (C::A_PLUS_ONE) {
    auto G = fe_norm(fe_add(A, B));                                    // A + B                (1,4)
    auto E = fe_norm(fe_sub(Sx, G));                                   // S - A - B = 2XY      (1,11)
    auto F = fe_norm(fe_sub(G, fe_dbl(ZZ)));                           // G - 2Z^2             (1,12)
    auto H = fe_norm(fe_sub(A, B));                                    // A - B                (1,6)
    static_assert(mul_v(11, 12) <= 5 && mul_v(4, 6) <= 5 && mul_v(12, 4) <= 5 && mul_v(11, 6) <= 5,
                  "quad doubling: products must stay inside FeP");
    place(E, F, G, H);
  } else {
    auto E = fe_norm(fe_sub(fe_add(A, B), Sx));                        // A + B - S            (1,8)
    auto aA = C::mul_aneg(A);
    auto H = fe_norm(fe_add(aA, B));                                   // -a*A + B             (1,12)
    auto G = fe_norm(fe_sub(aA, B));                                   // -a*A - B             (1,14)
    auto F = fe_norm(fe_add(G, fe_dbl(ZZ)));                           // G + 2Z^2             (1,18)
    static_assert(mul_v(8, 18) <= 5 && mul_v(14, 12) <= 5 && mul_v(18, 14) <= 5 && mul_v(8, 12) <= 5,
                  "quad doubling: products must stay inside FeP");
    place(E, F, G, H);
  }
  FeP prod = fe_mul(fe_unchecked(lhs), fe_unchecked(rhs));
  PtE r;
  r.X = quad_bcast<0>(prod); r.Y = quad_bcast<1>(prod); r.Z = quad_bcast<2>(prod); r.T = quad_bcast<3>(prod);
  return r;
}

constexpr int MSM_FINAL_BLOCK = 128;     // 23 windows x 4 lanes = 92 active lanes in two waves

template <class S>
__global__ void __launch_bounds__(MSM_FINAL_BLOCK) k_msm_final(const uint32_t* part, int groups, int groups_hi,
                                                               uint8_t* out_enc, uint8_t* out_xy,
                                                               const uint8_t* flags, uint8_t* status,
                                                               uint8_t* fail_flag, uint32_t sflags) {
  __shared__ uint32_t stage[32 * MSM_PT_WORDS];
  const int t = threadIdx.x, w = t >> 2, q = t & 3;
  // quad w < 23: R_w = sum over the groups (lane q takes groups q, q+4, ...; two butterfly steps join
  // them), then 2^(11 w) * R_w by 11 w quad-parallel doublings.  Quads run in lockstep: the critical path
  // is the top window's 242 doublings instead of a serial 253-doubling Horner.
  PtE acc = te_identity();
  if (w < MSM_W) {
    const int ng = w < MSM_W_SHORT ? groups : groups_hi;
    for (int g = q; g < ng; g += 4)
      acc = te_add<S>(acc, lds_load_pt(part + ((size_t)w * groups + g) * MSM_PT_WORDS));
  }
#pragma unroll 1
  for (int step = 1; step <= 2; step <<= 1) {
    PtE o;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      o.X.v[i] = __shfl_xor(acc.X.v[i], step, 64); o.Y.v[i] = __shfl_xor(acc.Y.v[i], step, 64);
      o.Z.v[i] = __shfl_xor(acc.Z.v[i], step, 64); o.T.v[i] = __shfl_xor(acc.T.v[i], step, 64);
    }
    acc = te_add<S>(acc, o);
  }
  // one representation for the whole quad (a + b and b + a agree only modulo q)
  acc.X = quad_bcast<0>(acc.X); acc.Y = quad_bcast<0>(acc.Y); acc.Z = quad_bcast<0>(acc.Z); acc.T = quad_bcast<0>(acc.T);
  const int nd = w < MSM_W ? MSM_C * w : 0;
#pragma unroll 1
  for (int j = 0; j < MSM_C * (MSM_W - 1); ++j)
    if (j < nd) acc = te_dbl_quad<S>(acc, q);
  // tree-sum of the 23 window results through LDS
  __syncthreads();
  if (q == 0 && w < 32) lds_store_pt(stage + w * MSM_PT_WORDS, acc);     // quads 23..31 hold the identity
  __syncthreads();
  if (t < 32) acc = lds_load_pt(stage + t * MSM_PT_WORDS);
#pragma unroll 1
  for (int s = 16; s >= 1; s >>= 1) {
    __syncthreads();
    if (t < 32) lds_store_pt(stage + t * MSM_PT_WORDS, acc);
    __syncthreads();
    if (t < s) acc = te_add<S>(acc, lds_load_pt(stage + (t + s) * MSM_PT_WORDS));
  }
  if (t == 0) {
    const bool bad = flags[0] != 0;
    if (!out_enc && !out_xy) {
      // verdict only (batched verification): the neutral element is X == 0, Y == Z; no inversion
      const bool neutral = fe_is_zero(acc.X) && fe_eq(acc.Y, acc.Z);
      if (fail_flag && (!neutral || bad)) fail_flag[0] = 1;
      if (status) status[0] = bad ? ST_INVALID_DATA : ST_OK;
      return;
    }
    FeN x, y;
    te_to_affine(x, y, acc);
    uint32_t e[8], xw[8], yw[8];
    te_encode_affine(e, x, y, sflags);
    fe_to_u256(xw, x); fe_to_u256(yw, y);
    // neutral element: x == 0 and y == 1
    uint32_t nz = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) nz |= xw[j] | (yw[j] ^ (j == 0 ? 1u : 0u));
    if (fail_flag && (nz != 0 || bad)) fail_flag[0] = 1;
    if (bad) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { e[j] = 0; xw[j] = 0; yw[j] = 0; }
    }
    if (out_enc) store32(out_enc, 0, e);
    if (out_xy) {
      uint32_t* p = reinterpret_cast<uint32_t*>(out_xy);
#pragma unroll
      for (int j = 0; j < 8; ++j) { p[j] = xw[j]; p[8 + j] = yw[j]; }
    }
    if (status) status[0] = bad ? ST_INVALID_DATA : ST_OK;
  }
}

// ------------------------------------------------------------------------------- host
static size_t pad256(size_t x) { return (x + 255) & ~size_t(255); }

// high-window groups for `groups` low-window groups: proportional to the share of full-size scalars
static int msm_groups_hi(size_t n, size_t n_long, int groups) {
  if (n_long >= n) return groups;
  int gh = (int)(((double)n_long / (double)n) * groups + 0.5);
  size_t min_g = (n_long + MSM_MAX_PER_GROUP - 1) / MSM_MAX_PER_GROUP;
  if ((size_t)gh < min_g) gh = (int)min_g;
  return gh < 1 ? 1 : gh;
}

int msm_groups(size_t n, size_t n_long, int cus) {
  // Workgroups = MSM_W_SHORT g + (MSM_W - MSM_W_SHORT) g_hi, as many per CU as their LDS allows (one with 1024 buckets):
  // aim at one full round of the chip for mid-sized inputs and two for large ones (the dispatcher evens out the tails), at
  // least 16 points per lane and group, never more than 2^21 points per group.
  const int slots = cus;                          // 1024 buckets x 144 B: one workgroup per CU
  const int rounds = n >= (size_t(1) << 22) ? 2 : 1;
  const double share = n ? (double)n_long / (double)n : 1.0;
  int g = (int)(rounds * slots / (MSM_W_SHORT + (MSM_W - MSM_W_SHORT) * share));
  if (g < 1) g = 1;
  while (g > 1 && MSM_W_SHORT * g + (MSM_W - MSM_W_SHORT) * msm_groups_hi(n, n_long, g) > rounds * slots) --g;
  size_t max_g = (n + 8191) / 8192;
  if ((size_t)g > max_g) g = (int)max_g;
  if (g < 1) g = 1;
  size_t min_g = (n + MSM_MAX_PER_GROUP - 1) / MSM_MAX_PER_GROUP;
  if ((size_t)g < min_g) g = (int)min_g;
  return g;
}

size_t msm_workspace_bytes(size_t n, int groups) {
  // sized for the worst case groups_hi == groups
  size_t per_group = (n + groups - 1) / groups, list_cap = per_group + MSM_BLOCK, wgs = (size_t)MSM_W * groups;
  return pad256(n * MSM_PTA_STRIDE * 4) + pad256((size_t)MSM_W * n * 2) + pad256(wgs * list_cap * 4) +
         pad256(wgs * MSM_BLOCK * MSM_PT_WORDS * 4) + pad256(wgs * MSM_PT_WORDS * 4) + 256;
}

MsmLayout msm_layout(size_t n, size_t n_long, int groups, void* ws) {
  MsmLayout L;
  L.sflags = 0;
  L.n = n;
  L.n_long = n_long < n ? n_long : n;
  L.groups = groups;
  L.groups_hi = msm_groups_hi(n, L.n_long, groups);
  if (L.groups_hi > groups) L.groups_hi = groups;
  L.per_group = (n + groups - 1) / groups;
  L.per_group_hi = (L.n_long + L.groups_hi - 1) / L.groups_hi;
  // groups_hi rounds to nearest, so a high-window group can be slightly larger than a low-window one;
  // the list capacity was sized for per_group: keep within it
  while (L.per_group_hi > L.per_group && L.groups_hi < groups) {
    ++L.groups_hi;
    L.per_group_hi = (L.n_long + L.groups_hi - 1) / L.groups_hi;
  }
  L.list_cap = L.per_group + MSM_BLOCK;
  const size_t wgs = (size_t)MSM_W * groups;
  uint8_t* p = static_cast<uint8_t*>(ws);
  L.pts = reinterpret_cast<uint32_t*>(p); p += pad256(n * MSM_PTA_STRIDE * 4);
  L.digits = reinterpret_cast<int16_t*>(p); p += pad256((size_t)MSM_W * n * 2);
  L.lists = reinterpret_cast<uint32_t*>(p); p += pad256(wgs * L.list_cap * 4);
  L.heads = reinterpret_cast<uint32_t*>(p); p += pad256(wgs * MSM_BLOCK * MSM_PT_WORDS * 4);
  L.part = reinterpret_cast<uint32_t*>(p); p += pad256(wgs * MSM_PT_WORDS * 4);
  L.flags = p;
  return L;
}

template <class S>
static void launch_msm_core_t(const MsmLayout& L, uint8_t* out_enc, uint8_t* out_xy, uint8_t* status,
                              uint8_t* fail_flag, hipStream_t st, hipEvent_t* ev) {
  const size_t lds_bytes = ((size_t)MSM_BUCKETS * MSM_PT_WORDS + 2 * MSM_BUCKETS + 16) * 4;   // 155,712 B of 160 KiB
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_msm_buckets<S>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    attr_set = true;
  }
  const int wgs = MSM_W_SHORT * L.groups + (MSM_W - MSM_W_SHORT) * L.groups_hi;
  hipLaunchKernelGGL(k_msm_buckets<S>, dim3(wgs), dim3(MSM_BLOCK), lds_bytes, st, L);
  if (ev) (void)hipEventRecord(ev[0], st);
  hipLaunchKernelGGL(k_msm_final<S>, dim3(1), dim3(MSM_FINAL_BLOCK), 0, st, L.part, L.groups, L.groups_hi, out_enc, out_xy, L.flags,
                     status, fail_flag, L.sflags);
  if (ev) { (void)hipEventRecord(ev[1], st); (void)hipEventRecord(ev[2], st); }
}

void launch_msm_core(int suite, const MsmLayout& L, uint8_t* out_enc, uint8_t* out_xy, uint8_t* status,
                     uint8_t* fail_flag, hipStream_t st, hipEvent_t* ev) {
  VRF_DISPATCH_SUITE(suite, launch_msm_core_t<S>(L, out_enc, out_xy, status, fail_flag, st, ev));
}

// mont256 != 0: the bases' coordinates are Montgomery images x 2^256 mod q (VRFHIP_FLAG_COORDS_MONT256); out_xy stays
// canonical here (api.hip converts it with launch_xy_to_mont256)
void launch_msm_coords(int suite, size_t n, const uint8_t* xy, const uint8_t* scalars, uint8_t* out_enc,
                       uint8_t* out_xy, uint8_t* status, void* ws, int groups, int mont256, uint32_t sflags,
                       hipStream_t st) {
  MsmLayout L = msm_layout(n, n, groups, ws);
  L.sflags = sflags;
  (void)hipMemsetAsync(L.flags, 0, 256, st);
  VRF_DISPATCH_SUITE(suite, hipLaunchKernelGGL(k_msm_prep<S>, grid_for(n), dim3(BLOCK), 0, st, n, xy, scalars,
                                               L.pts, L.digits, L.flags, mont256));
  launch_msm_core(suite, L, out_enc, out_xy, status, nullptr, st);
}
void launch_msm(int suite, size_t n, const uint8_t* xy, const uint8_t* scalars, uint8_t* out_enc,
                uint8_t* out_xy, uint8_t* status, void* ws, int groups, hipStream_t st) {
  launch_msm_coords(suite, n, xy, scalars, out_enc, out_xy, status, ws, groups, 0, 0, st);
}

VRF_NS_END
