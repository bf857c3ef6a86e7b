// k_msm_g1.hip -- multi-scalar multiplication on BLS12-381 G1 and the G1 side of the batched pairing check.
//
// [ref /root/reference src/lib.rs:14 `ring`]  A ring-VRF / KZG verifier ends in e(A, Q0) e(B, Q1) == 1 with the two
// G2 points fixed by the SRS.  For n such checks and secret weights z_i,
//     prod_i (e(A_i, Q0) e(B_i, Q1))^{z_i} = e(sum z_i A_i, Q0) e(sum z_i B_i, Q1),
// so the whole batch costs two G1 multi-scalar multiplications and ONE pairing check instead of n (a batch holding a
// false item passes with probability <= 2^-128; the host entry point falls back to the per-item kernel to name it).
// This file is the G1 part -- ark_ec's `VariableBaseMSM::msm` for ark-bls12-381 G1 (north_star: Pippenger with the
// bucket accumulation staged in LDS):
//   k_g1_prep_rlc / k_g1_prep_msm : wire points -> Montgomery affine, on-curve test; weights or scalars -> signed
//                                   10-bit digits
//   k_g1_buckets : one workgroup per (set, window, point group) owns the window's 512 buckets in LDS (84 KiB).
//                  Counting sort of the group's points by bucket (LDS atomics, wave-shuffle scan), the sorted list
//                  is cut into 512 equal chunks, every lane folds its chunk with complete mixed additions and
//                  flushes a run when the bucket changes (first run parked as a "head" and merged afterwards), then
//                  sum_j j B_j by a suffix scan and a tree over the 512 lanes, staged through LDS.
//   k_g1_final   : per set, one QUAD per window: sum of the groups, 10 w doublings (two rounds of one product per lane
//                  each), tree-sum, affine, wire format.
// Same schedule as k_msm.hip (twisted Edwards); the group law is g1.cuh's complete projective one.
#include "msm_g1.h"

#include "g1.cuh"
#include "bls12_quad.cuh"     // qperm: DPP moves inside a quad (k_g1_final)
#include "sha512.cuh"

namespace vrf {
using namespace bls;

static_assert(G1_PT_WORDS == G1P_WORDS && G1_AFF_WORDS == G1A_WORDS, "layout constants");

// signed radix-2^10 digits of a little-endian integer k < 2^(10 W - 1); zero = every digit 0
VRF_HD void g1_write_digits(int16_t* digits, size_t n, size_t i, int W, const uint32_t k[8], bool zero) {
  uint32_t carry = 0;
#pragma unroll 1
  for (int w = 0; w < W; ++w) {
    const int bit = w * G1_C, wi = bit >> 5, sh = bit & 31;
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j == wi) lo = k[j];
      if (j == wi + 1) hi = k[j];
    }
    uint32_t v = (sh ? ((lo >> sh) | (hi << (32 - sh))) : lo) & ((1u << G1_C) - 1);
    v += carry;
    int d = (int)v;
    carry = 0;
    if (v > (uint32_t)G1_BUCKETS) { d = (int)v - (1 << G1_C); carry = 1; }
    digits[(size_t)w * n + i] = (int16_t)(zero ? 0 : d);
  }
}

VRF_HD void g1_store_affine(uint32_t* dst, const G1Aff& P) {
#pragma unroll
  for (int j = 0; j < NLB; ++j) { dst[j] = (uint32_t)P.x.v[j]; dst[NLB + j] = (uint32_t)P.y.v[j]; }
}

// ------------------------------------------------------------------------------- prep
struct Seed32 { uint8_t b[32]; };
__global__ void __launch_bounds__(128) k_g1_prep_rlc(G1MsmLayout L, const uint8_t* g1, Seed32 seed,
                                                     const uint8_t* root, uint64_t index0, uint8_t* status) {
  const size_t i = (size_t)blockIdx.x * 128 + threadIdx.x;
  if (i >= L.n) return;
  const uint32_t* w = reinterpret_cast<const uint32_t*>(g1 + i * 192);
  G1Aff P[2];
  bool inf[2];
  bool ok = true;
#pragma unroll 1
  for (int s = 0; s < 2; ++s) {
    uint32_t words[24];
#pragma unroll
    for (int j = 0; j < 24; ++j) words[j] = w[24 * s + j];
    G1Aff Q; bool qi;
    ok = g1_load(Q, qi, words) && ok;
    if (s == 0) { P[0] = Q; inf[0] = qi; } else { P[1] = Q; inf[1] = qi; }
  }
  // z_i: 128 bits of SHA-512("vrfhip-pairing-rlc-v2" || seed || batch digest || u64_le(index))
  Sha512 h;
  sha512_init(h);
  constexpr char tag[] = "vrfhip-pairing-rlc-v2";
#pragma unroll
  for (int j = 0; j < 21; ++j) sha512_put_byte(h, (uint8_t)tag[j]);
  sha512_put_bytes(h, seed.b, 32);
  sha512_put_bytes(h, root, 32);
  const uint64_t idx = index0 + i;
#pragma unroll
  for (int j = 0; j < 8; ++j) sha512_put_byte(h, (uint8_t)(idx >> (8 * j)));
  sha512_final(h);
  uint32_t z[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) { z[j] = sha512_word_mem(h, j); z[4 + j] = 0; }
#pragma unroll 1
  for (int s = 0; s < 2; ++s) {
    g1_store_affine(L.pts + ((size_t)s * L.n + i) * G1_AFF_STRIDE, s == 0 ? P[0] : P[1]);
    g1_write_digits(L.digits + (size_t)s * L.windows * L.n, L.n, i, L.windows, z, !ok || (s == 0 ? inf[0] : inf[1]));
  }
  status[i] = ok ? 0 : 2;
}

__global__ void __launch_bounds__(128) k_g1_prep_msm(G1MsmLayout L, const uint8_t* bases, const uint8_t* scalars) {
  const size_t i = (size_t)blockIdx.x * 128 + threadIdx.x;
  if (i >= L.n) return;
  const uint32_t* w = reinterpret_cast<const uint32_t*>(bases + i * 96);
  uint32_t words[24], k[8];
#pragma unroll
  for (int j = 0; j < 24; ++j) words[j] = w[j];
  const uint32_t* kw = reinterpret_cast<const uint32_t*>(scalars + i * 32);
#pragma unroll
  for (int j = 0; j < 8; ++j) k[j] = kw[j];
  G1Aff P; bool inf;
  bool ok = g1_load(P, inf, words);
  // scalar < r (the BLS12-381 scalar field modulus = the base field of the VRF curves)
  bool lt = false, decided = false;
#pragma unroll
  for (int j = 7; j >= 0; --j)
    if (!decided && k[j] != vrfk::Q32[j]) { lt = k[j] < vrfk::Q32[j]; decided = true; }
  ok = ok && lt;
  if (!ok) L.flags[0] = 2;                    // InvalidData (copied into the caller's status byte)
  g1_store_affine(L.pts + i * G1_AFF_STRIDE, P);
  g1_write_digits(L.digits, L.n, i, L.windows, k, !ok || inf);
}

// ------------------------------------------------------------------------------- buckets
constexpr uint32_t G1_NONE = 0xffffffffu;
constexpr uint32_t G1_IDX_MASK = (1u << G1_IDX_BITS) - 1;

__global__ void __launch_bounds__(G1_BLOCK) k_g1_buckets(G1MsmLayout L) {
  extern __shared__ uint32_t lds[];
  uint32_t* bucket = lds;                                   // [512][42]
  uint32_t* counts = lds + G1_BUCKETS * G1_PT_WORDS;        // [512]
  uint32_t* cursor = counts + G1_BUCKETS;                   // [512] scatter cursors, later head bucket ids
  uint32_t* wsum = cursor + G1_BUCKETS;                     // [8]
  const int t = threadIdx.x;
  const int wg = blockIdx.x;
  const int per_set = L.windows * L.groups;
  const int set = wg / per_set, w = (wg % per_set) / L.groups, g = wg % L.groups;
  const size_t lo = (size_t)g * L.per_group;
  const size_t hi = lo + L.per_group < L.n ? lo + L.per_group : L.n;
  const uint32_t cnt_all = lo < hi ? (uint32_t)(hi - lo) : 0u;
  const int16_t* dig = L.digits + ((size_t)set * L.windows + w) * L.n + lo;
  const uint32_t* P = L.pts + ((size_t)set * L.n + lo) * G1_AFF_STRIDE;
  uint32_t* list = L.lists + (size_t)wg * L.list_cap;
  uint32_t* heads = L.heads + (size_t)wg * G1_BLOCK * G1_PT_WORDS;
  g1p_store(bucket + t * G1_PT_WORDS, g1_identity());
  // 1. histogram
  counts[t] = 0;
  __syncthreads();
  for (uint32_t j = t; j < cnt_all; j += G1_BLOCK) {
    const int d = dig[j];
    if (d != 0) atomicAdd(&counts[(d < 0 ? -d : d) - 1], 1u);
  }
  __syncthreads();
  // 2. exclusive scan of the 512 counts: waves scan by shuffles, wave totals through LDS
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
    for (int k = 0; k < G1_BLOCK / 64; ++k) {
      const uint32_t x = wsum[k];
      if (k < (t >> 6)) wbase += x;
      total += x;
    }
    m = total;
    cursor[t] = wbase + incl - v;
  }
  __syncthreads();
  // 3. scatter into bucket order, transposed: entry i of lane l sits at list[i * 512 + l]
  const uint32_t chunk = (m + G1_BLOCK - 1) / G1_BLOCK;
  for (uint32_t j = t; j < cnt_all; j += G1_BLOCK) {
    const int d = dig[j];
    if (d != 0) {
      const uint32_t b = (uint32_t)(d < 0 ? -d : d) - 1;
      const uint32_t pos = atomicAdd(&cursor[b], 1u);
      const uint32_t lane = pos / chunk, i = pos - lane * chunk;
      list[(size_t)i * G1_BLOCK + lane] = j | (b << G1_IDX_BITS) | (d < 0 ? 0x80000000u : 0u);
    }
  }
  __threadfence_block();
  __syncthreads();
  // 4. every lane folds its chunk (same trip count for all lanes)
  uint32_t head_b = G1_NONE;
  {
    const uint32_t my0 = (uint32_t)t * chunk;
    const uint32_t cnt = my0 >= m ? 0u : (m - my0 < chunk ? m - my0 : chunk);
    G1P acc = g1_identity();
    uint32_t cur = G1_NONE;
    bool first_run = true;
    uint32_t* myhead = heads + (size_t)t * G1_PT_WORDS;
    // the list entry names the point: entries are fetched two trips ahead, points one trip ahead (k_msm.hip)
    auto load_xy = [&](FpS& x, FpS& y, uint32_t e) {
      const uint32_t* src = P + (size_t)(e & G1_IDX_MASK) * G1_AFF_STRIDE;
#pragma unroll
      for (int j = 0; j < NLB; ++j) { x.v[j] = (int32_t)src[j]; y.v[j] = (int32_t)src[NLB + j]; }
    };
    uint32_t ent_n = 0, ent_nn = 0;
    FpS x_n = FpS(fp_zero()), y_n = FpS(fp_zero());
    if (cnt > 0) ent_n = list[t];
    if (cnt > 1) ent_nn = list[(size_t)G1_BLOCK + t];
    if (cnt > 0) load_xy(x_n, y_n, ent_n);
#pragma unroll 1
    for (uint32_t i = 0; i < chunk; ++i) {
      if (i < cnt) {
        const uint32_t ent = ent_n;
        const FpS x = x_n, y = y_n;
        ent_n = ent_nn;
        if (i + 2 < cnt) ent_nn = list[(size_t)(i + 2) * G1_BLOCK + t];
        if (i + 1 < cnt) load_xy(x_n, y_n, ent_n);
        const uint32_t b = (ent >> G1_IDX_BITS) & (G1_BUCKETS - 1);
        if (b != cur) {
          if (cur != G1_NONE) {
            if (first_run) { g1p_store(myhead, acc); head_b = cur; first_run = false; }
            else g1p_store(bucket + cur * G1_PT_WORDS, acc);
          }
          cur = b;
          acc = g1_identity();
        }
        acc = g1_madd(acc, x, y, (ent >> 31) != 0);
      }
    }
    if (cnt > 0) {
      if (first_run) { g1p_store(myhead, acc); head_b = cur; }
      else g1p_store(bucket + cur * G1_PT_WORDS, acc);
    }
  }
  cursor[t] = head_b;
  __threadfence_block();
  __syncthreads();
  // 5. merge the heads: the first lane of each chain of equal head buckets adds the chain to the bucket
  if (head_b != G1_NONE && (t == 0 || cursor[t - 1] != head_b)) {
    G1P h = g1p_load(heads + (size_t)t * G1_PT_WORDS);
    for (int k = t + 1; k < G1_BLOCK && cursor[k] == head_b; ++k) h = g1_add(h, g1p_load(heads + (size_t)k * G1_PT_WORDS));
    uint32_t* slot = bucket + head_b * G1_PT_WORDS;
    g1p_store(slot, g1_add(g1p_load(slot), h));
  }
  __syncthreads();
  // 6. R = sum_j j B_j: lane t holds B_{t+1}; suffix scan S_t = sum_{u >= t} B_{u+1}, then R = sum_t S_t
  G1P cur = g1p_load(bucket + t * G1_PT_WORDS);
  uint32_t* stage = lds;
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll 1
    for (int k = 0; k < 9; ++k) {
      const int s = pass == 0 ? (1 << k) : (G1_BLOCK >> (k + 1));
      __syncthreads();
      g1p_store(stage + t * G1_PT_WORDS, cur);
      __syncthreads();
      const bool active = pass == 0 ? (t + s < G1_BLOCK) : (t < s);
      if (active) cur = g1_add(cur, g1p_load(stage + (t + s) * G1_PT_WORDS));
    }
  }
  if (t == 0) g1p_store(L.part + (((size_t)set * L.windows + w) * L.groups + g) * G1_PT_WORDS, cur);
}

// ------------------------------------------------------------------------------- final
// One doubling on the four lanes of a DPP quad (the point replicated in all four): two rounds of ONE field product per
// lane -- Y^2 | YZ | Z^2 | XY, then (b3 Z^2)(8 Y^2) | (YZ)(8 Y^2) | t (Y^2 + b3 Z^2) | t (XY) with t = Y^2 - 3 b3 Z^2 -- instead
// of nine products in a row: the kernel is one latency chain of 10 (W - 1) doublings.  Same formulas as g1_dbl.
__device__ __forceinline__ G1P g1_dbl_quad(const G1P& p, int q) {
  const FpS a1 = fp_select(q == 3, p.X, fp_select(q == 2, p.Z, p.Y));
  const FpS b1 = fp_select(q == 1 || q == 2, p.Z, p.Y);
  const FpS m1 = fp_fit(fp_mul(a1, b1));
  const FpS t0 = qperm<QP_BC0>(m1), t1 = qperm<QP_BC1>(m1), zz = qperm<QP_BC2>(m1), xy = qperm<0xff>(m1);
  const FpS z8 = fp_fit(fp_dbl(fp_dbl(fp_dbl(t0))));                            // 8 Y^2
  const auto t2 = fp_reduce(fp_mul12(zz));                                      // b3 Z^2
  const FpS t2s = fp_fit(t2);
  const FpS y3 = fp_fit(fp_add(t0, t2));
  const FpS t0m = fp_fit(fp_sub(t0, fp_norm(fp_add(fp_dbl(t2), t2))));          // Y^2 - 3 b3 Z^2
  const FpS a2 = fp_select(q == 0, t2s, fp_select(q == 1, t1, t0m));
  const FpS b2 = fp_select(q < 2, z8, fp_select(q == 2, y3, xy));
  const FpS m2 = fp_fit(fp_mul(a2, b2));
  const FpS x3 = qperm<QP_BC0>(m2), z3 = qperm<QP_BC1>(m2), ty = qperm<QP_BC2>(m2), tx = qperm<0xff>(m2);
  G1P r;
  r.X = fp_fit(fp_dbl(tx));
  r.Y = fp_fit(fp_add(x3, ty));
  r.Z = z3;
  return r;
}

constexpr int G1_FINAL_BLOCK = 128;      // four lanes per window, up to 32 windows
__global__ void __launch_bounds__(G1_FINAL_BLOCK) k_g1_final(G1MsmLayout L) {
  __shared__ uint32_t stage[32 * G1_PT_WORDS];
  const int t = threadIdx.x, set = blockIdx.x;
  const int w = t >> 2, q = t & 3;                  // window, lane inside the window's quad
  G1P acc = g1_identity();
  if (w < L.windows) {
    // lane q sums the groups q, q + 4, ...; two butterfly steps (lane ^ 1, lane ^ 2) leave the total in all four lanes
    for (int g = q; g < L.groups; g += 4)
      acc = g1_add(acc, g1p_load(L.part + (((size_t)set * L.windows + w) * L.groups + g) * G1_PT_WORDS));
  }
  {
    G1P o;
    o.X = qperm<0xb1>(acc.X); o.Y = qperm<0xb1>(acc.Y); o.Z = qperm<0xb1>(acc.Z);      // quad_perm 1,0,3,2
    acc = g1_add(acc, o);
    o.X = qperm<0x4e>(acc.X); o.Y = qperm<0x4e>(acc.Y); o.Z = qperm<0x4e>(acc.Z);      // quad_perm 2,3,0,1
    acc = g1_add(acc, o);
  }
  // 2^(10 w) * R_w: the critical path is the top window's 10 (W - 1) doublings, each spread over its quad
  const int nd = w < L.windows ? G1_C * w : 0;
#pragma unroll 1
  for (int j = 0; j < G1_C * (L.windows - 1); ++j)
    if (j < nd) acc = g1_dbl_quad(acc, q);
  // window w's result (lane 0 of its quad) -> thread w, which runs the tree below
  __syncthreads();
  if (q == 0 && w < 32) g1p_store(stage + w * G1_PT_WORDS, acc);
  __syncthreads();
  acc = t < 32 ? g1p_load(stage + t * G1_PT_WORDS) : g1_identity();
  // tree-sum of the window results (<= 32 windows) through LDS
#pragma unroll 1
  for (int s = 16; s >= 1; s >>= 1) {
    __syncthreads();
    if (t < 32) g1p_store(stage + t * G1_PT_WORDS, acc);
    __syncthreads();
    if (t < s) acc = g1_add(acc, g1p_load(stage + (t + s) * G1_PT_WORDS));
  }
  if (t == 0) {
    uint32_t* out = reinterpret_cast<uint32_t*>(L.sums + set * 96);
    if (g1_is_identity(acc)) {
#pragma unroll
      for (int j = 0; j < 24; ++j) out[j] = 0;
    } else {
      FpS x, y;
      g1_to_affine(x, y, acc);
      uint32_t xw[12], yw[12];
      fp_to_words(xw, x); fp_to_words(yw, y);
#pragma unroll
      for (int j = 0; j < 12; ++j) { out[j] = xw[j]; out[12 + j] = yw[j]; }
    }
  }
}

// ------------------------------------------------------------------------------- host
static size_t pad256(size_t x) { return (x + 255) & ~size_t(255); }

int g1_msm_groups(size_t n, int sets, int windows, int cus) {
  int g = cus / (sets * windows);                 // about one round of the chip
  if (g < 1) g = 1;
  const size_t max_g = (n + 2047) / 2048;        // at least 4 points per lane and group
  if ((size_t)g > max_g) g = (int)max_g;
  if (g < 1) g = 1;
  const size_t min_g = (n + G1_MAX_PER_GROUP - 1) / G1_MAX_PER_GROUP;
  if ((size_t)g < min_g) g = (int)min_g;
  return g;
}
size_t g1_msm_workspace_bytes(size_t n, int sets, int windows, int groups) {
  const size_t per_group = (n + groups - 1) / groups, list_cap = per_group + G1_BLOCK, wgs = (size_t)sets * windows * groups;
  return pad256((size_t)sets * n * G1_AFF_STRIDE * 4) + pad256((size_t)sets * windows * n * 2) + pad256(wgs * list_cap * 4) +
         pad256(wgs * G1_BLOCK * G1_PT_WORDS * 4) + pad256(wgs * G1_PT_WORDS * 4) + 256 + 256;
}
G1MsmLayout g1_msm_layout(size_t n, int sets, int windows, int groups, void* ws) {
  G1MsmLayout L;
  L.n = n; L.sets = sets; L.windows = windows; L.groups = groups;
  L.per_group = (n + groups - 1) / groups;
  L.list_cap = L.per_group + G1_BLOCK;
  const size_t wgs = (size_t)sets * windows * groups;
  uint8_t* p = static_cast<uint8_t*>(ws);
  L.pts = reinterpret_cast<uint32_t*>(p); p += pad256((size_t)sets * n * G1_AFF_STRIDE * 4);
  L.digits = reinterpret_cast<int16_t*>(p); p += pad256((size_t)sets * windows * n * 2);
  L.lists = reinterpret_cast<uint32_t*>(p); p += pad256(wgs * L.list_cap * 4);
  L.heads = reinterpret_cast<uint32_t*>(p); p += pad256(wgs * G1_BLOCK * G1_PT_WORDS * 4);
  L.part = reinterpret_cast<uint32_t*>(p); p += pad256(wgs * G1_PT_WORDS * 4);
  L.sums = p; p += 256;
  L.flags = p;
  return L;
}

static void launch_core(const G1MsmLayout& L, hipStream_t st, hipEvent_t* ev) {
  const size_t lds_bytes = ((size_t)G1_BUCKETS * G1_PT_WORDS + 2 * G1_BUCKETS + 16) * 4;      // 90,176 B
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_g1_buckets), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds_bytes);
    attr_set = true;
  }
  hipLaunchKernelGGL(k_g1_buckets, dim3((unsigned)(L.sets * L.windows * L.groups)), dim3(G1_BLOCK), lds_bytes, st, L);
  if (ev) (void)hipEventRecord(ev[1], st);
  hipLaunchKernelGGL(k_g1_final, dim3((unsigned)L.sets), dim3(G1_FINAL_BLOCK), 0, st, L);
  if (ev) (void)hipEventRecord(ev[2], st);
}

void launch_g1_rlc(const G1MsmLayout& L, const uint8_t* g1, const uint8_t seed[32], const uint8_t* d_root, uint64_t index0,
                   uint8_t* status,
                   hipStream_t st, hipEvent_t* ev) {
  if (L.n == 0) return;
  (void)hipMemsetAsync(L.flags, 0, 256, st);
  Seed32 sd;
  for (int i = 0; i < 32; ++i) sd.b[i] = seed[i];
  hipLaunchKernelGGL(k_g1_prep_rlc, dim3((unsigned)((L.n + 127) / 128)), dim3(128), 0, st, L, g1, sd, d_root, index0, status);
  if (ev) (void)hipEventRecord(ev[0], st);
  launch_core(L, st, ev);
}

void launch_g1_msm(const G1MsmLayout& L, const uint8_t* bases, const uint8_t* scalars, uint8_t* status1, hipStream_t st) {
  (void)hipMemsetAsync(L.flags, 0, 256, st);
  if (L.n == 0) {
    (void)hipMemsetAsync(L.sums, 0, 96, st);
    (void)hipMemsetAsync(status1, 0, 1, st);
    return;
  }
  hipLaunchKernelGGL(k_g1_prep_msm, dim3((unsigned)((L.n + 127) / 128)), dim3(128), 0, st, L, bases, scalars);
  launch_core(L, st, nullptr);
  // status byte: 2 if any input was invalid (the sum is then meaningless and zeroed by the caller)
  (void)hipMemcpyAsync(status1, L.flags, 1, hipMemcpyDeviceToDevice, st);
}

}  // namespace vrf
