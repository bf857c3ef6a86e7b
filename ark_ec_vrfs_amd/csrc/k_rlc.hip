// k_rlc.hip -- batched Pedersen-VRF verification by random linear combination (SURVEY.md section 8
// row f2): n proofs are checked with ONE multi-scalar multiplication instead of 4n scalar
// multiplications.  Sits behind `pedersen::Verifier::verify` (/root/reference src/lib.rs:14) for callers
// that verify whole batches.
//
// A Pedersen proof (pk_com, R, Ok, s, sb) for (H, Gamma) is valid iff, with c = challenge(pk_com, H,
// Gamma, R, Ok, ad),
//     D1 = s*H - c*Gamma - Ok = O      and      D2 = s*G + sb*B - c*pk_com - R = O.
// c needs no curve arithmetic (every hashed point is part of the input), so for secret 128-bit weights
// z_i, z'_i the single check  sum_i (z_i*D1_i + z'_i*D2_i) = O  accepts a batch containing an invalid
// proof with probability <= 2^-125 (128-bit weights with three fixed bits; all points lie in the prime-order subgroup: checked by
// every verify entry point).  That sum is an MSM over 5n variable points plus the two fixed bases:
//     sum_i [ (z s)_i H_i - (z c)_i Gamma_i - (z' c)_i pk_com_i - z'_i R_i - z_i Ok_i ]
//       + (sum_i z'_i s_i) G + (sum_i z'_i sb_i) B.
//   k_rlc_decode : K proofs per lane share one inversion (5K decompression denominators); writes the
//                  Montgomery affine-cached points and the signed window digits straight into the MSM
//                  layout (rlc_index); accumulates the two fixed-base
//                  scalars as 64-bit limb columns (wave shuffles + one atomic per column and wave).
//   k_rlc_fixed  : normalises the columns mod r and appends G and B with their digits.
//   then k_msm_buckets / k_msm_final (k_msm.hip) and the verdict byte.
// Weights: (z_i, z'_i) = the first two 16-byte little-endian halves of
// SHA-512("vrfhip-rlc-v2" || seed[32] || digest of the launch group (digest.cuh) || u64_le(index of the proof in the
// caller's batch)) with the low
// three bits forced to 001; the seed must be unpredictable to whoever produced the proofs.
#include "kernels.h"
#include "msm.cuh"

VRF_NS_BEGIN

constexpr int RLC_SLOT = 4 * NL + 1;     // y | num | den | prefix | (flag, ok)

// challenge, weights, scalars and window digits of one proof; cp = compressed (pk_com, H, Gamma, R, Ok)
template <class S>
VRF_HD void rlc_emit_item(const RlcArgs& a, size_t item, bool pts_valid, const uint32_t (&cp)[5][8],
                          uint64_t (&cols)[16]) {
  const size_t n = a.n, N = a.L.n;
  uint32_t s[8], sb[8];
  load32(s, a.s, item); load32(sb, a.sb, item);
  const bool valid = pts_valid && fr_is_canonical<S>(s) && fr_is_canonical<S>(sb);
  if (!valid) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[j] = 0; sb[j] = 0; }
  }
  const uint8_t* adp; uint32_t adl;
  bytes_get(a.ad, item, adp, adl);
  uint32_t c[8], z[8], zp[8], t[8];
  challenge5<S>(c, cp, adp, adl, a.T.sq.str);
  rlc_weights<S>(z, zp, a.seed, a.root, a.index0 + item);
  fr_mul<S>(t, z, s);
  msm_write_digits<S>(a.L.digits, N, rlc_index(0, n, item), t, false, !valid);     // + (z s) H
  fr_mul<S>(t, z, c);
  msm_write_digits<S>(a.L.digits, N, rlc_index(1, n, item), t, true, !valid);      // - (z c) Gamma
  fr_mul<S>(t, zp, c);
  msm_write_digits<S>(a.L.digits, N, rlc_index(2, n, item), t, true, !valid);      // - (z' c) pk_com
  msm_write_digits<S>(a.L.digits, N, rlc_index(3, n, item), zp, true, !valid);     // - z' R   (128 bits)
  msm_write_digits<S>(a.L.digits, N, rlc_index(4, n, item), z, true, !valid);      // - z Ok   (128 bits)
  fr_mul<S>(t, zp, s);
#pragma unroll
  for (int j = 0; j < 8; ++j) cols[j] += t[j];
  fr_mul<S>(t, zp, sb);
#pragma unroll
  for (int j = 0; j < 8; ++j) cols[8 + j] += t[j];
  a.status[item] = (uint8_t)(valid ? ST_OK : ST_INVALID_DATA);
}

// wave reduction of the fixed-base limb columns, one atomic per column and wave
VRF_HD void rlc_flush_cols(const RlcArgs& a, const uint64_t (&cols)[16]) {
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    uint64_t v = cols[j];
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) v += __shfl_down(v, sft, 64);
    if ((threadIdx.x & 63) == 0 && v != 0)
      atomicAdd(reinterpret_cast<unsigned long long*>(a.fixed_cols) + j, (unsigned long long)v);
  }
}

template <class S, int MINW>
__global__ void __launch_bounds__(BLOCK, MINW) k_rlc_decode(RlcArgs a) {
  const size_t first = ((size_t)blockIdx.x * BLOCK + threadIdx.x) * a.k_lane;
  const int K = a.k_lane;
  const size_t n = a.n;
  uint64_t cols[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) cols[j] = 0;
  if (first < n) {
    // ---- phase a: y, y^2 - 1, d*y^2 - a of 5K points; running product of the denominators ----
    FeN run = fe_one();
#pragma unroll 1
    for (int j = 0; j < 5 * K; ++j) {
      const size_t item = first + j / 5;
      const int p = j % 5;
      if (item < n) {
        const uint8_t* src = p == 0 ? a.h : p == 1 ? a.gamma : p == 2 ? a.pk_com : p == 3 ? a.r : a.ok;
        uint32_t enc[8];
        load32(enc, src, item);
        DecodeA d = decode_phase_a<S>(enc);
        uint32_t* slot = a.scratch + item * (size_t)a.scratch_stride + p * RLC_SLOT;
        fe_store(slot, d.y);
        fe_store(slot + NL, d.num);
        fe_store(slot + 2 * NL, d.den);
        fe_store(slot + 3 * NL, run);
        slot[4 * NL] = (d.flag ? 1u : 0u) | (d.ok ? 2u : 0u);
        run = fe_mul(run, d.den);
      }
    }
    FeN inv = fe_inv(run);
    uint32_t valid_mask = 0xffffffffu;
    // ---- phase b: x by square root; Montgomery affine-cached point into the MSM layout ----
#pragma unroll 1
    for (int j = 5 * K - 1; j >= 0; --j) {
      const size_t item = first + j / 5;
      const int p = j % 5;
      if (item < n) {
        const uint32_t* slot = a.scratch + item * (size_t)a.scratch_stride + p * RLC_SLOT;
        DecodeA d;
        d.y = fe_load<1, 2>(slot);
        d.num = fe_load<1, 6>(slot + NL);
        d.den = fe_load<1, 2>(slot + 2 * NL);
        FeN prefix = fe_load<1, 2>(slot + 3 * NL);
        d.flag = slot[4 * NL] & 1u;
        d.ok = (slot[4 * NL] >> 1) & 1u;
        FeN di = fe_mul(inv, prefix);
        inv = fe_mul(inv, d.den);
        Fe<1, 4> x;
        bool ok = decode_phase_b<S>(x, d, di, a.T.sq);
        PtA pa;
        pa.x = fe_mul(x, fe_one());
        if (a.check_mask & (p == 0 ? CHK_INPUT : p == 1 ? CHK_OUTPUT : CHK_PROOF))
          ok = in_prime_subgroup<S>(pa.x, d.y, a.T.sq) && ok;
        if (!ok) valid_mask &= ~(1u << (j / 5));
        pa.y = d.y;
        pa.dt = fe_mul(fe_mul(pa.x, pa.y), S::d());
        pta_store(a.L.pts + rlc_index(p, n, item) * MSM_PTA_STRIDE, pa);
      }
    }
    // ---- challenge, weights, scalars, digits ----
#pragma unroll 1
    for (int k = 0; k < K; ++k) {
      const size_t item = first + k;
      if (item < n) {
        uint32_t cp[5][8];
        load32(cp[0], a.pk_com, item); load32(cp[1], a.h, item); load32(cp[2], a.gamma, item);
        load32(cp[3], a.r, item); load32(cp[4], a.ok, item);
        rlc_emit_item<S>(a, item, ((valid_mask >> k) & 1u) != 0, cp, cols);
      }
    }
  }
  rlc_flush_cols(a, cols);
}

// The same stage for callers that hold the points in memory as arkworks `Affine { x, y }` (64 bytes per
// point: x || y, 32-byte little-endian canonical): no square roots.  One lane per proof.  InvalidData =
// coordinate >= q or point off the curve.
template <class S, int MINW>
__global__ void __launch_bounds__(BLOCK, MINW) k_rlc_prep_affine(RlcArgs a) {
  const size_t item = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  const size_t n = a.n;
  uint64_t cols[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) cols[j] = 0;
  if (item < n) {
    uint32_t cp[5][8];
    bool valid = true;
#pragma unroll 1
    for (int p = 0; p < 5; ++p) {
      const uint8_t* src = p == 0 ? a.h : p == 1 ? a.gamma : p == 2 ? a.pk_com : p == 3 ? a.r : a.ok;
      const uint32_t* w = reinterpret_cast<const uint32_t*>(src + item * 64);
      uint32_t xin[8], yin[8], xw[8], yw[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { xin[j] = w[j]; yin[j] = w[8 + j]; }
      valid = valid && !u256_ge(xin, vrfk::Q32) && !u256_ge(yin, vrfk::Q32);
      PtA pa;
      pa.x = fe_from_abi(xw, xin, a.affine_in == 2);      // xw, yw: canonical words (sign of x, encoding of y)
      pa.y = fe_from_abi(yw, yin, a.affine_in == 2);
      FeN xyv = fe_mul(pa.x, pa.y);
      pa.dt = fe_mul(xyv, S::d());
      // a x^2 + y^2 = 1 + d x^2 y^2   <=>   y^2 - ANEG x^2 - 1 = (d x y)(x y)
      FeN x2 = fe_sqr(pa.x), y2 = fe_sqr(pa.y);
      auto lhs = te_curve_lhs<S>(x2, y2);
      valid = fe_eq(lhs, fe_mul(pa.dt, xyv)) && valid;
      if (a.check_mask & (p == 0 ? CHK_INPUT : p == 1 ? CHK_OUTPUT : CHK_PROOF))
        valid = in_prime_subgroup<S>(pa.x, pa.y, a.T.sq) && valid;
      pta_store(a.L.pts + rlc_index(p, n, item) * MSM_PTA_STRIDE, pa);
      // compressed encoding for the challenge: y, sign bit = (x > q - x); slot order pk_com, H, Gamma, R, Ok
      uint32_t e[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] = yw[j];
      if (te_x_sign(xw, a.T.sq.str.flags)) e[7] |= 0x80000000u;
      const int slot = p == 0 ? 1 : p == 1 ? 2 : p == 2 ? 0 : p;
#pragma unroll
      for (int q = 0; q < 5; ++q)
        if (q == slot) {
#pragma unroll
          for (int j = 0; j < 8; ++j) cp[q][j] = e[j];
        }
    }
    rlc_emit_item<S>(a, item, valid, cp, cols);
  }
  rlc_flush_cols(a, cols);
}

// one lane: columns -> scalars mod r; G and B (entry 1*256^0 of the fixed-base combs) become points 3n, 3n+1
template <class S>
__global__ void k_rlc_fixed(RlcArgs a) {
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
    fr_reduce512<S>(k, wide);
    const uint32_t* src = f == 0 ? a.T.g_comb : a.T.b_comb;
    uint32_t* dst = a.L.pts + (3 * a.n + f) * MSM_PTA_STRIDE;
    for (int j = 0; j < PTA_WORDS; ++j) dst[j] = src[j];
    msm_write_digits<S>(a.L.digits, N, 3 * a.n + f, k, false, false);
  }
}

// x || y (64 B) -> ArkworksCodec compressed (32 B); used when a failed affine batch falls back to the
// per-proof kernels
__global__ void __launch_bounds__(BLOCK) k_affine_compress(size_t n, const uint8_t* xy, uint8_t* enc, uint32_t sflags) {
  const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  const uint32_t* w = reinterpret_cast<const uint32_t*>(xy + i * 64);
  uint32_t xw[8], e[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { xw[j] = w[j]; e[j] = w[8 + j]; }
  // coordinates >= q cannot be encoded: poison the encoding (y = 2^255 - 1 >= q) so that decode rejects it
  const bool bad = u256_ge(xw, vrfk::Q32) || u256_ge(e, vrfk::Q32);
  if (te_x_sign(xw, sflags)) e[7] |= 0x80000000u;
  if (bad) {
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = 0xffffffffu;
  }
  store32(enc, i, e);
}
void launch_affine_compress(size_t n, const uint8_t* xy, uint8_t* enc, uint32_t sflags, hipStream_t st) {
  if (n) hipLaunchKernelGGL(k_affine_compress, grid_for(n), dim3(BLOCK), 0, st, n, xy, enc, sflags);
}

template <class S>
static void launch_rlc_t(const RlcArgs& a, uint8_t* fail_flag, hipStream_t st, hipEvent_t* ev) {
  (void)hipMemsetAsync(a.L.flags, 0, 256, st);
  (void)hipMemsetAsync(a.fixed_cols, 0, 16 * sizeof(uint64_t), st);
  if (ev) (void)hipEventRecord(ev[0], st);
  if (a.affine_in) VRF_LAUNCH_MINW(k_rlc_prep_affine, S, a.n, grid_for(a.n), 0, st, a);
  else {
    const size_t lanes_k_ = (a.n + a.k_lane - 1) / a.k_lane;
    const dim3 gk = grid_for(lanes_k_);
    VRF_LAUNCH_MINW(k_rlc_decode, S, lanes_k_, gk, spread_lds_bytes(gk.x), st, a);
  }
  hipLaunchKernelGGL(k_rlc_fixed<S>, dim3(1), dim3(64), 0, st, a);
  if (ev) (void)hipEventRecord(ev[1], st);
  launch_msm_core(a.suite, a.L, nullptr, nullptr, nullptr, fail_flag, st, ev ? ev + 2 : nullptr);
}

void launch_pedersen_rlc(const RlcArgs& a, uint8_t* fail_flag, hipStream_t st, hipEvent_t* ev) {
  if (a.n == 0) return;
  VRF_DISPATCH_SUITE(a.suite, launch_rlc_t<S>(a, fail_flag, st, ev));
}

VRF_NS_END
