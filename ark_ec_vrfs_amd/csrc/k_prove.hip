// k_prove.hip -- IETF ECVRF batch proving kernels.
// Replaces `Input::new`, `Secret::output` and `ietf::Prover::prove` (/root/reference src/lib.rs:14-16).
//
// Compiled once per (suite, stage) -- Makefile: -DVRF_FIELD=f -DVRF_PROVE_SUITE=1|2|3|4 -DVRF_PROVE_PART=1|2|3 ->
// k_prove_{bs,jj,ed,bj}_{1,2,3}.o -- so that the sets of kernels build in parallel (this file was the long pole of the build).
#include "kernels.h"
#include "tai_find.cuh"
#include <algorithm>

#ifndef VRF_PROVE_SUITE
#define VRF_PROVE_SUITE 1
#endif
#ifndef VRF_PROVE_PART
#define VRF_PROVE_PART 1
#endif

VRF_NS_BEGIN

#if VRF_PROVE_SUITE == 2
using ProveSuite = SuiteJJ;
#define PROVE_STAGE(name) name##_jj
#elif VRF_PROVE_SUITE == 3
using ProveSuite = SuiteED;
#define PROVE_STAGE(name) name##_ed
#elif VRF_PROVE_SUITE == 4
using ProveSuite = SuiteBJ;
#define PROVE_STAGE(name) name##_bj
#else
using ProveSuite = SuiteBS;
#define PROVE_STAGE(name) name##_bs
#endif
// the three stages of a suite, each defined by its own translation unit
void PROVE_STAGE(launch_prove_stage1)(const ProveArgs& a, hipStream_t st);
void PROVE_STAGE(launch_prove_stage2)(const ProveArgs& a, hipStream_t st);
void PROVE_STAGE(launch_prove_stage3)(const ProveArgs& a, hipStream_t st);

#if VRF_PROVE_PART == 1

// stage 1: H = hash_to_curve(msg) (or decode a given H), enc(H), nonce, window table of H
template <class S, int MINW>
__global__ void __launch_bounds__(BLOCK, MINW) k_prove_prepare(ProveArgs a) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= a.n) return;
  uint32_t sk[8], hg[8], h_enc[8], k[8];
  load32(sk, a.sk, i);
  const uint8_t* msg = nullptr; uint32_t msg_len = 0;
  if (a.h_given) load32(hg, a.h_given, i); else bytes_get(a.msg, i, msg, msg_len);
  // try-and-increment suites: k_tai_find left the first decodable counter in the item's flag byte
  const uint32_t tai_start = (!S::H2C_ELL2 && !a.h_given) ? a.ws.flags[i] : 0u;
  bool ok = prove_prepare_item<S>(h_enc, k, a.ws.tabs + i * ProveLayout<S>::TAB_WORDS, a.T, sk, msg,
                                        msg_len, a.h_given ? hg : nullptr, tai_start, a.check_mask);
  uint32_t* aux = a.ws.aux + i * AUX_WORDS;
#pragma unroll
  for (int j = 0; j < 8; ++j) { aux[j] = h_enc[j]; aux[8 + j] = k[j]; }
  if (a.pedersen) {
    const uint8_t* ad; uint32_t ad_len;
    bytes_get(a.ad, i, ad, ad_len);
    uint32_t b[8], kb[8];
    pedersen_blinding<S>(b, sk, h_enc, ad, ad_len, a.T.sq.str);
    nonce_rfc8032<S>(kb, b, h_enc);
#pragma unroll
    for (int j = 0; j < 8; ++j) { aux[16 + j] = b[j]; aux[24 + j] = kb[j]; }
  }
  a.ws.flags[i] = ok ? 1 : 0;
}

// stage 1 for the common case (Elligator suite, H from messages): PROVE_K proofs per lane share the two
// inversions of hash-to-curve.  The Pedersen extras (blinding, second nonce) are added per item.
template <class S, int MINW>
__global__ void __launch_bounds__(BLOCK, MINW) k_prove_prepare_multi(ProveArgs a) {
  if constexpr (S::H2C_ELL2) {
    size_t first = ((size_t)blockIdx.x * BLOCK + threadIdx.x) * a.k_lane;
    if (first >= a.n) return;
    prove_prepare_multi<S>(a.k_lane, a.T, first, a.n, a.sk, a.msg, a.ws.tabs, a.ws.pts, a.ws.aux, AUX_WORDS, a.ws.flags);
    if (a.pedersen) {
#pragma unroll 1
      for (int j = 0; j < a.k_lane; ++j) {
        size_t i = first + j;
        if (i >= a.n) break;
        uint32_t sk[8], h_enc[8], b[8], kb[8];
        load32(sk, a.sk, i);
        uint32_t* aux = a.ws.aux + i * AUX_WORDS;
#pragma unroll
        for (int t = 0; t < 8; ++t) h_enc[t] = aux[t];
        const uint8_t* ad; uint32_t ad_len;
        bytes_get(a.ad, i, ad, ad_len);
        pedersen_blinding<S>(b, sk, h_enc, ad, ad_len, a.T.sq.str);
        nonce_rfc8032<S>(kb, b, h_enc);
#pragma unroll
        for (int t = 0; t < 8; ++t) { aux[16 + t] = b[t]; aux[24 + t] = kb[t]; }
      }
    }
  }
}

void PROVE_STAGE(launch_prove_stage1)(const ProveArgs& a, hipStream_t st) {
  using S = ProveSuite;
  const size_t lanes_k_ = (a.n + a.k_lane - 1) / a.k_lane;      // K proofs per lane: a small grid
  const dim3 gk = grid_for(lanes_k_);
  if (S::H2C_ELL2 && !a.h_given) {
    VRF_LAUNCH_MINW(k_prove_prepare_multi, S, lanes_k_, gk, spread_lds_bytes(gk.x), st, a);
  } else {
    if (!S::H2C_ELL2 && !a.h_given) {
      launch_tai_find_t<S>(a.n, a.msg, a.ws.flags, a.T.sq, a.tai_queue, st);
    }
    VRF_LAUNCH_MINW(k_prove_prepare, S, a.n, grid_for(a.n), 0, st, a);
  }
}
#endif  // part 1

#if VRF_PROVE_PART == 2
// stage 2: two lanes per proof: lane 0 -> (sk*H, sk*G), lane 1 -> (k*H, k*G)
template <class S, bool CT>
__global__ void __launch_bounds__(BLOCK) k_prove_mul(ProveArgs a) {
  size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  size_t i = t >> 1;
  int half = (int)(t & 1);
  if (i >= a.n) return;
  uint32_t sc[8];
  if (half == 0) {
    load32(sc, a.sk, i);
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) sc[j] = a.ws.aux[i * AUX_WORDS + 8 + j];
  }
  // a non-canonical secret is reported InvalidData by stage 3; keep the digits in range here
  if (!fr_is_canonical<S>(sc)) {
#pragma unroll
    for (int j = 0; j < 8; ++j) sc[j] = 0;
  }
  prove_mul_item<S, CT>(a.ws.pts + i * PROVE_PTS_WORDS + half * 2 * UV_WORDS, a.T,
                          a.ws.tabs + i * ProveLayout<S>::TAB_WORDS, sc,
                          a.pedersen ? a.ws.aux + i * AUX_WORDS + 16 + half * 8 : nullptr);
}

void PROVE_STAGE(launch_prove_stage2)(const ProveArgs& a, hipStream_t st) {
  // CHK_CT_TABLES (VRFHIP_FLAG_CT_TABLES): the instantiation whose window lookups read all eight entries
  if (a.check_mask & CHK_CT_TABLES) hipLaunchKernelGGL((k_prove_mul<ProveSuite, true>), grid_for(2 * a.n), dim3(BLOCK), 0, st, a);
  else hipLaunchKernelGGL((k_prove_mul<ProveSuite, false>), grid_for(2 * a.n), dim3(BLOCK), 0, st, a);
}
#endif  // part 2

#if VRF_PROVE_PART == 3
// stage 3: PROVE_K proofs per lane share the inversion of their 4K projective Z; then per item the
// challenge and s = k + c*sk (Pedersen: also sb = kb + c*b).
// 64-byte affine output of item i: x (canonical words from the encode stage) || y (the encoding without its sign bit)
VRF_HD void store_xy(uint8_t* base, size_t i, const uint32_t* xw, const uint32_t enc[8], bool ok) {
  uint32_t* p = reinterpret_cast<uint32_t*>(base + i * 64);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    p[k] = ok ? xw[k] : 0u;
    p[8 + k] = ok ? (k == 7 ? enc[k] & 0x7fffffffu : enc[k]) : 0u;
  }
}

template <class S, int MINW>
__global__ void __launch_bounds__(BLOCK, MINW) k_prove_finish(ProveArgs a) {
  size_t first = ((size_t)blockIdx.x * BLOCK + threadIdx.x) * a.k_lane;
  if (first >= a.n) return;
  const int tstride = ProveLayout<S>::TAB_WORDS;
  prove_encode_multi<S>(a.k_lane, first, a.n, a.ws.pts, a.ws.tabs, tstride, a.T.sq.str.flags);
#pragma unroll 1
  for (int jj = 0; jj < a.k_lane; ++jj) {
    size_t i = first + jj;
    if (i >= a.n) break;
    uint32_t sk[8], h_enc[8], k[8], g[8], c[8], s[8], pk[8], rr[8], okp[8];
    load32(sk, a.sk, i);
    const uint32_t* aux = a.ws.aux + i * AUX_WORDS;
    const uint32_t* enc = a.ws.tabs + i * tstride + PROVE_ENC_OFF;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      h_enc[j] = aux[j]; k[j] = aux[8 + j];
      g[j] = enc[j]; pk[j] = enc[8 + j]; okp[j] = enc[16 + j]; rr[j] = enc[24 + j];
    }
    const uint8_t* ad; uint32_t ad_len;
    bytes_get(a.ad, i, ad, ad_len);
    prove_respond_item<S>(c, s, enc, h_enc, sk, k, ad, ad_len, a.T.sq.str);
    bool ok = a.ws.flags[i] != 0;
    if (a.pedersen) {
      uint32_t b[8], kb[8], cb[8], sb[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { b[j] = aux[16 + j]; kb[j] = aux[24 + j]; }
      fr_mul<S>(cb, c, b);
      fr_add<S>(sb, cb, kb);                    // sb = kb + c*b
      if (!ok) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { sb[j] = 0; b[j] = 0; rr[j] = 0; okp[j] = 0; }
      }
      if (a.out_affine) {
        store_xy(a.r_out, i, enc + 24 + PROVE_X_OFF, rr, ok);
        store_xy(a.ok_out, i, enc + 16 + PROVE_X_OFF, okp, ok);
      } else {
        store32(a.r_out, i, rr); store32(a.ok_out, i, okp);
      }
      store32(a.sb_out, i, sb);
      if (a.blinding_out) store32(a.blinding_out, i, b);
    }
    if (!ok) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { g[j] = 0; c[j] = 0; s[j] = 0; pk[j] = 0; }
    }
    store32(a.s, i, s);
    if (a.c) store32(a.c, i, c);
    if (a.out_affine) {
      // x || y instead of the compressed encoding: the caller builds typed points without a square root
      store_xy(a.gamma, i, enc + PROVE_X_OFF, g, ok);
      if (a.pk_out) store_xy(a.pk_out, i, enc + 8 + PROVE_X_OFF, pk, ok);
    } else {
      store32(a.gamma, i, g);
      if (a.pk_out) store32(a.pk_out, i, pk);
    }
    if (a.h_out) store32(a.h_out, i, h_enc);
    if (a.status) a.status[i] = ok ? ST_OK : ST_INVALID_DATA;
  }
}

void PROVE_STAGE(launch_prove_stage3)(const ProveArgs& a, hipStream_t st) {
  using S = ProveSuite;
  const size_t lanes_k_ = (a.n + a.k_lane - 1) / a.k_lane;
  const dim3 gk = grid_for(lanes_k_);
  VRF_LAUNCH_MINW(k_prove_finish, S, lanes_k_, gk, spread_lds_bytes(gk.x), st, a);
}
#endif  // part 3

// the dispatcher of a field lives in one object (stage 2 of the field's first suite)
#if VRF_PROVE_PART == 2 && (VRF_PROVE_SUITE == 1 || VRF_PROVE_SUITE == 3 || VRF_PROVE_SUITE == 4)
#if VRF_PROVE_SUITE == 1
void launch_prove_stage1_jj(const ProveArgs& a, hipStream_t st);
void launch_prove_stage2_jj(const ProveArgs& a, hipStream_t st);
void launch_prove_stage3_jj(const ProveArgs& a, hipStream_t st);
#endif
void launch_ietf_prove(const ProveArgs& a, hipStream_t st, hipEvent_t* ev) {
  if (a.n == 0) return;
#if VRF_PROVE_SUITE == 1
  const bool jj = a.suite == SUITE_JJ;
#define PROVE_CALL(stage) do { if (jj) stage##_jj(a, st); else stage##_bs(a, st); } while (0)
#else
#define PROVE_CALL(stage) PROVE_STAGE(stage)(a, st)
#endif
  if (ev) (void)hipEventRecord(ev[0], st);
  PROVE_CALL(launch_prove_stage1);
  if (ev) (void)hipEventRecord(ev[1], st);
  PROVE_CALL(launch_prove_stage2);
  if (ev) { (void)hipEventRecord(ev[2], st); (void)hipEventRecord(ev[3], st); }
  PROVE_CALL(launch_prove_stage3);
  if (ev) (void)hipEventRecord(ev[4], st);
#undef PROVE_CALL
}
#endif

VRF_NS_END
