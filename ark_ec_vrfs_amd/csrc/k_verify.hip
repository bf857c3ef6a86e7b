// k_verify.hip -- IETF ECVRF batch verification kernels.
// Replaces the body of `ietf::Verifier::verify` (/root/reference src/lib.rs:14).
#include "kernels.h"
#include "tai_find.cuh"

VRF_NS_BEGIN

// the proof's scalars for the Straus stages: c mod r (upstream decodes `Proof::c` with from_le_bytes_mod_order), s as is;
// a non-canonical s is reported InvalidData by the finish stage: keep the digits in range here
template <class S>
VRF_HD void load_cs_reduced(uint32_t c[8], uint32_t s[8]) {
  uint32_t cr[8];
  fr_reduce256<S>(cr, c);
  const bool s_ok = fr_is_canonical<S>(s);
#pragma unroll
  for (int j = 0; j < 8; ++j) { c[j] = s_ok ? cr[j] : 0u; s[j] = s_ok ? s[j] : 0u; }
}

// stage 1: one lane per proof.  Decompress pk, H, Gamma; build their GLV window-table pairs.
// VERIFY_K proofs per lane share one inversion (3K decompression denominators).
template <class S, int MINW>
__global__ void __launch_bounds__(BLOCK, MINW) k_verify_decode(VerifyArgs a) {
  size_t first = ((size_t)blockIdx.x * BLOCK + threadIdx.x) * a.k_lane;
  if (first >= a.n) return;
  verify_decode_multi<S>(a.k_lane, a.T, first, a.n, a.pk, a.h, a.gamma, a.ws.tabs, a.ws.pts, a.ws.flags, a.check_mask);
}

// Verification from alpha (`Input::new(alpha)` inside the call, as a deployed verifier holds pk, alpha and the proof):
// H leaves hash-to-curve as affine x, y -- its encoding goes where the finish stage reads it (`enc`, 32 B per item), its
// GLV table pair straight into slot 1 of the item's tables.  No compression, second square root or subgroup test for H:
// it is a cofactor multiple by construction.
template <class S>
__global__ void __launch_bounds__(BLOCK, 2) k_verify_input_from_alpha(size_t n, BytesView msg, uint8_t* enc, uint32_t* tabs,
                                                                      DevTables T, const uint8_t* tai_ctr) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  const uint8_t* m; uint32_t len;
  bytes_get(msg, i, m, len);
  PtE h = data_to_point<S>(m, len, T.sq, tai_ctr ? tai_ctr[i] : 0u);
  FeN x, y;
  te_to_affine(x, y, h);
  uint32_t e[8];
  te_encode_affine(e, x, y, T.sq.str.flags);
  store32(enc, i, e);
  build_glv_tables<S>(tabs + i * (VERIFY_TABS * WIN_TABLE_WORDS) + 2 * WIN_TABLE_WORDS, x, y);
}
// stage 1 after it: pk and Gamma only
template <class S, int MINW>
__global__ void __launch_bounds__(BLOCK, MINW) k_verify_decode_skip_h(VerifyArgs a) {
  size_t first = ((size_t)blockIdx.x * BLOCK + threadIdx.x) * a.k_lane;
  if (first >= a.n) return;
  verify_decode_multi<S, 2, true>(a.k_lane, a.T, first, a.n, a.pk, nullptr, a.gamma, a.ws.tabs, a.ws.pts, a.ws.flags,
                                  a.check_mask);
}

// stage 1, keyed: only H and Gamma are decompressed; validity also requires a valid, existing key
template <class S, int MINW>
__global__ void __launch_bounds__(BLOCK, MINW) k_verify_decode_keyed(VerifyArgs a) {
  size_t first = ((size_t)blockIdx.x * BLOCK + threadIdx.x) * a.k_lane;
  if (first >= a.n) return;
  verify_decode_multi<S, 2>(a.k_lane, a.T, first, a.n, nullptr, a.h, a.gamma, a.ws.tabs, a.ws.pts, a.ws.flags,
                            a.check_mask);
#pragma unroll 1
  for (int k = 0; k < a.k_lane; ++k) {
    const size_t i = first + k;
    if (i >= a.n) break;
    const uint32_t key = a.key_index[i];
    const bool key_ok = key < a.n_keys && a.key_valid[key] != 0;
    if (!key_ok) a.ws.flags[i] = 0;
  }
}

// stage 2, keyed U half: U = s*G - c*Y from two fixed-base combs (64 mixed additions, no doublings)
template <class S>
__global__ void __launch_bounds__(BLOCK) k_verify_comb_u(VerifyArgs a) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= a.n) return;
  uint32_t c[8], s[8];
  load32(c, a.c, i); load32(s, a.s, i);
  load_cs_reduced<S>(c, s);
  const uint32_t key = a.key_index[i] < a.n_keys ? a.key_index[i] : 0;
  PtE r = gcomb_mul<S>(a.T.g_comb, s);
  r = comb_add<S>(r, a.key_combs + (size_t)key * COMB_WORDS, c, true);
  uint32_t* out = a.ws.pts + i * PROVE_PTS_WORDS;
  fe_store(out, r.X); fe_store(out + NL, r.Y); fe_store(out + 2 * NL, r.Z);
}

// stage 1 for affine inputs (x || y, 64 bytes per point): no square roots.  The compressed encodings
// the challenge hash needs are written to the aux region.
template <class S>
__global__ void __launch_bounds__(BLOCK) k_verify_decode_affine(VerifyArgs a) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= a.n) return;
  uint32_t xy[3][16], enc[3][8];
  const uint32_t* p0 = reinterpret_cast<const uint32_t*>(a.pk + i * 64);
  const uint32_t* p1 = reinterpret_cast<const uint32_t*>(a.h + i * 64);
  const uint32_t* p2 = reinterpret_cast<const uint32_t*>(a.gamma + i * 64);
#pragma unroll
  for (int j = 0; j < 16; ++j) { xy[0][j] = p0[j]; xy[1][j] = p1[j]; xy[2][j] = p2[j]; }
  bool ok = verify_decode_affine_item<S>(enc, xy, a.ws.tabs + i * (VERIFY_TABS * WIN_TABLE_WORDS), a.T.sq, a.check_mask,
                                         a.affine_in == 2);
  uint32_t* aux = a.ws.aux + i * AUX_WORDS;
#pragma unroll
  for (int j = 0; j < 8; ++j) { aux[j] = enc[0][j]; aux[8 + j] = enc[1][j]; aux[16 + j] = enc[2][j]; }
  a.ws.flags[i] = ok ? 1 : 0;
}

// stage 2: one lane per proof and half (U, V in separate launches).  The Straus loops: ~65 % of the work.
template <class S, int HALF>
__global__ void __launch_bounds__(BLOCK) k_verify_straus(VerifyArgs a) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= a.n) return;
  uint32_t c[8], s[8];
  load32(c, a.c, i); load32(s, a.s, i);
  load_cs_reduced<S>(c, s);
  verify_straus_item<S, HALF>(a.ws.pts + i * PROVE_PTS_WORDS + HALF * UV_WORDS, a.T,
                                    a.ws.tabs + i * (VERIFY_TABS * WIN_TABLE_WORDS), c, s);
}

// stage 2 for small batches: both halves in one launch (blockIdx.y = 0: V, 1: U).  Below two waves per SIMD a
// launch lasts as long as one lane's chain of 128 doublings, so the halves overlap instead of queueing.
template <class S>
__global__ void __launch_bounds__(BLOCK) k_verify_straus_both(VerifyArgs a) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= a.n) return;
  uint32_t c[8], s[8];
  load32(c, a.c, i); load32(s, a.s, i);
  load_cs_reduced<S>(c, s);
  const uint32_t* tabs = a.ws.tabs + i * (VERIFY_TABS * WIN_TABLE_WORDS);
  if (blockIdx.y == 0) verify_straus_item<S, 1>(a.ws.pts + i * PROVE_PTS_WORDS + UV_WORDS, a.T, tabs, c, s);
  else verify_straus_item<S, 0>(a.ws.pts + i * PROVE_PTS_WORDS, a.T, tabs, c, s);
}

// stage 3: VERIFY_K proofs per lane share one inversion (2K Z coordinates).  Affine U, V; challenge
// hash; compare.
template <class S, int MINW>
__global__ void __launch_bounds__(BLOCK, MINW) k_verify_finish(VerifyArgs a) {
  size_t first = ((size_t)blockIdx.x * BLOCK + threadIdx.x) * a.k_lane;
  if (first >= a.n) return;
  verify_finish_multi<S>(a.k_lane, first, a.n, a.ws.pts, PROVE_PTS_WORDS, a.pk, a.h, a.gamma,
                         a.affine_in ? a.ws.aux : nullptr, AUX_WORDS, a.c, a.s, a.ad, a.ws.flags, a.status,
                         a.T.sq.str, a.key_index, a.n_keys);
}

template <class S>
static void launch_verify_t(const VerifyArgs& a, hipStream_t st, hipEvent_t* ev) {
  const size_t lanes_k_ = (a.n + a.k_lane - 1) / a.k_lane;      // K proofs per lane: a small grid
  const dim3 gk = grid_for(lanes_k_);
  if (ev) (void)hipEventRecord(ev[0], st);
  if (a.key_index) VRF_LAUNCH_MINW(k_verify_decode_keyed, S, lanes_k_, gk, spread_lds_bytes(gk.x), st, a);
  else if (a.h_in_tabs) VRF_LAUNCH_MINW(k_verify_decode_skip_h, S, lanes_k_, gk, spread_lds_bytes(gk.x), st, a);
  else if (a.affine_in) hipLaunchKernelGGL(k_verify_decode_affine<S>, grid_for(a.n), dim3(BLOCK), 0, st, a);
  else VRF_LAUNCH_MINW(k_verify_decode, S, lanes_k_, gk, spread_lds_bytes(gk.x), st, a);
  if (!ev && !a.key_index && a.n <= STRAUS_FUSE_MAX_ITEMS) {
    dim3 g2 = grid_for(a.n);
    g2.y = 2;
    hipLaunchKernelGGL(k_verify_straus_both<S>, g2, dim3(BLOCK), 0, st, a);
    VRF_LAUNCH_MINW(k_verify_finish, S, lanes_k_, gk, spread_lds_bytes(gk.x), st, a);
    return;
  }
  if (ev) (void)hipEventRecord(ev[1], st);
  hipLaunchKernelGGL((k_verify_straus<S, 1>), grid_for(a.n), dim3(BLOCK), 0, st, a);
  if (ev) (void)hipEventRecord(ev[2], st);
  if (a.key_index) hipLaunchKernelGGL(k_verify_comb_u<S>, grid_for(a.n), dim3(BLOCK), 0, st, a);
  else hipLaunchKernelGGL((k_verify_straus<S, 0>), grid_for(a.n), dim3(BLOCK), 0, st, a);
  if (ev) (void)hipEventRecord(ev[3], st);
  VRF_LAUNCH_MINW(k_verify_finish, S, lanes_k_, gk, spread_lds_bytes(gk.x), st, a);
  if (ev) (void)hipEventRecord(ev[4], st);
}
// tai_ctr / queue: as launch_hash_to_curve (the decode stage that follows overwrites the flag bytes the counters sit in)
void launch_verify_input_from_alpha(int suite, size_t n, BytesView msg, uint8_t* enc, uint32_t* tabs, DevTables T,
                                    hipStream_t st, uint8_t* tai_ctr, unsigned long long* queue) {
  if (!n) return;
  VRF_DISPATCH_SUITE(suite, {
    const uint8_t* ctr = nullptr;
    if constexpr (!S::H2C_ELL2) {
      if (tai_ctr && queue) { launch_tai_find_t<S>(n, msg, tai_ctr, T.sq, queue, st); ctr = tai_ctr; }
    }
    hipLaunchKernelGGL(k_verify_input_from_alpha<S>, grid_for(n), dim3(BLOCK), 0, st, n, msg, enc, tabs, T, ctr);
  });
}
void launch_ietf_verify(const VerifyArgs& a, hipStream_t st, hipEvent_t* ev) {
  if (a.n == 0) return;
  VRF_DISPATCH_SUITE(a.suite, launch_verify_t<S>(a, st, ev));
}

VRF_NS_END
