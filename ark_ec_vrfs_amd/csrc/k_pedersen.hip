// k_pedersen.hip -- Pedersen VRF batch verification kernels.
// Replaces `pedersen::Verifier::verify` (/root/reference src/lib.rs:14); proving shares the
// kernels of k_prove.hip (flag `pedersen`).
#include "kernels.h"

VRF_NS_BEGIN

template <class S, int MINW>
__global__ void __launch_bounds__(BLOCK, MINW) k_ped_verify_decode(PedersenVerifyArgs a) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= a.n) return;
  uint32_t enc[5][8];
  load32(enc[0], a.h, i); load32(enc[1], a.gamma, i); load32(enc[2], a.pk_com, i);
  load32(enc[3], a.r, i); load32(enc[4], a.ok, i);
  const uint8_t* ad; uint32_t ad_len;
  bytes_get(a.ad, i, ad, ad_len);
  uint32_t c[8];
  bool ok = pedersen_verify_decode_item<S>(c, a.T, enc, ad, ad_len,
                                                 a.ws.tabs + i * (VERIFY_TABS * WIN_TABLE_WORDS),
                                                 a.ws.pts + i * PROVE_PTS_WORDS, a.check_mask);
  uint32_t* aux = a.ws.aux + i * AUX_WORDS;
#pragma unroll
  for (int j = 0; j < 8; ++j) aux[j] = c[j];
  a.ws.flags[i] = ok ? 1 : 0;
}

// HALF 0: s*H - c*Gamma ; HALF 1: s*G - c*pk_com + sb*B.  Separate launches keep each wave uniform.
template <class S, int HALF>
__global__ void __launch_bounds__(BLOCK) k_ped_verify_straus(PedersenVerifyArgs a) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= a.n) return;
  uint32_t c[8], s[8], sb[8];
  load32(s, a.s, i); load32(sb, a.sb, i);
#pragma unroll
  for (int j = 0; j < 8; ++j) c[j] = a.ws.aux[i * AUX_WORDS + j];
  // out-of-range scalars are reported InvalidData by the finish stage; keep digits in range here
  if (!fr_is_canonical<S>(s) || !fr_is_canonical<S>(sb)) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[j] = 0; sb[j] = 0; }
  }
  pedersen_verify_straus_item<S, HALF>(a.ws.pts + i * PROVE_PTS_WORDS + HALF * UV_WORDS, a.T,
                                             a.ws.tabs + i * (VERIFY_TABS * WIN_TABLE_WORDS), c, s, sb);
}

// small batches: both halves in one launch (blockIdx.y picks the half), see k_verify_straus_both
template <class S>
__global__ void __launch_bounds__(BLOCK) k_ped_verify_straus_both(PedersenVerifyArgs a) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= a.n) return;
  uint32_t c[8], s[8], sb[8];
  load32(s, a.s, i); load32(sb, a.sb, i);
#pragma unroll
  for (int j = 0; j < 8; ++j) c[j] = a.ws.aux[i * AUX_WORDS + j];
  if (!fr_is_canonical<S>(s) || !fr_is_canonical<S>(sb)) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[j] = 0; sb[j] = 0; }
  }
  const uint32_t* tabs = a.ws.tabs + i * (VERIFY_TABS * WIN_TABLE_WORDS);
  if (blockIdx.y == 0) pedersen_verify_straus_item<S, 1>(a.ws.pts + i * PROVE_PTS_WORDS + UV_WORDS, a.T, tabs, c, s, sb);
  else pedersen_verify_straus_item<S, 0>(a.ws.pts + i * PROVE_PTS_WORDS, a.T, tabs, c, s, sb);
}

template <class S>
__global__ void __launch_bounds__(BLOCK) k_ped_verify_finish(PedersenVerifyArgs a) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= a.n) return;
  uint32_t s[8], sb[8];
  load32(s, a.s, i); load32(sb, a.sb, i);
  a.status[i] = (uint8_t)pedersen_verify_finish_item<S>(a.ws.pts + i * PROVE_PTS_WORDS, s, sb,
                                                              a.ws.flags[i] != 0);
}

template <class S>
static void launch_ped_t(const PedersenVerifyArgs& a, hipStream_t st, hipEvent_t* ev) {
  if (ev) (void)hipEventRecord(ev[0], st);
  VRF_LAUNCH_MINW(k_ped_verify_decode, S, a.n, grid_for(a.n), 0, st, a);
  if (ev) (void)hipEventRecord(ev[1], st);
  if (!ev && a.n <= STRAUS_FUSE_MAX_ITEMS) {
    dim3 g2 = grid_for(a.n);
    g2.y = 2;
    hipLaunchKernelGGL(k_ped_verify_straus_both<S>, g2, dim3(BLOCK), 0, st, a);
    hipLaunchKernelGGL(k_ped_verify_finish<S>, grid_for(a.n), dim3(BLOCK), 0, st, a);
    return;
  }
  hipLaunchKernelGGL((k_ped_verify_straus<S, 0>), grid_for(a.n), dim3(BLOCK), 0, st, a);
  if (ev) (void)hipEventRecord(ev[2], st);
  hipLaunchKernelGGL((k_ped_verify_straus<S, 1>), grid_for(a.n), dim3(BLOCK), 0, st, a);
  if (ev) (void)hipEventRecord(ev[3], st);
  hipLaunchKernelGGL(k_ped_verify_finish<S>, grid_for(a.n), dim3(BLOCK), 0, st, a);
  if (ev) (void)hipEventRecord(ev[4], st);
}
void launch_pedersen_verify(const PedersenVerifyArgs& a, hipStream_t st, hipEvent_t* ev) {
  if (a.n == 0) return;
  VRF_DISPATCH_SUITE(a.suite, launch_ped_t<S>(a, st, ev));
}

VRF_NS_END
