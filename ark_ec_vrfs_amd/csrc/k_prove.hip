// k_prove.hip -- IETF ECVRF batch proving kernels.
// Replaces `Input::new`, `Secret::output` and `ietf::Prover::prove` (/root/reference src/lib.rs:14-16).
#include "kernels.h"

namespace vrf {

// stage 1: H = hash_to_curve(msg) (or decode a given H), enc(H), nonce, window table of H
__global__ void __launch_bounds__(BLOCK) k_prove_prepare(ProveArgs a) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= a.n) return;
  uint32_t sk[8], hg[8], h_enc[8], k[8];
  load32(sk, a.sk, i);
  const uint8_t* msg = nullptr; uint32_t msg_len = 0;
  if (a.h_given) load32(hg, a.h_given, i); else bytes_get(a.msg, i, msg, msg_len);
  bool ok = prove_prepare_item<SuiteBS>(h_enc, k, a.ws.tabs + i * WIN_TABLE_WORDS, a.T, sk, msg,
                                        msg_len, a.h_given ? hg : nullptr);
  uint32_t* aux = a.ws.aux + i * 16;
#pragma unroll
  for (int j = 0; j < 8; ++j) { aux[j] = h_enc[j]; aux[8 + j] = k[j]; }
  a.ws.flags[i] = ok ? 1 : 0;
}

// stage 2: two lanes per proof: lane 0 -> (sk*H, sk*G), lane 1 -> (k*H, k*G)
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
    for (int j = 0; j < 8; ++j) sc[j] = a.ws.aux[i * 16 + 8 + j];
  }
  // a non-canonical secret is reported InvalidData by stage 3; keep the digits in range here
  if (!fr_is_canonical<SuiteBS>(sc)) {
#pragma unroll
    for (int j = 0; j < 8; ++j) sc[j] = 0;
  }
  prove_mul_item<SuiteBS>(a.ws.pts + i * PROVE_PTS_WORDS + half * 2 * UV_WORDS, a.T,
                          a.ws.tabs + i * WIN_TABLE_WORDS, sc);
}

// stage 3: encodings, challenge, s = k + c*sk
__global__ void __launch_bounds__(BLOCK) k_prove_finish(ProveArgs a) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= a.n) return;
  uint32_t sk[8], h_enc[8], k[8], g[8], c[8], s[8], pk[8];
  load32(sk, a.sk, i);
#pragma unroll
  for (int j = 0; j < 8; ++j) { h_enc[j] = a.ws.aux[i * 16 + j]; k[j] = a.ws.aux[i * 16 + 8 + j]; }
  const uint8_t* ad; uint32_t ad_len;
  bytes_get(a.ad, i, ad, ad_len);
  prove_finish_item<SuiteBS>(g, c, s, pk, a.ws.pts + i * PROVE_PTS_WORDS, h_enc, sk, k, ad, ad_len);
  bool ok = a.ws.flags[i] != 0;
  if (!ok) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { g[j] = 0; c[j] = 0; s[j] = 0; pk[j] = 0; }
  }
  store32(a.gamma, i, g); store32(a.c, i, c); store32(a.s, i, s);
  if (a.pk_out) store32(a.pk_out, i, pk);
  if (a.h_out) store32(a.h_out, i, h_enc);
  if (a.status) a.status[i] = ok ? ST_OK : ST_INVALID_DATA;
}

void launch_ietf_prove(const ProveArgs& a, hipStream_t st, hipEvent_t* ev) {
  if (a.n == 0) return;
  if (ev) (void)hipEventRecord(ev[0], st);
  hipLaunchKernelGGL(k_prove_prepare, grid_for(a.n), dim3(BLOCK), 0, st, a);
  if (ev) (void)hipEventRecord(ev[1], st);
  hipLaunchKernelGGL(k_prove_mul, grid_for(2 * a.n), dim3(BLOCK), 0, st, a);
  if (ev) (void)hipEventRecord(ev[2], st);
  hipLaunchKernelGGL(k_prove_finish, grid_for(a.n), dim3(BLOCK), 0, st, a);
  if (ev) (void)hipEventRecord(ev[3], st);
}

}  // namespace vrf
