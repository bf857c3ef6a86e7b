// k_verify.hip -- IETF ECVRF batch verification kernels.
// Replaces the body of `ietf::Verifier::verify` (/root/reference src/lib.rs:14).
#include "kernels.h"

namespace vrf {

// stage 1: one lane per proof.  Decompress pk, H, Gamma; build their radix-16 window tables.
__global__ void __launch_bounds__(BLOCK) k_verify_decode(VerifyArgs a) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= a.n) return;
  uint32_t pk[8], h[8], g[8];
  load32(pk, a.pk, i); load32(h, a.h, i); load32(g, a.gamma, i);
  bool ok = verify_decode_item<SuiteBS>(a.T, pk, h, g, a.ws.tabs + i * (VERIFY_TABS * WIN_TABLE_WORDS));
  a.ws.flags[i] = ok ? 1 : 0;
}

// stage 2: two lanes per proof (U and V).  The Straus loop: >90 % of the work.
__global__ void __launch_bounds__(BLOCK) k_verify_straus(VerifyArgs a) {
  size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  size_t i = t >> 1;
  int half = (int)(t & 1);
  if (i >= a.n) return;
  uint32_t c[8], s[8];
  load32(c, a.c, i); load32(s, a.s, i);
  verify_straus_item<SuiteBS>(a.ws.pts + i * (2 * UV_WORDS) + half * UV_WORDS, a.T,
                              a.ws.tabs + i * (VERIFY_TABS * WIN_TABLE_WORDS), c, s, half);
}

// stage 3: one lane per proof.  Affine U, V; challenge hash; compare.
__global__ void __launch_bounds__(BLOCK) k_verify_finish(VerifyArgs a) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= a.n) return;
  uint32_t pk[8], h[8], g[8], c[8], s[8];
  load32(pk, a.pk, i); load32(h, a.h, i); load32(g, a.gamma, i);
  load32(c, a.c, i); load32(s, a.s, i);
  const uint8_t* ad; uint32_t ad_len;
  bytes_get(a.ad, i, ad, ad_len);
  uint32_t st = verify_finish_item<SuiteBS>(a.ws.pts + i * (2 * UV_WORDS), pk, h, g, c, s,
                                            a.ws.flags[i] != 0, ad, ad_len);
  a.status[i] = (uint8_t)st;
}

void launch_ietf_verify(const VerifyArgs& a, hipStream_t st, hipEvent_t* ev) {
  if (a.n == 0) return;
  if (ev) (void)hipEventRecord(ev[0], st);
  hipLaunchKernelGGL(k_verify_decode, grid_for(a.n), dim3(BLOCK), 0, st, a);
  if (ev) (void)hipEventRecord(ev[1], st);
  hipLaunchKernelGGL(k_verify_straus, grid_for(2 * a.n), dim3(BLOCK), 0, st, a);
  if (ev) (void)hipEventRecord(ev[2], st);
  hipLaunchKernelGGL(k_verify_finish, grid_for(a.n), dim3(BLOCK), 0, st, a);
  if (ev) (void)hipEventRecord(ev[3], st);
}

}  // namespace vrf
