// bsw.h -- launch interface of `suites::bandersnatch_sw` (/root/reference src/lib.rs:14) between the C ABI (api.hip) and
// k_bsw.hip.  The suite runs on a Bandersnatch (base field 0) context: same tables, same workspace, same launch arguments
// (vrf_types.h) as the twisted-Edwards suite, with 33-byte compressed short-Weierstrass points in every point array
// (bsw_core.cuh).  Not built for this suite: x || y inputs / outputs of the schemes, key sets, MSM over caller-supplied bases.
#pragma once
#include "kernels.h"

VRF_NS_BEGIN

void launch_bsw_secret_from_seed(size_t n, const uint8_t* seeds, uint32_t seed_len, uint8_t* sk, uint8_t* pk, DevTables T,
                                 hipStream_t st);
// tai_ctr ([n] bytes) + queue (8-byte device counter): scratch of the try-and-increment counter search (tai_find.cuh)
void launch_bsw_hash_to_curve(size_t n, BytesView msg, uint8_t* points, DevTables T, hipStream_t st, uint8_t* tai_ctr,
                              unsigned long long* queue);
void launch_bsw_output_hash(size_t n, const uint8_t* gamma, uint8_t* hash, DevTables T, hipStream_t st);
// xy (nullable): the short-Weierstrass x || y of the valid points (zeros otherwise, and for the point at infinity)
void launch_bsw_point_validate(size_t n, const uint8_t* pts, uint8_t* xy, uint8_t* status, DevTables T, hipStream_t st);
// IETF and (a.pedersen) Pedersen proving; a.out_affine must be 0.  ev (nullable): 5 events as launch_ietf_prove
void launch_bsw_prove(const ProveArgs& a, hipStream_t st, hipEvent_t* ev = nullptr);
// a.affine_in, a.h_in_tabs, a.key_index must be 0 / NULL; a.k_lane is not read
void launch_bsw_ietf_verify(const VerifyArgs& a, hipStream_t st, hipEvent_t* ev = nullptr);
void launch_bsw_pedersen_verify(const PedersenVerifyArgs& a, hipStream_t st, hipEvent_t* ev = nullptr);
// the batched Pedersen verifier (launch_pedersen_rlc's contract, msm.cuh layout); a.affine_in must be 0
void launch_bsw_pedersen_rlc(const RlcArgs& a, uint8_t* fail_flag, hipStream_t st, hipEvent_t* ev = nullptr);

VRF_NS_END
