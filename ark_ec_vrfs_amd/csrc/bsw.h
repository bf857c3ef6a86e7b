// bsw.h -- launch interface of `suites::bandersnatch_sw` (/root/reference src/lib.rs:14) between the C ABI (api.hip) and
// k_bsw.hip.  The suite runs on a Bandersnatch (base field 0) context: same tables, same workspace, same launch arguments
// (vrf_types.h) as the twisted-Edwards suite, with 33-byte compressed short-Weierstrass points in every point array
// (bsw_core.cuh).
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
// IETF and (a.pedersen) Pedersen proving; a.out_affine: Gamma, pk / pk_com, R, Ok as Weierstrass x || y (64 B, canonical).
// ev (nullable): 5 events as launch_ietf_prove
void launch_bsw_prove(const ProveArgs& a, hipStream_t st, hipEvent_t* ev = nullptr);
// a.affine_in: pk, h, gamma as Weierstrass x || y; a.key_index: keyed verification (a.pk = the key set's encodings);
// a.h_in_tabs must be 0; a.k_lane is not read
void launch_bsw_ietf_verify(const VerifyArgs& a, hipStream_t st, hipEvent_t* ev = nullptr);
void launch_bsw_pedersen_verify(const PedersenVerifyArgs& a, hipStream_t st, hipEvent_t* ev = nullptr);
void launch_bsw_keyset_build(size_t n_keys, uint8_t* pks, uint32_t* xy, uint8_t* valid, uint32_t* combs, uint32_t* prefix, DevTables T,
                             hipStream_t st);
// the Edwards MSM's sum (te_xy: x || y canonical; status[0] its verdict) -> 33-byte encoding, Weierstrass x || y (nullable)
void launch_bsw_msm_out(const uint8_t* te_xy, uint8_t* out33, uint8_t* out_xy, uint8_t* status, hipStream_t st);
// the batched Pedersen verifier (launch_pedersen_rlc's contract, msm.cuh layout); a.affine_in: Weierstrass x || y points
// (and n x 64 B x || y -> n x 33 B encodings, for the per-proof fallback of a failed x || y batch; mont256: the form of xy)
void launch_bsw_affine_compress(size_t n, const uint8_t* xy, int mont256, uint8_t* enc33, hipStream_t st);
void launch_bsw_pedersen_rlc(const RlcArgs& a, uint8_t* fail_flag, hipStream_t st, hipEvent_t* ev = nullptr);

VRF_NS_END
