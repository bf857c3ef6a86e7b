// msm.cuh -- shared pieces of the Pippenger MSM (k_msm.hip) and of its callers that prepare points and
// digits themselves (k_rlc.hip: batched Pedersen verification by random linear combination).
//
// Replaces ark_ec::scalar_mul::VariableBaseMSM::msm (named in BASELINE.json north_star; reached from
// /root/reference through `reexports`, src/lib.rs:14).  Signed 11-bit windows: 23 windows x 1024
// buckets.  Scalars k > r/2 are replaced by r - k with the point's sign flipped, so k < 2^252 and the
// top window never carries out.
#pragma once
#include "vrf_core.cuh"

VRF_NS_BEGIN

// 11-bit windows: 1024 buckets x 144 B fill a CU's LDS, ONE workgroup of 8 waves per CU.  Round 4 tried 10-bit windows
// (512 buckets, 72 KB: two workgroups per CU, four waves per SIMD, 26 windows instead of 23) on the hypothesis that the
// 0.72 of the issue slots was an occupancy limit: the batched Pedersen verifier's bucket stage went 7.5 -> 11.4 ms and the
// plain MSM of 2^20 points 2.8 -> 4.8 ms -- the extra windows and the halved chunk per lane (the sort, the head merge and
// the 18-step bucket reduction are per workgroup) cost far more than the occupancy returned.  The code keeps both shapes
// (MSM_BPL buckets per lane); 11 bits is what ships.
constexpr int MSM_C = 11;                         // window bits
constexpr int MSM_W = 23;                         // windows: 11*23 = 253 bits
constexpr int MSM_BUCKETS = 1 << (MSM_C - 1);     // 1024 signed buckets per window
// 512 lanes per workgroup.  Round 4 also tried 1024 (the LDS footprint is the buckets' whatever the lane count, so the one
// workgroup per CU holds 16 waves instead of 8): the bucket stage of the batched Pedersen verifier went 7.6 -> 8.7 ms, the
// plain MSM of 2^20 points 2.86 -> 3.19 ms.  With the 10-bit experiment above and the group-major launch order (no change)
// that makes three measured answers to "why 0.72 of the issue slots": not occupancy, not HBM re-reads.
constexpr int MSM_BLOCK = 512;                    // lanes per workgroup
constexpr int MSM_LOG2_BLOCK = 9;
constexpr int MSM_BPL = MSM_BUCKETS / MSM_BLOCK;  // buckets per lane in the scan and the reduction (2; the code also runs with 1)
static_assert((1 << MSM_LOG2_BLOCK) == MSM_BLOCK && MSM_BPL >= 1 && MSM_BPL <= 2, "workgroup shape");
constexpr int MSM_PT_WORDS = 4 * NL;              // extended point staged in LDS / HBM: X, Y, Z, T
constexpr int MSM_PTA_STRIDE = 32;                // words between the affine-cached points of L.pts: 108 B of point in a
                                                  // 128-B slot, so that a bucket gather touches ONE cache line (27-word
                                                  // records straddled two: 26.7 GB of traffic per 2^20-proof batch)
constexpr int MSM_IDX_BITS = 21;                  // list entry: point index inside its group
constexpr uint32_t MSM_IDX_MASK = (1u << MSM_IDX_BITS) - 1;
constexpr size_t MSM_MAX_PER_GROUP = size_t(1) << MSM_IDX_BITS;

// Device-side layout of one MSM over n points (all regions inside one workspace allocation).
constexpr int MSM_W_SHORT = 12;                   // windows a scalar < 2^128 can reach (11*12 = 132 bits)

// MsmLayout (device-side layout of one MSM over n points): vrf_types.h

// canonical scalar k (< r) -> folded signed radix-2^11 digits of point i.  `negate` flips the sign of the
// term (callers that subtract a term); `zero` drops the point from the sum.
template <class S>
VRF_HD void msm_write_digits(int16_t* digits, size_t n, size_t i, const uint32_t k_in[8], bool negate,
                             bool zero) {
  uint32_t k[8], nk[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) k[j] = zero ? 0u : k_in[j];
  {
    uint32_t borrow = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      uint64_t d = (uint64_t)S::r32(j) - k[j] - borrow;
      nk[j] = (uint32_t)d;
      borrow = (uint32_t)(d >> 63);
    }
  }
  // fold: k > r - k  ->  use r - k and flip the sign of every digit
  bool flip = false, decided = false;
#pragma unroll
  for (int j = 7; j >= 0; --j)
    if (!decided && nk[j] != k[j]) { flip = nk[j] < k[j]; decided = true; }
  flip = flip && !zero;
#pragma unroll
  for (int j = 0; j < 8; ++j) k[j] = flip ? nk[j] : k[j];
  const bool neg = flip != negate;
  uint32_t carry = 0;
#pragma unroll 1
  for (int w = 0; w < MSM_W; ++w) {
    int bit = w * MSM_C, wi = bit >> 5, sh = bit & 31;
    uint32_t lo = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) if (j == wi) lo = k[j];
    uint32_t hi = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) if (j == wi + 1) hi = k[j];
    uint32_t v = (sh ? ((lo >> sh) | (hi << (32 - sh))) : lo) & ((1u << MSM_C) - 1);
    v += carry;
    int d = (int)v;
    carry = 0;
    if (v > (uint32_t)MSM_BUCKETS) { d = (int)v - (1 << MSM_C); carry = 1; }
    digits[(size_t)w * n + i] = (int16_t)(neg ? -d : d);
  }
}

// MSM index of point class p (H, Gamma, pk_com, R, Ok) of proof i: the classes with full-size scalars
// come first, then G and B (3n, 3n+1), then the two classes whose scalars are the 128-bit weights.
VRF_HD size_t rlc_index(int p, size_t n, size_t i) { return (size_t)p * n + i + (p >= 3 ? 2 : 0); }

template <class S>
VRF_HD void rlc_weights(uint32_t z[8], uint32_t zp[8], const uint8_t* seed, const uint8_t* root, uint64_t index) {
  Sha512 h;
  sha512_init(h);
  constexpr char tag[] = "vrfhip-rlc-v2";
#pragma unroll
  for (int i = 0; i < 13; ++i) sha512_put_byte(h, (uint8_t)tag[i]);
  sha512_put_bytes(h, seed, 32);
  sha512_put_bytes(h, root, 32);      // digest of the launch group's inputs (digest.cuh)
#pragma unroll
  for (int i = 0; i < 8; ++i) sha512_put_byte(h, (uint8_t)(index >> (8 * i)));
  sha512_final(h);
  uint32_t le[16];
  sha512_le512(le, h);
#pragma unroll
  for (int i = 0; i < 4; ++i) { z[i] = le[i]; zp[i] = le[4 + i]; z[4 + i] = 0; zp[4 + i] = 0; }
  // weights = 1 (mod 8): a weight then fixes every point of order dividing the cofactor (4 or 8), so a lone
  // small-order defect is never annihilated (defence in depth; the subgroup precondition remains)
  z[0] = (z[0] & ~7u) | 1u;
  zp[0] = (zp[0] & ~7u) | 1u;
}

// host side (k_msm.hip)
int msm_groups(size_t n, size_t n_long, int cus);
size_t msm_workspace_bytes(size_t n, int groups);
MsmLayout msm_layout(size_t n, size_t n_long, int groups, void* ws);
// launch_msm_core, launch_pedersen_rlc, launch_affine_compress: launchers.inc

VRF_NS_END
