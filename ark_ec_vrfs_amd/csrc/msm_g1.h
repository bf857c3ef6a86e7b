// msm_g1.h -- launch interface of the BLS12-381 G1 multi-scalar multiplication (k_msm_g1.hip) for api.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace vrf {

constexpr int G1_C = 10;                       // signed window bits: 512 buckets x 168 B = 84 KiB of LDS
constexpr int G1_BUCKETS = 1 << (G1_C - 1);
constexpr int G1_BLOCK = 512;                  // lanes per workgroup = buckets
constexpr int G1_W_FULL = 26;                  // windows of a 255-bit scalar (260 bits)
constexpr int G1_W_SHORT = 13;                 // windows of a 128-bit weight (130 bits)
constexpr int G1_PT_WORDS = 42;                // projective (X, Y, Z), 14 limbs each
constexpr int G1_AFF_WORDS = 28;               // Montgomery affine (x, y)
constexpr int G1_AFF_STRIDE = 32;              // words between the points of L.pts: 112 B in a 128-B slot (one cache line per gather)
constexpr int G1_IDX_BITS = 21;
constexpr size_t G1_MAX_PER_GROUP = size_t(1) << G1_IDX_BITS;

// Device layout of one call: `sets` point sets (1 = plain MSM, 2 = the A and B sides of a batched pairing check)
// that share nothing but the schedule.
struct G1MsmLayout {
  size_t n;
  int sets, windows, groups;
  size_t per_group, list_cap;
  uint32_t* pts;      // [sets][n][G1_AFF_STRIDE]  Montgomery affine coordinates
  int16_t* digits;    // [sets][windows][n]        signed digits in [-511, 512]; 0 = the point takes no part
  uint32_t* lists;    // [sets*windows*groups][list_cap] bucket-sorted entries, lane-transposed
  uint32_t* heads;    // [sets*windows*groups][G1_BLOCK][G1_PT_WORDS] first-run partial sums
  uint32_t* part;     // [sets][windows][groups][G1_PT_WORDS] per-workgroup window sums
  uint8_t* sums;      // [sets][96]  results in the wire format (x || y, 48-byte little-endian; all-zero = infinity)
  uint8_t* flags;     // [256]  flags[0] != 0: an input of the plain MSM was invalid
};
int g1_msm_groups(size_t n, int sets, int windows, int cus);
size_t g1_msm_workspace_bytes(size_t n, int sets, int windows, int groups);
G1MsmLayout g1_msm_layout(size_t n, int sets, int windows, int groups, void* ws);

// Batched pairing check, G1 side: validates the 2n points (g1: n x 192 B), derives the 128-bit weights
// z_i = SHA-512("vrfhip-pairing-rlc-v2" || seed || d_root[32] || u64_le(index0 + i))[0..16] (d_root: the batch digest of
// digest.cuh over the g1 items, device memory), writes status[i] in {0, 2} and leaves
// (sum z_i A_i, sum z_i B_i) in L.sums -- exactly one g1 item for the pairing kernel.  L: sets = 2, windows = 13.
// ev (nullable, 3 events): after prep, after buckets, after final.
void launch_g1_rlc(const G1MsmLayout& L, const uint8_t* g1, const uint8_t seed[32], const uint8_t* d_root, uint64_t index0,
                   uint8_t* status,
                   hipStream_t st, hipEvent_t* ev = nullptr);
// `VariableBaseMSM::msm` on G1: bases n x 96 B, scalars n x 32 B little-endian (< r); result in L.sums[0..96],
// status1[0] = 0 / 2 (a coordinate >= p, a point off the curve or a scalar >= r).  L: sets = 1, windows = 26.
void launch_g1_msm(const G1MsmLayout& L, const uint8_t* bases, const uint8_t* scalars, uint8_t* status1, hipStream_t st);

}  // namespace vrf
