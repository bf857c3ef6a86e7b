// msm.cuh -- shared pieces of the Pippenger MSM (k_msm.hip) and of its callers that prepare points and
// digits themselves (k_rlc.hip: batched Pedersen verification by random linear combination).
//
// Replaces ark_ec::scalar_mul::VariableBaseMSM::msm (named in BASELINE.json north_star; reached from
// /root/reference through `reexports`, src/lib.rs:14).  Signed 11-bit windows: 23 windows x 1024
// buckets.  Scalars k > r/2 are replaced by r - k with the point's sign flipped, so k < 2^252 and the
// top window never carries out.
#pragma once
#include "vrf_core.cuh"

namespace vrf {

constexpr int MSM_C = 11;                         // window bits
constexpr int MSM_W = 23;                         // windows: 11*23 = 253 bits
constexpr int MSM_BUCKETS = 1 << (MSM_C - 1);     // 1024 signed buckets per window
constexpr int MSM_BLOCK = 512;                    // lanes per workgroup
constexpr int MSM_PT_WORDS = 4 * NL;              // extended point staged in LDS / HBM: X, Y, Z, T
constexpr int MSM_IDX_BITS = 21;                  // list entry: point index inside its group
constexpr uint32_t MSM_IDX_MASK = (1u << MSM_IDX_BITS) - 1;
constexpr size_t MSM_MAX_PER_GROUP = size_t(1) << MSM_IDX_BITS;

// Device-side layout of one MSM over n points (all regions inside one workspace allocation).
constexpr int MSM_W_SHORT = 12;                   // windows a scalar < 2^128 can reach (11*12 = 132 bits)

struct MsmLayout {
  size_t n;
  // Points [0, n_long) carry full-size scalars, points [n_long, n) scalars < 2^128 whose digits in the
  // windows >= MSM_W_SHORT are zero by construction: those windows only partition [0, n_long), with
  // proportionally fewer groups, so that every workgroup sorts and folds about the same number of points.
  size_t n_long;
  int groups;            // point groups per low window (w < MSM_W_SHORT); also the stride of `part`
  int groups_hi;         // point groups per high window
  size_t per_group;      // points per group in the low windows (<= MSM_MAX_PER_GROUP)
  size_t per_group_hi;   // points per group in the high windows
  size_t list_cap;       // list entries reserved per workgroup (max per-group size + MSM_BLOCK)
  uint32_t* pts;         // [n][PTA_WORDS]   Montgomery affine-cached (x, y, d*x*y)
  int16_t* digits;       // [MSM_W][n]       signed digits in [-1024, 1024]
  uint32_t* lists;       // [MSM_W*groups][list_cap] bucket-sorted entries, lane-transposed
  uint32_t* heads;       // [MSM_W*groups][MSM_BLOCK][MSM_PT_WORDS] first-run partial sums
  uint32_t* part;        // [MSM_W][groups][MSM_PT_WORDS] per-workgroup window sums
  uint8_t* flags;        // [256] flags[0] != 0: some input was invalid
};

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
// buckets + final over prepared points / digits.  out_enc: 32 B compressed sum (nullable); out_xy: 64 B
// affine (nullable); status: 1 byte (nullable; 0 ok, 2 if flags[0] is set); fail_flag: 1 byte (nullable),
// set to 1 unless the sum is the neutral element (never cleared here: callers OR several MSMs into it).
// ev (nullable): ev[0] recorded after the bucket kernel, ev[1] and ev[2] after the final kernel.
void launch_msm_core(int suite, const MsmLayout& L, uint8_t* out_enc, uint8_t* out_xy, uint8_t* status,
                     uint8_t* fail_flag, hipStream_t st, hipEvent_t* ev = nullptr);

// Batched Pedersen verification by random linear combination (k_rlc.hip)
struct RlcArgs {
  int suite;
  int k_lane;                 // proofs per lane in the decode stage
  size_t n;                   // proofs in this launch group
  uint64_t index0;            // index of the first proof in the caller's batch (weights depend on it)
  const uint8_t *h, *gamma, *pk_com, *r, *ok, *s, *sb;   // affine_in: the five point arrays are 64-byte x || y
  int affine_in;
  uint32_t check_mask;        // CHK_INPUT | CHK_OUTPUT | CHK_PROOF: subgroup test of the decoded points
  BytesViewLite ad;
  uint8_t* status;            // [n] 0 = part of the batch sum, 2 = InvalidData (left out of it)
  uint32_t* scratch;          // per-proof scratch, scratch_stride words each (>= 5 * 37)
  int scratch_stride;
  MsmLayout L;                // over 5n + 2 points
  uint64_t* fixed_cols;       // [2][8] limb columns of sum z'_i s_i and sum z'_i sb_i
  DevTables T;
  uint8_t seed[32];
  const uint8_t* root;        // [32] device memory: batch digest of this launch group (digest.cuh)
};
// enqueues decode + MSM; fail_flag[0] becomes 1 if the batch equation does not hold.
// ev (nullable, 5 events): start | decode | buckets | final | final.
void launch_pedersen_rlc(const RlcArgs& a, uint8_t* fail_flag, hipStream_t st, hipEvent_t* ev = nullptr);
// n x 64 B affine x || y -> n x 32 B compressed encodings
void launch_affine_compress(size_t n, const uint8_t* xy, uint8_t* enc, hipStream_t st);

}  // namespace vrf
