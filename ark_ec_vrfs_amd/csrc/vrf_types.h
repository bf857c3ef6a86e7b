// vrf_types.h -- plain-data launch arguments shared by the C ABI (api.hip) and the kernel translation units of EVERY
// base field (field.h).  No arithmetic here: only pointers, sizes and byte strings, so the per-field builds of the
// kernels and the one build of the C ABI agree on these types by construction.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

#define VRF_HD __host__ __device__ __forceinline__

namespace vrf {

// Byte strings of the suite descriptor (include/vrfhip.h vrfhip_suite_desc), packed by the host into big-endian
// 64-bit words -- the unit SHA-512 absorbs -- and carried BY VALUE inside the kernel arguments: uniform reads from
// the kernarg segment are scalar loads, where a pointer into device memory cost every lane a vector load per byte.
struct SuiteStr {
  uint32_t suite_id_len;       // bytes, <= 64
  uint32_t dst_prime_len;      // bytes of DST' = DST || byte(len(DST)), <= 129 (Elligator suites; 0 otherwise)
  uint64_t suite_id_w[8];      // `Suite::SUITE_ID`, zero padded
  uint64_t dst_prime_w[17];    // RFC 9380 DST' (upstream DST: "ECVRF_" || h2c suite id || SUITE_ID)
  uint32_t challenge_len;      // `Suite::CHALLENGE_LEN`: bytes of the challenge hash that make c (1..32)
  uint32_t flags;              // SS_* below (vrfhip_suite_desc.flags)
};
// What a `Suite` impl may override besides its constants (`Suite::Codec`, `Suite::challenge`, `Suite::point_to_hash`):
// the three deviations that separate upstream's Ed25519 suite from RFC 9381's ECVRF-EDWARDS25519-SHA512-TAI, whose
// published vectors then run through the same kernels (tests/golden/rfc9381_edwards25519_sha512_tai.json).
enum : uint32_t {
  SS_SIGN_PARITY = 1,          // compressed points carry x mod 2 in bit 255 (RFC 8032) instead of arkworks' x > q - x
  SS_CHALLENGE_LE = 2,         // the truncated challenge hash is a little-endian integer (RFC 9381 edwards suites)
  SS_HASH_COFACTOR = 4         // Output::hash hashes cofactor * Gamma (RFC 9381 proof_to_hash)
};

struct SqrtTables {
  const uint32_t* P;     // [SQRT_TABLES][256][9]
  const uint8_t* lut;    // [1 << SQRT_LUT_BITS]
  // The suite's byte strings travel with these tables because the same helpers (hash-to-curve, point decoding)
  // receive them.
  SuiteStr str;
};

// variable-length byte strings: shared blob, per-item offsets, or fixed stride
struct BytesViewLite {
  const uint8_t* blob;
  const uint32_t* off;   // n+1 offsets, or nullptr
  uint32_t len;          // off == nullptr: length of every item
  uint32_t stride;       // off == nullptr: distance between items (0 = all items share blob)
};
VRF_HD void bytes_lite_get(const BytesViewLite& v, size_t i, const uint8_t*& p, uint32_t& n) {
  if (v.off) {
    uint32_t a = v.off[i], b = v.off[i + 1];
    p = v.blob + a;
    n = b - a;
  } else {
    p = v.blob + i * (size_t)v.stride;
    n = v.len;
  }
}
using BytesView = BytesViewLite;
VRF_HD void bytes_get(const BytesView& v, size_t i, const uint8_t*& p, uint32_t& n) { bytes_lite_get(v, i, p, n); }

// Shared, read-only device tables (built once per context, see k_misc.hip: k_init_tables)
struct DevTables {
  SqrtTables sq;
  const uint32_t* g_win;     // [2][8][PTC_WORDS]    j*G and j*psi(G), j = 1..8, cached form
  const uint32_t* g_comb;    // [GC_ROWS][GC_COLS][PTA_WORDS] j*2^(GCB*w)*G affine, signed windows (gcomb_*)
  const uint32_t* b_comb;    // same for the Pedersen blinding base
};

// per-context device workspace (capacity `cap` items)
struct Workspace {
  uint32_t* tabs;    // [cap][WS_TABS][WIN_TABLE_WORDS]  window tables (verify: 6 GLV tables per proof)
  uint32_t* pts;     // [cap][PROVE_PTS_WORDS]      projective intermediates (verify uses 2*UV_WORDS)
  uint32_t* aux;     // [cap][32]   prove: enc(H) | k | blinding b | kb ; Pedersen verify: challenge c
  uint8_t* flags;    // [cap]                       validity of decoded inputs
};

struct VerifyArgs {
  int suite;
  int k_lane;
  size_t n;
  const uint8_t *pk, *h, *gamma, *c, *s;   // affine_in != 0: pk, h, gamma are 64-byte x || y
  int affine_in;
  int h_in_tabs;                           // verification from alpha: h = the encodings hash-to-curve wrote, H's tables are built
  uint32_t check_mask;                     // CHK_* bits: which decoded points get the subgroup test
  BytesView ad;
  uint8_t* status;
  // keyed verification (key_index != nullptr): pk is the key set's encodings [n_keys][32]; the U half uses the
  // key's context-resident comb instead of per-proof tables
  const uint32_t* key_index;    // [n] index of each proof's key
  const uint32_t* key_combs;    // [n_keys][32][255][PTA_WORDS]
  const uint8_t* key_valid;     // [n_keys] 1 = decodes to a point of the prime-order subgroup
  size_t n_keys;
  Workspace ws;
  DevTables T;
};

struct ProveArgs {
  int suite;
  int k_lane;
  size_t n;
  const uint8_t* sk;
  BytesView msg;
  const uint8_t* h_given;     // nullable
  BytesView ad;
  uint8_t *gamma, *c, *s, *pk_out, *h_out, *status;
  // Pedersen (pedersen != 0): c is unused; pk_out receives pk_com; extra outputs below
  int pedersen;
  uint32_t check_mask;        // CHK_INPUT: subgroup test of a given H
  int out_affine;             // != 0: gamma, pk_out, r_out, ok_out are n x 64 B (x || y, canonical little-endian)
  uint8_t *r_out, *ok_out, *sb_out, *blinding_out;
  unsigned long long* tai_queue;   // 8-byte device counter for k_tai_find (try-and-increment suites)
  Workspace ws;
  DevTables T;
};

struct PedersenVerifyArgs {
  int suite;
  size_t n;
  const uint8_t *h, *gamma, *pk_com, *r, *ok, *s, *sb;
  uint32_t check_mask;        // CHK_INPUT | CHK_OUTPUT | CHK_PROOF
  BytesView ad;
  uint8_t* status;
  Workspace ws;
  DevTables T;
};

// Device-side layout of one MSM over n points (all regions inside one workspace allocation; msm.cuh).
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
  uint32_t sflags;       // SuiteStr::flags of the context (sign convention of the compressed result)
};

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

// suite ids follow vrfhip_suite (include/vrfhip.h); the field a suite lives in follows field.h
constexpr int SUITE_BS = 1, SUITE_JJ = 2, SUITE_ED = 3, SUITE_BJ = 4;
constexpr int suite_field(int suite) { return suite == SUITE_ED ? 1 : suite == SUITE_BJ ? 2 : 0; }

}  // namespace vrf
