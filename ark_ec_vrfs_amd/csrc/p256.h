// p256.h -- launch interface of the secp256r1 kernels (k_p256.hip, compiled with -DVRF_FIELD=3) for the C ABI (api.hip).
// Plain data only, like vrf_types.h: the suite has its own group law (sw.cuh) and wire format (33-byte Sec1 points,
// big-endian scalars, SHA-256), so it has its own workspace layout and argument blocks instead of the Edwards ones.
#pragma once
#include "vrf_types.h"

namespace vrf {
namespace p256 {

// Per-context device workspace for `cap` items.  The inter-stage regions are WORD-MAJOR over the batch -- word w of item i
// sits at base[w * cap + i] -- so that a wave's 64 lanes read 64 consecutive words whatever the per-item record size is;
// the window tables are contiguous per item, because a lookup is indexed by the item's own digit (sw.cuh).
struct Ws {
  uint32_t* tabs;    // [4][cap][224]  window tables (verify: Y, H, Gamma; prove: H, 2^64 H, 2^128 H, 2^192 H)
  uint32_t* pts;     // [4][27][cap]   projective results (verify: U, V; prove: pk, Gamma, U, V)
  uint32_t* aff;     // [5][18][cap]   decoded affine points, Montgomery limbs (IETF verify: pk, H, Gamma; Pedersen verify: H,
                     //                Gamma, pk_com, R, Ok; prove: H)
  uint32_t* sc;      // [4][8][cap]    scalars (verify: c, s [, sb]; prove: sk, k [, b, kb] -- wiped after every prove)
  uint32_t* enc;     // [3][9][cap]    tag + x of the wire encodings that enter the challenge hash (prove: H)
  uint8_t* flags;    // [cap]          1 = the item's inputs decoded
  size_t cap;
};
constexpr size_t WS_TAB_WORDS = 4 * 224, WS_PTS_WORDS = 4 * 27, WS_AFF_WORDS = 5 * 18, WS_SC_WORDS = 4 * 8, WS_ENC_WORDS = 3 * 9;
constexpr size_t WS_WORDS_PER_ITEM = WS_TAB_WORDS + WS_PTS_WORDS + WS_AFF_WORDS + WS_SC_WORDS + WS_ENC_WORDS;
inline size_t ws_bytes(size_t cap) { return cap * WS_WORDS_PER_ITEM * sizeof(uint32_t) + ((cap + 255) & ~size_t(255)); }
inline Ws ws_carve(void* base, size_t cap) {
  Ws w;
  uint32_t* p = static_cast<uint32_t*>(base);
  w.tabs = p; p += cap * WS_TAB_WORDS;
  w.pts = p; p += cap * WS_PTS_WORDS;
  w.aff = p; p += cap * WS_AFF_WORDS;
  w.sc = p; p += cap * WS_SC_WORDS;
  w.enc = p; p += cap * WS_ENC_WORDS;
  w.flags = reinterpret_cast<uint8_t*>(p);
  w.cap = cap;
  return w;
}

struct VerifyArgs {
  size_t n;
  const uint8_t *pk, *h, *gamma;     // n x 33 B Sec1 (affine_in != 0: n x 64 B x || y)
  int affine_in;                     // 0: Sec1; 1: x || y little-endian canonical; 2: x || y arkworks Montgomery limbs
  const uint8_t *c, *s;              // n x 32 B big-endian
  BytesViewLite ad;
  uint8_t* status;
  Ws ws;
  const uint32_t* comb;              // fixed-base comb of the generator
  SuiteStr str;
  // keyed verification (key_index != nullptr): pk is ignored; item i is verified under key key_index[i] of a key set
  const uint32_t* key_index;         // [n]
  size_t n_keys;
  const uint8_t* key_enc;            // [n_keys][33] Sec1 strings (hashed by the challenge)
  const uint32_t* key_aff;           // [n_keys][18] Montgomery affine coordinates
  const uint8_t* key_valid;          // [n_keys]
  const uint32_t* key_combs;         // [n_keys][key_rows][128][28] radix-256 comb rows 0 .. key_rows - 1 of every key
  int key_rows;                      // challenge_len + 1: the rows a challenge can reach
};
struct ProveArgs {
  size_t n;
  const uint8_t* sk;                 // n x 32 B big-endian
  BytesViewLite msg;
  const uint8_t* h_given;            // nullable: n x 33 B
  BytesViewLite ad;
  uint8_t *gamma, *c, *s;            // n x 33, n x 32, n x 32
  uint8_t *pk_out, *h_out, *status;  // nullable: n x 33, n x 33, n
  int out_affine;                    // != 0: gamma, pk_out, r_out, ok_out are n x 64 B x || y (2: arkworks Montgomery limbs)
  // Pedersen (pedersen != 0): c is unused, pk_out receives pk_com, and the outputs below are written
  int pedersen;
  int ct_tables;                     // != 0: window-table lookups by secret digits read all eight entries (VRFHIP_FLAG_CT_TABLES)
  uint8_t *r_out, *ok_out, *sb_out;  // n x 33, n x 33, n x 32
  uint8_t* blinding_out;             // nullable: n x 32
  const uint32_t* comb_b;            // fixed-base comb of the blinding base
  unsigned long long* tai_queue;     // 8-byte device counter of k_p256_tai_find
  Ws ws;
  const uint32_t* comb;
  SuiteStr str;
};

struct PedVerifyArgs {
  size_t n;
  const uint8_t *h, *gamma, *pk_com, *r, *ok;   // n x 33 B Sec1
  const uint8_t *s, *sb;                        // n x 32 B big-endian
  BytesViewLite ad;
  uint8_t* status;
  Ws ws;
  const uint32_t *comb, *comb_b;
  SuiteStr str;
};

// ---- multi-scalar multiplication and the batched Pedersen verifier (k_p256_msm.hip) ----
constexpr int MSM_C = 10;                   // signed window bits: 512 buckets x 108 B = 54 KiB of LDS
constexpr int MSM_W = 26;                   // windows of a folded 256-bit scalar (260 bits)
constexpr int MSM_AFF_STRIDE = 32;          // words per affine point slot (x, y: 18 words) -- one 128-B line per gather
constexpr int MSM_W_SHORT = 13;             // windows a 128-bit weight can reach (130 bits)
struct MsmL {
  size_t n;
  // Points [0, n_long) carry full-size scalars, points [n_long, n) bare 128-bit weights whose digits in the windows >=
  // MSM_W_SHORT are zero: those windows partition [0, n_long) only, into groups_hi <= groups groups of about the same size
  // as a low window's (the grid is MSM_W_SHORT groups + (MSM_W - MSM_W_SHORT) groups_hi workgroups).
  size_t n_long;
  int groups, groups_hi;
  size_t per_group, per_group_hi, list_cap;
  uint32_t* pts;             // [n][MSM_AFF_STRIDE]  Montgomery affine (x, y)
  int16_t* digits;           // [MSM_W][n]           signed digits in [-512, 512]; 0 = the point takes no part
  uint32_t* lists;           // [MSM_W*groups][list_cap] bucket-sorted entries, lane-transposed
  uint32_t* heads;           // [MSM_W*groups][512][27] first-run partial sums
  uint32_t* part;            // [MSM_W][groups][27]  per-workgroup window sums
  uint8_t* flags;            // [256] flags[0] != 0: an input of the plain MSM was invalid
  unsigned long long* cols;  // [2][8] limb columns of sum z'_i s_i and sum z'_i sb_i (batched verifier)
};
int msm_groups(size_t n, size_t n_long, int cus);
size_t msm_workspace_bytes(size_t n, int groups);
MsmL msm_layout(size_t n, size_t n_long, int groups, void* ws);
// `VariableBaseMSM::msm`: bases n x 64 B x || y (little-endian canonical; mont256: arkworks Montgomery limbs; all-zero = the
// point at infinity), scalars n x 32 B big-endian (< n).  out33: Sec1 (0x00 + zeros for the point at infinity), out_xy:
// x || y (all-zero for it); status1[0] = 0 / 2 (a coordinate >= p, a point off the curve, a scalar >= n: outputs zeroed).
void launch_msm(size_t n, const uint8_t* xy, int mont256, const uint8_t* scalars_be, uint8_t* out33, uint8_t* out_xy, uint8_t* status1,
                void* ws, int groups, hipStream_t st);
struct RlcArgs {
  size_t n;                                     // proofs in this launch group
  unsigned long long index0;                    // index of the first one in the caller's batch (the weights depend on it)
  const uint8_t *h, *gamma, *pk_com, *r, *ok;   // n x 33 B Sec1; affine_in: n x 64 B x || y (little-endian)
  int affine_in;                                // 0: Sec1 strings, 1: x || y canonical, 2: x || y Montgomery-256
  const uint8_t *s, *sb;                        // n x 32 B big-endian
  BytesViewLite ad;
  uint8_t* status;                              // [n] 0 = part of the batch sum, 2 = InvalidData (left out of it)
  MsmL L;                                       // over 5n + 2 points
  unsigned long long* fixed_cols;               // = L.cols (set by the launcher)
  uint8_t seed[32];
  const uint8_t* root;                          // [32] device memory: batch digest of this launch group (digest.cuh)
  uint8_t gen_xy[64], b_xy[64];                 // the descriptor's generator and blinding base
  SuiteStr str;
};
// fail_flag[0] = 1 unless sum_i z_i (s_i H_i - c_i Gamma_i - Ok_i) + z'_i (s_i G + sb_i B - c_i pk_com_i - R_i) = O over the
// decodable proofs.  ev: nullptr or 5 events (start, after decode, after buckets, after final, end).
void launch_pedersen_rlc(const RlcArgs& a, uint8_t* fail_flag, hipStream_t st, hipEvent_t* ev);
// n x 64 B x || y (little-endian; mont256: arkworks Montgomery limbs) -> n x 33 B Sec1 strings (a failed x || y batch falls back
// to the per-proof kernels); a coordinate >= p gives an undecodable string (tag 0xff)
void launch_affine_compress(size_t n, const uint8_t* xy, int mont256, uint8_t* enc33, hipStream_t st);

size_t comb_bytes();
// `Public` keys with resident combs (vrfhip_keyset_create on this suite): rows = challenge_len + 1 rows of 128 entries each
size_t key_comb_bytes(int rows);
void launch_keyset_build(size_t n_keys, const uint8_t* pks33, uint32_t* aff, uint8_t* valid, uint32_t* combs, int rows, hipStream_t st);
// comb of the point gen_xy (x || y, 32-byte little-endian canonical integers, as vrfhip_suite_desc carries it);
// ok[0] = 1 if it is a point of the curve other than the point at infinity
void launch_init_comb(uint32_t* comb, const uint8_t* d_gen_xy, uint8_t* d_ok, hipStream_t st);
// the built-in generator (SEC 2 2.4.2) and Pedersen blinding base (a nothing-up-my-sleeve point: tools/gen_constants.py)
// as the descriptor carries them
void default_generator(uint8_t xy[64]);
void default_blinding_base(uint8_t xy[64]);
void launch_verify(const VerifyArgs& a, hipStream_t st, hipEvent_t* ev);    // ev: nullptr or 5 events (stage boundaries)
void launch_prove(const ProveArgs& a, hipStream_t st, hipEvent_t* ev);      // IETF, or Pedersen when a.pedersen
void launch_pedersen_verify(const PedVerifyArgs& a, hipStream_t st, hipEvent_t* ev);
void launch_hash_to_curve(size_t n, BytesViewLite msg, uint8_t* points33, const SuiteStr& str, hipStream_t st,
                          uint8_t* tai_ctr = nullptr, unsigned long long* queue = nullptr);
void launch_output_hash(size_t n, const uint8_t* gamma33, uint8_t* hash32, const SuiteStr& str, hipStream_t st);
void launch_secret_from_seed(size_t n, const uint8_t* seeds, uint32_t seed_len, uint8_t* sk32, uint8_t* pk33,
                             const uint32_t* comb, hipStream_t st);
void launch_point_validate(size_t n, const uint8_t* points33, uint8_t* xy_out, int xy_mont256, uint8_t* status, hipStream_t st);

}  // namespace p256
}  // namespace vrf
