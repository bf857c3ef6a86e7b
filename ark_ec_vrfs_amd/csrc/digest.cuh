// digest.cuh -- a digest of a whole batch, hashed into the weights of the random-linear-combination checks
// (k_rlc.hip: batched Pedersen verification; k_msm_g1.hip: batched pairing check).
//
// Why: weights derived from (seed, index) alone are only as good as the caller's seed -- a reused or predictable
// seed lets a prover pick proofs after the weights are known.  With the digest in the derivation the weights of a
// launch group are fixed only once every byte of the group is (Fiat-Shamir style): a prover who knows the seed
// still has to find, by trial, batch contents whose own weights cancel its defect, 2^-125 per trial.
//
// Definition (normative; oracle/c/oracle_vrf.c restates it):
//   leaf_i = SHA-512("vrfhip-leaf-v1" || u64_le(index0 + i) || a_0[i] || ... || a_{k-1}[i] || ad_i || u32_le(|ad_i|))[0..32]
//            (a_j[i]: the item's w_j bytes of input array j, in the order of the entry point's arguments)
//   node   = SHA-512("vrfhip-node-v1" || u32_le(count) || child_0 || ... || child_{count-1})[0..32], count <= 16
//   level 0 = the leaves; every further level hashes runs of 16 consecutive children; the root is the single node
//   of the last level (a batch of <= 16 items has one level of nodes).
#pragma once
#include "vrf_core.cuh"

namespace vrf {

constexpr int DIGEST_FAN = 16;        // children per node: a node is 5 SHA-512 blocks, so a level's latency stays ~50 us
constexpr int DIGEST_MAX_ARRAYS = 8;

struct DigestSrc {
  const uint8_t* p[DIGEST_MAX_ARRAYS];
  uint32_t w[DIGEST_MAX_ARRAYS];       // bytes per item (a multiple of 4 needs a 4-byte aligned array; other widths are read bytewise)
  int n_arr;
  BytesViewLite ad;                    // blob == nullptr: no per-item string
};

// n 32-bit words of memory, in memory (byte) order; p 4-byte aligned
VRF_HD void sha512_put_mem32(Sha512& h, const uint32_t* p, uint32_t n) {
  uint32_t i = 0;
  for (; i + 2 <= n; i += 2) sha512_put(h, ((uint64_t)bswap32(p[i]) << 32) | bswap32(p[i + 1]), 8);
  if (i < n) sha512_put(h, (uint64_t)bswap32(p[i]) << 32, 4);
}

VRF_HD void digest_out32(uint8_t* out, const Sha512& h) {
  uint32_t* o = reinterpret_cast<uint32_t*>(out);
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = sha512_word_mem(h, j);
}

VRF_HD void digest_leaf(uint8_t* out, const DigestSrc& s, size_t i, uint64_t index) {
  Sha512 h;
  sha512_init(h);
  constexpr char tag[] = "vrfhip-leaf-v1";
#pragma unroll
  for (int j = 0; j < 14; ++j) sha512_put_byte(h, (uint8_t)tag[j]);
#pragma unroll
  for (int j = 0; j < 8; ++j) sha512_put_byte(h, (uint8_t)(index >> (8 * j)));
#pragma unroll 1
  for (int a = 0; a < s.n_arr; ++a) {
    // 4-byte multiples (the Edwards suites' 32 / 64-byte fields) go in as words, anything else (secp256r1's 33-byte Sec1
    // points) byte by byte: the same byte string either way
    if ((s.w[a] & 3u) == 0) sha512_put_mem32(h, reinterpret_cast<const uint32_t*>(s.p[a] + i * (size_t)s.w[a]), s.w[a] / 4);
    else sha512_put_bytes(h, s.p[a] + i * (size_t)s.w[a], s.w[a]);
  }
  uint32_t ad_len = 0;
  if (s.ad.blob) {
    const uint8_t* ad;
    bytes_lite_get(s.ad, i, ad, ad_len);
    sha512_put_bytes(h, ad, ad_len);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) sha512_put_byte(h, (uint8_t)(ad_len >> (8 * j)));
  sha512_final(h);
  digest_out32(out, h);
}

VRF_HD void digest_node(uint8_t* out, const uint8_t* children, uint32_t count) {
  Sha512 h;
  sha512_init(h);
  constexpr char tag[] = "vrfhip-node-v1";
#pragma unroll
  for (int j = 0; j < 14; ++j) sha512_put_byte(h, (uint8_t)tag[j]);
#pragma unroll
  for (int j = 0; j < 4; ++j) sha512_put_byte(h, (uint8_t)(count >> (8 * j)));
  sha512_put_mem32(h, reinterpret_cast<const uint32_t*>(children), count * 8);
  sha512_final(h);
  digest_out32(out, h);
}

// host side (k_digest.hip).  ws: digest_ws_bytes(n) bytes, 16-byte aligned; root: 32 bytes of device memory.
size_t digest_ws_bytes(size_t n);
void launch_batch_digest(const DigestSrc& src, size_t n, uint64_t index0, uint8_t* ws, uint8_t* root, hipStream_t st);

}  // namespace vrf
