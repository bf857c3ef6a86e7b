// sha256.cuh -- SHA-256 and HMAC-SHA-256 for the secp256r1 suite's transcript hashes (try-and-increment, RFC 6979 nonce,
// challenge, output hash).  Stands in for `sha2::Sha256` behind `Suite::Hasher` (/root/reference src/lib.rs:16) in the
// suite upstream calls `suites::secp256r1` (src/lib.rs:14).
//
// Same shape as sha512.cuh: the message block lives in sixteen 32-bit registers and every append goes through a 16-way
// predicated select on the word index, so a position known only at run time does not push the block into scratch.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#ifndef VRF_HD
#define VRF_HD __host__ __device__ __forceinline__
#endif

namespace vrf {

struct Sha256K {
  static VRF_HD uint32_t at(int i) {
    constexpr uint32_t K[64] = {
        0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u,
        0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u,
        0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau,
        0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u,
        0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u,
        0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u,
        0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u,
        0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
    return K[i];
  }
};

VRF_HD uint32_t rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

struct Sha256 {
  uint32_t h[8];
  uint32_t w[16];
  uint32_t pos;      // bytes in the current block (0..63)
  uint32_t total;    // total bytes absorbed
};

VRF_HD void sha256_init(Sha256& s) {
  s.h[0] = 0x6a09e667u; s.h[1] = 0xbb67ae85u; s.h[2] = 0x3c6ef372u; s.h[3] = 0xa54ff53au;
  s.h[4] = 0x510e527fu; s.h[5] = 0x9b05688cu; s.h[6] = 0x1f83d9abu; s.h[7] = 0x5be0cd19u;
#pragma unroll
  for (int i = 0; i < 16; ++i) s.w[i] = 0;
  s.pos = 0;
  s.total = 0;
}

// NOT inlined: the RFC 6979 nonce alone runs five HMACs, some thirty append sites each of which may close a block, and
// thirty inlined copies of the 64 rounds took the compiler more than half an hour per kernel.  One shared copy costs a
// call and keeps the 24-word state in memory across it -- nothing against the ladders these kernels wait for.
__host__ __device__ __attribute__((noinline)) inline void sha256_compress(Sha256& s) {
  uint32_t a = s.h[0], b = s.h[1], c = s.h[2], d = s.h[3];
  uint32_t e = s.h[4], f = s.h[5], g = s.h[6], hh = s.h[7];
  uint32_t w[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) w[i] = s.w[i];
#pragma unroll
  for (int r = 0; r < 64; ++r) {
    if (r >= 16) {
      const uint32_t w15 = w[(r + 1) & 15], w2 = w[(r + 14) & 15];
      const uint32_t s0 = rotr32(w15, 7) ^ rotr32(w15, 18) ^ (w15 >> 3);
      const uint32_t s1 = rotr32(w2, 17) ^ rotr32(w2, 19) ^ (w2 >> 10);
      w[r & 15] = w[r & 15] + s0 + w[(r + 9) & 15] + s1;
    }
    const uint32_t S1 = rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25);
    const uint32_t ch = (e & f) ^ (~e & g);
    const uint32_t t1 = hh + S1 + ch + Sha256K::at(r) + w[r & 15];
    const uint32_t S0 = rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22);
    const uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
    const uint32_t t2 = S0 + mj;
    hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  s.h[0] += a; s.h[1] += b; s.h[2] += c; s.h[3] += d;
  s.h[4] += e; s.h[5] += f; s.h[6] += g; s.h[7] += hh;
#pragma unroll
  for (int i = 0; i < 16; ++i) s.w[i] = 0;
}

VRF_HD void w32_or(uint32_t (&w)[16], uint32_t idx, uint32_t v) {
#pragma unroll
  for (int i = 0; i < 16; ++i)
    if (idx == (uint32_t)i) w[i] |= v;
}

// append the n (1..4) most-significant bytes of v
VRF_HD void sha256_put(Sha256& s, uint32_t v, uint32_t n) {
  if (n < 4) v &= ~0u << (8 * (4 - n));
  const uint32_t wi = s.pos >> 2, bo = s.pos & 3;
  w32_or(s.w, wi, v >> (8 * bo));
  const uint32_t first = 4 - bo;               // bytes that fit in word wi
  const uint32_t np = s.pos + n;
  if (n > first) {
    const uint32_t rest = v << (8 * first);
    if (wi == 15) {
      sha256_compress(s);
      s.w[0] = rest;
    } else {
      w32_or(s.w, wi + 1, rest);
    }
  } else if (np == 64) {
    sha256_compress(s);
  }
  s.pos = np & 63;
  s.total += n;
}

VRF_HD void sha256_put_byte(Sha256& s, uint8_t b) { sha256_put(s, (uint32_t)b << 24, 1); }

// a 256-bit integer given as 8 little-endian u32 words, appended as 32 big-endian bytes (I2OSP): ONE put site
VRF_HD void sha256_put_be256(Sha256& s, const uint32_t w[8]) {
#pragma unroll 1
  for (int j = 7; j >= 0; --j) {
    uint32_t v = w[0];
#pragma unroll
    for (int k = 1; k < 8; ++k)
      if (j == k) v = w[k];
    sha256_put(s, v, 4);
  }
}

// n bytes packed big-endian into 64-bit words (the SuiteStr packing of vrf_types.h)
VRF_HD void sha256_put_packed64(Sha256& s, const uint64_t* w, uint32_t n) {
  uint32_t i = 0;
#pragma unroll 1
  for (; i + 4 <= n; i += 4) sha256_put(s, (uint32_t)(w[i >> 3] >> (32 - 8 * (i & 4))), 4);
  if (i < n) sha256_put(s, (uint32_t)(w[i >> 3] >> (32 - 8 * (i & 4))), n - i);
}

// raw bytes from memory
VRF_HD void sha256_put_bytes(Sha256& s, const uint8_t* p, uint32_t n) {
  uint32_t i = 0;
  for (; i + 4 <= n; i += 4)
    sha256_put(s, ((uint32_t)p[i] << 24) | ((uint32_t)p[i + 1] << 16) | ((uint32_t)p[i + 2] << 8) | p[i + 3], 4);
  for (; i < n; ++i) sha256_put_byte(s, p[i]);
}

VRF_HD void sha256_final(Sha256& s) {
  const uint32_t total = s.total;
  sha256_put_byte(s, 0x80);
  if (s.pos > 56) {             // no room for the 8-byte length
    sha256_compress(s);
    s.pos = 0;
  }
  s.w[15] |= total * 8;         // messages here are far below 2^29 bytes
  sha256_compress(s);
}

// the digest read as a big-endian 256-bit integer -> 8 little-endian u32 words
VRF_HD void sha256_be256(uint32_t out[8], const Sha256& s) {
#pragma unroll
  for (int j = 0; j < 8; ++j) out[j] = s.h[7 - j];
}

// HMAC-SHA-256 with a 32-byte key: key and the 32-byte values are 256-bit big-endian integers held as 8 LE words
// (word 7 = the first four bytes).  Two-call form so that the message can be assembled from pieces:
//   hmac256_begin(s, key); sha256_put...(s, ...); hmac256_end(out, s, key);
VRF_HD void hmac256_pad(Sha256& s, const uint32_t key[8], uint32_t pad) {
  sha256_init(s);
#pragma unroll
  for (int j = 0; j < 8; ++j) s.w[j] = key[7 - j] ^ pad;
#pragma unroll
  for (int j = 8; j < 16; ++j) s.w[j] = pad;
  s.total = 64;
  sha256_compress(s);
}
VRF_HD void hmac256_begin(Sha256& s, const uint32_t key[8]) { hmac256_pad(s, key, 0x36363636u); }
VRF_HD void hmac256_end(uint32_t out[8], Sha256& s, const uint32_t key[8]) {
  sha256_final(s);
  uint32_t inner[8];
  sha256_be256(inner, s);
  hmac256_pad(s, key, 0x5c5c5c5cu);
  sha256_put_be256(s, inner);
  sha256_final(s);
  sha256_be256(out, s);
}

}  // namespace vrf
