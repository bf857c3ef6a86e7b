// sha512.cuh -- SHA-512 for the VRF transcript hashes (nonce, challenge, XMD, output hash).
// Stands in for the `sha2` crate used behind `Suite::Hasher` (/root/reference src/lib.rs:16).
//
// The message block lives in sixteen 64-bit registers.  Every append goes through w_or(),
// a 16-way predicated select on the word index: when the position is a compile-time
// constant (all fixed-layout prefixes) it folds to one OR; when it is only known at run
// time (after a variable-length `ad`) it stays in registers instead of spilling to scratch.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#ifndef VRF_HD
#define VRF_HD __host__ __device__ __forceinline__
#endif

namespace vrf {

struct Sha512K {
  static VRF_HD uint64_t at(int i) {
    constexpr uint64_t K[80] = {
        0x428a2f98d728ae22ULL, 0x7137449123ef65cdULL, 0xb5c0fbcfec4d3b2fULL, 0xe9b5dba58189dbbcULL,
        0x3956c25bf348b538ULL, 0x59f111f1b605d019ULL, 0x923f82a4af194f9bULL, 0xab1c5ed5da6d8118ULL,
        0xd807aa98a3030242ULL, 0x12835b0145706fbeULL, 0x243185be4ee4b28cULL, 0x550c7dc3d5ffb4e2ULL,
        0x72be5d74f27b896fULL, 0x80deb1fe3b1696b1ULL, 0x9bdc06a725c71235ULL, 0xc19bf174cf692694ULL,
        0xe49b69c19ef14ad2ULL, 0xefbe4786384f25e3ULL, 0x0fc19dc68b8cd5b5ULL, 0x240ca1cc77ac9c65ULL,
        0x2de92c6f592b0275ULL, 0x4a7484aa6ea6e483ULL, 0x5cb0a9dcbd41fbd4ULL, 0x76f988da831153b5ULL,
        0x983e5152ee66dfabULL, 0xa831c66d2db43210ULL, 0xb00327c898fb213fULL, 0xbf597fc7beef0ee4ULL,
        0xc6e00bf33da88fc2ULL, 0xd5a79147930aa725ULL, 0x06ca6351e003826fULL, 0x142929670a0e6e70ULL,
        0x27b70a8546d22ffcULL, 0x2e1b21385c26c926ULL, 0x4d2c6dfc5ac42aedULL, 0x53380d139d95b3dfULL,
        0x650a73548baf63deULL, 0x766a0abb3c77b2a8ULL, 0x81c2c92e47edaee6ULL, 0x92722c851482353bULL,
        0xa2bfe8a14cf10364ULL, 0xa81a664bbc423001ULL, 0xc24b8b70d0f89791ULL, 0xc76c51a30654be30ULL,
        0xd192e819d6ef5218ULL, 0xd69906245565a910ULL, 0xf40e35855771202aULL, 0x106aa07032bbd1b8ULL,
        0x19a4c116b8d2d0c8ULL, 0x1e376c085141ab53ULL, 0x2748774cdf8eeb99ULL, 0x34b0bcb5e19b48a8ULL,
        0x391c0cb3c5c95a63ULL, 0x4ed8aa4ae3418acbULL, 0x5b9cca4f7763e373ULL, 0x682e6ff3d6b2b8a3ULL,
        0x748f82ee5defb2fcULL, 0x78a5636f43172f60ULL, 0x84c87814a1f0ab72ULL, 0x8cc702081a6439ecULL,
        0x90befffa23631e28ULL, 0xa4506cebde82bde9ULL, 0xbef9a3f7b2c67915ULL, 0xc67178f2e372532bULL,
        0xca273eceea26619cULL, 0xd186b8c721c0c207ULL, 0xeada7dd6cde0eb1eULL, 0xf57d4f7fee6ed178ULL,
        0x06f067aa72176fbaULL, 0x0a637dc5a2c898a6ULL, 0x113f9804bef90daeULL, 0x1b710b35131c471bULL,
        0x28db77f523047d84ULL, 0x32caab7b40c72493ULL, 0x3c9ebe0a15c9bebcULL, 0x431d67c49c100d4cULL,
        0x4cc5d4becb3e42b6ULL, 0x597f299cfc657e2aULL, 0x5fcb6fab3ad6faecULL, 0x6c44198c4a475817ULL};
    return K[i];
  }
};

VRF_HD uint64_t rotr64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
VRF_HD uint32_t bswap32(uint32_t x) {
  return (x >> 24) | ((x >> 8) & 0xff00u) | ((x << 8) & 0xff0000u) | (x << 24);
}

struct Sha512 {
  uint64_t h[8];
  uint64_t w[16];
  uint32_t pos;      // bytes in the current block (0..127)
  uint32_t total;    // total bytes absorbed (messages here are < 4 GiB)
};

VRF_HD void sha512_init(Sha512& s) {
  s.h[0] = 0x6a09e667f3bcc908ULL; s.h[1] = 0xbb67ae8584caa73bULL;
  s.h[2] = 0x3c6ef372fe94f82bULL; s.h[3] = 0xa54ff53a5f1d36f1ULL;
  s.h[4] = 0x510e527fade682d1ULL; s.h[5] = 0x9b05688c2b3e6c1fULL;
  s.h[6] = 0x1f83d9abfb41bd6bULL; s.h[7] = 0x5be0cd19137e2179ULL;
#pragma unroll
  for (int i = 0; i < 16; ++i) s.w[i] = 0;
  s.pos = 0;
  s.total = 0;
}

VRF_HD void sha512_compress(Sha512& s) {
  uint64_t a = s.h[0], b = s.h[1], c = s.h[2], d = s.h[3];
  uint64_t e = s.h[4], f = s.h[5], g = s.h[6], hh = s.h[7];
  uint64_t w[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) w[i] = s.w[i];
#pragma unroll
  for (int r = 0; r < 80; ++r) {
    if (r >= 16) {
      uint64_t w15 = w[(r + 1) & 15], w2 = w[(r + 14) & 15];
      uint64_t s0 = rotr64(w15, 1) ^ rotr64(w15, 8) ^ (w15 >> 7);
      uint64_t s1 = rotr64(w2, 19) ^ rotr64(w2, 61) ^ (w2 >> 6);
      w[r & 15] = w[r & 15] + s0 + w[(r + 9) & 15] + s1;
    }
    uint64_t S1 = rotr64(e, 14) ^ rotr64(e, 18) ^ rotr64(e, 41);
    uint64_t ch = (e & f) ^ (~e & g);
    uint64_t t1 = hh + S1 + ch + Sha512K::at(r) + w[r & 15];
    uint64_t S0 = rotr64(a, 28) ^ rotr64(a, 34) ^ rotr64(a, 39);
    uint64_t mj = (a & b) ^ (a & c) ^ (b & c);
    uint64_t t2 = S0 + mj;
    hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  s.h[0] += a; s.h[1] += b; s.h[2] += c; s.h[3] += d;
  s.h[4] += e; s.h[5] += f; s.h[6] += g; s.h[7] += hh;
#pragma unroll
  for (int i = 0; i < 16; ++i) s.w[i] = 0;
}

VRF_HD void w_or(uint64_t (&w)[16], uint32_t idx, uint64_t v) {
#pragma unroll
  for (int i = 0; i < 16; ++i)
    if (idx == (uint32_t)i) w[i] |= v;
}

// append the n (1..8) most-significant bytes of v
VRF_HD void sha512_put(Sha512& s, uint64_t v, uint32_t n) {
  if (n < 8) v &= ~0ULL << (8 * (8 - n));
  const uint32_t wi = s.pos >> 3, bo = s.pos & 7;
  w_or(s.w, wi, v >> (8 * bo));
  const uint32_t first = 8 - bo;               // bytes that fit in word wi
  uint32_t np = s.pos + n;
  if (n > first) {
    // spill-over into the next word (or next block)
    uint64_t rest = v << (8 * first);
    if (wi == 15) {
      sha512_compress(s);
      s.w[0] = rest;
    } else {
      w_or(s.w, wi + 1, rest);
    }
  } else if (np == 128) {
    sha512_compress(s);
  }
  s.pos = np & 127;
  s.total += n;
}

VRF_HD void sha512_put_byte(Sha512& s, uint8_t b) { sha512_put(s, (uint64_t)b << 56, 1); }

// 32 bytes given as 8 little-endian u32 words (byte 0 = LSB of w[0])
VRF_HD void sha512_put_le32x8(Sha512& s, const uint32_t w[8]) {
#pragma unroll
  for (int j = 0; j < 4; ++j)
    sha512_put(s, ((uint64_t)bswap32(w[2 * j]) << 32) | bswap32(w[2 * j + 1]), 8);
}

// N 64-bit big-endian words held in registers, appended at a position known only at run time (after a
// descriptor-supplied suite string): ONE put site in a rolled loop, the word picked by a select chain, so the
// block-boundary compression is instantiated once instead of once per word.
template <int N>
VRF_HD void sha512_put_words(Sha512& s, const uint64_t (&w)[N]) {
#pragma unroll 1
  for (int i = 0; i < N; ++i) {
    uint64_t v = w[0];
#pragma unroll
    for (int j = 1; j < N; ++j)
      if (i == j) v = w[j];
    sha512_put(s, v, 8);
  }
}
// 32 bytes given as 8 little-endian u32 words -> the 4 big-endian message words
VRF_HD void sha512_words_le32x8(uint64_t out[4], const uint32_t w[8]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) out[j] = ((uint64_t)bswap32(w[2 * j]) << 32) | bswap32(w[2 * j + 1]);
}

// n bytes packed big-endian into 64-bit words (zero padded): whole words, then the tail
VRF_HD void sha512_put_packed(Sha512& s, const uint64_t* w, uint32_t n) {
  uint32_t i = 0;
#pragma unroll 1
  for (; i + 8 <= n; i += 8) sha512_put(s, w[i >> 3], 8);
  if (i < n) sha512_put(s, w[i >> 3], n - i);
}

// raw bytes from memory (host or device pointer valid in the calling context)
VRF_HD void sha512_put_bytes(Sha512& s, const uint8_t* p, uint32_t n) {
  uint32_t i = 0;
  for (; i + 8 <= n; i += 8) {
    uint64_t v = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) v = (v << 8) | p[i + k];
    sha512_put(s, v, 8);
  }
  for (; i < n; ++i) sha512_put_byte(s, p[i]);
}

VRF_HD void sha512_final(Sha512& s) {
  const uint32_t total = s.total;
  sha512_put_byte(s, 0x80);
  if (s.pos > 112) {            // no room for the 16-byte length
    sha512_compress(s);
    s.pos = 0;
  }
  s.w[15] |= (uint64_t)total * 8;
  sha512_compress(s);
}

// digest as 16 big-endian 32-bit chunks: chunk j = bytes 4j..4j+3 of the digest
VRF_HD uint32_t sha512_chunk_be(const Sha512& s, int j) {
  return (j & 1) ? (uint32_t)s.h[j >> 1] : (uint32_t)(s.h[j >> 1] >> 32);
}
// first `32` digest bytes read as a big-endian integer -> 8 LE u32 words
VRF_HD void sha512_be256(uint32_t out[8], const Sha512& s) {
#pragma unroll
  for (int j = 0; j < 8; ++j) out[j] = sha512_chunk_be(s, 7 - j);
}
// all 64 digest bytes read as a big-endian integer -> 16 LE u32 words
VRF_HD void sha512_be512(uint32_t out[16], const Sha512& s) {
#pragma unroll
  for (int j = 0; j < 16; ++j) out[j] = sha512_chunk_be(s, 15 - j);
}
// all 64 digest bytes read as a little-endian integer -> 16 LE u32 words
VRF_HD void sha512_le512(uint32_t out[16], const Sha512& s) {
#pragma unroll
  for (int j = 0; j < 16; ++j) out[j] = bswap32(sha512_chunk_be(s, j));
}
// digest bytes j*4.. as they sit in memory (for writing the 64-byte output hash)
VRF_HD uint32_t sha512_word_mem(const Sha512& s, int j) { return bswap32(sha512_chunk_be(s, j)); }

}  // namespace vrf
