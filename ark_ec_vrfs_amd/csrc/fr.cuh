// fr.cuh -- scalar-field (mod r) helpers: a handful of operations per proof, so a plain
// 8 x u32 CIOS Montgomery multiplier (R = 2^256) with the modulus taken from the curve tag.
// Stands in for ark_ff arithmetic on `ScalarField` (/root/reference src/lib.rs:15) where the
// VRF schemes reduce hashes mod r and form s = k + c*sk.
#pragma once
#include "fe.cuh"

VRF_NS_BEGIN

// a * b / 2^256 mod r, a < 2^256, b < r; result in [0, r)
template <class C>
VRF_HD void fr_montmul(uint32_t out[8], const uint32_t a[8], const uint32_t b[8]) {
  uint32_t t[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      uint64_t x = (uint64_t)a[j] * b[i] + t[j] + c;
      t[j] = (uint32_t)x;
      c = x >> 32;
    }
    uint64_t x = (uint64_t)t[8] + c;
    t[8] = (uint32_t)x;
    t[9] = (uint32_t)(x >> 32);
    uint32_t m = t[0] * C::R_NINV32;
    c = ((uint64_t)m * C::r32(0) + t[0]) >> 32;
#pragma unroll
    for (int j = 1; j < 8; ++j) {
      uint64_t y = (uint64_t)m * C::r32(j) + t[j] + c;
      t[j - 1] = (uint32_t)y;
      c = y >> 32;
    }
    x = (uint64_t)t[8] + c;
    t[7] = (uint32_t)x;
    t[8] = t[9] + (uint32_t)(x >> 32);
  }
  // t < 2r: one conditional subtraction (t[8] is the 257th bit, set only when r > 2^255: secp256r1's order)
  uint32_t d[8];
  uint32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint64_t y = (uint64_t)t[i] - C::r32(i) - borrow;
    d[i] = (uint32_t)y;
    borrow = (uint32_t)(y >> 63);
  }
  const bool keep = borrow && t[8] == 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = keep ? t[i] : d[i];
}

template <class C>
VRF_HD void fr_reduce256(uint32_t out[8], const uint32_t x[8]) {
  uint32_t r1[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) r1[i] = C::r_r1(i);
  fr_montmul<C>(out, x, r1);                 // x * (2^256 mod r) / 2^256 = x mod r
}

template <class C>
VRF_HD void fr_add(uint32_t out[8], const uint32_t a[8], const uint32_t b[8]) {   // a, b < r
  uint32_t s[8], d[8];
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint64_t x = (uint64_t)a[i] + b[i] + c;
    s[i] = (uint32_t)x;
    c = (uint32_t)(x >> 32);
  }
  uint32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint64_t y = (uint64_t)s[i] - C::r32(i) - borrow;
    d[i] = (uint32_t)y;
    borrow = (uint32_t)(y >> 63);
  }
  const bool keep = borrow && c == 0;          // c: the sum's 257th bit (r > 2^255 only)
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = keep ? s[i] : d[i];
}

template <class C>
VRF_HD void fr_reduce512(uint32_t out[8], const uint32_t x[16]) {
  uint32_t r1[8], r2[8], lo[8], hi[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { r1[i] = C::r_r1(i); r2[i] = C::r_r2(i); }
  fr_montmul<C>(lo, x, r1);
  fr_montmul<C>(hi, x + 8, r2);              // hi * 2^512 / 2^256 = hi * 2^256 mod r
  fr_add<C>(out, lo, hi);
}

template <class C>
VRF_HD void fr_mul(uint32_t out[8], const uint32_t a[8], const uint32_t b[8]) {   // a, b < r
  uint32_t t[8], r2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) r2[i] = C::r_r2(i);
  fr_montmul<C>(t, a, b);
  fr_montmul<C>(out, t, r2);
}

template <class C>
VRF_HD bool fr_is_canonical(const uint32_t a[8]) {    // a < r
  bool lt = false, decided = false;
#pragma unroll
  for (int i = 7; i >= 0; --i) {
    uint32_t ri = C::r32(i);
    if (!decided && a[i] != ri) { lt = a[i] < ri; decided = true; }
  }
  return lt;
}

// Signed radix-16 recoding: k + 0x88..8 has nibbles (d_w + 8) with d_w in [-8, 7] and
// sum d_w 16^w = k.  Requires k < 2^255.
VRF_HD void scalar_recode_signed4(uint32_t out[8], const uint32_t k[8]) {
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint64_t x = (uint64_t)k[i] + 0x88888888u + c;
    out[i] = (uint32_t)x;
    c = (uint32_t)(x >> 32);
  }
}
// digit w (0..63) of a recoded scalar: value in [-8, 7]
VRF_HD int scalar_digit4(const uint32_t rec[8], int w) {
  uint32_t word = rec[0];
#pragma unroll
  for (int i = 1; i < 8; ++i)
    if ((w >> 3) == i) word = rec[i];
  return (int)((word >> ((w & 7) * 4)) & 15u) - 8;
}

// ------------------------------------------------------------------------ GLV decomposition
// k = k1 + k2 * LAMBDA (mod r) with |k1|, |k2| < 2^127 (observed < 2^126, bound 2^126.3):
// Babai rounding against the short lattice basis in constants.gen.h.  Any pair satisfying the
// congruence gives the same group element, so the rounding details never affect results.
struct GlvHalf {
  uint32_t mag[4];   // |k_i|
  bool neg;
};

// acc (160-bit two's complement) +/-= x * y, low 160 bits
VRF_HD void acc5_muladd(uint32_t acc[5], const uint32_t x[5], const uint32_t (&y)[5], bool subtract) {
  uint32_t p[5];
  uint64_t carry = 0;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    uint64_t lo = carry & 0xffffffffu, hi = carry >> 32;
#pragma unroll
    for (int i = 0; i <= k; ++i) {
      uint64_t t = (uint64_t)x[i] * y[k - i];
      lo += t & 0xffffffffu;
      hi += t >> 32;
    }
    p[k] = (uint32_t)lo;
    carry = hi + (lo >> 32);
  }
  uint64_t c = subtract ? 1 : 0;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    uint64_t t = (uint64_t)acc[k] + (subtract ? ~p[k] : p[k]) + c;
    acc[k] = (uint32_t)t;
    c = t >> 32;
  }
}

// (k * g + 2^255) >> 256, k: 8 words, g: 5 words -> 5 words
VRF_HD void glv_round_mul(uint32_t out[5], const uint32_t k[8], const uint32_t (&g)[5]) {
  uint32_t prod[13];
  uint64_t carry = 0;
#pragma unroll
  for (int col = 0; col < 13; ++col) {
    uint64_t lo = carry & 0xffffffffu, hi = carry >> 32;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int j = col - i;
      if (j >= 0 && j < 5) {
        uint64_t t = (uint64_t)k[i] * g[j];
        lo += t & 0xffffffffu;
        hi += t >> 32;
      }
    }
    if (col == 7) lo += 0x80000000u;          // + 2^255
    prod[col] = (uint32_t)lo;
    carry = hi + (lo >> 32);
  }
#pragma unroll
  for (int i = 0; i < 5; ++i) out[i] = prod[8 + i];
}

VRF_HD GlvHalf glv_finish(const uint32_t v[5]) {
  GlvHalf h;
  h.neg = (v[4] >> 31) != 0;
  uint64_t c = h.neg ? 1 : 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint64_t t = (uint64_t)(h.neg ? ~v[i] : v[i]) + c;
    h.mag[i] = (uint32_t)t;
    c = t >> 32;
  }
  return h;
}

// Bandersnatch constants (the only GLV curve among the suites; the other fields' builds only see the declaration, inside
// discarded `if constexpr (S::HAS_GLV)` branches)
#if VRF_FIELD != 0
VRF_HD void glv_decompose_bs(GlvHalf& k1, GlvHalf& k2, const uint32_t k[8]);
#else
VRF_HD void glv_decompose_bs(GlvHalf& k1, GlvHalf& k2, const uint32_t k[8]) {
  uint32_t c1[5], c2[5];
  glv_round_mul(c1, k, vrfk::BS_GLV_G1);
  glv_round_mul(c2, k, vrfk::BS_GLV_G2);
  // k1 = k - c1*a1 - c2*a2 ; k2 = -c1*b1 - c2*b2   (c_i carry the signs C1_NEG / C2_NEG)
  uint32_t v1[5], v2[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) { v1[i] = k[i]; v2[i] = 0; }
  acc5_muladd(v1, c1, vrfk::BS_GLV_A1_MAG, !(vrfk::BS_GLV_C1_NEG ^ vrfk::BS_GLV_A1_NEG));
  acc5_muladd(v1, c2, vrfk::BS_GLV_A2_MAG, !(vrfk::BS_GLV_C2_NEG ^ vrfk::BS_GLV_A2_NEG));
  acc5_muladd(v2, c1, vrfk::BS_GLV_B1_MAG, !(vrfk::BS_GLV_C1_NEG ^ vrfk::BS_GLV_B1_NEG));
  acc5_muladd(v2, c2, vrfk::BS_GLV_B2_MAG, !(vrfk::BS_GLV_C2_NEG ^ vrfk::BS_GLV_B2_NEG));
  k1 = glv_finish(v1);
  k2 = glv_finish(v2);
}
#endif

// signed radix-16 recoding of a 128-bit magnitude (< 2^126.9): 32 digits in [-8, 7]
VRF_HD void scalar_recode_signed4_128(uint32_t out[4], const uint32_t k[4]) {
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint64_t x = (uint64_t)k[i] + 0x88888888u + c;
    out[i] = (uint32_t)x;
    c = (uint32_t)(x >> 32);
  }
}
VRF_HD int scalar_digit4_128(const uint32_t rec[4], int w) {   // w in 0..31
  uint32_t word = rec[0];
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if ((w >> 3) == i) word = rec[i];
  return (int)((word >> ((w & 7) * 4)) & 15u) - 8;
}

VRF_NS_END
