// fr.cuh -- scalar-field (mod r) helpers: a handful of operations per proof, so a plain
// 8 x u32 CIOS Montgomery multiplier (R = 2^256) with the modulus taken from the curve tag.
// Stands in for ark_ff arithmetic on `ScalarField` (/root/reference src/lib.rs:15) where the
// VRF schemes reduce hashes mod r and form s = k + c*sk.
#pragma once
#include "fe.cuh"

namespace vrf {

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
  // t < 2r < 2^256: one conditional subtraction
  uint32_t d[8];
  uint32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint64_t y = (uint64_t)t[i] - C::r32(i) - borrow;
    d[i] = (uint32_t)y;
    borrow = (uint32_t)(y >> 63);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = borrow ? t[i] : d[i];
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
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = borrow ? s[i] : d[i];
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

}  // namespace vrf
