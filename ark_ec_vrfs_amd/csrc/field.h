// field.h -- which BASE FIELD this translation unit is compiled for.
//
// The VRF kernels are compiled once per base field (Makefile: -DVRF_FIELD=n): the suites of `suites`
// (/root/reference src/lib.rs:14) live over four different fields, and a field's arithmetic -- Montgomery digit rule,
// reduction, square-root tables -- is chosen at compile time so that the inner loop carries no field switch:
//
//   VRF_FIELD 0  BLS12-381 Fr   Bandersnatch (a = -5), JubJub (a = -1)        constants.gen.h
//   VRF_FIELD 1  2^255 - 19     Ed25519 (a = -1)                              constants_f25519.gen.h
//   VRF_FIELD 2  BN254 Fr       Baby-JubJub (a = 1)                           constants_fbn254.gen.h
//   VRF_FIELD 3  NIST P-256 Fp  secp256r1 (short Weierstrass, sw.cuh)         constants_fp256.gen.h
//
// Everything field-dependent sits in an inline namespace named after the field, so the builds of the same source
// link into one library without sharing a symbol; the plain-data launch arguments (vrf_types.h) are common to all.
#pragma once

#ifndef VRF_FIELD
#define VRF_FIELD 0
#endif

#if VRF_FIELD == 0
#include "constants.gen.h"
#define VRF_FNS f_bls381fr
#elif VRF_FIELD == 1
#include "constants_f25519.gen.h"
#define VRF_FNS f_25519
#elif VRF_FIELD == 2
#include "constants_fbn254.gen.h"
#define VRF_FNS f_bn254fr
#elif VRF_FIELD == 3
#include "constants_fp256.gen.h"
#define VRF_FNS f_p256
#else
#error "VRF_FIELD must be 0 (BLS12-381 Fr), 1 (2^255 - 19), 2 (BN254 Fr) or 3 (NIST P-256 Fp)"
#endif

#define VRF_NS_BEGIN namespace vrf { inline namespace VRF_FNS {
#define VRF_NS_END } }
