/* vrfhip.h -- C ABI of libvrfhip: MI355X-native batch EC-VRF prove / verify.
 *
 * This is the drop-in boundary for the hot path of the `ark-ec-vrfs` / `ark-vrf` Rust API.
 * The reference (/root/reference) has no FFI of its own: its whole surface is the re-export
 * list at src/lib.rs:13-17.  Each entry point below names the Rust item of that list it
 * replaces; INTEGRATION.md shows the `extern "C"` block and the safe Rust wrapper a maintainer
 * would add to route batches through this library.
 *
 * Conventions
 *  - All arrays are struct-of-arrays, item-major, caller-allocated.
 *  - Points travel in the ArkworksCodec wire format (`codec`, src/lib.rs:14): 32 bytes,
 *    y little-endian, bit 255 = (x > q - x).  Scalars: 32 bytes little-endian, canonical (< r).
 *  - `*_dev` entry points take DEVICE pointers (hipMalloc'ed / torch CUDA tensors) and a
 *    hipStream_t passed as void*; they enqueue work and return without synchronising.
 *    The plain entry points take HOST pointers, copy in, run, copy out and synchronise.
 *    Device arrays of 32-, 64-, 96- and 192-byte items must be 4-byte aligned (the kernels read them as 32-bit
 *    words; hipMalloc and tensor allocators give 256 bytes); message and `ad` blobs may have any alignment.
 *  - `ad` (additional data): one blob.  If `ad_off` is NULL every item uses the whole blob
 *    (`ad_len` bytes); otherwise item i uses blob[ad_off[i] .. ad_off[i+1]) (n+1 offsets).
 *  - Return value: 0 on success, negative vrfhip_error on API / runtime failure (no partial
 *    output guarantee).  Per-item outcomes go to a status byte array mirroring `Error`
 *    (src/lib.rs:15): 0 = Ok, 1 = VerificationFailure, 2 = InvalidData.
 *  - Thread safety: a context may be shared; calls on one context serialise internally.
 *    A context owns ONE workspace: `*_dev` calls on the same context must all be enqueued on the
 *    same stream (or be ordered by the caller's events); for concurrent streams use one context each.
 *  - Points are wire data: every verify entry point (and prove with a given input point) decodes them with
 *    the semantics of arkworks' checked deserialisation, which is what `codec::point_decode` applies before a
 *    `Public` / `Input` / `Output` / proof value exists (src/lib.rs:14): on the curve AND in the prime-order
 *    subgroup.  A point that fails either test, and a non-canonical scalar, give InvalidData -- with the two laxities of
 *    upstream's own decoding: the challenge `c` of an IETF proof is taken mod r (`Proof::c` is decoded with
 *    from_le_bytes_mod_order; `s`, `sb` and secrets are strict), and a compressed point with x = 0 is accepted whatever
 *    its sign flag says and enters the transcript hashes in its canonical encoding (flag clear).  A caller that
 *    holds already validated points (typed arkworks values, an `Input` it hashed itself) can switch the
 *    subgroup test off per point class with vrfhip_ctx_set_flags.
 */
#ifndef VRFHIP_H
#define VRFHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VRFHIP_POINT_BYTES 32
#define VRFHIP_SCALAR_BYTES 32
#define VRFHIP_HASH_BYTES 64

typedef enum vrfhip_suite {
  VRFHIP_SUITE_BANDERSNATCH_SHA512_ELL2 = 1, /* `suites::bandersnatch` (src/lib.rs:14) */
  /* `suites::jubjub`: a = -1, cofactor 8, try-and-increment hash-to-curve.  Suite string, TAI
   * details and the Pedersen blinding base are recollections: parity unpinned.  The built-in
   * descriptor uses the suite string "JubJub_SHA-512_TAI" and, as blinding base, the TAI hash of
   * "vrfhip-jubjub-blinding-base" -- a caller that knows the upstream constants supplies them through
   * vrfhip_ctx_create_desc. */
  VRFHIP_SUITE_JUBJUB_SHA512_TAI = 2,
  /* `suites::ed25519` ("Ed25519_SHA-512_TAI"): edwards25519 over 2^255 - 19, cofactor 8, try-and-increment, ArkworksCodec,
   * `CHALLENGE_LEN = 16`.  The suite string, the challenge length and the (absent) salt are recollections of upstream and the
   * built-in blinding base is the TAI hash of "vrfhip-ed25519-blinding-base": parity with upstream unpinned.  The SAME
   * kernels run RFC 9381's ECVRF-EDWARDS25519-SHA512-TAI when the descriptor says so (suite string 0x03, the three
   * VRFHIP_SUITE_FLAG_* below, the public key prepended to the message as salt), and that suite's published vectors
   * (RFC 9381 Appendix B.3) are golden vectors of this library: tests/golden/rfc9381_edwards25519_sha512_tai.json. */
  VRFHIP_SUITE_ED25519_SHA512_TAI = 3,
  /* `suites::baby_jubjub` ("BabyJubJub_SHA-512_TAI"): ark-ed-on-bn254 (a = 1, d = 168696/168700 over BN254 Fr), cofactor 8,
   * try-and-increment, `CHALLENGE_LEN = 32`.  Suite string and blinding base as for JubJub: recollection / built-in TAI
   * point, replaceable through the descriptor; parity unpinned. */
  VRFHIP_SUITE_BABY_JUBJUB_SHA512_TAI = 4,
  /* `suites::secp256r1` ("P256_SHA256_TAI" = RFC 9381's ECVRF-P256-SHA256-TAI, suite_string 0x01): NIST P-256, a short
   * Weierstrass curve of prime order (cofactor 1), SHA-256, `Sec1Codec`, `nonce_rfc_6979`, try-and-increment,
   * `CHALLENGE_LEN = 16`.  This suite has its OWN WIRE FORMAT at every entry point that accepts it: points are 33-byte SEC1
   * compressed strings (0x02 / 0x03 || x big-endian), scalars (secret keys, `c`, `s`) 32-byte BIG-endian integers (`c` has
   * 16 significant bytes), `Output::hash` is 32 bytes (vrfhip_ctx_point_bytes / vrfhip_ctx_hash_bytes tell).  As in the
   * other suites a secret key and the proof's `s` must be canonical (< n, else InvalidData: RFC 9381 5.4.4) while `c` is taken
   * mod n; a point needs tag 0x02 / 0x03, x < p and to lie on the curve (cofactor 1: no subgroup test).  Entry points:
   * vrfhip_ietf_prove_batch / _verify_batch (+ _dev, _multi), vrfhip_ietf_verify_batch_alpha, vrfhip_hash_to_curve_batch, vrfhip_output_hash_batch,
   * vrfhip_secret_from_seed_batch, vrfhip_point_validate_batch (+ _dev), and the Pedersen scheme per proof:
   * vrfhip_pedersen_prove_batch / vrfhip_pedersen_verify_batch (+ _dev, _multi) when the descriptor carries a blinding base
   * (the built-in one is a nothing-up-my-sleeve point: upstream's `BLINDING_BASE` for this suite is not known here; a
   * descriptor with an all-zero base makes a context without the scheme).  The x || y forms work as for the other suites
   * (vrfhip_ietf_verify_batch_affine; VRFHIP_FLAG_PROVE_POINTS_AFFINE for the provers' Gamma / pk / pk_com / R / Ok;
   * VRFHIP_FLAG_COORDS_MONT256; always little-endian, as arkworks holds coordinates in memory).  Since round 4 also vrfhip_msm, the batched
   * Pedersen verifier (vrfhip_pedersen_verify_batch_rlc, both forms) and key sets: every entry point exists on this suite.  Pinned by RFC 9381 Appendix B.1, which upstream's own
   * tests run: tests/golden/rfc9381_p256_sha256_tai.json (the RFC's use: message = PK_string || alpha).  As upstream, the
   * RFC 6979 nonce takes h1 unreduced and the first HMAC_DRBG candidate mod n (each differs from the RFC text with
   * probability 2^-32). */
  VRFHIP_SUITE_SECP256R1_SHA256_TAI = 5,
  /* `suites::bandersnatch_sw` ("Bandersnatch_SW_SHA-512_TAI"): the Bandersnatch group on its short-Weierstrass model
   * (ark-ed-on-bls12-381-bandersnatch SWAffine: y^2 = x^3 + a' x + b', the Weierstrass form of the curve's Montgomery model),
   * SHA-512, ArkworksCodec, `nonce_rfc_8032`, try-and-increment, `CHALLENGE_LEN = 32`.  WIRE FORMAT: points are arkworks'
   * 33-byte compressed short-Weierstrass strings -- x as a 32-byte little-endian integer, then one flag byte (0x80: y is the
   * larger of {y, q - y}; 0x40: the point at infinity, written with x = 0; both = error; the six low bits are not looked at) --
   * scalars 32-byte little-endian, `Output::hash` 64 bytes.  The arithmetic is the twisted-Edwards suite's (same group:
   * points cross `utils::te_sw_map` at the codec, csrc/bsw_core.cuh), so speed and the checked-decode rules are that suite's.
   * Entry points: every one the twisted-Edwards suites have -- keys, hash-to-curve, output hash, point validation, the IETF
   * scheme (vrfhip_ietf_prove_batch / _verify_batch + _dev, _multi, _alpha, _affine, _keyed with vrfhip_keyset_create over
   * 33-byte keys), the Pedersen scheme (per proof and vrfhip_pedersen_verify_batch_rlc; needs a blinding base in the
   * descriptor), vrfhip_msm (bases and out_xy: WEIERSTRASS x || y; the sum as a 33-byte string), vrfhip_te_sw_map_batch, the
   * pairing / G1 entry points.  Every x || y this suite reads or writes (VRFHIP_FLAG_PROVE_POINTS_AFFINE, the _affine verifier,
   * vrfhip_point_validate_batch) is in Weierstrass coordinates, little-endian, Montgomery-256 under VRFHIP_FLAG_COORDS_MONT256.  PARITY UNPINNED: no vector of this suite is on this machine;
   * the suite string, the generator (= te_sw_map image of the twisted-Edwards suite's) and the flag convention are
   * recollections; what is checked is consistency with the vector-pinned twisted-Edwards suite through the map
   * (tests/test_bandersnatch_sw.py).  With the subgroup test switched off (VRFHIP_FLAG_PREVALIDATED_*) a point with y = 0 is
   * InvalidData here (it has no Edwards image) where upstream would compute with it. */
  VRFHIP_SUITE_BANDERSNATCH_SW_SHA512_TAI = 6
} vrfhip_suite;

/* Suite descriptor: what a `Suite` / `PedersenSuite` impl states as DATA (src/lib.rs:16 `Suite`, :14 `suites`):
 * `Suite::SUITE_ID`, the hash-to-curve domain separation tag, `Suite::generator()` and
 * `PedersenSuite::BLINDING_BASE`.  The arithmetic (field, curve coefficients, cofactor, subgroup order, hash-to-curve
 * construction, endomorphism) is compiled per `curve`.  Lets a caller run a suite whose constants this library does not
 * carry -- in particular the upstream JubJub suite string and blinding base, which could not be authenticated here
 *  -- and makes the built-in suites nothing more than two pre-filled descriptors. */
typedef enum vrfhip_curve {
  VRFHIP_CURVE_BANDERSNATCH = 1, /* ark-ed-on-bls12-381-bandersnatch: a = -5, cofactor 4; Elligator 2 (RFC 9380) */
  VRFHIP_CURVE_JUBJUB = 2,       /* ark-ed-on-bls12-381 (JubJub): a = -1, cofactor 8; try-and-increment (RFC 9381) */
  VRFHIP_CURVE_ED25519 = 3,      /* ark-ed25519: q = 2^255 - 19, a = -1, cofactor 8; try-and-increment */
  VRFHIP_CURVE_BABY_JUBJUB = 4,  /* ark-ed-on-bn254: q = BN254 Fr, a = 1, cofactor 8; try-and-increment */
  VRFHIP_CURVE_SECP256R1 = 5,    /* ark-secp256r1: y^2 = x^3 - 3x + b over the NIST P-256 prime, cofactor 1; try-and-increment,
                                    SHA-256, Sec1 wire format.  Descriptor: suite_id, challenge_len, generator and
                                    blinding_base are read (points still x || y little-endian; an all-zero blinding_base =
                                    no Pedersen scheme); flags must be 0 */
  VRFHIP_CURVE_BANDERSNATCH_SW = 6 /* ark-ed-on-bls12-381-bandersnatch SWAffine: Bandersnatch's short-Weierstrass model,
                                    cofactor 4; try-and-increment, 33-byte arkworks SW wire format.  Descriptor: suite_id,
                                    challenge_len, generator and blinding_base (x || y of the WEIERSTRASS points, little-endian;
                                    an all-zero blinding_base = no Pedersen scheme) are read; flags must be 0 */
} vrfhip_curve;

/* What a `Suite` impl may override besides its constants (`Suite::Codec`, `Suite::challenge`, `Suite::point_to_hash`):
 * the deviations of RFC 9381's edwards suites from upstream's built-in ones.  0 for every built-in descriptor. */
#define VRFHIP_SUITE_FLAG_SIGN_PARITY 1u   /* compressed points carry x mod 2 in bit 255 (RFC 8032) instead of x > q - x */
#define VRFHIP_SUITE_FLAG_CHALLENGE_LE 2u  /* the truncated challenge hash is a little-endian integer */
#define VRFHIP_SUITE_FLAG_HASH_COFACTOR 4u /* Output::hash hashes cofactor * Gamma (RFC 9381 proof_to_hash) */
#define VRFHIP_SUITE_FLAG_ALL 7u

typedef struct vrfhip_suite_desc {
  uint32_t struct_size;      /* sizeof(vrfhip_suite_desc) of the caller's header */
  int32_t curve;             /* vrfhip_curve; it also fixes the hash-to-curve construction (see the enum) */
  uint32_t suite_id_len;     /* 1..64 */
  uint8_t suite_id[64];      /* `Suite::SUITE_ID` */
  uint32_t h2c_dst_len;      /* Bandersnatch: 1..128, the RFC 9380 DST (upstream: "ECVRF_" || h2c suite id || SUITE_ID);
                                try-and-increment curves: ignored */
  uint8_t h2c_dst[128];
  uint8_t generator[64];     /* `Suite::generator()`: x || y, 32-byte little-endian canonical integers */
  uint8_t blinding_base[64]; /* `PedersenSuite::BLINDING_BASE`, same form */
  uint32_t challenge_len;    /* `Suite::CHALLENGE_LEN`, 1..32: c = the first challenge_len bytes of the challenge hash
                                (big-endian unless VRFHIP_SUITE_FLAG_CHALLENGE_LE) mod r; `Proof::c` stays a 32-byte scalar */
  uint32_t flags;            /* VRFHIP_SUITE_FLAG_* */
} vrfhip_suite_desc;

typedef enum vrfhip_status {
  VRFHIP_ST_OK = 0,
  VRFHIP_ST_VERIFICATION_FAILURE = 1, /* Error::VerificationFailure */
  VRFHIP_ST_INVALID_DATA = 2          /* Error::InvalidData */
} vrfhip_status;

typedef enum vrfhip_error {
  VRFHIP_SUCCESS = 0,
  VRFHIP_ERR_BAD_ARG = -1,
  VRFHIP_ERR_HIP = -2,
  VRFHIP_ERR_NO_DEVICE = -3,
  VRFHIP_ERR_OOM = -4,
  VRFHIP_ERR_UNSUPPORTED = -5
} vrfhip_error;

typedef struct vrfhip_ctx vrfhip_ctx;
typedef struct vrfhip_keyset vrfhip_keyset;

/* Library / context ------------------------------------------------------------------- */

/* ABI version of this header (major*100 + minor). */
int32_t vrfhip_abi_version(void);

/* Last error text for this thread ("" if none). */
const char* vrfhip_last_error(void);

/* Create a context on HIP device `device` for `suite`: uploads the square-root tables and
 * builds the fixed-base tables of G (and of the Pedersen blinding base) on the GPU.
 * Replaces the compile-time `Suite` selection (src/lib.rs:16). */
int32_t vrfhip_ctx_create(vrfhip_suite suite, int32_t device, vrfhip_ctx** out);
void vrfhip_ctx_destroy(vrfhip_ctx* ctx);

/* The built-in descriptor of `suite` (what vrfhip_ctx_create uses); a caller edits the fields it wants to replace.
 * `blinding_base` (`PedersenSuite::BLINDING_BASE`) is filled in for Bandersnatch only, where upstream's Pedersen vector
 * pins it.  For JubJub, Ed25519, Baby-JubJub and secp256r1 upstream's constant is not authenticated here, so the default
 * descriptor leaves it ALL ZERO: the context then has no Pedersen scheme (vrfhip_pedersen_* return
 * VRFHIP_ERR_UNSUPPORTED) until the caller writes the suite's constant into the descriptor -- the Rust crate does, from the
 * `PedersenSuite` trait.  (Rounds 1-3 shipped an invented point there: proofs made with it never verify upstream.) */
int32_t vrfhip_suite_desc_default(vrfhip_suite suite, vrfhip_suite_desc* out);
/* Test / bench only: a nothing-up-my-sleeve point of the suite's prime-order subgroup (x || y, 32-byte little-endian) to use
 * as blinding base where upstream's is unknown.  NOT upstream's constant; Bandersnatch returns its pinned one. */
int32_t vrfhip_test_blinding_base(vrfhip_suite suite, uint8_t out_xy[64]);
/* Create a context from a descriptor.  VRFHIP_ERR_BAD_ARG if a length is out of range or the generator / blinding
 * base is not a non-identity point of the curve's prime-order subgroup (checked on the device);
 * VRFHIP_ERR_UNSUPPORTED for an unknown curve or challenge length. */
int32_t vrfhip_ctx_create_desc(const vrfhip_suite_desc* desc, int32_t device, vrfhip_ctx** out);
/* The descriptor a context was created from. */
int32_t vrfhip_ctx_get_desc(const vrfhip_ctx* ctx, vrfhip_suite_desc* out);

/* Which point classes the caller vouches for (prime-order subgroup membership already established, e.g. typed
 * arkworks values or an input point the caller hashed itself): their subgroup test is skipped; the on-curve
 * test always runs.  Default 0: everything is checked, as arkworks' deserialisation does.  Skipping the test on
 * attacker-supplied bytes breaks VRF uniqueness (an output shifted by a 2-torsion point verifies with an even
 * challenge) and voids the bound of the batched Pedersen verifier. */
#define VRFHIP_FLAG_PREVALIDATED_PUBLIC 1u /* `Public` (pk of the IETF verifier) */
#define VRFHIP_FLAG_PREVALIDATED_INPUT 2u  /* `Input` (H) */
#define VRFHIP_FLAG_PREVALIDATED_OUTPUT 4u /* `Output` (Gamma) */
#define VRFHIP_FLAG_PREVALIDATED_PROOF 8u  /* `pedersen::Proof` points pk_com, R, Ok */
#define VRFHIP_FLAG_PREVALIDATED_ALL 15u
/* Output format of the provers (vrfhip_ietf_prove_batch*, vrfhip_pedersen_prove_batch*, their _dev and _multi forms): with
 * this flag every POINT they write -- output (Gamma), pk_out / pk_com, r, ok -- is n x 64 B, x || y as 32-byte little-endian
 * canonical integers, instead of the n x 32 B compressed encoding; the caller sizes those arrays accordingly.  A typed
 * `Output` / `pedersen::Proof` is then built with `Affine::new_unchecked(x, y)`: no square root and no subgroup test on the
 * CPU per point (decoding a compressed point costs the CPU more than the whole proof costs the GPU).  input_out (the
 * encoding of H), scalars and statuses are unchanged; a failed item's points are all-zero. */
#define VRFHIP_FLAG_PROVE_POINTS_AFFINE 16u
/* Coordinate format at the ABI (the optional fast path for arkworks' in-memory values).  With this
 * flag every x || y pair the library reads or writes -- the *_affine verify entry points, the bases and out_xy of vrfhip_msm,
 * xy_out of vrfhip_point_validate_batch, the provers' outputs under VRFHIP_FLAG_PROVE_POINTS_AFFINE -- holds the Montgomery
 * images x 2^256 mod q, y 2^256 mod q as four little-endian u64 each: the limbs of `ark_ff::Fp` (`p.x.0.0`), so the Rust side
 * copies memory instead of converting field elements (into_bigint / from_bigint are a Montgomery product each).  Compressed
 * points, scalars, the suite descriptor and vrfhip_fq_mul_batch are unaffected.  An all-zero pair (failed item) stays all-zero. */
#define VRFHIP_FLAG_COORDS_MONT256 32u
/* Hardening of the provers (not a parity matter: arkworks' mul_bigint is not constant-time either, `Secret` only zeroizes).
 * The provers multiply H by the secret key and by the nonce through per-proof window tables of 8 entries, and a lookup reads
 * the entry a secret digit names.  With this flag every such lookup reads ALL eight entries and keeps one by masks, so the
 * addresses the prover touches in its per-proof tables no longer follow the secret (cost: INTEGRATION.md section 4).  The
 * fixed-base combs of G and of the blinding base stay indexed by secret digits: they are shared read-only tables resident
 * in L2 / MALL, whose access pattern the flag does not change.  Proof bytes are identical with and without it. */
#define VRFHIP_FLAG_CT_TABLES 64u
int32_t vrfhip_ctx_set_flags(vrfhip_ctx* ctx, uint32_t flags);
uint32_t vrfhip_ctx_get_flags(const vrfhip_ctx* ctx);

/* Widths of the context's suite at this ABI: bytes of one compressed point (32; 33 for secp256r1: `Codec::point_encode`)
 * and of `Output::hash` (64 = SHA-512; 32 for secp256r1 = SHA-256).  Scalars are 32 bytes in every suite. */
size_t vrfhip_ctx_point_bytes(const vrfhip_ctx* ctx);
size_t vrfhip_ctx_hash_bytes(const vrfhip_ctx* ctx);

/* Size the internal HBM workspace for exactly `max_items` items per launch group (optional).
 * Larger batches are processed in chunks of `max_items`.  Without this call the workspace
 * grows on demand up to 2^20 items (about 7.3 KiB of HBM per item; a context also holds 113 MB of
 * fixed-base tables of the two generators). */
int32_t vrfhip_ctx_reserve(vrfhip_ctx* ctx, size_t max_items);

/* Bytes of device workspace currently held by the context. */
size_t vrfhip_ctx_workspace_bytes(const vrfhip_ctx* ctx);

/* Page-locked host memory for the arrays handed to the host-pointer entry points.  Those entry points cut a batch into
 * chunks of 2^18 items and send chunk k + 1 while chunk k is computed; arrays in pinned memory (from here, hipHostMalloc or
 * hipHostRegister) leave by DMA as they lie, pageable arrays are first gathered into a pinned staging ring by the calling
 * thread.  Replaces nothing in the reference: it is where a Rust caller would put its `Vec<u8>` of wire bytes. */
int32_t vrfhip_host_alloc(size_t bytes, void** out);
void vrfhip_host_free(void* p);

/* Per-stage device timing.  While enabled, every prove / verify launch group records hipEvents
 * on its launch stream around its kernels: verify = {decode, straus V, straus U, finish},
 * Pedersen verify = {decode, straus A, straus B, finish}, prove = {prepare, mul, (empty), finish}.
 * vrfhip_ctx_profile_read waits for the recorded events, returns the summed milliseconds per
 * stage and the number of launch groups, and clears the record. */
int32_t vrfhip_ctx_profile(vrfhip_ctx* ctx, int32_t enable);
int32_t vrfhip_ctx_profile_read(vrfhip_ctx* ctx, double stage_ms[4], uint64_t* launches);

/* IETF VRF ---------------------------------------------------------------------------- */

/* `ietf::Verifier::verify(&public, input, output, ad, &proof)` for n items (src/lib.rs:14).
 * pk, input (H), output (Gamma): n x 32 B points; c, s: n x 32 B proof scalars.
 * status: n bytes (vrfhip_status). */
int32_t vrfhip_ietf_verify_batch(vrfhip_ctx* ctx, size_t n, const uint8_t* pk, const uint8_t* input,
                                 const uint8_t* output, const uint8_t* c, const uint8_t* s,
                                 const uint8_t* ad, const uint32_t* ad_off, uint32_t ad_len,
                                 uint8_t* status);
int32_t vrfhip_ietf_verify_batch_dev(vrfhip_ctx* ctx, size_t n, const uint8_t* d_pk,
                                     const uint8_t* d_input, const uint8_t* d_output,
                                     const uint8_t* d_c, const uint8_t* d_s, const uint8_t* d_ad,
                                     const uint32_t* d_ad_off, uint32_t ad_len, uint8_t* d_status,
                                     void* stream);

/* The same verification for callers that hold the points in memory as arkworks `Affine { x, y }`
 * (the form `Public`, `Input`, `Output` wrap): pk_xy, input_xy, output_xy are n x 64 B, x || y as
 * 32-byte little-endian canonical integers.  No square roots are needed; InvalidData = coordinate
 * >= q or point off the curve. */
/* `Input::new(alpha)` + `ietf::Verifier::verify` (src/lib.rs:14-16) in one call: what a deployed verifier holds is the
 * public key, the message alpha and the proof -- H is not on the wire, the verifier hashes alpha to the curve itself.  msg /
 * msg_off / msg_len name the messages as in vrfhip_ietf_prove_batch (n+1 offsets into msg, or msg_off = NULL and msg_len
 * bytes each).  Statuses are those of vrfhip_hash_to_curve_batch followed by vrfhip_ietf_verify_batch; H stays on the device
 * as affine coordinates, so its compression, second square root and subgroup test (a cofactor multiple needs none) are not
 * paid.  (secp256r1: the two stages run as they are inside the one call -- cofactor 1, nothing to skip.) */
int32_t vrfhip_ietf_verify_batch_alpha(vrfhip_ctx* ctx, size_t n, const uint8_t* pk, const uint8_t* msg,
                                       const uint32_t* msg_off, uint32_t msg_len, const uint8_t* output,
                                       const uint8_t* c, const uint8_t* s, const uint8_t* ad,
                                       const uint32_t* ad_off, uint32_t ad_len, uint8_t* status);
int32_t vrfhip_ietf_verify_batch_alpha_dev(vrfhip_ctx* ctx, size_t n, const uint8_t* d_pk, const uint8_t* d_msg,
                                           const uint32_t* d_msg_off, uint32_t msg_len, const uint8_t* d_output,
                                           const uint8_t* d_c, const uint8_t* d_s, const uint8_t* d_ad,
                                           const uint32_t* d_ad_off, uint32_t ad_len, uint8_t* d_status,
                                           void* stream);

int32_t vrfhip_ietf_verify_batch_affine(vrfhip_ctx* ctx, size_t n, const uint8_t* pk_xy,
                                        const uint8_t* input_xy, const uint8_t* output_xy,
                                        const uint8_t* c, const uint8_t* s, const uint8_t* ad,
                                        const uint32_t* ad_off, uint32_t ad_len, uint8_t* status);
int32_t vrfhip_ietf_verify_batch_affine_dev(vrfhip_ctx* ctx, size_t n, const uint8_t* d_pk_xy,
                                            const uint8_t* d_input_xy, const uint8_t* d_output_xy,
                                            const uint8_t* d_c, const uint8_t* d_s, const uint8_t* d_ad,
                                            const uint32_t* d_ad_off, uint32_t ad_len,
                                            uint8_t* d_status, void* stream);

/* Keyed verification: many proofs per public key (`Public`, src/lib.rs:15).  A key set keeps, for each of
 * n_keys public keys, a validated point and its 8-bit fixed-base comb (881,280 bytes per key) resident in
 * HBM: 100,000 keys take 88 GB of the 288 GB.  Verification against a key of the set needs no decompression
 * of pk and computes U = s*G - c*Y with 48 mixed additions and no doublings.
 * vrfhip_keyset_create: pks = n_keys x 32 B compressed points (host); status (nullable, host) receives
 * 0 / 2 per key: a key must decode to a point of the prime-order subgroup (the checked decode of `codec`).
 * Proofs that name an invalid or out-of-range key are reported InvalidData.  The key set belongs to `ctx`
 * and must be destroyed before it. */
int32_t vrfhip_keyset_create(vrfhip_ctx* ctx, size_t n_keys, const uint8_t* pks, uint8_t* status,
                             vrfhip_keyset** out);
void vrfhip_keyset_destroy(vrfhip_keyset* keys);
size_t vrfhip_keyset_bytes(const vrfhip_keyset* keys);

/* `ietf::Verifier::verify` for n items whose public keys belong to `keys`: key_index[i] selects the key of
 * item i; everything else as vrfhip_ietf_verify_batch.  Same statuses as verifying against the key's bytes. */
int32_t vrfhip_ietf_verify_batch_keyed(vrfhip_ctx* ctx, const vrfhip_keyset* keys, size_t n,
                                       const uint32_t* key_index, const uint8_t* input, const uint8_t* output,
                                       const uint8_t* c, const uint8_t* s, const uint8_t* ad,
                                       const uint32_t* ad_off, uint32_t ad_len, uint8_t* status);
int32_t vrfhip_ietf_verify_batch_keyed_dev(vrfhip_ctx* ctx, const vrfhip_keyset* keys, size_t n,
                                           const uint32_t* d_key_index, const uint8_t* d_input,
                                           const uint8_t* d_output, const uint8_t* d_c, const uint8_t* d_s,
                                           const uint8_t* d_ad, const uint32_t* d_ad_off, uint32_t ad_len,
                                           uint8_t* d_status, void* stream);

/* `Input::new(msg)` + `Secret::output` + `ietf::Prover::prove` for n items (src/lib.rs:14-16).
 * sk: n x 32 B secret scalars.  Messages: blob `msg`; if msg_off is NULL item i is
 * msg[i*msg_len .. (i+1)*msg_len), else msg[msg_off[i] .. msg_off[i+1]).
 * If `input` is non-NULL it holds n pre-hashed inputs H (32 B points) and msg is ignored.
 * Outputs (each n x 32 B; any of pk_out / input_out may be NULL): output (Gamma), proof c,
 * proof s, the public key sk*G and the input point H.  status: n bytes. */
int32_t vrfhip_ietf_prove_batch(vrfhip_ctx* ctx, size_t n, const uint8_t* sk, const uint8_t* msg,
                                const uint32_t* msg_off, uint32_t msg_len, const uint8_t* input,
                                const uint8_t* ad, const uint32_t* ad_off, uint32_t ad_len,
                                uint8_t* output, uint8_t* c, uint8_t* s, uint8_t* pk_out,
                                uint8_t* input_out, uint8_t* status);
int32_t vrfhip_ietf_prove_batch_dev(vrfhip_ctx* ctx, size_t n, const uint8_t* d_sk,
                                    const uint8_t* d_msg, const uint32_t* d_msg_off,
                                    uint32_t msg_len, const uint8_t* d_input, const uint8_t* d_ad,
                                    const uint32_t* d_ad_off, uint32_t ad_len, uint8_t* d_output,
                                    uint8_t* d_c, uint8_t* d_s, uint8_t* d_pk_out,
                                    uint8_t* d_input_out, uint8_t* d_status, void* stream);

/* Pedersen VRF ------------------------------------------------------------------------ */

/* `Input::new` + `Secret::output` + `pedersen::Prover::prove` for n items (src/lib.rs:14).
 * Inputs as for vrfhip_ietf_prove_batch.  Outputs (n x 32 B each): output (Gamma); the proof
 * `pedersen::Proof { pk_com, r, ok, s, sb }`; the secret blinding factor the Rust API returns
 * next to the proof (blinding_out, may be NULL); the input point (input_out, may be NULL). */
int32_t vrfhip_pedersen_prove_batch(vrfhip_ctx* ctx, size_t n, const uint8_t* sk, const uint8_t* msg,
                                    const uint32_t* msg_off, uint32_t msg_len, const uint8_t* input,
                                    const uint8_t* ad, const uint32_t* ad_off, uint32_t ad_len,
                                    uint8_t* output, uint8_t* pk_com, uint8_t* r, uint8_t* ok,
                                    uint8_t* s, uint8_t* sb, uint8_t* blinding_out,
                                    uint8_t* input_out, uint8_t* status);
int32_t vrfhip_pedersen_prove_batch_dev(vrfhip_ctx* ctx, size_t n, const uint8_t* d_sk,
                                        const uint8_t* d_msg, const uint32_t* d_msg_off,
                                        uint32_t msg_len, const uint8_t* d_input,
                                        const uint8_t* d_ad, const uint32_t* d_ad_off,
                                        uint32_t ad_len, uint8_t* d_output, uint8_t* d_pk_com,
                                        uint8_t* d_r, uint8_t* d_ok, uint8_t* d_s, uint8_t* d_sb,
                                        uint8_t* d_blinding_out, uint8_t* d_input_out,
                                        uint8_t* d_status, void* stream);

/* `pedersen::Verifier::verify(input, output, ad, &proof)` for n items (src/lib.rs:14). */
int32_t vrfhip_pedersen_verify_batch(vrfhip_ctx* ctx, size_t n, const uint8_t* input,
                                     const uint8_t* output, const uint8_t* pk_com, const uint8_t* r,
                                     const uint8_t* ok, const uint8_t* s, const uint8_t* sb,
                                     const uint8_t* ad, const uint32_t* ad_off, uint32_t ad_len,
                                     uint8_t* status);
int32_t vrfhip_pedersen_verify_batch_dev(vrfhip_ctx* ctx, size_t n, const uint8_t* d_input,
                                         const uint8_t* d_output, const uint8_t* d_pk_com,
                                         const uint8_t* d_r, const uint8_t* d_ok, const uint8_t* d_s,
                                         const uint8_t* d_sb, const uint8_t* d_ad,
                                         const uint32_t* d_ad_off, uint32_t ad_len,
                                         uint8_t* d_status, void* stream);

/* `pedersen::Verifier::verify` for a whole batch with ONE multi-scalar multiplication (random linear
 * combination).  With c_i recomputed from the proof's own points, the batch is
 * accepted iff  sum_i z_i (s_i H_i - c_i Gamma_i - Ok_i) + z'_i (s_i G + sb_i B - c_i pk_com_i - R_i)
 * is the neutral element, (z_i, z'_i) = 2 x 128 bits of SHA-512("vrfhip-rlc-v2" || seed || D || u64_le(i)), each forced to
 * 1 (mod 8).  D is a digest of every input byte of the launch group (points, scalars, ad: vrfhip_test_batch_digest states
 * it), so the weights are fixed only once the batch is.  `seed` (32 bytes, host memory) should be fresh randomness per
 * call: a batch holding an invalid proof is then accepted with probability <= 2^-125 (the weights are 128-bit values with
 * their low three bits fixed).  A prover who knows or predicts the seed can no longer choose proofs after the weights: it
 * has to search for batch contents whose own weights cancel its defect, 2^-125 per trial.
 * The bound needs all five points of every proof in the prime-order subgroup, which the decode stage checks
 * (InvalidData otherwise) unless the caller vouched for them with vrfhip_ctx_set_flags.  With PREVALIDATED
 * flags set on unvalidated bytes the bound is void: a defect of small order (a proof point shifted by a 2- or
 * 4-torsion point) is annihilated by a weight divisible by its order.  Weights are 1 (mod 8), so a single such
 * proof is still caught; two colluding proofs whose small-order defects cancel are not.
 *
 * _dev form: enqueues the work and returns.  d_status[i] = 0 if proof i is part of the batch sum,
 * 2 (InvalidData) if it does not decode (it is then left out of the sum).  d_fail_flag[0] = 0 if every
 * proof with status 0 verifies, 1 if at least one does not (run vrfhip_pedersen_verify_batch_dev to
 * find which).
 * Host form: same inputs as vrfhip_pedersen_verify_batch and the same per-proof statuses: when the
 * batch equation fails it re-runs the per-proof kernels.  *batch_ok (nullable) reports whether the
 * single-MSM path sufficed. */
int32_t vrfhip_pedersen_verify_batch_rlc(vrfhip_ctx* ctx, size_t n, const uint8_t* input,
                                         const uint8_t* output, const uint8_t* pk_com, const uint8_t* r,
                                         const uint8_t* ok, const uint8_t* s, const uint8_t* sb,
                                         const uint8_t* ad, const uint32_t* ad_off, uint32_t ad_len,
                                         const uint8_t seed[32], uint8_t* status, int32_t* batch_ok);
int32_t vrfhip_pedersen_verify_batch_rlc_dev(vrfhip_ctx* ctx, size_t n, const uint8_t* d_input,
                                             const uint8_t* d_output, const uint8_t* d_pk_com,
                                             const uint8_t* d_r, const uint8_t* d_ok, const uint8_t* d_s,
                                             const uint8_t* d_sb, const uint8_t* d_ad,
                                             const uint32_t* d_ad_off, uint32_t ad_len,
                                             const uint8_t seed[32], uint8_t* d_status,
                                             uint8_t* d_fail_flag, void* stream);

/* The batched verifier for callers that hold the five points in memory as arkworks `Affine { x, y }`
 * (what `Input`, `Output` and `pedersen::Proof` wrap): *_xy arrays are n x 64 B, x || y as 32-byte
 * little-endian canonical integers; s, sb as above.  No square roots are needed.  InvalidData =
 * coordinate >= q, point off the curve, or scalar >= r. */
int32_t vrfhip_pedersen_verify_batch_rlc_affine(vrfhip_ctx* ctx, size_t n, const uint8_t* input_xy,
                                                const uint8_t* output_xy, const uint8_t* pk_com_xy,
                                                const uint8_t* r_xy, const uint8_t* ok_xy, const uint8_t* s,
                                                const uint8_t* sb, const uint8_t* ad, const uint32_t* ad_off,
                                                uint32_t ad_len, const uint8_t seed[32], uint8_t* status,
                                                int32_t* batch_ok);
int32_t vrfhip_pedersen_verify_batch_rlc_affine_dev(vrfhip_ctx* ctx, size_t n, const uint8_t* d_input_xy,
                                                    const uint8_t* d_output_xy, const uint8_t* d_pk_com_xy,
                                                    const uint8_t* d_r_xy, const uint8_t* d_ok_xy,
                                                    const uint8_t* d_s, const uint8_t* d_sb,
                                                    const uint8_t* d_ad, const uint32_t* d_ad_off,
                                                    uint32_t ad_len, const uint8_t seed[32],
                                                    uint8_t* d_status, uint8_t* d_fail_flag, void* stream);

/* Multi-scalar multiplication ---------------------------------------------------------- */

/* `VariableBaseMSM::msm(bases, scalars)` on the suite curve (ark_ec, named in BASELINE.json;
 * reached through `reexports`, src/lib.rs:14): result = sum_i scalars[i] * bases[i].
 * bases_xy: n x 64 B affine points (x || y, 32-byte little-endian canonical each -- what
 * vrfhip_point_validate_batch emits); scalars: n x 32 B LE canonical.  Works for both suites.
 * out_point: 32 B compressed result; out_xy: 64 B affine result (may be NULL);
 * status: 1 byte, 0 = Ok, 2 = InvalidData (a coordinate >= q, a point off the curve, or a
 * scalar >= r; the outputs are then zeroed).  n = 0 gives the identity. */
int32_t vrfhip_msm(vrfhip_ctx* ctx, size_t n, const uint8_t* bases_xy, const uint8_t* scalars,
                   uint8_t* out_point, uint8_t* out_xy, uint8_t* status);
int32_t vrfhip_msm_dev(vrfhip_ctx* ctx, size_t n, const uint8_t* d_bases_xy, const uint8_t* d_scalars,
                       uint8_t* d_out_point, uint8_t* d_out_xy, uint8_t* d_status, void* stream);

/* Pairing check (ring-VRF tail) ------------------------------------------------------- */

/* `Pairing::multi_miller_loop` over two (G1, G2) pairs + `final_exponentiation` == 1 on
 * BLS12-381, per item: the KZG equation that ends `ring::Verifier::verify` (src/lib.rs:14).
 * g1: n x 2 x 96 B, each point x || y as 48-byte little-endian canonical integers;
 * g2: n x 2 x 192 B, each point x.c0 || x.c1 || y.c0 || y.c1 (48-byte LE each); if g2_shared
 * is non-zero, g2 holds a single pair of points (384 B) used by every item (the SRS case):
 * their Miller-loop lines are computed once (arkworks' `G2Prepared`) and stay in the context, so
 * later calls with the same pair reuse them.
 * An all-zero encoding is the point at infinity.  status[i]: 0 = product is one,
 * 1 = VerificationFailure, 2 = InvalidData (coordinate >= p or point off its curve).
 * Subgroup membership of the G1 / G2 inputs is the caller's precondition, as for arkworks' prepared
 * points. */
int32_t vrfhip_pairing_check_batch(vrfhip_ctx* ctx, size_t n, const uint8_t* g1, const uint8_t* g2,
                                   int32_t g2_shared, uint8_t* status);
int32_t vrfhip_pairing_check_batch_dev(vrfhip_ctx* ctx, size_t n, const uint8_t* d_g1,
                                       const uint8_t* d_g2, int32_t g2_shared, uint8_t* d_status,
                                       void* stream);

/* The same n checks against ONE shared G2 pair (g2_shared: 384 B, a KZG verifier's SRS) as a batch: with secret
 * 128-bit weights z_i = SHA-512("vrfhip-pairing-rlc-v2" || seed || D || u64_le(i))[0..16] (D: the batch digest of the g1
 * items, vrfhip_test_batch_digest),
 *     prod_i (e(A_i,Q0) e(B_i,Q1))^{z_i} = e(sum z_i A_i, Q0) e(sum z_i B_i, Q1):
 * two G1 multi-scalar multiplications (Pippenger, buckets in LDS) and ONE pairing check for the whole batch -- the
 * aggregation step in front of the pairing tail of `ring::Verifier::verify` (src/lib.rs:14).  g1: n x 192 B as above.
 * `seed`: 32 bytes unpredictable to whoever made the items; a batch holding a false item passes with probability
 * <= 2^-128 (the G1 points lie in the prime-order subgroup: the caller's precondition, as for arkworks' `G1Prepared`).
 * _dev form: d_status[i] = 0 (part of the batch) or 2 (InvalidData: left out); d_verdict[0] = 0 if the batch equation
 * holds, 1 if it does not (vrfhip_pairing_check_batch_dev names the item), 2 if the shared pair is invalid.
 * Host form: the same per-item statuses as vrfhip_pairing_check_batch with g2_shared = 1 (a failing batch is
 * re-checked per item); *batch_ok (nullable) reports whether the single pairing sufficed. */
int32_t vrfhip_pairing_check_batch_rlc(vrfhip_ctx* ctx, size_t n, const uint8_t* g1, const uint8_t* g2_shared,
                                       const uint8_t seed[32], uint8_t* status, int32_t* batch_ok);
int32_t vrfhip_pairing_check_batch_rlc_dev(vrfhip_ctx* ctx, size_t n, const uint8_t* d_g1, const uint8_t* d_g2_shared,
                                           const uint8_t seed[32], uint8_t* d_status, uint8_t* d_verdict, void* stream);

/* `VariableBaseMSM::msm` on BLS12-381 G1 (ark-bls12-381; the KZG commitment / aggregation primitive of the ring
 * suite): out = sum_i scalars[i] * bases[i].  bases: n x 96 B (x || y, 48-byte little-endian; all-zero = infinity);
 * scalars: n x 32 B little-endian, < r; out: 96 B in the same form; status: 1 byte, 0 = Ok, 2 = InvalidData (a
 * coordinate >= p, a point off the curve or a scalar >= r; out is then zeroed by the host form).  n = 0: infinity. */
int32_t vrfhip_g1_msm(vrfhip_ctx* ctx, size_t n, const uint8_t* bases, const uint8_t* scalars, uint8_t* out,
                      uint8_t* status);
int32_t vrfhip_g1_msm_dev(vrfhip_ctx* ctx, size_t n, const uint8_t* d_bases, const uint8_t* d_scalars, uint8_t* d_out,
                          uint8_t* d_status, void* stream);

/* Building blocks --------------------------------------------------------------------- */

/* `Input::new(data)` = Suite::data_to_point = hash_to_curve_ell2_rfc_9380 (src/lib.rs:14-16):
 * n messages -> n x 32 B points. */
int32_t vrfhip_hash_to_curve_batch(vrfhip_ctx* ctx, size_t n, const uint8_t* msg,
                                   const uint32_t* msg_off, uint32_t msg_len, uint8_t* points);
int32_t vrfhip_hash_to_curve_batch_dev(vrfhip_ctx* ctx, size_t n, const uint8_t* d_msg,
                                       const uint32_t* d_msg_off, uint32_t msg_len,
                                       uint8_t* d_points, void* stream);

/* `Output::hash()` = point_to_hash_rfc_9381 (src/lib.rs:15): n x 32 B Gamma -> n x 64 B. */
int32_t vrfhip_output_hash_batch(vrfhip_ctx* ctx, size_t n, const uint8_t* output, uint8_t* hash);
int32_t vrfhip_output_hash_batch_dev(vrfhip_ctx* ctx, size_t n, const uint8_t* d_output,
                                     uint8_t* d_hash, void* stream);

/* `Secret::from_seed(seed)` + `Secret::public()` (src/lib.rs:16): seeds are fixed-stride
 * (seed_len bytes each).  sk_out, pk_out: n x 32 B (pk_out may be NULL). */
int32_t vrfhip_secret_from_seed_batch(vrfhip_ctx* ctx, size_t n, const uint8_t* seeds,
                                      uint32_t seed_len, uint8_t* sk_out, uint8_t* pk_out);
int32_t vrfhip_secret_from_seed_batch_dev(vrfhip_ctx* ctx, size_t n, const uint8_t* d_seeds,
                                          uint32_t seed_len, uint8_t* d_sk_out, uint8_t* d_pk_out,
                                          void* stream);

/* `codec::point_decode` with arkworks' checked-deserialisation semantics (src/lib.rs:14):
 * status[i] = 0 if points[i] decodes to a curve point of the prime-order subgroup, else 2.
 * If xy_out is non-NULL it receives n x 64 B affine coordinates (x || y, little-endian). */
int32_t vrfhip_point_validate_batch(vrfhip_ctx* ctx, size_t n, const uint8_t* points,
                                    uint8_t* xy_out, uint8_t* status);
int32_t vrfhip_point_validate_batch_dev(vrfhip_ctx* ctx, size_t n, const uint8_t* d_points,
                                        uint8_t* d_xy_out, uint8_t* d_status, void* stream);

/* `utils::te_sw_map::{te_to_sw, sw_to_te}` (src/lib.rs:14 `utils`): the point map between the context's twisted-Edwards
 * curve and the short-Weierstrass form of its Montgomery model (B v^2 = u^3 + A u^2 + u with A = 2(a+d)/(a-d),
 * B = 4/(a-d), arkworks' MontCurveConfig; SW point ((u + A/3)/B, v/B)) -- what `suites::bandersnatch_sw` values and the
 * ring plumbing are converted with.  to_te = 0: in_xy are TE points, out_xy their SW images; to_te = 1: the inverse.
 * Points are n x 64 B affine x || y, 32-byte little-endian canonical integers (arkworks' in-memory limbs with
 * VRFHIP_FLAG_COORDS_MONT256).  status[i] = 0, or 2 where upstream returns None -- TE x = 0 or y = 1 (the identity and the
 * point of order 2 have no affine image), SW y = 0 or B x - A/3 = -1 -- or a coordinate is not below the modulus; out_xy[i]
 * is then all zero.  As upstream (`new_unchecked`), no curve-membership test.  Twisted-Edwards suites only. */
int32_t vrfhip_te_sw_map_batch(vrfhip_ctx* ctx, size_t n, int32_t to_te, const uint8_t* in_xy, uint8_t* out_xy,
                               uint8_t* status);
int32_t vrfhip_te_sw_map_batch_dev(vrfhip_ctx* ctx, size_t n, int32_t to_te, const uint8_t* d_in_xy, uint8_t* d_out_xy,
                                   uint8_t* d_status, void* stream);

/* Test-only primitive: the quad-distributed Fp12 operations of the pairing kernel (one item per DPP quad)
 * against the one-lane tower operations on the same operands.  fp12_pairs: n x 2 x 12 field elements of
 * 48 bytes, little-endian (reduced mod p by the loader).  status[i] = bit mask of differing operations
 * (1 mul, 2 sqr, 4 cyclotomic sqr, 8 mul_by_014, 16 frobenius, 32 conj / gather-scatter); 0 = all equal. */
int32_t vrfhip_test_pairing_quad_ops(vrfhip_ctx* ctx, size_t n, const uint8_t* fp12_pairs, uint8_t* status);
/* The same for the 8-lanes-per-item layout (an Fp2 split over a lane pair; the throughput path from 2^11 items on) and its
 * cross-lane moves.  status[i]: 1 mul, 2 sqr, 4 cyclotomic sqr (of x^((p^6-1)(p^2+1))), 8 mul_by_014, 16 frobenius,
 * 32 inverse, 64 conj / scatter / easy part, 128 a cross-lane move; 0 = all equal. */
int32_t vrfhip_test_pairing_oct_ops(vrfhip_ctx* ctx, size_t n, const uint8_t* fp12_pairs, uint8_t* status);

/* Test / tuning knobs of a context.  The library never reads the environment; what rounds 1-3 steered with VRFHIP_PAIRING,
 * VRFHIP_PAIRING_ROW and VRFHIP_PIPE_*_LOG2 is set here, by tests and tuning scripts only.  Not part of the reference's
 * API (/root/reference src/lib.rs:13-17 has no counterpart); defaults are what production runs. */
enum {
  VRFHIP_DEBUG_PAIRING_LAYOUT = 1,  /* 0 by batch size (default); 1 one item per lane, 2 per DPP quad, 3 per 16-lane row,
                                       4 per wave, 5 per 8 lanes (per-item G2: lines kernel + Miller kernel), 6 per 8 lanes in
                                       one kernel; | 0x100: do not prepare the lines of a shared G2 pair */
  VRFHIP_DEBUG_PIPE_FIRST_LOG2 = 2, /* host-pointer verify pipeline: log2 items of the first chunk (12..18, default 17) */
  VRFHIP_DEBUG_PIPE_CHUNK_LOG2 = 3, /* ... of the following chunks (12..18, default 18) */
  VRFHIP_DEBUG_PROVE_K = 4,         /* proofs per lane in the provers' prepare / finish stages: 0 = by batch size, 1, 2, 4, 8 */
  VRFHIP_DEBUG_P256_MSM_GROUPS = 5  /* point groups per window of the secp256r1 MSM: 0 = by batch size */
};
int32_t vrfhip_debug_set(vrfhip_ctx* ctx, int32_t key, int32_t value);

/* One call, several devices ------------------------------------------------------------ */

/* The batch entry points above over n_ctx contexts (one per GPU, same suite descriptor): items are independent
 * (SURVEY.md section 8e), so context g gets the contiguous slice [g*n/n_ctx, (g+1)*n/n_ctx) and one host thread;
 * no data crosses devices.  Host pointers, same semantics and statuses as the single-context call on the whole
 * batch.  This is what a single-process Rust caller uses to spread `prove` / `verify` over the 8 GPUs of a node
 * (multi-process callers shard the same way and gather results over RCCL: ark_ec_vrfs_amd/sharding.py).
 * Returns the first failing slice's error.  rlc_seed (vrfhip_pedersen_verify_batch_multi): NULL = per-proof
 * verification; non-NULL (32 bytes) = one multi-scalar multiplication per slice with the per-proof fallback. */
int32_t vrfhip_ietf_verify_batch_multi(vrfhip_ctx* const* ctxs, int32_t n_ctx, size_t n, const uint8_t* pk,
                                       const uint8_t* input, const uint8_t* output, const uint8_t* c, const uint8_t* s,
                                       const uint8_t* ad, const uint32_t* ad_off, uint32_t ad_len, uint8_t* status);
int32_t vrfhip_ietf_prove_batch_multi(vrfhip_ctx* const* ctxs, int32_t n_ctx, size_t n, const uint8_t* sk,
                                      const uint8_t* msg, const uint32_t* msg_off, uint32_t msg_len, const uint8_t* input,
                                      const uint8_t* ad, const uint32_t* ad_off, uint32_t ad_len, uint8_t* output,
                                      uint8_t* c, uint8_t* s, uint8_t* pk_out, uint8_t* input_out, uint8_t* status);
int32_t vrfhip_pedersen_prove_batch_multi(vrfhip_ctx* const* ctxs, int32_t n_ctx, size_t n, const uint8_t* sk,
                                          const uint8_t* msg, const uint32_t* msg_off, uint32_t msg_len,
                                          const uint8_t* input, const uint8_t* ad, const uint32_t* ad_off, uint32_t ad_len,
                                          uint8_t* output, uint8_t* pk_com, uint8_t* r, uint8_t* ok, uint8_t* s,
                                          uint8_t* sb, uint8_t* blinding_out, uint8_t* input_out, uint8_t* status);
int32_t vrfhip_pedersen_verify_batch_multi(vrfhip_ctx* const* ctxs, int32_t n_ctx, size_t n, const uint8_t* input,
                                           const uint8_t* output, const uint8_t* pk_com, const uint8_t* r,
                                           const uint8_t* ok, const uint8_t* s, const uint8_t* sb, const uint8_t* ad,
                                           const uint32_t* ad_off, uint32_t ad_len, const uint8_t* rlc_seed,
                                           uint8_t* status);

/* Test-only primitives: the pieces under the batch calls, on their own, host pointers.
 * point_add: out[i] = a[i] + b[i] (`AffinePoint` addition, ark_ec TE group law; src/lib.rs:15), compressed points,
 *            status 2 if an operand does not decode (no subgroup test: the law is what is tested);
 * scalar_mul: out[i] = scalars[i] * points[i] (`mul_bigint` through the provers' variable-base path);
 * sha512:   out[i] = SHA-512(msg_i), 64 bytes (`Suite::Hasher`, src/lib.rs:16);
 * xmd:      out[i] = expand_message_xmd(msg_i, DST of the context's descriptor, 96 bytes) with arkworks' 48-byte
 *           Z_pad (`utils::hash_to_curve_ell2_rfc_9380`'s expander, src/lib.rs:14).  Messages as in
 *           vrfhip_hash_to_curve_batch. */
int32_t vrfhip_test_point_add(vrfhip_ctx* ctx, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out,
                              uint8_t* status);
int32_t vrfhip_test_scalar_mul(vrfhip_ctx* ctx, size_t n, const uint8_t* scalars, const uint8_t* points, uint8_t* out,
                               uint8_t* status);
int32_t vrfhip_test_sha512(vrfhip_ctx* ctx, size_t n, const uint8_t* msg, const uint32_t* msg_off, uint32_t msg_len,
                           uint8_t* out);
int32_t vrfhip_test_xmd(vrfhip_ctx* ctx, size_t n, const uint8_t* msg, const uint32_t* msg_off, uint32_t msg_len,
                        uint8_t* out);

/* Test-only: the batch digest that the random-linear-combination entry points hash into their weights, on its own.
 *   leaf_i = SHA-512("vrfhip-leaf-v1" || u64_le(index0 + i) || arrays[0][i] || ... || arrays[n_arr-1][i] || ad_i || u32_le(|ad_i|))[0..32]
 *   node   = SHA-512("vrfhip-node-v1" || u32_le(count) || child_0 || ... )[0..32] over runs of 16 consecutive children
 *   root   = the single node of the last level (n <= 16: one level of nodes).
 * arrays[j]: n x widths[j] bytes (host, widths a multiple of 4, 1 <= n_arr <= 8); ad as in the verify calls (NULL: empty).
 * vrfhip_pedersen_verify_batch_rlc* digest (input, output, pk_com, r, ok, s, sb, ad) of every launch group with index0 = the
 * group's first proof; vrfhip_pairing_check_batch_rlc* digest the g1 items.  n >= 1. */
int32_t vrfhip_test_batch_digest(vrfhip_ctx* ctx, size_t n, int32_t n_arr, const uint8_t* const* arrays,
                                 const uint32_t* widths, const uint8_t* ad, const uint32_t* ad_off, uint32_t ad_len,
                                 uint64_t index0, uint8_t root[32]);

/* Test-only: how many proofs one lane of the inversion-sharing stages (decode, finish, prepare) handles for a
 * launch group of n items (1, 2, 4 or 8): lets the parity tests assert that every kernel variant was exercised. */
int32_t vrfhip_debug_proofs_per_lane(size_t n);

/* Test-only primitive: r[i] = a[i] * b[i] mod q on n x 32 B little-endian field elements
 * (exercises ark_ff::Fp mul through the 29-bit Montgomery pipeline). */
int32_t vrfhip_fq_mul_batch(vrfhip_ctx* ctx, size_t n, const uint8_t* a, const uint8_t* b,
                            uint8_t* r);

#ifdef __cplusplus
}
#endif
#endif /* VRFHIP_H */
