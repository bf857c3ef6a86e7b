// vrfhip.hpp -- C++ host-side mirror of the `ark-ec-vrfs` / `ark-vrf` API over the C ABI (vrfhip.h).
//
// The reference is a Rust crate whose whole surface is the re-export list at /root/reference src/lib.rs:13-17
// (`Suite`, `Secret`, `Public`, `Input`, `Output`, `ietf`, `pedersen`, `Error`, `codec`).  There is no Rust
// toolchain in this image, so the host side above the C ABI is written in C++ with the reference's names,
// argument order and error behaviour; INTEGRATION.md shows the equivalent Rust binding.  Header-only; every
// call forwards to libvrfhip.so (hand-written HIP kernels) -- there is no CPU path.
//
//   Rust (ark-vrf)                                         here (namespace ark_vrf_hip)
//   Secret::<S>::from_seed(seed)                           Secret<S>::from_seed(ctx, seed)
//   Secret::<S>::from_scalar(scalar)                       Secret<S>::from_scalar(ctx, scalar)
//   secret.public()                                        secret.public_key()
//   Input::<S>::new(data) -> Option<Input>                 Input<S>::new_(ctx, data) -> std::optional<Input<S>>
//   secret.output(input)                                   secret.output(ctx, input)
//   output.hash()                                          output.hash(ctx)
//   ietf::Prover::prove(&secret, input, output, ad)        ietf::prove(ctx, secret, input, output, ad)
//   ietf::Verifier::verify(&public, input, output, ad, &p) ietf::verify(ctx, public, input, output, ad, p) -> Result
//   pedersen::Prover::prove(..) -> (Proof, blinding)       pedersen::prove(..) -> std::pair<Proof, Scalar>
//   pedersen::Verifier::verify(input, output, ad, &p)      pedersen::verify(ctx, input, output, ad, p) -> Result
//   utils::te_sw_map::{te_to_sw, sw_to_te}(point)          utils::te_to_sw(ctx, points) / utils::sw_to_te(ctx, points)
//   Error::{VerificationFailure, InvalidData}              enum class Error; Result = std::optional<Error> (nullopt = Ok(()))
// Suites: the four twisted-Edwards ones (32-byte ArkworksCodec points, SHA-512), `suites::bandersnatch_sw` (33-byte
// ArkworksCodec short-Weierstrass points) and `suites::secp256r1` (33-byte Sec1
// points, big-endian scalars, SHA-256) through the same templates: Point<S> / Hash<S> carry the widths.
// Several GPUs from one process: ietf::verify_batch_sharded over one Context per device (contiguous slices, one
// host thread per context, no exchange) -- the single-process form of `bench.py --gpus N`.
// Batch forms (what the GPU is for) take std::vector of the same types: ietf::verify_batch, ietf::prove_batch,
// pedersen::verify_batch (single-MSM random-linear-combination path with automatic per-proof fallback),
// KeySet + ietf::verify_batch_keyed (public keys with HBM-resident fixed-base tables).
#ifndef VRFHIP_HPP
#define VRFHIP_HPP

#include <array>
#include <cstdint>
#include <cstring>
#include <optional>
#include <random>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "vrfhip.h"

namespace ark_vrf_hip {

using Bytes32 = std::array<uint8_t, 32>;
using Scalar = Bytes32;                       // `ScalarField`, 32 bytes canonical: little-endian (ArkworksCodec), big-endian on secp256r1 (Sec1Codec)
using Bytes = std::vector<uint8_t>;

// `Error` (src/lib.rs:15)
enum class Error { VerificationFailure = VRFHIP_ST_VERIFICATION_FAILURE, InvalidData = VRFHIP_ST_INVALID_DATA };
using Result = std::optional<Error>;          // std::nullopt == Ok(())

inline Result result_of(uint8_t status) {
  if (status == VRFHIP_ST_OK) return std::nullopt;
  return status == VRFHIP_ST_VERIFICATION_FAILURE ? Error::VerificationFailure : Error::InvalidData;
}

// `Suite` (src/lib.rs:16): compile-time suite tags
// POINT_LEN / HASH_LEN: bytes of an encoded point (`Suite::Codec`) and of `Output::hash()` (the suite's hash); EDWARDS:
// the twisted-Edwards suites have the key sets and the single-MSM Pedersen verifier, secp256r1 verifies per proof.
struct EdwardsSha512 {
  static constexpr size_t POINT_LEN = 32, HASH_LEN = 64;
  static constexpr bool EDWARDS = true;
};
struct BandersnatchSha512Ell2 : EdwardsSha512 {
  static constexpr vrfhip_suite ID = VRFHIP_SUITE_BANDERSNATCH_SHA512_ELL2;
  static constexpr const char* SUITE_ID = "Bandersnatch_SHA-512_ELL2";
};
struct JubJubSha512Tai : EdwardsSha512 {
  static constexpr vrfhip_suite ID = VRFHIP_SUITE_JUBJUB_SHA512_TAI;
  static constexpr const char* SUITE_ID = "JubJub_SHA-512_TAI";
};
struct Ed25519Sha512Tai : EdwardsSha512 {
  static constexpr vrfhip_suite ID = VRFHIP_SUITE_ED25519_SHA512_TAI;
  static constexpr const char* SUITE_ID = "Ed25519_SHA-512_TAI";
};
struct BabyJubJubSha512Tai : EdwardsSha512 {
  static constexpr vrfhip_suite ID = VRFHIP_SUITE_BABY_JUBJUB_SHA512_TAI;
  static constexpr const char* SUITE_ID = "BabyJubJub_SHA-512_TAI";
};
// `suites::secp256r1` (upstream "P256_SHA256_TAI", RFC 9381 ECVRF-P256-SHA256-TAI): Sec1Codec -- 33-byte compressed
// points, big-endian scalars (`Proof::c` has 16 significant bytes), SHA-256
struct Secp256r1Sha256Tai {
  static constexpr vrfhip_suite ID = VRFHIP_SUITE_SECP256R1_SHA256_TAI;
  static constexpr const char* SUITE_ID = "\x01";
  static constexpr size_t POINT_LEN = 33, HASH_LEN = 32;
  static constexpr bool EDWARDS = false;
};
// `suites::bandersnatch_sw` (upstream "Bandersnatch_SW_SHA-512_TAI"): the Bandersnatch group on its short-Weierstrass model;
// ArkworksCodec over SWAffine -- 33-byte compressed points (x little-endian, then a flag byte), little-endian scalars, SHA-512.
struct BandersnatchSwSha512Tai {
  static constexpr vrfhip_suite ID = VRFHIP_SUITE_BANDERSNATCH_SW_SHA512_TAI;
  static constexpr const char* SUITE_ID = "Bandersnatch_SW_SHA-512_TAI";
  static constexpr size_t POINT_LEN = 33, HASH_LEN = 64;
  static constexpr bool EDWARDS = false;
};
template <class S> using Point = std::array<uint8_t, S::POINT_LEN>;     // an encoded point of suite S
template <class S> using Hash = std::array<uint8_t, S::HASH_LEN>;

// API / runtime failure of the library (negative vrfhip_error): not a per-item outcome
struct ApiError : std::runtime_error {
  int code;
  ApiError(int c, const std::string& what) : std::runtime_error(what + ": " + vrfhip_last_error()), code(c) {}
};
inline void check(int32_t rc, const char* what) {
  if (rc != VRFHIP_SUCCESS) throw ApiError(rc, what);
}

// one GPU, one suite: owns the device tables and workspace
template <class S>
class Context {
 public:
  explicit Context(int device = 0) { check(vrfhip_ctx_create(S::ID, device, &h_), "vrfhip_ctx_create"); }
  // a suite given as data (`Suite::SUITE_ID`, generator, `PedersenSuite::BLINDING_BASE`, hash-to-curve DST)
  Context(const vrfhip_suite_desc& desc, int device) { check(vrfhip_ctx_create_desc(&desc, device, &h_), "vrfhip_ctx_create_desc"); }
  static vrfhip_suite_desc default_descriptor() {
    vrfhip_suite_desc d;
    check(vrfhip_suite_desc_default(S::ID, &d), "vrfhip_suite_desc_default");
    return d;
  }
  // tests only: the default descriptor with the placeholder blinding base of vrfhip_test_blinding_base (upstream's
  // `PedersenSuite::BLINDING_BASE` is pinned for Bandersnatch alone; the other default descriptors carry none)
  static vrfhip_suite_desc test_descriptor() {
    vrfhip_suite_desc d = default_descriptor();
    check(vrfhip_test_blinding_base(S::ID, d.blinding_base), "vrfhip_test_blinding_base");
    return d;
  }
  // point classes whose subgroup membership the caller vouches for (typed, already validated values):
  // VRFHIP_FLAG_PREVALIDATED_*; default 0 = arkworks' checked deserialisation inside every verify
  void set_flags(uint32_t flags) { check(vrfhip_ctx_set_flags(h_, flags), "vrfhip_ctx_set_flags"); }
  uint32_t flags() const { return vrfhip_ctx_get_flags(h_); }
  ~Context() { vrfhip_ctx_destroy(h_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  vrfhip_ctx* handle() const { return h_; }

 private:
  vrfhip_ctx* h_ = nullptr;
};

template <class S> struct Public { Point<S> encoded; };    // `Public`: compressed point (`Suite::Codec`)

// A set of `Public` keys whose validated points and fixed-base tables stay resident in HBM (881 KB per key):
// the verifier of a validator set builds it once and names keys by index afterwards (ietf::verify_batch_keyed).
template <class S>
class KeySet {
 public:
  KeySet(const Context<S>& ctx, const std::vector<Public<S>>& keys) : valid_(keys.size()) {
    constexpr size_t W = S::POINT_LEN;                    // every suite has key sets (secp256r1, bandersnatch_sw: round 4)
    Bytes flat(keys.size() * W), st(keys.size());
    for (size_t i = 0; i < keys.size(); ++i) std::memcpy(flat.data() + W * i, keys[i].encoded.data(), W);
    check(vrfhip_keyset_create(ctx.handle(), keys.size(), flat.data(), st.data(), &h_), "vrfhip_keyset_create");
    for (size_t i = 0; i < keys.size(); ++i) valid_[i] = st[i] == VRFHIP_ST_OK;
  }
  ~KeySet() { vrfhip_keyset_destroy(h_); }
  KeySet(const KeySet&) = delete;
  KeySet& operator=(const KeySet&) = delete;
  const vrfhip_keyset* handle() const { return h_; }
  size_t size() const { return valid_.size(); }
  bool valid(size_t i) const { return valid_[i]; }          // key i is a point of the prime-order subgroup
  size_t bytes() const { return vrfhip_keyset_bytes(h_); }

 private:
  vrfhip_keyset* h_ = nullptr;
  std::vector<bool> valid_;
};
template <class S> struct Output;

template <class S>
struct Input {                                              // `Input`
  Point<S> encoded;
  // `Input::new(data)`: hash-to-curve (Elligator 2 / try-and-increment); None never happens for these suites
  static std::optional<Input> new_(const Context<S>& ctx, const Bytes& data) {
    Input in;
    check(vrfhip_hash_to_curve_batch(ctx.handle(), 1, data.empty() ? in.encoded.data() : data.data(), nullptr,
                                     (uint32_t)data.size(), in.encoded.data()), "vrfhip_hash_to_curve_batch");
    return in;
  }
};

template <class S>
struct Output {                                             // `Output`
  Point<S> encoded;
  Hash<S> hash(const Context<S>& ctx) const {                // `Output::hash()`
    Hash<S> h;
    check(vrfhip_output_hash_batch(ctx.handle(), 1, encoded.data(), h.data()), "vrfhip_output_hash_batch");
    return h;
  }
};

template <class S>
struct Secret {                                             // `Secret`
  Scalar scalar;
  Point<S> pk;
  static Secret from_seed(const Context<S>& ctx, const Bytes& seed) {          // `Secret::from_seed`
    Secret s;
    uint8_t dummy = 0;
    check(vrfhip_secret_from_seed_batch(ctx.handle(), 1, seed.empty() ? &dummy : seed.data(), (uint32_t)seed.size(),
                                        s.scalar.data(), s.pk.data()), "vrfhip_secret_from_seed_batch");
    return s;
  }
  static Secret from_scalar(const Context<S>& ctx, const Scalar& scalar);      // `Secret::from_scalar` (throws unless scalar < order)
  Public<S> public_key() const { return Public<S>{pk}; }                       // `Secret::public`
  Output<S> output(const Context<S>& ctx, const Input<S>& in) const;           // `Secret::output`
};

namespace detail {
inline const uint8_t* ad_ptr(const Bytes& ad) {
  static const uint8_t zero = 0;
  return ad.empty() ? &zero : ad.data();
}
template <class T, class F>
Bytes column(const std::vector<T>& v, F field) {          // one field of every item, back to back (32- or 33-byte arrays)
  Bytes out;
  for (const T& t : v) {
    const auto& a = field(t);
    out.insert(out.end(), a.begin(), a.end());
  }
  if (out.empty()) out.push_back(0);
  return out;
}
template <class A>
void take(A& dst, const Bytes& flat, size_t i) { std::memcpy(dst.data(), flat.data() + dst.size() * i, dst.size()); }
}  // namespace detail

// -------------------------------------------------------------------------------- `ietf` (src/lib.rs:14)
namespace ietf {
template <class S> struct Proof { Scalar c, s; };

template <class S>
struct Item { Public<S> pub; Input<S> input; Output<S> output; Proof<S> proof; };

// `ietf::Prover::prove`
template <class S>
Proof<S> prove(const Context<S>& ctx, const Secret<S>& sk, const Input<S>& in, const Output<S>& out, const Bytes& ad) {
  Proof<S> p;
  Point<S> gamma;
  uint8_t st = 0;
  check(vrfhip_ietf_prove_batch(ctx.handle(), 1, sk.scalar.data(), nullptr, nullptr, 0, in.encoded.data(),
                                detail::ad_ptr(ad), nullptr, (uint32_t)ad.size(), gamma.data(), p.c.data(), p.s.data(),
                                nullptr, nullptr, &st), "vrfhip_ietf_prove_batch");
  if (st != VRFHIP_ST_OK || gamma != out.encoded) throw std::invalid_argument("ietf::prove: secret/input/output mismatch");
  return p;
}
// `ietf::Verifier::verify`
template <class S>
Result verify(const Context<S>& ctx, const Public<S>& pub, const Input<S>& in, const Output<S>& out, const Bytes& ad,
              const Proof<S>& p) {
  uint8_t st = 0;
  check(vrfhip_ietf_verify_batch(ctx.handle(), 1, pub.encoded.data(), in.encoded.data(), out.encoded.data(),
                                 p.c.data(), p.s.data(), detail::ad_ptr(ad), nullptr, (uint32_t)ad.size(), &st),
        "vrfhip_ietf_verify_batch");
  return result_of(st);
}
// n x verify, one launch group
template <class S>
std::vector<Result> verify_batch(const Context<S>& ctx, const std::vector<Item<S>>& items, const Bytes& ad) {
  const size_t n = items.size();
  Bytes pk = detail::column(items, [](const Item<S>& t) -> const auto& { return t.pub.encoded; });
  Bytes h = detail::column(items, [](const Item<S>& t) -> const auto& { return t.input.encoded; });
  Bytes g = detail::column(items, [](const Item<S>& t) -> const auto& { return t.output.encoded; });
  Bytes c = detail::column(items, [](const Item<S>& t) -> const auto& { return t.proof.c; });
  Bytes s = detail::column(items, [](const Item<S>& t) -> const auto& { return t.proof.s; });
  Bytes st(n);
  check(vrfhip_ietf_verify_batch(ctx.handle(), n, pk.data(), h.data(), g.data(), c.data(), s.data(), detail::ad_ptr(ad),
                                 nullptr, (uint32_t)ad.size(), st.data()), "vrfhip_ietf_verify_batch");
  std::vector<Result> r(n);
  for (size_t i = 0; i < n; ++i) r[i] = result_of(st[i]);
  return r;
}
// n x (Input::new(alpha) + verify): what a verifier holding public keys, messages and proofs runs; the inputs are hashed
// to the curve on the GPU and stay there (items[i].input is not read).
template <class S>
std::vector<Result> verify_batch_from_alpha(const Context<S>& ctx, const std::vector<Item<S>>& items,
                                            const std::vector<Bytes>& alphas, const Bytes& ad) {
  const size_t n = items.size();
  if (alphas.size() != n) throw std::invalid_argument("verify_batch_from_alpha: ragged batch");
  Bytes pk = detail::column(items, [](const Item<S>& t) -> const auto& { return t.pub.encoded; });
  Bytes g = detail::column(items, [](const Item<S>& t) -> const auto& { return t.output.encoded; });
  Bytes c = detail::column(items, [](const Item<S>& t) -> const auto& { return t.proof.c; });
  Bytes s = detail::column(items, [](const Item<S>& t) -> const auto& { return t.proof.s; });
  Bytes blob, st(n + 1);
  std::vector<uint32_t> off(n + 1, 0);
  for (size_t i = 0; i < n; ++i) { blob.insert(blob.end(), alphas[i].begin(), alphas[i].end()); off[i + 1] = (uint32_t)blob.size(); }
  blob.push_back(0);
  check(vrfhip_ietf_verify_batch_alpha(ctx.handle(), n, pk.data(), blob.data(), off.data(), 0, g.data(), c.data(), s.data(),
                                       detail::ad_ptr(ad), nullptr, (uint32_t)ad.size(), st.data()),
        "vrfhip_ietf_verify_batch_alpha");
  std::vector<Result> r(n);
  for (size_t i = 0; i < n; ++i) r[i] = result_of(st[i]);
  return r;
}
// n x verify over several contexts (one per GPU): context g verifies items [g*n/G, (g+1)*n/G) on its own host
// thread; items are independent, so the only merge is the concatenation of the results.
template <class S>
std::vector<Result> verify_batch_sharded(const std::vector<const Context<S>*>& ctxs, const std::vector<Item<S>>& items,
                                         const Bytes& ad) {
  const size_t n = items.size(), G = ctxs.size();
  if (G == 0) throw std::invalid_argument("verify_batch_sharded: no context");
  std::vector<Result> r(n);
  std::vector<std::exception_ptr> err(G);
  std::vector<std::thread> th;
  for (size_t g = 0; g < G; ++g)
    th.emplace_back([&, g] {
      try {
        const size_t lo = n * g / G, hi = n * (g + 1) / G;
        std::vector<Item<S>> part(items.begin() + lo, items.begin() + hi);
        std::vector<Result> pr = verify_batch(*ctxs[g], part, ad);
        for (size_t i = lo; i < hi; ++i) r[i] = pr[i - lo];
      } catch (...) { err[g] = std::current_exception(); }
    });
  for (auto& t : th) t.join();
  for (auto& e : err) if (e) std::rethrow_exception(e);
  return r;
}
// n x verify against keys of a KeySet: key_index[i] names the key of items[i] (items[i].pub is not read)
template <class S>
std::vector<Result> verify_batch_keyed(const Context<S>& ctx, const KeySet<S>& keys, const std::vector<uint32_t>& key_index,
                                       const std::vector<Item<S>>& items, const Bytes& ad) {
  const size_t n = items.size();
  if (key_index.size() != n) throw std::invalid_argument("verify_batch_keyed: ragged batch");
  Bytes h = detail::column(items, [](const Item<S>& t) -> const auto& { return t.input.encoded; });
  Bytes g = detail::column(items, [](const Item<S>& t) -> const auto& { return t.output.encoded; });
  Bytes c = detail::column(items, [](const Item<S>& t) -> const auto& { return t.proof.c; });
  Bytes s = detail::column(items, [](const Item<S>& t) -> const auto& { return t.proof.s; });
  Bytes st(n);
  check(vrfhip_ietf_verify_batch_keyed(ctx.handle(), keys.handle(), n, key_index.data(), h.data(), g.data(), c.data(),
                                       s.data(), detail::ad_ptr(ad), nullptr, (uint32_t)ad.size(), st.data()),
        "vrfhip_ietf_verify_batch_keyed");
  std::vector<Result> r(n);
  for (size_t i = 0; i < n; ++i) r[i] = result_of(st[i]);
  return r;
}
// n x (Input::new + Secret::output + prove) from fixed-length messages
template <class S>
std::vector<Item<S>> prove_batch(const Context<S>& ctx, const std::vector<Secret<S>>& sks, const std::vector<Bytes>& msgs,
                                 const Bytes& ad) {
  const size_t n = sks.size();
  if (msgs.size() != n) throw std::invalid_argument("prove_batch: ragged batch");
  Bytes sk = detail::column(sks, [](const Secret<S>& t) -> const auto& { return t.scalar; });
  Bytes blob;
  std::vector<uint32_t> off(n + 1, 0);
  for (size_t i = 0; i < n; ++i) { blob.insert(blob.end(), msgs[i].begin(), msgs[i].end()); off[i + 1] = (uint32_t)blob.size(); }
  blob.push_back(0);
  constexpr size_t W = S::POINT_LEN;
  Bytes g(W * n + 1), c(32 * n + 1), s(32 * n + 1), pk(W * n + 1), h(W * n + 1), st(n + 1);
  check(vrfhip_ietf_prove_batch(ctx.handle(), n, sk.data(), blob.data(), off.data(), 0, nullptr, detail::ad_ptr(ad), nullptr,
                                (uint32_t)ad.size(), g.data(), c.data(), s.data(), pk.data(), h.data(), st.data()),
        "vrfhip_ietf_prove_batch");
  std::vector<Item<S>> items(n);
  for (size_t i = 0; i < n; ++i) {
    if (st[i] != VRFHIP_ST_OK) throw std::invalid_argument("prove_batch: invalid secret");
    detail::take(items[i].pub.encoded, pk, i);
    detail::take(items[i].input.encoded, h, i);
    detail::take(items[i].output.encoded, g, i);
    detail::take(items[i].proof.c, c, i);
    detail::take(items[i].proof.s, s, i);
  }
  return items;
}
}  // namespace ietf

template <class S>
Secret<S> Secret<S>::from_scalar(const Context<S>& ctx, const Scalar& scalar) {
  Secret s;
  s.scalar = scalar;
  Point<S> gamma;
  Scalar c, sv;
  uint8_t st = 0, msg = 0;                       // the public key comes back beside a proof over a one-byte message
  check(vrfhip_ietf_prove_batch(ctx.handle(), 1, scalar.data(), &msg, nullptr, 1, nullptr, detail::ad_ptr({}), nullptr, 0,
                                gamma.data(), c.data(), sv.data(), s.pk.data(), nullptr, &st), "vrfhip_ietf_prove_batch");
  if (st != VRFHIP_ST_OK) throw std::invalid_argument("Secret::from_scalar: not a canonical scalar");
  return s;
}

template <class S>
Output<S> Secret<S>::output(const Context<S>& ctx, const Input<S>& in) const {
  Output<S> out;
  Scalar c, s;
  uint8_t st = 0;
  check(vrfhip_ietf_prove_batch(ctx.handle(), 1, scalar.data(), nullptr, nullptr, 0, in.encoded.data(), detail::ad_ptr({}),
                                nullptr, 0, out.encoded.data(), c.data(), s.data(), nullptr, nullptr, &st),
        "vrfhip_ietf_prove_batch");
  if (st != VRFHIP_ST_OK) throw std::invalid_argument("Secret::output: invalid secret or input");
  return out;
}

// -------------------------------------------------------------------------------- `pedersen` (src/lib.rs:14)
namespace pedersen {
template <class S> struct Proof { Point<S> pk_com, r, ok; Scalar s, sb; };
template <class S> struct Item { Input<S> input; Output<S> output; Proof<S> proof; };

// `pedersen::Prover::prove` -> (proof, blinding factor)
template <class S>
std::pair<Proof<S>, Scalar> prove(const Context<S>& ctx, const Secret<S>& sk, const Input<S>& in, const Output<S>& out,
                                  const Bytes& ad) {
  Proof<S> p;
  Scalar blinding;
  Point<S> gamma;
  uint8_t st = 0;
  check(vrfhip_pedersen_prove_batch(ctx.handle(), 1, sk.scalar.data(), nullptr, nullptr, 0, in.encoded.data(),
                                    detail::ad_ptr(ad), nullptr, (uint32_t)ad.size(), gamma.data(), p.pk_com.data(),
                                    p.r.data(), p.ok.data(), p.s.data(), p.sb.data(), blinding.data(), nullptr, &st),
        "vrfhip_pedersen_prove_batch");
  if (st != VRFHIP_ST_OK || gamma != out.encoded) throw std::invalid_argument("pedersen::prove: secret/input/output mismatch");
  return {p, blinding};
}
// `pedersen::Verifier::verify`
template <class S>
Result verify(const Context<S>& ctx, const Input<S>& in, const Output<S>& out, const Bytes& ad, const Proof<S>& p) {
  uint8_t st = 0;
  check(vrfhip_pedersen_verify_batch(ctx.handle(), 1, in.encoded.data(), out.encoded.data(), p.pk_com.data(), p.r.data(),
                                     p.ok.data(), p.s.data(), p.sb.data(), detail::ad_ptr(ad), nullptr, (uint32_t)ad.size(),
                                     &st), "vrfhip_pedersen_verify_batch");
  return result_of(st);
}
// n x verify through ONE multi-scalar multiplication (random linear combination); a failed batch is re-checked
// per proof inside the library, so the results are those of n calls of verify().  *fast_path (optional) tells
// whether the single MSM sufficed.  secp256r1: one launch group of per-proof checks (*fast_path = false).
template <class S>
std::vector<Result> verify_batch(const Context<S>& ctx, const std::vector<Item<S>>& items, const Bytes& ad,
                                 bool* fast_path = nullptr) {
  const size_t n = items.size();
  Bytes h = detail::column(items, [](const Item<S>& t) -> const auto& { return t.input.encoded; });
  Bytes g = detail::column(items, [](const Item<S>& t) -> const auto& { return t.output.encoded; });
  Bytes pc = detail::column(items, [](const Item<S>& t) -> const auto& { return t.proof.pk_com; });
  Bytes r = detail::column(items, [](const Item<S>& t) -> const auto& { return t.proof.r; });
  Bytes ok = detail::column(items, [](const Item<S>& t) -> const auto& { return t.proof.ok; });
  Bytes s = detail::column(items, [](const Item<S>& t) -> const auto& { return t.proof.s; });
  Bytes sb = detail::column(items, [](const Item<S>& t) -> const auto& { return t.proof.sb; });
  Bytes st(n + 1);
  int32_t fast = 1;
  {
    // every suite has the single-MSM verifier (secp256r1 since round 4)
    std::array<uint8_t, 32> seed;
    std::random_device rd;                                 // must be unpredictable to the provers
    for (auto& b : seed) b = (uint8_t)rd();
    check(vrfhip_pedersen_verify_batch_rlc(ctx.handle(), n, h.data(), g.data(), pc.data(), r.data(), ok.data(), s.data(),
                                           sb.data(), detail::ad_ptr(ad), nullptr, (uint32_t)ad.size(), seed.data(), st.data(),
                                           &fast), "vrfhip_pedersen_verify_batch_rlc");
  }
  if (fast_path) *fast_path = fast != 0;
  std::vector<Result> res(n);
  for (size_t i = 0; i < n; ++i) res[i] = result_of(st[i]);
  return res;
}
}  // namespace pedersen

// -------------------------------------------------------------------------------- `utils::te_sw_map` (src/lib.rs:14)
namespace utils {
using XY = std::array<uint8_t, 64>;           // an affine point as x || y, 32-byte little-endian canonical integers
namespace detail {
template <class S>
std::vector<std::optional<XY>> te_sw_map(const Context<S>& ctx, const std::vector<XY>& pts, int32_t to_te) {
  static_assert(S::EDWARDS || S::ID == VRFHIP_SUITE_BANDERSNATCH_SW_SHA512_TAI, "te_sw_map: the suite's curve has no twisted-Edwards model");
  const size_t n = pts.size();
  Bytes in = ark_vrf_hip::detail::column(pts, [](const XY& t) -> const XY& { return t; }), out(64 * n + 1), st(n + 1);
  check(vrfhip_te_sw_map_batch(ctx.handle(), n, to_te, in.data(), out.data(), st.data()), "vrfhip_te_sw_map_batch");
  std::vector<std::optional<XY>> r(n);
  for (size_t i = 0; i < n; ++i)
    if (st[i] == VRFHIP_ST_OK) { r[i].emplace(); ark_vrf_hip::detail::take(*r[i], out, i); }
  return r;
}
}  // namespace detail
// `te_to_sw`: points of S's twisted-Edwards curve -> the short-Weierstrass form of its Montgomery model; nullopt = None
template <class S>
std::vector<std::optional<XY>> te_to_sw(const Context<S>& ctx, const std::vector<XY>& pts) { return detail::te_sw_map(ctx, pts, 0); }
// `sw_to_te`: the inverse map
template <class S>
std::vector<std::optional<XY>> sw_to_te(const Context<S>& ctx, const std::vector<XY>& pts) { return detail::te_sw_map(ctx, pts, 1); }
}  // namespace utils

}  // namespace ark_vrf_hip
#endif  // VRFHIP_HPP
