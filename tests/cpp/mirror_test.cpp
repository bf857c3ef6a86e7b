// tests/cpp/mirror_test.cpp -- the C++ host-side mirror (include/vrfhip.hpp) exercised the way upstream's own
// tests drive the Rust API: from_seed -> Input::new -> output -> prove -> verify, against the golden vector
// handed over by tests/test_cpp_mirror.py (hex on the command line), then batches through the GPU.
//   mirror_test <seed> <alpha> <ad> <pk> <h> <gamma> <beta> <c> <s>   <ped_ad> <blinding> <pk_com> <r> <ok> <ps> <psb>
//               <p256_sk> <p256_pk> <p256_alpha> <p256_h> <p256_pi> <p256_beta>       (RFC 9381 B.1, example 10)
// Exit code 0 = every check passed; prints the first failing check otherwise.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "vrfhip.hpp"

using namespace ark_vrf_hip;
using S = BandersnatchSha512Ell2;

static Bytes unhex(const std::string& h) {
  Bytes b(h.size() / 2);
  for (size_t i = 0; i < b.size(); ++i) b[i] = (uint8_t)std::stoul(h.substr(2 * i, 2), nullptr, 16);
  return b;
}
template <class A>
static std::string hex(const A& a) {
  static const char* d = "0123456789abcdef";
  std::string s;
  for (uint8_t v : a) { s.push_back(d[v >> 4]); s.push_back(d[v & 15]); }
  return s;
}
#define CHECK(cond)                                                        \
  do {                                                                     \
    if (!(cond)) { std::printf("FAILED: %s (line %d)\n", #cond, __LINE__); return 1; } \
  } while (0)

// `suites::secp256r1` through the same templates: 33-byte Sec1 points, big-endian scalars, 32-byte hashes.  RFC 9381 passes
// PK_string || alpha to encode_to_curve (the salt), so that is the message here.
static int p256_section(char** a) {
  using P = Secp256r1Sha256Tai;
  const std::string sk = a[0], pk = a[1], alpha = a[2], h = a[3], pi = a[4], beta = a[5];
  {
    // the default descriptor of this suite has no Pedersen blinding base (upstream's is not pinned): the scheme is refused
    Context<P> plain(0);
    bool refused = false;
    try {
      const auto sec0 = Secret<P>::from_seed(plain, Bytes{1});
      const auto in0 = Input<P>::new_(plain, Bytes{1});
      (void)pedersen::prove(plain, sec0, *in0, sec0.output(plain, *in0), Bytes{});
    }
    catch (const ApiError& e) { refused = e.code == VRFHIP_ERR_UNSUPPORTED; }
    CHECK(refused);
  }
  Context<P> ctx(Context<P>::test_descriptor(), 0);
  Scalar skb;
  const Bytes skv = unhex(sk);
  std::copy(skv.begin(), skv.end(), skb.begin());
  const auto secret = Secret<P>::from_scalar(ctx, skb);
  CHECK(hex(secret.public_key().encoded) == pk);
  Bytes msg = unhex(pk), al = unhex(alpha);
  msg.insert(msg.end(), al.begin(), al.end());
  const auto input = Input<P>::new_(ctx, msg);
  CHECK(input.has_value() && hex(input->encoded) == h);
  const auto output = secret.output(ctx, *input);
  CHECK(hex(output.encoded) == pi.substr(0, 66));
  CHECK(hex(output.hash(ctx)) == beta);
  const auto proof = ietf::prove(ctx, secret, *input, output, {});
  CHECK(hex(proof.c) == std::string(32, '0') + pi.substr(66, 32) && hex(proof.s) == pi.substr(98, 64));   // pi = Gamma || c (16 B) || s
  CHECK(!ietf::verify(ctx, secret.public_key(), *input, output, {}, proof).has_value());
  CHECK(ietf::verify(ctx, secret.public_key(), *input, output, Bytes{1}, proof) == Error::VerificationFailure);
  auto bad = proof;
  bad.s.fill(0xff);                                                                                     // s >= n
  CHECK(ietf::verify(ctx, secret.public_key(), *input, output, {}, bad) == Error::InvalidData);
  Scalar big;
  big.fill(0xff);                                                                                       // sk >= n
  bool threw = false;
  try { Secret<P>::from_scalar(ctx, big); } catch (const std::invalid_argument&) { threw = true; }
  CHECK(threw);
  // batches: ragged messages, one launch group; Pedersen per proof (no single-MSM verifier on this suite)
  const size_t n = 600;
  std::vector<Secret<P>> sks;
  std::vector<Bytes> msgs;
  for (size_t i = 0; i < n; ++i) {
    Bytes sd(8);
    for (int k = 0; k < 8; ++k) sd[k] = (uint8_t)(i >> (8 * k));
    sks.push_back(Secret<P>::from_seed(ctx, sd));
    msgs.push_back(Bytes(1 + i % 70, (uint8_t)(i * 3)));
  }
  auto items = ietf::prove_batch(ctx, sks, msgs, Bytes{7, 7});
  for (size_t i = 0; i < n; i += 59) {
    const auto in_i = Input<P>::new_(ctx, msgs[i]);
    CHECK(in_i->encoded == items[i].input.encoded && sks[i].pk == items[i].pub.encoded);
    const auto p_i = ietf::prove(ctx, sks[i], *in_i, sks[i].output(ctx, *in_i), Bytes{7, 7});
    CHECK(p_i.c == items[i].proof.c && p_i.s == items[i].proof.s);
    CHECK(Secret<P>::from_scalar(ctx, sks[i].scalar).pk == sks[i].pk);
  }
  items[5].proof.s[31] ^= 1;
  items[77].pub.encoded[0] = 0x04;                       // not a compressed-point tag
  items[300].output = items[301].output;
  const auto res = ietf::verify_batch(ctx, items, Bytes{7, 7});
  for (size_t i = 0; i < n; ++i) {
    if (i == 5 || i == 300) CHECK(res[i] == Error::VerificationFailure);
    else if (i == 77) CHECK(res[i] == Error::InvalidData);
    else CHECK(!res[i].has_value());
  }
  {
    auto alphas = msgs;
    alphas[9][0] ^= 1;
    const auto ares = ietf::verify_batch_from_alpha(ctx, items, alphas, Bytes{7, 7});
    for (size_t i = 0; i < n; ++i) CHECK(i == 9 ? ares[i] == Error::VerificationFailure : ares[i] == res[i]);
  }
  std::vector<pedersen::Item<P>> pitems;
  for (size_t i = 0; i < 40; ++i) {
    const auto in_i = Input<P>::new_(ctx, msgs[i]);
    const auto out_i = sks[i].output(ctx, *in_i);
    pitems.push_back({*in_i, out_i, pedersen::prove(ctx, sks[i], *in_i, out_i, Bytes{9}).first});
    CHECK(!pedersen::verify(ctx, *in_i, out_i, Bytes{9}, pitems.back().proof).has_value());
  }
  pitems[11].proof.sb[30] ^= 2;
  bool fast = true;
  const auto pres = pedersen::verify_batch(ctx, pitems, Bytes{9}, &fast);
  CHECK(!fast);
  for (size_t i = 0; i < pitems.size(); ++i) CHECK(i == 11 ? pres[i] == Error::VerificationFailure : !pres[i].has_value());
  // the valid batch goes through as ONE multi-scalar multiplication on this suite too (round 4: k_p256_msm.hip)
  pitems[11].proof.sb[30] ^= 2;
  fast = false;
  const auto pres2 = pedersen::verify_batch(ctx, pitems, Bytes{9}, &fast);
  CHECK(fast);
  for (const auto& r2 : pres2) CHECK(!r2.has_value());
  std::printf("mirror_test p256 ok: RFC 9381 B.1, %zu IETF proofs, %zu Pedersen proofs\n", n, pitems.size());
  return 0;
}

// `suites::bandersnatch_sw` through the same templates: 33-byte arkworks short-Weierstrass points.  No vector of this suite
// exists here (tests/test_bandersnatch_sw.py holds the library against its oracle); this is the API shape: round trips,
// tampering, the batched paths.
static int bsw_section() {
  using W = BandersnatchSwSha512Tai;
  Context<W> ctx(Context<W>::test_descriptor(), 0);
  const size_t n = 300;
  std::vector<Secret<W>> sks;
  std::vector<Bytes> msgs;
  for (size_t i = 0; i < n; ++i) {
    Bytes sd(8);
    for (int k = 0; k < 8; ++k) sd[k] = (uint8_t)(i >> (8 * k));
    sks.push_back(Secret<W>::from_seed(ctx, sd));
    msgs.push_back(Bytes(1 + i % 50, (uint8_t)(i * 5)));
  }
  CHECK(sks[0].pk.size() == 33 && (sks[0].pk[32] & 0x3f) == 0);
  auto items = ietf::prove_batch(ctx, sks, msgs, Bytes{1, 2});
  for (size_t i = 0; i < n; i += 37) {
    const auto in_i = Input<W>::new_(ctx, msgs[i]);
    CHECK(in_i->encoded == items[i].input.encoded && sks[i].pk == items[i].pub.encoded);
    const auto out_i = sks[i].output(ctx, *in_i);
    CHECK(out_i.encoded == items[i].output.encoded && out_i.hash(ctx).size() == 64);
    const auto p_i = ietf::prove(ctx, sks[i], *in_i, out_i, Bytes{1, 2});
    CHECK(p_i.c == items[i].proof.c && p_i.s == items[i].proof.s);
    CHECK(!ietf::verify(ctx, sks[i].public_key(), *in_i, out_i, Bytes{1, 2}, p_i).has_value());
  }
  items[4].proof.c[0] ^= 1;
  items[50].pub.encoded[32] = 0xC0;                      // both flags
  items[51].output.encoded[32] |= 0x07;                  // junk below the flags: not looked at
  items[200].output = items[201].output;
  const auto res = ietf::verify_batch(ctx, items, Bytes{1, 2});
  for (size_t i = 0; i < n; ++i) {
    if (i == 4 || i == 200) CHECK(res[i] == Error::VerificationFailure);
    else if (i == 50) CHECK(res[i] == Error::InvalidData);
    else CHECK(!res[i].has_value());
  }
  const auto ares = ietf::verify_batch_from_alpha(ctx, items, msgs, Bytes{1, 2});
  for (size_t i = 0; i < n; ++i) CHECK(ares[i] == res[i]);
  std::vector<pedersen::Item<W>> pitems;
  for (size_t i = 0; i < 48; ++i) {
    const auto in_i = Input<W>::new_(ctx, msgs[i]);
    const auto out_i = sks[i].output(ctx, *in_i);
    pitems.push_back({*in_i, out_i, pedersen::prove(ctx, sks[i], *in_i, out_i, Bytes{9}).first});
    CHECK(!pedersen::verify(ctx, *in_i, out_i, Bytes{9}, pitems.back().proof).has_value());
  }
  bool fast = false;
  for (const auto& r2 : pedersen::verify_batch(ctx, pitems, Bytes{9}, &fast)) CHECK(!r2.has_value());
  CHECK(fast);
  pitems[7].proof.s[3] ^= 8;
  const auto pres = pedersen::verify_batch(ctx, pitems, Bytes{9}, &fast);
  CHECK(!fast);
  for (size_t i = 0; i < pitems.size(); ++i) CHECK(i == 7 ? pres[i] == Error::VerificationFailure : !pres[i].has_value());
  {
    // key sets on this suite too: the items' keys, named by index
    std::vector<Public<W>> keys;
    for (size_t i = 0; i < 8; ++i) keys.push_back(sks[i].public_key());
    KeySet<W> ks(ctx, keys);
    std::vector<Secret<W>> who;
    std::vector<Bytes> what;
    std::vector<uint32_t> idx;
    for (size_t i = 0; i < 64; ++i) { idx.push_back((uint32_t)(i % 8)); who.push_back(sks[i % 8]); what.push_back(msgs[i]); }
    auto kitems = ietf::prove_batch(ctx, who, what, Bytes{3});
    kitems[9].proof.s[0] ^= 1;
    const auto kres = ietf::verify_batch_keyed(ctx, ks, idx, kitems, Bytes{3});
    for (size_t i = 0; i < kitems.size(); ++i) CHECK(i == 9 ? kres[i] == Error::VerificationFailure : !kres[i].has_value());
  }
  std::printf("mirror_test bandersnatch_sw ok: %zu IETF proofs, %zu Pedersen proofs\n", n, pitems.size());
  return 0;
}

int main(int argc, char** argv) {
  if (argc != 17 && argc != 23) { std::printf("usage: mirror_test <16 hex fields> [<6 secp256r1 fields>]\n"); return 2; }
  if (argc == 23 && p256_section(argv + 17)) return 1;
  if (bsw_section()) return 1;
  const std::string seed = argv[1], alpha = argv[2], ad = argv[3], pk = argv[4], h = argv[5], gamma = argv[6],
                    beta = argv[7], c = argv[8], s = argv[9], ped_ad = argv[10], blinding = argv[11], pk_com = argv[12],
                    pr = argv[13], pok = argv[14], ps = argv[15], psb = argv[16];
  Context<S> ctx(0);

  // ---- IETF VRF, single items (upstream: ietf::tests / testing::ietf_prove_verify) ----
  const auto secret = Secret<S>::from_seed(ctx, unhex(seed));
  const auto pub = secret.public_key();
  CHECK(hex(pub.encoded) == pk);
  const auto input = Input<S>::new_(ctx, unhex(alpha));
  CHECK(input.has_value() && hex(input->encoded) == h);
  const auto output = secret.output(ctx, *input);
  CHECK(hex(output.encoded) == gamma);
  CHECK(hex(output.hash(ctx)) == beta);
  const auto proof = ietf::prove(ctx, secret, *input, output, unhex(ad));
  CHECK(hex(proof.c) == c && hex(proof.s) == s);
  CHECK(!ietf::verify(ctx, pub, *input, output, unhex(ad), proof).has_value());                       // Ok(())
  Bytes other_ad = unhex(ad);
  other_ad.push_back(0x5a);
  CHECK(ietf::verify(ctx, pub, *input, output, other_ad, proof) == Error::VerificationFailure);
  auto bad = proof;
  for (auto& b : bad.s) b = 0xff;                                                                      // s >= r
  CHECK(ietf::verify(ctx, pub, *input, output, unhex(ad), bad) == Error::InvalidData);

  // ---- Pedersen VRF, single item (upstream: pedersen::tests) ----
  const auto [pproof, blind] = pedersen::prove(ctx, secret, *input, output, unhex(ped_ad));
  CHECK(hex(blind) == blinding && hex(pproof.pk_com) == pk_com && hex(pproof.r) == pr && hex(pproof.ok) == pok);
  CHECK(hex(pproof.s) == ps && hex(pproof.sb) == psb);
  CHECK(!pedersen::verify(ctx, *input, output, unhex(ped_ad), pproof).has_value());
  CHECK(pedersen::verify(ctx, *input, output, other_ad, pproof) == Error::VerificationFailure);

  // ---- batches ----
  const size_t n = 2000;
  std::vector<Secret<S>> sks;
  std::vector<Bytes> msgs;
  for (size_t i = 0; i < n; ++i) {
    Bytes sd(8);
    for (int k = 0; k < 8; ++k) sd[k] = (uint8_t)(i >> (8 * k));
    sks.push_back(Secret<S>::from_seed(ctx, sd));
    msgs.push_back(Bytes(1 + i % 40, (uint8_t)i));                                                      // ragged messages
  }
  auto items = ietf::prove_batch(ctx, sks, msgs, unhex(ad));
  for (size_t i = 0; i < n; i += 97) {                               // batch == single-item path
    const auto in_i = Input<S>::new_(ctx, msgs[i]);
    CHECK(in_i->encoded == items[i].input.encoded && sks[i].pk == items[i].pub.encoded);
    const auto out_i = sks[i].output(ctx, *in_i);
    CHECK(out_i.encoded == items[i].output.encoded);
    const auto p_i = ietf::prove(ctx, sks[i], *in_i, out_i, unhex(ad));
    CHECK(p_i.c == items[i].proof.c && p_i.s == items[i].proof.s);
  }
  items[7].proof.s[0] ^= 1;
  items[1234].output = items[1235].output;
  for (auto& b : items[1999].proof.c) b = 0xff;         // c >= r: decoded mod r, as `Proof::c` upstream -> a wrong challenge
  for (auto& b : items[1998].proof.s) b = 0xff;         // s >= r: strict -> InvalidData
  const auto res = ietf::verify_batch(ctx, items, unhex(ad));
  for (size_t i = 0; i < n; ++i) {
    if (i == 7 || i == 1234 || i == 1999) CHECK(res[i] == Error::VerificationFailure);
    else if (i == 1998) CHECK(res[i] == Error::InvalidData);
    else CHECK(!res[i].has_value());
  }
  // the same verdicts from (pk, alpha, proof): Input::new inside the call; another message is another H
  {
    auto alphas = msgs;
    alphas[55].push_back(0x21);
    const auto ares = ietf::verify_batch_from_alpha(ctx, items, alphas, unhex(ad));
    CHECK(ares.size() == n);
    for (size_t i = 0; i < n; ++i) CHECK(i == 55 ? ares[i] == Error::VerificationFailure : ares[i] == res[i]);
  }
  // several contexts from one process (one per GPU; here three on the same device): slices tile the batch
  {
    Context<S> c1(0), c2(0);
    const std::vector<const Context<S>*> set = {&ctx, &c1, &c2};
    const auto sres = ietf::verify_batch_sharded(set, items, unhex(ad));
    CHECK(sres.size() == n);
    for (size_t i = 0; i < n; ++i) CHECK(sres[i] == res[i]);
    const std::vector<ietf::Item<S>> few(items.begin(), items.begin() + 2);          // fewer items than contexts
    const auto fres = ietf::verify_batch_sharded(set, few, unhex(ad));
    CHECK(fres.size() == 2 && !fres[0].has_value() && !fres[1].has_value());
  }
  // keyed verification: 16 validators, every proof names its key by index; same verdicts as the plain call
  {
    const size_t nk = 16, m = 640;
    std::vector<Secret<S>> vals(sks.begin(), sks.begin() + nk);
    std::vector<Public<S>> pubs;
    for (const auto& v : vals) pubs.push_back(v.public_key());
    pubs[5].encoded.fill(0); pubs[5].encoded[0] = 3;                       // y = 3: not a curve point
    KeySet<S> keys(ctx, pubs);
    CHECK(keys.size() == nk && !keys.valid(5) && keys.valid(4) && keys.bytes() >= nk * 881280);
    std::vector<Secret<S>> who;
    std::vector<Bytes> what;
    std::vector<uint32_t> idx;
    for (size_t i = 0; i < m; ++i) { idx.push_back((uint32_t)((i * 7) % nk)); who.push_back(vals[idx[i]]); what.push_back(msgs[i]); }
    auto kitems = ietf::prove_batch(ctx, who, what, unhex(ad));
    kitems[100].proof.s[2] ^= 1;
    idx[200] = 99;                                                         // no such key
    const auto kres = ietf::verify_batch_keyed(ctx, keys, idx, kitems, unhex(ad));
    for (size_t i = 0; i < m; ++i) {
      if (i == 200 || idx[i] == 5) CHECK(kres[i] == Error::InvalidData);
      else if (i == 100) CHECK(kres[i] == Error::VerificationFailure);
      else CHECK(!kres[i].has_value());
    }
  }
  // batched Pedersen verification: fast path on a valid batch, per-proof verdicts on a tampered one
  std::vector<pedersen::Item<S>> pitems;
  for (size_t i = 0; i < 300; ++i) {
    const auto in_i = Input<S>::new_(ctx, msgs[i]);
    const auto out_i = sks[i].output(ctx, *in_i);
    pitems.push_back({*in_i, out_i, pedersen::prove(ctx, sks[i], *in_i, out_i, unhex(ped_ad)).first});
  }
  bool fast = false;
  auto pres = pedersen::verify_batch(ctx, pitems, unhex(ped_ad), &fast);
  CHECK(fast);
  for (const auto& r : pres) CHECK(!r.has_value());
  pitems[42].proof.sb[3] ^= 8;
  pres = pedersen::verify_batch(ctx, pitems, unhex(ped_ad), &fast);
  CHECK(!fast);
  for (size_t i = 0; i < pitems.size(); ++i) CHECK(i == 42 ? pres[i] == Error::VerificationFailure : !pres[i].has_value());
  // a suite given as data: a context made from the default descriptor proves the same bytes; flags round trip
  {
    vrfhip_suite_desc d = Context<S>::default_descriptor();
    CHECK(d.curve == VRFHIP_CURVE_BANDERSNATCH && d.suite_id_len == 25 && d.challenge_len == 32);
    // utils::te_sw_map: the generator and the blinding base go to the Weierstrass form and come back; the identity has no image
    utils::XY g, bb, id{};
    std::copy(d.generator, d.generator + 64, g.begin());
    std::copy(d.blinding_base, d.blinding_base + 64, bb.begin());
    id[32] = 1;                                                            // (0, 1)
    const auto sw = utils::te_to_sw(ctx, {g, bb, id});
    CHECK(sw.size() == 3 && sw[0].has_value() && sw[1].has_value() && !sw[2].has_value() && !(*sw[0] == g));
    const auto te = utils::sw_to_te(ctx, {*sw[0], *sw[1]});
    CHECK(te[0].has_value() && *te[0] == g && te[1].has_value() && *te[1] == bb);
    Context<S> from_desc(d, 0);
    const auto a = ietf::prove_batch(ctx, {sks[0], sks[1]}, {msgs[0], msgs[1]}, unhex(ad));
    const auto b = ietf::prove_batch(from_desc, {sks[0], sks[1]}, {msgs[0], msgs[1]}, unhex(ad));
    CHECK(a[0].proof.c == b[0].proof.c && a[1].proof.s == b[1].proof.s && a[1].output.encoded == b[1].output.encoded);
    d.suite_id[0] ^= 1;                                                    // another suite string: another challenge
    Context<S> other(d, 0);
    const auto c2 = ietf::prove_batch(other, {sks[0]}, {msgs[0]}, unhex(ad));
    CHECK(!(c2[0].proof.c == a[0].proof.c) && c2[0].output.encoded == a[0].output.encoded);
    CHECK(ctx.flags() == 0);
    from_desc.set_flags(VRFHIP_FLAG_PREVALIDATED_INPUT | VRFHIP_FLAG_PREVALIDATED_PUBLIC);
    CHECK(from_desc.flags() == 3);
    const auto r = ietf::verify_batch(from_desc, b, unhex(ad));
    CHECK(!r[0].has_value() && !r[1].has_value());
  }
  std::printf("mirror_test ok: KAT, %zu IETF proofs, %zu Pedersen proofs\n", n, pitems.size());
  return 0;
}
