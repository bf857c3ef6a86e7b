"""Verification from (pk, alpha, proof) -- `Input::new(alpha)` + `ietf::Verifier::verify` (/root/reference src/lib.rs:14-16)
as one call, `vrfhip_ietf_verify_batch_alpha`: H is hashed to the curve on the device and is never compressed, decompressed
or subgroup-tested.  GPU tier: statuses equal (a) the C oracle's verify on the oracle's own H and (b) the library's two-call
form (hash_to_curve_batch, then ietf_verify_batch), on every twisted-Edwards suite, with every kind of defect including a
wrong message; ragged messages with per-item ad; launch groups of several sizes; secp256r1 (the two stages inside one call)."""
import os

import numpy as np
import pytest

from oracle import c_oracle as co

NCPU = min(8, os.cpu_count() or 1)
SUITES = {"bandersnatch": ("BandersnatchSha512Ell2", 1), "jubjub": ("JubJubSha512Tai", 2), "ed25519": ("Ed25519Sha512Tai", 3),
          "babyjubjub": ("BabyJubJubSha512Tai", 4)}


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(SUITES))
def test_gpu_verify_from_alpha_equals_oracle_and_two_call_form(name):
    import ark_ec_vrfs_amd as pkg
    cls, sid = SUITES[name]
    ctx = pkg.Context(0, getattr(pkg, cls), test_blinding_base=True)
    co.set_suite(sid)
    try:
        rng = np.random.default_rng(sid)
        n = 3000
        seeds = np.arange(n, dtype=np.uint64).view(np.uint8).reshape(n, 8)
        sk, _ = ctx.secret_from_seed_batch(seeds)
        msg = rng.integers(0, 256, (n, 40), dtype=np.uint8)
        ad = b"alpha-test"
        ref = co.ietf_prove_batch(sk, msgs=msg, ad=ad, threads=NCPU)
        assert (ctx.ietf_verify_batch_alpha(ref["pk"], msg, ref["output"], ref["c"], ref["s"], ad=ad) == 0).all()
        pk, out, c, s, m2 = (x.copy() for x in (ref["pk"], ref["output"], ref["c"], ref["s"], msg))
        m2[::11, 7] ^= 1                                        # another message: another H
        s[3::13, 1] ^= 2                                        # wrong s
        c[5::17, 0] ^= 1                                        # wrong c
        s[7::101] = 0xFF                                        # s not canonical
        out[9::19] = ref["output"][10::19][: len(out[9::19])]   # another proof's output
        pk[2::23, 0] ^= 1                                       # a public key that (mostly) does not decode / is off the subgroup
        out[4::29, 3] ^= 4                                      # an output that (mostly) does not decode
        h2 = ctx.hash_to_curve_batch(m2)
        want = co.ietf_verify_batch(pk, h2, out, c, s, ad, threads=NCPU)
        assert (h2[1::11] == ref["input"][1::11]).all() and set(np.unique(want)) == {0, 1, 2}
        got = ctx.ietf_verify_batch_alpha(pk, m2, out, c, s, ad=ad)
        assert (got == want).all()
        assert (got == ctx.ietf_verify_batch(pk, h2, out, c, s, ad=ad)).all()
        # the validation flags still govern pk and the output: with everything declared pre-validated the verdicts are those
        # of the two-call form under the same flags
        ctx.set_flags(pkg.Context.PREVALIDATED_ALL)
        assert (ctx.ietf_verify_batch_alpha(pk, m2, out, c, s, ad=ad) == ctx.ietf_verify_batch(pk, h2, out, c, s, ad=ad)).all()
        ctx.set_flags(0)
        # ragged messages, per-item ad, group sizes around the lane packing
        for m in (1, 2, 63, 64, 65, 257):
            msgs = [bytes(rng.integers(0, 256, int(rng.integers(0, 90)), dtype=np.uint8)) for _ in range(m)]
            ads = [bytes(rng.integers(0, 256, int(rng.integers(0, 20)), dtype=np.uint8)) for _ in range(m)]
            pr = ctx.ietf_prove_batch(sk[:m], msgs=msgs, ad=ads)
            assert (ctx.ietf_verify_batch_alpha(pr["pk"], msgs, pr["output"], pr["c"], pr["s"], ad=ads) == 0).all()
            if m > 2:
                msgs[1] = msgs[1] + b"!"
                st = ctx.ietf_verify_batch_alpha(pr["pk"], msgs, pr["output"], pr["c"], pr["s"], ad=ads)
                assert st[1] == 1 and st.sum() == 1
        assert ctx.ietf_verify_batch_alpha(np.zeros((0, 32), np.uint8), [], np.zeros((0, 32), np.uint8), np.zeros((0, 32), np.uint8),
                                           np.zeros((0, 32), np.uint8)).shape == (0,)
    finally:
        co.set_suite(1)
        ctx.close()


@pytest.mark.gpu
def test_gpu_verify_from_alpha_on_device_pointers():
    import torch
    import ark_ec_vrfs_amd as pkg
    ctx = pkg.Context(0)
    n = (1 << 17) + 77                                          # past the fused-Straus threshold, ragged tail
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(17)
    seeds = np.arange(n, dtype=np.uint64).view(np.uint8).reshape(n, 8)
    sk_h, pk_h = ctx.secret_from_seed_batch(seeds)
    msg_h = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    sk, pk, msg = up(sk_h), up(pk_h), up(msg_h)
    z = lambda w=32: torch.empty((n, w), dtype=torch.uint8, device=dev)
    out, c, s, h = z(), z(), z(), z()
    st = torch.empty(n, dtype=torch.uint8, device=dev)
    ctx.ietf_prove_batch_dev(sk, msg, 32, out, c, s, input_out=h, status=st)
    torch.cuda.synchronize()
    assert int(st.sum()) == 0
    ctx.ietf_verify_batch_alpha_dev(pk, msg, 32, out, c, s, st)
    torch.cuda.synchronize()
    assert int(st.sum()) == 0
    s2 = s.clone(); s2[::997, 4] ^= 1
    msg2_h = msg_h.copy(); msg2_h[5::1009, 0] ^= 1
    ctx.ietf_verify_batch_alpha_dev(pk, up(msg2_h), 32, out, c, s2, st)
    h2 = up(ctx.hash_to_curve_batch(msg2_h))
    st2 = torch.empty_like(st)
    ctx.ietf_verify_batch_dev(pk, h2, out, c, s2, st2)
    torch.cuda.synchronize()
    assert bool((st == st2).all()) and int((st == 1).sum()) == len(range(0, n, 997)) + len(range(5, n, 1009))
    ctx.close()


@pytest.mark.gpu
def test_gpu_verify_from_alpha_secp256r1_equals_the_oracle():
    """secp256r1 (cofactor 1: nothing to skip): hash-to-curve and verify inside the one call; statuses equal the C oracle's
    on its own H, and hash_to_curve_batch (now through the work-queue counter search) equals the oracle's H."""
    import ark_ec_vrfs_amd as pkg
    ctx = pkg.Context(0, pkg.Secp256r1Sha256Tai, test_blinding_base=True)
    rng = np.random.default_rng(256)
    n = 2500
    seeds = np.arange(n, dtype=np.uint64).view(np.uint8).reshape(n, 8)
    sk, _ = ctx.secret_from_seed_batch(seeds)
    msg = rng.integers(0, 256, (n, 37), dtype=np.uint8)
    ref = co.p256_ietf_prove_batch(sk, msgs=msg, ad=b"p", threads=NCPU)
    assert (ctx.hash_to_curve_batch(msg) == ref["input"]).all()
    assert (ctx.ietf_verify_batch_alpha(ref["pk"], msg, ref["output"], ref["c"], ref["s"], ad=b"p") == 0).all()
    pk, out, c, s, m2 = (x.copy() for x in (ref["pk"], ref["output"], ref["c"], ref["s"], msg))
    m2[::11, 3] ^= 1
    s[3::13, 30] ^= 2
    c[5::17, 31] ^= 1
    s[7::101] = 0xFF
    pk[2::23, 0] = 5
    out[4::29, 9] ^= 4
    h2 = ctx.hash_to_curve_batch(m2)
    assert (h2[1::11] == ref["input"][1::11]).all()
    want = co.p256_ietf_verify_batch(pk, h2, out, c, s, ad=b"p", threads=NCPU)
    got = ctx.ietf_verify_batch_alpha(pk, m2, out, c, s, ad=b"p")
    assert (got == want).all() and set(np.unique(want)) == {0, 1, 2}
    msgs = [bytes(rng.integers(0, 256, int(rng.integers(0, 70)), dtype=np.uint8)) for _ in range(130)]
    pr = ctx.ietf_prove_batch(sk[:130], msgs=msgs, ad=b"")
    assert (ctx.ietf_verify_batch_alpha(pr["pk"], msgs, pr["output"], pr["c"], pr["s"]) == 0).all()
    ctx.close()
