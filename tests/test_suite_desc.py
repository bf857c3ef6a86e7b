"""Suite descriptor (`vrfhip_suite_desc`, [ref src/lib.rs:16 `Suite`, :14 `suites`]): SUITE_ID, the hash-to-curve DST,
the generator and the Pedersen blinding base are DATA supplied by the host; the built-in suites are two pre-filled
descriptors.  CPU: the two oracles agree under a descriptor; the device source (host build) follows it.  GPU: a context
made from the default Bandersnatch descriptor reproduces every golden vector; contexts made from descriptors with a
different suite string / DST / generator / blinding base equal the oracle given the same descriptor, on both curves;
invalid descriptors are refused."""
import ctypes
import dataclasses
import os
import subprocess

import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import vrf_oracle as o

S = o.BANDERSNATCH
HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim")
NCPU = min(16, os.cpu_count() or 1)
le = lambda v: int(v).to_bytes(32, "little")
xy = lambda P: le(P[0]) + le(P[1])


def custom_bandersnatch():
    G2 = o.te_mul(S, 5, (S.gx, S.gy))
    B2 = o.te_mul(S, 777, (S.gx, S.gy))
    return dataclasses.replace(S, suite_id=b"Custom-Bandersnatch/v9", h2c_dst=b"ECVRF_custom_dst_" + bytes(range(48, 48 + 70)),
                               gx=G2[0], gy=G2[1], bx=B2[0], by=B2[1])


def custom_jubjub():
    J = o.jubjub_params()
    B2 = o.te_mul(J, 123456789, (J.gx, J.gy))
    return dataclasses.replace(J, suite_id=b"JubJub_SHA-512_TAI_upstream-like-string", bx=B2[0], by=B2[1])


def oracle_select(Sx):
    co.set_suite_desc(1 if Sx.h2c == "ell2" else 2, Sx.suite_id, Sx.h2c_dst, xy((Sx.gx, Sx.gy)), xy((Sx.bx, Sx.by)))


def _items(n, start):
    sk = np.stack([np.frombuffer(co.secret_from_seed(o.synth_seed(start + i)), np.uint8) for i in range(n)])
    msg = np.stack([np.frombuffer(o.synth_msg(start + i), np.uint8) for i in range(n)])
    return sk, msg


# ------------------------------------------------------------------------------------------------ CPU
@pytest.mark.parametrize("make", [custom_bandersnatch, custom_jubjub])
def test_c_oracle_follows_the_descriptor_like_the_python_oracle(make):
    Sx = make()
    oracle_select(Sx)
    try:
        for i in range(3):
            sk = o.secret_from_seed(Sx, o.synth_seed(i)); msg = o.synth_msg(i); ad = b"d" * i
            H = o.data_to_point(Sx, msg)
            assert co.hash_to_curve(msg) == o.point_encode(Sx, H)
            gamma, c, s = o.ietf_prove(Sx, sk, H, ad)
            skb = np.frombuffer(o.scalar_encode(sk), np.uint8)
            r = co.ietf_prove_batch(skb, msgs=np.frombuffer(msg, np.uint8).reshape(1, -1), ad=ad)
            assert r["output"][0].tobytes() == o.point_encode(Sx, gamma)
            assert r["c"][0].tobytes() == o.scalar_encode(c) and r["s"][0].tobytes() == o.scalar_encode(s)
            assert r["pk"][0].tobytes() == o.point_encode(Sx, o.public_from_secret(Sx, sk))
            assert co.output_hash(r["output"][0].tobytes()) == o.output_hash(Sx, gamma)
            p = co.pedersen_prove_batch(skb, msgs=np.frombuffer(msg, np.uint8).reshape(1, -1), ad=ad)
            gm, proof, blinding = o.pedersen_prove(Sx, sk, H, ad)
            pk_com, R, Ok, ps, psb = proof
            assert p["pk_com"][0].tobytes() == o.point_encode(Sx, pk_com) and p["r"][0].tobytes() == o.point_encode(Sx, R)
            assert p["sb"][0].tobytes() == o.scalar_encode(psb) and p["blinding"][0].tobytes() == o.scalar_encode(blinding)
    finally:
        co.set_suite(1)


def test_device_source_host_build_follows_descriptor_strings_and_bases():
    so = os.path.join(HERE, "libhostsim.so")
    subprocess.run(["make", "-C", HERE, "-j4", "libhostsim.so"], check=True, stdout=subprocess.DEVNULL)
    hs = ctypes.CDLL(so)
    hs.hs_init()
    Sx = custom_bandersnatch()
    buf = lambda n=32: ctypes.create_string_buffer(n)
    try:
        hs.hs_set_suite_strings(Sx.suite_id, len(Sx.suite_id), Sx.h2c_dst, len(Sx.h2c_dst))
        hs.hs_set_bases(xy((Sx.gx, Sx.gy)), xy((Sx.bx, Sx.by)))
        hs.hs_set_generator_verify(xy((Sx.gx, Sx.gy)))
        for i in range(3):
            sk = o.secret_from_seed(Sx, o.synth_seed(40 + i)); msg = o.synth_msg(40 + i); ad = b"host" * i
            H = o.data_to_point(Sx, msg)
            out = buf()
            hs.hs_hash_to_curve(msg, len(msg), out)
            assert out.raw == o.point_encode(Sx, H)
            gamma, c, s = o.ietf_prove(Sx, sk, H, ad)
            g, cc, ss, hh, pk = buf(), buf(), buf(), buf(), buf()
            assert hs.hs_ietf_prove(o.scalar_encode(sk), msg, len(msg), None, ad, len(ad), g, cc, ss, hh, pk) == 1
            assert (g.raw, cc.raw, ss.raw) == (o.point_encode(Sx, gamma), o.scalar_encode(c), o.scalar_encode(s))
            assert pk.raw == o.point_encode(Sx, o.public_from_secret(Sx, sk))
            assert hs.hs_ietf_verify(pk.raw, hh.raw, g.raw, cc.raw, ss.raw, ad, len(ad)) == 0
            b = buf(64); hs.hs_output_hash(g.raw, b)
            assert b.raw == o.output_hash(Sx, gamma)
    finally:
        hs.hs_set_suite_strings(S.suite_id, len(S.suite_id), S.h2c_dst, len(S.h2c_dst))
        hs.hs_set_bases(xy((S.gx, S.gy)), xy((S.bx, S.by)))
        hs.hs_set_generator_verify(xy((S.gx, S.gy)))


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_default_descriptor_reproduces_every_golden_vector(kat):
    from ark_ec_vrfs_amd import BandersnatchSha512Ell2, Context, SuiteDesc, CURVE_BANDERSNATCH
    d = SuiteDesc.default(BandersnatchSha512Ell2)
    assert d.curve == CURVE_BANDERSNATCH and d.suite_id == S.suite_id and d.h2c_dst == S.h2c_dst
    assert d.generator == xy((S.gx, S.gy)) and d.blinding_base == xy((S.bx, S.by)) and d.challenge_len == 32
    ctx = Context(0, desc=d)
    try:
        assert ctx.desc() == d
        hx = lambda h: np.frombuffer(bytes.fromhex(h), np.uint8)
        for v in kat["ietf"]:
            pr = ctx.ietf_prove_batch(hx(v["sk"]), msgs=[bytes.fromhex(v["alpha"])], ad=bytes.fromhex(v["ad"]))
            for name, key in (("output", "gamma"), ("c", "c"), ("s", "s"), ("pk", "pk"), ("input", "h")):
                assert pr[name][0].tobytes().hex() == v[key], (name, v["comment"] if "comment" in v else "")
            assert ctx.output_hash_batch(pr["output"])[0].tobytes().hex() == v["beta"]
            st = ctx.ietf_verify_batch(pr["pk"], pr["input"], pr["output"], pr["c"], pr["s"], ad=bytes.fromhex(v["ad"]))
            assert st[0] == 0
        for v in kat["pedersen"]:
            base = next(w for w in kat["ietf"] if w["seed"] == v["seed"] and w["alpha"] == v["alpha"])
            pr = ctx.pedersen_prove_batch(hx(base["sk"]), msgs=[bytes.fromhex(v["alpha"])], ad=bytes.fromhex(v["ad"]))
            assert pr["output"][0].tobytes().hex() == base["gamma"]
            for name in ("pk_com", "r", "ok", "s", "sb", "blinding"):
                assert pr[name][0].tobytes().hex() == v[name], name
    finally:
        ctx.close()


def _gpu_vs_oracle(Sx, curve, n=192):
    from ark_ec_vrfs_amd import Context, SuiteDesc
    d = SuiteDesc(curve, Sx.suite_id, Sx.h2c_dst, xy((Sx.gx, Sx.gy)), xy((Sx.bx, Sx.by)))
    ctx = Context(0, desc=d)
    oracle_select(Sx)
    try:
        sk, msg = _items(n, 31000)
        ref = co.ietf_prove_batch(sk, msgs=msg, ad=b"desc", threads=NCPU)
        got = ctx.ietf_prove_batch(sk, msgs=msg, ad=b"desc")
        for k in ("output", "c", "s", "pk", "input"):
            assert (got[k] == ref[k]).all(), k
        assert (ctx.hash_to_curve_batch([m.tobytes() for m in msg]) == ref["input"]).all()
        assert ctx.output_hash_batch(ref["output"][:8])[3].tobytes() == co.output_hash(ref["output"][3].tobytes())
        s_bad = ref["s"].copy(); s_bad[::5, 0] ^= 1
        want = co.ietf_verify_batch(ref["pk"], ref["input"], ref["output"], ref["c"], s_bad, b"desc", threads=NCPU)
        assert (ctx.ietf_verify_batch(ref["pk"], ref["input"], ref["output"], ref["c"], s_bad, ad=b"desc") == want).all()
        assert want[::5].all() and want.sum() == len(want[::5])
        # Pedersen: the blinding base and the suite string enter the proof bytes
        pref = co.pedersen_prove_batch(sk, msgs=msg, ad=b"desc", threads=NCPU)
        pgot = ctx.pedersen_prove_batch(sk, msgs=msg, ad=b"desc")
        for k in ("output", "pk_com", "r", "ok", "s", "sb", "blinding"):
            assert (pgot[k] == pref[k]).all(), k
        args = [pref[k] for k in ("input", "output", "pk_com", "r", "ok", "s", "sb")]
        args[5] = args[5].copy(); args[5][::7, 1] ^= 2
        pw = co.pedersen_verify_batch(*args, b"desc", threads=NCPU)
        assert (ctx.pedersen_verify_batch(*args, ad=b"desc") == pw).all() and pw[::7].all()
        st, ok = ctx.pedersen_verify_batch_rlc(*args, ad=b"desc")
        assert (st == pw).all() and not ok
        # secrets / public keys use the descriptor's generator
        sks, pks = ctx.secret_from_seed_batch(np.stack([np.frombuffer(o.synth_seed(i), np.uint8) for i in range(4)]))
        for i in range(4):
            assert pks[i].tobytes() == co.public_from_secret(sks[i].tobytes())
        return ref, pref
    finally:
        co.set_suite(1)
        ctx.close()


@pytest.mark.gpu
def test_custom_bandersnatch_descriptor_equals_oracle_and_differs_from_builtin(ctx):
    from ark_ec_vrfs_amd import CURVE_BANDERSNATCH
    ref, pref = _gpu_vs_oracle(custom_bandersnatch(), CURVE_BANDERSNATCH)
    sk, msg = _items(8, 31000)
    builtin = ctx.ietf_prove_batch(sk, msgs=msg, ad=b"desc")
    assert not (builtin["input"] == ref["input"][:8]).all(axis=1).any()        # another DST: other input points
    assert not (builtin["pk"] == ref["pk"][:8]).all(axis=1).any()              # another generator: other keys


@pytest.mark.gpu
def test_custom_jubjub_descriptor_equals_oracle():
    from ark_ec_vrfs_amd import CURVE_JUBJUB, Context, JubJubSha512Tai
    ref, pref = _gpu_vs_oracle(custom_jubjub(), CURVE_JUBJUB)
    co.set_suite(2)
    sk, msg = _items(8, 31000)              # the same secrets (reduced mod JubJub's r) as inside _gpu_vs_oracle
    co.set_suite(1)
    cj = Context(0, suite=JubJubSha512Tai, test_blinding_base=True)
    try:
        b = cj.pedersen_prove_batch(sk, msgs=msg, ad=b"desc")
        assert not (b["pk_com"] == pref["pk_com"][:8]).all(axis=1).any()       # another blinding base
        assert not (b["input"] == pref["input"][:8]).all(axis=1).any()          # another suite string in the TAI hash
    finally:
        cj.close()


@pytest.mark.gpu
def test_invalid_descriptors_are_refused():
    from ark_ec_vrfs_amd import BandersnatchSha512Ell2, Context, SuiteDesc, VrfHipError
    good = SuiteDesc.default(BandersnatchSha512Ell2)
    T2 = (0, S.q - 1)
    off_curve = le(3) + le(5)
    not_subgroup = xy(o.te_add(S, (S.bx, S.by), T2))
    cases = [dataclasses.replace(good, blinding_base=off_curve), dataclasses.replace(good, blinding_base=not_subgroup),
             dataclasses.replace(good, generator=xy((0, 1))), dataclasses.replace(good, generator=le(S.q) + le(1)),
             dataclasses.replace(good, suite_id=b""), dataclasses.replace(good, h2c_dst=b""),
             dataclasses.replace(good, challenge_len=0), dataclasses.replace(good, challenge_len=33),
             dataclasses.replace(good, flags=8), dataclasses.replace(good, curve=7)]
    for d in cases:
        with pytest.raises(VrfHipError):
            Context(0, desc=d).close()
    Context(0, desc=good).close()


@pytest.mark.gpu
def test_default_descriptors_carry_only_the_pinned_blinding_base():
    """`PedersenSuite::BLINDING_BASE` is pinned by an upstream vector for Bandersnatch only.  The default
    descriptor of every other suite leaves it all-zero, such a context has the IETF scheme and every building block but
    refuses the Pedersen entry points (UNSUPPORTED) -- no proof is ever made with an invented base by default -- and the
    placeholder point is available by name (vrfhip_test_blinding_base) for tests and bench legs."""
    import numpy as np
    from ark_ec_vrfs_amd import (BabyJubJubSha512Tai, BandersnatchSha512Ell2, Context, Ed25519Sha512Tai, JubJubSha512Tai,
                                 Secp256r1Sha256Tai, SuiteDesc, VrfHipError)
    assert SuiteDesc.default(BandersnatchSha512Ell2).blinding_base == xy((S.bx, S.by)) != bytes(64)
    assert SuiteDesc.test_blinding_base(BandersnatchSha512Ell2) == xy((S.bx, S.by))
    for suite in (JubJubSha512Tai, Ed25519Sha512Tai, BabyJubJubSha512Tai, Secp256r1Sha256Tai):
        d = SuiteDesc.default(suite)
        assert d.blinding_base == bytes(64), suite.__name__
        tb = SuiteDesc.test_blinding_base(suite)
        assert tb != bytes(64) and SuiteDesc.with_test_blinding_base(suite).blinding_base == tb
        c = Context(0, suite=suite)
        try:
            assert c.desc().blinding_base == bytes(64)
            sk, _ = c.secret_from_seed_batch(np.arange(4 * 8, dtype=np.uint8).reshape(4, 8))
            msgs = [b"m%d" % i for i in range(4)]
            pr = c.ietf_prove_batch(sk, msgs=msgs, ad=b"x")                     # the IETF scheme is there
            assert not pr["status"].any()
            assert not c.ietf_verify_batch(pr["pk"], pr["input"], pr["output"], pr["c"], pr["s"], ad=b"x").any()
            with pytest.raises(VrfHipError, match="blinding base"):
                c.pedersen_prove_batch(sk, msgs=msgs, ad=b"x")
            pw = c.point_bytes()
            zp, zs = np.zeros((4, pw), np.uint8), np.zeros((4, 32), np.uint8)
            with pytest.raises(VrfHipError, match="blinding base"):
                c.pedersen_verify_batch(pr["input"], pr["output"], zp, zp, zp, zs, zs, ad=b"x")
            if suite is not Secp256r1Sha256Tai:
                with pytest.raises(VrfHipError, match="blinding base"):
                    c.pedersen_verify_batch_rlc(pr["input"], pr["output"], zp, zp, zp, zs, zs, ad=b"x")
        finally:
            c.close()
        c2 = Context(0, suite=suite, test_blinding_base=True)                    # opting in brings the scheme back
        try:
            p2 = c2.pedersen_prove_batch(sk, msgs=msgs, ad=b"x")
            assert not c2.pedersen_verify_batch(p2["input"], p2["output"], p2["pk_com"], p2["r"], p2["ok"], p2["s"], p2["sb"], ad=b"x").any()
        finally:
            c2.close()
