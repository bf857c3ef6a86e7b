"""`suites::bandersnatch_sw` ("Bandersnatch_SW_SHA-512_TAI"): the Bandersnatch group on its short-Weierstrass model
(SURVEY.md section 2.2 suite list; section 8 row f4).

PARITY UNPINNED: no vector of this suite is on this machine.  What the tiers check:
  CPU: the oracle (oracle/bsw_oracle.py, chord-and-tangent ON the Weierstrass curve) against the vector-pinned
       twisted-Edwards oracle through `utils::te_sw_map` -- same secret, same input point: Gamma, the public key, the Pedersen
       commitment are each other's images; a proof made on one model verifies there; the codec's rules.  And the device
       header of the suite compiled for the host (tests/hostsim/hostsim_bsw.hip: decode with the map and the subgroup test,
       projective encode, the transcript hashes, try-and-increment) against the oracle.
  GPU (-m gpu): the HIP path (which runs the group law on the EDWARDS model and crosses the map at the codec,
       csrc/bsw_core.cuh) against the oracle through the C ABI: prove bytes, verify statuses on tampered / undecodable /
       small-order / non-canonical inputs, Pedersen, verification from alpha, and the same consistency with the device's own
       twisted-Edwards suite at 2^13 items."""
import ctypes
import os
import random
import subprocess

import numpy as np
import pytest

from oracle import bsw_oracle as bo
from oracle import c_oracle as co
from oracle import vrf_oracle as vo

TE = vo.BANDERSNATCH
Q, R = bo.Q, bo.R


def le(x):
    return int(x).to_bytes(32, "little")


def xy(p):
    return le(p[0]) + le(p[1])


def _u8(rows):
    return np.stack([np.frombuffer(bytes(r), np.uint8) for r in rows])


def _torsion2():
    """The three points of order 2: (x, 0) with x a root of x^3 + a' x + b'.  One is the image's partner of the Edwards
    point (0, -1); the other two have no affine Edwards image."""
    # E[2] is rational (Z2 x Z2 x Zr): u in {0, roots of u^2 + A u + 1} on the Montgomery model; x = (u + A/3) / B
    A, B, _, _ = vo.te_sw_constants(TE)
    roots = []
    third = A * vo.finv(3, Q) % Q
    disc = vo.fsqrt((A * A - 4) % Q, Q)
    us = [0] + ([] if disc is None else [(-A + disc) * vo.finv(2, Q) % Q, (-A - disc) * vo.finv(2, Q) % Q])
    for u in us:
        x = (u + third) * vo.finv(B, Q) % Q
        assert (x ** 3 + bo.A * x + bo.B) % Q == 0
        roots.append((x, 0))
    return roots


# ------------------------------------------------------------------------------------------------------ CPU tier
def test_curve_and_constants():
    assert bo.is_on_curve(bo.G) and bo.is_on_curve(bo.BLINDING_BASE)
    assert bo.in_prime_subgroup(bo.G) and bo.in_prime_subgroup(bo.BLINDING_BASE)
    assert bo.mul(R, bo.G) is None and bo.mul(R - 1, bo.G) == bo.neg(bo.G)
    assert vo.sw_to_te(TE, bo.G) == (TE.gx, TE.gy)
    # group order 4 r: cofactor clearing lands in the subgroup; the 2-torsion is rational and full
    t2 = _torsion2()
    assert len(t2) == 3 and all(bo.mul(2, t) is None for t in t2)
    assert bo.add(t2[0], t2[1]) == t2[2]
    # the generated device constants are these numbers
    import re
    src = open(__import__("os").path.join(__import__("os").path.dirname(__file__), "..", "ark_ec_vrfs_amd", "csrc",
                                          "constants_bsw.gen.h")).read()
    g = bytes(int(v) for v in re.search(r"G_XY\[64\] = \{([^}]*)\}", src).group(1).split(","))
    b = bytes(int(v) for v in re.search(r"BB_XY\[64\] = \{([^}]*)\}", src).group(1).split(","))
    assert g == xy(bo.G) and b == xy(bo.BLINDING_BASE)


def test_oracle_group_law_is_the_map_image_of_the_edwards_law():
    import random
    rnd = random.Random(5)
    P = bo.G
    for _ in range(40):
        k1, k2 = rnd.randrange(1, R), rnd.randrange(1, R)
        a, b = bo.mul(k1, P), bo.mul(k2, P)
        ta, tb = vo.te_mul(TE, k1, (TE.gx, TE.gy)), vo.te_mul(TE, k2, (TE.gx, TE.gy))
        assert vo.te_to_sw(TE, ta) == a and vo.sw_to_te(TE, b) == tb
        assert vo.te_to_sw(TE, vo.te_add(TE, ta, tb)) == bo.add(a, b)
        assert bo.add(a, a) == bo.mul(2, a) and bo.add(a, bo.neg(a)) is None and bo.add(a, None) == a


def test_codec_rules():
    p = bo.mul(7, bo.G)
    e = bo.point_encode(p)
    assert len(e) == 33 and e[:32] == le(p[0]) and e[32] == (0x80 if p[1] > Q - p[1] else 0)
    assert bo.point_decode(e) == (True, p)
    assert bo.point_decode(bo.point_encode(bo.neg(p))) == (True, bo.neg(p)) and bo.point_encode(bo.neg(p))[32] == e[32] ^ 0x80
    assert bo.point_encode(None) == bytes(32) + b"\x40" and bo.point_decode(bytes(32) + b"\x40") == (True, None)
    assert bo.point_decode(e[:32] + bytes([e[32] | 0x40]))[0] == (e[32] == 0)           # both flags: an error
    assert bo.point_decode(e[:32] + bytes([e[32] | 0x2a])) == (True, p)                 # low flag-byte bits are not looked at
    assert bo.point_decode(le(123) + b"\x40") == (True, None)                           # infinity: x is not looked at...
    assert bo.point_decode(le(Q) + b"\x40")[0] is False                                 # ...beyond being a field element
    assert bo.point_decode(le(Q + p[0]) + e[32:])[0] is False
    x = next(x for x in range(2, 100) if vo.legendre((x ** 3 + bo.A * x + bo.B) % Q, Q) == -1)
    assert bo.point_decode(le(x) + b"\x00")[0] is False
    t = _torsion2()[0]
    assert bo.point_decode(le(t[0]) + b"\x80") == (True, t) and bo.point_encode(t)[32] == 0     # y = 0: either flag
    assert bo.point_decode_checked(le(t[0]) + b"\x00")[0] is False
    assert bo.point_decode_checked(bo.point_encode(bo.add(p, t)))[0] is False           # outside the prime-order subgroup


def test_oracle_schemes_against_the_pinned_edwards_oracle():
    """Same secret, same input point on both models: Gamma, pk and pk_com are each other's images (the nonce and the challenge
    hash the encodings, so they differ by design); proofs verify on their own model and not after tampering."""
    for i in range(6):
        sk = bo.secret_from_seed(b"seed %d" % i)
        assert sk == vo.secret_from_seed(TE, b"seed %d" % i)
        h = bo.hash_to_curve_tai(b"input %d" % i)
        assert bo.in_prime_subgroup(h)
        ht = vo.sw_to_te(TE, h)
        ad = b"ad" * i
        gamma, c, s = bo.ietf_prove(sk, h, ad)
        gt, _, _ = vo.ietf_prove(TE, sk, ht, ad)
        pk = bo.public_from_secret(sk)
        assert vo.te_to_sw(TE, gt) == gamma and vo.te_to_sw(TE, vo.public_from_secret(TE, sk)) == pk
        assert bo.ietf_verify(pk, h, gamma, ad, c, s) and bo.ietf_verify(pk, h, gamma, ad, c + R, s)
        assert not bo.ietf_verify(pk, h, gamma, ad + b"x", c, s) and not bo.ietf_verify(pk, h, gamma, ad, c, (s + 1) % R)
        assert not bo.ietf_verify(pk, h, bo.add(gamma, bo.G), ad, c, s)
        enc = [bo.point_encode(p) for p in (pk, h, gamma)]
        assert bo.ietf_verify_bytes(*enc, ad, le(c), le(s)) == 0 and bo.ietf_verify_bytes(*enc, ad, le(c), le(s + R)) == 2
        g2, proof, b = bo.pedersen_prove(sk, h, ad)
        assert g2 == gamma and bo.pedersen_verify(h, gamma, ad, proof)
        # the commitment with the Edwards oracle's arithmetic and THIS suite's blinding factor
        te_com = vo.te_add(TE, vo.te_mul(TE, sk, (TE.gx, TE.gy)), vo.te_mul(TE, b, (TE.bx, TE.by)))
        assert vo.te_to_sw(TE, te_com) == proof[0]
        bad = (proof[0], proof[1], proof[2], proof[3], (proof[4] + 1) % R)
        assert not bo.pedersen_verify(h, gamma, ad, bad)
        assert len(bo.output_hash(gamma)) == 64


def _defects(rnd, n, pk, h, g, c, s, proofs):
    """Every kind of defect on copies of a valid batch (proofs[i] = (Gamma, c, s) as the oracle's values)."""
    pkt, ht, gt, ct, st_ = (x.copy() for x in (pk, h, g, c, s))
    t2 = _torsion2()
    offx = next(x for x in range(2, 100) if vo.legendre((x ** 3 + bo.A * x + bo.B) % Q, Q) == -1)
    for i in range(n):
        kind = i % 16
        if kind == 1: st_[i, 3] ^= 1
        elif kind == 2: ct[i, 30] ^= 1
        elif kind == 3: gt[i] = g[(i + 1) % n]
        elif kind == 4: pkt[i] = np.frombuffer(le(offx) + b"\x00", np.uint8)                    # not on the curve
        elif kind == 5: ht[i, 32] |= 0xC0                                                        # both flags
        elif kind == 6: ct[i] = np.frombuffer(le(proofs[i][1] + R), np.uint8)                   # c + r: the same scalar mod r
        elif kind == 7: gt[i] = np.frombuffer(le(Q + 5) + b"\x00", np.uint8)                    # x >= q
        elif kind == 8: st_[i] = np.frombuffer(le(proofs[i][2] + R), np.uint8)                  # s not canonical
        elif kind == 9: pkt[i, 32] |= rnd.randrange(1, 64)                                       # junk in the flag byte's low bits: ignored
        elif kind == 10: gt[i] = np.frombuffer(bo.point_encode(bo.add(proofs[i][0], t2[i % 3])), np.uint8)    # + a point of order 2
        elif kind == 11: pkt[i] = np.frombuffer(le(t2[i % 3][0]) + b"\x00", np.uint8)           # a point of order 2 itself
        elif kind == 12: ht[i] = np.frombuffer(le(77) + b"\x40", np.uint8)                      # infinity (x not looked at)
        elif kind == 13: gt[i, 32] ^= 0x80                                                       # -Gamma
        elif kind == 14: pkt[i] = np.frombuffer(bytes(32) + b"\x40", np.uint8)                  # pk = infinity
    return pkt, ht, gt, ct, st_


def test_c_oracle_equals_the_python_oracle():
    """oracle/c/oracle_bsw.c (the batch-size checker and CPU baseline) against oracle/bsw_oracle.py: constants, keys,
    hash-to-curve, both schemes byte for byte, statuses on every kind of defect."""
    rnd = random.Random(31)
    ab, g, bb = co.bsw_constants()
    assert ab == le(bo.A) + le(bo.B) and g == xy(bo.G) and bb == xy(bo.BLINDING_BASE)
    n = 48
    seeds = [b"seed-%d" % i for i in range(n)]
    sk = _u8(co.bsw_secret_public(sd)[0] for sd in seeds)
    for i in range(0, n, 5):
        k = bo.secret_from_seed(seeds[i])
        assert co.bsw_secret_public(seeds[i]) == (le(k), bo.point_encode(bo.mul(k, bo.G)))
    msgs = np.frombuffer(b"".join(b"%020d" % i for i in range(n)), np.uint8).reshape(n, 20)
    ad = b"additional"
    r = co.bsw_ietf_prove_batch(sk, msgs=msgs, ad=ad, threads=4)
    proofs = []
    for i in range(n):
        k = int.from_bytes(sk[i].tobytes(), "little")
        h = bo.hash_to_curve_tai(msgs[i].tobytes())
        gamma, c, s = bo.ietf_prove(k, h, ad)
        proofs.append((gamma, c, s))
        assert co.bsw_hash_to_curve(msgs[i].tobytes()) == bo.point_encode(h) == r["input"][i].tobytes()
        assert r["output"][i].tobytes() == bo.point_encode(gamma) and r["c"][i].tobytes() == le(c) and r["s"][i].tobytes() == le(s)
        assert r["pk"][i].tobytes() == bo.point_encode(bo.mul(k, bo.G))
        assert co.bsw_output_hash(r["output"][i].tobytes()) == bo.output_hash(gamma)
    r2 = co.bsw_ietf_prove_batch(sk, inputs=r["input"], ad=ad, threads=2)
    assert all((r2[k_] == r[k_]).all() for k_ in ("output", "c", "s", "pk", "input"))
    t = _defects(rnd, n, r["pk"], r["input"], r["output"], r["c"], r["s"], proofs)
    got = co.bsw_ietf_verify_batch(*t, ad=ad, threads=4)
    want = np.array([bo.ietf_verify_bytes(t[0][i].tobytes(), t[1][i].tobytes(), t[2][i].tobytes(), ad, t[3][i].tobytes(),
                                          t[4][i].tobytes()) for i in range(n)], np.uint8)
    assert (got == want).all() and set(want) == {0, 1, 2}
    p = co.bsw_pedersen_prove_batch(sk, msgs=msgs, ad=ad, threads=4)
    for i in range(0, n, 3):
        k = int.from_bytes(sk[i].tobytes(), "little")
        gamma, (pc, rr, ok, s, sb), b = bo.pedersen_prove(k, bo.hash_to_curve_tai(msgs[i].tobytes()), ad)
        assert [p[k_][i].tobytes() for k_ in ("output", "pk_com", "r", "ok")] == [bo.point_encode(v) for v in (gamma, pc, rr, ok)]
        assert p["s"][i].tobytes() == le(s) and p["sb"][i].tobytes() == le(sb) and p["blinding"][i].tobytes() == le(b)
    args = [p[k_].copy() for k_ in ("input", "output", "pk_com", "r", "ok", "s", "sb")]
    args[5][1, 0] ^= 1; args[6][2, 0] ^= 1; args[3][3, 32] |= 0xC0; args[2][4] = args[2][5]
    got = co.bsw_pedersen_verify_batch(*args, ad=ad, threads=4)
    want = np.array([bo.pedersen_verify_bytes(*(args[j][i].tobytes() for j in range(7)), ad) for i in range(n)], np.uint8)
    assert (got == want).all() and list(want[:6]) == [0, 1, 1, 2, 1, 0]
    for enc in (bo.point_encode(bo.mul(5, bo.G)), bytes(32) + b"\x40", le(_torsion2()[1][0]) + b"\x00"):
        ok_, pt = bo.point_decode_checked(enc)
        assert (co.bsw_point_decode(enc) is not None) == ok_
        if ok_:
            assert co.bsw_point_decode(enc) == (bytes(64) if pt is None else xy(pt))


# ------------------------------------------------------------------------- the device header on the host (CPU tier)
@pytest.fixture(scope="module")
def hb():
    hs = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim")
    subprocess.run(["make", "-C", hs, "libhostsim_bsw.so"], check=True, stdout=subprocess.DEVNULL)
    return ctypes.CDLL(os.path.join(hs, "libhostsim_bsw.so"))


def _hb_decode(hb, enc):
    te, sw = ctypes.create_string_buffer(64), ctypes.create_string_buffer(64)
    fl = hb.hb_decode(bytes(enc), te, sw)
    return fl, te.raw, sw.raw


def _hb_encode(hb, te_pt, z):
    o = ctypes.create_string_buffer(33)
    hb.hb_encode(xy(te_pt), le(z), o)
    return o.raw


def test_host_build_codec_equals_the_oracle(hb):
    rnd = random.Random(17)
    t2 = _torsion2()
    pts = [bo.mul(rnd.randrange(1, R), bo.G) for _ in range(24)]
    pts += [bo.neg(pts[0]), bo.add(pts[1], t2[0]), bo.add(pts[2], t2[1]), bo.add(pts[3], t2[2])]
    for p in pts:
        e = bo.point_encode(p)
        fl, te, sw = _hb_decode(hb, e)
        assert fl & 1 and not fl & 2 and bool(fl & 4) == bo.in_prime_subgroup(p)
        assert sw == xy(p) and te == xy(vo.sw_to_te(TE, p))
        junk = e[:32] + bytes([e[32] | rnd.randrange(64)])                       # low flag-byte bits are not looked at
        assert _hb_decode(hb, junk)[1:] == (te, sw)
        o = ctypes.create_string_buffer(33); hb.hb_canonical(junk, o)
        assert o.raw == e
        for z in (1, 2, rnd.randrange(1, Q)):                                      # projective representatives
            assert _hb_encode(hb, vo.sw_to_te(TE, p), z) == e
    # infinity <-> (0, 1); the Edwards point of order 2 (0, -1) <-> (x0, 0)
    fl, te, _ = _hb_decode(hb, le(99) + b"\x40")
    assert fl == 7 and te == xy((0, 1))
    assert _hb_encode(hb, (0, 1), 5) == bytes(32) + b"\x40"
    enc_t = _hb_encode(hb, (0, Q - 1), 3)
    ok, t = bo.point_decode(enc_t)
    assert ok and t[1] == 0 and t in t2 and enc_t[32] == 0
    # rejected: both flags, x >= q, not on the curve, and y = 0 (no Edwards image: see csrc/bsw_core.cuh)
    e = bo.point_encode(pts[0])
    offx = next(x for x in range(2, 100) if vo.legendre((x ** 3 + bo.A * x + bo.B) % Q, Q) == -1)
    for bad in (e[:32] + bytes([e[32] | 0xC0]), le(Q + 3) + b"\x00", le(offx) + b"\x00", le(Q) + b"\x40", le(t2[0][0]) + b"\x00"):
        assert _hb_decode(hb, bad)[0] & 1 == 0


def test_host_build_shared_inversion_equals_the_point_by_point_decode(hb):
    """The way in as the verify kernels run it: the points of an item share one te_sw_map inversion (Montgomery's trick over
    the map denominators).  Mixtures of valid points, infinity, points outside the subgroup, undecodable strings and points
    with y = 0 (denominator zero: replaced by one, flagged) give what the point-by-point decode gives."""
    rnd = random.Random(29)
    t2 = _torsion2()
    offx = next(x for x in range(2, 100) if vo.legendre((x ** 3 + bo.A * x + bo.B) % Q, Q) == -1)
    pool = [bo.point_encode(bo.mul(rnd.randrange(1, R), bo.G)) for _ in range(12)]
    pool += [bytes(32) + b"\x40", le(5) + b"\x40", bo.point_encode(bo.add(bo.mul(9, bo.G), t2[1])), le(t2[0][0]) + b"\x00",
             le(t2[2][0]) + b"\x80", le(offx) + b"\x00", le(Q + 2) + b"\x00", pool[0][:32] + b"\xc0", pool[1][:32] + bytes([pool[1][32] | 0x11])]
    for trial in range(60):
        n = rnd.choice((1, 2, 3, 5, 8))
        encs = [rnd.choice(pool) for _ in range(n)]
        te, fl = ctypes.create_string_buffer(64 * n), ctypes.create_string_buffer(n)
        hb.hb_decode_shared(b"".join(encs), n, te, fl)
        for i, e in enumerate(encs):
            f1, te1, _ = _hb_decode(hb, e)
            assert fl.raw[i] == (f1 & 3), (trial, i)
            if f1 & 1:
                assert te.raw[64 * i:64 * i + 64] == te1, (trial, i)


def test_host_build_hashes_and_try_and_increment_equal_the_oracle(hb):
    rnd = random.Random(23)
    for i in range(12):
        P5 = [bo.mul(rnd.randrange(1, R), bo.G) for _ in range(5)]
        if i == 3:
            P5[2] = None
        ad = bytes(rnd.randrange(256) for _ in range(rnd.choice((0, 1, 7, 64, 111, 200))))
        c = ctypes.create_string_buffer(32)
        hb.hb_challenge(b"".join(bo.point_encode(p) for p in P5), ad, len(ad), c)
        assert c.raw == le(bo.challenge(P5, ad))
        sk = rnd.randrange(1, R)
        k = ctypes.create_string_buffer(32); hb.hb_nonce(le(sk), bo.point_encode(P5[0]), k)
        assert k.raw == le(bo.nonce_rfc8032(sk, P5[0]))
        b = ctypes.create_string_buffer(32); hb.hb_blinding(le(sk), bo.point_encode(P5[1]), ad, len(ad), b)
        assert b.raw == le(bo.pedersen_blinding(sk, P5[1], ad))
        o = ctypes.create_string_buffer(64); hb.hb_output_hash(bo.point_encode(P5[4]), o)
        assert o.raw == bo.output_hash(P5[4])
    for i in range(40):
        msg = bytes(rnd.randrange(256) for _ in range(rnd.choice((0, 3, 32, 100))))
        e, hint = ctypes.create_string_buffer(33), ctypes.c_uint32(0)
        hb.hb_hash_to_curve(msg, len(msg), 0, e, ctypes.byref(hint))
        want = bo.point_encode(bo.hash_to_curve_tai(msg))
        assert e.raw == want
        # the counter k_tai_find hands the kernels: the first candidate that is a finite curve point; starting there changes nothing
        first = next(c for c in range(256) if (lambda okp: okp[0] and okp[1] is not None)(
            bo.point_decode(bo.sha512(bo.SUITE_ID + b"\x01" + msg + bytes([c]) + b"\x00")[:33])))
        assert hint.value == first
        e2 = ctypes.create_string_buffer(33); hb.hb_hash_to_curve(msg, len(msg), hint.value, e2, None)
        assert e2.raw == want


# ------------------------------------------------------------------------------------------------------ GPU tier
@pytest.fixture(scope="module")
def gpu():
    from ark_ec_vrfs_amd import BandersnatchSwSha512Tai, Context
    ctx = Context(0, BandersnatchSwSha512Tai, test_blinding_base=True)
    yield ctx
    ctx.close()


@pytest.mark.gpu
def test_gpu_descriptor_keys_and_hash_to_curve(gpu):
    from ark_ec_vrfs_amd import BandersnatchSwSha512Tai, Context, SuiteDesc
    assert gpu.point_bytes() == 33 and gpu.hash_bytes() == 64
    d = gpu.desc()
    assert d.curve == 6 and d.suite_id == bo.SUITE_ID and d.challenge_len == 32 and d.flags == 0
    assert d.generator == xy(bo.G) and d.blinding_base == xy(bo.BLINDING_BASE)
    plain = Context(0, BandersnatchSwSha512Tai)                       # the built-in descriptor: no blinding base, no Pedersen scheme
    assert plain.desc().blinding_base == bytes(64) and plain.desc().generator == xy(bo.G)
    sk0 = np.zeros((1, 32), np.uint8); sk0[0, 0] = 3
    with pytest.raises(Exception):
        plain.pedersen_prove_batch(sk0, msgs=[b"m"], ad=b"")
    assert plain.ietf_prove_batch(sk0, msgs=[b"m"], ad=b"")["pk"][0].tobytes() == bo.point_encode(bo.mul(3, bo.G))
    plain.close()
    # context creation maps the descriptor's points to the Edwards model in front of the table kernels (same stream): repeated
    # creations, back to back, each give the right tables (a null-stream copy there once raced with them under the profiler)
    for _ in range(6):
        cx = Context(0, BandersnatchSwSha512Tai)
        assert cx.secret_from_seed_batch(np.zeros((1, 4), np.uint8))[1][0].tobytes() == co.bsw_secret_public(bytes(4))[1]
        cx.close()
    bad = SuiteDesc.with_test_blinding_base(BandersnatchSwSha512Tai)
    bad.generator = xy((bo.G[0], (bo.G[1] + 1) % Q))
    with pytest.raises(Exception):
        Context(0, desc=bad)
    n = 200
    seeds = _u8(b"seed-%05d" % i + bytes(i % 7) + bytes(22 - i % 7) for i in range(n))
    sk, pk = gpu.secret_from_seed_batch(seeds)
    for i in range(0, n, 9):
        k = bo.secret_from_seed(seeds[i].tobytes())
        assert sk[i].tobytes() == le(k) and pk[i].tobytes() == bo.point_encode(bo.mul(k, bo.G))
    msgs = [b"m%d" % i * (i % 6) for i in range(n)]
    hs = gpu.hash_to_curve_batch(msgs)
    for i in range(n):
        assert hs[i].tobytes() == bo.point_encode(bo.hash_to_curve_tai(msgs[i])), i
    fixed = np.frombuffer(b"".join(b"%032d" % i for i in range(64)), np.uint8).reshape(64, 32)
    hf = gpu.hash_to_curve_batch(fixed)
    assert all(hf[i].tobytes() == bo.point_encode(bo.hash_to_curve_tai(fixed[i].tobytes())) for i in range(64))


@pytest.mark.gpu
@pytest.mark.parametrize("given", [False, True])
def test_gpu_prove_bytes_and_verify_statuses_equal_the_oracle(gpu, given):
    n = 96
    seeds = _u8(b"k%07d" % i for i in range(n))
    sk, pk = gpu.secret_from_seed_batch(seeds)
    msgs = [b"alpha %d" % i * (1 + i % 4) for i in range(n)]
    ads = [b"ad" * (i % 5) for i in range(n)]
    H = [bo.hash_to_curve_tai(m) for m in msgs]
    if given:
        r = gpu.ietf_prove_batch(sk, inputs=_u8(bo.point_encode(h) for h in H), ad=ads)
    else:
        r = gpu.ietf_prove_batch(sk, msgs=msgs, ad=ads)
    assert (r["status"] == 0).all() and (r["pk"] == pk).all()
    proofs = []
    for i in range(n):
        k = int.from_bytes(sk[i].tobytes(), "little")
        gamma, c, s = bo.ietf_prove(k, H[i], ads[i])
        proofs.append((gamma, c, s))
        assert r["input"][i].tobytes() == bo.point_encode(H[i]) and r["output"][i].tobytes() == bo.point_encode(gamma), i
        assert r["c"][i].tobytes() == le(c) and r["s"][i].tobytes() == le(s), i
    assert (gpu.ietf_verify_batch(pk, r["input"], r["output"], r["c"], r["s"], ad=ads) == 0).all()
    beta = gpu.output_hash_batch(r["output"])
    assert all(beta[i].tobytes() == bo.output_hash(proofs[i][0]) for i in range(n))
    # every kind of defect; the oracle decides
    pkt, ht, gt, ct, st_ = (x.copy() for x in (pk, r["input"], r["output"], r["c"], r["s"]))
    t2 = _torsion2()
    offx = next(x for x in range(2, 100) if vo.legendre((x ** 3 + bo.A * x + bo.B) % Q, Q) == -1)
    for i in range(n):
        kind = i % 16
        if kind == 1: st_[i, 3] ^= 1
        elif kind == 2: ct[i, 30] ^= 1
        elif kind == 3: gt[i] = r["output"][(i + 1) % n]
        elif kind == 4: pkt[i] = np.frombuffer(le(offx) + b"\x00", np.uint8)                    # not on the curve
        elif kind == 5: ht[i, 32] |= 0xC0                                                        # both flags
        elif kind == 6: ct[i] = np.frombuffer(le(proofs[i][1] + R), np.uint8)                   # c + r: the same scalar mod r
        elif kind == 7: gt[i] = np.frombuffer(le(Q + 5) + b"\x00", np.uint8)                    # x >= q
        elif kind == 8: st_[i] = np.frombuffer(le(proofs[i][2] + R), np.uint8)                  # s not canonical
        elif kind == 9: pkt[i, 32] |= 0x15                                                       # junk in the flag byte's low bits: ignored
        elif kind == 10: gt[i] = np.frombuffer(bo.point_encode(bo.add(proofs[i][0], t2[i % 3])), np.uint8)    # + a point of order 2
        elif kind == 11: pkt[i] = np.frombuffer(le(t2[i % 3][0]) + b"\x00", np.uint8)           # a point of order 2 itself
        elif kind == 12: ht[i] = np.frombuffer(le(77) + b"\x40", np.uint8)                      # infinity (x not looked at)
        elif kind == 13: gt[i, 32] ^= 0x80                                                       # -Gamma
        elif kind == 14: pkt[i] = np.frombuffer(bytes(32) + b"\x40", np.uint8)                  # pk = infinity
    want = np.array([bo.ietf_verify_bytes(pkt[i].tobytes(), ht[i].tobytes(), gt[i].tobytes(), ads[i], ct[i].tobytes(),
                                          st_[i].tobytes()) for i in range(n)], np.uint8)
    got = gpu.ietf_verify_batch(pkt, ht, gt, ct, st_, ad=ads)
    assert (got == want).all(), (got, want)
    assert set(want) == {0, 1, 2} and want[9] == 0 and want[6] == 0 and want[10] == 2 and want[12] in (1, 2)
    # verification from alpha
    assert (gpu.ietf_verify_batch_alpha(pk, msgs, r["output"], r["c"], r["s"], ad=ads) == 0).all()
    got = gpu.ietf_verify_batch_alpha(pkt, msgs, gt, ct, st_, ad=ads)
    want_a = np.array([bo.ietf_verify_bytes(pkt[i].tobytes(), r["input"][i].tobytes(), gt[i].tobytes(), ads[i], ct[i].tobytes(),
                                            st_[i].tobytes()) for i in range(n)], np.uint8)
    assert (got == want_a).all()
    # point validation, x || y out
    pts = _u8([bo.point_encode(bo.mul(7, bo.G)), le(offx) + b"\x00", bytes(32) + b"\x40", bo.point_encode(bo.add(bo.G, t2[0])),
               le(bo.G[0]) + b"\xc0", bo.point_encode(bo.neg(bo.G))])
    stv, xyv = gpu.point_validate_batch(pts, want_xy=True)
    assert list(stv) == [0, 2, 0, 2, 2, 0]
    assert xyv[0].tobytes() == xy(bo.mul(7, bo.G)) and xyv[5].tobytes() == xy(bo.neg(bo.G)) and xyv[2].tobytes() == bytes(64)


@pytest.mark.gpu
def test_gpu_pedersen_equals_the_oracle(gpu):
    n = 64
    seeds = _u8(b"p%07d" % i for i in range(n))
    sk, _ = gpu.secret_from_seed_batch(seeds)
    msgs = [b"in %d" % i for i in range(n)]
    ads = [b"x" * (i % 3) for i in range(n)]
    r = gpu.pedersen_prove_batch(sk, msgs=msgs, ad=ads)
    assert (r["status"] == 0).all()
    for i in range(n):
        k = int.from_bytes(sk[i].tobytes(), "little")
        h = bo.hash_to_curve_tai(msgs[i])
        gamma, (pk_com, rr, ok, s, sb), b = bo.pedersen_prove(k, h, ads[i])
        for key, val in (("input", h), ("output", gamma), ("pk_com", pk_com), ("r", rr), ("ok", ok)):
            assert r[key][i].tobytes() == bo.point_encode(val), (i, key)
        assert r["s"][i].tobytes() == le(s) and r["sb"][i].tobytes() == le(sb) and r["blinding"][i].tobytes() == le(b)
    args = [r[k_] for k_ in ("input", "output", "pk_com", "r", "ok", "s", "sb")]
    assert (gpu.pedersen_verify_batch(*args, ad=ads) == 0).all()
    t = [a.copy() for a in args]
    t2 = _torsion2()
    for i in range(n):
        kind = i % 8
        if kind == 1: t[5][i, 0] ^= 1
        elif kind == 2: t[6][i, 0] ^= 1
        elif kind == 3: t[2][i] = args[2][(i + 1) % n]
        elif kind == 4: t[3][i, 32] |= 0xC0
        elif kind == 5: t[4][i] = np.frombuffer(bo.point_encode(bo.add(bo.point_decode(args[4][i].tobytes())[1], t2[0])), np.uint8)
        elif kind == 6: t[6][i] = np.frombuffer(le(int.from_bytes(args[6][i].tobytes(), "little") + R), np.uint8)
        elif kind == 7: t[1][i, 32] |= 0x03
    want = np.array([bo.pedersen_verify_bytes(*(t[j][i].tobytes() for j in range(7)), ads[i]) for i in range(n)], np.uint8)
    got = gpu.pedersen_verify_batch(*t, ad=ads)
    assert (got == want).all() and set(want) == {0, 1, 2}, (got, want)
    # the whole batch through ONE multi-scalar multiplication: same statuses; the valid batch takes the fast path
    st_b, fast = gpu.pedersen_verify_batch_rlc(*args, ad=ads, seed=bytes(range(32)))
    assert fast and (st_b == 0).all()
    st_b, fast = gpu.pedersen_verify_batch_rlc(*t, ad=ads, seed=bytes(range(32)))
    assert not fast and (st_b == want).all()
    # defects that cancel in an unweighted sum: sb_i + d, sb_j - d
    u = [a.copy() for a in args]
    d = 12345
    for i, sign in ((3, 1), (9, -1)):
        u[6][i] = np.frombuffer(le((int.from_bytes(args[6][i].tobytes(), "little") + sign * d) % R), np.uint8)
    st_b, fast = gpu.pedersen_verify_batch_rlc(*u, ad=ads)
    assert not fast and list(np.nonzero(st_b)[0]) == [3, 9]
    # ... and from Weierstrass x || y (typed callers), canonical and Montgomery-256; a failed batch falls back per proof
    for flags in (0, gpu.COORDS_MONT256):
        gpu.set_flags(flags)
        xy5 = []
        for a in args[:5]:
            stv, xyv = gpu.point_validate_batch(a, want_xy=True)
            assert not stv.any()
            xy5.append(xyv)
        st_b, fast = gpu.pedersen_verify_batch_rlc(*xy5, args[5], args[6], ad=ads, affine=True)
        assert fast and not st_b.any()
        badxy = [a.copy() for a in xy5]
        sb2 = args[6].copy()
        sb2[2, 0] ^= 1
        badxy[4][5, 33] ^= 1                                                   # Ok off the curve
        badxy[1][7] = xy5[1][8]
        st_b, fast = gpu.pedersen_verify_batch_rlc(*badxy, args[5], sb2, ad=ads, affine=True)
        assert not fast and st_b[2] == 1 and st_b[5] == 2 and st_b[7] == 1 and st_b.sum() == 4
    gpu.set_flags(0)


@pytest.mark.gpu
def test_gpu_whole_batch_equals_the_c_oracle(gpu):
    """2^13 items against the C oracle (which works on the Weierstrass curve): every proof byte, every status of a batch
    with all defect kinds, both schemes."""
    import multiprocessing
    nt = min(16, multiprocessing.cpu_count())
    rnd = random.Random(41)
    n = 1 << 13
    rng = np.random.default_rng(5)
    seeds = rng.integers(0, 256, (n, 12), dtype=np.uint8)
    sk, pk = gpu.secret_from_seed_batch(seeds)
    for i in range(0, n, 997):
        assert (sk[i].tobytes(), pk[i].tobytes()) == co.bsw_secret_public(seeds[i].tobytes())
    msgs = rng.integers(0, 256, (n, 28), dtype=np.uint8)
    ad = b"whole batch"
    r = gpu.ietf_prove_batch(sk, msgs=msgs, ad=ad)
    ref = co.bsw_ietf_prove_batch(sk, msgs=msgs, ad=ad, threads=nt)
    for k_ in ("output", "c", "s", "pk", "input"):
        assert (r[k_] == ref[k_]).all(), k_
    proofs = [(None, int.from_bytes(ref["c"][i].tobytes(), "little"), int.from_bytes(ref["s"][i].tobytes(), "little")) for i in range(n)]
    for i in range(10, n, 16):                                   # the defect that needs the typed Gamma
        proofs[i] = (bo.point_decode(ref["output"][i].tobytes())[1],) + proofs[i][1:]
    t = _defects(rnd, n, ref["pk"], ref["input"], ref["output"], ref["c"], ref["s"], proofs)
    want = co.bsw_ietf_verify_batch(*t, ad=ad, threads=nt)
    got = gpu.ietf_verify_batch(*t, ad=ad)
    assert (got == want).all() and set(np.unique(want)) == {0, 1, 2}
    assert (np.bincount(want, minlength=3) > n // 16).all()
    p = gpu.pedersen_prove_batch(sk, msgs=msgs, ad=ad)
    pref = co.bsw_pedersen_prove_batch(sk, msgs=msgs, ad=ad, threads=nt)
    for k_ in ("output", "pk_com", "r", "ok", "s", "sb", "blinding", "input"):
        assert (p[k_] == pref[k_]).all(), k_
    args = [pref[k_].copy() for k_ in ("input", "output", "pk_com", "r", "ok", "s", "sb")]
    args[5][::7, 2] ^= 1; args[6][3::11, 0] ^= 1; args[3][5::13, 32] |= 0xC0; args[2][6::17] = args[2][7::17][: len(args[2][6::17])]
    args[4][8::19, 32] ^= 0x80
    want = co.bsw_pedersen_verify_batch(*args, ad=ad, threads=nt)
    got = gpu.pedersen_verify_batch(*args, ad=ad)
    assert (got == want).all() and set(np.unique(want)) == {0, 1, 2}
    got_b, fast = gpu.pedersen_verify_batch_rlc(*args, ad=ad)
    assert not fast and (got_b == want).all()


@pytest.mark.gpu
def test_gpu_consistency_with_the_edwards_suite_at_scale(gpu):
    """2^13 proofs: every Gamma / pk of this suite is the te_sw_map image of what the device's own (vector-pinned)
    twisted-Edwards suite computes from the same secret and the mapped input point; everything verifies; a tampered stripe
    does not."""
    from ark_ec_vrfs_amd import BandersnatchSha512Ell2, Context
    n = 1 << 13
    rng = np.random.default_rng(11)
    seeds = rng.integers(0, 256, (n, 16), dtype=np.uint8)
    sk, pk = gpu.secret_from_seed_batch(seeds)
    msgs = rng.integers(0, 256, (n, 24), dtype=np.uint8)
    r = gpu.ietf_prove_batch(sk, msgs=msgs, ad=b"scale")
    assert (r["status"] == 0).all() and (r["pk"] == pk).all()
    assert (gpu.ietf_verify_batch(pk, r["input"], r["output"], r["c"], r["s"], ad=b"scale") == 0).all()
    assert (gpu.ietf_verify_batch_alpha(pk, msgs, r["output"], r["c"], r["s"], ad=b"scale") == 0).all()
    s2 = r["s"].copy(); s2[::5, 1] ^= 4
    st = gpu.ietf_verify_batch(pk, r["input"], r["output"], r["c"], s2, ad=b"scale")
    assert (st[::5] != 0).all() and (np.delete(st, np.arange(0, n, 5)) == 0).all()
    # the Weierstrass coordinates of H, Gamma, pk -> Edwards coordinates -> the Edwards suite's compressed form
    te = Context(0, BandersnatchSha512Ell2)
    def sw_xy(enc):
        stv, xyv = gpu.point_validate_batch(enc, want_xy=True)
        assert (stv == 0).all()
        return xyv
    def te_enc(xy_te):
        out = xy_te[:, 32:].copy()
        x = xy_te[:, :32]
        # arkworks' flag: x > q - x  <=>  x > (q - 1) / 2, compared as big-endian byte strings
        half = np.frombuffer(((Q - 1) // 2).to_bytes(32, "big"), np.uint8)
        xb = x[:, ::-1]
        diff = xb.astype(np.int16) - half.astype(np.int16)
        first = np.argmax(diff != 0, axis=1)
        gt = diff[np.arange(len(x)), first] > 0
        out[gt, 31] |= 0x80
        return out
    h_te, st_h = te.te_sw_map_batch(sw_xy(r["input"]), to_te=True)
    g_te, st_g = te.te_sw_map_batch(sw_xy(r["output"]), to_te=True)
    p_te, st_p = te.te_sw_map_batch(sw_xy(pk), to_te=True)
    assert (st_h == 0).all() and (st_g == 0).all() and (st_p == 0).all()
    rt = te.ietf_prove_batch(sk, inputs=te_enc(h_te), ad=b"scale")
    assert (rt["status"] == 0).all()
    assert (rt["output"] == te_enc(g_te)).all() and (rt["pk"] == te_enc(p_te)).all()
    te.close()


@pytest.mark.gpu
def test_gpu_multi_context(gpu):
    from ark_ec_vrfs_amd import BandersnatchSwSha512Tai, Context, ietf_prove_batch_multi, ietf_verify_batch_multi
    n = 700
    seeds = _u8(b"m%07d" % i for i in range(n))
    sk, pk = gpu.secret_from_seed_batch(seeds)
    msgs = np.frombuffer(b"".join(b"%016d" % i for i in range(n)), np.uint8).reshape(n, 16)
    r = gpu.ietf_prove_batch(sk, msgs=msgs, ad=b"")
    c2 = Context(0, BandersnatchSwSha512Tai)
    rm = ietf_prove_batch_multi([gpu, c2], sk, msgs=msgs, ad=b"")
    assert all((rm[k] == r[k]).all() for k in ("output", "c", "s", "pk", "input"))
    s2 = r["s"].copy(); s2[3::11, 0] ^= 1
    st = gpu.ietf_verify_batch(pk, r["input"], r["output"], r["c"], s2, ad=b"")
    stm = ietf_verify_batch_multi([gpu, c2], pk, r["input"], r["output"], r["c"], s2, ad=b"")
    assert (stm == st).all() and (st[3::11] == 1).all() and st.sum() == len(st[3::11])
    c2.close()
    # the curve-independent entry points run on this suite's contexts too: the pairing check of the ring verifier's tail
    import json
    import os as _os
    fx = json.load(open(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden", "pairing_items.json")))
    hx = lambda h: np.frombuffer(bytes.fromhex(h), np.uint8)
    g2 = hx(fx["shared_g2"])
    items = np.stack([hx(h) for h in fx["shared"][:4]] + [hx(h) for h in fx["shared_bad"][:2]])
    assert list(gpu.pairing_check_batch(items, g2, g2_shared=True)) == [0, 0, 0, 0, 1, 1]
@pytest.mark.gpu
def test_gpu_xy_forms_key_sets_and_msm(gpu):
    """The remaining entry points of the suite: x || y inputs (no square root) and outputs, verification against a resident
    key set, MSM over Weierstrass bases -- each against the compressed path or the oracle."""
    from ark_ec_vrfs_amd import BandersnatchSha512Ell2, Context
    rnd = random.Random(77)
    n = 512
    seeds = _u8(b"x%07d" % (i % 16) for i in range(n))                  # 16 distinct keys
    sk, pk = gpu.secret_from_seed_batch(seeds)
    msgs = np.frombuffer(b"".join(b"%024d" % i for i in range(n)), np.uint8).reshape(n, 24)
    ad = b"xy"
    r = gpu.ietf_prove_batch(sk, msgs=msgs, ad=ad)
    t2 = _torsion2()
    # ---- x || y inputs ----
    def sw_xy(enc):
        stv, xyv = gpu.point_validate_batch(enc, want_xy=True)
        assert (stv == 0).all()
        return xyv
    xs = [sw_xy(a) for a in (pk, r["input"], r["output"])]
    assert xs[2][5].tobytes() == xy(bo.point_decode(r["output"][5].tobytes())[1])
    assert (gpu.ietf_verify_batch_affine(*xs, r["c"], r["s"], ad=ad) == 0).all()
    bad = [a.copy() for a in xs]
    cc, ss = r["c"].copy(), r["s"].copy()
    want = np.zeros(n, np.uint8)
    for i in range(0, n, 7):
        kind = (i // 7) % 6
        if kind == 0: ss[i, 1] ^= 2; want[i] = 1
        elif kind == 1: bad[2][i, 40] ^= 1; want[i] = 2                                            # y off the curve
        elif kind == 2: bad[0][i, :32] = np.frombuffer(le(Q + 1), np.uint8); want[i] = 2           # x >= q
        elif kind == 3:                                                                             # on the curve, outside the subgroup
            bad[1][i] = np.frombuffer(xy(bo.add(bo.point_decode(r["input"][i].tobytes())[1], t2[i % 3])), np.uint8); want[i] = 2
        elif kind == 4: bad[2][i] = xs[2][(i + 1) % n]; want[i] = 1
        elif kind == 5: bad[0][i] = np.frombuffer(xy(t2[0]), np.uint8); want[i] = 2                # y = 0
    got = gpu.ietf_verify_batch_affine(*bad, cc, ss, ad=ad)
    assert (got == want).all(), (got[::7], want[::7])
    # ---- x || y outputs, canonical and in arkworks' in-memory Montgomery form; they feed the x || y verifier ----
    for flags in (gpu.PROVE_POINTS_AFFINE, gpu.PROVE_POINTS_AFFINE | gpu.COORDS_MONT256):
        gpu.set_flags(flags)
        ra = gpu.ietf_prove_batch(sk, msgs=msgs, ad=ad)
        pa = gpu.pedersen_prove_batch(sk[:64], msgs=msgs[:64], ad=ad)
        assert ra["output"].shape == (n, 64) and (ra["c"] == r["c"]).all() and (ra["s"] == r["s"]).all()
        if flags == gpu.PROVE_POINTS_AFFINE:
            assert (ra["output"] == xs[2]).all() and (ra["pk"] == xs[0]).all()
        else:
            R256 = (1 << 256) % Q
            g5 = bo.point_decode(r["output"][5].tobytes())[1]
            assert ra["output"][5].tobytes() == le(g5[0] * R256 % Q) + le(g5[1] * R256 % Q)
        h_xy = sw_xy(r["input"]) if flags == gpu.PROVE_POINTS_AFFINE else None
        if h_xy is not None:
            assert (gpu.ietf_verify_batch_affine(ra["pk"], h_xy, ra["output"], ra["c"], ra["s"], ad=ad) == 0).all()
        else:
            stv, hm = gpu.point_validate_batch(r["input"], want_xy=True)            # x || y out follows the flag too
            assert (gpu.ietf_verify_batch_affine(ra["pk"], hm, ra["output"], ra["c"], ra["s"], ad=ad) == 0).all()
        gpu.set_flags(0)
        pc = gpu.pedersen_prove_batch(sk[:64], msgs=msgs[:64], ad=ad)
        assert (pa["s"] == pc["s"]).all() and (pa["sb"] == pc["sb"]).all()
        if flags == gpu.PROVE_POINTS_AFFINE:
            for k_ in ("output", "pk_com", "r", "ok"):
                assert (pa[k_] == sw_xy(pc[k_])).all(), k_
    # ---- key sets ----
    keys = pk[:16].copy()
    keys[3, 32] |= 0x2a                                                      # junk under the flags: accepted, hashed canonically
    keys[7] = np.frombuffer(bo.point_encode(bo.add(bo.point_decode(pk[7].tobytes())[1], t2[1])), np.uint8)     # outside the subgroup
    ks, kst = gpu.keyset_create(keys)
    assert list(kst) == [0] * 7 + [2] + [0] * 8
    idx = (np.arange(n) % 16).astype(np.uint32)
    idx[11] = 99                                                             # no such key
    s2 = r["s"].copy(); s2[5::16, 0] ^= 1
    got = gpu.ietf_verify_batch_keyed(ks, idx, r["input"], r["output"], r["c"], s2, ad=ad)
    want = gpu.ietf_verify_batch(pk, r["input"], r["output"], r["c"], s2, ad=ad)
    want[idx == 7] = 2
    want[11] = 2
    assert (got == want).all() and (want[5::16] == 1).all()
    ks.close()
    # ---- MSM over Weierstrass bases ----
    m = 200
    ks_ = [rnd.randrange(1, R) for _ in range(m)]
    pts = [bo.mul(rnd.randrange(1, R), bo.G) for _ in range(m)]
    acc = None
    for k_, p_ in zip(ks_, pts):
        acc = bo.add(acc, bo.mul(k_, p_))
    enc, sxy = gpu.msm(_u8(xy(p_) for p_ in pts), _u8(le(k_) for k_ in ks_))
    assert enc == bo.point_encode(acc) and sxy == xy(acc)
    enc, sxy = gpu.msm(_u8([xy(pts[0]), xy(bo.neg(pts[0]))]), _u8([le(5), le(5)]))
    assert enc == bytes(32) + b"\x40" and sxy == bytes(64)
    enc, sxy = gpu.msm(np.zeros((0, 64), np.uint8), np.zeros((0, 32), np.uint8))
    assert enc == bytes(32) + b"\x40"
    for bad_base in (xy(t2[0]), le(Q) + le(1), xy((pts[0][0], (pts[0][1] + 1) % Q))):
        with pytest.raises(Exception):
            gpu.msm(_u8([xy(pts[0]), bad_base]), _u8([le(1), le(1)]))
    # 2^14 bases: equal to the Edwards suite's MSM over the mapped bases, mapped back
    te = Context(0, BandersnatchSha512Ell2)
    big = 1 << 14
    bsk, bpk = gpu.secret_from_seed_batch(np.arange(big, dtype=np.uint64).view(np.uint8).reshape(big, 8))
    bases = sw_xy(bpk)
    scal = bsk[::-1].copy()
    enc, sxy = gpu.msm(bases, scal)
    te_bases, stt = te.te_sw_map_batch(bases, to_te=True)
    assert (stt == 0).all()
    _, te_sum = te.msm(te_bases, scal)
    back, stb = te.te_sw_map_batch(np.frombuffer(te_sum, np.uint8).reshape(1, 64), to_te=False)
    assert stb[0] == 0 and back[0].tobytes() == sxy and enc[:32] == sxy[:32]
    te.close()
