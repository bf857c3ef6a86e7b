"""`utils::te_sw_map::{te_to_sw, sw_to_te}` (/root/reference src/lib.rs:14 `utils`): the twisted-Edwards <-> short-Weierstrass
point map, `vrfhip_te_sw_map_batch`.

CPU tier: the Python restatement (oracle/vrf_oracle.py) against the four arkworks coefficients of Bandersnatch (what pins the
map's scaling), the curve equation, the group laws of both models, upstream's None cases.
GPU tier (-m gpu): the HIP path through the C ABI against the restatement on every twisted-Edwards suite, both directions,
both coordinate formats, the None cases and non-canonical coordinates."""
import random

import numpy as np
import pytest

from oracle import vrf_oracle as o

SUITES = {"bandersnatch": o.BANDERSNATCH, "jubjub": o.jubjub_params(), "ed25519": o.ed25519_params(),
          "babyjubjub": o.baby_jubjub_params()}
# ark-ed-on-bls12-381-bandersnatch: `MontCurveConfig::{COEFF_A, COEFF_B}` and `SWCurveConfig::{COEFF_A, COEFF_B}`
ARK_BANDERSNATCH_MONT_A = 29978822694968839326280996386011761570173833766074948509196803838190355340952
ARK_BANDERSNATCH_MONT_B = 25465760566081946422412445027709227188579564747101592991722834452325077642517
ARK_BANDERSNATCH_SW_A = 10773120815616481058602537765553212789256758185246796157495669123169359657269
ARK_BANDERSNATCH_SW_B = 29569587568322301171008055308580903175558631321415017492731745847794083609535


def le(x):
    return int(x).to_bytes(32, "little")


def xy(P):
    return np.frombuffer(le(P[0]) + le(P[1]), np.uint8)


def un(row):
    b = bytes(row)
    return (int.from_bytes(b[:32], "little"), int.from_bytes(b[32:], "little"))


def test_the_models_coefficients_are_arkworks():
    """The Montgomery and short-Weierstrass coefficients the map implies for Bandersnatch are the ones arkworks states for
    its MontCurveConfig and SWCurveConfig: the scaling (u + A/3)/B, v/B is upstream's."""
    assert o.te_sw_constants(o.BANDERSNATCH) == (ARK_BANDERSNATCH_MONT_A, ARK_BANDERSNATCH_MONT_B, ARK_BANDERSNATCH_SW_A,
                                                  ARK_BANDERSNATCH_SW_B)


@pytest.mark.parametrize("name", list(SUITES))
def test_map_is_an_isomorphism_onto_the_weierstrass_form(name):
    S = SUITES[name]
    q = S.q
    _, _, a2, b2 = o.te_sw_constants(S)
    rnd = random.Random(name)
    G = (S.gx, S.gy)
    W = o.te_to_sw(S, G)
    for _ in range(12):
        P, Q = o.te_mul(S, rnd.randrange(1, S.r), G), o.te_mul(S, rnd.randrange(1, S.r), G)
        wp, wq = o.te_to_sw(S, P), o.te_to_sw(S, Q)
        assert (wp[1] ** 2 - (wp[0] ** 3 + a2 * wp[0] + b2)) % q == 0                 # on y^2 = x^3 + a' x + b'
        assert o.sw_to_te(S, wp) == P                                                   # round trip
        assert o.te_to_sw(S, o.te_add(S, P, Q)) == o.sw_add(S, wp, wq)                  # homomorphism
        assert o.te_to_sw(S, o.te_add(S, P, P)) == o.sw_add(S, wp, wp)
        assert o.te_to_sw(S, o.te_neg(S, P)) == (wp[0], (-wp[1]) % q)
    # r * G = infinity on the Weierstrass side too (double-and-add with the oracle's law)
    acc, base, k = None, W, S.r
    while k:
        if k & 1:
            acc = o.sw_add(S, acc, base)
        base, k = o.sw_add(S, base, base), k >> 1
    assert acc is None
    # upstream's None cases: the identity and the point of order 2 going out; y = 0 (the 2-torsion) coming back
    assert o.te_to_sw(S, (0, 1)) is None and o.te_to_sw(S, (0, q - 1)) is None
    A, B, _, _ = o.te_sw_constants(S)
    x2 = A * o.finv(3, q) * o.finv(B, q) % q                  # SW image of the Montgomery point (0, 0)
    assert (x2 ** 3 + a2 * x2 + b2) % q == 0 and o.sw_to_te(S, (x2, 0)) is None
    xm1 = (A * o.finv(3, q) - 1) * o.finv(B, q) % q           # B x - A/3 = -1
    assert o.sw_to_te(S, (xm1, 5)) is None


GPU_SUITES = {"bandersnatch": "BandersnatchSha512Ell2", "jubjub": "JubJubSha512Tai", "ed25519": "Ed25519Sha512Tai",
              "babyjubjub": "BabyJubJubSha512Tai"}


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(GPU_SUITES))
def test_gpu_te_sw_map_equals_the_restatement(name):
    import ark_ec_vrfs_amd as pkg
    S = SUITES[name]
    q = S.q
    ctx = pkg.Context(0, getattr(pkg, GPU_SUITES[name]), test_blinding_base=True)
    rnd = random.Random("gpu" + name)
    G = (S.gx, S.gy)
    pts = [o.te_mul(S, rnd.randrange(1, S.r), G) for _ in range(300)]
    pts += [(0, 1), (0, q - 1)]                                                 # None: identity, order 2
    pts += [(rnd.randrange(q), rnd.randrange(q)) for _ in range(40)]            # off the curve: mapped all the same
    te = np.stack([xy(P) for P in pts])
    bad = te[:3].copy()
    bad[0, :32] = np.frombuffer(le(q), np.uint8)                                # x = q: not canonical
    bad[1, 32:] = 0xFF
    te_all = np.concatenate([te, bad])
    out, st = ctx.te_sw_map_batch(te_all)
    want = [o.te_to_sw(S, P) for P in pts] + [None, None, o.te_to_sw(S, pts[2])]
    for i, w in enumerate(want):
        if w is None:
            assert st[i] == 2 and not out[i].any(), i
        else:
            assert st[i] == 0 and un(out[i]) == w, i
    # and back: the images return to the points they came from; the restatement agrees on arbitrary pairs too
    good = [i for i, w in enumerate(want) if w is not None]
    back, st2 = ctx.te_sw_map_batch(out[good], to_te=True)
    for j, i in enumerate(good):
        w = o.sw_to_te(S, want[i])
        if w is None:
            assert st2[j] == 2 and not back[j].any()
        else:
            assert st2[j] == 0 and un(back[j]) == w
            if i < 300:
                assert (back[j] == te_all[i]).all()
    A, B, a2, b2 = o.te_sw_constants(S)
    x2, xm1 = A * o.finv(3, q) * o.finv(B, q) % q, (A * o.finv(3, q) - 1) * o.finv(B, q) % q
    sw = [(x2, 0), (xm1, 5)] + [(rnd.randrange(q), rnd.randrange(1, q)) for _ in range(60)]
    got, st3 = ctx.te_sw_map_batch(np.stack([xy(P) for P in sw]), to_te=True)
    for i, P in enumerate(sw):
        w = o.sw_to_te(S, P)
        assert (st3[i] == 2 and not got[i].any()) if w is None else (st3[i] == 0 and un(got[i]) == w), i
    assert st3[0] == 2 and st3[1] == 2
    # arkworks' in-memory limbs (VRFHIP_FLAG_COORDS_MONT256): the same map on x * 2^256 mod q
    R = pow(2, 256, q)
    ctx.set_flags(pkg.Context.COORDS_MONT256)
    mont = np.stack([xy((P[0] * R % q, P[1] * R % q)) for P in pts[:64]])
    outm, stm = ctx.te_sw_map_batch(mont)
    for i in range(64):
        w = want[i]
        assert stm[i] == 0 and un(outm[i]) == (w[0] * R % q, w[1] * R % q)
    ctx.set_flags(0)
    # empty batch, and the refusal on the suite that is not twisted Edwards
    e, es = ctx.te_sw_map_batch(np.zeros((0, 64), np.uint8))
    assert e.shape == (0, 64) and es.shape == (0,)
    ctx.close()
    if name == "bandersnatch":
        p = pkg.Context(0, pkg.Secp256r1Sha256Tai, test_blinding_base=True)
        with pytest.raises(Exception):
            p.te_sw_map_batch(te[:2])
        p.close()
