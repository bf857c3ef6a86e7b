"""Ed25519 (2^255 - 19) and Baby-JubJub (BN254 Fr): SURVEY.md section 8 row f4.

CPU tier: the device headers compiled for the host per base field (tests/hostsim/hostsim_suite.hip, -DVRF_FIELD=1|2)
against Python big ints, the Python and C oracles and -- for the try-and-increment / challenge / output-hash code paths --
the published RFC 9381 ECVRF-EDWARDS25519-SHA512-TAI vectors (tests/golden/rfc9381_edwards25519_sha512_tai.json).
GPU tier (-m gpu): the HIP path through the C ABI against the C oracle: prove bytes, verify statuses, tampering,
non-subgroup / small-order forgeries, Pedersen, the batched Pedersen verifier, MSM, the RFC vectors."""
import ctypes
import json
import os
import random
import subprocess

import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import vrf_oracle as o

HERE = os.path.dirname(os.path.abspath(__file__))
HS = os.path.join(HERE, "hostsim")
NCPU = min(8, os.cpu_count() or 1)
RFC = json.load(open(os.path.join(HERE, "golden", "rfc9381_edwards25519_sha512_tai.json")))

ED, BJ = o.ed25519_params(), o.baby_jubjub_params()
SUITES = {"ed25519": (ED, 1, 3), "babyjubjub": (BJ, 2, 4)}        # params, field, suite / curve id


def le(x):
    return int(x).to_bytes(32, "little")


def xy(P):
    return le(P[0]) + le(P[1])


@pytest.fixture(scope="module", params=list(SUITES))
def hx(request):
    S, field, sid = SUITES[request.param]
    so = os.path.join(HS, "libhostsim_f%d.so" % field)
    subprocess.run(["make", "-C", HS, "-j4", os.path.basename(so)], check=True, stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(so)
    lib.hx_init()
    return lib, S, sid


def _call(f, *a):
    r = ctypes.create_string_buffer(32)
    ret = f(*[le(x) for x in a], r)
    return int.from_bytes(r.raw, "little"), ret


def _torsion8(S):
    """A generator of the rational 8-torsion (the curves here have cyclic 2-power torsion of order 8)."""
    y = 2
    while True:
        P = o.point_decode(S, le(y))
        if P is not None:
            T = o.te_mul(S, S.r, P)
            if o.te_mul(S, 4, T) != (0, 1):
                return T
        y += 1


# ---------------------------------------------------------------------------------------------- CPU: constants
def test_curve_constants_are_what_they_claim():
    """RFC 8032 5.1 / ark-ed-on-bn254: on the curve, generator of prime order r, #E = 8 r by Hasse, blinding bases in the
    subgroup and independent-looking (not a small multiple of G)."""
    for S in (ED, BJ):
        G, B = (S.gx, S.gy), (S.bx, S.by)
        assert o.te_is_on_curve(S, G) and o.te_mul(S, S.r, G) == (0, 1) and o.te_mul(S, S.cofactor, G) != (0, 1)
        assert o.te_in_prime_subgroup(S, B) and B != (0, 1)
        assert all(o.te_mul(S, k, G) != B for k in range(1, 64))
        assert abs(8 * S.r - (S.q + 1)) <= 2 * int(S.q ** 0.5) + 2                 # Hasse interval pins the group order
        assert pow(S.r, 1, 2) == 1 and all(S.r % p for p in (3, 5, 7, 11, 13, 17, 19, 23))
        T8 = _torsion8(S)
        assert o.te_mul(S, 8, T8) == (0, 1) and o.te_mul(S, 4, T8) == (0, S.q - 1)
    assert ED.gy == 4 * pow(5, ED.q - 2, ED.q) % ED.q and ED.gx % 2 == 0           # RFC 8032: y = 4/5, x even
    assert ED.d == (-121665 * pow(121666, ED.q - 2, ED.q)) % ED.q
    assert (BJ.d * 168700 - 168696) % BJ.q == 0 and BJ.a == 1


def test_rfc9381_vectors_pin_the_python_oracle():
    """ECVRF_prove from (SK, alpha) through the oracle's own functions reproduces RFC 9381 B.3 bit for bit; the proof verifies
    through ietf_verify (the code path of every suite) and a tampered one does not."""
    S = o.ed25519_rfc9381_params()
    for v in RFC["vectors"]:
        out = o.rfc9381_ed25519_prove(bytes.fromhex(v["sk"]), bytes.fromhex(v["alpha"]))
        for k in ("pk", "x", "h", "k", "u", "v", "pi", "beta"):
            if k in v:
                assert out[k].hex() == v[k], k
        pi = bytes.fromhex(v["pi"])
        Y, H, G = (o.point_decode_checked(S, b) for b in (bytes.fromhex(v["pk"]), out["h"], pi[:32]))
        c, s = int.from_bytes(pi[32:48], "little"), int.from_bytes(pi[48:], "little")
        assert o.ietf_verify(S, Y, H, G, b"", c, s)
        assert not o.ietf_verify(S, Y, H, G, b"", c, s ^ 1) and not o.ietf_verify(S, Y, H, G, b"x", c, s)


def test_c_oracle_equals_python_oracle_and_verifies_rfc9381():
    for name, (S, _, sid) in SUITES.items():
        co.set_suite(sid)
        try:
            n = 5
            sks = [o.secret_from_seed(S, o.synth_seed(i)) for i in range(n)]
            sk = np.stack([np.frombuffer(le(k), np.uint8) for k in sks])
            msg = np.stack([np.frombuffer(o.synth_msg(i), np.uint8) for i in range(n)])
            r = co.ietf_prove_batch(sk, msgs=msg, ad=b"ad")
            p = co.pedersen_prove_batch(sk, msgs=msg, ad=b"ad")
            for i in range(n):
                H = o.data_to_point(S, o.synth_msg(i))
                g, c, s = o.ietf_prove(S, sks[i], H, b"ad")
                assert r["input"][i].tobytes() == o.point_encode(S, H) and r["output"][i].tobytes() == o.point_encode(S, g)
                assert r["c"][i].tobytes() == le(c) and r["s"][i].tobytes() == le(s)
                assert c < (1 << (8 * S.challenge_len))
                g2, (pc, R, Ok, s_, sb), b = o.pedersen_prove(S, sks[i], H, b"ad")
                assert p["pk_com"][i].tobytes() == o.point_encode(S, pc) and p["r"][i].tobytes() == o.point_encode(S, R)
                assert p["s"][i].tobytes() == le(s_) and p["sb"][i].tobytes() == le(sb) and p["blinding"][i].tobytes() == le(b)
            assert (co.ietf_verify_batch(r["pk"], r["input"], r["output"], r["c"], r["s"], b"ad") == 0).all()
            assert (co.pedersen_verify_batch(p["input"], p["output"], p["pk_com"], p["r"], p["ok"], p["s"], p["sb"], b"ad") == 0).all()
        finally:
            co.set_suite(1)
    S = o.ed25519_rfc9381_params()
    co.set_suite_desc(3, b"\x03", b"", xy((S.gx, S.gy)), xy((S.bx, S.by)), challenge_len=16, flags=7)
    try:
        for v in RFC["vectors"]:
            pk, alpha, pi = bytes.fromhex(v["pk"]), bytes.fromhex(v["alpha"]), bytes.fromhex(v["pi"])
            h = co.hash_to_curve(pk + alpha)
            f = lambda b: np.frombuffer(b, np.uint8)
            assert co.ietf_verify_batch(f(pk), f(h), f(pi[:32]), f(pi[32:48] + bytes(16)), f(pi[48:]), b"")[0] == 0
            assert co.output_hash(pi[:32]).hex() == v["beta"]
    finally:
        co.set_suite(1)


def _posdivsteps(q, x, cap=4000):
    """Steps the positive-divsteps pair (fe.cuh jacobi_limbs) needs to reach f = g = 1."""
    f, g, eta, n = q, x, -1, 0
    while (f != 1 or g != 1) and n < cap:
        if g & 1 and eta < 0:
            f, g, eta = g, f, -eta
        if g & 1:
            g += f
        g >>= 1
        eta -= 1
        n += 1
    return n


# ---------------------------------------------------------------------------------------------- CPU: device headers on the host
def test_field_ops_against_python_ints(hx):
    lib, S, _ = hx
    Q = S.q
    rnd = random.Random(1)
    edge = [0, 1, 2, 19, Q - 1, Q - 2, Q, Q + 1, Q + 18, (1 << 255) - 1, (1 << 256) - 1, (1 << 255), (1 << 29) - 1, 1 << 232,
            (1 << 256) - 38, 2 * Q, 2 * Q + 1]
    for it in range(3000):
        x = rnd.choice(edge) if it % 7 == 0 else rnd.getrandbits(256)
        y = rnd.choice(edge) if it % 11 == 0 else rnd.getrandbits(256)
        assert _call(lib.hx_fe_mul, x, y)[0] == x * y % Q
        assert _call(lib.hx_fe_sqr, x)[0] == x * x % Q
        assert _call(lib.hx_fe_add, x, y)[0] == (x + y) % Q
        assert _call(lib.hx_fe_sub, x, y)[0] == (x - y) % Q
        a, b = x % Q, y % Q
        s, d, ab = a + b, a - b, a * b
        u2, w = 5 * ab + s * d - b, a - (ab + b)
        assert _call(lib.hx_fe_lazy, x, y)[0] == (u2 * w) % Q      # lazy-limb bound stress
        assert _call(lib.hx_fe_wide, x, y)[0] == 6 * a * b % Q
    for x in edge + [rnd.getrandbits(512) for _ in range(300)] + [(1 << 512) - 1, (1 << 384) - 1]:
        r = ctypes.create_string_buffer(32)
        lib.hx_from_u512(int(x % (1 << 512)).to_bytes(64, "little"), r)
        assert int.from_bytes(r.raw, "little") == (x % (1 << 512)) % Q
    for x in [0, 1, Q - 1] + [rnd.randrange(Q) for _ in range(200)]:
        c, bk = ctypes.create_string_buffer(32), ctypes.create_string_buffer(32)
        m = x * (1 << 256) % Q
        lib.hx_mont256_roundtrip(le(m), c, bk)
        assert int.from_bytes(c.raw, "little") == x and int.from_bytes(bk.raw, "little") == m


def test_inverse_sqrt_and_jacobi(hx):
    lib, S, _ = hx
    Q = S.q
    rnd = random.Random(2)
    z = 2 if Q % 8 == 5 else 5                                   # the field's non-residue (gen_constants.py)
    assert o.legendre(z, Q) == -1
    vals = [0, 1, 2, 3, Q - 1, Q - 2, (Q - 1) // 2, (Q + 1) // 2, Q + 5] + [1 << k for k in range(0, 256, 11)] + \
        [Q - (1 << k) for k in range(0, 250, 23)] + [rnd.getrandbits(256) for _ in range(1500)] + [rnd.getrandbits(40) for _ in range(50)]
    for x in vals:
        assert _call(lib.hx_fe_inv, x)[0] == pow(x % Q, Q - 2, Q), hex(x)
    for x in vals[:60]:
        assert _call(lib.hx_fe_inv_pow, x)[0] == pow(x % Q, Q - 2, Q), hex(x)
    vals += [pow(rnd.getrandbits(255), 2, Q) for _ in range(300)] + [z * pow(rnd.getrandbits(255), 2, Q) % Q for _ in range(300)]
    vals += [pow(rnd.getrandbits(255), 4, Q) for _ in range(50)] + [Q - pow(rnd.getrandbits(255), 2, Q) for _ in range(100)]
    for x in vals:
        xm = x % Q
        r, sq = _call(lib.hx_fe_sqrt, x)
        if xm == 0:
            assert r == 0
        elif o.legendre(xm, Q) == 1:
            assert sq == 1 and r * r % Q == xm, hex(x)
        else:
            assert sq == 0 and r * r % Q == z * xm % Q, hex(x)       # sqrt(Z * w)
        # the divsteps loop may give up (2) on the few values that converge slowly -- q - 2 needs 1564 steps for
        # 2^255 - 19 against 1160 allowed -- and fe_is_nonzero_square then falls back to the exponentiation
        j = lib.hx_fe_jacobi(le(x))
        assert j == o.legendre(xm, Q) or (j == 2 and _posdivsteps(Q, xm) > 39 * 29), hex(x)
        assert lib.hx_fe_is_nonzero_square(le(x)) == (1 if o.legendre(xm, Q) == 1 else 0)


def test_group_law_decode_and_subgroup_test(hx):
    lib, S, _ = hx
    rnd = random.Random(3)
    G = (S.gx, S.gy)
    T8 = _torsion8(S)
    pts = [o.te_mul(S, rnd.randrange(1, S.r), G) for _ in range(12)] + [(0, 1), G, o.te_neg(S, G)]
    buf = lambda: ctypes.create_string_buffer(32)
    for P in pts:
        for Qp in (pts[0], P, o.te_neg(S, P), (0, 1), o.te_add(S, P, T8)):
            ox, oy = buf(), buf()
            lib.hx_point_add(le(P[0]), le(P[1]), le(Qp[0]), le(Qp[1]), ox, oy)
            assert (int.from_bytes(ox.raw, "little"), int.from_bytes(oy.raw, "little")) == o.te_add(S, P, Qp)
        ox, oy = buf(), buf()
        lib.hx_point_dbl(le(P[0]), le(P[1]), ox, oy)
        assert (int.from_bytes(ox.raw, "little"), int.from_bytes(oy.raw, "little")) == o.te_add(S, P, P)
        k = rnd.choice([0, 1, 2, S.r - 1, rnd.randrange(S.r), rnd.randrange(S.r)])
        lib.hx_scalar_mul(le(k), le(P[0]), le(P[1]), ox, oy)
        assert (int.from_bytes(ox.raw, "little"), int.from_bytes(oy.raw, "little")) == o.te_mul(S, k, P)
        # decode of the encoding, both sign flags; the subgroup test on every coset of the 8-torsion
        enc = o.point_encode(S, P)
        x, y = buf(), buf()
        assert lib.hx_decode(enc, x, y) == 1 and (int.from_bytes(x.raw, "little"), int.from_bytes(y.raw, "little")) == P
        for j in range(8):
            Pj = o.te_add(S, P, o.te_mul(S, j, T8))
            want = 0 if o.te_in_prime_subgroup(S, Pj) else 2
            assert lib.hx_decode_checked(o.point_encode(S, Pj)) == want, (j, want)
    # undecodable: y >= q, non-square x^2; random strings agree with the oracle
    assert lib.hx_decode_checked(le(S.q)) == 2 and lib.hx_decode_checked(le(S.q + 1)) == 2
    for _ in range(300):
        raw = rnd.getrandbits(256).to_bytes(32, "little")
        want = 0 if o.point_decode_checked(S, raw) is not None else 2
        assert lib.hx_decode_checked(raw) == want
        P = o.point_decode(S, raw)
        x, y = buf(), buf()
        got = lib.hx_decode(raw, x, y)
        assert got == (1 if P is not None else 0)
        if P is not None:
            assert (int.from_bytes(x.raw, "little"), int.from_bytes(y.raw, "little")) == P


def test_subgroup_test_equals_r_times_p_on_every_coset_and_small_order_point(hx):
    """The cheap membership tests (Ed25519: one halving + the order-4 Tate pairing; Baby-JubJub: the order-8 pairing)
    against r * P = O through the compiled group law and through the oracle: 40 subgroup points on all 8 cosets of the
    torsion, the 8 torsion points themselves, points with special coordinates (y = 0, the generator, its negation)."""
    lib, S, _ = hx
    rnd = random.Random(11)
    G, T8 = (S.gx, S.gy), _torsion8(S)
    tors = [o.te_mul(S, j, T8) for j in range(8)]
    pts = [o.te_mul(S, rnd.randrange(1, S.r), G) for _ in range(40)] + [(0, 1), G, o.te_neg(S, G)]
    for P in pts:
        for j, Tj in enumerate(tors):
            Pj = o.te_add(S, P, Tj)
            want = 3 if (j == 0) else 0
            assert o.te_in_prime_subgroup(S, Pj) == (j == 0)
            assert lib.hx_subgroup_both(le(Pj[0]), le(Pj[1])) == want, (P, j)
    # every coset representative above has a random-looking y; points decoded from random strings have random cosets
    seen = set()
    for _ in range(200):
        P = o.point_decode(S, rnd.getrandbits(256).to_bytes(32, "little"))
        if P is None:
            continue
        got = lib.hx_subgroup_both(le(P[0]), le(P[1]))
        assert got in (0, 3) and (got == 3) == o.te_in_prime_subgroup(S, P)
        seen.add(got)
    assert seen == {0, 3}


def test_schemes_on_the_host_build_equal_the_oracle(hx):
    lib, S, _ = hx
    lib.hx_set_check_mask(15)
    buf = lambda n=32: ctypes.create_string_buffer(n)
    for i in range(6):
        seed, msg, ad = o.synth_seed(i), o.synth_msg(i)[: 5 + 7 * i], b"ad" * i
        sk = o.secret_from_seed(S, seed)
        b = buf(); lib.hx_secret_from_seed(seed, len(seed), b); assert b.raw == le(sk)
        b = buf(); lib.hx_public(le(sk), b); assert b.raw == o.point_encode(S, o.public_from_secret(S, sk))
        H = o.data_to_point(S, msg)
        b = buf(); lib.hx_hash_to_curve(msg, len(msg), b); assert b.raw == o.point_encode(S, H)
        g, c, s = o.ietf_prove(S, sk, H, ad)
        out = buf(160)
        assert lib.hx_prove(0, le(sk), msg, len(msg), ad, len(ad), out) == 1
        pk = o.point_encode(S, o.public_from_secret(S, sk))
        assert out.raw == o.point_encode(S, g) + le(c) + le(s) + pk + o.point_encode(S, H)
        b = buf(64); lib.hx_output_hash(o.point_encode(S, g), b); assert b.raw == o.output_hash(S, g)
        args = [pk, o.point_encode(S, H), o.point_encode(S, g), le(c), le(s)]
        assert lib.hx_ietf_verify(*args, ad, len(ad)) == 0
        assert lib.hx_ietf_verify(*args[:4], le(s ^ 1), ad, len(ad)) == 1
        assert lib.hx_ietf_verify(*args[:4], le(S.r), ad, len(ad)) == 2           # s not canonical
        assert lib.hx_ietf_verify(*args, ad + b"x", len(ad) + 1) == 1
        T2 = (0, S.q - 1)
        shifted = o.point_encode(S, o.te_add(S, g, T2))                                # Gamma + the 2-torsion point: not in the subgroup
        assert lib.hx_ietf_verify(args[0], args[1], shifted, args[3], args[4], ad, len(ad)) == 2
        g2, (pc, R, Ok, s_, sb), bl = o.pedersen_prove(S, sk, H, ad)
        out = buf(224)
        assert lib.hx_prove(1, le(sk), msg, len(msg), ad, len(ad), out) == 1
        enc = lambda P: o.point_encode(S, P)
        assert out.raw == enc(g2) + enc(pc) + enc(R) + enc(Ok) + le(s_) + le(sb) + le(bl)
        proof = enc(pc) + enc(R) + enc(Ok) + le(s_) + le(sb)
        assert lib.hx_pedersen_verify(enc(H), enc(g2), proof, ad, len(ad)) == 0
        assert lib.hx_pedersen_verify(enc(H), enc(g2), proof[:128] + le(sb ^ 1), ad, len(ad)) == 1
    lib.hx_set_check_mask(0)


def test_rfc9381_vectors_through_the_device_headers():
    """The kernels' own decode / try-and-increment / Straus / challenge / output-hash code, compiled for the host, run as RFC
    9381's suite 0x03 (descriptor: suite string 03, 16-byte little-endian challenge, parity sign bit, cofactor in the
    output hash; the public key prepended to alpha): hash-to-curve, verification and beta of the published vectors."""
    so = os.path.join(HS, "libhostsim_f1.so")
    subprocess.run(["make", "-C", HS, "-j4", "libhostsim_f1.so"], check=True, stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(so)
    S = o.ed25519_rfc9381_params()
    lib.hx_configure(b"\x03", 1, 16, 7, xy((S.gx, S.gy)), xy((S.bx, S.by)))
    lib.hx_set_check_mask(15)
    try:
        for v in RFC["vectors"]:
            pk, alpha, pi = bytes.fromhex(v["pk"]), bytes.fromhex(v["alpha"]), bytes.fromhex(v["pi"])
            h = ctypes.create_string_buffer(32)
            lib.hx_hash_to_curve(pk + alpha, len(pk + alpha), h)
            if "h" in v:
                assert h.raw.hex() == v["h"]
            assert lib.hx_tai_first_decodable(pk + alpha, len(pk + alpha)) <= v["ctr"]
            c32 = pi[32:48] + bytes(16)
            assert lib.hx_ietf_verify(pk, h.raw, pi[:32], c32, pi[48:], b"", 0) == 0
            assert lib.hx_ietf_verify(pk, h.raw, pi[:32], c32, le(int.from_bytes(pi[48:], "little") ^ 2), b"", 0) == 1
            b = ctypes.create_string_buffer(64)
            lib.hx_output_hash(pi[:32], b)
            assert b.raw.hex() == v["beta"]
    finally:
        lib.hx_configure(ED.suite_id, len(ED.suite_id), 16, 0, xy((ED.gx, ED.gy)), xy((ED.bx, ED.by)))
        lib.hx_set_check_mask(0)


# ---------------------------------------------------------------------------------------------- GPU: the HIP path through the C ABI
def _suite_class(name):
    from ark_ec_vrfs_amd import BabyJubJubSha512Tai, Ed25519Sha512Tai
    return {"ed25519": Ed25519Sha512Tai, "babyjubjub": BabyJubJubSha512Tai}[name]


@pytest.fixture(scope="module", params=list(SUITES))
def gpu(request):
    from ark_ec_vrfs_amd import Context
    S, _, sid = SUITES[request.param]
    ctx = Context(0, suite=_suite_class(request.param), test_blinding_base=True)
    co.set_suite(sid)
    yield ctx, S, sid
    co.set_suite(1)
    ctx.close()


def _synth(n, start=0):
    seeds = np.arange(start, start + n, dtype=np.uint64).view(np.uint8).reshape(n, 8)
    msg = np.stack([np.frombuffer(o.synth_msg(i), np.uint8) for i in range(start, start + n)])
    return seeds, msg


@pytest.mark.gpu
@pytest.mark.parametrize("ad", [b"", bytes(range(70))])
def test_gpu_prove_verify_pedersen_equal_the_c_oracle(gpu, ad):
    ctx, S, sid = gpu
    d = ctx.desc()
    assert d.suite_id == S.suite_id and d.challenge_len == S.challenge_len and d.curve == sid and d.flags == 0
    assert d.generator == xy((S.gx, S.gy)) and d.blinding_base == xy((S.bx, S.by))       # the built-in points are the oracle's
    n = 1536
    seeds, msg = _synth(n)
    sk, pk = ctx.secret_from_seed_batch(seeds)
    for i in range(0, n, 97):
        assert sk[i].tobytes() == co.secret_from_seed(seeds[i].tobytes())
        assert pk[i].tobytes() == co.public_from_secret(sk[i].tobytes())
    ref = co.ietf_prove_batch(sk, msgs=msg, ad=ad, threads=NCPU)
    got = ctx.ietf_prove_batch(sk, msgs=msg, ad=ad)
    for k in ("output", "c", "s", "pk", "input"):
        assert (got[k] == ref[k]).all(), k
    if S.challenge_len < 32:
        assert not got["c"][:, S.challenge_len:].any()                     # `Proof::c` is CHALLENGE_LEN bytes on the wire
    h = ctx.hash_to_curve_batch(msg[:256])
    assert (h == ref["input"][:256]).all()
    beta = ctx.output_hash_batch(ref["output"][:64])
    for i in range(0, 64, 9):
        assert beta[i].tobytes() == co.output_hash(ref["output"][i].tobytes())
    # verification: statuses equal the oracle's on a batch with every kind of tampering
    pkv, inp, outp, c, s = (ref[k].copy() for k in ("pk", "input", "output", "c", "s"))
    s[::7, 2] ^= 1                                     # wrong s
    c[3::31, 0] ^= 1                                   # wrong c
    c[4::37, 20] ^= 1                                  # a c above 2^(8 CHALLENGE_LEN): the 32-byte field has room for one
    for i in range(6, n, 41):                          # c + r: the same scalar, another string
        c[i] = np.frombuffer(le(int.from_bytes(ref["c"][i].tobytes(), "little") + S.r), np.uint8)
    s[5::61] = np.frombuffer(le(S.r), np.uint8)        # s not canonical
    outp[11::67] = ref["output"][12::67][: len(outp[11::67])]     # another proof's output
    want = co.ietf_verify_batch(pkv, inp, outp, c, s, ad, threads=NCPU)
    st = ctx.ietf_verify_batch(pkv, inp, outp, c, s, ad=ad)
    assert (st == want).all() and set(np.unique(want)) == {0, 1, 2}
    assert (ctx.ietf_verify_batch(ref["pk"], ref["input"], ref["output"], ref["c"], ref["s"], ad=ad) == 0).all()
    # c + r alone: the same scalar under another string.  A 32-byte challenge is a scalar (decoded mod r, as upstream); a
    # shorter one has CHALLENGE_LEN bytes on upstream's wire, so the larger field is no proof string and fails
    cr = np.frombuffer(b"".join(le(int.from_bytes(ref["c"][i].tobytes(), "little") + S.r) for i in range(16)), np.uint8).reshape(16, 32)
    st_cr = ctx.ietf_verify_batch(ref["pk"][:16], ref["input"][:16], ref["output"][:16], cr, ref["s"][:16], ad=ad)
    assert (st_cr == (1 if S.challenge_len < 32 else 0)).all()
    assert (co.ietf_verify_batch(ref["pk"][:16], ref["input"][:16], ref["output"][:16], cr, ref["s"][:16], ad, threads=NCPU) == st_cr).all()
    assert (ctx.ietf_verify_batch(ref["pk"], ref["input"], ref["output"], ref["c"], ref["s"], ad=ad + b"!") == 1).all()
    # Pedersen
    pref = co.pedersen_prove_batch(sk, msgs=msg, ad=ad, threads=NCPU)
    pgot = ctx.pedersen_prove_batch(sk, msgs=msg, ad=ad)
    for k in ("output", "pk_com", "r", "ok", "s", "sb", "blinding"):
        assert (pgot[k] == pref[k]).all(), k
    sb_bad = pref["sb"].copy(); sb_bad[::5, 0] ^= 2
    want = co.pedersen_verify_batch(pref["input"], pref["output"], pref["pk_com"], pref["r"], pref["ok"], pref["s"], sb_bad, ad, threads=NCPU)
    st = ctx.pedersen_verify_batch(pref["input"], pref["output"], pref["pk_com"], pref["r"], pref["ok"], pref["s"], sb_bad, ad=ad)
    assert (st == want).all() and (want[::5] == 1).all() and want.sum() == len(want[::5])
    # the batched verifier (one MSM): the valid batch passes as a whole; with defects it falls back and reports them
    args = [pref[k] for k in ("input", "output", "pk_com", "r", "ok", "s")]
    st, batch_ok = ctx.pedersen_verify_batch_rlc(*args, pref["sb"], ad=ad, seed=bytes(range(32)))
    assert batch_ok and (st == 0).all()
    st, batch_ok = ctx.pedersen_verify_batch_rlc(*args, sb_bad, ad=ad, seed=bytes(range(32)))
    assert not batch_ok and (st == want).all()


@pytest.mark.gpu
def test_gpu_point_validation_and_subgroup_forgeries(gpu):
    """Strict status equality on points outside the prime-order subgroup: every coset of the 8-torsion, small-order points,
    undecodable strings; and the uniqueness forgery (an output shifted by the 2-torsion point with an even challenge)."""
    ctx, S, sid = gpu
    rnd = random.Random(5)
    G = (S.gx, S.gy)
    T8 = _torsion8(S)
    encs, want = [], []
    for _ in range(24):
        P = o.te_mul(S, rnd.randrange(1, S.r), G)
        for j in range(8):
            Pj = o.te_add(S, P, o.te_mul(S, j, T8))
            encs.append(o.point_encode(S, Pj)); want.append(0 if j == 0 else 2)
    for j in range(8):                                  # the torsion points themselves (j = 0: the identity, accepted)
        encs.append(o.point_encode(S, o.te_mul(S, j, T8))); want.append(0 if j == 0 else 2)
    for _ in range(200):
        raw = rnd.getrandbits(256).to_bytes(32, "little")
        encs.append(raw); want.append(0 if o.point_decode_checked(S, raw) is not None else 2)
    encs += [le(S.q), le(S.q + 1), le((1 << 255) - 1)]; want += [2, 2, 2]
    arr = np.stack([np.frombuffer(e, np.uint8) for e in encs])
    st, xyo = ctx.point_validate_batch(arr, want_xy=True)
    assert (st == np.array(want, np.uint8)).all()
    for i in range(0, 24 * 8, 8):
        assert xyo[i].tobytes() == xy(o.point_decode(S, encs[i]))
    # IETF verify with a torsion-shifted output / input / key: InvalidData from GPU and oracle alike
    n = 64
    seeds, msg = _synth(n, 900)
    sk, _ = ctx.secret_from_seed_batch(seeds)
    ref = co.ietf_prove_batch(sk, msgs=msg, ad=b"t", threads=NCPU)
    shift = lambda arr, k: np.stack([np.frombuffer(o.point_encode(S, o.te_add(S, o.point_decode(S, a.tobytes()), o.te_mul(S, k, T8))), np.uint8) for a in arr])
    for which in ("pk", "input", "output"):
        bad = {k: ref[k].copy() for k in ("pk", "input", "output")}
        bad[which][::3] = shift(ref[which][::3], 4)            # + the point of order 2
        bad[which][1::3] = shift(ref[which][1::3], 1)          # + a point of order 8
        want = co.ietf_verify_batch(bad["pk"], bad["input"], bad["output"], ref["c"], ref["s"], b"t", threads=NCPU)
        st = ctx.ietf_verify_batch(bad["pk"], bad["input"], bad["output"], ref["c"], ref["s"], ad=b"t")
        assert (st == want).all() and (want[::3] == 2).all() and (want[1::3] == 2).all() and (want[2::3] == 0).all()
    # with the subgroup test switched off (PREVALIDATED_ALL) the same items are plain verification failures or worse:
    # what the flag means -- GPU and unchecked oracle still agree
    ctx.set_prevalidated(True); co.set_check_mask(0)
    try:
        bad_out = ref["output"].copy(); bad_out[::2] = shift(ref["output"][::2], 4)
        want = co.ietf_verify_batch(ref["pk"], ref["input"], bad_out, ref["c"], ref["s"], b"t", threads=NCPU)
        st = ctx.ietf_verify_batch(ref["pk"], ref["input"], bad_out, ref["c"], ref["s"], ad=b"t")
        assert (st == want).all()
    finally:
        ctx.set_prevalidated(False); co.set_check_mask(15)


@pytest.mark.gpu
def test_gpu_primitives_msm_and_affine_paths(gpu):
    ctx, S, sid = gpu
    rnd = random.Random(9)
    G = (S.gx, S.gy)
    n = 300
    ks = [rnd.randrange(S.r) for _ in range(n)]
    pts = [o.te_mul(S, rnd.randrange(1, S.r), G) for _ in range(24)]
    P = [pts[i % 24] for i in range(n)]
    enc = np.stack([np.frombuffer(o.point_encode(S, p), np.uint8) for p in P])
    kk = np.stack([np.frombuffer(le(k), np.uint8) for k in ks])
    out, st = ctx.test_scalar_mul(kk, enc)
    assert (st == 0).all()
    for i in range(0, n, 13):
        assert out[i].tobytes() == o.point_encode(S, o.te_mul(S, ks[i], P[i]))
    a, b = enc[:24], enc[12:36]
    out, st = ctx.test_point_add(a, b)
    for i in range(24):
        assert out[i].tobytes() == o.point_encode(S, o.te_add(S, P[i], P[12 + i]))
    # field product test primitive
    xs = [rnd.getrandbits(256) for _ in range(512)]; ys = [rnd.getrandbits(256) for _ in range(512)]
    r = ctx.fq_mul_batch(np.stack([np.frombuffer(le(x), np.uint8) for x in xs]), np.stack([np.frombuffer(le(y), np.uint8) for y in ys]))
    for i in range(512):
        assert int.from_bytes(r[i].tobytes(), "little") == xs[i] * ys[i] % S.q
    # MSM against the discrete-log sum
    m = 5000
    dl = [rnd.randrange(1, S.r) for _ in range(64)]
    base = [o.te_mul(S, d, G) for d in dl]
    bxy = np.stack([np.frombuffer(xy(base[i % 64]), np.uint8) for i in range(m)])
    sc = [rnd.randrange(S.r) for _ in range(m)]
    scb = np.stack([np.frombuffer(le(k), np.uint8) for k in sc])
    encp, pxy = ctx.msm(bxy, scb)
    tot = sum(sc[i] * dl[i % 64] for i in range(m)) % S.r
    assert pxy == xy(o.te_mul(S, tot, G)) and encp == o.point_encode(S, o.te_mul(S, tot, G))
    # affine verification entry point: x || y inputs give the statuses of the compressed path
    nn = 128
    seeds, msg = _synth(nn, 5000)
    sk, _ = ctx.secret_from_seed_batch(seeds)
    ref = co.ietf_prove_batch(sk, msgs=msg, ad=b"aff", threads=NCPU)
    toxy = lambda arr: np.stack([np.frombuffer(xy(o.point_decode(S, a.tobytes())), np.uint8) for a in arr])
    s_bad = ref["s"].copy(); s_bad[::4, 1] ^= 8
    st = ctx.ietf_verify_batch_affine(toxy(ref["pk"]), toxy(ref["input"]), toxy(ref["output"]), ref["c"], s_bad, ad=b"aff")
    assert (st[::4] == 1).all() and st.sum() == len(st[::4])


@pytest.mark.gpu
def test_gpu_rfc9381_vectors_through_the_hip_path():
    """RFC 9381 ECVRF-EDWARDS25519-SHA512-TAI, Appendix B.3: the published vectors through libvrfhip -- a context whose
    descriptor states the RFC's suite (suite string 03, 16-byte little-endian challenge, RFC 8032 sign bit, cofactor in the
    output hash) and the public key prepended to alpha as the salt.  hash-to-curve, verification and the output hash
    reproduce H, "VALID" and beta; a flipped bit anywhere in pi does not verify."""
    import dataclasses
    from ark_ec_vrfs_amd import Context, Ed25519Sha512Tai, SuiteDesc
    d = dataclasses.replace(SuiteDesc.default(Ed25519Sha512Tai), suite_id=b"\x03", challenge_len=16, flags=7)
    ctx = Context(0, desc=d)
    f = lambda b: np.frombuffer(b, np.uint8).reshape(1, -1)
    try:
        for v in RFC["vectors"]:
            pk, alpha, pi = bytes.fromhex(v["pk"]), bytes.fromhex(v["alpha"]), bytes.fromhex(v["pi"])
            h = ctx.hash_to_curve_batch([pk + alpha])[0].tobytes()
            if "h" in v:
                assert h.hex() == v["h"]
            c32 = pi[32:48] + bytes(16)
            assert ctx.ietf_verify_batch(f(pk), f(h), f(pi[:32]), f(c32), f(pi[48:]), ad=b"")[0] == 0
            assert ctx.output_hash_batch(f(pi[:32]))[0].tobytes().hex() == v["beta"]
            assert ctx.point_validate_batch(f(pk))[0] == 0 and ctx.point_validate_batch(f(pi[:32]))[0] == 0
            for bit in (0, 77, 255, 256 + 5, 256 + 127, 384 + 3, 384 + 250):        # Gamma | c | s
                bad = bytearray(pi); bad[bit // 8] ^= 1 << (bit % 8)
                st = ctx.ietf_verify_batch(f(pk), f(h), f(bytes(bad[:32])), f(bytes(bad[32:48]) + bytes(16)), f(bytes(bad[48:])), ad=b"")[0]
                assert st in (1, 2)
        # the same context proves and verifies among its own items (descriptor flags are consistent end to end), and
        # equals the C oracle run with the same descriptor
        S = o.ed25519_rfc9381_params()
        co.set_suite_desc(3, b"\x03", b"", xy((S.gx, S.gy)), xy((S.bx, S.by)), challenge_len=16, flags=7)
        n = 200
        seeds, msg = _synth(n, 77)
        sk, pk = ctx.secret_from_seed_batch(seeds)
        got = ctx.ietf_prove_batch(sk, msgs=msg, ad=b"rfc")
        ref = co.ietf_prove_batch(sk, msgs=msg, ad=b"rfc", threads=NCPU)
        for k in ("output", "c", "s", "pk", "input"):
            assert (got[k] == ref[k]).all(), k
        assert (ctx.ietf_verify_batch(got["pk"], got["input"], got["output"], got["c"], got["s"], ad=b"rfc") == 0).all()
    finally:
        co.set_suite(1)
        ctx.close()


@pytest.mark.gpu
def test_gpu_soak_2_18_plus_tail(gpu):
    """n = 2^18 + 5 (several proofs per lane, ragged tail): proof bytes on a strided sample and the statuses of a tampered
    batch against the C oracle; all-valid at full size."""
    import torch
    ctx, S, sid = gpu
    n = (1 << 18) + 5
    dev = torch.device("cuda:0")
    seeds = torch.arange(n, dtype=torch.int64, device=dev).view(torch.uint8).reshape(n, 8)
    sk = torch.empty((n, 32), dtype=torch.uint8, device=dev)
    from ark_ec_vrfs_amd import _lib
    lib = _lib.load()
    st0 = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.vrfhip_secret_from_seed_batch_dev(ctx.handle, n, seeds.data_ptr(), 8, sk.data_ptr(), None, st0), "seed")
    g = torch.Generator(device=dev); g.manual_seed(11)
    msg = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    mk = lambda: torch.empty((n, 32), dtype=torch.uint8, device=dev)
    out, c, s, pk, hh = mk(), mk(), mk(), mk(), mk()
    st = torch.empty(n, dtype=torch.uint8, device=dev)
    ctx.ietf_prove_batch_dev(sk, msg, 32, out, c, s, pk, hh, st)
    torch.cuda.synchronize()
    assert int(st.sum()) == 0
    idx = np.unique(np.concatenate([np.arange(0, n, 257), np.arange(n - 16, n)]))
    ti = torch.from_numpy(idx).to(dev)
    ref = co.ietf_prove_batch(sk[ti].cpu().numpy(), msgs=msg[ti].cpu().numpy(), ad=b"", threads=NCPU)
    for name, t in (("output", out), ("c", c), ("s", s), ("pk", pk), ("input", hh)):
        assert (t[ti].cpu().numpy() == ref[name]).all(), name
    ctx.ietf_verify_batch_dev(pk, hh, out, c, s, st)
    torch.cuda.synchronize()
    assert int(st.sum()) == 0
    s2 = s.clone(); s2[::1021, 5] ^= 4
    ctx.ietf_verify_batch_dev(pk, hh, out, c, s2, st)
    torch.cuda.synchronize()
    assert torch.equal(torch.nonzero(st).flatten(), torch.arange(0, n, 1021, device=dev)) and int(st.max()) == 1


@pytest.mark.gpu
def test_gpu_every_other_entry_point_on_the_new_suites(gpu):
    """The entry points the tests above do not touch, on Ed25519 and Baby-JubJub: ragged messages with per-item `ad`,
    prove from a given input point, empty batches, keyed verification, three contexts in one `_multi` call, the provers'
    x || y output mode and the Montgomery-256 coordinate format (x 2^256 mod q: arkworks' in-memory `Fp`)."""
    from ark_ec_vrfs_amd import (Context, ietf_prove_batch_multi, ietf_verify_batch_multi, pedersen_prove_batch_multi,
                                 pedersen_verify_batch_multi)
    ctx, S, sid = gpu
    rnd = random.Random(31)
    n = 257
    sk = np.stack([np.frombuffer(co.secret_from_seed(bytes([i & 255, i >> 8, 77])), np.uint8) for i in range(n)])
    msgs = [bytes(rnd.getrandbits(8) for _ in range(rnd.choice([0, 1, 31, 32, 33, 111, 200]))) for _ in range(n)]
    ads = [bytes(rnd.getrandbits(8) for _ in range(rnd.choice([0, 3, 16, 70, 129]))) for _ in range(n)]
    got = ctx.ietf_prove_batch(sk, msgs=msgs, ad=ads)
    for i in range(0, n, 5):                                   # per-item oracle calls (ragged shapes)
        if msgs[i]:
            r = co.ietf_prove_batch(sk[i:i + 1], msgs=np.frombuffer(msgs[i], np.uint8).reshape(1, -1), ad=ads[i])
            for k in ("output", "c", "s", "pk", "input"):
                assert (got[k][i] == r[k][0]).all(), (i, k)
    empty = [i for i in range(n) if not msgs[i]][:3]
    assert empty
    for i in empty:                                            # the empty message through the Python oracle
        H = o.data_to_point(S, b"")
        g, c, s_ = o.ietf_prove(S, int.from_bytes(sk[i].tobytes(), "little"), H, ads[i])
        assert got["input"][i].tobytes() == o.point_encode(S, H) and got["output"][i].tobytes() == o.point_encode(S, g)
        assert got["c"][i].tobytes() == le(c) and got["s"][i].tobytes() == le(s_)
    st = ctx.ietf_verify_batch(got["pk"], got["input"], got["output"], got["c"], got["s"], ad=ads)
    assert not st.any()
    ads2 = list(ads); ads2[7] = ads2[7] + b"x"
    st = ctx.ietf_verify_batch(got["pk"], got["input"], got["output"], got["c"], got["s"], ad=ads2)
    assert st[7] == 1 and st.sum() == 1
    # prove from given input points == prove from the messages
    again = ctx.ietf_prove_batch(sk, inputs=got["input"], ad=ads)
    for k in ("output", "c", "s"):
        assert (again[k] == got[k]).all(), k
    # empty batches are fine everywhere
    e32 = np.zeros((0, 32), np.uint8)
    assert ctx.ietf_verify_batch(e32, e32, e32, e32, e32, ad=b"").shape == (0,)
    assert ctx.ietf_prove_batch(e32, inputs=e32, ad=b"")["c"].shape == (0, 32)
    # keyed verification
    nk = 9
    ksk = sk[:nk]
    kpk = np.stack([np.frombuffer(co.public_from_secret(ksk[i].tobytes()), np.uint8) for i in range(nk)])
    key = (np.arange(n, dtype=np.uint32) * 5) % nk
    kp = ctx.ietf_prove_batch(ksk[key], msgs=msgs, ad=b"k")
    ks, kst = ctx.keyset_create(kpk)
    try:
        assert not kst.any()
        s_bad = kp["s"].copy(); s_bad[::6, 1] ^= 2
        key2 = key.copy(); key2[1::6] = (key2[1::6] + 1) % nk
        want = co.ietf_verify_batch(kpk[key2], kp["input"], kp["output"], kp["c"], s_bad, b"k", threads=NCPU)
        stk = ctx.ietf_verify_batch_keyed(ks, key2, kp["input"], kp["output"], kp["c"], s_bad, ad=b"k")
        assert (stk == want).all() and want[::6].all() and want[1::6].all() and not want[2::6].any()
    finally:
        ks.close()
    # three contexts, one call
    extra = [Context(0, suite=ctx.suite, test_blinding_base=True), Context(0, suite=ctx.suite, test_blinding_base=True)]
    ctxs = [ctx] + extra
    try:
        many = ietf_prove_batch_multi(ctxs, sk, msgs, ad=ads)
        for k in ("output", "c", "s", "pk", "input", "status"):
            assert (many[k] == got[k]).all() if k != "status" else not many[k].any(), k
        s_bad = got["s"].copy(); s_bad[::3, 2] ^= 1
        st1 = ctx.ietf_verify_batch(got["pk"], got["input"], got["output"], got["c"], s_bad, ad=ads)
        assert (ietf_verify_batch_multi(ctxs, got["pk"], got["input"], got["output"], got["c"], s_bad, ad=ads) == st1).all()
        p1 = ctx.pedersen_prove_batch(sk, msgs=msgs, ad=b"shared")
        pm = pedersen_prove_batch_multi(ctxs, sk, msgs, ad=b"shared")
        for k in ("output", "pk_com", "r", "ok", "s", "sb", "blinding", "input"):
            assert (p1[k] == pm[k]).all(), k
        args = [p1[k] for k in ("input", "output", "pk_com", "r", "ok", "s", "sb")]
        args[6] = args[6].copy(); args[6][::4, 0] ^= 8
        v1 = ctx.pedersen_verify_batch(*args, ad=b"shared")
        assert (pedersen_verify_batch_multi(ctxs, *args, ad=b"shared") == v1).all() and v1[::4].all()
        assert (pedersen_verify_batch_multi(ctxs, *args, ad=b"shared", rlc_seed=os.urandom(32)) == v1).all()
    finally:
        for c in extra:
            c.close()
    # x || y outputs and the Montgomery-256 coordinate format
    Q, R256 = S.q, (1 << 256) % S.q

    def to_mont(a):
        out = np.zeros_like(a)
        for i, row in enumerate(a):
            x, y = int.from_bytes(row[:32].tobytes(), "little"), int.from_bytes(row[32:].tobytes(), "little")
            out[i] = np.frombuffer(le(x * R256 % Q) + le(y * R256 % Q), np.uint8)
        return out
    c2 = Context(0, suite=ctx.suite, test_blinding_base=True)
    try:
        m = 64
        ref = c2.ietf_prove_batch(sk[:m], msgs=msgs[:m], ad=b"m")
        _, xys = zip(*[c2.point_validate_batch(ref[k], want_xy=True) for k in ("pk", "input", "output")])
        for k, a in zip(("pk", "input", "output"), xys):
            for i in (0, 17, 63):
                assert a[i].tobytes() == xy(o.point_decode(S, ref[k][i].tobytes()))
        s_bad = ref["s"].copy(); s_bad[::9, 0] ^= 1
        want_v = c2.ietf_verify_batch_affine(*xys, ref["c"], s_bad, ad=b"m")
        assert (want_v[::9] == 1).all() and want_v.sum() == len(want_v[::9])
        c2.set_flags(c2.PROVE_POINTS_AFFINE)
        ga = c2.ietf_prove_batch(sk[:m], msgs=msgs[:m], ad=b"m")
        assert (ga["output"] == xys[2]).all() and (ga["pk"] == xys[0]).all() and (ga["c"] == ref["c"]).all()
        c2.set_flags(c2.COORDS_MONT256)
        assert (c2.ietf_verify_batch_affine(*[to_mont(a) for a in xys], ref["c"], s_bad, ad=b"m") == want_v).all()
        _, vxy = c2.point_validate_batch(ref["output"], want_xy=True)
        assert (vxy == to_mont(xys[2])).all()
        k = np.stack([np.frombuffer(le(7 + 13 * i), np.uint8) for i in range(m)])
        got_m = c2.msm(to_mont(xys[2]), k)
        c2.set_flags(0)
        want_m = c2.msm(xys[2], k)
        assert got_m[0] == want_m[0] and got_m[1] == to_mont(np.frombuffer(want_m[1], np.uint8).reshape(1, 64)).tobytes()
    finally:
        c2.close()
