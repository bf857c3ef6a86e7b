"""secp256r1 ("P256_SHA256_TAI", RFC 9381 suite 0x01): SURVEY.md section 8 row f4, the short-Weierstrass suite.

CPU tier: RFC 9381 Appendix B.1 pins the Python oracle (oracle/sw_oracle.py); the device headers compiled for the host
(tests/hostsim/hostsim_p256.hip: P-256 field, complete projective law, SHA-256 / HMAC, Sec1 codec, try-and-increment,
RFC 6979 nonce, the per-item prove / verify steps) against Python big ints, hashlib, the oracle and the RFC vectors.
GPU tier (-m gpu): the HIP path through the C ABI: RFC vectors, prove bytes and verify statuses against the oracle,
tampering, undecodable points, batches that straddle launch groups."""
import ctypes
import hashlib
import hmac
import json
import os
import random
import subprocess

import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import sw_oracle as sw

HERE = os.path.dirname(os.path.abspath(__file__))
HS = os.path.join(HERE, "hostsim")
RFC = json.load(open(os.path.join(HERE, "golden", "rfc9381_p256_sha256_tai.json")))
P, N, G = sw.P, sw.N, sw.G


def be(x):
    return int(x).to_bytes(32, "big")


def xy(pt):
    return bytes(64) if pt is None else be(pt[0]) + be(pt[1])


def unxy(raw):
    return None if raw == bytes(64) else (int.from_bytes(raw[:32], "big"), int.from_bytes(raw[32:], "big"))


@pytest.fixture(scope="module")
def hp():
    so = os.path.join(HS, "libhostsim_p256.so")
    subprocess.run(["make", "-C", HS, "-j4", os.path.basename(so)], check=True, stdout=subprocess.DEVNULL)
    return ctypes.CDLL(so)


def _fe(f, *a):
    r = ctypes.create_string_buffer(32)
    ret = f(*[be(x) for x in a], r)
    return int.from_bytes(r.raw, "big"), ret


def _pt(f, *a):
    r = ctypes.create_string_buffer(64)
    f(*a, r)
    return unxy(r.raw)


# ---------------------------------------------------------------------------------------------- oracle pinned by the RFC
def test_rfc9381_b1_vectors_pin_the_python_oracle():
    for v in RFC["vectors"]:
        out = sw.rfc9381_prove(bytes.fromhex(v["sk"]), bytes.fromhex(v["alpha"]))
        for k in ("pk", "h", "k", "u", "v", "pi", "beta"):
            assert out[k].hex() == v[k], k
        assert out["ctr"] == v["ctr"]
        pi = bytes.fromhex(v["pi"])
        Y, H, Gm = (sw.point_decode(b) for b in (out["pk"], out["h"], pi[:33]))
        c, s = int.from_bytes(pi[33:49], "big"), int.from_bytes(pi[49:], "big")
        assert sw.ietf_verify(Y, H, Gm, b"", c, s)
        assert not sw.ietf_verify(Y, H, Gm, b"", c, s ^ 1) and not sw.ietf_verify(Y, H, Gm, b"x", c, s)
        assert not sw.ietf_verify(Y, H, sw.add(Gm, G), b"", c, s)


def test_curve_constants():
    assert sw.is_on_curve(G) and sw.mul(N, G) is None and sw.mul(N - 1, G) == sw.neg(G)
    assert abs(N - (P + 1)) <= 2 * int(P ** 0.5) + 2 and P % 4 == 3


def test_c_oracle_equals_the_python_oracle_and_the_rfc_vectors():
    """oracle/c/oracle_p256.c (the checker of the big GPU samples and the bench's CPU baseline) against RFC 9381 B.1 and
    against oracle/sw_oracle.py on random items, tampered proofs and undecodable points."""
    for v in RFC["vectors"]:
        sk, pk, alpha = bytes.fromhex(v["sk"]), bytes.fromhex(v["pk"]), bytes.fromhex(v["alpha"])
        assert co.p256_public_from_secret(sk).hex() == v["pk"] and co.p256_hash_to_curve(pk + alpha).hex() == v["h"]
        r = co.p256_ietf_prove_batch(np.frombuffer(sk, np.uint8), msgs=np.frombuffer(pk + alpha, np.uint8).reshape(1, -1))
        assert (r["output"][0].tobytes() + r["c"][0].tobytes()[16:] + r["s"][0].tobytes()).hex() == v["pi"]
        assert co.p256_output_hash(r["output"][0].tobytes()).hex() == v["beta"]
        assert co.p256_ietf_verify_batch(r["pk"], r["input"], r["output"], r["c"], r["s"])[0] == 0
    rng = np.random.default_rng(3)
    n = 12
    sk = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    msg = rng.integers(0, 256, (n, 19), dtype=np.uint8)
    r = co.p256_ietf_prove_batch(sk, msgs=msg, ad=b"ad", threads=4)
    for i in range(n):
        k = sw.scalar_decode(sk[i].tobytes())
        H, _ = sw.hash_to_curve_tai(msg[i].tobytes())
        g, c, s = sw.ietf_prove(k, H, b"ad")
        assert r["input"][i].tobytes() == sw.point_encode(H) and r["output"][i].tobytes() == sw.point_encode(g)
        assert r["c"][i].tobytes() == be(c) and r["s"][i].tobytes() == be(s) and r["pk"][i].tobytes() == sw.point_encode(sw.mul(k, G))
        assert co.p256_secret_from_seed(msg[i].tobytes()) == be(sw.secret_from_seed(msg[i].tobytes()))
    s2 = r["s"].copy(); s2[1, 5] ^= 1
    g2 = r["output"].copy(); g2[2, 0] = 7
    st = co.p256_ietf_verify_batch(r["pk"], r["input"], g2, r["c"], s2, ad=b"ad", threads=2)
    assert list(st[:4]) == [0, 1, 2, 0]
    # the group law's corner: s G = c Y (U at infinity) gets the oracle's verdict, whatever it is
    k0 = sw.scalar_decode(sk[0].tobytes())
    want = 0 if sw.ietf_verify(sw.mul(k0, G), sw.point_decode(r["input"][0].tobytes()), sw.point_decode(r["output"][0].tobytes()), b"ad", 1, k0) else 1
    assert co.p256_ietf_verify_batch(r["pk"][:1], r["input"][:1], r["output"][:1], np.frombuffer(be(1), np.uint8), np.frombuffer(be(k0), np.uint8), ad=b"ad")[0] == want


# ---------------------------------------------------------------------------------------------- device headers on the host
def test_field_ops_against_python_ints(hp):
    rnd = random.Random(1)
    edge = [0, 1, 2, 3, P - 1, P - 2, P, P + 1, (1 << 256) - 1, 1 << 255, (1 << 29) - 1, 1 << 232, 1 << 224, (1 << 224) - 1,
            1 << 192, 1 << 96, (1 << 96) - 1, P - (1 << 96), P - (1 << 224)]
    for it in range(3000):
        x = rnd.choice(edge) if it % 7 == 0 else rnd.getrandbits(256)
        y = rnd.choice(edge) if it % 11 == 0 else rnd.getrandbits(256)
        assert _fe(hp.hp_fe_mul, x, y)[0] == x * y % P
        assert _fe(hp.hp_fe_sqr, x)[0] == x * x % P
        assert _fe(hp.hp_fe_sub, x, y)[0] == (x - y) % P
    for it in range(400):
        x = rnd.choice(edge) if it % 5 == 0 else rnd.getrandbits(256)
        assert _fe(hp.hp_fe_inv, x)[0] == pow(x % P, P - 2, P)
        r, sq = _fe(hp.hp_fe_sqrt, x)
        want = pow(x % P, (P - 1) // 2, P) in (0, 1)
        assert bool(sq) == want
        assert r * r % P == (x % P if want else (-x) % P)
        assert hp.hp_fe_jacobi(be(x % P)) == {0: 0, 1: 1, P - 1: -1}[pow(x % P, (P - 1) // 2, P)]


def test_weak_reduction_over_its_whole_input_range(hp):
    """fe_wred takes any lazily accumulated value < 32 p back below 2 p: k * x for k = 1..15 (x < 2 p: up to 30 p) and x near 0, p and 2^256."""
    rnd = random.Random(2)
    xs = [0, 1, P - 1, P, P + 1, (1 << 256) - 1, (1 << 256) - (1 << 224), (1 << 224) + 5] + [rnd.getrandbits(256) for _ in range(40)]
    for x in xs:
        xm = x % P                                  # the value that enters is the FeN image of x (< 2 p)
        for k in list(range(1, 16)):
            r = ctypes.create_string_buffer(32)
            hp.hp_fe_wred_chain(be(x), k, r)
            assert int.from_bytes(r.raw, "big") == k * xm % P, (hex(x), k)


def test_scalar_ops(hp):
    rnd = random.Random(3)
    edge = [0, 1, N - 1, N, N + 1, (1 << 256) - 1, 1 << 255, N >> 1, (N >> 1) + 1]
    for it in range(600):
        a, b, c = (rnd.choice(edge) if (it + j) % 5 == 0 else rnd.getrandbits(256) for j in range(3))
        r = ctypes.create_string_buffer(32)
        hp.hp_fr_mul_add(be(a), be(b), be(c), r)
        assert int.from_bytes(r.raw, "big") == (a * b + c) % N


def test_complete_group_law(hp):
    rnd = random.Random(4)
    pts = [sw.mul(rnd.randrange(1, N), G) for _ in range(10)] + [None, G, sw.neg(G)]
    for A in pts:
        for B in (pts[0], A, sw.neg(A), None, sw.add(A, A), G):
            assert _pt(hp.hp_add, xy(A), xy(B)) == sw.add(A, B)
            za, zb = rnd.randrange(1, P), rnd.randrange(1, P)
            assert _pt(hp.hp_add_scaled, xy(A), xy(B), be(za), be(zb)) == sw.add(A, B)
        assert _pt(hp.hp_dbl, xy(A)) == sw.add(A, A)
        for z in (1, rnd.randrange(1, P), P - 1):       # the Jacobian doubling run between windows; infinity stays infinity
            assert _pt(hp.hp_dbl4, xy(A), be(z)) == sw.mul(16, A)
    special = [0, 1, 2, N - 1, N - 2, (1 << 255), (1 << 256) - 1 - (1 << 32), int("88" * 32, 16), int("77" * 32, 16),
               int("80" * 32, 16), int("7f" * 32, 16), int("7f80" * 16, 16), int("ff" * 31 + "00", 16) % N, 128, 127, 255, 256]
    for it in range(len(special) + 15):
        k = special[it] if it < len(special) else rnd.getrandbits(256)
        A = pts[it % 10]
        # scalars enter the ladders reduced mod n; k >= n is reduced by the caller (p256_scalar_decode)
        kk = k % N
        assert _pt(hp.hp_mul, be(kk), xy(A)) == sw.mul(kk, A)
        assert _pt(hp.hp_mul_base, be(kk)) == sw.mul(kk, G)
        assert _pt(hp.hp_mul_quad, be(kk), xy(A)) == sw.mul(kk, A)
    # the folded recoding's corner: |k'| just below n/2 with a carry into the top nibble (digit 63 becomes +8)
    for kk in (N >> 1, (N >> 1) + 1, (N >> 1) - 1, int("7" + "f" * 63, 16), int("78" + "8" * 62, 16), int("77" + "8" * 62, 16), N - int("78" + "8" * 62, 16)):
        assert _pt(hp.hp_mul_quad, be(kk % N), xy(pts[3])) == sw.mul(kk % N, pts[3]), hex(kk)


def test_sha256_and_hmac(hp):
    rnd = random.Random(5)
    for n in list(range(0, 140)) + [255, 256, 1000]:
        m = bytes(rnd.getrandbits(8) for _ in range(n))
        d = ctypes.create_string_buffer(32)
        hp.hp_sha256(m, n, d)
        assert d.raw == hashlib.sha256(m).digest(), n
        key = bytes(rnd.getrandbits(8) for _ in range(32))
        hp.hp_hmac256(key, m, n, d)
        assert d.raw == hmac.new(key, m, "sha256").digest(), n


def test_codec_and_suite_functions(hp):
    rnd = random.Random(6)
    # decode: both tags, x >= p, x off the curve, bad tags
    for _ in range(60):
        raw = bytes([rnd.choice([2, 3, 2, 3, 0, 4, 5])]) + rnd.getrandbits(256).to_bytes(32, "big")
        out = ctypes.create_string_buffer(64)
        ok = hp.hp_decode(raw, out)
        try:
            want = sw.point_decode(raw)
        except ValueError:
            want = None
        assert bool(ok) == (want is not None)
        if want is not None:
            assert unxy(out.raw) == want
    assert not hp.hp_decode(b"\x02" + be(P), ctypes.create_string_buffer(64))
    assert not hp.hp_decode(b"\x02" + be(P + 5), ctypes.create_string_buffer(64))      # x = 5 as an integer is valid, p + 5 is not
    for i in range(40):
        data = bytes(rnd.getrandbits(8) for _ in range(rnd.choice([0, 1, 31, 32, 33, 55, 56, 64, 100, 200])))
        enc = ctypes.create_string_buffer(33)
        assert hp.hp_hash_to_curve(data, len(data), enc)
        assert enc.raw == sw.point_encode(sw.hash_to_curve_tai(data)[0])
        sk = rnd.randrange(1, N)
        k = ctypes.create_string_buffer(32)
        hp.hp_nonce(be(sk), enc.raw, k)
        assert int.from_bytes(k.raw, "big") == sw.nonce_rfc6979(sk, sw.point_decode(enc.raw))
        beta = ctypes.create_string_buffer(32)
        hp.hp_output_hash(enc.raw, beta)
        assert beta.raw == sw.output_hash(sw.point_decode(enc.raw))
        s = ctypes.create_string_buffer(32)
        hp.hp_secret_from_seed(data, len(data), s)
        assert int.from_bytes(s.raw, "big") == sw.secret_from_seed(data)


def test_rfc9381_b1_vectors_through_the_host_build(hp):
    for v in RFC["vectors"]:
        sk, alpha, pk = bytes.fromhex(v["sk"]), bytes.fromhex(v["alpha"]), bytes.fromhex(v["pk"])
        out = ctypes.create_string_buffer(261)
        assert hp.hp_prove(sk, pk + alpha, len(pk + alpha), None, b"", 0, out) == 1
        o = out.raw
        assert o[:33].hex() == v["pk"] and o[33:66].hex() == v["h"] and o[163:195].hex() == v["k"]
        assert o[195:228].hex() == v["u"] and o[228:261].hex() == v["v"]
        pi = o[66:99] + o[99 + 16:131] + o[131:163]
        assert o[99:99 + 16] == bytes(16) and pi.hex() == v["pi"]
        beta = ctypes.create_string_buffer(32)
        hp.hp_output_hash(o[66:99], beta)
        assert beta.raw.hex() == v["beta"]
        assert hp.hp_verify(o[:33], o[33:66], o[66:99], o[99:131], o[131:163], b"", 0) == 0
        # given H instead of the message
        out2 = ctypes.create_string_buffer(261)
        assert hp.hp_prove(sk, None, 0, o[33:66], b"", 0, out2) == 1 and out2.raw == o


@pytest.mark.parametrize("clen", [20, 32, 1])
def test_host_build_with_another_suite_string_and_challenge_length(hp, clen):
    sid = b"custom-p256-suite"
    hp.hp_set_suite(sid, len(sid), clen)
    try:
        for i in range(3):
            sk = sw.secret_from_seed(b"q%d" % i)
            msg = b"abc" * (i + 1)
            H, _ = sw.hash_to_curve_tai(msg, sid)
            g, c, s = sw.ietf_prove(sk, H, b"x", sid, clen)
            out = ctypes.create_string_buffer(261)
            assert hp.hp_prove(be(sk), msg, len(msg), None, b"x", 1, out) == 1
            o = out.raw
            assert o[33:66] == sw.point_encode(H) and o[66:99] == sw.point_encode(g) and o[99:131] == be(c) and o[131:163] == be(s)
            assert (c >> 128) or clen < 17                      # a 20-byte challenge really is longer than 16
            assert hp.hp_verify(o[:33], o[33:66], o[66:99], o[99:131], o[131:163], b"x", 1) == 0
            assert hp.hp_verify(o[:33], o[33:66], o[66:99], be(c ^ (1 << (8 * clen - 1))), o[131:163], b"x", 1) == 1
            if clen < 32:
                assert hp.hp_verify(o[:33], o[33:66], o[66:99], be(c | (1 << (8 * clen + 3))), o[131:163], b"x", 1) == 1
            beta = ctypes.create_string_buffer(32)
            hp.hp_output_hash(o[66:99], beta)
            assert beta.raw == sw.output_hash(g, sid)
    finally:
        hp.hp_set_suite(b"\x01", 1, 16)


def test_host_build_prove_and_verify_equal_the_oracle(hp):
    rnd = random.Random(7)
    for i in range(6):
        sk = sw.secret_from_seed(b"seed%d" % i)
        msg, ad = b"message %d" % i * (i + 1), b"ad" * i
        H, _ = sw.hash_to_curve_tai(msg)
        gamma, c, s = sw.ietf_prove(sk, H, ad)
        out = ctypes.create_string_buffer(261)
        assert hp.hp_prove(be(sk), msg, len(msg), None, ad, len(ad), out) == 1
        o = out.raw
        assert o[:33] == sw.point_encode(sw.mul(sk, G)) and o[33:66] == sw.point_encode(H) and o[66:99] == sw.point_encode(gamma)
        assert o[99:131] == be(c) and o[131:163] == be(s)
        pk, h, g = o[:33], o[33:66], o[66:99]
        assert hp.hp_verify(pk, h, g, be(c), be(s), ad, len(ad)) == 0
        assert hp.hp_verify(pk, h, g, be(c), be(s), ad + b"x", len(ad) + 1) == 1
        assert hp.hp_verify(pk, h, g, be(c ^ 1), be(s), ad, len(ad)) == 1
        assert hp.hp_verify(pk, h, g, be(c), be((s + 1) % N), ad, len(ad)) == 1
        if s + N < (1 << 256):
            assert hp.hp_verify(pk, h, g, be(c), be(s + N), ad, len(ad)) == 2                                 # s >= n: RFC 9381 5.4.4
        if c + N < (1 << 256):
            assert hp.hp_verify(pk, h, g, be(c + N), be(s), ad, len(ad)) == 1                                # a c field above 16 bytes is no proof string
        assert hp.hp_verify(pk, h, g, be(c + (1 << 128)), be(s), ad, len(ad)) == 1                           # c is 16 bytes
        other = sw.point_encode(sw.mul(rnd.randrange(1, N), G))
        assert hp.hp_verify(other, h, g, be(c), be(s), ad, len(ad)) == 1
        assert hp.hp_verify(pk, other, g, be(c), be(s), ad, len(ad)) == 1
        assert hp.hp_verify(pk, h, other, be(c), be(s), ad, len(ad)) == 1
        bad = b"\x02" + be(next(x for x in range(2, 50) if not _on_curve_x(x)))
        assert hp.hp_verify(bad, h, g, be(c), be(s), ad, len(ad)) == 2
        assert hp.hp_verify(pk, h, b"\x04" + g[1:], be(c), be(s), ad, len(ad)) == 2
        # crafted corner of the group law: s G = c Y makes U the point at infinity; the verdict is still the oracle's
        assert hp.hp_verify(pk, h, g, be(1), be(sk), ad, len(ad)) == (0 if sw.ietf_verify(sw.mul(sk, G), H, gamma, ad, 1, sk) else 1)
        assert hp.hp_verify(pk, pk, pk, be(c), be(s), ad, len(ad)) == (0 if sw.ietf_verify(sw.mul(sk, G), sw.mul(sk, G), sw.mul(sk, G), ad, c, s) else 1)


def test_pedersen_host_build_and_c_oracle_equal_the_python_oracle(hp):
    """The Pedersen scheme on this suite (unpinned: no vector, upstream's blinding base unknown) -- three restatements of
    the same algorithm held against each other: Python, C, and the device headers on the host; then tampering."""
    B = sw.default_blinding_base()
    assert sw.is_on_curve(B) and sw.mul(N, B) is None
    hp.hp_set_blinding_base(xy(B))
    co.p256_set_blinding_base(B)
    rnd = random.Random(12)
    for i in range(4):
        sk = sw.secret_from_seed(b"ped%d" % i)
        msg, ad = b"pm%d" % i * (i + 2), b"a" * i
        H, _ = sw.hash_to_curve_tai(msg)
        g, (pc, R, Ok, s, sb), b = sw.pedersen_prove(sk, H, ad, B)
        assert sw.pedersen_verify(H, g, (pc, R, Ok, s, sb), ad, B) and not sw.pedersen_verify(H, g, (pc, R, Ok, s, sb ^ 1), ad, B)
        out = ctypes.create_string_buffer(261)
        assert hp.hp_ped_prove(be(sk), msg, len(msg), ad, len(ad), out) == 1
        o = out.raw
        enc = sw.point_encode
        assert o == enc(g) + enc(pc) + enc(R) + enc(Ok) + be(s) + be(sb) + be(b) + enc(H)
        rc = co.p256_pedersen_prove_batch(np.frombuffer(be(sk), np.uint8), msgs=np.frombuffer(msg, np.uint8).reshape(1, -1), ad=ad)
        assert b"".join(rc[k][0].tobytes() for k in ("output", "pk_com", "r", "ok", "s", "sb", "blinding", "input")) == o
        h33, g33, pc33, r33, ok33 = o[228:261], o[:33], o[33:66], o[66:99], o[99:132]
        args = (h33, g33, pc33, r33, ok33, o[132:164], o[164:196])
        assert hp.hp_ped_verify(*args, ad, len(ad)) == 0
        assert hp.hp_ped_verify(*args, ad + b"x", len(ad) + 1) == 1
        for j in range(7):                                    # every field tampered with: a wrong point, a wrong scalar
            bad = list(args)
            bad[j] = enc(sw.mul(rnd.randrange(1, N), G)) if j < 5 else be((int.from_bytes(args[j], "big") + 1) % N)
            want = co.p256_pedersen_verify_batch(*[np.frombuffer(x, np.uint8) for x in bad], ad=ad)[0]
            assert want == 1 and hp.hp_ped_verify(*bad, ad, len(ad)) == 1, j
        assert hp.hp_ped_verify(h33, g33, b"\x04" + pc33[1:], r33, ok33, o[132:164], o[164:196], ad, len(ad)) == 2
        assert hp.hp_ped_verify(h33, g33, pc33, r33, ok33, be(N), o[164:196], ad, len(ad)) == 2            # s >= n


def _on_curve_x(x):
    y2 = (x ** 3 - 3 * x + sw.B) % P
    return pow(y2, (P - 1) // 2, P) in (0, 1)


# ---------------------------------------------------------------------------------------------- GPU: the HIP path through the C ABI
@pytest.fixture(scope="module")
def gpu():
    from ark_ec_vrfs_amd import Context, Secp256r1Sha256Tai
    ctx = Context(0, Secp256r1Sha256Tai, test_blinding_base=True)
    yield ctx
    ctx.close()


def _u8(rows):
    return np.stack([np.frombuffer(bytes(r), np.uint8) for r in rows])


@pytest.mark.gpu
def test_gpu_rfc9381_b1_vectors(gpu):
    """RFC 9381 Appendix B.1 through the kernels: PK from SK, H, pi = (Gamma, c, s), beta; the proofs verify."""
    assert gpu.point_bytes() == 33 and gpu.hash_bytes() == 32
    V = RFC["vectors"]
    sk = _u8(bytes.fromhex(v["sk"]) for v in V)
    msgs = [bytes.fromhex(v["pk"]) + bytes.fromhex(v["alpha"]) for v in V]           # salt || alpha, the RFC's own use
    r = gpu.ietf_prove_batch(sk, msgs=msgs, ad=b"")
    assert (r["status"] == 0).all()
    for i, v in enumerate(V):
        assert r["pk"][i].tobytes().hex() == v["pk"] and r["input"][i].tobytes().hex() == v["h"]
        pi = r["output"][i].tobytes() + r["c"][i].tobytes()[16:] + r["s"][i].tobytes()
        assert r["c"][i].tobytes()[:16] == bytes(16) and pi.hex() == v["pi"]
    assert [b.tobytes().hex() for b in gpu.output_hash_batch(r["output"])] == [v["beta"] for v in V]
    assert [h.tobytes().hex() for h in gpu.hash_to_curve_batch(msgs)] == [v["h"] for v in V]
    st = gpu.ietf_verify_batch(r["pk"], r["input"], r["output"], r["c"], r["s"], ad=b"")
    assert (st == 0).all()
    r2 = gpu.ietf_prove_batch(sk, inputs=r["input"], ad=b"")                        # the pre-hashed input instead of the message
    assert all((r2[k] == r[k]).all() for k in ("output", "c", "s", "pk", "input"))


@pytest.mark.gpu
def test_gpu_prove_and_verify_equal_the_oracle(gpu):
    n = 24
    seeds = _u8(b"seed-%04d" % i + bytes(23) for i in range(n))
    sk, pk = gpu.secret_from_seed_batch(seeds)
    msgs = [b"msg %d " % i * (1 + i % 5) for i in range(n)]
    ads = [b"ad" * (i % 4) for i in range(n)]
    r = gpu.ietf_prove_batch(sk, msgs=msgs, ad=ads)
    assert (r["status"] == 0).all() and (r["pk"] == pk).all()
    for i in range(n):
        k = sw.secret_from_seed(seeds[i].tobytes())
        assert sk[i].tobytes() == be(k) and pk[i].tobytes() == sw.point_encode(sw.mul(k, G))
        H, _ = sw.hash_to_curve_tai(msgs[i])
        gamma, c, s = sw.ietf_prove(k, H, ads[i])
        assert r["input"][i].tobytes() == sw.point_encode(H) and r["output"][i].tobytes() == sw.point_encode(gamma)
        assert r["c"][i].tobytes() == be(c) and r["s"][i].tobytes() == be(s)
    assert (gpu.ietf_verify_batch(pk, r["input"], r["output"], r["c"], r["s"], ad=ads) == 0).all()
    # tampering: every field of every third item, statuses against the oracle's verdicts
    pkt, ht, gt, ct, st_ = (x.copy() for x in (pk, r["input"], r["output"], r["c"], r["s"]))
    other = sw.point_encode(sw.mul(12345, G))
    offx = next(x for x in range(2, 50) if not _on_curve_x(x))
    want = np.zeros(n, np.uint8)
    for i in range(n):
        kind = i % 8
        if kind == 1: st_[i, 31] ^= 1
        elif kind == 2: ct[i, 31] ^= 1
        elif kind == 3: gt[i] = np.frombuffer(other, np.uint8)
        elif kind == 4: pkt[i] = np.frombuffer(b"\x02" + be(offx), np.uint8)            # not on the curve
        elif kind == 5: ht[i, 0] = 4                                                      # bad tag
        elif kind == 6: ct[i, 3] = 1                                                      # c >= 2^128
        elif kind == 7: gt[i] = np.frombuffer(b"\x03" + be(P + 1), np.uint8)             # x >= p
        want[i] = [0, 1, 1, 1, 2, 2, 1, 2][kind]
    got = gpu.ietf_verify_batch(pkt, ht, gt, ct, st_, ad=ads)
    assert (got == want).all(), (got, want)
    # a non-canonical s is InvalidData (RFC 9381 5.4.4; upstream deserialises s strictly); a c field holding more than the
    # suite's 16 challenge bytes (here c + n: the same scalar) is no proof string and fails
    s0, c0 = int.from_bytes(r["s"][0].tobytes(), "big"), int.from_bytes(r["c"][0].tobytes(), "big")
    if s0 + N < (1 << 256):
        s2 = r["s"].copy(); s2[0] = np.frombuffer(be(s0 + N), np.uint8)
        assert gpu.ietf_verify_batch(pk, r["input"], r["output"], r["c"], s2, ad=ads)[0] == 2
    c2 = r["c"].copy(); c2[0] = np.frombuffer(be(c0 + N), np.uint8)
    assert gpu.ietf_verify_batch(pk, r["input"], r["output"], c2, r["s"], ad=ads)[0] == 1
    # corners of the group law built from valid bytes: c = 1, s = sk makes U = s G - c Y the point at infinity (hashed as
    # the single byte 0x00, as Sec1Codec encodes it); pk = H = Gamma makes both ladders add a point to itself
    k0 = sw.secret_from_seed(seeds[0].tobytes())
    one = np.frombuffer(be(1), np.uint8).reshape(1, 32)
    got = gpu.ietf_verify_batch(pk[:1], r["input"][:1], r["output"][:1], one, np.frombuffer(be(k0), np.uint8).reshape(1, 32), ad=ads[:1])
    H0, G0 = sw.point_decode(r["input"][0].tobytes()), sw.point_decode(r["output"][0].tobytes())
    assert got[0] == (0 if sw.ietf_verify(sw.mul(k0, G), H0, G0, ads[0], 1, k0) else 1)
    got = gpu.ietf_verify_batch(pk[:1], pk[:1], pk[:1], r["c"][:1], r["s"][:1], ad=ads[:1])
    c0, s0 = int.from_bytes(r["c"][0].tobytes(), "big"), int.from_bytes(r["s"][0].tobytes(), "big")
    Y0 = sw.mul(k0, G)
    assert got[0] == (0 if sw.ietf_verify(Y0, Y0, Y0, ads[0], c0, s0) else 1)
    # point validation
    pts = _u8([sw.point_encode(sw.mul(7, G)), b"\x02" + be(offx), b"\x05" + be(G[0]), b"\x03" + be(P), b"\x03" + be(G[0])])
    stv, xyv = gpu.point_validate_batch(pts, want_xy=True)
    assert list(stv) == [0, 2, 2, 2, 0]
    seven = sw.mul(7, G)
    assert xyv[0].tobytes() == int(seven[0]).to_bytes(32, "little") + int(seven[1]).to_bytes(32, "little")
    assert xyv[4].tobytes() == int(G[0]).to_bytes(32, "little") + int((P - G[1]) if G[1] % 2 == 0 else G[1]).to_bytes(32, "little")


@pytest.mark.gpu
def test_gpu_batch_round_trip_across_launch_groups(gpu):
    """2^15 + 7 items with a 2^13 workspace: prove, verify (all ok), one tampered byte per item of a slice (all rejected),
    the same proofs through two contexts at once (_multi) and against the single-context answers."""
    from ark_ec_vrfs_amd import Context, Secp256r1Sha256Tai, ietf_prove_batch_multi, ietf_verify_batch_multi
    n = (1 << 15) + 7
    gpu.reserve(1 << 13)
    rng = np.random.default_rng(9)
    sk = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    msg = rng.integers(0, 256, (n, 24), dtype=np.uint8)
    r = gpu.ietf_prove_batch(sk, msgs=msg, ad=b"batch")
    assert (r["status"] == 0).all()
    st = gpu.ietf_verify_batch(r["pk"], r["input"], r["output"], r["c"], r["s"], ad=b"batch")
    assert (st == 0).all()
    s2 = r["s"].copy(); s2[::3, 17] ^= 0x40
    st = gpu.ietf_verify_batch(r["pk"], r["input"], r["output"], r["c"], s2, ad=b"batch")
    assert (st[::3] == 1).all() and (np.delete(st, np.s_[::3]) == 0).all()
    assert (gpu.ietf_verify_batch(r["pk"], r["input"], r["output"], r["c"], r["s"], ad=b"other") == 1).all()
    # a sample against the oracle
    for i in (0, 1, (1 << 13) - 1, 1 << 13, n - 1):
        k = sw.scalar_decode(sk[i].tobytes())
        H, _ = sw.hash_to_curve_tai(msg[i].tobytes())
        gamma, c, s = sw.ietf_prove(k, H, b"batch")
        assert r["output"][i].tobytes() == sw.point_encode(gamma) and r["c"][i].tobytes() == be(c) and r["s"][i].tobytes() == be(s)
    other = Context(0, Secp256r1Sha256Tai, test_blinding_base=True)
    try:
        m = 4099
        rm = ietf_prove_batch_multi([gpu, other], sk[:m], [x.tobytes() for x in msg[:m]], ad=b"batch")
        assert all((rm[k] == r[k][:m]).all() for k in ("output", "c", "s", "pk", "input"))
        stm = ietf_verify_batch_multi([gpu, other], rm["pk"], rm["input"], rm["output"], rm["c"], s2[:m], ad=b"batch")
        assert (stm == st[:m]).all() and (stm[::3] == 1).all() and (np.delete(stm, np.s_[::3]) == 0).all()
        # the Pedersen scheme through two contexts: proof bytes equal the single context's, and a mixed batch (tampered s,
        # tampered sb, an undecodable R, another item's Ok) gets the single context's statuses item for item
        from ark_ec_vrfs_amd import pedersen_prove_batch_multi, pedersen_verify_batch_multi
        msgs_m = [x.tobytes() for x in msg[:m]]
        p1 = gpu.pedersen_prove_batch(sk[:m], msgs=msg[:m], ad=b"batch")
        pm = pedersen_prove_batch_multi([gpu, other], sk[:m], msgs_m, ad=b"batch")
        names = ("output", "pk_com", "r", "ok", "s", "sb", "input")
        assert all((pm[k] == p1[k]).all() for k in names)
        bad = {k: pm[k].copy() for k in names}
        bad["s"][::5, 9] ^= 0x10
        bad["sb"][1::5, 30] ^= 0x01
        bad["r"][2::7, 0] = 0x05                      # not a Sec1 tag: InvalidData
        bad["ok"][3::11] = np.roll(pm["ok"], 1, axis=0)[3::11]
        args = [bad[k] for k in ("input", "output", "pk_com", "r", "ok", "s", "sb")]
        st1 = gpu.pedersen_verify_batch(*args, ad=b"batch")
        stp = pedersen_verify_batch_multi([gpu, other], *args, ad=b"batch")
        assert (stp == st1).all() and (st1[::5] != 0).all() and (st1[2::7] == 2).all()
        clean = np.ones(m, bool); clean[::5] = clean[1::5] = clean[2::7] = clean[3::11] = False
        assert (st1[clean] == 0).all() and clean.sum() > m // 3
    finally:
        other.close()
        gpu.reserve(1 << 20)


@pytest.mark.gpu
def test_gpu_whole_batch_equals_the_c_oracle(gpu):
    """Every byte of 6000 proofs (ragged: two launch groups of a 4096 workspace) and every status of a batch with each
    kind of defect, against oracle/c/oracle_p256.c."""
    n = 6000
    gpu.reserve(4096)
    try:
        rng = np.random.default_rng(21)
        sk = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        sk[0] = 0; sk[0, 31] = 1                       # sk = 1
        sk[1] = np.frombuffer(be(N - 1), np.uint8)     # sk = n - 1
        sk[2] = np.frombuffer(be(N + 5), np.uint8)     # not a canonical scalar: InvalidData
        sk[3] = np.frombuffer(be(N), np.uint8)
        msg = rng.integers(0, 256, (n, 40), dtype=np.uint8)
        r = gpu.ietf_prove_batch(sk, msgs=msg, ad=b"soak")
        ref = co.p256_ietf_prove_batch(sk, msgs=msg, ad=b"soak", threads=8)
        good = np.ones(n, bool); good[2:4] = False
        for k in ("output", "c", "s", "pk", "input"):
            assert (r[k][good] == ref[k][good]).all(), k
            assert not r[k][~good].any()                      # a failed item's outputs are all-zero
        assert (r["status"] == ref["status"]).all() and list(r["status"][:5]) == [0, 0, 2, 2, 0]
        for k in ("output", "c", "s", "pk", "input"):         # the verifier below gets decodable bytes for those two
            r[k][~good] = r[k][4]
        pkt, ht, gt, ct, st_ = (x.copy() for x in (r["pk"], r["input"], r["output"], r["c"], r["s"]))
        kind = rng.integers(0, 9, n)
        for i in range(n):
            if kind[i] == 1: st_[i, rng.integers(0, 32)] ^= 1 << rng.integers(0, 8)
            elif kind[i] == 2: ct[i, rng.integers(16, 32)] ^= 1 << rng.integers(0, 8)
            elif kind[i] == 3: gt[i, rng.integers(1, 33)] ^= 1 << rng.integers(0, 8)      # a different x: another point or none
            elif kind[i] == 4: pkt[i, 0] ^= 1                                             # the other root: -Y
            elif kind[i] == 5: ht[i, 0] = rng.integers(4, 256)
            elif kind[i] == 6: ct[i, rng.integers(0, 16)] |= 1 << rng.integers(0, 8)      # c >= 2^128
            elif kind[i] == 7: gt[i, 1:] = 0xff                                           # x >= p
            elif kind[i] == 8: pkt[i] = r["pk"][(i + 1) % n]
        got = gpu.ietf_verify_batch(pkt, ht, gt, ct, st_, ad=b"soak")
        want = co.p256_ietf_verify_batch(pkt, ht, gt, ct, st_, ad=b"soak", threads=8)
        assert (got == want).all(), np.nonzero(got != want)[0][:10]
        assert set(np.unique(want)) == {0, 1, 2}
        assert (gpu.hash_to_curve_batch(msg[:512]) == ref["input"][:512])[good[:512]].all()
        assert [h.tobytes() for h in gpu.output_hash_batch(r["output"][:64])] == [co.p256_output_hash(x.tobytes()) for x in r["output"][:64]]
        stv = gpu.point_validate_batch(gt)
        assert list(stv) == [co.p256_point_decode(x.tobytes()) for x in gt]
    finally:
        gpu.reserve(1 << 20)


@pytest.mark.gpu
def test_gpu_descriptor_supplies_suite_string_challenge_length_and_generator():
    """The suite's data comes from the descriptor: a 17-byte suite string (not a multiple of the hash's word), a 20-byte
    challenge and another generator, against the oracle given the same data; a generator off the curve is refused."""
    from ark_ec_vrfs_amd import Context, SuiteDesc, CURVE_SECP256R1, VrfHipError
    sid, clen, gen = b"custom-p256-suite", 20, sw.mul(7, G)
    le64 = lambda pt: int(pt[0]).to_bytes(32, "little") + int(pt[1]).to_bytes(32, "little")
    ctx = Context(0, desc=SuiteDesc(CURVE_SECP256R1, sid, b"", le64(gen), bytes(64), challenge_len=clen))
    try:
        d = ctx.desc()
        assert d.suite_id == sid and d.challenge_len == clen and d.curve == CURVE_SECP256R1
        n = 6
        sks = [sw.secret_from_seed(b"k%d" % i) for i in range(n)]
        msgs = [b"m%d" % i * (i + 1) for i in range(n)]
        r = ctx.ietf_prove_batch(_u8(be(k) for k in sks), msgs=msgs, ad=b"dd")
        assert (r["status"] == 0).all()
        for i in range(n):
            H, _ = sw.hash_to_curve_tai(msgs[i], sid)
            g, c, s = sw.ietf_prove(sks[i], H, b"dd", sid, clen, gen)
            assert r["input"][i].tobytes() == sw.point_encode(H) and r["output"][i].tobytes() == sw.point_encode(g)
            assert r["pk"][i].tobytes() == sw.point_encode(sw.mul(sks[i], gen))
            assert r["c"][i].tobytes() == be(c) and r["s"][i].tobytes() == be(s) and c >= (1 << 128) or r["c"][i].tobytes() == be(c)
            assert ctx.output_hash_batch(r["output"][i:i + 1])[0].tobytes() == sw.output_hash(g, sid)
        assert (ctx.ietf_verify_batch(r["pk"], r["input"], r["output"], r["c"], r["s"], ad=b"dd") == 0).all()
        assert (ctx.ietf_verify_batch(r["pk"], r["input"], r["output"], r["c"], r["s"], ad=b"de") == 1).all()
        seeds = _u8(b"seed%04d" % i for i in range(4))
        sk2, pk2 = ctx.secret_from_seed_batch(seeds)
        for i in range(4):
            k = sw.secret_from_seed(seeds[i].tobytes())
            assert sk2[i].tobytes() == be(k) and pk2[i].tobytes() == sw.point_encode(sw.mul(k, gen))
    finally:
        ctx.close()
    with pytest.raises(VrfHipError):
        Context(0, desc=SuiteDesc(CURVE_SECP256R1, sid, b"", le64((gen[0], gen[1] ^ 1)), bytes(64), challenge_len=clen))
    with pytest.raises(VrfHipError):
        Context(0, desc=SuiteDesc(CURVE_SECP256R1, sid, b"", le64(gen), bytes(64), challenge_len=clen, flags=1))


@pytest.mark.gpu
def test_gpu_pedersen_equals_the_c_oracle(gpu):
    """Pedersen prove bytes and verify statuses on the GPU against oracle/c/oracle_p256.c (itself held against the Python
    oracle on CPU), with the built-in blinding base; mixed defects; a context whose descriptor has no base refuses."""
    from ark_ec_vrfs_amd import Context, SuiteDesc, CURVE_SECP256R1, VrfHipError, pedersen_prove_batch_multi, pedersen_verify_batch_multi
    B = sw.default_blinding_base()
    co.p256_set_blinding_base(B)
    d = gpu.desc()
    assert d.blinding_base == int(B[0]).to_bytes(32, "little") + int(B[1]).to_bytes(32, "little")
    n = 3000
    gpu.reserve(2048)
    try:
        rng = np.random.default_rng(33)
        sk = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        sk[5] = np.frombuffer(be(N + 1), np.uint8)                      # not canonical: InvalidData
        msg = rng.integers(0, 256, (n, 28), dtype=np.uint8)
        r = gpu.pedersen_prove_batch(sk, msgs=msg, ad=b"ped")
        ref = co.p256_pedersen_prove_batch(sk, msgs=msg, ad=b"ped", threads=8)
        names = ("output", "pk_com", "r", "ok", "s", "sb", "blinding", "input")
        good = np.ones(n, bool); good[5] = False
        assert (r["status"] == ref["status"]).all() and r["status"][5] == 2 and r["status"].sum() == 2
        for k in names:
            assert (r[k][good] == ref[k][good]).all(), k
            assert not r[k][5].any()
            r[k][5] = r[k][6]
        pts = [r[k].copy() for k in ("input", "output", "pk_com", "r", "ok")]
        sc = [r[k].copy() for k in ("s", "sb")]
        kind = rng.integers(0, 10, n)
        for i in range(n):
            j = kind[i]
            if j < 5: pts[j][i, rng.integers(1, 33)] ^= 1 << rng.integers(0, 8)          # another x: another point or none
            elif j < 7: sc[j - 5][i, rng.integers(0, 32)] ^= 1 << rng.integers(0, 8)
            elif j == 7: pts[rng.integers(0, 5)][i, 0] = 9                                # bad tag
        got = gpu.pedersen_verify_batch(*pts, *sc, ad=b"ped")
        want = co.p256_pedersen_verify_batch(*pts, *sc, ad=b"ped", threads=8)
        assert (got == want).all(), np.nonzero(got != want)[0][:10]
        assert set(np.unique(want)) == {0, 1, 2} and (want[kind >= 8] == 0).all()
        assert (gpu.pedersen_verify_batch(*[r[k] for k in ("input", "output", "pk_com", "r", "ok", "s", "sb")], ad=b"pee") == 1).all()
        other = Context(0, gpu.suite, test_blinding_base=True)
        try:
            m = 1001
            rm = pedersen_prove_batch_multi([gpu, other], sk[6:6 + m], [x.tobytes() for x in msg[6:6 + m]], ad=b"ped")
            assert all((rm[k] == ref[k][6:6 + m]).all() for k in names)
            stm = pedersen_verify_batch_multi([gpu, other], *[p_[6:6 + m] for p_ in pts], *[s_[6:6 + m] for s_ in sc], ad=b"ped")
            assert (stm == want[6:6 + m]).all()
        finally:
            other.close()
    finally:
        gpu.reserve(1 << 20)
    nob = Context(0, desc=SuiteDesc(CURVE_SECP256R1, b"\x01", b"", d.generator, bytes(64), challenge_len=16))
    try:
        with pytest.raises(VrfHipError):
            nob.pedersen_prove_batch(np.zeros((1, 32), np.uint8), msgs=[b"a"])
        assert nob.ietf_prove_batch(np.frombuffer(be(7), np.uint8), msgs=[b"a"])["status"][0] == 0
    finally:
        nob.close()


@pytest.mark.gpu
def test_gpu_xy_forms_for_typed_callers(gpu):
    """What a Rust caller holding arkworks values uses: the provers hand out points as x || y (VRFHIP_FLAG_PROVE_POINTS_AFFINE),
    the verifier takes pk / input / output as x || y (no square root on either side), canonical little-endian integers or
    arkworks' in-memory Montgomery limbs (VRFHIP_FLAG_COORDS_MONT256): same proofs, same verdicts as the Sec1 forms."""
    le = lambda v: int(v).to_bytes(32, "little")
    xyle = lambda pt: le(pt[0]) + le(pt[1])
    xym = lambda pt: le(pt[0] * (1 << 256) % P) + le(pt[1] * (1 << 256) % P)
    n = 9
    sks = [sw.secret_from_seed(b"xy%d" % i) for i in range(n)]
    msgs = [b"xy message %d" % i for i in range(n)]
    sk = _u8(be(k) for k in sks)
    ref = gpu.ietf_prove_batch(sk, msgs=msgs, ad=b"xy")
    pts = {k: [sw.point_decode(ref[k][i].tobytes()) for i in range(n)] for k in ("pk", "input", "output")}
    B = sw.default_blinding_base()
    try:
        for flags, enc in ((gpu.PROVE_POINTS_AFFINE, xyle), (gpu.PROVE_POINTS_AFFINE | gpu.COORDS_MONT256, xym)):
            gpu.set_flags(flags)
            assert gpu.prove_point_bytes() == 64
            r = gpu.ietf_prove_batch(sk, msgs=msgs, ad=b"xy")
            assert r["output"].shape == (n, 64) and r["pk"].shape == (n, 64) and r["input"].shape == (n, 33)
            assert (r["c"] == ref["c"]).all() and (r["s"] == ref["s"]).all() and (r["input"] == ref["input"]).all()
            for i in range(n):
                assert r["output"][i].tobytes() == enc(pts["output"][i]) and r["pk"][i].tobytes() == enc(pts["pk"][i])
            h_xy = _u8(enc(p_) for p_ in pts["input"])
            st = gpu.ietf_verify_batch_affine(r["pk"], h_xy, r["output"], r["c"], r["s"], ad=b"xy")
            assert (st == 0).all()
            bad_pk, bad_g, bad_s = r["pk"].copy(), r["output"].copy(), r["s"].copy()
            bad_pk[1, 5] ^= 1                                   # off the curve
            bad_g[2] = r["pk"][2]                               # another point of the curve
            bad_pk[3, :32] = np.frombuffer(le(P + 1) if flags == gpu.PROVE_POINTS_AFFINE else le(P), np.uint8)   # coordinate >= p
            bad_s[4, 31] ^= 1
            st = gpu.ietf_verify_batch_affine(bad_pk, h_xy, bad_g, r["c"], bad_s, ad=b"xy")
            assert list(st) == [0, 2, 1, 2, 1, 0, 0, 0, 0]
            stv, xyv = gpu.point_validate_batch(ref["output"], want_xy=True)
            assert (stv == 0).all() and all(xyv[i].tobytes() == enc(pts["output"][i]) for i in range(n))
            pr = gpu.pedersen_prove_batch(sk, msgs=msgs, ad=b"xy")
            for i in range(n):
                g, (pc, R, Ok, s_, sb_), b_ = sw.pedersen_prove(sks[i], pts["input"][i], b"xy", B)
                assert pr["output"][i].tobytes() == enc(g) and pr["pk_com"][i].tobytes() == enc(pc)
                assert pr["r"][i].tobytes() == enc(R) and pr["ok"][i].tobytes() == enc(Ok)
                assert pr["s"][i].tobytes() == be(s_) and pr["sb"][i].tobytes() == be(sb_) and pr["blinding"][i].tobytes() == be(b_)
    finally:
        gpu.set_flags(0)
    assert gpu.prove_point_bytes() == 33


@pytest.mark.gpu
def test_gpu_prevalidated_flags_are_accepted_without_effect(gpu):
    """Cofactor 1: there is no subgroup test to skip.  (Nothing is refused on this suite any more: round 4 built the MSM, the
    batched verifier in both forms and key sets -- tests below.)"""
    z32, z33 = np.zeros((2, 32), np.uint8), np.zeros((2, 33), np.uint8)
    gpu.set_prevalidated(True)
    assert gpu.ietf_verify_batch(z33, z33, z33, z32, z32)[0] == 2
    gpu.set_prevalidated(False)


@pytest.mark.gpu
def test_gpu_batched_pedersen_verifier_from_xy(gpu):
    """The batched verifier from typed values: the five points of every proof as x || y (canonical, and arkworks' in-memory
    Montgomery limbs): same statuses and verdicts as the Sec1 form; a failed batch falls back to the per-proof kernels through
    the compressed strings rebuilt from the coordinates."""
    B = sw.default_blinding_base()
    co.p256_set_blinding_base(B)
    n = 700
    rng = np.random.default_rng(45)
    sk = rng.integers(0, 256, (n, 32), dtype=np.uint8); sk[:, 0] &= 0x7f
    msg = rng.integers(0, 256, (n, 20), dtype=np.uint8)
    pr = gpu.pedersen_prove_batch(sk, msgs=msg, ad=b"xy")
    F = ("input", "output", "pk_com", "r", "ok")
    seed = bytes(range(32))
    for flags in (0, gpu.COORDS_MONT256):
        gpu.set_flags(flags)
        xy5 = []
        for k in F:
            stv, xyv = gpu.point_validate_batch(pr[k], want_xy=True)           # follows COORDS_MONT256
            assert not stv.any()
            xy5.append(xyv)
        st, ok = gpu.pedersen_verify_batch_rlc(*xy5, pr["s"], pr["sb"], ad=b"xy", seed=seed, affine=True)
        assert ok and not st.any()
        bad = [a.copy() for a in xy5]
        sb2 = pr["sb"].copy()
        sb2[9, 31] ^= 1                                                        # wrong response
        bad[1][20] = xy5[1][21]                                                # another proof's output
        bad[3][30, 40] ^= 1                                                    # R off the curve (or a coordinate >= p)
        st, ok = gpu.pedersen_verify_batch_rlc(*bad, pr["s"], sb2, ad=b"xy", seed=seed, affine=True)
        assert not ok and st[9] == 1 and st[20] == 1 and st[30] == 2 and st.sum() == 4
    gpu.set_flags(0)


def test_oracle_msm_against_discrete_logs():
    """oracle_p256.c p256_msm (one double-and-add per term) against sum k_i a_i * G from the discrete logs, with P and -P,
    a repeated point, the point at infinity, zero and extreme scalars, and the invalid inputs."""
    rnd = random.Random(70)
    le = lambda v: int(v).to_bytes(32, "little")
    a = [rnd.randrange(1, N) for _ in range(10)]
    pts = [sw.mul(x, sw.G) for x in a]
    rows = [le(p_[0]) + le(p_[1]) for p_ in pts]
    neg0 = sw.neg(pts[0])
    bases = rows + [rows[1], le(neg0[0]) + le(neg0[1]), bytes(64)]
    logs = a + [a[1], N - a[0], 0]
    ks = [rnd.randrange(N) for _ in range(10)] + [N - 1, 0, 12345]
    ks[2], ks[3] = 0, 1
    st, o33, oxy = co.p256_msm(_u8(bases), _u8(be(k) for k in ks))
    want = sw.mul(sum(x * y for x, y in zip(logs, ks)) % N, sw.G)
    assert st == 0 and o33 == sw.point_encode(want) and oxy == le(want[0]) + le(want[1])
    st, o33, oxy = co.p256_msm(_u8([rows[0], le(neg0[0]) + le(neg0[1])]), _u8([be(7), be(7)]))
    assert st == 0 and o33 == bytes(33) and oxy == bytes(64)                      # the point at infinity
    bad = bytearray(rows[4]); bad[1] ^= 2
    assert co.p256_msm(_u8([rows[0], bytes(bad)]), _u8([be(1), be(1)]))[0] == 2    # off the curve
    assert co.p256_msm(_u8([rows[0]]), _u8([be(N)]))[0] == 2                       # scalar not below n
    assert co.p256_msm(_u8([le(P) + le(1)]), _u8([be(1)]))[0] == 2                 # coordinate not below p


@pytest.mark.gpu
def test_gpu_msm_equals_the_oracle(gpu):
    """`VariableBaseMSM::msm` on secp256r1 (k_p256_msm.hip: Pippenger, 512 LDS buckets per window, the complete addition
    law) against oracle_p256.c and against discrete-log sums: sizes that cross the group boundaries, many equal points (one
    bucket meeting P, P again and -P), infinity, extreme scalars, the Montgomery-limb coordinate format, invalid inputs."""
    from ark_ec_vrfs_amd import InvalidData
    rnd = random.Random(71)
    le = lambda v: int(v).to_bytes(32, "little")
    a = [rnd.randrange(1, N) for _ in range(40)]
    pool = [sw.mul(x, sw.G) for x in a]
    enc = [le(p_[0]) + le(p_[1]) for p_ in pool]
    for n in (0, 1, 2, 63, 700, 5000, 40000):
        idx = [rnd.randrange(40) for _ in range(n)]
        ks = [rnd.randrange(N) for _ in range(n)]
        for j in range(0, n, 97):
            ks[j] = rnd.choice([0, 1, N - 1, (N - 1) // 2, (N + 1) // 2, 1 << 255, (1 << 128) - 1])
        bases = _u8(enc[i] for i in idx) if n else np.zeros((0, 64), np.uint8)
        sc = _u8(be(k) for k in ks) if n else np.zeros((0, 32), np.uint8)
        want = sw.mul(sum(a[i] * k for i, k in zip(idx, ks)) % N, sw.G) if n else None
        pt, xy = gpu.msm(bases, sc)
        if n == 0:
            assert pt == bytes(33) and xy == bytes(64)
        else:
            assert pt == sw.point_encode(want) and xy == le(want[0]) + le(want[1]), n
        if 0 < n <= 5000:
            st, o33, oxy = co.p256_msm(bases, sc)
            assert st == 0 and o33 == pt and oxy == xy
    # one bucket meets P twice and -P once; the point at infinity among the bases; everything cancels
    P0, nP0 = enc[0], le(sw.neg(pool[0])[0]) + le(sw.neg(pool[0])[1])
    pt, xy = gpu.msm(_u8([P0, P0, nP0, bytes(64)]), _u8([be(9), be(9), be(18), be(5)]))
    assert pt == bytes(33) and xy == bytes(64)
    pt, _ = gpu.msm(_u8([P0, P0, nP0]), _u8([be(9), be(9), be(9)]))
    assert pt == sw.point_encode(sw.mul(9 * a[0] % N, sw.G))
    # arkworks' in-memory coordinates
    gpu.set_flags(gpu.COORDS_MONT256)
    try:
        xym = lambda p_: le(p_[0] * (1 << 256) % P) + le(p_[1] * (1 << 256) % P)
        pt, xy = gpu.msm(_u8(xym(pool[i]) for i in range(5)), _u8(be(k + 3) for k in range(5)))
        want = sw.mul(sum(a[i] * (i + 3) for i in range(5)) % N, sw.G)
        assert pt == sw.point_encode(want) and xy == xym(want)
    finally:
        gpu.set_flags(0)
    bad = bytearray(enc[4]); bad[1] ^= 2
    for bases, sc in ((_u8([enc[0], bytes(bad)]), _u8([be(1), be(1)])), (_u8([enc[0]]), _u8([be(N)])),
                      (_u8([le(P) + le(1)]), _u8([be(1)]))):
        with pytest.raises(InvalidData):
            gpu.msm(bases, sc)


@pytest.mark.gpu
def test_gpu_batched_pedersen_verifier(gpu):
    """`pedersen::Verifier::verify` for a whole batch as ONE MSM over 5 n + 2 points (random linear combination, weights from
    a digest of every input byte).  Statuses always equal the per-proof verifier's (oracle_p256.c); the single-MSM verdict
    accepts a valid batch, rejects every kind of single defect, and is not fooled by two defects that cancel without
    weights (Ok_0 += D, Ok_1 -= D); undecodable proofs are left out of the sum and flagged."""
    import torch
    B = sw.default_blinding_base()
    co.p256_set_blinding_base(B)
    n = 2500
    rng = np.random.default_rng(44)
    sk = rng.integers(0, 256, (n, 32), dtype=np.uint8); sk[:, 0] &= 0x7f
    msg = rng.integers(0, 256, (n, 20), dtype=np.uint8)
    pr = gpu.pedersen_prove_batch(sk, msgs=msg, ad=b"rlc")
    assert not pr["status"].any()
    F = ("input", "output", "pk_com", "r", "ok", "s", "sb")
    args = lambda d: [d[k] for k in F]
    seed = bytes(range(32))
    st, ok = gpu.pedersen_verify_batch_rlc(*args(pr), ad=b"rlc", seed=seed)
    assert ok and not st.any()
    st, ok = gpu.pedersen_verify_batch_rlc(*args(pr), ad=b"other", seed=seed)            # every challenge changes
    assert not ok and (st == 1).all()
    for field, row, col in (("s", 7, 31), ("sb", 8, 0), ("output", 9, 5), ("pk_com", 10, 20), ("r", 11, 32), ("ok", 12, 1), ("input", 13, 9)):
        b = {k: v.copy() for k, v in pr.items()}
        if field in ("s", "sb"):
            b[field][row, col] ^= 4
        else:
            b[field][row] = pr[field][row + 1]                                             # another proof's (valid) point
        st, ok = gpu.pedersen_verify_batch_rlc(*args(b), ad=b"rlc", seed=seed)
        want = co.p256_pedersen_verify_batch(*args(b), b"rlc", threads=8)
        assert not ok and (st == want).all() and st[row] == 1 and st.sum() == 1, field
    # two defects that cancel in an unweighted sum
    D = sw.mul(123456789, sw.G)
    b = {k: v.copy() for k, v in pr.items()}
    ok0, ok1 = sw.point_decode(pr["ok"][0].tobytes()), sw.point_decode(pr["ok"][1].tobytes())
    b["ok"][0] = np.frombuffer(sw.point_encode(sw.add(ok0, D)), np.uint8)
    b["ok"][1] = np.frombuffer(sw.point_encode(sw.add(ok1, sw.neg(D))), np.uint8)
    st, ok = gpu.pedersen_verify_batch_rlc(*args(b), ad=b"rlc", seed=seed)
    assert not ok and st[0] == 1 and st[1] == 1 and st.sum() == 2
    # undecodable proofs: InvalidData, left out; the rest still passes as one MSM
    b = {k: v.copy() for k, v in pr.items()}
    b["s"][3] = np.frombuffer(be(N), np.uint8)
    b["r"][4, 0] = 5
    b["output"][5, 1:] = np.frombuffer(be(P), np.uint8)
    st, ok = gpu.pedersen_verify_batch_rlc(*args(b), ad=b"rlc", seed=seed)
    want = co.p256_pedersen_verify_batch(*args(b), b"rlc", threads=8)
    assert ok and (st == want).all() and list(np.flatnonzero(st)) == [3, 4, 5] and (st[3:6] == 2).all()
    # the device entry point: verdict flag only, per-item ad
    dev = torch.device("cuda:0")
    t = {k: torch.from_numpy(pr[k].copy()).to(dev) for k in F}
    status = torch.full((n,), 9, dtype=torch.uint8, device=dev)
    flag = torch.full((1,), 9, dtype=torch.uint8, device=dev)
    ad = torch.from_numpy(np.frombuffer(b"rlc", np.uint8).copy()).to(dev)
    gpu.pedersen_verify_batch_rlc_dev(*[t[k] for k in F], status, flag, seed, ad=ad, ad_len=3)
    torch.cuda.synchronize()
    assert int(flag[0]) == 0 and int(status.sum()) == 0
    t["sb"][77, 3] ^= 1
    gpu.pedersen_verify_batch_rlc_dev(*[t[k] for k in F], status, flag, seed, ad=ad, ad_len=3)
    torch.cuda.synchronize()
    assert int(flag[0]) == 1 and int(status.sum()) == 0           # the batch stage names no culprit: that is the fallback's job


@pytest.mark.gpu
def test_gpu_keyed_verification(gpu):
    """`ietf::Verifier::verify` against a resident key set (vrfhip_keyset_create on this suite: every key decoded once, the 17
    comb rows a 16-byte challenge can reach kept in HBM; U = s G - c Y from two combs).  Statuses equal the plain verifier's
    and the C oracle's on a batch with every kind of defect; an undecodable key and a key index out of range are InvalidData."""
    rng = np.random.default_rng(55)
    nk, n = 37, 5000
    ksk = rng.integers(0, 256, (nk, 32), dtype=np.uint8); ksk[:, 0] &= 0x7f
    kp = gpu.ietf_prove_batch(ksk, msgs=[b"k"] * nk, ad=b"")
    keys = kp["pk"].copy()
    keys[5] = 0; keys[5, 0] = 7                                       # not a Sec1 string
    ks, kst = gpu.keyset_create(keys)
    try:
        assert list(np.flatnonzero(kst)) == [5] and kst[5] == 2 and ks.bytes() > nk * 17 * 128 * 112
        idx = rng.integers(0, nk, n).astype(np.uint32)
        idx[idx == 5] = 6
        msg = rng.integers(0, 256, (n, 17), dtype=np.uint8)
        pr = gpu.ietf_prove_batch(ksk[idx], msgs=msg, ad=b"keyed")
        assert not pr["status"].any() and (pr["pk"] == keys[idx]).all()
        inp, out, c, s = (pr[k].copy() for k in ("input", "output", "c", "s"))
        s[::7, 30] ^= 1
        c[3::31, 31] ^= 1
        c[4::37, 2] ^= 1                                              # a c above 16 bytes
        out[11::67] = pr["output"][12::67][: len(out[11::67])]
        inp[13::71, 0] = 9                                             # undecodable H
        s[5::61] = np.frombuffer(be(N), np.uint8)
        idx2 = idx.copy(); idx2[17::97] = (idx2[17::97] + 1) % nk       # another signer's key
        idx2[idx2 == 5] = 4
        want = co.p256_ietf_verify_batch(keys[idx2], inp, out, c, s, b"keyed", threads=8)
        plain = gpu.ietf_verify_batch(keys[idx2], inp, out, c, s, ad=b"keyed")
        got = gpu.ietf_verify_batch_keyed(ks, idx2, inp, out, c, s, ad=b"keyed")
        assert (plain == want).all() and (got == want).all() and set(np.unique(want)) == {0, 1, 2}
        idx3 = idx[:64].copy(); idx3[1] = 5; idx3[2] = nk; idx3[3] = 0xffffffff
        got = gpu.ietf_verify_batch_keyed(ks, idx3, pr["input"][:64], pr["output"][:64], pr["c"][:64], pr["s"][:64], ad=b"keyed")
        assert list(got[1:4]) == [2, 2, 2] and not got[4:].any() and got[0] == 0
    finally:
        ks.close()
