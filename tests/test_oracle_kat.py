"""CPU: the oracles against the golden vectors and against each other."""
import random

import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import vrf_oracle as o
from conftest import hx

S = o.BANDERSNATCH


def test_constants_are_consistent():
    G, B = (S.gx, S.gy), (S.bx, S.by)
    assert o.te_is_on_curve(S, G) and o.te_is_on_curve(S, B)
    assert o.te_mul(S, S.r, G) == (0, 1) and o.te_mul(S, S.r, B) == (0, 1)
    assert (S.q - 1) % (1 << 32) == 0 and ((S.q - 1) >> 32) & 1
    assert o.legendre(5, S.q) == -1


def test_python_oracle_ietf_kat(kat):
    assert o.point_encode(S, (S.gx, S.gy)).hex() == kat["enc_G"]
    for v in kat["ietf"]:
        ad, alpha = bytes.fromhex(v["ad"]), bytes.fromhex(v["alpha"])
        sk = o.secret_from_seed(S, bytes.fromhex(v["seed"]))
        assert o.scalar_encode(sk).hex() == v["sk"]
        pk = o.public_from_secret(S, sk)
        assert o.point_encode(S, pk).hex() == v["pk"]
        H = o.data_to_point(S, alpha)
        assert o.point_encode(S, H).hex() == v["h"]
        g, c, s = o.ietf_prove(S, sk, H, ad)
        assert o.point_encode(S, g).hex() == v["gamma"]
        assert o.output_hash(S, g).hex() == v["beta"]
        assert o.scalar_encode(c).hex() == v["c"] and o.scalar_encode(s).hex() == v["s"]
        assert o.ietf_verify(S, pk, H, g, ad, c, s)
        assert not o.ietf_verify(S, pk, H, g, ad, c, s ^ 1)
        assert not o.ietf_verify(S, pk, H, g, ad + b"x", c, s)


def test_python_oracle_pedersen_kat(kat):
    for v in kat["pedersen"]:
        ad = bytes.fromhex(v["ad"])
        sk = o.secret_from_seed(S, bytes.fromhex(v["seed"]))
        H = o.data_to_point(S, bytes.fromhex(v["alpha"]))
        g, (pc, R, Ok, s, sb), b = o.pedersen_prove(S, sk, H, ad)
        assert o.scalar_encode(b).hex() == v["blinding"]
        assert o.point_encode(S, pc).hex() == v["pk_com"] and o.point_encode(S, R).hex() == v["r"]
        assert o.point_encode(S, Ok).hex() == v["ok"]
        assert o.scalar_encode(s).hex() == v["s"] and o.scalar_encode(sb).hex() == v["sb"]
        assert o.pedersen_verify(S, H, g, ad, (pc, R, Ok, s, sb))
        assert not o.pedersen_verify(S, H, g, ad, (pc, R, Ok, s, sb ^ 1))


def test_python_oracle_stage_kat(kat):
    for v in kat["stages"]:
        a = bytes.fromhex(v["alpha"])
        assert o.xmd_sha512_96(a, S.h2c_dst).hex() == v["xmd96"]
        u0, u1 = o.hash_to_field2(S, a)
        assert o.scalar_encode(u0).hex() == v["u0"] and o.scalar_encode(u1).hex() == v["u1"]
        q0, q1 = o.elligator2_te(S, u0), o.elligator2_te(S, u1)
        assert o.point_encode(S, q0).hex() == v["q0"] and o.point_encode(S, q1).hex() == v["q1"]
        assert o.point_encode(S, o.te_add(S, q0, q1)).hex() == v["q0q1"]


def test_stage_kat_nonce_and_challenge(kat):
    for v, st in zip(kat["ietf"][:2], kat["stages"]):
        sk = int.from_bytes(bytes.fromhex(v["sk"]), "little")
        H = o.point_decode(S, bytes.fromhex(v["h"]))
        k = o.nonce_rfc8032(S, sk, H)
        assert o.scalar_encode(k).hex() == st["k"]
        assert o.point_encode(S, o.te_mul(S, k, (S.gx, S.gy))).hex() == st["kG"]
        assert o.point_encode(S, o.te_mul(S, k, H)).hex() == st["kH"]
        pre = (S.suite_id + b"\x02" + bytes.fromhex(v["pk"]) + bytes.fromhex(v["h"]) + bytes.fromhex(v["gamma"])
               + bytes.fromhex(st["kG"]) + bytes.fromhex(st["kH"]) + b"\x00")
        assert len(pre) == 187 and o.sha512(pre).hex() == st["challenge_sha512"]


def test_c_oracle_kat(kat):
    for v in kat["ietf"]:
        ad = bytes.fromhex(v["ad"])
        assert co.secret_from_seed(bytes.fromhex(v["seed"])).hex() == v["sk"]
        assert co.public_from_secret(bytes.fromhex(v["sk"])).hex() == v["pk"]
        assert co.hash_to_curve(bytes.fromhex(v["alpha"])).hex() == v["h"]
        assert co.output_hash(bytes.fromhex(v["gamma"])).hex() == v["beta"]
        if len(v["alpha"]) > 0:
            r = co.ietf_prove_batch(hx(v["sk"]), msgs=hx(v["alpha"]).reshape(1, -1), ad=ad)
        else:
            r = co.ietf_prove_batch(hx(v["sk"]), inputs=hx(v["h"]), ad=ad)
        assert r["output"][0].tobytes().hex() == v["gamma"]
        assert r["c"][0].tobytes().hex() == v["c"] and r["s"][0].tobytes().hex() == v["s"]
        assert r["pk"][0].tobytes().hex() == v["pk"]
        assert co.ietf_verify_batch(hx(v["pk"]), hx(v["h"]), hx(v["gamma"]), hx(v["c"]), hx(v["s"]), ad)[0] == 0
        bad = bytearray(bytes.fromhex(v["s"])); bad[3] ^= 0x10
        assert co.ietf_verify_batch(hx(v["pk"]), hx(v["h"]), hx(v["gamma"]), hx(v["c"]),
                                    np.frombuffer(bytes(bad), np.uint8), ad)[0] == 1


def test_c_oracle_matches_python_oracle_on_random_items(synth):
    sk, msg = synth(6, start=1000)
    r = co.ietf_prove_batch(sk, msgs=msg, ad=b"ad-bytes", threads=2)
    for i in range(6):
        skv = int.from_bytes(sk[i].tobytes(), "little")
        H = o.data_to_point(S, msg[i].tobytes())
        g, c, s = o.ietf_prove(S, skv, H, b"ad-bytes")
        assert r["input"][i].tobytes() == o.point_encode(S, H)
        assert r["output"][i].tobytes() == o.point_encode(S, g)
        assert r["c"][i].tobytes() == o.scalar_encode(c) and r["s"][i].tobytes() == o.scalar_encode(s)
        assert r["pk"][i].tobytes() == o.point_encode(S, o.public_from_secret(S, skv))


def test_c_oracle_primitives_vs_python():
    rnd = random.Random(7)
    for _ in range(200):
        a, b = rnd.getrandbits(256), rnd.getrandbits(256)
        assert int.from_bytes(co.fq_mul(a.to_bytes(32, "little"), b.to_bytes(32, "little")), "little") == a * b % S.q
    for n in (0, 1, 111, 112, 113, 127, 128, 129, 255, 256, 1000):
        m = bytes(rnd.getrandbits(8) for _ in range(n))
        assert co.sha512(m) == o.sha512(m)


def test_decode_edge_cases_agree():
    rnd = random.Random(11)
    cases = [bytes(32), (1).to_bytes(32, "little"), (S.q - 1).to_bytes(32, "little"), S.q.to_bytes(32, "little"),
             (S.q + 1).to_bytes(32, "little"), b"\xff" * 32, ((1 << 255) | 1).to_bytes(32, "little")]
    cases += [rnd.getrandbits(256).to_bytes(32, "little") for _ in range(60)]
    n_ok = 0
    for enc in cases:
        p = o.point_decode(S, enc)
        c = co.point_decode(enc, subgroup=False)
        assert (p is None) == (c is None), enc.hex()
        if p is not None:
            n_ok += 1
            assert p == c and o.te_is_on_curve(S, p)
            c2 = co.point_decode(enc, subgroup=True)
            assert (c2 is not None) == o.te_in_prime_subgroup(S, p)
    assert n_ok > 10


def test_small_order_points_fail_subgroup_check():
    # (0, -1) has order 2; it decodes but is not in the prime-order subgroup
    enc = (S.q - 1).to_bytes(32, "little")
    assert o.point_decode(S, enc) == (0, S.q - 1)
    assert co.point_decode(enc, subgroup=False) is not None
    assert co.point_decode(enc, subgroup=True) is None


def test_c_oracle_pedersen_kat_and_python(kat, synth):
    v, iv = kat["pedersen"][0], kat["ietf"][0]
    r = co.pedersen_prove_batch(hx(iv["sk"]), inputs=hx(iv["h"]), ad=bytes.fromhex(v["ad"]))
    got = {k: r[k][0].tobytes().hex() for k in ("pk_com", "r", "ok", "s", "sb", "blinding")}
    assert got == {k: v[k] for k in got} and r["output"][0].tobytes().hex() == iv["gamma"]
    assert co.pedersen_verify_batch(hx(iv["h"]), r["output"], r["pk_com"], r["r"], r["ok"], r["s"], r["sb"])[0] == 0
    sk, msg = synth(4, start=40)
    r = co.pedersen_prove_batch(sk, msgs=msg, ad=b"pedersen ad", threads=2)
    for i in range(4):
        skv = int.from_bytes(sk[i].tobytes(), "little")
        H = o.data_to_point(S, msg[i].tobytes())
        g, (pc, R, Ok, s, sb), b = o.pedersen_prove(S, skv, H, b"pedersen ad")
        exp = [o.point_encode(S, g), o.point_encode(S, pc), o.point_encode(S, R), o.point_encode(S, Ok),
               o.scalar_encode(s), o.scalar_encode(sb), o.scalar_encode(b)]
        assert [r[k][i].tobytes() for k in ("output", "pk_com", "r", "ok", "s", "sb", "blinding")] == exp
    st = co.pedersen_verify_batch(r["input"], r["output"], r["pk_com"], r["r"], r["ok"], r["s"], r["sb"], b"pedersen ad")
    assert (st == 0).all()
    bad = r["sb"].copy(); bad[1, 0] ^= 1
    st = co.pedersen_verify_batch(r["input"], r["output"], r["pk_com"], r["r"], r["ok"], r["s"], bad, b"pedersen ad")
    assert list(st) == [0, 1, 0, 0]


def test_c_oracle_msm_against_python(synth):
    G = (S.gx, S.gy)
    pts = [o.te_mul(S, 3 + 5 * i, G) for i in range(5)]
    ks = [7, 0, S.r - 1, 123456789, 2 ** 200 + 17]
    acc = (0, 1)
    for P, k in zip(pts, ks):
        acc = o.te_add(S, acc, o.te_mul(S, k, P))
    xy = np.frombuffer(b"".join(P[0].to_bytes(32, "little") + P[1].to_bytes(32, "little") for P in pts), np.uint8)
    kk = np.frombuffer(b"".join(k.to_bytes(32, "little") for k in ks), np.uint8)
    enc, out_xy = co.msm(xy, kk)
    assert enc == o.point_encode(S, acc)
    assert out_xy == acc[0].to_bytes(32, "little") + acc[1].to_bytes(32, "little")
    bad = xy.copy(); bad[0] ^= 1
    assert co.msm(bad, kk) is None
