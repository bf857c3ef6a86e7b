"""CPU oracle (TEST INFRASTRUCTURE ONLY) for the secp256r1 suite of ark-ec-vrfs / ark-vrf (`suites::secp256r1`,
/root/reference src/lib.rs:14; upstream name "P256_SHA256_TAI", RFC 9381 suite_string 0x01).

This file is the checker, never the product: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it.

Parity status: like oracle/vrf_oracle.py the reference checkout holds no code or vector for this suite (it re-exports the
un-vendored crate ``ark-vrf``).  What pins this restatement: RFC 9381 Appendix B.1 (examples 10-12), which upstream's own
tests run for this suite -- tests/golden/rfc9381_p256_sha256_tai.json, authenticated by
tools/authenticate_rfc9381_p256_vectors.py -- reproduced field by field in tests/test_secp256r1.py.

Held the upstream way where upstream and the RFC differ (neither difference is reachable by a vector: both have
probability 2^-32 per proof):
  * nonce_rfc_6979 feeds h1 = Hash(point_to_string(H)) to HMAC as it is (RFC 6979's bits2octets reduces it mod n first);
  * the first HMAC_DRBG candidate is taken mod n (RFC 6979 retries while it is 0 or >= n).
Each function names the reference interface it stands in for as ``[ref src/lib.rs:LINE name]``.
"""
from __future__ import annotations

import hashlib
import hmac
from typing import Optional, Tuple

P = 2**256 - 2**224 + 2**192 + 2**96 - 1
A = P - 3
B = 0x5AC635D8AA3A93E7B3EBBD55769886BC651D06B0CC53B0F63BCE3C3E27D2604B
N = 0xFFFFFFFF00000000FFFFFFFFFFFFFFFFBCE6FAADA7179E84F3B9CAC2FC632551
G = (0x6B17D1F2E12C4247F8BCE6E563A440F277037D812DEB33A0F4A13945D898C296,
     0x4FE342E2FE1A7F9B8EE7EB4A7C0F9E162BCE33576B315ECECBB6406837BF51F5)
SUITE_ID = b"\x01"
CHALLENGE_LEN = 16
POINT_LEN = 33

Point = Optional[Tuple[int, int]]          # None = the point at infinity


def sha256(b: bytes) -> bytes:
    return hashlib.sha256(b).digest()


# ---- group law  [ref src/lib.rs:15 `AffinePoint`: ark_ec::short_weierstrass] ----
def is_on_curve(pt: Point) -> bool:
    return pt is None or (pt[1] * pt[1] - (pt[0] ** 3 + A * pt[0] + B)) % P == 0


def add(p1: Point, p2: Point) -> Point:
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    if p1[0] == p2[0]:
        if (p1[1] + p2[1]) % P == 0:
            return None
        lam = (3 * p1[0] * p1[0] + A) * pow(2 * p1[1], -1, P) % P
    else:
        lam = (p2[1] - p1[1]) * pow(p2[0] - p1[0], -1, P) % P
    x = (lam * lam - p1[0] - p2[0]) % P
    return (x, (lam * (p1[0] - x) - p1[1]) % P)


def neg(pt: Point) -> Point:
    return None if pt is None else (pt[0], (-pt[1]) % P)


def mul(k: int, pt: Point) -> Point:
    acc = None
    for bit in bin(k % N)[2:] if k % N else "":
        acc = add(acc, acc)
        if bit == "1":
            acc = add(acc, pt)
    return acc


# ---- codec  [ref src/lib.rs:14 `codec`: Sec1Codec] ----
def point_encode(pt: Point) -> bytes:
    if pt is None:
        return b"\x00"
    return bytes([2 + (pt[1] & 1)]) + pt[0].to_bytes(32, "big")


def point_decode(raw: bytes) -> Point:
    """33-byte compressed string -> point, or raises ValueError.  (The one-byte string 0x00 of the point at infinity
    is not a 33-byte string; the C ABI cannot carry it.)"""
    if len(raw) != 33 or raw[0] not in (2, 3):
        raise ValueError("tag")
    x = int.from_bytes(raw[1:], "big")
    if x >= P:
        raise ValueError("x >= p")
    y2 = (x * x * x + A * x + B) % P
    y = pow(y2, (P + 1) // 4, P)
    if y * y % P != y2:
        raise ValueError("not on the curve")
    return (x, y if (y & 1) == (raw[0] & 1) else P - y)


def scalar_encode(k: int) -> bytes:
    return (k % N).to_bytes(32, "big")


def scalar_decode(raw: bytes) -> int:
    return int.from_bytes(raw, "big") % N


# ---- suite functions ----
def secret_from_seed(seed: bytes) -> int:
    """[ref src/lib.rs:16 `Secret::from_seed`] Hash(seed) read little-endian, mod n."""
    return int.from_bytes(sha256(seed), "little") % N


def hash_to_curve_tai(data: bytes, suite_id: bytes = SUITE_ID) -> Tuple[Point, int]:
    """[ref src/lib.rs:14 `utils`] hash_to_curve_tai_rfc_9381 (RFC 9381 5.4.1.1)."""
    for ctr in range(256):
        try:
            return point_decode(b"\x02" + sha256(suite_id + b"\x01" + data + bytes([ctr]) + b"\x00")), ctr
        except ValueError:
            continue
    raise ValueError("no point within 256 attempts")


def nonce_rfc6979(sk: int, h: Point) -> int:
    """[ref src/lib.rs:14 `utils`] nonce_rfc_6979 as upstream runs it (module docstring)."""
    mac = lambda key, m: hmac.new(key, m, "sha256").digest()
    h1 = sha256(point_encode(h))
    x = scalar_encode(sk)
    v, k = b"\x01" * 32, b"\x00" * 32
    k = mac(k, v + b"\x00" + x + h1)
    v = mac(k, v)
    k = mac(k, v + b"\x01" + x + h1)
    v = mac(k, v)
    v = mac(k, v)
    return scalar_decode(v)


def challenge(points, ad: bytes, suite_id: bytes = SUITE_ID, challenge_len: int = CHALLENGE_LEN) -> int:
    """[ref src/lib.rs:14 `utils`] challenge_rfc_9381 (RFC 9381 5.4.3)."""
    buf = suite_id + b"\x02" + b"".join(point_encode(p) for p in points) + ad + b"\x00"
    return scalar_decode(sha256(buf)[:challenge_len])


def output_hash(gamma: Point, suite_id: bytes = SUITE_ID) -> bytes:
    """[ref src/lib.rs:16 `Output::hash`] point_to_hash_rfc_9381 (cofactor 1)."""
    return sha256(suite_id + b"\x03" + point_encode(gamma) + b"\x00")


def ietf_prove(sk: int, h: Point, ad: bytes, suite_id: bytes = SUITE_ID, challenge_len: int = CHALLENGE_LEN, gen: Point = G):
    """[ref src/lib.rs:14 `ietf::Prover::prove`] -> (gamma, c, s).  suite_id / challenge_len / gen: what a `Suite` impl
    states as data (the product takes them from the descriptor)."""
    gamma, k = mul(sk, h), nonce_rfc6979(sk, h)
    c = challenge([mul(sk, gen), h, gamma, mul(k, gen), mul(k, h)], ad, suite_id, challenge_len)
    return gamma, c, (k + c * sk) % N


def ietf_verify(pk: Point, h: Point, gamma: Point, ad: bytes, c: int, s: int, suite_id: bytes = SUITE_ID,
                challenge_len: int = CHALLENGE_LEN, gen: Point = G) -> bool:
    """[ref src/lib.rs:14 `ietf::Verifier::verify`]"""
    u = add(mul(s, gen), neg(mul(c, pk)))
    v = add(mul(s, h), neg(mul(c, gamma)))
    # a c field holding more than CHALLENGE_LEN bytes is no proof string (vrf_oracle.ietf_verify)
    return challenge([pk, h, gamma, u, v], ad, suite_id, challenge_len) == (c if challenge_len < 32 else c % N)


# ---- Pedersen VRF  [ref src/lib.rs:14 `pedersen`]  (with this suite's codec and hash) ----
# No vector pins this scheme on this suite, and upstream's BLINDING_BASE for it is not known here: the product's built-in
# descriptor carries a nothing-up-my-sleeve point (tools/gen_constants.py), a caller that knows upstream's passes it in.
def default_blinding_base() -> Point:
    for ctr in range(256):
        try:
            return point_decode(b"\x02" + sha256(b"\x01\x01" + b"vrfhip-p256-blinding-base" + bytes([ctr]) + b"\x00"))
        except ValueError:
            continue
    raise AssertionError


def pedersen_blinding(sk: int, h: Point, ad: bytes, suite_id: bytes = SUITE_ID) -> int:
    """[ref src/lib.rs:14 `pedersen::PedersenSuite::blinding`]"""
    return scalar_decode(sha256(suite_id + b"\xCC" + scalar_encode(sk) + point_encode(h) + ad + b"\x00"))


def pedersen_prove(sk: int, h: Point, ad: bytes, bb: Point, suite_id: bytes = SUITE_ID, challenge_len: int = CHALLENGE_LEN,
                   gen: Point = G):
    """[ref src/lib.rs:14 `pedersen::Prover::prove`] -> (gamma, (pk_com, r, ok, s, sb), blinding)."""
    gamma = mul(sk, h)
    b = pedersen_blinding(sk, h, ad, suite_id)
    k, kb = nonce_rfc6979(sk, h), nonce_rfc6979(b, h)
    pk_com = add(mul(sk, gen), mul(b, bb))
    r = add(mul(k, gen), mul(kb, bb))
    ok = mul(k, h)
    c = challenge([pk_com, h, gamma, r, ok], ad, suite_id, challenge_len)
    return gamma, (pk_com, r, ok, (k + c * sk) % N, (kb + c * b) % N), b


def pedersen_verify(h: Point, gamma: Point, proof, ad: bytes, bb: Point, suite_id: bytes = SUITE_ID,
                    challenge_len: int = CHALLENGE_LEN, gen: Point = G) -> bool:
    """[ref src/lib.rs:14 `pedersen::Verifier::verify`]"""
    pk_com, r, ok, s, sb = proof
    c = challenge([pk_com, h, gamma, r, ok], ad, suite_id, challenge_len)
    if add(mul(c, gamma), ok) != mul(s, h):
        return False
    return add(mul(c, pk_com), r) == add(mul(s, gen), mul(sb, bb))


def rfc9381_prove(sk_bytes: bytes, alpha: bytes) -> dict:
    """ECVRF_prove(SK, alpha) of RFC 9381 with every intermediate the appendix prints, through the functions above
    (encode_to_curve_salt = PK_string)."""
    sk = int.from_bytes(sk_bytes, "big")
    pk = mul(sk, G)
    h, ctr = hash_to_curve_tai(point_encode(pk) + alpha)
    k = nonce_rfc6979(sk, h)
    gamma, c, s = ietf_prove(sk, h, b"")
    return dict(pk=point_encode(pk), ctr=ctr, h=point_encode(h), k=scalar_encode(k), u=point_encode(mul(k, G)),
                v=point_encode(mul(k, h)), pi=point_encode(gamma) + c.to_bytes(16, "big") + scalar_encode(s),
                beta=output_hash(gamma))
