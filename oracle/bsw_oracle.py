"""CPU oracle (TEST INFRASTRUCTURE ONLY) for the short-Weierstrass Bandersnatch suite of ark-ec-vrfs / ark-vrf
(`suites::bandersnatch_sw`, /root/reference src/lib.rs:14; upstream name "Bandersnatch_SW_SHA-512_TAI").

This file is the checker, never the product: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it.

Parity status: UNPINNED.  The reference checkout holds no code or vector for this suite (it re-exports the un-vendored crate
``ark-vrf``), and no published vector for it is on this machine.  What this restatement rests on:
  * the curve: y^2 = x^3 + a' x + b' over the BLS12-381 scalar field with the coefficients te_sw_constants() derives from
    the twisted-Edwards Bandersnatch -- equal digit for digit to ark-ed-on-bls12-381-bandersnatch's SWConfig as recalled
    (tests/test_te_sw_map.py);
  * the scheme: the same RFC 9381 / ark-vrf recipe as oracle/vrf_oracle.py (pinned there by the upstream Bandersnatch vectors),
    with arkworks' short-Weierstrass point codec in place of the twisted-Edwards one, and try-and-increment hash-to-curve;
  * the ONLY check against something outside this file: every group element here is the te_sw_map image of the element the
    (vector-pinned) twisted-Edwards oracle computes from the same secret and input point (tests/test_bandersnatch_sw.py).
Recalled, not authenticated: the suite string; the generator and blinding base (taken as the te_sw_map images of the
twisted-Edwards suite's); the flag convention of the 33-byte compressed form (below).

The arithmetic is done ON THE WEIERSTRASS CURVE (Jacobian chord-and-tangent), independent of the device code, which runs the
group law on the twisted-Edwards model and crosses the map at the codec.

Each function names the reference interface it stands in for as ``[ref src/lib.rs:LINE name]``.
"""
from __future__ import annotations

import hashlib
from typing import Optional, Tuple

from . import vrf_oracle as vo

Q = vo.Q
R = vo.BANDERSNATCH.r
COFACTOR = 4
_, _, A, B = vo.te_sw_constants(vo.BANDERSNATCH)
SUITE_ID = b"Bandersnatch_SW_SHA-512_TAI"
CHALLENGE_LEN = 32
POINT_LEN = 33                              # 255-bit x + 2 flag bits do not fit 32 bytes
G = vo.te_to_sw(vo.BANDERSNATCH, (vo.BANDERSNATCH.gx, vo.BANDERSNATCH.gy))
BLINDING_BASE = vo.te_to_sw(vo.BANDERSNATCH, (vo.BANDERSNATCH.bx, vo.BANDERSNATCH.by))

Point = Optional[Tuple[int, int]]           # None = the point at infinity
FLAG_NEG, FLAG_INF = 0x80, 0x40             # SWFlags::YIsNegative (y > -y), SWFlags::PointAtInfinity


def sha512(b: bytes) -> bytes:
    return hashlib.sha512(b).digest()


# ---- group law  [ref src/lib.rs:15 `AffinePoint`: ark_ec::short_weierstrass::Affine<SWConfig>] ----
def is_on_curve(p: Point) -> bool:
    return p is None or (p[1] * p[1] - (p[0] ** 3 + A * p[0] + B)) % Q == 0


def _jdbl(p):
    X, Y, Z = p
    if Y == 0 or Z == 0:
        return (1, 1, 0)
    YY = Y * Y % Q
    S = 4 * X * YY % Q
    M = (3 * X * X + A * pow(Z, 4, Q)) % Q
    X3 = (M * M - 2 * S) % Q
    return (X3, (M * (S - X3) - 8 * YY * YY) % Q, 2 * Y * Z % Q)


def _jadd(p, q):
    if p[2] == 0:
        return q
    if q[2] == 0:
        return p
    Z1Z1, Z2Z2 = p[2] * p[2] % Q, q[2] * q[2] % Q
    U1, U2 = p[0] * Z2Z2 % Q, q[0] * Z1Z1 % Q
    S1, S2 = p[1] * q[2] * Z2Z2 % Q, q[1] * p[2] * Z1Z1 % Q
    if U1 == U2:
        return _jdbl(p) if S1 == S2 else (1, 1, 0)
    H, Rr = (U2 - U1) % Q, (S2 - S1) % Q
    HH = H * H % Q
    HHH, V = H * HH % Q, U1 * HH % Q
    X3 = (Rr * Rr - HHH - 2 * V) % Q
    return (X3, (Rr * (V - X3) - S1 * HHH) % Q, H * p[2] * q[2] % Q)


def _to_j(p: Point):
    return (1, 1, 0) if p is None else (p[0], p[1], 1)


def _from_j(p) -> Point:
    if p[2] == 0:
        return None
    zi = pow(p[2], -1, Q)
    return (p[0] * zi * zi % Q, p[1] * zi * zi * zi % Q)


def add(p: Point, q: Point) -> Point:
    return _from_j(_jadd(_to_j(p), _to_j(q)))


def neg(p: Point) -> Point:
    return None if p is None else (p[0], (Q - p[1]) % Q)


def mul(k: int, p: Point) -> Point:
    acc, base = (1, 1, 0), _to_j(p)
    for bit in bin(k)[2:] if k else "":
        acc = _jdbl(acc)
        if bit == "1":
            acc = _jadd(acc, base)
    return _from_j(acc)


def in_prime_subgroup(p: Point) -> bool:
    """arkworks' `is_in_correct_subgroup_assuming_on_curve`: r * P = O."""
    return mul(R, p) is None


# ---- codec  [ref src/lib.rs:14 `codec`: ArkworksCodec over short_weierstrass::Affine] ----
# 33 bytes: x as a 32-byte little-endian integer, then ONE byte whose top two bits are the flags (bit 7: y is the larger of
# {y, -y}; bit 6: the point at infinity, written with x = 0).  On reading, both flag bits set is an error, the low six bits of
# the flag byte are not looked at, x must be < q, and with the infinity flag x is not looked at further.
def point_encode(p: Point) -> bytes:
    if p is None:
        return bytes(32) + bytes([FLAG_INF])
    x, y = p
    return x.to_bytes(32, "little") + bytes([FLAG_NEG if y > (Q - y) % Q else 0])


def point_decode(b: bytes) -> Tuple[bool, Point]:
    """(ok, point) -- `deserialize_compressed_unchecked`: on the curve, any subgroup."""
    if len(b) != POINT_LEN:
        return False, None
    fl = b[32] & 0xC0
    x = int.from_bytes(b[:32], "little")
    if fl == 0xC0 or x >= Q:
        return False, None
    if fl == FLAG_INF:
        return True, None
    y = vo.fsqrt((x * x * x + A * x + B) % Q, Q)
    if y is None:
        return False, None
    lo, hi = sorted((y, (Q - y) % Q))
    return True, (x, hi if fl == FLAG_NEG else lo)


def point_decode_checked(b: bytes) -> Tuple[bool, Point]:
    ok, p = point_decode(b)
    if not ok or not in_prime_subgroup(p):
        return False, None
    return True, p


def scalar_encode(k: int) -> bytes:
    return int(k).to_bytes(32, "little")


# ---- [ref src/lib.rs:14 `utils`: hash_to_curve_tai_rfc_9381] ----
def hash_to_curve_tai(data: bytes) -> Point:
    for ctr in range(256):
        h = sha512(SUITE_ID + b"\x01" + data + bytes([ctr]) + b"\x00")
        ok, p = point_decode(h[:POINT_LEN])
        if not ok:
            continue
        p = mul(COFACTOR, p)
        if p is not None:
            return p
    raise ValueError("no point in 256 tries")


# ---- [ref src/lib.rs:16 `Secret`, `Suite::nonce`, `Suite::challenge`; :15 `Output::hash`] ----
def secret_from_seed(seed: bytes) -> int:
    return int.from_bytes(sha512(seed), "little") % R


def public_from_secret(sk: int) -> Point:
    return mul(sk, G)


def nonce_rfc8032(sk: int, h: Point) -> int:
    hsk = sha512(scalar_encode(sk))
    return int.from_bytes(sha512(hsk[32:64] + point_encode(h)), "little") % R


def challenge(pts, ad: bytes) -> int:
    buf = SUITE_ID + b"\x02" + b"".join(point_encode(p) for p in pts) + ad + b"\x00"
    return int.from_bytes(sha512(buf)[:CHALLENGE_LEN], "big") % R


def output_hash(gamma: Point) -> bytes:
    return sha512(SUITE_ID + b"\x03" + point_encode(gamma) + b"\x00")


# ---- [ref src/lib.rs:14 `ietf`] ----
def ietf_prove(sk: int, h: Point, ad: bytes):
    pk, gamma = mul(sk, G), mul(sk, h)
    k = nonce_rfc8032(sk, h)
    c = challenge([pk, h, gamma, mul(k, G), mul(k, h)], ad)
    return gamma, c, (k + c * sk) % R


def ietf_verify(pk: Point, h: Point, gamma: Point, ad: bytes, c: int, s: int) -> bool:
    u = add(mul(s, G), neg(mul(c, pk)))
    v = add(mul(s, h), neg(mul(c, gamma)))
    return challenge([pk, h, gamma, u, v], ad) == c % R


def ietf_verify_bytes(pk: bytes, h: bytes, gamma: bytes, ad: bytes, c: bytes, s: bytes) -> int:
    """Wire form -> status as the C ABI reports it: 0 verified, 1 VerificationFailure, 2 InvalidData."""
    ok0, p0 = point_decode_checked(pk)
    ok1, p1 = point_decode_checked(h)
    ok2, p2 = point_decode_checked(gamma)
    sv = int.from_bytes(s, "little")
    if not (ok0 and ok1 and ok2) or sv >= R:
        return 2
    return 0 if ietf_verify(p0, p1, p2, ad, int.from_bytes(c, "little"), sv) else 1


# ---- [ref src/lib.rs:14 `pedersen`] ----
def pedersen_blinding(sk: int, h: Point, ad: bytes) -> int:
    return int.from_bytes(sha512(SUITE_ID + b"\xCC" + scalar_encode(sk) + point_encode(h) + ad + b"\x00"), "big") % R


def pedersen_prove(sk: int, h: Point, ad: bytes):
    """Returns (gamma, (pk_com, R, Ok, s, sb), blinding)."""
    gamma = mul(sk, h)
    b = pedersen_blinding(sk, h, ad)
    k, kb = nonce_rfc8032(sk, h), nonce_rfc8032(b, h)
    pk_com = add(mul(sk, G), mul(b, BLINDING_BASE))
    rr = add(mul(k, G), mul(kb, BLINDING_BASE))
    ok = mul(k, h)
    c = challenge([pk_com, h, gamma, rr, ok], ad)
    return gamma, (pk_com, rr, ok, (k + c * sk) % R, (kb + c * b) % R), b


def pedersen_verify(h: Point, gamma: Point, ad: bytes, proof) -> bool:
    pk_com, rr, ok, s, sb = proof
    c = challenge([pk_com, h, gamma, rr, ok], ad)
    if add(ok, mul(c, gamma)) != mul(s, h):
        return False
    return add(rr, mul(c, pk_com)) == add(mul(s, G), mul(sb, BLINDING_BASE))


def pedersen_verify_bytes(h: bytes, gamma: bytes, pk_com: bytes, rr: bytes, ok: bytes, s: bytes, sb: bytes, ad: bytes) -> int:
    pts = [point_decode_checked(b) for b in (h, gamma, pk_com, rr, ok)]
    sv, sbv = int.from_bytes(s, "little"), int.from_bytes(sb, "little")
    if not all(p[0] for p in pts) or sv >= R or sbv >= R:
        return 2
    return 0 if pedersen_verify(pts[0][1], pts[1][1], ad, (pts[2][1], pts[3][1], pts[4][1], sv, sbv)) else 1
