"""CPU oracle (TEST INFRASTRUCTURE ONLY) for the EC-VRF hot path of ark-ec-vrfs / ark-vrf.

This file is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.

Parity status
-------------
The reference checkout (/root/reference) is a 17-line deprecated re-export shim:
``src/lib.rs:13-17`` re-exports ``ark_vrf::{codec, ietf, pedersen, ring, suites, utils,
Input, Output, Public, Secret, Suite, ...}`` and ``src/lib.rs:9-11`` is an unconditional
``compile_error!``.  The arithmetic lives in the un-vendored third-party crate ``ark-vrf``
(``Cargo.toml:11-12``: ``ark-vrf = "0.1.0"``, caret requirement, Cargo.lock git-ignored
at ``.gitignore:8`` => exact version unpinned) and its arkworks dependencies (ark-ec,
ark-ff, ark-serialize, ark-ed-on-bls12-381-bandersnatch, sha2), none of which is on this
machine.  By the reference's *own* contents parity is therefore **unpinned**: it holds no
test, golden vector or fixture for this path.

What pins this oracle instead: it restates the published algorithms the upstream crate
implements (RFC 9381 section 5 ECVRF with additional data, RFC 9380 section 5.3.1
expand_message_xmd / section 6.7.1 + 6.8.2 Elligator 2 and the Montgomery->Edwards map,
RFC 8032-style deterministic nonce, arkworks' compressed twisted-Edwards serialisation)
exactly as written down in SURVEY.md Appendix A, and is checked field by field against
the upstream ``bandersnatch_sha-512_ell2`` IETF vectors 1-3 and Pedersen vector 1
reproduced in SURVEY.md Appendix B (tests/golden/bandersnatch_sha512_ell2_kat.json).

Each function names the reference interface it stands in for as
``[ref src/lib.rs:LINE name]`` (the only place that name exists in /root/reference).
"""
from __future__ import annotations

import hashlib
from dataclasses import dataclass, field
from typing import Optional, Tuple

# --------------------------------------------------------------------------------------
# Suite descriptors  [ref src/lib.rs:14 `suites`, src/lib.rs:16 `Suite`]
# --------------------------------------------------------------------------------------

Q = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001  # BLS12-381 Fr


@dataclass(frozen=True)
class SuiteParams:
    name: str
    suite_id: bytes
    q: int            # base field modulus
    a: int            # TE coefficient a (mod q)
    d: int            # TE coefficient d (mod q)
    r: int            # prime subgroup order
    cofactor: int
    gx: int
    gy: int
    bx: int           # Pedersen blinding base
    by: int
    h2c: str          # "ell2" | "tai"
    h2c_dst: bytes = b""
    # Elligator-2 constants on the birationally equivalent Montgomery curve
    mont_j: int = 0
    mont_k: int = 0
    ell2_z: int = 0
    challenge_len: int = 32
    # What a `Suite` impl may override besides its constants (vrfhip_suite_desc.flags): the three deviations that separate
    # RFC 9381's ECVRF-EDWARDS25519-SHA512-TAI from upstream's built-in Ed25519 suite.  False for every upstream suite.
    sign_parity: bool = False      # compressed points carry x mod 2 in bit 255 (RFC 8032) instead of x > q - x
    challenge_le: bool = False     # the truncated challenge hash is a little-endian integer
    hash_cofactor: bool = False    # Output::hash hashes cofactor * Gamma (RFC 9381 proof_to_hash)


BANDERSNATCH = SuiteParams(
    name="bandersnatch_sha512_ell2",
    suite_id=b"Bandersnatch_SHA-512_ELL2",
    q=Q,
    a=Q - 5,
    d=45022363124591815672509500913686876175488063829319466900776701791074614335719,
    r=0x1CFB69D4CA675F520CCE760202687600FF8F87007419047174FD06B52876E7E1,
    cofactor=4,
    gx=18886178867200960497001835917649091219057080094937609519140440539760939937304,
    gy=19188667384257783945677642223292697773471335439753913231509108946878080696678,
    bx=6150229251051246713677296363717454238956877613358614224171740096471278798312,
    by=28442734166467795856797249030329035618871580593056783094884474814923353898473,
    h2c="ell2",
    h2c_dst=b"ECVRF_" + b"Bandersnatch_XMD:SHA-512_ELL2_RO_" + b"Bandersnatch_SHA-512_ELL2",
    mont_j=29978822694968839326280996386011761570173833766074948509196803838190355340952,
    mont_k=25465760566081946422412445027709227188579564747101592991722834452325077642517,
    ell2_z=5,
)

# JubJub: SURVEY.md Appendix A.6 -- [RECALL], parity unpinned (suite id / TAI details /
# blinding base are not authenticated by any vector).  The blinding base below is NOT an
# upstream constant: it is derived deterministically in `jubjub_params()` and travels to
# the device through the suite descriptor, exactly as SURVEY.md section 8b prescribes.
_JUBJUB_D = 19257038036680949359750312669786877991949435402254120286184196891950884077233
_JUBJUB_R = 6554484396890773809930967563523245729705921265872317281365359162392183254199
_JUBJUB_GX = 8076246640662884909881801758704306714034609987455869804520522091855516602923
_JUBJUB_GY = 13262374693698910701929044844600465831413122818447359594527400194675274060458


# --------------------------------------------------------------------------------------
# Field helpers  [ref src/lib.rs:15 `BaseField`, `ScalarField`] (ark-ff Fp semantics)
# --------------------------------------------------------------------------------------

def finv(x: int, p: int) -> int:
    return pow(x, p - 2, p)


def legendre(x: int, p: int) -> int:
    """1 if x is a non-zero square, 0 if x == 0, -1 otherwise."""
    x %= p
    if x == 0:
        return 0
    return 1 if pow(x, (p - 1) // 2, p) == 1 else -1


def fsqrt(x: int, p: int) -> Optional[int]:
    """Tonelli-Shanks; returns *a* root (caller fixes the sign) or None."""
    x %= p
    if x == 0:
        return 0
    if legendre(x, p) != 1:
        return None
    s, t = 0, p - 1
    while t % 2 == 0:
        s += 1
        t //= 2
    z = 2
    while legendre(z, p) != -1:
        z += 1
    m, c, tt, rr = s, pow(z, t, p), pow(x, t, p), pow(x, (t + 1) // 2, p)
    while tt != 1:
        i, t2 = 0, tt
        while t2 != 1:
            t2 = t2 * t2 % p
            i += 1
        b = pow(c, 1 << (m - i - 1), p)
        m, c = i, b * b % p
        tt, rr = tt * c % p, rr * b % p
    return rr


# --------------------------------------------------------------------------------------
# Twisted Edwards group  [ref src/lib.rs:15 `AffinePoint`] (ark-ec TE model semantics)
# Points are affine tuples (x, y); identity = (0, 1).
# --------------------------------------------------------------------------------------

Point = Tuple[int, int]


def te_identity() -> Point:
    return (0, 1)


def te_is_on_curve(S: SuiteParams, P: Point) -> bool:
    x, y = P
    q = S.q
    return (S.a * x * x + y * y - 1 - S.d * x * x % q * y * y) % q == 0


def te_add(S: SuiteParams, P: Point, Qp: Point) -> Point:
    x1, y1 = P
    x2, y2 = Qp
    q = S.q
    dxy = S.d * x1 % q * x2 % q * y1 % q * y2 % q
    x3 = (x1 * y2 + y1 * x2) * finv((1 + dxy) % q, q) % q
    y3 = (y1 * y2 - S.a * x1 % q * x2) * finv((1 - dxy) % q, q) % q
    return (x3 % q, y3 % q)


def te_neg(S: SuiteParams, P: Point) -> Point:
    return ((-P[0]) % S.q, P[1])


def _ext_add(S, P, Qe):
    # extended coordinates, add-2008-hwcd (unified)
    q = S.q
    X1, Y1, Z1, T1 = P
    X2, Y2, Z2, T2 = Qe
    A = X1 * X2 % q
    B = Y1 * Y2 % q
    C = S.d * T1 % q * T2 % q
    D = Z1 * Z2 % q
    E = ((X1 + Y1) * (X2 + Y2) - A - B) % q
    F = (D - C) % q
    G = (D + C) % q
    H = (B - S.a * A) % q
    return (E * F % q, G * H % q, F * G % q, E * H % q)


def te_mul(S: SuiteParams, k: int, P: Point) -> Point:
    """k*P by left-to-right double-and-add on extended coordinates (any k >= 0)."""
    q = S.q
    acc = (0, 1, 1, 0)
    base = (P[0], P[1], 1, P[0] * P[1] % q)
    for bit in bin(k)[2:] if k else "":
        acc = _ext_add(S, acc, acc)
        if bit == "1":
            acc = _ext_add(S, acc, base)
    X, Y, Z, _ = acc
    zi = finv(Z, q)
    return (X * zi % q, Y * zi % q)


def te_in_prime_subgroup(S: SuiteParams, P: Point) -> bool:
    return te_is_on_curve(S, P) and te_mul(S, S.r, P) == (0, 1)


# --------------------------------------------------------------------------------------
# Codec  [ref src/lib.rs:14 `codec`] (ArkworksCodec: LE scalars, compressed TE points)
# SURVEY.md Appendix A.1
# --------------------------------------------------------------------------------------

def scalar_encode(k: int) -> bytes:
    return int(k).to_bytes(32, "little")


def scalar_decode(b: bytes, r: int) -> Optional[int]:
    v = int.from_bytes(b, "little")
    return v if v < r else None


def challenge_decode(b: bytes, r: int) -> int:
    """`ietf::Proof::c` as upstream decodes it: `codec::scalar_decode` = from_le_bytes_mod_order (never fails);
    `s` goes through canonical deserialisation (scalar_decode above: None when >= r)."""
    return int.from_bytes(b, "little") % r


def point_encode(S: SuiteParams, P: Point) -> bytes:
    x, y = P
    out = bytearray(y.to_bytes(32, "little"))
    if (x & 1) if S.sign_parity else (x > (S.q - x) % S.q):
        out[31] |= 0x80
    return bytes(out)


def point_decode(S: SuiteParams, b: bytes) -> Optional[Point]:
    """Decompress; no subgroup check (that is `point_decode_checked`)."""
    if len(b) != 32:
        return None
    raw = bytearray(b)
    flag = bool(raw[31] & 0x80)
    raw[31] &= 0x7F
    y = int.from_bytes(raw, "little")
    q = S.q
    if y >= q:
        return None
    y2 = y * y % q
    den = (S.a - S.d * y2) % q
    if den == 0:
        return None
    x2 = (1 - y2) * finv(den, q) % q
    x = fsqrt(x2, q)
    if x is None:
        return None
    neg = (q - x) % q
    if S.sign_parity:               # RFC 8032 5.1.3: the root whose low bit is the flag; x = 0 with the flag set fails
        if x == 0 and flag:
            return None
        return (x if (x & 1) == flag else neg, y)
    lo, hi = (x, neg) if x <= neg else (neg, x)
    x = hi if flag else lo          # x == 0 with the flag set is accepted, as arkworks does
    return (x, y)


def point_decode_checked(S: SuiteParams, b: bytes) -> Optional[Point]:
    P = point_decode(S, b)
    if P is None or not te_in_prime_subgroup(S, P):
        return None
    return P


# --------------------------------------------------------------------------------------
# SHA-512 / expand_message_xmd  [ref src/lib.rs:14 `utils`] (ark-ec::hashing quirk:
# Z_pad = 48 bytes)
# --------------------------------------------------------------------------------------

def sha512(b: bytes) -> bytes:
    return hashlib.sha512(b).digest()


def xmd_sha512_96(msg: bytes, dst: bytes, z_pad_len: int = 48) -> bytes:
    assert len(dst) <= 255
    dst_prime = dst + bytes([len(dst)])
    b0 = sha512(bytes(z_pad_len) + msg + b"\x00\x60" + b"\x00" + dst_prime)
    b1 = sha512(b0 + b"\x01" + dst_prime)
    b2 = sha512(bytes(x ^ y for x, y in zip(b0, b1)) + b"\x02" + dst_prime)
    return (b1 + b2)[:96]


def hash_to_field2(S: SuiteParams, msg: bytes) -> Tuple[int, int]:
    u = xmd_sha512_96(msg, S.h2c_dst)
    return int.from_bytes(u[:48], "big") % S.q, int.from_bytes(u[48:], "big") % S.q


# --------------------------------------------------------------------------------------
# Elligator 2 -> twisted Edwards  [ref src/lib.rs:14 `utils::hash_to_curve_ell2_rfc_9380`]
# SURVEY.md Appendix A.3
# --------------------------------------------------------------------------------------

def elligator2_te(S: SuiteParams, u: int) -> Point:
    q = S.q
    J, K, Z = S.mont_j, S.mont_k, S.ell2_z
    jk = J * finv(K, q) % q            # J/K
    k2i = finv(K * K % q, q)            # 1/K^2
    den = (1 + Z * u % q * u) % q
    if den == 0:
        den = 1
    x1 = (-jk) * finv(den, q) % q
    gx1 = (x1 * x1 % q * x1 + jk * x1 % q * x1 + x1 * k2i) % q
    x2 = (-x1 - jk) % q
    gx2 = (x2 * x2 % q * x2 + jk * x2 % q * x2 + x2 * k2i) % q
    if legendre(gx1, q) >= 0:          # gx1 square (zero counts as square)
        x = x1
        y = fsqrt(gx1, q)
        if y % 2 == 0:                  # want y odd
            y = (q - y) % q
    else:
        x = x2
        y = fsqrt(gx2, q)
        if y % 2 == 1:                  # want y even
            y = (q - y) % q
    s = x * K % q
    t = y * K % q
    if t * (s + 1) % q == 0:
        return (0, 1)
    v = s * finv(t, q) % q
    w = (s - 1) * finv((s + 1) % q, q) % q
    return (v, w)


def hash_to_curve_ell2(S: SuiteParams, data: bytes) -> Point:
    u0, u1 = hash_to_field2(S, data)
    q0 = elligator2_te(S, u0)
    q1 = elligator2_te(S, u1)
    return te_mul(S, S.cofactor, te_add(S, q0, q1))


def hash_to_curve_tai(S: SuiteParams, data: bytes) -> Optional[Point]:
    """RFC 9381 5.4.1.1 try-and-increment as SURVEY.md A.6 recalls it [unpinned]."""
    for ctr in range(256):
        h = sha512(S.suite_id + b"\x01" + data + bytes([ctr]) + b"\x00")
        P = point_decode(S, h[:32])
        if P is None:
            continue
        P = te_mul(S, S.cofactor, P)
        if P != (0, 1):
            return P
    return None


def data_to_point(S: SuiteParams, data: bytes) -> Optional[Point]:
    """[ref src/lib.rs:15-16 `Input::new` / `Suite::data_to_point`]"""
    if S.h2c == "ell2":
        return hash_to_curve_ell2(S, data)
    return hash_to_curve_tai(S, data)


# --------------------------------------------------------------------------------------
# Secret / nonce / challenge / output hash
# [ref src/lib.rs:16 `Secret`, `Suite::nonce`, `Suite::challenge`; src/lib.rs:15 `Output`]
# SURVEY.md Appendix A.2, A.4
# --------------------------------------------------------------------------------------

def secret_from_seed(S: SuiteParams, seed: bytes) -> int:
    return int.from_bytes(sha512(seed), "little") % S.r


def public_from_secret(S: SuiteParams, sk: int) -> Point:
    return te_mul(S, sk, (S.gx, S.gy))


def nonce_rfc8032(S: SuiteParams, sk: int, H: Point) -> int:
    hsk = sha512(scalar_encode(sk))
    return int.from_bytes(sha512(hsk[32:64] + point_encode(S, H)), "little") % S.r


def challenge_rfc9381(S: SuiteParams, pts, ad: bytes) -> int:
    buf = S.suite_id + b"\x02" + b"".join(point_encode(S, P) for P in pts) + ad + b"\x00"
    return int.from_bytes(sha512(buf)[: S.challenge_len], "little" if S.challenge_le else "big") % S.r


def output_hash(S: SuiteParams, gamma: Point) -> bytes:
    """[ref src/lib.rs:15 `Output::hash`] point_to_hash_rfc_9381, no cofactor clearing (upstream; RFC 9381 clears it)."""
    if S.hash_cofactor:
        gamma = te_mul(S, S.cofactor, gamma)
    return sha512(S.suite_id + b"\x03" + point_encode(S, gamma) + b"\x00")


# --------------------------------------------------------------------------------------
# IETF VRF  [ref src/lib.rs:14 `ietf`]  SURVEY.md Appendix A.4
# --------------------------------------------------------------------------------------

def ietf_prove(S: SuiteParams, sk: int, H: Point, ad: bytes):
    """Returns (gamma, c, s). [ref src/lib.rs:14 `ietf::Prover::prove`]"""
    G = (S.gx, S.gy)
    pk = te_mul(S, sk, G)
    gamma = te_mul(S, sk, H)
    k = nonce_rfc8032(S, sk, H)
    kG = te_mul(S, k, G)
    kH = te_mul(S, k, H)
    c = challenge_rfc9381(S, [pk, H, gamma, kG, kH], ad)
    s = (k + c * sk) % S.r
    return gamma, c, s


def ietf_verify(S: SuiteParams, pk: Point, H: Point, gamma: Point, ad: bytes, c: int, s: int) -> bool:
    """[ref src/lib.rs:14 `ietf::Verifier::verify`]"""
    G = (S.gx, S.gy)
    U = te_add(S, te_mul(S, s, G), te_neg(S, te_mul(S, c, pk)))
    V = te_add(S, te_mul(S, s, H), te_neg(S, te_mul(S, c, gamma)))
    # `Proof::c` is a field element (mod r) when CHALLENGE_LEN = 32; upstream writes a shorter challenge with CHALLENGE_LEN
    # bytes, so a 32-byte field holding more is no proof string: compared as it stands, it never matches
    return challenge_rfc9381(S, [pk, H, gamma, U, V], ad) == (c if S.challenge_len < 32 else c % S.r)


# --------------------------------------------------------------------------------------
# Pedersen VRF  [ref src/lib.rs:14 `pedersen`]  SURVEY.md Appendix A.5
# --------------------------------------------------------------------------------------

def pedersen_blinding(S: SuiteParams, sk: int, H: Point, ad: bytes) -> int:
    buf = S.suite_id + b"\xCC" + scalar_encode(sk) + point_encode(S, H) + ad + b"\x00"
    return int.from_bytes(sha512(buf), "big") % S.r


def pedersen_prove(S: SuiteParams, sk: int, H: Point, ad: bytes):
    """Returns (gamma, (pk_com, R, Ok, s, sb), blinding)."""
    G, B = (S.gx, S.gy), (S.bx, S.by)
    gamma = te_mul(S, sk, H)
    b = pedersen_blinding(S, sk, H, ad)
    k = nonce_rfc8032(S, sk, H)
    kb = nonce_rfc8032(S, b, H)
    pk_com = te_add(S, te_mul(S, sk, G), te_mul(S, b, B))
    R = te_add(S, te_mul(S, k, G), te_mul(S, kb, B))
    Ok = te_mul(S, k, H)
    c = challenge_rfc9381(S, [pk_com, H, gamma, R, Ok], ad)
    s = (k + c * sk) % S.r
    sb = (kb + c * b) % S.r
    return gamma, (pk_com, R, Ok, s, sb), b


def pedersen_verify(S: SuiteParams, H: Point, gamma: Point, ad: bytes, proof) -> bool:
    pk_com, R, Ok, s, sb = proof
    G, B = (S.gx, S.gy), (S.bx, S.by)
    c = challenge_rfc9381(S, [pk_com, H, gamma, R, Ok], ad)
    if te_add(S, Ok, te_mul(S, c, gamma)) != te_mul(S, s, H):
        return False
    lhs = te_add(S, R, te_mul(S, c, pk_com))
    rhs = te_add(S, te_mul(S, s, G), te_mul(S, sb, B))
    return lhs == rhs


# --------------------------------------------------------------------------------------
# JubJub descriptor (unpinned, see note above)
# --------------------------------------------------------------------------------------

# ------------------------------------------------------------------ TE <-> SW map
# [ref src/lib.rs:14 `utils`: te_sw_map::{te_to_sw, sw_to_te}]  The point map between a twisted-Edwards curve and the
# short-Weierstrass form of its Montgomery model (what the `bandersnatch_sw` suite and the ring plumbing sit on).
# Montgomery model  B v^2 = u^3 + A u^2 + u  with  A = 2(a + d)/(a - d),  B = 4/(a - d)  (arkworks' MontCurveConfig);
#   TE -> Mont: (u, v) = ((1 + y)/(1 - y), (1 + y)/((1 - y) x));  Mont -> SW: ((u + A/3)/B, v/B);
#   SW: y^2 = x^3 + a' x + b',  a' = (3 - A^2)/(3 B^2),  b' = (2 A^3 - 9 A)/(27 B^3).
# Pinned as far as this environment allows: for Bandersnatch these formulas give exactly the four coefficients of
# ark-ed-on-bls12-381-bandersnatch (MontCurveConfig COEFF_A / COEFF_B, SWConfig COEFF_A / COEFF_B) as recalled digit for
# digit -- tests/test_te_sw_map.py -- which fixes the scaling of the map (the only freedom left is y -> -y, and upstream
# takes v/B with the sign above).  None where upstream's `inverse()?` returns None: x = 0 or y = 1 (TE -> SW: the identity
# and the point of order 2 have no affine image), y = 0 or B x - A/3 = -1 (SW -> TE).
def te_sw_constants(S: SuiteParams):
    """(A, B, a', b') of the Montgomery and short-Weierstrass models of S's curve."""
    q = S.q
    A = 2 * (S.a + S.d) * finv(S.a - S.d, q) % q
    B = 4 * finv(S.a - S.d, q) % q
    return (A, B, (3 - A * A) * finv(3 * B * B, q) % q, (2 * A ** 3 - 9 * A) * finv(27 * B ** 3, q) % q)


def te_to_sw(S: SuiteParams, P: Point) -> Optional[Point]:
    q = S.q
    x, y = P
    A, B, _, _ = te_sw_constants(S)
    if (1 - y) % q == 0 or (x - x * y) % q == 0:
        return None
    u = (1 + y) * finv(1 - y, q) % q
    v = (1 + y) * finv(x - x * y, q) % q
    return ((u + A * finv(3, q)) * finv(B, q) % q, v * finv(B, q) % q)


def sw_to_te(S: SuiteParams, P: Point) -> Optional[Point]:
    q = S.q
    x, y = P
    A, B, _, _ = te_sw_constants(S)
    mx, my = (B * x - A * finv(3, q)) % q, B * y % q
    if my == 0 or (mx + 1) % q == 0:
        return None
    return (mx * finv(my, q) % q, (mx - 1) * finv(mx + 1, q) % q)


def sw_add(S: SuiteParams, P: Optional[Point], Q: Optional[Point]) -> Optional[Point]:
    """The chord-and-tangent law on the short-Weierstrass form (None = infinity): the tests check the map against it."""
    q = S.q
    _, _, a2, _ = te_sw_constants(S)
    if P is None:
        return Q
    if Q is None:
        return P
    if P[0] == Q[0]:
        if (P[1] + Q[1]) % q == 0:
            return None
        lam = (3 * P[0] * P[0] + a2) * finv(2 * P[1], q) % q
    else:
        lam = (Q[1] - P[1]) * finv(Q[0] - P[0], q) % q
    x = (lam * lam - P[0] - Q[0]) % q
    return (x, (lam * (P[0] - x) - P[1]) % q)


def jubjub_params() -> SuiteParams:
    base = SuiteParams(
        name="jubjub_sha512_tai", suite_id=b"JubJub_SHA-512_TAI", q=Q, a=Q - 1, d=_JUBJUB_D,
        r=_JUBJUB_R, cofactor=8, gx=_JUBJUB_GX, gy=_JUBJUB_GY, bx=0, by=1, h2c="tai")
    # host-supplied blinding base: TAI hash of a fixed label (nothing-up-my-sleeve)
    B = hash_to_curve_tai(base, b"vrfhip-jubjub-blinding-base")
    return SuiteParams(**{**base.__dict__, "bx": B[0], "by": B[1]})


# --------------------------------------------------------------------------------------
# Ed25519 and Baby-JubJub descriptors  [ref src/lib.rs:14 `suites`]  
# Curve constants: RFC 8032 section 5.1 / ark-ed-on-bn254, checked algebraically (tests/test_oracle_kat.py: on the curve,
# order r, cofactor).  Suite strings, CHALLENGE_LEN and the absence of a salt are recollections of upstream; the blinding
# bases are this repository's (TAI hash of a fixed label, as for JubJub): parity with upstream unpinned.
# --------------------------------------------------------------------------------------

Q_25519 = (1 << 255) - 19
Q_BN254 = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def _with_tai_blinding(base: SuiteParams, label: bytes) -> SuiteParams:
    B = hash_to_curve_tai(base, label)
    return SuiteParams(**{**base.__dict__, "bx": B[0], "by": B[1]})


def ed25519_params() -> SuiteParams:
    q = Q_25519
    base = SuiteParams(
        name="ed25519_sha512_tai", suite_id=b"Ed25519_SHA-512_TAI", q=q, a=q - 1,
        d=(-121665 * pow(121666, q - 2, q)) % q, r=(1 << 252) + 27742317777372353535851937790883648493, cofactor=8,
        gx=15112221349535400772501151409588531511454012693041857206046113283949847762202,
        gy=46316835694926478169428394003475163141307993866256225615783033603165251855960,
        bx=0, by=1, h2c="tai", challenge_len=16)
    return _with_tai_blinding(base, b"vrfhip-ed25519-blinding-base")


def baby_jubjub_params() -> SuiteParams:
    q = Q_BN254
    base = SuiteParams(
        name="babyjubjub_sha512_tai", suite_id=b"BabyJubJub_SHA-512_TAI", q=q, a=1,
        d=168696 * pow(168700, q - 2, q) % q,
        r=2736030358979909402780800718157159386076813972158567259200215660948447373041, cofactor=8,
        gx=19698561148652590122159747500897617769866003486955115824547446575314762165298,
        gy=19298250018296453272277890825869354524455968081175474282777126169995084727839,
        bx=0, by=1, h2c="tai")
    return _with_tai_blinding(base, b"vrfhip-babyjubjub-blinding-base")


# --------------------------------------------------------------------------------------
# RFC 9381 ECVRF-EDWARDS25519-SHA512-TAI (section 5.5, suite_string 0x03) through the functions above.
# The suite differs from upstream's Ed25519 suite in data (suite string 0x03; the public key prepended to alpha as the
# encode_to_curve salt) and in the three flags of SuiteParams -- and in how a secret key becomes a scalar and a nonce
# prefix, which is the only part restated here on its own (RFC 8032 key expansion).  Its published vectors (RFC 9381
# Appendix B.3, tests/golden/rfc9381_edwards25519_sha512_tai.json) therefore pin try-and-increment, the codec with
# cofactor clearing, the challenge layout, s = k + c x and the output hash of this oracle -- the code paths that the
# JubJub-family suites have no external vector for.
# --------------------------------------------------------------------------------------

def ed25519_rfc9381_params() -> SuiteParams:
    e = ed25519_params()
    return SuiteParams(**{**e.__dict__, "name": "rfc9381_edwards25519_sha512_tai", "suite_id": b"\x03",
                          "sign_parity": True, "challenge_le": True, "hash_cofactor": True})


def rfc9381_ed25519_expand_key(S: SuiteParams, sk_seed: bytes):
    """RFC 8032 5.1.5: (x, nonce prefix, public key point)."""
    h = sha512(sk_seed)
    a = bytearray(h[:32])
    a[0] &= 248
    a[31] &= 127
    a[31] |= 64
    x = int.from_bytes(a, "little")
    return x, h[32:], te_mul(S, x, (S.gx, S.gy))


def rfc9381_ed25519_prove(sk_seed: bytes, alpha: bytes):
    """ECVRF_prove of RFC 9381 section 5.1 for suite 0x03 -> dict of the vector's fields (bytes)."""
    S = ed25519_rfc9381_params()
    x, prefix, Y = rfc9381_ed25519_expand_key(S, sk_seed)
    pk = point_encode(S, Y)
    H = hash_to_curve_tai(S, pk + alpha)                       # encode_to_curve_salt = PK_string
    h_string = point_encode(S, H)
    gamma = te_mul(S, x, H)
    k = int.from_bytes(sha512(prefix + h_string), "little") % S.r      # RFC 8032-style nonce (section 5.4.2.2)
    U, V = te_mul(S, k, (S.gx, S.gy)), te_mul(S, k, H)
    c = challenge_rfc9381(S, [Y, H, gamma, U, V], b"")
    s = (k + c * x) % S.r
    pi = point_encode(S, gamma) + c.to_bytes(16, "little") + s.to_bytes(32, "little")
    return dict(pk=pk, x=x.to_bytes(32, "little"), h=h_string, k=k.to_bytes(32, "little"), u=point_encode(S, U),
                v=point_encode(S, V), pi=pi, beta=output_hash(S, gamma))


# --------------------------------------------------------------------------------------
# Synthetic benchmark inputs
# --------------------------------------------------------------------------------------

def synth_seed(i: int) -> bytes:
    return int(i).to_bytes(8, "little")


def synth_msg(i: int) -> bytes:
    return sha512(b"vrfhip-msg" + int(i).to_bytes(8, "little"))[:32]
