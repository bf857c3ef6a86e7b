#!/usr/bin/env python3
"""Build-time generator for the constant headers of ark_ec_vrfs_amd/csrc:

    constants.gen.h          field 0: BLS12-381 Fr      (Bandersnatch, JubJub; + the BLS12-381 Fp tower constants)
    constants_f25519.gen.h   field 1: 2^255 - 19        (Ed25519)
    constants_fbn254.gen.h   field 2: BN254 Fr          (Baby-JubJub)
    constants_fp256.gen.h    field 3: NIST P-256 Fp     (secp256r1, short Weierstrass: sw.cuh)

One header per BASE FIELD: the kernels are compiled once per field (-DVRF_FIELD=n, fe.cuh) and every header defines
the same names in `namespace vrfk`.  Everything is derived with Python big ints from the curve parameters (SURVEY.md
Appendix C for field 0; RFC 8032 / ark-ed-on-bn254 for the others, each checked algebraically below): radix-2^29 limb
forms, the images x*R mod q (R = 2^261 for the Montgomery fields, R = 1 for the pseudo-Mersenne field 2^255 - 19), the
lazy-subtraction bias words, the fixed exponents, the 2-Sylow discrete-log tables of the table-driven square root, the
subgroup-test constants.  Build tooling only: nothing here runs on the product path, and it does not import the oracle.

Run:  python tools/gen_constants.py   (rewrites the headers in place; output is committed)
"""
import hashlib
import math
import os

NL, W = 9, 29
MASK = (1 << W) - 1
RBITS = NL * W                      # 261
U_SLACK = 1 << 13                   # "L = 1" means every limb < 2^29 + 2^13

Q_BLS = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
Q_25519 = (1 << 255) - 19
Q_BN254 = 21888242871839275222246405745257275088548364400416034343698204186575808495617
Q_P256 = (1 << 256) - (1 << 224) + (1 << 192) + (1 << 96) - 1

KIND_MONT_Q1, KIND_MONT, KIND_PM25519 = 0, 1, 2


def words32(x, n=8):
    return [(x >> (32 * i)) & 0xFFFFFFFF for i in range(n)]


def carr(name, vals, ty="uint32_t"):
    body = ", ".join("0x%08xu" % v for v in vals)
    return "constexpr %s %s[%d] = {%s};" % (ty, name, len(vals), body)


def pow_program(e, w=4):
    """Left-to-right sliding-window program for x^e with the odd powers x^1..x^(2^w - 1): a list of
    (squarings << 4) | table_index words; entry 0 only selects the start value; index 15 = no multiply."""
    bits = bin(e)[2:]
    ops, pend, i, first = [], 0, 0, True
    while i < len(bits):
        if bits[i] == "0":
            pend += 1
            i += 1
            continue
        l = min(w, len(bits) - i)
        while bits[i + l - 1] == "0":
            l -= 1
        v = int(bits[i:i + l], 2)
        nsq = 0 if first else pend + l
        assert nsq < 4096 and (v - 1) // 2 < 8
        ops.append((nsq << 4) | ((v - 1) // 2))
        first, pend = False, 0
        i += l
    if pend:
        ops.append((pend << 4) | 15)
    # self-check
    acc = None
    for k, op in enumerate(ops):
        nsq, idx = op >> 4, op & 15
        if k == 0:
            acc = 2 * idx + 1
            continue
        acc <<= nsq
        if idx != 15:
            acc += 2 * idx + 1
    assert acc == e
    return ops


class Field:
    def __init__(self, q, kind, z):
        self.q, self.kind, self.z = q, kind, z
        self.R = 1 if kind == KIND_PM25519 else 1 << RBITS
        assert pow(z, (q - 1) // 2, q) == q - 1, "z must be a non-residue"
        s, t = 0, q - 1
        while t % 2 == 0:
            s, t = s + 1, t // 2
        self.s, self.t = s, t
        if kind == KIND_MONT_Q1:
            assert q % (1 << W) == 1
        if kind == KIND_PM25519:
            assert q == (1 << 255) - 19

    def inv(self, x):
        return pow(x % self.q, self.q - 2, self.q)

    def mont(self, x):
        return x * self.R % self.q

    def limbs(self, x, n=NL):
        out = []
        for i in range(n):
            if i == n - 1:
                out.append(x)
            else:
                out.append(x & MASK)
                x >>= W
        assert out[-1] < (1 << 32)
        return out

    def lm(self, x):
        return self.limbs(self.mont(x % self.q))

    def bias(self, k, lb):
        """k*q written with limbs 0..7 >= lb*(2^29 + 2^13) (so a limb-wise subtraction of any
        value whose limbs are < lb*(2^29+2^13) and whose top limb is <= ours never borrows)."""
        off = lb * ((1 << W) + U_SLACK)
        t = k * self.q - sum(off << (W * i) for i in range(NL - 1))
        assert t >= 0
        dig = self.limbs(t)
        out = [off + dig[i] for i in range(NL - 1)] + [dig[NL - 1]]
        assert sum(v << (W * i) for i, v in enumerate(out)) == k * self.q
        assert all(v < (lb + 1) * ((1 << W) + U_SLACK) for v in out[:-1])
        return out

    def sqrt(self, n):
        """Tonelli-Shanks (generator tooling); None for a non-residue."""
        q = self.q
        n %= q
        if n == 0:
            return 0
        if pow(n, (q - 1) // 2, q) != 1:
            return None
        m_, c_, tt, rr = self.s, pow(self.z, self.t, q), pow(n, self.t, q), pow(n, (self.t + 1) // 2, q)
        while tt != 1:
            i, t2 = 0, tt
            while t2 != 1:
                t2 = t2 * t2 % q
                i += 1
            b_ = pow(c_, 1 << (m_ - i - 1), q)
            m_, c_, tt, rr = i, b_ * b_ % q, tt * b_ * b_ % q, rr * b_ % q
        return rr

    def sqrt_tables(self):
        """2-Sylow subgroup (order 2^s) tables for the table-driven Tonelli-Shanks (fe.cuh fe_sqrt_or_zsqrt).
        s = 32: byte digits, tables h^(j 2^(8k)), k = 0..3.  s = 28: digits 8 | 8 | 8 | 4, the same four tables plus
        h^(j 2^4) and h^(j 2^12) (indices 4, 5).  s = 2: no tables, the four torsion elements are constants."""
        q, s, t, z = self.q, self.s, self.t, self.z
        g = pow(z, t, q)                                # generator of the 2^s subgroup
        h = self.inv(g)
        cz = pow(z, (t + 1) // 2, q)                    # sqrt(Z * w) correction: Z^((t+1)/2)
        if s <= 2:      # s = 1 (q = 3 mod 4): sqrt(w) = w^((q+1)/4), no torsion to walk
            return dict(t=t, g=g, h=h, P=[], lut=[0, 0], lut_bits=1, lut_mult=1, cz=cz, levels=[])
        assert s in (28, 32)
        levels = [0, 8, 16, 24] + ([4, 12] if s == 28 else [])
        P = [[pow(h, j << lv, q) for j in range(256)] for lv in levels]
        # order-256 subgroup generated by g^(2^(s-8)); lookup canonical value -> exponent
        gg = pow(g, 1 << (s - 8), q)
        sub = [pow(gg, j, q) for j in range(256)]
        assert len(set(sub)) == 256
        # perfect hash on the low 29-bit limb of the canonical image (x R mod q)
        found = None
        for bits in (9, 10, 11, 12):
            for mult in range(1, 200000, 2):
                seen = {}
                ok = True
                for j, v in enumerate(sub):
                    hsh = (((self.mont(v) & MASK) * mult) & 0xFFFFFFFF) >> (32 - bits)
                    if hsh in seen:
                        ok = False
                        break
                    seen[hsh] = j
                if ok:
                    found = (bits, mult, seen)
                    break
            if found:
                break
        assert found, "no perfect hash"
        bits, mult, seen = found
        lut = [0] * (1 << bits)
        for hsh, j in seen.items():
            lut[hsh] = j
        return dict(t=t, g=g, h=h, P=P, lut=lut, lut_bits=bits, lut_mult=mult, cz=cz, levels=levels)


# ------------------------------------------------------------------------------------------------ curves
class Curve:
    """Twisted Edwards a x^2 + y^2 = 1 + d x^2 y^2 over F (affine tuples, generator tooling only)."""

    def __init__(self, F, a, d, r, cofactor, gx, gy):
        self.F, self.a, self.d, self.r, self.cofactor, self.gx, self.gy = F, a % F.q, d % F.q, r, cofactor, gx, gy
        assert self.on_curve((gx, gy)), "generator not on the curve"
        assert self.mul(r, (gx, gy)) == (0, 1), "generator order"
        assert self.mul(cofactor, (gx, gy)) != (0, 1)

    def on_curve(self, P):
        x, y = P
        q = self.F.q
        return (self.a * x * x + y * y - 1 - self.d * x * x % q * y * y) % q == 0

    def add(self, P1, P2):
        q, inv = self.F.q, self.F.inv
        x1, y1 = P1
        x2, y2 = P2
        k = self.d * x1 % q * x2 % q * y1 % q * y2 % q
        return ((x1 * y2 + y1 * x2) * inv(1 + k) % q, (y1 * y2 - self.a * x1 % q * x2) * inv(1 - k) % q)

    def mul(self, k, P):
        acc = (0, 1)
        for bit in bin(k)[2:]:
            acc = self.add(acc, acc)
            if bit == "1":
                acc = self.add(acc, P)
        return acc

    def decode(self, raw32):
        """ArkworksCodec point decoding (y little-endian, bit 255 = x > q - x), no subgroup test."""
        q = self.F.q
        raw = bytearray(raw32)
        flag = bool(raw[31] & 0x80)
        raw[31] &= 0x7F
        y = int.from_bytes(raw, "little")
        if y >= q:
            return None
        den = (self.a - self.d * y * y) % q
        if den == 0:
            return None
        x = self.F.sqrt((1 - y * y) * self.F.inv(den) % q)
        if x is None:
            return None
        lo, hi = min(x, q - x), max(x, q - x) % q
        return (hi if flag else lo, y)

    def tai_base(self, suite_id, label):
        """Nothing-up-my-sleeve point: try-and-increment hash (RFC 9381 5.4.1.1 shape) of a fixed label.  Used as the
        built-in Pedersen blinding base of the suites whose upstream constant could not be authenticated (SURVEY.md
        A.6): the suite descriptor carries it, a caller that knows upstream's replaces it."""
        for ctr in range(256):
            h = hashlib.sha512(suite_id + b"\x01" + label + bytes([ctr]) + b"\x00").digest()
            P = self.decode(h[:32])
            if P is None:
                continue
            P = self.mul(self.cofactor, P)
            if P != (0, 1):
                return P
        raise AssertionError

    def tate8(self):
        """Constants of the subgroup test by the reduced Tate pairing with a point of order 8 (vrf_core.cuh
        subgroup_by_tate8), for curves whose rational 2-power torsion is cyclic of order 8 and whose field holds the
        8th roots of unity.  E(Fq)[2^inf] = Z8 (one rational point of order 2), so P is in the prime-order subgroup 8E iff
        t_8(T8, P) = f_{8,T8}(P)^((q-1)/8) = 1 for a generator T8 of the 8-torsion.  On the short Weierstrass model
        (via Montgomery: u = (1+y)/(1-y), v = u/x; X = (u + A/3)/B, Y = v/B) the Miller function is
          f = l8^4 l4^2 / (v4^4 v2),   l_R = tangent at R, v_R = vertical through R, R in {T8, T4 = 2T8, T2 = 4T8},
        and with Yn = 1 + y, W = (1 - y) x, Xn = (1 + y) x + (A/3) W every line is a polynomial over B W:
          V2 = (1+y) x, V4 = Xn - c4 W, L4 = Yn - y4 W - lam4 V4, V8 = Xn - c8 W, L8 = Yn - y8 W - lam8 V8,
          f = L8^4 L4^2 V4^4 (B V2 W)^7   modulo 8th powers."""
        F, q, inv = self.F, self.F.q, self.F.inv
        assert self.cofactor == 8 and (q - 1) % 8 == 0
        ja, jd = self.a, self.d
        jA = 2 * (ja + jd) * inv(ja - jd) % q
        jB = 4 * inv(ja - jd) % q
        jaW = (3 - jA * jA) * inv(3 * jB * jB) % q
        T8 = None
        yy = 2
        while T8 is None:
            den = (ja - jd * yy * yy) % q
            x2 = (1 - yy * yy) * inv(den) % q
            if den and pow(x2, (q - 1) // 2, q) == 1:
                xx = F.sqrt(x2)
                cand = self.mul(self.r, (min(xx, q - xx), yy))
                if self.mul(4, cand) != (0, 1):
                    T8 = cand
            yy += 1
        T4 = self.mul(2, T8)
        assert self.mul(4, T8) == (0, q - 1) and self.mul(8, T8) == (0, 1)

        def te2w(P):
            u = (1 + P[1]) * inv(1 - P[1]) % q
            v = u * inv(P[0]) % q
            return ((u + jA * inv(3)) * inv(jB) % q, v * inv(jB) % q)
        W8, W4 = te2w(T8), te2w(T4)
        lam = lambda R: (3 * R[0] * R[0] + jaW) * inv(2 * R[1]) % q
        c = dict(A3=jA * inv(3) % q, B=jB, C8=W8[0] * jB % q, Y8=W8[1] * jB % q, LAM8=lam(W8),
                 C4=W4[0] * jB % q, Y4=W4[1] * jB % q, LAM4=lam(W4))
        self._check_tate8(c, T8)
        return c

    def _check_tate8(self, c, T8):
        """The formula of subgroup_by_tate8 in Python ints against r*P = O on every coset of the 8-torsion."""
        import random
        q = self.F.q
        rnd = random.Random(8)

        def test(P):
            x, y = P
            if (x, y) == (0, 1):
                return True
            Yn, Wv, V2 = (1 + y) % q, (1 - y) * x % q, (1 + y) * x % q
            Xn = (V2 + Wv * c["A3"]) % q
            V4, V8 = (Xn - Wv * c["C4"]) % q, (Xn - Wv * c["C8"]) % q
            L4 = (Yn - Wv * c["Y4"] - V4 * c["LAM4"]) % q
            L8 = (Yn - Wv * c["Y8"] - V8 * c["LAM8"]) % q
            Z = V2 * Wv % q * c["B"] % q
            f = pow(L8, 4, q) * pow(L4, 2, q) % q * pow(V4, 4, q) % q * pow(Z, 7, q) % q
            return f != 0 and pow(f, (q - 1) // 8, q) == 1
        G = (self.gx, self.gy)
        for _ in range(6):
            P = self.mul(rnd.randrange(1, self.r), G)
            for j in range(8):
                Pj = self.add(P, self.mul(j, T8))
                assert test(Pj) == (self.mul(self.r, Pj) == (0, 1)), "tate8 formula disagrees with r*P = O"

    def halve_tate4(self):
        """Constants of the subgroup test by one halving and the reduced Tate pairing with a point of order 4 (vrf_core.cuh
        subgroup_by_halving_tate4), for a = -1 curves whose rational 2-power torsion is cyclic of order 8 over a field with
        q = 5 (mod 8) (Ed25519): Fq holds the 4th roots of unity but not the 8th, so the order-8 pairing of tate8() does not
        exist.  On the Montgomery model v^2 = u^3 + A u^2 + u (u = (1+y)/(1-y), v = c u/x, c = sqrt(-(A+2))):
          P in 2E  iff  u is a square (the curve has ONE rational point of order 2, (0,0), and u generates its descent);
          a half Q of P through the 2-isogeny with kernel (0,0) and its dual: s = sqrt(u), X = A + 2u +- 2v/s (the square
          one of the two), T = X - A, x_Q = (T + sqrt(T^2 - 4))/2, Y' = 8 X^2 v/((A^2-4) - X^2), y_Q = Y' x_Q^2/(1 - x_Q^2);
          Q is defined up to the point of order 2, which lies in 4E, so: P in 8E iff Q in 4E iff t_4(T4, Q) = 1 with
          T4 = (1, y4), y4 = sqrt(A+2): f_{4,T4}(Q) = l^2 / (x_Q ...) = x_Q l^2 modulo 4th powers, l = y_Q - y4 x_Q the
          tangent at T4 (it passes through (0,0)), so the test is chi_4(x_Q l^2) = 1.
        The device code runs the same steps without inversions (every quotient only feeds a character)."""
        F, q, inv = self.F, self.F.q, self.F.inv
        assert self.cofactor == 8 and q % 8 == 5 and self.a == q - 1
        A = 2 * (self.a + self.d) * inv(self.a - self.d) % q
        c = F.sqrt((-(A + 2)) % q)
        y4 = F.sqrt((A + 2) % q)
        assert c is not None and y4 is not None
        k = dict(A=A, C=c, C8=8 * c % q, Y4=y4, BP=(A * A - 4) % q)
        self._check_halve_tate4(k)
        return k

    def _check_halve_tate4(self, k):
        """The inversion-free formula of subgroup_by_halving_tate4 in Python ints against r*P = O on every coset of the
        8-torsion, on the torsion points themselves and on random decodable strings."""
        import random
        F, q = self.F, self.F.q
        rnd = random.Random(4)
        chi = lambda z: pow(z % q, (q - 1) // 2, q)

        def test(P):
            x, y = P
            if (x, y) == (0, 1):
                return True
            n, m = (1 + y) % q, (1 - y) % q
            N = n * m % q
            if chi(N) != 1:
                return False
            sp = F.sqrt(N)
            Td = m * x % q * sp % q
            Tnp, Tnm = 2 * n * (x * sp + k["C"] * m) % q, 2 * n * (x * sp - k["C"] * m) % q
            ATd = k["A"] * Td % q
            Tn = Tnp if chi((ATd + Tnp) * Td) == 1 else Tnm
            Xn = (ATd + Tn) % q
            Dn = (Tn - 2 * Td) * (Tn + 2 * Td) % q
            if chi(Dn) != 1:
                return False
            xn, xd = (Tn + F.sqrt(Dn)) % q, 2 * Td % q
            Bd = Td * ((k["BP"] * Td * Td - Xn * Xn) % q) % q * ((xd * xd - xn * xn) % q) % q
            Bn = (k["C8"] * Xn * Xn % q * n % q * sp % q * xn % q * xd - k["Y4"] * Bd) % q
            BB = Bn * Bd % q
            W = pow(xn, 3, q) * xd % q * BB % q * BB % q
            return W != 0 and pow(W, (q - 1) // 4, q) == 1
        G = (self.gx, self.gy)
        T8 = None
        while T8 is None:
            P = self.decode(rnd.getrandbits(256).to_bytes(32, "little"))
            if P is not None and self.mul(4, self.mul(self.r, P)) != (0, 1):
                T8 = self.mul(self.r, P)
        for j in range(8):
            assert test(self.mul(j, T8)) == (j == 0), "halving/tate4 on the torsion"
        for _ in range(6):
            P = self.mul(rnd.randrange(1, self.r), G)
            for j in range(8):
                Pj = self.add(P, self.mul(j, T8))
                assert test(Pj) == (j == 0), "halving/tate4 formula disagrees with r*P = O"
        for _ in range(24):
            P = self.decode(rnd.getrandbits(256).to_bytes(32, "little"))
            if P is not None:
                assert test(P) == (self.mul(self.r, P) == (0, 1))


F0 = Field(Q_BLS, KIND_MONT_Q1, 5)
F1 = Field(Q_25519, KIND_PM25519, 2)
F2 = Field(Q_BN254, KIND_MONT, 5)
F3 = Field(Q_P256, KIND_MONT, Q_P256 - 1)         # q = 3 (mod 4): -1 is the non-residue

BS = dict(
    a=Q_BLS - 5,
    d=45022363124591815672509500913686876175488063829319466900776701791074614335719,
    r=0x1CFB69D4CA675F520CCE760202687600FF8F87007419047174FD06B52876E7E1,
    gx=18886178867200960497001835917649091219057080094937609519140440539760939937304,
    gy=19188667384257783945677642223292697773471335439753913231509108946878080696678,
    bx=6150229251051246713677296363717454238956877613358614224171740096471278798312,
    by=28442734166467795856797249030329035618871580593056783094884474814923353898473,
    J=29978822694968839326280996386011761570173833766074948509196803838190355340952,
    K=25465760566081946422412445027709227188579564747101592991722834452325077642517,
    Z=5,
)
JJ = dict(
    a=Q_BLS - 1,
    d=19257038036680949359750312669786877991949435402254120286184196891950884077233,
    r=6554484396890773809930967563523245729705921265872317281365359162392183254199,
    gx=8076246640662884909881801758704306714034609987455869804520522091855516602923,
    gy=13262374693698910701929044844600465831413122818447359594527400194675274060458,
)
# Ed25519 (RFC 8032 section 5.1): a = -1, d = -121665/121666, base point with y = 4/5 and even x, order
# l = 2^252 + 27742317777372353535851937790883648493, cofactor 8.
ED = dict(
    a=Q_25519 - 1,
    d=(-121665 * pow(121666, Q_25519 - 2, Q_25519)) % Q_25519,
    r=(1 << 252) + 27742317777372353535851937790883648493,
    gx=15112221349535400772501151409588531511454012693041857206046113283949847762202,
    gy=46316835694926478169428394003475163141307993866256225615783033603165251855960,
)
# Baby-JubJub as ark-ed-on-bn254 states it: a = 1, d = 168696/168700 over BN254 Fr, cofactor 8.
BJ = dict(
    a=1,
    d=168696 * pow(168700, Q_BN254 - 2, Q_BN254) % Q_BN254,
    r=2736030358979909402780800718157159386076813972158567259200215660948447373041,
    gx=19698561148652590122159747500897617769866003486955115824547446575314762165298,
    gy=19298250018296453272277890825869354524455968081175474282777126169995084727839,
)


def glv_constants():
    """Bandersnatch endomorphism psi (degree 2, psi^2 = -2), derived numerically and self-checked:
    psi(x, y) = (c (1 - y^2) / (x y), b (y^2 + b) / (y^2 - b)), psi(P) = lambda P on the subgroup."""
    Q, inv = Q_BLS, F0.inv
    r = BS["r"]
    lam = 0x13b4f3dc4a39a493edf849562b38c72bcfc49db970a5056ed13d21408783df05
    b = 37446463827641770816307242315180085052603635617490163568005256780843403514036
    c = 49199877423542878313146170939139662862850515542392585932876811575731455068989
    assert lam * lam % r == r - 2
    # self-check on a curve point: psi(G) must be on the curve and psi(psi(G)) = -2 G is implied by lambda^2 = -2
    x, y = BS["gx"], BS["gy"]
    px = c * (1 - y * y) % Q * inv(x * y % Q) % Q
    py = b * (y * y + b) % Q * inv((y * y - b) % Q) % Q
    assert (BS["a"] * px * px + py * py - 1 - BS["d"] * px * px % Q * py * py) % Q == 0
    # short lattice basis of {(u, v): u + v*lambda = 0 mod r} by the extended Euclid on (r, lambda)
    rs = [(r, 1, 0), (lam, 0, 1)]
    while rs[-1][0] != 0:
        (r0, s0, t0), (r1, s1, t1) = rs[-2], rs[-1]
        qq = r0 // r1
        rs.append((r0 - qq * r1, s0 - qq * s1, t0 - qq * t1))
    sq = math.isqrt(r)
    i = next(k for k, e in enumerate(rs) if e[0] < sq)
    v1 = (rs[i][0], -rs[i][2])
    cand = [(rs[i - 1][0], -rs[i - 1][2]), (rs[i + 1][0], -rs[i + 1][2])]
    v2 = min(cand, key=lambda v: v[0] * v[0] + v[1] * v[1])
    for v in (v1, v2):
        assert (v[0] + v[1] * lam) % r == 0
    # orient: det = a1*b2 - a2*b1 = +r
    (a1, b1), (a2, b2) = v1, v2
    det = a1 * b2 - a2 * b1
    assert abs(det) == r
    if det < 0:
        a2, b2 = -a2, -b2
    # beta1 = k*b2/r, beta2 = -k*b1/r ; multipliers with a 2^256 shift
    g1 = (abs(b2) << 256) // r
    g2 = (abs(b1) << 256) // r
    return dict(lam=lam, b=b, c=c, a1=a1, b1=b1, a2=a2, b2=b2, g1=g1, g2=g2,
                sg1=1 if b2 >= 0 else -1, sg2=1 if -b1 >= 0 else -1)


def bls_constants():
    """BLS12-381 base field in radix 2^28 (14 limbs, Montgomery R = 2^392) + tower constants."""
    P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
    NLB, WB = 14, 28
    RB = 1 << (NLB * WB)
    MB = (1 << WB) - 1

    def lim(x):
        out = []
        for i in range(NLB):
            if i == NLB - 1:
                out.append(x)
            else:
                out.append(x & MB)
                x >>= WB
        assert out[-1] < (1 << 32)
        return out

    def mont(x):
        return x * RB % P

    def bias(k, lb):
        off = lb * ((1 << WB) + (1 << 12))
        t = k * P - sum(off << (WB * i) for i in range(NLB - 1))
        assert t >= 0
        dig = lim(t)
        out = [off + dig[i] for i in range(NLB - 1)] + [dig[NLB - 1]]
        assert sum(v << (WB * i) for i, v in enumerate(out)) == k * P
        return out

    # Fp2 helpers on tuples for the Frobenius constants xi^(i (p-1)/6), xi = 1 + u
    def f2mul(a, b):
        return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)

    def f2pow(a, e):
        r = (1, 0)
        for bit in bin(e)[2:]:
            r = f2mul(r, r)
            if bit == "1":
                r = f2mul(r, a)
        return r
    gam = [f2pow((1, 1), (P - 1) * i // 6) for i in range(6)]
    return dict(P=P, lim=lim, mont=mont, bias=bias, gam=gam, NLB=NLB, WB=WB,
                pinv=(-pow(P, -1, 1 << WB)) % (1 << WB))


def xy_le(x, y):
    b = int(x).to_bytes(32, "little") + int(y).to_bytes(32, "little")
    return ",".join(str(v) for v in b)


def emit_field_core(ap, F, st, title):
    """The names fe.cuh reads: identical in every field's header."""
    q, R = F.q, F.R
    ap("// GENERATED by tools/gen_constants.py -- do not edit by hand.")
    for line in title:
        ap(line)
    ap("#pragma once")
    ap("#include <cstdint>")
    ap("namespace vrfk {")
    ap("constexpr int NL = 9;")
    ap("constexpr int LW = 29;")
    ap("constexpr uint32_t LMASK = 0x%08xu;" % MASK)
    ap(carr("Q29", F.limbs(q)))
    ap(carr("Q32", words32(q)))
    ap(carr("QM1H32", words32((q - 1) // 2)))
    ap(carr("ONE_M", F.lm(1)))
    if F.kind == KIND_MONT_Q1:
        ap(carr("FIVE_M", F.lm(5)))
    ap(carr("R2_29", F.limbs(R * R % q)))
    ap(carr("R3_29", F.limbs(R * R * R % q)))
    ap("// 2^256 * R mod q: multiplier that folds the high half of a 512-bit integer")
    ap(carr("TWO256_R2", F.limbs((1 << 256) * R * R % q)))
    # coordinates in arkworks' in-memory form (Montgomery, radix 2^256) at the ABI: x 2^256 -> x R is a product by
    # R^2 / 2^256 (2^266 when R = 2^261); x R -> x 2^256 (as a plain integer) is a product by 2^256
    ap(carr("TWO266_29", F.limbs(R * R * F.inv(1 << 256) % q)))
    ap(carr("TWO256_29", F.limbs((1 << 256) % q)))
    for lb in (1, 2):
        for k in (4, 8, 16, 32, 64):
            ap(carr("BIAS_L%d_K%d" % (lb, k), F.bias(k, lb)))
    # exponents (uniform across lanes -> scalar loop control), little-endian 32-bit words
    ap(carr("EXP_INV", words32(q - 2)))
    ap(carr("EXP_SQRT", words32((st["t"] - 1) // 2)))
    ap(carr("EXP_LEGENDRE", words32((q - 1) // 2)))
    ap("// sliding-window (w = 4) programs for the fixed exponents: (squarings << 4) | odd-power index, 15 = none")
    ap(carr("POW_INV_PROG", pow_program(q - 2), "uint16_t"))
    ap(carr("POW_SQRT_PROG", pow_program((st["t"] - 1) // 2), "uint16_t"))
    ap(carr("SQRT_CZ_M", F.lm(st["cz"])))
    ap(carr("SQRT_G_M", F.lm(st["g"])))
    ap("constexpr int SQRT_LUT_BITS = %d;" % st["lut_bits"])
    ap("constexpr uint32_t SQRT_LUT_MULT = %du;" % st["lut_mult"])


def emit_field_traits(ap, F, st):
    """What makes fe.cuh field-generic (appended after the historical block so that field 0's header only grows)."""
    q = F.q
    ap("// ---- field traits (fe.cuh) ----")
    ap("// FIELD_KIND 0: Montgomery R = 2^261 with q = 1 (mod 2^29) (digit = -column, no multiply); 1: Montgomery R = 2^261,")
    ap("// general q (digit = column * NINV29); 2: q = 2^255 - 19, R = 1 (2^261 = 1216 folds the high columns)")
    ap("constexpr int FIELD_KIND = %d;" % F.kind)
    ap("constexpr uint32_t NINV29 = 0x%08xu;   // -q^-1 mod 2^29" % ((-pow(q, -1, 1 << W)) % (1 << W)))
    mulv = 0 if F.kind == KIND_PM25519 else -((-q * 100000) // F.R)      # ceil(q / R * 1e5)
    ap("constexpr int MULV_NUM = %d;           // ceil(1e5 q / R): value bound of a product, mul_v()" % mulv)
    ap("constexpr int V256 = %d;               // 2^256 < V256 q" % ((1 << 256) // q + 1))
    ap("constexpr int VMAX = %d;               // largest value bound V: V q < 2^261 (the top limb stays below 2^29)" % min(64, (1 << RBITS) // q))
    chi_r = 1 if F.R == 1 else (1 if pow(2, (q - 1) // 2, q) == 1 else -1)     # R = 2 (2^130)^2
    ap("constexpr int CHI_R = %d;              // quadratic character of R: symbol(x R) = CHI_R symbol(x)" % chi_r)
    ap("constexpr int SQRT_S = %d;             // 2-adicity of q - 1" % F.s)
    ap("constexpr int SQRT_TABLES = %d;        // 256-entry tables h^(j 2^level); levels %s" % (len(st["levels"]), st["levels"]))
    if F.s == 2:
        g, h = st["g"], st["h"]
        ap("// the 4-torsion g^e, e = 1..3 (canonical images) and h^e = g^-e, e = 1, 2")
        for e in (1, 2, 3):
            ap(carr("SQRT_G%d_M" % e, F.lm(pow(g, e, q))))
        for e in (1, 2):
            ap(carr("SQRT_H%d_M" % e, F.lm(pow(h, e, q))))


def emit_curve(ap, F, tag, name, C, with_gdt=True):
    ap("// ---- %s ----" % name)
    q = F.q
    ap(carr(tag + "_D_M", F.lm(C["d"])))
    ap(carr(tag + "_GX_M", F.lm(C["gx"])))
    ap(carr(tag + "_GY_M", F.lm(C["gy"])))
    ap(carr(tag + "_GDT_M", F.lm(C["d"] * C["gx"] * C["gy"] % q)))
    r = C["r"]
    ap(carr(tag + "_R32", words32(r)))
    ap("constexpr uint32_t %s_R_NINV32 = 0x%08xu;" % (tag, (-pow(r, -1, 1 << 32)) % (1 << 32)))
    ap(carr(tag + "_R_R1", words32((1 << 256) % r)))
    ap(carr(tag + "_R_R2", words32((1 << 512) % r)))


def emit_tate8(ap, F, tag, name, c):
    ap("// %s subgroup test by the Tate pairing with a point of order 8 (vrf_core.cuh subgroup_by_tate8)" % name)
    for k in ("A3", "B", "C8", "Y8", "LAM8", "C4", "Y4", "LAM4"):
        ap(carr("%s_TATE_%s_M" % (tag, k), F.lm(c[k])))


def emit_tables(ap, F, st, points):
    ap("} // namespace vrfk")
    ap("")
    ap("// ---- square-root tables (host copies; uploaded to HBM at context creation) ----")
    ap("namespace vrfk_tables {")
    flat = []
    for tab in st["P"]:
        for j in range(256):
            flat += F.lm(tab[j])
    if not flat:
        flat = [0] * NL                      # fields without tables (2-adicity 2): one unused entry
    ap("// SQRT_P[k][j] = h^(j*2^level_k) (image x R), h = 1/g, j=0..255, 9 limbs each; levels: see SQRT_TABLES")
    ap("static const uint32_t SQRT_P[%d] = {%s};" % (len(flat), ",".join("0x%xu" % v for v in flat)))
    ap("static const uint8_t SQRT_LUT[%d] = {%s};" % (len(st["lut"]), ",".join(str(v) for v in st["lut"])))
    ap("// built-in suite descriptors (vrfhip_suite_desc_default): generator and Pedersen blinding base as x || y,")
    ap("// 32-byte little-endian canonical integers")
    for nm, (x, y) in points:
        ap("static const uint8_t %s[64] = {%s};" % (nm, xy_le(x, y)))
    ap("} // namespace vrfk_tables")


def write(name, out):
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ark_ec_vrfs_amd", "csrc", name)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")
    print("wrote", os.path.normpath(path))


def gen_field0():
    F, Q, inv, mont = F0, Q_BLS, F0.inv, F0.mont
    st = F.sqrt_tables()
    out = []
    ap = out.append
    emit_field_core(ap, F, st, [
        "// Radix-2^29, 9-limb field constants for Fq = BLS12-381 Fr (SURVEY.md Appendix C),",
        "// Montgomery radix R = 2^261.  *_M = Montgomery image (x*R mod q), 29-bit limbs."])
    cbs = Curve(F, BS["a"], BS["d"], BS["r"], 4, BS["gx"], BS["gy"])
    cjj = Curve(F, JJ["a"], JJ["d"], JJ["r"], 8, JJ["gx"], JJ["gy"])
    emit_curve(ap, F, "BS", "Bandersnatch", BS)
    emit_curve(ap, F, "JJ", "JubJub", JJ)
    jbx, jby = cjj.tai_base(b"JubJub_SHA-512_TAI", b"vrfhip-jubjub-blinding-base")
    ap(carr("JJ_BX_M", F.lm(jbx)))
    ap(carr("JJ_BY_M", F.lm(jby)))
    ap(carr("BS_BX_M", F.lm(BS["bx"])))
    ap(carr("BS_BY_M", F.lm(BS["by"])))
    ap(carr("BS_BDT_M", F.lm(BS["d"] * BS["bx"] * BS["by"] % Q)))
    J, K, Z = BS["J"], BS["K"], BS["Z"]
    jk = J * inv(K) % Q
    ap("// Elligator 2 (SURVEY.md A.3): J/K, 1/K^2, K, Z (all Montgomery)")
    ap(carr("BS_ELL2_JK_M", F.lm(jk)))
    ap(carr("BS_ELL2_NJK_M", F.lm(Q - jk)))
    ap(carr("BS_ELL2_K2I_M", F.lm(inv(K * K % Q))))
    ap(carr("BS_ELL2_K_M", F.lm(K)))
    ap(carr("BS_ELL2_Z_M", F.lm(Z)))
    # ---- subgroup membership by 2-descent (Bandersnatch: E(Fq) = Z2 x Z2 x Zr, so the prime-order subgroup is 2E) ----
    # Montgomery model B v^2 = u (u - e2)(u - e3), u = (1 + y)/(1 - y), A = 2(a + d)/(a - d), B = 4/(a - d);
    # P in 2E  <=>  B u, B (u - e2), B (u - e3) are squares (their product is one, so two tests suffice).
    a_bs, d_bs = Q - 5, BS["d"]
    A_m = 2 * (a_bs + d_bs) * inv((a_bs - d_bs) % Q) % Q
    B_m = 4 * inv((a_bs - d_bs) % Q) % Q
    disc = (A_m * A_m - 4) % Q
    assert pow(disc, (Q - 1) // 2, Q) == 1, "full rational 2-torsion expected"
    sd = F.sqrt(disc)
    assert sd * sd % Q == disc
    e2 = (-A_m + sd) * inv(2) % Q
    assert (e2 * e2 + A_m * e2 + 1) % Q == 0
    ap("// subgroup test by 2-descent: B = 4/(a-d), 1 + e2, 1 - e2 with e2 a root of u^2 + A u + 1 (Montgomery model)")
    ap(carr("BS_DESC_B_M", F.lm(B_m)))
    ap(carr("BS_DESC_1PE2_M", F.lm((1 + e2) % Q)))
    ap(carr("BS_DESC_1ME2_M", F.lm((1 - e2) % Q)))
    emit_tate8(ap, F, "JJ", "JubJub", cjj.tate8())
    gl = glv_constants()
    ap("// ---- GLV (Bandersnatch endomorphism psi, psi(P) = LAMBDA * P on the prime-order subgroup) ----")
    ap(carr("BS_GLV_LAMBDA32", words32(gl["lam"])))
    ap(carr("BS_PSI_B_M", F.lm(gl["b"])))
    ap(carr("BS_PSI_C_M", F.lm(gl["c"])))
    ap("// k = k1 + k2*LAMBDA: c1 = sign1 * ((k*G1) >> 256), c2 = sign2 * ((k*G2) >> 256) (rounded),")
    ap("// k1 = k - c1*A1 - c2*A2, k2 = -c1*B1 - c2*B2; lattice vectors as magnitude (5 words) + sign")

    def w5(x):
        assert abs(x) < (1 << 160)
        return [(abs(x) >> (32 * i)) & 0xFFFFFFFF for i in range(5)]
    for nm in ("a1", "b1", "a2", "b2"):
        ap(carr("BS_GLV_%s_MAG" % nm.upper(), w5(gl[nm])))
        ap("constexpr int BS_GLV_%s_NEG = %d;" % (nm.upper(), 1 if gl[nm] < 0 else 0))
    ap(carr("BS_GLV_G1", w5(gl["g1"])))
    ap(carr("BS_GLV_G2", w5(gl["g2"])))
    ap("constexpr int BS_GLV_C1_NEG = %d;" % (1 if gl["sg1"] < 0 else 0))
    ap("constexpr int BS_GLV_C2_NEG = %d;" % (1 if gl["sg2"] < 0 else 0))
    bl = bls_constants()
    ap("// ---- BLS12-381 base field: 14 limbs x 28 bits, Montgomery radix 2^392 ----")
    ap(carr("BLS_P28", bl["lim"](bl["P"])))
    ap("constexpr uint32_t BLS_PINV28 = 0x%08xu;   // -p^-1 mod 2^28" % bl["pinv"])
    ap(carr("BLS_ONE_M", bl["lim"](bl["mont"](1))))
    ap(carr("BLS_R2", bl["lim"]((1 << 784) % bl["P"])))
    ap(carr("BLS_INV2_M", bl["lim"](bl["mont"](pow(2, -1, bl["P"])))))
    ap(carr("BLS_P_WORDS", [(bl["P"] >> (32 * i)) & 0xFFFFFFFF for i in range(12)]))
    ap(carr("BLS_EXP_INV", [((bl["P"] - 2) >> (32 * i)) & 0xFFFFFFFF for i in range(12)]))
    for lb in (1, 2, 4):
        for k in (4, 8, 16, 32, 64):
            ap(carr("BLS_BIAS_L%d_K%d" % (lb, k), bl["bias"](k, lb)))
    for i in range(1, 6):
        ap(carr("BLS_GAMMA%d_RE_M" % i, bl["lim"](bl["mont"](bl["gam"][i][0]))))
        ap(carr("BLS_GAMMA%d_IM_M" % i, bl["lim"](bl["mont"](bl["gam"][i][1]))))
    emit_field_traits(ap, F, st)
    emit_tables(ap, F, st, [("BS_G_XY", (BS["gx"], BS["gy"])), ("BS_B_XY", (BS["bx"], BS["by"])),
                            ("JJ_G_XY", (JJ["gx"], JJ["gy"])), ("JJ_B_XY", (jbx, jby))])
    write("constants.gen.h", out)
    print("  field 0: lut_bits", st["lut_bits"], "mult", st["lut_mult"])


def gen_field1():
    F = F1
    st = F.sqrt_tables()
    out = []
    ap = out.append
    emit_field_core(ap, F, st, [
        "// Radix-2^29, 9-limb field constants for Fq = 2^255 - 19 (Ed25519, RFC 8032 section 5.1).",
        "// Pseudo-Mersenne arithmetic: R = 1, *_M = the value itself (x mod q), 29-bit limbs."])
    ced = Curve(F, ED["a"], ED["d"], ED["r"], 8, ED["gx"], ED["gy"])
    assert ED["gy"] == 4 * F.inv(5) % F.q and ED["gx"] % 2 == 0
    emit_curve(ap, F, "ED", "Ed25519", ED)
    bx, by = ced.tai_base(b"Ed25519_SHA-512_TAI", b"vrfhip-ed25519-blinding-base")
    ap(carr("ED_BX_M", F.lm(bx)))
    ap(carr("ED_BY_M", F.lm(by)))
    ap("// ---- subgroup test by one halving + the order-4 Tate pairing (vrf_core.cuh subgroup_by_halving_tate4) ----")
    ap("// Montgomery A, c = sqrt(-(A+2)), 8c, y4 = sqrt(A+2) (T4 = (1, y4) has order 4), A^2 - 4")
    for nm, v in sorted(ced.halve_tate4().items()):
        ap(carr("ED_HALVE_%s_M" % nm, F.lm(v)))
    emit_field_traits(ap, F, st)
    emit_tables(ap, F, st, [("ED_G_XY", (ED["gx"], ED["gy"])), ("ED_B_XY", (bx, by))])
    write("constants_f25519.gen.h", out)


def gen_field2():
    F = F2
    st = F.sqrt_tables()
    out = []
    ap = out.append
    emit_field_core(ap, F, st, [
        "// Radix-2^29, 9-limb field constants for Fq = BN254 Fr (Baby-JubJub, ark-ed-on-bn254),",
        "// Montgomery radix R = 2^261.  *_M = Montgomery image (x*R mod q), 29-bit limbs."])
    cbj = Curve(F, BJ["a"], BJ["d"], BJ["r"], 8, BJ["gx"], BJ["gy"])
    emit_curve(ap, F, "BJ", "Baby-JubJub", BJ)
    bx, by = cbj.tai_base(b"BabyJubJub_SHA-512_TAI", b"vrfhip-babyjubjub-blinding-base")
    ap(carr("BJ_BX_M", F.lm(bx)))
    ap(carr("BJ_BY_M", F.lm(by)))
    emit_tate8(ap, F, "BJ", "Baby-JubJub", cbj.tate8())
    emit_field_traits(ap, F, st)
    emit_tables(ap, F, st, [("BJ_G_XY", (BJ["gx"], BJ["gy"])), ("BJ_B_XY", (bx, by))])
    write("constants_fbn254.gen.h", out)
    print("  field 2: lut_bits", st["lut_bits"], "mult", st["lut_mult"])


# NIST P-256 / secp256r1 (SP 800-186 3.2.1.3, SEC 2 2.4.2): y^2 = x^3 - 3x + b over Fp, prime order n, cofactor 1.
P256 = dict(
    b=0x5AC635D8AA3A93E7B3EBBD55769886BC651D06B0CC53B0F63BCE3C3E27D2604B,
    n=0xFFFFFFFF00000000FFFFFFFFFFFFFFFFBCE6FAADA7179E84F3B9CAC2FC632551,
    gx=0x6B17D1F2E12C4247F8BCE6E563A440F277037D812DEB33A0F4A13945D898C296,
    gy=0x4FE342E2FE1A7F9B8EE7EB4A7C0F9E162BCE33576B315ECECBB6406837BF51F5,
)


def sw_add(F, P1, P2):
    """Affine short-Weierstrass addition with a = -3 (None = the point at infinity); generator tooling."""
    q = F.q
    if P1 is None:
        return P2
    if P2 is None:
        return P1
    if P1[0] == P2[0]:
        if (P1[1] + P2[1]) % q == 0:
            return None
        lam = (3 * P1[0] * P1[0] - 3) * F.inv(2 * P1[1]) % q
    else:
        lam = (P2[1] - P1[1]) * F.inv(P2[0] - P1[0]) % q
    x = (lam * lam - P1[0] - P2[0]) % q
    return (x, (lam * (P1[0] - x) - P1[1]) % q)


def sw_mul(F, k, P):
    acc = None
    for bit in bin(k)[2:]:
        acc = sw_add(F, acc, acc)
        if bit == "1":
            acc = sw_add(F, acc, P)
    return acc


def gen_field3():
    F = F3
    q = F.q
    st = F.sqrt_tables()
    out = []
    ap = out.append
    emit_field_core(ap, F, st, [
        "// Radix-2^29, 9-limb field constants for Fp = NIST P-256 (secp256r1), Montgomery radix R = 2^261.",
        "// *_M = Montgomery image (x*R mod p), 29-bit limbs.  p = -1 (mod 2^96): the Montgomery digit is the column itself."])
    G = (P256["gx"], P256["gy"])
    n = P256["n"]
    assert (G[1] * G[1] - (G[0] ** 3 - 3 * G[0] + P256["b"])) % q == 0, "generator not on the curve"
    assert sw_mul(F, n, G) is None and sw_mul(F, n - 1, G) == (G[0], q - G[1]), "generator order"
    assert abs(n - (q + 1)) <= 2 * math.isqrt(q) + 1 and all(n % s for s in (2, 3, 5, 7, 11, 13))   # cofactor 1 by Hasse
    ap("// ---- secp256r1: y^2 = x^3 - 3x + b, prime order n, cofactor 1 ----")
    ap(carr("P256_B_M", F.lm(P256["b"])))
    ap(carr("P256_GX_M", F.lm(G[0])))
    ap(carr("P256_GY_M", F.lm(G[1])))
    ap(carr("P256_R32", words32(n)))
    ap("constexpr uint32_t P256_R_NINV32 = 0x%08xu;" % ((-pow(n, -1, 1 << 32)) % (1 << 32)))
    ap(carr("P256_R_R1", words32((1 << 256) % n)))
    ap(carr("P256_R_R2", words32((1 << 512) % n)))
    # built-in Pedersen blinding base: upstream's constant for this suite is not known here, so -- as for JubJub, Ed25519
    # and Baby-JubJub -- a nothing-up-my-sleeve point: try-and-increment (RFC 9381 5.4.1.1 with this suite's own string
    # and interpret_hash_value_as_a_point = 0x02 || hash) of a fixed label.  A caller that knows upstream's passes it in
    # the descriptor.
    Bp = None
    for ctr in range(256):
        h = hashlib.sha256(b"\x01\x01" + b"vrfhip-p256-blinding-base" + bytes([ctr]) + b"\x00").digest()
        x = int.from_bytes(h, "big")
        if x >= q:
            continue
        y = F.sqrt((x * x * x - 3 * x + P256["b"]) % q)
        if y is None:
            continue
        Bp = (x, y if y % 2 == 0 else q - y)
        break
    assert Bp and sw_mul(F, n, Bp) is None and Bp != G
    emit_field_traits(ap, F, st)
    emit_tables(ap, F, st, [("P256_G_XY", G), ("P256_B_XY", Bp)])
    write("constants_fp256.gen.h", out)


def main():
    gen_field3()
    gen_field0()
    gen_field1()
    gen_field2()


if __name__ == "__main__":
    main()
