"""BLS12-381 pairing oracle (TEST INFRASTRUCTURE ONLY): the tail of `ring::Verifier::verify`
(/root/reference src/lib.rs:14 `ring`), i.e. ark_ec::pairing::Pairing::{multi_miller_loop,
final_exponentiation} on ark-bls12-381, restated from the published construction (optimal ate
pairing, loop count |x| = 0xd201000000010000, x < 0; tower Fp2 = Fp[u]/(u^2+1),
Fp6 = Fp2[v]/(v^3 - (1+u)), Fp12 = Fp6[w]/(w^2 - v); M-type sextic twist).

Parity status: unpinned by the reference (no vectors).  Pinned here by
algebra: bilinearity e(aP, bQ) = e(P, Q)^(ab), non-degeneracy, e^r = 1, and agreement between two
independent formulations (affine Miller loop with explicit untwisted lines + exponentiation by
(p^12-1)/r as an integer, versus the projective/x-chain formulation the device mirrors).
The device returns only the boolean  prod_i e(P_i, Q_i) == 1, which does not depend on the
normalisation of the pairing.
"""
from __future__ import annotations

P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
X_ABS = 0xD201000000010000          # the curve parameter is x = -X_ABS

G1X = 0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB
G1Y = 0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1
G2X = (0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
       0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E)
G2Y = (0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
       0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE)


# ---------------------------------------------------------------------------- Fp2
class Fp2:
    __slots__ = ("a", "b")

    def __init__(self, a=0, b=0):
        self.a, self.b = a % P, b % P

    def __add__(s, o): return Fp2(s.a + o.a, s.b + o.b)
    def __sub__(s, o): return Fp2(s.a - o.a, s.b - o.b)
    def __neg__(s): return Fp2(-s.a, -s.b)
    def __eq__(s, o): return s.a == o.a and s.b == o.b
    def __mul__(s, o):
        if isinstance(o, int):
            return Fp2(s.a * o, s.b * o)
        return Fp2(s.a * o.a - s.b * o.b, s.a * o.b + s.b * o.a)
    def sq(s): return s * s
    def conj(s): return Fp2(s.a, -s.b)
    def inv(s):
        d = pow(s.a * s.a + s.b * s.b, P - 2, P)
        return Fp2(s.a * d, -s.b * d)
    def mul_xi(s):                       # * (1 + u)
        return Fp2(s.a - s.b, s.a + s.b)
    def is_zero(s): return s.a == 0 and s.b == 0
    def __repr__(s): return "Fp2(%x, %x)" % (s.a, s.b)


F2_0, F2_1 = Fp2(0, 0), Fp2(1, 0)
XI = Fp2(1, 1)


def fp2_pow(a, e):
    r = F2_1
    for bit in bin(e)[2:]:
        r = r.sq()
        if bit == "1":
            r = r * a
    return r


# ---------------------------------------------------------------------------- Fp6 = Fp2[v]/(v^3 - xi)
class Fp6:
    __slots__ = ("c",)

    def __init__(self, c0=F2_0, c1=F2_0, c2=F2_0):
        self.c = (c0, c1, c2)

    def __add__(s, o): return Fp6(*[x + y for x, y in zip(s.c, o.c)])
    def __sub__(s, o): return Fp6(*[x - y for x, y in zip(s.c, o.c)])
    def __neg__(s): return Fp6(*[-x for x in s.c])
    def __eq__(s, o): return all(x == y for x, y in zip(s.c, o.c))
    def __mul__(s, o):
        a0, a1, a2 = s.c
        b0, b1, b2 = o.c
        return Fp6(a0 * b0 + (a1 * b2 + a2 * b1).mul_xi(),
                   a0 * b1 + a1 * b0 + (a2 * b2).mul_xi(),
                   a0 * b2 + a1 * b1 + a2 * b0)
    def mul_v(s):                        # * v
        return Fp6(s.c[2].mul_xi(), s.c[0], s.c[1])
    def inv(s):
        a0, a1, a2 = s.c
        t0 = a0.sq() - (a1 * a2).mul_xi()
        t1 = a2.sq().mul_xi() - a0 * a1
        t2 = a1.sq() - a0 * a2
        d = (a0 * t0 + (a2 * t1 + a1 * t2).mul_xi()).inv()
        return Fp6(t0 * d, t1 * d, t2 * d)


F6_0, F6_1 = Fp6(), Fp6(F2_1)


# ---------------------------------------------------------------------------- Fp12 = Fp6[w]/(w^2 - v)
class Fp12:
    __slots__ = ("c0", "c1")

    def __init__(self, c0=F6_0, c1=F6_0):
        self.c0, self.c1 = c0, c1

    def __mul__(s, o):
        return Fp12(s.c0 * o.c0 + (s.c1 * o.c1).mul_v(), s.c0 * o.c1 + s.c1 * o.c0)
    def sq(s): return s * s
    def __eq__(s, o): return s.c0 == o.c0 and s.c1 == o.c1
    def conj(s): return Fp12(s.c0, -s.c1)
    def inv(s):
        d = (s.c0 * s.c0 - (s.c1 * s.c1).mul_v()).inv()
        return Fp12(s.c0 * d, -(s.c1 * d))
    def coeffs(s):                       # 12 Fp values: c0.c0.a, c0.c0.b, c0.c1.a, ...
        out = []
        for c6 in (s.c0, s.c1):
            for c2 in c6.c:
                out += [c2.a, c2.b]
        return out


F12_1 = Fp12(F6_1, F6_0)


def fp12_pow(a, e):
    r = F12_1
    for bit in bin(e)[2:]:
        r = r.sq()
        if bit == "1":
            r = r * a
    return r


def fp12_from_014(c0, c1, c4):
    """c0 + c1*v + c4*v*w"""
    return Fp12(Fp6(c0, c1, F2_0), Fp6(F2_0, c4, F2_0))


# Frobenius: (a + b u)^p = a - b u; v^p = gamma_v * v with gamma = xi^((p-1)/3); w^p = xi^((p-1)/6) w
GAMMA = [fp2_pow(XI, (P - 1) * i // 6) for i in range(6)]     # xi^(i (p-1)/6)


def fp12_frob(f):
    """f^p"""
    cs = [f.c0.c[0], f.c1.c[0], f.c0.c[1], f.c1.c[1], f.c0.c[2], f.c1.c[2]]    # coefficients of w^0..w^5
    out = [cs[i].conj() * GAMMA[i] for i in range(6)]
    return Fp12(Fp6(out[0], out[2], out[4]), Fp6(out[1], out[3], out[5]))


def fp12_frob_k(f, k):
    for _ in range(k):
        f = fp12_frob(f)
    return f


# ---------------------------------------------------------------------------- curves
def g1_add(p1, p2):
    if p1 is None: return p2
    if p2 is None: return p1
    x1, y1 = p1; x2, y2 = p2
    if x1 == x2:
        if (y1 + y2) % P == 0: return None
        lam = 3 * x1 * x1 * pow(2 * y1, P - 2, P) % P
    else:
        lam = (y2 - y1) * pow(x2 - x1, P - 2, P) % P
    x3 = (lam * lam - x1 - x2) % P
    return (x3, (lam * (x1 - x3) - y1) % P)


def g1_mul(k, pt):
    acc = None
    for bit in bin(k)[2:] if k else "":
        acc = g1_add(acc, acc)
        if bit == "1":
            acc = g1_add(acc, pt)
    return acc


def g1_neg(pt): return None if pt is None else (pt[0], (-pt[1]) % P)


def g2_add(p1, p2):
    if p1 is None: return p2
    if p2 is None: return p1
    x1, y1 = p1; x2, y2 = p2
    if x1 == x2:
        if (y1 + y2).is_zero(): return None
        lam = x1.sq() * 3 * (y1 * 2).inv()
    else:
        lam = (y2 - y1) * (x2 - x1).inv()
    x3 = lam.sq() - x1 - x2
    return (x3, lam * (x1 - x3) - y1)


def g2_mul(k, pt):
    acc = None
    for bit in bin(k)[2:] if k else "":
        acc = g2_add(acc, acc)
        if bit == "1":
            acc = g2_add(acc, pt)
    return acc


G1 = (G1X, G1Y)
G2 = (Fp2(*G2X), Fp2(*G2Y))
B2 = XI * 4                              # twist coefficient b' = 4(1 + u)


def g1_on_curve(pt): return pt is None or (pt[1] * pt[1] - pt[0] ** 3 - 4) % P == 0
def g2_on_curve(pt): return pt is None or (pt[1].sq() - pt[0].sq() * pt[0] - B2).is_zero()


# ---------------------------------------------------------------------------- reference pairing (affine)
def miller_affine(p1, q2):
    """f_{|x|,Q}(P) with untwisted lines scaled by w^3 (subfield factors die in the final
    exponentiation): l = (lam*x_T - y_T) + (-lam*x_P) v + y_P v w."""
    if p1 is None or q2 is None:
        return F12_1
    xp, yp = p1
    t = q2
    f = F12_1
    for bit in bin(X_ABS)[3:]:
        lam = t[0].sq() * 3 * (t[1] * 2).inv()
        line = fp12_from_014(lam * t[0] - t[1], -(lam * xp), Fp2(yp, 0))
        f = f.sq() * line
        t = g2_add(t, t)
        if bit == "1":
            lam = (t[1] - q2[1]) * (t[0] - q2[0]).inv()
            line = fp12_from_014(lam * t[0] - t[1], -(lam * xp), Fp2(yp, 0))
            f = f * line
            t = g2_add(t, q2)
    return f.conj()                       # x < 0


def final_exp_reference(f):
    return fp12_pow(f, (P ** 12 - 1) // R)


def pairing_reference(p1, q2):
    return final_exp_reference(miller_affine(p1, q2))


# ---------------------------------------------------------------------------- device-shaped formulation
def g2_double_step(T):
    """Homogeneous projective doubling on the twist; returns (T2, (c0, c1', c4')) where the line is
    c0 + (c1' * x_P) v + (c4' * y_P) v w  (scaled by an Fp2 factor)."""
    X, Y, Z = T
    inv2 = pow(2, P - 2, P)
    a = X * Y * inv2
    b = Y.sq()
    c = Z.sq()
    e = B2 * (c * 3)
    f = e * 3
    g = (b + f) * inv2
    h = (Y + Z).sq() - (b + c)            # 2YZ
    j = X.sq()
    X3 = a * (b - f)
    Y3 = g.sq() - e.sq() * 3
    Z3 = b * h
    return (X3, Y3, Z3), (b - e, -(j * 3), h)     # (Y^2 - 3b'Z^2, -3X^2, 2YZ)


def g2_add_step(T, Q):
    X, Y, Z = T
    xq, yq = Q
    theta = Y - yq * Z
    lam = X - xq * Z
    c = theta.sq()
    d = lam.sq()
    e = lam * d
    f = Z * c
    g = X * d
    h = e + f - g * 2
    X3 = lam * h
    Y3 = theta * (g - h) - e * Y
    Z3 = Z * e
    return (X3, Y3, Z3), (theta * xq - lam * yq, -theta, lam)


def miller_projective(pairs):
    """prod_i f_{|x|,Q_i}(P_i), conjugated; pairs with a point at infinity contribute 1."""
    pairs = [(p, q) for p, q in pairs if p is not None and q is not None]
    f = F12_1
    Ts = [(q[0], q[1], F2_1) for _, q in pairs]
    for bit in bin(X_ABS)[3:]:
        f = f.sq()
        for i, (p, q) in enumerate(pairs):
            Ts[i], (c0, c1, c4) = g2_double_step(Ts[i])
            f = f * fp12_from_014(c0, c1 * p[0], c4 * p[1])
        if bit == "1":
            for i, (p, q) in enumerate(pairs):
                Ts[i], (c0, c1, c4) = g2_add_step(Ts[i], q)
                f = f * fp12_from_014(c0, c1 * p[0], c4 * p[1])
    return f.conj()


def exp_by_x(f):
    """f^x for x = -X_ABS (f in the cyclotomic subgroup: inverse = conjugate)."""
    return fp12_pow(f, X_ABS).conj()


def final_exp_chain(f):
    """f^(3 (p^12-1)/r): easy part, then 3*lambda = (x-1)^2 (x+p) (x^2+p^2-1) + 3."""
    f1 = f.conj() * f.inv()                        # ^(p^6 - 1)
    f2 = fp12_frob_k(f1, 2) * f1                   # ^(p^2 + 1)
    y0 = exp_by_x(f2) * f2.conj()                  # ^(x - 1)
    y1 = exp_by_x(y0) * y0.conj()                  # ^(x - 1)^2
    y2 = exp_by_x(y1) * fp12_frob(y1)              # ^(x + p)
    y3 = exp_by_x(exp_by_x(y2)) * fp12_frob_k(y2, 2) * y2.conj()   # ^(x^2 + p^2 - 1)
    return y3 * f2.sq() * f2


def pairing_check(pairs) -> bool:
    """prod e(P_i, Q_i) == 1 (what the device kernel returns)."""
    return final_exp_chain(miller_projective(pairs)) == F12_1


def selfcheck():
    assert (P ** 4 - P ** 2 + 1) % R == 0
    lam = (P ** 4 - P ** 2 + 1) // R
    x = -X_ABS
    assert 3 * lam == (x - 1) ** 2 * (x + P) * (x * x + P * P - 1) + 3
    assert g1_on_curve(G1) and g2_on_curve(G2)
    assert g1_mul(R, G1) is None and g2_mul(R, G2) is None


if __name__ == "__main__":
    import time
    selfcheck()
    t = time.time()
    e = pairing_reference(G1, G2)
    print("ref pairing %.2fs" % (time.time() - t))
    assert e != F12_1 and fp12_pow(e, R) == F12_1
    a, b = 0x1234567, 0x89ABCDE
    assert pairing_reference(g1_mul(a, G1), g2_mul(b, G2)) == fp12_pow(e, a * b)
    t = time.time()
    e3 = final_exp_chain(miller_projective([(G1, G2)]))
    print("chain pairing %.2fs" % (time.time() - t))
    assert e3 == e * e * e
    assert pairing_check([(g1_mul(a, G1), G2), (g1_neg(G1), g2_mul(a, G2))])
    assert not pairing_check([(g1_mul(a, G1), G2), (g1_neg(G1), g2_mul(a + 1, G2))])
    print("bls oracle ok")
