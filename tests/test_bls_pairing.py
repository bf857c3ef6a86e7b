"""BLS12-381 pairing check: the Python oracle's own algebraic pins,
the device source compiled for the host against it, and (gpu) the kernel through the C ABI."""
import ctypes
import os
import random
import subprocess

import numpy as np
import pytest

from oracle import bls_oracle as b

P = b.P
HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim")


def w(x):
    return int(x).to_bytes(48, "little")


def enc_g1(p):
    return bytes(96) if p is None else w(p[0]) + w(p[1])


def enc_g2(q):
    return bytes(192) if q is None else w(q[0].a) + w(q[0].b) + w(q[1].a) + w(q[1].b)


def kzg_like_items(n, seed=1):
    """n valid items (a1 G1, b1 G2), (a2 G1, b2 G2) with a1 b1 + a2 b2 = 0 mod r, then corrupted copies."""
    rnd = random.Random(seed)
    items = []
    for _ in range(n):
        a1, b1, b2 = (rnd.randrange(1, b.R) for _ in range(3))
        a2 = (-a1 * b1 * pow(b2, -1, b.R)) % b.R
        items.append(((b.g1_mul(a1, b.G1), b.g2_mul(b1, b.G2)), (b.g1_mul(a2, b.G1), b.g2_mul(b2, b.G2))))
    return items


def pack(items):
    g1 = b"".join(enc_g1(p0) + enc_g1(p1) for (p0, _), (p1, _) in items)
    g2 = b"".join(enc_g2(q0) + enc_g2(q1) for (_, q0), (_, q1) in items)
    return np.frombuffer(g1, np.uint8).reshape(-1, 192), np.frombuffer(g2, np.uint8).reshape(-1, 384)


def test_oracle_algebra():
    b.selfcheck()
    e = b.pairing_reference(b.G1, b.G2)
    assert e != b.F12_1 and b.fp12_pow(e, b.R) == b.F12_1            # non-degenerate, order r
    a, c = 0x1234567, 0x89ABCDE
    assert b.pairing_reference(b.g1_mul(a, b.G1), b.g2_mul(c, b.G2)) == b.fp12_pow(e, a * c)   # bilinear
    e3 = b.final_exp_chain(b.miller_projective([(b.G1, b.G2)]))
    assert e3 == e * e * e                                              # projective + x-chain == reference^3
    items = kzg_like_items(2)
    for it in items:
        assert b.pairing_check(list(it))
        (p0, q0), (p1, q1) = it
        assert not b.pairing_check([(p0, q0), (b.g1_add(p1, b.G1), q1)])


@pytest.fixture(scope="module")
def hb():
    so = os.path.join(HERE, "libhostsim_bls.so")
    subprocess.run(["make", "-C", HERE, "libhostsim_bls.so"], check=True, stdout=subprocess.DEVNULL)
    return ctypes.CDLL(so)


def test_g1_group_law_host_build(hb):
    """g1.cuh (complete projective formulas, device source compiled for the host) against the oracle's affine law:
    addition, mixed addition of +/- Q, doubling, and every exceptional case the complete formulas must absorb."""
    rnd = random.Random(31)
    G = b.G1
    pts = [b.g1_mul(rnd.randrange(1, b.R), G) for _ in range(12)]
    cases = [(pts[i], pts[(i + 1) % 12]) for i in range(12)]
    cases += [(pts[0], pts[0]), (pts[1], b.g1_neg(pts[1])), (None, pts[2]), (pts[3], None), (None, None)]
    out = ctypes.create_string_buffer(96)
    for P1, P2 in cases:
        hb.hb_g1_op(0, enc_g1(P1), enc_g1(P2), out)
        assert out.raw == enc_g1(b.g1_add(P1, P2)), ("add", P1 is None, P2 is None)
        if P2 is not None:
            hb.hb_g1_op(1, enc_g1(P1), enc_g1(P2), out)
            assert out.raw == enc_g1(b.g1_add(P1, P2)), "madd"
            hb.hb_g1_op(2, enc_g1(P1), enc_g1(P2), out)
            assert out.raw == enc_g1(b.g1_add(P1, b.g1_neg(P2))), "msub"
        hb.hb_g1_op(3, enc_g1(P1), enc_g1(P1), out)
        assert out.raw == enc_g1(b.g1_add(P1, P1)), "dbl"
    # long chains keep the lazy limb bounds honest: 200 mixed additions (repeats included) then 64 doublings
    seq = [pts[rnd.randrange(12)] for _ in range(200)]
    signs = bytes(rnd.getrandbits(1) for _ in range(200))
    hb.hb_g1_chain(200, b"".join(enc_g1(P) for P in seq), signs, 64, out)
    acc = None
    for P, sg in zip(seq, signs):
        acc = b.g1_add(acc, b.g1_neg(P) if sg else P)
    assert out.raw == enc_g1(b.g1_mul(1 << 64, acc) if acc is not None else None)


def _call(f, *a, n=48):
    r = ctypes.create_string_buffer(n)
    f(*a, r)
    return r.raw


def _enc12(f):
    return b"".join(w(c) for c in f.coeffs())


def _dec12(raw):
    c = [int.from_bytes(raw[48 * i:48 * i + 48], "little") for i in range(12)]
    f2 = [b.Fp2(c[2 * i], c[2 * i + 1]) for i in range(6)]
    return b.Fp12(b.Fp6(*f2[:3]), b.Fp6(*f2[3:]))


def test_hostsim_field_and_tower(hb):
    rnd = random.Random(1)
    edge = [0, 1, 2, P - 1, P - 2, (P - 1) // 2]
    for it in range(1500):
        x = rnd.choice(edge) if it % 7 == 0 else rnd.randrange(P)
        y = rnd.choice(edge) if it % 11 == 0 else rnd.randrange(P)
        assert int.from_bytes(_call(hb.hb_fp_mul, w(x), w(y)), "little") == x * y % P
        assert int.from_bytes(_call(hb.hb_fp_sqr, w(x)), "little") == x * x % P
        t, ww = ((x - y) * (x + y) - 3 * x * y) % P, (y - 2 * x) % P
        assert int.from_bytes(_call(hb.hb_fp_lazy, w(x), w(y)), "little") == t * ww % P      # signed lazy limbs
    # inversion by positive divsteps with cofactors (bls12.cuh fp_inv): random values, edge values, powers of two, small
    # values, values near p, and 0 -> 0 as the power a^(p-2) gives
    inv_cases = [0, 1, 2, 3, P - 1, P - 2, (P - 1) // 2, (P + 1) // 2, (1 << 380), (1 << 381) - 1 - (1 << 200)] + \
        [1 << k for k in range(0, 381, 19)] + [P - (1 << k) for k in range(0, 380, 37)] + [rnd.randrange(1, P) for _ in range(1500)] + \
        [rnd.randrange(1, 1 << 64) for _ in range(40)]
    for x in inv_cases:
        x %= P
        assert int.from_bytes(_call(hb.hb_fp_inv, w(x)), "little") == pow(x, P - 2, P), hex(x)
    rf2 = lambda: b.Fp2(rnd.randrange(P), rnd.randrange(P))
    rf12 = lambda: b.Fp12(b.Fp6(rf2(), rf2(), rf2()), b.Fp6(rf2(), rf2(), rf2()))
    for _ in range(10):
        x, y = rf12(), rf12()
        assert _dec12(_call(hb.hb_fp12_mul, _enc12(x), _enc12(y), n=576)) == x * y
        assert _dec12(_call(hb.hb_fp12_sqr, _enc12(x), n=576)) == x.sq()
        assert _dec12(_call(hb.hb_fp12_frob, _enc12(x), n=576)) == b.fp12_frob(x)
        c0, c1, c4 = rf2(), rf2(), rf2()
        cc = b"".join(w(v) for v in (c0.a, c0.b, c1.a, c1.b, c4.a, c4.b))
        assert _dec12(_call(hb.hb_fp12_mul014, _enc12(x), cc, n=576)) == x * b.fp12_from_014(c0, c1, c4)
    x = rf12()
    assert _dec12(_call(hb.hb_fp12_inv, _enc12(x), n=576)) == x.inv()


def test_hostsim_pairing_values_and_checks(hb):
    ml, fe = ctypes.create_string_buffer(576), ctypes.create_string_buffer(576)
    a = 0x1234567
    P1, Q2 = b.g1_mul(a, b.G1), b.g2_mul(a + 5, b.G2)
    assert hb.hb_pairing(enc_g1(P1), enc_g2(Q2), ml, fe) == 1
    f = b.miller_projective([(P1, Q2)])
    assert _dec12(ml.raw) == f and _dec12(fe.raw) == b.final_exp_chain(f)
    items = kzg_like_items(2, seed=5)
    for (p0, q0), (p1, q1) in items:
        assert hb.hb_pairing_check2(enc_g1(p0) + enc_g1(p1), enc_g2(q0) + enc_g2(q1)) == 0
        assert hb.hb_pairing_check2(enc_g1(p0) + enc_g1(b.g1_add(p1, b.G1)), enc_g2(q0) + enc_g2(q1)) == 1
        bad = bytearray(enc_g1(p0)); bad[3] ^= 1
        assert hb.hb_pairing_check2(bytes(bad) + enc_g1(p1), enc_g2(q0) + enc_g2(q1)) == 2
        assert hb.hb_pairing_check2(enc_g1(p0) + enc_g1(p1), enc_g2(q0) + w(P) + bytes(144)) == 2   # coordinate >= p
    assert hb.hb_pairing_check2(bytes(192), enc_g2(b.G2) + enc_g2(b.G2)) == 0          # infinity pairs with anything


def test_hostsim_pairing_prepared_lines(hb):
    """The shared-G2 route (lines of pairing_prepare_g2_pair, scaled per item) gives the verdicts of the plain
    check: valid, wrong, invalid G1 / G2, points at infinity on either side."""
    hb.hb_pairing_check2_prepared.restype = ctypes.c_uint32
    cases = []
    for (p0, q0), (p1, q1) in kzg_like_items(2, seed=6):
        g2 = enc_g2(q0) + enc_g2(q1)
        cases += [(enc_g1(p0) + enc_g1(p1), g2), (enc_g1(p0) + enc_g1(b.g1_add(p1, b.G1)), g2),
                  (enc_g1(p0) + bytes(96), g2), (bytes(192), g2), (enc_g1(p0) + enc_g1(p1), enc_g2(q0) + bytes(192)),
                  (enc_g1(p0) + enc_g1(p1), bytes(384))]
        bad1 = bytearray(enc_g1(p0) + enc_g1(p1)); bad1[100] ^= 1
        bad2 = bytearray(g2); bad2[200] ^= 1
        cases += [(bytes(bad1), g2), (enc_g1(p0) + enc_g1(p1), bytes(bad2))]
    seen = set()
    for g1, g2 in cases:
        want = hb.hb_pairing_check2(g1, g2)
        assert hb.hb_pairing_check2_prepared(g1, g2) == want
        seen.add(want)
    assert seen == {0, 1, 2}


@pytest.mark.gpu
def test_gpu_pairing_check_matches_oracle(ctx):
    items = kzg_like_items(6, seed=9)
    g1, g2 = pack(items)
    g1 = g1.copy(); g2 = g2.copy()
    want = [0] * 6
    # corrupt some items in oracle-checkable ways
    (p0, q0), (p1, q1) = items[1]
    g1[1] = np.frombuffer(enc_g1(p0) + enc_g1(b.g1_add(p1, b.G1)), np.uint8); want[1] = 1
    g1[3, 5] ^= 1; want[3] = 2                                         # off the curve
    g2[4, :48] = np.frombuffer(w(P), np.uint8); want[4] = 2            # coordinate >= p
    g1[5] = 0                                                           # both G1 points at infinity: product is one
    st = ctx.pairing_check_batch(g1, g2)
    assert list(st) == want
    for i in (0, 2):
        assert b.pairing_check(list(items[i]))
    # shared G2 pair (the SRS case): e(a G1, Q0) e(-(a c) G1, Q1) with Q1 = c^-1 ... built from one scalar relation
    c = 0xABCDEF12345
    Q0, Q1 = b.G2, b.g2_mul(c, b.G2)
    rows = []
    for a in (3, 5, 7, 11):
        rows.append(enc_g1(b.g1_mul(a * c % b.R, b.G1)) + enc_g1(b.g1_neg(b.g1_mul(a, b.G1))))
    rows.append(enc_g1(b.g1_mul(13, b.G1)) + enc_g1(b.g1_neg(b.g1_mul(13, b.G1))))       # wrong relation
    sh = np.frombuffer(enc_g2(Q0) + enc_g2(Q1), np.uint8)
    st = ctx.pairing_check_batch(np.frombuffer(b"".join(rows), np.uint8).reshape(-1, 192), sh, g2_shared=True)
    assert list(st) == [0, 0, 0, 0, 1]


@pytest.mark.gpu
def test_gpu_pairing_shared_g2_prepared_lines(ctx):
    """Shared G2 points (g2_shared=True) go through lines prepared once; the verdicts must equal the per-item
    path fed the same G2 points in every row, on valid, wrong, off-curve and infinity items, and with shared
    G2 points that are invalid or at infinity."""
    c = 0x1234567FEDCBA987
    Q0, Q1 = b.g2_mul(7, b.G2), b.g2_mul(7 * c % b.R, b.G2)
    rows, want = [], []
    for a in range(1, 41):
        good = a % 5 != 0
        p0 = b.g1_mul(a * c % b.R, b.G1)
        p1 = b.g1_neg(b.g1_mul(a if good else a + 1, b.G1))
        rows.append(enc_g1(p0) + enc_g1(p1)); want.append(0 if good else 1)
    g1 = np.frombuffer(b"".join(rows), np.uint8).reshape(-1, 192).copy()
    g1[7, 3] ^= 1; want[7] = 2                                  # off the curve
    g1[11] = 0; want[11] = 0                                    # both at infinity
    g1[13, 96:] = 0; want[13] = 1                               # second point at infinity: e(P0, Q0) != 1
    sh = np.frombuffer(enc_g2(Q0) + enc_g2(Q1), np.uint8).copy()
    per_item = np.tile(sh, (g1.shape[0], 1))
    got_shared = ctx.pairing_check_batch(g1, sh, g2_shared=True)
    got_items = ctx.pairing_check_batch(g1, per_item)
    assert list(got_items) == want
    assert list(got_shared) == want
    # a batch that does not fill its last wave / workgroup
    assert list(ctx.pairing_check_batch(g1[:3], sh, g2_shared=True)) == want[:3]
    # shared points: Q1 at infinity (only e(P0, Q0) counts), Q0 off its curve (every item invalid)
    sh_inf = sh.copy(); sh_inf[192:] = 0
    a = ctx.pairing_check_batch(g1, sh_inf, g2_shared=True)
    bb = ctx.pairing_check_batch(g1, np.tile(sh_inf, (g1.shape[0], 1)))
    assert list(a) == list(bb) and a[11] == 0 and a[0] == 1 and a[7] == 2
    sh_bad = sh.copy(); sh_bad[5] ^= 1
    a = ctx.pairing_check_batch(g1, sh_bad, g2_shared=True)
    assert (a == 2).all()
    # the context keeps the lines of the last shared pair: a changed pair replaces them, an unchanged one reuses them
    for _ in range(2):
        assert list(ctx.pairing_check_batch(g1, sh, g2_shared=True)) == want
    # the verdicts also agree with the path that does not prepare lines, and -- small batches run one item per
    # 16-lane row (bls12_row.cuh), large ones one item per quad -- with the quad kernel forced on this small batch
    # and with the row / quad switch crossed by a tiled batch
    for mode in ("noprep", "quad", "oct"):
        ctx.debug_pairing_layout(mode)
        try:
            assert list(ctx.pairing_check_batch(g1, sh, g2_shared=True)) == want, mode
            a = ctx.pairing_check_batch(g1, sh_inf, g2_shared=True)
            assert list(a) == list(bb), mode
        finally:
            ctx.debug_pairing_layout()
    for reps in (25, 26, 102, 103):        # 1000 .. 4120 items, all through the default (8 lanes per item); rounds 1-3 switched layouts here
        big = np.tile(g1, (reps, 1))
        assert list(ctx.pairing_check_batch(big, sh, g2_shared=True)) == want * reps
    for rowmode in ("tri", "row"):                              # both row layouts forced on the small batch
        ctx.debug_pairing_layout(rowmode)
        try:
            assert list(ctx.pairing_check_batch(g1, sh, g2_shared=True)) == want, rowmode
            assert list(ctx.pairing_check_batch(g1, sh_inf, g2_shared=True)) == list(bb), rowmode
        finally:
            ctx.debug_pairing_layout()


@pytest.mark.gpu
def test_gpu_pairing_batch_2_14_tiled(ctx):
    """BASELINE.json config 5 size: 2^14 checks (8 distinct oracle-made items tiled), every 97th corrupted; the soak below
    runs 2^12 DISTINCT items."""
    import torch
    items = kzg_like_items(8, seed=21)
    g1, g2 = pack(items)
    n = 1 << 14
    reps = n // 8
    G1 = np.tile(g1, (reps, 1)).copy(); G2 = np.tile(g2, (reps, 1)).copy()
    bad = np.arange(0, n, 97)
    G1[bad, 96:] = G1[(bad + 1) % n, 96:]                               # swap in a different item's second G1 point
    dev = torch.device("cuda:0")
    d1, d2 = torch.from_numpy(G1).to(dev), torch.from_numpy(G2).to(dev)
    st = torch.full((n,), 9, dtype=torch.uint8, device=dev)
    ctx.pairing_check_batch_dev(d1, d2, st)
    torch.cuda.synchronize()
    got = st.cpu().numpy()
    want = np.zeros(n, np.uint8); want[bad] = 1
    assert (got == want).all()


@pytest.mark.gpu
def test_gpu_quad_tower_ops_match_one_lane_ops(ctx):
    """The pairing kernel spreads one item over a DPP quad (bls12_quad.cuh); its Fp12 operations must equal the
    one-lane operations of bls12.cuh (themselves pinned against the oracle by the hostsim tests) on random
    operands, in every quad of a wave and with edge values (0, 1, p-1)."""
    from ark_ec_vrfs_amd import _lib
    lib = _lib.load()
    rnd = np.random.default_rng(3)
    n = 300
    raw = rnd.integers(0, 256, (n, 2, 12, 48), dtype=np.uint8)
    raw[..., 47] &= 0x0f                                   # < 2^380 < p: canonical inputs
    raw[0] = 0
    raw[1, :, :, :] = 0; raw[1, :, 0, 0] = 1               # x = y = 1
    pm1 = np.frombuffer(int(P - 1).to_bytes(48, "little"), np.uint8)
    raw[2, :, :, :] = pm1
    raw[3, 0] = 0                                          # x = 0
    st = np.full(n, 255, np.uint8)
    _lib.check(lib.vrfhip_test_pairing_quad_ops(ctx.handle, n, raw.ctypes.data, st.ctypes.data), "quad selftest")
    assert (st == 0).all(), {int(i): int(v) for i, v in enumerate(st) if v}


@pytest.mark.gpu
def test_gpu_oct_tower_equals_lane_tower(ctx):
    """The throughput pairing kernel spreads one item over 8 lanes with every Fp2 split over a lane pair
    (bls12_oct.cuh).  Its cross-lane moves (quad_perm, row_shl:4 / row_shr:4 with bank masks, the ballot) and its Fp12
    operations -- product, squaring, cyclotomic squaring, sparse product, Frobenius, inversion, the easy part -- must equal
    the one-lane operations of bls12.cuh on random operands, in every item slot of a wave and with edge values."""
    from ark_ec_vrfs_amd import _lib
    lib = _lib.load()
    rnd = np.random.default_rng(4)
    n = 300
    raw = rnd.integers(0, 256, (n, 2, 12, 48), dtype=np.uint8)
    raw[..., 47] &= 0x0f                                   # < 2^380 < p: canonical inputs
    raw[1, :, :, :] = 0; raw[1, :, 0, 0] = 1               # x = y = 1
    pm1 = np.frombuffer(int(P - 1).to_bytes(48, "little"), np.uint8)
    raw[2, :, :, :] = pm1
    st = np.full(n, 255, np.uint8)
    _lib.check(lib.vrfhip_test_pairing_oct_ops(ctx.handle, n, raw.ctypes.data, st.ctypes.data), "oct selftest")
    assert (st == 0).all(), {int(i): int(v) for i, v in enumerate(st) if v}


@pytest.fixture(scope="module")
def ho():
    so = os.path.join(HERE, "libhostsim_bls_oct.so")
    subprocess.run(["make", "-C", HERE, "libhostsim_bls_oct.so"], check=True, stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(so)
    for f in (lib.hb_oct_moves, lib.hb_oct_selftest, lib.hb_oct_pairing_check2, lib.hb_oct_pairing_check2_prepared,
              lib.hb_oct_pairing_check2_split):
        f.restype = ctypes.c_uint32
    return lib


def test_hostsim_oct_layout_tower_and_checks(hb, ho):
    """bls12_oct.cuh on the CPU: the 8 lanes of an item are 8 threads that meet at a barrier in every cross-lane move.
    Tower operations against the one-lane tower (itself held against the Python oracle above), then whole pairing checks --
    valid, wrong, invalid, infinity on either side -- against pairing_check2_item, with per-item and with prepared lines."""
    assert ho.hb_oct_moves() == 0
    rnd = random.Random(11)
    rf2 = lambda: b.Fp2(rnd.randrange(P), rnd.randrange(P))
    rf12 = lambda: b.Fp12(b.Fp6(rf2(), rf2(), rf2()), b.Fp6(rf2(), rf2(), rf2()))
    edge = b.Fp12(b.Fp6(b.Fp2(P - 1, P - 1), b.Fp2(1, 0), b.Fp2(0, P - 1)), b.Fp6(b.Fp2(P - 2, 1), b.Fp2(0, 0), b.Fp2(P - 1, 0)))
    for it in range(24):
        x, y = (edge, rf12()) if it == 0 else (rf12(), edge) if it == 1 else (rf12(), rf12())
        c0, c1, c4 = rf2(), rf2(), rf2()
        cc = b"".join(w(v) for v in (c0.a, c0.b, c1.a, c1.b, c4.a, c4.b))
        assert ho.hb_oct_selftest(_enc12(x), _enc12(y), cc) == 0, it
    cases = []
    for (p0, q0), (p1, q1) in kzg_like_items(2, seed=8):
        g1, g2 = enc_g1(p0) + enc_g1(p1), enc_g2(q0) + enc_g2(q1)
        bad1 = bytearray(g1); bad1[100] ^= 1
        bad2 = bytearray(g2); bad2[200] ^= 1
        cases += [(g1, g2), (enc_g1(p0) + enc_g1(b.g1_add(p1, b.G1)), g2), (enc_g1(p0) + bytes(96), g2), (bytes(192), g2),
                  (g1, enc_g2(q0) + bytes(192)), (g1, bytes(384)), (bytes(bad1), g2), (g1, bytes(bad2)),
                  (g1, enc_g2(q0) + w(P) + bytes(144))]
    seen = set()
    for g1, g2 in cases:
        want = hb.hb_pairing_check2(g1, g2)
        assert ho.hb_oct_pairing_check2(g1, g2) == want
        assert ho.hb_oct_pairing_check2_split(g1, g2) == want                  # lines kernel + Miller kernel
        assert ho.hb_oct_pairing_check2_prepared(g1, g2) == want
        seen.add(want)
    assert seen == {0, 1, 2}


# ------------------------------------------------------------------------------------------ G1 MSM, batched check
def _g1_pool(k=48, seed=51):
    rnd = random.Random(seed)
    a = [rnd.randrange(1, b.R) for _ in range(k)]
    return a, [b.g1_mul(x, b.G1) for x in a]


@pytest.mark.gpu
def test_gpu_g1_msm_matches_discrete_log_oracle(ctx):
    """`VariableBaseMSM::msm` on G1 (k_msm_g1.hip) against sum_i k_i a_i * G computed from the discrete logs: random
    scalars, equal points in one bucket (doubling through the complete law), P and -P, infinity, zero and extreme
    scalars, sizes that cross the point-group boundaries, and invalid inputs."""
    from ark_ec_vrfs_amd import InvalidData
    rnd = random.Random(52)
    a, pts = _g1_pool()
    encs = [enc_g1(P_) for P_ in pts]
    le32 = lambda v: int(v).to_bytes(32, "little")
    assert ctx.g1_msm(np.zeros((0, 96), np.uint8), np.zeros((0, 32), np.uint8)) == bytes(96)      # empty sum: infinity
    for n in (1, 2, 37, 1000, 8191, 8193, 20000):
        idx = [rnd.randrange(len(pts)) for _ in range(n)]
        ks = [rnd.randrange(b.R) for _ in range(n)]
        if n >= 37:
            ks[0], ks[1], ks[2] = 0, b.R - 1, 1
            idx[3] = idx[4]; ks[3] = ks[4]                              # the same point twice in every window's bucket
            ks[5] = ks[6]; idx[5] = idx[6]
        bases = [encs[i] for i in idx]
        acc = sum(k * a[i] for k, i in zip(ks, idx)) % b.R
        if n >= 37:
            bases[5] = enc_g1(b.g1_neg(pts[idx[5]]))                    # P and -P with one scalar: cancel
            acc = (acc - 2 * ks[5] * a[idx[5]]) % b.R
            bases[7] = bytes(96)                                        # the point at infinity
            acc = (acc - ks[7] * a[idx[7]]) % b.R
        got = ctx.g1_msm(np.frombuffer(b"".join(bases), np.uint8).reshape(-1, 96),
                         np.frombuffer(b"".join(le32(k) for k in ks), np.uint8).reshape(-1, 32))
        assert got == enc_g1(b.g1_mul(acc, b.G1) if acc else None), n
    # every scalar equal (all points of a window in ONE bucket) and a sum that is the identity
    n = 5000
    idx = [i % len(pts) for i in range(n)]
    k = rnd.randrange(1, b.R)
    got = ctx.g1_msm(np.frombuffer(b"".join(encs[i] for i in idx), np.uint8).reshape(-1, 96),
                     np.frombuffer(le32(k) * n, np.uint8).reshape(-1, 32))
    assert got == enc_g1(b.g1_mul(k * sum(a[i] for i in idx) % b.R, b.G1))
    two = np.frombuffer(encs[0] + enc_g1(b.g1_neg(pts[0])), np.uint8).reshape(2, 96)
    assert ctx.g1_msm(two, np.frombuffer(le32(77) * 2, np.uint8).reshape(2, 32)) == bytes(96)
    # invalid inputs
    bad = np.frombuffer(encs[0] + encs[1], np.uint8).reshape(2, 96).copy()
    kk = np.frombuffer(le32(3) + le32(4), np.uint8).reshape(2, 32).copy()
    bad2 = bad.copy(); bad2[1, 7] ^= 1
    with pytest.raises(InvalidData):
        ctx.g1_msm(bad2, kk)                                            # off the curve
    kk2 = kk.copy(); kk2[0] = np.frombuffer(le32(b.R), np.uint8)
    with pytest.raises(InvalidData):
        ctx.g1_msm(bad, kk2)                                            # scalar >= r
    bad3 = bad.copy(); bad3[0, :48] = np.frombuffer(w(P), np.uint8)
    with pytest.raises(InvalidData):
        ctx.g1_msm(bad3, kk)                                            # coordinate >= p


@pytest.mark.gpu
def test_gpu_g1_msm_2_17_discrete_log(ctx):
    rnd = np.random.default_rng(53)
    a, pts = _g1_pool()
    encs = np.stack([np.frombuffer(enc_g1(P_), np.uint8) for P_ in pts])
    n = 1 << 17
    idx = rnd.integers(0, len(pts), n)
    ks = rnd.integers(0, 256, (n, 32), dtype=np.uint8)
    ks[:, 31] &= 0x3f                                                   # < 2^254 < r
    acc = 0
    for i in range(n):
        acc += int.from_bytes(ks[i].tobytes(), "little") * a[idx[i]]
    got = ctx.g1_msm(encs[idx], ks)
    assert got == enc_g1(b.g1_mul(acc % b.R, b.G1))


def _shared_fixture():
    import json
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pairing_items.json")))
    hx = lambda h: np.frombuffer(bytes.fromhex(h), np.uint8)
    return hx(fx["shared_g2"]), np.stack([hx(h) for h in fx["shared"]]), np.stack([hx(h) for h in fx["shared_bad"]])


@pytest.mark.gpu
def test_gpu_batched_pairing_check_statuses_equal_per_item_path(ctx):
    """vrfhip_pairing_check_batch_rlc (two G1 MSMs + ONE pairing) against the per-item kernel and the oracle: all-valid
    batches pass on the fast path; a false item makes the batch fail and is named by the fallback; InvalidData items
    are left out; infinity points; sizes 1 .. 3000; the device form's verdict byte."""
    import torch
    sh, good, bad = _shared_fixture()
    dec1 = lambda raw: [int.from_bytes(bytes(raw[48 * i:48 * i + 48]), "little") for i in range(4)]
    dec2 = lambda raw: [int.from_bytes(bytes(raw[48 * i:48 * i + 48]), "little") for i in range(8)]
    q = dec2(sh)
    Q0, Q1 = (b.Fp2(q[0], q[1]), b.Fp2(q[2], q[3])), (b.Fp2(q[4], q[5]), b.Fp2(q[6], q[7]))
    for row, want in ((good[0], True), (bad[0], False)):                 # the fixture means what it says (oracle)
        v = dec1(row)
        assert b.pairing_check([((v[0], v[1]), Q0), ((v[2], v[3]), Q1)]) == want
    for n in (1, 2, 9, 300, 3000):
        g1 = np.tile(good, (n // 8 + 1, 1))[:n].copy()
        st, ok = ctx.pairing_check_batch_rlc(g1, sh)
        assert ok and (st == 0).all(), n
        assert (ctx.pairing_check_batch(g1, sh, g2_shared=True) == 0).all()
    n = 1000
    g1 = np.tile(good, (n // 8, 1)).copy()
    g1[17] = bad[0]; g1[500] = bad[1]                                     # false items
    g1[33, 5] ^= 1                                                        # off the curve: InvalidData
    g1[44] = 0                                                            # both points at infinity: trivially true
    g1[55, 96:] = 0                                                       # B at infinity: e(A, Q0) != 1
    want = ctx.pairing_check_batch(g1, sh, g2_shared=True)
    assert want[17] == 1 and want[500] == 1 and want[33] == 2 and want[44] == 0 and want[55] == 1 and want.sum() == 5
    st, ok = ctx.pairing_check_batch_rlc(g1, sh)
    assert not ok and (st == want).all()
    g2 = g1.copy(); g2[17] = good[1]; g2[500] = good[2]; g2[55] = good[3]  # only the InvalidData item left
    st, ok = ctx.pairing_check_batch_rlc(g2, sh)
    assert ok and st[33] == 2 and st.sum() == 2
    # device form: statuses {0, 2} and the verdict byte; a different seed gives the same verdicts
    dev = torch.device("cuda:0")
    d1, dsh = torch.from_numpy(g1).to(dev), torch.from_numpy(sh.copy()).to(dev)
    dst = torch.full((n,), 9, dtype=torch.uint8, device=dev)
    dv = torch.full((1,), 9, dtype=torch.uint8, device=dev)
    for seed in (bytes(32), os.urandom(32)):
        ctx.pairing_check_batch_rlc_dev(d1, dsh, dst, dv, seed)
        torch.cuda.synchronize()
        assert int(dv[0]) == 1 and int(dst[33]) == 2 and int(dst.sum()) == 2
        ctx.pairing_check_batch_rlc_dev(torch.from_numpy(g2).to(dev), dsh, dst, dv, seed)
        torch.cuda.synchronize()
        assert int(dv[0]) == 0
    # an invalid shared pair: every item InvalidData, as on the per-item path
    sh_bad = sh.copy(); sh_bad[5] ^= 1
    st, ok = ctx.pairing_check_batch_rlc(g2, sh_bad)
    assert not ok and (st == 2).all()


@pytest.mark.gpu
def test_gpu_batched_pairing_check_2_14_and_2_18(ctx):
    import torch
    sh, good, bad = _shared_fixture()
    dev = torch.device("cuda:0")
    dsh = torch.from_numpy(sh.copy()).to(dev)
    for lg in (14, 18):
        n = 1 << lg
        g1 = np.tile(good, (n // 8, 1)).copy()
        d1 = torch.from_numpy(g1).to(dev)
        dst = torch.full((n,), 9, dtype=torch.uint8, device=dev)
        dv = torch.full((1,), 9, dtype=torch.uint8, device=dev)
        ctx.pairing_check_batch_rlc_dev(d1, dsh, dst, dv, os.urandom(32))
        torch.cuda.synchronize()
        assert int(dv[0]) == 0 and int(dst.sum()) == 0
        d1[n // 3] = torch.from_numpy(bad[0]).to(dev)                      # one false item anywhere fails the batch
        ctx.pairing_check_batch_rlc_dev(d1, dsh, dst, dv, os.urandom(32))
        torch.cuda.synchronize()
        assert int(dv[0]) == 1 and int(dst.sum()) == 0



# ---------------------------------------------------------------------------------------------- native oracle + soak
def test_c_pairing_oracle_equals_the_python_oracle():
    """oracle/c/oracle_bls.c (the native twin used for the soak below and as bench.py's cpu_baseline) against
    oracle/bls_oracle.py: scalar multiplication on G1 / G2, the committed fixture, and constructed items of every kind."""
    import json
    from oracle import c_oracle as co
    rnd = random.Random(4)
    G1b, G2b = enc_g1(b.G1), enc_g2(b.G2)
    for k in (0, 1, 2, b.R - 1, b.R, rnd.randrange(b.R), rnd.randrange(b.R)):
        assert co.g1_mul(k, G1b) == enc_g1(b.g1_mul(k % b.R, b.G1)) and co.g2_mul(k, G2b) == enc_g2(b.g2_mul(k % b.R, b.G2))
    items = kzg_like_items(4, seed=33)
    g1, g2 = pack(items)
    g1, g2 = g1.copy(), g2.copy()
    (p0, q0), (p1, q1) = items[1]
    g1[1] = np.frombuffer(enc_g1(p0) + enc_g1(b.g1_add(p1, b.G1)), np.uint8)      # wrong relation
    g1[2, 7] ^= 1                                                                  # off the curve
    g2[3, 48:96] = np.frombuffer(w(P), np.uint8)                                   # coordinate >= p
    st = co.pairing_check_batch(g1, g2, threads=2)
    assert list(st) == [0, 1, 2, 2]
    assert b.pairing_check(list(items[0])) and not b.pairing_check([(p0, q0), (b.g1_add(p1, b.G1), q1)])
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pairing_items.json")))
    hx = lambda h: np.frombuffer(bytes.fromhex(h), np.uint8)
    f1 = np.stack([hx(it["g1"]) for it in fx["per_item"]]); f2 = np.stack([hx(it["g2"]) for it in fx["per_item"]])
    assert not co.pairing_check_batch(f1, f2).any()
    sh = np.stack([hx(h) for h in fx["shared"]])
    assert not co.pairing_check_batch(sh, hx(fx["shared_g2"]), shared=True).any()
    if fx.get("shared_bad"):
        assert co.pairing_check_batch(np.stack([hx(h) for h in fx["shared_bad"]]), hx(fx["shared_g2"]), shared=True).all()
    # infinity conventions: both G1 points at infinity -> product one; one of them -> the other pairing alone
    z = g1[:1].copy(); z[:] = 0
    assert co.pairing_check_batch(z, g2[:1])[0] == 0
    z = g1[:1].copy(); z[0, 96:] = 0
    assert co.pairing_check_batch(z, g2[:1])[0] == 1


TAMPER = ("none", "wrong_scalar", "swapped", "shifted", "off_curve_g1", "coord_ge_p", "inf_p0", "inf_both", "inf_q1",
          "off_curve_g2")


def _soak_items(n, shared_pair=None, seed=5):
    """n DISTINCT items built from scalar relations with the native oracle's g1_mul / g2_mul, every third one tampered
    with (kind cycling through TAMPER).  shared_pair = (Q0, Q1 = tau Q0) bytes and tau: the KZG shape A = -tau b G1,
    B = b G1 against one pair; else a fresh pair of G2 points per item.  Returns g1 (n, 192), g2 (n, 384) or (384,),
    and the status each item must get -- known by construction, and confirmed by the native oracle in the test."""
    from oracle import c_oracle as co
    rnd = random.Random(seed)
    G1b, G2b = enc_g1(b.G1), enc_g2(b.G2)
    g1 = np.empty((n, 192), np.uint8)
    g2 = None if shared_pair else np.empty((n, 384), np.uint8)
    want = np.zeros(n, np.uint8)
    kinds = []
    for i in range(n):
        kind = TAMPER[(i // 3) % len(TAMPER)] if i % 3 == 0 else "none"
        if shared_pair:
            (pair, tau) = shared_pair
            bsc = rnd.randrange(1, b.R)
            a_sc = (-tau * bsc) % b.R
            if kind == "wrong_scalar":
                a_sc = (a_sc + 1 + rnd.randrange(1 << 64)) % b.R
            p0, p1 = co.g1_mul(a_sc, G1b), co.g1_mul(bsc, G1b)
            q = pair
        else:
            a1, b1, b2 = (rnd.randrange(1, b.R) for _ in range(3))
            a2 = (-a1 * b1 * pow(b2, -1, b.R)) % b.R
            if kind == "wrong_scalar":
                a2 = (a2 + 1 + rnd.randrange(1 << 64)) % b.R
            p0, p1 = co.g1_mul(a1, G1b), co.g1_mul(a2, G1b)
            q = co.g2_mul(b1, G2b) + co.g2_mul(b2, G2b)
        row, st = bytearray(p0 + p1), 0
        qq = bytearray(q)
        if kind == "wrong_scalar":
            st = 1
        elif kind == "swapped":
            row = bytearray(p1 + p0); st = 1                       # (P1, Q0), (P0, Q1): another relation (distinct scalars)
        elif kind == "shifted":
            row[96:] = co.g1_add(bytes(row[96:]), G1b); st = 1     # P1 + G1
        elif kind == "off_curve_g1":
            row[48 + rnd.randrange(40)] ^= 1 << rnd.randrange(8); st = 2
        elif kind == "coord_ge_p":
            row[96:144] = w(P + rnd.randrange(1000)); st = 2
        elif kind == "inf_p0":
            row[:96] = bytes(96); st = 1                           # only e(P1, Q1) is left: not one
        elif kind == "inf_both":
            row[:] = bytes(192); st = 0                            # empty product
        elif kind == "inf_q1" and not shared_pair:
            qq[192:] = bytes(192); st = 1                          # only e(P0, Q0) is left
        elif kind == "off_curve_g2" and not shared_pair:
            qq[96 + rnd.randrange(40)] ^= 1 << rnd.randrange(8); st = 2
        g1[i] = np.frombuffer(bytes(row), np.uint8)
        if g2 is not None:
            g2[i] = np.frombuffer(bytes(qq), np.uint8)
        want[i] = st
        kinds.append(kind)
    if shared_pair:
        g2 = np.frombuffer(shared_pair[0], np.uint8).copy()
    return g1, g2, want, kinds


@pytest.mark.gpu
def test_gpu_pairing_soak_distinct_items_every_path(ctx):
    """2^12 DISTINCT items per shape, a third of them tampered with in ten ways (wrong scalar, swapped and shifted points,
    off-curve and out-of-range coordinates on either side, infinity on each side and on both): the verdict of EVERY device
    path equals the native oracle's and the status known by construction.  Paths: one item per lane, per quad (independent
    G2 points); for a shared pair the prepared lines with one item per wave (tri, n <= 1024), per row (n <= 4096), per quad
    (beyond), the unprepared kernel, and the batched check (two G1 MSMs + one pairing, per-item fallback)."""
    from oracle import c_oracle as co
    ncpu = min(8, os.cpu_count() or 1)
    n = 1 << 12
    # independent G2 points per item
    g1, g2, want, kinds = _soak_items(n, seed=6)
    ref = co.pairing_check_batch(g1, g2, threads=ncpu)
    assert (ref == want).all(), [(k, int(a), int(c)) for k, a, c in zip(kinds, ref, want) if a != c][:5]
    assert len({bytes(r) for r in g1}) > n - n // 20 and len({bytes(r) for r in g2}) > n - n // 20     # distinct (but for the zeroed ones)
    assert (ctx.pairing_check_batch(g1, g2) == want).all()               # 2^12 items: 8 lanes per item
    for mode in ("lane", "quad", "oct", "oct1"):
        ctx.debug_pairing_layout(mode)
        try:
            assert (ctx.pairing_check_batch(g1, g2) == want).all(), mode
        finally:
            ctx.debug_pairing_layout()
    assert (ctx.pairing_check_batch(g1[:1500], g2[:1500]) == want[:1500]).all()    # below the switch: one item per quad
    # one shared pair (Q0, tau Q0): the KZG verifier's shape
    tau = 0x1D3A5F7C9B2E4F6A8C0E1B3D5F7A9C0E2B4D6F8A1C3E5A7C9E0B2D4F6A8C1E3
    q0 = co.g2_mul(0xC0FFEE1234567, enc_g2(b.G2))
    pair = q0 + co.g2_mul(tau, q0)
    m = n + 64                                                         # crosses the row / quad switch (4096)
    s1, sg2, swant, skinds = _soak_items(m, shared_pair=(pair, tau), seed=7)
    sref = co.pairing_check_batch(s1, sg2, shared=True, threads=ncpu)
    assert (sref == swant).all(), [(k, int(a), int(c)) for k, a, c in zip(skinds, sref, swant) if a != c][:5]
    assert set(np.unique(swant)) == {0, 1, 2}
    for cnt in (1, 7, 1000, 4096, m):                                  # the default layout (8 lanes per item) at every size
        assert (ctx.pairing_check_batch(s1[:cnt], sg2, g2_shared=True) == swant[:cnt]).all(), cnt
    for mode in ("noprep", "quad", "row", "oct"):
        ctx.debug_pairing_layout(mode)
        try:
            assert (ctx.pairing_check_batch(s1[:1500], sg2, g2_shared=True) == swant[:1500]).all(), mode
        finally:
            ctx.debug_pairing_layout()
    ctx.debug_pairing_layout("noprep", "oct")                            # shared pair through the unprepared 8-lane kernel
    try:
        assert (ctx.pairing_check_batch(s1[:700], sg2, g2_shared=True) == swant[:700]).all()
    finally:
        ctx.debug_pairing_layout()
    assert (ctx.pairing_check_batch(s1, np.tile(sg2, (m, 1))) == swant).all()       # the same items through the per-item kernel
    # batched: a batch with false items fails as a whole and falls back to the per-item verdicts ...
    st, batch_ok = ctx.pairing_check_batch_rlc(s1, sg2, seed=bytes(range(32)))
    assert not batch_ok and (st == swant).all()
    # ... the valid items alone (and the harmless all-infinity ones) pass as ONE pairing; one false item among 2^12 is caught
    good = np.flatnonzero(swant == 0)
    st, batch_ok = ctx.pairing_check_batch_rlc(s1[good], sg2, seed=bytes(range(1, 33)))
    assert batch_ok and not st.any() and len(good) > 2800
    one_bad = s1[good].copy()
    one_bad[1234, 96:] = np.frombuffer(co.g1_add(bytes(one_bad[1234, 96:]), enc_g1(b.G1)), np.uint8)
    st, batch_ok = ctx.pairing_check_batch_rlc(one_bad, sg2, seed=bytes(range(2, 34)))
    assert not batch_ok and st[1234] == 1 and st.sum() == 1
