"""CPU: the device headers (fe.cuh, te.cuh, sha512.cuh, fr.cuh, vrf_core.cuh) compiled for the
host (tests/hostsim) against Python big ints, the oracle and the golden vectors.  This checks
the exact kernel source without a GPU; the GPU parity tests proper are in test_gpu_parity.py."""
import ctypes
import os
import random
import subprocess

import pytest

from oracle import vrf_oracle as o

S = o.BANDERSNATCH
Q = S.q
HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim")


@pytest.fixture(scope="module")
def hs():
    so = os.path.join(HERE, "libhostsim.so")
    subprocess.run(["make", "-C", HERE, "-j4", "libhostsim.so"], check=True, stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(so)
    lib.hs_init()
    return lib


def _b(x):
    return x.to_bytes(32, "little")


def _call(f, *a):
    r = ctypes.create_string_buffer(32)
    ret = f(*[_b(x) for x in a], r)
    return int.from_bytes(r.raw, "little"), ret


EDGE = [0, 1, 2, Q - 1, Q - 2, (1 << 255) - 1, (1 << 256) - 1, Q, Q + 1, (1 << 29) - 1, 1 << 232]


def test_field_ops_against_python_ints(hs):
    rnd = random.Random(1)
    for it in range(2000):
        x = rnd.choice(EDGE) if it % 7 == 0 else rnd.getrandbits(256)
        y = rnd.choice(EDGE) if it % 11 == 0 else rnd.getrandbits(256)
        assert _call(hs.hs_fe_mul, x, y)[0] == x * y % Q
        assert _call(hs.hs_fe_sqr, x)[0] == x * x % Q
        assert _call(hs.hs_fe_add, x, y)[0] == (x + y) % Q
        assert _call(hs.hs_fe_sub, x, y)[0] == (x - y) % Q
        a, b = x % Q, y % Q
        s, d, ab = a + b, a - b, a * b
        u2, w = 5 * ab + s * d - b, a - (ab + b)
        assert _call(hs.hs_fe_lazy, x, y)[0] == (u2 * w) % Q      # lazy-limb bound stress


def test_inverse_and_table_driven_sqrt(hs):
    rnd = random.Random(2)
    # fe_inv: positive divsteps with cofactors (fe.cuh); 0 -> 0, powers of two, values near q, small values, random
    for x in [0, 1, 2, 3, Q - 1, Q - 2, (Q - 1) // 2, (Q + 1) // 2, Q + 5] + [1 << k for k in range(0, 256, 11)] + \
            [Q - (1 << k) for k in range(0, 250, 23)] + [rnd.getrandbits(256) for _ in range(2500)] + [rnd.getrandbits(40) for _ in range(50)]:
        assert _call(hs.hs_fe_inv, x)[0] == pow(x % Q, Q - 2, Q), hex(x)
    for it in range(200):
        x = rnd.choice(EDGE) if it % 7 == 0 else rnd.getrandbits(256)
        assert _call(hs.hs_fe_inv, x)[0] == pow(x % Q, Q - 2, Q)
        r, sq = _call(hs.hs_fe_sqrt, x)
        xm = x % Q
        if xm == 0:
            assert r == 0
        elif o.legendre(xm, Q) == 1:
            assert sq == 1 and r * r % Q == xm
        else:
            assert sq == 0 and r * r % Q == 5 * xm % Q            # sqrt(Z * w), Z = 5


def test_jacobi_symbol_against_euler_criterion(hs):
    """fe.cuh jacobi_limbs (positive divsteps on the Montgomery image) == w^((q-1)/2), never 'rounds exhausted'."""
    rnd = random.Random(20)
    vals = [rnd.getrandbits(256) for _ in range(3000)] + EDGE + [4, 9, 5, 7, Q - 4, Q - 5, 2 * Q, 2 * Q + 4]
    vals += [pow(rnd.getrandbits(255), 2, Q) for _ in range(200)]           # squares
    vals += [5 * pow(rnd.getrandbits(255), 2, Q) % Q for _ in range(200)]   # non-squares (5 is one)
    vals += [1 << k for k in range(0, 256, 7)] + [Q - (1 << k) for k in range(0, 250, 11)]
    for x in vals:
        a = x % Q
        want = 0 if a == 0 else (1 if pow(a, (Q - 1) // 2, Q) == 1 else -1)
        got = hs.hs_fe_jacobi(_b(x))
        assert got == want, (hex(x), got, want)
        assert hs.hs_fe_is_nonzero_square(_b(x)) == (1 if want == 1 else 0)


def test_sha512_all_padding_boundaries(hs):
    rnd = random.Random(3)
    for n in list(range(0, 20)) + [55, 111, 112, 113, 119, 120, 127, 128, 129, 239, 240, 241, 255, 256, 257, 300]:
        m = bytes(rnd.getrandbits(8) for _ in range(n))
        out = ctypes.create_string_buffer(64)
        hs.hs_sha512(m, n, out)
        assert out.raw == o.sha512(m), n


def test_stage_and_scheme_kats(hs, kat):
    buf = lambda n=32: ctypes.create_string_buffer(n)
    for v in kat["stages"]:
        a = bytes.fromhex(v["alpha"]); u0, u1 = buf(), buf()
        hs.hs_h2f(a, len(a), u0, u1)
        assert u0.raw.hex() == v["u0"] and u1.raw.hex() == v["u1"]
    for v in kat["ietf"]:
        a, ad = bytes.fromhex(v["alpha"]), bytes.fromhex(v["ad"])
        h = buf(); hs.hs_hash_to_curve(a, len(a), h); assert h.raw.hex() == v["h"]
        sd = bytes.fromhex(v["seed"]); sk = buf(); hs.hs_secret_from_seed(sd, len(sd), sk); assert sk.raw.hex() == v["sk"]
        pk = buf(); hs.hs_public(bytes.fromhex(v["sk"]), pk); assert pk.raw.hex() == v["pk"]
        b = buf(64); hs.hs_output_hash(bytes.fromhex(v["gamma"]), b); assert b.raw.hex() == v["beta"]
        args = [bytes.fromhex(v[k]) for k in ("pk", "h", "gamma", "c", "s")]
        assert hs.hs_ietf_verify(*args, ad, len(ad)) == 0
        bad = bytearray(args[4]); bad[0] ^= 1
        assert hs.hs_ietf_verify(*args[:4], bytes(bad), ad, len(ad)) == 1
        g, c, s, hh, pk = buf(), buf(), buf(), buf(), buf()
        assert hs.hs_ietf_prove(bytes.fromhex(v["sk"]), a, len(a), None, ad, len(ad), g, c, s, hh, pk) == 1
        assert (g.raw.hex(), c.raw.hex(), s.raw.hex(), hh.raw.hex(), pk.raw.hex()) == \
               (v["gamma"], v["c"], v["s"], v["h"], v["pk"])
        # pre-hashed input path
        assert hs.hs_ietf_prove(bytes.fromhex(v["sk"]), None, 0, bytes.fromhex(v["h"]), ad, len(ad), g, c, s, hh, pk) == 1
        assert (g.raw.hex(), c.raw.hex(), s.raw.hex()) == (v["gamma"], v["c"], v["s"])


def test_decode_and_verify_status_against_oracle(hs):
    rnd = random.Random(5)
    x, y = ctypes.create_string_buffer(32), ctypes.create_string_buffer(32)
    cases = [bytes(32), (Q - 1).to_bytes(32, "little"), Q.to_bytes(32, "little"), b"\xff" * 32]
    cases += [rnd.getrandbits(256).to_bytes(32, "little") for _ in range(40)]
    for enc in cases:
        ok = hs.hs_decode(enc, x, y)
        p = o.point_decode(S, enc)
        assert bool(ok) == (p is not None), enc.hex()
        if p:
            assert (int.from_bytes(x.raw, "little"), int.from_bytes(y.raw, "little")) == p
    # random proofs with long / empty ad, verified and tampered
    for i in range(3):
        sk = o.secret_from_seed(S, o.synth_seed(50 + i))
        H = o.data_to_point(S, o.synth_msg(50 + i))
        ad = bytes(rnd.getrandbits(8) for _ in range([0, 70, 200][i]))
        g, c, s = o.ietf_prove(S, sk, H, ad)
        enc = [o.point_encode(S, o.public_from_secret(S, sk)), o.point_encode(S, H), o.point_encode(S, g),
               o.scalar_encode(c), o.scalar_encode(s)]
        assert hs.hs_ietf_verify(*enc, ad, len(ad)) == 0
        assert hs.hs_ietf_verify(*enc, ad + b"!", len(ad) + 1) == 1
        bad = list(enc); bad[3] = (c + S.r).to_bytes(32, "little")  # `Proof::c` is decoded mod r: the same field element
        assert hs.hs_ietf_verify(*bad, ad, len(ad)) == 0
        bad = list(enc); bad[4] = (s + S.r).to_bytes(32, "little")  # s is strict
        assert hs.hs_ietf_verify(*bad, ad, len(ad)) == 2
        bad = list(enc); bad[0] = Q.to_bytes(32, "little")          # y >= q
        assert hs.hs_ietf_verify(*bad, ad, len(ad)) == 2


def test_generator_table_entries_match_the_ladder(hs):
    """The signed-window generator tables are built by the device's own segment builder (one inversion per
    segment); entries must equal j * 2^(bits w) * G computed by the branch-free ladder.  The host simulation
    uses 8-bit windows (the device 16-bit: same code, GC_* constants differ)."""
    rows, cols = ctypes.c_int(), ctypes.c_int()
    bits = hs.hs_gcomb_geometry(ctypes.byref(rows), ctypes.byref(cols))
    assert (bits, rows.value, cols.value) == (8, 32, 128)
    for (w, j) in [(0, 1), (0, 2), (0, 128), (3, 17), (31, 1), (31, 7), (17, 127)]:
        assert hs.hs_comb_entry_check(w, j) == 1


def test_pedersen_kat_and_oracle(hs, kat):
    v, iv = kat["pedersen"][0], kat["ietf"][0]
    out = ctypes.create_string_buffer(224)
    a, ad = bytes.fromhex(v["alpha"]), bytes.fromhex(v["ad"])
    assert hs.hs_pedersen_prove(bytes.fromhex(iv["sk"]), a, len(a), ad, len(ad), out) == 1
    parts = [out.raw[32 * i:32 * i + 32].hex() for i in range(7)]
    assert parts == [iv["gamma"], v["pk_com"], v["r"], v["ok"], v["s"], v["sb"], v["blinding"]]
    proof = out.raw[32:192]
    h, g = bytes.fromhex(iv["h"]), bytes.fromhex(iv["gamma"])
    assert hs.hs_pedersen_verify(h, g, proof, ad, len(ad)) == 0
    for pos in (5, 40, 70, 100, 130):                    # pk_com, R, Ok, s, sb
        bad = bytearray(proof); bad[pos] ^= 1
        assert hs.hs_pedersen_verify(h, g, bytes(bad), ad, len(ad)) in (1, 2)
    sk = o.secret_from_seed(S, o.synth_seed(3)); msg = o.synth_msg(3); ad = bytes(range(90))
    H = o.data_to_point(S, msg)
    gm, (pc, R, Ok, s, sb), b = o.pedersen_prove(S, sk, H, ad)
    assert hs.hs_pedersen_prove(o.scalar_encode(sk), msg, len(msg), ad, len(ad), out) == 1
    exp = b"".join([o.point_encode(S, gm), o.point_encode(S, pc), o.point_encode(S, R), o.point_encode(S, Ok),
                    o.scalar_encode(s), o.scalar_encode(sb), o.scalar_encode(b)])
    assert out.raw == exp
    assert hs.hs_pedersen_verify(o.point_encode(S, H), o.point_encode(S, gm), out.raw[32:192], ad, len(ad)) == 0


def test_glv_decomposition_and_endomorphism(hs):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "..", "tools"))
    import gen_constants as gc
    lam = gc.glv_constants()["lam"]
    r = S.r
    rnd = random.Random(8)
    out = ctypes.create_string_buffer(34)
    for it in range(5000):
        k = [0, 1, r - 1, r - 2, lam, r - lam, (r - 1) // 2][it] if it < 7 else rnd.randrange(r)
        hs.hs_glv_decompose(k.to_bytes(32, "little"), out)
        k1 = int.from_bytes(out.raw[0:16], "little") * (-1 if out.raw[16] else 1)
        k2 = int.from_bytes(out.raw[17:33], "little") * (-1 if out.raw[33] else 1)
        assert (k1 + k2 * lam - k) % r == 0
        assert abs(k1) < (1 << 127) - (1 << 123) and abs(k2) < (1 << 127) - (1 << 123)   # 32 signed nibbles suffice
    buf = ctypes.create_string_buffer(32)
    G = (S.gx, S.gy)
    for i in (1, 2, 3, 77, 12345):
        P = o.te_mul(S, i, G)
        assert hs.hs_psi(o.point_encode(S, P), buf) == 1
        assert buf.raw == o.point_encode(S, o.te_mul(S, lam, P))
    assert hs.hs_psi(o.point_encode(S, (0, 1)), buf) == 1 and buf.raw == o.point_encode(S, (0, 1))


def test_zero_secret_proof_is_accepted(hs):
    """sk = 0: pk and Gamma are the identity (a subgroup point on which psi's formula degenerates)."""
    H = o.data_to_point(S, b"zero-key")
    g, c, s = o.ietf_prove(S, 0, H, b"")
    assert o.ietf_verify(S, (0, 1), H, g, b"", c, s)
    enc = [o.point_encode(S, (0, 1)), o.point_encode(S, H), o.point_encode(S, g), o.scalar_encode(c), o.scalar_encode(s)]
    assert hs.hs_ietf_verify(*enc, b"", 0) == 0


def test_affine_input_verify(hs, kat):
    def xy(enc):
        Pt = o.point_decode(S, bytes.fromhex(enc))
        return Pt[0].to_bytes(32, "little") + Pt[1].to_bytes(32, "little")
    for v in kat["ietf"]:
        ad = bytes.fromhex(v["ad"])
        c, s = bytes.fromhex(v["c"]), bytes.fromhex(v["s"])
        assert hs.hs_ietf_verify_affine(xy(v["pk"]), xy(v["h"]), xy(v["gamma"]), c, s, ad, len(ad)) == 0
        bad = bytearray(s); bad[0] ^= 1
        assert hs.hs_ietf_verify_affine(xy(v["pk"]), xy(v["h"]), xy(v["gamma"]), c, bytes(bad), ad, len(ad)) == 1
        off = bytearray(xy(v["pk"])); off[0] ^= 1
        assert hs.hs_ietf_verify_affine(bytes(off), xy(v["h"]), xy(v["gamma"]), c, s, ad, len(ad)) == 2


def test_multi_proof_lanes_match_oracle(hs):
    """The K-proofs-per-lane decode/finish (shared inversions) on a ragged batch with bad items."""
    import numpy as np
    from oracle import c_oracle as co
    n = 19                                           # not a multiple of VERIFY_K = 8
    sk = np.stack([np.frombuffer(co.secret_from_seed(o.synth_seed(700 + i)), np.uint8) for i in range(n)])
    msg = np.stack([np.frombuffer(o.synth_msg(700 + i), np.uint8) for i in range(n)])
    r = co.ietf_prove_batch(sk, msgs=msg, ad=b"multi", threads=4)
    a = {k: r[k].copy() for k in ("pk", "input", "output", "c", "s")}
    a["s"][3, 0] ^= 1
    a["pk"][9] = np.frombuffer(Q.to_bytes(32, "little"), np.uint8)            # y >= q in the middle of a lane group
    a["output"][10] = r["output"][11]
    a["c"][17] = np.frombuffer(S.r.to_bytes(32, "little"), np.uint8)         # c = r is the field element 0: wrong, not invalid
    a["s"][18] = np.frombuffer(S.r.to_bytes(32, "little"), np.uint8)         # s is strict: InvalidData
    want = co.ietf_verify_batch(a["pk"], a["input"], a["output"], a["c"], a["s"], b"multi", threads=4)
    st = (ctypes.c_uint8 * n)()
    hs.hs_ietf_verify_multi(n, a["pk"].tobytes(), a["input"].tobytes(), a["output"].tobytes(), a["c"].tobytes(),
                            a["s"].tobytes(), b"multi", 5, st)
    assert list(st) == list(want)
    assert list(want[[3, 9, 10, 17, 18]]) == [1, 2, 1, 1, 2] and want.sum() == 7


def test_multi_proof_prepare_matches_oracle(hs):
    import numpy as np
    from oracle import c_oracle as co
    n = 11
    sk = np.stack([np.frombuffer(co.secret_from_seed(o.synth_seed(900 + i)), np.uint8) for i in range(n)])
    msg = np.stack([np.frombuffer(o.synth_msg(900 + i), np.uint8) for i in range(n)])
    r = co.ietf_prove_batch(sk, msgs=msg, ad=b"pm", threads=4)
    out = ctypes.create_string_buffer(160 * n)
    hs.hs_ietf_prove_multi(n, sk.tobytes(), msg.tobytes(), 32, b"pm", 2, out)
    for i in range(n):
        row = out.raw[160 * i:160 * i + 160]
        assert row == b"".join(r[k][i].tobytes() for k in ("output", "c", "s", "pk", "input")), i


def test_msm_digit_recoding_and_rlc_weights(hs):
    """msm.cuh: signed radix-2^11 digits (folded at r/2) rebuild +-k mod r, stay in [-1024, 1024], leave the
    windows >= 12 empty for 128-bit scalars; the batched-verification weights are the two 16-byte halves of
    SHA-512("vrfhip-rlc-v2" || seed || batch digest || u64_le(i)) with the low three bits forced to 001; the MSM index map puts G and B after the three classes
    with full-size scalars."""
    import hashlib
    rnd = random.Random(11)
    jj_r = o.jubjub_params().r
    for suite, r in ((1, S.r), (2, jj_r)):
        cases = [0, 1, 2, 1024, 1025, 2047, 2048, r - 1, r - 2, (r - 1) // 2, (r + 1) // 2, (1 << 128) - 1, 1 << 127,
                 (1 << 132) - 1] + [rnd.randrange(r) for _ in range(300)] + [rnd.randrange(1 << 128) for _ in range(100)]
        for k in cases:
            for negate in (0, 1):
                d = (ctypes.c_int16 * 23)()
                hs.hs_msm_digits(suite, _b(k), negate, 0, d)
                digs = list(d)
                assert all(-1024 <= x <= 1024 for x in digs)
                val = sum(x << (11 * w) for w, x in enumerate(digs))
                assert val % r == (-k if negate else k) % r
                if k < (1 << 128):
                    assert all(x == 0 for x in digs[12:])
        d = (ctypes.c_int16 * 23)()
        hs.hs_msm_digits(suite, _b(r - 5), 0, 1, d)
        assert not any(d)
    seed, root = bytes(range(100, 132)), bytes(range(7, 39))
    for idx in (0, 1, 255, 1 << 20, (1 << 63) + 12345):
        z, zp = ctypes.create_string_buffer(32), ctypes.create_string_buffer(32)
        hs.hs_rlc_weights(seed, root, ctypes.c_uint64(idx), z, zp)
        dg = hashlib.sha512(b"vrfhip-rlc-v2" + seed + root + idx.to_bytes(8, "little")).digest()
        fix = lambda b: bytes([(b[0] & 0xf8) | 1]) + b[1:]
        assert z.raw == fix(dg[:16]) + bytes(16) and zp.raw == fix(dg[16:32]) + bytes(16)
    hs.hs_rlc_index.restype = ctypes.c_uint64
    n = 1000
    idx = sorted(hs.hs_rlc_index(p, ctypes.c_uint64(n), ctypes.c_uint64(i)) for p in range(5) for i in range(n))
    assert idx == list(range(3 * n)) + list(range(3 * n + 2, 5 * n + 2))          # 3n, 3n+1 are G and B


def py_batch_digest(arrays, ads, index0=0):
    """digest.cuh / vrfhip_test_batch_digest restated with hashlib (ads: one byte string per item)."""
    import hashlib
    n = len(arrays[0])
    level = [hashlib.sha512(b"vrfhip-leaf-v1" + (index0 + i).to_bytes(8, "little") + b"".join(bytes(a[i]) for a in arrays)
                            + ads[i] + len(ads[i]).to_bytes(4, "little")).digest()[:32] for i in range(n)]
    while True:
        level = [hashlib.sha512(b"vrfhip-node-v1" + len(level[t:t + 16]).to_bytes(4, "little")
                                + b"".join(level[t:t + 16])).digest()[:32] for t in range(0, len(level), 16)]
        if len(level) == 1:
            return level[0]


def test_batch_digest_header_and_oracle_equal_the_definition(hs):
    """digest.cuh (host build) and oracle_batch_digest against the hashlib restatement: 1, 15, 16, 17, 257 and 16500 items
    (one to four node levels), several arrays of different widths, shared / per-item / no ad."""
    import numpy as np
    from oracle import c_oracle as co
    rng = np.random.default_rng(5)
    hs.hs_batch_digest.restype = None
    for n, widths, mode in ((1, (32,), "none"), (15, (32, 64), "shared"), (16, (192,), "per"), (17, (32, 32, 32, 32, 32, 32, 32), "per"),
                            (257, (36,), "per"), (16500, (32, 32), "shared")):
        arrays = [rng.integers(0, 256, (n, w), dtype=np.uint8) for w in widths]
        if mode == "per":
            ads = [bytes(rng.integers(0, 256, int(rng.integers(0, 40)), dtype=np.uint8)) for _ in range(n)]
            ad_arg = ads
        elif mode == "shared":
            ads, ad_arg = [b"shared ad"] * n, b"shared ad"
        else:
            ads, ad_arg = [b""] * n, None
        want = py_batch_digest(arrays, ads, index0=77)
        assert co.batch_digest(arrays, ad_arg, index0=77) == want, (n, mode)
        ptrs = (ctypes.c_void_p * len(arrays))(*[a.ctypes.data for a in arrays])
        w = np.array(widths, np.uint32)
        root = ctypes.create_string_buffer(32)
        if mode == "per":
            off = np.concatenate([[0], np.cumsum([len(a) for a in ads])]).astype(np.uint32)
            blob = b"".join(ads) + b"\0"
            hs.hs_batch_digest(ctypes.c_uint64(n), ctypes.c_uint64(77), len(arrays), ptrs, w.ctypes.data_as(ctypes.c_void_p), blob,
                               off.ctypes.data_as(ctypes.c_void_p), 0, root)
        else:
            hs.hs_batch_digest(ctypes.c_uint64(n), ctypes.c_uint64(77), len(arrays), ptrs, w.ctypes.data_as(ctypes.c_void_p),
                               ad_arg if ad_arg is not None else None, None, len(ads[0]), root)
        assert root.raw == want, (n, mode)


def test_subgroup_by_2descent_equals_r_times_p(hs):
    """codec checked decode: the two-square-roots subgroup test (vrf_core.cuh subgroup_by_2descent) against
    r*P = O (C oracle) on decodable encodings of every coset of the cofactor-4 group, and on the special
    points: identity, the point of order 2, y = 0."""
    from oracle import c_oracle as co
    rnd = random.Random(21)
    cases = [(1).to_bytes(32, "little"), (Q - 1).to_bytes(32, "little"), bytes(32)]
    while len(cases) < 1200:
        enc = bytearray(rnd.randrange(Q).to_bytes(32, "little"))
        enc[31] |= 0x80 * rnd.randrange(2)
        cases.append(bytes(enc))
    cases += [co.hash_to_curve(bytes([i, 7])) for i in range(60)]
    seen = {0: 0, 2: 0}
    n_decodable_outside = 0
    for enc in cases:
        want = 0 if co.point_decode(enc, subgroup=True) is not None else 2
        assert hs.hs_decode_checked(enc) == want, enc.hex()
        seen[want] += 1
        if want == 2 and co.point_decode(enc, subgroup=False) is not None:
            n_decodable_outside += 1
    assert seen[0] > 150 and n_decodable_outside > 300        # members, and curve points outside the subgroup


def test_keyset_comb_row_matches_oracle_multiples(hs):
    """comb_build_row (keyed verification): row w of a key's comb holds j * 256^w * Y for j = 1..255, made affine
    with one shared inversion; spot-checked against the oracle's scalar multiplication."""
    Y = o.te_mul(S, 0x1234567, (S.gx, S.gy))
    buf = ctypes.create_string_buffer(64 * 255)
    for w in (0, 1, 7, 31):
        hs.hs_comb_row(_b(Y[0]), _b(Y[1]), w, buf)
        for j in (1, 2, 3, 100, 254, 255):
            want = o.te_mul(S, (j << (8 * w)) % S.r, Y)
            got = buf.raw[64 * (j - 1): 64 * j]
            assert (int.from_bytes(got[:32], "little"), int.from_bytes(got[32:], "little")) == want, (w, j)
