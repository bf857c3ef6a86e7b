"""GPU (-m gpu): (1) the test primitives of SURVEY.md section 8b -- the twisted-Edwards group law, variable-base scalar
multiplication, SHA-512 and expand_message_xmd on their own through the C ABI against Python big ints / hashlib / the
stage vectors of the golden file; (2) the one-call multi-device entry points (vrfhip_*_batch_multi): several contexts
on device 0 must give byte-for-byte the single-context results (contiguous slices, ragged messages, per-item ad,
batches smaller than the context count)."""
import hashlib
import os
import random

import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import vrf_oracle as o

pytestmark = pytest.mark.gpu
S = o.BANDERSNATCH
NCPU = min(16, os.cpu_count() or 1)
enc = lambda Sx, P: np.frombuffer(o.point_encode(Sx, P), np.uint8)
le = lambda v: np.frombuffer(int(v).to_bytes(32, "little"), np.uint8)


def _law_cases(Sx, rnd, n):
    G = (Sx.gx, Sx.gy)
    pts = [o.te_mul(Sx, rnd.randrange(1, Sx.r), G) for _ in range(n)]
    a = pts + [pts[0], pts[1], (0, 1), pts[2], (0, 1)]
    b = pts[1:] + pts[:1] + [pts[0], o.te_neg(Sx, pts[1]), pts[3], (0, 1), (0, 1)]      # doubling, inverse, identities
    return a, b


def test_point_add_and_scalar_mul_bandersnatch(ctx):
    rnd = random.Random(41)
    a, b = _law_cases(S, rnd, 24)
    out, st = ctx.test_point_add(np.stack([enc(S, P) for P in a]), np.stack([enc(S, P) for P in b]))
    assert (st == 0).all()
    for i, (P, Qp) in enumerate(zip(a, b)):
        assert out[i].tobytes() == o.point_encode(S, o.te_add(S, P, Qp)), i
    # an undecodable operand
    bad = np.stack([enc(S, a[0]), le(S.q)])
    out, st = ctx.test_point_add(bad, np.stack([enc(S, a[1]), enc(S, a[1])]))
    assert list(st) == [0, 2]
    # k * P against the double-and-add oracle, edge scalars included
    ks = [0, 1, 2, S.r - 1, S.r - 2, (1 << 128) - 1, 1 << 128, (1 << 252) + 12345] + [rnd.randrange(S.r) for _ in range(24)]
    ks = [k % S.r for k in ks]
    pts = [o.te_mul(S, rnd.randrange(1, S.r), (S.gx, S.gy)) for _ in ks]
    out, st = ctx.test_scalar_mul(np.stack([le(k) for k in ks]), np.stack([enc(S, P) for P in pts]))
    assert (st == 0).all()
    for i, (k, P) in enumerate(zip(ks, pts)):
        assert out[i].tobytes() == o.point_encode(S, o.te_mul(S, k, P)), i
    # k * G == Secret::public for a thousand scalars (C oracle), and k >= r is InvalidData
    n = 1000
    sk = np.stack([np.frombuffer(co.secret_from_seed(bytes([i & 255, i >> 8, 3])), np.uint8) for i in range(n)])
    G = np.tile(enc(S, (S.gx, S.gy)), (n, 1))
    out, st = ctx.test_scalar_mul(sk, G)
    for i in range(n):
        assert out[i].tobytes() == co.public_from_secret(sk[i].tobytes()), i
    out, st = ctx.test_scalar_mul(np.stack([le(S.r), le(5)]), G[:2])
    assert list(st) == [2, 0]


def test_point_add_and_scalar_mul_jubjub():
    from ark_ec_vrfs_amd import Context, JubJubSha512Tai
    J = o.jubjub_params()
    rnd = random.Random(42)
    cj = Context(0, suite=JubJubSha512Tai, test_blinding_base=True)
    try:
        a, b = _law_cases(J, rnd, 12)
        out, st = cj.test_point_add(np.stack([enc(J, P) for P in a]), np.stack([enc(J, P) for P in b]))
        assert (st == 0).all()
        for i, (P, Qp) in enumerate(zip(a, b)):
            assert out[i].tobytes() == o.point_encode(J, o.te_add(J, P, Qp)), i
        ks = [0, 1, J.r - 1] + [rnd.randrange(J.r) for _ in range(13)]
        pts = [o.te_mul(J, rnd.randrange(1, J.r), (J.gx, J.gy)) for _ in ks]
        out, st = cj.test_scalar_mul(np.stack([le(k) for k in ks]), np.stack([enc(J, P) for P in pts]))
        for i, (k, P) in enumerate(zip(ks, pts)):
            assert out[i].tobytes() == o.point_encode(J, o.te_mul(J, k, P)), i
    finally:
        cj.close()


def test_sha512_and_xmd_primitives(ctx, kat):
    rnd = random.Random(43)
    lens = list(range(0, 20)) + [47, 48, 55, 56, 63, 64, 110, 111, 112, 113, 127, 128, 129, 239, 240, 255, 256, 300, 1000]
    msgs = [bytes(rnd.getrandbits(8) for _ in range(k)) for k in lens]
    d = ctx.test_sha512(msgs)
    for i, m in enumerate(msgs):
        assert d[i].tobytes() == hashlib.sha512(m).digest(), lens[i]
    x = ctx.test_xmd(msgs)
    for i, m in enumerate(msgs):
        assert x[i].tobytes() == o.xmd_sha512_96(m, S.h2c_dst), lens[i]
    # stage vectors of the golden file
    alphas = [bytes.fromhex(v["alpha"]) for v in kat["stages"]]
    x = ctx.test_xmd(alphas)
    for i, v in enumerate(kat["stages"]):
        assert x[i].tobytes().hex() == v["xmd96"]


def test_multi_context_calls_equal_single_context(ctx):
    from ark_ec_vrfs_amd import (Context, ietf_prove_batch_multi, ietf_verify_batch_multi, pedersen_prove_batch_multi,
                                 pedersen_verify_batch_multi)
    rnd = random.Random(44)
    extra = [Context(0), Context(0)]
    ctxs = [ctx] + extra
    try:
        for n in (1, 2, 1000):
            sk = np.stack([np.frombuffer(co.secret_from_seed(bytes([i & 255, i >> 8, 9])), np.uint8) for i in range(n)])
            msgs = [bytes(rnd.getrandbits(8) for _ in range(rnd.choice([0, 1, 32, 33, 100]))) for _ in range(n)]
            ads = [bytes(rnd.getrandbits(8) for _ in range(rnd.choice([0, 5, 70]))) for _ in range(n)]
            one = ctx.ietf_prove_batch(sk, msgs=msgs, ad=ads)
            many = ietf_prove_batch_multi(ctxs, sk, msgs, ad=ads)
            for k in ("output", "c", "s", "pk", "input", "status"):
                assert (one[k] == many[k]).all(), (n, k)
            s_bad = one["s"].copy(); s_bad[::3, 2] ^= 1
            st1 = ctx.ietf_verify_batch(one["pk"], one["input"], one["output"], one["c"], s_bad, ad=ads)
            stm = ietf_verify_batch_multi(ctxs, one["pk"], one["input"], one["output"], one["c"], s_bad, ad=ads)
            assert (st1 == stm).all() and st1[::3].all()
            p1 = ctx.pedersen_prove_batch(sk, msgs=msgs, ad=b"shared")
            pm = pedersen_prove_batch_multi(ctxs, sk, msgs, ad=b"shared")
            for k in ("output", "pk_com", "r", "ok", "s", "sb", "blinding", "input"):
                assert (p1[k] == pm[k]).all(), (n, k)
            args = [p1[k] for k in ("input", "output", "pk_com", "r", "ok", "s", "sb")]
            args[6] = args[6].copy(); args[6][::4, 0] ^= 8
            v1 = ctx.pedersen_verify_batch(*args, ad=b"shared")
            assert (pedersen_verify_batch_multi(ctxs, *args, ad=b"shared") == v1).all()
            assert (pedersen_verify_batch_multi(ctxs, *args, ad=b"shared", rlc_seed=os.urandom(32)) == v1).all()
            assert v1[::4].all()
    finally:
        for c in extra:
            c.close()


@pytest.mark.gpu
def test_gpu_batch_digest_equals_the_oracle_and_the_definition(ctx):
    """vrfhip_test_batch_digest (the digest hashed into the weights of the batched verifiers) against the C oracle and
    the hashlib restatement: one to four node levels, seven arrays as the Pedersen entry point passes them,
    shared / per-item / no ad, a non-zero first index."""
    from oracle import c_oracle as co
    from test_hostsim import py_batch_digest
    rng = np.random.default_rng(9)
    for n, widths, mode in ((1, (32,), "none"), (16, (192,), "none"), (17, (32,) * 7, "per"), (257, (36,), "per"),
                            (4097, (64,) * 5 + (32, 32), "shared"), (40000, (32,) * 7, "per")):
        arrays = [rng.integers(0, 256, (n, w), dtype=np.uint8) for w in widths]
        if mode == "per":
            ads = [bytes(rng.integers(0, 256, int(rng.integers(0, 24)), dtype=np.uint8)) for _ in range(n)]
            ad_arg = ads
        elif mode == "shared":
            ads, ad_arg = [b"one ad for all"] * n, b"one ad for all"
        else:
            ads, ad_arg = [b""] * n, None
        got = ctx.test_batch_digest(arrays, ad_arg, index0=5)
        assert got == co.batch_digest(arrays, ad_arg, index0=5), (n, mode)
        if n <= 4097:
            assert got == py_batch_digest(arrays, ads, index0=5), (n, mode)
    # every input byte matters
    arrays = [rng.integers(0, 256, (300, 32), dtype=np.uint8) for _ in range(3)]
    base = ctx.test_batch_digest(arrays, b"x")
    for j, i in ((0, 0), (1, 150), (2, 299)):
        mod = [a.copy() for a in arrays]
        mod[j][i, 31] ^= 1
        assert ctx.test_batch_digest(mod, b"x") != base
    assert ctx.test_batch_digest(arrays, b"y") != base and ctx.test_batch_digest(arrays, b"x", index0=1) != base


@pytest.mark.gpu
@pytest.mark.parametrize("suite_name", ["bandersnatch", "jubjub"])
def test_gpu_prove_points_affine_equal_the_decoded_compressed_outputs(suite_name):
    """VRFHIP_FLAG_PROVE_POINTS_AFFINE: the provers write output / pk / pk_com / r / ok as x || y.  Those must be the
    coordinates of exactly the points the default (compressed) mode returns -- decoded here by the C oracle -- for the
    IETF and Pedersen provers, hashed and given inputs, 1 and several proofs per lane, one and three contexts, and a
    failed item (sk >= r) must come back all-zero."""
    from ark_ec_vrfs_amd import Context, JubJubSha512Tai, BandersnatchSha512Ell2
    from ark_ec_vrfs_amd import api as A
    jj = suite_name == "jubjub"
    suite = JubJubSha512Tai if jj else BandersnatchSha512Ell2
    if jj:
        co.set_suite(2)
    try:
        ctxs = [Context(0, suite=suite, test_blinding_base=True) for _ in range(3)]
        c0 = ctxs[0]
        n = 300
        sk = np.stack([np.frombuffer(co.secret_from_seed(o.synth_seed(4000 + i)), np.uint8) for i in range(n)])
        sk[7] = 0xff                                      # >= r: InvalidData
        msgs = [o.synth_msg(i)[: (i % 40)] for i in range(n)]

        def xy_of(enc):
            out = np.zeros((enc.shape[0], 64), np.uint8)
            for i, e in enumerate(enc):
                r = co.point_decode(bytes(e), subgroup=False)
                if r is not None and bytes(e) != bytes(32):
                    out[i] = np.frombuffer(r[0].to_bytes(32, "little") + r[1].to_bytes(32, "little"), np.uint8)
            return out

        ref_i = c0.ietf_prove_batch(sk, msgs=msgs, ad=b"xy")
        ref_p = c0.pedersen_prove_batch(sk, msgs=msgs, ad=b"xy")
        assert ref_i["status"][7] == 2 and (ref_i["status"] == 0).sum() == n - 1
        for c in ctxs:
            c.set_flags(c.PROVE_POINTS_AFFINE)
        assert c0.prove_point_bytes() == 64
        got_i = c0.ietf_prove_batch(sk, msgs=msgs, ad=b"xy")
        got_p = c0.pedersen_prove_batch(sk, msgs=msgs, ad=b"xy")
        for k in ("c", "s", "input", "status"):
            assert (got_i[k] == ref_i[k]).all(), k
        for k in ("s", "sb", "blinding", "input", "status"):
            assert (got_p[k] == ref_p[k]).all(), k
        for k in ("output", "pk"):
            assert got_i[k].shape == (n, 64) and (got_i[k] == xy_of(ref_i[k])).all(), k
        for k in ("output", "pk_com", "r", "ok"):
            assert got_p[k].shape == (n, 64) and (got_p[k] == xy_of(ref_p[k])).all(), k
        assert not got_i["output"][7].any() and not got_p["r"][7].any()
        # given inputs (the Rust crate's path) and the multi-context calls
        gi = c0.ietf_prove_batch(sk, inputs=ref_i["input"], ad=b"xy")
        assert (gi["output"] == got_i["output"]).all() and (gi["s"] == ref_i["s"]).all()
        mi = A.ietf_prove_batch_multi(ctxs, sk, msgs, ad=b"xy")
        mp = A.pedersen_prove_batch_multi(ctxs, sk, msgs, ad=b"xy")
        for k in ("output", "pk", "c", "s"):
            assert (mi[k] == got_i[k]).all(), k
        for k in ("output", "pk_com", "r", "ok", "s", "sb"):
            assert (mp[k] == got_p[k]).all(), k
        # several proofs per lane (K > 1 needs >= 2^18 items): x || y of a strided sample against the compressed run
        big = 1 << 18
        skb = np.tile(sk[:256], (big // 256, 1))
        mb = np.frombuffer(np.random.default_rng(3).bytes(big * 16), np.uint8).reshape(big, 16)
        c0.set_flags(0)
        rb = c0.ietf_prove_batch(skb, msgs=mb, ad=b"")
        c0.set_flags(c0.PROVE_POINTS_AFFINE)
        gb = c0.ietf_prove_batch(skb, msgs=mb, ad=b"")
        sel = np.arange(0, big, 1021)
        assert (gb["output"][sel] == xy_of(rb["output"][sel])).all() and (gb["pk"][sel] == xy_of(rb["pk"][sel])).all()
        assert (gb["c"] == rb["c"]).all() and (gb["s"] == rb["s"]).all()
        # contexts that disagree on the flag are refused
        ctxs[1].set_flags(0)
        with pytest.raises(Exception):
            A.ietf_prove_batch_multi(ctxs, sk, msgs, ad=b"xy")
        for c in ctxs:
            c.close()
    finally:
        if jj:
            co.set_suite(1)


@pytest.mark.gpu
def test_gpu_coords_mont256_equal_the_canonical_format(ctx):
    """VRFHIP_FLAG_COORDS_MONT256 (arkworks' in-memory field elements at the ABI): every x || y pair the
    library reads or writes is x 2^256 mod q.  With the flag set and inputs converted, the affine verifier, the batched
    Pedersen verifier on affine points and the MSM must give the results of the canonical format; the provers' x || y
    outputs, point_validate's xy_out and the MSM's out_xy must be the Montgomery images of the canonical ones."""
    from ark_ec_vrfs_amd import Context
    Q = o.Q
    R256 = (1 << 256) % Q
    def to_mont(xy):                                   # (n, 64) canonical -> Montgomery-256
        out = np.zeros_like(xy)
        for i, row in enumerate(xy):
            x, y = int.from_bytes(row[:32].tobytes(), "little"), int.from_bytes(row[32:].tobytes(), "little")
            out[i] = np.frombuffer((x * R256 % Q).to_bytes(32, "little") + (y * R256 % Q).to_bytes(32, "little"), np.uint8)
        return out
    c = Context(0)
    try:
        n = 200
        sk = np.stack([np.frombuffer(co.secret_from_seed(o.synth_seed(6000 + i)), np.uint8) for i in range(n)])
        msgs = [o.synth_msg(i)[: 5 + i % 20] for i in range(n)]
        ref = c.ietf_prove_batch(sk, msgs=msgs, ad=b"m256")
        ped = c.pedersen_prove_batch(sk, msgs=msgs, ad=b"m256")
        # canonical x || y of every point
        _, xy = zip(*[c.point_validate_batch(ref[k], want_xy=True) for k in ("pk", "input", "output")])
        pk_xy, in_xy, out_xy = xy
        _, pxy = zip(*[c.point_validate_batch(ped[k], want_xy=True) for k in ("input", "output", "pk_com", "r", "ok")])
        s_bad = ref["s"].copy(); s_bad[::9, 0] ^= 1
        want_v = c.ietf_verify_batch_affine(pk_xy, in_xy, out_xy, ref["c"], s_bad, ad=b"m256")
        sb_bad = ped["sb"].copy(); sb_bad[5, 1] ^= 4
        want_r, want_ok = c.pedersen_verify_batch_rlc(*pxy, ped["s"], sb_bad, ad=b"m256", seed=bytes(32), affine=True)
        k = np.stack([np.frombuffer(int(7 + 13 * i).to_bytes(32, "little"), np.uint8) for i in range(n)])
        want_m = c.msm(out_xy, k)
        c.set_flags(c.COORDS_MONT256)
        assert (c.ietf_verify_batch_affine(to_mont(pk_xy), to_mont(in_xy), to_mont(out_xy), ref["c"], s_bad, ad=b"m256") == want_v).all()
        assert (want_v[::9] == 1).all() and want_v.sum() == len(want_v[::9])
        got_r, got_ok = c.pedersen_verify_batch_rlc(*[to_mont(a) for a in pxy], ped["s"], sb_bad, ad=b"m256", seed=bytes(32), affine=True)
        assert (got_r == want_r).all() and got_ok == want_ok and want_r[5] == 1
        got_m = c.msm(to_mont(out_xy), k)
        assert got_m[0] == want_m[0] and got_m[1] == to_mont(np.frombuffer(want_m[1], np.uint8).reshape(1, 64)).tobytes()
        st, vxy = c.point_validate_batch(ref["output"], want_xy=True)
        assert (vxy == to_mont(out_xy)).all()
        # coordinates >= q are InvalidData in this format too
        bad = to_mont(pk_xy); bad[3, :32] = 0xff
        assert c.ietf_verify_batch_affine(bad, to_mont(in_xy), to_mont(out_xy), ref["c"], ref["s"], ad=b"m256")[3] == 2
        # provers: x || y outputs in Montgomery-256 form
        c.set_flags(c.COORDS_MONT256 | c.PROVE_POINTS_AFFINE)
        gi = c.ietf_prove_batch(sk, msgs=msgs, ad=b"m256")
        gp = c.pedersen_prove_batch(sk, msgs=msgs, ad=b"m256")
        assert (gi["output"] == to_mont(out_xy)).all() and (gi["pk"] == to_mont(pk_xy)).all()
        assert (gi["c"] == ref["c"]).all() and (gi["s"] == ref["s"]).all()
        for key, idx in (("output", 1), ("pk_com", 2), ("r", 3), ("ok", 4)):
            assert (gp[key] == to_mont(pxy[idx])).all(), key
    finally:
        c.close()


@pytest.mark.gpu
def test_ct_table_lookups_give_the_same_proofs():
    """VRFHIP_FLAG_CT_TABLES: the provers' per-proof window-table lookups read all eight entries and keep
    one by masks.  Proof bytes must not change -- every suite, both schemes, against the same context without the flag (which
    the parity tests hold against the oracles)."""
    import numpy as np
    from ark_ec_vrfs_amd import (BabyJubJubSha512Tai, BandersnatchSha512Ell2, Context, Ed25519Sha512Tai, JubJubSha512Tai,
                                 Secp256r1Sha256Tai)
    n = 3000
    for suite in (BandersnatchSha512Ell2, JubJubSha512Tai, Ed25519Sha512Tai, BabyJubJubSha512Tai, Secp256r1Sha256Tai):
        c = Context(0, suite=suite, test_blinding_base=True)
        try:
            sk, _ = c.secret_from_seed_batch(np.arange(n * 8, dtype=np.uint8).reshape(n, 8))
            sk[7] = 255                                   # not canonical: InvalidData either way
            msgs = np.random.default_rng(5).integers(0, 256, (n, 19), dtype=np.uint8)
            a = c.ietf_prove_batch(sk, msgs=msgs, ad=b"ct")
            p = c.pedersen_prove_batch(sk, msgs=msgs, ad=b"ct")
            c.set_flags(c.CT_TABLES)
            a2 = c.ietf_prove_batch(sk, msgs=msgs, ad=b"ct")
            p2 = c.pedersen_prove_batch(sk, msgs=msgs, ad=b"ct")
            c.set_flags(0)
            assert a["status"][7] == 2 and a["status"].sum() == 2
            for k in a:
                assert (a[k] == a2[k]).all(), (suite.__name__, k)
            for k in p:
                assert (p[k] == p2[k]).all(), (suite.__name__, k)
        finally:
            c.close()
