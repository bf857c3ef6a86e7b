"""GPU (-m gpu): the HIP path, called through the C ABI of libvrfhip.so, against the oracles.

Bit-exact bar (integer/byte work): every output byte equals the CPU oracle's on the same inputs.
Sizes: Python oracle on tens of items, C oracle on thousands, and size-independent properties
(prove->verify round trip, tamper detection, determinism, chunk invariance) at 2^20.
"""
import os
import random

import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import vrf_oracle as o
from conftest import hx

pytestmark = pytest.mark.gpu
S = o.BANDERSNATCH
Q, R = S.q, S.r
NCPU = min(16, os.cpu_count() or 1)


def test_native_library_is_the_one_running(ctx):
    from ark_ec_vrfs_amd import LIB_PATH
    maps = open("/proc/self/maps").read()
    assert LIB_PATH in maps, "libvrfhip.so is not mapped in this process"


def test_fq_mul_against_python_ints(ctx):
    rnd = np.random.default_rng(0)
    a = rnd.integers(0, 256, (20000, 32), dtype=np.uint8)
    b = rnd.integers(0, 256, (20000, 32), dtype=np.uint8)
    edge = [0, 1, Q - 1, Q, Q + 1, (1 << 256) - 1, (1 << 255), (1 << 29) - 1]
    for i, e in enumerate(edge):
        a[i] = np.frombuffer(e.to_bytes(32, "little"), np.uint8)
        b[-1 - i] = np.frombuffer(e.to_bytes(32, "little"), np.uint8)
    r = ctx.fq_mul_batch(a, b)
    for i in range(a.shape[0]):
        x = int.from_bytes(a[i].tobytes(), "little"); y = int.from_bytes(b[i].tobytes(), "little")
        assert int.from_bytes(r[i].tobytes(), "little") == x * y % Q, i


def test_golden_vectors_through_the_c_abi(ctx, kat):
    for v in kat["ietf"]:
        ad, alpha = bytes.fromhex(v["ad"]), bytes.fromhex(v["alpha"])
        sk, pk = ctx.secret_from_seed_batch(hx(v["seed"]).reshape(1, -1))
        assert sk[0].tobytes().hex() == v["sk"] and pk[0].tobytes().hex() == v["pk"]
        assert ctx.hash_to_curve_batch([alpha])[0].tobytes().hex() == v["h"]
        pr = ctx.ietf_prove_batch(hx(v["sk"]), msgs=[alpha], ad=ad)
        got = {k: pr[k][0].tobytes().hex() for k in ("output", "c", "s", "pk", "input")}
        assert got == dict(output=v["gamma"], c=v["c"], s=v["s"], pk=v["pk"], input=v["h"])
        pr2 = ctx.ietf_prove_batch(hx(v["sk"]), inputs=hx(v["h"]), ad=ad)          # pre-hashed input
        assert pr2["c"][0].tobytes().hex() == v["c"] and pr2["s"][0].tobytes().hex() == v["s"]
        assert ctx.output_hash_batch(hx(v["gamma"]))[0].tobytes().hex() == v["beta"]
        assert ctx.ietf_verify_batch(hx(v["pk"]), hx(v["h"]), hx(v["gamma"]), hx(v["c"]), hx(v["s"]), ad=ad)[0] == 0
    assert ctx.point_validate_batch(hx(kat["enc_G"]))[0] == 0


def test_mirror_api_reads_like_the_reference(ctx, kat):
    """Same call sequence as upstream's own tests: from_seed -> Input::new -> output -> prove -> verify."""
    from ark_ec_vrfs_amd import Input, Secret, VerificationFailure, ietf
    v = kat["ietf"][1]
    secret = Secret.from_seed(bytes.fromhex(v["seed"]), ctx=ctx)
    public = secret.public()
    inp = Input.new(bytes.fromhex(v["alpha"]), ctx=ctx)
    out = secret.output(inp, ctx=ctx)
    assert (secret.scalar.hex(), public.encoded.hex(), inp.encoded.hex(), out.encoded.hex()) == \
           (v["sk"], v["pk"], v["h"], v["gamma"])
    assert out.hash(ctx=ctx).hex() == v["beta"]
    proof = ietf.Prover.prove(secret, inp, out, b"", ctx=ctx)
    assert (proof.c.hex(), proof.s.hex()) == (v["c"], v["s"])
    assert ietf.Verifier.verify(public, inp, out, b"", proof, ctx=ctx) is None
    with pytest.raises(VerificationFailure):
        ietf.Verifier.verify(public, inp, out, b"other ad", proof, ctx=ctx)


def test_prove_matches_python_oracle(ctx, synth):
    sk, msg = synth(24, start=0)
    pr = ctx.ietf_prove_batch(sk, msgs=msg, ad=b"xy")
    for i in range(24):
        skv = int.from_bytes(sk[i].tobytes(), "little")
        H = o.data_to_point(S, msg[i].tobytes())
        g, c, s = o.ietf_prove(S, skv, H, b"xy")
        assert pr["input"][i].tobytes() == o.point_encode(S, H)
        assert pr["output"][i].tobytes() == o.point_encode(S, g)
        assert pr["c"][i].tobytes() == o.scalar_encode(c) and pr["s"][i].tobytes() == o.scalar_encode(s)


@pytest.mark.parametrize("ad", [b"", b"additional-data-16", bytes(range(200))])
def test_prove_and_verify_match_c_oracle_at_4096(ctx, synth, ad):
    n = 4096
    sk, msg = synth(n, start=5000)
    ref = co.ietf_prove_batch(sk, msgs=msg, ad=ad, threads=NCPU)
    pr = ctx.ietf_prove_batch(sk, msgs=msg, ad=ad)
    for k in ("output", "c", "s", "pk", "input"):
        assert (pr[k] == ref[k]).all(), k
    assert (pr["status"] == 0).all()
    # verification statuses on a mix of valid and corrupted proofs
    rnd = np.random.default_rng(1)
    arrs = {k: ref[k].copy() for k in ("pk", "input", "output", "c", "s")}
    kinds = rnd.integers(0, 8, n)
    for i in range(n):
        k = kinds[i]
        if k == 1: arrs["s"][i, rnd.integers(0, 31)] ^= 1 << rnd.integers(0, 8)
        elif k == 2: arrs["c"][i, rnd.integers(0, 31)] ^= 1 << rnd.integers(0, 8)
        elif k == 3: arrs["pk"][i] = ref["pk"][(i + 1) % n]
        elif k == 4: arrs["output"][i] = ref["output"][(i + 1) % n]
        elif k == 5: arrs["input"][i] = ref["input"][(i + 1) % n]
    want = co.ietf_verify_batch(arrs["pk"], arrs["input"], arrs["output"], arrs["c"], arrs["s"], ad, threads=NCPU)
    got = ctx.ietf_verify_batch(arrs["pk"], arrs["input"], arrs["output"], arrs["c"], arrs["s"], ad=ad)
    assert (got == want).all()
    assert (want[kinds == 0] == 0).all() and (want[(kinds >= 1) & (kinds <= 5)] != 0).all()


def test_challenge_mod_r_and_noncanonical_identity_encoding(ctx, synth):
    """Two corners where arkworks' decoding is laxer than a strict reading: `ietf::Proof::c` is decoded with
    from_le_bytes_mod_order, so c + k r verifies like c (s is strict); a compressed point with x = 0 decodes whatever its
    sign flag says and is hashed in its canonical form.  GPU == C oracle on both, on every verify entry point."""
    sk, msg = synth(64, start=1300)
    ref = co.ietf_prove_batch(sk, msgs=msg, ad=b"m", threads=4)
    a = {k: ref[k].copy() for k in ("pk", "input", "output", "c", "s")}
    enc = lambda v: np.frombuffer(int(v).to_bytes(32, "little"), np.uint8)
    ival = lambda b: int.from_bytes(b.tobytes(), "little")
    kmax = ((1 << 256) - 1) // R
    for i in range(0, 32):
        k = 1 + i % kmax
        if ival(a["c"][i]) + k * R < (1 << 256):
            a["c"][i] = enc(ival(a["c"][i]) + k * R)                 # still the same field element
    a["s"][40] = enc(ival(a["s"][40]) + R)                           # s is strict: InvalidData
    want = co.ietf_verify_batch(a["pk"], a["input"], a["output"], a["c"], a["s"], b"m", threads=4)
    assert (want[:32] == 0).all() and want[40] == 2 and (np.delete(want, 40) == 0).all()
    got = ctx.ietf_verify_batch(a["pk"], a["input"], a["output"], a["c"], a["s"], ad=b"m")
    assert (got == want).all()
    xy = np.zeros((3, 64, 64), np.uint8)
    for j, k in enumerate(("pk", "input", "output")):
        for i in range(64):
            x, y = co.point_decode(a[k][i].tobytes())
            xy[j, i] = np.frombuffer(x.to_bytes(32, "little") + y.to_bytes(32, "little"), np.uint8)
    assert (ctx.ietf_verify_batch_affine(xy[0], xy[1], xy[2], a["c"], a["s"], ad=b"m") == want).all()
    # identity public key, canonical and with the sign flag set: the same typed point, so the same challenge bytes.
    # A proof for pk = O: sk = 0 is not a valid secret, but U = s G, V = s H - c Gamma can be met with Gamma = O too:
    # take k, U = k G, V = k H, c = challenge(O, H, O, U, V), s = k.
    Sg = S
    H = o.data_to_point(Sg, b"identity corner")
    kk = 123456789
    U, V = o.te_mul(Sg, kk, (Sg.gx, Sg.gy)), o.te_mul(Sg, kk, H)
    ident = (0, 1)
    c = o.challenge_rfc9381(Sg, [ident, H, ident, U, V], b"")
    row = lambda b: np.frombuffer(b, np.uint8).reshape(1, 32)
    ide, flagged = (1).to_bytes(32, "little"), ((1 << 255) | 1).to_bytes(32, "little")
    for pk_b, g_b in ((ide, ide), (flagged, ide), (ide, flagged), (flagged, flagged)):
        args = (row(pk_b), row(o.point_encode(Sg, H)), row(g_b), row(o.scalar_encode(c)), row(o.scalar_encode(kk)))
        w = co.ietf_verify_batch(*args, b"", threads=1)
        g = ctx.ietf_verify_batch(*args, ad=b"")
        assert w[0] == 0 and g[0] == 0, (pk_b[-1], g_b[-1])
    # Output::hash of the flagged identity = hash of the canonical one
    assert (ctx.output_hash_batch(row(flagged)) == ctx.output_hash_batch(row(ide))).all()
    assert ctx.output_hash_batch(row(ide))[0].tobytes() == co.output_hash(ide)


def test_invalid_encodings_and_scalars(ctx, synth):
    sk, msg = synth(16, start=900)
    ref = co.ietf_prove_batch(sk, msgs=msg, ad=b"", threads=4)
    a = {k: ref[k].copy() for k in ("pk", "input", "output", "c", "s")}
    enc = lambda v: np.frombuffer(int(v).to_bytes(32, "little"), np.uint8)
    a["pk"][0] = enc(Q)                    # y >= q
    a["input"][1] = enc((1 << 256) - 1)    # y >= q with flag
    a["c"][2] = enc(R)                     # c == r: `Proof::c` is decoded mod r upstream -> c = 0 -> VerificationFailure
    a["s"][3] = enc((1 << 256) - 1)        # scalar >= r
    # an encoding whose x^2 is a non-square
    y = 2
    while o.point_decode(S, int(y).to_bytes(32, "little")) is not None:
        y += 1
    a["output"][4] = enc(y)
    a["pk"][5] = enc(1)                    # identity (0, 1): decodes; verification simply fails
    a["pk"][6] = enc(Q - 1)                # (0, -1), order 2: decodes
    want = co.ietf_verify_batch(a["pk"], a["input"], a["output"], a["c"], a["s"], b"", threads=4)
    got = ctx.ietf_verify_batch(a["pk"], a["input"], a["output"], a["c"], a["s"], ad=b"")
    assert list(want[:5]) == [2, 2, 1, 2, 2] and (want[7:] == 0).all()
    assert (got[:5] == want[:5]).all() and (got[7:] == want[7:]).all()
    assert got[5] != 0 and got[6] != 0     # small-order pk: precondition violated, but never accepted
    # prove with a non-canonical secret reports InvalidData
    bad_sk = sk.copy(); bad_sk[0] = enc(R)
    pr = ctx.ietf_prove_batch(bad_sk, msgs=msg, ad=b"")
    assert pr["status"][0] == 2 and (pr["status"][1:] == 0).all()
    assert (pr["c"][1:] == ref["c"][1:]).all()


def test_ragged_messages_and_per_item_ad(ctx):
    rnd = random.Random(3)
    n = 40
    msgs = [bytes(rnd.getrandbits(8) for _ in range(rnd.choice([0, 1, 31, 32, 33, 79, 80, 81, 127, 128, 300]))) for _ in range(n)]
    ads = [bytes(rnd.getrandbits(8) for _ in range(rnd.choice([0, 1, 7, 8, 9, 68, 69, 70, 200]))) for _ in range(n)]
    sks = np.stack([np.frombuffer(co.secret_from_seed(bytes([i, 7])), np.uint8) for i in range(n)])
    pr = ctx.ietf_prove_batch(sks, msgs=msgs, ad=ads)
    for i in range(n):
        r = co.ietf_prove_batch(sks[i], inputs=np.frombuffer(co.hash_to_curve(msgs[i]), np.uint8), ad=ads[i])
        assert pr["input"][i].tobytes() == co.hash_to_curve(msgs[i])
        for k in ("output", "c", "s"):
            assert pr[k][i].tobytes() == r[k][0].tobytes(), (i, k)
    st = ctx.ietf_verify_batch(pr["pk"], pr["input"], pr["output"], pr["c"], pr["s"], ad=ads)
    assert (st == 0).all()
    ads2 = list(ads); ads2[5] = ads[5] + b"\x00"
    st = ctx.ietf_verify_batch(pr["pk"], pr["input"], pr["output"], pr["c"], pr["s"], ad=ads2)
    assert st[5] == 1 and st.sum() == 1
    assert (ctx.hash_to_curve_batch(msgs) == pr["input"]).all()


def test_empty_and_single_item_batches(ctx, kat):
    e = np.empty((0, 32), np.uint8)
    assert ctx.ietf_verify_batch(e, e, e, e, e).shape == (0,)
    assert ctx.ietf_prove_batch(e, msgs=[])["c"].shape == (0, 32)
    assert ctx.hash_to_curve_batch([]).shape == (0, 32)
    assert ctx.output_hash_batch(e).shape == (0, 64)
    v = kat["ietf"][0]
    assert ctx.ietf_verify_batch(hx(v["pk"]), hx(v["h"]), hx(v["gamma"]), hx(v["c"]), hx(v["s"]))[0] == 0


def test_point_validation_matches_checked_decode(ctx):
    rnd = random.Random(9)
    encs = [bytes(32), (1).to_bytes(32, "little"), (Q - 1).to_bytes(32, "little"), Q.to_bytes(32, "little"), b"\xff" * 32]
    encs += [rnd.getrandbits(256).to_bytes(32, "little") for _ in range(200)]
    encs += [co.hash_to_curve(bytes([i])) for i in range(20)]
    arr = np.frombuffer(b"".join(encs), np.uint8).reshape(-1, 32)
    st, xy = ctx.point_validate_batch(arr, want_xy=True)
    n_valid = 0
    for i, e in enumerate(encs):
        p = co.point_decode(e, subgroup=True)
        assert (st[i] == 0) == (p is not None), i
        if p:
            n_valid += 1
            assert int.from_bytes(xy[i, :32].tobytes(), "little") == p[0]
            assert int.from_bytes(xy[i, 32:].tobytes(), "little") == p[1]
    assert n_valid >= 21   # identity + the 20 hashed points (random encodings are almost never in the subgroup)


def test_output_hash_and_secret_from_seed_batches(ctx):
    n = 300
    seeds = np.arange(n, dtype=np.uint64).view(np.uint8).reshape(n, 8)
    sk, pk = ctx.secret_from_seed_batch(seeds)
    for i in range(0, n, 7):
        assert sk[i].tobytes() == co.secret_from_seed(seeds[i].tobytes())
        assert pk[i].tobytes() == co.public_from_secret(sk[i].tobytes())
    hs = ctx.output_hash_batch(pk)
    for i in range(0, n, 11):
        assert hs[i].tobytes() == co.output_hash(pk[i].tobytes())


def test_full_size_properties_2_20(ctx):
    """BASELINE.json config sizes: 2^20 items stay on the GPU; checked by size-independent properties
    plus a strided oracle sample."""
    import torch
    from ark_ec_vrfs_amd import _lib
    n = 1 << 20
    dev = torch.device("cuda:0")
    lib = _lib.load()
    st0 = torch.cuda.current_stream().cuda_stream
    seeds = torch.arange(n, dtype=torch.int64, device=dev).view(torch.uint8).reshape(n, 8)
    sk = torch.empty((n, 32), dtype=torch.uint8, device=dev)
    _lib.check(lib.vrfhip_secret_from_seed_batch_dev(ctx.handle, n, seeds.data_ptr(), 8, sk.data_ptr(), None, st0), "seed")
    g = torch.Generator(device=dev); g.manual_seed(1234)
    msg = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    mk = lambda: torch.empty((n, 32), dtype=torch.uint8, device=dev)
    out, c, s, pk, hh = mk(), mk(), mk(), mk(), mk()
    pst = torch.empty(n, dtype=torch.uint8, device=dev)
    ctx.ietf_prove_batch_dev(sk, msg, 32, out, c, s, pk, hh, pst)
    torch.cuda.synchronize()
    assert int(pst.sum()) == 0
    # determinism: same batch twice -> identical bytes
    out2, c2, s2 = mk(), mk(), mk()
    ctx.ietf_prove_batch_dev(sk, msg, 32, out2, c2, s2, None, None, None)
    torch.cuda.synchronize()
    assert torch.equal(out, out2) and torch.equal(c, c2) and torch.equal(s, s2)
    # strided sample against the C oracle
    idx = torch.arange(0, n, 4099, device=dev)
    ref = co.ietf_prove_batch(sk[idx].cpu().numpy(), msgs=msg[idx].cpu().numpy(), ad=b"", threads=NCPU)
    assert (out[idx].cpu().numpy() == ref["output"]).all() and (c[idx].cpu().numpy() == ref["c"]).all()
    assert (s[idx].cpu().numpy() == ref["s"]).all() and (pk[idx].cpu().numpy() == ref["pk"]).all()
    # round trip: every proof verifies
    vs = torch.full((n,), 9, dtype=torch.uint8, device=dev)
    ctx.ietf_verify_batch_dev(pk, hh, out, c, s, vs)
    torch.cuda.synchronize()
    assert int((vs != 0).sum()) == 0
    # tamper every 1024th proof (flip bit 0 of s): exactly those fail
    s_bad = s.clone(); s_bad[::1024, 0] ^= 1
    ctx.ietf_verify_batch_dev(pk, hh, out, c, s_bad, vs)
    torch.cuda.synchronize()
    bad = torch.nonzero(vs).flatten()
    assert torch.equal(bad, torch.arange(0, n, 1024, device=dev)) and int(vs.max()) == 1


def test_chunked_batches_equal_unchunked(ctx, synth):
    """A context whose workspace is smaller than the batch processes it in chunks: same bytes."""
    from ark_ec_vrfs_amd import Context
    sk, msg = synth(700, start=70000)
    a = ctx.ietf_prove_batch(sk, msgs=msg, ad=b"q")
    small = Context(0)
    try:
        small.reserve(256)          # 700 items -> chunks of 256, 256, 188
        assert 0 < small.workspace_bytes() < 2 * 1024 * 1024
        b = small.ietf_prove_batch(sk, msgs=msg, ad=b"q")
        st = small.ietf_verify_batch(a["pk"], a["input"], a["output"], a["c"], a["s"], ad=b"q")
    finally:
        small.close()
    for k in ("output", "c", "s", "pk", "input"):
        assert (a[k] == b[k]).all(), k
    assert (st == 0).all()


def test_pedersen_kat_and_mirror(ctx, kat):
    from ark_ec_vrfs_amd import Input, Secret, VerificationFailure, pedersen
    v, iv = kat["pedersen"][0], kat["ietf"][0]
    secret = Secret.from_seed(bytes.fromhex(v["seed"]), ctx=ctx)
    inp = Input.new(bytes.fromhex(v["alpha"]), ctx=ctx)
    out = secret.output(inp, ctx=ctx)
    proof, blinding = pedersen.Prover.prove(secret, inp, out, bytes.fromhex(v["ad"]), ctx=ctx)
    assert blinding.hex() == v["blinding"] and out.encoded.hex() == iv["gamma"]
    assert (proof.pk_com.hex(), proof.r.hex(), proof.ok.hex(), proof.s.hex(), proof.sb.hex()) == \
           (v["pk_com"], v["r"], v["ok"], v["s"], v["sb"])
    assert pedersen.Verifier.verify(inp, out, bytes.fromhex(v["ad"]), proof, ctx=ctx) is None
    with pytest.raises(VerificationFailure):
        pedersen.Verifier.verify(inp, out, b"x", proof, ctx=ctx)


@pytest.mark.parametrize("ad", [b"", bytes(range(100))])
def test_pedersen_matches_c_oracle_at_2048(ctx, synth, ad):
    n = 2048
    sk, msg = synth(n, start=20000)
    ref = co.pedersen_prove_batch(sk, msgs=msg, ad=ad, threads=NCPU)
    got = ctx.pedersen_prove_batch(sk, msgs=msg, ad=ad)
    for k in ("output", "pk_com", "r", "ok", "s", "sb", "blinding", "input"):
        assert (got[k] == ref[k]).all(), k
    assert (got["status"] == 0).all()
    rnd = np.random.default_rng(2)
    a = {k: ref[k].copy() for k in ("input", "output", "pk_com", "r", "ok", "s", "sb")}
    kinds = rnd.integers(0, 9, n)
    names = [None, "s", "sb", "pk_com", "r", "ok", "output", "input"]
    for i in range(n):
        k = kinds[i]
        if 1 <= k <= 2:
            a[names[k]][i, rnd.integers(0, 31)] ^= 1 << rnd.integers(0, 8)
        elif 3 <= k <= 7:
            a[names[k]][i] = ref[names[k]][(i + 1) % n]
        elif k == 8:
            a["s"][i] = np.frombuffer(int(R).to_bytes(32, "little"), np.uint8)      # non-canonical -> InvalidData
    want = co.pedersen_verify_batch(a["input"], a["output"], a["pk_com"], a["r"], a["ok"], a["s"], a["sb"], ad, threads=NCPU)
    got = ctx.pedersen_verify_batch(a["input"], a["output"], a["pk_com"], a["r"], a["ok"], a["s"], a["sb"], ad=ad)
    assert (got == want).all()
    assert (want[kinds == 0] == 0).all() and (want[kinds == 8] == 2).all() and (want[(kinds >= 1) & (kinds <= 7)] != 0).all()


def test_identity_points_zero_secret(ctx):
    """sk = 0 gives pk = Gamma = identity; arkworks accepts such a proof, so must the GLV path."""
    zero = np.zeros((1, 32), np.uint8)
    msg = np.frombuffer(b"zero-key-message-0123456789abcdef"[:32], np.uint8).reshape(1, 32)
    ref = co.ietf_prove_batch(zero, msgs=msg, ad=b"")
    got = ctx.ietf_prove_batch(zero, msgs=msg, ad=b"")
    for k in ("output", "c", "s", "pk"):
        assert (got[k] == ref[k]).all(), k
    assert got["pk"][0].tobytes() == (1).to_bytes(32, "little")
    assert co.ietf_verify_batch(ref["pk"], ref["input"], ref["output"], ref["c"], ref["s"], b"")[0] == 0
    assert ctx.ietf_verify_batch(ref["pk"], ref["input"], ref["output"], ref["c"], ref["s"], ad=b"")[0] == 0


def _xy_of(ctx, enc):
    st, xy = ctx.point_validate_batch(enc, want_xy=True)
    assert (st == 0).all()
    return xy


@pytest.mark.parametrize("n", [1, 2, 3, 17, 1000, 5000])
def test_msm_matches_naive_oracle(ctx, n):
    rnd = np.random.default_rng(n)
    seeds = rnd.integers(0, 256, (n, 8), dtype=np.uint8)
    sk, pk = ctx.secret_from_seed_batch(seeds)
    xy = _xy_of(ctx, pk)
    k, _ = ctx.secret_from_seed_batch(rnd.integers(0, 256, (n, 9), dtype=np.uint8), with_public=False)
    if n >= 3:
        k[0] = 0                                              # zero scalar
        k[1] = np.frombuffer(int(R - 1).to_bytes(32, "little"), np.uint8)
        xy[2] = xy[1]                                         # repeated base
    got = ctx.msm(xy, k)
    want = co.msm(xy, k)
    assert got == want


def test_msm_edge_cases(ctx):
    from ark_ec_vrfs_amd import InvalidData
    out, xy = ctx.msm(np.empty((0, 64), np.uint8), np.empty((0, 32), np.uint8))
    assert out == (1).to_bytes(32, "little") and xy == bytes(32) + (1).to_bytes(32, "little")
    sk, pk = ctx.secret_from_seed_batch(np.arange(16, dtype=np.uint8).reshape(4, 4))
    xy4 = _xy_of(ctx, pk)
    one = np.zeros((4, 32), np.uint8); one[:, 0] = 1
    # P - P + Q - Q = identity (buckets that cancel)
    neg = xy4.copy()
    for i in (1, 3):
        x = int.from_bytes(xy4[i - 1, :32].tobytes(), "little")
        neg[i, :32] = np.frombuffer(((Q - x) % Q).to_bytes(32, "little"), np.uint8)
        neg[i, 32:] = xy4[i - 1, 32:]
    assert ctx.msm(neg, one)[0] == (1).to_bytes(32, "little")
    bad = xy4.copy(); bad[2, 0] ^= 1                          # off the curve
    with pytest.raises(InvalidData):
        ctx.msm(bad, one)
    bigk = one.copy(); bigk[0] = np.frombuffer(int(R).to_bytes(32, "little"), np.uint8)
    with pytest.raises(InvalidData):
        ctx.msm(xy4, bigk)


def test_msm_full_size_2_20_linearity(ctx):
    """bases a_i*G with known a_i: MSM(bases, k) must equal (sum a_i k_i mod r)*G."""
    import torch
    from ark_ec_vrfs_amd import _lib
    n = 1 << 20
    dev = torch.device("cuda:0")
    lib = _lib.load()
    st0 = torch.cuda.current_stream().cuda_stream
    seeds = torch.arange(n, dtype=torch.int64, device=dev).view(torch.uint8).reshape(n, 8)
    a = torch.empty((n, 32), dtype=torch.uint8, device=dev)
    pk = torch.empty((n, 32), dtype=torch.uint8, device=dev)
    _lib.check(lib.vrfhip_secret_from_seed_batch_dev(ctx.handle, n, seeds.data_ptr(), 8, a.data_ptr(), pk.data_ptr(), st0), "seed")
    xy = torch.empty((n, 64), dtype=torch.uint8, device=dev)
    vst = torch.empty(n, dtype=torch.uint8, device=dev)
    _lib.check(lib.vrfhip_point_validate_batch_dev(ctx.handle, n, pk.data_ptr(), xy.data_ptr(), vst.data_ptr(), st0), "validate")
    seeds2 = (torch.arange(n, dtype=torch.int64, device=dev) + (1 << 40)).view(torch.uint8).reshape(n, 8)
    k = torch.empty((n, 32), dtype=torch.uint8, device=dev)
    _lib.check(lib.vrfhip_secret_from_seed_batch_dev(ctx.handle, n, seeds2.data_ptr(), 8, k.data_ptr(), None, st0), "seed2")
    out = torch.empty(32, dtype=torch.uint8, device=dev); st = torch.empty(1, dtype=torch.uint8, device=dev)
    ctx.msm_dev(xy, k, out, None, st)
    torch.cuda.synchronize()
    assert int(vst.sum()) == 0 and int(st[0]) == 0
    an = a.cpu().numpy(); kn = k.cpu().numpy()
    tot = 0
    for i in range(n):
        tot += int.from_bytes(an[i].tobytes(), "little") * int.from_bytes(kn[i].tobytes(), "little")
    want = co.public_from_secret((tot % R).to_bytes(32, "little"))
    assert out.cpu().numpy().tobytes() == want


@pytest.mark.parametrize("shape", ["all_equal", "two_values", "short_128", "one_bucket_per_window"])
def test_msm_skewed_digit_distributions(ctx, shape):
    """Digit distributions that put long runs into few buckets (the balanced chunking splits them across
    lanes) or leave the top windows empty: results still equal the naive sum."""
    n = 6000
    rnd = np.random.default_rng(77)
    sk, pk = ctx.secret_from_seed_batch(rnd.integers(0, 256, (n, 8), dtype=np.uint8))
    xy = _xy_of(ctx, pk)
    k = np.zeros((n, 32), np.uint8)
    if shape == "all_equal":
        k[:] = np.frombuffer(int(R - 12345).to_bytes(32, "little"), np.uint8)
    elif shape == "two_values":
        k[::2] = np.frombuffer((0x0123456789abcdef0123456789abcdef0123456789abcdef01234567 % R).to_bytes(32, "little"), np.uint8)
        k[1::2] = np.frombuffer((3).to_bytes(32, "little"), np.uint8)
    elif shape == "short_128":
        k[:, :16] = rnd.integers(0, 256, (n, 16), dtype=np.uint8)
    else:
        k[:, 0] = 1024 & 0xff; k[:, 1] = 1024 >> 8                # digit +1024 -> bucket 1023 of window 0 only
    assert ctx.msm(xy, k) == co.msm(xy, k)


def test_msm_2_22_two_rounds_and_extreme_skew(ctx):
    """2^22 points (two rounds of workgroups, 22 groups per window): random scalars against the discrete-log
    identity, then every scalar equal (all points of a window in ONE bucket): k * sum(P) two ways."""
    import torch
    from ark_ec_vrfs_amd import _lib
    n = 1 << 22
    dev = torch.device("cuda:0")
    lib = _lib.load()
    st0 = torch.cuda.current_stream().cuda_stream
    seeds = torch.arange(n, dtype=torch.int64, device=dev).view(torch.uint8).reshape(n, 8)
    a = torch.empty((n, 32), dtype=torch.uint8, device=dev)
    pk = torch.empty((n, 32), dtype=torch.uint8, device=dev)
    _lib.check(lib.vrfhip_secret_from_seed_batch_dev(ctx.handle, n, seeds.data_ptr(), 8, a.data_ptr(), pk.data_ptr(), st0), "seed")
    xy = torch.empty((n, 64), dtype=torch.uint8, device=dev)
    vst = torch.empty(n, dtype=torch.uint8, device=dev)
    _lib.check(lib.vrfhip_point_validate_batch_dev(ctx.handle, n, pk.data_ptr(), xy.data_ptr(), vst.data_ptr(), st0), "validate")
    k = torch.empty((n, 32), dtype=torch.uint8, device=dev)
    seeds2 = (torch.arange(n, dtype=torch.int64, device=dev) + (1 << 41)).view(torch.uint8).reshape(n, 8)
    _lib.check(lib.vrfhip_secret_from_seed_batch_dev(ctx.handle, n, seeds2.data_ptr(), 8, k.data_ptr(), None, st0), "seed2")
    out = torch.empty(32, dtype=torch.uint8, device=dev); st = torch.empty(1, dtype=torch.uint8, device=dev)
    ctx.msm_dev(xy, k, out, None, st)
    torch.cuda.synchronize()
    assert int(vst.sum()) == 0 and int(st[0]) == 0
    le = lambda t: [int.from_bytes(row.tobytes(), "little") for row in t.cpu().numpy()]
    av, kv = le(a), le(k)
    want = co.public_from_secret((sum(x * y for x, y in zip(av, kv)) % R).to_bytes(32, "little"))
    assert out.cpu().numpy().tobytes() == want
    # all scalars equal: one bucket per window holds all 2^22 points
    kc = 0x1234567890abcdef1234567890abcdef1234567890abcdef1234567890abcd % R
    keq = torch.from_numpy(np.frombuffer(kc.to_bytes(32, "little"), np.uint8).copy()).to(dev).repeat(n, 1)
    ctx.msm_dev(xy, keq, out, None, st)
    torch.cuda.synchronize()
    want = co.public_from_secret((sum(av) % R * kc % R).to_bytes(32, "little"))
    assert int(st[0]) == 0 and out.cpu().numpy().tobytes() == want


@pytest.mark.parametrize("n", [511, 513, 8191, 8193, 16385, 70001])
def test_msm_group_boundaries(ctx, n):
    """Sizes around the points-per-group and lanes-per-workgroup boundaries, discrete-log identity."""
    rnd = np.random.default_rng(n)
    a, pk = ctx.secret_from_seed_batch(rnd.integers(0, 256, (n, 8), dtype=np.uint8))
    xy = _xy_of(ctx, pk)
    k, _ = ctx.secret_from_seed_batch(rnd.integers(0, 256, (n, 9), dtype=np.uint8), with_public=False)
    le = lambda t: [int.from_bytes(row.tobytes(), "little") for row in t]
    want = co.public_from_secret((sum(x * y for x, y in zip(le(a), le(k))) % R).to_bytes(32, "little"))
    assert ctx.msm(xy, k)[0] == want


def test_msm_jubjub_matches_naive_oracle():
    from ark_ec_vrfs_amd import Context, JubJubSha512Tai
    cj = Context(0, suite=JubJubSha512Tai, test_blinding_base=True)
    co.set_suite(2)
    try:
        n = 3000
        rnd = np.random.default_rng(9)
        sk, pk = cj.secret_from_seed_batch(rnd.integers(0, 256, (n, 8), dtype=np.uint8))
        st, xy = cj.point_validate_batch(pk, want_xy=True)
        assert (st == 0).all()
        k, _ = cj.secret_from_seed_batch(rnd.integers(0, 256, (n, 9), dtype=np.uint8), with_public=False)
        assert cj.msm(xy, k) == co.msm(xy, k)
    finally:
        co.set_suite(1)
        cj.close()


def test_affine_input_verify_matches_compressed(ctx, synth):
    n = 1024
    sk, msg = synth(n, start=91000)
    pr = ctx.ietf_prove_batch(sk, msgs=msg, ad=b"aff")
    xy = {}
    for k in ("pk", "input", "output"):
        st, v = ctx.point_validate_batch(pr[k], want_xy=True)
        assert (st == 0).all()
        xy[k] = v
    s_bad = pr["s"].copy(); s_bad[::5, 1] ^= 4
    want = ctx.ietf_verify_batch(pr["pk"], pr["input"], pr["output"], pr["c"], s_bad, ad=b"aff")
    got = ctx.ietf_verify_batch_affine(xy["pk"], xy["input"], xy["output"], pr["c"], s_bad, ad=b"aff")
    assert (got == want).all() and (want[::5] == 1).all() and want.sum() == len(want[::5])
    ref = co.ietf_verify_batch(pr["pk"], pr["input"], pr["output"], pr["c"], s_bad, b"aff", threads=NCPU)
    assert (ref == want).all()
    off = xy["pk"].copy(); off[3, 0] ^= 1                                  # off the curve
    big = xy["output"].copy(); big[4, :32] = np.frombuffer(int(Q).to_bytes(32, "little"), np.uint8)   # x >= q
    got = ctx.ietf_verify_batch_affine(off, xy["input"], big, pr["c"], pr["s"], ad=b"aff")
    assert got[3] == 2 and got[4] == 2 and (np.delete(got, [3, 4]) == 0).all()


def test_fused_and_separate_straus_launches_agree(ctx, synth):
    """Batches up to 2^17 run the U and V halves in one launch; a profiled context keeps the separate launches
    (the only path of larger batches).  Same statuses on a mixed batch, IETF and Pedersen."""
    n = 3000
    sk, msg = synth(n, start=123000)
    p = ctx.ietf_prove_batch(sk, msgs=msg, ad=b"f")
    s = p["s"].copy(); s[::7, 0] ^= 1
    s[5] = np.frombuffer(int(R).to_bytes(32, "little"), np.uint8)
    out = p["output"].copy(); out[11] = out[12]
    fused = ctx.ietf_verify_batch(p["pk"], p["input"], out, p["c"], s, ad=b"f")
    ctx.profile(True)
    try:
        separate = ctx.ietf_verify_batch(p["pk"], p["input"], out, p["c"], s, ad=b"f")
    finally:
        ctx.profile(False); ctx.profile_read()
    assert (fused == separate).all()
    assert fused[5] == 2 and fused[11] == 1 and fused[7] == 1 and fused[1] == 0 and int((fused == 0).sum()) > n // 2
    q = ctx.pedersen_prove_batch(sk, msgs=msg, ad=b"f")
    sb = q["sb"].copy(); sb[::5, 3] ^= 4
    fused = ctx.pedersen_verify_batch(q["input"], q["output"], q["pk_com"], q["r"], q["ok"], q["s"], sb, ad=b"f")
    ctx.profile(True)
    try:
        separate = ctx.pedersen_verify_batch(q["input"], q["output"], q["pk_com"], q["r"], q["ok"], q["s"], sb, ad=b"f")
    finally:
        ctx.profile(False); ctx.profile_read()
    assert (fused == separate).all() and fused[5] == 1 and fused[1] == 0
