"""Batched Pedersen verification by random linear combination.

CPU tier: the naive oracle of the batch equation (oracle/c: oracle_pedersen_rlc_check) against the
per-proof oracle: accepts valid batches, rejects any tampering, and is not fooled by two defects that
cancel without weights.  GPU tier: vrfhip_pedersen_verify_batch_rlc(_dev) through the C ABI against both
oracles (verdict, InvalidData statuses, fallback statuses), on Bandersnatch and JubJub, up to 2^20.
"""
import os

import numpy as np
import pytest

from oracle import c_oracle as co, vrf_oracle as o

NCPU = min(8, os.cpu_count() or 1)
SEED = bytes(range(32))
S = o.BANDERSNATCH
FIELDS = ("input", "output", "pk_com", "r", "ok", "s", "sb")


def _proofs(synth, n, start, ad):
    sk, msg = synth(n, start=start)
    ref = co.pedersen_prove_batch(sk, msgs=msg, ad=ad, threads=NCPU)
    return {k: ref[k].copy() for k in FIELDS}


def _args(a):
    return [a[k] for k in FIELDS]


def test_oracle_rlc_accepts_valid_and_rejects_tampered(synth):
    a = _proofs(synth, 24, 500, b"rlc")
    st, fail = co.pedersen_rlc_check(*_args(a), seed=SEED, ad=b"rlc")
    assert fail == 0 and (st == 0).all()
    # any single tampering flips the verdict
    for field in ("s", "sb"):
        b = {k: v.copy() for k, v in a.items()}
        b[field][7, 3] ^= 4
        st, fail = co.pedersen_rlc_check(*_args(b), seed=SEED, ad=b"rlc")
        assert fail == 1 and (st == 0).all()
    for field in ("input", "output", "pk_com", "r", "ok"):
        b = {k: v.copy() for k, v in a.items()}
        b[field][5] = a[field][6]
        st, fail = co.pedersen_rlc_check(*_args(b), seed=SEED, ad=b"rlc")
        assert fail == 1
    # wrong ad: every challenge changes
    st, fail = co.pedersen_rlc_check(*_args(a), seed=SEED, ad=b"other")
    assert fail == 1
    # an undecodable proof is left out of the sum and flagged; the rest still passes
    b = {k: v.copy() for k, v in a.items()}
    b["s"][3] = np.frombuffer(int(S.r).to_bytes(32, "little"), np.uint8)
    st, fail = co.pedersen_rlc_check(*_args(b), seed=SEED, ad=b"rlc")
    assert fail == 0 and st[3] == 2 and st.sum() == 2


def _swap_defects(a):
    """Ok_0 += D and Ok_1 -= D for a subgroup point D: sum_i D1_i stays O although both proofs are wrong."""
    D = o.te_mul(S, 123456789, (S.gx, S.gy))
    b = {k: v.copy() for k, v in a.items()}
    ok0 = o.point_decode(S, a["ok"][0].tobytes())
    ok1 = o.point_decode(S, a["ok"][1].tobytes())
    b["ok"][0] = np.frombuffer(o.point_encode(S, o.te_add(S, ok0, D)), np.uint8)
    b["ok"][1] = np.frombuffer(o.point_encode(S, o.te_add(S, ok1, o.te_neg(S, D))), np.uint8)
    return b


def test_oracle_rlc_weights_defeat_cancelling_defects(synth):
    a = _proofs(synth, 6, 900, b"")
    b = _swap_defects(a)
    per_item = co.pedersen_verify_batch(*_args(b), b"", threads=1)
    assert per_item[0] == 1 and per_item[1] == 1 and (per_item[2:] == 0).all()
    st, fail = co.pedersen_rlc_check(*_args(b), seed=SEED, ad=b"")
    assert fail == 1
    # different seeds, same verdict
    for sd in (b"\x01" * 32, os.urandom(32)):
        assert co.pedersen_rlc_check(*_args(b), seed=sd, ad=b"")[1] == 1
        assert co.pedersen_rlc_check(*_args(a), seed=sd, ad=b"")[1] == 0


# ----------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("n,ad", [(1, b""), (2, b"x"), (37, b""), (1000, bytes(range(90))), (4096, b"ad")])
def test_gpu_rlc_accepts_valid_batches(ctx, synth, n, ad):
    a = _proofs(synth, n, 3000, ad)
    st, ok = ctx.pedersen_verify_batch_rlc(*_args(a), ad=ad, seed=SEED)
    assert ok and (st == 0).all()
    st, ok = ctx.pedersen_verify_batch_rlc(*_args(a), ad=ad)          # fresh random seed
    assert ok and (st == 0).all()


@pytest.mark.gpu
def test_gpu_rlc_verdict_and_statuses_match_the_oracles(ctx, synth):
    import torch
    n, ad = 512, b"mixed"
    a = _proofs(synth, n, 7000, ad)
    rnd = np.random.default_rng(5)
    b = {k: v.copy() for k, v in a.items()}
    for i in rnd.choice(n, 40, replace=False):
        kind = rnd.integers(0, 4)
        if kind == 0:
            b["s"][i, rnd.integers(0, 31)] ^= 1 << rnd.integers(0, 8)
        elif kind == 1:
            b["ok"][i] = a["ok"][(i + 1) % n]
        elif kind == 2:
            b["sb"][i] = np.frombuffer(int(S.r).to_bytes(32, "little"), np.uint8)          # InvalidData
        else:
            b["r"][i] = np.frombuffer((2).to_bytes(32, "little"), np.uint8)                  # y = 2: maybe off-curve
    want_items = co.pedersen_verify_batch(*_args(b), ad, threads=NCPU)
    want_st, want_fail = co.pedersen_rlc_check(*_args(b), seed=SEED, ad=ad)
    assert want_fail == 1
    # device-pointer form: statuses {0, 2} and the verdict byte
    dev = torch.device("cuda:0")
    d = [torch.from_numpy(b[k]).to(dev) for k in FIELDS]
    adt = torch.from_numpy(np.frombuffer(ad, np.uint8).copy()).to(dev)
    st = torch.full((n,), 9, dtype=torch.uint8, device=dev)
    flag = torch.full((1,), 9, dtype=torch.uint8, device=dev)
    ctx.pedersen_verify_batch_rlc_dev(*d, st, flag, SEED, ad=adt, ad_len=len(ad))
    torch.cuda.synchronize()
    assert (st.cpu().numpy() == want_st).all() and int(flag[0]) == 1
    # host form: falls back to the per-proof kernels, statuses equal the per-proof oracle
    got, ok = ctx.pedersen_verify_batch_rlc(*_args(b), ad=ad, seed=SEED)
    assert not ok and (got == want_items).all()
    # only InvalidData items, the rest valid: the batch equation holds and the fast path suffices
    c = {k: v.copy() for k, v in a.items()}
    c["s"][11] = np.frombuffer(int(S.r).to_bytes(32, "little"), np.uint8)
    c["pk_com"][300] = np.frombuffer((3).to_bytes(32, "little"), np.uint8)      # y = 3: no such point
    want_items = co.pedersen_verify_batch(*_args(c), ad, threads=NCPU)
    assert want_items[11] == 2 and want_items[300] == 2 and want_items.sum() == 4
    got, ok = ctx.pedersen_verify_batch_rlc(*_args(c), ad=ad, seed=SEED)
    assert (got == want_items).all() and ok
    c["pk_com"][300] = np.frombuffer((2).to_bytes(32, "little"), np.uint8)      # y = 2 decodes, outside the subgroup
    want_items = co.pedersen_verify_batch(*_args(c), ad, threads=NCPU)
    assert want_items[300] == 2
    got, ok = ctx.pedersen_verify_batch_rlc(*_args(c), ad=ad, seed=SEED)
    assert (got == want_items).all() and ok
    c["pk_com"][300] = a["pk_com"][301]                                         # a valid point, the wrong one
    want_items = co.pedersen_verify_batch(*_args(c), ad, threads=NCPU)
    assert want_items[300] == 1
    got, ok = ctx.pedersen_verify_batch_rlc(*_args(c), ad=ad, seed=SEED)
    assert (got == want_items).all() and not ok


@pytest.mark.gpu
def test_gpu_rlc_rejects_cancelling_defects(ctx, synth):
    a = _proofs(synth, 64, 900, b"")
    b = _swap_defects(a)
    st, ok = ctx.pedersen_verify_batch_rlc(*_args(b), ad=b"", seed=SEED)
    assert not ok and st[0] == 1 and st[1] == 1 and (st[2:] == 0).all()


@pytest.mark.gpu
def test_gpu_rlc_per_item_ad_and_chunking(ctx, synth):
    from ark_ec_vrfs_amd import Context
    n = 300
    sk, msg = synth(n, start=12000)
    ads = [bytes([i % 251]) * (i % 7) for i in range(n)]
    got = ctx.pedersen_prove_batch(sk, msgs=msg, ad=ads)
    a = {k: got[k] for k in FIELDS}
    st, ok = ctx.pedersen_verify_batch_rlc(*_args(a), ad=ads, seed=SEED)
    assert ok and (st == 0).all()
    small = Context(0)
    try:
        small.reserve(128)                      # chunks of 128, 128, 44: one MSM each, verdicts are OR-ed
        st, ok = small.pedersen_verify_batch_rlc(*_args(a), ad=ads, seed=SEED)
        assert ok and (st == 0).all()
        bad = {k: v.copy() for k, v in a.items()}
        bad["sb"][299, 0] ^= 1
        st, ok = small.pedersen_verify_batch_rlc(*_args(bad), ad=ads, seed=SEED)
        assert not ok and st[299] == 1 and st.sum() == 1
    finally:
        small.close()


@pytest.mark.gpu
def test_gpu_rlc_affine_inputs(ctx, synth):
    """x || y inputs (arkworks `Affine`): same verdicts and statuses as the compressed form; off-curve
    points and coordinates >= q are InvalidData; a failed batch falls back to the per-proof kernels."""
    n, ad = 600, b"affine"
    a = _proofs(synth, n, 40000, ad)
    xy = {}
    for k in FIELDS[:5]:
        st, xy[k] = ctx.point_validate_batch(a[k], want_xy=True)
        assert (st == 0).all()
    args = lambda d: [d[k] for k in FIELDS[:5]] + [a["s"], a["sb"]]
    st, ok = ctx.pedersen_verify_batch_rlc(*args(xy), ad=ad, seed=SEED, affine=True)
    assert ok and (st == 0).all()
    bad = {k: v.copy() for k, v in xy.items()}
    bad["ok"][17, 3] ^= 1                                              # off the curve
    bad["r"][99, :32] = np.frombuffer(int(S.q).to_bytes(32, "little"), np.uint8)   # x = q: not canonical
    st, ok = ctx.pedersen_verify_batch_rlc(*args(bad), ad=ad, seed=SEED, affine=True)
    assert ok and st[17] == 2 and st[99] == 2 and st.sum() == 4
    bad["output"][5] = xy["output"][6]                                 # a valid point, the wrong one
    st, ok = ctx.pedersen_verify_batch_rlc(*args(bad), ad=ad, seed=SEED, affine=True)
    assert not ok and st[5] == 1 and st[17] == 2 and st[99] == 2 and st.sum() == 5
    # negated x: still on the curve, different compressed sign bit -> challenge changes -> rejected
    neg = {k: v.copy() for k, v in xy.items()}
    x = int.from_bytes(xy["pk_com"][8, :32].tobytes(), "little")
    neg["pk_com"][8, :32] = np.frombuffer(((S.q - x) % S.q).to_bytes(32, "little"), np.uint8)
    st, ok = ctx.pedersen_verify_batch_rlc(*args(neg), ad=ad, seed=SEED, affine=True)
    assert not ok and st[8] == 1 and st.sum() == 1


@pytest.mark.gpu
def test_gpu_rlc_jubjub(synth):
    from ark_ec_vrfs_amd import Context, JubJubSha512Tai
    cj = Context(0, suite=JubJubSha512Tai, test_blinding_base=True)
    co.set_suite(2)
    try:
        n = 700
        sk, msg = synth(n, start=100)
        got = cj.pedersen_prove_batch(sk, msgs=msg, ad=b"jj")
        a = {k: got[k] for k in FIELDS}
        want_st, want_fail = co.pedersen_rlc_check(*_args(a), seed=SEED, ad=b"jj")
        assert want_fail == 0
        st, ok = cj.pedersen_verify_batch_rlc(*_args(a), ad=b"jj", seed=SEED)
        assert ok and (st == want_st).all()
        a["s"][5, 0] ^= 2
        st, ok = cj.pedersen_verify_batch_rlc(*_args(a), ad=b"jj", seed=SEED)
        assert not ok and st[5] == 1 and st.sum() == 1
    finally:
        co.set_suite(1)
        cj.close()


@pytest.mark.gpu
def test_gpu_rlc_full_size_2_20(ctx):
    """2^20 proofs made by the GPU prover: the single MSM accepts them; one flipped bit anywhere rejects."""
    import torch
    from ark_ec_vrfs_amd import _lib
    dev = torch.device("cuda:0")
    n = 1 << 20
    lib = _lib.load()
    st0 = torch.cuda.current_stream().cuda_stream
    seeds = torch.arange(n, dtype=torch.int64, device=dev).view(torch.uint8).reshape(n, 8)
    sk = torch.empty((n, 32), dtype=torch.uint8, device=dev)
    _lib.check(lib.vrfhip_secret_from_seed_batch_dev(ctx.handle, n, seeds.data_ptr(), 8, sk.data_ptr(), None, st0), "seed")
    msg = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev)
    mk = lambda: torch.empty((n, 32), dtype=torch.uint8, device=dev)
    g, pc, r, ok, s, sb, hh = (mk() for _ in range(7))
    pst = torch.empty(n, dtype=torch.uint8, device=dev)
    ctx.pedersen_prove_batch_dev(sk, msg, 32, g, pc, r, ok, s, sb, None, hh, pst)
    st = torch.full((n,), 9, dtype=torch.uint8, device=dev)
    flag = torch.full((1,), 9, dtype=torch.uint8, device=dev)
    ctx.pedersen_verify_batch_rlc_dev(hh, g, pc, r, ok, s, sb, st, flag, SEED)
    torch.cuda.synchronize()
    assert int(flag[0]) == 0 and int(st.max()) == 0
    for victim, field in ((n - 1, s), (123457, sb), (0, ok)):
        saved = field[victim].clone()
        if field is ok:
            field[victim] = ok[victim + 1]
        else:
            field[victim, 5] ^= 16
        ctx.pedersen_verify_batch_rlc_dev(hh, g, pc, r, ok, s, sb, st, flag, SEED)
        torch.cuda.synchronize()
        assert int(flag[0]) == 1 and int(st.max()) == 0
        field[victim] = saved
    ctx.pedersen_verify_batch_rlc_dev(hh, g, pc, r, ok, s, sb, st, flag, os.urandom(32))
    torch.cuda.synchronize()
    assert int(flag[0]) == 0


def _shift_by_order2(enc):
    """enc(P + (0, -1)) = enc((-x, -y)): the same point plus the rational 2-torsion point (outside the subgroup)."""
    x, y = o.point_decode(S, enc.tobytes())
    return np.frombuffer(o.point_encode(S, ((-x) % S.q, (-y) % S.q)), np.uint8)


def test_oracle_rlc_lone_small_order_defect_is_caught(synth):
    """Weights are 1 (mod 8): a single proof whose Ok (or R) is shifted by a point of order 2 fails the batch
    equation for every seed, although an even weight would have annihilated the defect."""
    a = _proofs(synth, 8, 4321, b"t")
    for field in ("ok", "r"):
        b = {k: v.copy() for k, v in a.items()}
        b[field][3] = _shift_by_order2(a[field][3])
        # checked decode (the oracle's default, as upstream): the shifted point is InvalidData and left out of the sum
        assert co.pedersen_verify_batch(*_args(b), b"t", threads=1)[3] == 2
        st, fail = co.pedersen_rlc_check(*_args(b), seed=SEED, ad=b"t")
        assert st[3] == 2 and fail == 0
        co.set_check_mask(0)                       # points vouched for by the caller: the odd weights catch it
        try:
            assert co.pedersen_verify_batch(*_args(b), b"t", threads=1)[3] == 1
            for sd in (SEED, b"\x07" * 32, os.urandom(32), os.urandom(32)):
                assert co.pedersen_rlc_check(*_args(b), seed=sd, ad=b"t")[1] == 1
        finally:
            co.set_check_mask(15)


@pytest.mark.gpu
def test_gpu_rlc_lone_small_order_defect_is_caught(ctx, synth):
    a = _proofs(synth, 200, 4321, b"t")
    b = {k: v.copy() for k, v in a.items()}
    b["ok"][77] = _shift_by_order2(a["ok"][77])
    # default (checked decode, as arkworks' deserialisation): the shifted point is InvalidData and left out
    st, ok = ctx.pedersen_verify_batch_rlc(*_args(b), ad=b"t")
    assert ok and st[77] == 2 and st.sum() == 2
    # points declared pre-validated by the caller: the odd weights still catch the lone defect
    ctx.set_prevalidated(True)
    try:
        for _ in range(4):
            st, ok = ctx.pedersen_verify_batch_rlc(*_args(b), ad=b"t")          # fresh random seeds
            assert not ok and st[77] == 1 and st.sum() == 1
    finally:
        ctx.set_prevalidated(False)
