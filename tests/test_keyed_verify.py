"""Keyed IETF verification (key sets: validated public keys with context-resident fixed-base combs).
The statuses must equal those of the plain verifier / the C oracle on the same proofs, whatever mix of keys,
and invalid keys (undecodable, outside the prime-order subgroup, out-of-range index) give InvalidData."""
import os

import numpy as np
import pytest

from oracle import c_oracle as co, vrf_oracle as o

pytestmark = pytest.mark.gpu
NCPU = min(8, os.cpu_count() or 1)
S = o.BANDERSNATCH


def _batch(synth, n_keys, n, ad):
    sk, _ = synth(n_keys, start=300)
    _, msg = synth(n, start=9000)
    key = np.arange(n, dtype=np.uint32) * 7 % n_keys
    ref = co.ietf_prove_batch(sk[key], msgs=msg, ad=ad, threads=NCPU)
    pks = np.stack([np.frombuffer(co.public_from_secret(sk[i].tobytes()), np.uint8) for i in range(n_keys)])
    return pks, key, ref


def test_keyed_equals_plain_verifier(ctx, synth):
    ad = b"keyed"
    pks, key, ref = _batch(synth, 37, 3000, ad)
    ks, kst = ctx.keyset_create(pks)
    try:
        assert (kst == 0).all() and ks.bytes() >= 37 * 881280
        st = ctx.ietf_verify_batch_keyed(ks, key, ref["input"], ref["output"], ref["c"], ref["s"], ad=ad)
        assert (st == 0).all()
        # tamper: wrong key, flipped s, swapped output, c >= r, s >= r, wrong ad
        rnd = np.random.default_rng(4)
        a = {k: ref[k].copy() for k in ("input", "output", "c", "s")}
        key2 = key.copy()
        for i in rnd.choice(3000, 300, replace=False):
            kind = rnd.integers(0, 4)
            if kind == 0:
                key2[i] = (key2[i] + 1) % 37
            elif kind == 1:
                a["s"][i, rnd.integers(0, 31)] ^= 1 << rnd.integers(0, 8)
            elif kind == 2:
                a["output"][i] = ref["output"][(i + 1) % 3000]
            elif i % 2:
                a["c"][i] = 0xff                  # c >= r: decoded mod r (a wrong challenge, not invalid data)
            else:
                a["s"][i] = 0xff                  # s >= r: strict -> InvalidData
        want = co.ietf_verify_batch(pks[key2], a["input"], a["output"], a["c"], a["s"], ad, threads=NCPU)
        got = ctx.ietf_verify_batch_keyed(ks, key2, a["input"], a["output"], a["c"], a["s"], ad=ad)
        assert (got == want).all() and (want == 1).sum() > 100 and (want == 2).sum() > 15
        plain = ctx.ietf_verify_batch(pks[key2], a["input"], a["output"], a["c"], a["s"], ad=ad)
        assert (plain == got).all()
        assert (ctx.ietf_verify_batch_keyed(ks, key, ref["input"], ref["output"], ref["c"], ref["s"], ad=b"other") == 1).all()
    finally:
        ks.close()


def test_keyed_invalid_keys_and_indices(ctx, synth):
    ad = b""
    pks, key, ref = _batch(synth, 8, 64, ad)
    bad = pks.copy()
    bad[2] = np.frombuffer((3).to_bytes(32, "little"), np.uint8)             # y = 3: not on the curve
    # a decodable point outside the prime-order subgroup (found by search against the checked decode)
    y = 2
    while True:
        enc = y.to_bytes(32, "little")
        if co.point_decode(enc, subgroup=False) is not None and co.point_decode(enc, subgroup=True) is None:
            break
        y += 1
    bad[5] = np.frombuffer(enc, np.uint8)
    ks, kst = ctx.keyset_create(bad)
    try:
        assert list(kst) == [0, 0, 2, 0, 0, 2, 0, 0]
        idx = key.copy()
        idx[10] = 8                                                           # out of range
        idx[11] = 0xffffffff
        st = ctx.ietf_verify_batch_keyed(ks, idx, ref["input"], ref["output"], ref["c"], ref["s"], ad=ad)
        for i in range(64):
            if i in (10, 11) or idx[i] in (2, 5):
                assert st[i] == 2, i
            else:
                assert st[i] == 0, i
    finally:
        ks.close()


def test_keyed_full_size_2_20(ctx):
    """2^20 proofs over 1024 keys made by the GPU prover: all verify; every 1024th tampered proof is caught;
    statuses equal the plain verifier's."""
    import torch
    from ark_ec_vrfs_amd import _lib
    dev = torch.device("cuda:0")
    n, nk = 1 << 20, 1024
    lib = _lib.load()
    st0 = torch.cuda.current_stream().cuda_stream
    seeds = torch.arange(nk, dtype=torch.int64, device=dev).view(torch.uint8).reshape(nk, 8)
    ksk = torch.empty((nk, 32), dtype=torch.uint8, device=dev); kpk = torch.empty((nk, 32), dtype=torch.uint8, device=dev)
    _lib.check(lib.vrfhip_secret_from_seed_batch_dev(ctx.handle, nk, seeds.data_ptr(), 8, ksk.data_ptr(), kpk.data_ptr(), st0), "seed")
    idx = (torch.arange(n, device=dev) * 2654435761 % nk).to(torch.int32)
    sk = ksk[idx.long()].contiguous()
    msg = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev)
    mk = lambda: torch.empty((n, 32), dtype=torch.uint8, device=dev)
    g, c, s, pk, hh = (mk() for _ in range(5))
    pst = torch.empty(n, dtype=torch.uint8, device=dev)
    ctx.ietf_prove_batch_dev(sk, msg, 32, g, c, s, pk, hh, pst)
    torch.cuda.synchronize()
    ks, kst = ctx.keyset_create(kpk.cpu().numpy())
    try:
        assert (kst == 0).all()
        st = torch.full((n,), 9, dtype=torch.uint8, device=dev)
        ctx.ietf_verify_batch_keyed_dev(ks, idx, hh, g, c, s, st)
        torch.cuda.synchronize()
        assert int((st != 0).sum()) == 0
        s_bad = s.clone(); s_bad[::1024, 0] ^= 1
        ctx.ietf_verify_batch_keyed_dev(ks, idx, hh, g, c, s_bad, st)
        st2 = torch.full((n,), 9, dtype=torch.uint8, device=dev)
        ctx.ietf_verify_batch_dev(pk, hh, g, c, s_bad, st2)
        torch.cuda.synchronize()
        assert torch.equal(st, st2) and torch.equal(torch.nonzero(st).flatten(), torch.arange(0, n, 1024, device=dev))
    finally:
        ks.close()


def test_keyed_jubjub(synth):
    """The other suite: keys are validated by r*P = O (no 2-descent on a cofactor-8 curve), same statuses as the
    plain verifier."""
    from ark_ec_vrfs_amd import Context, JubJubSha512Tai
    cj = Context(0, suite=JubJubSha512Tai, test_blinding_base=True)
    co.set_suite(2)
    try:
        n, nk = 500, 9
        sk, _ = synth(nk, start=40)
        _, msg = synth(n, start=777)
        key = (np.arange(n, dtype=np.uint32) * 5) % nk
        got = cj.ietf_prove_batch(sk[key], msgs=msg, ad=b"jj")
        pks = np.stack([got["pk"][np.nonzero(key == k)[0][0]] for k in range(nk)])
        bad = pks.copy()
        bad[4] = np.frombuffer((5).to_bytes(32, "little"), np.uint8)
        ks, kst = cj.keyset_create(bad)
        try:
            want_key_ok = [co.point_decode(bad[k].tobytes(), subgroup=True) is not None for k in range(nk)]
            assert [s == 0 for s in kst] == want_key_ok and not want_key_ok[4]
            s2 = got["s"].copy(); s2[::50, 1] ^= 4
            st = cj.ietf_verify_batch_keyed(ks, key, got["input"], got["output"], got["c"], s2, ad=b"jj")
            plain = cj.ietf_verify_batch(bad[key], got["input"], got["output"], got["c"], s2, ad=b"jj")
            want = co.ietf_verify_batch(bad[key], got["input"], got["output"], got["c"], s2, b"jj", threads=NCPU)
            assert (plain == want).all()
            # the plain verifier does not test subgroup membership of pk; the key set does (checked decode):
            # proofs under the rejected key are InvalidData there, everything else agrees
            for i in range(n):
                assert st[i] == (2 if key[i] == 4 else want[i]), i
        finally:
            ks.close()
    finally:
        co.set_suite(1)
        cj.close()
