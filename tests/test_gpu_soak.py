"""GPU (-m gpu): bounded parity soak at the batch sizes that pick the other kernel variants.

`lanes_k` (csrc/kernels.h) runs up to K proofs per lane in the inversion-sharing stages: the PROVERS K = 1 / 2 / 4 / 8 for
batches below 2^18 / 2^19 / 2^20 / from 2^20, the VERIFIERS K = 1 / 2 (api.hip VERIFY_K_POLICY, since inversions became
cheap).  The remaining suites cover K = 1 (<= 4096 items) and the exact 2^20.  Here: n = 2^18 + 5 (K = 2), 2^19 + 3
(provers K = 4) and 2^20 + 7 inside ONE chunk (reserve(2^21): provers K = 8 with a ragged last lane), each with
  * proof bytes against the C oracle on a strided sample plus the whole tail (where the ragged lanes are),
  * statuses of a tampered batch against the oracle on the same sample, and the exact set of rejected items,
and the exact 2^16 prove batch of BASELINE.json configs[1] byte for byte.  The same for Pedersen on JubJub.
The oracle is the checker only."""
import hashlib
import os

import numpy as np
import pytest

from oracle import c_oracle as co

pytestmark = pytest.mark.gpu
NCPU = min(16, os.cpu_count() or 1)


def _msgs(n, lo=0):
    out = np.empty((n, 32), np.uint8)
    for k in range(n):
        out[k] = np.frombuffer(hashlib.sha512(b"vrfhip-msg" + int(lo + k).to_bytes(8, "little")).digest()[:32], np.uint8)
    return out


def _sample(n, stride):
    """strided sample + the last 4096 items (ragged tail lanes) + the first 64"""
    idx = np.unique(np.concatenate([np.arange(0, n, stride), np.arange(max(0, n - 4096), n), np.arange(0, min(64, n))]))
    return idx


def _seeded_sk(torch, lib, _lib, ctx, n, dev, lo=0):
    seeds = (torch.arange(n, dtype=torch.int64, device=dev) + lo).view(torch.uint8).reshape(n, 8)
    sk = torch.empty((n, 32), dtype=torch.uint8, device=dev)
    st0 = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.vrfhip_secret_from_seed_batch_dev(ctx.handle, n, seeds.data_ptr(), 8, sk.data_ptr(), None, st0), "seed")
    return sk


@pytest.mark.parametrize("n,reserve", [((1 << 18) + 5, 0), ((1 << 19) + 3, 0), ((1 << 20) + 7, 1 << 21)])
def test_ietf_prove_and_verify_soak_all_lane_variants(n, reserve):
    import torch
    from ark_ec_vrfs_amd import Context, _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    ctx = Context(0)
    try:
        if reserve:
            ctx.reserve(reserve)
        assert lib.vrfhip_debug_proofs_per_lane(n if not reserve else min(n, reserve)) == {(1 << 18) + 5: 2, (1 << 19) + 3: 2, (1 << 20) + 7: 2}[n]     # verifiers: K <= 2 (api.hip VERIFY_K_POLICY); the provers reach 4 and 8
        sk = _seeded_sk(torch, lib, _lib, ctx, n, dev)
        msg_h = _msgs(n)
        msg = torch.from_numpy(msg_h).to(dev)
        mk = lambda: torch.empty((n, 32), dtype=torch.uint8, device=dev)
        out, c, s, pk, hh = mk(), mk(), mk(), mk(), mk()
        pst = torch.empty(n, dtype=torch.uint8, device=dev)
        ad = torch.from_numpy(np.frombuffer(b"soak-ad", np.uint8).copy()).to(dev)
        ctx.ietf_prove_batch_dev(sk, msg, 32, out, c, s, pk, hh, pst, ad=ad, ad_len=7)
        torch.cuda.synchronize()
        assert int(pst.sum()) == 0
        idx = _sample(n, 13)                       # >= 2^14 items
        assert idx.size >= (1 << 14)
        ti = torch.from_numpy(idx).to(dev)
        ref = co.ietf_prove_batch(sk[ti].cpu().numpy(), msgs=msg_h[idx], ad=b"soak-ad", threads=NCPU)
        for name, t in (("output", out), ("c", c), ("s", s), ("pk", pk), ("input", hh)):
            assert (t[ti].cpu().numpy() == ref[name]).all(), name
        # tampered batch: bits of s, c, pk; swapped outputs; a scalar out of range
        g = torch.Generator(device=dev); g.manual_seed(n)
        kind = torch.randint(0, 8, (n,), device=dev, generator=g)
        s2, c2, pk2, out2 = s.clone(), c.clone(), pk.clone(), out.clone()
        s2[kind == 1, 5] ^= 2
        c2[kind == 2, 17] ^= 64
        pk2[kind == 3, 9] ^= 1
        sw = torch.nonzero(kind == 4).flatten()
        out2[sw] = out[(sw + 1) % n]
        s2[kind == 5, 31] |= 0x80                  # >= r: InvalidData
        vs = torch.full((n,), 9, dtype=torch.uint8, device=dev)
        ctx.ietf_verify_batch_dev(pk2, hh, out2, c2, s2, vs, ad=ad, ad_len=7)
        torch.cuda.synchronize()
        want = co.ietf_verify_batch(pk2[ti].cpu().numpy(), hh[ti].cpu().numpy(), out2[ti].cpu().numpy(),
                                    c2[ti].cpu().numpy(), s2[ti].cpu().numpy(), b"soak-ad", threads=NCPU)
        assert (vs[ti].cpu().numpy() == want).all()
        # size-independent: untouched items verify, every tampered item is rejected, s >= r is InvalidData
        vs_h, kind_h = vs.cpu().numpy(), kind.cpu().numpy()
        assert (vs_h[(kind_h == 0) | (kind_h >= 6)] == 0).all()
        assert (vs_h[(kind_h >= 1) & (kind_h <= 5)] != 0).all() and (vs_h[kind_h == 5] == 2).all()
    finally:
        ctx.close()


def test_ietf_prove_exact_2_16_batch_bytes_equal_oracle(ctx):
    """BASELINE.json configs[1]: the whole 2^16 batch, every byte."""
    import torch
    from ark_ec_vrfs_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    n = 1 << 16
    sk = _seeded_sk(torch, lib, _lib, ctx, n, dev)
    msg_h = _msgs(n)
    msg = torch.from_numpy(msg_h).to(dev)
    mk = lambda: torch.empty((n, 32), dtype=torch.uint8, device=dev)
    out, c, s, pk, hh = mk(), mk(), mk(), mk(), mk()
    pst = torch.empty(n, dtype=torch.uint8, device=dev)
    ctx.ietf_prove_batch_dev(sk, msg, 32, out, c, s, pk, hh, pst)
    torch.cuda.synchronize()
    ref = co.ietf_prove_batch(sk.cpu().numpy(), msgs=msg_h, ad=b"", threads=NCPU)
    for name, t in (("output", out), ("c", c), ("s", s), ("pk", pk), ("input", hh)):
        assert (t.cpu().numpy() == ref[name]).all(), name
    assert int(pst.sum()) == 0


@pytest.mark.parametrize("n", [(1 << 18) + 5, (1 << 19) + 3])
def test_pedersen_jubjub_soak_lane_variants(n):
    import torch
    from ark_ec_vrfs_amd import Context, JubJubSha512Tai, _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    cj = Context(0, suite=JubJubSha512Tai, test_blinding_base=True)
    co.set_suite(2)
    try:
        sk = _seeded_sk(torch, lib, _lib, cj, n, dev)
        msg_h = _msgs(n)
        msg = torch.from_numpy(msg_h).to(dev)
        mk = lambda: torch.empty((n, 32), dtype=torch.uint8, device=dev)
        g_, pc, r, ok, ss, sb, hj, bl = (mk() for _ in range(8))
        pst = torch.empty(n, dtype=torch.uint8, device=dev)
        cj.pedersen_prove_batch_dev(sk, msg, 32, g_, pc, r, ok, ss, sb, bl, hj, pst)
        torch.cuda.synchronize()
        assert int(pst.sum()) == 0
        idx = _sample(n, 97)
        ti = torch.from_numpy(idx).to(dev)
        ref = co.pedersen_prove_batch(sk[ti].cpu().numpy(), msgs=msg_h[idx], ad=b"", threads=NCPU)
        for name, t in (("output", g_), ("pk_com", pc), ("r", r), ("ok", ok), ("s", ss), ("sb", sb), ("blinding", bl), ("input", hj)):
            assert (t[ti].cpu().numpy() == ref[name]).all(), name
        # tampered batch through the per-proof verifier and the batched verifier (decode runs K proofs per lane)
        gen = torch.Generator(device=dev); gen.manual_seed(n)
        kind = torch.randint(0, 6, (n,), device=dev, generator=gen)
        ss2, ok2, pc2 = ss.clone(), ok.clone(), pc.clone()
        ss2[kind == 1, 3] ^= 1
        sw = torch.nonzero(kind == 2).flatten()
        ok2[sw] = ok[(sw + 1) % n]
        pc2[kind == 3, 31] = 0xff                   # y >= q: InvalidData
        vs = torch.full((n,), 9, dtype=torch.uint8, device=dev)
        cj.pedersen_verify_batch_dev(hj, g_, pc2, r, ok2, ss2, sb, vs)
        torch.cuda.synchronize()
        host = lambda t: t[ti].cpu().numpy()
        want = co.pedersen_verify_batch(host(hj), host(g_), host(pc2), host(r), host(ok2), host(ss2), host(sb), b"", threads=NCPU)
        assert (vs[ti].cpu().numpy() == want).all()
        kind_h, vs_h = kind.cpu().numpy(), vs.cpu().numpy()
        assert (vs_h[(kind_h == 0) | (kind_h >= 4)] == 0).all() and (vs_h[(kind_h == 1) | (kind_h == 2)] == 1).all()
        assert (vs_h[kind_h == 3] == 2).all()
        # batched verifier on the same tampered batch: InvalidData items flagged, batch equation fails
        flag = torch.zeros(1, dtype=torch.uint8, device=dev)
        bs = torch.full((n,), 9, dtype=torch.uint8, device=dev)
        cj.pedersen_verify_batch_rlc_dev(hj, g_, pc2, r, ok2, ss2, sb, bs, flag, os.urandom(32))
        torch.cuda.synchronize()
        assert int(flag[0]) == 1 and torch.equal(bs == 2, vs == 2) and int((bs == 1).sum()) == 0
        # and on the untouched batch: passes
        cj.pedersen_verify_batch_rlc_dev(hj, g_, pc, r, ok, ss, sb, bs, flag, os.urandom(32))
        torch.cuda.synchronize()
        assert int(flag[0]) == 0 and int(bs.sum()) == 0
    finally:
        co.set_suite(1)
        cj.close()
