"""Cost of VRFHIP_FLAG_CT_TABLES (the provers' per-proof window lookups read all eight entries): IETF prove 2^20 with and
without the flag, stage times from the context's profile events.  usage (GPU box): python tools/gpu_ct_cost.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ark_ec_vrfs_amd import (BandersnatchSha512Ell2, Context, Ed25519Sha512Tai, JubJubSha512Tai, Secp256r1Sha256Tai, _lib)
dev = torch.device("cuda:0"); lib = _lib.load()
n = 1 << 20
for suite in (BandersnatchSha512Ell2, JubJubSha512Tai, Ed25519Sha512Tai, Secp256r1Sha256Tai):
    ctx = Context(0, suite=suite)
    st0 = torch.cuda.current_stream().cuda_stream
    pw = ctx.point_bytes()
    seeds = torch.arange(n, dtype=torch.int64, device=dev).view(torch.uint8).reshape(n, 8)
    sk = torch.empty((n, 32), dtype=torch.uint8, device=dev)
    _lib.check(lib.vrfhip_secret_from_seed_batch_dev(ctx.handle, n, seeds.data_ptr(), 8, sk.data_ptr(), None, st0), "seed")
    msg = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev)
    mk = lambda w: torch.empty((n, w), dtype=torch.uint8, device=dev)
    g, c, s, pk, hh = mk(pw), mk(32), mk(32), mk(pw), mk(pw)
    pst = torch.empty(n, dtype=torch.uint8, device=dev)
    ref = None
    for flag in (0, ctx.CT_TABLES):
        ctx.set_flags(flag)
        fn = lambda: ctx.ietf_prove_batch_dev(sk, msg, 32, g, c, s, pk, hh, pst)
        fn(); torch.cuda.synchronize()
        ctx.profile(True)
        best = 1e9
        for _ in range(4):
            torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t)
        ctx.profile(False)
        ms, groups = ctx.profile_read()
        assert int(pst.sum()) == 0
        out = (g.clone(), c.clone(), s.clone())
        if ref is None:
            ref = out
        else:
            assert all(bool((a == b).all()) for a, b in zip(ref, out)), "proof bytes changed with the flag"
        print("%-24s ct=%d: prove 2^20 %.2f ms (%.3e proofs/s); mul stage %.2f ms" % (suite.__name__, 1 if flag else 0, best * 1e3,
              n / best, ms[1] / max(groups, 1)), flush=True)
    ctx.close()
