"""Timing of the batched (single-MSM) Pedersen verifier next to the per-proof one, device-resident inputs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ark_ec_vrfs_amd import Context, JubJubSha512Tai, BandersnatchSha512Ell2, _lib
dev = torch.device('cuda:0'); lib = _lib.load()
logs = [int(x) for x in sys.argv[1:]] or [16, 18, 20]
for suite in (BandersnatchSha512Ell2, JubJubSha512Tai):
    ctx = Context(0, suite=suite, test_blinding_base=True)
    st0 = torch.cuda.current_stream().cuda_stream
    for logn in logs:
        n = 1 << logn
        seeds = torch.arange(n, dtype=torch.int64, device=dev).view(torch.uint8).reshape(n, 8)
        sk = torch.empty((n, 32), dtype=torch.uint8, device=dev)
        _lib.check(lib.vrfhip_secret_from_seed_batch_dev(ctx.handle, n, seeds.data_ptr(), 8, sk.data_ptr(), None, st0), "seed")
        msg = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev)
        mk = lambda: torch.empty((n, 32), dtype=torch.uint8, device=dev)
        g, pc, r, ok, s, sb, hh = (mk() for _ in range(7))
        pst = torch.empty(n, dtype=torch.uint8, device=dev)
        ctx.pedersen_prove_batch_dev(sk, msg, 32, g, pc, r, ok, s, sb, None, hh, pst)
        st = torch.empty(n, dtype=torch.uint8, device=dev); flag = torch.empty(1, dtype=torch.uint8, device=dev)
        seed = os.urandom(32)
        def best(fn, reps=4):
            b = 1e9
            for _ in range(reps):
                torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize()
                b = min(b, time.perf_counter() - t)
            return b
        t_item = best(lambda: ctx.pedersen_verify_batch_dev(hh, g, pc, r, ok, s, sb, st))
        assert int(st.max()) == 0
        t_rlc = best(lambda: ctx.pedersen_verify_batch_rlc_dev(hh, g, pc, r, ok, s, sb, st, flag, seed))
        assert int(flag[0]) == 0 and int(st.max()) == 0
        xy = []
        for src in (hh, g, pc, r, ok):
            t = torch.empty((n, 64), dtype=torch.uint8, device=dev)
            _lib.check(lib.vrfhip_point_validate_batch_dev(ctx.handle, n, src.data_ptr(), t.data_ptr(), st.data_ptr(), st0), "validate")
            xy.append(t)
        t_aff = best(lambda: ctx.pedersen_verify_batch_rlc_dev(*xy, s, sb, st, flag, seed, affine=True))
        assert int(flag[0]) == 0 and int(st.max()) == 0
        ctx.profile(True)
        ctx.pedersen_verify_batch_rlc_dev(hh, g, pc, r, ok, s, sb, st, flag, seed)
        torch.cuda.synchronize(); ctx.profile(False)
        ms, groups = ctx.profile_read()
        print(f"{suite.__name__} n=2^{logn}: per-proof {n/t_item:.3e}/s ({t_item*1e3:.2f} ms)  rlc {n/t_rlc:.3e}/s ({t_rlc*1e3:.2f} ms)  rlc-affine {n/t_aff:.3e}/s ({t_aff*1e3:.2f} ms)"
              f"  stages decode={ms[0]:.2f} buckets={ms[1]:.2f} final={ms[2]:.2f} ms", flush=True)
    ctx.close()
