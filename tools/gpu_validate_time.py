"""Timing of vrfhip_point_validate_batch_dev (checked decode: on curve + prime-order subgroup)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ark_ec_vrfs_amd import Context, JubJubSha512Tai, BandersnatchSha512Ell2, _lib
dev = torch.device('cuda:0'); lib = _lib.load()
for suite in (BandersnatchSha512Ell2, JubJubSha512Tai):
    ctx = Context(0, suite=suite, test_blinding_base=True); st0 = torch.cuda.current_stream().cuda_stream
    n = 1 << 20
    seeds = torch.arange(n, dtype=torch.int64, device=dev).view(torch.uint8).reshape(n, 8)
    sk = torch.empty((n, 32), dtype=torch.uint8, device=dev); pk = torch.empty((n, 32), dtype=torch.uint8, device=dev)
    _lib.check(lib.vrfhip_secret_from_seed_batch_dev(ctx.handle, n, seeds.data_ptr(), 8, sk.data_ptr(), pk.data_ptr(), st0), "seed")
    xy = torch.empty((n, 64), dtype=torch.uint8, device=dev); st = torch.empty(n, dtype=torch.uint8, device=dev)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        _lib.check(lib.vrfhip_point_validate_batch_dev(ctx.handle, n, pk.data_ptr(), xy.data_ptr(), st.data_ptr(), st0), "validate")
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    assert int(st.sum()) == 0
    print(f"{suite.__name__} point_validate n=2^20: {best*1e3:.2f} ms ({n/best:.3e} points/s)", flush=True)
    ctx.close()
