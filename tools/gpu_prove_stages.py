"""Stage times (prepare / mul / finish) of IETF prove by batch size: where a 2^16 batch spends its latency."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ark_ec_vrfs_amd import Context, _lib
dev = torch.device('cuda:0'); lib = _lib.load(); ctx = Context(0); st0 = torch.cuda.current_stream().cuda_stream
N = 1 << 20
seeds = torch.arange(N, dtype=torch.int64, device=dev).view(torch.uint8).reshape(N, 8)
sk = torch.empty((N, 32), dtype=torch.uint8, device=dev)
_lib.check(lib.vrfhip_secret_from_seed_batch_dev(ctx.handle, N, seeds.data_ptr(), 8, sk.data_ptr(), 0, st0), "seed")
msg = torch.randint(0, 256, (N, 32), dtype=torch.uint8, device=dev)
mk = lambda: torch.empty((N, 32), dtype=torch.uint8, device=dev)
g, c, s, pk, hh = (mk() for _ in range(5)); pst = torch.empty(N, dtype=torch.uint8, device=dev)
for lg in (12, 14, 16, 17, 18, 20):
    n = 1 << lg
    f = lambda: ctx.ietf_prove_batch_dev(sk[:n], msg[:n], 32, g[:n], c[:n], s[:n], pk[:n], hh[:n], pst[:n])
    f(); torch.cuda.synchronize()
    b = 1e9
    for _ in range(4):
        torch.cuda.synchronize(); t = time.perf_counter(); f(); torch.cuda.synchronize(); b = min(b, time.perf_counter() - t)
    ctx.profile(True); f(); torch.cuda.synchronize(); ctx.profile(False)
    ms, _ = ctx.profile_read()
    print(f"prove 2^{lg}: {b*1e3:.2f} ms ({n/b:.3e}/s) stages " + " ".join(f"{x:.2f}" for x in ms), flush=True)
