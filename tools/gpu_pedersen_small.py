"""Per-proof Pedersen verification by batch size (fused Straus halves up to 2^17), profiled call = separate launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ark_ec_vrfs_amd import Context
ctx = Context(0); dev = torch.device('cuda:0')
N = 1 << 17
seeds = np.arange(N, dtype=np.uint64).view(np.uint8).reshape(N, 8)
sk, _ = ctx.secret_from_seed_batch(seeds)
msg = np.random.default_rng(1).integers(0, 256, (N, 32), dtype=np.uint8)
p = ctx.pedersen_prove_batch(sk, msgs=msg)
d = {k: torch.from_numpy(p[k]).to(dev) for k in ("input", "output", "pk_com", "r", "ok", "s", "sb")}
st = torch.empty(N, dtype=torch.uint8, device=dev)
for lg in (12, 14, 16, 17):
    n = 1 << lg
    f = lambda: ctx.pedersen_verify_batch_dev(*(d[k][:n] for k in ("input", "output", "pk_com", "r", "ok", "s", "sb")), st[:n])
    res = []
    for prof in (False, True):
        ctx.profile(prof); f(); torch.cuda.synchronize(); b = 1e9
        for _ in range(4):
            torch.cuda.synchronize(); t = time.perf_counter(); f(); torch.cuda.synchronize(); b = min(b, time.perf_counter() - t)
        ctx.profile(False); ctx.profile_read(); res.append(b)
        assert int(st[:n].sum()) == 0
    print(f"pedersen verify 2^{lg}: fused {res[0]*1e3:.2f} ms, separate {res[1]*1e3:.2f} ms", flush=True)
