"""PCIe-inclusive timing of the host-buffer entry point (vrfhip_ietf_verify_batch) next to the
device-resident one, and device-resident throughput by batch size (slice planning)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ark_ec_vrfs_amd import Context, _lib
dev = torch.device('cuda:0'); lib = _lib.load(); ctx = Context(0); st0 = torch.cuda.current_stream().cuda_stream
n = 1 << 20
seeds = torch.arange(n, dtype=torch.int64, device=dev).view(torch.uint8).reshape(n, 8)
sk = torch.empty((n, 32), dtype=torch.uint8, device=dev)
_lib.check(lib.vrfhip_secret_from_seed_batch_dev(ctx.handle, n, seeds.data_ptr(), 8, sk.data_ptr(), 0, st0), "seed")
msg = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev)
mk = lambda: torch.empty((n, 32), dtype=torch.uint8, device=dev)
g, c, s, pk, hh = (mk() for _ in range(5)); pst = torch.empty(n, dtype=torch.uint8, device=dev)
ctx.ietf_prove_batch_dev(sk, msg, 32, g, c, s, pk, hh, pst); torch.cuda.synchronize()
st = torch.empty(n, dtype=torch.uint8, device=dev)
def best(fn, reps=4):
    b = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); b = min(b, time.perf_counter() - t)
    return b
for lg in (14, 16, 17, 18, 19, 20):
    m = 1 << lg
    t = best(lambda: ctx.ietf_verify_batch_dev(pk[:m], hh[:m], g[:m], c[:m], s[:m], st[:m]))
    assert int(st[:m].sum()) == 0
    print(f"device-resident 2^{lg}: {m/t:.3e}/s ({t*1e3:.2f} ms)", flush=True)
host = [x.cpu().numpy() for x in (pk, hh, g, c, s)]
th = best(lambda: ctx.ietf_verify_batch(*host))
r = ctx.ietf_verify_batch(*host); assert int(r.sum()) == 0
print(f"host buffers (pageable) 2^20: {n/th:.3e}/s ({th*1e3:.2f} ms)", flush=True)
pinned = [x.cpu().pin_memory() for x in (pk, hh, g, c, s)]
pn = [x.numpy() for x in pinned]
tp = best(lambda: ctx.ietf_verify_batch(*pn))
print(f"host buffers (pinned)   2^20: {n/tp:.3e}/s ({tp*1e3:.2f} ms)", flush=True)
bad = [a.copy() for a in host]; bad[4][::1024, 0] ^= 1
r = ctx.ietf_verify_batch(*bad)
assert (r[::1024] == 1).all() and int(r.sum()) == n // 1024
print("host path statuses ok (every 1024th tampered proof rejected)", flush=True)
