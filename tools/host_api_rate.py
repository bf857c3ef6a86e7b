"""PCIe-inclusive rate of the host-pointer entry points: the 2^20 headline batch handed over as
host numpy arrays -- vrfhip_ietf_verify_batch copies 160 MiB in, runs the same kernels, copies 1 MiB of statuses out.
usage (GPU box): python tools/host_api_rate.py"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ark_ec_vrfs_amd import Context, PinnedBuffer, _lib
n = 1 << 20
ctx = Context(0)
lib = _lib.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
seeds = torch.arange(n, dtype=torch.int64, device=dev).view(torch.uint8).reshape(n, 8)
sk = torch.empty((n, 32), dtype=torch.uint8, device=dev)
_lib.check(lib.vrfhip_secret_from_seed_batch_dev(ctx.handle, n, seeds.data_ptr(), 8, sk.data_ptr(), None, st), "seed")
msg = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev)
mk = lambda: torch.empty((n, 32), dtype=torch.uint8, device=dev)
g, c, s, pk, hh = mk(), mk(), mk(), mk(), mk()
pst = torch.empty(n, dtype=torch.uint8, device=dev)
ctx.ietf_prove_batch_dev(sk, msg, 32, g, c, s, pk, hh, pst)
torch.cuda.synchronize()
host = [t.cpu().numpy() for t in (pk, hh, g, c, s)]
pins = [PinnedBuffer((n, 32)) for _ in range(5)]
for p, h in zip(pins, host):
    p.array[:] = h
pinned = [p.array for p in pins]
status = torch.empty(n, dtype=torch.uint8, device=dev)
import threading
others = [Context(0) for _ in range(2)]          # the _multi shape: three contexts on one device, a third of the batch each


def three_contexts():
    cs = [ctx] + others
    th = []
    for k, cx in enumerate(cs):
        lo, hi = n * k // 3, n * (k + 1) // 3
        th.append(threading.Thread(target=lambda cx=cx, lo=lo, hi=hi: cx.ietf_verify_batch(*[h[lo:hi] for h in host], ad=b"")))
    for t in th:
        t.start()
    for t in th:
        t.join()


from ark_ec_vrfs_amd import ietf_verify_batch_multi  # noqa: E402
extra8 = [Context(0) for _ in range(5)]         # 8 contexts in ONE process: what the Rust crate's _multi call does on an 8-GPU node
ctx8 = [ctx] + others + extra8


def eight_contexts_multi():
    st8 = ietf_verify_batch_multi(ctx8, *host, ad=b"")
    assert not st8.any()


for name, fn in (("device pointers", lambda: (ctx.ietf_verify_batch_dev(pk, hh, g, c, s, status), torch.cuda.synchronize())),
                 ("host pointers (pageable numpy)", lambda: ctx.ietf_verify_batch(*host, ad=b"")),
                 ("host pointers (vrfhip_host_alloc)", lambda: ctx.ietf_verify_batch(*pinned, ad=b"")),
                 ("3 contexts x 1/3 batch, pageable", three_contexts),
                 ("vrfhip_ietf_verify_batch_multi, 8 contexts (one device), pageable", eight_contexts_multi)):
    fn()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); r = fn(); best = min(best, time.perf_counter() - t0)
    print("%-66s %.2f ms per 2^20 verifies = %.3e verifies/s" % (name, best * 1e3, n / best), flush=True)
