"""JubJub (try-and-increment suite) IETF and Pedersen prove timing with per-stage profile."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ark_ec_vrfs_amd import Context, JubJubSha512Tai, _lib
ctx = Context(0, suite=JubJubSha512Tai, test_blinding_base=True); dev = torch.device('cuda:0'); lib = _lib.load()
for logn in (16, 20):
    n = 1 << logn
    seeds = torch.arange(n, dtype=torch.int64, device=dev).view(torch.uint8).reshape(n, 8)
    sk = torch.empty((n, 32), dtype=torch.uint8, device=dev)
    _lib.check(lib.vrfhip_secret_from_seed_batch_dev(ctx.handle, n, seeds.data_ptr(), 8, sk.data_ptr(), None, torch.cuda.current_stream().cuda_stream), "seed")
    msg = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev)
    mk = lambda: torch.empty((n, 32), dtype=torch.uint8, device=dev)
    out, pkc, r, ok, s, sb, hh, c = (mk() for _ in range(8))
    st = torch.empty(n, dtype=torch.uint8, device=dev)
    for name, fn in (("ietf", lambda: ctx.ietf_prove_batch_dev(sk, msg, 32, out, c, s, pkc, hh, st)),
                     ("pedersen", lambda: ctx.pedersen_prove_batch_dev(sk, msg, 32, out, pkc, r, ok, s, sb, None, hh, st))):
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t)
        assert int(st.sum()) == 0
        ctx.profile(True); fn(); torch.cuda.synchronize(); ctx.profile(False)
        ms, g = ctx.profile_read()
        print(f"jubjub {name} prove n=2^{logn}: {n/best:.3e}/s ({best*1e3:.2f} ms) stages prepare={ms[0]:.2f} mul={ms[1]:.2f} finish={ms[3]:.2f}", flush=True)
