import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ark_ec_vrfs_amd import Context, _lib
ctx = Context(0); dev = torch.device('cuda:0'); lib = _lib.load()
for logn in (16, 20):
    n = 1 << logn
    seeds = torch.arange(n, dtype=torch.int64, device=dev).view(torch.uint8).reshape(n, 8)
    sk = torch.empty((n, 32), dtype=torch.uint8, device=dev)
    _lib.check(lib.vrfhip_secret_from_seed_batch_dev(ctx.handle, n, seeds.data_ptr(), 8, sk.data_ptr(), None, torch.cuda.current_stream().cuda_stream), "seed")
    msg = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev)
    mk = lambda: torch.empty((n, 32), dtype=torch.uint8, device=dev)
    out, pkc, r, ok, s, sb, hh = mk(), mk(), mk(), mk(), mk(), mk(), mk()
    st = torch.empty(n, dtype=torch.uint8, device=dev)
    for rep in range(2):
        torch.cuda.synchronize(); t = time.time()
        ctx.pedersen_prove_batch_dev(sk, msg, 32, out, pkc, r, ok, s, sb, None, hh, st)
        torch.cuda.synchronize(); tp = time.time() - t
    assert int(st.sum()) == 0
    ctx.profile(True)
    for rep in range(2):
        torch.cuda.synchronize(); t = time.time()
        ctx.pedersen_verify_batch_dev(hh, out, pkc, r, ok, s, sb, st)
        torch.cuda.synchronize(); tv = time.time() - t
    ctx.profile(False)
    ms, g = ctx.profile_read()
    assert int(st.sum()) == 0, int(st.sum())
    print(f"pedersen n=2^{logn}: prove {n/tp:.3e}/s verify {n/tv:.3e}/s stages(ms/launch) {[m/g for m in ms]}", flush=True)
