"""Timing of keyed IETF verification (1024 keys, 2^20 proofs) next to the plain verifier."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ark_ec_vrfs_amd import Context, _lib
dev = torch.device('cuda:0'); lib = _lib.load(); ctx = Context(0); st0 = torch.cuda.current_stream().cuda_stream
n, nk = 1 << 20, 1024
seeds = torch.arange(nk, dtype=torch.int64, device=dev).view(torch.uint8).reshape(nk, 8)
ksk = torch.empty((nk, 32), dtype=torch.uint8, device=dev); kpk = torch.empty((nk, 32), dtype=torch.uint8, device=dev)
_lib.check(lib.vrfhip_secret_from_seed_batch_dev(ctx.handle, nk, seeds.data_ptr(), 8, ksk.data_ptr(), kpk.data_ptr(), st0), "seed")
idx = (torch.arange(n, device=dev) * 2654435761 % nk).to(torch.int32)
sk = ksk[idx.long()].contiguous()
msg = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev)
mk = lambda: torch.empty((n, 32), dtype=torch.uint8, device=dev)
g, c, s, pk, hh = (mk() for _ in range(5)); pst = torch.empty(n, dtype=torch.uint8, device=dev)
ctx.ietf_prove_batch_dev(sk, msg, 32, g, c, s, pk, hh, pst); torch.cuda.synchronize()
t = time.perf_counter(); ks, kst = ctx.keyset_create(kpk.cpu().numpy()); tb = time.perf_counter() - t
assert (kst == 0).all()
st = torch.empty(n, dtype=torch.uint8, device=dev)
def best(fn, reps=4):
    b = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); b = min(b, time.perf_counter() - t)
    return b
tk = best(lambda: ctx.ietf_verify_batch_keyed_dev(ks, idx, hh, g, c, s, st)); assert int(st.sum()) == 0
tp = best(lambda: ctx.ietf_verify_batch_dev(pk, hh, g, c, s, st)); assert int(st.sum()) == 0
ctx.profile(True); ctx.ietf_verify_batch_keyed_dev(ks, idx, hh, g, c, s, st); torch.cuda.synchronize(); ctx.profile(False)
ms, _ = ctx.profile_read()
print(f"keyset build {nk} keys: {tb*1e3:.1f} ms ({ks.bytes()/1e6:.0f} MB); keyed verify 2^20: {n/tk:.3e}/s ({tk*1e3:.2f} ms) "
      f"stages decode={ms[0]:.2f} V={ms[1]:.2f} U={ms[2]:.2f} finish={ms[3]:.2f}; plain {n/tp:.3e}/s ({tp*1e3:.2f} ms)", flush=True)
ks.close()
