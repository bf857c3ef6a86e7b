"""One profiled pass of the batched Pedersen verifier at 2^N (argv[1], default 20): for rocprofv3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ark_ec_vrfs_amd import Context, _lib
dev = torch.device('cuda:0'); lib = _lib.load()
ctx = Context(0); st0 = torch.cuda.current_stream().cuda_stream
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 20)
seeds = torch.arange(n, dtype=torch.int64, device=dev).view(torch.uint8).reshape(n, 8)
sk = torch.empty((n, 32), dtype=torch.uint8, device=dev)
_lib.check(lib.vrfhip_secret_from_seed_batch_dev(ctx.handle, n, seeds.data_ptr(), 8, sk.data_ptr(), None, st0), "seed")
msg = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev)
mk = lambda: torch.empty((n, 32), dtype=torch.uint8, device=dev)
g, pc, r, ok, s, sb, hh = (mk() for _ in range(7))
pst = torch.empty(n, dtype=torch.uint8, device=dev)
ctx.pedersen_prove_batch_dev(sk, msg, 32, g, pc, r, ok, s, sb, None, hh, pst)
st = torch.empty(n, dtype=torch.uint8, device=dev); flag = torch.empty(1, dtype=torch.uint8, device=dev)
for _ in range(3):
    ctx.pedersen_verify_batch_rlc_dev(hh, g, pc, r, ok, s, sb, st, flag, os.urandom(32))
torch.cuda.synchronize()
assert int(flag[0]) == 0
