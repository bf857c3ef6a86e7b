"""One profiled pass of BASELINE.json config 4's shape on one GPU: 2^20 Pedersen proofs on JubJub, per-proof
verification and the batched verifier (for rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ark_ec_vrfs_amd import Context, JubJubSha512Tai, _lib
ctx = Context(0, suite=JubJubSha512Tai, test_blinding_base=True); dev = torch.device('cuda:0'); lib = _lib.load()
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 20)
seeds = torch.arange(n, dtype=torch.int64, device=dev).view(torch.uint8).reshape(n, 8)
sk = torch.empty((n, 32), dtype=torch.uint8, device=dev)
_lib.check(lib.vrfhip_secret_from_seed_batch_dev(ctx.handle, n, seeds.data_ptr(), 8, sk.data_ptr(), None, torch.cuda.current_stream().cuda_stream), "seed")
msg = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev)
mk = lambda: torch.empty((n, 32), dtype=torch.uint8, device=dev)
out, pkc, r, ok, s, sb, hh = (mk() for _ in range(7))
st = torch.empty(n, dtype=torch.uint8, device=dev); ff = torch.zeros(1, dtype=torch.uint8, device=dev)
for _ in range(2):
    ctx.pedersen_prove_batch_dev(sk, msg, 32, out, pkc, r, ok, s, sb, None, hh, st)
    ctx.pedersen_verify_batch_dev(hh, out, pkc, r, ok, s, sb, st)
    ctx.pedersen_verify_batch_rlc_dev(hh, out, pkc, r, ok, s, sb, st, ff, bytes(range(32)))
torch.cuda.synchronize()
assert int(st.sum()) == 0 and int(ff[0]) == 0
