import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ark_ec_vrfs_amd import Context, _lib
ctx = Context(0); dev = torch.device('cuda:0'); lib = _lib.load()
st0 = torch.cuda.current_stream().cuda_stream
n = 1 << int(sys.argv[1])
seeds = torch.arange(n, dtype=torch.int64, device=dev).view(torch.uint8).reshape(n, 8)
a = torch.empty((n, 32), dtype=torch.uint8, device=dev); pk = torch.empty((n, 32), dtype=torch.uint8, device=dev)
_lib.check(lib.vrfhip_secret_from_seed_batch_dev(ctx.handle, n, seeds.data_ptr(), 8, a.data_ptr(), pk.data_ptr(), st0), "seed")
xy = torch.empty((n, 64), dtype=torch.uint8, device=dev); vst = torch.empty(n, dtype=torch.uint8, device=dev)
_lib.check(lib.vrfhip_point_validate_batch_dev(ctx.handle, n, pk.data_ptr(), xy.data_ptr(), vst.data_ptr(), st0), "validate")
out = torch.empty(32, dtype=torch.uint8, device=dev); st = torch.empty(1, dtype=torch.uint8, device=dev)
for rep in range(3):
    ctx.msm_dev(xy, a, out, None, st)
torch.cuda.synchronize()
