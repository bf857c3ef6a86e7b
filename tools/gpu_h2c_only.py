"""hash-to-curve alone (the try-and-increment search + the decode of the found candidate) on every try-and-increment suite, 2^20
messages: the workload tools/profile_lane_util.sh counts active lanes on.  usage (GPU box): python tools/gpu_h2c_only.py [libvrfhip variant .so]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ark_ec_vrfs_amd import _lib as _l0
if len(sys.argv) > 1:
    _l0.LIB_PATH = os.path.abspath(sys.argv[1])          # an A/B build of the library
import torch
from ark_ec_vrfs_amd import (BabyJubJubSha512Tai, BandersnatchSwSha512Tai, Context, Ed25519Sha512Tai, JubJubSha512Tai,
                             Secp256r1Sha256Tai, _lib)
dev = torch.device("cuda:0"); lib = _lib.load()
n = 1 << 20
msg = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev)
for suite in (JubJubSha512Tai, Ed25519Sha512Tai, BabyJubJubSha512Tai, Secp256r1Sha256Tai, BandersnatchSwSha512Tai):
    ctx = Context(0, suite=suite)
    out = torch.empty((n, ctx.point_bytes()), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    fn = lambda: _lib.check(lib.vrfhip_hash_to_curve_batch_dev(ctx.handle, n, msg.data_ptr(), None, 32, out.data_ptr(), st), "h2c")
    fn(); torch.cuda.synchronize()
    t = time.perf_counter(); fn(); fn(); torch.cuda.synchronize()
    print("%s hash-to-curve 2^20: %.2f ms" % (suite.__name__, (time.perf_counter() - t) / 2 * 1e3), flush=True)
    ctx.close()
