"""Sweeps of two launch-shape knobs (vrfhip_debug_set): proofs per lane in the provers' prepare / finish stages, and point
groups per window of the secp256r1 MSM inside the batched Pedersen verifier.  usage (GPU box): python tools/gpu_knob_sweep.py [p256]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ark_ec_vrfs_amd import BandersnatchSha512Ell2, Context, JubJubSha512Tai, Secp256r1Sha256Tai, _lib
dev = torch.device("cuda:0"); lib = _lib.load()
n = 1 << 20


def best(fn, reps=4):
    b = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); b = min(b, time.perf_counter() - t)
    return b


def batch(ctx, suite):
    st0 = torch.cuda.current_stream().cuda_stream
    seeds = torch.arange(n, dtype=torch.int64, device=dev).view(torch.uint8).reshape(n, 8)
    sk = torch.empty((n, 32), dtype=torch.uint8, device=dev)
    _lib.check(lib.vrfhip_secret_from_seed_batch_dev(ctx.handle, n, seeds.data_ptr(), 8, sk.data_ptr(), None, st0), "seed")
    msg = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev)
    return sk, msg


for suite in (() if "p256" in sys.argv[1:] else (BandersnatchSha512Ell2, JubJubSha512Tai)):
    ctx = Context(0, suite=suite, test_blinding_base=True)
    sk, msg = batch(ctx, suite)
    mk = lambda: torch.empty((n, 32), dtype=torch.uint8, device=dev)
    g, c, s, pk, hh = (mk() for _ in range(5)); pst = torch.empty(n, dtype=torch.uint8, device=dev)
    for k in (0, 8, 4, 2, 1):
        ctx.debug_set(4, k)
        fn = lambda: ctx.ietf_prove_batch_dev(sk, msg, 32, g, c, s, pk, hh, pst)
        fn(); torch.cuda.synchronize()
        ctx.profile(True); t = best(fn); ctx.profile(False)
        ms, grp = ctx.profile_read()
        assert int(pst.sum()) == 0
        print("%s prove 2^20, proofs per lane %d: %.2f ms  stages %s" % (suite.__name__, k, t * 1e3, [round(x / grp, 2) for x in ms]), flush=True)
    ctx.close()

cp = Context(0, suite=Secp256r1Sha256Tai, test_blinding_base=True)
sk, msg = batch(cp, Secp256r1Sha256Tai)
mk = lambda w: torch.empty((n, w), dtype=torch.uint8, device=dev)
g, pc, rr, okp, hh = (mk(33) for _ in range(5)); s_, sbb = mk(32), mk(32)
st = torch.empty(n, dtype=torch.uint8, device=dev); flag = torch.empty(1, dtype=torch.uint8, device=dev)
cp.pedersen_prove_batch_dev(sk, msg, 32, g, pc, rr, okp, s_, sbb, None, hh, st)
torch.cuda.synchronize()
assert int(st.sum()) == 0
seed = os.urandom(32)
for groups in (0, 11, 12, 16, 20, 24, 25, 26, 32, 40, 48, 50, 52, 64, 72):
    cp.debug_set(5, groups)
    fn = lambda: cp.pedersen_verify_batch_rlc_dev(hh, g, pc, rr, okp, s_, sbb, st, flag, seed)
    fn(); torch.cuda.synchronize()
    cp.profile(True); t = best(fn); cp.profile(False)
    ms, grp = cp.profile_read()
    assert int(flag[0]) == 0 and int(st.sum()) == 0
    print("secp256r1 batched Pedersen verify 2^20, MSM groups %d: %.2f ms  stages %s" % (groups, t * 1e3, [round(x / grp, 2) for x in ms]), flush=True)
cp.close()
