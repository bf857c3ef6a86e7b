"""A/B of the batch-digest kernels' register cap (k_digest.hip DIGEST_MINW): the batched Pedersen verifier at 2^20 on JubJub
(32-byte rows: the word path of the leaf hash) and secp256r1 (33-byte rows: the byte path), wall time and what the stage
events do not cover (digest + launch gaps).  usage (GPU box): python tools/gpu_digest_ab.py <libvrfhip variant .so>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ark_ec_vrfs_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
from ark_ec_vrfs_amd import Context, JubJubSha512Tai, Secp256r1Sha256Tai
dev = torch.device("cuda:0"); lib = _lib.load()
n = 1 << 20
for suite in (JubJubSha512Tai, Secp256r1Sha256Tai):
    ctx = Context(0, suite=suite, test_blinding_base=True)
    pw = ctx.point_bytes()
    st0 = torch.cuda.current_stream().cuda_stream
    seeds = torch.arange(n, dtype=torch.int64, device=dev).view(torch.uint8).reshape(n, 8)
    sk = torch.empty((n, 32), dtype=torch.uint8, device=dev)
    _lib.check(lib.vrfhip_secret_from_seed_batch_dev(ctx.handle, n, seeds.data_ptr(), 8, sk.data_ptr(), None, st0), "seed")
    msg = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev)
    mkp = lambda: torch.empty((n, pw), dtype=torch.uint8, device=dev)
    mk = lambda: torch.empty((n, 32), dtype=torch.uint8, device=dev)
    g, pc, r, ok, hh = (mkp() for _ in range(5)); s, sb = mk(), mk()
    st = torch.empty(n, dtype=torch.uint8, device=dev); flag = torch.empty(1, dtype=torch.uint8, device=dev)
    ctx.pedersen_prove_batch_dev(sk, msg, 32, g, pc, r, ok, s, sb, None, hh, st)
    seed = os.urandom(32)
    fn = lambda: ctx.pedersen_verify_batch_rlc_dev(hh, g, pc, r, ok, s, sb, st, flag, seed)
    fn(); torch.cuda.synchronize()
    assert int(flag[0]) == 0 and int(st.max()) == 0
    best = 1e9
    for _ in range(6):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    ctx.profile(True); fn(); torch.cuda.synchronize(); ctx.profile(False)
    ms, groups = ctx.profile_read()
    stages = sum(ms) / groups
    print("%s %s: wall %.2f ms, stage events %.2f ms, outside them (digest + gaps) %.2f ms" % (
        os.path.basename(_lib.LIB_PATH), suite.__name__, best * 1e3, stages, best * 1e3 - stages), flush=True)
    ctx.close()
