import os, sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from ark_ec_vrfs_amd import Context
from test_bls_pairing import enc_g1, enc_g2
from oracle import bls_oracle as b
ctx = Context(0); dev = torch.device('cuda:0')
c = 0x1234567FEDCBA987
sh = np.frombuffer(enc_g2(b.g2_mul(7, b.G2)) + enc_g2(b.g2_mul(7 * c % b.R, b.G2)), np.uint8).copy()
rows = [enc_g1(b.g1_mul(a * c % b.R, b.G1)) + enc_g1(b.g1_neg(b.g1_mul(a, b.G1))) for a in range(1, 9)]
s1 = np.frombuffer(b"".join(rows), np.uint8).reshape(-1, 192)
dsh = torch.from_numpy(sh).to(dev)
def best(fn, reps=6):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    return min(ts)
for n in (1, 2, 8):
    d1 = torch.from_numpy(np.tile(s1, (1, 1))[:n].copy()).to(dev)
    st = torch.full((n,), 9, dtype=torch.uint8, device=dev)
    for mode in ("oct", "tri", "row"):
        ctx.debug_pairing_layout(mode)
        t = best(lambda: ctx.pairing_check_batch_dev(d1, dsh, st, g2_shared=True))
        assert int(st.sum()) == 0
        print(f"prepared [{mode}] n={n}: {t*1e3:.2f} ms", flush=True)
    ctx.debug_pairing_layout()
m = 1 << 14
dm = torch.from_numpy(np.tile(s1, (m // 8, 1)).copy()).to(dev)
mst = torch.empty(m, dtype=torch.uint8, device=dev); verdict = torch.empty(1, dtype=torch.uint8, device=dev)
seed = os.urandom(32)
for mode in ("oct", "tri"):
    ctx.debug_pairing_layout(mode)
    fn = lambda: ctx.pairing_check_batch_rlc_dev(dm, dsh, mst, verdict, seed)
    fn(); torch.cuda.synchronize()
    ctx.profile(True); t = best(fn); ctx.profile(False)
    ms, g = ctx.profile_read()
    print(f"batched 2^14 [{mode}]: {t*1e3:.2f} ms stages {[round(x/g,2) for x in ms]}", flush=True)
