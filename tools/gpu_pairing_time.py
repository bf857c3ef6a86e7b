"""GPU timing of the pairing-check layouts (one item per quad / per 8 lanes / per row / per wave), per-item G2 points and a
shared G2 pair, at the batch sizes given on the command line (log2; default 12 13 14 16)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from ark_ec_vrfs_amd import Context
from test_bls_pairing import kzg_like_items, pack, enc_g1, enc_g2
from oracle import bls_oracle as b  # noqa: E402  (tools script: the oracle only builds test inputs)

logs = [int(x) for x in sys.argv[1:]] or [12, 13, 14, 16]
ctx = Context(0); dev = torch.device('cuda:0')
g1, g2 = pack(kzg_like_items(8, seed=21))


def best(fn, reps=4):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter()
        fn()
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    return min(ts)


for logn in logs:
    n = 1 << logn
    d1 = torch.from_numpy(np.tile(g1, (n // 8, 1)).copy()).to(dev)
    d2 = torch.from_numpy(np.tile(g2, (n // 8, 1)).copy()).to(dev)
    st = torch.full((n,), 9, dtype=torch.uint8, device=dev)
    for mode in ("quad", "oct1", "oct"):
        ctx.debug_pairing_layout(mode)
        t = best(lambda: ctx.pairing_check_batch_dev(d1, d2, st))
        assert int(st.sum()) == 0
        print(f"pairing check [{mode}] n=2^{logn}: {t*1e3:.2f} ms ({n/t:.3e} checks/s)", flush=True)
    ctx.debug_pairing_layout()

c = 0x1234567FEDCBA987
sh = np.frombuffer(enc_g2(b.g2_mul(7, b.G2)) + enc_g2(b.g2_mul(7 * c % b.R, b.G2)), np.uint8).copy()
rows = [enc_g1(b.g1_mul(a * c % b.R, b.G1)) + enc_g1(b.g1_neg(b.g1_mul(a, b.G1))) for a in range(1, 9)]
s1 = np.frombuffer(b"".join(rows), np.uint8).reshape(-1, 192)
dsh = torch.from_numpy(sh).to(dev)
for logn in logs:
    n = 1 << logn
    d1 = torch.from_numpy(np.tile(s1, (n // 8, 1)).copy()).to(dev)
    st = torch.full((n,), 9, dtype=torch.uint8, device=dev)
    for mode in ("quad", "oct", "row", "tri") if logn <= 12 else ("quad", "oct"):
        ctx.debug_pairing_layout(mode)
        t = best(lambda: ctx.pairing_check_batch_dev(d1, dsh, st, g2_shared=True))
        assert int(st.sum()) == 0
        print(f"shared-G2 pairing check [{mode}, prepared lines] n=2^{logn}: {t*1e3:.2f} ms ({n/t:.3e} checks/s)", flush=True)
    ctx.debug_pairing_layout()
