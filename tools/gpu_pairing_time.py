import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from ark_ec_vrfs_amd import Context
from test_bls_pairing import kzg_like_items, pack
ctx = Context(0); dev = torch.device('cuda:0')
items = kzg_like_items(8, seed=21)
g1, g2 = pack(items)
for logn in (10, 12, 14, 16, 18):
    n = 1 << logn
    d1 = torch.from_numpy(np.tile(g1, (n // 8, 1)).copy()).to(dev)
    d2 = torch.from_numpy(np.tile(g2, (n // 8, 1)).copy()).to(dev)
    st = torch.full((n,), 9, dtype=torch.uint8, device=dev)
    ts = []
    for rep in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        ctx.pairing_check_batch_dev(d1, d2, st)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    assert int(st.sum()) == 0
    print(f"pairing check n=2^{logn}: {min(ts)*1e3:.2f} ms ({n/min(ts):.3e} checks/s)", flush=True)

# shared G2 points (the SRS case): lines prepared once per call, items scale them
from oracle import bls_oracle as b  # noqa: E402  (tools script: the oracle only builds test inputs)
from test_bls_pairing import enc_g1, enc_g2
c = 0x1234567FEDCBA987
sh = np.frombuffer(enc_g2(b.g2_mul(7, b.G2)) + enc_g2(b.g2_mul(7 * c % b.R, b.G2)), np.uint8).copy()
rows = [enc_g1(b.g1_mul(a * c % b.R, b.G1)) + enc_g1(b.g1_neg(b.g1_mul(a, b.G1))) for a in range(1, 9)]
s1 = np.frombuffer(b"".join(rows), np.uint8).reshape(-1, 192)
dsh = torch.from_numpy(sh).to(dev)
for logn in (10, 14, 16, 18):
    n = 1 << logn
    d1 = torch.from_numpy(np.tile(s1, (n // 8, 1)).copy()).to(dev)
    st = torch.full((n,), 9, dtype=torch.uint8, device=dev)
    for mode in ("prepared", "noprep"):
        if mode == "noprep": os.environ["VRFHIP_PAIRING"] = "noprep"
        ts = []
        for rep in range(3):
            torch.cuda.synchronize(); t = time.perf_counter()
            ctx.pairing_check_batch_dev(d1, dsh, st, g2_shared=True)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        os.environ.pop("VRFHIP_PAIRING", None)
        assert int(st.sum()) == 0
        print(f"shared-G2 pairing check ({mode}) n=2^{logn}: {min(ts)*1e3:.2f} ms ({n/min(ts):.3e} checks/s)", flush=True)
