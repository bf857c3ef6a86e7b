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
