"""One profiled pass of 2^N (argv[1], default 16) pairing checks: for rocprofv3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from ark_ec_vrfs_amd import Context
from test_bls_pairing import kzg_like_items, pack
ctx = Context(0); dev = torch.device('cuda:0')
g1, g2 = pack(kzg_like_items(8, seed=21))
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 16)
d1 = torch.from_numpy(np.tile(g1, (n // 8, 1)).copy()).to(dev)
d2 = torch.from_numpy(np.tile(g2, (n // 8, 1)).copy()).to(dev)
st = torch.full((n,), 9, dtype=torch.uint8, device=dev)
for _ in range(2):
    ctx.pairing_check_batch_dev(d1, d2, st)
torch.cuda.synchronize()
assert int(st.sum()) == 0
# the same size against one shared G2 pair: k_pairing_prepare_g2 once, then k_pairing_check2_quad_prepared
from oracle import bls_oracle as b  # noqa: E402  (tools script: the oracle only builds inputs)
from test_bls_pairing import enc_g1, enc_g2  # noqa: E402
c = 0x1234567FEDCBA987
sh = np.frombuffer(enc_g2(b.g2_mul(7, b.G2)) + enc_g2(b.g2_mul(7 * c % b.R, b.G2)), np.uint8).copy()
rows = [enc_g1(b.g1_mul(a * c % b.R, b.G1)) + enc_g1(b.g1_neg(b.g1_mul(a, b.G1))) for a in range(1, 9)]
s1 = np.frombuffer(b"".join(rows), np.uint8).reshape(-1, 192)
d1 = torch.from_numpy(np.tile(s1, (n // 8, 1)).copy()).to(dev)
dsh = torch.from_numpy(sh).to(dev)
for _ in range(2):
    ctx.pairing_check_batch_dev(d1, dsh, st, g2_shared=True)
torch.cuda.synchronize()
assert int(st.sum()) == 0
