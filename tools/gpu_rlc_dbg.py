import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ark_ec_vrfs_amd import Context
from oracle import c_oracle as co, vrf_oracle as o
ctx = Context(0); dev = torch.device('cuda:0')
FIELDS = ("input", "output", "pk_com", "r", "ok", "s", "sb")
n = 512; ad = b"mixed"; SEED = bytes(range(32))
sk = np.stack([np.frombuffer(co.secret_from_seed(o.synth_seed(7000 + i)), np.uint8) for i in range(n)])
msg = np.stack([np.frombuffer(o.synth_msg(7000 + i), np.uint8) for i in range(n)])
ref = co.pedersen_prove_batch(sk, msgs=msg, ad=ad, threads=8)
a = {k: ref[k].copy() for k in FIELDS}
def run(c, label):
    d = [torch.from_numpy(c[k]).to(dev) for k in FIELDS]
    adt = torch.from_numpy(np.frombuffer(ad, np.uint8).copy()).to(dev)
    st = torch.full((n,), 9, dtype=torch.uint8, device=dev); flag = torch.full((1,), 9, dtype=torch.uint8, device=dev)
    ctx.pedersen_verify_batch_rlc_dev(*d, st, flag, SEED, ad=adt, ad_len=len(ad))
    torch.cuda.synchronize()
    want_st, want_fail = co.pedersen_rlc_check(*[c[k] for k in FIELDS], seed=SEED, ad=ad)
    items = co.pedersen_verify_batch(*[c[k] for k in FIELDS], ad, threads=8)
    stn = st.cpu().numpy()
    print(label, "gpu flag", int(flag[0]), "oracle fail", want_fail, "status equal", (stn == want_st).all(), "nonzero", np.nonzero(stn)[0], "per-item nonzero", np.nonzero(items)[0], items[np.nonzero(items)[0]])
run(a, "valid")
c = {k: v.copy() for k, v in a.items()}; c["s"][11] = np.frombuffer(int(o.BANDERSNATCH.r).to_bytes(32, "little"), np.uint8)
run(c, "s[11]=r")
c = {k: v.copy() for k, v in a.items()}; c["pk_com"][300] = np.frombuffer((2).to_bytes(32, "little"), np.uint8)
run(c, "pk_com[300]=2")
for y in (3, 4, 5, 6, 7):
    c = {k: v.copy() for k, v in a.items()}; c["r"][7] = np.frombuffer((y).to_bytes(32, "little"), np.uint8)
    run(c, "r[7]=%d" % y)
