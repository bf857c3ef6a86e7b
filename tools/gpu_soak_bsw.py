"""One-off GPU soak (not a test) of `suites::bandersnatch_sw` at BASELINE.json's full size: 2^20 + 5 proofs (a ragged last launch
group) proved and verified on the device, stripes of every field disturbed, a 2^15-item random sample of proofs and of the
disturbed batch's statuses held against the C oracle (which works on the Weierstrass curve).
usage (GPU box): python tools/gpu_soak_bsw.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ark_ec_vrfs_amd import BandersnatchSwSha512Tai, Context
from oracle import c_oracle as co

n = (1 << 20) + 5
rng = np.random.default_rng(20)
ctx = Context(0, BandersnatchSwSha512Tai, test_blinding_base=True)
seeds = rng.integers(0, 256, (n, 16), dtype=np.uint8)
sk, pk = ctx.secret_from_seed_batch(seeds)
msg = rng.integers(0, 256, (n, 40), dtype=np.uint8)
ad = b"soak"
r = ctx.ietf_prove_batch(sk, msgs=msg, ad=ad)
assert not r["status"].any() and (r["pk"] == pk).all()
st = ctx.ietf_verify_batch(pk, r["input"], r["output"], r["c"], r["s"], ad=ad)
assert not st.any()
sta = ctx.ietf_verify_batch_alpha(pk, msg, r["output"], r["c"], r["s"], ad=ad)
assert not sta.any()
t = [x.copy() for x in (pk, r["input"], r["output"], r["c"], r["s"])]
t[4][0::11, 5] ^= 1; t[3][1::13, 9] ^= 2; t[2][2::17, 32] ^= 0x80; t[0][3::19, 32] |= 0xC0; t[1][4::23, 3] ^= 4
st = ctx.ietf_verify_batch(*t, ad=ad)
touched = np.zeros(n, bool)
for a, b in ((0, 11), (1, 13), (2, 17), (3, 19), (4, 23)):
    touched[a::b] = True
assert (st[~touched] == 0).all() and (st[touched] != 0).all()
idx = np.sort(rng.choice(n, 1 << 15, replace=False))
nt = min(16, os.cpu_count() or 1)
ref = co.bsw_ietf_prove_batch(sk[idx], msgs=msg[idx], ad=ad, threads=nt)
for k in ("output", "c", "s", "pk", "input"):
    assert (ref[k] == r[k][idx]).all(), k
want = co.bsw_ietf_verify_batch(*[x[idx] for x in t], ad=ad, threads=nt)
assert (want == st[idx]).all()
p = ctx.pedersen_prove_batch(sk, msgs=msg, ad=ad)
args = [p[k] for k in ("input", "output", "pk_com", "r", "ok", "s", "sb")]
assert not ctx.pedersen_verify_batch(*args, ad=ad).any()
stb, fast = ctx.pedersen_verify_batch_rlc(*args, ad=ad)
assert fast and not stb.any()
pref = co.bsw_pedersen_prove_batch(sk[idx], msgs=msg[idx], ad=ad, threads=nt)
for k in ("output", "pk_com", "r", "ok", "s", "sb"):
    assert (pref[k] == p[k][idx]).all(), k
print("bandersnatch_sw soak ok: %d proofs proved and verified (per proof, from alpha, Pedersen per proof and as one MSM); %d disturbed "
      "items rejected, none else; %d sampled proofs and statuses equal the C oracle's (statuses %s)" % (
          n, int(touched.sum()), len(idx), np.bincount(want, minlength=3).tolist()))
