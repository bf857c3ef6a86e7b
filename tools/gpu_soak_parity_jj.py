"""Parity soak for BASELINE.json config 4's shape (Pedersen on JubJub): 2^N (default 18) GPU proofs byte-compared
with the C oracle's, and the statuses of a randomly tampered verify batch (per-proof AND batched verifier)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ark_ec_vrfs_amd import Context, JubJubSha512Tai
from oracle import c_oracle as co
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 18)
th = min(os.cpu_count() or 8, 64)
ctx = Context(0, suite=JubJubSha512Tai, test_blinding_base=True)
co.set_suite(2)
seeds = np.arange(n, dtype=np.uint64).view(np.uint8).reshape(n, 8)
sk, pk = ctx.secret_from_seed_batch(seeds)
msg = np.random.default_rng(17).integers(0, 256, (n, 32), dtype=np.uint8)
got = ctx.pedersen_prove_batch(sk, msgs=msg, ad=b"soak")
t = time.time(); step = 1 << 15
ref = {k: [] for k in ("input", "output", "pk_com", "r", "ok", "s", "sb", "blinding")}
for lo in range(0, n, step):
    part = co.pedersen_prove_batch(sk[lo:lo + step], msgs=msg[lo:lo + step], ad=b"soak", threads=th)
    for k in ref:
        assert (got[k][lo:lo + step] == part[k]).all(), (k, lo)
print(f"JubJub Pedersen prove parity: {n} proofs byte-equal ({time.time() - t:.0f} s of oracle time)", flush=True)
rnd = np.random.default_rng(6)
kind = rnd.integers(0, 8, n)                     # 0..3 untouched; 4: sb bit; 5: s bit; 6: output swapped; 7: pk_com byte
a = {k: got[k].copy() for k in ("input", "output", "pk_com", "r", "ok", "s", "sb")}
for kk, name in ((4, "sb"), (5, "s"), (7, "pk_com")):
    i = np.nonzero(kind == kk)[0]
    a[name][i, rnd.integers(0, 32, i.size)] ^= (1 << rnd.integers(0, 8, i.size)).astype(np.uint8)
i6 = np.nonzero(kind == 6)[0]; a["output"][i6] = got["output"][(i6 + 1) % n]
args = (a["input"], a["output"], a["pk_com"], a["r"], a["ok"], a["s"], a["sb"])
st = ctx.pedersen_verify_batch(*args, ad=b"soak")
t = time.time()
want = np.concatenate([co.pedersen_verify_batch(*(x[lo:lo + step] for x in args), b"soak", threads=th) for lo in range(0, n, step)])
assert (st == want).all(), np.nonzero(st != want)[0][:10]
r = ctx.pedersen_verify_batch_rlc(*args, ad=b"soak")
st_rlc = r[0] if isinstance(r, tuple) else r
assert (np.asarray(st_rlc) == want).all()
print(f"JubJub Pedersen verify parity: {n} statuses equal the oracle's, per-proof and batched-with-fallback "
      f"({time.time() - t:.0f} s of oracle time); ok={int((want == 0).sum())} failure={int((want == 1).sum())} "
      f"invalid={int((want == 2).sum())}", flush=True)
co.set_suite(1)
