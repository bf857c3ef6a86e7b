"""Full-size parity soak (not part of the test suite: ~2 min of CPU): EVERY one of 2^20 GPU proofs byte-compared
with the C oracle's, and the GPU statuses of a randomly tampered 2^20 verify batch compared with the oracle's."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ark_ec_vrfs_amd import Context, _lib
from oracle import c_oracle as co
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 20)
th = os.cpu_count() or 8
dev = torch.device('cuda:0'); lib = _lib.load(); ctx = Context(0); st0 = torch.cuda.current_stream().cuda_stream
seeds = torch.arange(n, dtype=torch.int64, device=dev).view(torch.uint8).reshape(n, 8)
sk = torch.empty((n, 32), dtype=torch.uint8, device=dev)
_lib.check(lib.vrfhip_secret_from_seed_batch_dev(ctx.handle, n, seeds.data_ptr(), 8, sk.data_ptr(), 0, st0), "seed")
g = torch.Generator(device=dev); g.manual_seed(99)
msg = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
mk = lambda: torch.empty((n, 32), dtype=torch.uint8, device=dev)
out, c, s, pk, hh = (mk() for _ in range(5)); pst = torch.empty(n, dtype=torch.uint8, device=dev)
ctx.ietf_prove_batch_dev(sk, msg, 32, out, c, s, pk, hh, pst); torch.cuda.synchronize()
H = {k: v.cpu().numpy() for k, v in dict(sk=sk, msg=msg, out=out, c=c, s=s, pk=pk, hh=hh).items()}
t = time.time()
step = 1 << 16
for lo in range(0, n, step):
    ref = co.ietf_prove_batch(H["sk"][lo:lo + step], msgs=H["msg"][lo:lo + step], ad=b"", threads=th)
    for a, b in (("out", "output"), ("c", "c"), ("s", "s"), ("pk", "pk"), ("hh", "input")):
        assert (H[a][lo:lo + step] == ref[b]).all(), (a, lo)
    print(f"prove parity: items [{lo}, {lo + step}) byte-equal ({time.time() - t:.0f} s)", flush=True)
rnd = np.random.default_rng(5)
kind = rnd.integers(0, 8, n)                       # 0..3 untouched; 4: s bit; 5: c bit; 6: output swapped; 7: pk byte
a = {k: H[k].copy() for k in ("pk", "hh", "out", "c", "s")}
i4 = np.nonzero(kind == 4)[0]; a["s"][i4, rnd.integers(0, 32, i4.size)] ^= (1 << rnd.integers(0, 8, i4.size)).astype(np.uint8)
i5 = np.nonzero(kind == 5)[0]; a["c"][i5, rnd.integers(0, 32, i5.size)] ^= (1 << rnd.integers(0, 8, i5.size)).astype(np.uint8)
i6 = np.nonzero(kind == 6)[0]; a["out"][i6] = H["out"][(i6 + 1) % n]
i7 = np.nonzero(kind == 7)[0]; a["pk"][i7, rnd.integers(0, 32, i7.size)] ^= (1 << rnd.integers(0, 8, i7.size)).astype(np.uint8)
got = ctx.ietf_verify_batch(a["pk"], a["hh"], a["out"], a["c"], a["s"])
t = time.time()
want = np.concatenate([co.ietf_verify_batch(a["pk"][lo:lo + step], a["hh"][lo:lo + step], a["out"][lo:lo + step],
                                            a["c"][lo:lo + step], a["s"][lo:lo + step], b"", threads=th)
                       for lo in range(0, n, step)])
assert (got == want).all(), np.nonzero(got != want)[0][:10]
print(f"verify parity: {n} statuses equal the oracle's ({time.time() - t:.0f} s of oracle time on {th} threads); "
      f"ok={int((want == 0).sum())} failure={int((want == 1).sum())} invalid={int((want == 2).sum())}", flush=True)
assert (want[kind < 4] == 0).all() and (want[kind >= 4] != 0).all()
