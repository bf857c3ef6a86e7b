"""Quick GPU bring-up: KATs, small random parity vs the Python oracle, first timings."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ark_ec_vrfs_amd import Context
from oracle import vrf_oracle as o

S = o.BANDERSNATCH
Q = S.q
t0 = time.time()
ctx = Context(0)
print("ctx create %.2fs" % (time.time() - t0), flush=True)
rng = np.random.default_rng(1)
a = rng.integers(0, 256, (1000, 32), dtype=np.uint8); b = rng.integers(0, 256, (1000, 32), dtype=np.uint8)
r = ctx.fq_mul_batch(a, b)
for i in range(1000):
    x = int.from_bytes(a[i].tobytes(), 'little'); y = int.from_bytes(b[i].tobytes(), 'little')
    assert int.from_bytes(r[i].tobytes(), 'little') == x * y % Q, i
print("fq_mul ok", flush=True)
k = json.load(open(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'bandersnatch_sha512_ell2_kat.json')))
H = lambda s: np.frombuffer(bytes.fromhex(s), dtype=np.uint8)
for v in k['ietf']:
    ad = bytes.fromhex(v['ad'])
    pr = ctx.ietf_prove_batch(H(v['sk']), msgs=[bytes.fromhex(v['alpha'])], ad=ad)
    got = {kk: pr[kk][0].tobytes().hex() for kk in ('output', 'c', 's', 'pk', 'input')}
    exp = dict(output=v['gamma'], c=v['c'], s=v['s'], pk=v['pk'], input=v['h'])
    assert got == exp, (got, exp)
    st = ctx.ietf_verify_batch(H(v['pk']), H(v['h']), H(v['gamma']), H(v['c']), H(v['s']), ad=ad)
    assert st[0] == 0, st
    assert ctx.output_hash_batch(H(v['gamma']))[0].tobytes().hex() == v['beta']
print("KAT ok", flush=True)
# random parity vs oracle
N = 64
sks, msgs = [], []
for i in range(N):
    sk = o.secret_from_seed(S, o.synth_seed(i)); sks.append(o.scalar_encode(sk)); msgs.append(o.synth_msg(i))
pr = ctx.ietf_prove_batch(np.frombuffer(b"".join(sks), np.uint8).reshape(N, 32), msgs=msgs, ad=b"xy")
for i in range(N):
    sk = int.from_bytes(sks[i], 'little')
    Hh = o.data_to_point(S, msgs[i]); g, c, s = o.ietf_prove(S, sk, Hh, b"xy")
    assert pr['output'][i].tobytes() == o.point_encode(S, g) and pr['c'][i].tobytes() == o.scalar_encode(c) and pr['s'][i].tobytes() == o.scalar_encode(s), i
st = ctx.ietf_verify_batch(pr['pk'], pr['input'], pr['output'], pr['c'], pr['s'], ad=b"xy")
assert (st == 0).all(), st
bad = pr['s'].copy(); bad[::2, 0] ^= 1
st = ctx.ietf_verify_batch(pr['pk'], pr['input'], pr['output'], pr['c'], bad, ad=b"xy")
assert (st[::2] == 1).all() and (st[1::2] == 0).all(), st
print("random parity ok", flush=True)
# timings on device buffers
dev = torch.device('cuda:0')
for logn in (12, 16, 18, 20):
    n = 1 << logn
    sk_h, _ = ctx.secret_from_seed_batch(np.arange(n, dtype=np.uint64).view(np.uint8).reshape(n, 8), with_public=False) if logn <= 16 else (None, None)
    if sk_h is None:
        seeds = torch.arange(n, dtype=torch.int64, device=dev).view(torch.uint8).reshape(n, 8)
        sk = torch.empty((n, 32), dtype=torch.uint8, device=dev)
        import ctypes
        from ark_ec_vrfs_amd import _lib
        _lib.check(_lib.load().vrfhip_secret_from_seed_batch_dev(ctx.handle, n, seeds.data_ptr(), 8, sk.data_ptr(), None, torch.cuda.current_stream().cuda_stream), "seed")
    else:
        sk = torch.from_numpy(sk_h).to(dev)
    msg = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev)
    out = torch.empty((n, 32), dtype=torch.uint8, device=dev); c = torch.empty_like(out); s = torch.empty_like(out)
    pk = torch.empty_like(out); hh = torch.empty_like(out); stt = torch.empty(n, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize(); t = time.time()
    ctx.ietf_prove_batch_dev(sk, msg, 32, out, c, s, pk, hh, stt)
    torch.cuda.synchronize(); tp = time.time() - t
    t = time.time()
    ctx.ietf_prove_batch_dev(sk, msg, 32, out, c, s, pk, hh, stt)
    torch.cuda.synchronize(); tp2 = time.time() - t
    assert int(stt.sum()) == 0
    vs = torch.empty(n, dtype=torch.uint8, device=dev)
    t = time.time()
    ctx.ietf_verify_batch_dev(pk, hh, out, c, s, vs)
    torch.cuda.synchronize(); tv = time.time() - t
    t = time.time()
    ctx.ietf_verify_batch_dev(pk, hh, out, c, s, vs)
    torch.cuda.synchronize(); tv2 = time.time() - t
    assert int(vs.sum()) == 0, int(vs.sum())
    print(f"n=2^{logn}: prove {tp:.3f}s/{tp2:.3f}s ({n/tp2:.3e}/s)  verify {tv:.3f}s/{tv2:.3f}s ({n/tv2:.3e}/s) ws={ctx.workspace_bytes()/2**20:.0f}MiB", flush=True)
print("done")
