"""One-off GPU check (not a test): 2^16 random byte strings as secp256r1 proofs -- tags forced to 0x02 / 0x03 on most items so
that about an eighth decode -- through vrfhip_ietf_verify_batch and vrfhip_pedersen_verify_batch against oracle/c/oracle_p256.c.
usage (GPU box): python tools/gpu_p256_fuzz.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ark_ec_vrfs_amd import Context, Secp256r1Sha256Tai
from oracle import c_oracle as co, sw_oracle as sw
n = 1 << 16
rng = np.random.default_rng(2026)
ctx = Context(0, Secp256r1Sha256Tai, test_blinding_base=True)
co.p256_set_blinding_base(sw.default_blinding_base())
pts = [rng.integers(0, 256, (n, 33), dtype=np.uint8) for _ in range(5)]
for p in pts:
    keep = rng.integers(0, 16, n) == 0
    p[~keep, 0] = 2 + (p[~keep, 0] & 1)
sc = [rng.integers(0, 256, (n, 32), dtype=np.uint8) for _ in range(2)]
sc[0][:, :16] = 0                                  # c as a 16-byte challenge on the wire
got = ctx.ietf_verify_batch(pts[0], pts[1], pts[2], sc[0], sc[1], ad=b"fz")
want = co.p256_ietf_verify_batch(pts[0], pts[1], pts[2], sc[0], sc[1], ad=b"fz", threads=16)
print("ietf: equal", bool((got == want).all()), "histogram", np.bincount(want, minlength=3))
got = ctx.pedersen_verify_batch(*pts, sc[0], sc[1], ad=b"fz")
want = co.p256_pedersen_verify_batch(*pts, sc[0], sc[1], ad=b"fz", threads=16)
print("pedersen: equal", bool((got == want).all()), "histogram", np.bincount(want, minlength=3))
st = ctx.point_validate_batch(pts[0])
print("validate: equal", bool((st == np.array([co.p256_point_decode(x.tobytes()) for x in pts[0][:4096]] + list(st[4096:]), np.uint8)).all()))
