"""One-off GPU check (not a test): the PROVERS of every suite against the C oracle at the hash functions' block boundaries.
Message lengths and additional-data lengths are chosen so that the hash-to-curve, nonce, blinding and challenge preimages
end just before, on and just after a SHA-512 / SHA-256 block or padding boundary; secret keys include 1, 2, r - 1 and
r - 2 beside random ones.  Every byte the provers write (output, c, s, pk, input; pk_com, r, ok, s, sb, blinding) is compared.
usage (GPU box): python tools/gpu_prove_fuzz.py      -- one line per suite, exit 1 on a difference"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ark_ec_vrfs_amd import (BabyJubJubSha512Tai, BandersnatchSha512Ell2, BandersnatchSwSha512Tai, Context, Ed25519Sha512Tai,
                             JubJubSha512Tai, Secp256r1Sha256Tai)
from oracle import c_oracle as co, sw_oracle as sw

N = 1024
THREADS = min(16, os.cpu_count() or 1)
rng = np.random.default_rng(6979)
MSG_LENS = [1, 20, 31, 32, 33, 54, 55, 56, 63, 64, 65, 86, 100, 110, 111, 112, 119, 120, 127, 128, 129, 200, 300]
AD_LENS = [0, 1, 9, 21, 22, 23, 54, 55, 56, 64, 85, 86, 87, 119, 128, 200]


def secrets(order, big_endian):
    sk = rng.integers(0, 256, (N, 32), dtype=np.uint8)
    for i in range(N):                                     # random values below the order
        v = int.from_bytes(sk[i].tobytes(), "little") % order or 1
        sk[i] = np.frombuffer(v.to_bytes(32, "big" if big_endian else "little"), np.uint8)
    for i, v in enumerate((1, 2, order - 1, order - 2, order >> 1)):
        sk[i] = np.frombuffer(v.to_bytes(32, "big" if big_endian else "little"), np.uint8)
    return sk


CT = "--ct" in sys.argv        # VRFHIP_FLAG_CT_TABLES: the provers' window lookups read all eight table entries


def run(name, suite, sid, order, big_endian, p256, bsw=False):
    ctx = Context(0, suite, test_blinding_base=True)
    if CT:
        ctx.set_flags(ctx.CT_TABLES)
    if p256:
        co.p256_set_blinding_base(sw.default_blinding_base())
        ip, pp = co.p256_ietf_prove_batch, co.p256_pedersen_prove_batch
    elif bsw:
        ip, pp = co.bsw_ietf_prove_batch, co.bsw_pedersen_prove_batch
    else:
        co.set_suite(sid)
        ip, pp = co.ietf_prove_batch, co.pedersen_prove_batch
    sk = secrets(order, big_endian)
    bad, proofs = 0, 0
    for t, ml in enumerate(MSG_LENS):
        al = AD_LENS[t % len(AD_LENS)]
        msg = rng.integers(0, 256, (N, ml), dtype=np.uint8)
        ad = bytes(rng.integers(0, 256, al, dtype=np.uint8))
        got = ctx.ietf_prove_batch(sk, msgs=msg, ad=ad)
        pgot = ctx.pedersen_prove_batch(sk, msgs=msg, ad=ad)
        ref, pref = ip(sk, msgs=msg, ad=ad, threads=THREADS), pp(sk, msgs=msg, ad=ad, threads=THREADS)
        for k in ("output", "c", "s", "pk", "input"):
            bad += int((got[k] != ref[k]).any(axis=1).sum())
        for k in ("output", "pk_com", "r", "ok", "s", "sb", "blinding"):
            bad += int((pgot[k] != pref[k]).any(axis=1).sum())
        proofs += 2 * N
    print("%-15s %7d proofs (IETF + Pedersen) over %d message / ad length pairs, differing fields %d" % (name, proofs, len(MSG_LENS), bad), flush=True)
    ctx.close()
    return bad


R_BS = 0x1CFB69D4CA675F520CCE760202687600FF8F87007419047174FD06B52876E7E1
R_JJ = 0x0E7DB4EA6533AFA906673B0101343B00A6682093CCC81082D0970E5ED6F72CB7
R_ED = (1 << 252) + 27742317777372353535851937790883648493
R_BJ = 2736030358979909402780800718157159386076813972158567259200215660948447373041
total = 0
total += run("bandersnatch", BandersnatchSha512Ell2, 1, R_BS, False, False)
total += run("jubjub", JubJubSha512Tai, 2, R_JJ, False, False)
total += run("ed25519", Ed25519Sha512Tai, 3, R_ED, False, False)
total += run("babyjubjub", BabyJubJubSha512Tai, 4, R_BJ, False, False)
total += run("secp256r1", Secp256r1Sha256Tai, 5, sw.N, True, True)
total += run("bandersnatch_sw", BandersnatchSwSha512Tai, 6, R_BS, False, False, bsw=True)
co.set_suite(1)
print("TOTAL differing fields", total)
sys.exit(1 if total else 0)
