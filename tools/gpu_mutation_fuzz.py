"""One-off GPU check (not a test): MUTATED valid proofs, every suite, IETF and Pedersen, statuses against the C oracle.
Random byte strings almost never decode, so they exercise the codec and little else; a valid proof with one field disturbed
-- a flipped bit, a byte replaced, a field of another item, a scalar plus the group order, a scalar of all ones -- decodes
far more often and reaches the ladders, the challenge hash and the subgroup tests with adversarial values.
usage (GPU box): python tools/gpu_mutation_fuzz.py [rounds]      -- prints one line per suite and scheme, exit 1 on a difference"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ark_ec_vrfs_amd import (BabyJubJubSha512Tai, BandersnatchSha512Ell2, BandersnatchSwSha512Tai, Context, Ed25519Sha512Tai,
                             JubJubSha512Tai, Secp256r1Sha256Tai)
from oracle import c_oracle as co, sw_oracle as sw

ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 12
N = 1 << 13
THREADS = min(16, os.cpu_count() or 1)
rng = np.random.default_rng(9381)


def mutate(fields, orders, big_endian):
    """fields: list of [N, w] uint8 arrays (valid values).  One field per item gets one of five disturbances.
    orders[j]: the group order when field j is a scalar (None for points)."""
    out = [f.copy() for f in fields]
    which = rng.integers(0, len(fields), N)
    kind = rng.integers(0, 5, N)
    for j, f in enumerate(out):
        w = f.shape[1]
        rows = np.nonzero(which == j)[0]
        for i in rows:
            k = kind[i]
            if k == 0:                                        # one bit
                f[i, rng.integers(0, w)] ^= 1 << rng.integers(0, 8)
            elif k == 1:                                      # one byte
                f[i, rng.integers(0, w)] = rng.integers(0, 256)
            elif k == 2:                                      # the same field of another item
                f[i] = fields[j][(i + 1 + rng.integers(0, N - 1)) % N]
            elif k == 3 and orders[j] is not None:            # scalar + order (another string of the same residue), if it fits
                v = int.from_bytes(f[i].tobytes(), "big" if big_endian else "little") + orders[j]
                if v < 1 << (8 * w):
                    f[i] = np.frombuffer(v.to_bytes(w, "big" if big_endian else "little"), np.uint8)
            else:                                             # extreme values
                f[i] = 0xFF if rng.integers(0, 2) else 0
    return out


def run(name, suite, sid, order, big_endian, p256, bsw=False):
    ctx = Context(0, suite, test_blinding_base=True)
    seeds = np.arange(N, dtype=np.uint64).view(np.uint8).reshape(N, 8)
    msg = rng.integers(0, 256, (N, 32), dtype=np.uint8)
    sk, _ = ctx.secret_from_seed_batch(seeds)
    if p256:
        co.p256_set_blinding_base(sw.default_blinding_base())
        iv, pv = co.p256_ietf_verify_batch, co.p256_pedersen_verify_batch
    elif bsw:
        iv, pv = co.bsw_ietf_verify_batch, co.bsw_pedersen_verify_batch
    else:
        co.set_suite(sid)
        iv, pv = co.ietf_verify_batch, co.pedersen_verify_batch
    bad = 0
    r = ctx.ietf_prove_batch(sk, msgs=msg, ad=b"mf")
    base = [r[k] for k in ("pk", "input", "output", "c", "s")]
    assert (ctx.ietf_verify_batch(*base, ad=b"mf") == 0).all()
    hist = np.zeros(3, np.int64)
    for _ in range(ROUNDS):
        m = mutate(base, [None, None, None, order, order], big_endian)
        got, want = ctx.ietf_verify_batch(*m, ad=b"mf"), iv(*m, ad=b"mf", threads=THREADS)
        bad += int((got != want).sum())
        hist += np.bincount(want, minlength=3)
    print("%-15s ietf     %8d mutated proofs, differences %d, oracle statuses %s" % (name, ROUNDS * N, bad, hist.tolist()), flush=True)
    if not p256:
        # the same through verification from (pk, alpha, proof): the message takes the input's place among the mutated fields
        # (its H for the oracle comes from the library's hash-to-curve, itself held against the oracle by the tests)
        base_a = [base[0], msg, base[2], base[3], base[4]]
        bad_a = 0
        hist[:] = 0
        for _ in range(ROUNDS):
            m = mutate(base_a, [None, None, None, order, order], big_endian)
            got = ctx.ietf_verify_batch_alpha(m[0], m[1], m[2], m[3], m[4], ad=b"mf")
            want = iv(m[0], ctx.hash_to_curve_batch(m[1]), m[2], m[3], m[4], ad=b"mf", threads=THREADS)
            bad_a += int((got != want).sum())
            hist += np.bincount(want, minlength=3)
        print("%-15s alpha    %8d mutated proofs, differences %d, oracle statuses %s" % (name, ROUNDS * N, bad_a, hist.tolist()), flush=True)
        bad += bad_a
    r = ctx.pedersen_prove_batch(sk, msgs=msg, ad=b"mf")
    base = [r[k] for k in ("input", "output", "pk_com", "r", "ok", "s", "sb")]
    assert (ctx.pedersen_verify_batch(*base, ad=b"mf") == 0).all()
    hist[:] = 0
    badp = 0
    for _ in range(ROUNDS):
        m = mutate(base, [None] * 5 + [order, order], big_endian)
        got, want = ctx.pedersen_verify_batch(*m, ad=b"mf"), pv(*m, ad=b"mf", threads=THREADS)
        badp += int((got != want).sum())
        hist += np.bincount(want, minlength=3)
    print("%-15s pedersen %8d mutated proofs, differences %d, oracle statuses %s" % (name, ROUNDS * N, badp, hist.tolist()), flush=True)
    ctx.close()
    return bad + badp


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
print("TOTAL differences", total)
sys.exit(1 if total else 0)
