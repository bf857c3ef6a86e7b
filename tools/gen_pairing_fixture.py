#!/usr/bin/env python3
"""Generates tests/golden/pairing_items.json: small BLS12-381 pairing-check inputs (data only) made with the
Python oracle (oracle/bls_oracle.py, test infrastructure).  bench.py and the GPU tests tile these items; the
product path never imports the oracle.

  per_item : 8 items, each two (G1, G2) pairs with e(P0,Q0) e(P1,Q1) = 1          (g1 192 B, g2 384 B per item)
  shared   : one G2 pair (Q0, Q1) = (x G2, x cc G2) -- the shape of a KZG verifier's SRS -- and 8 items
             (A_i, B_i) = (a_i cc G1, -a_i G1) with e(A_i,Q0) e(B_i,Q1) = 1       (g1 192 B per item)
  shared_bad: 2 items for the same pair that do NOT satisfy the equation

Run from the repo root:  python tools/gen_pairing_fixture.py
"""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import bls_oracle as b  # noqa: E402


def w(x):
    return int(x).to_bytes(48, "little")


def enc_g1(p):
    return bytes(96) if p is None else w(p[0]) + w(p[1])


def enc_g2(q):
    return bytes(192) if q is None else w(q[0].a) + w(q[0].b) + w(q[1].a) + w(q[1].b)


def main():
    rnd = random.Random(20261004)
    per_item = []
    for _ in range(8):
        a1, b1, b2 = (rnd.randrange(1, b.R) for _ in range(3))
        a2 = (-a1 * b1 * pow(b2, -1, b.R)) % b.R
        p0, q0, p1, q1 = b.g1_mul(a1, b.G1), b.g2_mul(b1, b.G2), b.g1_mul(a2, b.G1), b.g2_mul(b2, b.G2)
        assert b.pairing_check([(p0, q0), (p1, q1)])
        per_item.append({"g1": (enc_g1(p0) + enc_g1(p1)).hex(), "g2": (enc_g2(q0) + enc_g2(q1)).hex()})
    x, cc = rnd.randrange(1, b.R), rnd.randrange(1, b.R)
    q0, q1 = b.g2_mul(x, b.G2), b.g2_mul(x * cc % b.R, b.G2)
    shared, bad = [], []
    for k in range(10):
        a = rnd.randrange(1, b.R)
        A, B = b.g1_mul(a * cc % b.R, b.G1), b.g1_neg(b.g1_mul(a, b.G1))
        if k >= 8:
            B = b.g1_add(B, b.G1)
            assert not b.pairing_check([(A, q0), (B, q1)])
            bad.append((enc_g1(A) + enc_g1(B)).hex())
        else:
            assert b.pairing_check([(A, q0), (B, q1)])
            shared.append((enc_g1(A) + enc_g1(B)).hex())
    out = {"provenance": "tools/gen_pairing_fixture.py (oracle/bls_oracle.py), seed 20261004; encodings as include/vrfhip.h "
                         "vrfhip_pairing_check_batch: 48-byte little-endian coordinates",
           "per_item": per_item, "shared_g2": (enc_g2(q0) + enc_g2(q1)).hex(), "shared": shared, "shared_bad": bad}
    path = os.path.join(ROOT, "tests", "golden", "pairing_items.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
