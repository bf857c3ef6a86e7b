"""ctypes wrapper of oracle/c/liboracle_vrf.so (plain-C CPU restatement).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product."""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import c_int, c_size_t, c_void_p

import numpy as np

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c")
LIB_PATH = os.path.join(_DIR, "liboracle_vrf.so")
_lib = None


def build() -> None:
    subprocess.run(["make", "-C", _DIR], check=True, stdout=subprocess.DEVNULL)


def load() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        lib = ctypes.CDLL(LIB_PATH)
        P = c_void_p
        lib.oracle_ietf_verify_batch.argtypes = [c_size_t, P, P, P, P, P, P, c_size_t, P, c_int]
        lib.oracle_ietf_verify_batch.restype = None
        lib.oracle_ietf_prove_batch.argtypes = [c_size_t, P, P, c_size_t, P, P, c_size_t, P, P, P, P, P, P, c_int]
        lib.oracle_ietf_prove_batch.restype = None
        lib.oracle_pedersen_verify_batch.argtypes = [c_size_t, P, P, P, P, c_size_t, P, c_int]
        lib.oracle_pedersen_verify_batch.restype = None
        lib.oracle_pedersen_prove_batch.argtypes = [c_size_t, P, P, c_size_t, P, P, c_size_t, P, P, P, P, P, c_int]
        lib.oracle_pedersen_prove_batch.restype = None
        lib.oracle_pedersen_rlc_check.argtypes = [c_size_t, P, P, P, P, c_size_t, P, ctypes.c_uint64, P]
        lib.oracle_pedersen_rlc_check.restype = c_int
        lib.oracle_msm.argtypes = [c_size_t, P, P, P, P]
        lib.oracle_msm.restype = c_int
        lib.oracle_hash_to_curve.argtypes = [P, c_size_t, P]
        lib.oracle_output_hash.argtypes = [P, P]
        lib.oracle_secret_from_seed.argtypes = [P, c_size_t, P]
        lib.oracle_public_from_secret.argtypes = [P, P]
        lib.oracle_point_decode.argtypes = [P, c_int, P]
        lib.oracle_point_decode.restype = c_int
        lib.oracle_fq_mul.argtypes = [P, P, P]
        lib.oracle_sha512.argtypes = [P, c_size_t, P]
        lib.oracle_set_suite_desc.argtypes = [c_int, P, c_size_t, P, c_size_t, P, P]
        lib.oracle_set_suite_desc.restype = c_int
        lib.oracle_set_suite_desc2.argtypes = [c_int, P, c_size_t, P, c_size_t, P, P, c_int, c_int]
        lib.oracle_set_suite_desc2.restype = c_int
        lib.oracle_set_check_mask.argtypes = [c_int]
        lib.oracle_pairing_check2.argtypes = [P, P]
        lib.oracle_pairing_check2.restype = c_int
        lib.oracle_pairing_check2_batch.argtypes = [c_size_t, P, P, c_size_t, P, c_int]
        lib.oracle_pairing_check2_batch.restype = None
        lib.p256_ietf_verify_batch.argtypes = [c_size_t, P, P, P, P, P, P, c_size_t, P, c_int]
        lib.p256_ietf_verify_batch.restype = None
        lib.p256_ietf_prove_batch.argtypes = [c_size_t, P, P, c_size_t, P, P, c_size_t, P, P, P, P, P, P, c_int]
        lib.p256_ietf_prove_batch.restype = None
        lib.p256_set_blinding_base.argtypes = [P, P]
        lib.p256_pedersen_verify_batch.argtypes = [c_size_t, P, P, P, P, P, P, P, P, c_size_t, P, c_int]
        lib.p256_pedersen_verify_batch.restype = None
        lib.p256_pedersen_prove_batch.argtypes = [c_size_t, P, P, c_size_t, P, P, c_size_t, P, P, P, P, P, P, P, P, P, c_int]
        lib.p256_pedersen_prove_batch.restype = None
        lib.p256_secret_from_seed.argtypes = [P, c_size_t, P]
        lib.p256_public_from_secret.argtypes = [P, P]
        lib.p256_hash_to_curve.argtypes = [P, c_size_t, P]
        lib.p256_output_hash.argtypes = [P, P]
        lib.p256_point_decode.argtypes = [P]
        lib.p256_msm.argtypes = [c_size_t, P, P, P, P]
        lib.p256_msm.restype = c_int
        lib.oracle_g1_mul.argtypes = [P, P, P]
        lib.oracle_g1_mul.restype = c_int
        lib.oracle_g2_mul.argtypes = [P, P, P]
        lib.oracle_g2_mul.restype = c_int
        lib.oracle_g1_add.argtypes = [P, P, P]
        lib.oracle_g1_add.restype = c_int
        _lib = lib
    return _lib


def set_suite(suite_id: int) -> None:
    """1 = Bandersnatch_SHA-512_ELL2 (default), 2 = JubJub_SHA-512_TAI, 3 = Ed25519_SHA-512_TAI,
    4 = BabyJubJub_SHA-512_TAI (process-global selection)."""
    if load().oracle_set_suite(int(suite_id)) != 0:
        raise ValueError("unknown suite")


def set_suite_desc(curve: int, suite_id: bytes, h2c_dst: bytes, g_xy: bytes, b_xy: bytes, challenge_len: int = None,
                   flags: int = 0) -> None:
    """Select a suite given as a descriptor (the fields of include/vrfhip.h vrfhip_suite_desc): curve 1 = Bandersnatch,
    2 = JubJub, 3 = Ed25519, 4 = Baby-JubJub; g_xy / b_xy = generator / blinding base as x || y, 32-byte little-endian;
    challenge_len defaults to the curve's built-in suite (16 for Ed25519, else 32); flags = vrfhip_suite_desc.flags.
    set_suite(1) goes back."""
    a = lambda b: np.frombuffer(bytes(b) + b"\0", np.uint8)
    sid, dst, g, bb = a(suite_id), a(h2c_dst), a(g_xy), a(b_xy)
    assert len(g_xy) == 64 and len(b_xy) == 64
    if challenge_len is None:
        challenge_len = 16 if int(curve) == 3 else 32
    if load().oracle_set_suite_desc2(int(curve), sid.ctypes.data, len(suite_id), dst.ctypes.data, len(h2c_dst),
                                     g.ctypes.data, bb.ctypes.data, int(challenge_len), int(flags)) != 0:
        raise ValueError("bad suite descriptor")


def set_check_mask(mask: int) -> None:
    """Which point classes the verifiers subgroup-check on decode (arkworks' checked deserialisation): bit 1 public
    key, 2 input, 4 output, 8 proof points; 15 (default) = all, as upstream; 0 = on-curve only.  Process-global."""
    load().oracle_set_check_mask(int(mask))


def _a(x):
    return np.ascontiguousarray(x, dtype=np.uint8)


def ietf_verify_batch(pk, h, gamma, c, s, ad: bytes = b"", threads: int = 1) -> np.ndarray:
    pk, h, gamma, c, s = (_a(x).reshape(-1, 32) for x in (pk, h, gamma, c, s))
    n = pk.shape[0]
    st = np.empty(n, dtype=np.uint8)
    adb = np.frombuffer(bytes(ad) + b"\0", dtype=np.uint8)
    load().oracle_ietf_verify_batch(n, pk.ctypes.data, h.ctypes.data, gamma.ctypes.data, c.ctypes.data,
                                    s.ctypes.data, adb.ctypes.data, len(ad), st.ctypes.data, threads)
    return st


def ietf_prove_batch(sk, msgs: np.ndarray = None, inputs=None, ad: bytes = b"", threads: int = 1):
    sk = _a(sk).reshape(-1, 32)
    n = sk.shape[0]
    res = {k: np.empty((n, 32), dtype=np.uint8) for k in ("output", "c", "s", "pk", "input")}
    st = np.empty(n, dtype=np.uint8)
    adb = np.frombuffer(bytes(ad) + b"\0", dtype=np.uint8)
    mp, ml, ip = None, 0, None
    if inputs is not None:
        inputs = _a(inputs).reshape(n, 32)
        ip = inputs.ctypes.data
    else:
        msgs = _a(msgs).reshape(n, -1)
        ml = msgs.shape[1]
        msgs = np.concatenate([msgs.reshape(-1), np.zeros(1, np.uint8)])
        mp = msgs.ctypes.data
    load().oracle_ietf_prove_batch(n, sk.ctypes.data, mp, ml, ip, adb.ctypes.data, len(ad),
                                   res["output"].ctypes.data, res["c"].ctypes.data, res["s"].ctypes.data,
                                   res["pk"].ctypes.data, res["input"].ctypes.data, st.ctypes.data, threads)
    res["status"] = st
    return res


def pedersen_prove_batch(sk, msgs: np.ndarray = None, inputs=None, ad: bytes = b"", threads: int = 1):
    """-> dict(output, pk_com, r, ok, s, sb, blinding, input, status)"""
    sk = _a(sk).reshape(-1, 32)
    n = sk.shape[0]
    gamma = np.empty((n, 32), np.uint8); proof = np.empty((n, 160), np.uint8)
    bl = np.empty((n, 32), np.uint8); hh = np.empty((n, 32), np.uint8); st = np.empty(n, np.uint8)
    adb = np.frombuffer(bytes(ad) + b"\0", dtype=np.uint8)
    mp, ml, ip = None, 0, None
    if inputs is not None:
        inputs = _a(inputs).reshape(n, 32)
        ip = inputs.ctypes.data
    else:
        msgs = _a(msgs).reshape(n, -1)
        ml = msgs.shape[1]
        msgs = np.concatenate([msgs.reshape(-1), np.zeros(1, np.uint8)])
        mp = msgs.ctypes.data
    load().oracle_pedersen_prove_batch(n, sk.ctypes.data, mp, ml, ip, adb.ctypes.data, len(ad), gamma.ctypes.data,
                                       proof.ctypes.data, bl.ctypes.data, hh.ctypes.data, st.ctypes.data, threads)
    return dict(output=gamma, pk_com=proof[:, 0:32].copy(), r=proof[:, 32:64].copy(), ok=proof[:, 64:96].copy(),
                s=proof[:, 96:128].copy(), sb=proof[:, 128:160].copy(), blinding=bl, input=hh, status=st)


def pedersen_verify_batch(h, gamma, pk_com, r, ok, s, sb, ad: bytes = b"", threads: int = 1) -> np.ndarray:
    arrs = [_a(x).reshape(-1, 32) for x in (h, gamma, pk_com, r, ok, s, sb)]
    n = arrs[0].shape[0]
    proof = np.ascontiguousarray(np.concatenate(arrs[2:], axis=1))
    st = np.empty(n, np.uint8)
    adb = np.frombuffer(bytes(ad) + b"\0", dtype=np.uint8)
    load().oracle_pedersen_verify_batch(n, arrs[0].ctypes.data, arrs[1].ctypes.data, proof.ctypes.data,
                                        adb.ctypes.data, len(ad), st.ctypes.data, threads)
    return st


def pedersen_rlc_check(h, gamma, pk_com, r, ok, s, sb, seed: bytes, ad: bytes = b"", index0: int = 0):
    """Batch equation of the random-linear-combination verifier, evaluated naively.
    Returns (status, fail): status[i] in {0, 2}; fail = 1 if the weighted sum is not the identity."""
    arrs = [_a(x).reshape(-1, 32) for x in (h, gamma, pk_com, r, ok, s, sb)]
    n = arrs[0].shape[0]
    proof = np.ascontiguousarray(np.concatenate(arrs[2:], axis=1))
    st = np.empty(n, np.uint8)
    adb = np.frombuffer(bytes(ad) + b"\0", dtype=np.uint8)
    sd = np.frombuffer(bytes(seed), dtype=np.uint8)
    assert sd.size == 32
    fail = load().oracle_pedersen_rlc_check(n, arrs[0].ctypes.data, arrs[1].ctypes.data, proof.ctypes.data,
                                            adb.ctypes.data, len(ad), sd.ctypes.data, index0, st.ctypes.data)
    return st, int(fail)


def batch_digest(arrays, ad=None, index0: int = 0) -> bytes:
    """Batch digest of the random-linear-combination verifiers (csrc/digest.cuh): arrays = list of (n, w_j) uint8
    arrays; ad = None, bytes (shared by all items) or a list of n byte strings."""
    arrs = [np.ascontiguousarray(a, dtype=np.uint8) for a in arrays]
    n = arrs[0].shape[0]
    ptrs = (ctypes.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    widths = np.array([a.size // n for a in arrs], np.uint32)
    blob, off, ad_len = np.zeros(1, np.uint8), None, 0
    if isinstance(ad, (bytes, bytearray)):
        blob, ad_len = np.frombuffer(bytes(ad) + b"\0", np.uint8), len(ad)
    elif ad is not None:
        lens = [len(a) for a in ad]
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
        blob = np.frombuffer(b"".join(bytes(a) for a in ad) + b"\0", np.uint8)
    root = np.empty(32, np.uint8)
    lib = load()
    P = c_void_p
    lib.oracle_batch_digest.argtypes = [c_size_t, ctypes.c_uint64, ctypes.c_int, P, P, P, P, P, ctypes.c_uint32, P]
    lib.oracle_batch_digest.restype = None
    lib.oracle_batch_digest(n, index0, len(arrs), ctypes.cast(ptrs, ctypes.c_void_p), widths.ctypes.data, widths.ctypes.data,
                            blob.ctypes.data, off.ctypes.data if off is not None else None, ad_len, root.ctypes.data)
    return root.tobytes()


def msm(bases_xy, scalars):
    """Naive sum_i k_i * P_i.  Returns (point32, xy64) or None on invalid input."""
    b = _a(bases_xy).reshape(-1, 64)
    k = _a(scalars).reshape(-1, 32)
    out, xy = np.empty(32, np.uint8), np.empty(64, np.uint8)
    b2 = np.concatenate([b.reshape(-1), np.zeros(1, np.uint8)])
    k2 = np.concatenate([k.reshape(-1), np.zeros(1, np.uint8)])
    rc = load().oracle_msm(b.shape[0], b2.ctypes.data, k2.ctypes.data, out.ctypes.data, xy.ctypes.data)
    return None if rc else (out.tobytes(), xy.tobytes())


def hash_to_curve(msg: bytes) -> bytes:
    out = np.empty(32, np.uint8)
    m = np.frombuffer(bytes(msg) + b"\0", np.uint8)
    load().oracle_hash_to_curve(m.ctypes.data, len(msg), out.ctypes.data)
    return out.tobytes()


def output_hash(gamma: bytes) -> bytes:
    out = np.empty(64, np.uint8)
    g = np.frombuffer(bytes(gamma), np.uint8)
    load().oracle_output_hash(g.ctypes.data, out.ctypes.data)
    return out.tobytes()


def secret_from_seed(seed: bytes) -> bytes:
    out = np.empty(32, np.uint8)
    m = np.frombuffer(bytes(seed) + b"\0", np.uint8)
    load().oracle_secret_from_seed(m.ctypes.data, len(seed), out.ctypes.data)
    return out.tobytes()


def public_from_secret(sk: bytes) -> bytes:
    out = np.empty(32, np.uint8)
    m = np.frombuffer(bytes(sk), np.uint8)
    load().oracle_public_from_secret(m.ctypes.data, out.ctypes.data)
    return out.tobytes()


def point_decode(enc: bytes, subgroup: bool = False):
    """Returns (x, y) ints or None."""
    out = np.empty(64, np.uint8)
    m = np.frombuffer(bytes(enc), np.uint8)
    rc = load().oracle_point_decode(m.ctypes.data, int(subgroup), out.ctypes.data)
    if rc != 0:
        return None
    return int.from_bytes(out[:32].tobytes(), "little"), int.from_bytes(out[32:].tobytes(), "little")


def fq_mul(a: bytes, b: bytes) -> bytes:
    out = np.empty(32, np.uint8)
    x, y = np.frombuffer(bytes(a), np.uint8), np.frombuffer(bytes(b), np.uint8)
    load().oracle_fq_mul(x.ctypes.data, y.ctypes.data, out.ctypes.data)
    return out.tobytes()


def sha512(m: bytes) -> bytes:
    out = np.empty(64, np.uint8)
    x = np.frombuffer(bytes(m) + b"\0", np.uint8)
    load().oracle_sha512(x.ctypes.data, len(m), out.ctypes.data)
    return out.tobytes()


# ---- BLS12-381 pairing-product check (oracle/c/oracle_bls.c): the native twin of oracle/bls_oracle.py ----
def pairing_check_batch(g1, g2, shared: bool = False, threads: int = 1) -> np.ndarray:
    """status per item of e(P0, Q0) e(P1, Q1) == 1 (0 / 1 / 2 as vrfhip_pairing_check_batch).  g1: (n, 192) bytes;
    g2: (n, 384) bytes, or one 384-byte pair when shared."""
    g1 = _a(g1).reshape(-1, 192)
    g2 = _a(g2).reshape(-1, 384)
    n = g1.shape[0]
    assert g2.shape[0] == (1 if shared else n)
    st = np.empty(n, np.uint8)
    load().oracle_pairing_check2_batch(n, g1.ctypes.data, g2.ctypes.data, 0 if shared else 384, st.ctypes.data, int(threads))
    return st


def g1_mul(k: int, pt96: bytes) -> bytes:
    out = np.empty(96, np.uint8)
    kk, p = np.frombuffer(int(k).to_bytes(32, "little"), np.uint8), np.frombuffer(bytes(pt96), np.uint8)
    if load().oracle_g1_mul(kk.ctypes.data, p.ctypes.data, out.ctypes.data) != 0:
        raise ValueError("G1 point does not decode")
    return out.tobytes()


def g2_mul(k: int, pt192: bytes) -> bytes:
    out = np.empty(192, np.uint8)
    kk, p = np.frombuffer(int(k).to_bytes(32, "little"), np.uint8), np.frombuffer(bytes(pt192), np.uint8)
    if load().oracle_g2_mul(kk.ctypes.data, p.ctypes.data, out.ctypes.data) != 0:
        raise ValueError("G2 point does not decode")
    return out.tobytes()


def g1_add(a96: bytes, b96: bytes) -> bytes:
    out = np.empty(96, np.uint8)
    x, y = np.frombuffer(bytes(a96), np.uint8), np.frombuffer(bytes(b96), np.uint8)
    if load().oracle_g1_add(x.ctypes.data, y.ctypes.data, out.ctypes.data) != 0:
        raise ValueError("G1 point does not decode")
    return out.tobytes()


# ---- secp256r1 (oracle/c/oracle_p256.c; wire format: 33-byte Sec1 points, 32-byte big-endian scalars) ----
def p256_ietf_verify_batch(pk, h, gamma, c, s, ad: bytes = b"", threads: int = 1) -> np.ndarray:
    pk, h, gamma = (_a(x).reshape(-1, 33) for x in (pk, h, gamma))
    c, s = (_a(x).reshape(-1, 32) for x in (c, s))
    n = pk.shape[0]
    st = np.empty(n, dtype=np.uint8)
    adb = np.frombuffer(bytes(ad) + b"\0", dtype=np.uint8)
    load().p256_ietf_verify_batch(n, pk.ctypes.data, h.ctypes.data, gamma.ctypes.data, c.ctypes.data, s.ctypes.data,
                                  adb.ctypes.data, len(ad), st.ctypes.data, threads)
    return st


def p256_ietf_prove_batch(sk, msgs: np.ndarray = None, inputs=None, ad: bytes = b"", threads: int = 1):
    sk = _a(sk).reshape(-1, 32)
    n = sk.shape[0]
    res = {k: np.empty((n, 32 if k in ("c", "s") else 33), dtype=np.uint8) for k in ("output", "c", "s", "pk", "input")}
    st = np.empty(n, dtype=np.uint8)
    adb = np.frombuffer(bytes(ad) + b"\0", dtype=np.uint8)
    mp, ml, ip = None, 0, None
    if inputs is not None:
        inputs = _a(inputs).reshape(n, 33)
        ip = inputs.ctypes.data
    else:
        msgs = _a(msgs).reshape(n, -1)
        ml = msgs.shape[1]
        msgs = np.concatenate([msgs.reshape(-1), np.zeros(1, np.uint8)])
        mp = msgs.ctypes.data
    load().p256_ietf_prove_batch(n, sk.ctypes.data, mp, ml, ip, adb.ctypes.data, len(ad), res["output"].ctypes.data,
                                 res["c"].ctypes.data, res["s"].ctypes.data, res["pk"].ctypes.data, res["input"].ctypes.data,
                                 st.ctypes.data, threads)
    res["status"] = st
    return res


def p256_secret_from_seed(seed: bytes) -> bytes:
    out = ctypes.create_string_buffer(32)
    load().p256_secret_from_seed(bytes(seed), len(seed), out)
    return out.raw


def p256_public_from_secret(sk_be: bytes) -> bytes:
    out = ctypes.create_string_buffer(33)
    load().p256_public_from_secret(bytes(sk_be), out)
    return out.raw


def p256_hash_to_curve(data: bytes) -> bytes:
    out = ctypes.create_string_buffer(33)
    load().p256_hash_to_curve(bytes(data), len(data), out)
    return out.raw


def p256_output_hash(gamma33: bytes) -> bytes:
    out = ctypes.create_string_buffer(32)
    load().p256_output_hash(bytes(gamma33), out)
    return out.raw


def p256_point_decode(enc33: bytes) -> int:
    return load().p256_point_decode(bytes(enc33))


def p256_msm(bases_xy, scalars_be):
    """sum_i k_i P_i on secp256r1 (oracle_p256.c p256_msm): bases (n, 64) x || y little-endian, scalars (n, 32) big-endian.
    Returns (status, sec1 33 bytes, xy 64 bytes)."""
    b = np.ascontiguousarray(bases_xy, dtype=np.uint8).reshape(-1, 64)
    k = np.ascontiguousarray(scalars_be, dtype=np.uint8).reshape(-1, 32)
    o33, oxy = ctypes.create_string_buffer(33), ctypes.create_string_buffer(64)
    st = load().p256_msm(b.shape[0], b.ctypes.data if b.size else None, k.ctypes.data if k.size else None, o33, oxy)
    return st, o33.raw, oxy.raw


def p256_set_blinding_base(pt) -> None:
    """The Pedersen blinding base (affine (x, y) ints) the p256_pedersen_* functions use."""
    load().p256_set_blinding_base(int(pt[0]).to_bytes(32, "big"), int(pt[1]).to_bytes(32, "big"))


def p256_pedersen_prove_batch(sk, msgs: np.ndarray = None, inputs=None, ad: bytes = b"", threads: int = 1):
    sk = _a(sk).reshape(-1, 32)
    n = sk.shape[0]
    res = {k: np.empty((n, 32 if k in ("s", "sb", "blinding") else 33), dtype=np.uint8)
           for k in ("output", "pk_com", "r", "ok", "s", "sb", "blinding", "input")}
    st = np.empty(n, dtype=np.uint8)
    adb = np.frombuffer(bytes(ad) + b"\0", dtype=np.uint8)
    mp, ml, ip = None, 0, None
    if inputs is not None:
        inputs = _a(inputs).reshape(n, 33)
        ip = inputs.ctypes.data
    else:
        msgs = _a(msgs).reshape(n, -1)
        ml = msgs.shape[1]
        msgs = np.concatenate([msgs.reshape(-1), np.zeros(1, np.uint8)])
        mp = msgs.ctypes.data
    load().p256_pedersen_prove_batch(n, sk.ctypes.data, mp, ml, ip, adb.ctypes.data, len(ad), *[res[k].ctypes.data for k in
                                     ("output", "pk_com", "r", "ok", "s", "sb", "blinding", "input")], st.ctypes.data, threads)
    res["status"] = st
    return res


def p256_pedersen_verify_batch(h, gamma, pk_com, r, ok, s, sb, ad: bytes = b"", threads: int = 1) -> np.ndarray:
    pts = [_a(x).reshape(-1, 33) for x in (h, gamma, pk_com, r, ok)]
    sc = [_a(x).reshape(-1, 32) for x in (s, sb)]
    n = pts[0].shape[0]
    st = np.empty(n, dtype=np.uint8)
    adb = np.frombuffer(bytes(ad) + b"\0", dtype=np.uint8)
    load().p256_pedersen_verify_batch(n, *[x.ctypes.data for x in pts + sc], adb.ctypes.data, len(ad), st.ctypes.data, threads)
    return st


# ------------------------------------------------------------------ `suites::bandersnatch_sw` (oracle/c/oracle_bsw.c)
# 33-byte arkworks short-Weierstrass points, little-endian scalars; arithmetic on the Weierstrass curve itself.
def _bsw():
    lib = load()
    if not getattr(lib, "_bsw_typed", False):
        P = c_void_p
        lib.oracle_bsw_ietf_verify_batch.argtypes = [c_size_t, P, P, P, P, P, P, c_size_t, P, c_int]
        lib.oracle_bsw_ietf_verify_batch.restype = None
        lib.oracle_bsw_ietf_prove_batch.argtypes = [c_size_t, P, P, c_size_t, P, P, c_size_t, P, P, P, P, P, P, c_int]
        lib.oracle_bsw_ietf_prove_batch.restype = None
        lib.oracle_bsw_pedersen_verify_batch.argtypes = [c_size_t, P, P, P, P, c_size_t, P, c_int]
        lib.oracle_bsw_pedersen_verify_batch.restype = None
        lib.oracle_bsw_pedersen_prove_batch.argtypes = [c_size_t, P, P, c_size_t, P, P, c_size_t, P, P, P, P, P, c_int]
        lib.oracle_bsw_pedersen_prove_batch.restype = None
        lib.oracle_bsw_secret_public.argtypes = [P, c_size_t, P, P]
        lib.oracle_bsw_hash_to_curve.argtypes = [P, c_size_t, P]
        lib.oracle_bsw_hash_to_curve.restype = c_int
        lib.oracle_bsw_output_hash.argtypes = [P, P]
        lib.oracle_bsw_point_decode.argtypes = [P, c_int, P]
        lib.oracle_bsw_point_decode.restype = c_int
        lib.oracle_bsw_constants.argtypes = [P, P, P]
        lib._bsw_typed = True
    return lib


def bsw_constants():
    """(a', b'), generator, blinding base as the C oracle derives them: three 64-byte x || y strings."""
    ab, g, bb = (ctypes.create_string_buffer(64) for _ in range(3))
    _bsw().oracle_bsw_constants(ab, g, bb)
    return ab.raw, g.raw, bb.raw


def bsw_secret_public(seed: bytes):
    sk, pk = ctypes.create_string_buffer(32), ctypes.create_string_buffer(33)
    _bsw().oracle_bsw_secret_public(bytes(seed), len(seed), sk, pk)
    return sk.raw, pk.raw


def bsw_hash_to_curve(msg: bytes) -> bytes:
    out = ctypes.create_string_buffer(33)
    _bsw().oracle_bsw_hash_to_curve(bytes(msg), len(msg), out)
    return out.raw


def bsw_output_hash(gamma: bytes) -> bytes:
    out = ctypes.create_string_buffer(64)
    _bsw().oracle_bsw_output_hash(bytes(gamma), out)
    return out.raw


def bsw_point_decode(enc: bytes, subgroup: bool = True):
    """x || y (64 bytes; zeros for the point at infinity) or None."""
    xy = ctypes.create_string_buffer(64)
    return xy.raw if _bsw().oracle_bsw_point_decode(bytes(enc), 1 if subgroup else 0, xy) == 0 else None


def bsw_ietf_prove_batch(sk, msgs: np.ndarray = None, inputs=None, ad: bytes = b"", threads: int = 1):
    sk = _a(sk).reshape(-1, 32)
    n = sk.shape[0]
    res = {k: np.empty((n, 32 if k in ("c", "s") else 33), dtype=np.uint8) for k in ("output", "c", "s", "pk", "input")}
    st = np.empty(n, dtype=np.uint8)
    adb = np.frombuffer(bytes(ad) + b"\0", dtype=np.uint8)
    mp, ml, ip = None, 0, None
    if inputs is not None:
        inputs = _a(inputs).reshape(n, 33)
        ip = inputs.ctypes.data
    else:
        msgs = _a(msgs).reshape(n, -1)
        ml = msgs.shape[1]
        msgs = np.concatenate([msgs.reshape(-1), np.zeros(1, np.uint8)])
        mp = msgs.ctypes.data
    _bsw().oracle_bsw_ietf_prove_batch(n, sk.ctypes.data, mp, ml, ip, adb.ctypes.data, len(ad), res["output"].ctypes.data,
                                       res["c"].ctypes.data, res["s"].ctypes.data, res["pk"].ctypes.data, res["input"].ctypes.data,
                                       st.ctypes.data, threads)
    res["status"] = st
    return res


def bsw_ietf_verify_batch(pk, h, gamma, c, s, ad: bytes = b"", threads: int = 1) -> np.ndarray:
    pts = [_a(x).reshape(-1, 33) for x in (pk, h, gamma)]
    sc = [_a(x).reshape(-1, 32) for x in (c, s)]
    n = pts[0].shape[0]
    st = np.empty(n, dtype=np.uint8)
    adb = np.frombuffer(bytes(ad) + b"\0", dtype=np.uint8)
    _bsw().oracle_bsw_ietf_verify_batch(n, *[x.ctypes.data for x in pts + sc], adb.ctypes.data, len(ad), st.ctypes.data, threads)
    return st


def bsw_pedersen_prove_batch(sk, msgs: np.ndarray = None, inputs=None, ad: bytes = b"", threads: int = 1):
    """dict(output, pk_com, r, ok (n x 33), s, sb, blinding (n x 32), input, status)."""
    sk = _a(sk).reshape(-1, 32)
    n = sk.shape[0]
    gamma, proof = np.empty((n, 33), np.uint8), np.empty((n, 163), np.uint8)
    blind, hh, st = np.empty((n, 32), np.uint8), np.empty((n, 33), np.uint8), np.empty(n, np.uint8)
    adb = np.frombuffer(bytes(ad) + b"\0", dtype=np.uint8)
    mp, ml, ip = None, 0, None
    if inputs is not None:
        inputs = _a(inputs).reshape(n, 33)
        ip = inputs.ctypes.data
    else:
        msgs = _a(msgs).reshape(n, -1)
        ml = msgs.shape[1]
        msgs = np.concatenate([msgs.reshape(-1), np.zeros(1, np.uint8)])
        mp = msgs.ctypes.data
    _bsw().oracle_bsw_pedersen_prove_batch(n, sk.ctypes.data, mp, ml, ip, adb.ctypes.data, len(ad), gamma.ctypes.data, proof.ctypes.data,
                                           blind.ctypes.data, hh.ctypes.data, st.ctypes.data, threads)
    return {"output": gamma, "pk_com": proof[:, :33].copy(), "r": proof[:, 33:66].copy(), "ok": proof[:, 66:99].copy(),
            "s": proof[:, 99:131].copy(), "sb": proof[:, 131:163].copy(), "blinding": blind, "input": hh, "status": st}


def bsw_pedersen_verify_batch(h, gamma, pk_com, r, ok, s, sb, ad: bytes = b"", threads: int = 1) -> np.ndarray:
    h, gamma = _a(h).reshape(-1, 33), _a(gamma).reshape(-1, 33)
    n = h.shape[0]
    proof = np.ascontiguousarray(np.concatenate([_a(x).reshape(n, -1) for x in (pk_com, r, ok, s, sb)], axis=1))
    assert proof.shape[1] == 163
    st = np.empty(n, dtype=np.uint8)
    adb = np.frombuffer(bytes(ad) + b"\0", dtype=np.uint8)
    _bsw().oracle_bsw_pedersen_verify_batch(n, h.ctypes.data, gamma.ctypes.data, proof.ctypes.data, adb.ctypes.data, len(ad),
                                            st.ctypes.data, threads)
    return st
