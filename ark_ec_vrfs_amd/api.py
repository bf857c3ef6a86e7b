"""Python mirror of the ark-ec-vrfs / ark-vrf API for the IETF VRF hot path.

Rust name (src/lib.rs:13-17)            here
--------------------------------------  -------------------------------------------------
Suite (suites::bandersnatch)            BandersnatchSha512Ell2 (class attribute constants)
Secret::from_seed / from_scalar/public  Secret.from_seed / Secret.from_scalar / .public()
Input::new(data) -> Option<Input>       Input.new(data)
Secret::output(input)                   Secret.output(input)
Output::hash()                          Output.hash()
ietf::Prover::prove(&sk, in, out, ad)   ietf.Prover.prove(secret, input, output, ad)
ietf::Verifier::verify(&pk, ...)        ietf.Verifier.verify(public, input, output, ad, proof)
Error::{VerificationFailure,InvalidData} exceptions VerificationFailure / InvalidData
(batch forms, no Rust counterpart)      Context.ietf_prove_batch / ietf_verify_batch / ...

Single-item methods are batches of one; the batch methods are the product.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib

ST_OK, ST_VERIFICATION_FAILURE, ST_INVALID_DATA = 0, 1, 2


class Error(Exception):
    """`Error` (src/lib.rs:15)."""


class VerificationFailure(Error):
    pass


class InvalidData(Error):
    pass


class Suite:
    """`Suite` (src/lib.rs:16): compile-time constants of a cipher suite."""
    SUITE_ID: bytes = b""
    CHALLENGE_LEN: int = 32
    SUITE_ENUM: int = 0


class BandersnatchSha512Ell2(Suite):
    SUITE_ID = b"Bandersnatch_SHA-512_ELL2"
    CHALLENGE_LEN = 32
    SUITE_ENUM = 1


class JubJubSha512Tai(Suite):
    """`suites::jubjub` as SURVEY.md A.6 recalls it (parity unpinned)."""
    SUITE_ID = b"JubJub_SHA-512_TAI"
    CHALLENGE_LEN = 32
    SUITE_ENUM = 2


class Ed25519Sha512Tai(Suite):
    """`suites::ed25519`: edwards25519, try-and-increment, 16-byte challenge (suite string / blinding base unpinned)."""
    SUITE_ID = b"Ed25519_SHA-512_TAI"
    CHALLENGE_LEN = 16
    SUITE_ENUM = 3


class BabyJubJubSha512Tai(Suite):
    """`suites::baby_jubjub`: ark-ed-on-bn254 (a = 1), try-and-increment (suite string / blinding base unpinned)."""
    SUITE_ID = b"BabyJubJub_SHA-512_TAI"
    CHALLENGE_LEN = 32
    SUITE_ENUM = 4


class Secp256r1Sha256Tai(Suite):
    """`suites::secp256r1` = RFC 9381 ECVRF-P256-SHA256-TAI (suite_string 0x01).  Own wire format: points are 33-byte SEC1
    compressed strings, scalars 32-byte BIG-endian integers, `Output::hash` is 32 bytes (SHA-256).  Both schemes; the
    Pedersen scheme needs the caller's `PedersenSuite::BLINDING_BASE` in the descriptor (upstream's constant is not pinned
    here: the default descriptor leaves it zero and pedersen_* then raise)."""
    SUITE_ID = b"\x01"
    CHALLENGE_LEN = 16
    SUITE_ENUM = 5


class BandersnatchSwSha512Tai(Suite):
    """`suites::bandersnatch_sw`: the Bandersnatch group on its short-Weierstrass model, try-and-increment.  Own wire format:
    points are arkworks' 33-byte compressed short-Weierstrass strings (x little-endian, then a flag byte); scalars and the
    output hash as for the twisted-Edwards suite.  Parity unpinned (include/vrfhip.h); the Pedersen scheme needs a blinding
    base in the descriptor."""
    SUITE_ID = b"Bandersnatch_SW_SHA-512_TAI"
    CHALLENGE_LEN = 32
    SUITE_ENUM = 6


CURVE_BANDERSNATCH, CURVE_JUBJUB, CURVE_ED25519, CURVE_BABY_JUBJUB, CURVE_SECP256R1, CURVE_BANDERSNATCH_SW = 1, 2, 3, 4, 5, 6
# vrfhip_suite_desc.flags (VRFHIP_SUITE_FLAG_*): what separates RFC 9381's edwards suites from upstream's built-in ones
SUITE_FLAG_SIGN_PARITY, SUITE_FLAG_CHALLENGE_LE, SUITE_FLAG_HASH_COFACTOR = 1, 2, 4


@dataclass
class SuiteDesc:
    """`vrfhip_suite_desc`: what a `Suite` / `PedersenSuite` impl states as data (src/lib.rs:16): SUITE_ID, the
    hash-to-curve DST, the generator and the Pedersen blinding base (x || y, 32-byte little-endian each)."""
    curve: int
    suite_id: bytes
    h2c_dst: bytes
    generator: bytes
    blinding_base: bytes
    challenge_len: int = 32
    flags: int = 0

    @staticmethod
    def default(suite: type) -> "SuiteDesc":
        """The built-in descriptor of a suite class (what Context(suite=...) uses)."""
        d = _lib.SuiteDescStruct()
        _lib.check(_lib.load().vrfhip_suite_desc_default(suite.SUITE_ENUM, ctypes.byref(d)), "vrfhip_suite_desc_default")
        return SuiteDesc._from_struct(d)

    @staticmethod
    def test_blinding_base(suite: type) -> bytes:
        """vrfhip_test_blinding_base: a nothing-up-my-sleeve subgroup point (x || y) -- NOT upstream's constant; for tests
        and bench legs of suites whose `PedersenSuite::BLINDING_BASE` is unpinned."""
        out = (ctypes.c_uint8 * 64)()
        _lib.check(_lib.load().vrfhip_test_blinding_base(suite.SUITE_ENUM, out), "vrfhip_test_blinding_base")
        return bytes(out)

    @staticmethod
    def with_test_blinding_base(suite: type) -> "SuiteDesc":
        d = SuiteDesc.default(suite)
        d.blinding_base = SuiteDesc.test_blinding_base(suite)
        return d

    @staticmethod
    def _from_struct(d) -> "SuiteDesc":
        return SuiteDesc(int(d.curve), bytes(d.suite_id[:d.suite_id_len]), bytes(d.h2c_dst[:d.h2c_dst_len]),
                         bytes(d.generator), bytes(d.blinding_base), int(d.challenge_len), int(d.flags))

    def _to_struct(self):
        if len(self.suite_id) > 64 or len(self.h2c_dst) > 128 or len(self.generator) != 64 or len(self.blinding_base) != 64:
            raise ValueError("descriptor field out of range")
        d = _lib.SuiteDescStruct()
        d.struct_size = ctypes.sizeof(_lib.SuiteDescStruct)
        d.curve = self.curve
        d.suite_id_len = len(self.suite_id)
        ctypes.memmove(d.suite_id, bytes(self.suite_id), len(self.suite_id))
        d.h2c_dst_len = len(self.h2c_dst)
        ctypes.memmove(d.h2c_dst, bytes(self.h2c_dst), len(self.h2c_dst))
        ctypes.memmove(d.generator, bytes(self.generator), 64)
        ctypes.memmove(d.blinding_base, bytes(self.blinding_base), 64)
        d.challenge_len = self.challenge_len
        d.flags = self.flags
        return d


class PinnedBuffer:
    """Page-locked host memory (vrfhip_host_alloc) viewed as a numpy uint8 array: arrays handed to the host-pointer
    entry points from here leave by DMA as they lie instead of through the pinned staging ring."""

    def __init__(self, shape):
        self.shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        n = int(np.prod(self.shape))
        self._p = ctypes.c_void_p()
        _lib.check(_lib.load().vrfhip_host_alloc(max(n, 1), ctypes.byref(self._p)), "vrfhip_host_alloc")
        self.array = np.ctypeslib.as_array((ctypes.c_uint8 * max(n, 1)).from_address(self._p.value))[:n].reshape(self.shape)

    def free(self) -> None:
        if self._p:
            self.array = None
            _lib.load().vrfhip_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _np_u8(b, n_bytes: Optional[int] = None) -> np.ndarray:
    a = np.frombuffer(bytes(b), dtype=np.uint8) if not isinstance(b, np.ndarray) else b
    a = np.ascontiguousarray(a, dtype=np.uint8).reshape(-1)
    if n_bytes is not None and a.size != n_bytes:
        raise ValueError(f"expected {n_bytes} bytes, got {a.size}")
    return a


def _ptr(a: Optional[np.ndarray]) -> Optional[int]:
    return None if a is None else a.ctypes.data


def _pack_var(items: Sequence[bytes]) -> Tuple[np.ndarray, np.ndarray]:
    off = np.zeros(len(items) + 1, dtype=np.uint32)
    off[1:] = np.cumsum([len(x) for x in items], dtype=np.uint64).astype(np.uint32)
    blob = np.frombuffer(b"".join(items) + b"\x00", dtype=np.uint8).copy()
    return blob, off


class Context:
    """One GPU + one suite: owns the device tables and HBM workspace (vrfhip_ctx)."""

    def __init__(self, device: int = 0, suite: type = BandersnatchSha512Ell2, desc: Optional[SuiteDesc] = None,
                 test_blinding_base: bool = False):
        """suite: a built-in suite class; desc: a suite descriptor (overrides `suite`), e.g. the upstream JubJub
        constants filled in by the caller.  test_blinding_base: tests / bench only -- the built-in descriptor with the
        placeholder blinding base of vrfhip_test_blinding_base (the default descriptor of every suite but Bandersnatch has
        none, and no Pedersen scheme)."""
        self._lib = _lib.load()
        self.suite = suite
        if desc is None and test_blinding_base:
            desc = SuiteDesc.with_test_blinding_base(suite)
        h = ctypes.c_void_p()
        if desc is not None:
            d = desc._to_struct()
            _lib.check(self._lib.vrfhip_ctx_create_desc(ctypes.byref(d), device, ctypes.byref(h)), "vrfhip_ctx_create_desc")
        else:
            _lib.check(self._lib.vrfhip_ctx_create(suite.SUITE_ENUM, device, ctypes.byref(h)), "vrfhip_ctx_create")
        self._h = h
        self.device = device

    def desc(self) -> SuiteDesc:
        d = _lib.SuiteDescStruct()
        _lib.check(self._lib.vrfhip_ctx_get_desc(self._h, ctypes.byref(d)), "vrfhip_ctx_get_desc")
        return SuiteDesc._from_struct(d)

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.vrfhip_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self) -> ctypes.c_void_p:
        return self._h

    def reserve(self, max_items: int) -> None:
        _lib.check(self._lib.vrfhip_ctx_reserve(self._h, max_items), "vrfhip_ctx_reserve")

    # VRFHIP_FLAG_PREVALIDATED_* (include/vrfhip.h)
    PREVALIDATED_PUBLIC, PREVALIDATED_INPUT, PREVALIDATED_OUTPUT, PREVALIDATED_PROOF, PREVALIDATED_ALL = 1, 2, 4, 8, 15
    PROVE_POINTS_AFFINE = 16        # VRFHIP_FLAG_PROVE_POINTS_AFFINE: the provers write points as x || y (64 B)
    CT_TABLES = 64                  # VRFHIP_FLAG_CT_TABLES: the provers' per-proof table lookups read all eight entries
    COORDS_MONT256 = 32             # VRFHIP_FLAG_COORDS_MONT256: x || y pairs are arkworks' in-memory Montgomery limbs

    def point_bytes(self) -> int:
        """Bytes of one compressed point on the wire: 32 (ArkworksCodec), 33 for secp256r1 (Sec1Codec)."""
        return int(self._lib.vrfhip_ctx_point_bytes(self._h))

    def hash_bytes(self) -> int:
        """Bytes of `Output::hash`: 64 (SHA-512), 32 for secp256r1 (SHA-256)."""
        return int(self._lib.vrfhip_ctx_hash_bytes(self._h))

    def prove_point_bytes(self) -> int:
        """Bytes per point in the provers' outputs (output, pk / pk_com, r, ok): the wire width, 64 with PROVE_POINTS_AFFINE."""
        return 64 if self.get_flags() & self.PROVE_POINTS_AFFINE else self.point_bytes()

    def set_flags(self, flags: int) -> None:
        """Point classes whose prime-order-subgroup membership the caller vouches for (their test is skipped)."""
        _lib.check(self._lib.vrfhip_ctx_set_flags(self._h, int(flags)), "vrfhip_ctx_set_flags")

    def get_flags(self) -> int:
        return int(self._lib.vrfhip_ctx_get_flags(self._h))

    # test / tuning knobs (vrfhip_debug_set): the library never reads the environment
    PAIRING_LAYOUTS = {"auto": 0, "lane": 1, "quad": 2, "row": 3, "tri": 4, "oct": 5, "oct1": 6, "noprep": 0x100}

    def debug_set(self, key: int, value: int) -> None:
        _lib.check(self._lib.vrfhip_debug_set(self._h, int(key), int(value)), "vrfhip_debug_set")

    def debug_pairing_layout(self, *names: str) -> None:
        """Force a pairing-check layout (tests): any of PAIRING_LAYOUTS, OR-ed; no name = back to the default."""
        v = 0
        for nm in names:
            v |= self.PAIRING_LAYOUTS[nm]
        self.debug_set(1, v)

    def debug_pipeline(self, first_log2: int = 17, chunk_log2: int = 18) -> None:
        self.debug_set(2, first_log2)
        self.debug_set(3, chunk_log2)

    def set_prevalidated(self, on: bool = True) -> None:
        """All points declared validated by the caller (on) / everything checked as arkworks' deserialisation does (off)."""
        self.set_flags(self.PREVALIDATED_ALL if on else 0)

    def profile(self, enable: bool) -> None:
        """Record hipEvents around the three kernels of every prove/verify launch group."""
        _lib.check(self._lib.vrfhip_ctx_profile(self._h, int(enable)), "vrfhip_ctx_profile")

    def profile_read(self):
        """-> ([ms per stage: decode|prepare, straus V|mul, straus U, finish] summed over launch groups, n)."""
        ms = (ctypes.c_double * 4)()
        n = ctypes.c_uint64()
        _lib.check(self._lib.vrfhip_ctx_profile_read(self._h, ms, ctypes.byref(n)), "vrfhip_ctx_profile_read")
        return [ms[0], ms[1], ms[2], ms[3]], int(n.value)

    def workspace_bytes(self) -> int:
        return int(self._lib.vrfhip_ctx_workspace_bytes(self._h))

    # ---- host-buffer batch API (numpy uint8 arrays of shape (n, 32)) -------------------
    @staticmethod
    def _ad_args(ad, n):
        """ad: bytes (shared) or a sequence of n byte strings (per item)."""
        if ad is None:
            ad = b""
        if isinstance(ad, (bytes, bytearray, memoryview)):
            blob = np.frombuffer(bytes(ad) + b"\x00", dtype=np.uint8).copy()
            return blob, None, len(ad)
        if len(ad) != n:
            raise ValueError("per-item ad must have n entries")
        blob, off = _pack_var([bytes(x) for x in ad])
        return blob, off, 0

    def ietf_verify_batch(self, pk, inp, out, c, s, ad=b"") -> np.ndarray:
        pw = self.point_bytes()
        pk, inp, out = (np.ascontiguousarray(x, dtype=np.uint8).reshape(-1, pw) for x in (pk, inp, out))
        c, s = (np.ascontiguousarray(x, dtype=np.uint8).reshape(-1, 32) for x in (c, s))
        n = pk.shape[0]
        if not all(x.shape[0] == n for x in (inp, out, c, s)):
            raise ValueError("ragged batch")
        status = np.empty(n, dtype=np.uint8)
        blob, off, ad_len = self._ad_args(ad, n)
        _lib.check(self._lib.vrfhip_ietf_verify_batch(
            self._h, n, _ptr(pk), _ptr(inp), _ptr(out), _ptr(c), _ptr(s), _ptr(blob), _ptr(off), ad_len,
            _ptr(status)), "vrfhip_ietf_verify_batch")
        return status

    def ietf_verify_batch_alpha(self, pk, msgs, out, c, s, ad=b"") -> np.ndarray:
        """`Input::new(alpha)` + `ietf::Verifier::verify` in one call (vrfhip_ietf_verify_batch_alpha): msgs is a sequence of
        byte strings or an (n, L) uint8 array; H is hashed to the curve on the device and stays there as affine coordinates."""
        pw = self.point_bytes()
        pk, out = (np.ascontiguousarray(x, dtype=np.uint8).reshape(-1, pw) for x in (pk, out))
        c, s = (np.ascontiguousarray(x, dtype=np.uint8).reshape(-1, 32) for x in (c, s))
        n = pk.shape[0]
        if not all(x.shape[0] == n for x in (out, c, s)):
            raise ValueError("ragged batch")
        msg_blob, msg_off, msg_len, _ = self._msg_args(n, msgs, None)
        status = np.empty(n, dtype=np.uint8)
        blob, off, ad_len = self._ad_args(ad, n)
        _lib.check(self._lib.vrfhip_ietf_verify_batch_alpha(
            self._h, n, _ptr(pk), _ptr(msg_blob), _ptr(msg_off), msg_len, _ptr(out), _ptr(c), _ptr(s), _ptr(blob), _ptr(off),
            ad_len, _ptr(status)), "vrfhip_ietf_verify_batch_alpha")
        return status

    # ---- keyed verification: public keys with context-resident fixed-base tables --------------
    def keyset_create(self, pks):
        """`Public` keys -> KeySet (validated points + one 881 KB comb each, resident in HBM).  Returns
        (keyset, status): status[i] = 0, or 2 if key i is not a point of the prime-order subgroup."""
        k = np.ascontiguousarray(pks, dtype=np.uint8).reshape(-1, self.point_bytes())
        st = np.empty(k.shape[0], dtype=np.uint8)
        h = ctypes.c_void_p()
        _lib.check(self._lib.vrfhip_keyset_create(self._h, k.shape[0], _ptr(k), _ptr(st), ctypes.byref(h)),
                   "vrfhip_keyset_create")
        return KeySet(self, h, k.shape[0]), st

    def ietf_verify_batch_keyed(self, keyset, key_index, inp, out, c, s, ad=b"") -> np.ndarray:
        """`ietf::Verifier::verify` against keys of a KeySet; key_index[i] names the key of proof i."""
        idx = np.ascontiguousarray(key_index, dtype=np.uint32).reshape(-1)
        inp, out = (np.ascontiguousarray(x, dtype=np.uint8).reshape(-1, self.point_bytes()) for x in (inp, out))
        c, s = (np.ascontiguousarray(x, dtype=np.uint8).reshape(-1, 32) for x in (c, s))
        n = idx.shape[0]
        if not all(x.shape[0] == n for x in (inp, out, c, s)):
            raise ValueError("ragged batch")
        status = np.empty(n, dtype=np.uint8)
        blob, off, ad_len = self._ad_args(ad, n)
        _lib.check(self._lib.vrfhip_ietf_verify_batch_keyed(
            self._h, keyset.handle, n, _ptr(idx), _ptr(inp), _ptr(out), _ptr(c), _ptr(s), _ptr(blob), _ptr(off), ad_len,
            _ptr(status)), "vrfhip_ietf_verify_batch_keyed")
        return status

    def ietf_verify_batch_keyed_dev(self, keyset, key_index, inp, out, c, s, status, ad=None, ad_off=None, ad_len=0,
                                    stream=None):
        import torch
        st = torch.cuda.current_stream().cuda_stream if stream is None else stream
        _lib.check(self._lib.vrfhip_ietf_verify_batch_keyed_dev(
            self._h, keyset.handle, inp.shape[0], key_index.data_ptr(), inp.data_ptr(), out.data_ptr(), c.data_ptr(),
            s.data_ptr(), None if ad is None else ad.data_ptr(), None if ad_off is None else ad_off.data_ptr(), ad_len,
            status.data_ptr(), st), "vrfhip_ietf_verify_batch_keyed_dev")

    def ietf_verify_batch_affine(self, pk_xy, inp_xy, out_xy, c, s, ad=b"") -> np.ndarray:
        """Verification from in-memory affine points: (n, 64) arrays x || y (LE canonical)."""
        pk, inp, out = (np.ascontiguousarray(x, dtype=np.uint8).reshape(-1, 64) for x in (pk_xy, inp_xy, out_xy))
        c, s = (np.ascontiguousarray(x, dtype=np.uint8).reshape(-1, 32) for x in (c, s))
        n = pk.shape[0]
        if not all(x.shape[0] == n for x in (inp, out, c, s)):
            raise ValueError("ragged batch")
        status = np.empty(n, dtype=np.uint8)
        blob, off, ad_len = self._ad_args(ad, n)
        _lib.check(self._lib.vrfhip_ietf_verify_batch_affine(
            self._h, n, _ptr(pk), _ptr(inp), _ptr(out), _ptr(c), _ptr(s), _ptr(blob), _ptr(off), ad_len,
            _ptr(status)), "vrfhip_ietf_verify_batch_affine")
        return status

    def ietf_verify_batch_affine_dev(self, pk_xy, inp_xy, out_xy, c, s, status, ad=None, ad_off=None, ad_len=0,
                                     stream=None):
        import torch
        st = torch.cuda.current_stream().cuda_stream if stream is None else stream
        _lib.check(self._lib.vrfhip_ietf_verify_batch_affine_dev(
            self._h, pk_xy.shape[0], pk_xy.data_ptr(), inp_xy.data_ptr(), out_xy.data_ptr(), c.data_ptr(),
            s.data_ptr(), None if ad is None else ad.data_ptr(), None if ad_off is None else ad_off.data_ptr(),
            ad_len, status.data_ptr(), st), "vrfhip_ietf_verify_batch_affine_dev")

    def ietf_prove_batch(self, sk, msgs=None, inputs=None, ad=b""):
        """Returns dict(output, c, s, pk, input, status).  Either `msgs` (sequence of byte strings
        or an (n, L) uint8 array) or `inputs` (n x 32 pre-hashed points) must be given."""
        sk = np.ascontiguousarray(sk, dtype=np.uint8).reshape(-1, 32)
        n = sk.shape[0]
        msg_blob = msg_off = inp = None
        msg_len = 0
        ipw = self.point_bytes()
        if inputs is not None:
            inp = np.ascontiguousarray(inputs, dtype=np.uint8).reshape(-1, ipw)
            if inp.shape[0] != n:
                raise ValueError("ragged batch")
        elif isinstance(msgs, np.ndarray):
            m = np.ascontiguousarray(msgs, dtype=np.uint8).reshape(n, -1)
            msg_len = m.shape[1]
            msg_blob = np.concatenate([m.reshape(-1), np.zeros(1, np.uint8)])
        else:
            if msgs is None or len(msgs) != n:
                raise ValueError("msgs must have n entries")
            msg_blob, msg_off = _pack_var([bytes(x) for x in msgs])
        pw = self.prove_point_bytes()
        res = {k: np.empty((n, pw if k in ("output", "pk") else ipw if k == "input" else 32), dtype=np.uint8)
               for k in ("output", "c", "s", "pk", "input")}
        status = np.empty(n, dtype=np.uint8)
        blob, off, ad_len = self._ad_args(ad, n)
        _lib.check(self._lib.vrfhip_ietf_prove_batch(
            self._h, n, _ptr(sk), _ptr(msg_blob), _ptr(msg_off), msg_len, _ptr(inp), _ptr(blob), _ptr(off),
            ad_len, _ptr(res["output"]), _ptr(res["c"]), _ptr(res["s"]), _ptr(res["pk"]), _ptr(res["input"]),
            _ptr(status)), "vrfhip_ietf_prove_batch")
        res["status"] = status
        return res

    def _msg_args(self, n, msgs, inputs):
        msg_blob = msg_off = inp = None
        msg_len = 0
        if inputs is not None:
            inp = np.ascontiguousarray(inputs, dtype=np.uint8).reshape(-1, self.point_bytes())
            if inp.shape[0] != n:
                raise ValueError("ragged batch")
        elif isinstance(msgs, np.ndarray):
            m = np.ascontiguousarray(msgs, dtype=np.uint8).reshape(n, -1)
            msg_len = m.shape[1]
            msg_blob = np.concatenate([m.reshape(-1), np.zeros(1, np.uint8)])
        else:
            if msgs is None or len(msgs) != n:
                raise ValueError("msgs must have n entries")
            msg_blob, msg_off = _pack_var([bytes(x) for x in msgs])
        return msg_blob, msg_off, msg_len, inp

    def pedersen_prove_batch(self, sk, msgs=None, inputs=None, ad=b""):
        """`pedersen::Prover::prove`.  Returns dict(output, pk_com, r, ok, s, sb, blinding, input, status)."""
        sk = np.ascontiguousarray(sk, dtype=np.uint8).reshape(-1, 32)
        n = sk.shape[0]
        msg_blob, msg_off, msg_len, inp = self._msg_args(n, msgs, inputs)
        pw, ipw = self.prove_point_bytes(), self.point_bytes()
        res = {k: np.empty((n, pw if k in ("output", "pk_com", "r", "ok") else ipw if k == "input" else 32), dtype=np.uint8)
               for k in ("output", "pk_com", "r", "ok", "s", "sb", "blinding", "input")}
        status = np.empty(n, dtype=np.uint8)
        blob, off, ad_len = self._ad_args(ad, n)
        _lib.check(self._lib.vrfhip_pedersen_prove_batch(
            self._h, n, _ptr(sk), _ptr(msg_blob), _ptr(msg_off), msg_len, _ptr(inp), _ptr(blob), _ptr(off), ad_len,
            _ptr(res["output"]), _ptr(res["pk_com"]), _ptr(res["r"]), _ptr(res["ok"]), _ptr(res["s"]),
            _ptr(res["sb"]), _ptr(res["blinding"]), _ptr(res["input"]), _ptr(status)), "vrfhip_pedersen_prove_batch")
        res["status"] = status
        return res

    def pedersen_verify_batch(self, inp, out, pk_com, r, ok, s, sb, ad=b"") -> np.ndarray:
        """`pedersen::Verifier::verify`."""
        pw = self.point_bytes()
        arrs = [np.ascontiguousarray(x, dtype=np.uint8).reshape(-1, pw) for x in (inp, out, pk_com, r, ok)]
        arrs += [np.ascontiguousarray(x, dtype=np.uint8).reshape(-1, 32) for x in (s, sb)]
        n = arrs[0].shape[0]
        if not all(x.shape[0] == n for x in arrs):
            raise ValueError("ragged batch")
        status = np.empty(n, dtype=np.uint8)
        blob, off, ad_len = self._ad_args(ad, n)
        _lib.check(self._lib.vrfhip_pedersen_verify_batch(
            self._h, n, *[_ptr(x) for x in arrs], _ptr(blob), _ptr(off), ad_len, _ptr(status)),
            "vrfhip_pedersen_verify_batch")
        return status

    def pedersen_verify_batch_rlc(self, inp, out, pk_com, r, ok, s, sb, ad=b"", seed: Optional[bytes] = None,
                                  affine: bool = False):
        """`pedersen::Verifier::verify` for a whole batch through one MSM (random linear combination).
        Returns (status, batch_ok): the same per-proof statuses as pedersen_verify_batch; batch_ok tells
        whether the single-MSM path sufficed (False: the per-proof kernels were run to locate failures).
        seed: 32 secret random bytes (default os.urandom).  affine: the five point arrays are (n, 64)
        x || y little-endian (arkworks `Affine`) instead of compressed encodings."""
        import os
        pw = 64 if affine else self.point_bytes()
        arrs = [np.ascontiguousarray(x, dtype=np.uint8).reshape(-1, pw) for x in (inp, out, pk_com, r, ok)]
        arrs += [np.ascontiguousarray(x, dtype=np.uint8).reshape(-1, 32) for x in (s, sb)]
        n = arrs[0].shape[0]
        if not all(x.shape[0] == n for x in arrs):
            raise ValueError("ragged batch")
        seed_a = _np_u8(os.urandom(32) if seed is None else seed, 32)
        status = np.empty(n, dtype=np.uint8)
        blob, off, ad_len = self._ad_args(ad, n)
        okf = ctypes.c_int32(0)
        fn = self._lib.vrfhip_pedersen_verify_batch_rlc_affine if affine else self._lib.vrfhip_pedersen_verify_batch_rlc
        _lib.check(fn(self._h, n, *[_ptr(x) for x in arrs], _ptr(blob), _ptr(off), ad_len, _ptr(seed_a), _ptr(status),
                      ctypes.byref(okf)), "vrfhip_pedersen_verify_batch_rlc")
        return status, bool(okf.value)

    def pedersen_verify_batch_rlc_dev(self, inp, out, pk_com, r, ok, s, sb, status, fail_flag, seed: bytes,
                                      ad=None, ad_off=None, ad_len=0, stream=None, affine: bool = False):
        """Device-pointer form: status[i] in {0, 2}; fail_flag[0] = 1 if the batch equation fails."""
        import torch
        st = torch.cuda.current_stream().cuda_stream if stream is None else stream
        dp = lambda t: None if t is None else t.data_ptr()
        seed_a = _np_u8(seed, 32)
        fn = self._lib.vrfhip_pedersen_verify_batch_rlc_affine_dev if affine else self._lib.vrfhip_pedersen_verify_batch_rlc_dev
        _lib.check(fn(
            self._h, inp.shape[0], inp.data_ptr(), out.data_ptr(), pk_com.data_ptr(), r.data_ptr(), ok.data_ptr(),
            s.data_ptr(), sb.data_ptr(), dp(ad), dp(ad_off), ad_len, _ptr(seed_a), status.data_ptr(),
            fail_flag.data_ptr(), st), "vrfhip_pedersen_verify_batch_rlc_dev")

    def pedersen_prove_batch_dev(self, sk, msg, msg_len, out, pk_com, r, ok, s, sb, blinding=None, input_out=None,
                                 status=None, inputs=None, ad=None, ad_off=None, ad_len=0, stream=None):
        import torch
        st = torch.cuda.current_stream().cuda_stream if stream is None else stream
        dp = lambda t: None if t is None else t.data_ptr()
        _lib.check(self._lib.vrfhip_pedersen_prove_batch_dev(
            self._h, sk.shape[0], sk.data_ptr(), dp(msg), None, msg_len, dp(inputs), dp(ad), dp(ad_off), ad_len,
            out.data_ptr(), pk_com.data_ptr(), r.data_ptr(), ok.data_ptr(), s.data_ptr(), sb.data_ptr(),
            dp(blinding), dp(input_out), dp(status), st), "vrfhip_pedersen_prove_batch_dev")

    def pedersen_verify_batch_dev(self, inp, out, pk_com, r, ok, s, sb, status, ad=None, ad_off=None, ad_len=0,
                                  stream=None):
        import torch
        st = torch.cuda.current_stream().cuda_stream if stream is None else stream
        dp = lambda t: None if t is None else t.data_ptr()
        _lib.check(self._lib.vrfhip_pedersen_verify_batch_dev(
            self._h, inp.shape[0], inp.data_ptr(), out.data_ptr(), pk_com.data_ptr(), r.data_ptr(), ok.data_ptr(),
            s.data_ptr(), sb.data_ptr(), dp(ad), dp(ad_off), ad_len, status.data_ptr(), st),
            "vrfhip_pedersen_verify_batch_dev")

    def msm(self, bases_xy, scalars):
        """`VariableBaseMSM::msm`: sum_i scalars[i] * bases[i].  bases_xy: (n, 64) affine x||y LE;
        scalars: (n, 32) in the suite's scalar encoding (LE; big-endian on secp256r1).  Returns (point, xy64) with the
        point in the suite's encoding (32 bytes; 33-byte Sec1 on secp256r1); raises InvalidData on bad inputs."""
        b = np.ascontiguousarray(bases_xy, dtype=np.uint8).reshape(-1, 64)
        k = np.ascontiguousarray(scalars, dtype=np.uint8).reshape(-1, 32)
        if b.shape[0] != k.shape[0]:
            raise ValueError("ragged batch")
        out, xy, st = np.empty(self.point_bytes(), np.uint8), np.empty(64, np.uint8), np.empty(1, np.uint8)
        _lib.check(self._lib.vrfhip_msm(self._h, b.shape[0], _ptr(b) if b.size else None, _ptr(k) if k.size else None,
                                        _ptr(out), _ptr(xy), _ptr(st)), "vrfhip_msm")
        if st[0] != ST_OK:
            raise InvalidData()
        return out.tobytes(), xy.tobytes()

    def msm_dev(self, bases_xy, scalars, out_point, out_xy, status, stream=None):
        import torch
        st = torch.cuda.current_stream().cuda_stream if stream is None else stream
        _lib.check(self._lib.vrfhip_msm_dev(self._h, bases_xy.shape[0], bases_xy.data_ptr(), scalars.data_ptr(),
                                            out_point.data_ptr(), None if out_xy is None else out_xy.data_ptr(),
                                            status.data_ptr(), st), "vrfhip_msm_dev")

    def pairing_check_batch(self, g1, g2, g2_shared: bool = False) -> np.ndarray:
        """BLS12-381: e(P0,Q0) e(P1,Q1) == 1 per item.  g1: (n, 192) bytes; g2: (n, 384) or (384,) if shared."""
        a = np.ascontiguousarray(g1, dtype=np.uint8).reshape(-1, 192)
        b = np.ascontiguousarray(g2, dtype=np.uint8).reshape(-1, 384)
        n = a.shape[0]
        if (g2_shared and b.shape[0] != 1) or (not g2_shared and b.shape[0] != n):
            raise ValueError("ragged batch")
        st = np.empty(n, dtype=np.uint8)
        _lib.check(self._lib.vrfhip_pairing_check_batch(self._h, n, _ptr(a), _ptr(b), int(g2_shared), _ptr(st)),
                   "vrfhip_pairing_check_batch")
        return st

    def pairing_check_batch_dev(self, g1, g2, status, g2_shared: bool = False, stream=None):
        import torch
        st = torch.cuda.current_stream().cuda_stream if stream is None else stream
        _lib.check(self._lib.vrfhip_pairing_check_batch_dev(self._h, g1.shape[0], g1.data_ptr(), g2.data_ptr(),
                                                            int(g2_shared), status.data_ptr(), st),
                   "vrfhip_pairing_check_batch_dev")

    def pairing_check_batch_rlc(self, g1, g2_shared, seed: Optional[bytes] = None):
        """n checks against ONE shared G2 pair as a batch: two G1 MSMs + one pairing.  -> (status, batch_ok)"""
        import os as _os
        a = np.ascontiguousarray(g1, dtype=np.uint8).reshape(-1, 192)
        b = np.ascontiguousarray(g2_shared, dtype=np.uint8).reshape(-1)
        if b.size != 384:
            raise ValueError("g2_shared must be 384 bytes")
        n = a.shape[0]
        sd = _np_u8(seed if seed is not None else _os.urandom(32), 32)
        st = np.empty(n, dtype=np.uint8)
        okf = ctypes.c_int32(1)
        _lib.check(self._lib.vrfhip_pairing_check_batch_rlc(self._h, n, _ptr(a) if n else None, _ptr(b), _ptr(sd), _ptr(st),
                                                            ctypes.byref(okf)), "vrfhip_pairing_check_batch_rlc")
        return st, bool(okf.value)

    def pairing_check_batch_rlc_dev(self, g1, g2_shared, status, verdict, seed: bytes, stream=None):
        import torch
        st = torch.cuda.current_stream().cuda_stream if stream is None else stream
        sd = _np_u8(seed, 32)
        _lib.check(self._lib.vrfhip_pairing_check_batch_rlc_dev(self._h, g1.shape[0], g1.data_ptr(), g2_shared.data_ptr(),
                                                                _ptr(sd), status.data_ptr(), verdict.data_ptr(), st),
                   "vrfhip_pairing_check_batch_rlc_dev")

    def g1_msm(self, bases, scalars):
        """`VariableBaseMSM::msm` on BLS12-381 G1: bases (n, 96), scalars (n, 32) -> 96-byte point; raises InvalidData."""
        bs = np.ascontiguousarray(bases, dtype=np.uint8).reshape(-1, 96)
        k = np.ascontiguousarray(scalars, dtype=np.uint8).reshape(-1, 32)
        if bs.shape[0] != k.shape[0]:
            raise ValueError("ragged batch")
        out, st = np.empty(96, np.uint8), np.empty(1, np.uint8)
        _lib.check(self._lib.vrfhip_g1_msm(self._h, bs.shape[0], _ptr(bs) if bs.size else None, _ptr(k) if k.size else None,
                                           _ptr(out), _ptr(st)), "vrfhip_g1_msm")
        if st[0] != ST_OK:
            raise InvalidData()
        return out.tobytes()

    def hash_to_curve_batch(self, msgs) -> np.ndarray:
        if isinstance(msgs, np.ndarray):
            m = np.ascontiguousarray(msgs, dtype=np.uint8)
            n, msg_len = m.shape
            blob, off = np.concatenate([m.reshape(-1), np.zeros(1, np.uint8)]), None
        else:
            n, msg_len = len(msgs), 0
            blob, off = _pack_var([bytes(x) for x in msgs])
        pts = np.empty((n, self.point_bytes()), dtype=np.uint8)
        _lib.check(self._lib.vrfhip_hash_to_curve_batch(self._h, n, _ptr(blob), _ptr(off), msg_len, _ptr(pts)),
                   "vrfhip_hash_to_curve_batch")
        return pts

    def output_hash_batch(self, outputs) -> np.ndarray:
        o = np.ascontiguousarray(outputs, dtype=np.uint8).reshape(-1, self.point_bytes())
        h = np.empty((o.shape[0], self.hash_bytes()), dtype=np.uint8)
        _lib.check(self._lib.vrfhip_output_hash_batch(self._h, o.shape[0], _ptr(o), _ptr(h)),
                   "vrfhip_output_hash_batch")
        return h

    def secret_from_seed_batch(self, seeds: np.ndarray, with_public: bool = True):
        sd = np.ascontiguousarray(seeds, dtype=np.uint8)
        n, seed_len = sd.shape
        sk = np.empty((n, 32), dtype=np.uint8)
        pk = np.empty((n, self.point_bytes()), dtype=np.uint8) if with_public else None
        _lib.check(self._lib.vrfhip_secret_from_seed_batch(self._h, n, _ptr(sd), seed_len, _ptr(sk), _ptr(pk)),
                   "vrfhip_secret_from_seed_batch")
        return sk, pk

    def point_validate_batch(self, points, want_xy: bool = False):
        p = np.ascontiguousarray(points, dtype=np.uint8).reshape(-1, self.point_bytes())
        n = p.shape[0]
        st = np.empty(n, dtype=np.uint8)
        xy = np.empty((n, 64), dtype=np.uint8) if want_xy else None
        _lib.check(self._lib.vrfhip_point_validate_batch(self._h, n, _ptr(p), _ptr(xy), _ptr(st)),
                   "vrfhip_point_validate_batch")
        return (st, xy) if want_xy else st

    def te_sw_map_batch(self, points_xy, to_te: bool = False):
        """`utils::te_sw_map::te_to_sw` (to_te=False) / `sw_to_te` (to_te=True) on n x 64 B affine x || y points of the
        context's twisted-Edwards curve / its short-Weierstrass form.  Returns (out_xy, status): status 2 = upstream's None
        (or a coordinate not below the modulus), the row is then zero."""
        p = np.ascontiguousarray(points_xy, dtype=np.uint8).reshape(-1, 64)
        n = p.shape[0]
        out, st = np.empty((n, 64), np.uint8), np.empty(n, np.uint8)
        _lib.check(self._lib.vrfhip_te_sw_map_batch(self._h, n, 1 if to_te else 0, _ptr(p), _ptr(out), _ptr(st)),
                   "vrfhip_te_sw_map_batch")
        return out, st

    def fq_mul_batch(self, a, b) -> np.ndarray:
        a = np.ascontiguousarray(a, dtype=np.uint8).reshape(-1, 32)
        b = np.ascontiguousarray(b, dtype=np.uint8).reshape(-1, 32)
        r = np.empty_like(a)
        _lib.check(self._lib.vrfhip_fq_mul_batch(self._h, a.shape[0], _ptr(a), _ptr(b), _ptr(r)),
                   "vrfhip_fq_mul_batch")
        return r

    # ---- test primitives (vrfhip_test_*): the pieces under the batch calls ---------------
    def test_point_add(self, a, b):
        a, b = (np.ascontiguousarray(x, dtype=np.uint8).reshape(-1, 32) for x in (a, b))
        n = a.shape[0]
        out, st = np.empty((n, 32), np.uint8), np.empty(n, np.uint8)
        _lib.check(self._lib.vrfhip_test_point_add(self._h, n, _ptr(a), _ptr(b), _ptr(out), _ptr(st)), "vrfhip_test_point_add")
        return out, st

    def test_scalar_mul(self, scalars, points):
        k, p = (np.ascontiguousarray(x, dtype=np.uint8).reshape(-1, 32) for x in (scalars, points))
        n = k.shape[0]
        out, st = np.empty((n, 32), np.uint8), np.empty(n, np.uint8)
        _lib.check(self._lib.vrfhip_test_scalar_mul(self._h, n, _ptr(k), _ptr(p), _ptr(out), _ptr(st)), "vrfhip_test_scalar_mul")
        return out, st

    def _test_hash(self, fn, name, msgs, width):
        blob, off = _pack_var([bytes(m) for m in msgs])
        out = np.empty((len(msgs), width), np.uint8)
        _lib.check(fn(self._h, len(msgs), _ptr(blob), _ptr(off), 0, _ptr(out)), name)
        return out

    def test_sha512(self, msgs):
        return self._test_hash(self._lib.vrfhip_test_sha512, "vrfhip_test_sha512", msgs, 64)

    def test_xmd(self, msgs):
        return self._test_hash(self._lib.vrfhip_test_xmd, "vrfhip_test_xmd", msgs, 96)

    def test_batch_digest(self, arrays, ad=None, index0: int = 0) -> bytes:
        """The digest the batched (random-linear-combination) verifiers hash into their weights: arrays = list of
        (n, w_j) uint8 arrays; ad = None, bytes (shared) or a list of n byte strings."""
        import ctypes
        arrs = [np.ascontiguousarray(a, dtype=np.uint8) for a in arrays]
        n = arrs[0].shape[0]
        ptrs = (ctypes.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        widths = np.array([a.size // n for a in arrs], np.uint32)
        blob, off, ad_len = None, None, 0
        if isinstance(ad, (bytes, bytearray)):
            blob, ad_len = np.frombuffer(bytes(ad) + b"\0", np.uint8), len(ad)
        elif ad is not None:
            blob, off = _pack_var([bytes(a) for a in ad])
        root = np.empty(32, np.uint8)
        _lib.check(self._lib.vrfhip_test_batch_digest(self._h, n, len(arrs), ctypes.cast(ptrs, ctypes.c_void_p), _ptr(widths),
                                                      _ptr(blob) if blob is not None else None,
                                                      _ptr(off) if off is not None else None, ad_len, index0, _ptr(root)),
                   "vrfhip_test_batch_digest")
        return root.tobytes()

    # ---- device-pointer batch API (torch CUDA uint8 tensors, current stream) -----------
    def ietf_verify_batch_dev(self, pk, inp, out, c, s, status, ad=None, ad_off=None, ad_len=0, stream=None):
        import torch
        st = torch.cuda.current_stream().cuda_stream if stream is None else stream
        n = pk.shape[0]
        _lib.check(self._lib.vrfhip_ietf_verify_batch_dev(
            self._h, n, pk.data_ptr(), inp.data_ptr(), out.data_ptr(), c.data_ptr(), s.data_ptr(),
            None if ad is None else ad.data_ptr(), None if ad_off is None else ad_off.data_ptr(), ad_len,
            status.data_ptr(), st), "vrfhip_ietf_verify_batch_dev")

    def ietf_verify_batch_alpha_dev(self, pk, msg, msg_len, out, c, s, status, msg_off=None, ad=None, ad_off=None, ad_len=0,
                                    stream=None):
        import torch
        st = torch.cuda.current_stream().cuda_stream if stream is None else stream
        dp = lambda t: None if t is None else t.data_ptr()
        _lib.check(self._lib.vrfhip_ietf_verify_batch_alpha_dev(
            self._h, pk.shape[0], pk.data_ptr(), dp(msg), dp(msg_off), msg_len, out.data_ptr(), c.data_ptr(), s.data_ptr(),
            dp(ad), dp(ad_off), ad_len, status.data_ptr(), st), "vrfhip_ietf_verify_batch_alpha_dev")

    def ietf_prove_batch_dev(self, sk, msg, msg_len, out, c, s, pk_out=None, input_out=None, status=None,
                             msg_off=None, inputs=None, ad=None, ad_off=None, ad_len=0, stream=None):
        import torch
        st = torch.cuda.current_stream().cuda_stream if stream is None else stream
        n = sk.shape[0]
        dp = lambda t: None if t is None else t.data_ptr()
        _lib.check(self._lib.vrfhip_ietf_prove_batch_dev(
            self._h, n, sk.data_ptr(), dp(msg), dp(msg_off), msg_len, dp(inputs), dp(ad), dp(ad_off), ad_len,
            out.data_ptr(), c.data_ptr(), s.data_ptr(), dp(pk_out), dp(input_out), dp(status), st),
            "vrfhip_ietf_prove_batch_dev")


def _ctx_array(ctxs):
    arr = (ctypes.c_void_p * len(ctxs))(*[c.handle for c in ctxs])
    return arr, len(ctxs)


def ietf_verify_batch_multi(ctxs, pk, inp, out, c, s, ad=b"") -> np.ndarray:
    """vrfhip_ietf_verify_batch_multi: one call, one context per GPU, contiguous slices, one host thread each."""
    pw = ctxs[0].point_bytes()
    pk, inp, out = (np.ascontiguousarray(x, dtype=np.uint8).reshape(-1, pw) for x in (pk, inp, out))
    c, s = (np.ascontiguousarray(x, dtype=np.uint8).reshape(-1, 32) for x in (c, s))
    n = pk.shape[0]
    blob, off, ad_len = Context._ad_args(ad, n)
    st = np.empty(n, np.uint8)
    arr, k = _ctx_array(ctxs)
    _lib.check(_lib.load().vrfhip_ietf_verify_batch_multi(arr, k, n, _ptr(pk), _ptr(inp), _ptr(out), _ptr(c), _ptr(s),
                                                          _ptr(blob), _ptr(off), ad_len, _ptr(st)),
               "vrfhip_ietf_verify_batch_multi")
    return st


def ietf_prove_batch_multi(ctxs, sk, msgs, ad=b""):
    sk = np.ascontiguousarray(sk, dtype=np.uint8).reshape(-1, 32)
    n = sk.shape[0]
    mblob, moff = _pack_var([bytes(m) for m in msgs])
    blob, off, ad_len = Context._ad_args(ad, n)
    pw, ipw = ctxs[0].prove_point_bytes(), ctxs[0].point_bytes()
    res = {k: np.empty((n, pw if k in ("output", "pk") else ipw if k == "input" else 32), np.uint8) for k in ("output", "c", "s", "pk", "input")}
    res["status"] = np.empty(n, np.uint8)
    arr, k = _ctx_array(ctxs)
    _lib.check(_lib.load().vrfhip_ietf_prove_batch_multi(
        arr, k, n, _ptr(sk), _ptr(mblob), _ptr(moff), 0, None, _ptr(blob), _ptr(off), ad_len, _ptr(res["output"]),
        _ptr(res["c"]), _ptr(res["s"]), _ptr(res["pk"]), _ptr(res["input"]), _ptr(res["status"])),
        "vrfhip_ietf_prove_batch_multi")
    return res


def pedersen_prove_batch_multi(ctxs, sk, msgs, ad=b""):
    sk = np.ascontiguousarray(sk, dtype=np.uint8).reshape(-1, 32)
    n = sk.shape[0]
    mblob, moff = _pack_var([bytes(m) for m in msgs])
    blob, off, ad_len = Context._ad_args(ad, n)
    names = ("output", "pk_com", "r", "ok", "s", "sb", "blinding", "input")
    pw, ipw = ctxs[0].prove_point_bytes(), ctxs[0].point_bytes()
    res = {k: np.empty((n, pw if k in ("output", "pk_com", "r", "ok") else ipw if k == "input" else 32), np.uint8) for k in names}
    res["status"] = np.empty(n, np.uint8)
    arr, k = _ctx_array(ctxs)
    _lib.check(_lib.load().vrfhip_pedersen_prove_batch_multi(
        arr, k, n, _ptr(sk), _ptr(mblob), _ptr(moff), 0, None, _ptr(blob), _ptr(off), ad_len,
        *[_ptr(res[x]) for x in names], _ptr(res["status"])), "vrfhip_pedersen_prove_batch_multi")
    return res


def pedersen_verify_batch_multi(ctxs, inp, out, pk_com, r, ok, s, sb, ad=b"", rlc_seed: Optional[bytes] = None) -> np.ndarray:
    pw = ctxs[0].point_bytes()
    arrs = [np.ascontiguousarray(x, dtype=np.uint8).reshape(-1, pw) for x in (inp, out, pk_com, r, ok)]
    arrs += [np.ascontiguousarray(x, dtype=np.uint8).reshape(-1, 32) for x in (s, sb)]
    n = arrs[0].shape[0]
    blob, off, ad_len = Context._ad_args(ad, n)
    st = np.empty(n, np.uint8)
    seed = None if rlc_seed is None else _np_u8(rlc_seed, 32)
    arr, k = _ctx_array(ctxs)
    _lib.check(_lib.load().vrfhip_pedersen_verify_batch_multi(arr, k, n, *[_ptr(x) for x in arrs], _ptr(blob), _ptr(off),
                                                              ad_len, _ptr(seed), _ptr(st)),
               "vrfhip_pedersen_verify_batch_multi")
    return st


class KeySet:
    """Device-resident tables of a set of public keys (vrfhip_keyset); destroy before its Context."""

    def __init__(self, ctx: Context, handle, n_keys: int):
        self._ctx, self.handle, self.n_keys = ctx, handle, n_keys

    def bytes(self) -> int:
        return int(self._ctx._lib.vrfhip_keyset_bytes(self.handle))

    def close(self) -> None:
        if self.handle:
            self._ctx._lib.vrfhip_keyset_destroy(self.handle)
            self.handle = None


_default_ctx: Optional[Context] = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


# ---- single-item mirror of the Rust types --------------------------------------------------

@dataclass(frozen=True)
class Public:
    """`Public` (src/lib.rs:15): a public key point, ArkworksCodec encoding."""
    encoded: bytes


@dataclass(frozen=True)
class Input:
    """`Input` (src/lib.rs:15): VRF input point."""
    encoded: bytes

    @staticmethod
    def new(data: bytes, ctx: Optional[Context] = None) -> Optional["Input"]:
        ctx = ctx or default_context()
        return Input(bytes(ctx.hash_to_curve_batch([bytes(data)])[0]))


@dataclass(frozen=True)
class Output:
    """`Output` (src/lib.rs:15): VRF output point (Gamma)."""
    encoded: bytes

    def hash(self, ctx: Optional[Context] = None) -> bytes:
        ctx = ctx or default_context()
        return bytes(ctx.output_hash_batch(np.frombuffer(self.encoded, dtype=np.uint8))[0])


@dataclass(frozen=True)
class IetfProof:
    """`ietf::Proof { c, s }` (src/lib.rs:14)."""
    c: bytes
    s: bytes


@dataclass(frozen=True)
class Secret:
    """`Secret` (src/lib.rs:16): secret scalar (32 B little-endian) with its cached public key."""
    scalar: bytes
    _public: bytes

    @staticmethod
    def from_seed(seed: bytes, ctx: Optional[Context] = None) -> "Secret":
        ctx = ctx or default_context()
        sk, pk = ctx.secret_from_seed_batch(np.frombuffer(bytes(seed), dtype=np.uint8).reshape(1, -1))
        return Secret(bytes(sk[0]), bytes(pk[0]))

    def public(self) -> Public:
        return Public(self._public)

    def output(self, inp: Input, ctx: Optional[Context] = None) -> Output:
        ctx = ctx or default_context()
        r = ctx.ietf_prove_batch(np.frombuffer(self.scalar, dtype=np.uint8), inputs=np.frombuffer(inp.encoded, dtype=np.uint8))
        if r["status"][0] != ST_OK:
            raise InvalidData()
        return Output(bytes(r["output"][0]))


@dataclass(frozen=True)
class PedersenProof:
    """`pedersen::Proof { pk_com, r, ok, s, sb }` (src/lib.rs:14)."""
    pk_com: bytes
    r: bytes
    ok: bytes
    s: bytes
    sb: bytes


class pedersen:  # noqa: N801  (mirrors the Rust module name)
    """`pedersen` module (src/lib.rs:14)."""
    Proof = PedersenProof

    class Prover:
        @staticmethod
        def prove(secret: "Secret", inp: "Input", out: "Output", ad: bytes = b"", ctx: Optional[Context] = None):
            """-> (proof, blinding), as the Rust `prove` returns `(Proof, ScalarField)`."""
            ctx = ctx or default_context()
            r = ctx.pedersen_prove_batch(np.frombuffer(secret.scalar, dtype=np.uint8),
                                         inputs=np.frombuffer(inp.encoded, dtype=np.uint8), ad=bytes(ad))
            if r["status"][0] != ST_OK:
                raise InvalidData()
            g = lambda k: bytes(r[k][0])
            return PedersenProof(g("pk_com"), g("r"), g("ok"), g("s"), g("sb")), g("blinding")

    class Verifier:
        @staticmethod
        def verify(inp: "Input", out: "Output", ad: bytes, proof: PedersenProof, ctx: Optional[Context] = None) -> None:
            ctx = ctx or default_context()
            f = lambda b: np.frombuffer(b, dtype=np.uint8)
            st = ctx.pedersen_verify_batch(f(inp.encoded), f(out.encoded), f(proof.pk_com), f(proof.r), f(proof.ok),
                                           f(proof.s), f(proof.sb), ad=bytes(ad))[0]
            if st == ST_VERIFICATION_FAILURE:
                raise VerificationFailure()
            if st != ST_OK:
                raise InvalidData()


class ietf:  # noqa: N801  (mirrors the Rust module name)
    """`ietf` module (src/lib.rs:14)."""
    Proof = IetfProof

    class Prover:
        @staticmethod
        def prove(secret: Secret, inp: Input, out: Output, ad: bytes = b"", ctx: Optional[Context] = None) -> IetfProof:
            ctx = ctx or default_context()
            r = ctx.ietf_prove_batch(np.frombuffer(secret.scalar, dtype=np.uint8),
                                     inputs=np.frombuffer(inp.encoded, dtype=np.uint8), ad=bytes(ad))
            if r["status"][0] != ST_OK:
                raise InvalidData()
            return IetfProof(bytes(r["c"][0]), bytes(r["s"][0]))

    class Verifier:
        @staticmethod
        def verify(public: Public, inp: Input, out: Output, ad: bytes, proof: IetfProof,
                   ctx: Optional[Context] = None) -> None:
            """Returns None on success (Rust `Ok(())`), raises `VerificationFailure` / `InvalidData`."""
            ctx = ctx or default_context()
            f = lambda b: np.frombuffer(b, dtype=np.uint8)
            st = ctx.ietf_verify_batch(f(public.encoded), f(inp.encoded), f(out.encoded), f(proof.c), f(proof.s),
                                       ad=bytes(ad))[0]
            if st == ST_VERIFICATION_FAILURE:
                raise VerificationFailure()
            if st != ST_OK:
                raise InvalidData()
