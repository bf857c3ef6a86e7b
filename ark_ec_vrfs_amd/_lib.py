"""ctypes binding of libvrfhip.so (include/vrfhip.h).

There is no CPU path: if the HIP library cannot be loaded, or no MI355X is visible, every
entry point raises.  The library is built in-tree by ``__graft_entry__.build()`` /
``make -C ark_ec_vrfs_amd/csrc``.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_int32, c_size_t, c_uint8, c_uint32, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libvrfhip.so")

# Every symbol include/vrfhip.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "vrfhip_abi_version", "vrfhip_last_error", "vrfhip_ctx_create", "vrfhip_ctx_destroy",
    "vrfhip_suite_desc_default", "vrfhip_ctx_create_desc", "vrfhip_ctx_get_desc",
    "vrfhip_ctx_reserve", "vrfhip_ctx_workspace_bytes", "vrfhip_host_alloc", "vrfhip_host_free",
    "vrfhip_ctx_profile", "vrfhip_ctx_profile_read",
    "vrfhip_ctx_set_flags", "vrfhip_ctx_get_flags", "vrfhip_ctx_point_bytes", "vrfhip_ctx_hash_bytes",
    "vrfhip_ietf_verify_batch", "vrfhip_ietf_verify_batch_dev",
    "vrfhip_ietf_verify_batch_affine", "vrfhip_ietf_verify_batch_affine_dev",
    "vrfhip_ietf_verify_batch_alpha", "vrfhip_ietf_verify_batch_alpha_dev",
    "vrfhip_keyset_create", "vrfhip_keyset_destroy", "vrfhip_keyset_bytes",
    "vrfhip_ietf_verify_batch_keyed", "vrfhip_ietf_verify_batch_keyed_dev",
    "vrfhip_ietf_prove_batch", "vrfhip_ietf_prove_batch_dev",
    "vrfhip_pedersen_prove_batch", "vrfhip_pedersen_prove_batch_dev",
    "vrfhip_pedersen_verify_batch", "vrfhip_pedersen_verify_batch_dev",
    "vrfhip_pedersen_verify_batch_rlc", "vrfhip_pedersen_verify_batch_rlc_dev",
    "vrfhip_pedersen_verify_batch_rlc_affine", "vrfhip_pedersen_verify_batch_rlc_affine_dev",
    "vrfhip_msm", "vrfhip_msm_dev",
    "vrfhip_pairing_check_batch", "vrfhip_pairing_check_batch_dev",
    "vrfhip_pairing_check_batch_rlc", "vrfhip_pairing_check_batch_rlc_dev", "vrfhip_g1_msm", "vrfhip_g1_msm_dev",
    "vrfhip_hash_to_curve_batch", "vrfhip_hash_to_curve_batch_dev",
    "vrfhip_output_hash_batch", "vrfhip_output_hash_batch_dev",
    "vrfhip_secret_from_seed_batch", "vrfhip_secret_from_seed_batch_dev",
    "vrfhip_point_validate_batch", "vrfhip_point_validate_batch_dev",
    "vrfhip_te_sw_map_batch", "vrfhip_te_sw_map_batch_dev",
    "vrfhip_fq_mul_batch", "vrfhip_test_pairing_quad_ops", "vrfhip_test_pairing_oct_ops", "vrfhip_debug_set", "vrfhip_test_blinding_base",
    "vrfhip_debug_proofs_per_lane",
    "vrfhip_ietf_verify_batch_multi", "vrfhip_ietf_prove_batch_multi",
    "vrfhip_pedersen_prove_batch_multi", "vrfhip_pedersen_verify_batch_multi",
    "vrfhip_test_point_add", "vrfhip_test_scalar_mul", "vrfhip_test_sha512", "vrfhip_test_xmd", "vrfhip_test_batch_digest",
]


class VrfHipError(RuntimeError):
    pass


class SuiteDescStruct(ctypes.Structure):
    """`vrfhip_suite_desc` (include/vrfhip.h)."""
    _fields_ = [("struct_size", c_uint32), ("curve", c_int32), ("suite_id_len", c_uint32),
                ("suite_id", c_uint8 * 64), ("h2c_dst_len", c_uint32), ("h2c_dst", c_uint8 * 128),
                ("generator", c_uint8 * 64), ("blinding_base", c_uint8 * 64), ("challenge_len", c_uint32),
                ("flags", c_uint32)]


_lib = None


def source_stamp() -> str:
    """sha256 over the kernel sources (csrc/*.hip, *.cuh, *.h, *.inc, Makefile, sorted by name).  A profile summary
    (profiles/pmc_kernels.json) carries the stamp of the sources it was measured on; bench.py attaches its counters to a
    fresh number only when the stamp still matches (a stale profile must not decorate a new kernel)."""
    import glob
    import hashlib
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.cuh")) + glob.glob(os.path.join(d, "*.h")) +
                    glob.glob(os.path.join(d, "*.inc")) + [os.path.join(d, "Makefile")]):
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()


ABI_VERSION = 144      # vrfhip_abi_version() of the library this binding was written against


def load() -> ctypes.CDLL:
    """Load libvrfhip.so (once).  torch is imported first when present so that the library
    binds to the HIP runtime already in the process (same SONAME libamdhip64.so.7) and device
    pointers of torch tensors are valid in our launches."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VrfHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    try:
        import torch  # noqa: F401  (loads torch's libamdhip64 first)
    except Exception:
        pass
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    u8p, u32p = POINTER(c_uint8), POINTER(c_uint32)
    lib.vrfhip_abi_version.restype = c_int32
    if lib.vrfhip_abi_version() != ABI_VERSION:
        raise VrfHipError(f"{LIB_PATH} has ABI version {lib.vrfhip_abi_version()}, this binding expects {ABI_VERSION}: rebuild "
                          "(`make -C ark_ec_vrfs_amd/csrc`)")
    lib.vrfhip_last_error.restype = c_char_p
    lib.vrfhip_ctx_create.argtypes = [c_int32, c_int32, POINTER(c_void_p)]
    lib.vrfhip_ctx_create.restype = c_int32
    lib.vrfhip_suite_desc_default.argtypes = [c_int32, POINTER(SuiteDescStruct)]
    lib.vrfhip_ctx_create_desc.argtypes = [POINTER(SuiteDescStruct), c_int32, POINTER(c_void_p)]
    lib.vrfhip_ctx_get_desc.argtypes = [c_void_p, POINTER(SuiteDescStruct)]
    lib.vrfhip_ctx_destroy.argtypes = [c_void_p]
    lib.vrfhip_ctx_destroy.restype = None
    lib.vrfhip_ctx_reserve.argtypes = [c_void_p, c_size_t]
    lib.vrfhip_ctx_reserve.restype = c_int32
    lib.vrfhip_ctx_workspace_bytes.argtypes = [c_void_p]
    lib.vrfhip_ctx_workspace_bytes.restype = c_size_t
    lib.vrfhip_host_alloc.argtypes = [c_size_t, POINTER(c_void_p)]
    lib.vrfhip_host_free.argtypes = [c_void_p]
    lib.vrfhip_host_free.restype = None
    lib.vrfhip_ctx_profile.argtypes = [c_void_p, c_int32]
    lib.vrfhip_ctx_profile_read.argtypes = [c_void_p, POINTER(ctypes.c_double), POINTER(ctypes.c_uint64)]
    lib.vrfhip_ctx_set_flags.argtypes = [c_void_p, c_uint32]
    lib.vrfhip_ctx_get_flags.argtypes = [c_void_p]
    lib.vrfhip_ctx_get_flags.restype = c_uint32
    lib.vrfhip_ctx_point_bytes.argtypes = [c_void_p]
    lib.vrfhip_ctx_point_bytes.restype = c_size_t
    lib.vrfhip_ctx_hash_bytes.argtypes = [c_void_p]
    lib.vrfhip_ctx_hash_bytes.restype = c_size_t
    P = c_void_p  # raw addresses (host buffers or device pointers)
    lib.vrfhip_ietf_verify_batch.argtypes = [c_void_p, c_size_t, P, P, P, P, P, P, P, c_uint32, P]
    lib.vrfhip_ietf_verify_batch_dev.argtypes = [c_void_p, c_size_t, P, P, P, P, P, P, P, c_uint32, P, c_void_p]
    lib.vrfhip_ietf_verify_batch_alpha.argtypes = [c_void_p, c_size_t, P, P, P, c_uint32, P, P, P, P, P, c_uint32, P]
    lib.vrfhip_ietf_verify_batch_alpha_dev.argtypes = [c_void_p, c_size_t, P, P, P, c_uint32, P, P, P, P, P, c_uint32, P, c_void_p]
    lib.vrfhip_ietf_verify_batch_affine.argtypes = lib.vrfhip_ietf_verify_batch.argtypes
    lib.vrfhip_ietf_verify_batch_affine_dev.argtypes = lib.vrfhip_ietf_verify_batch_dev.argtypes
    lib.vrfhip_keyset_create.argtypes = [c_void_p, c_size_t, P, P, POINTER(c_void_p)]
    lib.vrfhip_keyset_destroy.argtypes = [c_void_p]
    lib.vrfhip_keyset_destroy.restype = None
    lib.vrfhip_keyset_bytes.argtypes = [c_void_p]
    lib.vrfhip_keyset_bytes.restype = c_size_t
    lib.vrfhip_ietf_verify_batch_keyed.argtypes = [c_void_p, c_void_p, c_size_t, P, P, P, P, P, P, P, c_uint32, P]
    lib.vrfhip_ietf_verify_batch_keyed_dev.argtypes = [c_void_p, c_void_p, c_size_t, P, P, P, P, P, P, P, c_uint32, P,
                                                       c_void_p]
    lib.vrfhip_ietf_prove_batch.argtypes = [c_void_p, c_size_t, P, P, P, c_uint32, P, P, P, c_uint32,
                                            P, P, P, P, P, P]
    lib.vrfhip_ietf_prove_batch_dev.argtypes = [c_void_p, c_size_t, P, P, P, c_uint32, P, P, P, c_uint32,
                                                P, P, P, P, P, P, c_void_p]
    lib.vrfhip_pedersen_prove_batch.argtypes = [c_void_p, c_size_t, P, P, P, c_uint32, P, P, P, c_uint32,
                                                P, P, P, P, P, P, P, P, P]
    lib.vrfhip_pedersen_prove_batch_dev.argtypes = [c_void_p, c_size_t, P, P, P, c_uint32, P, P, P, c_uint32,
                                                    P, P, P, P, P, P, P, P, P, c_void_p]
    lib.vrfhip_pedersen_verify_batch.argtypes = [c_void_p, c_size_t, P, P, P, P, P, P, P, P, P, c_uint32, P]
    lib.vrfhip_pedersen_verify_batch_dev.argtypes = [c_void_p, c_size_t, P, P, P, P, P, P, P, P, P, c_uint32, P,
                                                     c_void_p]
    lib.vrfhip_pedersen_verify_batch_rlc.argtypes = [c_void_p, c_size_t, P, P, P, P, P, P, P, P, P, c_uint32, P, P,
                                                     POINTER(c_int32)]
    lib.vrfhip_pedersen_verify_batch_rlc_dev.argtypes = [c_void_p, c_size_t, P, P, P, P, P, P, P, P, P, c_uint32,
                                                         P, P, P, c_void_p]
    lib.vrfhip_pedersen_verify_batch_rlc_affine.argtypes = lib.vrfhip_pedersen_verify_batch_rlc.argtypes
    lib.vrfhip_pedersen_verify_batch_rlc_affine_dev.argtypes = lib.vrfhip_pedersen_verify_batch_rlc_dev.argtypes
    lib.vrfhip_msm.argtypes = [c_void_p, c_size_t, P, P, P, P, P]
    lib.vrfhip_msm_dev.argtypes = [c_void_p, c_size_t, P, P, P, P, P, c_void_p]
    lib.vrfhip_pairing_check_batch.argtypes = [c_void_p, c_size_t, P, P, c_int32, P]
    lib.vrfhip_pairing_check_batch_dev.argtypes = [c_void_p, c_size_t, P, P, c_int32, P, c_void_p]
    lib.vrfhip_pairing_check_batch_rlc.argtypes = [c_void_p, c_size_t, P, P, P, P, POINTER(c_int32)]
    lib.vrfhip_pairing_check_batch_rlc_dev.argtypes = [c_void_p, c_size_t, P, P, P, P, P, c_void_p]
    lib.vrfhip_g1_msm.argtypes = [c_void_p, c_size_t, P, P, P, P]
    lib.vrfhip_g1_msm_dev.argtypes = [c_void_p, c_size_t, P, P, P, P, c_void_p]
    lib.vrfhip_hash_to_curve_batch.argtypes = [c_void_p, c_size_t, P, P, c_uint32, P]
    lib.vrfhip_hash_to_curve_batch_dev.argtypes = [c_void_p, c_size_t, P, P, c_uint32, P, c_void_p]
    lib.vrfhip_output_hash_batch.argtypes = [c_void_p, c_size_t, P, P]
    lib.vrfhip_output_hash_batch_dev.argtypes = [c_void_p, c_size_t, P, P, c_void_p]
    lib.vrfhip_secret_from_seed_batch.argtypes = [c_void_p, c_size_t, P, c_uint32, P, P]
    lib.vrfhip_secret_from_seed_batch_dev.argtypes = [c_void_p, c_size_t, P, c_uint32, P, P, c_void_p]
    lib.vrfhip_point_validate_batch.argtypes = [c_void_p, c_size_t, P, P, P]
    lib.vrfhip_point_validate_batch_dev.argtypes = [c_void_p, c_size_t, P, P, P, c_void_p]
    lib.vrfhip_te_sw_map_batch.argtypes = [c_void_p, c_size_t, c_int32, P, P, P]
    lib.vrfhip_te_sw_map_batch_dev.argtypes = [c_void_p, c_size_t, c_int32, P, P, P, c_void_p]
    lib.vrfhip_fq_mul_batch.argtypes = [c_void_p, c_size_t, P, P, P]
    lib.vrfhip_test_pairing_quad_ops.argtypes = [c_void_p, c_size_t, P, P]
    lib.vrfhip_test_pairing_oct_ops.argtypes = [c_void_p, c_size_t, P, P]
    lib.vrfhip_debug_set.argtypes = [c_void_p, c_int32, c_int32]
    lib.vrfhip_test_blinding_base.argtypes = [c_int32, P]
    lib.vrfhip_debug_proofs_per_lane.argtypes = [c_size_t]
    CP = POINTER(c_void_p)   # vrfhip_ctx* const*
    lib.vrfhip_ietf_verify_batch_multi.argtypes = [CP, c_int32, c_size_t, P, P, P, P, P, P, P, c_uint32, P]
    lib.vrfhip_ietf_prove_batch_multi.argtypes = [CP, c_int32, c_size_t, P, P, P, c_uint32, P, P, P, c_uint32,
                                                  P, P, P, P, P, P]
    lib.vrfhip_pedersen_prove_batch_multi.argtypes = [CP, c_int32, c_size_t, P, P, P, c_uint32, P, P, P, c_uint32,
                                                      P, P, P, P, P, P, P, P, P]
    lib.vrfhip_pedersen_verify_batch_multi.argtypes = [CP, c_int32, c_size_t, P, P, P, P, P, P, P, P, P, c_uint32, P, P]
    lib.vrfhip_test_point_add.argtypes = [c_void_p, c_size_t, P, P, P, P]
    lib.vrfhip_test_scalar_mul.argtypes = [c_void_p, c_size_t, P, P, P, P]
    lib.vrfhip_test_sha512.argtypes = [c_void_p, c_size_t, P, P, c_uint32, P]
    lib.vrfhip_test_xmd.argtypes = [c_void_p, c_size_t, P, P, c_uint32, P]
    lib.vrfhip_test_batch_digest.argtypes = [c_void_p, c_size_t, c_int32, P, P, P, P, c_uint32, ctypes.c_uint64, P]
    for name in SYMBOLS:
        fn = getattr(lib, name)
        if name not in ("vrfhip_last_error", "vrfhip_ctx_destroy", "vrfhip_ctx_workspace_bytes", "vrfhip_host_free",
                        "vrfhip_keyset_destroy", "vrfhip_keyset_bytes", "vrfhip_ctx_get_flags", "vrfhip_ctx_point_bytes",
                        "vrfhip_ctx_hash_bytes"):
            fn.restype = c_int32
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().vrfhip_last_error().decode("utf-8", "replace")
        raise VrfHipError(f"{what} failed with {rc}: {msg}")
