"""ark_ec_vrfs_amd -- MI355X-native batch EC-VRF prove/verify behind the ark-ec-vrfs API.

Host-side mirror (Python) of the Rust surface re-exported at /root/reference src/lib.rs:13-17
(`Suite`, `Secret`, `Public`, `Input`, `Output`, `ietf::{Prover, Verifier, Proof}`, `Error`)
over the C ABI of ``csrc/libvrfhip.so``.  All arithmetic runs in hand-written HIP kernels.
"""
from .api import (  # noqa: F401
    KeySet, SuiteDesc, PinnedBuffer, ietf_verify_batch_multi, ietf_prove_batch_multi, pedersen_prove_batch_multi,
    pedersen_verify_batch_multi, CURVE_BANDERSNATCH, CURVE_JUBJUB, CURVE_ED25519, CURVE_BABY_JUBJUB, CURVE_SECP256R1, CURVE_BANDERSNATCH_SW,
    BandersnatchSha512Ell2, JubJubSha512Tai, Ed25519Sha512Tai, BabyJubJubSha512Tai, Secp256r1Sha256Tai, BandersnatchSwSha512Tai, Context, Error, IetfProof, Input, Output, Public, Secret, Suite,
    VerificationFailure, InvalidData, ietf, pedersen, PedersenProof, default_context,
)
from ._lib import VrfHipError, LIB_PATH  # noqa: F401
