//! `ark-ec-vrfs-hip` -- the `ark-ec-vrfs` API with an MI355X batch backend.
//!
//! The reference crate (`/root/reference/src/lib.rs:13-17`) is a re-export of `ark_vrf`; this crate re-exports the
//! same list, so existing code keeps compiling, and adds [`GpuBatch`]: batches of `prove` / `verify` routed through
//! libvrfhip's C ABI (`include/vrfhip.h`, raw bindings in [`ffi`]).  Single items keep using the arkworks CPU path
//! (`BASELINE.json` configs[0]); a batch is where the GPU pays.
//!
//! **Status: source only.**  The build image has no Rust toolchain and the arithmetic crates are not vendored, so
//! this file has never been compiled.  The `ark_vrf` items it names are the ones the reference re-exports; their exact
//! signatures (`codec::point_encode`, `Secret::scalar`, `PedersenSuite::BLINDING_BASE`, ...) are recollections of
//! upstream and may need touching up against the real crate.  What is machine-checked here is the FFI boundary:
//! `src/ffi.rs` is generated from the C header and `tests/test_rust_shim.py` compares the two item by item.
#![cfg_attr(not(feature = "std"), no_std)]
extern crate alloc;

use alloc::vec;
use alloc::vec::Vec;
use core::marker::PhantomData;

// the reference's whole public surface (src/lib.rs:13-17), unchanged
pub use ark_vrf::{
    codec, ietf, pedersen, reexports, ring, ring_suite_types, suite_types, suites, utils, AffinePoint, BaseField,
    CurveConfig, Error, HashOutput, Input, Output, Public, ScalarField, Secret, Suite,
};

pub mod ffi;

use ark_vrf::pedersen::PedersenSuite;
use ark_vrf::reexports::ark_serialize::{CanonicalDeserialize, CanonicalSerialize};

/// Failure of the library itself (bad argument, HIP error, out of memory, no device): `vrfhip_error` + message.
/// Per-item outcomes are `ark_vrf::Error` values, exactly as the CPU API reports them.
#[derive(Debug, Clone)]
pub struct GpuError {
    pub code: i32,
    pub message: alloc::string::String,
}

fn check(rc: i32) -> Result<(), GpuError> {
    if rc == ffi::VRFHIP_SUCCESS {
        return Ok(());
    }
    let msg = unsafe { core::ffi::CStr::from_ptr(ffi::vrfhip_last_error()) };
    Err(GpuError { code: rc, message: msg.to_string_lossy().into_owned() })
}

fn status_to_result(st: u8) -> Result<(), Error> {
    match st as i32 {
        ffi::VRFHIP_ST_OK => Ok(()),
        ffi::VRFHIP_ST_VERIFICATION_FAILURE => Err(Error::VerificationFailure),
        _ => Err(Error::InvalidData),
    }
}

/// A suite libvrfhip has arithmetic for.  Everything a `Suite` states as data travels in the descriptor, filled from
/// the trait's own constants, so the GPU runs exactly the suite the CPU code runs (in particular the upstream JubJub
/// suite string and blinding base, which libvrfhip's built-in JubJub descriptor does not claim to know).
pub trait GpuSuite: Suite + PedersenSuite {
    /// `ffi::VRFHIP_CURVE_*`: which compiled curve arithmetic (and hash-to-curve construction) applies.
    const CURVE: i32;
    /// RFC 9380 domain separation tag of `Suite::data_to_point` (Elligator suites); empty for try-and-increment.
    fn h2c_dst() -> Vec<u8>;

    fn descriptor() -> ffi::vrfhip_suite_desc {
        let mut d = ffi::vrfhip_suite_desc {
            struct_size: core::mem::size_of::<ffi::vrfhip_suite_desc>() as u32,
            curve: Self::CURVE,
            suite_id_len: Self::SUITE_ID.len() as u32,
            suite_id: [0u8; 64],
            h2c_dst_len: 0,
            h2c_dst: [0u8; 128],
            generator: [0u8; 64],
            blinding_base: [0u8; 64],
            challenge_len: Self::CHALLENGE_LEN as u32,
            flags: 0, // upstream's suites: ArkworksCodec sign flag, big-endian challenge, no cofactor in `Output::hash`
        };
        d.suite_id[..Self::SUITE_ID.len()].copy_from_slice(Self::SUITE_ID);
        let dst = Self::h2c_dst();
        d.h2c_dst_len = dst.len() as u32;
        d.h2c_dst[..dst.len()].copy_from_slice(&dst);
        affine_xy::<Self>(&Self::generator(), &mut d.generator);
        affine_xy::<Self>(&Self::BLINDING_BASE, &mut d.blinding_base);
        d
    }
}

impl GpuSuite for suites::bandersnatch::BandersnatchSha512Ell2 {
    const CURVE: i32 = ffi::VRFHIP_CURVE_BANDERSNATCH;
    fn h2c_dst() -> Vec<u8> {
        // "ECVRF_" || h2c suite id || SUITE_ID (authenticated by the golden vectors)
        [b"ECVRF_".as_slice(), b"Bandersnatch_XMD:SHA-512_ELL2_RO_", Self::SUITE_ID].concat()
    }
}

impl GpuSuite for suites::jubjub::JubJubSha512Tai {
    const CURVE: i32 = ffi::VRFHIP_CURVE_JUBJUB;
    fn h2c_dst() -> Vec<u8> {
        Vec::new()
    }
}

// The suites over the other base fields (`full` feature upstream): Ed25519 (2^255 - 19, `CHALLENGE_LEN` 16) and
// Baby-JubJub (BN254 Fr).  Suite string, challenge length, generator and blinding base travel in the descriptor,
// taken from the trait, so the library's own recollections of them (vrfhip_suite_desc_default) never matter here.
impl GpuSuite for suites::ed25519::Ed25519Sha512Tai {
    const CURVE: i32 = ffi::VRFHIP_CURVE_ED25519;
    fn h2c_dst() -> Vec<u8> {
        Vec::new()
    }
}

impl GpuSuite for suites::baby_jubjub::BabyJubJubSha512Tai {
    const CURVE: i32 = ffi::VRFHIP_CURVE_BABY_JUBJUB;
    fn h2c_dst() -> Vec<u8> {
        Vec::new()
    }
}

/// x || y, 32-byte little-endian canonical integers (arkworks' uncompressed field encoding, twice)
fn affine_xy<S: Suite>(p: &AffinePoint<S>, out: &mut [u8; 64]) {
    p.x.serialize_uncompressed(&mut out[..32]).expect("32-byte base field");
    p.y.serialize_uncompressed(&mut out[32..]).expect("32-byte base field");
}

fn point32<S: Suite>(p: &AffinePoint<S>, out: &mut [u8]) {
    let mut buf = Vec::with_capacity(32);
    codec::point_encode::<S>(p, &mut buf);
    out.copy_from_slice(&buf);
}

/// A point the GPU produced, from its x || y form (`VRFHIP_FLAG_PROVE_POINTS_AFFINE`): two field-element loads, no
/// square root and no subgroup test -- the prover's outputs are multiples of points that were validated on the way in.
fn point_from_xy<S: Suite>(xy: &[u8]) -> AffinePoint<S> {
    let x = BaseField::<S>::deserialize_uncompressed_unchecked(&xy[..32]).expect("canonical x from the GPU");
    let y = BaseField::<S>::deserialize_uncompressed_unchecked(&xy[32..64]).expect("canonical y from the GPU");
    AffinePoint::<S>::new_unchecked(x, y)
}

fn scalar32<S: Suite>(k: &ScalarField<S>, out: &mut [u8]) {
    let mut buf = Vec::with_capacity(32);
    codec::scalar_encode::<S>(k, &mut buf);
    out.copy_from_slice(&buf);
}

/// One context (one GPU) per element; batches are cut into contiguous slices, one host thread per device
/// (`vrfhip_*_batch_multi`).  `&self` only and internally synchronised, like the CPU API's pure functions.
pub struct GpuBatch<S: GpuSuite> {
    ctxs: Vec<*mut ffi::vrfhip_ctx>,
    _suite: PhantomData<S>,
}

unsafe impl<S: GpuSuite> Send for GpuBatch<S> {}
unsafe impl<S: GpuSuite> Sync for GpuBatch<S> {}

impl<S: GpuSuite> GpuBatch<S> {
    /// One context per listed device, all created from `S`'s own constants.
    pub fn new(devices: &[i32]) -> Result<Self, GpuError> {
        let desc = S::descriptor();
        let mut this = GpuBatch { ctxs: Vec::new(), _suite: PhantomData };
        for &dev in devices {
            let mut ctx: *mut ffi::vrfhip_ctx = core::ptr::null_mut();
            check(unsafe { ffi::vrfhip_ctx_create_desc(&desc, dev, &mut ctx) })?;
            this.ctxs.push(ctx);
            // the provers hand back points as x || y: a typed `Output` / `Proof` then costs two field loads per point
            // instead of a square root (`codec::point_decode`), which would dwarf the GPU's own time
            check(unsafe { ffi::vrfhip_ctx_set_flags(ctx, ffi::VRFHIP_FLAG_PROVE_POINTS_AFFINE) })?;
        }
        Ok(this)
    }

    /// The points handed to `verify` are typed arkworks values: their subgroup membership was established when they
    /// were deserialised, so the library may skip its own test (`VRFHIP_FLAG_PREVALIDATED_*`).  Off by default.
    pub fn trust_typed_points(&self, on: bool) -> Result<(), GpuError> {
        let flags = ffi::VRFHIP_FLAG_PROVE_POINTS_AFFINE | if on { ffi::VRFHIP_FLAG_PREVALIDATED_ALL } else { 0 };
        for &c in &self.ctxs {
            check(unsafe { ffi::vrfhip_ctx_set_flags(c, flags) })?;
        }
        Ok(())
    }

    /// `Secret::output` + `ietf::Prover::prove` for every (secret, input) pair.  Per item: the proof, or the `Error` the
    /// library's status byte names (a failed item's points come back all-zero and are never turned into typed values).
    pub fn ietf_prove(
        &self,
        secrets: &[Secret<S>],
        inputs: &[Input<S>],
        ad: &[u8],
    ) -> Result<Vec<Result<(Output<S>, ietf::Proof<S>), Error>>, GpuError> {
        let n = secrets.len();
        assert_eq!(n, inputs.len());
        let (mut sk, mut h) = (vec![0u8; n * 32], vec![0u8; n * 32]);
        for i in 0..n {
            scalar32::<S>(&secrets[i].scalar, &mut sk[i * 32..(i + 1) * 32]);
            point32::<S>(&inputs[i].0, &mut h[i * 32..(i + 1) * 32]);
        }
        let (mut gamma, mut c, mut s) = (vec![0u8; n * 64], vec![0u8; n * 32], vec![0u8; n * 32]); // gamma: x || y
        let mut status = vec![0u8; n];
        check(unsafe {
            ffi::vrfhip_ietf_prove_batch_multi(
                self.ctxs.as_ptr(), self.ctxs.len() as i32, n, sk.as_ptr(), core::ptr::null(), core::ptr::null(), 0,
                h.as_ptr(), ad.as_ptr(), core::ptr::null(), ad.len() as u32, gamma.as_mut_ptr(), c.as_mut_ptr(),
                s.as_mut_ptr(), core::ptr::null_mut(), core::ptr::null_mut(), status.as_mut_ptr(),
            )
        })?;
        sk.iter_mut().for_each(|b| *b = 0); // the staged copy of the secrets
        Ok((0..n)
            .map(|i| {
                status_to_result(status[i])?;
                let out = Output::<S>::from(point_from_xy::<S>(&gamma[i * 64..(i + 1) * 64]));
                let proof = ietf::Proof::<S> {
                    c: codec::scalar_decode::<S>(&c[i * 32..(i + 1) * 32]),
                    s: codec::scalar_decode::<S>(&s[i * 32..(i + 1) * 32]),
                };
                Ok((out, proof))
            })
            .collect())
    }

    /// `ietf::Verifier::verify(&public, input, output, ad, &proof)` for every item: `Ok(())`,
    /// `Err(Error::VerificationFailure)` or `Err(Error::InvalidData)` per item, as the CPU verifier reports.
    pub fn ietf_verify(
        &self,
        publics: &[Public<S>],
        inputs: &[Input<S>],
        outputs: &[Output<S>],
        ad: &[u8],
        proofs: &[ietf::Proof<S>],
    ) -> Result<Vec<Result<(), Error>>, GpuError> {
        let n = publics.len();
        assert!(n == inputs.len() && n == outputs.len() && n == proofs.len());
        let mut buf = vec![0u8; 5 * n * 32];
        {
            let (pk, rest) = buf.split_at_mut(n * 32);
            let (h, rest) = rest.split_at_mut(n * 32);
            let (g, rest) = rest.split_at_mut(n * 32);
            let (c, s) = rest.split_at_mut(n * 32);
            for i in 0..n {
                let r = i * 32..(i + 1) * 32;
                point32::<S>(&publics[i].0, &mut pk[r.clone()]);
                point32::<S>(&inputs[i].0, &mut h[r.clone()]);
                point32::<S>(&outputs[i].0, &mut g[r.clone()]);
                scalar32::<S>(&proofs[i].c, &mut c[r.clone()]);
                scalar32::<S>(&proofs[i].s, &mut s[r]);
            }
        }
        let mut status = vec![0u8; n];
        let p = buf.as_ptr();
        check(unsafe {
            ffi::vrfhip_ietf_verify_batch_multi(
                self.ctxs.as_ptr(), self.ctxs.len() as i32, n, p, p.add(n * 32), p.add(2 * n * 32), p.add(3 * n * 32),
                p.add(4 * n * 32), ad.as_ptr(), core::ptr::null(), ad.len() as u32, status.as_mut_ptr(),
            )
        })?;
        Ok(status.into_iter().map(status_to_result).collect())
    }

    /// `Input::new(alpha)` followed by `ietf::Verifier::verify` for every item -- what a verifier that holds public keys,
    /// messages and proofs runs.  The inputs are hashed to the curve on the GPU and never leave it
    /// (`vrfhip_ietf_verify_batch_alpha`); one launch group on the first device.
    pub fn ietf_verify_from_alpha(
        &self,
        publics: &[Public<S>],
        alphas: &[&[u8]],
        outputs: &[Output<S>],
        ad: &[u8],
        proofs: &[ietf::Proof<S>],
    ) -> Result<Vec<Result<(), Error>>, GpuError> {
        let n = publics.len();
        assert!(n == alphas.len() && n == outputs.len() && n == proofs.len());
        let mut buf = vec![0u8; 4 * n * 32];
        {
            let (pk, rest) = buf.split_at_mut(n * 32);
            let (g, rest) = rest.split_at_mut(n * 32);
            let (c, s) = rest.split_at_mut(n * 32);
            for i in 0..n {
                let r = i * 32..(i + 1) * 32;
                point32::<S>(&publics[i].0, &mut pk[r.clone()]);
                point32::<S>(&outputs[i].0, &mut g[r.clone()]);
                scalar32::<S>(&proofs[i].c, &mut c[r.clone()]);
                scalar32::<S>(&proofs[i].s, &mut s[r]);
            }
        }
        let mut msg = Vec::new();
        let mut off = vec![0u32; n + 1];
        for (i, a) in alphas.iter().enumerate() {
            msg.extend_from_slice(a);
            off[i + 1] = msg.len() as u32;
        }
        msg.push(0);
        let mut status = vec![0u8; n];
        let p = buf.as_ptr();
        check(unsafe {
            ffi::vrfhip_ietf_verify_batch_alpha(
                self.ctxs[0], n, p, msg.as_ptr(), off.as_ptr(), 0, p.add(n * 32), p.add(2 * n * 32), p.add(3 * n * 32),
                ad.as_ptr(), core::ptr::null(), ad.len() as u32, status.as_mut_ptr(),
            )
        })?;
        Ok(status.into_iter().map(status_to_result).collect())
    }

    /// `pedersen::Prover::prove`: (output, proof, blinding factor) per item, or the item's `Error`.
    pub fn pedersen_prove(
        &self,
        secrets: &[Secret<S>],
        inputs: &[Input<S>],
        ad: &[u8],
    ) -> Result<Vec<Result<(Output<S>, pedersen::Proof<S>, ScalarField<S>), Error>>, GpuError> {
        let n = secrets.len();
        assert_eq!(n, inputs.len());
        let (mut sk, mut h) = (vec![0u8; n * 32], vec![0u8; n * 32]);
        for i in 0..n {
            scalar32::<S>(&secrets[i].scalar, &mut sk[i * 32..(i + 1) * 32]);
            point32::<S>(&inputs[i].0, &mut h[i * 32..(i + 1) * 32]);
        }
        let mut p = vec![0u8; 4 * n * 64]; // gamma | pk_com | r | ok, x || y each
        let mut o = vec![0u8; 3 * n * 32]; // s | sb | blinding
        let mut status = vec![0u8; n];
        let (q, w) = (p.as_mut_ptr(), o.as_mut_ptr());
        check(unsafe {
            ffi::vrfhip_pedersen_prove_batch_multi(
                self.ctxs.as_ptr(), self.ctxs.len() as i32, n, sk.as_ptr(), core::ptr::null(), core::ptr::null(), 0,
                h.as_ptr(), ad.as_ptr(), core::ptr::null(), ad.len() as u32, q, q.add(n * 64), q.add(2 * n * 64),
                q.add(3 * n * 64), w, w.add(n * 32), w.add(2 * n * 32), core::ptr::null_mut(),
                status.as_mut_ptr(),
            )
        })?;
        sk.iter_mut().for_each(|b| *b = 0);
        let pt = |k: usize, i: usize| point_from_xy::<S>(&p[(k * n + i) * 64..(k * n + i + 1) * 64]);
        let sc = |k: usize, i: usize| codec::scalar_decode::<S>(&o[(k * n + i) * 32..(k * n + i + 1) * 32]);
        let res = (0..n)
            .map(|i| {
                status_to_result(status[i])?;
                let proof = pedersen::Proof::<S> { pk_com: pt(1, i), r: pt(2, i), ok: pt(3, i), s: sc(0, i), sb: sc(1, i) };
                Ok((Output::<S>::from(pt(0, i)), proof, sc(2, i)))
            })
            .collect();
        o.iter_mut().for_each(|b| *b = 0); // blinding factors
        Ok(res)
    }

    /// `pedersen::Verifier::verify` per item.  `batched`: one multi-scalar multiplication per device slice (random
    /// linear combination, fresh weights from `seed`), falling back to the per-proof kernels when the batch fails --
    /// same statuses either way.
    pub fn pedersen_verify(
        &self,
        inputs: &[Input<S>],
        outputs: &[Output<S>],
        ad: &[u8],
        proofs: &[pedersen::Proof<S>],
        batched_seed: Option<&[u8; 32]>,
    ) -> Result<Vec<Result<(), Error>>, GpuError> {
        let n = inputs.len();
        assert!(n == outputs.len() && n == proofs.len());
        let mut buf = vec![0u8; 7 * n * 32]; // h | gamma | pk_com | r | ok | s | sb
        for i in 0..n {
            let at = |k: usize| (k * n + i) * 32..(k * n + i + 1) * 32;
            point32::<S>(&inputs[i].0, &mut buf[at(0)]);
            point32::<S>(&outputs[i].0, &mut buf[at(1)]);
            point32::<S>(&proofs[i].pk_com, &mut buf[at(2)]);
            point32::<S>(&proofs[i].r, &mut buf[at(3)]);
            point32::<S>(&proofs[i].ok, &mut buf[at(4)]);
            scalar32::<S>(&proofs[i].s, &mut buf[at(5)]);
            scalar32::<S>(&proofs[i].sb, &mut buf[at(6)]);
        }
        let mut status = vec![0u8; n];
        let p = buf.as_ptr();
        let seed = batched_seed.map_or(core::ptr::null(), |s| s.as_ptr());
        check(unsafe {
            ffi::vrfhip_pedersen_verify_batch_multi(
                self.ctxs.as_ptr(), self.ctxs.len() as i32, n, p, p.add(n * 32), p.add(2 * n * 32), p.add(3 * n * 32),
                p.add(4 * n * 32), p.add(5 * n * 32), p.add(6 * n * 32), ad.as_ptr(), core::ptr::null(),
                ad.len() as u32, seed, status.as_mut_ptr(),
            )
        })?;
        Ok(status.into_iter().map(status_to_result).collect())
    }
}

impl<S: GpuSuite> GpuBatch<S> {
    /// `utils::te_sw_map::te_to_sw` for every point: the coordinates of its image on the short-Weierstrass form of the
    /// curve (the caller wraps them: `WeierstrassAffine::new_unchecked(x, y)`), or `None` where upstream's map is `None`
    /// (the identity and the point of order 2).  One launch on the first device.
    pub fn te_to_sw(&self, points: &[AffinePoint<S>]) -> Result<Vec<Option<(BaseField<S>, BaseField<S>)>>, GpuError> {
        let n = points.len();
        let mut inp = vec![0u8; n * 64];
        for (i, p) in points.iter().enumerate() {
            let mut xy = [0u8; 64];
            affine_xy::<S>(p, &mut xy);
            inp[i * 64..(i + 1) * 64].copy_from_slice(&xy);
        }
        let (out, status) = self.te_sw_map(n, 0, &inp)?;
        Ok((0..n)
            .map(|i| {
                (status[i] == 0).then(|| {
                    let x = BaseField::<S>::deserialize_uncompressed_unchecked(&out[i * 64..i * 64 + 32]).expect("canonical x");
                    let y = BaseField::<S>::deserialize_uncompressed_unchecked(&out[i * 64 + 32..i * 64 + 64]).expect("canonical y");
                    (x, y)
                })
            })
            .collect())
    }

    /// `utils::te_sw_map::sw_to_te`: short-Weierstrass coordinates back to points of `S`'s twisted-Edwards curve (no
    /// curve-membership test, as upstream's `new_unchecked`).
    pub fn sw_to_te(&self, points: &[(BaseField<S>, BaseField<S>)]) -> Result<Vec<Option<AffinePoint<S>>>, GpuError> {
        let n = points.len();
        let mut inp = vec![0u8; n * 64];
        for (i, (x, y)) in points.iter().enumerate() {
            x.serialize_uncompressed(&mut inp[i * 64..i * 64 + 32]).expect("32-byte base field");
            y.serialize_uncompressed(&mut inp[i * 64 + 32..i * 64 + 64]).expect("32-byte base field");
        }
        let (out, status) = self.te_sw_map(n, 1, &inp)?;
        Ok((0..n).map(|i| (status[i] == 0).then(|| point_from_xy::<S>(&out[i * 64..(i + 1) * 64]))).collect())
    }

    fn te_sw_map(&self, n: usize, to_te: i32, inp: &[u8]) -> Result<(Vec<u8>, Vec<u8>), GpuError> {
        let mut out = vec![0u8; n * 64];
        let mut status = vec![0u8; n];
        check(unsafe {
            ffi::vrfhip_te_sw_map_batch(self.ctxs[0], n, to_te, inp.as_ptr(), out.as_mut_ptr(), status.as_mut_ptr())
        })?;
        Ok((out, status))
    }
}

impl<S: GpuSuite> Drop for GpuBatch<S> {
    fn drop(&mut self) {
        for &c in &self.ctxs {
            unsafe { ffi::vrfhip_ctx_destroy(c) }; // wipes staged secrets before freeing (api.hip)
        }
    }
}

/// `suites::secp256r1` ("P256_SHA256_TAI"): the suite whose codec is `Sec1Codec` -- 33-byte compressed points, big-endian
/// scalars -- and whose hash is SHA-256.  libvrfhip runs it behind the same entry points with that wire format
/// (`VRFHIP_SUITE_SECP256R1_SHA256_TAI`; `vrfhip_ctx_point_bytes` = 33): points go in as `codec::point_encode` strings
/// (compressing costs no square root) and come back as x || y (`VRFHIP_FLAG_PROVE_POINTS_AFFINE`), so that a typed
/// `Output` / proof point is two field loads, not a `codec::point_decode` (a square root per point on a CPU core).
pub struct GpuBatchSec1 {
    ctxs: Vec<*mut ffi::vrfhip_ctx>,
}

unsafe impl Send for GpuBatchSec1 {}
unsafe impl Sync for GpuBatchSec1 {}

type P256 = suites::secp256r1::P256Sha256Tai;
const SEC1: usize = 33;

impl GpuBatchSec1 {
    pub fn new(devices: &[i32]) -> Result<Self, GpuError> {
        let mut this = GpuBatchSec1 { ctxs: Vec::new() };
        for &dev in devices {
            let mut ctx: *mut ffi::vrfhip_ctx = core::ptr::null_mut();
            check(unsafe { ffi::vrfhip_ctx_create(ffi::VRFHIP_SUITE_SECP256R1_SHA256_TAI, dev, &mut ctx) })?;
            debug_assert_eq!(unsafe { ffi::vrfhip_ctx_point_bytes(ctx) }, SEC1);
            this.ctxs.push(ctx);
            check(unsafe { ffi::vrfhip_ctx_set_flags(ctx, ffi::VRFHIP_FLAG_PROVE_POINTS_AFFINE) })?;
        }
        Ok(this)
    }

    fn put_point(p: &AffinePoint<P256>, out: &mut [u8]) {
        let mut v = Vec::with_capacity(SEC1);
        codec::point_encode_into::<P256>(p, &mut v);
        out.copy_from_slice(&v); // a `Public` / `Input` / `Output` is never the point at infinity: always 33 bytes
    }

    fn put_scalar(k: &ScalarField<P256>, out: &mut [u8]) {
        let mut v = Vec::with_capacity(32);
        codec::scalar_encode_into::<P256>(k, &mut v);
        out.copy_from_slice(&v);
    }

    /// `Secret::output` + `ietf::Prover::prove` per (secret, input) pair.
    pub fn ietf_prove(
        &self,
        secrets: &[Secret<P256>],
        inputs: &[Input<P256>],
        ad: &[u8],
    ) -> Result<Vec<Result<(Output<P256>, ietf::Proof<P256>), Error>>, GpuError> {
        let n = secrets.len();
        assert_eq!(n, inputs.len());
        let (mut sk, mut h) = (vec![0u8; n * 32], vec![0u8; n * SEC1]);
        for i in 0..n {
            Self::put_scalar(&secrets[i].scalar, &mut sk[i * 32..(i + 1) * 32]);
            Self::put_point(&inputs[i].0, &mut h[i * SEC1..(i + 1) * SEC1]);
        }
        let (mut gamma, mut c, mut s) = (vec![0u8; n * 64], vec![0u8; n * 32], vec![0u8; n * 32]); // gamma: x || y
        let mut status = vec![0u8; n];
        check(unsafe {
            ffi::vrfhip_ietf_prove_batch_multi(
                self.ctxs.as_ptr(), self.ctxs.len() as i32, n, sk.as_ptr(), core::ptr::null(), core::ptr::null(), 0,
                h.as_ptr(), ad.as_ptr(), core::ptr::null(), ad.len() as u32, gamma.as_mut_ptr(), c.as_mut_ptr(),
                s.as_mut_ptr(), core::ptr::null_mut(), core::ptr::null_mut(), status.as_mut_ptr(),
            )
        })?;
        sk.iter_mut().for_each(|b| *b = 0);
        Ok((0..n)
            .map(|i| {
                status_to_result(status[i])?;
                let out = Output::<P256>::from(point_from_xy::<P256>(&gamma[i * 64..(i + 1) * 64]));
                let proof = ietf::Proof::<P256> {
                    c: codec::scalar_decode::<P256>(&c[i * 32 + 16..(i + 1) * 32]), // the challenge's 16 significant bytes
                    s: codec::scalar_decode::<P256>(&s[i * 32..(i + 1) * 32]),
                };
                Ok((out, proof))
            })
            .collect())
    }

    /// `ietf::Verifier::verify` per item.
    pub fn ietf_verify(
        &self,
        publics: &[Public<P256>],
        inputs: &[Input<P256>],
        outputs: &[Output<P256>],
        ad: &[u8],
        proofs: &[ietf::Proof<P256>],
    ) -> Result<Vec<Result<(), Error>>, GpuError> {
        let n = publics.len();
        assert!(n == inputs.len() && n == outputs.len() && n == proofs.len());
        let (mut pk, mut h, mut g) = (vec![0u8; n * SEC1], vec![0u8; n * SEC1], vec![0u8; n * SEC1]);
        let (mut c, mut s) = (vec![0u8; n * 32], vec![0u8; n * 32]);
        for i in 0..n {
            Self::put_point(&publics[i].0, &mut pk[i * SEC1..(i + 1) * SEC1]);
            Self::put_point(&inputs[i].0, &mut h[i * SEC1..(i + 1) * SEC1]);
            Self::put_point(&outputs[i].0, &mut g[i * SEC1..(i + 1) * SEC1]);
            Self::put_scalar(&proofs[i].c, &mut c[i * 32..(i + 1) * 32]);
            Self::put_scalar(&proofs[i].s, &mut s[i * 32..(i + 1) * 32]);
        }
        let mut status = vec![0u8; n];
        check(unsafe {
            ffi::vrfhip_ietf_verify_batch_multi(
                self.ctxs.as_ptr(), self.ctxs.len() as i32, n, pk.as_ptr(), h.as_ptr(), g.as_ptr(), c.as_ptr(), s.as_ptr(),
                ad.as_ptr(), core::ptr::null(), ad.len() as u32, status.as_mut_ptr(),
            )
        })?;
        Ok(status.into_iter().map(status_to_result).collect())
    }

    /// `Input::new(alpha)` followed by `ietf::Verifier::verify` for every item (`vrfhip_ietf_verify_batch_alpha`): the
    /// inputs are hashed to the curve on the GPU; one launch group on the first device.
    pub fn ietf_verify_from_alpha(
        &self,
        publics: &[Public<P256>],
        alphas: &[&[u8]],
        outputs: &[Output<P256>],
        ad: &[u8],
        proofs: &[ietf::Proof<P256>],
    ) -> Result<Vec<Result<(), Error>>, GpuError> {
        let n = publics.len();
        assert!(n == alphas.len() && n == outputs.len() && n == proofs.len());
        let (mut pk, mut g) = (vec![0u8; n * SEC1], vec![0u8; n * SEC1]);
        let (mut c, mut s) = (vec![0u8; n * 32], vec![0u8; n * 32]);
        let mut msg = Vec::new();
        let mut off = vec![0u32; n + 1];
        for i in 0..n {
            Self::put_point(&publics[i].0, &mut pk[i * SEC1..(i + 1) * SEC1]);
            Self::put_point(&outputs[i].0, &mut g[i * SEC1..(i + 1) * SEC1]);
            Self::put_scalar(&proofs[i].c, &mut c[i * 32..(i + 1) * 32]);
            Self::put_scalar(&proofs[i].s, &mut s[i * 32..(i + 1) * 32]);
            msg.extend_from_slice(alphas[i]);
            off[i + 1] = msg.len() as u32;
        }
        msg.push(0);
        let mut status = vec![0u8; n];
        check(unsafe {
            ffi::vrfhip_ietf_verify_batch_alpha(
                self.ctxs[0], n, pk.as_ptr(), msg.as_ptr(), off.as_ptr(), 0, g.as_ptr(), c.as_ptr(), s.as_ptr(), ad.as_ptr(),
                core::ptr::null(), ad.len() as u32, status.as_mut_ptr(),
            )
        })?;
        Ok(status.into_iter().map(status_to_result).collect())
    }
}

impl GpuBatchSec1 {
    /// `pedersen::Prover::prove` per (secret, input) pair: (output, proof, blinding factor) or the item's `Error`.  The
    /// contexts must have been made from a descriptor that carries upstream's `BLINDING_BASE` for the suite
    /// ([`GpuBatchSec1::with_blinding_base`]): the library's default descriptor of this suite carries none, and these calls
    /// then answer `VRFHIP_ERR_UNSUPPORTED` instead of proving with an invented point.
    pub fn pedersen_prove(
        &self,
        secrets: &[Secret<P256>],
        inputs: &[Input<P256>],
        ad: &[u8],
    ) -> Result<Vec<Result<(Output<P256>, pedersen::Proof<P256>, ScalarField<P256>), Error>>, GpuError> {
        let n = secrets.len();
        assert_eq!(n, inputs.len());
        let (mut sk, mut h) = (vec![0u8; n * 32], vec![0u8; n * SEC1]);
        for i in 0..n {
            Self::put_scalar(&secrets[i].scalar, &mut sk[i * 32..(i + 1) * 32]);
            Self::put_point(&inputs[i].0, &mut h[i * SEC1..(i + 1) * SEC1]);
        }
        let mut p = vec![0u8; 4 * n * 64]; // gamma | pk_com | r | ok, x || y each
        let mut o = vec![0u8; 3 * n * 32]; // s | sb | blinding
        let mut status = vec![0u8; n];
        let (q, w) = (p.as_mut_ptr(), o.as_mut_ptr());
        check(unsafe {
            ffi::vrfhip_pedersen_prove_batch_multi(
                self.ctxs.as_ptr(), self.ctxs.len() as i32, n, sk.as_ptr(), core::ptr::null(), core::ptr::null(), 0,
                h.as_ptr(), ad.as_ptr(), core::ptr::null(), ad.len() as u32, q, q.add(n * 64), q.add(2 * n * 64),
                q.add(3 * n * 64), w, w.add(n * 32), w.add(2 * n * 32), core::ptr::null_mut(),
                status.as_mut_ptr(),
            )
        })?;
        sk.iter_mut().for_each(|b| *b = 0);
        let pt = |k: usize, i: usize| point_from_xy::<P256>(&p[(k * n + i) * 64..(k * n + i + 1) * 64]);
        let sc = |k: usize, i: usize| codec::scalar_decode::<P256>(&o[(k * n + i) * 32..(k * n + i + 1) * 32]);
        let res = (0..n)
            .map(|i| {
                status_to_result(status[i])?;
                let proof = pedersen::Proof::<P256> { pk_com: pt(1, i), r: pt(2, i), ok: pt(3, i), s: sc(0, i), sb: sc(1, i) };
                Ok((Output::<P256>::from(pt(0, i)), proof, sc(2, i)))
            })
            .collect();
        o.iter_mut().for_each(|b| *b = 0); // blinding factors
        Ok(res)
    }

    /// `pedersen::Verifier::verify` per item (per-proof kernels; the batched verifier is not built for this suite).
    pub fn pedersen_verify(
        &self,
        inputs: &[Input<P256>],
        outputs: &[Output<P256>],
        ad: &[u8],
        proofs: &[pedersen::Proof<P256>],
    ) -> Result<Vec<Result<(), Error>>, GpuError> {
        let n = inputs.len();
        assert!(n == outputs.len() && n == proofs.len());
        let mut pts = vec![0u8; 5 * n * SEC1]; // h | gamma | pk_com | r | ok
        let mut sc = vec![0u8; 2 * n * 32]; // s | sb
        for i in 0..n {
            let at = |k: usize| (k * n + i) * SEC1..(k * n + i + 1) * SEC1;
            Self::put_point(&inputs[i].0, &mut pts[at(0)]);
            Self::put_point(&outputs[i].0, &mut pts[at(1)]);
            Self::put_point(&proofs[i].pk_com, &mut pts[at(2)]);
            Self::put_point(&proofs[i].r, &mut pts[at(3)]);
            Self::put_point(&proofs[i].ok, &mut pts[at(4)]);
            Self::put_scalar(&proofs[i].s, &mut sc[i * 32..(i + 1) * 32]);
            Self::put_scalar(&proofs[i].sb, &mut sc[(n + i) * 32..(n + i + 1) * 32]);
        }
        let mut status = vec![0u8; n];
        let (p, q) = (pts.as_ptr(), sc.as_ptr());
        check(unsafe {
            ffi::vrfhip_pedersen_verify_batch_multi(
                self.ctxs.as_ptr(), self.ctxs.len() as i32, n, p, p.add(n * SEC1), p.add(2 * n * SEC1), p.add(3 * n * SEC1),
                p.add(4 * n * SEC1), q, q.add(n * 32), ad.as_ptr(), core::ptr::null(), ad.len() as u32, core::ptr::null(),
                status.as_mut_ptr(),
            )
        })?;
        Ok(status.into_iter().map(status_to_result).collect())
    }

    /// Contexts whose descriptor carries the suite's own `PedersenSuite::BLINDING_BASE` (x || y, little-endian, as the
    /// descriptor states every point).  The default descriptor leaves the field all-zero (no Pedersen scheme): upstream's
    /// constant is pinned by a vector for Bandersnatch only, so every other suite gets it from the trait, here.
    pub fn with_blinding_base(devices: &[i32]) -> Result<Self, GpuError> {
        let mut desc: ffi::vrfhip_suite_desc = unsafe { core::mem::zeroed() };
        check(unsafe { ffi::vrfhip_suite_desc_default(ffi::VRFHIP_SUITE_SECP256R1_SHA256_TAI, &mut desc) })?;
        let b = <P256 as pedersen::PedersenSuite>::BLINDING_BASE;
        b.x.serialize_uncompressed(&mut desc.blinding_base[..32]).expect("32-byte base field");
        b.y.serialize_uncompressed(&mut desc.blinding_base[32..]).expect("32-byte base field");
        let mut this = GpuBatchSec1 { ctxs: Vec::new() };
        for &dev in devices {
            let mut ctx: *mut ffi::vrfhip_ctx = core::ptr::null_mut();
            check(unsafe { ffi::vrfhip_ctx_create_desc(&desc, dev, &mut ctx) })?;
            this.ctxs.push(ctx);
            check(unsafe { ffi::vrfhip_ctx_set_flags(ctx, ffi::VRFHIP_FLAG_PROVE_POINTS_AFFINE) })?;
        }
        Ok(this)
    }
}

impl Drop for GpuBatchSec1 {
    fn drop(&mut self) {
        for &c in &self.ctxs {
            unsafe { ffi::vrfhip_ctx_destroy(c) };
        }
    }
}

/// `suites::bandersnatch_sw` ("Bandersnatch_SW_SHA-512_TAI"): Bandersnatch on its short-Weierstrass model, ArkworksCodec over
/// `SWAffine` -- 33-byte compressed points.  libvrfhip runs the group law on the twisted-Edwards model and crosses
/// `utils::te_sw_map` at the codec (csrc/bsw_core.cuh), so this suite has the twisted-Edwards suite's speed; it has no x || y
/// forms, which is why it does not go through `GpuBatch<S>`: points travel as `codec::point_encode` strings both ways (an
/// output costs the caller one `codec::point_decode`).  The suite's constants -- suite string, generator, blinding base --
/// are taken from the `Suite` impl through the descriptor, as for every suite.
pub struct GpuBatchSw {
    ctxs: Vec<*mut ffi::vrfhip_ctx>,
}

unsafe impl Send for GpuBatchSw {}
unsafe impl Sync for GpuBatchSw {}

type BSw = suites::bandersnatch_sw::BandersnatchSha512Tai;
const SWPT: usize = 33;

impl GpuBatchSw {
    pub fn new(devices: &[i32]) -> Result<Self, GpuError> {
        let mut desc = ffi::vrfhip_suite_desc {
            struct_size: core::mem::size_of::<ffi::vrfhip_suite_desc>() as u32,
            curve: ffi::VRFHIP_CURVE_BANDERSNATCH_SW,
            suite_id_len: BSw::SUITE_ID.len() as u32,
            suite_id: [0u8; 64],
            h2c_dst_len: 0,
            h2c_dst: [0u8; 128],
            generator: [0u8; 64],
            blinding_base: [0u8; 64],
            challenge_len: BSw::CHALLENGE_LEN as u32,
            flags: 0,
        };
        desc.suite_id[..BSw::SUITE_ID.len()].copy_from_slice(BSw::SUITE_ID);
        affine_xy::<BSw>(&BSw::generator(), &mut desc.generator); // the Weierstrass coordinates
        affine_xy::<BSw>(&<BSw as PedersenSuite>::BLINDING_BASE, &mut desc.blinding_base);
        let mut this = GpuBatchSw { ctxs: Vec::new() };
        for &dev in devices {
            let mut ctx: *mut ffi::vrfhip_ctx = core::ptr::null_mut();
            check(unsafe { ffi::vrfhip_ctx_create_desc(&desc, dev, &mut ctx) })?;
            debug_assert_eq!(unsafe { ffi::vrfhip_ctx_point_bytes(ctx) }, SWPT);
            this.ctxs.push(ctx);
        }
        Ok(this)
    }

    fn put_point(p: &AffinePoint<BSw>, out: &mut [u8]) {
        let mut v = Vec::with_capacity(SWPT);
        codec::point_encode_into::<BSw>(p, &mut v);
        out.copy_from_slice(&v);
    }

    /// `Secret::output` + `ietf::Prover::prove` per (secret, input) pair.
    pub fn ietf_prove(
        &self,
        secrets: &[Secret<BSw>],
        inputs: &[Input<BSw>],
        ad: &[u8],
    ) -> Result<Vec<Result<(Output<BSw>, ietf::Proof<BSw>), Error>>, GpuError> {
        let n = secrets.len();
        assert_eq!(n, inputs.len());
        let (mut sk, mut h) = (vec![0u8; n * 32], vec![0u8; n * SWPT]);
        for i in 0..n {
            scalar32::<BSw>(&secrets[i].scalar, &mut sk[i * 32..(i + 1) * 32]);
            Self::put_point(&inputs[i].0, &mut h[i * SWPT..(i + 1) * SWPT]);
        }
        let (mut gamma, mut c, mut s) = (vec![0u8; n * SWPT], vec![0u8; n * 32], vec![0u8; n * 32]);
        let mut status = vec![0u8; n];
        check(unsafe {
            ffi::vrfhip_ietf_prove_batch_multi(
                self.ctxs.as_ptr(), self.ctxs.len() as i32, n, sk.as_ptr(), core::ptr::null(), core::ptr::null(), 0,
                h.as_ptr(), ad.as_ptr(), core::ptr::null(), ad.len() as u32, gamma.as_mut_ptr(), c.as_mut_ptr(),
                s.as_mut_ptr(), core::ptr::null_mut(), core::ptr::null_mut(), status.as_mut_ptr(),
            )
        })?;
        sk.iter_mut().for_each(|b| *b = 0);
        Ok((0..n)
            .map(|i| {
                status_to_result(status[i])?;
                let out = Output::<BSw>::from(codec::point_decode::<BSw>(&gamma[i * SWPT..(i + 1) * SWPT])?);
                let proof = ietf::Proof::<BSw> {
                    c: codec::scalar_decode::<BSw>(&c[i * 32..(i + 1) * 32]),
                    s: codec::scalar_decode::<BSw>(&s[i * 32..(i + 1) * 32]),
                };
                Ok((out, proof))
            })
            .collect())
    }

    /// `ietf::Verifier::verify` per item.
    pub fn ietf_verify(
        &self,
        publics: &[Public<BSw>],
        inputs: &[Input<BSw>],
        outputs: &[Output<BSw>],
        ad: &[u8],
        proofs: &[ietf::Proof<BSw>],
    ) -> Result<Vec<Result<(), Error>>, GpuError> {
        let n = publics.len();
        assert!(n == inputs.len() && n == outputs.len() && n == proofs.len());
        let (mut pk, mut h, mut g) = (vec![0u8; n * SWPT], vec![0u8; n * SWPT], vec![0u8; n * SWPT]);
        let (mut c, mut s) = (vec![0u8; n * 32], vec![0u8; n * 32]);
        for i in 0..n {
            Self::put_point(&publics[i].0, &mut pk[i * SWPT..(i + 1) * SWPT]);
            Self::put_point(&inputs[i].0, &mut h[i * SWPT..(i + 1) * SWPT]);
            Self::put_point(&outputs[i].0, &mut g[i * SWPT..(i + 1) * SWPT]);
            scalar32::<BSw>(&proofs[i].c, &mut c[i * 32..(i + 1) * 32]);
            scalar32::<BSw>(&proofs[i].s, &mut s[i * 32..(i + 1) * 32]);
        }
        let mut status = vec![0u8; n];
        check(unsafe {
            ffi::vrfhip_ietf_verify_batch_multi(
                self.ctxs.as_ptr(), self.ctxs.len() as i32, n, pk.as_ptr(), h.as_ptr(), g.as_ptr(), c.as_ptr(), s.as_ptr(),
                ad.as_ptr(), core::ptr::null(), ad.len() as u32, status.as_mut_ptr(),
            )
        })?;
        Ok(status.into_iter().map(status_to_result).collect())
    }
}

impl Drop for GpuBatchSw {
    fn drop(&mut self) {
        for &c in &self.ctxs {
            unsafe { ffi::vrfhip_ctx_destroy(c) };
        }
    }
}
