//! Criterion benchmark mirroring upstream's per-suite benches (`ark-vrf/benches/*`: ietf prove / verify, pedersen prove /
//! verify per item on the CPU) with the batch sizes of /root/repo/BASELINE.json next to them: the same inputs through
//! `GpuBatch` (libvrfhip) and through a loop over the arkworks CPU API, so `cargo bench` prints the two side by side.
//!
//! SOURCE ONLY: like the rest of this crate it has never been compiled (no Rust toolchain in the build image).  The
//! measured counterpart that does run here is /root/repo/bench.py, which times the same operations through the C ABI and
//! the C restatement of the CPU path.
use ark_ec_vrfs_hip::{ietf, pedersen, suites, GpuBatch, GpuSuite, Input, Public, Secret};
use criterion::{criterion_group, criterion_main, BatchSize, BenchmarkId, Criterion, Throughput};

fn inputs<S: GpuSuite>(n: usize) -> (Vec<Secret<S>>, Vec<Public<S>>, Vec<Input<S>>) {
    // seed_i = u64_le(i), msg_i = SHA-512("vrfhip-msg" || u64_le(i))[..32]
    use sha2::{Digest, Sha512};
    let secrets: Vec<_> = (0..n as u64).map(|i| Secret::<S>::from_seed(&i.to_le_bytes())).collect();
    let publics = secrets.iter().map(|s| s.public()).collect();
    let inputs = (0..n as u64)
        .map(|i| {
            let mut h = Sha512::new();
            h.update(b"vrfhip-msg");
            h.update(i.to_le_bytes());
            Input::<S>::new(&h.finalize()[..32]).expect("hash-to-curve")
        })
        .collect();
    (secrets, publics, inputs)
}

fn bench_suite<S: GpuSuite>(c: &mut Criterion, name: &str, sizes: &[usize])
where
    Secret<S>: ietf::Prover<S> + pedersen::Prover<S>,
    Public<S>: ietf::Verifier<S> + pedersen::Verifier<S>,
{
    let gpu = GpuBatch::<S>::new(&[0]).expect("an MI355X and libvrfhip.so");
    for &n in sizes {
        let (sk, pk, inp) = inputs::<S>(n);
        let proofs: Vec<_> = gpu.ietf_prove(&sk, &inp, b"").unwrap().into_iter().map(|r| r.unwrap()).collect();
        let (outs, prfs): (Vec<_>, Vec<_>) = proofs.into_iter().unzip();
        let mut g = c.benchmark_group(format!("{name}/ietf"));
        g.throughput(Throughput::Elements(n as u64)).sample_size(10);
        g.bench_with_input(BenchmarkId::new("prove/gpu", n), &n, |b, _| b.iter(|| gpu.ietf_prove(&sk, &inp, b"").unwrap()));
        g.bench_with_input(BenchmarkId::new("verify/gpu", n), &n, |b, _| {
            b.iter(|| gpu.ietf_verify(&pk, &inp, &outs, b"", &prfs).unwrap())
        });
        // the CPU path on a bounded sample (upstream's own benches time one item)
        let m = n.min(256);
        g.throughput(Throughput::Elements(m as u64));
        g.bench_with_input(BenchmarkId::new("prove/cpu", m), &m, |b, _| {
            b.iter_batched(
                || (),
                |_| (0..m).map(|i| { use ietf::Prover; let o = sk[i].output(inp[i]); sk[i].prove(inp[i], o, b"") }).count(),
                BatchSize::SmallInput,
            )
        });
        g.bench_with_input(BenchmarkId::new("verify/cpu", m), &m, |b, _| {
            b.iter(|| (0..m).map(|i| { use ietf::Verifier; pk[i].verify(inp[i], outs[i], b"", &prfs[i]).is_ok() }).count())
        });
        g.finish();
        let ped = gpu.pedersen_prove(&sk, &inp, b"").unwrap();
        let (pouts, pprfs): (Vec<_>, Vec<_>) = ped.into_iter().map(|r| r.unwrap()).map(|(o, p, _)| (o, p)).unzip();
        let mut g = c.benchmark_group(format!("{name}/pedersen"));
        g.throughput(Throughput::Elements(n as u64)).sample_size(10);
        g.bench_with_input(BenchmarkId::new("prove/gpu", n), &n, |b, _| b.iter(|| gpu.pedersen_prove(&sk, &inp, b"").unwrap()));
        g.bench_with_input(BenchmarkId::new("verify/gpu", n), &n, |b, _| {
            b.iter(|| gpu.pedersen_verify(&inp, &pouts, b"", &pprfs, None).unwrap())
        });
        g.bench_with_input(BenchmarkId::new("verify_batched/gpu", n), &n, |b, _| {
            b.iter(|| gpu.pedersen_verify(&inp, &pouts, b"", &pprfs, Some(&[7u8; 32])).unwrap())
        });
        g.finish();
    }
}

fn benches(c: &mut Criterion) {
    // BASELINE.json: configs[1] 2^16 prove, configs[2] 2^20 verify (Bandersnatch); configs[3] 2^20 Pedersen (JubJub)
    bench_suite::<suites::bandersnatch::BandersnatchSha512Ell2>(c, "bandersnatch_sha-512_ell2", &[1 << 16, 1 << 20]);
    bench_suite::<suites::jubjub::JubJubSha512Tai>(c, "jubjub_sha-512_tai", &[1 << 20]);
    bench_suite::<suites::ed25519::Ed25519Sha512Tai>(c, "ed25519_sha-512_tai", &[1 << 20]);
    bench_suite::<suites::baby_jubjub::BabyJubJubSha512Tai>(c, "babyjubjub_sha-512_tai", &[1 << 20]);
}

criterion_group!(vrf_batches, benches);
criterion_main!(vrf_batches);
