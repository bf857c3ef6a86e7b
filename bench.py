#!/usr/bin/env python3
"""bench.py -- headline benchmark: batch 2^20 Bandersnatch IETF-ECVRF verification on MI355X.

A "step" is one pass of the verify hot path (vrfhip_ietf_verify_batch_dev: decode -> Straus ->
finish) over one batch of 2^20 synthetic proofs per GPU that already sit in HBM (SURVEY.md
section 8d: seed_i = u64_le(i), sk_i = from_seed, msg_i = SHA512("vrfhip-msg" || u64_le(i))[..32],
ad = "").  The proofs themselves are produced by the GPU prove path before the timed region.
N > 1: one process per GPU (torch.distributed / RCCL); every rank verifies its own 2^20 items
(weak scaling), the only collective is the gather of the status bytes.

Prints ONE JSON line (rank 0).  `roofline` prices the longest kernel (k_verify_straus<1>) against
HBM as the contract asks -- the path is VALU-integer bound, so `valu` gives the meaningful
ceiling: executed vector instructions per second against the measured v_mad_u64_u32 issue peak.
`cpu_baseline` times the plain-C oracle (oracle/c, kind "port") on this box's host cores.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LOG2_BATCH = 20
BYTES_PER_VERIFY = 161            # pk 32 + H 32 + Gamma 32 + c 32 + s 32 in, 1 status byte out
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: 8 TB/s
VALU_INT_PEAK = 256 * 4 * 16 * 2.4e9   # lanes/s: 256 CU x 4 SIMD x 16 int lanes/clk x 2.4 GHz (microbench: profiles/r01_instr_rate_microbench.jsonl)


def synth_msgs(lo, n):
    out = np.empty((n, 32), dtype=np.uint8)
    pre = b"vrfhip-msg"
    for k in range(n):
        out[k] = np.frombuffer(hashlib.sha512(pre + int(lo + k).to_bytes(8, "little")).digest()[:32], np.uint8)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log2-batch", type=int, default=LOG2_BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--secondary", action="store_true",
                    help="also time the other BASELINE.json configs (prove 2^16, Pedersen 2^20, MSM 2^20, "
                         "pairing 2^14, affine-input verify) and attach them under \"secondary\"")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from ark_ec_vrfs_amd import Context, _lib
    from ark_ec_vrfs_amd.sharding import gather_results

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # rehearsal hook: VRFHIP_BENCH_BACKEND=gloo lets several ranks share the GPUs that exist
    # (rank -> device modulo device_count, status gather through host memory)
    backend = os.environ.get("VRFHIP_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    n = 1 << args.log2_batch
    lo = rank * n                                  # weak scaling: rank g owns items [g*n, (g+1)*n)
    ctx = Context(local)
    lib = _lib.load()
    stream = torch.cuda.current_stream().cuda_stream

    # ---- synthetic inputs, produced on the GPU and left resident in HBM ----
    seeds = (torch.arange(n, dtype=torch.int64, device=dev) + lo).view(torch.uint8).reshape(n, 8)
    sk = torch.empty((n, 32), dtype=torch.uint8, device=dev)
    _lib.check(lib.vrfhip_secret_from_seed_batch_dev(ctx.handle, n, seeds.data_ptr(), 8, sk.data_ptr(), None, stream), "seed")
    msg = torch.from_numpy(synth_msgs(lo, n)).to(dev)
    mk = lambda: torch.empty((n, 32), dtype=torch.uint8, device=dev)
    gamma, c, s, pk, hh = mk(), mk(), mk(), mk(), mk()
    pst = torch.empty(n, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.ietf_prove_batch_dev(sk, msg, 32, gamma, c, s, pk, hh, pst)
    torch.cuda.synchronize()
    prove_s = time.perf_counter() - t0             # includes first-touch of the workspace
    t0 = time.perf_counter()
    ctx.ietf_prove_batch_dev(sk, msg, 32, gamma, c, s, pk, hh, pst)
    torch.cuda.synchronize()
    prove_s = min(prove_s, time.perf_counter() - t0)
    assert int(pst.sum()) == 0
    status = torch.full((n,), 255, dtype=torch.uint8, device=dev)

    def step():
        ctx.ietf_verify_batch_dev(pk, hh, gamma, c, s, status)
        if world > 1:
            if backend == "nccl":
                return gather_results(status, world * n, rank, world)   # RCCL: result gather only
            return gather_results(status.cpu(), world * n, rank, world).to(dev)
        return status

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ctx.profile(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        full = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ctx.profile(False)
    stage_ms, groups = ctx.profile_read()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    n_bad = int((full != 0).sum())
    assert n_bad == 0, f"{n_bad} synthetic proofs failed to verify"

    if rank == 0:
        total_items = world * n * args.steps
        value = total_items / elapsed
        straus_s = stage_ms[1] / max(groups, 1) / 1e3          # average launch duration of k_verify_straus<1>
        achieved_gbs = BYTES_PER_VERIFY * n / straus_s / 1e9
        # measured constants of the same command line (rocprofv3 --pmc passes, see profiles/)
        traffic = None
        valu = None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_k_verify_straus.json")
        if os.path.exists(pmc_path):
            pmc = json.load(open(pmc_path))
            if pmc.get("log2_batch") == args.log2_batch:
                traffic = pmc.get("hbm_bytes_per_launch")
                lanes = pmc.get("valu_lane_instructions_per_launch")
                if lanes:
                    valu = {"bound": "valu_int", "achieved": lanes / straus_s, "peak": VALU_INT_PEAK,
                            "unit": "lane-instr/s", "frac": lanes / straus_s / VALU_INT_PEAK,
                            "source": pmc.get("source")}
        out = {
            "metric": "Bandersnatch IETF-ECVRF verifies/sec, 2^%d batch per GPU" % args.log2_batch,
            "value": value, "unit": "verifies/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "IETF ECVRF verify, Bandersnatch_SHA-512_ELL2, batch 2^%d per GPU, compressed "
                                   "points (161 B/verify), ad=\"\" (BASELINE.json configs[2])" % args.log2_batch,
                       "global_batch": world * n, "parallelism": "items sharded x%d, result gather only" % world},
            "roofline": {"bound": "hbm", "kernel": "k_verify_straus<1> (V = s*H - c*Gamma, the longest kernel)", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic,
                         "avg_launch_ms": straus_s * 1e3, "launches_timed": groups},
            "valu": valu,
            "stage_ms_per_step": {"decode": stage_ms[0] / max(groups, 1), "straus_v": stage_ms[1] / max(groups, 1),
                                  "straus_u": stage_ms[2] / max(groups, 1), "finish": stage_ms[3] / max(groups, 1)},
            "proofs_per_sec": n / prove_s,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, pk, hh, gamma, c, s, status)
        if args.secondary and world == 1:
            out["secondary"] = secondary(ctx, lib, dev, sk, msg, pk, hh, gamma, c, s)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


def _time(fn, reps=3):
    import torch
    best = 1e30
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best


def secondary(ctx, lib, dev, sk, msg, pk, hh, gamma, c, s):
    """Other configs of BASELINE.json, same synthetic items, device-resident; best of 3."""
    import torch
    from ark_ec_vrfs_amd import _lib
    n = sk.shape[0]
    res = {}
    stream = torch.cuda.current_stream().cuda_stream
    mk = lambda m=n, w=32: torch.empty((m, w), dtype=torch.uint8, device=dev)
    # config 1: IETF prove at 2^16 (160 B/proof)
    m = min(n, 1 << 16)
    o1, o2, o3, o4, o5 = mk(m), mk(m), mk(m), mk(m), mk(m)
    st = torch.empty(m, dtype=torch.uint8, device=dev)
    t = _time(lambda: ctx.ietf_prove_batch_dev(sk[:m], msg[:m], 32, o1, o2, o3, o4, o5, st))
    res["ietf_prove_2^16"] = {"proofs_per_s": m / t, "ms": t * 1e3, "bytes_per_item": 160}
    # affine-input verify (257 B/verify)
    xy = [mk(n, 64) for _ in range(3)]
    vst = torch.empty(n, dtype=torch.uint8, device=dev)
    for src, dst in zip((pk, hh, gamma), xy):
        _lib.check(lib.vrfhip_point_validate_batch_dev(ctx.handle, n, src.data_ptr(), dst.data_ptr(), vst.data_ptr(), stream), "validate")
    t = _time(lambda: ctx.ietf_verify_batch_affine_dev(xy[0], xy[1], xy[2], c, s, vst))
    assert int(vst.sum()) == 0
    res["ietf_verify_affine_2^%d" % (n.bit_length() - 1)] = {"verifies_per_s": n / t, "ms": t * 1e3, "bytes_per_item": 257}
    # keyed verification: the same 2^n proofs re-made under 1024 keys whose combs stay resident in HBM (129 B/verify + 4 B index)
    try:
        nk = 1024
        kseeds = torch.arange(nk, dtype=torch.int64, device=dev).view(torch.uint8).reshape(nk, 8)
        ksk, kpk = mk(nk), mk(nk)
        _lib.check(lib.vrfhip_secret_from_seed_batch_dev(ctx.handle, nk, kseeds.data_ptr(), 8, ksk.data_ptr(), kpk.data_ptr(), stream), "seed")
        kidx = (torch.arange(n, device=dev) * 2654435761 % nk).to(torch.int32)
        g2, c2, s2, h2 = mk(), mk(), mk(), mk()
        ctx.ietf_prove_batch_dev(ksk[kidx.long()].contiguous(), msg, 32, g2, c2, s2, None, h2, vst)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ks, kst = ctx.keyset_create(kpk.cpu().numpy())
        tb = time.perf_counter() - t0
        t = _time(lambda: ctx.ietf_verify_batch_keyed_dev(ks, kidx, h2, g2, c2, s2, vst))
        assert int(vst.sum()) == 0 and int(kst.sum()) == 0
        res["ietf_verify_keyed_2^%d" % (n.bit_length() - 1)] = {"verifies_per_s": n / t, "ms": t * 1e3, "keys": nk,
                                                                "keyset_build_ms": tb * 1e3, "keyset_bytes": ks.bytes(),
                                                                "bytes_per_item": 133}
        ks.close()
    except Exception as e:
        res["ietf_verify_keyed"] = {"error": repr(e)}
    # config 3: Pedersen prove + verify (Bandersnatch; 288 / 225 B)
    g, pc, r, ok, ss, sb = (mk() for _ in range(6))
    pst = torch.empty(n, dtype=torch.uint8, device=dev)
    tp = _time(lambda: ctx.pedersen_prove_batch_dev(sk, msg, 32, g, pc, r, ok, ss, sb, None, None, pst))
    tv = _time(lambda: ctx.pedersen_verify_batch_dev(hh, g, pc, r, ok, ss, sb, pst))
    assert int(pst.sum()) == 0
    # batched verification: one MSM over 5n + 2 points (random linear combination), compressed and affine inputs
    flag = torch.empty(1, dtype=torch.uint8, device=dev)
    seed = os.urandom(32)
    tb = _time(lambda: ctx.pedersen_verify_batch_rlc_dev(hh, g, pc, r, ok, ss, sb, pst, flag, seed))
    assert int(flag[0]) == 0 and int(pst.sum()) == 0
    pxy = [mk(n, 64) for _ in range(5)]
    for src, dst in zip((hh, g, pc, r, ok), pxy):
        _lib.check(lib.vrfhip_point_validate_batch_dev(ctx.handle, n, src.data_ptr(), dst.data_ptr(), vst.data_ptr(), stream), "validate")
    ta = _time(lambda: ctx.pedersen_verify_batch_rlc_dev(*pxy, ss, sb, pst, flag, seed, affine=True))
    assert int(flag[0]) == 0 and int(pst.sum()) == 0
    res["pedersen_2^%d" % (n.bit_length() - 1)] = {"proofs_per_s": n / tp, "verifies_per_s": n / tv,
                                                   "batched_verifies_per_s": n / tb, "batched_affine_verifies_per_s": n / ta,
                                                   "bytes_per_proof": 288, "bytes_per_verify": 225,
                                                   "bytes_per_verify_affine": 385}
    # config 3 as BASELINE.json words it: Pedersen on JubJub (suite parity unpinned)
    try:
        from ark_ec_vrfs_amd import Context, JubJubSha512Tai
        cj = Context(dev.index or 0, suite=JubJubSha512Tai)
        skj = torch.empty((n, 32), dtype=torch.uint8, device=dev)
        seeds = torch.arange(n, dtype=torch.int64, device=dev).view(torch.uint8).reshape(n, 8)
        _lib.check(lib.vrfhip_secret_from_seed_batch_dev(cj.handle, n, seeds.data_ptr(), 8, skj.data_ptr(), None, stream), "seed")
        hj = mk()
        tp = _time(lambda: cj.pedersen_prove_batch_dev(skj, msg, 32, g, pc, r, ok, ss, sb, None, hj, pst))
        tv = _time(lambda: cj.pedersen_verify_batch_dev(hj, g, pc, r, ok, ss, sb, pst))
        assert int(pst.sum()) == 0
        tb = _time(lambda: cj.pedersen_verify_batch_rlc_dev(hj, g, pc, r, ok, ss, sb, pst, flag, seed))
        assert int(flag[0]) == 0 and int(pst.sum()) == 0
        res["pedersen_jubjub_2^%d" % (n.bit_length() - 1)] = {"proofs_per_s": n / tp, "verifies_per_s": n / tv,
                                                              "batched_verifies_per_s": n / tb,
                                                              "bytes_per_proof": 288, "bytes_per_verify": 225}
        cj.close()
    except Exception as e:
        res["pedersen_jubjub"] = {"error": repr(e)}
    # MSM over the public keys with the secrets as scalars (96 B/term)
    out = torch.empty(32, dtype=torch.uint8, device=dev); mst = torch.empty(1, dtype=torch.uint8, device=dev)
    t = _time(lambda: ctx.msm_dev(xy[0], sk, out, None, mst))
    assert int(mst[0]) == 0
    res["msm_2^%d" % (n.bit_length() - 1)] = {"points_per_s": n / t, "ms": t * 1e3, "bytes_per_item": 96}
    # config 4: pairing checks at 2^14 (577 B/check); 8 oracle-made items tiled
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from test_bls_pairing import kzg_like_items, pack
        g1, g2 = pack(kzg_like_items(8, seed=21))
        k = 1 << 14
        d1 = torch.from_numpy(np.tile(g1, (k // 8, 1)).copy()).to(dev)
        d2 = torch.from_numpy(np.tile(g2, (k // 8, 1)).copy()).to(dev)
        pstat = torch.empty(k, dtype=torch.uint8, device=dev)
        t = _time(lambda: ctx.pairing_check_batch_dev(d1, d2, pstat))
        assert int(pstat.sum()) == 0
        res["pairing_check_2^14"] = {"checks_per_s": k / t, "ms": t * 1e3, "bytes_per_item": 577}
        # the same size against one shared G2 pair (a KZG verifier's SRS): lines prepared once per context
        from test_bls_pairing import enc_g1, enc_g2, b as bo
        cc = 0x1234567FEDCBA987
        sh = np.frombuffer(enc_g2(bo.g2_mul(7, bo.G2)) + enc_g2(bo.g2_mul(7 * cc % bo.R, bo.G2)), np.uint8).copy()
        rows = [enc_g1(bo.g1_mul(a * cc % bo.R, bo.G1)) + enc_g1(bo.g1_neg(bo.g1_mul(a, bo.G1))) for a in range(1, 9)]
        s1 = np.frombuffer(b"".join(rows), np.uint8).reshape(-1, 192)
        d1 = torch.from_numpy(np.tile(s1, (k // 8, 1)).copy()).to(dev)
        dsh = torch.from_numpy(sh).to(dev)
        t = _time(lambda: ctx.pairing_check_batch_dev(d1, dsh, pstat, g2_shared=True))
        assert int(pstat.sum()) == 0
        res["pairing_check_shared_g2_2^14"] = {"checks_per_s": k / t, "ms": t * 1e3, "bytes_per_item": 193}
    except Exception as e:                                      # the headline number must not depend on this leg
        res["pairing_check_2^14"] = {"error": repr(e)}
    return res


def cpu_baseline(args, pk, hh, gamma, c, s, status):
    """The plain-C oracle (a port, not arkworks) on this box's host cores, bounded sample of the same batch."""
    from oracle import c_oracle as co
    # a one-GPU box's CPU share is 16 cores (more threads than that only oversubscribe the cgroup)
    cores = min(16, os.cpu_count() or 1, len(os.sched_getaffinity(0)))
    host = lambda t, m: t[:m].cpu().numpy()
    probe = 64 * cores
    t0 = time.perf_counter()
    st = co.ietf_verify_batch(host(pk, probe), host(hh, probe), host(gamma, probe), host(c, probe), host(s, probe), b"", threads=cores)
    rate = probe / (time.perf_counter() - t0)
    m = int(min(1 << 18, max(probe, rate * args.cpu_seconds)))
    t0 = time.perf_counter()
    st = co.ietf_verify_batch(host(pk, m), host(hh, m), host(gamma, m), host(c, m), host(s, m), b"", threads=cores)
    dt = time.perf_counter() - t0
    assert (st == status[:m].cpu().numpy()).all(), "GPU statuses differ from the CPU oracle on the sample"
    t0 = time.perf_counter()
    m1 = max(64, int(m / cores / 8))
    co.ietf_verify_batch(host(pk, m1), host(hh, m1), host(gamma, m1), host(c, m1), host(s, m1), b"", threads=1)
    dt1 = time.perf_counter() - t0
    return {"value": m / dt, "unit": "verifies/s", "cores": cores, "kind": "port",
            "sample": "first %d items of the same 2^%d batch, %.1f s wall on %d threads; statuses equal the GPU's" % (m, args.log2_batch, dt, cores),
            "single_core": m1 / dt1,
            "note": "CPU restatement in C (oracle/c/oracle_vrf.c), not arkworks: no Rust toolchain on this box"}


if __name__ == "__main__":
    main()
