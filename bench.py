#!/usr/bin/env python3
"""bench.py -- the BASELINE.json benchmark of libvrfhip on MI355X.

Headline (`metric`, `value`): batch 2^20 Bandersnatch IETF-ECVRF verification from the wire format
(BASELINE.json configs[2]).  A "step" is one pass of the verify hot path (vrfhip_ietf_verify_batch_dev: checked
decode -> Straus -> finish) over one batch of 2^20 synthetic proofs per GPU that already sit in HBM (SURVEY.md
section 8d: seed_i = u64_le(i), sk_i = from_seed, msg_i = SHA512("vrfhip-msg" || u64_le(i))[..32], ad = "").
The proofs themselves are produced by the GPU prove path before the timed region.

`python bench.py --gpus N` starts its own ranks when it is not already running under torch.distributed.run (the
parent touches no GPU: it only spawns `python -m torch.distributed.run ... bench.py` and relays its output).
N > 1: one process per GPU over RCCL; every rank works on its own 2^20 items (weak scaling); the only
collective is the gather of the result bytes.

The JSON line also carries, under "configs", the other BASELINE.json configurations measured in the same run
(IETF prove 2^16; Pedersen prove + verify on JubJub 2^20, per proof and batched; pairing checks 2^14, per item
and against a shared G2 pair), each with its own `roofline` (dominant kernel, HIP events on the launch stream)
and, at N = 1, a `cpu_baseline` leg (the plain-C / Python oracle under oracle/, kind "port", bounded sample).

`roofline` prices the longest kernel against HBM as the contract asks; the path is VALU-integer bound, so
`valu` gives the ceiling that matters: the fraction of the SIMDs' vector issue slots the kernel fills, from the
counters of the profiled run (SQ_INSTS_VALU, SQ_INSTS_VALU_INT64, GRBM_GUI_ACTIVE: profiles/pmc_kernels.json) at the
clock the chip actually held.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# HIP maps streams onto this many hardware queues (default 4); streams that share a queue run in order.  The
# "in flight" configuration below wants its four streams on four queues (measured: two streams on one queue = no overlap).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

LOG2_BATCH = 20
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: 8 TB/s
VALU_INT_PEAK = 256 * 4 * 16 * 2.4e9   # lanes/s (microbench: profiles/r01_instr_rate_microbench.jsonl)
# algorithmic bytes per unit
B_VERIFY, B_PROVE, B_PED_PROVE, B_PED_VERIFY, B_PAIRING, B_PAIRING_SHARED = 161, 160, 288, 225, 577, 193


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log2-batch", type=int, default=LOG2_BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="headline only (skip the other BASELINE.json configs)")
    ap.add_argument("--config-steps", type=int, default=5, help="timed steps per secondary config")
    ap.add_argument("--prevalidated", action="store_true",
                    help="headline without the subgroup check (inputs declared validated by the caller)")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="CPU time budget per cpu_baseline leg")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="what `value` measures at N > 1: weak = 2^log2_batch items per GPU (global batch grows with N); strong = "
                         "ONE global batch of 2^log2_batch items cut into N contiguous shards (BASELINE.json: '2^20 ... sharded "
                         "8x').  The other mode is measured in the same run and reported under `scaling_modes`.")
    ap.add_argument("--only", default="", help="comma-separated secondary legs to run (default: all)")
    return ap.parse_args()


def self_launch(args):
    """--gpus N > 1 outside torch.distributed.run: spawn the ranks.  Nothing here touches a GPU (no torch import)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    if os.environ.get("VRFHIP_BENCH_DRYRUN"):            # tests/test_bench_launch.py: show the launch, start nothing
        print(json.dumps({"launch": cmd, "HSA_ENABLE_IPC_MODE_LEGACY": env["HSA_ENABLE_IPC_MODE_LEGACY"]}))
        return 0
    return subprocess.run(cmd, env=env).returncode


def synth_msgs(lo, n):
    out = np.empty((n, 32), dtype=np.uint8)
    pre = b"vrfhip-msg"
    for k in range(n):
        out[k] = np.frombuffer(hashlib.sha512(pre + int(lo + k).to_bytes(8, "little")).digest()[:32], np.uint8)
    return out


def slim_results(res):
    """What a rank tells the others about a leg: per config its rate and step time, or its error."""
    return {k: ({"error": v["error"]} if "error" in v else {"value": v["value"], "ms_per_step": v["ms_per_step"]})
            for k, v in res.items()}

FINAL_LINE_MAX = 4096             # the driver keeps an 8 KB stdout tail: the last line must fit it whole, with margin


def _round_floats(x, sig=6):
    if isinstance(x, float):
        return float("%.*g" % (sig, x))
    if isinstance(x, dict):
        return {k: _round_floats(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_round_floats(v, sig) for v in x]
    return x


def assemble_final_line(out, configs_file=None):
    """The ONE line the driver parses: headline metric + roofline + valu + cpu_baseline, <= FINAL_LINE_MAX bytes whatever
    the number of secondary legs.  `out` is the full record (with "configs"); the legs themselves go to an earlier stdout
    line and to `configs_file`.  Pure: no I/O, so tests/test_bench_launch.py can size-check it on CPU."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "scaling_modes",
            "vs_baseline", "dtype", "data", "config", "roofline", "valu", "pmc_stale", "stage_ms_per_step", "proofs_per_sec",
            "cpu_baseline")
    line = {k: out[k] for k in keep if k in out}
    if "prevalidated" in out:
        line["prevalidated"] = {k: out["prevalidated"][k] for k in ("value", "ms_per_step") if k in out["prevalidated"]}
    if line.get("cpu_baseline"):
        cb = dict(line["cpu_baseline"])
        for k in ("sample", "note"):
            if k in cb and len(cb[k]) > 160:
                cb[k] = cb[k][:157] + "..."
        line["cpu_baseline"] = cb
    line["configs_file"] = configs_file
    line = _round_floats(line)
    if line.get("roofline") and line["roofline"].get("peak"):          # frac stays exactly achieved / peak after rounding
        line["roofline"]["frac"] = line["roofline"]["achieved"] / line["roofline"]["peak"]
    cfgs = out.get("configs") or {}
    # one number per secondary leg, as many as fit (full records: configs_file and the earlier stdout line)
    summary = {}
    for k, v in cfgs.items():
        summary[k] = "error" if "error" in v else [float("%.4g" % v["value"]), float("%.4g" % v["ms_per_step"])]
    line["configs_summary"] = summary
    while summary and len(json.dumps(line)) > FINAL_LINE_MAX:
        summary.pop(next(reversed(summary)))
    if cfgs and len(summary) < len(cfgs):
        line["configs_summary_truncated"] = len(cfgs) - len(summary)
    if len(json.dumps(line)) > FINAL_LINE_MAX:      # cannot happen with the fields above; never print an unparseable tail
        for k in ("configs_summary", "prevalidated", "stage_ms_per_step", "scaling_modes"):
            line.pop(k, None)
            if len(json.dumps(line)) <= FINAL_LINE_MAX:
                break
    return line


def print_result(out, configs, world):
    """Secondary legs: full records on an EARLIER stdout line and in a file; the LAST line is the headline alone and stays
    under FINAL_LINE_MAX bytes (a 25 KB line overflowed the driver's stdout tail)."""
    configs_file = None
    if configs:
        configs_file = os.environ.get("VRFHIP_BENCH_CONFIGS_FILE") or os.path.join(ROOT, "bench_configs_n%d.json" % world)
        try:
            with open(configs_file, "w") as f:
                json.dump({"headline": {k: v for k, v in out.items() if k != "configs"}, "configs": configs}, f, indent=1)
            if configs_file.startswith(ROOT + os.sep):
                configs_file = os.path.relpath(configs_file, ROOT)
        except OSError as e:
            configs_file = "not written: %r" % (e,)
        print(json.dumps({"bench_configs": configs}), flush=True)
    print(json.dumps(assemble_final_line(out, configs_file)), flush=True)


def merge_rank_results(own, allr):
    """own: this rank's full results of a leg; allr: slim_results of every rank.  A config's rate is set by its slowest
    rank (each rank computed world x units / its own elapsed, so the job's figure is the minimum); an error or a missing
    config on any rank is reported instead of a number."""
    out = dict(own)
    for k in list(out):
        errs = [r[k]["error"] for r in allr if k in r and "error" in r[k]] + ["missing on a rank" for r in allr if k not in r]
        if errs:
            out[k] = {"error": errs[0]}
        elif "value" in out[k]:
            out[k] = dict(out[k], value=min(r[k]["value"] for r in allr), ms_per_step=max(r[k]["ms_per_step"] for r in allr))
    for r in allr:
        for k in r:
            if k not in out:
                out[k] = {"error": r[k].get("error", "missing on rank 0")}
    return out


class Dist:
    """Rank bookkeeping + the two collectives the bench needs (result gather, max of elapsed)."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={self.world}")
        # rehearsal hook: VRFHIP_BENCH_BACKEND=gloo lets several ranks share the GPUs that exist
        # (rank -> device modulo device_count, result gather through host memory)
        self.backend = os.environ.get("VRFHIP_BENCH_BACKEND", "nccl")
        ndev = torch.cuda.device_count()
        if ndev == 0:
            raise SystemExit("no GPU visible: libvrfhip has no CPU path")
        if self.backend == "nccl" and self.world > ndev:
            raise SystemExit(f"--gpus {self.world} but only {ndev} device(s) visible (VRFHIP_BENCH_BACKEND=gloo shares them)")
        self.local = local % ndev
        torch.cuda.set_device(self.local)
        self.dev = torch.device("cuda", self.local)
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.backend == "nccl":
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.dev)
            else:
                dist.init_process_group(self.backend, rank=self.rank, world_size=self.world)

    local_legs = False      # set while a secondary config runs at world > 1 (see timed)

    def merge_leg_results(self, res):
        """Every rank's {config: result} of one leg -> rank 0's view of the whole job (merge_rank_results).  One
        collective, reached by every rank whether its leg failed or not."""
        if self.world == 1:
            return res
        allr = [None] * self.world
        self.dist.all_gather_object(allr, slim_results(res))
        return merge_rank_results(res, allr)

    def barrier(self):
        self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def gather(self, t, n_per_rank, n_global=None):
        """n_global: items of the whole job when the ranks hold the contiguous shards of ONE batch (strong scaling:
        sharding.shard_range sizes); default world * n_per_rank (weak scaling: every rank its own batch)."""
        if self.world == 1:
            return t
        from ark_ec_vrfs_amd.sharding import gather_results
        total = self.world * n_per_rank if n_global is None else n_global
        if self.backend == "nccl":
            return gather_results(t, total, self.rank, self.world)   # RCCL: result gather only
        return gather_results(t.cpu(), total, self.rank, self.world).to(self.dev)

    def max_elapsed(self, elapsed):
        if self.world == 1:
            return elapsed
        t = self.torch.tensor([elapsed], dtype=self.torch.float64, device=self.dev if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.world > 1:
            self.dist.barrier()
            self.dist.destroy_process_group()


def timed(D, fn, steps, warmup, gather_t=None, n_per_rank=0, n_global=None):
    """warmup untimed calls, then exactly `steps` calls between barrier + synchronize; max over ranks.
    Inside a secondary leg of a multi-rank run (D.local_legs) the bracket is this rank's own synchronize and nothing
    is gathered: a leg that fails on one rank must not leave the others waiting at a barrier -- the ranks' results meet
    once, after the leg (merge_leg_results)."""
    if D.local_legs:
        for _ in range(warmup):
            fn()
        D.torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        D.torch.cuda.synchronize()
        return time.perf_counter() - t0, gather_t
    out = None
    for _ in range(warmup):
        fn()
        if gather_t is not None:
            D.gather(gather_t, n_per_rank, n_global)
    D.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
        if gather_t is not None:
            out = D.gather(gather_t, n_per_rank, n_global)
    D.barrier()
    return D.max_elapsed(time.perf_counter() - t0), out


_PMC = None


def pmc_table():
    """profiles/pmc_kernels.json if it was measured on THESE kernel sources, else {} (and why).  The file carries the
    sha256 of csrc/ at profiling time ("_stamp": tools/summarize_profiles3.py); a bench run on edited kernels reports
    `valu` / `roofline.traffic` as null instead of replaying counters of other code."""
    global _PMC
    if _PMC is None:
        path = os.environ.get("VRFHIP_BENCH_PMC_FILE") or os.path.join(ROOT, "profiles", "pmc_kernels.json")
        tab, why = {}, "no profiles/pmc_kernels.json"
        if os.path.exists(path):
            from ark_ec_vrfs_amd._lib import source_stamp
            raw = json.load(open(path))
            st = raw.get("_stamp") or {}
            if st.get("source_sha256") == source_stamp():
                tab, why = raw, None
            else:
                why = "profiles/pmc_kernels.json was measured on other kernel sources (stamp %s..., now %s...): re-profile" % (
                    str(st.get("source_sha256"))[:12], source_stamp()[:12])
        _PMC = (tab, why)
    return _PMC


def pmc_for(name, log2_batch):
    """HBM traffic and executed vector instructions per launch of a config's dominant kernel, measured with
    rocprofv3 --pmc on the same command line (profiles/pmc_kernels.json; tools/summarize_profiles3.py)."""
    e = pmc_table()[0].get(name)
    if e and e.get("log2_batch") == log2_batch:
        return e
    return None


def roofline(kernel, bytes_per_unit, units, kernel_ms, launches, pmc):
    sec = kernel_ms / 1e3
    achieved = bytes_per_unit * units / sec / 1e9 if sec > 0 else 0.0
    r = {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc.get("hbm_bytes_per_launch") if pmc else None,
         "avg_launch_ms": kernel_ms, "launches_timed": launches, "algorithmic_bytes_per_launch": bytes_per_unit * units}
    v = None
    if pmc and pmc.get("valu_lane_instructions_per_launch") and sec > 0:
        # The ceiling that matters for this integer path: how many of the SIMDs' VALU issue slots the kernel fills.  From
        # the counters of the profiled run (tools/profile_round3.sh): clock = GRBM_GUI_ACTIVE / 8 XCDs / duration,
        # issue_slot_frac = SQ_INSTS_VALU x 4 cycles / (cycles x 1024 SIMDs); issue_cycle_frac prices the non-64-bit
        # instructions at 2 cycles (the lower bound).  `achieved` scales the profiled instruction count to THIS run's
        # kernel time; no nominal-clock peak is assumed any more.
        lanes = pmc["valu_lane_instructions_per_launch"]
        v = {"bound": "valu_issue", "achieved": lanes / sec, "unit": "lane-instr/s",
             "clock_ghz_observed": pmc.get("clock_ghz_observed"), "issue_slot_frac": pmc.get("issue_slot_frac"),
             "issue_cycle_frac": pmc.get("issue_cycle_frac"), "frac": pmc.get("issue_slot_frac"),
             "scratch_bytes_per_lane": pmc.get("scratch_bytes_per_lane"), "source": pmc.get("source")}
    return r, v


def stage_avg(ctx):
    ms, groups = ctx.profile_read()
    g = max(groups, 1)
    return [m / g for m in ms], groups


# ------------------------------------------------------------------------------------------- cpu legs
def cpu_cores():
    # a one-GPU box's CPU share is 16 cores (more threads than that only oversubscribe the cgroup)
    return min(16, os.cpu_count() or 1, len(os.sched_getaffinity(0)))


def cpu_leg(fn, probe, budget_s, cap, unit, what, src="oracle/c/oracle_vrf.c"):
    """fn(m) runs the oracle on the first m items.  Probe, then one bounded run of about budget_s seconds."""
    cores = cpu_cores()
    t0 = time.perf_counter()
    fn(probe)
    rate = probe / (time.perf_counter() - t0)
    m = int(min(cap, max(probe, rate * budget_s)))
    t0 = time.perf_counter()
    fn(m)
    dt = time.perf_counter() - t0
    return {"value": m / dt, "unit": unit, "cores": cores, "kind": "port",
            "sample": "first %d items of the same batch, %.1f s wall on %d threads; %s" % (m, dt, cores, what),
            "note": "CPU restatement in C (%s), not arkworks: no Rust toolchain on this box" % src}


BLS_P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab


def _g1_multiples(items, reps):
    """items: (m, 192) uint8, two affine BLS12-381 G1 points each (x || y, 48-byte little-endian).  Returns (m * reps, 192):
    row m r + t holds (r + 1) x both points of item t -- running sums P, 2P, 3P, ... in affine coordinates (synthetic-input
    generation on the host: Python integers, no library code)."""
    p = BLS_P
    m = items.shape[0]
    out = np.empty((m * reps, 192), np.uint8)

    def dec(b):
        return int.from_bytes(b[:48].tobytes(), "little"), int.from_bytes(b[48:96].tobytes(), "little")

    def add(P, Q):
        (x1, y1), (x2, y2) = P, Q
        lam = (3 * x1 * x1) * pow(2 * y1, -1, p) % p if P == Q else (y2 - y1) * pow(x2 - x1, -1, p) % p
        x3 = (lam * lam - x1 - x2) % p
        return x3, (lam * (x1 - x3) - y1) % p

    for t in range(m):
        for half in range(2):
            base = dec(items[t, 96 * half:96 * half + 96])
            cur = base
            for r in range(reps):
                if r:
                    cur = add(cur, base)
                out[m * r + t, 96 * half:96 * half + 48] = np.frombuffer(cur[0].to_bytes(48, "little"), np.uint8)
                out[m * r + t, 96 * half + 48:96 * half + 96] = np.frombuffer(cur[1].to_bytes(48, "little"), np.uint8)
    return out


# ------------------------------------------------------------------------------------------- configs
def cfg_ietf_prove(D, args, ctx, sk, msg, want_cpu):
    """BASELINE.json configs[1]: batch 2^16 IETF prove, Bandersnatch (Elligator 2 + fixed-base + GLV)."""
    torch = D.torch
    m = min(sk.shape[0], 1 << 16)
    mk = lambda: torch.empty((m, 32), dtype=torch.uint8, device=D.dev)
    g, c, s, pk, hh = mk(), mk(), mk(), mk(), mk()
    st = torch.empty(m, dtype=torch.uint8, device=D.dev)
    fn = lambda: ctx.ietf_prove_batch_dev(sk[:m], msg[:m], 32, g, c, s, pk, hh, st)
    fn(); torch.cuda.synchronize()
    ctx.profile(True)
    el, _ = timed(D, fn, args.config_steps, 1)
    ctx.profile(False)
    ms, groups = stage_avg(ctx)
    assert int(st.sum()) == 0
    r, v = roofline("k_prove_mul (sk*H, sk*G, k*H, k*G: 2 lanes per proof)", B_PROVE, m, ms[1], groups,
                    pmc_for("ietf_prove", 16))
    out = {"workload": "IETF ECVRF prove (Input::new + Secret::output + prove), Bandersnatch_SHA-512_ELL2, batch 2^16 per GPU "
                       "(BASELINE.json configs[1])",
           "value": D.world * m * args.config_steps / el, "unit": "proofs/s", "ms_per_step": el / args.config_steps * 1e3,
           "bytes_per_unit": B_PROVE, "roofline": r, "valu": v,
           "stage_ms_per_step": {"prepare": ms[0], "mul": ms[1], "finish": ms[3]}}
    if want_cpu:
        from oracle import c_oracle as co
        skh, msgh = sk[:m].cpu().numpy(), msg[:m].cpu().numpy()
        gh, ch, sh = g.cpu().numpy(), c.cpu().numpy(), s.cpu().numpy()

        def leg(k):
            ref = co.ietf_prove_batch(skh[:k], msgs=msgh[:k], ad=b"", threads=cpu_cores())
            assert (ref["output"] == gh[:k]).all() and (ref["c"] == ch[:k]).all() and (ref["s"] == sh[:k]).all(), \
                "GPU proofs differ from the CPU oracle on the sample"
        out["cpu_baseline"] = cpu_leg(leg, 32 * cpu_cores(), args.cpu_seconds, m, "proofs/s", "proof bytes equal the GPU's")
    return out


def cfg_ietf_keyed(D, args, ctx, msg, lo):
    """The headline's workload for a verifier that knows its signers (`Public` keys of a validator set): 2^20 proofs
    under 1024 keys whose validated points and fixed-base tables stay resident in HBM (vrfhip_keyset_create)."""
    torch = D.torch
    from ark_ec_vrfs_amd import _lib
    lib = _lib.load()
    n = msg.shape[0]
    nk = 1024
    stream = torch.cuda.current_stream().cuda_stream
    mk = lambda m=n: torch.empty((m, 32), dtype=torch.uint8, device=D.dev)
    kseeds = (torch.arange(nk, dtype=torch.int64, device=D.dev) + (1 << 40)).view(torch.uint8).reshape(nk, 8)
    ksk, kpk = mk(nk), mk(nk)
    _lib.check(lib.vrfhip_secret_from_seed_batch_dev(ctx.handle, nk, kseeds.data_ptr(), 8, ksk.data_ptr(), kpk.data_ptr(), stream), "seed")
    kidx = ((torch.arange(n, device=D.dev) + lo) * 2654435761 % nk).to(torch.int32)
    g2, c2, s2, h2 = mk(), mk(), mk(), mk()
    vst = torch.empty(n, dtype=torch.uint8, device=D.dev)
    ctx.ietf_prove_batch_dev(ksk[kidx.long()].contiguous(), msg, 32, g2, c2, s2, None, h2, vst)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ks, kst = ctx.keyset_create(kpk.cpu().numpy())
    build_ms = (time.perf_counter() - t0) * 1e3
    try:
        fn = lambda: ctx.ietf_verify_batch_keyed_dev(ks, kidx, h2, g2, c2, s2, vst)
        fn(); torch.cuda.synchronize()
        ctx.profile(True)
        el, _ = timed(D, fn, args.config_steps, 1, gather_t=vst, n_per_rank=n)
        ctx.profile(False)
        ms, groups = stage_avg(ctx)
        assert int(vst.sum()) == 0 and int(kst.sum()) == 0
        lg = n.bit_length() - 1
        rf, v = roofline("k_verify_straus<1> (V = s*H - c*Gamma)", 133, n, ms[1], groups, pmc_for("ietf_verify", lg))
        return {"workload": "IETF ECVRF verify against a resident key set (1024 validated keys, 881 KB comb each), batch 2^%d per "
                            "GPU: 4-byte key index + H, Gamma, c, s per proof; H and Gamma checked-decoded" % lg,
                "value": D.world * n * args.config_steps / el, "unit": "verifies/s", "ms_per_step": el / args.config_steps * 1e3,
                "bytes_per_unit": 133, "roofline": rf, "valu": v, "keyset_build_ms": build_ms, "keyset_bytes": ks.bytes(),
                "stage_ms_per_step": {"decode": ms[0], "straus_v": ms[1], "comb_u": ms[2], "finish": ms[3]}}
    finally:
        ks.close()


def cfg_pedersen_jubjub(D, args, msg, lo, want_cpu):
    """BASELINE.json configs[3]: Pedersen VRF batch 2^20 prove + verify on JubJub (sharded: every rank its own batch)."""
    torch = D.torch
    from ark_ec_vrfs_amd import Context, JubJubSha512Tai, _lib
    lib = _lib.load()
    n = msg.shape[0]
    cj = Context(D.local, suite=JubJubSha512Tai, test_blinding_base=True)
    stream = torch.cuda.current_stream().cuda_stream
    mk = lambda: torch.empty((n, 32), dtype=torch.uint8, device=D.dev)
    seeds = (torch.arange(n, dtype=torch.int64, device=D.dev) + lo).view(torch.uint8).reshape(n, 8)
    skj = mk()
    _lib.check(lib.vrfhip_secret_from_seed_batch_dev(cj.handle, n, seeds.data_ptr(), 8, skj.data_ptr(), None, stream), "seed")
    g, pc, r, ok, ss, sb, hj = (mk() for _ in range(7))
    pst = torch.empty(n, dtype=torch.uint8, device=D.dev)
    flag = torch.empty(1, dtype=torch.uint8, device=D.dev)
    seed = os.urandom(32)
    res = {}
    lg = n.bit_length() - 1
    # prove
    fn = lambda: cj.pedersen_prove_batch_dev(skj, msg, 32, g, pc, r, ok, ss, sb, None, hj, pst)
    fn(); torch.cuda.synchronize()
    cj.profile(True)
    el, _ = timed(D, fn, args.config_steps, 1)
    cj.profile(False)
    ms, groups = stage_avg(cj)
    assert int(pst.sum()) == 0
    rf, v = roofline("k_prove_mul (253-bit Straus: JubJub has no endomorphism)", B_PED_PROVE, n, ms[1], groups,
                     pmc_for("pedersen_prove_jubjub", lg))
    res["pedersen_prove_jubjub"] = {
        "workload": "Pedersen VRF prove, JubJub_SHA-512_TAI, batch 2^%d per GPU (BASELINE.json configs[3], prove half); non-upstream "
                    "blinding base (vrfhip_test_blinding_base: upstream's JubJub constant is not pinned)" % lg,
        "value": D.world * n * args.config_steps / el, "unit": "proofs/s", "ms_per_step": el / args.config_steps * 1e3,
        "bytes_per_unit": B_PED_PROVE, "roofline": rf, "valu": v,
        "stage_ms_per_step": {"tai_find+prepare": ms[0], "mul": ms[1], "finish": ms[3]}}
    # per-proof verify
    fn = lambda: cj.pedersen_verify_batch_dev(hj, g, pc, r, ok, ss, sb, pst)
    fn(); torch.cuda.synchronize()
    cj.profile(True)
    el, _ = timed(D, fn, args.config_steps, 1, gather_t=pst, n_per_rank=n)
    cj.profile(False)
    ms, groups = stage_avg(cj)
    assert int(pst.sum()) == 0
    rf, v = roofline("k_ped_verify_straus<0> (s*H - c*Gamma)", B_PED_VERIFY, n, ms[1], groups,
                     pmc_for("pedersen_verify_jubjub", lg))
    res["pedersen_verify_jubjub"] = {
        "workload": "Pedersen VRF verify per proof, JubJub_SHA-512_TAI, batch 2^%d per GPU (BASELINE.json configs[3], verify half)" % lg,
        "value": D.world * n * args.config_steps / el, "unit": "verifies/s", "ms_per_step": el / args.config_steps * 1e3,
        "bytes_per_unit": B_PED_VERIFY, "roofline": rf, "valu": v,
        "stage_ms_per_step": {"decode": ms[0], "straus_a": ms[1], "straus_b": ms[2], "finish": ms[3]}}
    # batched verify: one MSM over 5n + 2 points (random linear combination)
    fn = lambda: cj.pedersen_verify_batch_rlc_dev(hj, g, pc, r, ok, ss, sb, pst, flag, seed)
    fn(); torch.cuda.synchronize()
    cj.profile(True)
    el, _ = timed(D, fn, args.config_steps, 1, gather_t=pst, n_per_rank=n)
    cj.profile(False)
    ms, groups = stage_avg(cj)
    assert int(flag[0]) == 0 and int(pst.sum()) == 0
    rf, v = roofline("k_rlc_decode (five decompressions per proof, digits and points into the MSM layout)", B_PED_VERIFY, n,
                     ms[0], groups, pmc_for("pedersen_rlc_jubjub", lg))
    res["pedersen_verify_batched_jubjub"] = {
        "workload": "Pedersen VRF verify, whole batch by one Pippenger MSM (random linear combination), JubJub, 2^%d per GPU" % lg,
        "value": D.world * n * args.config_steps / el, "unit": "verifies/s", "ms_per_step": el / args.config_steps * 1e3,
        "bytes_per_unit": B_PED_VERIFY, "roofline": rf, "valu": v,
        "stage_ms_per_step": {"decode": ms[0], "msm_buckets": ms[1], "msm_final": ms[2] + ms[3]}}
    if want_cpu:
        from oracle import c_oracle as co
        host = lambda t: t.cpu().numpy()
        skh, msgh = host(skj[:1 << 16]), host(msg[:1 << 16])
        ref_names = (("output", g), ("pk_com", pc), ("r", r), ("ok", ok), ("s", ss), ("sb", sb))
        refs = {k: host(t[:1 << 16]) for k, t in ref_names}
        hjh = host(hj[:1 << 16])
        co.set_suite(2)
        try:
            def leg_p(k):
                ref = co.pedersen_prove_batch(skh[:k], msgs=msgh[:k], ad=b"", threads=cpu_cores())
                for name in refs:
                    assert (ref[name] == refs[name][:k]).all(), "GPU Pedersen proofs differ from the CPU oracle: " + name
            res["pedersen_prove_jubjub"]["cpu_baseline"] = cpu_leg(leg_p, 16 * cpu_cores(), args.cpu_seconds, 1 << 16,
                                                                   "proofs/s", "proof bytes equal the GPU's")

            def leg_v(k):
                st = co.pedersen_verify_batch(hjh[:k], refs["output"][:k], refs["pk_com"][:k], refs["r"][:k], refs["ok"][:k],
                                              refs["s"][:k], refs["sb"][:k], b"", threads=cpu_cores())
                assert not st.any(), "CPU oracle rejects GPU-made proofs"
            leg = cpu_leg(leg_v, 16 * cpu_cores(), args.cpu_seconds, 1 << 16, "verifies/s", "statuses equal the GPU's")
            res["pedersen_verify_jubjub"]["cpu_baseline"] = leg
            res["pedersen_verify_batched_jubjub"]["cpu_baseline"] = dict(leg, sample=leg["sample"] + " (per-proof verifier)")
        finally:
            co.set_suite(1)
    cj.close()
    return res


def cfg_suite_ietf(D, args, tag, suite_cls, oracle_suite, title, lo, want_cpu):
    """IETF prove + verify for a suite of another base field (Ed25519 over 2^255 - 19, Baby-JubJub over
    BN254 Fr), batch 2^20 per GPU, wire format, checked decode -- the headline's operation on that curve."""
    torch = D.torch
    from ark_ec_vrfs_amd import Context, _lib
    lib = _lib.load()
    n = 1 << args.log2_batch
    cx = Context(D.local, suite=suite_cls, test_blinding_base=True)
    res = {}
    try:
        stream = torch.cuda.current_stream().cuda_stream
        mk = lambda: torch.empty((n, 32), dtype=torch.uint8, device=D.dev)
        seeds = (torch.arange(n, dtype=torch.int64, device=D.dev) + lo).view(torch.uint8).reshape(n, 8)
        sk = mk()
        _lib.check(lib.vrfhip_secret_from_seed_batch_dev(cx.handle, n, seeds.data_ptr(), 8, sk.data_ptr(), None, stream), "seed")
        gen = torch.Generator(device=D.dev); gen.manual_seed(1234 + lo)
        msg = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=D.dev, generator=gen)
        pw = cx.point_bytes()                      # 32; 33 for secp256r1 (Sec1)
        mkp = lambda: torch.empty((n, pw), dtype=torch.uint8, device=D.dev)
        g, c, s_, pk, hh = mkp(), mk(), mk(), mkp(), mkp()
        st = torch.empty(n, dtype=torch.uint8, device=D.dev)
        lg = args.log2_batch
        sw = tag == "secp256r1"
        bsw = tag == "bandersnatch_sw"           # the Edwards arithmetic behind arkworks' 33-byte short-Weierstrass codec
        b_prove, b_verify = B_PROVE + (pw - 32), B_VERIFY + 3 * (pw - 32)      # one / three points of pw bytes instead of 32
        fn = lambda: cx.ietf_prove_batch_dev(sk, msg, 32, g, c, s_, pk, hh, st)
        fn(); torch.cuda.synchronize()
        cx.profile(True)
        el, _ = timed(D, fn, args.config_steps, 1)
        cx.profile(False)
        ms, groups = stage_avg(cx)
        assert int(st.sum()) == 0
        rf, v = roofline("k_p256_prove_mul (sk*G, sk*H, k*G, k*H: 4 lanes per proof, grid.y = 4)" if sw else
                         "k_prove_mul (sk*H, sk*G, k*H, k*G: 2 lanes per proof; the twisted-Edwards suite's kernel)" if bsw else
                         "k_prove_mul (sk*H, sk*G, k*H, k*G: 2 lanes per proof)", b_prove, n, ms[1], groups, pmc_for("ietf_prove_" + tag, lg))
        res["ietf_prove_" + tag] = {
            "workload": "IETF ECVRF prove, %s, batch 2^%d per GPU (SURVEY.md section 8 f4)" % (title, lg),
            "value": D.world * n * args.config_steps / el, "unit": "proofs/s", "ms_per_step": el / args.config_steps * 1e3,
            "bytes_per_unit": b_prove, "roofline": rf, "valu": v,
            "stage_ms_per_step": {"tai_find+prepare": ms[0], "mul": ms[1], "finish": ms[3]}}
        fn = lambda: cx.ietf_verify_batch_dev(pk, hh, g, c, s_, st)
        fn(); torch.cuda.synchronize()
        cx.profile(True)
        el, _ = timed(D, fn, args.config_steps, 1, gather_t=st, n_per_rank=n)
        cx.profile(False)
        ms, groups = stage_avg(cx)
        assert int(st.sum()) == 0
        heavy = max(range(4), key=lambda k: ms[k])
        kname = ("k_verify_decode (3 decompressions + subgroup tests + tables)", "k_verify_straus<1> (V = s*H - c*Gamma)",
                 "k_verify_straus<0> (U = s*G - c*Y)", "k_verify_finish")[heavy]
        if bsw:
            kname = ("k_bsw_verify_decode (3 decompressions + te_sw_map + subgroup tests + tables)", "k_bsw_verify_straus<1> (V = s*H - c*Gamma)",
                     "k_bsw_verify_straus<0> (U = s*G - c*Y)", "k_bsw_verify_finish")[heavy]
        if sw:
            kname = ("k_p256_verify_decode", "k_p256_verify_mul<1> (V = s*H - c*Gamma)", "k_p256_verify_mul<0> (U = s*G - c*Y)",
                     "k_p256_verify_finish")[heavy]
        rf, v = roofline(kname, b_verify, n, ms[heavy], groups, pmc_for("ietf_verify_" + tag, lg))
        res["ietf_verify_" + tag] = {
            "workload": "IETF ECVRF verify, %s, batch 2^%d per GPU, compressed points, checked decode (SURVEY.md section 8 f4)" % (title, lg),
            "value": D.world * n * args.config_steps / el, "unit": "verifies/s", "ms_per_step": el / args.config_steps * 1e3,
            "bytes_per_unit": b_verify, "roofline": rf, "valu": v,
            "stage_ms_per_step": {"decode": ms[0], "straus_v": ms[1], "straus_u": ms[2], "finish": ms[3]}}
        cx.set_prevalidated(True)
        el_p, _ = timed(D, fn, max(2, args.config_steps), 1)
        cx.set_prevalidated(False)
        res["ietf_verify_" + tag]["prevalidated"] = {"value": D.world * n * max(2, args.config_steps) / el_p, "unit": "verifies/s",
                                                     "ms_per_step": el_p / max(2, args.config_steps) * 1e3}
        if True:
            # as a deployed verifier runs it: from (pk, alpha, proof), H hashed inside the call (cfg_from_alpha; kept affine on
            # the Edwards suites)
            fn_a = lambda: cx.ietf_verify_batch_alpha_dev(pk, msg, 32, g, c, s_, st)
            st.fill_(255)
            fn_a(); torch.cuda.synchronize()
            assert int(st.sum()) == 0
            el_a, _ = timed(D, fn_a, max(2, args.config_steps), 1)
            res["ietf_verify_" + tag]["from_alpha"] = {"value": D.world * n * max(2, args.config_steps) / el_a, "unit": "verifies/s",
                                                       "ms_per_step": el_a / max(2, args.config_steps) * 1e3}
        if sw or bsw:
            # the same verification from typed values: pk, input, output as x || y (no decompression)
            from ark_ec_vrfs_amd import _lib as _l
            xy = [torch.empty((n, 64), dtype=torch.uint8, device=D.dev) for _ in range(3)]
            for src_, dst_ in zip((pk, hh, g), xy):
                _l.check(lib.vrfhip_point_validate_batch_dev(cx.handle, n, src_.data_ptr(), dst_.data_ptr(), st.data_ptr(), stream), "validate")
            fn_xy = lambda: cx.ietf_verify_batch_affine_dev(xy[0], xy[1], xy[2], c, s_, st)
            fn_xy(); torch.cuda.synchronize()
            el_xy, _ = timed(D, fn_xy, max(2, args.config_steps), 1)
            assert int(st.sum()) == 0
            res["ietf_verify_" + tag]["affine_inputs"] = {"value": D.world * n * max(2, args.config_steps) / el_xy, "unit": "verifies/s",
                                                          "ms_per_step": el_xy / max(2, args.config_steps) * 1e3}
        if sw or bsw:
            # the Pedersen scheme on this suite (placeholder blinding base), per proof
            pc, rr, okp, sbb = mkp(), mkp(), mkp(), mk()
            fn = lambda: cx.pedersen_prove_batch_dev(sk, msg, 32, g, pc, rr, okp, s_, sbb, None, hh, st)
            fn(); torch.cuda.synchronize()
            cx.profile(True)
            el, _ = timed(D, fn, args.config_steps, 1)
            cx.profile(False)
            ms, groups = stage_avg(cx)
            assert int(st.sum()) == 0
            res["pedersen_prove_" + tag] = {
                "workload": "Pedersen VRF prove, %s, batch 2^%d per GPU; non-upstream blinding base (vrfhip_test_blinding_base)" % (title, lg),
                "value": D.world * n * args.config_steps / el, "unit": "proofs/s", "ms_per_step": el / args.config_steps * 1e3,
                "stage_ms_per_step": {"tai_find+prepare": ms[0], "mul": ms[1], "finish": ms[3]}}
            fn = lambda: cx.pedersen_verify_batch_dev(hh, g, pc, rr, okp, s_, sbb, st)
            fn(); torch.cuda.synchronize()
            cx.profile(True)
            el, _ = timed(D, fn, args.config_steps, 1)
            cx.profile(False)
            ms, groups = stage_avg(cx)
            assert int(st.sum()) == 0
            res["pedersen_verify_" + tag] = {
                "workload": "Pedersen VRF verify, %s, batch 2^%d per GPU, per proof, %d-byte points; non-upstream blinding base" % (title, lg, pw),
                "value": D.world * n * args.config_steps / el, "unit": "verifies/s", "ms_per_step": el / args.config_steps * 1e3,
                "stage_ms_per_step": {"decode": ms[0], "eq_h": ms[1], "eq_g": ms[2], "finish": ms[3]}}
        if sw or bsw:
            # ... and the whole batch through ONE MSM over 5 n + 2 points (k_p256_msm.hip; bandersnatch_sw: k_msm.hip behind k_bsw_rlc_decode)
            flag = torch.empty(1, dtype=torch.uint8, device=D.dev)
            seed = os.urandom(32)
            fn = lambda: cx.pedersen_verify_batch_rlc_dev(hh, g, pc, rr, okp, s_, sbb, st, flag, seed)
            fn(); torch.cuda.synchronize()
            cx.profile(True)
            el, _ = timed(D, fn, args.config_steps, 1)
            cx.profile(False)
            ms, groups = stage_avg(cx)
            assert int(flag[0]) == 0 and int(st.sum()) == 0
            res["pedersen_verify_batched_" + tag] = {
                "workload": "Pedersen VRF verify, %s, whole batch by one Pippenger MSM (random linear combination), 2^%d per GPU" % (title, lg),
                "value": D.world * n * args.config_steps / el, "unit": "verifies/s", "ms_per_step": el / args.config_steps * 1e3,
                "stage_ms_per_step": {"decode": ms[0], "msm_buckets": ms[1], "msm_final": ms[2] + ms[3]},
                "note": "non-upstream blinding base (vrfhip_test_blinding_base)"}
        if sw or bsw:
            fn = lambda: cx.ietf_prove_batch_dev(sk, msg, 32, g, c, s_, pk, hh, st)        # the IETF proofs back for the CPU leg
            fn(); torch.cuda.synchronize()
        if want_cpu and bsw:
            # oracle/c/oracle_bsw.c: the suite on the Weierstrass curve itself (Jacobian double-and-add), 16 threads
            from oracle import c_oracle as co
            cap = 1 << 15
            host = lambda t: t[:cap].cpu().numpy()
            skh, msgh, gh, ch, sh, pkh, hhh = (host(t) for t in (sk, msg, g, c, s_, pk, hh))

            def leg_p(k):
                ref = co.bsw_ietf_prove_batch(skh[:k], msgs=msgh[:k], ad=b"", threads=cpu_cores())
                assert (ref["output"] == gh[:k]).all() and (ref["c"] == ch[:k]).all() and (ref["s"] == sh[:k]).all(), \
                    "GPU proofs differ from the CPU oracle on the sample"
            res["ietf_prove_" + tag]["cpu_baseline"] = cpu_leg(leg_p, 16 * cpu_cores(), args.cpu_seconds / 2, cap, "proofs/s",
                                                               "proof bytes equal the GPU's", "oracle/c/oracle_bsw.c")

            def leg_v(k):
                stv = co.bsw_ietf_verify_batch(pkh[:k], hhh[:k], gh[:k], ch[:k], sh[:k], b"", threads=cpu_cores())
                assert not stv.any(), "CPU oracle rejects GPU-made proofs"
            res["ietf_verify_" + tag]["cpu_baseline"] = cpu_leg(leg_v, 16 * cpu_cores(), args.cpu_seconds / 2, cap, "verifies/s",
                                                                "statuses equal the GPU's", "oracle/c/oracle_bsw.c")
        elif want_cpu:
            from oracle import c_oracle as co
            cap = 1 << 15
            host = lambda t: t[:cap].cpu().numpy()
            skh, msgh, gh, ch, sh, pkh, hhh = (host(t) for t in (sk, msg, g, c, s_, pk, hh))
            if not sw:
                co.set_suite(oracle_suite)
            c_prove = co.p256_ietf_prove_batch if sw else co.ietf_prove_batch
            c_verify = co.p256_ietf_verify_batch if sw else co.ietf_verify_batch
            try:
                def leg_p(k):
                    ref = c_prove(skh[:k], msgs=msgh[:k], ad=b"", threads=cpu_cores())
                    assert (ref["output"] == gh[:k]).all() and (ref["c"] == ch[:k]).all() and (ref["s"] == sh[:k]).all(), \
                        "GPU proofs differ from the CPU oracle on the sample"
                csrc = "oracle/c/oracle_p256.c" if sw else "oracle/c/oracle_vrf.c"
                res["ietf_prove_" + tag]["cpu_baseline"] = cpu_leg(leg_p, 16 * cpu_cores(), args.cpu_seconds / 2, cap, "proofs/s",
                                                                   "proof bytes equal the GPU's", csrc)

                def leg_v(k):
                    stv = c_verify(pkh[:k], hhh[:k], gh[:k], ch[:k], sh[:k], b"", threads=cpu_cores())
                    assert not stv.any(), "CPU oracle rejects GPU-made proofs"
                res["ietf_verify_" + tag]["cpu_baseline"] = cpu_leg(leg_v, 16 * cpu_cores(), args.cpu_seconds / 2, cap, "verifies/s",
                                                                    "statuses equal the GPU's", csrc)
            finally:
                if not sw:
                    co.set_suite(1)
    finally:
        cx.close()
    return res


def cfg_shard_sizes(D, args, ctx, pk, hh, gamma, c, s):
    """The headline's operation at the per-GPU shard sizes of a strong-scaled 2^20 job (BASELINE.json: "2^20 ... sharded"):
    2^19 (N = 2), 2^18 (N = 4), 2^17 (N = 8) items on ONE GPU.  value / 2^20-per-GPU value is the per-GPU efficiency a
    strong-scaled job can reach at that N: what SCALE_rNN will show once an 8-GPU node is available."""
    torch = D.torch
    out = {}
    for lg in (19, 18, 17):
        m = 1 << lg
        if m >= pk.shape[0]:
            continue
        st = torch.empty(m, dtype=torch.uint8, device=D.dev)
        fn = lambda: ctx.ietf_verify_batch_dev(pk[:m], hh[:m], gamma[:m], c[:m], s[:m], st)
        fn(); torch.cuda.synchronize()
        steps = max(3, args.config_steps)
        el, _ = timed(D, fn, steps, 1)
        assert int(st.sum()) == 0
        out["ietf_verify_shard_2^%d" % lg] = {
            "workload": "the headline's verify on a 2^%d-item shard (per-GPU share of a 2^20 batch strong-scaled over %d GPUs)"
                        % (lg, 1 << (20 - lg)),
            "value": D.world * m * steps / el, "unit": "verifies/s", "ms_per_step": el / steps * 1e3}
    return out


def cfg_from_alpha(D, args, ctx, lib, pk, msg, hh, gamma, c, s):
    """Verification as a deployed verifier runs it: it holds pk, alpha and the proof, H is not on the wire.  Two forms on the
    headline batch: `vrfhip_hash_to_curve_batch_dev` then verify with the input declared validated (a cofactor multiple needs
    no subgroup test), and the fused `vrfhip_ietf_verify_batch_alpha_dev`, which keeps H on the device as affine coordinates
    (no compression, second square root or test).  145 B per verify cross the boundary (pk, a 32-byte alpha, Gamma, c, s)."""
    from ark_ec_vrfs_amd import _lib
    torch = D.torch
    n = pk.shape[0]
    st = torch.empty(n, dtype=torch.uint8, device=D.dev)
    h2 = torch.empty_like(hh)
    stream = torch.cuda.current_stream().cuda_stream
    steps = max(3, args.config_steps)

    def two_calls():
        _lib.check(lib.vrfhip_hash_to_curve_batch_dev(ctx.handle, n, msg.data_ptr(), None, msg.shape[1], h2.data_ptr(), stream), "h2c")
        ctx.ietf_verify_batch_dev(pk, h2, gamma, c, s, st)

    fused = lambda: ctx.ietf_verify_batch_alpha_dev(pk, msg, msg.shape[1], gamma, c, s, st)
    flags0 = ctx.get_flags() if hasattr(ctx, "get_flags") else 0
    ctx.set_flags(flags0 | ctx.PREVALIDATED_INPUT)
    two_calls(); torch.cuda.synchronize()
    assert int(st.sum()) == 0 and bool((h2 == hh).all())
    el2, _ = timed(D, two_calls, steps, 1)
    ctx.set_flags(flags0)
    st.fill_(255)
    fused(); torch.cuda.synchronize()
    assert int(st.sum()) == 0
    el1, _ = timed(D, fused, steps, 1)
    return {"ietf_verify_from_alpha": {
        "workload": "IETF ECVRF verify from (pk, alpha, proof), Bandersnatch, batch 2^%d per GPU: Input::new(alpha) inside the "
                    "call, H kept on the device as affine coordinates; pk and Gamma checked as in the headline"
                    % (n.bit_length() - 1),
        "value": D.world * n * steps / el1, "unit": "verifies/s", "ms_per_step": el1 / steps * 1e3, "bytes_per_unit": 145,
        "two_calls": {"workload": "vrfhip_hash_to_curve_batch_dev, then verify with VRFHIP_FLAG_PREVALIDATED_INPUT",
                      "value": D.world * n * steps / el2, "unit": "verifies/s", "ms_per_step": el2 / steps * 1e3}}}


def cfg_pairing(D, args, ctx, want_cpu):
    """BASELINE.json configs[4]: 2^14 pairing checks e(P0,Q0) e(P1,Q1) == 1 (Miller loop + final exponentiation)."""
    torch = D.torch
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "pairing_items.json")))
    hx = lambda h: np.frombuffer(bytes.fromhex(h), np.uint8)
    k = 1 << 14
    g1 = np.stack([hx(it["g1"]) for it in fx["per_item"]])
    g2 = np.stack([hx(it["g2"]) for it in fx["per_item"]])
    reps = k // g1.shape[0]
    # 2^14 DISTINCT items from the 8 fixture items: (A, B) passes against (Q0, Q1) iff (jA, jB) does (bilinearity), so item
    # 8 r + t is (r + 1) x fixture item t -- distinct G1 points in every item, the fixture's 8 G2 pairs in turn.  The
    # multiples are a running sum in plain affine arithmetic over the BLS12-381 base field (no timed run used
    # distinct items; the kernels have no data-dependent shortcuts, so the time is the same)
    g1_all = _g1_multiples(g1, reps)
    d1 = torch.from_numpy(g1_all).to(D.dev)
    d2 = torch.from_numpy(np.tile(g2, (reps, 1)).copy()).to(D.dev)
    pstat = torch.empty(k, dtype=torch.uint8, device=D.dev)
    res = {}

    def run(name, fn, bytes_per, kernel, workload, pmc_name):
        fn(); torch.cuda.synchronize()
        evs = []

        def evfn():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record()       # the kernel is launched on torch's current stream (api.py)
            evs.append((a, b))
        el, _ = timed(D, evfn, args.config_steps, 1, gather_t=pstat, n_per_rank=k)
        assert int(pstat.sum()) == 0
        kms = sum(a.elapsed_time(b) for a, b in evs[-args.config_steps:]) / args.config_steps
        rf, v = roofline(kernel, bytes_per, k, kms, args.config_steps, pmc_for(pmc_name, 14))
        res[name] = {"workload": workload, "value": D.world * k * args.config_steps / el, "unit": "checks/s",
                     "ms_per_step": el / args.config_steps * 1e3, "bytes_per_unit": bytes_per, "roofline": rf, "valu": v}

    run("pairing_check", lambda: ctx.pairing_check_batch_dev(d1, d2, pstat), B_PAIRING,
        "k_pairing_lines_oct + k_pairing_check2_oct_lines (one item per 8 lanes, Fp2 split over lane pairs: G2 lines -> HBM, then 2-pair Miller loop + final exponentiation)",
        "BLS12-381 pairing check, 2 (G1, G2) pairs per item, batch 2^14 per GPU (BASELINE.json configs[4]); "
        "2^14 distinct items (multiples of 8 fixture items: distinct G1 points, 8 G2 pairs in turn)", "pairing_check")
    s1 = np.stack([hx(h) for h in fx["shared"]])
    s1_all = _g1_multiples(s1, k // s1.shape[0])
    ds1 = torch.from_numpy(s1_all).to(D.dev)
    dsh = torch.from_numpy(hx(fx["shared_g2"]).copy()).to(D.dev)
    run("pairing_check_shared_g2", lambda: ctx.pairing_check_batch_dev(ds1, dsh, pstat, g2_shared=True), B_PAIRING_SHARED,
        "k_pairing_check2_oct_prepared (lines of the shared G2 pair prepared once per context)",
        "the same check against ONE shared G2 pair (a KZG verifier's SRS), batch 2^14 per GPU", "pairing_check_shared")
    # the same shared-G2 checks as ONE batch: two G1 multi-scalar multiplications + one pairing (random linear
    # combination).  At 2^14 the single pairing's latency (one quad) is the whole cost; the amortised rate shows at 2^18.
    for lg in (14, 18):
        m = 1 << lg
        dm = torch.from_numpy(np.tile(s1_all, (m // s1_all.shape[0], 1)).copy()).to(D.dev)      # the 2^14 distinct items (tiled at 2^18)
        mst = torch.empty(m, dtype=torch.uint8, device=D.dev)
        verdict = torch.empty(1, dtype=torch.uint8, device=D.dev)
        seed = os.urandom(32)
        fn = lambda: ctx.pairing_check_batch_rlc_dev(dm, dsh, mst, verdict, seed)
        fn(); torch.cuda.synchronize()
        ctx.profile(True)
        el, _ = timed(D, fn, args.config_steps, 1, gather_t=mst, n_per_rank=m)
        ctx.profile(False)
        ms, groups = stage_avg(ctx)
        assert int(verdict[0]) == 0 and int(mst.sum()) == 0
        heavy = max(range(4), key=lambda k: ms[k])
        kname = ("k_g1_prep_rlc", "k_g1_buckets (Pippenger, 512 buckets per window in LDS)", "k_g1_final",
                 "k_pairing_check2_quad_prepared (ONE check for the batch)")[heavy]
        rf, v = roofline(kname, B_PAIRING_SHARED, m, ms[heavy], groups, pmc_for("pairing_check_batched", lg))
        res["pairing_check_batched_shared_g2_2^%d" % lg] = {
            "workload": "2^%d checks against one shared G2 pair verified as ONE batch: two BLS12-381 G1 MSMs + one pairing "
                        "check (random linear combination, 128-bit weights)" % lg,
            "value": D.world * m * args.config_steps / el, "unit": "checks/s", "ms_per_step": el / args.config_steps * 1e3,
            "bytes_per_unit": B_PAIRING_SHARED, "roofline": rf, "valu": v,
            "stage_ms_per_step": {"prep": ms[0], "msm_buckets": ms[1], "msm_final": ms[2], "pairing": ms[3]}}
    # steady state of a verifier that keeps four batches in flight: one context and one stream each, so that a
    # batch's latency-bound tail (bucket reduction + the single pairing: two waves, 12 ms) overlaps the others'
    # MSMs.  Needs the streams on distinct hardware queues: GPU_MAX_HW_QUEUES (set at the top of this file).
    m = 1 << 18
    extra = [type(ctx)(D.local) for _ in range(3)]
    lanes = []
    for cx in [ctx] + extra:
        lanes.append((cx, torch.cuda.Stream(), torch.from_numpy(np.tile(s1_all, (m // s1_all.shape[0], 1)).copy()).to(D.dev),
                      torch.empty(m, dtype=torch.uint8, device=D.dev), torch.empty(1, dtype=torch.uint8, device=D.dev)))
    seed = os.urandom(32)
    torch.cuda.synchronize()

    def fn2():
        for cx, st, dm2, mst2, vd2 in lanes:
            with torch.cuda.stream(st):
                cx.pairing_check_batch_rlc_dev(dm2, dsh, mst2, vd2, seed)
    fn2(); torch.cuda.synchronize()
    el, _ = timed(D, fn2, args.config_steps, 1)      # statuses stay on their ranks: checked below
    assert all(int(l[4][0]) == 0 and int(l[3].sum()) == 0 for l in lanes)
    res["pairing_check_batched_shared_g2_2^18_four_in_flight"] = {
        "workload": "as above at 2^18, four batches in flight (four contexts, four streams): a step is all four batches",
        "value": D.world * len(lanes) * m * args.config_steps / el, "unit": "checks/s", "ms_per_step": el / args.config_steps * 1e3,
        "bytes_per_unit": B_PAIRING_SHARED, "roofline": None, "valu": None}
    for cx in extra:
        cx.close()
    if want_cpu:
        from oracle import c_oracle as co
        g1h, g2h = g1_all, np.tile(g2, (reps, 1))

        def leg_c(m):
            stc = co.pairing_check_batch(g1h[:m], g2h[:m], threads=cpu_cores())
            assert not stc.any(), "CPU oracle rejects the fixture items"
        leg = cpu_leg(leg_c, 4 * cpu_cores(), args.cpu_seconds, k, "checks/s", "verdicts equal the GPU's")
        leg["note"] = ("CPU restatement in C (oracle/c/oracle_bls.c: 6 x 64-bit Montgomery limbs, Karatsuba tower, plain "
                       "square-and-multiply final exponentiation), not arkworks: no Rust toolchain on this box")
        res["pairing_check"]["cpu_baseline"] = leg          # one record; the other pairing legs point at it
        for k in res:
            if k != "pairing_check":
                res[k]["cpu_baseline_ref"] = "pairing_check"
    return res


def headline_cpu_baseline(args, pk, hh, gamma, c, s, status):
    """The plain-C oracle (a port, not arkworks) on this box's host cores, bounded sample of the same batch."""
    from oracle import c_oracle as co
    cap = 1 << 18
    host = lambda t: t[:cap].cpu().numpy()
    pkh, hhh, gh, ch, sh, sth = host(pk), host(hh), host(gamma), host(c), host(s), host(status)

    def leg(k):
        st = co.ietf_verify_batch(pkh[:k], hhh[:k], gh[:k], ch[:k], sh[:k], b"", threads=cpu_cores())
        assert (st == sth[:k]).all(), "GPU statuses differ from the CPU oracle on the sample"
    out = cpu_leg(leg, 64 * cpu_cores(), args.cpu_seconds, cap, "verifies/s", "statuses equal the GPU's")
    m1 = 256
    t0 = time.perf_counter()
    co.ietf_verify_batch(pkh[:m1], hhh[:m1], gh[:m1], ch[:m1], sh[:m1], b"", threads=1)
    out["single_core"] = m1 / (time.perf_counter() - t0)
    out["note"] += ("; like the GPU path it validates the three decoded points (r*P = O, arkworks' own subgroup test)"
                    if not args.prevalidated else "; run with the oracle's default checked decode")
    return out


def run_rank(args):
    D = Dist(args)
    torch = D.torch
    from ark_ec_vrfs_amd import Context, _lib
    rank, world, dev = D.rank, D.world, D.dev
    n = 1 << args.log2_batch
    lo = rank * n                                  # weak scaling: rank g owns items [g*n, (g+1)*n)
    ctx = Context(D.local)
    can_check = hasattr(ctx, "set_prevalidated")    # the fused subgroup check (vrfhip_ctx_set_flags)
    if args.prevalidated and can_check:
        ctx.set_prevalidated(True)
    lib = _lib.load()
    stream = torch.cuda.current_stream().cuda_stream

    def make_batch(first, m):
        """Synthetic items [first, first + m), produced on the GPU and left resident in HBM."""
        seeds = (torch.arange(m, dtype=torch.int64, device=dev) + first).view(torch.uint8).reshape(m, 8)
        sk_ = torch.empty((m, 32), dtype=torch.uint8, device=dev)
        _lib.check(lib.vrfhip_secret_from_seed_batch_dev(ctx.handle, m, seeds.data_ptr(), 8, sk_.data_ptr(), None, stream), "seed")
        msg_ = torch.from_numpy(synth_msgs(first, m)).to(dev)
        mk_ = lambda: torch.empty((m, 32), dtype=torch.uint8, device=dev)
        b = {"sk": sk_, "msg": msg_, "gamma": mk_(), "c": mk_(), "s": mk_(), "pk": mk_(), "hh": mk_()}
        pst_ = torch.empty(m, dtype=torch.uint8, device=dev)
        best = 1e30
        for _ in range(2):                              # the first call includes first-touch of the workspace
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.ietf_prove_batch_dev(sk_, msg_, 32, b["gamma"], b["c"], b["s"], b["pk"], b["hh"], pst_)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        assert int(pst_.sum()) == 0
        b["prove_s"] = best
        b["status"] = torch.full((m,), 255, dtype=torch.uint8, device=dev)
        return b

    def verify_step(b):
        return lambda: ctx.ietf_verify_batch_dev(b["pk"], b["hh"], b["gamma"], b["c"], b["s"], b["status"])

    # weak: every rank its own 2^log2_batch items.  strong: ONE global batch of 2^log2_batch items, rank g verifies the
    # contiguous shard sharding.shard_range gives it (BASELINE.json configs: "2^20 ... sharded 8x").  At N = 1 they coincide.
    from ark_ec_vrfs_amd.sharding import shard_range
    weak = make_batch(lo, n)
    s_lo, s_hi = shard_range(n, rank, world)
    strong = weak if world == 1 else make_batch(s_lo, s_hi - s_lo)
    main_b, other_b = (strong, weak) if args.scaling == "strong" else (weak, strong)
    main_n = (s_hi - s_lo) if args.scaling == "strong" else n
    main_global = n if args.scaling == "strong" else world * n
    sk, msg, gamma, c, s, pk, hh, status = (weak[k] for k in ("sk", "msg", "gamma", "c", "s", "pk", "hh", "status"))
    prove_s = weak["prove_s"]

    step = verify_step(main_b)
    for _ in range(args.warmup):
        step()
        D.gather(main_b["status"], main_n, main_global)
    ctx.profile(True)
    elapsed, full = timed(D, step, args.steps, 0, gather_t=main_b["status"], n_per_rank=main_n, n_global=main_global)
    ctx.profile(False)
    stage_ms, groups = stage_avg(ctx)
    n_bad = int((full != 0).sum())
    assert n_bad == 0 and full.shape[0] == main_global, f"{n_bad} synthetic proofs failed to verify"
    modes = {args.scaling: {"value": main_global * args.steps / elapsed, "ms_per_step": elapsed / args.steps * 1e3,
                            "global_batch": main_global, "items_per_gpu": main_global // world, "steps": args.steps}}
    other = "weak" if args.scaling == "strong" else "strong"
    if world == 1:
        modes[other] = dict(modes[args.scaling])
    else:
        o_n = n if other == "weak" else (s_hi - s_lo)
        o_global = world * n if other == "weak" else n
        o_steps = max(3, args.config_steps)
        el_o, full_o = timed(D, verify_step(other_b), o_steps, 1, gather_t=other_b["status"], n_per_rank=o_n, n_global=o_global)
        assert int((full_o != 0).sum()) == 0 and full_o.shape[0] == o_global
        modes[other] = {"value": o_global * o_steps / el_o, "ms_per_step": el_o / o_steps * 1e3, "global_batch": o_global,
                        "items_per_gpu": o_global // world, "steps": o_steps}
    step = verify_step(weak)                         # the secondary legs below run on the rank's own 2^log2_batch items

    # the same batch with the points declared validated (typed Public/Input/Output values): round 1's operation
    preval = None
    if can_check and not args.prevalidated:
        ctx.set_prevalidated(True)
        el_p, _ = timed(D, step, max(2, args.config_steps), 1, gather_t=status, n_per_rank=n)
        ctx.set_prevalidated(False)
        preval = {"workload": "the headline batch with VRFHIP_FLAG_PREVALIDATED_* set: no subgroup test (typed, already "
                              "validated values -- the operation round 1 reported)",
                  "value": world * n * max(2, args.config_steps) / el_p, "unit": "verifies/s",
                  "ms_per_step": el_p / max(2, args.config_steps) * 1e3}

    want_cpu = (not args.no_cpu_baseline) and world == 1
    configs = {}
    if not args.no_configs:
        from ark_ec_vrfs_amd import BabyJubJubSha512Tai, BandersnatchSwSha512Tai, Ed25519Sha512Tai, Secp256r1Sha256Tai
        legs = (("ietf_prove", lambda: {"ietf_prove": cfg_ietf_prove(D, args, ctx, sk, msg, want_cpu and rank == 0)}),
                ("shard_sizes", lambda: cfg_shard_sizes(D, args, ctx, pk, hh, gamma, c, s)),
                ("from_alpha", lambda: cfg_from_alpha(D, args, ctx, lib, pk, msg, hh, gamma, c, s)),
                ("ietf_verify_keyed", lambda: {"ietf_verify_keyed": cfg_ietf_keyed(D, args, ctx, msg, lo)}),
                ("pedersen_jubjub", lambda: cfg_pedersen_jubjub(D, args, msg, lo, want_cpu and rank == 0)),
                ("ed25519", lambda: cfg_suite_ietf(D, args, "ed25519", Ed25519Sha512Tai, 3, "Ed25519_SHA-512_TAI", lo,
                                                   want_cpu and rank == 0)),
                ("babyjubjub", lambda: cfg_suite_ietf(D, args, "babyjubjub", BabyJubJubSha512Tai, 4, "BabyJubJub_SHA-512_TAI", lo,
                                                      want_cpu and rank == 0)),
                ("secp256r1", lambda: cfg_suite_ietf(D, args, "secp256r1", Secp256r1Sha256Tai, 5, "P256_SHA256_TAI (RFC 9381 suite 0x01)", lo,
                                                     want_cpu and rank == 0)),
                ("bandersnatch_sw", lambda: cfg_suite_ietf(D, args, "bandersnatch_sw", BandersnatchSwSha512Tai, 1,
                                                           "Bandersnatch_SW_SHA-512_TAI (33-byte arkworks SW points)", lo, want_cpu and rank == 0)),
                ("pairing", lambda: cfg_pairing(D, args, ctx, want_cpu and rank == 0)))
        only = [x for x in args.only.split(",") if x]
        D.local_legs = world > 1
        for name, leg in legs:
            if only and name not in only:
                continue
            try:
                res = leg()
            except Exception as e:                  # the headline number must not depend on a secondary leg
                res = {name: {"error": repr(e)}}
            configs.update(D.merge_leg_results(res))
        D.local_legs = False

    if rank == 0:
        value = main_global * args.steps / elapsed
        checked = can_check and not args.prevalidated
        r, v = roofline("k_verify_straus<1> (V = s*H - c*Gamma, the longest kernel)", B_VERIFY, main_n, stage_ms[1], groups,
                        pmc_for("ietf_verify", args.log2_batch))
        out = {
            "metric": "Bandersnatch IETF-ECVRF verifies/sec, 2^%d batch %s" % (
                args.log2_batch, "sharded over the GPUs" if args.scaling == "strong" else "per GPU"),
            "value": value, "unit": "verifies/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "scaling_modes": modes,
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "IETF ECVRF verify, Bandersnatch_SHA-512_ELL2, batch 2^%d %s, compressed points "
                                   "(161 B/verify), %s, ad=\"\" (BASELINE.json configs[2])"
                                   % (args.log2_batch, "in total, cut into contiguous shards" if args.scaling == "strong" else "per GPU",
                                      "checked decode: on curve + prime-order subgroup, as arkworks' "
                                      "deserialisation" if checked else "inputs declared pre-validated (no subgroup check)"),
                       "global_batch": main_global, "parallelism": "items sharded x%d, result gather only" % world,
                       "subgroup_check": checked},
            "roofline": r, "valu": v, "pmc_stale": pmc_table()[1],
            "stage_ms_per_step": {"decode": stage_ms[0], "straus_v": stage_ms[1], "straus_u": stage_ms[2],
                                  "finish": stage_ms[3]},
            "proofs_per_sec": n / prove_s,
        }
        if preval:
            out["prevalidated"] = preval
        if want_cpu:
            out["cpu_baseline"] = headline_cpu_baseline(args, pk, hh, gamma, c, s, status)
        if configs:
            out["configs"] = configs
        print_result(out, configs, world)
    D.close()
    ctx.close()


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    run_rank(args)


if __name__ == "__main__":
    main()
