"""CPU: bench.py's launch plumbing.  `python bench.py --gpus N` (N > 1, not under torch.distributed.run) must start
its own ranks through `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`
BEFORE anything touches a GPU (the parent imports neither torch nor the library), pass its own flags through, and
under torch.distributed.run it must not launch again.  Also: the JSON line's contract keys on a recorded run."""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_n_self_launches_through_torch_distributed_run():
    env = dict(os.environ, VRFHIP_BENCH_DRYRUN="1")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "7", "--warmup", "2"],
                         env=env, capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    d = json.loads(out.stdout.strip().splitlines()[-1])
    cmd = d["launch"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert int(cmd[cmd.index("--master-port") + 1]) > 0
    tail = cmd[cmd.index(os.path.join(ROOT, "bench.py")) + 1:]
    assert tail == ["--gpus", "4", "--steps", "7", "--warmup", "2"]
    assert d["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_scaling_flag_reaches_the_ranks_and_both_modes_are_in_the_line():
    """--scaling strong (ONE 2^20 batch cut into N shards, BASELINE.json "2^20 ... sharded 8x") is passed through to the
    ranks; the JSON line states which mode `value` is and carries both under `scaling_modes`."""
    env = dict(os.environ, VRFHIP_BENCH_DRYRUN="1")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--scaling", "strong"],
                         env=env, capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    cmd = json.loads(out.stdout.strip().splitlines()[-1])["launch"]
    assert cmd[cmd.index(os.path.join(ROOT, "bench.py")) + 1:] == ["--gpus", "8", "--scaling", "strong"]
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '"scaling": args.scaling' in src and '"scaling_modes": modes' in src and "shard_range(n, rank, world)" in src
    # a recorded two-rank strong-scaling rehearsal (gloo ranks sharing the one GPU of the box): both modes, shard sizes add up
    f = os.path.join(ROOT, "profiles", "r03", "bench_n2_gloo_strong_rehearsal_2_18.json")
    d = json.loads(open(f).read().strip().splitlines()[-1])
    assert d["scaling"] == "strong" and d["n_gpus"] == 2 and set(d["scaling_modes"]) == {"weak", "strong"}
    st, wk = d["scaling_modes"]["strong"], d["scaling_modes"]["weak"]
    assert st["global_batch"] == 1 << 18 and st["items_per_gpu"] == 1 << 17 and wk["global_batch"] == 2 << 18
    assert abs(d["value"] - st["value"]) < 1e-6 and d["config"]["global_batch"] == 1 << 18


def test_parent_process_imports_no_gpu_runtime_before_spawning():
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[:src.index("def self_launch")]
    assert "import torch" not in head and "ark_ec_vrfs_amd" not in head     # module level: numpy / stdlib only
    body = src[src.index("def self_launch"):src.index("def synth_msgs")]
    assert "import torch" not in body and "Context(" not in body and "subprocess.run" in body
    main = src[src.index("def main():"):]
    assert main.index("self_launch(args)") < main.index("run_rank(args)")


def test_recorded_bench_lines_carry_the_contract_keys():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r02", "bench_n1*.json")) +
                   glob.glob(os.path.join(ROOT, "profiles", "r03", "bench_n1*.json")))
    assert any("r03" in f for f in files), "no recorded bench line under profiles/r03"
    for f in files:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                  "vs_baseline", "dtype", "data", "config", "roofline"):
            assert k in d, (f, k)
        assert d["config"]["workload"] and d["roofline"]["bound"] == "hbm" and d["vs_baseline"] is None
        r = d["roofline"]
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
        if "r03" in f:              # this round's line: the CPU leg, the issue-slot block, both scaling modes, every f4 suite
            assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1
            assert 0.5 < d["valu"]["issue_slot_frac"] < 1.2 and set(d["scaling_modes"]) == {"weak", "strong"}
            for leg in ("ietf_prove", "pedersen_verify_jubjub", "ietf_verify_ed25519", "ietf_verify_babyjubjub",
                        "ietf_prove_secp256r1", "ietf_verify_secp256r1", "pairing_check"):
                assert d["configs"][leg]["value"] > 0, leg


def test_secondary_leg_results_merge_across_ranks():
    """bench.merge_rank_results: the slowest rank sets a config's rate; an error or a missing config on ANY rank replaces
    the number (the headline line must still be printed)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    r0 = {"a": {"value": 10.0, "ms_per_step": 1.0, "unit": "x/s"}, "b": {"value": 5.0, "ms_per_step": 2.0}, "c": {"value": 1.0, "ms_per_step": 1.0}}
    r1 = {"a": {"value": 8.0, "ms_per_step": 1.25}, "b": {"error": "boom"}, "d": {"error": "leg failed early"}}
    m = b.merge_rank_results(r0, [b.slim_results(r0), b.slim_results(r1)])
    assert m["a"] == {"value": 8.0, "ms_per_step": 1.25, "unit": "x/s"}
    assert m["b"] == {"error": "boom"} and "error" in m["c"] and m["d"] == {"error": "leg failed early"}
    assert b.merge_rank_results(r0, [b.slim_results(r0)] * 3)["b"]["value"] == 5.0


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_final_line_stays_under_4_kb_with_every_leg_present():
    """round 3's 25.7 KB line overflowed the driver's 8 KB stdout tail and the headline went unparsed.
    assemble_final_line must bring ANY full record (here: round 3's, 30 legs with cpu_baseline notes, and a doubled copy of
    it) under 4096 bytes with the headline's roofline, valu and cpu_baseline intact."""
    b = _load_bench()
    full = json.loads(open(os.path.join(ROOT, "profiles", "r03", "bench_n1_default.json")).read().strip().splitlines()[-1])
    assert len(json.dumps(full)) > 20000 and len(full["configs"]) >= 20
    fat = dict(full, configs=dict(full["configs"], **{k + "_again": v for k, v in full["configs"].items()}))
    fat["cpu_baseline"] = dict(full["cpu_baseline"], note=full["cpu_baseline"]["note"] * 8)
    for rec in (full, fat):
        line = b.assemble_final_line(rec, "bench_configs_n1.json")
        text = json.dumps(line)
        assert len(text) <= b.FINAL_LINE_MAX == 4096, len(text)
        back = json.loads(text)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                  "scaling_modes", "vs_baseline", "dtype", "data", "config", "roofline", "valu", "stage_ms_per_step",
                  "cpu_baseline", "configs_file"):
            assert k in back, k
        assert "configs" not in back
        assert abs(back["roofline"]["frac"] / full["roofline"]["frac"] - 1) < 1e-5
        assert abs(back["roofline"]["frac"] - back["roofline"]["achieved"] / back["roofline"]["peak"]) < 1e-12
        assert abs(back["cpu_baseline"]["value"] / full["cpu_baseline"]["value"] - 1) < 1e-5
        assert abs(back["value"] / full["value"] - 1) < 1e-5 and back["config"]["workload"] == full["config"]["workload"]
    # with round 3's real leg count every leg also has its one-number summary in the line
    line = b.assemble_final_line(full, "x")
    assert set(line["configs_summary"]) == set(full["configs"]) and "configs_summary_truncated" not in line


def test_last_stdout_line_of_a_run_is_the_short_headline(tmp_path):
    """What the driver does: take the LAST stdout line and parse it.  run_rank's printing, replayed on a recorded full
    record (no GPU here): the legs go out on an earlier line and into the configs file, the last line parses and carries
    roofline.frac and cpu_baseline.value."""
    code = (
        "import json, sys, os\n"
        "sys.argv = ['bench.py']\n"
        "import importlib.util\n"
        "spec = importlib.util.spec_from_file_location('bench_mod', %r)\n"
        "b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)\n"
        "full = json.loads(open(%r).read().strip().splitlines()[-1])\n"
        "b.print_result(full, full.get('configs'), 1)\n"
    ) % (os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "profiles", "r03", "bench_n1_default.json"))
    cf = str(tmp_path / "cfg.json")
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, VRFHIP_BENCH_CONFIGS_FILE=cf),
                         capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    assert len(lines) == 2 and len(lines[-1]) <= 4096 and "bench_configs" in json.loads(lines[0])
    d = json.loads(lines[-1])
    assert d["roofline"]["frac"] > 0 and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["kind"] == "port"
    saved = json.load(open(cf))
    assert saved["configs"]["pairing_check"]["value"] > 0 and saved["headline"]["value"] == json.loads(
        open(os.path.join(ROOT, "profiles", "r03", "bench_n1_default.json")).read().strip().splitlines()[-1])["value"]
    # the driver's 8 KB tail always contains the whole last line
    assert len(lines[-1].encode()) + 1 < 8192


def test_stale_profile_counters_are_not_attached(tmp_path):
    """`valu` and `roofline.traffic` are replayed from profiles/pmc_kernels.json.  The file carries the
    sha256 of the kernel sources it was measured on; a stamp that does not match the tree (or no stamp) gives no counters
    and a reason, a matching stamp gives them."""
    from ark_ec_vrfs_amd._lib import source_stamp
    b = _load_bench()
    entry = {"ietf_verify": {"log2_batch": 20, "hbm_bytes_per_launch": 1.0, "valu_lane_instructions_per_launch": 2.0}}
    f = tmp_path / "pmc.json"
    for stamp, live in (({"source_sha256": source_stamp()}, True), ({"source_sha256": "0" * 64}, False), (None, False)):
        d = dict(entry)
        if stamp is not None:
            d["_stamp"] = stamp
        f.write_text(json.dumps(d))
        os.environ["VRFHIP_BENCH_PMC_FILE"] = str(f)
        try:
            b._PMC = None
            tab, why = b.pmc_table()
            assert (b.pmc_for("ietf_verify", 20) is not None) == live and (why is None) == live
            assert b.pmc_for("ietf_verify", 19) is None
            r, v = b.roofline("k", 161, 1 << 20, 14.5, 20, b.pmc_for("ietf_verify", 20))
            assert (r["traffic"] is not None) == live and (v is not None) == live
        finally:
            del os.environ["VRFHIP_BENCH_PMC_FILE"]
            b._PMC = None
    # the committed summary either matches the committed sources or is ignored -- never silently replayed
    tab, why = b.pmc_table()
    committed = json.load(open(os.path.join(ROOT, "profiles", "pmc_kernels.json")))
    assert bool(tab) == ((committed.get("_stamp") or {}).get("source_sha256") == source_stamp()) and bool(tab) == (why is None)
