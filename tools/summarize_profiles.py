"""Condense the rocprofv3 passes of tools/profile_round.sh (gpurun_out/<tag>/) into the tracked summaries:
profiles/<round>/rocprofv3_kernel_stats_bench_steps3.csv, profiles/<round>/rocprofv3_pmc_summary.json and
profiles/pmc_k_verify_straus.json (the measured constants bench.py attaches to its roofline object).
usage: python tools/summarize_profiles.py gpurun_out/r01b profiles/r01"""
import collections, csv, json, os, re, shutil, sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "stats", "run_kernel_stats.csv"), os.path.join(dst, "rocprofv3_kernel_stats_bench_steps3.csv"))
def collect(prefix):
    """Counters (averages per launch) and kernel-trace durations of one profiled workload."""
    summary = collections.OrderedDict()
    for sub in ("sq", "fetch", "write", "tcc"):
        path = os.path.join(src, prefix + sub, "run_counter_collection.csv")
        if not os.path.exists(path):
            continue
        acc = collections.defaultdict(list)
        disp = {}
        for r in csv.DictReader(open(path)):
            name = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
            if not name.startswith("vrf::"):
                continue
            acc[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
            disp[name] = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count",
                                            "Accum_VGPR_Count", "SGPR_Count")}
        for (name, ctr), vals in acc.items():
            e = summary.setdefault(name, collections.OrderedDict())
            e["dispatch"] = disp[name]
            e[ctr] = sum(vals) / len(vals)          # average per launch
            e.setdefault("launches", len(vals))
    spath = os.path.join(src, (prefix + "stats") if prefix else "stats", "run_kernel_stats.csv")
    stats = {}
    if os.path.exists(spath):
        stats = {re.sub(r"^void ", "", r["Name"]).split("(")[0]: float(r["AverageNs"]) for r in csv.DictReader(open(spath))}
    for name, e in summary.items():
        if name in stats:
            e["avg_duration_ns_kernel_trace"] = stats[name]
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            # gfx950: FETCH_SIZE counts 128-B requests of 16 B/lane loads at 64 B -> x2; units of 1 KiB
            e["hbm_bytes_per_launch_corrected"] = e["FETCH_SIZE"] * 2 * 1024 + e["WRITE_SIZE"] * 1024
        if "SQ_INSTS_VALU" in e and "avg_duration_ns_kernel_trace" in e:
            e["valu_frac_of_peak"] = e["SQ_INSTS_VALU"] * 64 / (e["avg_duration_ns_kernel_trace"] * 1e-9) / 3.93216e13
    return summary


workloads = collections.OrderedDict()
workloads["bench_ietf_verify_2^20 (python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline)"] = collect("")
for prefix, label in (("rlc_", "pedersen_batched_verify_2^20 (tools/gpu_rlc_prof.py 20)"),
                      ("pairing_", "pairing_check_2^16 (tools/gpu_pairing_prof.py 16)"),
                      ("msm_", "msm_2^20 (tools/gpu_msm_prof.py 20)")):
    w = collect(prefix)
    if w:
        workloads[label] = w
        sp = os.path.join(src, prefix + "stats", "run_kernel_stats.csv")
        if os.path.exists(sp):
            shutil.copy(sp, os.path.join(dst, "rocprofv3_kernel_stats_%s.csv" % prefix.rstrip("_")))
summary = next(iter(workloads.values()))
json.dump({"command": "rocprofv3 --pmc <one group per pass> --kernel-trace -- <workload>",
           "note": "averages per launch; passes: {SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE}, {FETCH_SIZE}, {WRITE_SIZE}, {TCC_HIT_sum TCC_MISS_sum}; "
                   "valu_frac_of_peak = SQ_INSTS_VALU x 64 / duration / (256 CU x 4 SIMD x 16 lanes x 2.4 GHz)",
           "workloads": workloads}, open(os.path.join(dst, "rocprofv3_pmc_summary.json"), "w"), indent=1)
k = "vrf::k_verify_straus<vrf::SuiteBS, 1>"
e = summary[k]
out = {"kernel": k, "log2_batch": 20,
       "hbm_bytes_per_launch": e["hbm_bytes_per_launch_corrected"],
       "fetch_size_kb_raw": e["FETCH_SIZE"], "write_size_kb_raw": e["WRITE_SIZE"],
       "correction": "read bytes = 2 x FETCH_SIZE x 1024 (gfx950 FETCH_SIZE tallies 128-B requests at 64 B for 16 B/lane loads); write bytes = WRITE_SIZE x 1024",
       "valu_wave_instructions_per_launch": e["SQ_INSTS_VALU"], "valu_lane_instructions_per_launch": e["SQ_INSTS_VALU"] * 64,
       "sq_waves": e["SQ_WAVES"], "grbm_gui_active_sum8xcd": e["GRBM_GUI_ACTIVE"],
       "tcc_hit": e.get("TCC_HIT_sum"), "tcc_miss": e.get("TCC_MISS_sum"),
       "avg_duration_ns_kernel_trace": e.get("avg_duration_ns_kernel_trace"),
       "source": "%s/rocprofv3_pmc_summary.json (rocprofv3 --pmc passes of `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline`)" % dst}
json.dump(out, open(os.path.join(os.path.dirname(dst.rstrip("/")), "pmc_k_verify_straus.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
for label, w in workloads.items():
    print(label)
    for name, e in w.items():
        print("  %-58s valu=%.4g dur=%.3f ms hbm=%.3g B valu_frac=%.2f" % (name[:58], e.get("SQ_INSTS_VALU", 0),
              e.get("avg_duration_ns_kernel_trace", 0) / 1e6, e.get("hbm_bytes_per_launch_corrected", 0), e.get("valu_frac_of_peak", 0)))
for f in os.listdir(src):
    if f.startswith("gpu_") and f.endswith(".log"):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f[4:].replace("_time", "_timing")))
