"""Condense tools/profile_round3.sh's rocprofv3 output (gpurun_out/r03prof/) into the tracked summaries:
  <dst>/rocprofv3_kernel_stats_bench.csv       the --stats table of the bench command, as rocprofv3 wrote it
  <dst>/rocprofv3_kernels_by_grid.json         per (kernel, grid size): launches, average duration from the kernel trace,
                                               counters per launch, corrected HBM bytes, clock and issue-slot occupancy
  <dst>/pmc_kernels.json                       what bench.py attaches to each config's roofline / valu objects (copied to
                                               profiles/pmc_kernels.json when the round's profile is committed)
Kernels are keyed by (name, grid) because the bench launches the same kernel at several batch sizes in one process.  The
inline namespace of the base field (vrf::f_bls381fr:: ...) is dropped from the names: the suite tag tells the field.

Issue-slot occupancy instead of a nominal-clock "fraction of peak": GRBM_GUI_ACTIVE sums the busy
cycles of the 8 XCDs, so cycles = GRBM_GUI_ACTIVE / 8 and clock = cycles / duration; a SIMD starts one wave-instruction
per 4 cycles for the 64-bit / VOP3 class (v_mad_u64_u32: what these kernels are made of) and per 2 cycles for plain
32-bit VOP1/VOP2 (profiles/r01_instr_rate_microbench.jsonl), so with 256 CU x 4 SIMD
    issue_slot_frac  = SQ_INSTS_VALU * 4 / (cycles * 1024)                          every VALU instruction priced at 4 cycles
    issue_cycle_frac = (INT64 * 4 + (SQ_INSTS_VALU - INT64) * 2) / (cycles * 1024)  the mixed-rate lower bound
The truth lies between the two (not every non-INT64 instruction dual-issues).
usage: python tools/summarize_profiles3.py gpurun_out/r03prof profiles/r03"""
import collections, csv, glob, json, os, re, shutil, sys

src, dst = sys.argv[1], sys.argv[2]
ROUND = sys.argv[3] if len(sys.argv) > 3 else "r03"
os.makedirs(dst, exist_ok=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ark_ec_vrfs_amd._lib import source_stamp, LIB_PATH  # noqa: E402  (what the counters were measured on)
import hashlib  # noqa: E402
STAMP = {"source_sha256": source_stamp(), "lib_sha256": hashlib.sha256(open(LIB_PATH, "rb").read()).hexdigest(),
         "round": ROUND, "note": "sha256 of ark_ec_vrfs_amd/csrc sources (ark_ec_vrfs_amd._lib.source_stamp) and of libvrfhip.so "
                                 "at profiling time; bench.py attaches these counters only while the source stamp matches"}
VALU_PEAK = 256 * 4 * 16 * 2.4e9


def find(sub, suffix):
    hits = glob.glob(os.path.join(src, sub, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


def short(name):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "").split("(")[0]
    return re.sub(r"f_(bls381fr|25519|bn254fr|p256)::", "", name)


stats = find("stats", "kernel_stats.csv")
if stats:
    shutil.copy(stats, os.path.join(dst, "rocprofv3_kernel_stats_bench.csv"))
table = collections.OrderedDict()
trace = find("stats", "kernel_trace.csv")
dur = collections.defaultdict(list)
if trace:
    for r in csv.DictReader(open(trace)):
        name = short(r["Kernel_Name"])
        if name.startswith("vrf::"):
            grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])       # work-items, as Grid_Size in the counter files
            dur[(name, grid)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for key, v in dur.items():
    table[key] = collections.OrderedDict(kernel=key[0], grid=key[1], launches=len(v), avg_duration_ns=sum(v) / len(v),
                                         min_duration_ns=min(v))
for sub in ("sq", "fetch", "write", "mix"):
    path = find(sub, "counter_collection.csv")
    if not path:
        continue
    acc = collections.defaultdict(list)
    disp = {}
    for r in csv.DictReader(open(path)):
        name = short(r["Kernel_Name"])
        if not name.startswith("vrf::"):
            continue
        key = (name, int(r["Grid_Size"]))
        acc[(key, r["Counter_Name"])].append(float(r["Counter_Value"]))
        disp[key] = {k: r[k] for k in ("Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count") if k in r}
    for (key, ctr), vals in acc.items():
        e = table.setdefault(key, collections.OrderedDict(kernel=key[0], grid=key[1]))
        e["dispatch"] = disp[key]
        e[ctr] = sum(vals) / len(vals)
for e in table.values():
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        # MI355X_MICROARCH.md, HBM section: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies the 128-B
        # requests of 16 B/lane loads at 64 B -> x2 for these kernels (every global load here is a dwordx4)
        e["hbm_bytes_per_launch"] = e["FETCH_SIZE"] * 2 * 1024 + e["WRITE_SIZE"] * 1024
    if "SQ_INSTS_VALU" in e and "avg_duration_ns" in e:
        e["valu_lane_instructions_per_launch"] = e["SQ_INSTS_VALU"] * 64
        e["valu_frac_of_peak"] = e["SQ_INSTS_VALU"] * 64 / (e["avg_duration_ns"] * 1e-9) / VALU_PEAK
    if "GRBM_GUI_ACTIVE" in e and "avg_duration_ns" in e and "SQ_INSTS_VALU" in e:
        cycles = e["GRBM_GUI_ACTIVE"] / 8.0
        e["clock_ghz_observed"] = cycles / e["avg_duration_ns"]
        e["issue_slot_frac"] = e["SQ_INSTS_VALU"] * 4.0 / (cycles * 1024.0) if cycles else None
        if "SQ_INSTS_VALU_INT64" in e and cycles:
            n64 = e["SQ_INSTS_VALU_INT64"]
            e["issue_cycle_frac"] = (n64 * 4.0 + max(e["SQ_INSTS_VALU"] - n64, 0.0) * 2.0) / (cycles * 1024.0)
rows = sorted(table.values(), key=lambda e: -e.get("avg_duration_ns", 0) * e.get("launches", 1))
json.dump({"command": "rocprofv3 {--kernel-trace --stats | --pmc <one group per pass> --kernel-trace} -- python3 bench.py --steps 3 "
                      "--warmup 1 --config-steps 2 --no-cpu-baseline  (tools/profile_round3.sh)",
           "passes": ["SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE", "FETCH_SIZE", "WRITE_SIZE",
                      "SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_INSTS_SALU SQ_WAVE_CYCLES"],
           "note": "averages per launch; valu_frac_of_peak = SQ_INSTS_VALU x 64 / duration / (256 CU x 4 SIMD x 16 lanes x 2.4 GHz); "
                   "clock_ghz_observed = GRBM_GUI_ACTIVE / 8 / duration; issue_slot_frac = SQ_INSTS_VALU x 4 / (cycles x 1024 SIMDs); "
                   "issue_cycle_frac prices non-INT64 instructions at 2 cycles",
           "_stamp": STAMP, "kernels": rows}, open(os.path.join(dst, "rocprofv3_kernels_by_grid.json"), "w"), indent=1)

# config -> (kernel substring, grid) of its dominant kernel
N20, N16, N14 = 1 << 20, 1 << 16, 1 << 14
CONFIGS = [("ietf_verify", "k_verify_straus<vrf::SuiteBS, 1>", N20, 20), ("ietf_prove", "k_prove_mul<vrf::SuiteBS, false>", 2 * N16, 16),
           ("ietf_prove_ed25519", "k_prove_mul<vrf::SuiteED, false>", 2 * N20, 20), ("ietf_verify_ed25519", "k_verify_decode<vrf::SuiteED, 2>", None, 20),
           ("ietf_prove_babyjubjub", "k_prove_mul<vrf::SuiteBJ, false>", 2 * N20, 20),
           ("ietf_verify_babyjubjub", "k_verify_straus<vrf::SuiteBJ, 1>", N20, 20),
           ("ietf_prove_secp256r1", "k_p256_prove_mul<0>", 4 * N20, 20), ("ietf_verify_secp256r1", "k_p256_verify_mul<1>", N20, 20),
           ("pedersen_prove_jubjub", "k_prove_mul<vrf::SuiteJJ, false>", 2 * N20, 20),
           ("pedersen_verify_jubjub", "k_ped_verify_straus<vrf::SuiteJJ, 0>", N20, 20),
           ("pedersen_rlc_jubjub", "k_rlc_decode<vrf::SuiteJJ, 2>", None, 20),
           ("ietf_prove_bandersnatch_sw", "k_prove_mul<vrf::SuiteBS, false>", 2 * N20, 20),
           ("ietf_verify_bandersnatch_sw", "k_bsw_verify_decode", None, 20),
           ("pairing_check", "k_pairing_check2_oct_lines", 8 * N14, 14), ("pairing_check_shared", "k_pairing_check2_oct_prepared", 8 * N14, 14)]
out = collections.OrderedDict()
for cfg, pat, grid, lg in CONFIGS:
    pat = pat.rstrip("(")
    cand = [e for e in rows if (e["kernel"].endswith(pat) or pat in e["kernel"]) and (grid is None or abs(e["grid"] - grid) <= 256)
            and not (pat.endswith(("quad", "oct")) and "prepared" in e["kernel"])]
    if not cand:
        continue
    e = max(cand, key=lambda x: x.get("launches", 0))
    out[cfg] = {"kernel": e["kernel"], "grid": e["grid"], "log2_batch": lg, "hbm_bytes_per_launch": e.get("hbm_bytes_per_launch"),
                "valu_lane_instructions_per_launch": e.get("valu_lane_instructions_per_launch"),
                "avg_duration_ns_kernel_trace": e.get("avg_duration_ns"), "valu_frac_of_peak": e.get("valu_frac_of_peak"),
                "clock_ghz_observed": e.get("clock_ghz_observed"), "issue_slot_frac": e.get("issue_slot_frac"),
                "issue_cycle_frac": e.get("issue_cycle_frac"),
                "scratch_bytes_per_lane": (e.get("dispatch") or {}).get("Scratch_Size"),
                "source": "profiles/%s/rocprofv3_kernels_by_grid.json (tools/profile_round4.sh)" % ROUND}
out["_stamp"] = STAMP
json.dump(out, open(os.path.join(dst, "pmc_kernels.json"), "w"), indent=1)
for e in rows[:60]:
    print("%-62s grid %-9d x%-3d %9.3f ms  hbm %.3g B  %.2f GHz  slot %.2f  cyc %.2f  scratch %s" % (
        e["kernel"][:62], e["grid"], e.get("launches", 0), e.get("avg_duration_ns", 0) / 1e6,
        e.get("hbm_bytes_per_launch", 0), e.get("clock_ghz_observed") or 0, e.get("issue_slot_frac") or 0,
        e.get("issue_cycle_frac") or 0, (e.get("dispatch") or {}).get("Scratch_Size")))
print(json.dumps(out, indent=1)[:1500])
