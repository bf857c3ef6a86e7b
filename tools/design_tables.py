"""Rewrites the two measurement tables of DESIGN.md section 4 (and the sentences that quote them) from profiles/r02/bench_n1_default.json and
profiles/r02/rocprofv3_kernels_by_grid.json (so the document quotes the committed run, not a remembered one).
usage: python tools/design_tables.py"""
import json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.loads(open(os.path.join(ROOT, "profiles/r02/bench_n1_default.json")).read().strip().splitlines()[-1])
c = d["configs"]
st = lambda x, keys: " · ".join("%s %.2f" % (k.replace("_", " "), x[k]) for k in keys)
rows = []
rows.append("| configs[2] **IETF verify 2^20, Bandersnatch, wire format, checked** (headline) | **%.2fe7 verifies/s** | %.1f ms | %s |" % (d["value"] / 1e7, d["ms_per_step"], st(d["stage_ms_per_step"], ["decode", "straus_v", "straus_u", "finish"])))
rows.append("| … the same, points declared pre-validated (round 1's operation: 2.85e7) | %.2fe7 verifies/s | %.1f ms | |" % (d["prevalidated"]["value"] / 1e7, d["prevalidated"]["ms_per_step"]))
k = c["ietf_verify_keyed"]; rows.append("| … keyed (`vrfhip_keyset_create`, 1024 resident keys) | %.2fe7 verifies/s | %.1f ms | %s |" % (k["value"] / 1e7, k["ms_per_step"], st(k["stage_ms_per_step"], ["decode", "straus_v", "comb_u", "finish"])))
k = c["ietf_prove"]; rows.append("| configs[1] IETF prove 2^16 | %.2fe7 proofs/s | %.2f ms | %s |" % (k["value"] / 1e7, k["ms_per_step"], st(k["stage_ms_per_step"], ["prepare", "mul", "finish"])))
rows.append("| … prove 2^20 (produces the headline's inputs) | %.2fe7 proofs/s | %.1f ms | kernel table below |" % (d["proofs_per_sec"] / 1e7, (1 << 20) / d["proofs_per_sec"] * 1e3))
k = c["pedersen_prove_jubjub"]; rows.append("| configs[3] Pedersen prove 2^20, JubJub | %.2fe7 proofs/s | %.1f ms | %s |" % (k["value"] / 1e7, k["ms_per_step"], st(k["stage_ms_per_step"], ["tai_find+prepare", "mul", "finish"])))
k = c["pedersen_verify_jubjub"]; rows.append("| configs[3] Pedersen verify 2^20, JubJub, per proof, checked | %.2fe7 verifies/s | %.1f ms | %s |" % (k["value"] / 1e7, k["ms_per_step"], st(k["stage_ms_per_step"], ["decode", "straus_a", "straus_b", "finish"])))
k = c["pedersen_verify_batched_jubjub"]; rows.append("| … batched (digest + one MSM) | %.2fe7 verifies/s | %.1f ms | digest 1.0 · %s |" % (k["value"] / 1e7, k["ms_per_step"], st(k["stage_ms_per_step"], ["decode", "msm_buckets", "msm_final"])))
k = c["pairing_check"]; rows.append("| configs[4] pairing check 2^14, per item | %.2fe6 checks/s | %.2f ms | one kernel (one item per quad) |" % (k["value"] / 1e6, k["ms_per_step"]))
k = c["pairing_check_shared_g2"]; rows.append("| … shared G2 pair (prepared lines) | %.2fe6 checks/s | %.2f ms | one kernel |" % (k["value"] / 1e6, k["ms_per_step"]))
k = c["pairing_check_batched_shared_g2_2^14"]; rows.append("| … shared G2 pair, ONE batch (two G1 MSMs + one pairing on one wave), 2^14 | **%.2fe6 checks/s** | %.2f ms | %s |" % (k["value"] / 1e6, k["ms_per_step"], st(k["stage_ms_per_step"], ["prep", "msm_buckets", "msm_final", "pairing"])))
k = c["pairing_check_batched_shared_g2_2^18"]; rows.append("| … the same at 2^18 | %.2fe7 checks/s | %.2f ms | %s |" % (k["value"] / 1e7, k["ms_per_step"], st(k["stage_ms_per_step"], ["prep", "msm_buckets", "msm_final", "pairing"])))
k = c["pairing_check_batched_shared_g2_2^18_four_in_flight"]; rows.append("| … 2^18, four batches in flight (4 contexts, 4 streams) | %.2fe7 checks/s | %.1f ms / 4 batches | |" % (k["value"] / 1e7, k["ms_per_step"]))
cfg = "| config | result | step | stages (ms) |\n|---|---|---|---|\n" + "\n".join(rows)
prof = json.load(open(os.path.join(ROOT, "profiles/r02/rocprofv3_kernels_by_grid.json")))
def find(name, grid):
    for v in prof["kernels"]:
        if v["kernel"].startswith(name) and v["grid"] == grid:
            return v
want = [("vrf::k_verify_decode<vrf::SuiteBS, 2>", 524288, ""), ("vrf::k_verify_straus<vrf::SuiteBS, 1>", 1048576, ""), ("vrf::k_verify_straus<vrf::SuiteBS, 0>", 1048576, ""), ("vrf::k_verify_finish<vrf::SuiteBS, 2>", 524288, ""),
        ("vrf::k_verify_decode_keyed<vrf::SuiteBS, 2>", 524288, ""), ("vrf::k_verify_comb_u<vrf::SuiteBS>", 1048576, ""),
        ("vrf::k_prove_prepare_multi<vrf::SuiteBS, 1>", 65536, " (2^16)"), ("vrf::k_prove_mul<vrf::SuiteBS>", 131072, " (2^16)"), ("vrf::k_prove_prepare_multi<vrf::SuiteBS, 2>", 131072, " (2^20)"), ("vrf::k_prove_mul<vrf::SuiteBS>", 2097152, " (2^20)"),
        ("vrf::k_prove_finish<vrf::SuiteBS, 2>", 131072, " (2^20)"),
        ("vrf::k_tai_find<vrf::SuiteJJ>", 262144, ""), ("vrf::k_prove_prepare<vrf::SuiteJJ, 2>", 1048576, ""), ("vrf::k_prove_mul<vrf::SuiteJJ>", 2097152, ""),
        ("vrf::k_ped_verify_decode<vrf::SuiteJJ, 2>", 1048576, ""), ("vrf::k_ped_verify_straus<vrf::SuiteJJ, 0>", 1048576, ""), ("vrf::k_ped_verify_straus<vrf::SuiteJJ, 1>", 1048576, ""),
        ("vrf::k_rlc_decode<vrf::SuiteJJ, 2>", 524288, ""), ("vrf::k_msm_buckets<vrf::SuiteJJ>", 261632, ""), ("vrf::k_digest_leaves", 1048576, " (2^20 × 225 B)"),
        ("vrf::k_pairing_check2_quad", 65536, ""), ("vrf::k_pairing_check2_quad_prepared", 65536, ""), ("void vrf::k_pairing_check2_row_prepared<true>", 64, " (ONE item: 48 lanes)"),
        ("vrf::k_g1_buckets", 119808, " (2^18 × 2 sets)"), ("vrf::k_g1_final", 256, " (2 sets × 128 lanes)")]
kr = []
for name, grid, note in want:
    v = find(name, grid) or find(name.replace("void ", ""), grid) or find("void " + name, grid)
    if not v:
        print("MISSING", name, grid); continue
    lat = (v.get("valu_frac_of_peak") or 0) < 0.01
    kr.append("| `%s`%s | %d | %.2f | %s | %s | %s / %s / %s |" % (name.replace("void ", "").replace("vrf::", "").replace("Suite", ""), note, grid, v["avg_duration_ns"] / 1e6,
              "latency" if lat else "%.2f" % v["valu_frac_of_peak"], "–" if lat else "%.1f" % ((v.get("hbm_bytes_per_launch") or 0) / 1e9),
              v["dispatch"]["VGPR_Count"], v["dispatch"]["LDS_Block_Size"], v["dispatch"]["Scratch_Size"]))
ktab = "| kernel | grid | ms | VALU | HBM GB | VGPR / LDS / scratch |\n|---|---|---|---|---|---|\n" + "\n".join(kr)
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
a = s.index("| config | result | step | stages (ms) |"); b = s.index("\n\nPer kernel (`profiles/r02/rocprofv3_kernels_by_grid.json`")
s = s[:a] + cfg + s[b:]
a = s.index("| kernel | grid | ms | VALU | HBM GB | VGPR / LDS / scratch |"); b = s.index("\n\n(The profiled runs are a few per cent slower")
s = s[:a] + ktab + s[b:]
# the sentences of section 4 that quote the same run
import re
f = lambda x, n=2: ("%." + str(n) + "f") % x
s = re.sub(r"(\*\*Checked decode costs 10 %\*\*: )[0-9.]+(e7 verifies/s \()[0-9.]+( ms per 2\^20 in the committed run; 39\.5–)[0-9.]+( ms across boxes\) against\n)[0-9.]+(e7 \()[0-9.]+( ms\) with the)",
           lambda m: m.group(1) + f(d["value"] / 1e7) + m.group(2) + f(d["ms_per_step"], 1) + m.group(3) + m.group(0).split("39.5–")[1].split(" ms")[0] + m.group(4) +
           f(d["prevalidated"]["value"] / 1e7) + m.group(5) + f(d["prevalidated"]["ms_per_step"], 1) + m.group(6), s)
s = re.sub(r"(161 B × 2\^20 = 169 MB per launch ÷ )[0-9.]+( ms = )[0-9.]+( GB/s: \*\*frac )[0-9.]+(\*\* of 8 TB/s\.)",
           lambda m: m.group(1) + f(d["roofline"]["avg_launch_ms"], 1) + m.group(2) + f(d["roofline"]["achieved"], 1) + m.group(3) + f(d["roofline"]["frac"], 4) + m.group(4), s)
s = re.sub(r"(7\.5e9 wave-instructions in )[0-9.]+( ms = \*\*)[0-9.]+( of the VALU ceiling\*\*)",
           lambda m: m.group(1) + f(d["roofline"]["avg_launch_ms"], 1) + m.group(2) + f(d["valu"]["frac"]) + m.group(3), s)
b14, b18, b4 = c["pairing_check_batched_shared_g2_2^14"], c["pairing_check_batched_shared_g2_2^18"], c["pairing_check_batched_shared_g2_2^18_four_in_flight"]
sg = b14["stage_ms_per_step"]
s = re.sub(r"Measured: 2\^14 checks [0-9.]+ ms \(digest \+ prep [0-9.]+, buckets [0-9.]+, final [0-9.]+, the one pairing [0-9.]+\) = \*\*[0-9.]+e6 checks/s",
           lambda m: "Measured: 2^14 checks " + f(b14["ms_per_step"]) + " ms (digest + prep " + f(sg["prep"]) + ", buckets " + f(sg["msm_buckets"]) + ", final " + f(sg["msm_final"]) +
           ", the one pairing " + f(sg["pairing"]) + ") = **" + f(b14["value"] / 1e6) + "e6 checks/s", s)
s = re.sub(r"2\^18 checks [0-9.]+ ms = \*\*[0-9.]+e7 checks/s, 20× the", lambda m: "2^18 checks " + f(b18["ms_per_step"]) + " ms = **" + f(b18["value"] / 1e7) + "e7 checks/s, 20× the", s)
s = re.sub(r"(overlaps one batch's tail with the others' MSMs: )[0-9.]+(e7 checks/s with)", lambda m: m.group(1) + f(b4["value"] / 1e7, 1) + m.group(2), s)
s = re.sub(r"(the same 2\^14 checks now take )[0-9.]+( ms as one batch)", lambda m: m.group(1) + f(b14["ms_per_step"], 1) + m.group(2), s)
open(p, "w").write(s)
print("tables and quoted numbers rewritten: %d config rows, %d kernel rows" % (len(rows), len(kr)))
