"""Rewrites the two measurement tables of DESIGN.md section 4 from profiles/r02/bench_n1_default.json and
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
want = [("vrf::k_verify_decode<vrf::SuiteBS, 2>", 131072, ""), ("vrf::k_verify_straus<vrf::SuiteBS, 1>", 1048576, ""), ("vrf::k_verify_straus<vrf::SuiteBS, 0>", 1048576, ""), ("vrf::k_verify_finish<vrf::SuiteBS, 2>", 131072, ""),
        ("vrf::k_verify_decode_keyed<vrf::SuiteBS, 2>", 131072, ""), ("vrf::k_verify_comb_u<vrf::SuiteBS>", 1048576, ""),
        ("vrf::k_prove_prepare_multi<vrf::SuiteBS, 1>", 65536, " (2^16)"), ("vrf::k_prove_mul<vrf::SuiteBS>", 131072, " (2^16)"), ("vrf::k_prove_prepare_multi<vrf::SuiteBS, 2>", 131072, " (2^20)"), ("vrf::k_prove_mul<vrf::SuiteBS>", 2097152, " (2^20)"),
        ("vrf::k_prove_finish<vrf::SuiteBS, 2>", 131072, " (2^20)"),
        ("vrf::k_tai_find<vrf::SuiteJJ>", 262144, ""), ("vrf::k_prove_prepare<vrf::SuiteJJ, 2>", 1048576, ""), ("vrf::k_prove_mul<vrf::SuiteJJ>", 2097152, ""),
        ("vrf::k_ped_verify_decode<vrf::SuiteJJ, 2>", 1048576, ""), ("vrf::k_ped_verify_straus<vrf::SuiteJJ, 0>", 1048576, ""), ("vrf::k_ped_verify_straus<vrf::SuiteJJ, 1>", 1048576, ""),
        ("vrf::k_rlc_decode<vrf::SuiteJJ, 2>", 131072, ""), ("vrf::k_msm_buckets<vrf::SuiteJJ>", 261632, ""), ("vrf::k_digest_leaves", 1048576, " (2^20 × 225 B)"),
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
open(p, "w").write(s)
print("tables rewritten: %d config rows, %d kernel rows" % (len(rows), len(kr)))
