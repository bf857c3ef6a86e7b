"""Prints the two measurement tables of DESIGN.md section 4 from the committed round-4 records, so that the document quotes
the committed run, not a remembered one:
  profiles/r04/bench_n1_default.json            the default `python bench.py` line (one MI355X)
  profiles/r04/rocprofv3_kernels_by_grid.json   per-kernel counters of the profiled bench (tools/profile_round4.sh)
usage: python tools/design_tables.py  > /tmp/tables.md"""
import json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.loads(open(os.path.join(ROOT, "profiles/r04/bench_n1_default.json")).read().strip().splitlines()[-1])
c = d["configs"]
st = lambda x, keys: " · ".join("%s %.2f" % (k.replace("_", " "), x[k]) for k in keys)
cpu = lambda e: ("%.1fe4/s on %d threads" % (e["cpu_baseline"]["value"] / 1e4, e["cpu_baseline"]["cores"])) if e.get("cpu_baseline") else ""
rows = []
def row(name, e, unit, stages=None, div=1e7, suf="e7"):
    rows.append("| %s | %.2f%s %s | %.2f ms | %s | %s |" % (name, e["value"] / div, suf, unit, e["ms_per_step"],
                st(e["stage_ms_per_step"], stages) if stages and e.get("stage_ms_per_step") else "", cpu(e)))
row("configs[2] **IETF verify 2^20, Bandersnatch, wire format, checked** (headline)", d, "verifies/s", ["decode", "straus_v", "straus_u", "finish"])
row("… points declared pre-validated", d["prevalidated"], "verifies/s")
for lg in (19, 18, 17):
    row("… on a 2^%d shard (per-GPU share of a 2^20 batch at N = %d)" % (lg, 1 << (20 - lg)), c["ietf_verify_shard_2^%d" % lg], "verifies/s")
if "ietf_verify_from_alpha" in c:
    row("… from (pk, alpha, proof): hash-to-curve inside the call, H kept affine (`vrfhip_ietf_verify_batch_alpha`)", c["ietf_verify_from_alpha"], "verifies/s")
    row("… … the two-call form (hash_to_curve_batch, verify with the input pre-validated)", c["ietf_verify_from_alpha"]["two_calls"], "verifies/s")
row("… keyed (`vrfhip_keyset_create`, 1024 resident keys)", c["ietf_verify_keyed"], "verifies/s", ["decode", "straus_v", "comb_u", "finish"])
row("configs[1] IETF prove 2^16", c["ietf_prove"], "proofs/s", ["prepare", "mul", "finish"])
rows.append("| … prove 2^20 (produces the headline's inputs) | %.2fe7 proofs/s | %.1f ms | | |" % (d["proofs_per_sec"] / 1e7, (1 << 20) / d["proofs_per_sec"] * 1e3))
row("configs[3] Pedersen prove 2^20, JubJub", c["pedersen_prove_jubjub"], "proofs/s", ["tai_find+prepare", "mul", "finish"])
row("configs[3] Pedersen verify 2^20, JubJub, per proof, checked", c["pedersen_verify_jubjub"], "verifies/s", ["decode", "straus_a", "straus_b", "finish"])
row("… batched (digest + one MSM)", c["pedersen_verify_batched_jubjub"], "verifies/s", ["decode", "msm_buckets", "msm_final"])
row("f4 IETF prove 2^20, Ed25519", c["ietf_prove_ed25519"], "proofs/s", ["tai_find+prepare", "mul", "finish"])
row("f4 IETF verify 2^20, Ed25519, checked", c["ietf_verify_ed25519"], "verifies/s", ["decode", "straus_v", "straus_u", "finish"])
row("… pre-validated", c["ietf_verify_ed25519"]["prevalidated"], "verifies/s")
if "from_alpha" in c["ietf_verify_ed25519"]:
    row("… from (pk, alpha, proof), checked", c["ietf_verify_ed25519"]["from_alpha"], "verifies/s")
row("f4 IETF prove 2^20, Baby-JubJub", c["ietf_prove_babyjubjub"], "proofs/s", ["tai_find+prepare", "mul", "finish"])
row("f4 IETF verify 2^20, Baby-JubJub, checked", c["ietf_verify_babyjubjub"], "verifies/s", ["decode", "straus_v", "straus_u", "finish"])
row("… pre-validated", c["ietf_verify_babyjubjub"]["prevalidated"], "verifies/s")
if "from_alpha" in c["ietf_verify_babyjubjub"]:
    row("… from (pk, alpha, proof), checked", c["ietf_verify_babyjubjub"]["from_alpha"], "verifies/s")
row("f4 IETF prove 2^20, secp256r1 (RFC 9381 P256-SHA256-TAI)", c["ietf_prove_secp256r1"], "proofs/s", ["tai_find+prepare", "mul", "finish"])
row("f4 IETF verify 2^20, secp256r1, Sec1 wire format", c["ietf_verify_secp256r1"], "verifies/s", ["decode", "straus_v", "straus_u", "finish"])
if "from_alpha" in c["ietf_verify_secp256r1"]:
    row("… from (pk, alpha, proof)", c["ietf_verify_secp256r1"]["from_alpha"], "verifies/s")
if "affine_inputs" in c["ietf_verify_secp256r1"]:
    row("… pk, input, output as x ‖ y (typed callers)", c["ietf_verify_secp256r1"]["affine_inputs"], "verifies/s")
row("… Pedersen prove 2^20, secp256r1 (unpinned; placeholder blinding base)", c["pedersen_prove_secp256r1"], "proofs/s", ["tai_find+prepare", "mul", "finish"])
row("… Pedersen verify 2^20, secp256r1, per proof", c["pedersen_verify_secp256r1"], "verifies/s", ["decode", "eq_h", "eq_g", "finish"])
if "pedersen_verify_batched_secp256r1" in c:
    row("… batched (digest + one MSM over 5n + 2 points)", c["pedersen_verify_batched_secp256r1"], "verifies/s", ["decode", "msm_buckets", "msm_final"])
if "ietf_prove_bandersnatch_sw" in c:
    row("f4 IETF prove 2^20, bandersnatch_sw (33-byte SW points; Edwards arithmetic behind the codec)", c["ietf_prove_bandersnatch_sw"], "proofs/s", ["tai_find+prepare", "mul", "finish"])
    row("f4 IETF verify 2^20, bandersnatch_sw, checked", c["ietf_verify_bandersnatch_sw"], "verifies/s", ["decode", "straus_v", "straus_u", "finish"])
    if "from_alpha" in c["ietf_verify_bandersnatch_sw"]:
        row("… from (pk, alpha, proof)", c["ietf_verify_bandersnatch_sw"]["from_alpha"], "verifies/s")
    if "affine_inputs" in c["ietf_verify_bandersnatch_sw"]:
        row("… pk, input, output as Weierstrass x ‖ y (typed callers)", c["ietf_verify_bandersnatch_sw"]["affine_inputs"], "verifies/s")
    row("… Pedersen prove 2^20, bandersnatch_sw (placeholder blinding base)", c["pedersen_prove_bandersnatch_sw"], "proofs/s", ["tai_find+prepare", "mul", "finish"])
    row("… Pedersen verify 2^20, bandersnatch_sw, per proof", c["pedersen_verify_bandersnatch_sw"], "verifies/s", ["decode", "eq_h", "eq_g", "finish"])
    row("… batched (digest + one MSM over 5n + 2 points)", c["pedersen_verify_batched_bandersnatch_sw"], "verifies/s", ["decode", "msm_buckets", "msm_final"])
row("configs[4] pairing check 2^14, per item", c["pairing_check"], "checks/s", None, 1e6, "e6")
row("… shared G2 pair (prepared lines)", c["pairing_check_shared_g2"], "checks/s", None, 1e6, "e6")
row("… shared G2 pair, ONE batch (two G1 MSMs + one pairing), 2^14", c["pairing_check_batched_shared_g2_2^14"], "checks/s", ["prep", "msm_buckets", "msm_final", "pairing"], 1e6, "e6")
row("… the same at 2^18", c["pairing_check_batched_shared_g2_2^18"], "checks/s", ["prep", "msm_buckets", "msm_final", "pairing"])
row("… 2^18, four batches in flight (4 contexts, 4 streams; a step is four batches)", c["pairing_check_batched_shared_g2_2^18_four_in_flight"], "checks/s")
print("| config | result | step | stages (ms) | CPU restatement, same run |\n|---|---|---|---|---|\n" + "\n".join(rows))
prof = json.load(open(os.path.join(ROOT, "profiles/r04/rocprofv3_kernels_by_grid.json")))
def find(name, grid):
    best = None
    for v in prof["kernels"]:
        if v["kernel"] == name and (grid is None or v["grid"] == grid):
            if best is None or v.get("avg_duration_ns", 0) > best.get("avg_duration_ns", 0):
                best = v
    return best
want = [("vrf::k_verify_decode<vrf::SuiteBS, 2>", 524288, " (checked and pre-validated launches averaged)"), ("vrf::k_verify_straus<vrf::SuiteBS, 1>", 1048576, ""), ("vrf::k_verify_straus<vrf::SuiteBS, 0>", 1048576, ""), ("vrf::k_verify_finish<vrf::SuiteBS, 2>", 524288, ""),
        ("vrf::k_verify_input_from_alpha<vrf::SuiteBS>", 1048576, " (hash-to-curve + H's tables)"), ("vrf::k_verify_decode_skip_h<vrf::SuiteBS, 2>", 524288, " (pk, Γ)"),
        ("vrf::k_verify_decode_keyed<vrf::SuiteBS, 2>", 524288, ""), ("vrf::k_verify_comb_u<vrf::SuiteBS>", 1048576, ""),
        ("vrf::k_prove_mul<vrf::SuiteBS, false>", 131072, " (2^16)"), ("vrf::k_prove_prepare_multi<vrf::SuiteBS, 2>", 131072, " (2^20)"), ("vrf::k_prove_mul<vrf::SuiteBS, false>", 2097152, " (2^20)"),
        ("vrf::k_prove_finish<vrf::SuiteBS, 2>", 131072, " (2^20)"),
        ("vrf::k_tai_find<vrf::SuiteJJ>", 262144, ""), ("vrf::k_prove_prepare<vrf::SuiteJJ, 2>", 1048576, ""), ("vrf::k_prove_mul<vrf::SuiteJJ, false>", 2097152, ""),
        ("vrf::k_ped_verify_decode<vrf::SuiteJJ, 2>", 1048576, ""), ("vrf::k_ped_verify_straus<vrf::SuiteJJ, 0>", 1048576, ""), ("vrf::k_ped_verify_straus<vrf::SuiteJJ, 1>", 1048576, ""),
        ("vrf::k_rlc_decode<vrf::SuiteJJ, 2>", 524288, ""), ("vrf::k_msm_buckets<vrf::SuiteJJ>", None, ""), ("vrf::k_digest_leaves", 1048576, " (2^20 × 225 B)"),
        ("vrf::k_verify_decode<vrf::SuiteED, 2>", 524288, " (checked and pre-validated averaged)"), ("vrf::k_verify_straus<vrf::SuiteED, 1>", 1048576, ""), ("vrf::k_verify_straus<vrf::SuiteED, 0>", 1048576, ""),
        ("vrf::k_prove_prepare<vrf::SuiteED, 2>", 1048576, ""), ("vrf::k_prove_mul<vrf::SuiteED, false>", 2097152, ""),
        ("vrf::k_verify_decode<vrf::SuiteBJ, 2>", 524288, " (averaged)"), ("vrf::k_verify_straus<vrf::SuiteBJ, 1>", 1048576, ""), ("vrf::k_verify_straus<vrf::SuiteBJ, 0>", 1048576, ""),
        ("vrf::k_prove_prepare<vrf::SuiteBJ, 2>", 1048576, ""), ("vrf::k_prove_mul<vrf::SuiteBJ, false>", 2097152, ""),
        ("vrf::k_p256_verify_decode", 1048576, ""), ("vrf::k_p256_verify_mul<1>", 1048576, " (V = sH − cΓ)"), ("vrf::k_p256_verify_mul<0>", 1048576, " (U = sG − cY)"),
        ("vrf::k_p256_verify_finish", 1048576, ""), ("vrf::k_p256_tai_find", 262144, " (work queue, 4096 persistent waves)"), ("vrf::k_p256_prove_prepare<0>", 1048576, ""), ("vrf::k_p256_prove_tables", 1048576, " (H, 2^64 H, 2^128 H, 2^192 H)"), ("vrf::k_p256_prove_mul<0>", 4194304, " (4 ladders per proof)"),
        ("vrf::k_p256_prove_finish<0>", 1048576, ""), ("vrf::k_p256_prove_mul<1>", 4194304, " (Pedersen)"),
        ("vrf::k_p256_ped_verify_decode", 1048576, ""), ("vrf::k_p256_ped_verify_mul<0>", 1048576, " (sH − cΓ − Ok = O)"),
        ("vrf::k_p256_ped_verify_mul<1>", 1048576, " (sG + sbB − c·pk_com − R = O)"),
        ("vrf::k_pm_rlc_decode", 1048576, " (secp256r1: five decompressions, challenge, weights, digits)"), ("vrf::k_pm_buckets", None, " (secp256r1, 5·2^20 + 2 points)"),
        ("vrf::k_tai_find<vrf::SuiteBW>", 262144, " (bandersnatch_sw: 8.8 expected attempts per input)"), ("vrf::k_bsw_prove_prepare", 1048576, ""),
        ("vrf::k_bsw_prove_finish", 1048576, ""), ("vrf::k_bsw_verify_decode", 1048576, " (3 x: sqrt, te_sw_map, 2-descent, GLV tables)"),
        ("vrf::k_bsw_verify_straus<1>", 1048576, " (= verify_straus_item<BS, 1>)"), ("vrf::k_bsw_verify_straus<0>", 1048576, ""), ("vrf::k_bsw_verify_finish", 1048576, ""),
        ("vrf::k_bsw_ped_verify_decode", 1048576, ""), ("vrf::k_bsw_rlc_decode", 1048576, " (in front of k_msm_buckets<BS>)"),
        ("vrf::k_pairing_lines_oct", 131072, " (2^14 items x 8 lanes: G2 walk, lines -> HBM)"), ("vrf::k_pairing_check2_oct_lines", 131072, " (Miller loop over them + final exponentiation)"),
        ("vrf::k_pairing_check2_oct_prepared", 131072, " (shared G2 pair)"), ("vrf::k_pairing_check2_row_prepared<true>", 64, " (ONE item: 48 lanes)"),
        ("vrf::k_g1_buckets", 119808, " (2^18 × 2 sets)"), ("vrf::k_g1_final", 256, " (2 sets × 128 lanes)")]
kr = []
for name, grid, note in want:
    v = find(name, grid)
    if not v:
        print("MISSING", name, grid); continue
    lat = (v.get("issue_slot_frac") or 0) < 0.02
    kr.append("| `%s`%s | %d | %.2f | %s | %s | %s | %s / %s / %s |" % (name.replace("vrf::", "").replace("Suite", ""), note, v["grid"], v["avg_duration_ns"] / 1e6,
              "–" if lat else "%.2f" % v["clock_ghz_observed"],
              "latency" if lat else "%.2f / %.2f" % (v["issue_slot_frac"], v.get("issue_cycle_frac") or 0), "–" if lat else "%.1f" % ((v.get("hbm_bytes_per_launch") or 0) / 1e9),
              v["dispatch"]["VGPR_Count"], v["dispatch"]["LDS_Block_Size"], v["dispatch"]["Scratch_Size"]))
print()
print("| kernel | grid | ms | GHz | issue slots (all 4-cycle / mixed) | HBM GB | VGPR / LDS / scratch |\n|---|---|---|---|---|---|---|\n" + "\n".join(kr))
