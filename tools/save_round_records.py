"""Copies what a GPU run left under gpurun_out/ into the tracked profiles/ tree (round 4):
  gpurun_out/r04/{rocprofv3_kernel_stats_bench.csv, rocprofv3_kernels_by_grid.json, pmc_kernels.json} -> profiles/r04/ (and
  pmc_kernels.json also to profiles/, where bench.py looks for it)
  the bench's configs file + final line -> profiles/r04/bench_n1_default.json (ONE object: the headline record with its
  "configs", the format of rounds 2-3) and profiles/r04/bench_n1_final_line.json (the short line the driver parses)
usage: python tools/save_round_records.py gpurun_out/<bench stdout log> gpurun_out/<bench configs json>"""
import json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
log, cfg = sys.argv[1], sys.argv[2]
dst = os.path.join(ROOT, "profiles", "r04")
os.makedirs(dst, exist_ok=True)
for f in ("rocprofv3_kernel_stats_bench.csv", "rocprofv3_kernels_by_grid.json", "pmc_kernels.json"):
    shutil.copy(os.path.join(ROOT, "gpurun_out", "r04", f), os.path.join(dst, f))
shutil.copy(os.path.join(dst, "pmc_kernels.json"), os.path.join(ROOT, "profiles", "pmc_kernels.json"))
last = open(log).read().strip().splitlines()[-1]
line = json.loads(last)
c = json.load(open(cfg))
full = dict(c["headline"], configs=c["configs"])
assert abs(full["value"] / line["value"] - 1) < 1e-4, "configs file and final line come from different runs"
open(os.path.join(dst, "bench_n1_default.json"), "w").write(json.dumps(full) + "\n")
open(os.path.join(dst, "bench_n1_final_line.json"), "w").write(last + "\n")
print("saved: headline %.4g %s, %.2f ms per step, %d legs, final line %d bytes" % (full["value"], full["unit"], full["ms_per_step"],
      len(c["configs"]), len(last)))
