# Active lanes per vector instruction of EVERY kernel of the default bench command:
# SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU (64 = no lane ever idle).  usage (GPU box): bash tools/profile_lane_util_bench.sh
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/lane_util_bench
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 560 rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $O -o run -- python3 $R/bench.py --steps 2 --warmup 1 --config-steps 1 --no-cpu-baseline > $O/run.log 2>&1; echo pmc_ok
python3 - <<PY
import csv, collections, glob, re
f = glob.glob("$O/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    k = re.sub(r"^void ", "", r["Kernel_Name"]).replace("(anonymous namespace)::", "").split("(")[0]
    k = re.sub(r"f_(bls381fr|25519|bn254fr|p256)::", "", k).replace("vrf::", "")
    acc[(k, int(r["Grid_Size"]))][r["Counter_Name"]] += float(r["Counter_Value"])
rows = []
for (k, g), v in acc.items():
    insts, thr = v.get("SQ_INSTS_VALU", 0), v.get("SQ_THREAD_CYCLES_VALU", 0)
    if insts > 1e6:
        rows.append((insts, k, g, thr / insts))
for insts, k, g, l in sorted(rows, reverse=True):
    print("%-64s grid %-9d SQ_INSTS_VALU %.3e  lanes/instr %.1f" % (k[:64], g, insts, l))
PY
