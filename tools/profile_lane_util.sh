# Active lanes per vector instruction of the try-and-increment search kernels (k_tai_find<*>, k_p256_tai_find):
# SQ_THREAD_CYCLES_VALU / (SQ_INSTS_VALU x 4 cycles x 64 lanes) over tools/gpu_h2c_only.py.  usage (GPU box): bash tools/profile_lane_util.sh
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/lane_util
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O -o run -- python3 $R/tools/gpu_h2c_only.py > $O/run.log 2>&1; echo pmc_ok
python3 - <<PY
import csv, collections, glob
f = glob.glob("$O/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "tai_find" in k or "hash_to_curve" in k:
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(acc.items()):
    insts, thr, act = v.get("SQ_INSTS_VALU", 0), v.get("SQ_THREAD_CYCLES_VALU", 0), v.get("SQ_ACTIVE_INST_VALU", 0)
    print("%-70s SQ_INSTS_VALU %.3e  THREAD_CYCLES_VALU %.3e  ACTIVE_INST_VALU %.3e  lanes/instr = %.1f of 64" % (
        k[-70:], insts, thr, act, thr / (insts * 4) if insts else 0))
PY
