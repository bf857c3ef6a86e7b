# Instruction mix / stall counters of the bench kernels (three SQ passes); summaries -> gpurun_out/sqmix/
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/sqmix
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $R/gpurun_out/sqmix/a -o run -- $B > $R/gpurun_out/sqmix/a.log 2>&1; echo a_ok
timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/sqmix/b -o run -- $B > $R/gpurun_out/sqmix/b.log 2>&1; echo b_ok
timeout -k 10 200 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 --kernel-trace --output-format csv -d $R/gpurun_out/sqmix/c -o run -- $B > $R/gpurun_out/sqmix/c.log 2>&1; echo c_ok
