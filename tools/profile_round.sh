set -e
mkdir -p gpurun_out/r01b
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r01b/stats -o run -- $B > $R/gpurun_out/r01b/stats.log 2>&1; echo stats_ok
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/r01b/sq -o run -- $B > $R/gpurun_out/r01b/sq.log 2>&1; echo sq_ok
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r01b/fetch -o run -- $B > $R/gpurun_out/r01b/fetch.log 2>&1; echo fetch_ok
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r01b/write -o run -- $B > $R/gpurun_out/r01b/write.log 2>&1; echo write_ok
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $R/gpurun_out/r01b/tcc -o run -- $B > $R/gpurun_out/r01b/tcc.log 2>&1; echo tcc_ok
cd $R && python3 bench.py --steps 20 --secondary > gpurun_out/r01b/bench_secondary.json 2> gpurun_out/r01b/bench_secondary.err; tail -c 1500 gpurun_out/r01b/bench_secondary.json
