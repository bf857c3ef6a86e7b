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
for T in "rlc tools/gpu_rlc_prof.py 20" "pairing tools/gpu_pairing_prof.py 16" "msm tools/gpu_msm_prof.py 20"; do
  set -- $T
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r01b/$1_stats -o run -- python3 $R/$2 $3 > $R/gpurun_out/r01b/$1_stats.log 2>&1; echo $1_stats_ok
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/r01b/$1_sq -o run -- python3 $R/$2 $3 > $R/gpurun_out/r01b/$1_sq.log 2>&1; echo $1_sq_ok
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r01b/$1_fetch -o run -- python3 $R/$2 $3 > $R/gpurun_out/r01b/$1_fetch.log 2>&1; echo $1_fetch_ok
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r01b/$1_write -o run -- python3 $R/$2 $3 > $R/gpurun_out/r01b/$1_write.log 2>&1; echo $1_write_ok
done
cd $R && for tool in gpu_msm_time gpu_rlc_time gpu_pairing_time gpu_keyed_time gpu_validate_time gpu_jj_prove_time; do timeout -k 10 300 python3 tools/$tool.py > gpurun_out/r01b/$tool.log 2>/dev/null; echo $tool done; done
cd $R && python3 bench.py --steps 20 --secondary > gpurun_out/r01b/bench_secondary.json 2> gpurun_out/r01b/bench_secondary.err; tail -c 1500 gpurun_out/r01b/bench_secondary.json
