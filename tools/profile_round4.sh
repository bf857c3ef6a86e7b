# Round-4 profiling: rocprofv3 passes over ONE command -- the default bench (headline + every BASELINE config and the
# Ed25519 / Baby-JubJub / secp256r1 legs in the same process).  Kernel trace + stats first, then one --pmc group per pass (the guide's
# HBM recipe: FETCH_SIZE and WRITE_SIZE in separate passes).  Raw output under gpurun_out/r04prof/;
# tools/summarize_profiles3.py condenses it into profiles/r04/ and profiles/pmc_kernels.json.
# usage (on the GPU box): bash tools/profile_round4.sh
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 3 --warmup 1 --config-steps 2 --no-cpu-baseline"
timeout -k 10 560 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- $B > $O/stats.log 2>&1; echo stats_ok
timeout -k 10 560 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq -o run -- $B > $O/sq.log 2>&1; echo sq_ok
timeout -k 10 560 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o run -- $B > $O/fetch.log 2>&1; echo fetch_ok
timeout -k 10 560 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o run -- $B > $O/write.log 2>&1; echo write_ok
timeout -k 10 560 rocprofv3 --pmc SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_INSTS_SALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/mix -o run -- $B > $O/mix.log 2>&1; echo mix_ok
# keep what travels back small: the raw per-dispatch csv files are condensed on the box
python3 $R/tools/summarize_profiles3.py $O $R/gpurun_out/r04 r04 > $O/summary.log 2>&1; echo summary_ok
find $O -name "*.csv" -size +8M -delete
ls $R/gpurun_out/r04
