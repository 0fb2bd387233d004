# Run ON THE GPU BOX at the end of round 3: every record of profiles/r03_* from the final build.
#   gpurun --timeout 1200 -- bash tools/profile_r03_final.sh
set -e -o pipefail
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT
bash $ROOT/tools/profile_round.sh r03 2>&1 | grep -E "^==|done"
f=$(find $OUT/r03_prof_default -name "*kernel_stats.csv" | head -1); test -n "$f"; cp "$f" $OUT/r03_default_bench_kernel_stats.csv
find $OUT/r03_prof_default $OUT/r03_prof_train_bf16 -name "*kernel_trace.csv" -delete
bash $ROOT/tools/profile_16bit.sh r03 2>&1 | grep -E "^==|conv_c8|wgrad|cac|done" | cut -c1-130
cd /tmp && export TMPDIR=/tmp
echo "== C1 trace"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r03_c1_trace -- python3 $ROOT/tools/time_c1.py > $OUT/r03_c1_trace.log 2>&1
f=$(find $OUT/r03_c1_trace -name "*kernel_stats.csv" | head -1); test -n "$f"; cp "$f" $OUT/r03_c1_1x128x128_f32_kernel_stats.csv
find $OUT/r03_c1_trace -name "*kernel_trace.csv" -delete
tail -2 $OUT/r03_c1_trace.log
echo finished
