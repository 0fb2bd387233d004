# kernel-trace summary of two bf16 training steps; prints the rows matching $1 (regex)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/tt; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --mode train --dtype bf16 --no-cpu-baseline --steps 2 --warmup 1 > $OUT/trace.log 2>&1
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); test -n "$f" && cp "$f" $OUT/kernel_stats.csv
find $OUT/trace -name "*kernel_trace.csv" -delete
grep -E "$1" $OUT/kernel_stats.csv | cut -c1-200
