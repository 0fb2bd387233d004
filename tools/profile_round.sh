#!/bin/bash
# Run ON THE GPU BOX (gpurun -- bash tools/profile_round.sh r05): bench records + rocprofv3 summaries of one round.
# Everything lands under gpurun_out/<tag>_*; copy what is to be judged into profiles/.
# Every rocprofv3 command puts the program itself (python3 bench.py ...) directly after "--"; counters are collected in
# their own runs (no trace domains besides --kernel-trace), FETCH_SIZE / WRITE_SIZE / clock+busy in separate passes.
# One stderr file per command (round 3 wrote five runs into one .err: only the last survived).
set -e -o pipefail
tag=${1:-r05}
# (the box is fresh on every gpurun call; LOCALLY gpurun_out/ accumulates: regenerate tables only from the newest counter CSV per directory)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py"
run() {   # run <name> <bench args...>: JSON line -> <tag>_<name>.json, stderr -> <tag>_<name>.err
  local name=$1; shift
  echo "== $name"; $B "$@" > $OUT/${tag}_${name}.json 2> $OUT/${tag}_${name}.err
}
pmc3() {  # pmc3 <label> <out.json> <mode> <esize> <bench args...>: FETCH / WRITE / busy passes + the merged table
  # (PMC_SHAPE="B H W" for a profiled command that is not 32 x 480 x 640)
  local label=$1 out=$2 mode=$3 es=$4; shift 4
  local shape=${PMC_SHAPE:-32 480 640}
  for c in FETCH_SIZE WRITE_SIZE; do
    echo "== $label: pmc $c"
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${tag}_${label}_$c -- python3 $ROOT/bench.py "$@" --steps 1 --warmup 0 > $OUT/${tag}_${label}_$c.log 2>&1
  done
  echo "== $label: pmc clock + matrix-pipe busy"
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/${tag}_${label}_busy -- python3 $ROOT/bench.py "$@" --steps 1 --warmup 0 > $OUT/${tag}_${label}_busy.log 2>&1
  python3 $ROOT/tools/pmc_report.py $OUT/${tag}_${label}_FETCH_SIZE $OUT/${tag}_${label}_WRITE_SIZE $OUT/${tag}_${label}_busy $OUT/$out $shape $mode $es | tee $OUT/${tag}_${label}_pmc.txt
  find $OUT/${tag}_${label}_FETCH_SIZE $OUT/${tag}_${label}_WRITE_SIZE $OUT/${tag}_${label}_busy -name "*.csv" -size +8M -delete
}
trace() { # trace <label> <stats.csv name> <bench args...>
  local label=$1 out=$2; shift 2
  echo "== $label: kernel trace"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${tag}_${label}_trace -- python3 $ROOT/bench.py "$@" > $OUT/${tag}_${label}_trace.log 2>&1
  f=$(find $OUT/${tag}_${label}_trace -name "*kernel_stats.csv" | head -1); test -n "$f"; cp "$f" $OUT/$out
  find $OUT/${tag}_${label}_trace -name "*kernel_trace.csv" -delete
}

# LINES_ONLY=1: only the bench lines (the PMC tables of this library are already in profiles/)
# PART=a|b|c (round 6: a gpurun call is at most 20 minutes): a = traces + PMC of the three 32 x 480 x 640 workloads, b = PMC of
# configs[3] / [4], c = the bench lines + the one-image timelines.  Copy gpurun_out/<tag>_*_pmc.json into profiles/ between
# the calls (every call starts on a fresh box); without PART everything runs in one go.
# (--no-strong-shards: the default line's b16 / b8 / b4 shard timings launch the same kernels on smaller batches and would
# be averaged into the per-kernel rows)
part=${PART:-abc}
if [ -z "$LINES_ONLY" ]; then
if [[ $part == *a* ]]; then
trace default ${tag}_default_bench_kernel_stats.csv --steps 3 --warmup 1 --no-cpu-baseline --no-script-pattern --no-strong-shards
pmc3 fwd ${tag}_fwd_b32_480x640_pmc.json fwd 4 --no-cpu-baseline --no-fwd-bwd --no-script-pattern --no-strong-shards
trace bf16fwd ${tag}_bf16_fwd_b32_480x640_kernel_stats.csv --dtype bf16 --no-cpu-baseline --no-fwd-bwd --steps 3 --warmup 1
pmc3 bf16fwd ${tag}_bf16_fwd_b32_480x640_pmc.json fwd 2 --dtype bf16 --no-cpu-baseline --no-fwd-bwd
trace bf16train ${tag}_bf16_train_b32_480x640_kernel_stats.csv --mode train --dtype bf16 --no-cpu-baseline --steps 3 --warmup 1
pmc3 bf16train ${tag}_bf16_train_b32_480x640_pmc.json train 2 --mode train --dtype bf16 --no-cpu-baseline
fi
if [[ $part == *b* ]]; then
# round 6: one PMC pass each for BASELINE configs[3] (x8, b16, 960 x 1280, fp32) and configs[4] (x16, b8, 1920 x 2560, bf16), so
# that their bench lines carry roofline.traffic too
PMC_SHAPE="16 960 1280" pmc3 x8fwd ${tag}_x8_fwd_b16_960x1280_pmc.json fwd 4 --scale 8 --batch 16 --height 960 --width 1280 --no-cpu-baseline --no-fwd-bwd
PMC_SHAPE="8 1920 2560" pmc3 x16fwd ${tag}_x16_bf16_fwd_b8_1920x2560_pmc.json fwd 2 --scale 16 --dtype bf16 --batch 8 --height 1920 --width 2560 --no-cpu-baseline --no-fwd-bwd
fi
# the lines below read roofline.traffic from profiles/: give them the tables of THIS library (traffic_from_hash == lib_source_hash)
for f in $OUT/${tag}_fwd_b32_480x640_pmc.json $OUT/${tag}_bf16_fwd_b32_480x640_pmc.json $OUT/${tag}_bf16_train_b32_480x640_pmc.json \
   $OUT/${tag}_x8_fwd_b16_960x1280_pmc.json $OUT/${tag}_x16_bf16_fwd_b8_1920x2560_pmc.json; do test -f $f && cp $f $ROOT/profiles/; done
fi
[[ $part == *c* ]] || { echo "done (part $part)"; exit 0; }
run bench_default --steps 5 --warmup 2
run bench_bf16 --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline
run train_bf16 --mode train --dtype bf16 --steps 10 --warmup 2
run bench_x8_f32 --scale 8 --batch 16 --height 960 --width 1280 --steps 3 --warmup 1 --no-cpu-baseline --no-fwd-bwd
run bench_x16_bf16 --scale 16 --dtype bf16 --batch 8 --height 1920 --width 2560 --steps 5 --warmup 2 --no-cpu-baseline
run bench_rmcr_f32 --model rmcr --steps 3 --warmup 1 --no-cpu-baseline
# one image per call (the reference script's own pattern): kernel statistics + one forward's timeline
cd $ROOT
test -n "$LINES_ONLY" && { echo done; exit 0; }
bash tools/probes/trace_b1.sh ${tag}_b1_fp16_370x463 fp16 370 463 > $OUT/${tag}_b1_fp16.log 2>&1
bash tools/probes/trace_b1.sh ${tag}_b1_fp32_128x128 fp32 128 128 > $OUT/${tag}_b1_fp32.log 2>&1
echo done
