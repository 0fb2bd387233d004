#!/bin/bash
# Run ON THE GPU BOX: kernel-trace summary + three PMC passes of the bf16 forward and of the bf16 training step.
#   gpurun -- bash tools/profile_16bit.sh <tag> [tree]      tree = repo root to profile (default: this one; ab/r02 = round-2 tree)
# Every rocprofv3 command puts the program itself (python3 bench.py ...) directly after "--"; counters are collected in
# their own runs (no trace domains besides --kernel-trace).  Output: gpurun_out/<tag>_*.
set -e -o pipefail
tag=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TREE=$(cd "${2:-$ROOT}" && pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for mode in fwd train; do
  if [ $mode = fwd ]; then A="--dtype bf16 --no-cpu-baseline --no-fwd-bwd"; else A="--mode train --dtype bf16 --no-cpu-baseline"; fi
  echo "== $tag $mode: kernel trace"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${tag}_${mode}_trace -- python3 $TREE/bench.py $A --steps 3 --warmup 1 > $OUT/${tag}_${mode}_trace.log 2>&1
  f=$(find $OUT/${tag}_${mode}_trace -name "*kernel_stats.csv" | head -1); test -n "$f"; cp "$f" $OUT/${tag}_bf16_${mode}_b32_480x640_kernel_stats.csv
  find $OUT/${tag}_${mode}_trace -name "*kernel_trace.csv" -delete
  for c in FETCH_SIZE WRITE_SIZE; do
    echo "== $tag $mode: pmc $c"
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${tag}_${mode}_$c -- python3 $TREE/bench.py $A --steps 1 --warmup 0 > $OUT/${tag}_${mode}_$c.log 2>&1
  done
  echo "== $tag $mode: pmc clock + matrix-pipe busy"
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/${tag}_${mode}_busy -- python3 $TREE/bench.py $A --steps 1 --warmup 0 > $OUT/${tag}_${mode}_busy.log 2>&1
  python3 $ROOT/tools/pmc_report.py $OUT/${tag}_${mode}_FETCH_SIZE $OUT/${tag}_${mode}_WRITE_SIZE $OUT/${tag}_${mode}_busy $OUT/${tag}_bf16_${mode}_b32_480x640_pmc.json | tee $OUT/${tag}_bf16_${mode}_pmc.txt
  find $OUT/${tag}_${mode}_FETCH_SIZE $OUT/${tag}_${mode}_WRITE_SIZE $OUT/${tag}_${mode}_busy -name "*.csv" -size +8M -delete
done
echo done
