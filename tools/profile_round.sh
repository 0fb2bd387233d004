#!/bin/bash
# Run ON THE GPU BOX (gpurun -- bash tools/profile_round.sh r02): bench records + rocprofv3 summaries of one round.
# Everything lands under gpurun_out/<tag>_*; copy what is to be judged into profiles/.
set -e -o pipefail
tag=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py"
echo "== default bench line"; $B --steps 5 --warmup 2 > $OUT/${tag}_bench_default.json 2> $OUT/${tag}_bench_default.err
echo "== kernel trace, forward + fwd_bwd leg"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${tag}_prof_default -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/${tag}_prof_default.log 2>&1
echo "== PMC FETCH_SIZE"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${tag}_pmc_fetch -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-fwd-bwd > $OUT/${tag}_pmc_fetch.log 2>&1
echo "== PMC WRITE_SIZE"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${tag}_pmc_write -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-fwd-bwd > $OUT/${tag}_pmc_write.log 2>&1
python3 $ROOT/tools/pmc_hbm.py $OUT/${tag}_pmc_fetch $OUT/${tag}_pmc_write $OUT/${tag}_fwd_b32_480x640_pmc_hbm.json > $OUT/${tag}_pmc_hbm.txt
echo "== other configs"
$B --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/${tag}_bench_bf16.json 2> $OUT/${tag}_other.err
$B --mode train --dtype bf16 --steps 5 --warmup 2 > $OUT/${tag}_train_bf16.json 2> $OUT/${tag}_other.err
$B --scale 8 --batch 16 --height 960 --width 1280 --steps 3 --warmup 1 --no-cpu-baseline --no-fwd-bwd > $OUT/${tag}_bench_x8_f32.json 2> $OUT/${tag}_other.err
$B --scale 16 --dtype bf16 --batch 8 --height 1920 --width 2560 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/${tag}_bench_x16_bf16.json 2> $OUT/${tag}_other.err
$B --model rmcr --steps 3 --warmup 1 --no-cpu-baseline > $OUT/${tag}_bench_rmcr_f32.json 2> $OUT/${tag}_other.err
echo "== bf16 training kernel trace"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${tag}_prof_train_bf16 -- python3 $ROOT/bench.py --mode train --dtype bf16 --steps 3 --warmup 1 > $OUT/${tag}_prof_train_bf16.log 2>&1
find $OUT/${tag}_prof_default $OUT/${tag}_prof_train_bf16 -name "*kernel_stats.csv" | head
echo done
