#!/bin/bash
# Run ON THE GPU BOX: kernel statistics of the fp32 one-image forward at 370 x 463 in the round-4 tree and the current one
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/f32mid; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for t in r04 r05; do
  if [ $t = r04 ]; then TB=$ROOT/ab/r04/trace_b1.py; else TB=$ROOT/tools/trace_b1.py; fi
  rm -rf $OUT/$t
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$t -- python3 $TB fp32 370 463 20 > $OUT/$t.log 2>&1
  f=$(find $OUT/$t -name "*kernel_stats.csv" | head -1); cp $f $OUT/${t}_stats.csv
  find $OUT/$t -name "*kernel_trace.csv" -delete
  echo "== $t"; head -14 $OUT/${t}_stats.csv | cut -c1-150
done
