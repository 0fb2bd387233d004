#!/bin/bash
# Run ON THE GPU BOX: the round-4 tree (ab/r04, built from commit 553c3be) against the current one, same box, alternating
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT
for rep in 1 2; do
  for t in r04 r05; do
    if [ $t = r04 ]; then D=$ROOT/ab/r04; TB=$D/trace_b1.py; else D=$ROOT; TB=$ROOT/tools/trace_b1.py; fi
    echo "== $t"
    (cd $D && python3 bench.py --mode train --dtype bf16 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed 's/^/bf16 train /')
    (cd $D && python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fwd-bwd --no-script-pattern 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed 's/^/fp32 fwd /')
    (cd $D && python3 bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-fwd-bwd --no-script-pattern 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed 's/^/bf16 fwd /')
    for c in "fp32 370 463" "fp16 370 463" "fp32 128 128" "fp16 247 343"; do CODON_B1_GRAPH=1 python3 $TB $c 60 2>/dev/null | grep "ms/forward"; done
  done
done
