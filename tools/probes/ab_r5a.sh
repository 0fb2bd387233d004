#!/bin/bash
# Run ON THE GPU BOX: round-5 pass removals, same-box alternating A/B on the bf16 training step
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT
run() { echo -n "== $*: "; env "$@" python3 bench.py --mode train --dtype bf16 --steps 8 --warmup 2 --no-cpu-baseline 2>>gpurun_out/ab_r5a.err | grep -o '"ms_per_step": [0-9.]*' | head -1 || exit 1; }
for rep in 1 2; do
  run CODON_DEFER_REDUCE=1 CODON_SUM_GFUSE_IN_DGRAD=1 CODON_GRAD_DIRECT=1
  run CODON_DEFER_REDUCE=0 CODON_SUM_GFUSE_IN_DGRAD=0 CODON_GRAD_DIRECT=0
  run CODON_DEFER_REDUCE=1 CODON_SUM_GFUSE_IN_DGRAD=0 CODON_GRAD_DIRECT=1
  run CODON_DEFER_REDUCE=0 CODON_SUM_GFUSE_IN_DGRAD=1 CODON_GRAD_DIRECT=0
done
