#!/bin/bash
# Run ON THE GPU BOX: same-box alternating A/B of two library builds on the bf16 training step and the bf16 forward
#   ab_step.sh <tagA> <tagB>
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT
B=$ROOT/tools/probes/bin
for rep in 1 2; do for v in $1 $2; do
  echo "== $v: bf16 train"; CODON_AMD_LIB=$B/libcodon_hip_$v.so python3 bench.py --mode train --dtype bf16 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 || exit 1
  echo "== $v: bf16 forward"; CODON_AMD_LIB=$B/libcodon_hip_$v.so python3 bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-fwd-bwd --no-script-pattern 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 || exit 1
done; done
