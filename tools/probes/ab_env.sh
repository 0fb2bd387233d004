#!/bin/bash
# Run ON THE GPU BOX: same-box alternating A/B of an environment switch on the bf16 training step
#   ab_env.sh VAR valueA valueB
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT
for rep in 1 2 3; do for v in $2 $3; do
  echo "== $1=$v: bf16 train"; env $1=$v python3 bench.py --mode train --dtype bf16 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"peak_mem_gb": [0-9.]*' | head -2 | tr '\n' ' ' || exit 1; echo
done; done
