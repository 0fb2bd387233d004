#!/bin/bash
# Run ON THE GPU BOX
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
for t in $ROOT/ab/r04 $ROOT; do
  for k in 0 64 256 1024 1536 2048 3072 5120 20480 65536; do
    python3 $ROOT/tools/probes/perturb_b1.py $t $k 2>&1 | tail -1
  done
done
