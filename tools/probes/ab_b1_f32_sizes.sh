#!/bin/bash
# Run ON THE GPU BOX: fp32 one-image eager forwards at several sizes, round-4 tree vs current
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
for s in "200 300" "247 343" "300 400" "370 463" "480 640" "600 800" "768 1024"; do
  a=$(python3 $ROOT/ab/r04/trace_b1.py fp32 $s 20 2>&1 | grep forward)
  b=$(python3 $ROOT/tools/trace_b1.py fp32 $s 20 2>&1 | grep forward)
  echo "r04: $a   |   now: $b"
done
