#!/bin/bash
# Run ON THE GPU BOX: s_setprio around the prologue / epilogue of the fp32 conv kernel (same box, alternating)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT
B=$ROOT/tools/probes/bin
for rep in 1 2; do for v in base prio; do echo "== $v"; CODON_AMD_LIB=$B/libcodon_hip_$v.so python3 tools/time_conv.py f32 2>&1 | grep "^conv" || exit 1; done; done
echo "== timeline prio"; CODON_AMD_LIB=$B/libcodon_hip_priot.so python3 tools/probes/cu_timeline.py 3 64 64 2>&1 | grep -v "amdgpu.ids\|CODON_AMD_LIB"
