#!/bin/bash
# Run ON THE GPU BOX
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export CODON_AMD_LIB=$ROOT/tools/probes/bin/libcodon_hip_gridmode.so
CODON_PROBE_SMALL=0 python3 $ROOT/tools/probes/grid_mode_sweep.py "8x32 tiles, two workgroups per CU"
CODON_PROBE_SMALL=1 CODON_PROBE_NOSOLO=1 python3 $ROOT/tools/probes/grid_mode_sweep.py "4x32 tiles, two workgroups per CU"
CODON_PROBE_SMALL=1 python3 $ROOT/tools/probes/grid_mode_sweep.py "4x32 tiles, one workgroup per CU"
