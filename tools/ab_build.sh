#!/bin/bash
# Build a variant of libcodon_hip.so that differs in ONE translation unit's -D flags, for same-box A/B timing:
#   tools/ab_build.sh <tag> <file.hip> [-DFOO=1 ...]   ->  tools/probes/bin/libcodon_hip_<tag>.so
# Use it with CODON_AMD_LIB=$PWD/tools/probes/bin/libcodon_hip_<tag>.so python tools/time_conv.py ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
tag=$1; src=$2; shift 2
cd "$ROOT/codon_amd/csrc"
make -j8 >/dev/null
mkdir -p "$ROOT/tools/probes/bin" build
base=${src%.hip}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I"$ROOT/include" -Wall -Wno-unused-function "$@" -c "$src" -o "build/ab_${tag}.o"
objs=$(ls build/*.o | grep -v "build/ab_" | grep -v "build/${base}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/probes/bin/libcodon_hip_${tag}.so" $objs "build/ab_${tag}.o"
echo "$ROOT/tools/probes/bin/libcodon_hip_${tag}.so"
