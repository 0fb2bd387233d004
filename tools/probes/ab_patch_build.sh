#!/bin/bash
# Build a variant of libcodon_hip.so from a PATCHED copy of codon_amd/csrc (timing-only ablations that do not belong in the
# product sources):   ab_patch_build.sh <tag> <patch file> <file.hip> [-DFOO=1 ...]  ->  tools/probes/bin/libcodon_hip_<tag>.so
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
tag=$1; patch=$(realpath "$2"); src=$3; shift 3
T=$(mktemp -d)
cp -r "$ROOT/codon_amd/csrc" "$T/csrc"
(cd "$T/csrc" && patch -s "$src" < "$patch")
mkdir -p "$ROOT/tools/probes/bin"
base=${src%.hip}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I"$ROOT/include" -I"$T/csrc" -Wall -Wno-unused-function "$@" -c "$T/csrc/$src" -o "$T/ab_$base.o"
objs=$(ls "$ROOT"/codon_amd/csrc/build/*.o | grep -v "build/ab_" | grep -v "build/${base}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/probes/bin/libcodon_hip_${tag}.so" $objs "$T/ab_$base.o"
rm -rf "$T"
echo "$ROOT/tools/probes/bin/libcodon_hip_${tag}.so"
