#!/bin/bash
# Build a variant of libcodon_hip.so that differs in the -D flags of ONE OR MORE translation units, for same-box A/B timing:
#   tools/ab_build.sh <tag> <file.hip[,file2.hip,...]> [-DFOO=1 ...]   ->  tools/probes/bin/libcodon_hip_<tag>.so
# Use it with CODON_AMD_LIB=$PWD/tools/probes/bin/libcodon_hip_<tag>.so python tools/time_conv.py ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
tag=$1; srcs=$2; shift 2
cd "$ROOT/codon_amd/csrc"
make -j8 >/dev/null
mkdir -p "$ROOT/tools/probes/bin" build
objs=$(ls build/*.o | grep -v "build/ab_")
new=""
for src in ${srcs//,/ }; do
  base=${src%.hip}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I"$ROOT/include" -Wall -Wno-unused-function "$@" -c "$src" -o "build/ab_${tag}_${base}.o" &
  objs=$(echo "$objs" | grep -v "build/${base}.o")
  new="$new build/ab_${tag}_${base}.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/probes/bin/libcodon_hip_${tag}.so" $objs $new
echo "$ROOT/tools/probes/bin/libcodon_hip_${tag}.so"
