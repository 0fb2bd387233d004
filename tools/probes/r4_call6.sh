set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4e
for lib in "" p364 p564 "" p364 p564; do
  if [ -n "$lib" ]; then export CODON_AMD_LIB=$GRAFT_REPO_ROOT/tools/probes/bin/libcodon_hip_$lib.so; else unset CODON_AMD_LIB; fi
  echo "== lib=${lib:-default}"; DATA=relu python tools/time_conv.py f32 2 2>&1 | grep conv; DATA=relu python tools/time_conv.py f32 1 2>&1 | grep conv
done
for lib in "" p364 p564; do
  if [ -n "$lib" ]; then export CODON_AMD_LIB=$GRAFT_REPO_ROOT/tools/probes/bin/libcodon_hip_$lib.so; else unset CODON_AMD_LIB; fi
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fwd-bwd --no-script-pattern > gpurun_out/r4e/bench_f32_${lib:-default}.json 2> gpurun_out/r4e/bench_f32_${lib:-default}.err; python -c "
import json; d=json.load(open('gpurun_out/r4e/bench_f32_${lib:-default}.json')); print('f32 fwd ${lib:-default}', d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
