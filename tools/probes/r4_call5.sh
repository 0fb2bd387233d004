set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4d
bash tools/probes/r4_guard_cost.sh 2>&1 | grep GUARD
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_forward.py -q -x -p no:cacheprovider > gpurun_out/r4d/t2.log 2>&1; echo "kernels/forward rc=$?"; tail -3 gpurun_out/r4d/t2.log
for lib in "" rs1 "" rs1; do
  if [ -n "$lib" ]; then export CODON_AMD_LIB=$GRAFT_REPO_ROOT/tools/probes/bin/libcodon_hip_$lib.so; else unset CODON_AMD_LIB; fi
  echo "== lib=${lib:-default(RS3=3)}"; B=16 DATA=relu python tools/time_conv.py f32 2 2>&1 | grep conv; B=16 DATA=relu python tools/time_conv.py f32 3 2>&1 | grep conv
done
unset CODON_AMD_LIB
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fwd-bwd --no-script-pattern > gpurun_out/r4d/bench_f32.json 2> gpurun_out/r4d/bench_f32.err; python -c "
import json; d=json.load(open('gpurun_out/r4d/bench_f32.json')); print('f32 fwd RS3=3', d['ms_per_step'], d['roofline']['avg_launch_ms'])"
export CODON_AMD_LIB=$GRAFT_REPO_ROOT/tools/probes/bin/libcodon_hip_rs1.so
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fwd-bwd --no-script-pattern > gpurun_out/r4d/bench_f32_rs1.json 2> gpurun_out/r4d/bench_f32_rs1.err; python -c "
import json; d=json.load(open('gpurun_out/r4d/bench_f32_rs1.json')); print('f32 fwd RS3=1', d['ms_per_step'], d['roofline']['avg_launch_ms'])"
