set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4h
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_c8.py -q -x -p no:cacheprovider -k "wgrad or training_schedules" > gpurun_out/r4h/t1.log 2>&1; echo "wgrad tests rc=$?"; tail -3 gpurun_out/r4h/t1.log | cut -c1-300
python -m pytest tests/test_gpu_backward.py -q -x -p no:cacheprovider -k "deterministic or golden or bf16" > gpurun_out/r4h/t2.log 2>&1; echo "backward rc=$?"; tail -3 gpurun_out/r4h/t2.log | cut -c1-300
for lib in "" nbuf2 "" nbuf2; do
  if [ -n "$lib" ]; then export CODON_AMD_LIB=$GRAFT_REPO_ROOT/tools/probes/bin/libcodon_hip_$lib.so; else unset CODON_AMD_LIB; fi
  echo "== lib=${lib:-default(nbuf3)}"; DATA=relu python tools/time_wgrad.py bf16 0 2>&1 | grep wgrad; DATA=relu python tools/time_wgrad.py bf16 1 2>&1 | grep wgrad
done
for lib in "" nbuf2 "" nbuf2; do
  if [ -n "$lib" ]; then export CODON_AMD_LIB=$GRAFT_REPO_ROOT/tools/probes/bin/libcodon_hip_$lib.so; else unset CODON_AMD_LIB; fi
  python bench.py --mode train --dtype bf16 --steps 6 --warmup 2 > gpurun_out/r4h/train_${lib:-nbuf3}.json 2> gpurun_out/r4h/train_${lib:-nbuf3}.err; python -c "
import json; d=json.load(open('gpurun_out/r4h/train_${lib:-nbuf3}.json')); print('bf16 train ${lib:-nbuf3}', d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
