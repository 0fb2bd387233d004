set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4l
timeout -k 10 300 python -m pytest tests/test_gpu_c8.py -q -x -p no:cacheprovider -k "resident" > gpurun_out/r4l/t0.log 2>&1; rc=$?; echo "resident test rc=$rc"; tail -5 gpurun_out/r4l/t0.log | cut -c1-300
if [ $rc -ne 0 ]; then exit $rc; fi
for lib in "" staged3 "" staged3; do
  if [ -n "$lib" ]; then export CODON_AMD_LIB=$GRAFT_REPO_ROOT/tools/probes/bin/libcodon_hip_$lib.so; else unset CODON_AMD_LIB; fi
  echo "== lib=${lib:-default(resident)}"; DATA=relu timeout -k 10 120 python tools/time_conv.py bf16 2 2>&1 | grep conv
done
for lib in "" staged3 "" staged3; do
  if [ -n "$lib" ]; then export CODON_AMD_LIB=$GRAFT_REPO_ROOT/tools/probes/bin/libcodon_hip_$lib.so; else unset CODON_AMD_LIB; fi
  timeout -k 10 200 python bench.py --mode train --dtype bf16 --steps 6 --warmup 2 > gpurun_out/r4l/train_${lib:-res}.json 2> gpurun_out/r4l/train_${lib:-res}.err; python -c "
import json; d=json.load(open('gpurun_out/r4l/train_${lib:-res}.json')); print('bf16 train ${lib:-resident}', d['ms_per_step'])"
  timeout -k 10 200 python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4l/fwd_${lib:-res}.json 2> gpurun_out/r4l/fwd_${lib:-res}.err; python -c "
import json; d=json.load(open('gpurun_out/r4l/fwd_${lib:-res}.json')); print('bf16 fwd ${lib:-resident}', d['ms_per_step'])"
done
