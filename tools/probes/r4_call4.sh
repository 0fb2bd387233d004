set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c
python -m pytest tests/test_gpu_backward.py -q -p no:cacheprovider -k "bf16_gradients_vs_reference" -s > gpurun_out/r4c/t1.log 2>&1; echo "tests rc=$?"; grep -E "^\[bf16grad|passed|failed|Error" gpurun_out/r4c/t1.log | cut -c1-600
python -m pytest tests/test_gpu_c8.py tests/test_gpu_kernels.py tests/test_gpu_forward.py tests/test_gpu_fullsize.py -q -x -p no:cacheprovider > gpurun_out/r4c/t2.log 2>&1; echo "c8/kernels/forward/fullsize rc=$?"; tail -3 gpurun_out/r4c/t2.log
bash tools/probes/r4_guard_cost.sh 2>&1 | grep GUARD
for d in relu; do
for lib in "" nopersist; do
  if [ -n "$lib" ]; then export CODON_AMD_LIB=$GRAFT_REPO_ROOT/tools/probes/bin/libcodon_hip_$lib.so; else unset CODON_AMD_LIB; fi
  echo "== DATA=$d lib=${lib:-default(persist3)}"; DATA=$d python tools/time_conv.py bf16 2 2>&1 | grep conv; DATA=$d python tools/time_conv.py bf16 2 2>&1 | grep conv
done; done
unset CODON_AMD_LIB
python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4c/bench_bf16.json 2> gpurun_out/r4c/bench_bf16.err; python -c "
import json; d=json.load(open('gpurun_out/r4c/bench_bf16.json')); print('bf16 fwd', d['ms_per_step'])"
python bench.py --mode train --dtype bf16 --steps 5 --warmup 2 > gpurun_out/r4c/train_bf16.json 2> gpurun_out/r4c/train_bf16.err; python -c "
import json; d=json.load(open('gpurun_out/r4c/train_bf16.json')); print('bf16 train', d['ms_per_step'])"
export CODON_AMD_LIB=$GRAFT_REPO_ROOT/tools/probes/bin/libcodon_hip_nopersist.so
python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4c/bench_bf16_np.json 2> gpurun_out/r4c/bench_bf16_np.err; python -c "
import json; d=json.load(open('gpurun_out/r4c/bench_bf16_np.json')); print('bf16 fwd nopersist', d['ms_per_step'])"
python bench.py --mode train --dtype bf16 --steps 5 --warmup 2 > gpurun_out/r4c/train_bf16_np.json 2> gpurun_out/r4c/train_bf16_np.err; python -c "
import json; d=json.load(open('gpurun_out/r4c/train_bf16_np.json')); print('bf16 train nopersist', d['ms_per_step'])"
