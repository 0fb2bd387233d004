set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
python -m pytest tests/test_gpu_backward.py -q -p no:cacheprovider -k "bf16_gradients_vs_reference or packed_weight or weight_guard" -s > gpurun_out/r4b/t1.log 2>&1; echo "tests rc=$?"; grep -E "^\[bf16grad|passed|failed|Error" gpurun_out/r4b/t1.log | cut -c1-700
python tools/diag_bf16grad.py > gpurun_out/r4b/diag_bf16grad.txt 2>&1
python -m pytest tests/test_gpu_c8.py tests/test_gpu_kernels.py tests/test_gpu_forward.py -q -p no:cacheprovider > gpurun_out/r4b/t2.log 2>&1; echo "c8/kernels/forward rc=$?"; tail -3 gpurun_out/r4b/t2.log
bash tools/probes/r4_guard_cost.sh 2>&1 | grep GUARD
python tools/time_conv.py > gpurun_out/r4b/time_conv.txt 2>&1; cat gpurun_out/r4b/time_conv.txt | tail -20
for d in relu rand; do
for lib in "" nopersist persist5; do
  if [ -n "$lib" ]; then export CODON_AMD_LIB=$GRAFT_REPO_ROOT/tools/probes/bin/libcodon_hip_$lib.so; else unset CODON_AMD_LIB; fi
  echo "== DATA=$d lib=${lib:-default(persist3)}"; DATA=$d python tools/time_conv.py bf16 2 2>&1 | grep conv; DATA=$d python tools/time_conv.py bf16 1 2>&1 | grep conv
done; done
unset CODON_AMD_LIB
python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4b/bench_bf16.json 2> gpurun_out/r4b/bench_bf16.err; python -c "
import json; d=json.load(open('gpurun_out/r4b/bench_bf16.json')); print('bf16 fwd', d['ms_per_step'])"
python bench.py --mode train --dtype bf16 --steps 5 --warmup 2 > gpurun_out/r4b/train_bf16.json 2> gpurun_out/r4b/train_bf16.err; python -c "
import json; d=json.load(open('gpurun_out/r4b/train_bf16.json')); print('bf16 train', d['ms_per_step'])"
export CODON_AMD_LIB=$GRAFT_REPO_ROOT/tools/probes/bin/libcodon_hip_nopersist.so
python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4b/bench_bf16_np.json 2> gpurun_out/r4b/bench_bf16_np.err; python -c "
import json; d=json.load(open('gpurun_out/r4b/bench_bf16_np.json')); print('bf16 fwd nopersist', d['ms_per_step'])"
python bench.py --mode train --dtype bf16 --steps 5 --warmup 2 > gpurun_out/r4b/train_bf16_np.json 2> gpurun_out/r4b/train_bf16_np.err; python -c "
import json; d=json.load(open('gpurun_out/r4b/train_bf16_np.json')); print('bf16 train nopersist', d['ms_per_step'])"
