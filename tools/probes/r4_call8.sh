set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4g
python -m pytest tests/test_gpu_c8.py tests/test_gpu_kernels.py tests/test_gpu_forward.py -q -x -p no:cacheprovider > gpurun_out/r4g/t1.log 2>&1; echo "c8/kernels/forward rc=$?"; tail -3 gpurun_out/r4g/t1.log | cut -c1-300
DATA=relu python tools/time_conv.py bf16 2>&1 | grep conv
python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4g/bench_bf16.json 2> gpurun_out/r4g/bench_bf16.err; python -c "
import json; d=json.load(open('gpurun_out/r4g/bench_bf16.json')); print('bf16 fwd', d['ms_per_step'])"
python bench.py --mode train --dtype bf16 --steps 6 --warmup 2 > gpurun_out/r4g/train_bf16.json 2> gpurun_out/r4g/train_bf16.err; python -c "
import json; d=json.load(open('gpurun_out/r4g/train_bf16.json')); print('bf16 train', d['ms_per_step'])"
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fwd-bwd --no-script-pattern > gpurun_out/r4g/bench_f32.json 2> gpurun_out/r4g/bench_f32.err; python -c "
import json; d=json.load(open('gpurun_out/r4g/bench_f32.json')); print('f32 fwd', d['ms_per_step'], d['roofline']['avg_launch_ms'])"
export CODON_AMD_LIB=$GRAFT_REPO_ROOT/tools/probes/bin/libcodon_hip_noskip.so
echo "== all pieces issued (before)"
DATA=relu python tools/time_conv.py bf16 2>&1 | grep conv
python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4g/bench_bf16_ns.json 2> gpurun_out/r4g/bench_bf16_ns.err; python -c "
import json; d=json.load(open('gpurun_out/r4g/bench_bf16_ns.json')); print('bf16 fwd', d['ms_per_step'])"
python bench.py --mode train --dtype bf16 --steps 6 --warmup 2 > gpurun_out/r4g/train_bf16_ns.json 2> gpurun_out/r4g/train_bf16_ns.err; python -c "
import json; d=json.load(open('gpurun_out/r4g/train_bf16_ns.json')); print('bf16 train', d['ms_per_step'])"
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fwd-bwd --no-script-pattern > gpurun_out/r4g/bench_f32_ns.json 2> gpurun_out/r4g/bench_f32_ns.err; python -c "
import json; d=json.load(open('gpurun_out/r4g/bench_f32_ns.json')); print('f32 fwd', d['ms_per_step'], d['roofline']['avg_launch_ms'])"
