set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
python tools/diag_bf16grad.py > gpurun_out/r4a/diag_bf16grad.txt 2>&1
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_gpu_backward.py::test_bf16_gradients_vs_reference_bf16_autograd > gpurun_out/r4a/gpu_tests.log 2>&1; rc=$?; tail -15 gpurun_out/r4a/gpu_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
bash tools/probes/r4_guard_cost.sh
