set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2e
timeout -k 10 900 python3 -m pytest tests/test_gpu_round2.py tests/test_gpu_stages.py -m gpu -x -q > gpurun_out/r2e/gpu_tests.log 2>&1 && \
timeout -k 10 300 python3 tests/gpu_tune.py --config H --reps 3 --opts "apply_nat=0" "apply_nat=1" "apply_nat=0" "apply_nat=1" > gpurun_out/r2e/tune_H.txt 2>&1 && \
timeout -k 10 300 python3 tests/gpu_gram_trace.py --config H --opts "gram_taper=1" > gpurun_out/r2e/gram_trace_H.txt 2>&1 && \
timeout -k 10 200 python3 tests/gpu_tune.py --config H --dtype f64 --rows 500000 --reps 2 --opts "apply_nat=0" "apply_nat=1" > gpurun_out/r2e/tune_H_f64.txt 2>&1 && \
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r2e/gpu_tests_all.log 2>&1
echo "rc=$?"
tail -5 gpurun_out/r2e/gpu_tests.log gpurun_out/r2e/gpu_tests_all.log
