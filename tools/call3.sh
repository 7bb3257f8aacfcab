set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2c
timeout -k 10 300 python3 tests/gpu_gram_trace.py --config H --opts "gram_taper=0" "gram_taper=1" "gram_taper=1,gram_nsplit=64" "gram_taper=1,gram_nsplit=32" > gpurun_out/r2c/gram_trace_H.txt 2>&1 && \
timeout -k 10 300 python3 tests/gpu_tune.py --config H --reps 3 --opts "gram_taper=0" "gram_taper=1" "gram_taper=1,gram_nsplit=64" "gram_taper=1,gram_nsplit=40" "gram_taper=1,gram_nsplit=32" "gram_taper=0,gram_nsplit=48" > gpurun_out/r2c/tune_H.txt 2>&1 && \
timeout -k 10 200 python3 tests/gpu_tune.py --config H --dtype f64 --rows 500000 --reps 2 --opts "gram_taper=0" "gram_taper=1" > gpurun_out/r2c/tune_H_f64.txt 2>&1 && \
timeout -k 10 200 python3 tests/gpu_tune.py --config H --rows 125000 --reps 3 --opts "gram_taper=0" "gram_taper=1" > gpurun_out/r2c/tune_H_125k.txt 2>&1 && \
timeout -k 10 900 python3 -m pytest tests/test_gpu_round2.py tests/test_gpu_stages.py -m gpu -x -q > gpurun_out/r2c/gpu_tests.log 2>&1
echo "rc=$?"
tail -5 gpurun_out/r2c/gpu_tests.log
