set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2f
timeout -k 10 900 python3 -m pytest tests/test_gpu_stages.py tests/test_gpu_parity.py -m gpu -x -q -k "stages or artifact or oracle_kats or ragged or error" > gpurun_out/r2f/gpu_tests.log 2>&1 && \
timeout -k 10 300 python3 tests/gpu_tune.py --config H --reps 3 > gpurun_out/r2f/tune_H.txt 2>&1 && \
timeout -k 10 300 python3 tests/gpu_tune.py --config C1 --reps 5 > gpurun_out/r2f/tune_C1.txt 2>&1 && \
timeout -k 10 300 python3 tests/gpu_tune.py --config C2 --reps 5 > gpurun_out/r2f/tune_C2.txt 2>&1
echo "rc=$?"
tail -n 5 gpurun_out/r2f/gpu_tests.log
