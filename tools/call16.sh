set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2o
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -n 40 $O/gpu_tests.log; exit 1; }
timeout -k 10 300 python3 tests/gpu_tune.py --config H --dtype bf16x3 --reps 3 > $O/tune_bf3.txt 2>&1 || { tail $O/tune_bf3.txt; exit 1; }
timeout -k 10 300 python3 tests/gpu_tune.py --config H --reps 3 > $O/tune_f32.txt 2>&1 || exit 1
echo ALLDONE
