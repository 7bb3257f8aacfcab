#!/bin/bash
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/quick
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_stages.py tests/test_gpu_parity.py -m gpu -x -q -k "stages or fp32_mode or ragged" > $O/gpu_tests.log 2>&1 || { tail -n 40 $O/gpu_tests.log; exit 1; }
timeout -k 10 300 python3 tests/gpu_tune.py --config H --dtype bf16x3 --reps 3 > $O/tune_bf3.txt 2>&1 || { tail $O/tune_bf3.txt; exit 1; }
echo ALLDONE
