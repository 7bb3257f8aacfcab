#!/bin/bash
# scratch driver for a quick GPU check: stage/parity tests then per-stage timings of the given configs
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/quick
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_stages.py tests/test_gpu_parity.py -m gpu -x -q -k "stages or artifact or oracle_kats or ragged or error or predict" > $O/gpu_tests.log 2>&1 || { tail -n 40 $O/gpu_tests.log; exit 1; }
for cfg in "$@"; do
  timeout -k 10 300 python3 tests/gpu_tune.py --config $cfg --reps 3 > $O/tune_$cfg.txt 2>&1 || { tail $O/tune_$cfg.txt; exit 1; }
done
echo ALLDONE
