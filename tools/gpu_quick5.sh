#!/bin/bash
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/quick
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_round2.py -m gpu -x -q -k "random_shapes" --durations=3 > $O/rand.log 2>&1 || { tail -n 40 $O/rand.log; exit 1; }
echo ALLDONE
