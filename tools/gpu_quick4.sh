#!/bin/bash
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/quick
mkdir -p $O
timeout -k 10 300 python3 tests/gpu_tune.py --config H --rows 262144 --reps 3 > $O/tune_k1.txt 2>&1 || exit 1
SCFGP_LIB_VARIANT=_krep timeout -k 10 300 python3 tests/gpu_tune.py --config H --rows 262144 --reps 3 > $O/tune_k4.txt 2>&1 || { tail $O/tune_k4.txt; exit 1; }
echo ALLDONE
