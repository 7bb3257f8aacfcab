#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -x -q -m gpu -k "predict or kat or facade" > gpurun_out/predict_tests.log 2>&1 || { tail -30 gpurun_out/predict_tests.log; exit 1; }
tail -3 gpurun_out/predict_tests.log
timeout -k 10 200 python tools/predict_time.py > gpurun_out/predict_time.log 2>&1 || { tail -30 gpurun_out/predict_time.log; exit 1; }
tail -2 gpurun_out/predict_time.log
