#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "bf16x3" > gpurun_out/bf3_tests.log 2>&1 || { tail -30 gpurun_out/bf3_tests.log; exit 1; }
tail -2 gpurun_out/bf3_tests.log
