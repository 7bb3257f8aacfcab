#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for v in "" _prio; do
    SCFGP_LIB_VARIANT=$v timeout -k 10 300 python tests/gpu_tune.py --config H --dtype bf16x3 --reps 3 --opts "bf3_dma=1" > gpurun_out/diag_dma$v.log 2>&1 || { tail -20 gpurun_out/diag_dma$v.log; exit 1; }
    echo "variant '$v': $(grep -o 'apply_v=[0-9.]*' gpurun_out/diag_dma$v.log | tail -1) $(grep -o 'apply_phibar=[0-9.]*' gpurun_out/diag_dma$v.log | tail -1)"
done
