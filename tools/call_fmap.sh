#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_stages.py tests/test_gpu_parity.py -x -q -m gpu -k "not headline and not full_size" > gpurun_out/fmap_tests.log 2>&1 || { tail -30 gpurun_out/fmap_tests.log; exit 1; }
tail -2 gpurun_out/fmap_tests.log
for cfg in H C3 C2 C5; do
    timeout -k 10 300 python tests/gpu_tune.py --config $cfg --reps 3 > gpurun_out/fmap_$cfg.log 2>&1 || { tail -20 gpurun_out/fmap_$cfg.log; exit 1; }
    echo "$cfg: $(grep -o 'featuremap=[0-9.]*' gpurun_out/fmap_$cfg.log | tail -1) $(grep -o 'total [0-9.]* ms' gpurun_out/fmap_$cfg.log | tail -1)"
done
