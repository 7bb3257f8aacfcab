#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_stages.py tests/test_gpu_parity.py -x -q -m gpu -k "not full_size" > gpurun_out/ks_tests.log 2>&1 || { tail -30 gpurun_out/ks_tests.log; exit 1; }
tail -2 gpurun_out/ks_tests.log
for v in ""; do
for cfg in H C5 C1; do
  SCFGP_LIB_VARIANT=$v timeout -k 10 300 python tests/gpu_tune.py --config $cfg --rows 65536 --reps 5 > gpurun_out/ks_$cfg$v.log 2>&1 || { tail -20 gpurun_out/ks_$cfg$v.log; exit 1; }
  echo "$cfg '$v': $(grep -o 'kstage_factor=[0-9.]*' gpurun_out/ks_$cfg$v.log | tail -1)"
done
done
