#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for v in "" _fns _fnc; do
  for cfg in H C3; do
    SCFGP_LIB_VARIANT=$v timeout -k 10 300 python tests/gpu_tune.py --config $cfg --reps 3 > gpurun_out/diag_fmap${v}_$cfg.log 2>&1 || { tail -20 gpurun_out/diag_fmap${v}_$cfg.log; exit 1; }
    echo "variant '$v' $cfg: $(grep -o 'featuremap=[0-9.]*' gpurun_out/diag_fmap${v}_$cfg.log | tail -1)"
  done
done
