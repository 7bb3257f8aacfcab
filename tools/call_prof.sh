#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/prof_ks
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_ks -o ks --output-format csv -- python3 $R/tests/gpu_tune.py --config H --rows 65536 --reps 2 > $R/gpurun_out/prof_ks/run.log 2>&1 || { tail -20 $R/gpurun_out/prof_ks/run.log; exit 1; }
echo done
