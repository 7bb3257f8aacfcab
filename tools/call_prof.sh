#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/prof_fm
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_fm -o fm --output-format csv -- python3 $R/tests/gpu_tune.py --config H --reps 2 > $R/gpurun_out/prof_fm/run.log 2>&1 || { tail -20 $R/gpurun_out/prof_fm/run.log; exit 1; }
f=$(find $R/gpurun_out/prof_fm -name "*kernel_stats.csv" | head -1)
grep -E "featuremap|project_kernel|xtz|reduce_tri|Name" $f | cut -c1-60,200-400
