#!/bin/bash
# artefacts of the secondary mode f16x3 at the headline shape: the bench line (fp32 headline + secondary.f16x3), rocprofv3 kernel stats of
# f16x3 evaluations, PMC passes over the f16x3 kernels (gpurun_out/prof/); --no-bench: after refresh_profiles.sh, which wrote the line
set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/prof
mkdir -p $O
[ "$1" = "--no-bench" ] || { timeout -k 10 400 python3 bench.py --config H --steps 20 --warmup 5 > $O/bench_H.json 2> $O/bench_H.err || { tail $O/bench_H.err; exit 1; }; }
export TMPDIR=/tmp
( cd /tmp && rm -rf $O/stats16 && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats16 -o h16 --output-format csv -- python3 $GRAFT_REPO_ROOT/tests/gpu_tune.py --config H --dtype f16x3 --reps 10 > $O/stats16.log 2>&1 ) || { tail $O/stats16.log; exit 1; }
timeout -k 10 400 bash tools/pmc_f16.sh > $O/pmc_f16.txt 2>&1; rm -rf gpurun_out/pmc_f16/*/*/*.db
echo ALLDONE
