#!/bin/bash
# the headline artefacts only: bench line (H, fp32), rocprofv3 kernel stats of the same command, PMC passes, Gram trace (gpurun_out/prof/)
set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/prof
mkdir -p $O
timeout -k 10 400 python3 bench.py --config H --steps 20 --warmup 5 > $O/bench_H.json 2> $O/bench_H.err || { tail $O/bench_H.err; exit 1; }
export TMPDIR=/tmp
( cd /tmp && rm -rf $O/stats && timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/stats -o h --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --config H --steps 10 --warmup 3 --no-cpu --no-secondary > $O/stats.log 2>&1 ) || { tail $O/stats.log; exit 1; }
timeout -k 10 300 bash profiles/collect_pmc.sh H_f32 --config H > $O/pmc_h.log 2>&1
python3 profiles/pmc_summarize.py gpurun_out/pmc_H_f32 H_f32 gpurun_out/r05_pmc_traffic.json > gpurun_out/pmc_H_f32.summary.txt; rm -rf gpurun_out/pmc_H_f32/*/*/*.db
SCFGP_LIB_VARIANT=_trace timeout -k 10 200 python3 tests/gpu_gram_trace.py > gpurun_out/gram_trace_final.txt 2>&1
echo ALLDONE
