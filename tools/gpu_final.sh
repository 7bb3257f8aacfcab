#!/bin/bash
# final validation on the GPU box: the two driver tiers (pytest -m gpu, smoke) and the default bench line
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/final
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -n 40 $O/gpu_tests.log; exit 1; }
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail $O/smoke.log; exit 1; }
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_H.json 2> $O/bench_H.err || { tail $O/bench_H.err; exit 1; }
echo ALLDONE
