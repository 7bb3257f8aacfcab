#!/bin/bash
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/quick
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/gpu_tests_all.log 2>&1 || { tail -n 40 $O/gpu_tests_all.log; exit 1; }
timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu --config C2 > $O/bench_nccl1.json 2> $O/bench_nccl1.err || { tail -n 20 $O/bench_nccl1.err; exit 1; }
timeout -k 10 300 python3 tests/gpu_tune.py --config C1 --reps 5 > $O/tune_C1.txt 2>&1 || exit 1
timeout -k 10 300 python3 bench.py --config C1 > $O/bench_C1.json 2> $O/bench_C1.err || exit 1
echo ALLDONE
