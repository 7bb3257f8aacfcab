#!/bin/bash
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/quick
mkdir -p $O
timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu > $O/bench_nccl1.json 2> $O/bench_nccl1.err || { tail -n 20 $O/bench_nccl1.err; exit 1; }
timeout -k 10 400 python3 bench.py --gpus 4 --backend gloo --config C2 --steps 3 --warmup 1 --no-cpu > $O/bench_gloo4.json 2> $O/bench_gloo4.err || { tail -n 20 $O/bench_gloo4.err; exit 1; }
timeout -k 10 400 python3 bench.py --config C2 --steps 3 --warmup 1 --no-cpu > $O/bench_c2_1.json 2> $O/bench_c2_1.err || exit 1
echo ALLDONE
