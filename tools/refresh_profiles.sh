#!/bin/bash
# regenerates the bench lines (H, H fp64, C1..C5) and the rocprofv3 kernel stats under gpurun_out/prof/ (copy to profiles/ by hand)
set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/prof
mkdir -p $O
timeout -k 10 400 python3 bench.py --config H --steps 20 --warmup 5 > $O/bench_H.json 2> $O/bench_H.err || { tail $O/bench_H.err; exit 1; }
timeout -k 10 400 python3 bench.py --config H --dtype f64 --steps 10 --warmup 3 --no-secondary > $O/bench_H_f64.json 2> $O/bench_H_f64.err || { tail $O/bench_H_f64.err; exit 1; }
timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --rows 250000 --steps 5 --warmup 2 --no-cpu --no-secondary > $O/bench_gloo2_250k.json 2> $O/bench_gloo2_250k.err || { tail $O/bench_gloo2_250k.err; exit 1; }
python3 tools/box_probe.py 3 > $O/box_probe.txt 2>&1
for cfg in C1 C2 C3; do
    timeout -k 10 300 python3 bench.py --config $cfg --steps 10 --warmup 3 > $O/bench_${cfg}.json 2> $O/bench_${cfg}.err || { tail $O/bench_${cfg}.err; exit 1; }
done
for cfg in C4 C5; do
    timeout -k 10 400 python3 bench.py --config $cfg --steps 5 --warmup 2 > $O/bench_${cfg}.json 2> $O/bench_${cfg}.err || { tail $O/bench_${cfg}.err; exit 1; }
done
export TMPDIR=/tmp
cd /tmp
rm -rf $O/stats
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/stats -o h --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --config H --steps 10 --warmup 3 --no-cpu --no-secondary > $O/stats.log 2>&1 || { tail $O/stats.log; exit 1; }
echo ALLDONE
