#!/bin/bash
# one GPU-box call: the -m gpu tier, then the default bench line and the fp64 line (outputs under gpurun_out/$1)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-check}
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -n 40 $O/gpu_tests.log; exit 1; }
tail -n 3 $O/gpu_tests.log
timeout -k 10 500 python3 bench.py --gpus 1 --steps 10 --warmup 3 > $O/bench_H.json 2> $O/bench_H.err || { tail $O/bench_H.err; exit 1; }
timeout -k 10 300 python3 bench.py --config H --dtype f64 --steps 5 --warmup 2 --no-secondary --no-cpu > $O/bench_H_f64.json 2> $O/bench_H_f64.err || { tail $O/bench_H_f64.err; exit 1; }
python3 - <<PY
import json
for f in ('bench_H', 'bench_H_f64'):
    o = json.load(open('$O/%s.json' % f))
    print(f, o['ms_per_step'], o['roofline']['frac'], {k: round(v, 2) for k, v in o['stages_ms'].items()})
    if 'secondary' in o and 'f64' in o['secondary']:
        print('  f64 leg', o['secondary']['f64']['ms_per_step'], o['secondary']['f64']['roofline']['frac'])
        print('  parity', {k: v for k, v in o['parity_at_size'].items() if k != 'what'})
PY
echo ALLDONE
