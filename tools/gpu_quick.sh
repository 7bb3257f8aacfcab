#!/bin/bash
# quick GPU check: selected test files, then the bench line without the secondary legs
# usage: bash tools/gpu_quick.sh <outdir> "<pytest args>" [bench args...]
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-quick}; shift
T=${1:-tests}; shift
mkdir -p $O
timeout -k 10 900 python3 -m pytest $T -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -n 40 $O/gpu_tests.log; exit 1; }
tail -n 3 $O/gpu_tests.log
timeout -k 10 400 python3 bench.py --gpus 1 --steps 10 --warmup 3 --no-secondary --no-cpu "$@" > $O/bench.json 2> $O/bench.err || { tail $O/bench.err; exit 1; }
python3 - <<PY
import json
o = json.load(open('$O/bench.json'))
print('bench', o['ms_per_step'], o['roofline']['frac'], {k: round(v, 2) for k, v in o['stages_ms'].items()})
print('gram frac', o['secondary']['gram']['frac'])
PY
echo ALLDONE
